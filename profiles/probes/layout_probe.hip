// layout_probe.hip -- would an AoSoA state layout lift the streaming step's bandwidth?  
//
// The k = 1 launch (k_newton_stream16, F3 fp64, zero end velocities) loads 14 fields and stores 11 per problem, 16 B per lane
// (two consecutive problems), nontemporal, in place.  With the SoA layout base[f * stride + i] a wave touches 25 regions
// 8 MiB apart.  AoSoA: [chunk of W problems][16 fields][W]: the same 25 accesses of a wave fall into ONE contiguous
// 16 * W * 8 B region.  This probe runs exactly that access pattern with no arithmetic (the kernel's k = 0 form) for both
// layouts and a flat copy of the same byte count, every launch on a cold 128 MiB set.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2 __attribute__((ext_vector_type(2)));
constexpr int NF = 16, NLOAD = 14, NSTORE = 11;
static __device__ __forceinline__ bool skipped(int f) { return f == 12 || f == 15; }      // vel0X, vel2X are not read

// SoA, field stride `stride` elements
__global__ void __launch_bounds__(256) k_soa(double *__restrict__ base, size_t stride)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    v2 f[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q)
        if (!skipped(q)) f[q] = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(base + (size_t)q * stride + i));
#pragma unroll
    for (int q = 0; q < NSTORE; ++q) __builtin_nontemporal_store(f[q] + f[(q + 3) % 11 == 1 ? 13 : 14], reinterpret_cast<v2 *>(base + (size_t)q * stride + i));
}

// AoSoA: chunk c = i / W holds fields 0..15 of W consecutive problems, field q at chunk * 16 W + q W
template <int W>
__global__ void __launch_bounds__(256) k_aosoa(double *__restrict__ base)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    double *p = base + (i / W) * (size_t)(NF * W) + (i % W);
    v2 f[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q)
        if (!skipped(q)) f[q] = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(p + q * W));
#pragma unroll
    for (int q = 0; q < NSTORE; ++q) __builtin_nontemporal_store(f[q] + f[(q + 3) % 11 == 1 ? 13 : 14], reinterpret_cast<v2 *>(p + q * W));
}

// flat: 112 B read + 88 B written per problem from / to contiguous buffers (what a copy of the same byte count does)
__global__ void __launch_bounds__(256) k_flat(const double *__restrict__ in, double *__restrict__ out, size_t n)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
    v2 acc = {0.0, 0.0};
    for (size_t j = t; j < n * NLOAD / 2; j += nt) {
        const v2 x = __builtin_nontemporal_load(reinterpret_cast<const v2 *>(in) + j);
        if (j < n * NSTORE / 2) __builtin_nontemporal_store(x, reinterpret_cast<v2 *>(out) + j);
        else acc += x;
    }
    if (acc[0] == 12345.678) out[0] = acc[1];
}

int main()
{
    const size_t n = 1 << 20;
    const int sets = 8;
    size_t stride = (n + 511) / 512 * 512;
    if ((stride / 512) % 2 == 0) stride += 512;
    const size_t set_elems = NF * stride;
    double *buf, *buf2;
    hipMalloc(&buf, sets * set_elems * 8);
    hipMalloc(&buf2, sets * set_elems * 8);
    hipMemset(buf, 0, sets * set_elems * 8);
    hipMemset(buf2, 0, sets * set_elems * 8);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const unsigned grid = (unsigned)(n / 512);
    auto timed = [&](const char *name, auto launch) {
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(a);
            for (int s = 0; s < sets; ++s) launch(s);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep) { sum += ms; if (ms < best) best = ms; }
        }
        printf("%-34s best %.4f ms  mean %.4f ms per launch   %.0f / %.0f GB/s on 200 B per problem\n", name, best / sets, sum / 4 / sets,
               (double)n * 200 * sets / best / 1e6, (double)n * 200 * sets / (sum / 4) / 1e6);
    };
    for (int round = 0; round < 2; ++round) {
        timed("SoA, odd-multiple-of-512 stride", [&](int s) { hipLaunchKernelGGL(k_soa, dim3(grid), dim3(256), 0, 0, buf + s * set_elems, stride); });
        timed("SoA, stride = n (aliased)", [&](int s) { hipLaunchKernelGGL(k_soa, dim3(grid), dim3(256), 0, 0, buf + s * set_elems, n); });
        timed("AoSoA, W = 128 (16 KiB chunks)", [&](int s) { hipLaunchKernelGGL(k_aosoa<128>, dim3(grid), dim3(256), 0, 0, buf + s * set_elems); });
        timed("AoSoA, W = 512 (64 KiB chunks)", [&](int s) { hipLaunchKernelGGL(k_aosoa<512>, dim3(grid), dim3(256), 0, 0, buf + s * set_elems); });
        timed("AoSoA, W = 2048 (256 KiB chunks)", [&](int s) { hipLaunchKernelGGL(k_aosoa<2048>, dim3(grid), dim3(256), 0, 0, buf + s * set_elems); });
        timed("flat copy of the same bytes", [&](int s) { hipLaunchKernelGGL(k_flat, dim3(2048), dim3(256), 0, 0, buf + s * set_elems, buf2 + s * set_elems, n); });
    }
    return 0;
}
