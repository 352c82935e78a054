// bw_probe3.hip -- why does the real step kernel stream at 4.3 TB/s when a bare 16-in/11-out kernel reaches 5.4?
// Same access pattern, varying: in-place update, occupancy (LDS padding limits blocks per CU), problems per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_BYTES, bool INPLACE, bool PREFETCH>
__global__ void __launch_bounds__(256) k(const double *__restrict__ in, double *__restrict__ out, size_t n, size_t stride)
{
    extern __shared__ char pad[];
    if (LDS_BYTES && threadIdx.x == 9999) pad[0] = 1;
    const size_t step = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double cur[16], nxt[16];
#pragma unroll
    for (int f = 0; f < 16; ++f) cur[f] = in[(size_t)f * stride + i];
    for (;;) {
        const size_t inext = i + step;
        const bool have = inext < n;
        const size_t src = have ? inext : i;
        if (PREFETCH) {
#pragma unroll
            for (int f = 0; f < 16; ++f) nxt[f] = in[(size_t)f * stride + src];
        }
        double *o = INPLACE ? const_cast<double *>(in) : out;
#pragma unroll
        for (int f = 0; f < 11; ++f) o[(size_t)f * stride + i] = cur[f] + cur[(f + 5) % 16];
        if (!have) break;
        if (!PREFETCH) {
#pragma unroll
            for (int f = 0; f < 16; ++f) nxt[f] = in[(size_t)f * stride + src];
        }
#pragma unroll
        for (int f = 0; f < 16; ++f) cur[f] = nxt[f];
        i = inext;
    }
}
template <int LDS_BYTES, bool INPLACE, bool PREFETCH>
void run(const char *tag, int grid, double *in, double *out, size_t n, int sets)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const size_t set_elems = 16 * n;
    hipFuncSetAttribute((const void *)k<LDS_BYTES, INPLACE, PREFETCH>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) hipEventRecord(a);
        for (int s = 0; s < sets; ++s)
            hipLaunchKernelGGL((k<LDS_BYTES, INPLACE, PREFETCH>), dim3(grid), dim3(256), LDS_BYTES, 0, in + s * set_elems, out + s * set_elems, n, n);
    }
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-64s grid %5d  %.4f ms/launch  %.0f GB/s\n", tag, grid, ms / sets, (double)n * 216 * sets / ms / 1e6);
}
int main()
{
    const size_t n = 1 << 20; const int sets = 6;
    double *in, *out;
    hipMalloc(&in, sets * 16 * n * 8); hipMalloc(&out, sets * 16 * n * 8);
    hipMemset(in, 0, sets * 16 * n * 8); hipMemset(out, 0, sets * 16 * n * 8);
    for (int grid : {512, 4096}) {
        run<0, false, true>("separate out, full occupancy, prefetch", grid, in, out, n, sets);
        run<0, true, true>("in place, full occupancy, prefetch", grid, in, out, n, sets);
        run<65536, true, true>("in place, 2 blocks/CU (64 KiB LDS pad), prefetch", grid, in, out, n, sets);
        run<65536, true, false>("in place, 2 blocks/CU, no prefetch", grid, in, out, n, sets);
        run<65536, false, true>("separate out, 2 blocks/CU, prefetch", grid, in, out, n, sets);
    }
    return 0;
}
