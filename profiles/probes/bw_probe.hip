// bw_probe.hip -- what HBM rate does the step's access pattern allow?  (tuning aid, not product)
//   A: 16 read streams + 11 write streams of 8 B per lane (the SoA-of-doubles layout), one problem per lane
//   B: 8 read streams + 6 write streams of 16 B per lane (fields paired into double2)
//   C: plain copy, 16 B per lane, 1 stream in / 1 out (the chip's practical ceiling for a read+write mix)
// build: hipcc -O3 --offload-arch=gfx950 -o bw_probe bw_probe.hip ; run: ./bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NIN, int NOUT, typename V>
__global__ void __launch_bounds__(256) k_streams(const V *__restrict__ in, V *__restrict__ out, size_t n, size_t stride)
{
    const size_t step = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step) {
        V v[NIN];
#pragma unroll
        for (int f = 0; f < NIN; ++f) v[f] = in[(size_t)f * stride + i];
#pragma unroll
        for (int f = 0; f < NOUT; ++f) out[(size_t)f * stride + i] = v[f] + v[(f + 1) % NIN];
    }
}

template <int NIN, int NOUT, typename V>
double run(const char *tag, size_t n, int grid, V *in, V *out, size_t stride, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_streams<NIN, NOUT, V>), dim3(grid), dim3(256), 0, 0, in, out, n, stride);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (r > 0 && ms < best) best = ms;
    }
    const double bytes = (double)n * sizeof(V) * (NIN + NOUT);
    printf("%-44s grid %5d  %.4f ms  %.0f GB/s\n", tag, grid, best, bytes / best / 1e6);
    return best;
}

int main()
{
    const size_t n = 1 << 20;
    // rotate over several buffer sets so nothing is served from the 256 MiB Infinity Cache
    const int sets = 6;
    double *in, *out;
    const size_t stride = n;
    CK(hipMalloc(&in, sets * 16 * n * sizeof(double)));
    CK(hipMalloc(&out, sets * 16 * n * sizeof(double)));
    CK(hipMemset(in, 0, sets * 16 * n * sizeof(double)));
    CK(hipMemset(out, 0, sets * 16 * n * sizeof(double)));
    for (int grid : {512, 1024, 2048, 4096}) {
        for (int s = 0; s < 2; ++s) {
            double *i0 = in + (size_t)(s % sets) * 16 * n, *o0 = out + (size_t)(s % sets) * 16 * n;
            if (s == 0) continue;
            run<16, 11, double>("A: 16 in + 11 out streams, 8 B/lane", n, grid, i0, o0, stride, 4);
            run<8, 6, double2>("B: 8 in + 6 out streams, 16 B/lane", n, grid, (double2 *)i0, (double2 *)o0, stride, 4);
        }
    }
    // single-stream copy of the same volume (128 B in, 96 B ~ use 112 B each way): 7 Mi double2 elements
    run<1, 1, double2>("C: 1 in + 1 out stream, 16 B/lane, 7 Mi elems", (size_t)7 << 20, 2048, (double2 *)in, (double2 *)out, 0, 4);
    run<1, 1, double>("C': 1 in + 1 out stream, 8 B/lane, 14 Mi elems", (size_t)14 << 20, 2048, in, out, 0, 4);
    run<1, 1, double2>("C: same, 32 Mi elems (512 MiB each way)", (size_t)32 << 20, 4096, (double2 *)in, (double2 *)out, 0, 3);
    return 0;
}
