// ilp_probe.hip -- how many independent fp64 chains does ONE wave per SIMD need to keep the vector ALU busy?  (tuning aid for the
// post-convergence loop of small fixed-step batches, which run one wave per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 8192
template <int CH>
__global__ void __launch_bounds__(256) k(double *out, double a, double b)
{
    double x[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) x[j] = a + threadIdx.x * 1e-9 + j;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < CH; ++j) x[j] = __builtin_fma(x[j], a, b);
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < CH; ++j) s += x[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH> void run(int blocks_per_cu, double *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(256), 0, 0, d, 1.0000001, 1e-7);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double winst = (double)blocks_per_cu * ITER * CH;      // wave-instructions per SIMD
    printf("%d independent chain(s), %d wave(s)/SIMD: %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", CH, blocks_per_cu,
           best * 1e6 / winst, best * 1e6 / winst * 2.4);
}
int main()
{
    double *d; hipMalloc(&d, 256 * 8 * 256 * 8);
    for (int w : {1, 2, 3}) { run<1>(w, d); run<2>(w, d); run<3>(w, d); run<4>(w, d); run<8>(w, d); }
    return 0;
}
