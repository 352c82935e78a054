// valu_probe.hip -- issue cost of the fp64 instructions the step is made of (tuning aid)
// Each kernel runs ITER x 8 independent chains per lane; waves/SIMD is set by the grid (8 waves per CU = 2 per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
template <int OP>
__global__ void __launch_bounds__(256) k(double *out, double a, double b)
{
    double x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = a + threadIdx.x * 1e-9 + j;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) x[j] = __builtin_fma(x[j], a, b);
            if (OP == 1) x[j] = x[j] * a;
            if (OP == 2) x[j] = x[j] + b;
            if (OP == 3) x[j] = __builtin_amdgcn_rcp(x[j]);
            if (OP == 4) x[j] = (x[j] > b) ? a : x[j];          // v_cmp_f64 + 2 v_cndmask_b32
            if (OP == 5) x[j] = __builtin_fmin(x[j], b);
            if (OP == 6) { float f = (float)x[j]; f = __builtin_fmaf(f, 1.0001f, 0.5f); x[j] = f; }   // cvt + fma32 + cvt
        }
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char *name, int blocks_per_cu, double *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 1.0000001, 1e-7);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    // wave-instructions per SIMD = blocks_per_cu (waves per SIMD) * ITER * 8
    const double winst = (double)blocks_per_cu * ITER * 8;
    printf("%-34s %d wave(s)/SIMD  %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, blocks_per_cu,
           best, best * 1e6 / winst, best * 1e6 / winst * 2.4);
}
int main()
{
    double *d; hipMalloc(&d, 256 * 8 * 256 * 8);
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64", w, d);
        run<1>("v_mul_f64", w, d);
        run<2>("v_add_f64", w, d);
        run<3>("v_rcp_f64", w, d);
        run<4>("v_cmp_f64 + 2 v_cndmask_b32", w, d);
        run<5>("v_min_f64", w, d);
        run<6>("cvt_f32_f64 + fma_f32 + cvt_f64_f32", w, d);
    }
    return 0;
}
