// bw_probe4.hip -- the bare 16-in/11-out streaming loop at restricted occupancy (dynamic-LDS padding limits
// blocks per CU) and with different amounts of work per lane: is it occupancy or loop structure that costs
// the real step kernel its last 20 % of bandwidth?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(const double *__restrict__ in, double *__restrict__ out, size_t n, size_t stride)
{
    extern __shared__ char pad[];
    if (threadIdx.x == 9999) pad[0] = 1;
    const size_t step = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step) {
        double v[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) v[f] = in[(size_t)f * stride + i];
#pragma unroll
        for (int f = 0; f < 11; ++f) out[(size_t)f * stride + i] = v[f] + v[(f + 5) % 16];
    }
}
void run(int lds, int grid, double *in, double *out, size_t n, int sets, bool inplace)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const size_t set_elems = 16 * n;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) hipEventRecord(a);
        for (int s = 0; s < sets; ++s)
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, in + s * set_elems, (inplace ? in : out) + s * set_elems, n, n);
    }
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("blocks/CU <= %d  grid %5d  %s  %.4f ms/launch  %.0f GB/s\n", lds ? 160 * 1024 / lds : 8, grid, inplace ? "in place " : "separate ", ms / sets,
           (double)n * 216 * sets / ms / 1e6);
}
int main()
{
    const size_t n = 1 << 20; const int sets = 6;
    double *in, *out;
    hipMalloc(&in, sets * 16 * n * 8); hipMalloc(&out, sets * 16 * n * 8);
    hipMemset(in, 0, sets * 16 * n * 8); hipMemset(out, 0, sets * 16 * n * 8);
    for (int lds : {0, 40 * 1024, 80 * 1024})          // 8, 4, 2 blocks per CU
        for (int grid : {512, 1024, 2048, 4096})
            for (bool ip : {false, true}) run(lds, grid, in, out, n, sets, ip);
    return 0;
}
