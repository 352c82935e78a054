// bw_probe2.hip -- HBM read+write rate of the step's access pattern at HBM scale (buffers rotate through
// 6 x 226 MB so nothing is served from the 256 MiB Infinity Cache), with plain vs non-temporal accesses,
// 8 B vs 16 B per lane, and a few grid sizes.  Tuning aid, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int NIN, int NOUT, typename V, bool NT>
__global__ void __launch_bounds__(256) k_streams(const V *__restrict__ in, V *__restrict__ out, size_t n, size_t stride)
{
    const size_t step = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step) {
        V v[NIN];
#pragma unroll
        for (int f = 0; f < NIN; ++f) v[f] = NT ? __builtin_nontemporal_load(&in[(size_t)f * stride + i]) : in[(size_t)f * stride + i];
#pragma unroll
        for (int f = 0; f < NOUT; ++f) {
            V r = v[f] + v[(f + 1) % NIN];
            if (NT) __builtin_nontemporal_store(r, &out[(size_t)f * stride + i]); else out[(size_t)f * stride + i] = r;
        }
    }
}
template <int NIN, int NOUT, typename V, bool NT>
void run(const char *tag, size_t n, int grid, V *in, V *out, size_t set_elems, int sets)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    // warm-up over all sets, then time one pass over all sets (each set touched once per pass)
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) hipEventRecord(a);
        for (int s = 0; s < sets; ++s)
            hipLaunchKernelGGL((k_streams<NIN, NOUT, V, NT>), dim3(grid), dim3(256), 0, 0, in + s * set_elems, out + s * set_elems, n, n);
    }
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)n * sizeof(V) * (NIN + NOUT) * sets;
    printf("%-52s grid %5d  %.4f ms/launch  %.0f GB/s\n", tag, grid, ms / sets, bytes / ms / 1e6);
}
int main()
{
    const size_t n = 1 << 20;
    const int sets = 6;
    const size_t set_elems = 16 * n;                 // doubles per set (128 MiB in, 128 MiB out region)
    double *in, *out;
    hipMalloc(&in, sets * set_elems * 8); hipMalloc(&out, sets * set_elems * 8);
    hipMemset(in, 0, sets * set_elems * 8); hipMemset(out, 0, sets * set_elems * 8);
    for (int grid : {512, 1024, 4096}) {
        run<16, 11, double, false>("16 in + 11 out, 8 B/lane, plain", n, grid, in, out, set_elems, sets);
        run<16, 11, double, true>("16 in + 11 out, 8 B/lane, non-temporal", n, grid, in, out, set_elems, sets);
        run<8, 6, d2, false>("8 in + 6 out, 16 B/lane, plain", n, grid, (d2 *)in, (d2 *)out, set_elems / 2, sets);
        run<8, 6, d2, true>("8 in + 6 out, 16 B/lane, non-temporal", n, grid, (d2 *)in, (d2 *)out, set_elems / 2, sets);
    }
    run<1, 1, d2, false>("copy 1 in + 1 out, 16 B/lane, 8 Mi elems x 6, plain", (size_t)8 << 20, 4096, (d2 *)in, (d2 *)out, set_elems / 2, sets);
    run<1, 1, d2, true>("copy 1 in + 1 out, 16 B/lane, 8 Mi elems x 6, nt", (size_t)8 << 20, 4096, (d2 *)in, (d2 *)out, set_elems / 2, sets);
    run<1, 0, d2, false>("read only, 16 B/lane, 8 Mi elems x 6", (size_t)8 << 20, 4096, (d2 *)in, (d2 *)out, set_elems / 2, sets);
    return 0;
}
