// rcp_probe.hip -- accuracy of v_rcp_f64, of 1 / 2 Newton refinements and of one cubic (Halley) refinement against IEEE division (tuning aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, double *o, int n)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r0 = __builtin_amdgcn_rcp(v);
    double r1 = __builtin_fma(__builtin_fma(-v, r0, 1.0), r0, r0);
    double r2 = __builtin_fma(__builtin_fma(-v, r1, 1.0), r1, r1);
    // one cubic step: e = 1 - v r0, r = r0 (1 + e + e^2): three multiply-adds, error e^3
    double e = __builtin_fma(-v, r0, 1.0);
    double r3 = __builtin_fma(r0, __builtin_fma(e, e, e), r0);
    o[5 * i + 0] = 1.0 / v; o[5 * i + 1] = r0; o[5 * i + 2] = r1; o[5 * i + 3] = r2; o[5 * i + 4] = r3;
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> h(n);
    unsigned long long s = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); h[i] = std::ldexp(1.0 + u, (int)(s % 80) - 40) * ((s & 1) ? 1 : -1); }
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, (size_t)n * 40);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    std::vector<double> o(5 * (size_t)n);
    hipMemcpy(o.data(), dout, (size_t)n * 40, hipMemcpyDeviceToHost);
    double e[5] = {0, 0, 0, 0, 0};
    long long differ[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) for (int j = 1; j < 5; ++j) {
        double r = std::fabs(o[5 * (size_t)i + j] / o[5 * (size_t)i] - 1.0); if (r > e[j]) e[j] = r;
        differ[j] += o[5 * (size_t)i + j] != o[5 * (size_t)i];
    }
    printf("max rel err vs IEEE 1/x: raw v_rcp_f64 %.3e  1 Newton %.3e (%.2f ulp)  2 Newton %.3e (%.2f ulp, %lld of %d differ)  1 cubic %.3e (%.2f ulp, %lld differ)\n",
           e[1], e[2], e[2] / 2.22e-16, e[3], e[3] / 2.22e-16, differ[3], n, e[4], e[4] / 2.22e-16, differ[4]);
    return 0;
}
