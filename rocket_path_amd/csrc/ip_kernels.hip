// ip_kernels.hip -- HIP kernels of the batched interior-point path, gfx950 only.
//
// Data layout in HBM: structure of arrays.  Field f of problem i lives at
// base[f * stride + i]; the field order is the reference's enum order (enum V,
// onedpath_ip.cpp:15-43; enum V2, onedpath2_ip.cpp:15-39), so field 0..2 are the variables,
// 3..3+m-1 the multipliers and the last five the constants.  A wave touches 64 consecutive
// elements of each field: every load/store instruction is one fully used 512 B (f64) or
// 256 B (f32) segment.  One Newton step reads 16 and writes 11 fields (F3): 216 B per
// problem per step, the algorithmic traffic the roofline is priced against.
//
// Launch shape: 256-thread blocks (4 waves), one problem per lane, grid = ceil(n / 256).
// At n = 1 Mi that is 4096 blocks, 16 per CU, dealt round-robin over the 8 XCDs; problems
// are independent and nothing is re-read, so there is no L2 locality to arrange and the
// plain blockIdx -> problem-range map is already XCD-neutral.
#include "ip_kernels.h"

#include "../../include/rp_batch.h"
#include "feas_core.h"
#include "ip_core.h"

namespace rp {

namespace {

constexpr int kBlock = 256;

template <typename T>
__device__ __forceinline__ int wave_sum(int x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

template <typename T> KParams<T> make_kparams(const HostParams &hp, int variant)
{
    KParams<T> kp;
    kp.limit = (T)hp.accel_limit;
    kp.inv_mu_den = (T)(1.0 / (num_constraints(variant) * hp.mu_divisor));
    kp.boundary = (T)hp.boundary_fraction;
    kp.backtrack = (T)hp.backtrack;
    kp.armijo = (T)hp.armijo;
    kp.c_floor = (T)hp.accel_limit * (sizeof(T) == 8 ? (T)8.673617379884035e-19 : (T)4.656612873077393e-10);   // L * eps / 256
    kp.max_bt = hp.max_backtracks;
    return kp;
}

// ---------------------------------------------------------------------------------------
// The hot kernel: up to k Newton steps per problem, state in registers between steps.
//   GATED = false : exactly k steps (k presses of 'n', onedpath_ip.cpp:269-272)
//   GATED = true  : before each step stop if gap < tol or the problem's step count reached
//                   max_iter; problems already finished are skipped
//                   without touching their state.
#ifndef RP_NEWTON_WAVES
#define RP_NEWTON_WAVES 2     // minimum waves per SIMD the register allocator must leave room for
#endif
template <typename T, int VARIANT, bool GATED>
__global__ void __launch_bounds__(kBlock, RP_NEWTON_WAVES)
k_newton(T *__restrict__ base, size_t stride, size_t n, int k, KParams<T> kp, T tol, int max_iter,
         int32_t *__restrict__ iters, uint32_t *__restrict__ status, unsigned long long *__restrict__ counters)
{
    constexpr int NC = CMap<VARIANT>::NC;
    constexpr int CB = 3 + NC;   // first constant field: pos0, vel0, pos1, pos2, vel2
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = i < n;

    int it = 0;
    uint32_t st = 0;
    bool active = valid;
    if (GATED && valid) {
        it = iters[i];
        st = status[i];
        active = (st & (RP_ST_CONVERGED | RP_ST_MAXITER)) == 0;
    }
    int steps_here = 0;
    bool still_open = false;

    if (active) {
        T *f = base + i;
        T v = f[0 * stride], t0 = f[1 * stride], t1 = f[2 * stride];
        T lam[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) lam[c] = f[(3 + c) * stride];
        Prob<T> pr;
        {
            const T p0 = f[(CB + 0) * stride], p1 = f[(CB + 2) * stride], p2 = f[(CB + 3) * stride];
            pr.v0 = f[(CB + 1) * stride];
            pr.v2 = f[(CB + 4) * stride];
            pr.dx0 = p1 - p0;
            pr.dx1 = p2 - p1;
        }

        Acc<T> e;
        accel_values(pr, v, t0, t1, e);
        accel_grads(pr, v, e);

        bool done = false;
        for (int s = 0; s < k; ++s) {
            const T gap = duality_gap<T, VARIANT>(e, lam, kp.limit);
            if (GATED) {
                if (gap < tol) { st |= RP_ST_CONVERGED; done = true; break; }
                if (it >= max_iter) { st |= RP_ST_MAXITER; done = true; break; }
            }
            newton_step<T, VARIANT>(pr, kp, gap, v, t0, t1, lam, e);
            ++it;
            ++steps_here;
        }

        if (GATED) {
            if (!done) {   // settle the status now so the host knows whether to launch again
                const T gap = duality_gap<T, VARIANT>(e, lam, kp.limit);
                if (gap < tol) { st |= RP_ST_CONVERGED; done = true; }
                else if (it >= max_iter) { st |= RP_ST_MAXITER; done = true; }
            }
            st &= ~(RP_ST_NONFINITE | RP_ST_INFEASIBLE);
            if (!(finite_(v) && finite_(t0) && finite_(t1))) st |= RP_ST_NONFINITE;
            if (!all_satisfied<T, VARIANT>(e, kp.limit)) st |= RP_ST_INFEASIBLE;
            iters[i] = it;
            status[i] = st;
            still_open = !done;
        }

        if (steps_here > 0) {
            f[0 * stride] = v;
            f[1 * stride] = t0;
            f[2 * stride] = t1;
#pragma unroll
            for (int c = 0; c < NC; ++c) f[(3 + c) * stride] = lam[c];
        }
    }

    if (GATED) {
        const unsigned long long open_mask = __ballot(still_open);
        const int steps_wave = wave_sum<int>(steps_here);
        if ((threadIdx.x & 63) == 0) {
            if (open_mask) atomicAdd(&counters[0], (unsigned long long)__popcll(open_mask));
            if (steps_wave) atomicAdd(&counters[1], (unsigned long long)steps_wave);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Batch reduction: per problem the surrogate gap, ||r||^2 at p = gap / (10 m), converged bit.
// Block partials go to d_partials[4 * blockIdx]; k_reduce_final folds them (max, max, sum, sum).
__device__ __forceinline__ double nan_max(double a, double b) { return (a > b || a != a) ? a : b; }

template <typename T, int VARIANT>
__global__ void __launch_bounds__(kBlock)
k_reduce_partial(const T *__restrict__ base, size_t stride, size_t n, KParams<T> kp,
                 const uint32_t *__restrict__ status, double *__restrict__ partials)
{
    constexpr int NC = CMap<VARIANT>::NC;
    constexpr int CB = 3 + NC;
    double mr = 0.0, mg = -1.7976931348623157e308, nc = 0.0;
    bool any = false;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const T *f = base + i;
        T lam[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) lam[c] = f[(3 + c) * stride];
        Prob<T> pr;
        const T p0 = f[(CB + 0) * stride], p1 = f[(CB + 2) * stride], p2 = f[(CB + 3) * stride];
        pr.v0 = f[(CB + 1) * stride];
        pr.v2 = f[(CB + 4) * stride];
        pr.dx0 = p1 - p0;
        pr.dx1 = p2 - p1;
        const T v = f[0];
        Acc<T> e;
        accel_values(pr, v, f[1 * stride], f[2 * stride], e);
        accel_grads(pr, v, e);
        const T gap = duality_gap<T, VARIANT>(e, lam, kp.limit);
        const T rn = residual_norm<T, VARIANT, false>(e, lam, lam, T(0), gap * kp.inv_mu_den, kp.limit);
        mr = any ? nan_max((double)rn, mr) : (double)rn;
        mg = any ? nan_max((double)gap, mg) : (double)gap;
        any = true;
        nc += (status[i] & RP_ST_CONVERGED) ? 1.0 : 0.0;
    }
    __shared__ double sh[3][kBlock / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mr = nan_max(__shfl_xor(mr, o), mr);
        mg = nan_max(__shfl_xor(mg, o), mg);
        nc += __shfl_xor(nc, o);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = mr; sh[1][w] = mg; sh[2][w] = nc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int j = 1; j < kBlock / 64; ++j) { mr = nan_max(sh[0][j], mr); mg = nan_max(sh[1][j], mg); nc += sh[2][j]; }
        partials[4 * blockIdx.x + 0] = mr;
        partials[4 * blockIdx.x + 1] = mg;
        partials[4 * blockIdx.x + 2] = nc;
        partials[4 * blockIdx.x + 3] = 0.0;
    }
}

__global__ void __launch_bounds__(64)
k_reduce_final(const double *__restrict__ partials, int nblocks, const unsigned long long *__restrict__ counters,
               double host_steps, double *__restrict__ out4)
{
    double mr = 0.0, mg = -1.7976931348623157e308, nc = 0.0;
    for (int j = threadIdx.x; j < nblocks; j += 64) {
        mr = nan_max(partials[4 * j + 0], mr);
        mg = nan_max(partials[4 * j + 1], mg);
        nc += partials[4 * j + 2];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mr = nan_max(__shfl_xor(mr, o), mr);
        mg = nan_max(__shfl_xor(mg, o), mg);
        nc += __shfl_xor(nc, o);
    }
    if (threadIdx.x == 0) {
        out4[0] = mr;
        out4[1] = mg;
        out4[2] = nc;
        out4[3] = (double)counters[1] + host_steps;
    }
}

// ---------------------------------------------------------------------------------------
// AoS (reference layout, double var[M] per problem) <-> SoA (compute type), staged through
// LDS so that both the global reads and the global writes are fully coalesced.  Rows are
// padded by one double: a lane reading its problem's field f then hits bank (34 l + 2 f) % 64,
// a 2-way conflict instead of the 32-way one of an unpadded 16-double row.
template <typename T, int M>
__global__ void __launch_bounds__(kBlock)
k_aos_to_soa(const double *__restrict__ aos, T *__restrict__ base, size_t stride, size_t n)
{
    __shared__ double tile[kBlock * (M + 1)];
    const size_t first = (size_t)blockIdx.x * kBlock;
    const size_t count = (n - first < (size_t)kBlock) ? (n - first) : (size_t)kBlock;
    const double *src = aos + first * M;
    for (size_t j = threadIdx.x; j < count * M; j += kBlock) tile[(j / M) * (M + 1) + (j % M)] = src[j];
    __syncthreads();
    if (threadIdx.x < count) {
#pragma unroll
        for (int f = 0; f < M; ++f) base[(size_t)f * stride + first + threadIdx.x] = (T)tile[threadIdx.x * (M + 1) + f];
    }
}

template <typename T, int M>
__global__ void __launch_bounds__(kBlock)
k_soa_to_aos(const T *__restrict__ base, size_t stride, size_t n, double *__restrict__ aos)
{
    __shared__ double tile[kBlock * (M + 1)];
    const size_t first = (size_t)blockIdx.x * kBlock;
    const size_t count = (n - first < (size_t)kBlock) ? (n - first) : (size_t)kBlock;
    if (threadIdx.x < count) {
#pragma unroll
        for (int f = 0; f < M; ++f) tile[threadIdx.x * (M + 1) + f] = (double)base[(size_t)f * stride + first + threadIdx.x];
    }
    __syncthreads();
    double *dst = aos + first * M;
    for (size_t j = threadIdx.x; j < count * M; j += kBlock) dst[j] = tile[(j / M) * (M + 1) + (j % M)];
}

// Feasible start (build-defined): vel1 = 0, t_i = (3.5/sqrt 12) sqrt(6 |dX_i| / L),
// multipliers 1, vel0 = vel2 = 0.  Computed in double, stored in the compute type.
template <typename T, int VARIANT>
__global__ void __launch_bounds__(kBlock)
k_init_feasible(T *__restrict__ base, size_t stride, size_t n, double limit, const double *__restrict__ pos0,
                const double *__restrict__ pos1, const double *__restrict__ pos2)
{
    constexpr int NC = CMap<VARIANT>::NC;
    constexpr int CB = 3 + NC;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double p0 = pos0[i], p1 = pos1[i], p2 = pos2[i];
    const double scale = 3.5 / __builtin_sqrt(12.0);
    T *f = base + i;
    f[0 * stride] = T(0);
    f[1 * stride] = (T)(scale * __builtin_sqrt(6.0 * __builtin_fabs(p1 - p0) / limit));
    f[2 * stride] = (T)(scale * __builtin_sqrt(6.0 * __builtin_fabs(p2 - p1) / limit));
#pragma unroll
    for (int c = 0; c < NC; ++c) f[(3 + c) * stride] = T(1);
    f[(CB + 0) * stride] = (T)p0;
    f[(CB + 1) * stride] = T(0);
    f[(CB + 2) * stride] = (T)p1;
    f[(CB + 3) * stride] = (T)p2;
    f[(CB + 4) * stride] = T(0);
}

// Every problem gets the same state (initDefault / initStuck broadcast).
struct ConstState { double v[16]; };

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_init_const(T *__restrict__ base, size_t stride, size_t n, int m, ConstState cs)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    for (int f = 0; f < m; ++f) base[(size_t)f * stride + i] = (T)cs.v[f];
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_nudge(T *__restrict__ field, size_t n, T delta)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) field[i] += delta;
}

__global__ void __launch_bounds__(kBlock)
k_clear_progress(int32_t *__restrict__ iters, uint32_t *__restrict__ status, size_t n, unsigned long long *counters)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) { iters[i] = 0; status[i] = 0; }
    if (i == 0) { counters[0] = 0; counters[1] = 0; }
}

__global__ void k_zero_counter(unsigned long long *c) { c[0] = 0; }

// ---------------------------------------------------------------------------------------
// moveTowardFeasibility (onedpath_ip.cpp:648-721 / onedpath2_ip.cpp:536-609), one lane per problem.
template <typename T, int VARIANT>
__global__ void __launch_bounds__(kBlock)
k_move_toward_feasibility(T *__restrict__ base, size_t stride, size_t n, KParams<T> kp)
{
    constexpr int NC = CMap<VARIANT>::NC;
    constexpr int CB = 3 + NC;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    T *f = base + i;
    Prob<T> pr;
    const T p0 = f[(CB + 0) * stride], p1 = f[(CB + 2) * stride], p2 = f[(CB + 3) * stride];
    pr.v0 = f[(CB + 1) * stride];
    pr.v2 = f[(CB + 4) * stride];
    pr.dx0 = p1 - p0;
    pr.dx1 = p2 - p1;
    T v = f[0], t0 = f[1 * stride], t1 = f[2 * stride];
    Acc<T> e;
    accel_values(pr, v, t0, t1, e);
    accel_grads(pr, v, e);
    T dxv, dx0, dx1;
    if (feasibility_move<T, VARIANT>(e, kp.limit, dxv, dx0, dx1)) {
        f[0] = v + dxv;
        f[1 * stride] = t0 + dx0;
        f[2 * stride] = t1 + dx1;
    }
}

// Plot data: one thread per output value.  Per problem 66 positions (drawSegment,
// onedpath_ip.cpp:1065-1088, 33 per segment) and 4 end accelerations (plotAcceleration, 1024-1027).
template <typename T, int VARIANT>
__global__ void __launch_bounds__(kBlock)
k_sample(const T *__restrict__ base, size_t stride, size_t n, double *__restrict__ pos66, double *__restrict__ acc4)
{
    constexpr int CB = 3 + CMap<VARIANT>::NC;
    const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t i = idx / 70;
    const int slot = (int)(idx % 70);
    if (i >= n) return;
    const T *f = base + i;
    const double v1 = (double)f[0], t0 = (double)f[1 * stride], t1 = (double)f[2 * stride];
    const double p0 = (double)f[(CB + 0) * stride], v0 = (double)f[(CB + 1) * stride], p1 = (double)f[(CB + 2) * stride];
    const double p2 = (double)f[(CB + 3) * stride], v2 = (double)f[(CB + 4) * stride];
    if (slot < 66) {
        const int seg = slot / 33, j = slot % 33;
        const double x0 = seg ? p1 : p0, x1 = seg ? p2 : p1, va = seg ? v1 : v0, vb = seg ? v2 : v1, h = seg ? t1 : t0;
        double out;
        if (j == 0) out = x0;
        else if (j == 32) out = x1;
        else {
            const double acc0 = (x1 - x0) * (6.0 / (h * h)) - (va * 4.0 + vb * 2.0) / h;
            const double jrk0 = (vb - va) * (2.0 / (h * h)) - acc0 * (2.0 / h);
            const double t = h * (double)j / 32.0;
            out = x0 + (va + (acc0 + jrk0 * (t / 3.0)) * (t / 2.0)) * t;
        }
        pos66[i * 66 + slot] = out;
    } else {
        const int a = slot - 66;
        double out;
        if (a == 0)      out = ((p1 - p0) * 6.0 / t0 + v0 * -4.0 + v1 * -2.0) / t0;
        else if (a == 1) out = ((p1 - p0) * -6.0 / t0 + v0 * 2.0 + v1 * 4.0) / t0;
        else if (a == 2) out = ((p2 - p1) * 6.0 / t1 + v1 * -4.0 + v2 * -2.0) / t1;
        else             out = ((p2 - p1) * -6.0 / t1 + v1 * 2.0 + v2 * 4.0) / t1;
        acc4[i * 4 + a] = out;
    }
}

inline unsigned grid_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

// dispatch on (dtype, variant)
#define RP_DISPATCH(b, ...)                                               \
    do {                                                                  \
        if ((b).dtype == 0) {                                             \
            using T = double;                                             \
            if ((b).variant == 3) { constexpr int V = 3; __VA_ARGS__; }   \
            else                  { constexpr int V = 4; __VA_ARGS__; }   \
        } else {                                                          \
            using T = float;                                              \
            if ((b).variant == 3) { constexpr int V = 3; __VA_ARGS__; }   \
            else                  { constexpr int V = 4; __VA_ARGS__; }   \
        }                                                                 \
    } while (0)

}  // namespace

hipError_t launch_steps(const BatchView &b, const HostParams &hp, int k, hipStream_t stream)
{
    if (k <= 0 || b.n == 0) return hipSuccess;
    RP_DISPATCH(b, hipLaunchKernelGGL((k_newton<T, V, false>), dim3(grid_for(b.n)), dim3(kBlock), 0, stream,
                                       (T *)b.base, b.stride, b.n, k, make_kparams<T>(hp, V), T(0), 0,
                                       (int32_t *)nullptr, (uint32_t *)nullptr, (unsigned long long *)nullptr));
    return hipGetLastError();
}

hipError_t launch_solve(const BatchView &b, const HostParams &hp, int k, double gap_tol, int max_iter, hipStream_t stream)
{
    if (b.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_zero_counter, dim3(1), dim3(1), 0, stream, b.counters);
    RP_DISPATCH(b, hipLaunchKernelGGL((k_newton<T, V, true>), dim3(grid_for(b.n)), dim3(kBlock), 0, stream,
                                       (T *)b.base, b.stride, b.n, k, make_kparams<T>(hp, V), (T)gap_tol, max_iter,
                                       b.iters, b.status, b.counters));
    return hipGetLastError();
}

hipError_t launch_reduce(const BatchView &b, const HostParams &hp, double host_steps, double *d_partials,
                         double *d_out4, hipStream_t stream)
{
    unsigned blocks = grid_for(b.n);
    if (blocks > 1024) blocks = 1024;
    RP_DISPATCH(b, hipLaunchKernelGGL((k_reduce_partial<T, V>), dim3(blocks), dim3(kBlock), 0, stream,
                                       (const T *)b.base, b.stride, b.n, make_kparams<T>(hp, V), b.status, d_partials));
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(64), 0, stream, d_partials, (int)blocks, b.counters, host_steps, d_out4);
    return hipGetLastError();
}

hipError_t launch_aos_to_soa(const BatchView &b, const double *d_aos, hipStream_t stream)
{
    const dim3 g(grid_for(b.n)), t(kBlock);
    if (b.dtype == 0) {
        if (b.variant == 3) hipLaunchKernelGGL((k_aos_to_soa<double, 16>), g, t, 0, stream, d_aos, (double *)b.base, b.stride, b.n);
        else                hipLaunchKernelGGL((k_aos_to_soa<double, 12>), g, t, 0, stream, d_aos, (double *)b.base, b.stride, b.n);
    } else {
        if (b.variant == 3) hipLaunchKernelGGL((k_aos_to_soa<float, 16>), g, t, 0, stream, d_aos, (float *)b.base, b.stride, b.n);
        else                hipLaunchKernelGGL((k_aos_to_soa<float, 12>), g, t, 0, stream, d_aos, (float *)b.base, b.stride, b.n);
    }
    return hipGetLastError();
}

hipError_t launch_soa_to_aos(const BatchView &b, double *d_aos, hipStream_t stream)
{
    const dim3 g(grid_for(b.n)), t(kBlock);
    if (b.dtype == 0) {
        if (b.variant == 3) hipLaunchKernelGGL((k_soa_to_aos<double, 16>), g, t, 0, stream, (const double *)b.base, b.stride, b.n, d_aos);
        else                hipLaunchKernelGGL((k_soa_to_aos<double, 12>), g, t, 0, stream, (const double *)b.base, b.stride, b.n, d_aos);
    } else {
        if (b.variant == 3) hipLaunchKernelGGL((k_soa_to_aos<float, 16>), g, t, 0, stream, (const float *)b.base, b.stride, b.n, d_aos);
        else                hipLaunchKernelGGL((k_soa_to_aos<float, 12>), g, t, 0, stream, (const float *)b.base, b.stride, b.n, d_aos);
    }
    return hipGetLastError();
}

hipError_t launch_init_feasible(const BatchView &b, const HostParams &hp, const double *d_pos0, const double *d_pos1,
                                const double *d_pos2, hipStream_t stream)
{
    RP_DISPATCH(b, hipLaunchKernelGGL((k_init_feasible<T, V>), dim3(grid_for(b.n)), dim3(kBlock), 0, stream,
                                       (T *)b.base, b.stride, b.n, hp.accel_limit, d_pos0, d_pos1, d_pos2));
    return hipGetLastError();
}

hipError_t launch_init_const(const BatchView &b, const double *host_state, hipStream_t stream)
{
    ConstState cs;
    const int m = state_len(b.variant);
    for (int f = 0; f < 16; ++f) cs.v[f] = f < m ? host_state[f] : 0.0;
    if (b.dtype == 0) hipLaunchKernelGGL((k_init_const<double>), dim3(grid_for(b.n)), dim3(kBlock), 0, stream, (double *)b.base, b.stride, b.n, m, cs);
    else              hipLaunchKernelGGL((k_init_const<float>), dim3(grid_for(b.n)), dim3(kBlock), 0, stream, (float *)b.base, b.stride, b.n, m, cs);
    return hipGetLastError();
}

hipError_t launch_nudge(const BatchView &b, int field, double delta, hipStream_t stream)
{
    if (b.dtype == 0) hipLaunchKernelGGL((k_nudge<double>), dim3(grid_for(b.n)), dim3(kBlock), 0, stream, (double *)b.base + (size_t)field * b.stride, b.n, delta);
    else              hipLaunchKernelGGL((k_nudge<float>), dim3(grid_for(b.n)), dim3(kBlock), 0, stream, (float *)b.base + (size_t)field * b.stride, b.n, (float)delta);
    return hipGetLastError();
}

hipError_t launch_clear_progress(const BatchView &b, hipStream_t stream)
{
    hipLaunchKernelGGL(k_clear_progress, dim3(grid_for(b.n)), dim3(kBlock), 0, stream, b.iters, b.status, b.n, b.counters);
    return hipGetLastError();
}

hipError_t launch_move_toward_feasibility(const BatchView &b, const HostParams &hp, hipStream_t stream)
{
    RP_DISPATCH(b, hipLaunchKernelGGL((k_move_toward_feasibility<T, V>), dim3(grid_for(b.n)), dim3(kBlock), 0, stream,
                                       (T *)b.base, b.stride, b.n, make_kparams<T>(hp, V)));
    return hipGetLastError();
}

hipError_t launch_sample(const BatchView &b, double *d_pos66, double *d_acc4, hipStream_t stream)
{
    const size_t total = b.n * 70;
    RP_DISPATCH(b, hipLaunchKernelGGL((k_sample<T, V>), dim3(grid_for(total)), dim3(kBlock), 0, stream,
                                       (const T *)b.base, b.stride, b.n, d_pos66, d_acc4));
    return hipGetLastError();
}

}  // namespace rp
