// ip_core.h -- per-lane device code of the batched interior-point Newton step (gfx950).
//
// One wavefront lane owns one problem; everything below is straight-line register code for
// that lane.  What it computes is the reference's moveInteriorPoint
// (onedpath_ip.cpp:810-953 for F3, onedpath2_ip.cpp:698-841 for F4) with one algebraic
// change: the (3+m)x(3+m) KKT system the reference hands to Eigen's column-pivoted QR
// (onedpath_ip.cpp:886-887) is condensed to the 3x3 Schur complement on the three free
// variables and solved in registers with partial pivoting:
//
//     M = [ S lam_i H_i     G^T     ]      d lam_i = -(lam_i + p/c_i) - (lam_i/c_i) g_i.dx
//         [ diag(lam) G   diag(c)   ]      K dx = -grad f + S_i g_i p/c_i
//                                          K = S lam_i H_i - S (lam_i/c_i) g_i g_i^T
//
// K is not positive definite in early iterations (about 1 % of steps), hence pivoted LU
// rather than Cholesky.  K(t0,t1) is structurally zero (no constraint touches both
// durations).  Everything after the solve -- fraction-to-boundary on the multipliers,
// feasibility backtracking, residual backtracking, the update -- is the reference's logic
// decision for decision.
//
// Arithmetic notes (tolerance of the path is 1e-10 relative, not bitwise):
//  * compiled with -ffp-contract=off; every fused multiply-add below is written out, so the
//    same expression gives the same bits wherever it is inlined.  That is what makes the two
//    memoisations exact: a trial point that is bitwise the current point re-uses the current
//    point's evaluation (the reference recomputes the same numbers), and the evaluation at
//    the accepted trial point is carried into the next step instead of being recomputed.
//  * 1/t0 and 1/t1 are formed once per trial point (the reference divides by t about 20
//    times per constraint sweep, onedpath_ip.cpp:385-391, 404-410).
#pragma once

#include <hip/hip_runtime.h>

namespace rp {

template <typename T> __device__ __forceinline__ T fma_(T a, T b, T c);
template <> __device__ __forceinline__ double fma_<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <> __device__ __forceinline__ float fma_<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename T> __device__ __forceinline__ T abs_(T a);
template <> __device__ __forceinline__ double abs_<double>(double a) { return __builtin_fabs(a); }
template <> __device__ __forceinline__ float abs_<float>(float a) { return __builtin_fabsf(a); }

template <typename T> __device__ __forceinline__ T sqrt_(T a);
template <> __device__ __forceinline__ double sqrt_<double>(double a) { return __builtin_sqrt(a); }
template <> __device__ __forceinline__ float sqrt_<float>(float a) { return __builtin_sqrtf(a); }

template <typename T> __device__ __forceinline__ bool finite_(T a) { return abs_(a) <= T(1.7976931348623157e308) && a == a; }
template <> __device__ __forceinline__ bool finite_<float>(float a) { return abs_(a) <= 3.4028234e38f && a == a; }

// Solver constants in the compute type (rp_params, include/rp_batch.h).
template <typename T> struct KParams {
    T limit;        // L
    T mu_den;       // m * mu_divisor: perturbation = gap / mu_den   (onedpath_ip.cpp:812)
    T boundary;     // 0.99
    T backtrack;    // 0.5
    T armijo;       // 0.01
    T c_floor;      // |c_i| below this is rounding noise of (a - L): condensed as -c_floor (see newton_step)
    int max_bt;     // 100
};

// The five per-problem constants (enum V 11..15) plus the two position deltas.
template <typename T> struct Prob {
    T v0, v2;      // vel0X, vel2X
    T dx0, dx1;    // pos1X - pos0X, pos2X - pos1X
};

// End accelerations of both segments and their first derivatives at one (v, t0, t1).
// index j: 0 = segment 0 initial, 1 = segment 0 final, 2 = segment 1 initial, 3 = segment 1 final
// (evalAccelInit / evalAccelFinal, onedpath_ip.cpp:372-392, 413-433).
template <typename T> struct Acc {
    T r0, r1;      // 1/t0, 1/t1
    T a[4];
    T gt[4];       // d a_j / d t_seg(j)
    T gv[4];       // d a_j / d vel1
};

template <int VARIANT> struct CMap;
template <> struct CMap<3> { static constexpr int NC = 8; };   // onedpath_ip.cpp:101
template <> struct CMap<4> { static constexpr int NC = 4; };   // onedpath2_ip.cpp

// accelerations only: enough for constraintsSatisfied (onedpath_ip.cpp:738-751)
template <typename T>
__device__ __forceinline__ void accel_values(const Prob<T> &k, T v, T t0, T t1, Acc<T> &e)
{
    const T r0 = T(1) / t0, r1 = T(1) / t1;
    e.r0 = r0;
    e.r1 = r1;
    const T u0 = k.dx0 * r0, u1 = k.dx1 * r1;                 // dX / t
    const T m0 = fma_(T(-4), k.v0, T(-2) * v);                // v0*-4 + v1*-2   (segment 0: v1 = vel1)
    const T n0 = fma_(T(2), k.v0, T(4) * v);                  // v0*2 + v1*4
    const T m1 = fma_(T(-4), v, T(-2) * k.v2);                // segment 1: v0 = vel1, v1 = vel2
    const T n1 = fma_(T(2), v, T(4) * k.v2);
    e.a[0] = fma_(T(6), u0, m0) * r0;
    e.a[1] = fma_(T(-6), u0, n0) * r0;
    e.a[2] = fma_(T(6), u1, m1) * r1;
    e.a[3] = fma_(T(-6), u1, n1) * r1;
}

// first derivatives, from the reciprocals already in e (dAdT, dAdV0/dAdV1 of :389-391, :430-432)
template <typename T>
__device__ __forceinline__ void accel_grads(const Prob<T> &k, T v, Acc<T> &e)
{
    const T r0 = e.r0, r1 = e.r1;
    const T u0 = k.dx0 * r0, u1 = k.dx1 * r1;
    const T m0 = fma_(T(-4), k.v0, T(-2) * v);
    const T n0 = fma_(T(2), k.v0, T(4) * v);
    const T m1 = fma_(T(-4), v, T(-2) * k.v2);
    const T n1 = fma_(T(2), v, T(4) * k.v2);
    const T q0 = r0 * r0, q1 = r1 * r1;
    e.gt[0] = fma_(T(-12), u0, -m0) * q0;
    e.gt[1] = fma_(T(12), u0, -n0) * q0;
    e.gt[2] = fma_(T(-12), u1, -m1) * q1;
    e.gt[3] = fma_(T(12), u1, -n1) * q1;
    e.gv[0] = T(-2) * r0;      // dAdV1 of the initial end, segment 0
    e.gv[1] = T(4) * r0;       // dAdV1 of the final end
    e.gv[2] = T(-4) * r1;      // dAdV0 of the initial end, segment 1
    e.gv[3] = T(2) * r1;       // dAdV0 of the final end
}

// second derivatives (evalAccelSecondDerivInit / Final, onedpath_ip.cpp:394-411, 435-452)
template <typename T>
__device__ __forceinline__ void accel_hess(const Prob<T> &k, T v, const Acc<T> &e, T (&htt)[4], T (&htv)[4])
{
    const T r0 = e.r0, r1 = e.r1;
    const T u0 = k.dx0 * r0, u1 = k.dx1 * r1;
    const T m0 = fma_(T(-4), k.v0, T(-2) * v);
    const T n0 = fma_(T(2), k.v0, T(4) * v);
    const T m1 = fma_(T(-4), v, T(-2) * k.v2);
    const T n1 = fma_(T(2), v, T(4) * k.v2);
    const T q0 = r0 * r0, q1 = r1 * r1;
    const T c0 = q0 * r0, c1 = q1 * r1;
    htt[0] = fma_(T(36), u0, T(2) * m0) * c0;      // (36 dX/t - 8 v0 - 4 v1) / t^3
    htt[1] = fma_(T(-36), u0, T(2) * n0) * c0;     // (-36 dX/t + 4 v0 + 8 v1) / t^3
    htt[2] = fma_(T(36), u1, T(2) * m1) * c1;
    htt[3] = fma_(T(-36), u1, T(2) * n1) * c1;
    htv[0] = T(2) * q0;       // sTV1, initial end
    htv[1] = T(-4) * q0;      // sTV1, final end
    htv[2] = T(4) * q1;       // sTV0, initial end
    htv[3] = T(-2) * q1;      // sTV0, final end
}

// ---- constraints built on the accelerations -------------------------------------------
// F3 (evalConstraint0..7, onedpath_ip.cpp:454-625): i -> accel i/2, even i: -a - L, odd i: a - L.
// F4 (evalConstraint0..3, onedpath2_ip.cpp:414-511): i -> accel i, (a^2 - L^2)/2.
template <typename T, int VARIANT>
__device__ __forceinline__ T c_value(int i, const Acc<T> &e, T L)
{
    if constexpr (VARIANT == 3) {
        const T a = e.a[i >> 1];
        return (i & 1) ? a - L : -a - L;
    } else {
        const T a = e.a[i];
        return (a * a - L * L) * T(0.5);
    }
}

// gradient of constraint i in (vel1, t_seg) coordinates; the other duration's entry is 0
template <typename T, int VARIANT>
__device__ __forceinline__ void c_grad(int i, const Acc<T> &e, T &gv, T &gt)
{
    if constexpr (VARIANT == 3) {
        const int j = i >> 1;
        gv = (i & 1) ? e.gv[j] : -e.gv[j];
        gt = (i & 1) ? e.gt[j] : -e.gt[j];
    } else {
        gv = e.a[i] * e.gv[i];
        gt = e.a[i] * e.gt[i];
    }
}

template <int VARIANT> __device__ __forceinline__ constexpr int c_segment(int i)
{
    return VARIANT == 3 ? (i >> 2) : (i >> 1);
}

// constraintsSatisfied (onedpath_ip.cpp:738-751): false iff some error > 0 (NaN passes, as there)
template <typename T, int VARIANT>
__device__ __forceinline__ bool all_satisfied(const Acc<T> &e, T L)
{
    bool ok = true;
#pragma unroll
    for (int i = 0; i < CMap<VARIANT>::NC; ++i) ok = ok && !(c_value<T, VARIANT>(i, e, L) > T(0));
    return ok;
}

// surrogateDualityGap (onedpath_ip.cpp:794-808)
template <typename T, int VARIANT>
__device__ __forceinline__ T duality_gap(const Acc<T> &e, const T (&lam)[CMap<VARIANT>::NC], T L)
{
    T mu = T(0);
#pragma unroll
    for (int i = 0; i < CMap<VARIANT>::NC; ++i) mu = fma_(-c_value<T, VARIANT>(i, e, L), lam[i], mu);
    return mu;
}

// residualNorm (onedpath_ip.cpp:753-792): || (grad f + G^T lam ; lam.c + p) ||^2
template <typename T, int VARIANT>
__device__ __forceinline__ T residual_norm(const Acc<T> &e, const T (&lam)[CMap<VARIANT>::NC], T p, T L)
{
    T rv = T(0), rt0 = T(1), rt1 = T(1), acc = T(0);
#pragma unroll
    for (int i = 0; i < CMap<VARIANT>::NC; ++i) {
        T gv, gt;
        c_grad<T, VARIANT>(i, e, gv, gt);
        rv = fma_(lam[i], gv, rv);
        if (c_segment<VARIANT>(i) == 0) rt0 = fma_(lam[i], gt, rt0);
        else                            rt1 = fma_(lam[i], gt, rt1);
        const T rc = fma_(lam[i], c_value<T, VARIANT>(i, e, L), p);
        acc = fma_(rc, rc, acc);
    }
    acc = fma_(rv, rv, acc);
    acc = fma_(rt0, rt0, acc);
    acc = fma_(rt1, rt1, acc);
    return acc;
}

// 3x3 solve, Gaussian elimination with partial pivoting (row of largest magnitude, first
// wins ties), branch-free.  A is symmetric with A[1][2] = 0 on entry but is treated as general.
template <typename T>
__device__ __forceinline__ void swap_if(bool c, T &a, T &b)
{
    const T ta = c ? b : a, tb = c ? a : b;
    a = ta;
    b = tb;
}

template <typename T> __device__ __forceinline__ T sdiv(T num, T den) { return den != T(0) ? num / den : T(0); }

template <typename T>
__device__ __forceinline__ void solve3(T a00, T a01, T a02, T a10, T a11, T a12, T a20, T a21, T a22,
                                       T b0, T b1, T b2, T &x0, T &x1, T &x2)
{
    // column 0
    {
        const T m0 = abs_(a00), m1 = abs_(a10), m2 = abs_(a20);
        const bool s1 = (m1 > m0) && !(m2 > m1);      // row 1 is the pivot
        const bool s2 = (m2 > m0) && (m2 > m1);       // row 2 is the pivot
        swap_if(s1, a00, a10); swap_if(s1, a01, a11); swap_if(s1, a02, a12); swap_if(s1, b0, b1);
        swap_if(s2, a00, a20); swap_if(s2, a01, a21); swap_if(s2, a02, a22); swap_if(s2, b0, b2);
    }
    // A zero pivot means a zero column: the reference's rank-revealing QR sets that component of
    // the step to 0 (ColPivHouseholderQR.h:609-610); dividing "by zero -> 0" does the same here.
    {
        const T l1 = sdiv(a10, a00), l2 = sdiv(a20, a00);
        a11 = fma_(-l1, a01, a11); a12 = fma_(-l1, a02, a12); b1 = fma_(-l1, b0, b1);
        a21 = fma_(-l2, a01, a21); a22 = fma_(-l2, a02, a22); b2 = fma_(-l2, b0, b2);
    }
    // column 1
    {
        const bool s = abs_(a21) > abs_(a11);
        swap_if(s, a11, a21); swap_if(s, a12, a22); swap_if(s, b1, b2);
    }
    {
        const T l2 = sdiv(a21, a11);
        a22 = fma_(-l2, a12, a22); b2 = fma_(-l2, b1, b2);
    }
    x2 = sdiv(b2, a22);
    x1 = sdiv(fma_(-a12, x2, b1), a11);
    x0 = sdiv(fma_(-a02, x2, fma_(-a01, x1, b0)), a00);
}

// ---- one Newton step -------------------------------------------------------------------
// In:  x = (v, t0, t1), lam, and e = values + grads at x.   Out: the same at the new point.
template <typename T, int VARIANT>
__device__ __forceinline__ void newton_step(const Prob<T> &k, const KParams<T> &kp,
                                            T &v, T &t0, T &t1, T (&lam)[CMap<VARIANT>::NC], Acc<T> &e)
{
    constexpr int NC = CMap<VARIANT>::NC;
    const T L = kp.limit;

    // -- assemble (onedpath_ip.cpp:812-861, condensed) --
    T htt[4], htv[4];
    accel_hess(k, v, e, htt, htv);

    T c[NC], gv[NC], gt[NC];
    bool feasible_here = true;
    T gap = T(0);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        c[i] = c_value<T, VARIANT>(i, e, L);
        c_grad<T, VARIANT>(i, e, gv[i], gt[i]);
        feasible_here = feasible_here && !(c[i] > T(0));
        gap = fma_(-c[i], lam[i], gap);
    }
    const T p = gap / kp.mu_den;

    T kvv = T(0), kv0 = T(0), kv1 = T(0), k00 = T(0), k11 = T(0);
    T bv = T(0), b0 = T(-1), b1 = T(-1);
    T w[NC], pc[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        // An active constraint converges to c_i = 0, sometimes exactly (the default problem ends at
        // a = L bit for bit).  The reference's 11x11 QR is indifferent to a zero on the diagonal;
        // the condensation divides by c_i, so values inside the rounding noise of (a - L) are
        // condensed as the negative number of that size.  The reference row lam_i g_i.dx + c_i dlam_i
        // = -(lam_i c_i + p) is then solved with c_i perturbed by less than one ulp of L.
        const T cs = (abs_(c[i]) < kp.c_floor) ? -kp.c_floor : c[i];
        const T ic = T(1) / cs;
        w[i] = lam[i] * ic;
        pc[i] = p * ic;
        // Hessian of constraint i: (t,t) and (t,v) entries only; F4 leaves (v,v) at zero
        // although d2/dv2 of a^2/2 is not (onedpath2_ip.cpp:446-448) -- reproduced.
        T Htt, Htv;
        if constexpr (VARIANT == 3) {
            const int j = i >> 1;
            Htt = (i & 1) ? htt[j] : -htt[j];
            Htv = (i & 1) ? htv[j] : -htv[j];
        } else {
            Htt = fma_(e.gt[i], e.gt[i], e.a[i] * htt[i]);
            Htv = fma_(e.gt[i], e.gv[i], e.a[i] * htv[i]);
        }
        const T wgv = w[i] * gv[i], wgt = w[i] * gt[i];
        kvv = fma_(-wgv, gv[i], kvv);
        bv = fma_(gv[i], pc[i], bv);
        if (c_segment<VARIANT>(i) == 0) {
            kv0 = fma_(lam[i], Htv, fma_(-wgv, gt[i], kv0));
            k00 = fma_(lam[i], Htt, fma_(-wgt, gt[i], k00));
            b0 = fma_(gt[i], pc[i], b0);
        } else {
            kv1 = fma_(lam[i], Htv, fma_(-wgv, gt[i], kv1));
            k11 = fma_(lam[i], Htt, fma_(-wgt, gt[i], k11));
            b1 = fma_(gt[i], pc[i], b1);
        }
    }

    // -- solve K dx = rhs, recover d lam --
    T dxv, dx0, dx1;
    solve3<T>(kvv, kv0, kv1, kv0, k00, T(0), kv1, T(0), k11, bv, b0, b1, dxv, dx0, dx1);

    T dl[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const T gdx = fma_(gv[i], dxv, gt[i] * (c_segment<VARIANT>(i) == 0 ? dx0 : dx1));
        dl[i] = fma_(-w[i], gdx, -(lam[i] + pc[i]));
    }

    // -- fraction to the boundary on the multipliers (onedpath_ip.cpp:903-915) --
    T s = T(1);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        if (dl[i] < T(0)) {
            const T q = -lam[i] / dl[i];
            s = (q < s) ? q : s;
        }
    }
    s *= kp.boundary;

    // -- backtrack until primal feasible (onedpath_ip.cpp:919-928) --
    Acc<T> et;                 // evaluation at the current trial point
    T tv = v, tt0 = t0, tt1 = t1;
    bool et_valid = false;     // et holds accel values of (tv,tt0,tt1) == x + s*dx
    for (int it = 0; it < kp.max_bt; ++it) {
        tv = fma_(dxv, s, v);
        tt0 = fma_(dx0, s, t0);
        tt1 = fma_(dx1, s, t1);
        bool ok;
        if (tv == v && tt0 == t0 && tt1 == t1) {
            ok = feasible_here;            // same point, same answer
            et_valid = false;
        } else {
            accel_values(k, tv, tt0, tt1, et);
            ok = all_satisfied<T, VARIANT>(et, L);
            et_valid = true;
        }
        if (ok) break;
        s *= kp.backtrack;
        et_valid = false;
    }

    // -- backtrack until the residual decreases (onedpath_ip.cpp:932-945) --
    const T r0n = residual_norm<T, VARIANT>(e, lam, p, L);
    T tl[NC];
    bool accepted_eval = false;    // et = values + grads at the point the loop broke on
    bool accepted_same = false;    // ... which is bitwise the current point
    for (int it = 0; it < kp.max_bt; ++it) {
        tv = fma_(dxv, s, v);
        tt0 = fma_(dx0, s, t0);
        tt1 = fma_(dx1, s, t1);
        bool same_x = (tv == v && tt0 == t0 && tt1 == t1);
        bool same_l = true;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            tl[i] = fma_(dl[i], s, lam[i]);
            same_l = same_l && (tl[i] == lam[i]);
        }
        T rn;
        if (same_x && same_l) {
            rn = r0n;
        } else if (same_x) {
            rn = residual_norm<T, VARIANT>(e, tl, p, L);
        } else {
            if (!et_valid) accel_values(k, tv, tt0, tt1, et);
            accel_grads(k, tv, et);
            rn = residual_norm<T, VARIANT>(et, tl, p, L);
        }
        et_valid = false;
        if (rn <= r0n * (T(1) - kp.armijo * s)) {
            accepted_eval = !same_x;
            accepted_same = same_x;
            break;
        }
        s *= kp.backtrack;
    }

    // -- take the step (onedpath_ip.cpp:949-952) --
    v = fma_(dxv, s, v);
    t0 = fma_(dx0, s, t0);
    t1 = fma_(dx1, s, t1);
#pragma unroll
    for (int i = 0; i < NC; ++i) lam[i] = fma_(dl[i], s, lam[i]);

    if (accepted_eval) {
        e = et;
    } else if (!accepted_same) {
        // loop ran out of halvings: the accepted s was never evaluated
        accel_values(k, v, t0, t1, e);
        accel_grads(k, v, e);
    }
}

}  // namespace rp
