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
// F3's eight constraints are four (-a_j - L, a_j - L) pairs on the four end accelerations,
// with opposite gradients and Hessians, so the assembly works per acceleration:
//     S lam g      = (lp - lm) grad a          S lam H = (lp - lm) hess a
//     S (lam/c) gg = (lm/cm + lp/cp) grad a grad a^T
//     S g p/c      = p (1/cp - 1/cm) grad a
// and all eight 1/cm_j, 1/cp_j come from ONE reciprocal (of the product of the four cm_j cp_j).
//
// Arithmetic notes (tolerance of the path is 1e-10 relative, not bitwise):
//  * compiled with -ffp-contract=off; every fused multiply-add below is written out, so the
//    same expression gives the same bits wherever it is inlined.  That is what makes the
//    re-uses exact: a trial point that is bitwise the current point re-uses the current
//    point's evaluation (the reference recomputes the same numbers; MEMO), the evaluation at
//    the accepted trial point is carried into the next step instead of being recomputed, and in
//    the gated kernels so are the sums its residual test was made of (residual_sums).
//  * the kernels are fp64-ALU bound (the vector ALU issues ~80 % of the cycles of a power-capped clock), so the
//    currency is instructions: ~300 per gated Newton step (profiles/r4_sq_counters.json has the count of the
//    shipped kernels).  An IEEE fp64 division costs 11 of them on gfx950 (v_div_scale x2, v_rcp, 5 fma,
//    v_div_fmas, v_div_fixup); every division on the path is a reciprocal, v_rcp_f64 + ONE cubic refinement
//    (rcp_ below: 4 instructions, correctly rounded but for ~1e-7 of the inputs, the issue time of ~6
//    multiplications), and reciprocals are batched: one for the two durations of a trial point (the reference
//    divides by t about 20 times per constraint sweep, onedpath_ip.cpp:385-391, 404-410), one for all
//    constraints, one for the two arrow pivots, one for the boundary fraction -- 4-5 per step where the first
//    version of this file had 19.  Define RP_EXACT_DIV to build with correctly rounded divisions instead (A/B builds).
//  * two forms of the step.  newton_step_inplace (the gated solve, and since round 4 every fixed-step launch of F3): the trial
//    overwrites the state, the step's start waits in LDS (or, for small batches, registers), residual sums carried, loops tested
//    by ballots over comparisons; FROZEN adds the post-convergence regime's search on affine pieces with its certain failures
//    counted in closed form.  newton_step_to (F4's fixed-step launches, mu_mode 1): template switches MEMO (exact memoisation
//    for the reference's post-convergence regime), AFFINE (with MEMO: that regime's residual loop on affine pieces), MU
//    (rp_params.mu_mode), WAVE (stragglers of the residual loop served by the whole wave).  Both take a bookkeeping hook
//    (halving counts for rp_batch_step_counted).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

namespace rp {

template <typename T> __device__ __forceinline__ T fma_(T a, T b, T c);
template <> __device__ __forceinline__ double fma_<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <> __device__ __forceinline__ float fma_<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename T> __device__ __forceinline__ T abs_(T a);
template <> __device__ __forceinline__ double abs_<double>(double a) { return __builtin_fabs(a); }
template <> __device__ __forceinline__ float abs_<float>(float a) { return __builtin_fabsf(a); }

template <typename T> __device__ __forceinline__ T ldexp_(T a, int e);
template <> __device__ __forceinline__ double ldexp_<double>(double a, int e) { return __builtin_ldexp(a, e); }
template <> __device__ __forceinline__ float ldexp_<float>(float a, int e) { return __builtin_ldexpf(a, e); }

template <typename T> __device__ __forceinline__ T sqrt_(T a);
template <> __device__ __forceinline__ double sqrt_<double>(double a) { return __builtin_sqrt(a); }
template <> __device__ __forceinline__ float sqrt_<float>(float a) { return __builtin_sqrtf(a); }

template <typename T> __device__ __forceinline__ T min_(T a, T b);      // NaN in one operand -> the other
template <> __device__ __forceinline__ double min_<double>(double a, double b) { return __builtin_fmin(a, b); }
template <> __device__ __forceinline__ float min_<float>(float a, float b) { return __builtin_fminf(a, b); }

template <typename T> __device__ __forceinline__ T max_(T a, T b);      // NaN in one operand -> the other
template <> __device__ __forceinline__ double max_<double>(double a, double b) { return __builtin_fmax(a, b); }
template <> __device__ __forceinline__ float max_<float>(float a, float b) { return __builtin_fmaxf(a, b); }

template <typename T> __device__ __forceinline__ bool finite_(T a) { return abs_(a) <= T(1.7976931348623157e308) && a == a; }
template <> __device__ __forceinline__ bool finite_<float>(float a) { return abs_(a) <= 3.4028234e38f && a == a; }

// reciprocal
template <typename T> __device__ __forceinline__ T rcp_(T x);
#ifdef RP_EXACT_DIV
template <> __device__ __forceinline__ double rcp_<double>(double x) { return 1.0 / x; }
template <> __device__ __forceinline__ float rcp_<float>(float x) { return 1.0f / x; }
#else
template <> __device__ __forceinline__ double rcp_<double>(double x)
{
    // v_rcp_f64 is good to 2^-24 (4.6e-8 measured); ONE cubic refinement, r (1 + e + e^2) with e = 1 - x r, leaves a truncation
    // error e^3 ~ 1e-22 relative before the final rounding: three multiply-adds.  That is CORRECTLY ROUNDED EXCEPT WITH
    // PROBABILITY ~1e-7 per input (e^3 / half an ulp; no misrounding among 4.19 M samples, profiles/probes/rcp_probe.hip) -- not
    // IEEE-exact: a 1 Mi-problem solve makes ~1e8 reciprocals and will contain a few 1-ulp differences from true division, from
    // the RP_EXACT_DIV build and from the two Newton steps (four multiply-adds, misrounding ~1e-13) this was until late in
    // round 3.  "Bit-identical" claims elsewhere are A/B comparisons between builds that share this function.
    // (One Newton step, 10 ulp, fails parity: profiles/r3_tuning.md.)
    const double r = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, __builtin_fma(e, e, e), r);
}
template <> __device__ __forceinline__ float rcp_<float>(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    r = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
    return r;
}
#endif
// One Newton refinement: <= 10 ulp (measured on gfx950: raw v_rcp_f64 is 2^-24, one Newton step 2.2e-15 --
// profiles/probes/rcp_probe.hip).  Used only where the quotient feeds a bound, not the iterate: the
// fraction-to-boundary ratios.
template <typename T> __device__ __forceinline__ T rcp1_(T x);
#ifdef RP_EXACT_DIV
template <> __device__ __forceinline__ double rcp1_<double>(double x) { return 1.0 / x; }
#else
template <> __device__ __forceinline__ double rcp1_<double>(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}
#endif
template <> __device__ __forceinline__ float rcp1_<float>(float x) { return rcp_<float>(x); }
// The 32-bit word of x that holds its sign bit, and the OR of three such words as ONE opaque instruction (written in C the
// compiler widens an OR of high words back into 64-bit ORs of the whole numbers: twice the instructions for the same bit).
template <typename T> __device__ __forceinline__ unsigned sign_word(T x)
{
    if constexpr (sizeof(T) == 8) return (unsigned)(__builtin_bit_cast(unsigned long long, x) >> 32);
    else return __builtin_bit_cast(unsigned, x);
}
__device__ __forceinline__ unsigned or3_(unsigned a, unsigned b, unsigned c)
{
    unsigned o;
    asm("v_or3_b32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
}
// 1/den, or 0 for a zero denominator (rank-deficient pivot: that component of the step is 0)
template <typename T> __device__ __forceinline__ T srcp_(T den) { return den != T(0) ? rcp_(den) : T(0); }

// x of lane `src` (wave-uniform index) in every lane
__device__ __forceinline__ int bcast_(int x, int src) { return __builtin_amdgcn_readlane(x, src); }
__device__ __forceinline__ float bcast_(float x, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src)); }
__device__ __forceinline__ double bcast_(double x, int src)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

#ifndef RP_FROZEN_PROOF
#define RP_FROZEN_PROOF 1      // 0: every trial of the frozen search is evaluated (A/B builds: no decision may change)
#endif
#ifndef RP_FROZEN_MASKS
#define RP_FROZEN_MASKS 1     // the fixed-step kernels' feasibility loop (newton_step_inplace<FROZEN>) keeps its per-lane flag as a 64-bit lane mask in scalar
                              // registers (0: the bool form, A/B).  Measured at 65,536 x 50 (profiles/r6_tuning.md): 289 -> 246 scalar instructions per
                              // post-convergence wave-step, 0.200 -> 0.194 ms.  The same treatment of the residual loop costs 40-70 VGPRs (spills at every
                              // occupancy) whichever way the mask becomes a predicate again, and a wave-uniform form of the frozen search trades its 17
                              // scalar instructions per trip for 5 vector ones and is slower: neither is in the source.
#endif
// lane mask <-> lane predicate without a vector instruction (the mask must be wave-uniform, which a ballot's result is)
__device__ __forceinline__ unsigned long long ballot_(bool c) { return __builtin_amdgcn_ballot_w64(c); }
__device__ __forceinline__ bool in_mask_(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// Solver constants in the compute type (rp_params, include/rp_batch.h).
template <typename T> struct KParams {
    T limit;        // L
    T inv_mu_den;   // 1 / (m * mu_divisor): perturbation = gap / (m * 10)   (onedpath_ip.cpp:812)
    T boundary;     // 0.99
    T backtrack;    // 0.5
    T armijo;       // 0.01
    T c_floor;      // L * eps / 256: the shift applied to every c_i before it is inverted (see c_guard)
    T x_floor;      // 2 L c_floor: the same shift on the product cm cp of an F3 constraint pair
    int max_bt;     // 100
    int stall_window;   // 0 = off; see rp_params.stall_window
    T sigma_try[2];     // mu_mode 1: centring parameters tried (ascending) before the reference's 1/mu_divisor; see newton_step
    int handoff_lanes, handoff_patience;      // the gated solve's straggler hand-off (k_solve_chunks<ROUNDS>): a wave whose stepping lanes have numbered <=
                                              // handoff_lanes for more than handoff_patience steps stops and leaves them open (0 lanes: never)
};

// The per-problem constants the step needs (enum V 11..15 reduced to velocities and deltas).
// ZV = "vel0X and vel2X are zero for every problem of the batch".  Every start state the reference has
// leaves them at zero (initDefault / initStuck, onedpath_ip.cpp:179-186, 203-210) and no key changes them, so
// this is the normal case; the batch keeps a flag and only a set_state / nudge with non-zero end velocities
// selects the general instantiation.  With ZV the two fields are neither read nor held, and the four velocity
// combinations lose a multiply-add each; the results are bit-identical to the general code on zero inputs
// (fma(c, 0, x) == x exactly).
// LEAN: the instantiating kernel has no registers to spare (F4 in double precision with double storage, 166 of the 168 that three
// waves allow): optional accelerators that change no result (ray_proof) are left out.
template <typename T, bool ZV = false, bool LEAN = false> struct Prob {
    static constexpr bool zero_vel = ZV;
    static constexpr bool lean = LEAN;
    T v0, v2;      // vel0X, vel2X (unused when ZV)
    T dx0, dx1;    // pos1X - pos0X, pos2X - pos1X
};
// The same with the two deltas parked in LDS (the gated kernel, which needs its registers for a fourth wave per SIMD): a use is
// a ds_read, which issues beside the arithmetic.  `at` points at this lane's value; volatile for the same reason as LdsBackup.
template <typename T> using LdsBackup = __attribute__((address_space(3))) volatile T *;      // LDS address space kept in the type: ds_read / ds_write, not flat accesses
template <typename T> struct LdsConst {
    __attribute__((address_space(3))) volatile T *at;
    __device__ __forceinline__ operator T() const { return *at; }
};
template <typename T, bool ZV = false> struct ProbLds {
    static constexpr bool zero_vel = ZV;
    static constexpr bool lean = false;
    T v0, v2;
    LdsConst<T> dx0, dx1;
};
// The velocity combinations of the four end accelerations (v0*-4 + v1*-2 etc., onedpath_ip.cpp:387, 428).  With zero end
// velocities they are multiples of vel1 by powers of two, formed once (v2 = 2v, v4 = 4v, v8 = 8v: exact) and signed for free.
template <typename T, class P> __device__ __forceinline__ T seg0_m(const P &k, T v) { if constexpr (P::zero_vel) return -(v + v); else return fma_(T(-4), k.v0, T(-2) * v); }   // v0*-4 + v1*-2
template <typename T, class P> __device__ __forceinline__ T seg0_n(const P &k, T v) { if constexpr (P::zero_vel) return T(4) * v; else return fma_(T(2), k.v0, T(4) * v); }      // v0*2 + v1*4
template <typename T, class P> __device__ __forceinline__ T seg1_m(const P &k, T v) { if constexpr (P::zero_vel) return -(T(4) * v); else return fma_(T(-4), v, T(-2) * k.v2); }   // segment 1: v0 = vel1, v1 = vel2
template <typename T, class P> __device__ __forceinline__ T seg1_n(const P &k, T v) { if constexpr (P::zero_vel) return v + v; else return fma_(T(2), v, T(4) * k.v2); }

// End accelerations of both segments and their first derivatives at one (v, t0, t1).
// index j: 0 = segment 0 initial, 1 = segment 0 final, 2 = segment 1 initial, 3 = segment 1 final
// (evalAccelInit / evalAccelFinal, onedpath_ip.cpp:372-392, 413-433).
template <typename T> struct Acc {
    T r0, r1;      // 1/t0, 1/t1
    T a[4];
    T gt[4];       // d a_j / d t_seg(j)   (not carried from step to step: rebuilt by accel_grads, see newton_step)
};
// What the evaluation of a point leaves behind for the stages that follow it (the time derivatives, the next direction's second
// derivatives): u = dX / t of both segments and the four velocity combinations -- exact re-uses, the same products wherever they
// are formed.  (With zero end velocities the combinations are +-2v and +-4v: two registers.)
template <typename T> struct PointAux {
    T u0, u1;
    T m0, n0, m1, n1;      // (zero end velocities: only n0 = 4v and n1 = 2v are kept -- m0 = -n1 and m1 = -n0 are sign modifiers at their uses, not values)
    template <class P> __device__ __forceinline__ T M0() const { if constexpr (P::zero_vel) return -n1; else return m0; }
    template <class P> __device__ __forceinline__ T M1() const { if constexpr (P::zero_vel) return -n0; else return m1; }
};
// A small constant held in a scalar register: as a literal a multiply-add has to be the two-address form, which overwrites its
// addend -- and the addends here (the velocity combinations) are needed again, so each would be copied first.
template <typename T> __device__ __forceinline__ T scalar_const(T c)
{
    asm("" : "+s"(c));
    return c;
}

// What a lane carries from one step to the next: the reciprocals and the four accelerations.
// GT = false: the time derivatives are rebuilt at the start of the step (14 flop) -- eight registers that are then free in
// the residual backtracking loop, the register peak of the fixed-step kernels (memoisation state on top).  GT = true
// (gated kernels, which have the registers): they are carried too.
// RS = true (gated reference-mode kernels): also the three sums the residual test and the gate are made of, taken at
// the accepted trial point -- X = |grad f + G^T lam|^2, Q1 = S lam_i c_i (= -gap), Q2 = S (lam_i c_i)^2; see residual_sums.
template <typename T, bool GT = false, bool RS = false> struct AccCarry {
    static constexpr bool has_sums = RS;
    T r0, r1;
    T a[4];
    T gt[GT ? 4 : 1];
    T X, Q1, Q2;      // unused (and not live) unless RS
    T cm[RS ? 4 : 1], cp[RS ? 4 : 1];      // with RS, F3: the constraint values -a_j - L, a_j - L the sums were formed from -- the next direction starts from them
    PointAux<T> x;                         // with RS: dX / t and the velocity combinations of the point (the direction's second derivatives re-use them)
};

// d a_j / d vel1: dAdV1 of segment 0's ends (-2/t0, 4/t0), dAdV0 of segment 1's ends (-4/t1, 2/t1)
// (onedpath_ip.cpp:390-391, 431-432).  Not stored: one multiply from the reciprocals.
template <typename T> __device__ __forceinline__ T acc_gv(const Acc<T> &e, int j)
{
    return j == 0 ? T(-2) * e.r0 : j == 1 ? T(4) * e.r0 : j == 2 ? T(-4) * e.r1 : T(2) * e.r1;
}

template <int VARIANT> struct CMap;
template <> struct CMap<3> { static constexpr int NC = 8; };   // onedpath_ip.cpp:101
template <> struct CMap<4> { static constexpr int NC = 4; };   // onedpath2_ip.cpp

// accelerations only: enough for constraintsSatisfied (onedpath_ip.cpp:738-751)
// (kdx0, kdx1: the problem's two deltas, passed in by callers that have read them from LDS together with other values)
// KEEP: the caller carries x on (the in-place step), so the constants come from scalar registers (scalar_const)
template <typename T, class P, bool KEEP = true>
__device__ __forceinline__ void accel_values_u(const P &k, T v, T t0, T t1, Acc<T> &e, PointAux<T> &x, T kdx0, T kdx1)
{
    const T rr = rcp_(t0 * t1);                               // one reciprocal for both durations: 1/t0 = t1/(t0 t1)
    const T r0 = t1 * rr, r1 = t0 * rr;
    e.r0 = r0;
    e.r1 = r1;
    x.u0 = kdx0 * r0;                                         // dX / t
    x.u1 = kdx1 * r1;
    x.n0 = seg0_n<T>(k, v);                                   // segment 0: v1 = vel1
    x.n1 = seg1_n<T>(k, v);                                   // segment 1: v0 = vel1
    if constexpr (!P::zero_vel) { x.m0 = seg0_m<T>(k, v); x.m1 = seg1_m<T>(k, v); }
    const T six = KEEP ? scalar_const(T(6)) : T(6);
    e.a[0] = fma_(six, x.u0, x.template M0<P>()) * r0;
    e.a[1] = fma_(-six, x.u0, x.n0) * r0;
    e.a[2] = fma_(six, x.u1, x.template M1<P>()) * r1;
    e.a[3] = fma_(-six, x.u1, x.n1) * r1;
}
template <typename T, class P, bool KEEP = true>
__device__ __forceinline__ void accel_values_u(const P &k, T v, T t0, T t1, Acc<T> &e, PointAux<T> &x)
{
    const T kdx0 = k.dx0, kdx1 = k.dx1;      // (read first: from LDS in the in-place kernels, under the reciprocal)
    accel_values_u<T, P, KEEP>(k, v, t0, t1, e, x, kdx0, kdx1);
}
template <typename T, class P>
__device__ __forceinline__ void accel_values(const P &k, T v, T t0, T t1, Acc<T> &e)
{
    PointAux<T> x;
    accel_values_u<T, P, false>(k, v, t0, t1, e, x);
}

// first derivatives, from the reciprocals already in e (dAdT, dAdV0/dAdV1 of :389-391, :430-432) and what the values left behind
template <typename T, class P, bool KEEP = true>
__device__ __forceinline__ void accel_grads_u(Acc<T> &e, const PointAux<T> &x)
{
    const T r0 = e.r0, r1 = e.r1;
    const T q0 = r0 * r0, q1 = r1 * r1;
    const T twelve = KEEP ? scalar_const(T(12)) : T(12);
    e.gt[0] = fma_(-twelve, x.u0, -x.template M0<P>()) * q0;
    e.gt[1] = fma_(twelve, x.u0, -x.n0) * q0;
    e.gt[2] = fma_(-twelve, x.u1, -x.template M1<P>()) * q1;
    e.gt[3] = fma_(twelve, x.u1, -x.n1) * q1;
}
template <typename T, class P>
__device__ __forceinline__ void accel_grads(const P &k, T v, Acc<T> &e)
{
    PointAux<T> x;
    x.u0 = k.dx0 * e.r0; x.u1 = k.dx1 * e.r1;
    x.n0 = seg0_n<T>(k, v); x.n1 = seg1_n<T>(k, v);
    if constexpr (!P::zero_vel) { x.m0 = seg0_m<T>(k, v); x.m1 = seg1_m<T>(k, v); }
    accel_grads_u<T, P, false>(e, x);
}

// second derivatives (evalAccelSecondDerivInit / Final, onedpath_ip.cpp:394-411, 435-452)
template <typename T, class P>
__device__ __forceinline__ void accel_hess(const P &k, T v, const Acc<T> &e, T (&htt)[4], T (&htv)[4])
{
    const T r0 = e.r0, r1 = e.r1;
    const T u0 = k.dx0 * r0, u1 = k.dx1 * r1;
    const T m0 = seg0_m<T>(k, v), n0 = seg0_n<T>(k, v);
    const T m1 = seg1_m<T>(k, v), n1 = seg1_n<T>(k, v);
    const T q0 = r0 * r0, q1 = r1 * r1;
    const T c0 = q0 * r0, c1 = q1 * r1;
    T m0d, n0d, m1d, n1d;      // 2 m0, 2 n0, 2 m1, 2 n1
    if constexpr (P::zero_vel) { m0d = m1; n0d = T(8) * v; m1d = -n0d; n1d = n0; }      // -4v, 8v, -8v, 4v: already there
    else { m0d = T(2) * m0; n0d = T(2) * n0; m1d = T(2) * m1; n1d = T(2) * n1; }
    htt[0] = fma_(T(36), u0, m0d) * c0;      // (36 dX/t - 8 v0 - 4 v1) / t^3
    htt[1] = fma_(T(-36), u0, n0d) * c0;     // (-36 dX/t + 4 v0 + 8 v1) / t^3
    htt[2] = fma_(T(36), u1, m1d) * c1;
    htt[3] = fma_(T(-36), u1, n1d) * c1;
    htv[0] = T(2) * q0;       // sTV1, initial end
    htv[1] = T(-4) * q0;      // sTV1, final end
    htv[2] = T(4) * q1;       // sTV0, initial end
    htv[3] = T(-2) * q1;      // sTV0, final end
}

// ---- constraints built on the accelerations -------------------------------------------
// F3 (evalConstraint0..7, onedpath_ip.cpp:454-625): i -> accel i/2, even i: -a - L, odd i: a - L.
// F4 (evalConstraint0..3, onedpath2_ip.cpp:414-511): i -> accel i, (a^2 - L^2)/2.
template <typename T, int VARIANT, class A = Acc<T>>
__device__ __forceinline__ T c_value(int i, const A &e, T L)
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
        gv = (i & 1) ? acc_gv(e, j) : -acc_gv(e, j);
        gt = (i & 1) ? e.gt[j] : -e.gt[j];
    } else {
        gv = e.a[i] * acc_gv(e, i);
        gt = e.a[i] * e.gt[i];
    }
}

template <int VARIANT> __device__ __forceinline__ constexpr int c_segment(int i)
{
    return VARIANT == 3 ? (i >> 2) : (i >> 1);
}

// constraintsSatisfied (onedpath_ip.cpp:738-751): false iff some error > 0 (NaN passes, as there).
// F3: -a - L > 0 or a - L > 0  <=>  |a| > L exactly (a floating-point difference has the sign of
// the exact difference), so one compare per acceleration.
template <typename T, int VARIANT, class A = Acc<T>>
__device__ __forceinline__ bool all_satisfied(const A &e, T L)
{
    bool ok = true;
    if constexpr (VARIANT == 3) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ok = ok && !(abs_(e.a[j]) > L);
    } else {
        // (a^2 - L^2) / 2 > 0  <=>  fl(a^2) > fl(L^2): a floating-point difference has the sign of the comparison of its
        // operands and halving keeps it (NaN: both false) -- one multiply and one compare per acceleration
        const T l2 = L * L;
#pragma unroll
        for (int i = 0; i < 4; ++i) ok = ok && !(e.a[i] * e.a[i] > l2);
    }
    return ok;
}

// "This trial point violates an acceleration limit BEYOND DOUBT" -- a division-free proof, used to walk the feasibility loop
// (onedpath_ip.cpp:919-928) past the halvings whose verdict cannot be in question.  F4 never converges: it sits on an
// acceleration limit with a Newton direction that points outward, and on alternate steps the reference halves the step 12 to 19
// times before the trial point is back inside -- the violation shrinks with s, |a| / L - 1 ~ c 2^-k at the k-th halving, and stays
// far above rounding until the very end.  In exact arithmetic the four end accelerations are
//     a_0 = (6 dX0 + m0 t0) / t0^2,  a_1 = (-6 dX0 + n0 t0) / t0^2   (and the same with dX1, m1, n1, t1 for segment 1),
// m, n the velocity combinations of accel_values, so |a_j| > L  <=>  |N_j| > L t^2 with the numerators N_j: four multiply-adds
// and two squares where the evaluation proper costs a reciprocal and 20 operations.  The evaluation proper computes
// a_j = ((+-6 (dX r) + m) r) with r = 1/t good to 3 eps, i.e. with an absolute error below 8 eps (|6 dX| / t^2 + |a_j|); N_j and
// L t^2 here carry 2 eps (|N_j| + |6 dX|) and 2 eps L t^2.  The proof therefore demands
//     |N_j| > L t^2 (1 + 64 eps) + 64 eps |6 dX|,
// several times the sum of both errors: whenever it holds, the evaluation proper finds |a_j| > L as well (and so, up to the
// rounding-level differences the parity tests measure anyway, does the reference: constraintsSatisfied, error > 0).  Anything
// else -- including NaNs, which compare false -- is left to the evaluation proper.  No decision changes; trials whose outcome
// is certain are not evaluated.
template <typename T, class P>
__device__ __forceinline__ bool infeasible_beyond_doubt(const P &k, T L, T v, T t0, T t1)
{
    constexpr T margin = sizeof(T) == 8 ? T(64.0 * 2.220446049250313e-16) : T(64.0 * 1.1920929e-7);
    const T six0 = T(6) * k.dx0, six1 = T(6) * k.dx1;
    const T lim = L * (T(1) + margin);
    const T b0 = fma_(lim, t0 * t0, margin * abs_(six0)), b1 = fma_(lim, t1 * t1, margin * abs_(six1));
    const T n0 = max_(abs_(fma_(seg0_m<T>(k, v), t0, six0)), abs_(fma_(seg0_n<T>(k, v), t0, -six0)));
    const T n1 = max_(abs_(fma_(seg1_m<T>(k, v), t1, six1)), abs_(fma_(seg1_n<T>(k, v), t1, -six1)));
    return n0 > b0 || n1 > b1;
}

// The same proof along the whole ray x + s dx, for ONE limit, in closed form.  In 99.4 % of F4's long halving sequences the
// limit that is still broken at the last rejected trial is broken at every trial before it (oracle trajectories,
// profiles/r3_tuning.md), and along the ray its numerator and L t^2 are quadratics in s:
//     N(s) = S + (M0 + s M1)(T0 + s T1),      M0 = m(v) (the velocity combination at x), M1 = B dv, (T0, T1) = (t, dt) of the
//     segment, S = +-6 dX;      the proof's condition   sg N(s) > L (1 + 64 eps) T(s)^2 + 64 eps |6 dX|
// is g(s) = g0 + s (g1 + s g2) > 0 with three coefficients per lane: two multiply-adds and a compare per halving instead of the 21
// instructions of infeasible_beyond_doubt.  Which limit: the one most broken at a probe point eight halvings down the ray (none
// broken there: no closed form for this lane, the per-trial proof takes over); sg = the sign of its numerator there.
// Rounding: the six coefficients carry at most 3 eps of their terms each, Horner's rule 2 eps, and the trial point the loop would
// really form (a rounded multiply-add per coordinate) moves N and L T^2 by 2 eps of theirs: all below 8 eps Q with
//     Q = |S| + (|M0| + s |M1|)(|T0| + s |T1|) + L (|T0| + s |T1|)^2 at the first, largest s;
// 32 eps Q is subtracted from g0.  Where g(s) > 0 the per-trial proof's condition holds for the point the loop would form, so
// the evaluation proper would find the limit broken (see infeasible_beyond_doubt): the halving is certain.
template <typename T> struct RayProof {
    T g0, g1, g2;
    bool on;
    __device__ __forceinline__ bool holds(T s) const { return fma_(fma_(g2, s, g1), s, g0) > T(0); }
};
template <typename T, class P>
__device__ __forceinline__ RayProof<T> ray_proof(const P &k, T L, T v, T t0, T t1, T dv, T d0, T d1, T s)
{
    constexpr T eps = sizeof(T) == 8 ? T(2.220446049250313e-16) : T(1.1920929e-7);
    constexpr T margin = T(64) * eps;
    const T lim = L * (T(1) + margin);
    const T six0 = T(6) * k.dx0, six1 = T(6) * k.dx1;
    // the probe: which limit is (most) broken eight halvings down the ray
    const T sp = ldexp_(s, -8);
    const T pv = fma_(dv, sp, v), p0 = fma_(d0, sp, t0), p1 = fma_(d1, sp, t1);
    const T n0 = fma_(seg0_m<T>(k, pv), p0, six0), n1 = fma_(seg0_n<T>(k, pv), p0, -six0);
    const T n2 = fma_(seg1_m<T>(k, pv), p1, six1), n3 = fma_(seg1_n<T>(k, pv), p1, -six1);
    const T b0 = fma_(lim, p0 * p0, margin * abs_(six0)), b1 = fma_(lim, p1 * p1, margin * abs_(six1));
    const T e0 = abs_(n0) - b0, e1 = abs_(n1) - b0, e2 = abs_(n2) - b1, e3 = abs_(n3) - b1;
    const T e01 = max_(e0, e1), e23 = max_(e2, e3);
    const bool seg = e23 > e01;                       // segment 1
    const bool fin = seg ? e3 > e2 : e1 > e0;         // the final end of that segment
    RayProof<T> r;
    r.on = max_(e01, e23) > T(0);
    // that limit's line: N(s) = S + (M0 + s M1)(T0 + s T1)
    const T T0 = seg ? t1 : t0, T1 = seg ? d1 : d0;
    const T six = seg ? six1 : six0;
    const T S = fin ? -six : six;
    const T M0 = seg ? (fin ? seg1_n<T>(k, v) : seg1_m<T>(k, v)) : (fin ? seg0_n<T>(k, v) : seg0_m<T>(k, v));
    const T B = seg ? (fin ? T(2) : T(-4)) : (fin ? T(4) : T(-2));      // d m / d vel1 of seg0_m, seg0_n, seg1_m, seg1_n
    const T M1 = B * dv;
    const T nsel = seg ? (fin ? n3 : n2) : (fin ? n1 : n0);
    const T sg = nsel < T(0) ? T(-1) : T(1);
    const T c0 = fma_(M0, T0, S), c1 = fma_(M0, T1, M1 * T0), c2 = M1 * T1;
    const T lt1 = lim * T1;
    const T q0 = fma_(lim * T0, T0, margin * abs_(six)), q1 = (lim * T0) * (T1 + T1), q2 = lt1 * T1;
    const T ta = fma_(s, abs_(T1), abs_(T0));                            // |T0| + s |T1|
    const T Q = abs_(six) + fma_(fma_(s, abs_(M1), abs_(M0)), ta, (L * ta) * ta);
    r.g0 = fma_(sg, c0, -q0) - (T(32) * eps) * Q;
    r.g1 = fma_(sg, c1, -q1);
    r.g2 = fma_(sg, c2, -q2);
    return r;
}

// surrogateDualityGap (onedpath_ip.cpp:794-808)
template <typename T, int VARIANT, class A = Acc<T>>
__device__ __forceinline__ T duality_gap(const A &e, const T (&lam)[CMap<VARIANT>::NC], T L)
{
    T mu = T(0);
#pragma unroll
    for (int i = 0; i < CMap<VARIANT>::NC; ++i) mu = fma_(-c_value<T, VARIANT, A>(i, e, L), lam[i], mu);
    return mu;
}

// residualNorm (onedpath_ip.cpp:753-792): || (grad f + G^T lam ; lam.c + p) ||^2, evaluated at the
// multipliers lam + s*dl (TRIAL) or lam (not TRIAL; dl and s unused) without materialising them.
template <typename T, int VARIANT, bool TRIAL>
__device__ __forceinline__ T residual_norm(const Acc<T> &e, const T (&lam)[CMap<VARIANT>::NC],
                                           const T (&dl)[CMap<VARIANT>::NC], T s, T p, T L)
{
    // The sum of squares runs in independent partial sums (the constraints' minus and plus halves, then the three
    // gradient terms): a lone wave on a SIMD -- the post-convergence regime of a fixed-step run -- is bound by the
    // latency of a dependent chain, not by issue.  Every caller goes through this one function, so the r(x) a step
    // starts from and the r(x + s d) it is compared with are rounded alike.
    T rv = T(0), rt0 = T(1), rt1 = T(1), accm = T(0), accp = T(0);
    if constexpr (VARIANT == 3) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const T lm = TRIAL ? fma_(dl[2 * j], s, lam[2 * j]) : lam[2 * j];
            const T lp = TRIAL ? fma_(dl[2 * j + 1], s, lam[2 * j + 1]) : lam[2 * j + 1];
            const T d = lp - lm;
            rv = fma_(d, acc_gv(e, j), rv);
            if (j < 2) rt0 = fma_(d, e.gt[j], rt0);
            else       rt1 = fma_(d, e.gt[j], rt1);
            const T rm = fma_(lm, -e.a[j] - L, p);
            const T rp = fma_(lp, e.a[j] - L, p);
            accm = fma_(rm, rm, accm);
            accp = fma_(rp, rp, accp);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const T li = TRIAL ? fma_(dl[i], s, lam[i]) : lam[i];
            T gv, gt;
            c_grad<T, 4>(i, e, gv, gt);
            rv = fma_(li, gv, rv);
            if (i < 2) rt0 = fma_(li, gt, rt0);
            else       rt1 = fma_(li, gt, rt1);
            const T rc = fma_(li, c_value<T, 4>(i, e, L), p);
            if (i & 1) accp = fma_(rc, rc, accp);
            else       accm = fma_(rc, rc, accm);
        }
    }
    const T accx = fma_(rt1, rt1, fma_(rt0, rt0, rv * rv));
    return (accm + accp) + accx;
}

// The same residual in the form the gated kernels carry from step to step.  With t_i = lam_i c_i,
//     |r(p)|^2 = X + S (t_i + p)^2 = X + Q2 + p (2 Q1 + m p),      X = |grad f + G^T lam|^2, Q1 = S t_i, Q2 = S t_i^2,
// and the surrogate gap of the same point is -Q1.  X, Q1 and Q2 do not depend on p: evaluated once at the accepted trial
// point they serve that step's residual test, the next step's gate and perturbation (gap = -Q1, p = gap / (10 m)) AND the
// next step's r(x) under its new p -- the 8-constraint sweeps the reference spends on surrogateDualityGap and on
// residualNorm at x (onedpath_ip.cpp:812, 932) cost a handful of operations.  (Q2 + p (2 Q1 + m p) cancels where the
// point is well centred, t_i ~ -p: the complementarity part is then resolved to ~1e-15 p^2, far below anything the Armijo
// comparison against r(x) can see -- it has to be beaten by a factor (1 - 0.01 s), not by rounding.)
template <typename T, int VARIANT, bool TRIAL>
__device__ __forceinline__ void residual_sums(const Acc<T> &e, const T (&lam)[CMap<VARIANT>::NC], const T (&dl)[CMap<VARIANT>::NC], T s, T L,
                                              T &X, T &Q1, T &Q2, T (&cm_out)[4], T (&cp_out)[4])
{
    // (one chain per sum: the gated kernels keep four waves on a SIMD and are bound by issue, not by the latency of a chain)
    T rv = T(0), rt0 = T(1), rt1 = T(1), q1m = T(0), q1p = T(0), q2m = T(0), q2p = T(0);
    if constexpr (VARIANT == 3) {
        T d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const T lm = TRIAL ? fma_(dl[2 * j], s, lam[2 * j]) : lam[2 * j];
            const T lp = TRIAL ? fma_(dl[2 * j + 1], s, lam[2 * j + 1]) : lam[2 * j + 1];
            d[j] = lp - lm;
            if (j < 2) rt0 = fma_(d[j], e.gt[j], rt0);
            else       rt1 = fma_(d[j], e.gt[j], rt1);
            const T cm = -e.a[j] - L, cp = e.a[j] - L;
            cm_out[j] = cm;
            cp_out[j] = cp;
            const T tm = lm * cm, tp = lp * cp;
            q1m = j == 0 ? tm + tp : (q1m + tm) + tp;
            q2m = j == 0 ? fma_(tp, tp, tm * tm) : fma_(tp, tp, fma_(tm, tm, q2m));
        }
        // S d_j grad_v a_j with the constant coefficients (-2, 4) / t0, (-4, 2) / t1 folded in (acc_gv): half of it, doubled exactly
        const T rvh = fma_(e.r0, fma_(T(2), d[1], -d[0]), e.r1 * fma_(T(-2), d[2], d[3]));
        rv = rvh + rvh;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const T li = TRIAL ? fma_(dl[i], s, lam[i]) : lam[i];
            T gv, gt;
            c_grad<T, 4>(i, e, gv, gt);
            rv = fma_(li, gv, rv);
            if (i < 2) rt0 = fma_(li, gt, rt0);
            else       rt1 = fma_(li, gt, rt1);
            const T t = li * c_value<T, 4>(i, e, L);
            if (i & 1) { q1p += t; q2p = fma_(t, t, q2p); }
            else       { q1m += t; q2m = fma_(t, t, q2m); }
        }
    }
    X = fma_(rt1, rt1, fma_(rt0, rt0, rv * rv));
    if constexpr (VARIANT == 3) { Q1 = q1m; Q2 = q2m; }
    else { Q1 = q1m + q1p; Q2 = q2m + q2p; }
}

template <typename T, int NC>
__device__ __forceinline__ T residual_from_sums(T X, T Q1, T Q2, T p) { return X + fma_(p, fma_(T(NC), p, Q1 + Q1), Q2); }

// 3x3 solve, Gaussian elimination with partial pivoting (row of largest magnitude, first
// wins ties), branch-free, three reciprocals.  A zero pivot means a zero column: the
// reference's rank-revealing QR sets that component of the step to 0
// (ColPivHouseholderQR.h:609-610); srcp_ does the same here.
template <typename T>
__device__ __forceinline__ void swap_if(bool c, T &a, T &b)
{
    const T ta = c ? b : a, tb = c ? a : b;
    a = ta;
    b = tb;
}

template <typename T>
__device__ __forceinline__ void solve3(T a00, T a01, T a02, T a10, T a11, T a12, T a20, T a21, T a22,
                                       T b0, T b1, T b2, T &x0, T &x1, T &x2)
{
    {
        const T m0 = abs_(a00), m1 = abs_(a10), m2 = abs_(a20);
        const bool s1 = (m1 > m0) && !(m2 > m1);      // row 1 is the pivot
        const bool s2 = (m2 > m0) && (m2 > m1);       // row 2 is the pivot
        swap_if(s1, a00, a10); swap_if(s1, a01, a11); swap_if(s1, a02, a12); swap_if(s1, b0, b1);
        swap_if(s2, a00, a20); swap_if(s2, a01, a21); swap_if(s2, a02, a22); swap_if(s2, b0, b2);
    }
    const T i0 = srcp_(a00);
    {
        const T l1 = a10 * i0, l2 = a20 * i0;
        a11 = fma_(-l1, a01, a11); a12 = fma_(-l1, a02, a12); b1 = fma_(-l1, b0, b1);
        a21 = fma_(-l2, a01, a21); a22 = fma_(-l2, a02, a22); b2 = fma_(-l2, b0, b2);
    }
    {
        const bool s = abs_(a21) > abs_(a11);
        swap_if(s, a11, a21); swap_if(s, a12, a22); swap_if(s, b1, b2);
    }
    const T i1 = srcp_(a11);
    {
        const T l2 = a21 * i1;
        a22 = fma_(-l2, a12, a22); b2 = fma_(-l2, b1, b2);
    }
    x2 = b2 * srcp_(a22);
    x1 = fma_(-a12, x2, b1) * i1;
    x0 = fma_(-a02, x2, fma_(-a01, x1, b0)) * i0;
}

// An active constraint converges to c_i = 0, sometimes exactly (the default problem ends at
// a = L bit for bit).  The reference's 11x11 QR is indifferent to a zero on the diagonal; the
// condensation divides by c_i.  So every c_i is shifted by -c_floor = -L*eps/256 before it is
// inverted: about 1/80 of the rounding error c_i = a - L already carries (half an ulp of L), it
// cannot cancel c_i (values of a - L are multiples of ulp(L)/2 >> c_floor), and an exact zero
// becomes the negative number of that size.  One add instead of a compare and two selects.
template <typename T> __device__ __forceinline__ T c_guard(T c, T c_floor) { return c - c_floor; }
// F3 inverts its constraints pairwise through x = cm cp (direction): there the shift rides on the product, x + 2 L c_floor, in
// the multiply-add that forms it.  With cp -> 0 (cm -> -2L) that is cm (cp - c_floor) to first order, i.e. the same guarded
// 1/cp = cm / x; the inactive partner's 1/cm = cp / x inherits a relative change c_floor / |cp|, which matters only below
// |cp| ~ 1e-13 where its multiplier (~ p / 2L) has no influence on K or the right-hand side.

// K dx = rhs for the arrow matrix K = [[a, b, c], [b, d, 0], [c, 0, e]] (K(t0,t1) is structurally
// zero).  d and e are eliminated first when they pass the Bunch-Kaufman 1x1 pivot test
// |pivot| >= alpha * |off-diagonal of its column|, alpha = (1 + sqrt 17)/8 -- bounded growth, the
// symmetric counterpart of partial pivoting -- which they do in all but the early, indefinite
// iterations; otherwise (or when a pivot is zero, or their product leaves the number range) the general
// partial-pivot elimination runs.  The branch is per lane; a wave pays for both paths only while one of its
// lanes is in the indefinite regime.
// HALF: the caller passes the system scaled by D = diag(1/2, 1, 1) (a / 4, b / 2, c / 2, rv / 2: F3's assembly produces
// them in that form) and gets 2 xv back; the pivot test compares against the unscaled off-diagonals, so it decides as it
// would on the unscaled system.
// (Measured and rejected in round 3: Cramer's rule instead of both paths -- 31 instructions, no branch, no compare-and-select
// -- loses eps * w^2 where elimination loses eps * w while ONE constraint dominates K with weight w = lam / c (steps 2-3 of a
// solve): 2.6e-7 off the oracle on 16 of 4,096 problems with non-zero end velocities, profiles/r3_tuning.md.)
template <typename T, bool HALF = false>
__device__ __forceinline__ void solve_arrow(T a, T b, T c, T d, T e, T rv, T r0, T r1, T &xv, T &x0, T &x1)
{
    const T alpha = HALF ? T(2.0 * 0.6403882032022076) : T(0.6403882032022076);
    const T de = d * e;
    // (one flag from four comparisons: written with && the compiler branches on the first and evaluates it twice, once negated)
    const bool bounded = (int)(abs_(d) >= alpha * abs_(b)) & (int)(abs_(e) >= alpha * abs_(c)) & (int)(de != T(0)) & (int)finite_(de);
    if (bounded) {
        const T ide = rcp_(de);                              // one reciprocal for both pivots: 1/d = e/(d e)
        const T id = e * ide, ie = d * ide;
        const T lb = b * id, lc = c * ie;
        const T sc = fma_(-lb, b, fma_(-lc, c, a));          // Schur complement on vel1
        const T rs = fma_(-lb, r0, fma_(-lc, r1, rv));
        xv = rs * srcp_(sc);
        x0 = fma_(-b, xv, r0) * id;
        x1 = fma_(-c, xv, r1) * ie;
    } else {
        T cc = c;
        asm volatile("" : "+v"(cc));      // (opaque: or the compiler forms |c| for this rare branch's selects in front of the test, on everybody's path)
        solve3<T>(a, b, cc, b, d, T(0), cc, T(0), e, rv, r0, r1, xv, x0, x1);
    }
}

// ---- the Newton direction (onedpath_ip.cpp:812-887, condensed) --------------------------
// In: point (v; e = values + grads there), multipliers, perturbation p.  Out: dx, d lam.
// HAVE_C: cm_in / cp_in hold -a_j - L and a_j - L of this very point (gated kernels carry them from the residual sums of the
// accepted trial); otherwise they are formed here.
// sg[i]: a number whose SIGN BIT is set iff the full step would drive multiplier i negative (lam_i + dl_i < 0) -- what the
// fraction-to-boundary rule screens on.  F3 gets it for nothing: lam_i + dl_i = (1/c_i) (lam_i g_i.dx -+ p) and 1/c_i < 0 at a
// feasible point, so the sign is that of the bracket, an intermediate of dl_i.  `suspect` (F3; a word whose sign bit says it): some pair product cm cp is
// not positive, i.e. the point violates a constraint by a rounding (the residual loop accepts points the feasibility loop
// never saw, as the reference's does) and the sign argument does not hold for that pair: the caller then tests every multiplier.
template <typename T, int VARIANT, class P, bool HAVE_C = false>
__device__ __forceinline__ void direction(const P &k, const KParams<T> &kp, T v, const T (&lam)[CMap<VARIANT>::NC],
                                          const Acc<T> &e, T p, T &dxv, T &dx0, T &dx1, T (&dl)[CMap<VARIANT>::NC],
                                          T (&sg)[CMap<VARIANT>::NC], unsigned &suspect,
                                          const T *cm_in = nullptr, const T *cp_in = nullptr, const PointAux<T> *aux = nullptr)
{
    const T L = kp.limit;
    [[maybe_unused]] T htt[4], htv[4];
    if constexpr (VARIANT != 3) accel_hess(k, v, e, htt, htv);      // (F3 folds its second derivatives into the assembly below)
    suspect = 0u;
    [[maybe_unused]] T kvv = T(0), kv0 = T(0), kv1 = T(0);
    T k00 = T(0), k11 = T(0);
    [[maybe_unused]] T bv = T(0);
    T b0 = T(-1), b1 = T(-1);

    if constexpr (VARIANT == 3) {
        // 1/cm_j and 1/cp_j for the four accelerations from ONE reciprocal: with x_j = cm_j cp_j (= L^2 - a_j^2 > 0 inside
        // the feasible set) 1/(x0 x1 x2 x3) gives every 1/x_j by two multiplications, and 1/cm = cp/x, 1/cp = cm/x.  A
        // v_rcp_f64 with its two Newton steps costs what seven multiplications cost.  x_j ranges over [L^2 eps/128, 4 L^2]
        // for feasible points, so the product of four stays far inside the double range.
        T icm[4], icp[4];
        {
            T cm[4], cp[4], x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cm[j] = HAVE_C ? cm_in[j] : -e.a[j] - L;
                cp[j] = HAVE_C ? cp_in[j] : e.a[j] - L;
                x[j] = fma_(cm[j], cp[j], kp.x_floor);      // the c_guard shift, applied to the pair's product (see c_guard)
            }
            suspect = or3_(sign_word(x[0]), sign_word(x[1]), sign_word(x[2])) | sign_word(x[3]);
            const T x01 = x[0] * x[1], x23 = x[2] * x[3];
            const T iall = rcp_(x01 * x23);
            const T i01 = x23 * iall, i23 = x01 * iall;
            const T ix[4] = {x[1] * i01, x[0] * i01, x[3] * i23, x[2] * i23};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                icm[j] = cp[j] * ix[j];
                icp[j] = cm[j] * ix[j];
            }
        }
        // Assembly with the velocity gradients' constant coefficients folded in: d a_j / d vel1 = (-2, 4) / t0, (-4, 2) / t1
        // (acc_gv), so S w_j gv_j^2 = 4 [q0 (w0 + 4 w1) + q1 (4 w2 + w3)] with q = 1/t^2, and likewise for the mixed entries
        // and the right-hand side.  The common factors 4 and 2 are not multiplied out: the system is solved in the scaled
        // unknown 2 dxv, i.e. for K' = D K D, rhs' = D rhs with D = diag(1/2, 1, 1) -- powers of two, so exact.
        T w[4], dlt[4], qq[4], wgt[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const T lm = lam[2 * j], lp = lam[2 * j + 1];
            w[j] = fma_(lm, icm[j], lp * icp[j]);      // lm/cm + lp/cp
            dlt[j] = lp - lm;
            qq[j] = icp[j] - icm[j];                   // x p: rhs weight of grad a_j
            wgt[j] = w[j] * e.gt[j];
        }
        const T r0 = e.r0, r1 = e.r1;
        const T q0 = r0 * r0, q1 = r1 * r1;
        const T kvv_q = fma_(-q1, fma_(T(4), w[2], w[3]), -(q0 * fma_(T(4), w[1], w[0])));                       // kvv / 4
        const T kv0_h = fma_(q0, fma_(T(-2), dlt[1], dlt[0]), r0 * fma_(T(-2), wgt[1], wgt[0]));                  // kv0 / 2
        const T kv1_h = fma_(q1, fma_(T(2), dlt[2], -dlt[3]), r1 * fma_(T(2), wgt[2], -wgt[3]));                  // kv1 / 2
        {
            // S dlt_j d2a_j/dt2 with the common 1/t^3 taken out of each segment's pair (evalAccelSecondDerivInit / Final,
            // onedpath_ip.cpp:404-410, 445-451: (36 dX/t - 8 v0 - 4 v1) / t^3 and (-36 dX/t + 4 v0 + 8 v1) / t^3)
            T u0, u1;
            T m0d, n0d, m1d, n1d;      // 2 m0, 2 n0, 2 m1, 2 n1 of accel_values
            if constexpr (HAVE_C) {      // ... which the carried evaluation of this very point has left behind (same products: same bits)
                u0 = aux->u0; u1 = aux->u1;
                if constexpr (P::zero_vel) { m0d = -aux->n0; n0d = aux->n0 + aux->n0; m1d = -n0d; n1d = aux->n0; }      // -4v, 8v, -8v, 4v
                else { m0d = T(2) * seg0_m<T>(k, v); n0d = T(2) * seg0_n<T>(k, v); m1d = T(2) * seg1_m<T>(k, v); n1d = T(2) * seg1_n<T>(k, v); }      // (four more values to carry: no registers)
            } else {
                u0 = k.dx0 * r0; u1 = k.dx1 * r1;
                if constexpr (P::zero_vel) { m0d = T(-4) * v; n0d = T(8) * v; m1d = -n0d; n1d = -m0d; }
                else { m0d = T(2) * seg0_m<T>(k, v); n0d = T(2) * seg0_n<T>(k, v); m1d = T(2) * seg1_m<T>(k, v); n1d = T(2) * seg1_n<T>(k, v); }
            }
            // (36 is no inline constant: as a literal the multiply-add has to be the two-address form, which overwrites its
            // addend, and each addend is needed twice -- from a scalar register it is the three-address form, no copies)
            T c36 = T(36);
            asm("" : "+s"(c36));
            const T h0 = fma_(dlt[0], fma_(c36, u0, m0d), dlt[1] * fma_(-c36, u0, n0d));
            const T h1 = fma_(dlt[2], fma_(c36, u1, m1d), dlt[3] * fma_(-c36, u1, n1d));
            k00 = fma_(q0 * r0, h0, fma_(-wgt[0], e.gt[0], -(wgt[1] * e.gt[1])));
            k11 = fma_(q1 * r1, h1, fma_(-wgt[2], e.gt[2], -(wgt[3] * e.gt[3])));
        }
        const T bv_h = p * fma_(r0, fma_(T(2), qq[1], -qq[0]), r1 * fma_(T(-2), qq[2], qq[3]));                   // bv / 2
        b0 = fma_(p, fma_(e.gt[0], qq[0], e.gt[1] * qq[1]), T(-1));
        b1 = fma_(p, fma_(e.gt[2], qq[2], e.gt[3] * qq[3]), T(-1));
        T xv2;                                                                                                    // 2 dxv
        solve_arrow<T, true>(kvv_q, kv0_h, kv1_h, k00, k11, bv_h, b0, b1, xv2, dx0, dx1);
        dxv = T(0.5) * xv2;
        const T rx0 = r0 * xv2, rx1 = r1 * xv2;      // grad_v a_j . dxv = (-1, 2) rx0, (-2, 1) rx1
        const T da[4] = {fma_(e.gt[0], dx0, -rx0), fma_(e.gt[1], dx0, rx0 + rx0), fma_(e.gt[2], dx1, -(rx1 + rx1)), fma_(e.gt[3], dx1, rx1)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const T lm = lam[2 * j], lp = lam[2 * j + 1];
            sg[2 * j] = fma_(-lm, da[j], p);
            sg[2 * j + 1] = fma_(lp, da[j], p);
            dl[2 * j] = fma_(-icm[j], sg[2 * j], -lm);                    // -lm + (lm da - p)/cm
            dl[2 * j + 1] = fma_(-icp[j], sg[2 * j + 1], -lp);            // -lp - (lp da + p)/cp
        }
    } else {
        T w[4], pc[4], gv[4], gt[4], icv[4];
        {   // the four 1/c_i from two reciprocals (pairwise: single precision has no range for a product of four)
            T cg[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) cg[i] = c_guard((e.a[i] * e.a[i] - L * L) * T(0.5), kp.c_floor * L);
            const T i01 = rcp_(cg[0] * cg[1]), i23 = rcp_(cg[2] * cg[3]);
            icv[0] = cg[1] * i01; icv[1] = cg[0] * i01; icv[2] = cg[3] * i23; icv[3] = cg[2] * i23;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const T a = e.a[i];
            const T ic = icv[i];
            gv[i] = a * acc_gv(e, i);
            gt[i] = a * e.gt[i];
            w[i] = lam[i] * ic;
            pc[i] = p * ic;
            // Hessian of (a^2 - L^2)/2: (t,t) and (t,v) entries only.  The reference never writes the
            // (vel1X,vel1X) entry (dAdV)^2 (onedpath2_ip.cpp:446-448) -- reproduced: nothing on kvv here.
            const T Htt = fma_(e.gt[i], e.gt[i], a * htt[i]);
            const T Htv = fma_(e.gt[i], acc_gv(e, i), a * htv[i]);
            const T wgv = w[i] * gv[i], wgt = w[i] * gt[i];
            kvv = fma_(-wgv, gv[i], kvv);
            bv = fma_(gv[i], pc[i], bv);
            if (i < 2) {
                kv0 = fma_(lam[i], Htv, fma_(-wgv, gt[i], kv0));
                k00 = fma_(lam[i], Htt, fma_(-wgt, gt[i], k00));
                b0 = fma_(gt[i], pc[i], b0);
            } else {
                kv1 = fma_(lam[i], Htv, fma_(-wgv, gt[i], kv1));
                k11 = fma_(lam[i], Htt, fma_(-wgt, gt[i], k11));
                b1 = fma_(gt[i], pc[i], b1);
            }
        }
        solve_arrow<T>(kvv, kv0, kv1, k00, k11, bv, b0, b1, dxv, dx0, dx1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const T gdx = fma_(gv[i], dxv, gt[i] * (i < 2 ? dx0 : dx1));
            dl[i] = fma_(-w[i], gdx, -(lam[i] + pc[i]));
            sg[i] = lam[i] + dl[i];
        }
    }
}

// ---- the same direction split by its dependence on the perturbation (mu_mode 1) -----------
// The right-hand side of the condensed system is affine in p:  K dx = -grad f + p * S_i g_i / c_i, and so are dx and
// d lam:  d(p) = d_a + p d_c with the "affine-scaling" direction d_a (p = 0, the predictor of Mehrotra's method) and the
// centring direction d_c, both from the ONE matrix K.  newton_step then picks the centring parameter by trial.
template <typename T, int VARIANT, class P>
__device__ __forceinline__ void direction_split(const P &k, const KParams<T> &kp, T v, const T (&lam)[CMap<VARIANT>::NC], const Acc<T> &e,
                                                T (&dxa)[3], T (&dla)[CMap<VARIANT>::NC], T (&dxc)[3], T (&dlc)[CMap<VARIANT>::NC])
{
    constexpr int NC = CMap<VARIANT>::NC;
    const T L = kp.limit;
    T htt[4], htv[4];
    accel_hess(k, v, e, htt, htv);
    T kvv = T(0), kv0 = T(0), kv1 = T(0), k00 = T(0), k11 = T(0);
    T cv = T(0), c0 = T(0), c1 = T(0);                 // b_c = S g_i / c_i
    T ic[NC], wgt_[NC];                                // 1/c_i and lam_i/c_i
    T gvv[NC], gtt[NC];                                // gradients in (vel1, t_seg) coordinates
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const T ci = c_guard(c_value<T, VARIANT>(i, e, L), VARIANT == 3 ? kp.c_floor : kp.c_floor * L);
        ic[i] = rcp_(ci);
        wgt_[i] = lam[i] * ic[i];
        c_grad<T, VARIANT>(i, e, gvv[i], gtt[i]);
        const int seg = c_segment<VARIANT>(i);
        T Htt, Htv;
        if constexpr (VARIANT == 3) {
            const int a = i >> 1;
            Htt = (i & 1) ? htt[a] : -htt[a];
            Htv = (i & 1) ? htv[a] : -htv[a];
        } else {
            Htt = fma_(e.gt[i], e.gt[i], e.a[i] * htt[i]);
            Htv = fma_(e.gt[i], acc_gv(e, i), e.a[i] * htv[i]);
        }
        const T wgv = wgt_[i] * gvv[i], wg = wgt_[i] * gtt[i];
        kvv = fma_(-wgv, gvv[i], kvv);
        cv = fma_(gvv[i], ic[i], cv);
        if (seg == 0) {
            kv0 = fma_(lam[i], Htv, fma_(-wgv, gtt[i], kv0));
            k00 = fma_(lam[i], Htt, fma_(-wg, gtt[i], k00));
            c0 = fma_(gtt[i], ic[i], c0);
        } else {
            kv1 = fma_(lam[i], Htv, fma_(-wgv, gtt[i], kv1));
            k11 = fma_(lam[i], Htt, fma_(-wg, gtt[i], k11));
            c1 = fma_(gtt[i], ic[i], c1);
        }
    }
    solve_arrow<T>(kvv, kv0, kv1, k00, k11, T(0), T(-1), T(-1), dxa[0], dxa[1], dxa[2]);
    solve_arrow<T>(kvv, kv0, kv1, k00, k11, cv, c0, c1, dxc[0], dxc[1], dxc[2]);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int seg = c_segment<VARIANT>(i);
        const T ga = fma_(gvv[i], dxa[0], gtt[i] * (seg == 0 ? dxa[1] : dxa[2]));
        const T gc = fma_(gvv[i], dxc[0], gtt[i] * (seg == 0 ? dxc[1] : dxc[2]));
        dla[i] = fma_(-wgt_[i], ga, -lam[i]);          // -lam_i - (lam_i/c_i) g_i.dx_a
        dlc[i] = fma_(-wgt_[i], gc, -ic[i]);           // -1/c_i  - (lam_i/c_i) g_i.dx_c
    }
}

// ---- fraction to the boundary on the multipliers (onedpath_ip.cpp:903-915): s = min(1, min_{dl_i < 0} -lam_i/dl_i).
// The smallest ratio is found on cross-multiplied pairs (lam_i / -dl_i < nb / -db  <=>  lam_i db > nb dl_i for negative
// dl_i, db; a non-negative dl_i never wins against a negative db, a NaN compares false and is skipped as std::min skips
// it) and divided once, instead of eight divisions and a running minimum.
// Only a multiplier that the full step would drive negative (lam_i + dl_i < 0, i.e. ratio < 1) can bind, which on the
// benchmark distribution happens in steps 1-5 of a solve and for a third of the wave-steps -- and then for two or three
// of the eight multipliers, the same ones in neighbouring problems of the scheduled order.  So the arg-min runs behind two
// screens, both on the sign bits direction() hands over: the lane skips it when no multiplier of its own would go
// negative, and inside it the WAVE skips multiplier i when it would in none of its lanes (one compare and a scalar branch
// instead of two multiplications, a compare and four selects).  A skipped multiplier has ratio >= 1 and could not have
// changed the minimum below 1; a -0 or a NaN with its sign bit set passes the screens and loses the comparison.
// WAVE_SCREEN = false (RP_FROZEN_BF, the fixed-step kernels' build switch): behind the lane screen every multiplier is compared, no
// per-multiplier wave votes -- eight ballots, scalar compares and branches that a lone wave in the post-convergence regime (where the
// directions are rounding noise and sign bits are set at random, so the votes skip little) pays an issue slot each for.
template <typename T, int NC, bool WAVE_SCREEN = true>
__device__ __forceinline__ T boundary_fraction(const KParams<T> &kp, const T (&lam)[NC], const T (&dl)[NC], const T (&sg)[NC], unsigned suspect)
{
    T s = kp.boundary;
    {
        unsigned hi[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) hi[i] = sign_word(sg[i]);
        unsigned bits;
        if constexpr (NC == 8) bits = or3_(or3_(hi[0], hi[1], hi[2]), or3_(hi[3], hi[4], hi[5]), or3_(hi[6], hi[7], suspect));
        else bits = or3_(or3_(hi[0], hi[1], hi[2]), hi[3], suspect);
        if constexpr (!WAVE_SCREEN) {
            if ((int)bits < 0) {
                T nb = T(1), db = T(-1);
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    const bool take = lam[i] * db > nb * dl[i];
                    nb = take ? lam[i] : nb;
                    db = take ? dl[i] : db;
                }
                s = min_(nb * rcp1_(-db), T(1)) * kp.boundary;
            }
        } else
        if ((int)bits < 0) {
            const bool every = __builtin_amdgcn_ballot_w64((int)suspect < 0) != 0ull;      // a pair product not positive somewhere in the wave: no screen
            T nb = T(1), db = T(-1);
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                if (every || __builtin_amdgcn_ballot_w64((int)hi[i] < 0) != 0ull) {
                    const bool take = lam[i] * db > nb * dl[i];
                    nb = take ? lam[i] : nb;
                    db = take ? dl[i] : db;
                }
            }
            s = min_(nb * rcp1_(-db), T(1)) * kp.boundary;
        }
    }
    return s;
}

// Line-search bookkeeping for the diagnostic kernel (rp_batch_step_counted): executions of `s *= 0.5` in the feasibility
// loop (onedpath_ip.cpp:927) and in the residual loop (:944) -- the two numbers the oracle's orc_step_info reports.
struct NoDiag {
    __device__ __forceinline__ void feas() {}
    __device__ __forceinline__ void resid() {}
    __device__ __forceinline__ void moving() {}
};
struct HalvingDiag {
    unsigned nf = 0, nr = 0, nm = 0;      // nm: residual trials that needed a full evaluation (trial point != x); tuning only
    __device__ __forceinline__ void feas() { ++nf; }
    __device__ __forceinline__ void resid() { ++nr; }
    __device__ __forceinline__ void moving() { ++nm; }
};

// The residual at a FIXED point x as a function of the step length (the reference's post-convergence regime: the trial point
// x + s dx has become x bit for bit and only the multipliers lam + s dl still move).  Every component of the residual vector is
// then affine in s, r_i(0) + s r_i': the pieces are formed once per step and a halving is 11 multiply-adds for the components
// plus the sum of squares, in residual_norm's order -- so that once s r_i' drops below half an ulp the value is the value at
// s = 0 bit for bit, which is what ends the reference's loop after ~48 halvings (onedpath_ip.cpp:941).
// lam_at(i): the multipliers the step starts from (registers, or the in-place step's LDS column).
template <typename T, int VARIANT> struct AffineResidual {
    static constexpr int NC = CMap<VARIANT>::NC;
    T rv0, rv1, ra0, ra1, rb0, rb1, c0[NC], c1[NC];
    template <class LamAt>
    __device__ __forceinline__ void setup(const Acc<T> &et, LamAt lam_at, const T (&dl)[NC], T p, T L)
    {
        rv0 = T(0); rv1 = T(0); ra0 = T(1); ra1 = T(0); rb0 = T(1); rb1 = T(0);
        if constexpr (VARIANT == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const T lm = lam_at(2 * q), lp = lam_at(2 * q + 1);
                const T d0 = lp - lm, d1 = dl[2 * q + 1] - dl[2 * q];
                rv0 = fma_(d0, acc_gv(et, q), rv0);
                rv1 = fma_(d1, acc_gv(et, q), rv1);
                if (q < 2) { ra0 = fma_(d0, et.gt[q], ra0); ra1 = fma_(d1, et.gt[q], ra1); }
                else       { rb0 = fma_(d0, et.gt[q], rb0); rb1 = fma_(d1, et.gt[q], rb1); }
                const T cm = -et.a[q] - L, cp = et.a[q] - L;
                c0[2 * q] = fma_(lm, cm, p);
                c1[2 * q] = dl[2 * q] * cm;
                c0[2 * q + 1] = fma_(lp, cp, p);
                c1[2 * q + 1] = dl[2 * q + 1] * cp;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                T gv, gt;
                c_grad<T, 4>(i, et, gv, gt);
                const T li = lam_at(i);
                rv0 = fma_(li, gv, rv0);
                rv1 = fma_(dl[i], gv, rv1);
                if (i < 2) { ra0 = fma_(li, gt, ra0); ra1 = fma_(dl[i], gt, ra1); }
                else       { rb0 = fma_(li, gt, rb0); rb1 = fma_(dl[i], gt, rb1); }
                const T ci = c_value<T, 4>(i, et, L);
                c0[i] = fma_(li, ci, p);
                c1[i] = dl[i] * ci;
            }
        }
    }
    __device__ __forceinline__ T operator()(T sq) const
    {
        T accm = T(0), accp = T(0);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const T ri = fma_(c1[i], sq, c0[i]);
            if (i & 1) accp = fma_(ri, ri, accp);
            else       accm = fma_(ri, ri, accm);
        }
        const T rv = fma_(rv1, sq, rv0), ra = fma_(ra1, sq, ra0), rb = fma_(rb1, sq, rb0);
        return (accm + accp) + fma_(rb, rb, fma_(ra, ra, rv * rv));
    }
    // How many of the next trials (s, s/2, s/4, ...) FAIL the residual test beyond doubt -- counted, not evaluated.  In exact
    // arithmetic the affine model's value is R(s) = R(0) + B s + C s^2 with B = 2 S r_i(0) r_i', C = S r_i'^2 >= 0, and the test
    // "R(s) <= R(0) (1 - armijo s)" fails iff s (B + armijo R(0) + C s) > 0.  The evaluation (operator()) is a sum of squares of
    // singly rounded terms: its relative error is below 10 u (u = eps / 2), the bound's below 14 u, B's absolute error below
    // 13 u Babs (Babs = 2 S |r_i(0) r_i'|), C's relative error below 13 u.  So wherever
    //     h(s) = s (B + armijo R(0)) + C s^2 - 64 eps (R(0) + s Babs + C s^2) > 0
    // -- a margin four times the sum of those errors -- the evaluated test fails as well.  h(0) < 0 and h is convex, so h > 0
    // exactly on the step lengths above its positive root: "h(s 2^-k) > 0" is true up to some k and false from there on, and
    // seven probes of a bisection find the count K in [0, 127]: every counted trial lies between two probes at which h > 0 was
    // EVALUATED.  The reference's loop halves ~50-60 times per post-convergence step before s r' drops below the rounding of r;
    // all but the last few of those halvings are decided here in ~80 instructions instead of ~27 each.  (Backtrack factor 1/2
    // only; a NaN anywhere makes every comparison false: K = 0.)
    __device__ __forceinline__ int certain_failures(const KParams<T> &kp, T r0, T s) const
    {
        constexpr T eps = sizeof(T) == 8 ? T(2.220446049250313e-16) : T(1.1920929e-7);
        constexpr T margin = T(64) * eps;
        T B = rv0 * rv1, Babs = abs_(B), C = rv1 * rv1;
        B = fma_(ra0, ra1, B); Babs = fma_(abs_(ra0), abs_(ra1), Babs); C = fma_(ra1, ra1, C);
        B = fma_(rb0, rb1, B); Babs = fma_(abs_(rb0), abs_(rb1), Babs); C = fma_(rb1, rb1, C);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            B = fma_(c0[i], c1[i], B);
            Babs = fma_(abs_(c0[i]), abs_(c1[i]), Babs);
            C = fma_(c1[i], c1[i], C);
        }
        const T h0 = -margin * r0;
        const T h1 = fma_(kp.armijo, r0, B + B) - margin * (Babs + Babs);
        const T h2 = C * (T(1) - T(2) * margin);
        auto holds = [&](T q) { return fma_(fma_(h2, q, h1), q, h0) > T(0); };
        int last = 0;
#pragma unroll
        for (int b = 64; b >= 1; b >>= 1) last += holds(ldexp_(s, -(last + b))) ? b : 0;
        return holds(s) ? last + 1 : 0;
    }
    // The reference's loop from trial number `it` on (s is that trial's step length, not yet tested): W step lengths per trip
    // (s and its next W - 1 halvings), accepted in the reference's order.  A lone wave on its SIMD is bound by the latency of
    // one evaluation's dependent chain; all W evaluations are complete before the one branch of the trip, so they fill each
    // other's gaps.  r0: the residual the trials have to beat.  Returns whether a trial was accepted.
    template <class D>
    __device__ __forceinline__ bool search(const KParams<T> &kp, T r0, T &s, int &it, D &diag) const
    {
        // Two step lengths per trip (measured at 65,536 problems x 50 steps: one 0.545 ms, two 0.332 ms, four 0.352 ms), written out
        // BRANCH-FREE (round 6): which of the two trials is the first to pass, how many halvings that makes and the next step length are
        // selects on two comparisons -- as a loop over q with `q < valid && ...` the compiler nested two exec-mask regions per trip, eight
        // scalar instructions and two branches that a lone wave pays an issue slot each for (profiles/r6_tuning.md).
#ifdef RP_SEARCH_BRANCHY      // A/B: the round-5 form of the trip
        constexpr int W = 2;
        while (it < kp.max_bt) {
            T sk[W], rn[W];
            sk[0] = s;
            for (int q = 1; q < W; ++q) sk[q] = sk[q - 1] * kp.backtrack;
            for (int q = 0; q < W; ++q) rn[q] = (*this)(sk[q]);
            for (int q = 0; q < W; ++q) asm volatile("" : "+v"(rn[q]));
            const int valid = (kp.max_bt - it < W) ? kp.max_bt - it : W;
            int first = W;
            for (int q = W - 1; q >= 0; --q)
                if (q < valid && rn[q] <= r0 * (T(1) - kp.armijo * sk[q])) first = q;
            const bool got = first < W;
            const int halved = got ? first : valid;
            T snew = sk[W - 1] * kp.backtrack;
            for (int q = W - 1; q >= 0; --q)
                if (halved == q) snew = sk[q];
            s = snew;
            if constexpr (!std::is_same<D, NoDiag>::value)
                for (int q = 0; q < halved; ++q) diag.resid();
            it += halved;
            if (got) return true;
        }
        return false;
#endif
        while (it < kp.max_bt) {
            const T s0 = s, s1 = s * kp.backtrack;
            T r0v = (*this)(s0), r1v = (*this)(s1);
            asm volatile("" : "+v"(r0v), "+v"(r1v));      // (opaque: or the compiler sinks the second evaluation behind the test of the first and serialises them)
            const int left = kp.max_bt - it;              // trials the reference would still make: >= 1 here
            const bool ok0 = r0v <= r0 * (T(1) - kp.armijo * s0);
            const bool ok1 = ((int)(left > 1) & (int)(r1v <= r0 * (T(1) - kp.armijo * s1))) != 0;
            const int none = left < 2 ? left : 2;         // halvings when neither passes
            const int halved = ok0 ? 0 : ok1 ? 1 : none;
            s = ok0 ? s0 : (ok1 | (left < 2)) ? s1 : s1 * kp.backtrack;
            if constexpr (!std::is_same<D, NoDiag>::value)
                for (int q = 0; q < halved; ++q) diag.resid();
            it += halved;
            if (ok0 | ok1) return true;
        }
        return false;
    }
};

#ifndef RP_FEAS_SCREEN
#define RP_FEAS_SCREEN 1      // F4: walk the feasibility loop past trials that are infeasible beyond doubt (0: evaluate every trial, for A/B runs)
#endif
#ifndef RP_FEAS_RAY
#define RP_FEAS_RAY 1         // ... first in closed form along the ray (0: trial by trial only)
#endif

// The feasibility loop's certain halvings (F4): s is walked past every trial that is infeasible beyond doubt -- first along the
// ray in closed form (ray_proof), then trial by trial (infeasible_beyond_doubt) -- and the number of halvings made is returned;
// the caller's loop takes over at the first trial that needs a proper look.  Wave-uniform loops.
template <typename T, int VARIANT, class P, class D>
__device__ __forceinline__ int skip_certain_halvings(const P &k, const KParams<T> &kp, T L, T v, T t0, T t1, T dxv, T dx0, T dx1, T &s, D &diag)
{
    int it_feas = 0;
    // (not in the one instantiation that has no registers for it -- double precision with non-zero end velocities, 167 of the 168
    // three waves allow: the proof changes no decision, so leaving it out there changes no result either)
    if constexpr (VARIANT == 4 && RP_FEAS_SCREEN && (P::zero_vel || sizeof(T) == 4)) {
        // the halvings whose trial misses the limits beyond doubt: the reference evaluates them, finds an error > 0 and halves; this
        // halves.  Wave-uniform loops, every lane to its first trial that needs a proper look: first along the ray in closed form
        // for the one limit that stays broken longest (ray_proof), then trial by trial for all four (infeasible_beyond_doubt).
        if constexpr (RP_FEAS_RAY && !P::lean) {
            const RayProof<T> ray = ray_proof<T, P>(k, L, v, t0, t1, dxv, dx0, dx1, s);
            if (kp.backtrack == T(0.5)) {
                // The number of proven halvings by bisection, no loop over them.  The bisection needs "g(s 2^-k) > 0" to be true up to
                // some k and false from there on.  With g(0) <= 0 (x itself not beyond doubt infeasible) and g(s) > 0 at the first trial
                // that holds for any quadratic: convex, g increases beyond its one positive root; concave, g >= the smaller of two
                // positive values in between.  With g(0) > 0 (x itself beyond doubt infeasible: a nudged or set state, an
                // RP_ST_INFEASIBLE start, a point the residual loop accepted outside) it still holds for a concave or linear g, which is
                // then positive on all of [0, s] -- all 127 probes hold and the budget caps the count, as the reference's loop runs
                // out of halvings -- but NOT for a convex one, which may dip below zero in between (10 - 140 s + 200 s^2): there the
                // closed form is not used and the per-trial proof below, sequential and always sound, walks the trials.
                // Seven probes find the last true k in [0, 127]; every counted halving lies between two EVALUATED positives.
#ifdef RP_RAY_ASSUME_MONOTONE      // the round-3 form, for showing that tests/test_gpu_parity.py::test_f4_steps_from_points_that_are_infeasible_beyond_doubt bites
                const bool monotone = true;
#else
                const bool monotone = !(ray.g0 > T(0)) || !(ray.g2 > T(0));
#endif
                int last = 0;
#pragma unroll
                for (int b = 64; b >= 1; b >>= 1) last += ray.holds(ldexp_(s, -(last + b))) ? b : 0;
                int proven = (ray.on && monotone && ray.holds(s)) ? last + 1 : 0;
                proven = proven < kp.max_bt ? proven : kp.max_bt;
                s = ldexp_(s, -proven);
                it_feas = proven;
                if constexpr (!std::is_same<D, NoDiag>::value)
                    for (int q = 0; q < proven; ++q) diag.feas();
            } else {
                bool more = ray.on;
                while (__builtin_amdgcn_ballot_w64(more) != 0ull) {      // (branch-free body: two multiply-adds, a compare, the selects)
                    more = more && it_feas < kp.max_bt && ray.holds(s);
                    s *= more ? kp.backtrack : T(1);
                    it_feas += more ? 1 : 0;
                    if (more) diag.feas();
                }
            }
        }
        bool more = true;
        do {
            if (more) {
                more = it_feas < kp.max_bt && infeasible_beyond_doubt<T, P>(k, L, fma_(dxv, s, v), fma_(dx0, s, t0), fma_(dx1, s, t1));
                if (more) {
                    s *= kp.backtrack;
                    ++it_feas;
                    diag.feas();
                }
            }
        } while (__builtin_amdgcn_ballot_w64(more) != 0ull);
    }
    return it_feas;
}

// ---- one Newton step -------------------------------------------------------------------
// In:  x = (v, t0, t1), lam, c = reciprocals + accelerations at x, gap = surrogate duality gap at x.
// Out: the same at the new point (the caller recomputes the gap from c).
//
// MEMO: exact memoisation of evaluations at trial points that are bitwise the current point.  It changes no result
// (the reference recomputes the same numbers); what it buys is the reference's post-convergence regime, where x no
// longer moves and every step still walks ~48 residual halvings (onedpath_ip.cpp:932-945) -- fixed-step runs.  A gated
// solve stops long before that regime, so its kernels are built without the checks.
// AFFINE (with MEMO): the post-convergence loop runs on the affine pieces of the residual (44 more
// registers: the one-problem-per-lane streaming kernels, which run the small fixed-step batches, have them; the tiled
// kernels at 168 VGPRs do not).
//
// MU (rp_params.mu_mode): 0 = the reference's fixed centring, p = gap / (m * mu_divisor) (onedpath_ip.cpp:812) -- the only
// mode the parity tests are about.  1 = centring by trial: with the split direction d(p) = d_a + p d_c, the candidates
// sigma = kp.sigma_try[0] < sigma_try[1] (p = sigma * gap / m) are tried in turn and the first one is taken whose FULL
// step (s = boundary fraction, i.e. no multiplier would leave the positive orthant, the point is primal feasible and the
// residual of its own p passes the reference's Armijo test) succeeds; otherwise the reference step is taken with the
// reference's line search.  Fewer steps to the same optimum (measured: 15.4 -> 12.7 mean on the benchmark distribution);
// each step costs more, and results are NOT the reference's iterates -- opt-in, off by default.
// WAVE (ungated kernels whose lanes all take the same number of steps: every live lane of the wave is inside this function at
// the same time; the value is the instantiating kernel's block size and must be 64): the residual loop's stragglers are served by
// the whole wave, see "wave-parallel line search" below.
template <typename T, int VARIANT, class P, bool MEMO = true, bool AFFINE = false, int MU = 0, class D = NoDiag, int WAVE = 0>
__device__ __forceinline__ void newton_step_to(const P &k, const KParams<T> &kp, T gap,
                                               const T v, const T t0, const T t1, const T (&lam)[CMap<VARIANT>::NC],
                                               const AccCarry<T, !MEMO, !MEMO && MU == 0> &c,
                                               T &nv, T &nt0, T &nt1, T (&nlam)[CMap<VARIANT>::NC], AccCarry<T, !MEMO, !MEMO && MU == 0> &nc,
                                               D &diag)
{
    constexpr int NC = CMap<VARIANT>::NC;
    constexpr bool SUMS = !MEMO && MU == 0;      // the residual in its carried form (residual_sums)
    const T L = kp.limit;
    const T p = gap * kp.inv_mu_den;                      // onedpath_ip.cpp:812

    T dxv, dx0, dx1, dl[NC], r0n;
    T sg[NC];                      // sign bit set: the full step would take multiplier i below zero (see direction)
    unsigned suspect = 0u;
    bool feasible_here = true;
    {
        Acc<T> e;
        e.r0 = c.r0; e.r1 = c.r1;
#pragma unroll
        for (int j = 0; j < 4; ++j) e.a[j] = c.a[j];
        if constexpr (MEMO) {
            accel_grads(k, v, e);
            feasible_here = all_satisfied<T, VARIANT>(e, L);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) e.gt[j] = c.gt[j];
        }
        if constexpr (MU == 0) {
#ifndef RP_NO_CARRIED_C      // tuning knob of newton_step_to (A/B builds with RP_GATED_IN_PLACE=0): 18 more VGPRs for 8 fewer instructions per step
            if constexpr (SUMS && VARIANT == 3) direction<T, VARIANT, P, true>(k, kp, v, lam, e, p, dxv, dx0, dx1, dl, sg, suspect, c.cm, c.cp, &c.x);
#else
            if constexpr (false) {}
#endif
            else direction<T, VARIANT, P>(k, kp, v, lam, e, p, dxv, dx0, dx1, dl, sg, suspect);
            if constexpr (SUMS) r0n = residual_from_sums<T, NC>(c.X, c.Q1, c.Q2, p);
            else r0n = residual_norm<T, VARIANT, false>(e, lam, dl, T(0), p, L);      // onedpath_ip.cpp:932
        } else {
            T dxa[3], dla[NC], dxc[3], dlc[NC];
            direction_split<T, VARIANT, P>(k, kp, v, lam, e, dxa, dla, dxc, dlc);
            const T mu = gap * (T(1) / T(NC));
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const T pq = kp.sigma_try[q] * mu;
                T tl[NC], tmin = T(0);
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    tl[i] = fma_(pq, dlc[i], dla[i]);
                    tmin = min_(tmin, lam[i] + tl[i]);
                }
                const T sq = kp.boundary;
                const T qv = fma_(fma_(pq, dxc[0], dxa[0]), sq, v), q0 = fma_(fma_(pq, dxc[1], dxa[1]), sq, t0), q1 = fma_(fma_(pq, dxc[2], dxa[2]), sq, t1);
                Acc<T> eq;
                accel_values(k, qv, q0, q1, eq);
                bool good = !(tmin < T(0)) && all_satisfied<T, VARIANT>(eq, L);
                if (good) {
                    accel_grads(k, qv, eq);
                    const T rq = residual_norm<T, VARIANT, true>(eq, lam, tl, sq, pq, L);
                    const T rx = residual_norm<T, VARIANT, false>(e, lam, tl, T(0), pq, L);
                    good = rq <= rx * (T(1) - kp.armijo * sq);
                }
                if (good) {                 // take it: the full step of the smaller centring parameter
                    nv = qv; nt0 = q0; nt1 = q1;
#pragma unroll
                    for (int i = 0; i < NC; ++i) nlam[i] = fma_(tl[i], sq, lam[i]);
                    nc.r0 = eq.r0; nc.r1 = eq.r1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) nc.a[j] = eq.a[j];
                    if constexpr (!MEMO) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) nc.gt[j] = eq.gt[j];
                    }
                    return;
                }
            }
            // the reference's centring with the reference's line search, on the split direction
            dxv = fma_(p, dxc[0], dxa[0]);
            dx0 = fma_(p, dxc[1], dxa[1]);
            dx1 = fma_(p, dxc[2], dxa[2]);
#pragma unroll
            for (int i = 0; i < NC; ++i) { dl[i] = fma_(p, dlc[i], dla[i]); sg[i] = lam[i] + dl[i]; }
            r0n = residual_norm<T, VARIANT, false>(e, lam, dl, T(0), p, L);
        }
    }

    T s = boundary_fraction<T, NC>(kp, lam, dl, sg, suspect);      // onedpath_ip.cpp:903-915

    // -- backtrack until primal feasible (onedpath_ip.cpp:919-928) --
    Acc<T> et;                 // evaluation at the current trial point
    T tv = v, tt0 = t0, tt1 = t1;
    bool et_valid = false;     // et holds the accelerations of (tv,tt0,tt1) == x + s*dx
    const int it_feas = skip_certain_halvings<T, VARIANT, P, D>(k, kp, L, v, t0, t1, dxv, dx0, dx1, s, diag);
    for (int it = it_feas; it < kp.max_bt; ++it) {
        tv = fma_(dxv, s, v);
        tt0 = fma_(dx0, s, t0);
        tt1 = fma_(dx1, s, t1);
        bool ok;
        if (MEMO && tv == v && tt0 == t0 && tt1 == t1) {
            ok = feasible_here;            // same point, same answer
            et_valid = false;
            if (!ok) {
                // x itself fails the test by a rounding (|a| = L + 1 ulp: the residual loop of the previous step accepts
                // points the feasibility loop never saw, as the reference's does) and every smaller s gives x again: the
                // reference walks all its remaining halvings to the same verdict.  So does this, without the evaluations.
                if (kp.backtrack == T(0.5) && std::is_same<D, NoDiag>::value) {
                    s = ldexp_(s, it - kp.max_bt);      // max_bt - it exact halvings at once
                } else {
                    for (; it < kp.max_bt; ++it) {
                        s *= kp.backtrack;
                        diag.feas();
                    }
                }
                break;
            }
        } else {
            accel_values(k, tv, tt0, tt1, et);
            ok = all_satisfied<T, VARIANT>(et, L);
            et_valid = true;
        }
        if (ok) break;
        s *= kp.backtrack;
        diag.feas();
        et_valid = false;
    }

    // -- backtrack until the residual decreases (onedpath_ip.cpp:932-945) --
    bool accepted = false;         // et = values at the point the loop broke on
    T tl[SUMS ? NC : 1];           // gated kernels: the trial multipliers lam + s dl of the last evaluated trial
    int it = 0;
    bool frozen = false;           // the trial point has become bitwise x (and stays so: s only shrinks)
    if constexpr (WAVE != 0 && !SUMS) {
        // ---- wave-parallel line search ----
        // WAVE = the THREADS PER BLOCK of the kernel that instantiates this form (0: the serial search).  The service below broadcasts
        // through one LDS area per block with no barrier, which is correct only in a single-wave block (a wave's LDS operations
        // complete in order): any other launch shape fails to compile here.
        static_assert(WAVE == 64, "newton_step_to<WAVE> broadcasts through LDS without a barrier: single-wave blocks (64 threads) only");
        // F4 never converges (README.md:34): from step ~6 a few per cent of the problems walk ~50 residual halvings at every
        // step, each with a full evaluation (the trial point still moves), while the other lanes of their wave have long
        // accepted -- per 64-lane wave 847 residual trials over 50 steps where a lane needs 21
        // (profiles/r2_f4_halving_probe.log).  Here the loop is wave-uniform, and once only a few lanes are still searching
        // each of them is served by the WHOLE wave: its search state is broadcast and the live lane of rank q evaluates
        // the trial the serial loop would make q halvings later (s 2^-q, exact for backtrack = 1/2) with the very same
        // functions; a ballot gives the first trial at which the serial loop would stop halving moving points -- accepted,
        // or the trial point has become x (then the frozen-regime loop below takes over, as in the serial form), or the
        // halvings are used up.  The straggler jumps there (s, it) and re-evaluates that one trial in the common loop, so
        // everything downstream is the serial code and every decision and every bit is the serial loop's
        // (onedpath2_ip.cpp:791-833).
#ifndef RP_WAVE_SERIAL_FIRST
#define RP_WAVE_SERIAL_FIRST 1
#endif
#ifndef RP_WAVE_SERVE_AT_MOST
#define RP_WAVE_SERVE_AT_MOST 16
#endif
        constexpr int kSerialFirst = RP_WAVE_SERIAL_FIRST, kServeAtMost = RP_WAVE_SERVE_AT_MOST;      // trials in lock step before anyone is served; stragglers worth serving one by one
        const unsigned long long alive = __ballot(true);
        const int lane_id = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const int rank = __popcll(alive & ((1ull << lane_id) - 1ull)), nalive = __popcll(alive);
        bool open = true;
        for (int round = 0;; ++round) {
            const unsigned long long om = __ballot(open);
            if (om == 0ull) break;
            if (round >= kSerialFirst && kp.backtrack == T(0.5) && __popcll(om) <= kServeAtMost) {
                unsigned long long todo = om;
                while (todo) {
                    const int src = __ffsll((long long)todo) - 1;
                    todo &= todo - 1ull;
                    P bk = k;
                    T bv, bt0, bt1, bdv, bd0, bd1, bs, bp, br0n, blam[NC], bdl[NC];
                    const int bit = bcast_(it, src);
#ifndef RP_WAVE_BCAST_LDS
#define RP_WAVE_BCAST_LDS 1      // the straggler's search state reaches the other lanes through LDS (0: 2 x 21 v_readlane, for A/B runs)
#endif
                    if constexpr (RP_WAVE_BCAST_LDS != 0) {
                        // One lane writes its 13 + 2 NC values, every lane reads them back (a broadcast read: one address, no bank
                        // conflict): LDS instructions, which issue beside the other waves' arithmetic, where 2 x 21 v_readlane and the
                        // moves that bring their scalar results back into vector registers were a third of the service's vector
                        // instructions.  The kernel's blocks are single waves (k_steps_chunks: kChunkBlock, with a static_assert at the one
                        // place this WAVE form is instantiated), and a wave's LDS operations complete in order: no barrier.  Any other launch
                        // shape (the 256-thread streaming kernels share run_lane) must not instantiate WAVE: four waves would race on s_bc.
                        __shared__ T s_bc[13 + 2 * NC];
                        LdsBackup<T> bc = (LdsBackup<T>)s_bc;
                        if (lane_id == src) {
                            bc[0] = k.dx0; bc[1] = k.dx1;
                            if constexpr (!P::zero_vel) { bc[2] = k.v0; bc[3] = k.v2; }
                            bc[4] = v; bc[5] = t0; bc[6] = t1; bc[7] = dxv; bc[8] = dx0; bc[9] = dx1; bc[10] = s; bc[11] = p; bc[12] = r0n;
#pragma unroll
                            for (int i = 0; i < NC; ++i) { bc[13 + i] = lam[i]; bc[13 + NC + i] = dl[i]; }
                        }
                        bk.dx0 = bc[0]; bk.dx1 = bc[1];
                        if constexpr (!P::zero_vel) { bk.v0 = bc[2]; bk.v2 = bc[3]; }
                        bv = bc[4]; bt0 = bc[5]; bt1 = bc[6]; bdv = bc[7]; bd0 = bc[8]; bd1 = bc[9]; bs = bc[10]; bp = bc[11]; br0n = bc[12];
#pragma unroll
                        for (int i = 0; i < NC; ++i) { blam[i] = bc[13 + i]; bdl[i] = bc[13 + NC + i]; }
                    } else {
                        bk.dx0 = bcast_(k.dx0, src);
                        bk.dx1 = bcast_(k.dx1, src);
                        if constexpr (!P::zero_vel) { bk.v0 = bcast_(k.v0, src); bk.v2 = bcast_(k.v2, src); }
                        bv = bcast_(v, src); bt0 = bcast_(t0, src); bt1 = bcast_(t1, src);
                        bdv = bcast_(dxv, src); bd0 = bcast_(dx0, src); bd1 = bcast_(dx1, src);
                        bs = bcast_(s, src); bp = bcast_(p, src); br0n = bcast_(r0n, src);
#pragma unroll
                        for (int i = 0; i < NC; ++i) { blam[i] = bcast_(lam[i], src); bdl[i] = bcast_(dl[i], src); }
                    }
                    const T sq = ldexp_(bs, -rank);                    // the step length of the serial loop's trial number bit + rank
                    const bool beyond = bit + rank >= kp.max_bt;       // ... which it would not make
                    const T qv = fma_(bdv, sq, bv), q0 = fma_(bd0, sq, bt0), q1 = fma_(bd1, sq, bt1);
                    const bool still = (qv == bv && q0 == bt0 && q1 == bt1);
                    Acc<T> eq;
                    accel_values(bk, qv, q0, q1, eq);
                    accel_grads(bk, qv, eq);
                    const T rq = residual_norm<T, VARIANT, true>(eq, blam, bdl, sq, bp, L);
                    const bool pass = rq <= br0n * (T(1) - kp.armijo * sq);
                    const unsigned long long stop = __ballot(beyond || still || pass);
                    // live lanes are ranked by lane number, so the lowest stopping lane holds the earliest stopping trial
                    const int adv = stop ? __popcll(alive & ((1ull << (__ffsll((long long)stop) - 1)) - 1ull)) : nalive;
                    if (lane_id == src) {
                        s = ldexp_(s, -adv);
                        it += adv;
                        if constexpr (!std::is_same<D, NoDiag>::value)
                            for (int q = 0; q < adv; ++q) diag.resid();
                        et_valid = false;
                    }
                }
            }
            if (open) {      // one trial in lock step: the serial loop's body
                if (!(it < kp.max_bt)) {
                    open = false;                                  // halvings used up: the last s is never evaluated
                } else {
                    if (!et_valid) {
                        tv = fma_(dxv, s, v);
                        tt0 = fma_(dx0, s, t0);
                        tt1 = fma_(dx1, s, t1);
                    }
                    if (MEMO && tv == v && tt0 == t0 && tt1 == t1) {
                        frozen = true;
                        open = false;
                    } else {
                        if (!et_valid) accel_values(k, tv, tt0, tt1, et);
                        accel_grads(k, tv, et);
                        diag.moving();
                        const T rn = residual_norm<T, VARIANT, true>(et, lam, dl, s, p, L);
                        et_valid = false;
                        if (rn <= r0n * (T(1) - kp.armijo * s)) {
                            accepted = true;
                            open = false;
                        } else {
                            s *= kp.backtrack;
                            diag.resid();
                            ++it;
                        }
                    }
                }
            }
        }
    } else if constexpr (!MEMO) {
        // Kernels without memoisation (the gated ones): the trial point and its accelerations are formed where s changes --
        // by the feasibility loop, or after a failed residual trial -- so the common case, first trial accepted, runs straight
        // through with nothing to select.  (The feasibility loop leaves no evaluation only when it ran out of halvings.)
        if (!et_valid && it < kp.max_bt) {
            tv = fma_(dxv, s, v);
            tt0 = fma_(dx0, s, t0);
            tt1 = fma_(dx1, s, t1);
            accel_values(k, tv, tt0, tt1, et);
        }
        while (it < kp.max_bt) {
            accel_grads(k, tv, et);
            diag.moving();
            T rn;
            if constexpr (SUMS) {
#pragma unroll
                for (int i = 0; i < NC; ++i) tl[i] = fma_(dl[i], s, lam[i]);              // kept: the accepted trial IS the update
                residual_sums<T, VARIANT, false>(et, tl, dl, T(0), L, nc.X, nc.Q1, nc.Q2, nc.cm, nc.cp);   // overwritten by every trial: the accepted one stays
                rn = residual_from_sums<T, NC>(nc.X, nc.Q1, nc.Q2, p);
            } else {
                rn = residual_norm<T, VARIANT, true>(et, lam, dl, s, p, L);
            }
            if (rn <= r0n * (T(1) - kp.armijo * s)) {
                accepted = true;
                break;
            }
            s *= kp.backtrack;
            diag.resid();
            ++it;
            if (it < kp.max_bt) {
                tv = fma_(dxv, s, v);
                tt0 = fma_(dx0, s, t0);
                tt1 = fma_(dx1, s, t1);
                accel_values(k, tv, tt0, tt1, et);
            }
        }
    } else
    for (; it < kp.max_bt; ++it) {
        if (!et_valid) {          // (et_valid: the feasibility loop ended on this very point with this very s -- nothing to redo)
            tv = fma_(dxv, s, v);
            tt0 = fma_(dx0, s, t0);
            tt1 = fma_(dx1, s, t1);
        }
        if (MEMO && tv == v && tt0 == t0 && tt1 == t1) { frozen = true; break; }
        if (!et_valid) accel_values(k, tv, tt0, tt1, et);
        accel_grads(k, tv, et);
        diag.moving();
        T rn;
        if constexpr (SUMS) {
#pragma unroll
            for (int i = 0; i < NC; ++i) tl[i] = fma_(dl[i], s, lam[i]);              // kept: the accepted trial IS the update
            residual_sums<T, VARIANT, false>(et, tl, dl, T(0), L, nc.X, nc.Q1, nc.Q2, nc.cm, nc.cp);   // overwritten by every trial: the accepted one stays
            rn = residual_from_sums<T, NC>(nc.X, nc.Q1, nc.Q2, p);
        } else {
            rn = residual_norm<T, VARIANT, true>(et, lam, dl, s, p, L);
        }
        et_valid = false;
        if (rn <= r0n * (T(1) - kp.armijo * s)) {
            accepted = true;
            break;
        }
        s *= kp.backtrack;
        diag.resid();
    }
    if constexpr (MEMO) {
        if (frozen) {
            // x + s dx == x from here on: the evaluation at x (bit for bit what the step started from) is loop-invariant,
            // only the multipliers lam + s dl still depend on s.  What is left per halving is the sum of squares itself.
            et.r0 = c.r0; et.r1 = c.r1;
#pragma unroll
            for (int j = 0; j < 4; ++j) et.a[j] = c.a[j];
            accel_grads(k, v, et);
            if constexpr (!AFFINE) {      // the kernels without the registers for the affine pieces: the direct evaluation
                for (; it < kp.max_bt; ++it) {
                    const T rn = residual_norm<T, VARIANT, true>(et, lam, dl, s, p, L);
                    if (rn <= r0n * (T(1) - kp.armijo * s)) {
                        accepted = true;
                        break;
                    }
                    s *= kp.backtrack;
                    diag.resid();
                }
            } else {
            // With x fixed the residual vector is AFFINE in s (AffineResidual above): pieces once per step, 27 instructions per halving
            AffineResidual<T, VARIANT> ar;
            ar.setup(et, [&](int i) { return lam[i]; }, dl, p, L);
            accepted = ar.search(kp, r0n, s, it, diag);
            }
        }
    }

    // -- take the step (onedpath_ip.cpp:949-952) --
    if constexpr (SUMS) {
        // Gated kernels: the accepted trial point and multipliers ARE the update, bit for bit.  A loop that ran out of
        // halvings (its last s was never evaluated) forms that last trial here, into the same registers, so that one set
        // of values leaves the step whichever way it ended.
        if (!accepted) {
            tv = fma_(dxv, s, v);
            tt0 = fma_(dx0, s, t0);
            tt1 = fma_(dx1, s, t1);
#pragma unroll
            for (int i = 0; i < NC; ++i) tl[i] = fma_(dl[i], s, lam[i]);
            accel_values(k, tv, tt0, tt1, et);
            accel_grads(k, tv, et);
            residual_sums<T, VARIANT, false>(et, tl, dl, T(0), L, nc.X, nc.Q1, nc.Q2, nc.cm, nc.cp);
        }
        nv = tv; nt0 = tt0; nt1 = tt1;
#pragma unroll
        for (int i = 0; i < NC; ++i) nlam[i] = tl[SUMS ? i : 0];
    } else {
        nv = fma_(dxv, s, v);
        nt0 = fma_(dx0, s, t0);
        nt1 = fma_(dx1, s, t1);
#pragma unroll
        for (int i = 0; i < NC; ++i) nlam[i] = fma_(dl[i], s, lam[i]);
        if (!accepted) {                               // the loop ran out of halvings: its last s was never evaluated
            accel_values(k, nv, nt0, nt1, et);
            if constexpr (!MEMO) accel_grads(k, nv, et);
        }
    }
    nc.r0 = et.r0; nc.r1 = et.r1;
#pragma unroll
    for (int j = 0; j < 4; ++j) nc.a[j] = et.a[j];
    if constexpr (!MEMO) {
#pragma unroll
        for (int j = 0; j < 4; ++j) nc.gt[j] = et.gt[j];
    }
}

// the same in place
template <typename T, int VARIANT, class P, bool MEMO = true, bool AFFINE = false, int MU = 0, class D = NoDiag, int WAVE = 0>
__device__ __forceinline__ void newton_step(const P &k, const KParams<T> &kp, T gap,
                                            T &v, T &t0, T &t1, T (&lam)[CMap<VARIANT>::NC], AccCarry<T, !MEMO, !MEMO && MU == 0> &c,
                                            D &diag)
{
    constexpr int NC = CMap<VARIANT>::NC;
    T nv, nt0, nt1, nlam[NC];
    AccCarry<T, !MEMO, !MEMO && MU == 0> nc;
    newton_step_to<T, VARIANT, P, MEMO, AFFINE, MU, D, WAVE>(k, kp, gap, v, t0, t1, lam, c, nv, nt0, nt1, nlam, nc, diag);
    v = nv; t0 = nt0; t1 = nt1;
#pragma unroll
    for (int i = 0; i < NC; ++i) lam[i] = nlam[i];
    c = nc;
}

// ---- the same step IN PLACE, for the gated solve ----------------------------------------------------------------------
// newton_step_to keeps the point and multipliers a step starts from next to the trial it is evaluating (a rejected trial is
// followed by another from the same start), and its loops are left by every lane when that lane is done -- for which the
// compiler copies every value that is live after such a loop once per trip, for the lanes that have left: 11 + 9 64-bit moves
// per step on a kernel that is bound by vector-instruction issue (and, back to back, by the energy of a step).  Here the trial
// overwrites the state (a multiply-add onto itself), the start of the step waits in LDS instead -- 11 ds_write per step, which
// issue on the LDS port beside the other waves' arithmetic -- to be read back only when a trial is rejected (one step in five
// has a rejected feasibility trial, a rejected residual trial is rare before convergence), and the loops run while ANY lane of
// the wave is inside (wave-uniform exits: nothing to copy).  Same functions, same operands, same order of decisions as
// newton_step_to<MEMO = false, MU = 0>: every bit of every iterate is the same (tests/checks/inplace_ab.py, 36 cases).
// bk: where the step's start waits (LdsColumn: this lane's column of the block's backup area, field q at p[q * 64]; volatile so that the compiler neither forwards the
// stored values to the reads (keeping them in registers is what this form is there to avoid) nor drops the stores.
// The loops are written with the trial formed where s is set (at the bottom, from the backed-up start), so that a trial
// is a multiply-add INTO the state registers.
// Where the step's start waits: this lane's column of the block's LDS area (field q at p[q * 64]) -- the form the large-batch
// kernels are built around, four waves per SIMD -- or, for batches too small to fill the chip (a lone wave per SIMD has nothing
// to run under an LDS round trip: BASELINE configs[1]), plain registers at three.
template <typename T> struct LdsColumn {
    LdsBackup<T> p;
    __device__ __forceinline__ T get(int q) const { return p[q * 64]; }
    __device__ __forceinline__ void put(int q, T x) const { p[q * 64] = x; }
};
template <typename T, int N> struct RegColumn {
    T r[N];
    __device__ __forceinline__ T get(int q) const { return r[q]; }
    __device__ __forceinline__ void put(int q, T x) { r[q] = x; }
};

// FROZEN (the fixed-step kernels: a gated solve stops long before): the reference's post-convergence regime, where x no longer
// moves and every step still walks ~48 residual halvings (onedpath_ip.cpp:932-945), gets its own search -- see "frozen" below.
// D: line-search bookkeeping (rp_batch_step_counted).
//
// Loop shape: a loop's exit test is a ballot over a COMPARISON made in the same block ("some lane's trial failed"), not over a
// flag carried round the loop -- the flag form costs a select and a compare per test to turn the carried bit back into a lane
// mask; a lane whose trial has passed simply re-evaluates its unchanged trial (same values, the instructions issue for the wave
// anyway) while others retry.  The halving counter `it` is touched only where a trial fails: it enters every loop as zero and is
// put back to zero behind a wave-uniform branch when some lane used it.
#ifndef RP_USED_INT
#define RP_USED_INT 1      // the wave-uniform "some lane moved its halving counter" flag of the in-place step's loops as an int (0: a bool, A/B)
#endif
#if RP_USED_INT
typedef int used_t;
#else
typedef bool used_t;
#endif
template <typename T, int VARIANT, class P, bool FROZEN = false, class D = NoDiag, class BK = LdsColumn<T>>
__device__ __forceinline__ void newton_step_inplace(const P &k, const KParams<T> &kp, T gap, T &v, T &t0, T &t1, T (&lam)[CMap<VARIANT>::NC],
                                                    AccCarry<T, true, true> &c, BK &bk, int &it, D &diag)
{
    constexpr int NC = CMap<VARIANT>::NC;
    const T L = kp.limit;
    const T p = gap * kp.inv_mu_den;                      // onedpath_ip.cpp:812

    T dxv, dx0, dx1, dl[NC], sg[NC];
    unsigned suspect = 0u;
    {
        Acc<T> e;
        e.r0 = c.r0; e.r1 = c.r1;
#pragma unroll
        for (int j = 0; j < 4; ++j) { e.a[j] = c.a[j]; e.gt[j] = c.gt[j]; }
        if constexpr (VARIANT == 3) direction<T, VARIANT, P, true>(k, kp, v, lam, e, p, dxv, dx0, dx1, dl, sg, suspect, c.cm, c.cp, &c.x);
        else direction<T, VARIANT, P>(k, kp, v, lam, e, p, dxv, dx0, dx1, dl, sg, suspect);
    }
    const T r0n = residual_from_sums<T, NC>(c.X, c.Q1, c.Q2, p);      // onedpath_ip.cpp:932
    bk.put(0, v); bk.put(1, t0); bk.put(2, t1);
#pragma unroll
    for (int i = 0; i < NC; ++i) bk.put(3 + i, lam[i]);

#ifndef RP_FROZEN_BF
#define RP_FROZEN_BF 1      // the fixed-step kernels (FROZEN) take the boundary fraction without the per-multiplier wave votes (0: with them, A/B): 0.194 -> 0.183 ms
                            // at 65,536 x 50 and 0.0203 -> 0.0185 ms at 65,536 x 12 -- a lone wave pays an issue slot for every vote, compare and branch
#endif
    T s = boundary_fraction<T, NC, !(FROZEN && RP_FROZEN_BF != 0)>(kp, lam, dl, sg, suspect);         // onedpath_ip.cpp:903-915

    // -- backtrack until primal feasible (onedpath_ip.cpp:919-928); a trial is formed where s is set --
    Acc<T> et;
    PointAux<T> xt;      // dX / t and the velocity combinations of the trial point et belongs to
    if constexpr (VARIANT == 4) it = skip_certain_halvings<T, VARIANT, P, D>(k, kp, L, v, t0, t1, dxv, dx0, dx1, s, diag);      // (F4: its certain halvings)
    v = fma_(dxv, s, v);
    t0 = fma_(dx0, s, t0);
    t1 = fma_(dx1, s, t1);
    if constexpr (FROZEN && RP_FROZEN_MASKS != 0) {
        // The same loop with its per-lane flags (the trial has become x: at_x) as LANE MASKS in scalar registers: a flag carried round a
        // loop as a bool comes back as a select and a compare every time it is tested or merged; a mask is combined by scalar and / or
        // and becomes a predicate again at no cost (in_mask_).  Same decisions per lane.
        used_t used = VARIANT == 4;
        unsigned long long atx_m = 0ull;
        for (;;) {
            accel_values_u(k, v, t0, t1, et, xt);
            const unsigned long long bad_m = ballot_(!all_satisfied<T, VARIANT>(et, L));
            if (bad_m == 0ull) break;
            const unsigned long long go_m = bad_m & ballot_(it < kp.max_bt);      // (out of halvings: the reference goes on with an s it has not tested)
            if (go_m == 0ull) break;
            used = 1;
            const unsigned long long jump_m = go_m & atx_m, step_m = go_m & ~atx_m;
            if (jump_m != 0ull) {
                if (in_mask_(jump_m)) {      // x itself fails the test by a rounding and every smaller s gives x again: the remaining halvings at once
                    if (kp.backtrack == T(0.5) && std::is_same<D, NoDiag>::value) {
                        s = ldexp_(s, it - kp.max_bt);
                    } else {
                        for (int q = it; q < kp.max_bt; ++q) { s *= kp.backtrack; diag.feas(); }
                    }
                    it = kp.max_bt;
                }
            }
            if (step_m != 0ull) {
                if (in_mask_(step_m)) {
                    s *= kp.backtrack;
                    ++it;
                    diag.feas();
                    v = fma_(dxv, s, (T)bk.get(0));
                    t0 = fma_(dx0, s, (T)bk.get(1));
                    t1 = fma_(dx1, s, (T)bk.get(2));
                }
                // (one ballot per comparison, the masks combined as integers: a ballot of `a && b && c` is lowered through a select and a
                // compare -- two vector instructions to turn three lane masks into one)
#ifndef RP_SPLIT_BALLOTS
#define RP_SPLIT_BALLOTS 1      // 0: one ballot of the conjunction (A/B)
#endif
                if constexpr (RP_SPLIT_BALLOTS != 0) atx_m |= step_m & ballot_(v == (T)bk.get(0)) & ballot_(t0 == (T)bk.get(1)) & ballot_(t1 == (T)bk.get(2));
                else atx_m |= step_m & ballot_(v == (T)bk.get(0) && t0 == (T)bk.get(1) && t1 == (T)bk.get(2));
            }
        }
        if (used != 0) { asm volatile(""); it = 0; }
    } else {
        used_t used = VARIANT == 4;      // wave-uniform: some lane has moved its counter
        [[maybe_unused]] bool at_x = false;      // FROZEN: the trial point has become x itself (and every later one will be)
        for (;;) {
            accel_values_u(k, v, t0, t1, et, xt);
            const bool bad = !all_satisfied<T, VARIANT>(et, L);
            const unsigned long long any_bad = __builtin_amdgcn_ballot_w64(bad);
            if (any_bad == 0ull) break;
            // (out of halvings: the reference goes on with an s it has not tested)
            const bool room = it < kp.max_bt;
            if ((any_bad & __builtin_amdgcn_ballot_w64(room)) == 0ull) break;
            used = 1;
            if (bad && room) {
                if (FROZEN && at_x) {
                    // x itself fails the test by a rounding (the residual loop accepts points the feasibility loop never saw, as the
                    // reference's does) and every smaller s gives x again: the reference walks its remaining halvings to the same
                    // verdict.  So does this, without the evaluations.
                    if (kp.backtrack == T(0.5) && std::is_same<D, NoDiag>::value) {
                        s = ldexp_(s, it - kp.max_bt);
                    } else {
                        for (int q = it; q < kp.max_bt; ++q) { s *= kp.backtrack; diag.feas(); }
                    }
                    it = kp.max_bt;
                } else {
                    s *= kp.backtrack;
                    ++it;
                    diag.feas();
                    const T x0 = bk.get(0), x1 = bk.get(1), x2 = bk.get(2);
                    v = fma_(dxv, s, x0);
                    t0 = fma_(dx0, s, x1);
                    t1 = fma_(dx1, s, x2);
                    if constexpr (FROZEN) at_x = v == x0 && t0 == x1 && t1 == x2;
                }
            }
        }
        if (used != 0) { asm volatile(""); it = 0; }      // (a branch, not a select: the common path does not touch the counter)
    }
    // -- backtrack until the residual decreases (onedpath_ip.cpp:932-945), and take the step (:949-952): the trial that ends
    // the loop -- accepted, or the last s, which the reference takes untested -- is the new state, its sums the next step's --
#pragma unroll
    for (int i = 0; i < NC; ++i) lam[i] = fma_(dl[i], s, lam[i]);
    {
        used_t used = 0;
#ifndef RP_FROZEN_INTFLAG
#define RP_FROZEN_INTFLAG 1      // the flag lives in a vector register as 0 / 1 and every test of it is a FRESH compare (0: a bool, A/B) -- a bool carried round
                                 // the loop is merged by three scalar mask operations per trip and comes back through v_cndmask / v_cmp at each ballot:
                                 // 230 -> 187 scalar instructions per post-convergence wave-step of a lone wave (profiles/r6_tuning.md).  Where the step's
                                 // start waits in registers only (the small-batch kernels, which are the ones a lone wave runs): the LDS-column kernels sit
                                 // at 128 VGPRs for their fourth wave and have none to spare for it
#endif
        constexpr bool kIntFlag = RP_FROZEN_INTFLAG != 0 && !std::is_same<BK, LdsColumn<T>>::value;
        [[maybe_unused]] std::conditional_t<kIntFlag, int, bool> frozen = 0;      // FROZEN: the trial point has become bitwise x (and stays so: s only shrinks)
        for (;;) {
            accel_grads_u<T, P>(et, xt);
            residual_sums<T, VARIANT, false>(et, lam, dl, T(0), L, c.X, c.Q1, c.Q2, c.cm, c.cp);      // (r0n and the direction have taken what they needed from c)
            const T rn = residual_from_sums<T, NC>(c.X, c.Q1, c.Q2, p);
            diag.moving();
            const bool bad = !(rn <= r0n * (T(1) - kp.armijo * s));
            const unsigned long long any_bad = __builtin_amdgcn_ballot_w64(bad);
            if (any_bad == 0ull) break;
            const bool room = it < kp.max_bt;
            const bool again = bad && room && !(FROZEN && frozen != 0);
            [[maybe_unused]] unsigned long long again_m = 0ull;      // = ballot(again), from one ballot per comparison (see the feasibility loop)
            if constexpr (FROZEN) {
                if constexpr (RP_SPLIT_BALLOTS != 0) again_m = any_bad & __builtin_amdgcn_ballot_w64(room) & __builtin_amdgcn_ballot_w64(frozen == 0);
                else again_m = __builtin_amdgcn_ballot_w64(again);
                if (again_m == 0ull) break;
            } else { if ((any_bad & __builtin_amdgcn_ballot_w64(room)) == 0ull) break; }
            used = 1;
            if (again) {
                s *= kp.backtrack;
                ++it;
                diag.resid();
                // (all eleven reads are issued before the first use: the multipliers arrive under the trial point's evaluation)
                const T x0 = bk.get(0), x1 = bk.get(1), x2 = bk.get(2);
                const T kdx0 = k.dx0, kdx1 = k.dx1;
#pragma unroll
                for (int i = 0; i < NC; ++i) lam[i] = bk.get(3 + i);
                v = fma_(dxv, s, x0);
                t0 = fma_(dx0, s, x1);
                t1 = fma_(dx1, s, x2);
                if constexpr (FROZEN) frozen = (v == x0 && t0 == x1 && t1 == x2) ? 1 : 0;
                accel_values_u(k, v, t0, t1, et, xt, kdx0, kdx1);
#pragma unroll
                for (int i = 0; i < NC; ++i) lam[i] = fma_(dl[i], s, lam[i]);
            }
            // FROZEN: when every lane that is still searching has frozen, the affine search below takes over at once (its first
            // candidate is the trial just formed); the sums of such a trial are only wanted for the one the search ends on
            if constexpr (FROZEN) {
                if constexpr (RP_SPLIT_BALLOTS != 0) { if ((again_m & __builtin_amdgcn_ballot_w64(frozen == 0)) == 0ull) break; }
                else { if (__builtin_amdgcn_ballot_w64(again && frozen == 0) == 0ull) break; }
            }
        }
        if constexpr (FROZEN) {
            // ---- frozen: x + s dx == x from here on ----
            // The evaluation of the point is loop-invariant (et, formed above from the very bits of x); only the multipliers
            // lam + s dl still depend on s.  The rest of the search runs on the affine pieces of the residual (AffineResidual:
            // 27 instructions per halving, two step lengths per trip) against the value the same pieces give at s = 0 -- the
            // regime's own, self-consistent evaluation, as the reference's loop is: it ends where s r' drops below half an
            // ulp of r.  The trial it ends on is the new state; its sums, in the carried form, are the next step's.
            if (__builtin_amdgcn_ballot_w64(frozen != 0) != 0ull) {
                used = 1;
                if (frozen != 0) {
                    accel_grads_u<T, P>(et, xt);      // (the loop above may have been left before it came round to this trial)
                    AffineResidual<T, VARIANT> ar;
                    ar.setup(et, [&](int i) { return (T)bk.get(3 + i); }, dl, p, L);
                    const T r0a = ar(T(0));
                    if (RP_FROZEN_PROOF && kp.backtrack == T(0.5)) {
                        // the trials that fail beyond doubt, counted in closed form (AffineResidual::certain_failures)
                        int skip = ar.certain_failures(kp, r0a, s);
                        const int room = kp.max_bt - it;
                        skip = skip < room ? skip : room;
                        skip = skip > 0 ? skip : 0;
                        s = ldexp_(s, -skip);
                        it += skip;
                        if constexpr (!std::is_same<D, NoDiag>::value)
                            for (int q = 0; q < skip; ++q) diag.resid();
                    }
                    ar.search(kp, r0a, s, it, diag);
#pragma unroll
                    for (int i = 0; i < NC; ++i) lam[i] = fma_(dl[i], s, bk.get(3 + i));
                    residual_sums<T, VARIANT, false>(et, lam, dl, T(0), L, c.X, c.Q1, c.Q2, c.cm, c.cp);
                }
            }
        }
        if (used != 0) { asm volatile(""); it = 0; }
    }
    c.r0 = et.r0; c.r1 = et.r1;
    c.x = xt;
#pragma unroll
    for (int j = 0; j < 4; ++j) { c.a[j] = et.a[j]; c.gt[j] = et.gt[j]; }
}

// (the gated solve's call: no bookkeeping, no frozen regime)
template <typename T, int VARIANT, class P, class BK>
__device__ __forceinline__ void newton_step_inplace(const P &k, const KParams<T> &kp, T gap, T &v, T &t0, T &t1, T (&lam)[CMap<VARIANT>::NC],
                                                    AccCarry<T, true, true> &c, BK &bk, int &it)
{
    NoDiag none;
    newton_step_inplace<T, VARIANT, P, false, NoDiag, BK>(k, kp, gap, v, t0, t1, lam, c, bk, it, none);
}

// the common call: no bookkeeping
template <typename T, int VARIANT, class P, bool MEMO = true, bool AFFINE = false, int MU = 0>
__device__ __forceinline__ void newton_step(const P &k, const KParams<T> &kp, T gap,
                                            T &v, T &t0, T &t1, T (&lam)[CMap<VARIANT>::NC], AccCarry<T, !MEMO, !MEMO && MU == 0> &c)
{
    NoDiag none;
    newton_step<T, VARIANT, P, MEMO, AFFINE, MU, NoDiag>(k, kp, gap, v, t0, t1, lam, c, none);
}

}  // namespace rp
