// ip_kernels.hip -- HIP kernels of the batched interior-point path, gfx950 only.
//
// Data layout in HBM: structure of arrays.  Field f of the problem at position s lives at
// base[f * stride + s]; the field order is the reference's enum order (enum V,
// onedpath_ip.cpp:15-43; enum V2, onedpath2_ip.cpp:15-39), so field 0..2 are the variables,
// 3..3+m-1 the multipliers and the last five the constants.  The stride is an odd multiple of 512
// elements (rp_batch.cpp: a power-of-two distance between the fields puts all of them on one HBM
// channel).  One Newton step reads 16 and writes 11 fields (F3): 216 B per problem per step, the
// algorithmic traffic of SURVEY.md 8d; the kernels instantiated for zero end velocities read 14.
//
// Within the batch the problems lie in SCHEDULED order (schedule.hip, run when positions are set): sorted by what predicts
// their gated step count, the segment-length ratio min|dX| / max|dX| (64 classes; similar segments take longest) and,
// within a class, the longer segment's length (32 levels).  slot_of[] / prob_of[] map problem index <-> position; only the
// kernels at the ABI boundary (state in / out, read-backs by problem index) and the first kernel after set_problems look at
// them, the Newton kernels walk positions.
//
// Launch shapes, all sharing the per-lane step of ip_core.h:
//   k_solve_chunks     the gated solve (the benchmark's kernel): one 64-problem chunk of the scheduled order per
//                      single-wave block, state in registers from its first load to its last store (LDS only as the in-place
//                      step's backup column, no staging), longest chunks dispatched first
//   k_steps_chunks     k >= 2 ungated steps: the same shape, fixed step count.  F3 runs the gated solve's in-place step here too
//                      (round 4; below three waves per SIMD an instantiation that keeps the step's start in registers); F4 keeps
//                      newton_step_to with the wave-parallel line search -- a wave's stragglers in the residual loop are served
//                      by the whole wave
//   k_newton_stream16  k = 1 (one launch per Newton step), the HBM-streaming form: 16 B per lane (two doubles / four floats =
//                      that many consecutive problems per lane), one global_load/store_dwordx4 per field
//   k_newton_stream    the same with one problem per lane and a register prefetch: ragged remainders, mu_mode 1
// Every fixed-step kernel of a variant runs the same step function, so step(k) is k x step(1) bit for bit in every shape.
// Problems are independent and nothing is re-read, so there is no L2 locality to arrange: consecutive blocks are dealt
// round-robin over the 8 XCDs and touch disjoint cache lines.
#include "ip_kernels.h"

#include <cstdlib>

#include "../../include/rp_batch.h"
#include "feas_core.h"
#include "ip_core.h"

namespace rp {

namespace {

constexpr int kBlock = 256;
// once-per-launch global accesses (state in, results out): nontemporal, they are never re-read through the caches
#if !defined(RP_STREAM_PLAIN) && !defined(RP_TILE_PLAIN)
template <typename S> __device__ __forceinline__ S ld_once(const S *p) { return __builtin_nontemporal_load(p); }
template <typename S> __device__ __forceinline__ void st_once(S *p, S v) { __builtin_nontemporal_store(v, p); }
#else
template <typename S> __device__ __forceinline__ S ld_once(const S *p) { return *p; }
template <typename S> __device__ __forceinline__ void st_once(S *p, S v) { *p = v; }
#endif
// Progress counters are sharded over 64 words each (open lanes: [0,64), gated steps: [64,128)):
// one word takes ~88 atomics/us, and a k = 1 gated launch ends with 16384 waves arriving at once
// (measured: 0.34 ms of a 0.40 ms launch was the two single-word atomics).
constexpr int kShards = 64;

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
    kp.x_floor = T(2) * kp.limit * kp.c_floor;
    kp.max_bt = hp.max_backtracks;
    kp.stall_window = hp.stall_window;
    kp.sigma_try[0] = (T)hp.mu_sigma_try[0];
    kp.sigma_try[1] = (T)hp.mu_sigma_try[1];
    kp.handoff_lanes = 0;
    kp.handoff_patience = 0;
    return kp;
}

// one problem's answer as its 32-byte record: two 16-byte nontemporal stores = one whole sector
__device__ __forceinline__ void store_solution(Solution *rec, double v, double t0, double t1, int it, uint32_t st)
{
    typedef double v2 __attribute__((ext_vector_type(2)));
    const unsigned long long words = ((unsigned long long)st << 32) | (unsigned long long)(uint32_t)it;      // little endian: iters, then status
    const v2 a = {v, t0}, b = {t1, __longlong_as_double((long long)words)};
    v2 *dst = reinterpret_cast<v2 *>(rec);
#ifdef RP_SOLUTION_NT
    __builtin_nontemporal_store(a, dst);
    __builtin_nontemporal_store(b, dst + 1);
#else
    dst[0] = a;      // plain stores: the two halves of the sector meet in L2 and leave it as one write -- 29.5 us per 1 Mi records
    dst[1] = b;      // where nontemporal stores take 79 us and a gather through slot_of 46 us (profiles/r4_solution_probe.log)
#endif
}

// ---------------------------------------------------------------------------------------
// run_lane: up to k Newton steps of one problem, state in registers between steps.
//   GATED = false : exactly k steps (k presses of 'n', onedpath_ip.cpp:269-272)
//   GATED = true  : before each step stop if gap < tol or the problem's step count reached
//                   max_iter; problems already finished are skipped
//                   without touching their state.
#ifndef RP_NEWTON_WAVES
#define RP_NEWTON_WAVES 2     // minimum waves per SIMD the register allocator must leave room for
#endif
#ifndef RP_GATED_WAVES
#define RP_GATED_WAVES 4     // 128 VGPRs (112-128 used): the in-place step with the cold values parked in LDS fits without a spill in the loop; 1 Mi problems = 16,384
                             // waves = exactly four full rounds of the chip's 4,096 wave slots (three per SIMD: 5.33 rounds).  The stall-detector
                             // twin (two more live values, 10 spilled at 128) stays at three
#endif
// Which residual-loop form the fixed-step kernels use once the trial point has become x (newton_step's AFFINE): every kernel of
// a variant uses the same one, so that all launch shapes agree bit for bit.  This switch concerns newton_step_to, i.e. F4 (affine
// pieces: it reaches that regime within a dozen steps and has the registers) and mu_mode 1; since round 4 F3's fixed-step
// launches run newton_step_inplace<FROZEN>, whose post-convergence search is on affine pieces as well (ip_core.h).
#ifdef RP_AFFINE_ALL      // A/B build (profiles/r3_tuning.md: no gain for F3, 168 VGPRs + 2 spilled in its chunk kernel)
template <int VARIANT> constexpr bool kAffine = true;
#else
template <int VARIANT> constexpr bool kAffine = (VARIANT == 4);
#endif
#ifndef RP_WAVE_LS
#define RP_WAVE_LS 1      // wave-parallel line search in F4's fixed-step chunk kernel (0: the serial loop, for A/B runs)
#endif
#ifndef RP_GATED_IN_PLACE
#define RP_GATED_IN_PLACE 1     // the gated kernel's step overwrites the state, its start backed up in LDS (0: newton_step_to, for A/B runs)
#endif
#ifndef RP_TILED_WAVES
#define RP_TILED_WAVES 3     // F4's fixed-step chunk kernels (156-168 VGPRs) and F3's register-column instantiation; the in-place kernels have RP_GATED_WAVES
#endif

// The per-lane body shared by both Newton kernels: up to k steps on the state held in registers.
// STALL: compile the stall detector in (two more live registers); the tiled solve instantiates both
// forms and picks by rp_params.stall_window, so the default (off) pays nothing for it.
// S = storage type of the batch.  When it differs from the compute type T (fp32 state, fp64 arithmetic) the state is
// rounded to S after every step, so that a step is a function "S state -> S state" whatever the launch shape:
// step(k) stays bit-identical to k x step(1).
// WAVE (ungated launches only, whose live lanes all take the same k steps together): the residual loop's stragglers are served
// by the whole wave (newton_step_to, "wave-parallel line search").
// Which step a launch runs.  In place (newton_step_inplace: the step's start waits in LDS, residual sums carried, wave-uniform
// loops): the gated solve in the reference's mu mode, and -- round 4 -- EVERY fixed-step launch of F3, whose kernels ran
// newton_step_to with per-lane loop exits until then (389 VALU instructions per step against the gated kernel's 307).  Since all
// of F3's fixed-step kernels run the one function, step(k) is k x step(1) bit for bit in every launch shape as before, and a
// gated launch whose gate never closes now takes the very same steps as well.  F4's fixed-step launches keep newton_step_to (the
// wave-parallel line search and the affine post-convergence loop are built on it); mu_mode 1 likewise.
template <int VARIANT, bool GATED, int MU, class D>
constexpr bool kStepInPlace = RP_GATED_IN_PLACE && MU == 0 && (GATED ? std::is_same<D, NoDiag>::value : VARIANT == 3);

// PARK (RP_PARK_FIXED_POINTS; F4's fused fixed-step launches on an fp32 state with fp64 arithmetic): a lane whose step has left its STORED
// state bit for bit where it was sits the remaining steps of the launch out -- see the ungated loop below.  (Pure fp32 arithmetic keeps
// the plain loop: there the stuck problems are few -- its Armijo test stops resolving long before a state freezes -- and the bookkeeping
// cost more than it saved: 50.0 against 53.8 G steps/s at 1 Mi x 50, profiles/r6_f4_park_ab.log.  The code below still handles S == T.)
#ifndef RP_PARK_FIXED_POINTS
#define RP_PARK_FIXED_POINTS 1      // 0: every lane takes every step (A/B builds: no bit may change)
#endif
template <typename T, int VARIANT, bool GATED, bool STALL = GATED, class P = Prob<T>, typename S = T, bool AFFINE = false, int MU = 0, class D = NoDiag, int WAVE = 0,
          class BK = LdsColumn<T>, bool PARK = false, bool ROUNDS = false>
__device__ __forceinline__ void run_lane(const P &pr, const KParams<T> &kp, int k, T tol, int max_iter,
                                         T &v, T &t0, T &t1, T (&lam)[CMap<VARIANT>::NC],
                                         int &it, uint32_t &st, int &steps_here, bool &still_open, D &diag, BK backup = BK{})
{
    static_assert(!(WAVE != 0 && GATED), "lanes of a gated solve leave the loop at different steps");
    constexpr bool INPLACE = kStepInPlace<VARIANT, GATED, MU, D>;
    // gated kernels carry the time derivatives as well (newton_step's MEMO = !GATED) and, in the reference's mu mode, the
    // residual sums, from which the gap of the current point comes for free; so does every launch that steps in place
    using Carry = AccCarry<T, GATED || INPLACE, (GATED && MU == 0) || INPLACE>;
    Carry e;            // the evaluation at the current point, carried from step to step
    auto evaluate = [&]() {
        Acc<T> e0;
        PointAux<T> x0;
        accel_values_u(pr, v, t0, t1, e0, x0);
        e.r0 = e0.r0; e.r1 = e0.r1;
#pragma unroll
        for (int j = 0; j < 4; ++j) e.a[j] = e0.a[j];
        if constexpr (GATED || INPLACE) {
            accel_grads_u<T, P>(e0, x0);
#pragma unroll
            for (int j = 0; j < 4; ++j) e.gt[j] = e0.gt[j];
            if constexpr (Carry::has_sums) {
                residual_sums<T, VARIANT, false>(e0, lam, lam, T(0), kp.limit, e.X, e.Q1, e.Q2, e.cm, e.cp);
                e.x = x0;
            }
        }
    };
    auto current_gap = [&]() -> T {
        if constexpr (Carry::has_sums) return -e.Q1;
        else return duality_gap<T, VARIANT, Carry>(e, lam, kp.limit);
    };
    evaluate();

    bool done = false;
    T best_gap = T(3.0e38);      // stall detector state (off unless kp.stall_window > 0)
    int since_best = 0;
    const T objective_in = t0 + t1;      // the objective (total duration) this launch started from: RP_ST_WRONG_WAY below
    [[maybe_unused]] int left = max_iter - it;      // GATED: steps this problem may still take -- the ONE per-lane counter of the solve (it = max_iter - left
                                                    // afterwards; the caller has the count it came in with and takes the difference as steps_here)
    [[maybe_unused]] int taken = 0;                 // (STALL only: whether this launch moved the problem at all)
    [[maybe_unused]] int halvings = 0;              // the in-place step's line-search counter: zero whenever a loop is entered (newton_step_inplace)
    if constexpr (GATED && !STALL) {
        // Wave-uniform loop: it runs while any lane of the wave still has steps to take, and a lane that has reached its gate
        // sits the rest out under the execution mask.  (Lanes leaving a loop one by one make the compiler copy every value that
        // is live after it -- the whole state -- at every trip.)  The test is a ballot over the gate's own comparisons -- a lane
        // that has passed its gate passes it again, its state no longer moves -- and the status bits are read off the final
        // point afterwards: nothing is carried round the loop but the state.
        [[maybe_unused]] int lonely = 0;      // ROUNDS: steps this wave has taken with no more than kp.handoff_lanes of its lanes stepping
        for (int s = 0; s < k; ++s) {
            const T gap = current_gap();
            const bool above = !(gap < tol), may = left > 0;
            const unsigned long long stepping = __builtin_amdgcn_ballot_w64(above) & __builtin_amdgcn_ballot_w64(may);      // (two ballots, each its comparison's own lane mask)
            if (stepping == 0ull) break;
            if constexpr (ROUNDS) {
                // Straggler hand-off: a wave runs until its slowest lane is done, and with step counts that nothing predicted (states that
                // did not come from the feasible-start rule: nudged, set, restored) one lane that needs 200 steps holds 63 finished ones for
                // 185.  When only a few lanes are still stepping, and have been for `patience` steps, the wave stops; those lanes stay open,
                // their positions go onto a list (k_solve_chunks) and the next launch of the solve packs them densely into new waves.
                if (__popcll(stepping) <= kp.handoff_lanes) { if (++lonely > kp.handoff_patience) break; }
            }
            if (above && may) {
                if constexpr (INPLACE)
                    newton_step_inplace<T, VARIANT, P, BK>(pr, kp, gap, v, t0, t1, lam, e, backup, halvings);      // the step's start waits in LDS, the accepted trial is the state
                else
                    newton_step<T, VARIANT, P, false, AFFINE, MU, D, WAVE>(pr, kp, gap, v, t0, t1, lam, e, diag);      // gated solves never reach the regime the memoisation is for
                if constexpr (sizeof(S) != sizeof(T)) {
                    v = (T)(S)v; t0 = (T)(S)t0; t1 = (T)(S)t1;
#pragma unroll
                    for (int c = 0; c < CMap<VARIANT>::NC; ++c) lam[c] = (T)(S)lam[c];
                    evaluate();                      // the carried evaluation belongs to the unrounded point
                }
                --left;
                if constexpr (ROUNDS && INPLACE) {
                    // Fixed points use their budget up at once (exact).  A start outside the feasible set -- a nudged velocity, durations
                    // cut short: what SURVEY 8f's callers feed in -- makes no progress in the reference: 100 feasibility halvings
                    // (onedpath_ip.cpp:919-928), x + s dx is x bit for bit, and so are the multipliers; the next step starts from the same
                    // bits and does the same again, up to the step budget: 200 steps of a hundred evaluations each for one lane, holding its
                    // wave.  The step is a function of the state (the carried evaluation is a function of it too: formed by the same
                    // expressions at the same point), so a state it maps onto itself stays: the lane takes its remaining steps as read --
                    // the count, the status and every bit it stores are what stepping on would have left.  The step's start is still in the
                    // LDS column: one read and one compare per step screen (vel1 unchanged, which a moving iterate never shows), the other
                    // ten behind a branch that is taken when some lane passes the screen.
                    const bool v_same = __builtin_bit_cast(unsigned long long, (double)v) == __builtin_bit_cast(unsigned long long, (double)(T)backup.get(0));
                    if (__builtin_amdgcn_ballot_w64(v_same) != 0ull) {
                        bool same = v_same && t0 == (T)backup.get(1) && t1 == (T)backup.get(2);      // (durations are positive and finite: value equality is bit equality)
#pragma unroll
                        for (int c = 0; c < CMap<VARIANT>::NC; ++c)
                            same = same && __builtin_bit_cast(unsigned long long, (double)lam[c]) == __builtin_bit_cast(unsigned long long, (double)(T)backup.get(3 + c));
                        if (same) left = 0;
                    }
                }
            }
        }
    } else if constexpr (GATED) {
        bool open = true;
        for (int s = 0; s < k; ++s) {
            if (open) {
                const T gap = current_gap();
                if (gap < tol) { st |= RP_ST_CONVERGED; done = true; open = false; }
                else if (left <= 0) { st |= RP_ST_MAXITER; done = true; open = false; }
                else if (STALL && kp.stall_window > 0) {
                    if (gap < T(0.5) * best_gap) { best_gap = gap; since_best = 0; }
                    else if (++since_best >= kp.stall_window) { st |= RP_ST_STALLED; done = true; open = false; }
                }
                if (open) {
                    if constexpr (INPLACE)
                        newton_step_inplace<T, VARIANT, P, BK>(pr, kp, gap, v, t0, t1, lam, e, backup, halvings);
                    else
                        newton_step<T, VARIANT, P, false, AFFINE, MU, D, WAVE>(pr, kp, gap, v, t0, t1, lam, e, diag);
                    if constexpr (sizeof(S) != sizeof(T)) {
                        v = (T)(S)v; t0 = (T)(S)t0; t1 = (T)(S)t1;
#pragma unroll
                        for (int c = 0; c < CMap<VARIANT>::NC; ++c) lam[c] = (T)(S)lam[c];
                        evaluate();                      // the carried evaluation belongs to the unrounded point
                    }
                    --left;
                    ++taken;
                }
            }
            if (__builtin_amdgcn_ballot_w64(open) == 0ull) break;
        }
    } else if constexpr (PARK && RP_PARK_FIXED_POINTS != 0) {
        // ---- fixed points sit out (exact) ----
        // F4 never converges (README.md:34): from step ~6 on a growing share of its problems -- 2.4 % by step 48 on the benchmark
        // distribution -- is STUCK: the residual loop (onedpath2_ip.cpp:820-833) halves the step ~52 times until x + s dx is x bit for
        // bit and the multipliers move by less than an ulp of their fp32 storage, so the step stores the state it loaded.  The step is
        // a function of the stored state (and of the evaluation carried with it, which is a function of the state as well): a
        // state it maps onto itself it maps onto itself for ever, and every further step of the launch is the same ~52-halving
        // walk to the same bits -- the lanes the wave-parallel line search exists for.  Such a lane is parked: it takes no further
        // steps in this launch, and what it stores is bit for bit what the steps would have left (tests/checks/fixed_step_ab.py
        // against a build with RP_PARK_FIXED_POINTS=0; measured on the oracle, profiles/r6_tuning.md: the stuck problems and the
        // fixed points of the fp32-state step are the same problems, none ever leaves).  Bit patterns are compared, not values
        // (-0 against +0 is a change, NaN against the same NaN is none), and with S == T (no re-evaluation after the step) the
        // carried evaluation is compared too.  A wave whose lanes are all parked leaves the loop.
        static_assert(!INPLACE && MU == 0 && !GATED && sizeof(S) == 4, "parking is built on newton_step_to's separate outputs, for states stored in fp32");
        constexpr int NCc = CMap<VARIANT>::NC;
        auto bits = [](T x) { if constexpr (sizeof(T) == 8) return __builtin_bit_cast(unsigned long long, x); else return __builtin_bit_cast(unsigned, x); };
        bool parked = false;
        for (int s = 0; s < k; ++s) {
            if (__builtin_amdgcn_ballot_w64(!parked) == 0ull) { steps_here += k - s; break; }
            if (!parked) {
                const T gap = current_gap();
                T nv, nt0, nt1, nlam[NCc];
                Carry ne;
                // (the problem's two deltas made opaque once per step: otherwise the compiler hoists every loop-invariant function of them the
                // proofs use -- 6 dX, its negation, its magnitude, the margins: seven register pairs -- out of the step loop, and the
                // fp32-state instantiation, 166 of the 168 VGPRs three waves allow, spills; recomputing them is six instructions per step)
                P pq = pr;
                if constexpr (sizeof(T) == 8) asm volatile("" : "+v"(pq.dx0), "+v"(pq.dx1));
                newton_step_to<T, VARIANT, P, true, AFFINE, MU, D, WAVE>(pq, kp, gap, v, t0, t1, lam, e, nv, nt0, nt1, nlam, ne, diag);
                // Round to the storage type and take the new state; behind a screen on vel1 alone (a moving iterate never repeats its velocity
                // bit for bit: one conversion and one compare per step) the whole state is compared with what was stored, value by value
                // before it is overwritten (the kernel has no registers to hold two states side by side).
                bool same = false;
                auto put = [&](T &cur, T nxt) { if constexpr (sizeof(S) != sizeof(T)) cur = (T)(S)nxt; else cur = nxt; };
                auto sbits = [](T x) { if constexpr (sizeof(S) == 4) return __builtin_bit_cast(unsigned, (S)x); else return __builtin_bit_cast(unsigned long long, (S)x); };      // (of the value as stored)
                const bool v_same = sbits(nv) == sbits(v);
                if (__builtin_amdgcn_ballot_w64(v_same) != 0ull) {
                    same = v_same && sbits(nt0) == sbits(t0) && sbits(nt1) == sbits(t1);
#pragma unroll
                    for (int c = 0; c < NCc; ++c) same = same && sbits(nlam[c]) == sbits(lam[c]);
                    if constexpr (sizeof(S) == sizeof(T)) {      // (otherwise evaluate() below rebuilds the carried evaluation from the rounded state: a function of it)
                        same = same && bits(ne.r0) == bits(e.r0) && bits(ne.r1) == bits(e.r1);
#pragma unroll
                        for (int j = 0; j < 4; ++j) same = same && bits(ne.a[j]) == bits(e.a[j]);
                    }
                }
                put(v, nv); put(t0, nt0); put(t1, nt1);
#pragma unroll
                for (int c = 0; c < NCc; ++c) put(lam[c], nlam[c]);
                parked = same;
                if constexpr (sizeof(S) != sizeof(T)) evaluate();      // the carried evaluation belongs to the unrounded point
                else e = ne;
            }
            ++steps_here;
        }
    } else
    for (int s = 0; s < k; ++s) {
        const T gap = current_gap();
        if constexpr (INPLACE)
            newton_step_inplace<T, VARIANT, P, true, D, BK>(pr, kp, gap, v, t0, t1, lam, e, backup, halvings, diag);      // FROZEN: with the post-convergence regime's loops
        else
            newton_step<T, VARIANT, P, true, AFFINE, MU, D, WAVE>(pr, kp, gap, v, t0, t1, lam, e, diag);
        if constexpr (sizeof(S) != sizeof(T)) {
            v = (T)(S)v; t0 = (T)(S)t0; t1 = (T)(S)t1;
#pragma unroll
            for (int c = 0; c < CMap<VARIANT>::NC; ++c) lam[c] = (T)(S)lam[c];
            evaluate();                      // the carried evaluation belongs to the unrounded point
        }
        ++steps_here;
    }
    if constexpr (GATED) it = max_iter - left;
    else it += steps_here;
    if (GATED) {
        if (!done) {   // the status of the point the launch leaves (and so the host knows whether to launch again)
            const T gap = current_gap();
            if (gap < tol) { st |= RP_ST_CONVERGED; done = true; }
            else if (left <= 0) { st |= RP_ST_MAXITER; done = true; }
        }
        st &= ~(RP_ST_NONFINITE | RP_ST_INFEASIBLE);
        // F4 "tends to settle the wrong direction" (README.md:34): from the feasible start its total duration GROWS (7.0 ->
        // 7.07 on the default problem, optimum 4.0) while the gap sticks near 0.47.  With the stall detector on, a problem that
        // stops unconverged with an objective no better than the one this launch started from is flagged.
        if (STALL && kp.stall_window > 0 && taken > 0 && !(st & RP_ST_CONVERGED) && !(t0 + t1 < objective_in)) st |= RP_ST_WRONG_WAY;
        if (st & RP_ST_CONVERGED) st &= ~RP_ST_WRONG_WAY;      // the flag is per launch: a short launch (host-polled rounds) may not lower the objective of a problem that converges later
        if (!(finite_(v) && finite_(t0) && finite_(t1))) st |= RP_ST_NONFINITE;
        if constexpr (Carry::has_sums && VARIANT == 3) {
            // from the carried constraint values -a - L, a - L (exact signs: a floating-point difference has the sign of the
            // comparison of its operands), so that the accelerations themselves need not stay in registers through the solve
            bool ok = true;
#pragma unroll
            for (int j = 0; j < 4; ++j) ok = ok && !(e.cm[j] > T(0)) && !(e.cp[j] > T(0));
            if (!ok) st |= RP_ST_INFEASIBLE;
        } else {
            if (!all_satisfied<T, VARIANT, Carry>(e, kp.limit)) st |= RP_ST_INFEASIBLE;
        }
        still_open = !done;
    }
}

// the common call: no line-search bookkeeping
template <typename T, int VARIANT, bool GATED, bool STALL = GATED, class P = Prob<T>, typename S = T, bool AFFINE = false, int MU = 0, int WAVE = 0, bool PARK = false>
__device__ __forceinline__ void run_lane(const P &pr, const KParams<T> &kp, int k, T tol, int max_iter,
                                         T &v, T &t0, T &t1, T (&lam)[CMap<VARIANT>::NC],
                                         int &it, uint32_t &st, int &steps_here, bool &still_open, LdsColumn<T> backup = LdsColumn<T>{})
{
    NoDiag none;
    run_lane<T, VARIANT, GATED, STALL, P, S, AFFINE, MU, NoDiag, WAVE, LdsColumn<T>, PARK>(pr, kp, k, tol, max_iter, v, t0, t1, lam, it, st, steps_here, still_open, none, backup);
}

// ---------------------------------------------------------------------------------------
// The gated solve: up to k gated steps per problem (k = max_iter: every problem to its gate in one launch, the
// benchmark's form; smaller k: one round of a host-polled or globally checked loop).  A wave runs until its slowest lane
// has converged and gated step counts differ from problem to problem (12-20 on the benchmark distribution), so in arbitrary
// order ~20 % of the lane-steps of a fused solve are idle.  The batch is therefore kept in scheduled order (see the top
// of this file) and each single-wave block takes one 64-problem chunk of it: lanes of similar expected length, loaded and
// stored as full coalesced segments with no staging and no block barrier in between.  With one wave per block the hardware
// dispatcher IS the work queue: a wave that finishes early frees its slot for the next chunk, so the tail of the grid is
// one chunk long instead of one 512-problem tile (which cost 11 % at 1 Mi problems: 2,048 tiles over 768 slots), and
// the blocks walk the order from its end, so the longest chunks start first.
// Which lane solves which problem changes nothing in any problem's result (lanes never interact); a stale order
// (positions nudged after it was computed) is merely a less effective schedule.
#ifdef RP_TRACE      // tuning build (profiles/probes/chunk_trace.hip): per chunk (SIMD, start, end, steps) in 100 MHz ticks
__device__ unsigned long long g_trace[4 * 32768];
#define RP_TRACE_BEGIN() const unsigned long long trace_t0 = wall_clock64()
#define RP_TRACE_END(c, steps)                                                                      \
    do {                                                                                            \
        const unsigned long long trace_t1 = wall_clock64();                                         \
        int smax = (steps);                                                                         \
        for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(smax, o); smax = other > smax ? other : smax; } \
        if (threadIdx.x == 0 && (c) < 32768u) {                                                     \
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       /* HW_REG_HW_ID: simd [5:4], cu [11:8], se [15:13] */ \
            const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u;   /* HW_REG_XCC_ID */ \
            g_trace[4 * (c) + 0] = ((unsigned long long)xcc << 32) | hw;                            \
            g_trace[4 * (c) + 1] = trace_t0;                                                        \
            g_trace[4 * (c) + 2] = trace_t1;                                                        \
            g_trace[4 * (c) + 3] = (unsigned long long)smax;                                        \
        }                                                                                           \
    } while (0)
#else
#define RP_TRACE_BEGIN() do {} while (0)
#define RP_TRACE_END(c, steps) do {} while (0)
#endif

// START: the batch has just been given its problems (rp_batch_set_problems_device: scheduled order computed, each problem's
// three positions in its 32-byte record, nothing else materialised) and this launch is its first: every lane loads the
// record of the problem that lies at its position (a gather through prob_of, hidden under the other waves' arithmetic),
// stores the positions into the constant fields, forms the feasible start (k_restart_feasible's rule, same
// arithmetic, so the same bits) in registers instead of loading it, takes iteration count and status as zero, and stores
// unconditionally.  That spares a fresh batch the 128 B per problem the feasible start would write, the 88 B of them this
// kernel would read back, and the progress words' clearing pass.
// (mu_mode 1 carries the split direction: ~210 VGPRs, two waves per SIMD)
// ROUNDS (rp_params.handoff_rounds; never the benchmark's fresh batches): the launch is one round of a solve in rounds.  `list_in` (null: the
// whole batch) holds the POSITIONS this round walks, `list_count` their number; a wave that stops with lanes still open (run_lane's hand-off)
// appends their positions to `list_out` (null in the last round, which runs every lane to its end).  Gathered 8-byte accesses instead of
// coalesced ones from the second round on -- for the few per cent of a batch that get that far.
template <typename S, typename T, int VARIANT, bool STALL, bool ZV, int MU = 0, bool START = false, bool ROUNDS = false>
__global__ void __launch_bounds__(64, MU == 1 ? RP_NEWTON_WAVES : (STALL && RP_GATED_WAVES > 3) ? 3 : RP_GATED_WAVES)
k_solve_chunks(S *__restrict__ base, size_t stride, size_t n, int k, KParams<T> kp, T tol, int max_iter,
               int32_t *__restrict__ iters, uint32_t *__restrict__ status, unsigned long long *__restrict__ counters,
               const StartRecord *__restrict__ records, const uint32_t *__restrict__ prob_of, double start_limit,
               Solution *__restrict__ solution, const uint32_t *__restrict__ sol_prob_of, int iters_add,
               const uint32_t *__restrict__ list_in = nullptr, const uint32_t *__restrict__ list_count = nullptr,
               uint32_t *__restrict__ list_out = nullptr, uint32_t *__restrict__ list_out_count = nullptr)
{
    constexpr int NC = CMap<VARIANT>::NC;
    constexpr int CB = 3 + NC;   // first constant field: pos0, vel0, pos1, pos2, vel2
    static_assert(!(ROUNDS && (START || STALL || MU != 0)), "rounds exist for the plain gated solve of a materialised batch");
    RP_TRACE_BEGIN();
    const unsigned chunk = gridDim.x - 1 - blockIdx.x;      // the scheduled order ends with the longest problems: they start first
    size_t i = (size_t)chunk * 64 + threadIdx.x;
    if constexpr (ROUNDS) {
        if (list_in) {      // this round's problems: entry i of the previous round's list (the grid is sized for the most that list can hold)
            const size_t count = (size_t)*list_count;
            if ((size_t)chunk * 64 >= count) return;
            const size_t e = i;
            i = e < count ? (size_t)list_in[e] : n;      // (n: no problem)
        }
    }

    int it = 0;
    uint32_t st = 0;
    bool active = i < n;
    if (!START && active) {
        it = iters[i];
        st = status[i];
        active = (st & (RP_ST_CONVERGED | RP_ST_MAXITER | RP_ST_STALLED)) == 0;
    }
    if (__ballot(active) == 0) return;      // a chunk that finished in an earlier launch
    int steps_here = 0;
    bool still_open = false;
    // where the in-place step (newton_step_inplace) parks the point and multipliers a step started from: 11 (F4: 7) fields x 64 lanes
    // ... and the problem's two deltas (ProbLds): 13 (9) fields
    constexpr bool kInPlace = kStepInPlace<VARIANT, true, MU, NoDiag>;
    __shared__ T s_backup[kInPlace ? (3 + NC + 2) * 64 : 1];

    if (active) {
        S *f = base + i;
        T v, t0, t1, lam[NC];
        Prob<T, ZV> pr;
        if constexpr (START) {
            static_assert(ZV, "the feasible start has zero end velocities");
            typedef double v2 __attribute__((ext_vector_type(2)));
            const v2 *rec = reinterpret_cast<const v2 *>(records + prob_of[i]);      // the problem that lies here: one 32-byte sector
            const v2 ra = rec[0], rb = rec[1];
            const S s0 = (S)ra[0], s1 = (S)ra[1], s2 = (S)rb[0];      // what the constant fields hold from here on
            f[(CB + 0) * stride] = s0;
            f[(CB + 2) * stride] = s1;
            f[(CB + 3) * stride] = s2;
            const double scale = 3.5 / __builtin_sqrt(12.0);      // k_restart_feasible, operation for operation
            v = T(0);
            t0 = (T)(S)(scale * __builtin_sqrt(6.0 * __builtin_fabs((double)s1 - (double)s0) / start_limit));
            t1 = (T)(S)(scale * __builtin_sqrt(6.0 * __builtin_fabs((double)s2 - (double)s1) / start_limit));
#pragma unroll
            for (int c = 0; c < NC; ++c) lam[c] = T(1);
            pr.dx0 = (T)s1 - (T)s0;
            pr.dx1 = (T)s2 - (T)s1;
        } else {
            v = (T)ld_once(f + 0 * stride); t0 = (T)ld_once(f + 1 * stride); t1 = (T)ld_once(f + 2 * stride);
#pragma unroll
            for (int c = 0; c < NC; ++c) lam[c] = (T)ld_once(f + (3 + c) * stride);
            const T p0 = (T)f[(CB + 0) * stride], p1 = (T)f[(CB + 2) * stride], p2 = (T)f[(CB + 3) * stride];
            if constexpr (!ZV) {
                pr.v0 = (T)f[(CB + 1) * stride];
                pr.v2 = (T)f[(CB + 4) * stride];
            }
            pr.dx0 = p1 - p0;
            pr.dx1 = p2 - p1;
        }
        // The iteration count and the status word are not held through the steps: the lane takes its budget and a clean flag word
        // in, and both are read again (START: known) when the results are merged below.
        int it_new = it;
        uint32_t flags = 0;
        if constexpr (kInPlace) {
            LdsBackup<T> col = (LdsBackup<T>)&s_backup[threadIdx.x];
            col[(3 + NC) * 64] = pr.dx0;
            col[(3 + NC + 1) * 64] = pr.dx1;
            ProbLds<T, ZV> pl;
            if constexpr (!ZV) { pl.v0 = pr.v0; pl.v2 = pr.v2; }
            pl.dx0.at = col + (3 + NC) * 64;
            pl.dx1.at = col + (3 + NC + 1) * 64;
            NoDiag none;
            run_lane<T, VARIANT, true, STALL, ProbLds<T, ZV>, S, false, MU, NoDiag, 0, LdsColumn<T>, false, ROUNDS>(pl, kp, k, tol, max_iter, v, t0, t1, lam, it_new, flags, steps_here, still_open, none, LdsColumn<T>{col});
        } else {
            static_assert(!ROUNDS, "rounds are built on the in-place step's launch shape");
            run_lane<T, VARIANT, true, STALL, Prob<T, ZV>, S, false, MU>(pr, kp, k, tol, max_iter, v, t0, t1, lam, it_new, flags, steps_here, still_open);
        }
        // the store addresses are formed only now (the barrier keeps the compiler from holding them in registers across the steps),
        // from the lane number the hardware counts rather than the thread index that came in a register
        unsigned zero = 0u;
        asm volatile("" : "+v"(zero));      // (opaque, or the count is merged with one taken before the steps and held in a register through them)
        size_t j = (size_t)chunk * 64 + __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
        asm volatile("" : "+v"(j));
        if constexpr (ROUNDS) { if (list_in) j = (size_t)list_in[j]; }      // (read again rather than held through the steps; an active lane's entry exists)
        if constexpr (START) {
            steps_here = it_new;
            it = it_new;
            st = flags;
        } else {
            steps_here = it_new - *(volatile int32_t *)(iters + j);
            it = it_new;
            // run_lane clears NONFINITE / INFEASIBLE (re-derived from the final point) and, on convergence, WRONG_WAY; everything else of the old word stays
            uint32_t old = *(volatile uint32_t *)(status + j);
            old &= ~(RP_ST_NONFINITE | RP_ST_INFEASIBLE);
            if (flags & RP_ST_CONVERGED) old &= ~RP_ST_WRONG_WAY;
            st = old | flags;
        }
        iters[j] = it;
        status[j] = st;
        if (START || steps_here > 0) {
            S *g = base + j;
            st_once(g + 0 * stride, (S)v);
            st_once(g + 1 * stride, (S)t0);
            st_once(g + 2 * stride, (S)t1);
#pragma unroll
            for (int c = 0; c < NC; ++c) st_once(g + (3 + c) * stride, (S)lam[c]);
        }
        // A bound solution buffer (rp_batch_bind_solution): the answer of the problem that lies here -- what the reference leaves in
        // var[vel1X, duration0, duration1] of its Trajectory (onedpath_ip.cpp:47-52) -- goes straight to that problem's 32-byte
        // record, in PROBLEM order: one whole sector per problem, scattered, under the other waves' arithmetic.  (Wave-uniform
        // branch on a kernel argument; the problem index is loaded here, after the steps, not held through them.)
        if (solution) {
            const size_t prob = sol_prob_of ? (size_t)sol_prob_of[j] : j;
            store_solution(solution + prob, (double)(S)v, (double)(S)t0, (double)(S)t1, it + iters_add, st);
        }
    }

    RP_TRACE_END(blockIdx.x, steps_here);
    const unsigned long long open_mask = __ballot(still_open);
    if constexpr (ROUNDS) {
        if (list_out && open_mask) {      // this wave's open lanes onto the next round's list: one atomic per wave, positions packed in lane order
            const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            uint32_t at = 0;
            if (lane == (unsigned)(__ffsll((long long)open_mask) - 1)) at = atomicAdd(list_out_count, (uint32_t)__popcll(open_mask));
            at = __shfl(at, __ffsll((long long)open_mask) - 1);
            if (still_open) {
                size_t mine = (size_t)chunk * 64 + lane;
                if (list_in) mine = (size_t)list_in[mine];
                list_out[at + (uint32_t)__popcll(open_mask & ((1ull << lane) - 1ull))] = (uint32_t)mine;
            }
        }
    }
    const int steps_wave = wave_sum<int>(steps_here);
    if (threadIdx.x == 0) {
        const unsigned shard = blockIdx.x & (kShards - 1);
        if (open_mask) atomicAdd(&counters[shard], (unsigned long long)__popcll(open_mask));
        if (steps_wave) atomicAdd(&counters[kShards + shard], (unsigned long long)steps_wave);
    }
}

// ---------------------------------------------------------------------------------------
// k >= 2 ungated steps per problem: arithmetic-bound like the gated solve, and the same shape -- one
// single-wave block per 64 consecutive positions, state loaded straight into registers, k steps, stored; no LDS.  (Until
// late in round 2 this was a 512-problem tile staged in LDS, the only form that fitted three waves per SIMD: the compiler
// kept the eleven store addresses in registers across the steps.  Forming them after the steps, below, saved 16 VGPRs and
// made the direct form both fit and win: 57.5 -> 65.5 G steps/s at k = 12.)
#ifndef RP_SMALL_WAVES
#define RP_SMALL_WAVES RP_TILED_WAVES     // the register-column instantiation (batches that leave SIMDs with a lone wave): A/B knob, profiles/r5_tuning.md
#endif
template <int VARIANT, bool REGBK> constexpr int kStepsChunkWaves = (kStepInPlace<VARIANT, false, 0, NoDiag> && !REGBK) ? RP_GATED_WAVES : (REGBK ? RP_SMALL_WAVES : RP_TILED_WAVES);

// REGBK (F3): the in-place step's start waits in registers instead of LDS, three waves per SIMD -- the instantiation for batches
// that cannot fill three waves per SIMD anyway (launch_steps picks it below 196,608 problems): a lone wave has nothing to run
// under an LDS round trip, and the post-convergence regime of a fixed-step run (BASELINE configs[1], 65,536 problems x 50 steps)
// makes one per halving.  Same arithmetic, same bits.
// kChunkBlock: threads per block of the chunk kernels = ONE wave.  newton_step_to<WAVE> (F4's fixed-step launches) broadcasts a straggler's
// search state through one LDS area per BLOCK with no barrier -- correct only because the block is a single wave, whose LDS operations complete
// in order; the static_assert below ties that code to this launch shape (in a 256-thread kernel four waves would race on it).
constexpr int kChunkBlock = 64;
template <typename S, typename T, int VARIANT, bool ZV, bool REGBK = false>
__global__ void __launch_bounds__(kChunkBlock, (kStepsChunkWaves<VARIANT, REGBK>))
k_steps_chunks(S *__restrict__ base, size_t stride, size_t n, int k, KParams<T> kp, unsigned lanes)
{
    constexpr int NC = CMap<VARIANT>::NC;
    constexpr int CB = 3 + NC;
    // F3 steps in place, as the gated solve does (round 4): the step's start and the problem's two deltas wait in LDS, 128 VGPRs,
    // four waves per SIMD; F4 keeps newton_step_to with the wave-parallel line search at three
    constexpr bool kInPlace = kStepInPlace<VARIANT, false, 0, NoDiag>;
    static_assert(kInPlace || !REGBK, "the register column belongs to the in-place step");
    __shared__ T s_backup[(kInPlace && !REGBK) ? (3 + NC + 2) * 64 : 1];
    // lanes: problems per wave -- 64 in every shipped launch; tuning builds can leave the upper lanes of every wave empty
    // (RP_LANES_PER_WAVE: part-filled waves for batches that cannot fill the chip, measured and not kept: profiles/r5_tuning.md)
    const size_t i = (size_t)blockIdx.x * lanes + threadIdx.x;
    if (threadIdx.x >= lanes || i >= n) return;
    S *f = base + i;
    T v = (T)ld_once(f + 0 * stride), t0 = (T)ld_once(f + 1 * stride), t1 = (T)ld_once(f + 2 * stride);
    T lam[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) lam[c] = (T)ld_once(f + (3 + c) * stride);
    using Pk = Prob<T, ZV, VARIANT == 4 && sizeof(S) == 8 && sizeof(T) == 8>;
    Pk pr;
    {
        const T p0 = (T)f[(CB + 0) * stride], p1 = (T)f[(CB + 2) * stride], p2 = (T)f[(CB + 3) * stride];
        if constexpr (!ZV) {
            pr.v0 = (T)f[(CB + 1) * stride];
            pr.v2 = (T)f[(CB + 4) * stride];
        }
        pr.dx0 = p1 - p0;
        pr.dx1 = p2 - p1;
    }
    int it = 0, steps_here = 0;
    uint32_t st = 0u;
    bool still_open = false;
    if constexpr (kInPlace && REGBK) {
        NoDiag none;
        Prob<T, ZV> pq;
        if constexpr (!ZV) { pq.v0 = pr.v0; pq.v2 = pr.v2; }
        pq.dx0 = pr.dx0;
        pq.dx1 = pr.dx1;
        run_lane<T, VARIANT, false, false, Prob<T, ZV>, S, false, 0, NoDiag, false, RegColumn<T, 3 + NC>>(pq, kp, k, T(0), 0, v, t0, t1, lam, it, st, steps_here, still_open, none);
    } else if constexpr (kInPlace) {
        LdsBackup<T> col = (LdsBackup<T>)&s_backup[threadIdx.x];
        col[(3 + NC) * 64] = pr.dx0;
        col[(3 + NC + 1) * 64] = pr.dx1;
        ProbLds<T, ZV> pl;
        if constexpr (!ZV) { pl.v0 = pr.v0; pl.v2 = pr.v2; }
        pl.dx0.at = col + (3 + NC) * 64;
        pl.dx1.at = col + (3 + NC + 1) * 64;
        NoDiag none;
        run_lane<T, VARIANT, false, false, ProbLds<T, ZV>, S, false, 0, NoDiag, false>(pl, kp, k, T(0), 0, v, t0, t1, lam, it, st, steps_here, still_open, none, LdsColumn<T>{col});
    } else {
    // F4 has the registers for the affine post-convergence loop (and reaches "the trial point is x" within a dozen steps:
    // its stalled problems), so all its fixed-step kernels use it and agree bit for bit
    // (newton_step_to<WAVE> broadcasts through LDS without a barrier: single-wave blocks only -- the block size travels into the step as
    // the template argument WAVE and is asserted THERE, inside the WAVE branch, so that no multi-wave kernel can instantiate it: ADVICE r5)
    run_lane<T, VARIANT, false, false, Pk, S, kAffine<VARIANT>, 0, (RP_WAVE_LS && VARIANT == 4) ? kChunkBlock : 0, (VARIANT == 4 && sizeof(S) == 4 && sizeof(T) == 8)>(pr, kp, k, T(0), 0, v, t0, t1, lam, it, st, steps_here, still_open);
    }
    // the store addresses are formed only now: the barrier keeps the compiler from holding eleven of them in registers
    // across the steps (168 VGPRs and 4-10 spilled without it, 152 with it)
    size_t j = (size_t)blockIdx.x * lanes + threadIdx.x;
    asm volatile("" : "+v"(j));
    S *g = base + j;
    st_once(g + 0 * stride, (S)v);
    st_once(g + 1 * stride, (S)t0);
    st_once(g + 2 * stride, (S)t1);
#pragma unroll
    for (int c = 0; c < NC; ++c) st_once(g + (3 + c) * stride, (S)lam[c]);
}

// ---------------------------------------------------------------------------------------
// The streaming form of the ungated step: k Newton steps per problem (k = 1 is "one launch per
// press of 'n'"), where 216 B per problem per step really cross HBM.  With k small a lane's
// life is load (~2 us of HBM latency) -> ~1.5 us of arithmetic -> store, and at 2 waves per SIMD
// (the step needs ~190 VGPRs) nothing overlaps the two.  So the grid is sized to what is
// resident (2 blocks per CU) and every lane walks the batch with stride gridDim.x * 256,
// loading problem i + stride into a second register set before it computes problem i: the
// next state streams in under the arithmetic of the current one.
template <typename T, int NF> struct LaneState { T f[NF]; };

template <typename S, typename T, int VARIANT, bool ZV, int MU = 0>
__global__ void __launch_bounds__(kBlock, RP_NEWTON_WAVES)
k_newton_stream(S *__restrict__ base, size_t stride, size_t n, int k, KParams<T> kp)
{
    constexpr int NC = CMap<VARIANT>::NC;
    constexpr int CB = 3 + NC;
    constexpr int NF = CB + 5;
    constexpr bool kInPlace = kStepInPlace<VARIANT, false, MU, NoDiag>;      // F3 in the reference's mu mode: the step's start waits in LDS
    __shared__ T s_backup[kInPlace ? (3 + NC) * kBlock : 1];
    [[maybe_unused]] LdsBackup<T> col = (LdsBackup<T>)&s_backup[(threadIdx.x >> 6) * (3 + NC) * 64 + (threadIdx.x & 63)];      // per wave: fields 64 apart
    const size_t step = (size_t)gridDim.x * kBlock;
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    LaneState<S, NF> cur, nxt;      // the prefetched state waits in the storage type
    if (i >= n) return;
#pragma unroll
    for (int f = 0; f < NF; ++f)
        if (!(ZV && (f == CB + 1 || f == CB + 4))) cur.f[f] = base[(size_t)f * stride + i];     // ZV: end velocities not read
    for (;;) {
        // The prefetch is unconditional (the last round re-reads its own problem): a conditional
        // load makes the compiler's s_waitcnt pass assume the worst at the join and drain the
        // queue (vmcnt(0)) before the arithmetic, which is exactly the overlap this kernel is for.
        const size_t inext = i + step;
        const bool have_next = inext < n;
        const size_t src = have_next ? inext : i;
#pragma unroll
        for (int f = 0; f < NF; ++f)
            if (!(ZV && (f == CB + 1 || f == CB + 4))) nxt.f[f] = base[(size_t)f * stride + src];
        T v = (T)cur.f[0], t0 = (T)cur.f[1], t1 = (T)cur.f[2];
        T lam[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) lam[c] = (T)cur.f[3 + c];
        Prob<T, ZV> pr;
        if constexpr (!ZV) {
            pr.v0 = (T)cur.f[CB + 1];
            pr.v2 = (T)cur.f[CB + 4];
        }
        pr.dx0 = (T)cur.f[CB + 2] - (T)cur.f[CB + 0];
        pr.dx1 = (T)cur.f[CB + 3] - (T)cur.f[CB + 2];
        int it = 0, steps_here = 0;
        uint32_t st = 0;
        bool still_open = false;
        run_lane<T, VARIANT, false, false, Prob<T, ZV>, S, kAffine<VARIANT>, MU>(pr, kp, k, T(0), 0, v, t0, t1, lam, it, st, steps_here, still_open, LdsColumn<T>{col});
        S *f = base + i;
        f[0 * stride] = (S)v;
        f[1 * stride] = (S)t0;
        f[2 * stride] = (S)t1;
#pragma unroll
        for (int c = 0; c < NC; ++c) f[(3 + c) * stride] = (S)lam[c];
        if (!have_next) break;
        cur = nxt;
        i = inext;
    }
}

// ---------------------------------------------------------------------------------------
// Diagnostic twin of the ungated step (rp_batch_step_counted): one problem per lane, k steps, the same arithmetic as
// k_newton_stream, plus the per-problem totals of feasibility and residual halvings -- what the oracle's orc_step_info
// counts -- so that the line search can be compared decision for decision, not only through the states.
template <typename S, typename T, int VARIANT>
__global__ void __launch_bounds__(kBlock)
k_newton_counted(S *__restrict__ base, size_t stride, size_t n, int k, KParams<T> kp, uint32_t *__restrict__ nfeas, uint32_t *__restrict__ nresid)
{
    constexpr int NC = CMap<VARIANT>::NC;
    constexpr int CB = 3 + NC;
    constexpr bool kInPlace = kStepInPlace<VARIANT, false, 0, HalvingDiag>;
    __shared__ T s_backup[kInPlace ? (3 + NC) * kBlock : 1];
    [[maybe_unused]] LdsBackup<T> col = (LdsBackup<T>)&s_backup[(threadIdx.x >> 6) * (3 + NC) * 64 + (threadIdx.x & 63)];
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    S *f = base + i;
    T v = (T)f[0 * stride], t0 = (T)f[1 * stride], t1 = (T)f[2 * stride];
    T lam[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) lam[c] = (T)f[(3 + c) * stride];
    Prob<T, false> pr;
    const T p0 = (T)f[(CB + 0) * stride], p1 = (T)f[(CB + 2) * stride], p2 = (T)f[(CB + 3) * stride];
    pr.v0 = (T)f[(CB + 1) * stride];
    pr.v2 = (T)f[(CB + 4) * stride];
    pr.dx0 = p1 - p0;
    pr.dx1 = p2 - p1;
    HalvingDiag diag;
    int it = 0, steps_here = 0;
    uint32_t st = 0u;
    bool still_open = false;
    run_lane<T, VARIANT, false, false, Prob<T, false>, S, kAffine<VARIANT>, 0, HalvingDiag, false>(pr, kp, k, T(0), 0, v, t0, t1, lam, it, st, steps_here, still_open, diag,
                                                                                                   LdsColumn<T>{col});
    f[0 * stride] = (S)v;
    f[1 * stride] = (S)t0;
    f[2 * stride] = (S)t1;
#pragma unroll
    for (int c = 0; c < NC; ++c) f[(3 + c) * stride] = (S)lam[c];
#ifdef RP_DIAG_MOVING      // tuning build: full residual evaluations instead of feasibility halvings
    nfeas[i] = diag.nm;
#else
    nfeas[i] = diag.nf;
#endif
    nresid[i] = diag.nr;
}

// ---------------------------------------------------------------------------------------
// The same streaming step with 16-byte accesses: a lane owns 16 / sizeof(S) CONSECUTIVE problems (two doubles, four
// floats), loads each field of all of them with one global_load_dwordx4 (a wave moves 1 KiB per instruction), steps
// them one after the other and stores each mutable field with one global_store_dwordx4.  No prefetch registers:
// the 14 x 16 B a lane holds for its problems are the latency cover (57 KiB in flight per resident block).
// Loads and stores are NONTEMPORAL: every byte is touched once per launch, and keeping it out of the caches' way is
// worth 7 % at k = 1 (same box, alternating runs: 4.66 -> 4.98 TB/s; k = 0: 4.79 -> 5.33; -DRP_STREAM_PLAIN for the A/B).
// Covers floor(n / (256 PER)) full blocks; launch_steps hands the ragged remainder to k_newton_stream.
// S = storage, T = arithmetic.  fp32 state with fp64 arithmetic takes 8 B per lane (two problems): four sequential fp64
// problems per lane would leave the launch compute-bound with half the waves (measured at 1 Mi F4: 26 -> see DESIGN tuning log).
template <typename S, typename T> struct Vec16;
template <> struct Vec16<double, double> { using type = double __attribute__((ext_vector_type(2))); static constexpr int PER = 2; };
template <> struct Vec16<float, float> { using type = float __attribute__((ext_vector_type(4))); static constexpr int PER = 4; };
template <> struct Vec16<float, double> { using type = float __attribute__((ext_vector_type(2))); static constexpr int PER = 2; };

template <typename S, typename T, int VARIANT, bool ZV>
__global__ void __launch_bounds__(kBlock, RP_NEWTON_WAVES)
k_newton_stream16(S *__restrict__ base, size_t stride, int k, KParams<T> kp)
{
    constexpr int NC = CMap<VARIANT>::NC;
    constexpr int CB = 3 + NC;
    constexpr int NF = CB + 5;
    constexpr int PER = Vec16<S, T>::PER;
    using V = typename Vec16<S, T>::type;
    constexpr bool kInPlace = kStepInPlace<VARIANT, false, 0, NoDiag>;      // F3: the step's start waits in LDS
    __shared__ T s_backup[kInPlace ? (3 + NC) * kBlock : 1];
    [[maybe_unused]] LdsBackup<T> col = (LdsBackup<T>)&s_backup[(threadIdx.x >> 6) * (3 + NC) * 64 + (threadIdx.x & 63)];
    const size_t i = ((size_t)blockIdx.x * kBlock + threadIdx.x) * PER;
    V f[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q)
#ifndef RP_STREAM_PLAIN
        if (!(ZV && (q == CB + 1 || q == CB + 4))) f[q] = __builtin_nontemporal_load(reinterpret_cast<const V *>(base + (size_t)q * stride + i));
#else
        if (!(ZV && (q == CB + 1 || q == CB + 4))) f[q] = *reinterpret_cast<const V *>(base + (size_t)q * stride + i);
#endif
    // the segment lengths of all PER problems up front: the three position vectors are dead from here on
    T dx0[PER], dx1[PER];
#pragma unroll
    for (int c = 0; c < PER; ++c) {
        dx0[c] = (T)f[CB + 2][c] - (T)f[CB + 0][c];
        dx1[c] = (T)f[CB + 3][c] - (T)f[CB + 2][c];
    }
#pragma unroll
    for (int c = 0; c < PER; ++c) {
        T v = (T)f[0][c], t0 = (T)f[1][c], t1 = (T)f[2][c];
        T lam[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) lam[q] = (T)f[3 + q][c];
        Prob<T, ZV> pr;
        if constexpr (!ZV) {
            pr.v0 = (T)f[CB + 1][c];
            pr.v2 = (T)f[CB + 4][c];
        }
        pr.dx0 = dx0[c];
        pr.dx1 = dx1[c];
        int it = 0, steps_here = 0;
        uint32_t st = 0;
        bool still_open = false;
        run_lane<T, VARIANT, false, false, Prob<T, ZV>, S, kAffine<VARIANT>>(pr, kp, k, T(0), 0, v, t0, t1, lam, it, st, steps_here, still_open, LdsColumn<T>{col});
        f[0][c] = (S)v;
        f[1][c] = (S)t0;
        f[2][c] = (S)t1;
#pragma unroll
        for (int q = 0; q < NC; ++q) f[3 + q][c] = (S)lam[q];
    }
    size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) * PER;      // store addresses formed only now (see k_steps_chunks)
    asm volatile("" : "+v"(j));
#pragma unroll
#ifndef RP_STREAM_PLAIN
    for (int q = 0; q < CB; ++q) __builtin_nontemporal_store(f[q], reinterpret_cast<V *>(base + (size_t)q * stride + j));
#else
    for (int q = 0; q < CB; ++q) *reinterpret_cast<V *>(base + (size_t)q * stride + j) = f[q];
#endif
}

// ---------------------------------------------------------------------------------------
// Batch reduction: per problem the surrogate gap, ||r||^2 at p = gap / (10 m), converged bit.
// Block partials go to d_partials[4 * blockIdx]; k_reduce_final folds them (max, max, sum, sum).
__device__ __forceinline__ double nan_max(double a, double b) { return (a > b || a != a) ? a : b; }

template <typename S, typename T, int VARIANT>
__global__ void __launch_bounds__(kBlock)
k_reduce_partial(const S *__restrict__ base, size_t stride, size_t n, KParams<T> kp,
                 const uint32_t *__restrict__ status, double *__restrict__ partials)
{
    constexpr int NC = CMap<VARIANT>::NC;
    constexpr int CB = 3 + NC;
    double mr = 0.0, mg = -1.7976931348623157e308, nc = 0.0;
    bool any = false;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const S *f = base + i;
        T lam[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) lam[c] = (T)f[(3 + c) * stride];
        Prob<T> pr;
        const T p0 = (T)f[(CB + 0) * stride], p1 = (T)f[(CB + 2) * stride], p2 = (T)f[(CB + 3) * stride];
        pr.v0 = (T)f[(CB + 1) * stride];
        pr.v2 = (T)f[(CB + 4) * stride];
        pr.dx0 = p1 - p0;
        pr.dx1 = p2 - p1;
        const T v = (T)f[0];
        Acc<T> e;
        accel_values(pr, v, (T)f[1 * stride], (T)f[2 * stride], e);
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
    double steps = (double)counters[kShards + threadIdx.x];       // 64 threads, 64 shards
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) steps += __shfl_xor(steps, o);
    if (threadIdx.x == 0) {
        out4[0] = mr;
        out4[1] = mg;
        out4[2] = nc;
        out4[3] = steps + host_steps;
    }
}

// ---------------------------------------------------------------------------------------
// AoS (reference layout, double var[M] per problem) <-> SoA (compute type), staged through
// LDS so that both the global reads and the global writes are fully coalesced.  Rows are
// padded by one double: a lane reading its problem's field f then hits bank (34 l + 2 f) % 64,
// a 2-way conflict instead of the 32-way one of an unpadded 16-double row.
// `slot_of` (null = identity) maps problem index -> position in the batch: the AoS side is always in problem order.
template <typename T, int M>
__global__ void __launch_bounds__(kBlock)
k_aos_to_soa(const double *__restrict__ aos, T *__restrict__ base, size_t stride, size_t n, const uint32_t *__restrict__ slot_of)
{
    __shared__ double tile[kBlock * (M + 1)];
    const size_t first = (size_t)blockIdx.x * kBlock;
    const size_t count = (n - first < (size_t)kBlock) ? (n - first) : (size_t)kBlock;
    const double *src = aos + first * M;
    for (size_t j = threadIdx.x; j < count * M; j += kBlock) tile[(j / M) * (M + 1) + (j % M)] = src[j];
    __syncthreads();
    if (threadIdx.x < count) {
        const size_t at = slot_of ? (size_t)slot_of[first + threadIdx.x] : first + threadIdx.x;
#pragma unroll
        for (int f = 0; f < M; ++f) base[(size_t)f * stride + at] = (T)tile[threadIdx.x * (M + 1) + f];
    }
}

// problems [first, first + count) -> aos rows 0 .. count-1
template <typename T, int M>
__global__ void __launch_bounds__(kBlock)
k_soa_to_aos(const T *__restrict__ base, size_t stride, size_t first, size_t count, const uint32_t *__restrict__ slot_of, double *__restrict__ aos)
{
    __shared__ double tile[kBlock * (M + 1)];
    const size_t row0 = (size_t)blockIdx.x * kBlock;
    const size_t rows = (count - row0 < (size_t)kBlock) ? (count - row0) : (size_t)kBlock;
    if (threadIdx.x < rows) {
        const size_t i = first + row0 + threadIdx.x;
        const size_t at = slot_of ? (size_t)slot_of[i] : i;
#pragma unroll
        for (int f = 0; f < M; ++f) tile[threadIdx.x * (M + 1) + f] = (double)base[(size_t)f * stride + at];
    }
    __syncthreads();
    double *dst = aos + row0 * M;
    for (size_t j = threadIdx.x; j < rows * M; j += kBlock) dst[j] = tile[(j / M) * (M + 1) + (j % M)];
}

// per-problem words kept in batch order (iters, status, halving counts) -> problem order
__global__ void __launch_bounds__(kBlock)
k_gather_u32(const uint32_t *__restrict__ src, const uint32_t *__restrict__ slot_of, size_t n, uint32_t *__restrict__ dst)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[i] = src[slot_of[i]];
}

// The same two transposes for a WHOLE batch in scheduled order, walking POSITIONS: a block takes 256 consecutive positions, so
// the SoA side is coalesced as before, and the AoS side is whole rows -- problem prob_of[s]'s row of M doubles (128 B for F3:
// four whole sectors; 96 B for F4: three), M consecutive threads per row.  The forms above, which walk problems and reach the
// state through slot_of, touch one 32-byte sector per 8-byte field: 1,458 MB of traffic for the 268 MB a 1 Mi-problem read-back
// is made of (5.4 x, profiles/r3_hbm_traffic.json); they stay for identity order and for the ranges a host watches.
template <typename T, int M>
__global__ void __launch_bounds__(kBlock)
k_soa_to_aos_rows(const T *__restrict__ base, size_t stride, size_t n, const uint32_t *__restrict__ prob_of, double *__restrict__ aos)
{
    __shared__ double tile[kBlock * (M + 1)];
    __shared__ uint32_t s_prob[kBlock];
    const size_t row0 = (size_t)blockIdx.x * kBlock;
    const size_t rows = (n - row0 < (size_t)kBlock) ? (n - row0) : (size_t)kBlock;
    if (threadIdx.x < rows) {
        const size_t at = row0 + threadIdx.x;
        s_prob[threadIdx.x] = prob_of[at];
#pragma unroll
        for (int f = 0; f < M; ++f) tile[threadIdx.x * (M + 1) + f] = (double)ld_once(base + (size_t)f * stride + at);
    }
    __syncthreads();
    for (size_t j = threadIdx.x; j < rows * M; j += kBlock) {
        const size_t r = j / M, f = j % M;
        aos[(size_t)s_prob[r] * M + f] = tile[r * (M + 1) + f];      // plain stores: the 8-byte pieces of a sector meet in L2 (nontemporal ones do not: profiles/r4_solution_probe.log)
    }
}

template <typename T, int M>
__global__ void __launch_bounds__(kBlock)
k_aos_rows_to_soa(const double *__restrict__ aos, T *__restrict__ base, size_t stride, size_t n, const uint32_t *__restrict__ prob_of)
{
    __shared__ double tile[kBlock * (M + 1)];
    __shared__ uint32_t s_prob[kBlock];
    const size_t row0 = (size_t)blockIdx.x * kBlock;
    const size_t rows = (n - row0 < (size_t)kBlock) ? (n - row0) : (size_t)kBlock;
    if (threadIdx.x < rows) s_prob[threadIdx.x] = prob_of[row0 + threadIdx.x];
    __syncthreads();
    for (size_t j = threadIdx.x; j < rows * M; j += kBlock) {
        const size_t r = j / M, f = j % M;
        tile[r * (M + 1) + f] = __builtin_nontemporal_load(aos + (size_t)s_prob[r] * M + f);
    }
    __syncthreads();
    if (threadIdx.x < rows) {
        const size_t at = row0 + threadIdx.x;
#pragma unroll
        for (int f = 0; f < M; ++f) base[(size_t)f * stride + at] = (T)tile[threadIdx.x * (M + 1) + f];
    }
}

// rp_batch_solution_device: every problem's answer -- (vel1, duration0, duration1), iteration count, status word -- as a 32-byte
// record in PROBLEM order.  Walks positions: 28 B per problem read coalesced + the 4-byte inverse map, one whole sector written
// where prob_of says.  (k_solve_chunks writes the same records itself when a buffer is bound.)
template <typename S>
__global__ void __launch_bounds__(kBlock)
k_solution(const S *__restrict__ base, size_t stride, size_t n, const int32_t *__restrict__ iters, const uint32_t *__restrict__ status,
           const uint32_t *__restrict__ prob_of, int iters_add, Solution *__restrict__ out)
{
    const size_t s = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    const size_t prob = prob_of ? (size_t)prob_of[s] : s;
    store_solution(out + prob, (double)ld_once(base + s), (double)ld_once(base + stride + s), (double)ld_once(base + 2 * stride + s),
                   iters[s] + iters_add, status[s]);
}

// the same walking PROBLEMS: five gathered sectors read per problem, records written coalesced (A/B form, -DRP_SOLUTION_GATHER)
template <typename S>
__global__ void __launch_bounds__(kBlock)
k_solution_gather(const S *__restrict__ base, size_t stride, size_t n, const int32_t *__restrict__ iters, const uint32_t *__restrict__ status,
                  const uint32_t *__restrict__ slot_of, int iters_add, Solution *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const size_t s = slot_of[i];
    store_solution(out + i, (double)base[s], (double)base[stride + s], (double)base[2 * stride + s], iters[s] + iters_add, status[s]);
}

// Feasible start (build-defined): vel1 = 0, t_i = (3.5/sqrt 12) sqrt(6 |dX_i| / L), multipliers 1,
// vel0 = vel2 = 0, computed in double from the positions the batch holds in its own constant fields (put there, at each
// problem's position in the scheduled order, by schedule.hip) and stored in the batch's storage type.  This is what
// materialises a batch that has been given its problems, and what rp_batch_restart runs.
template <typename S, int VARIANT>
__global__ void __launch_bounds__(kBlock)
k_restart_feasible(S *__restrict__ base, size_t stride, size_t n, double limit)
{
    constexpr int NC = CMap<VARIANT>::NC;
    constexpr int CB = 3 + NC;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    S *f = base + i;
    const double p0 = (double)f[(CB + 0) * stride], p1 = (double)f[(CB + 2) * stride], p2 = (double)f[(CB + 3) * stride];
    const double scale = 3.5 / __builtin_sqrt(12.0);
    f[0 * stride] = S(0);
    f[1 * stride] = (S)(scale * __builtin_sqrt(6.0 * __builtin_fabs(p1 - p0) / limit));
    f[2 * stride] = (S)(scale * __builtin_sqrt(6.0 * __builtin_fabs(p2 - p1) / limit));
#pragma unroll
    for (int c = 0; c < NC; ++c) f[(3 + c) * stride] = S(1);
    f[(CB + 1) * stride] = S(0);
    f[(CB + 4) * stride] = S(0);
}

// The same start for a batch that has just been scheduled: positions from the 32-byte records the scheduling pass kept per
// problem (schedule.hip), found through prob_of, into the constant fields, the start computed from the STORED positions exactly as above, progress
// words cleared -- what k_solve_chunks<START> forms in registers, written out for every other consumer of the state.
template <typename S, int VARIANT>
__global__ void __launch_bounds__(kBlock)
k_start_from_records(S *__restrict__ base, size_t stride, size_t n, double limit, const StartRecord *__restrict__ records,
                     const uint32_t *__restrict__ prob_of, int32_t *__restrict__ iters, uint32_t *__restrict__ status)
{
    constexpr int NC = CMap<VARIANT>::NC;
    constexpr int CB = 3 + NC;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    typedef double v2 __attribute__((ext_vector_type(2)));
    const v2 *rec = reinterpret_cast<const v2 *>(records + prob_of[i]);
    const v2 ra = rec[0], rb = rec[1];
    const S s0 = (S)ra[0], s1 = (S)ra[1], s2 = (S)rb[0];
    const double p0 = (double)s0, p1 = (double)s1, p2 = (double)s2;
    const double scale = 3.5 / __builtin_sqrt(12.0);
    S *f = base + i;
    f[0 * stride] = S(0);
    f[1 * stride] = (S)(scale * __builtin_sqrt(6.0 * __builtin_fabs(p1 - p0) / limit));
    f[2 * stride] = (S)(scale * __builtin_sqrt(6.0 * __builtin_fabs(p2 - p1) / limit));
#pragma unroll
    for (int c = 0; c < NC; ++c) f[(3 + c) * stride] = S(1);
    f[(CB + 0) * stride] = s0;
    f[(CB + 1) * stride] = S(0);
    f[(CB + 2) * stride] = s1;
    f[(CB + 3) * stride] = s2;
    f[(CB + 4) * stride] = S(0);
    iters[i] = 0;
    status[i] = 0;
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
    if (i < 2 * kShards) counters[i] = 0;
}

__global__ void __launch_bounds__(64) k_zero_counter(unsigned long long *c) { c[threadIdx.x] = 0; }   // the open-lane shards

// ---------------------------------------------------------------------------------------
// moveTowardFeasibility (onedpath_ip.cpp:648-721 / onedpath2_ip.cpp:536-609), one lane per problem.
template <typename S, typename T, int VARIANT>
__global__ void __launch_bounds__(kBlock)
k_move_toward_feasibility(S *__restrict__ base, size_t stride, size_t n, KParams<T> kp)
{
    constexpr int NC = CMap<VARIANT>::NC;
    constexpr int CB = 3 + NC;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    S *f = base + i;
    Prob<T> pr;
    const T p0 = (T)f[(CB + 0) * stride], p1 = (T)f[(CB + 2) * stride], p2 = (T)f[(CB + 3) * stride];
    pr.v0 = (T)f[(CB + 1) * stride];
    pr.v2 = (T)f[(CB + 4) * stride];
    pr.dx0 = p1 - p0;
    pr.dx1 = p2 - p1;
    T v = (T)f[0], t0 = (T)f[1 * stride], t1 = (T)f[2 * stride];
    T dxv, dx0, dx1;
    if (feasibility_move<T, VARIANT>(pr.dx0, pr.dx1, pr.v0, pr.v2, v, t0, t1, kp.limit, dxv, dx0, dx1)) {
        f[0] = (S)(v + dxv);
        f[1 * stride] = (S)(t0 + dx0);
        f[2 * stride] = (S)(t1 + dx1);
    }
}

// Plot data: per problem 66 positions (drawSegment, onedpath_ip.cpp:1065-1088, 33 per segment) and 4 end accelerations
// (plotAcceleration, 1024-1027).  The launch moves 64 B of state in and 560 B out per problem: it has to be an HBM-write
// kernel.  A 256-thread block takes 128 problems: its first 128 threads read one problem each (eight fields through the slot
// map) and leave, per segment, the six numbers a sample is made of in LDS; then all threads write the block's 8,448 positions
// and 512 accelerations as consecutive elements (full coalesced segments), each from the constants of its problem.  The
// reference's divisions are multiplications by a refined reciprocal (rcp_: IEEE 1/x), one per segment instead of nine per
// sample; the parity test allows 1e-13.  (The first form -- one thread per output element, every thread reading the eight
// fields and dividing for itself -- ran at 0.17 of the HBM peak: it was bound by its 9 broadcast loads per wave.)
// (Walking POSITIONS instead -- coalesced field reads, each problem's 528-byte row scattered to where prob_of says -- was measured
// and is slower, 0.227 against 0.194 ms at 1 Mi problems: rows start on alternating 16-byte offsets, so every row ends in two
// partial sectors.  The gather through slot_of costs eight -- with zero end velocities, which are then not read, six --
// 32-byte sectors per problem on top of the 560 B written; that traffic is what the launch time is made of.)
constexpr int kSampleProblems = 128;
template <typename T, int VARIANT, bool ZV>
__global__ void __launch_bounds__(kBlock)
k_sample(const T *__restrict__ base, size_t stride, size_t first, size_t count, const uint32_t *__restrict__ slot_of,
         double *__restrict__ pos66, double *__restrict__ acc4)
{
    constexpr int CB = 3 + CMap<VARIANT>::NC;
    __shared__ double s_seg[2][6][kSampleProblems];      // per segment: x0, x1, va, acc0, jrk0, h / 32
    __shared__ double s_acc[4][kSampleProblems];
    const size_t p_first = (size_t)blockIdx.x * kSampleProblems;      // output row of the block's first problem
    const int here = (int)(count - p_first < (size_t)kSampleProblems ? count - p_first : (size_t)kSampleProblems);
    if (threadIdx.x < here) {
        const int q = threadIdx.x;
        const T *f = base + (slot_of ? (size_t)slot_of[first + p_first + q] : first + p_first + q);
        const double v1 = (double)f[0], t0 = (double)f[1 * stride], t1 = (double)f[2 * stride];
        const double p0 = (double)f[(CB + 0) * stride], p1 = (double)f[(CB + 2) * stride], p2 = (double)f[(CB + 3) * stride];
        const double v0 = ZV ? 0.0 : (double)f[(CB + 1) * stride], v2 = ZV ? 0.0 : (double)f[(CB + 4) * stride];
#pragma unroll
        for (int seg = 0; seg < 2; ++seg) {
            const double x0 = seg ? p1 : p0, x1 = seg ? p2 : p1, va = seg ? v1 : v0, vb = seg ? v2 : v1, h = seg ? t1 : t0;
            const double ih = rcp_<double>(h), ih2 = ih * ih;
            const double acc0 = (x1 - x0) * (6.0 * ih2) - (va * 4.0 + vb * 2.0) * ih;
            const double jrk0 = (vb - va) * (2.0 * ih2) - acc0 * (2.0 * ih);
            s_seg[seg][0][q] = x0; s_seg[seg][1][q] = x1; s_seg[seg][2][q] = va;
            s_seg[seg][3][q] = acc0; s_seg[seg][4][q] = jrk0; s_seg[seg][5][q] = h * 0.03125;
            // end accelerations of the segment (evalAccelInit / evalAccelFinal's formulas)
            s_acc[2 * seg][q] = ((x1 - x0) * 6.0 * ih + va * -4.0 + vb * -2.0) * ih;
            s_acc[2 * seg + 1][q] = ((x1 - x0) * -6.0 * ih + va * 2.0 + vb * 4.0) * ih;
        }
    }
    __syncthreads();
    // two consecutive positions per thread and trip (a problem's 66 are 33 pairs): 16-byte nontemporal stores, 1 KiB per wave
    typedef double v2 __attribute__((ext_vector_type(2)));
    auto position = [&](int q, int slot) -> double {
        const int seg = slot >= 33, j = slot - 33 * seg;
        if (j == 0) return s_seg[seg][0][q];
        if (j == 32) return s_seg[seg][1][q];
        const double t = s_seg[seg][5][q] * (double)j;      // h j / 32
        return s_seg[seg][0][q] + (s_seg[seg][2][q] + (s_seg[seg][3][q] + s_seg[seg][4][q] * (t * (1.0 / 3.0))) * (t * 0.5)) * t;
    };
    v2 *out_pos = reinterpret_cast<v2 *>(pos66 + p_first * 66);      // 16-byte aligned: 66 doubles per problem, 128 problems per block
    for (int pr = threadIdx.x; pr < here * 33; pr += kBlock) {
        const int q = pr / 33, pair = pr - q * 33;
        const v2 both = {position(q, 2 * pair), position(q, 2 * pair + 1)};
        __builtin_nontemporal_store(both, out_pos + pr);
    }
    double *out_acc = acc4 + p_first * 4;
    for (int o = threadIdx.x; o < here * 4; o += kBlock) out_acc[o] = s_acc[o & 3][o >> 2];
}

// The same plot data for a WHOLE scheduled batch from two problem-order records per problem (round 4): the positions the batch
// was given (StartRecord, kept by the scheduling pass) and the problem's solution record (k_solution writes them into a scratch
// first: 68 B per problem).  Both reads are coalesced whole sectors -- 64 B per problem where the gather through slot_of touches
// six 32-byte sectors for 48 B -- and the arithmetic and the stores are k_sample's (same bits).  Zero end velocities only, and
// only while the records are the batch's positions (rp_batch.cpp keeps the flag).
template <typename S>
__global__ void __launch_bounds__(kBlock)
k_sample_records(const StartRecord *__restrict__ records, const Solution *__restrict__ sol, size_t count,
                 double *__restrict__ pos66, double *__restrict__ acc4)
{
    __shared__ double s_seg[2][6][kSampleProblems];      // per segment: x0, x1, va, acc0, jrk0, h / 32
    __shared__ double s_acc[4][kSampleProblems];
    const size_t p_first = (size_t)blockIdx.x * kSampleProblems;
    const int here = (int)(count - p_first < (size_t)kSampleProblems ? count - p_first : (size_t)kSampleProblems);
    if (threadIdx.x < here) {
        const int q = threadIdx.x;
        typedef double v2 __attribute__((ext_vector_type(2)));
        const v2 *rec = reinterpret_cast<const v2 *>(records + p_first + q), *so = reinterpret_cast<const v2 *>(sol + p_first + q);
        const v2 ra = rec[0], rb = rec[1], sa = so[0], sb = so[1];
        const double p0 = (double)(S)ra[0], p1 = (double)(S)ra[1], p2 = (double)(S)rb[0];      // what the constant fields hold
        const double v1 = sa[0], t0 = sa[1], t1 = sb[0];
#pragma unroll
        for (int seg = 0; seg < 2; ++seg) {
            const double x0 = seg ? p1 : p0, x1 = seg ? p2 : p1, va = seg ? v1 : 0.0, vb = seg ? 0.0 : v1, h = seg ? t1 : t0;
            const double ih = rcp_<double>(h), ih2 = ih * ih;
            const double acc0 = (x1 - x0) * (6.0 * ih2) - (va * 4.0 + vb * 2.0) * ih;
            const double jrk0 = (vb - va) * (2.0 * ih2) - acc0 * (2.0 * ih);
            s_seg[seg][0][q] = x0; s_seg[seg][1][q] = x1; s_seg[seg][2][q] = va;
            s_seg[seg][3][q] = acc0; s_seg[seg][4][q] = jrk0; s_seg[seg][5][q] = h * 0.03125;
            s_acc[2 * seg][q] = ((x1 - x0) * 6.0 * ih + va * -4.0 + vb * -2.0) * ih;
            s_acc[2 * seg + 1][q] = ((x1 - x0) * -6.0 * ih + va * 2.0 + vb * 4.0) * ih;
        }
    }
    __syncthreads();
    typedef double v2 __attribute__((ext_vector_type(2)));
    auto position = [&](int q, int slot) -> double {
        const int seg = slot >= 33, j = slot - 33 * seg;
        if (j == 0) return s_seg[seg][0][q];
        if (j == 32) return s_seg[seg][1][q];
        const double t = s_seg[seg][5][q] * (double)j;      // h j / 32
        return s_seg[seg][0][q] + (s_seg[seg][2][q] + (s_seg[seg][3][q] + s_seg[seg][4][q] * (t * (1.0 / 3.0))) * (t * 0.5)) * t;
    };
    v2 *out_pos = reinterpret_cast<v2 *>(pos66 + p_first * 66);
    for (int pr = threadIdx.x; pr < here * 33; pr += kBlock) {
        const int q = pr / 33, pair = pr - q * 33;
        const v2 both = {position(q, 2 * pair), position(q, 2 * pair + 1)};
        __builtin_nontemporal_store(both, out_pos + pr);
    }
    double *out_acc = acc4 + p_first * 4;
    for (int o = threadIdx.x; o < here * 4; o += kBlock) out_acc[o] = s_acc[o & 3][o >> 2];
}

// printState's per-problem part for a (small) range of problems: the surrogate gap and, per constraint, what
// printConstraints shows (onedpath_ip.cpp:955-995, 1008-1010): error, gradient, the 3x3 second-derivative
// matrix and dot = (0,-1,-1).gradient.  Row layout per problem, in doubles:
//     [0] gap, then for constraint i at 1 + 14 i: error, deriv[3], second[3][3] (row-major), dot.
// Same device functions as the step (accel_values / accel_grads / accel_hess, c_value, c_grad), so what is
// printed is what the kernels step on.
template <typename S, typename T, int VARIANT>
__global__ void __launch_bounds__(kBlock)
k_constraint_table(const S *__restrict__ base, size_t stride, size_t first, size_t count, const uint32_t *__restrict__ slot_of,
                   KParams<T> kp, double *__restrict__ out)
{
    constexpr int NC = CMap<VARIANT>::NC;
    constexpr int CB = 3 + NC;
    constexpr int ROW = 1 + 14 * NC;
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= count) return;
    const S *f = base + (slot_of ? (size_t)slot_of[first + j] : first + j);
    Prob<T> pr;
    const T p0 = (T)f[(CB + 0) * stride], p1 = (T)f[(CB + 2) * stride], p2 = (T)f[(CB + 3) * stride];
    pr.v0 = (T)f[(CB + 1) * stride];
    pr.v2 = (T)f[(CB + 4) * stride];
    pr.dx0 = p1 - p0;
    pr.dx1 = p2 - p1;
    const T v = (T)f[0];
    T lam[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) lam[c] = (T)f[(3 + c) * stride];
    Acc<T> e;
    accel_values(pr, v, (T)f[1 * stride], (T)f[2 * stride], e);
    accel_grads(pr, v, e);
    T htt[4], htv[4];
    accel_hess(pr, v, e, htt, htv);
    double *o = out + j * ROW;
    o[0] = (double)duality_gap<T, VARIANT>(e, lam, kp.limit);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        T gv, gt;
        c_grad<T, VARIANT>(i, e, gv, gt);
        const int seg = c_segment<VARIANT>(i);
        T Htt, Htv;      // second derivatives of c_i: (t,t) and (t,v) = (v,t); everything else 0
        if constexpr (VARIANT == 3) {
            const int a = i >> 1;
            Htt = (i & 1) ? htt[a] : -htt[a];
            Htv = (i & 1) ? htv[a] : -htv[a];
        } else {      // (a^2 - L^2)/2; the (vel1,vel1) entry is never written by the reference (onedpath2_ip.cpp:446-448)
            Htt = fma_(e.gt[i], e.gt[i], e.a[i] * htt[i]);
            Htv = fma_(e.gt[i], acc_gv(e, i), e.a[i] * htv[i]);
        }
        double *r = o + 1 + 14 * i;
        r[0] = (double)c_value<T, VARIANT>(i, e, kp.limit);
        r[1] = (double)gv;
        r[2] = seg == 0 ? (double)gt : 0.0;
        r[3] = seg == 0 ? 0.0 : (double)gt;
        for (int q = 0; q < 9; ++q) r[4 + q] = 0.0;
        const int t = 1 + seg;      // index of this constraint's duration among (vel1, t0, t1)
        r[4 + 3 * t + t] = (double)Htt;
        r[4 + 3 * 0 + t] = (double)Htv;
        r[4 + 3 * t + 0] = (double)Htv;
        r[13] = -(r[2] + r[3]);     // obj = (0, -1, -1)
    }
}

inline unsigned grid_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

// dispatch on (dtype, variant): S = storage type in HBM, T = arithmetic type in registers
//   RP_DTYPE_F64        double / double
//   RP_DTYPE_F32        float  / float
//   RP_DTYPE_F32_STATE  float  / double   (fp32 state and traffic, fp64 arithmetic)
#define RP_DISPATCH_V(b, ...)                                             \
    do {                                                                  \
        if ((b).variant == 3) { constexpr int V = 3; __VA_ARGS__; }       \
        else                  { constexpr int V = 4; __VA_ARGS__; }       \
    } while (0)
#define RP_DISPATCH(b, ...)                                                                              \
    do {                                                                                                 \
        if ((b).dtype == 0)      { using S [[maybe_unused]] = double; using T [[maybe_unused]] = double; RP_DISPATCH_V(b, __VA_ARGS__); } \
        else if ((b).dtype == 1) { using S [[maybe_unused]] = float;  using T [[maybe_unused]] = float;  RP_DISPATCH_V(b, __VA_ARGS__); } \
        else                     { using S [[maybe_unused]] = float;  using T [[maybe_unused]] = double; RP_DISPATCH_V(b, __VA_ARGS__); } \
    } while (0)

// dispatch on the dtype alone (S, T)
#define RP_DISPATCH_ST(b, ...)                                                                           \
    do {                                                                                                 \
        if ((b).dtype == 0)      { using S [[maybe_unused]] = double; using T [[maybe_unused]] = double; __VA_ARGS__; } \
        else if ((b).dtype == 1) { using S [[maybe_unused]] = float;  using T [[maybe_unused]] = float;  __VA_ARGS__; } \
        else                     { using S [[maybe_unused]] = float;  using T [[maybe_unused]] = double; __VA_ARGS__; } \
    } while (0)

}  // namespace

// the three Newton kernels are also instantiated for "end velocities are zero" (BatchView::zero_end_vel)
#define RP_DISPATCH_Z(b, ...)                                             \
    do {                                                                  \
        if ((b).zero_end_vel) { constexpr bool Z = true; RP_DISPATCH(b, __VA_ARGS__); }   \
        else                  { constexpr bool Z = false; RP_DISPATCH(b, __VA_ARGS__); }  \
    } while (0)

// mu_mode 1 (centring by trial) exists for double arithmetic (dtypes f64 and f32-state) in the one-problem-per-lane
// kernels; rp_batch_set_params refuses it for pure fp32.
#define RP_DISPATCH_MU1(b, ...)                                                                          \
    do {                                                                                                 \
        using T [[maybe_unused]] = double;                                                               \
        if ((b).dtype == 0) { using S [[maybe_unused]] = double; RP_DISPATCH_V(b, __VA_ARGS__); }          \
        else                { using S [[maybe_unused]] = float;  RP_DISPATCH_V(b, __VA_ARGS__); }          \
    } while (0)
#define RP_DISPATCH_MU1_Z(b, ...)                                         \
    do {                                                                  \
        if ((b).zero_end_vel) { constexpr bool Z = true; RP_DISPATCH_MU1(b, __VA_ARGS__); }   \
        else                  { constexpr bool Z = false; RP_DISPATCH_MU1(b, __VA_ARGS__); }  \
    } while (0)

hipError_t launch_steps(const BatchView &b, const HostParams &hp, int k, hipStream_t stream)
{
    if (k < 0 || b.n == 0) return hipSuccess;     // k == 0: load/store only (bandwidth probe, see rp_batch_step)
    if (hp.mu_mode == 1) {
        unsigned grid = grid_for(b.n);
        if (grid > 2048u) grid = 2048u;
        RP_DISPATCH_MU1_Z(b, hipLaunchKernelGGL((k_newton_stream<S, T, V, Z, 1>), dim3(grid), dim3(kBlock), 0, stream,
                                                 (S *)b.base, b.stride, b.n, k, make_kparams<T>(hp, V)));
        return hipGetLastError();
    }
    // k = 1 is memory-bound (200 B per problem cross HBM for one step's arithmetic): the 16-byte streaming kernel.  From k = 2 on
    // the arithmetic dominates and k_steps_chunks runs (one problem per lane, no second register set, 152-158 VGPRs: 3 waves per
    // SIMD) at every batch size -- measured at 1 Mi problems: k = 1 5.3 TB/s (stream16) against 5.1, k = 2 0.042 ms (chunks)
    // against 0.047 (profiles/r3_k1_ab_probe.log); round 2 had drawn the line at k = 3 against the LDS-tiled form of the day.
    // The shipped library reads nothing from the environment: the knobs below exist in tuning builds only (-DRP_TUNING, what
    // the probes under profiles/probes build), where they override the launch shape for A/B runs.
#ifdef RP_TUNING
    static const char *grid_env = getenv("RP_STREAM_GRID");     // forces the streaming kernel
    static const int chunks_from = getenv("RP_CHUNKS_FROM_K") ? atoi(getenv("RP_CHUNKS_FROM_K")) : 2;      // smallest k that takes k_steps_chunks
    static const bool scalar_only = getenv("RP_STREAM_SCALAR") != nullptr;      // everything through the 8-byte prefetching kernel
#else
    constexpr const char *grid_env = nullptr;
    constexpr int chunks_from = 2;
    constexpr bool scalar_only = false;
#endif
    if (k >= chunks_from && k >= 1 && !grid_env) {
        // F3 below three waves per SIMD (3 x 1,024 SIMDs x 64 lanes): the instantiation whose step keeps its start in registers
#ifdef RP_TUNING
        static const size_t reg_column_upto = getenv("RP_REG_COLUMN_UPTO") ? (size_t)atol(getenv("RP_REG_COLUMN_UPTO")) : (size_t)3 * 1024 * 64;      // A/B: where the register column ends
#else
        constexpr size_t reg_column_upto = (size_t)3 * 1024 * 64;
#endif
        if (b.variant == 3 && b.n <= reg_column_upto) {
            constexpr int V3 = 3;
#ifdef RP_TUNING
            static const unsigned lanes_env = getenv("RP_LANES_PER_WAVE") ? (unsigned)atoi(getenv("RP_LANES_PER_WAVE")) : 64u;      // A/B: part-filled waves
            if (lanes_env < 1u || lanes_env > 64u) return hipErrorInvalidValue;      // (0 would divide by zero below, > 64 would leave problems unprocessed: ADVICE r5)
            const unsigned lanes = lanes_env;
#else
            constexpr unsigned lanes = 64u;
#endif
            const dim3 grid((unsigned)((b.n + lanes - 1) / lanes));
            if (b.zero_end_vel) { constexpr bool Z = true;  RP_DISPATCH_ST(b, hipLaunchKernelGGL((k_steps_chunks<S, T, V3, Z, true>), grid, dim3(kChunkBlock), 0, stream, (S *)b.base, b.stride, b.n, k, make_kparams<T>(hp, V3), lanes)); }
            else                { constexpr bool Z = false; RP_DISPATCH_ST(b, hipLaunchKernelGGL((k_steps_chunks<S, T, V3, Z, true>), grid, dim3(kChunkBlock), 0, stream, (S *)b.base, b.stride, b.n, k, make_kparams<T>(hp, V3), lanes)); }
            return hipGetLastError();
        }
        RP_DISPATCH_Z(b, hipLaunchKernelGGL((k_steps_chunks<S, T, V, Z>), dim3((unsigned)((b.n + 63) / 64)), dim3(kChunkBlock), 0, stream,
                                             (S *)b.base, b.stride, b.n, k, make_kparams<T>(hp, V), 64u));
        return hipGetLastError();
    }
    // k = 1: memory-bound.  Full blocks of 256 lanes x 16 B go through k_newton_stream16; the ragged remainder (and, in a
    // tuning build under RP_STREAM_SCALAR=1, everything: the A/B switch of the tuning log) through the 8-byte prefetching kernel.
    const size_t per_block = (size_t)kBlock * (b.dtype == 1 ? 4 : 2);      // problems per lane: Vec16<S, T>::PER
    const size_t nfull = (scalar_only || grid_env || k > 2) ? 0 : b.n / per_block * per_block;      // (k > 2 only under RP_STREAM_GRID)
    if (nfull > 0)
        RP_DISPATCH_Z(b, hipLaunchKernelGGL((k_newton_stream16<S, T, V, Z>), dim3((unsigned)(nfull / per_block)), dim3(kBlock), 0, stream,
                                             (S *)b.base, b.stride, k, make_kparams<T>(hp, V)));
    if (nfull < b.n) {
        BatchView w = b;
        w.base = (char *)b.base + nfull * storage_size(b.dtype);
        w.n = b.n - nfull;
        const unsigned cap = grid_env ? (unsigned)atoi(grid_env) : 512u;
        unsigned grid = grid_for(w.n);
        if (grid > cap) grid = cap;
        RP_DISPATCH_Z(w, hipLaunchKernelGGL((k_newton_stream<S, T, V, Z>), dim3(grid), dim3(kBlock), 0, stream,
                                             (S *)w.base, w.stride, w.n, k, make_kparams<T>(hp, V)));
    }
    return hipGetLastError();
}

hipError_t launch_steps_counted(const BatchView &b, const HostParams &hp, int k, uint32_t *d_nfeas, uint32_t *d_nresid, hipStream_t stream)
{
    RP_DISPATCH(b, hipLaunchKernelGGL((k_newton_counted<S, T, V>), dim3(grid_for(b.n)), dim3(kBlock), 0, stream,
                                       (S *)b.base, b.stride, b.n, k, make_kparams<T>(hp, V), d_nfeas, d_nresid));
    return hipGetLastError();
}

// the gated solve: up to k gated steps per problem, one 64-problem chunk of the scheduled order per single-wave block
static hipError_t launch_chunks(const BatchView &b, const HostParams &hp, int k, double gap_tol, int max_iter, bool from_start, hipStream_t stream)
{
    const dim3 grid((unsigned)((b.n + 63) / 64)), block(64);
#ifdef RP_TUNING
    // tuning builds only: dynamic LDS per block, to cap the waves per SIMD -- passed to whichever instantiation the batch selects
    static const unsigned lds_pad = getenv("RP_GATED_LDS_PAD") ? (unsigned)atoi(getenv("RP_GATED_LDS_PAD")) : 0u;
#else
    constexpr unsigned lds_pad = 0u;
#endif
    if (from_start) {      // reference mode, no stall detector, zero end velocities: rp_batch.cpp only asks for this form then
        if (hp.mu_mode != 0 || hp.stall_window > 0 || !b.zero_end_vel || !b.records) return hipErrorInvalidValue;
        RP_DISPATCH(b, hipLaunchKernelGGL((k_solve_chunks<S, T, V, false, true, 0, true>), grid, block, lds_pad, stream, (S *)b.base, b.stride, b.n, k,
                                           make_kparams<T>(hp, V), (T)gap_tol, max_iter, b.iters, b.status, b.counters, b.records, b.prob_of, hp.accel_limit, b.solution, b.scheduled ? (const uint32_t *)b.prob_of : nullptr, b.iters_add));
    } else if (hp.mu_mode == 1)
        RP_DISPATCH_MU1_Z(b, hipLaunchKernelGGL((k_solve_chunks<S, T, V, true, Z, 1>), grid, block, lds_pad, stream, (S *)b.base, b.stride, b.n, k,
                                                 make_kparams<T>(hp, V), (T)gap_tol, max_iter, b.iters, b.status, b.counters, b.records, b.prob_of, hp.accel_limit, b.solution, b.scheduled ? (const uint32_t *)b.prob_of : nullptr, b.iters_add));
    else if (hp.stall_window > 0)
        RP_DISPATCH_Z(b, hipLaunchKernelGGL((k_solve_chunks<S, T, V, true, Z>), grid, block, lds_pad, stream, (S *)b.base, b.stride, b.n, k,
                                             make_kparams<T>(hp, V), (T)gap_tol, max_iter, b.iters, b.status, b.counters, b.records, b.prob_of, hp.accel_limit, b.solution, b.scheduled ? (const uint32_t *)b.prob_of : nullptr, b.iters_add));
    else
        RP_DISPATCH_Z(b, hipLaunchKernelGGL((k_solve_chunks<S, T, V, false, Z>), grid, block, lds_pad, stream, (S *)b.base, b.stride, b.n, k,
                                             make_kparams<T>(hp, V), (T)gap_tol, max_iter, b.iters, b.status, b.counters, b.records, b.prob_of, hp.accel_limit, b.solution, b.scheduled ? (const uint32_t *)b.prob_of : nullptr, b.iters_add));
    return hipGetLastError();
}

hipError_t launch_solve_fused(const BatchView &b, const HostParams &hp, double gap_tol, int max_iter, bool from_start, hipStream_t stream)
{
    if (b.n == 0) return hipSuccess;
    // (the open-lane shards are not zeroed here: only the host-polled loop reads them, and launch_solve zeroes them itself)
    return launch_chunks(b, hp, max_iter > 0 ? max_iter : 1, gap_tol, max_iter, from_start, stream);
}

hipError_t launch_solve_rounds(const BatchView &b, const HostParams &hp, double gap_tol, int max_iter, int rounds, int lanes, int patience, hipStream_t stream)
{
    if (b.n == 0) return hipSuccess;
    if ((rounds > 1 && !b.lists) || hp.mu_mode != 0 || hp.stall_window > 0 || rounds < 1 || rounds > 8 || lanes < 1 || lanes > 48 || max_iter <= 0) return hipErrorInvalidValue;
    uint32_t *list[2] = {b.lists, b.lists ? b.lists + b.n : nullptr}, *counts = b.lists ? b.lists + 2 * b.n : nullptr;      // counts[r]: entries of the list round r wrote
    hipError_t e = rounds > 1 ? hipMemsetAsync(counts, 0, 16 * sizeof(uint32_t), stream) : hipSuccess;
    if (e != hipSuccess) return e;
    size_t bound = b.n;      // the most problems this round can be given
    for (int r = 0; r < rounds; ++r) {
        const bool last = r == rounds - 1;
        const unsigned waves = (unsigned)((bound + 63) / 64);
        const uint32_t *in = r == 0 ? nullptr : list[(r - 1) & 1], *in_count = r == 0 ? nullptr : counts + (r - 1);
        uint32_t *out = last ? nullptr : list[r & 1], *out_count = last ? nullptr : counts + r;
        RP_DISPATCH_Z(b, {
            KParams<T> kp = make_kparams<T>(hp, V);
            kp.handoff_lanes = last ? 0 : lanes;
            kp.handoff_patience = patience;
            hipLaunchKernelGGL((k_solve_chunks<S, T, V, false, Z, 0, false, true>), dim3(waves), dim3(64), 0, stream, (S *)b.base, b.stride, b.n, max_iter,
                               kp, (T)gap_tol, max_iter, b.iters, b.status, b.counters, b.records, b.prob_of, hp.accel_limit, b.solution,
                               b.scheduled ? (const uint32_t *)b.prob_of : nullptr, b.iters_add, in, in_count, out, out_count);
        });
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        bound = (size_t)waves * (size_t)lanes;      // every wave hands off at most `lanes` of its lanes
        if (bound > b.n) bound = b.n;
    }
    return hipSuccess;
}

hipError_t launch_solve(const BatchView &b, const HostParams &hp, int k, double gap_tol, int max_iter, hipStream_t stream)
{
    if (b.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_zero_counter, dim3(1), dim3(kShards), 0, stream, b.counters);
    return launch_chunks(b, hp, k, gap_tol, max_iter, false, stream);
}

hipError_t launch_gather_u32(const BatchView &b, const uint32_t *d_src, uint32_t *d_dst, hipStream_t stream)
{
    if (b.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_gather_u32, dim3(grid_for(b.n)), dim3(kBlock), 0, stream, d_src, (const uint32_t *)b.slot_of, b.n, d_dst);
    return hipGetLastError();
}

hipError_t launch_reduce(const BatchView &b, const HostParams &hp, double host_steps, double *d_partials,
                         double *d_out4, hipStream_t stream)
{
    unsigned blocks = grid_for(b.n);
    if (blocks > 1024) blocks = 1024;
    RP_DISPATCH(b, hipLaunchKernelGGL((k_reduce_partial<S, T, V>), dim3(blocks), dim3(kBlock), 0, stream,
                                       (const S *)b.base, b.stride, b.n, make_kparams<T>(hp, V), b.status, d_partials));
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(64), 0, stream, d_partials, (int)blocks, b.counters, host_steps, d_out4);
    return hipGetLastError();
}

static const uint32_t *slots(const BatchView &b) { return b.scheduled ? b.slot_of : nullptr; }

hipError_t launch_solution(const BatchView &b, Solution *d_out, hipStream_t stream)
{
    if (b.n == 0) return hipSuccess;
    const uint32_t *inv = b.scheduled ? (const uint32_t *)b.prob_of : nullptr;
#ifdef RP_SOLUTION_GATHER
    if (b.scheduled) {
        if (b.dtype == 0) hipLaunchKernelGGL((k_solution_gather<double>), dim3(grid_for(b.n)), dim3(kBlock), 0, stream, (const double *)b.base, b.stride, b.n, b.iters, b.status, (const uint32_t *)b.slot_of, b.iters_add, d_out);
        else              hipLaunchKernelGGL((k_solution_gather<float>), dim3(grid_for(b.n)), dim3(kBlock), 0, stream, (const float *)b.base, b.stride, b.n, b.iters, b.status, (const uint32_t *)b.slot_of, b.iters_add, d_out);
        return hipGetLastError();
    }
#endif
    if (b.dtype == 0) hipLaunchKernelGGL((k_solution<double>), dim3(grid_for(b.n)), dim3(kBlock), 0, stream, (const double *)b.base, b.stride, b.n, b.iters, b.status, inv, b.iters_add, d_out);
    else              hipLaunchKernelGGL((k_solution<float>), dim3(grid_for(b.n)), dim3(kBlock), 0, stream, (const float *)b.base, b.stride, b.n, b.iters, b.status, inv, b.iters_add, d_out);
    return hipGetLastError();
}

hipError_t launch_aos_to_soa(const BatchView &b, const double *d_aos, hipStream_t stream)
{
    const dim3 g(grid_for(b.n)), t(kBlock);
    if (b.scheduled) {      // whole rows in, walking positions
        const uint32_t *inv = (const uint32_t *)b.prob_of;
        if (b.dtype == 0) {
            if (b.variant == 3) hipLaunchKernelGGL((k_aos_rows_to_soa<double, 16>), g, t, 0, stream, d_aos, (double *)b.base, b.stride, b.n, inv);
            else                hipLaunchKernelGGL((k_aos_rows_to_soa<double, 12>), g, t, 0, stream, d_aos, (double *)b.base, b.stride, b.n, inv);
        } else {
            if (b.variant == 3) hipLaunchKernelGGL((k_aos_rows_to_soa<float, 16>), g, t, 0, stream, d_aos, (float *)b.base, b.stride, b.n, inv);
            else                hipLaunchKernelGGL((k_aos_rows_to_soa<float, 12>), g, t, 0, stream, d_aos, (float *)b.base, b.stride, b.n, inv);
        }
        return hipGetLastError();
    }
    if (b.dtype == 0) {      // storage type only: dtype 1 and 2 both keep floats
        if (b.variant == 3) hipLaunchKernelGGL((k_aos_to_soa<double, 16>), g, t, 0, stream, d_aos, (double *)b.base, b.stride, b.n, slots(b));
        else                hipLaunchKernelGGL((k_aos_to_soa<double, 12>), g, t, 0, stream, d_aos, (double *)b.base, b.stride, b.n, slots(b));
    } else {
        if (b.variant == 3) hipLaunchKernelGGL((k_aos_to_soa<float, 16>), g, t, 0, stream, d_aos, (float *)b.base, b.stride, b.n, slots(b));
        else                hipLaunchKernelGGL((k_aos_to_soa<float, 12>), g, t, 0, stream, d_aos, (float *)b.base, b.stride, b.n, slots(b));
    }
    return hipGetLastError();
}

hipError_t launch_soa_to_aos_range(const BatchView &b, size_t first, size_t count, double *d_aos, hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    const dim3 g(grid_for(count)), t(kBlock);
    if (b.dtype == 0) {
        if (b.variant == 3) hipLaunchKernelGGL((k_soa_to_aos<double, 16>), g, t, 0, stream, (const double *)b.base, b.stride, first, count, slots(b), d_aos);
        else                hipLaunchKernelGGL((k_soa_to_aos<double, 12>), g, t, 0, stream, (const double *)b.base, b.stride, first, count, slots(b), d_aos);
    } else {
        if (b.variant == 3) hipLaunchKernelGGL((k_soa_to_aos<float, 16>), g, t, 0, stream, (const float *)b.base, b.stride, first, count, slots(b), d_aos);
        else                hipLaunchKernelGGL((k_soa_to_aos<float, 12>), g, t, 0, stream, (const float *)b.base, b.stride, first, count, slots(b), d_aos);
    }
    return hipGetLastError();
}

hipError_t launch_soa_to_aos(const BatchView &b, double *d_aos, hipStream_t stream)
{
    if (!b.scheduled || b.n == 0) return launch_soa_to_aos_range(b, 0, b.n, d_aos, stream);
    const dim3 g(grid_for(b.n)), t(kBlock);      // whole rows out, walking positions
    const uint32_t *inv = (const uint32_t *)b.prob_of;
    if (b.dtype == 0) {
        if (b.variant == 3) hipLaunchKernelGGL((k_soa_to_aos_rows<double, 16>), g, t, 0, stream, (const double *)b.base, b.stride, b.n, inv, d_aos);
        else                hipLaunchKernelGGL((k_soa_to_aos_rows<double, 12>), g, t, 0, stream, (const double *)b.base, b.stride, b.n, inv, d_aos);
    } else {
        if (b.variant == 3) hipLaunchKernelGGL((k_soa_to_aos_rows<float, 16>), g, t, 0, stream, (const float *)b.base, b.stride, b.n, inv, d_aos);
        else                hipLaunchKernelGGL((k_soa_to_aos_rows<float, 12>), g, t, 0, stream, (const float *)b.base, b.stride, b.n, inv, d_aos);
    }
    return hipGetLastError();
}

hipError_t launch_restart_feasible(const BatchView &b, const HostParams &hp, hipStream_t stream)
{
    RP_DISPATCH(b, hipLaunchKernelGGL((k_restart_feasible<S, V>), dim3(grid_for(b.n)), dim3(kBlock), 0, stream,
                                       (S *)b.base, b.stride, b.n, hp.accel_limit));
    return hipGetLastError();
}

hipError_t launch_start_from_records(const BatchView &b, const HostParams &hp, hipStream_t stream)
{
    if (!b.records) return hipErrorInvalidValue;
    RP_DISPATCH(b, hipLaunchKernelGGL((k_start_from_records<S, V>), dim3(grid_for(b.n)), dim3(kBlock), 0, stream,
                                       (S *)b.base, b.stride, b.n, hp.accel_limit, (const StartRecord *)b.records, (const uint32_t *)b.prob_of, b.iters, b.status));
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
    // Always in double, whatever the batch's arithmetic type: the move squares the conditioning of the constraint
    // gradients (Gram matrix), which single precision cannot carry (measured: 10 % of fp32 moves off by > 6e-3), and
    // it is a one-off between solves, not the hot path.  An fp32 state gets the fp64 move rounded to fp32.
    RP_DISPATCH(b, hipLaunchKernelGGL((k_move_toward_feasibility<S, double, V>), dim3(grid_for(b.n)), dim3(kBlock), 0, stream,
                                       (S *)b.base, b.stride, b.n, make_kparams<double>(hp, V)));
    return hipGetLastError();
}

// ---- read-backs by problem index: a range [first, first + count) of problems, wherever they lie in the batch ----
hipError_t launch_sample_range(const BatchView &b, size_t first, size_t count, double *d_pos66, double *d_acc4, hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    const dim3 grid((unsigned)((count + kSampleProblems - 1) / kSampleProblems));
    if (b.zero_end_vel) {
        RP_DISPATCH(b, hipLaunchKernelGGL((k_sample<S, V, true>), grid, dim3(kBlock), 0, stream,
                                           (const S *)b.base, b.stride, first, count, slots(b), d_pos66, d_acc4));
    } else {
        RP_DISPATCH(b, hipLaunchKernelGGL((k_sample<S, V, false>), grid, dim3(kBlock), 0, stream,
                                           (const S *)b.base, b.stride, first, count, slots(b), d_pos66, d_acc4));
    }
    return hipGetLastError();
}

hipError_t launch_sample(const BatchView &b, double *d_pos66, double *d_acc4, hipStream_t stream) { return launch_sample_range(b, 0, b.n, d_pos66, d_acc4, stream); }

hipError_t launch_sample_from_records(const BatchView &b, Solution *d_solution_scratch, double *d_pos66, double *d_acc4, hipStream_t stream)
{
    if (b.n == 0) return hipSuccess;
    if (!b.scheduled || !b.zero_end_vel || !b.records) return hipErrorInvalidValue;
    hipError_t e = launch_solution(b, d_solution_scratch, stream);      // every problem's (vel1, duration0, duration1) in problem order
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)((b.n + kSampleProblems - 1) / kSampleProblems));
    if (b.dtype == 0) hipLaunchKernelGGL((k_sample_records<double>), grid, dim3(kBlock), 0, stream, (const StartRecord *)b.records, (const Solution *)d_solution_scratch, b.n, d_pos66, d_acc4);
    else              hipLaunchKernelGGL((k_sample_records<float>), grid, dim3(kBlock), 0, stream, (const StartRecord *)b.records, (const Solution *)d_solution_scratch, b.n, d_pos66, d_acc4);
    return hipGetLastError();
}

hipError_t launch_constraint_table(const BatchView &b, const HostParams &hp, size_t first, size_t count, double *d_rows, hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    RP_DISPATCH(b, hipLaunchKernelGGL((k_constraint_table<S, T, V>), dim3(grid_for(count)), dim3(kBlock), 0, stream,
                                       (const S *)b.base, b.stride, first, count, slots(b), make_kparams<T>(hp, V), d_rows));
    return hipGetLastError();
}

}  // namespace rp
