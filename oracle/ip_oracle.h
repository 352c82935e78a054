/*
 * ip_oracle.h -- CPU restatement of rocket-path's interior-point Newton step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call it, and there only as the checker / reported CPU baseline.
 *
 * What it restates (all citations into /root/reference):
 *   F3  onedpath_ip.cpp:372-452   cubic-segment end accelerations + derivatives
 *       onedpath_ip.cpp:454-636   8 constraints  c_i = -/+a - L, gradients, Hessians
 *       onedpath_ip.cpp:723-808   trajectoryStep / constraintsSatisfied / residual /
 *                                 residualNorm / surrogateDualityGap
 *       onedpath_ip.cpp:810-953   moveInteriorPoint (11x11 KKT, QR solve, 3 backtracks)
 *       onedpath_ip.cpp:648-721   moveTowardFeasibility (the "Space" key)
 *       onedpath_ip.cpp:177-228   initStuck / initDefault
 *       onedpath_ip.cpp:1015-1088 plotAcceleration / drawSegment sample values
 *   F4  onedpath2_ip.cpp:414-524  4 constraints c_i = (a^2 - L^2)/2 (incl. the missing
 *                                 (v,v) Hessian entry), 698-841 moveInteriorPoint (7x7)
 *   QR  libs/eigen/Eigen/src/QR/ColPivHouseholderQR.h:480-611,
 *       libs/eigen/Eigen/src/Householder/Householder.h:65-131
 *
 * Parity pin (pinned): the reference's own hot-path functions, compiled in the build container from the
 * files where they lie -- `make -C oracle ref` pipes the GL-free line ranges of onedpath_ip.cpp /
 * onedpath2_ip.cpp into the compiler (real headers, the vendored Eigen 3.3.0, nothing stood in for;
 * ref_build/ref_hotpath.cpp) -> oracle/_ref/libref_hotpath.so.  tests/test_oracle_reference.py holds
 * this restatement to it BIT FOR BIT (model functions, moveInteriorPoint with the reference's Eigen QR
 * doing the solve, gated solves, moveTowardFeasibility, every committed fixture); the survey's
 * known-answer vectors (tests/golden/survey_kat.json) are a second, independent record
 * of the same outputs.  See DESIGN.md "Oracle".
 *
 * Build: plain C99, no FMA contraction (-ffp-contract=off, no -march flags), so the
 * arithmetic is the reference's x86-64 SSE2 arithmetic operation for operation.
 */
#ifndef RP_IP_ORACLE_H
#define RP_IP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Index layout of the F3 state, identical to enum V (onedpath_ip.cpp:15-43). */
enum {
    ORC3_VEL1 = 0, ORC3_DUR0 = 1, ORC3_DUR1 = 2,
    ORC3_LAM0 = 3,                      /* 8 multipliers: 3..10 */
    ORC3_POS0 = 11, ORC3_VEL0 = 12, ORC3_POS1 = 13, ORC3_POS2 = 14, ORC3_VEL2 = 15,
    ORC3_M = 16
};
/* Index layout of the F4 state, identical to enum V2 (onedpath2_ip.cpp:15-39). */
enum {
    ORC4_VEL1 = 0, ORC4_DUR0 = 1, ORC4_DUR1 = 2,
    ORC4_LAM0 = 3,                      /* 4 multipliers: 3..6 */
    ORC4_POS0 = 7, ORC4_VEL0 = 8, ORC4_POS1 = 9, ORC4_POS2 = 10, ORC4_VEL2 = 11,
    ORC4_M = 12
};

#define ORC_VARIANT_F3 3
#define ORC_VARIANT_F4 4

/* Per-step diagnostics (not in the reference; the counts the survey probed by hand). */
typedef struct {
    int feas_halvings;    /* s *= 0.5 executions at onedpath_ip.cpp:927 */
    int resid_halvings;   /* s *= 0.5 executions at onedpath_ip.cpp:944 */
    int nonzero_pivots;   /* ColPivHouseholderQR::nonzeroPivots() of the KKT matrix */
    double step_scale;    /* final s */
    double perturbation;  /* p = gap / (10 m), onedpath_ip.cpp:812 */
} orc_step_info;

/* ---- spline end accelerations (identical in both TUs) ---- */
void orc_accel_init(double x0, double v0, double x1, double v1, double t,
                    double *a, double *dAdT, double *dAdV0, double *dAdV1);
void orc_accel_init_2nd(double x0, double v0, double x1, double v1, double t,
                        double *sTT, double *sTV0, double *sTV1);
void orc_accel_final(double x0, double v0, double x1, double v1, double t,
                     double *a, double *dAdT, double *dAdV0, double *dAdV1);
void orc_accel_final_2nd(double x0, double v0, double x1, double v1, double t,
                         double *sTT, double *sTV0, double *sTV1);

/* ---- per-variant pieces; `var` has ORC3_M / ORC4_M doubles ---- */
int    orc_num_constraints(int variant);             /* 8 / 4 */
int    orc_state_len(int variant);                   /* 16 / 12 */
void   orc_constraint(int variant, int i, const double *var, double *err, double grad[3]);
void   orc_constraint_hess(int variant, int i, const double *var, double H[9]); /* row-major 3x3 */
double orc_gap(int variant, const double *var);
void   orc_residual(int variant, const double *var, double perturbation, double *r /* 3+m */);
double orc_residual_norm(int variant, const double *var, double perturbation);
int    orc_constraints_satisfied(int variant, const double *var);
void   orc_kkt(int variant, const double *var, double *m_colmajor /* (3+m)^2 */, double *r /* 3+m */,
               double *perturbation);
void   orc_step(int variant, double *var, orc_step_info *info /* may be NULL */);
/* step with the Newton direction returned (for solver-level comparisons) */
void   orc_step_dir(int variant, double *var, double *d /* 3+m */, orc_step_info *info);
/* same step with the linear solve delegated (tests pass oracle/_ref's real Eigen QR here);
   solver(n, A_colmajor, b, x, force_dynamic) returns nonzero pivots; NULL = orc_colpiv_qr_solve */
typedef int (*orc_qr_solver)(int n, const double *A, const double *b, double *x, int force_dynamic);
void   orc_step_ex(int variant, double *var, double *d, orc_step_info *info, orc_qr_solver solver);
/* the step with the reference's two compiled-in line-search constants (backtrack factor 0.5, 100 halvings per loop) as
   parameters -- the oracle for rp_params.backtrack / max_backtracks; (0.5, 100) is orc_step */
void   orc_step_params(int variant, double *var, orc_step_info *info, double backtrack, int max_bt);
void   orc_batch_steps_params(int variant, size_t n, double *aos, int k, int threads, double backtrack, int max_bt);
void   orc_move_toward_feasibility(int variant, double *var);
/* the residual test of the second backtracking loop (onedpath_ip.cpp:941 / onedpath2_ip.cpp:828) laid open at the trial made after
   `halvings` halvings: out = { |r(trial)|^2, |r(x)|^2 (1 - 0.01 s), s, largest change of out[0] under a one-ulp move of one trial
   coordinate, largest change of out[1] under a one-ulp move of one coordinate of x, change of out[0] with the direction of a second
   backward-stable solver (Gaussian elimination) }; 0 if the loop's budget ends before that trial.  For certifying decisions that
   differ by a rounding. */
int    orc_armijo_sides(int variant, const double *var, int halvings, orc_qr_solver solver, double out[6]);
/* the feasibility test of the first backtracking loop (onedpath_ip.cpp:919-928) at the trial made after `halvings` halvings:
   out = { largest constraint value there (> 0: rejected), s, largest change of a constraint value under a one-ulp move of one variable,
   ... and with the direction of a second backward-stable solver } */
int    orc_feasibility_margin(int variant, const double *var, int halvings, orc_qr_solver solver, double out[4]);

void   orc_init_default(int variant, double *var);
void   orc_init_stuck_f3(double *var);
/* Build-defined feasible start: vel1 = 0, t_i = (3.5/sqrt 12) sqrt(6|dX_i|/L), lambda = 1. */
void   orc_init_feasible(int variant, double pos0, double pos1, double pos2, double *var);

/* Gate convention of SURVEY.md appendix A.5: before each step, stop if gap < tol.  Returns steps taken. */
int    orc_solve_gated(int variant, double *var, double gap_tol, int max_iter);

/* ---- column-pivoted Householder QR solve, any n <= 16, column-major A ---- */
int    orc_colpiv_qr_solve(int n, const double *A_colmajor, const double *b, double *x);
/* the same through Eigen's DYNAMIC-size code paths (MatrixXd: moveTowardFeasibility, onedpath_ip.cpp:693): reductions
   ordered as Redux.h's linear-vectorised traversal and GeneralMatrixVector.h's row-major kernel order them */
int    orc_colpiv_qr_solve_dynamic(int n, const double *A_colmajor, const double *b, double *x);

/* ---- batches (AoS, stride = orc_state_len) ; threads <= 0 means all cores ---- */
void   orc_batch_init_feasible(int variant, size_t n, const double *pos0, const double *pos1,
                               const double *pos2, double *aos);
void   orc_batch_steps(int variant, size_t n, double *aos, int k, int threads);
/* iters[i] receives the gated step count of problem i; returns the sum over the batch */
int64_t orc_batch_solve_gated(int variant, size_t n, double *aos, double gap_tol, int max_iter,
                              int32_t *iters, int threads);
/* the same with the linear solve delegated (oracle/_ref's Eigen QR is re-entrant: all state on the stack) */
int    orc_solve_gated_ex(int variant, double *var, double gap_tol, int max_iter, orc_qr_solver solver);
int64_t orc_batch_solve_gated_ex(int variant, size_t n, double *aos, double gap_tol, int max_iter,
                                 int32_t *iters, int threads, orc_qr_solver solver);
int    orc_hw_threads(void);

/* ---- trajectory sampling (plot data), onedpath_ip.cpp:1015-1088 ---- */
/* out_pos: 2*33 positions (segment 0 samples j=0..32, then segment 1), out_acc: 4 end accelerations */
void   orc_sample_trajectory(int variant, const double *var, double *out_pos, double *out_acc);

/* Synthetic problem generator shared by tests / bench (SplitMix64, documented in DESIGN.md).
   dist 0: monotone, 1: reference-like, 2: non-monotone stress. */
void   orc_gen_problems(uint64_t seed, size_t first, size_t n, int dist,
                        double *pos0, double *pos1, double *pos2);

#ifdef __cplusplus
}
#endif
#endif
