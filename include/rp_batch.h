/*
 * rp_batch.h -- C ABI of the MI355X-native batched interior-point path.
 *
 * This is the one boundary between host code (the C++ `Problem` plug-in that drops into
 * rocket_path.cpp, or the Python mirror used by tests/bench) and the HIP kernels.  Plain
 * pointers and sizes only; no C++ or torch types cross it.  The library never throws:
 * every entry point returns an rp_status, and rp_last_error() has the text.
 *
 * A batch is N independent problems of one variant:
 *   RP_VARIANT_F3  two-segment cubic, 3 variables + 8 multipliers, c_i = -/+a - L
 *                  (replaces the file-static `Trajectory g_trajectory`, onedpath_ip.cpp:47-52)
 *   RP_VARIANT_F4  same spline, 4 multipliers, c_i = (a^2 - L^2)/2
 *                  (replaces `Trajectory2 g_trajectory`, onedpath2_ip.cpp:50-55)
 * Host-visible state is the reference's own array-of-structs layout, `double var[16]`
 * (enum V, onedpath_ip.cpp:15-43) or `double var[12]` (enum V2, onedpath2_ip.cpp:15-39)
 * per problem; on the device it is structure-of-arrays in the batch's compute type.
 *
 * Threading: a handle is not thread-safe (the reference is single-threaded: one GLUT
 * thread calls Problem::onKey).  Different handles may be used from different threads.
 * Calls that enqueue work are asynchronous on the batch's stream unless the doc says
 * "synchronous"; rp_batch_sync() waits.
 */
#ifndef RP_BATCH_H
#define RP_BATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define RP_API __attribute__((visibility("default")))
#else
#define RP_API
#endif

typedef struct rp_batch rp_batch; /* opaque, owned by the caller between create and destroy */

/* ABI revision of this header: bumped whenever a struct that crosses the boundary changes size or an entry point changes
 * meaning.  rp_abi_version() returns what the LIBRARY was built with; a binding compares the two before its first call
 * (rocket_path_amd/capi.py and BatchedOneDPathIP do) -- rp_params grew in revision 2 (mu_mode, mu_sigma_try), and a caller
 * compiled against the older header would have handed rp_batch_set_params a shorter struct.
 *   1  round 1      2  round 2: rp_params + mu_mode / mu_sigma_try; rp_batch_field_ptr returns batch order (rp_batch_slot_map)
 *   3  round 3: sizeof(rp_params) returned by rp_params_size(); set_problems defers the feasible start
 *   4  round 4: rp_solution records (rp_batch_solution_device, rp_batch_bind_solution); rp_batch_traffic_probe replaces an
 *      environment switch; the library reads nothing from the environment; rp_batch_sample_device checks its alignment
 *   5  round 5: rp_device_id; a bound solution buffer is seeded before a gated launch that skips finished problems
 *   6  round 6: rp_pipeline_* (positions in -> solutions out over several streams); rp_params + handoff_rounds / handoff_lanes (the gated
 *      solve in rounds); a raw pointer to a mutable field keeps the seeding pass on for every later gated launch */
#define RP_ABI_VERSION 6

typedef enum {
    RP_OK = 0,
    RP_ERR_INVALID = 1,     /* bad argument (null handle, unknown variant/dtype, n == 0, ...) */
    RP_ERR_DEVICE = 2,      /* a HIP call failed; rp_last_error() has hipGetErrorString */
    RP_ERR_NOMEM = 3,       /* host or device allocation failed */
    RP_ERR_UNSUPPORTED = 4, /* valid request this build does not implement */
    RP_ERR_NO_DEVICE = 5    /* no HIP device visible: the product path has no CPU fallback */
} rp_status;

#define RP_VARIANT_F3 3
#define RP_VARIANT_F4 4
#define RP_DTYPE_F64 0
#define RP_DTYPE_F32 1
/* fp32 state in HBM (the traffic of RP_DTYPE_F32: 76 B per F4 step), fp64 arithmetic in registers; the state is
 * rounded to fp32 after every step.  One step from an fp32 state then agrees with the fp64 reference step from the
 * same state to fp32 rounding (6e-8 relative) for every problem, where pure fp32 arithmetic cannot: its Armijo test
 * (onedpath_ip.cpp:941) needs a relative residual decrease of 0.01 s, below fp32 resolution for s < 1e-5, and its 3x3
 * solve loses cond(K) * 6e-8 -- see DESIGN.md section 4, "Single precision". */
#define RP_DTYPE_F32_STATE 2

/* Per-problem status bits (new; the reference has no error reporting). */
#define RP_ST_CONVERGED 1u  /* gap < tol seen at a gate check */
#define RP_ST_MAXITER 2u    /* step cap reached before the gate */
#define RP_ST_NONFINITE 4u  /* a variable became NaN/inf */
#define RP_ST_INFEASIBLE 8u /* some c_i > 0 at the last gate check (constraintsSatisfied false) */
#define RP_ST_STALLED 16u   /* stall detector fired (only when rp_params.stall_window > 0); the problem is left alone from then on */
#define RP_ST_WRONG_WAY 32u /* with the stall detector on: a gated launch left the problem unconverged with a total duration no smaller than
                               the one it started the launch with -- F4's "settles the wrong direction" (README.md:34).  Cleared again
                               should a later launch converge the problem */

/* Solver constants, defaults = the reference's compile-time values. */
typedef struct {
    double accel_limit;       /* 100.0   onedpath_ip.cpp:54 */
    double mu_divisor;        /* 10.0    perturbation = gap / (m * 10), onedpath_ip.cpp:812 */
    double boundary_fraction; /* 0.99    onedpath_ip.cpp:915 */
    double backtrack;         /* 0.5     onedpath_ip.cpp:927, 944 */
    double armijo;            /* 0.01    onedpath_ip.cpp:941 */
    int32_t max_backtracks;   /* 100     onedpath_ip.cpp:919, 934 */
    int32_t stall_window;     /* 0 = off (the reference's behaviour: it keeps stepping, e.g. from initStuck, onedpath_ip.cpp:177-199).
                                 w > 0: in a gated solve, a problem whose surrogate gap has not halved for w consecutive steps of
                                 one launch is marked RP_ST_STALLED and stops -- SURVEY.md 8f row 4; never changes a converging run */
    int32_t mu_mode;          /* 0 = the reference's centring, perturbation = gap / (m * mu_divisor) at every step (onedpath_ip.cpp:812); the
                                 default, and the only mode the parity guarantees are about.
                                 1 = centring by trial (a Mehrotra-style predictor/centring split): the step is
                                 d_a + p d_c from one factorisation; sigma = mu_sigma_try[0], then [1] (p = sigma * gap / m) is taken if its
                                 full step keeps the multipliers positive, is primal feasible and passes the reference's residual test,
                                 else the reference step.  Same optimum in fewer steps (15.4 -> 12.7 mean on the benchmark distribution);
                                 double arithmetic only */
    double mu_sigma_try[2];   /* 0.01, 0.03 */
    int32_t handoff_rounds;   /* How the fused gated solve (rp_batch_solve with steps_per_launch <= 0) treats states its internal order says
                                 nothing about -- states that were set, nudged, moved or handed out raw since the last set_problems / init
                                 (reference mode only: mu_mode 0, no stall detector).
                                 0 (default): such a batch runs the kernel that WATCHES FOR FIXED POINTS: a problem whose step leaves
                                 its state bit for bit unchanged -- a start outside the feasible set: 100 feasibility halvings, no
                                 movement, onedpath_ip.cpp:919-928 -- takes its remaining step budget as read; count, status and state are
                                 exactly what stepping on would leave, without the ~20,000 evaluations.  Everything else (and the
                                 benchmark's path: set_problems -> solve) runs the plain kernel.
                                 2..8: additionally IN ROUNDS: a wave runs until its slowest lane is done, so one problem that needs 60
                                 steps holds 63 finished neighbours; in rounds a wave whose stepping lanes have numbered <= handoff_lanes
                                 for more than one step stops and leaves them open, the next launch packs all open problems densely,
                                 the last round runs to the end.  Same steps per problem, same results.  Pays only where most lane-steps
                                 would idle (measured: profiles/r6_state_families.log); costs 10-30 % where they would not.
                                 -1: the plain kernel always. */
    int32_t handoff_lanes;    /* 24 (1..48) */
} rp_params;

/* Batch-wide reduction, the payload of the one cross-GPU collective (max / max / sum / sum). */
typedef struct {
    double max_residual_sq; /* max_i ||r_i||^2 with r as residual(), onedpath_ip.cpp:753-792, p = gap_i/(10 m) */
    double max_gap;         /* max_i surrogateDualityGap, onedpath_ip.cpp:794-808 */
    double n_converged;     /* problems with RP_ST_CONVERGED (double so one dtype all-reduces) */
    double total_steps;     /* Newton steps executed since the last init/set_state */
} rp_reduction;

/* One problem's answer: what the reference leaves in var[vel1X], var[duration0], var[duration1] of its Trajectory
 * (onedpath_ip.cpp:47-52, read by printState 997-1006), plus the two progress words.  32 bytes = one HBM sector. */
typedef struct {
    double vel1, duration0, duration1;
    int32_t iters;   /* Newton steps taken since the last init / set_state / set_problems (gated + ungated) */
    uint32_t status; /* RP_ST_* */
} rp_solution;

/* ---- library ---- */
RP_API const char *rp_version(void);
RP_API int rp_abi_version(void);       /* RP_ABI_VERSION of the library's own build */
RP_API size_t rp_params_size(void);    /* sizeof(rp_params) in the library: must equal the caller's */
RP_API const char *rp_last_error(void); /* thread-local text of the last failure */
RP_API const char *rp_status_string(int status);
RP_API int rp_device_count(int *count);
/* "pci <domain:bus:device.function> uuid <hex>" of HIP device `device` into out (len >= 64).  New -- the reference is one process
 * on no device; a multi-GPU host (bench.py, ShardedOneDPathIP) reports with it that its N shards really sat on N different GPUs. */
RP_API int rp_device_id(int device, char *out, size_t len);
RP_API void rp_params_default(rp_params *p);

/* ---- lifetime ---- */
/* `stream` is a hipStream_t the caller owns (e.g. torch's current stream) or NULL for a
 * stream the batch creates.  `device` is the HIP device ordinal. */
RP_API int rp_batch_create(rp_batch **out, int variant, int dtype, size_t n, int device, void *stream);
RP_API int rp_batch_destroy(rp_batch *b);
RP_API int rp_batch_set_params(rp_batch *b, const rp_params *p);
RP_API int rp_batch_get_params(const rp_batch *b, rp_params *p);
RP_API int rp_batch_size(const rp_batch *b, size_t *n);
RP_API int rp_batch_info(const rp_batch *b, int *variant, int *dtype, int *device);

/* ---- init: Problem::init() / onKey('i') / onKey('j') for every problem of the batch ---- */
RP_API int rp_batch_init_default(rp_batch *b); /* initDefault, onedpath_ip.cpp:201-228 / onedpath2_ip.cpp:164-193 */
RP_API int rp_batch_init_stuck(rp_batch *b);   /* initStuck, onedpath_ip.cpp:177-199 (F3 only) */
/* Per-problem positions (host arrays of n doubles), then the feasible start rule of
 * SURVEY.md 8d on the device: vel = 0, t_i = (3.5/sqrt 12) sqrt(6 |dX_i| / L), multipliers 1.
 * This (and rp_batch_set_state) is also where the batch decides in which order it keeps its problems internally
 * (sorted by expected step count, for the gated solve); callers never see that order except through
 * rp_batch_field_ptr: problem i of every call below is the problem of element i of these arrays. */
RP_API int rp_batch_set_problems(rp_batch *b, const double *pos0, const double *pos1, const double *pos2);
/* Same with device-resident inputs (no PCIe in the path).  Asynchronous: the three arrays are read, in the batch's stream order,
 * by this call's kernels only (they are copied), so they may be overwritten by later work on that stream.  The call computes the
 * batch's internal order and stops there: the start state itself is formed in registers by a fused rp_batch_solve
 * (steps_per_launch <= 0) that follows, or written out by whichever other call touches the state first -- same bits either way.
 * What a caller can observe of the deferral: "positions in, solutions out" costs one launch of state traffic less; the start
 * is the one of rp_params.accel_limit AS IT WAS when the problems were set (rp_batch_set_params writes the start out before
 * it changes the limit); raw pointers from rp_batch_field_ptr are undefined until the next state-touching call. */
RP_API int rp_batch_set_problems_device(rp_batch *b, const double *d_pos0, const double *d_pos1, const double *d_pos2);
/* Back to the feasible start of the positions the batch already holds (the `I` key for per-problem positions): nothing
 * crosses the boundary.  Asynchronous. */
RP_API int rp_batch_restart(rp_batch *b);
/* Whole state in the reference's AoS layout, n * 16 (F3) or n * 12 (F4) doubles.  Synchronous. */
RP_API int rp_batch_set_state(rp_batch *b, const double *aos);
RP_API int rp_batch_get_state(rp_batch *b, double *aos);
/* The same for problems [first, first + count): count * 16 (or 12) doubles.  No per-call allocation and only the
 * requested rows cross PCIe -- what a host that watches ONE problem of the batch calls (printState, onedpath_ip.cpp:997-1006). */
RP_API int rp_batch_get_state_range(rp_batch *b, size_t first, size_t count, double *aos);
/* Special-key nudges (onSpecialKey, onedpath_ip.cpp:280-324): var[index] += delta for all problems. */
RP_API int rp_batch_nudge(rp_batch *b, int var_index, double delta);

/* ---- the hot path ---- */
/* k times onKey('n') = moveInteriorPoint (onedpath_ip.cpp:810-953 / onedpath2_ip.cpp:698-841)
 * on every problem, ungated, fused into one launch (state stays in registers between steps). */
RP_API int rp_batch_step(rp_batch *b, int k);
/* Diagnostic twin of rp_batch_step: the same k steps (same arithmetic: results are bit-identical to rp_batch_step's, state and
 * multipliers, for every k, variant and number mode -- every fixed-step kernel of a variant evaluates the reference's
 * post-convergence residual loop in the same form), returning per problem how often the feasibility loop
 * (onedpath_ip.cpp:927) and the residual loop (:944) halved the step over those k steps.  For decision-level comparisons with the
 * reference; one problem per lane, synchronous, not a fast path. */
RP_API int rp_batch_step_counted(rp_batch *b, int k, uint32_t *feas_halvings, uint32_t *resid_halvings);
/* Gated solve, the convention of SURVEY.md appendix A.5 per problem:
 *     for (it = 0; it < max_iter; ++it) { if (gap < gap_tol) break; step; }
 * steps_per_launch <= 0: one fused launch (each lane loops until its own gate);
 * steps_per_launch = s > 0: launches of s steps until every problem is done (host polls a
 * device counter after each launch; synchronous).  Iteration counts accumulate across calls
 * until the next init/set_state. */
RP_API int rp_batch_solve(rp_batch *b, double gap_tol, int max_iter, int steps_per_launch);
/* ONE asynchronous launch of up to k gated steps per still-open problem, no host polling: the building block of a solve
 * whose convergence check is global (multi-GPU: launch on every shard, all-reduce the summaries, repeat -- SURVEY.md 8d, C4;
 * rocket_path_amd/sharding.py, solve_with_global_checks). */
RP_API int rp_batch_solve_launch(rp_batch *b, double gap_tol, int max_iter, int k);
/* The Space key, moveTowardFeasibility (onedpath_ip.cpp:648-721), on every problem. */
RP_API int rp_batch_move_toward_feasibility(rp_batch *b);

/* ---- results ---- */
RP_API int rp_batch_get_iters(rp_batch *b, int32_t *iters, uint32_t *status); /* either may be NULL; synchronous */
/* Solutions in PROBLEM order in DEVICE memory the caller owns (n records, 32-byte aligned): record i is problem i of the arrays
 * handed to set_problems / set_state, whatever order the batch keeps internally -- the device-side counterpart of reading
 * var[] of trajectory i in the reference (onedpath_ip.cpp:47-52, 997-1006); nothing crosses PCIe.  Asynchronous on the batch
 * stream; works on any state (after steps, solves, nudges).  68 B of HBM traffic per problem. */
RP_API int rp_batch_solution_device(rp_batch *b, rp_solution *d_out);
/* Bind (NULL: unbind) a buffer of n records: from now on every gated solve (rp_batch_solve, rp_batch_solve_launch) writes the
 * record of each problem it works on as that problem leaves the launch -- "positions in, solutions out in problem order" then
 * costs no extra pass (the one 32-byte sector per problem is written under the solve's arithmetic).  After a gated solve of a
 * batch EVERY record is current, also those of problems that had finished in an earlier launch and that this one skipped: a
 * launch that may skip problems is preceded by one pass that writes all records from the state as it is (k_solution, 68 B per
 * problem) whenever the buffer is new or anything but gated solves has touched the state since the records were written; the
 * first solve of a batch that has just been given its problems stores every record itself and needs no such pass.  Calls other
 * than gated solves (rp_batch_step, nudges, set_state ...) do not update the buffer -- use rp_batch_solution_device for the
 * state they leave.  The buffer must outlive the binding. */
RP_API int rp_batch_bind_solution(rp_batch *b, rp_solution *d_out);
RP_API int rp_batch_reduce(rp_batch *b, rp_reduction *out);                   /* synchronous */
/* Writes the 4 doubles of rp_reduction to device memory the caller owns, asynchronously on
 * the batch stream: the buffer a multi-GPU caller hands to its RCCL all-reduce. */
RP_API int rp_batch_reduce_device(rp_batch *b, double *d_out4);
/* For single-process multi-GPU hosts (csrc/host/sharded_problem.cpp): the same reduction into the batch's OWN
 * 4-double device slot, whose address is returned (asynchronous; all-reduce it in place on rp_batch_stream()),
 * and the synchronous read-back of that slot afterwards. */
RP_API int rp_batch_summary_device(rp_batch *b, double **d_out4);
RP_API int rp_batch_summary_read(rp_batch *b, rp_reduction *out);
/* Plot data (plotTrajectory/plotAcceleration, onedpath_ip.cpp:1015-1088): per problem 66
 * positions (33 per segment) and 4 end accelerations, host arrays, synchronous. */
RP_API int rp_batch_sample(rp_batch *b, double *pos66, double *acc4);

/* The same into DEVICE memory the caller owns (n x 66 and n x 4 doubles), asynchronously on the batch stream: for a consumer
 * that draws from device memory, and what bench.py times (nothing crosses PCIe). */
/* d_pos66 must be 16-byte aligned (positions are written as 16-byte vectors): RP_ERR_INVALID otherwise. */
RP_API int rp_batch_sample_device(rp_batch *b, double *d_pos66, double *d_acc4);

/* The same for problems [first, first + count) only (what onDraw needs for the watched problem). Synchronous. */
RP_API int rp_batch_sample_range(rp_batch *b, size_t first, size_t count, double *pos66, double *acc4);
/* The rest of printState for problems [first, first + count): `Surrogate gap` and the `Constraints:` table
 * (printConstraints, onedpath_ip.cpp:955-995, 1008-1010).  Per problem 1 + 14 m doubles (m = 8 for F3, 4 for F4):
 * [0] = surrogate gap; then for constraint i at 1 + 14 i: error, deriv[3], second[3][3] row-major, dot = (0,-1,-1).deriv.
 * Variable order (vel1X, duration0, duration1) as in enum V.  Synchronous. */
RP_API int rp_batch_constraints_range(rp_batch *b, size_t first, size_t count, double *rows);

/* ---- stream / timing plumbing ---- */
/* Bandwidth calibration: the one-launch-per-step kernel with NO step -- its loads and stores of every problem's state and
 * nothing else (the state is rewritten unchanged).  What bench.py prices the k = 1 launch against.  Asynchronous. */
RP_API int rp_batch_traffic_probe(rp_batch *b);
RP_API int rp_batch_sync(rp_batch *b);
RP_API int rp_batch_stream(rp_batch *b, void **stream);
/* HIP events on the batch's own stream: record slot 0..7, elapsed between two recorded slots. */
RP_API int rp_batch_event_record(rp_batch *b, int slot);
RP_API int rp_batch_event_elapsed_ms(rp_batch *b, int slot_start, int slot_stop, float *ms);
/* Device pointer of one SoA field (0..15 / 0..11) for callers that manage their own copies.  Elements are in BATCH
 * order: the batch keeps its problems sorted for the gated solve (set_problems / set_state decide the order), problem i's
 * element is ptr[slot_of_problem[i]] with the map of rp_batch_slot_map.  Asking for an end-velocity field
 * (vel0X / vel2X) makes the batch assume they may become non-zero (general kernels) until the next init /
 * set_problems / set_state.  A pointer taken earlier is undefined between rp_batch_set_problems(_device) and the next call
 * that touches the state (the feasible start is written lazily): take it again after set_problems.  Once a pointer to a
 * CONSTANT field (positions, end velocities) has been handed out the batch assumes for the rest of its life that positions may be
 * written behind its back: rp_batch_sample_device then always reads them from the fields (the gather path), never from the
 * records set_problems kept. */
RP_API int rp_batch_field_ptr(rp_batch *b, int field, void **d_ptr);
/* slot_of_problem[i] = position of problem i inside the batch's field arrays (n words; the identity after
 * init_default / init_stuck).  Every other entry point takes and returns problem order; only rp_batch_field_ptr
 * exposes batch order.  Synchronous. */
RP_API int rp_batch_slot_map(rp_batch *b, uint32_t *slot_of_problem);

/* ---- positions in -> solutions out, batch after batch (new: the reference solves ONE problem per key press, onedpath_ip.cpp:269-272;
 * this is the caller either side of the batched path) ----
 * A pipeline owns `depth` batches of n problems on one device and `n_streams` (1..4, depth a multiple of it) streams; job i goes to
 * batch i % depth on stream (i % depth) % n_streams.  rp_pipeline_submit enqueues, for one job and without synchronising the host,
 * exactly what a caller would issue by hand -- rp_batch_bind_solution, rp_batch_set_problems_device (the scheduling pass),
 * rp_batch_solve (fused, from the feasible start formed in registers) -- so every result is bit for bit the one-stream path's.  What
 * the arrangement buys: with n_streams >= 2 the scheduling pass of job i + 1 (memory- and latency-bound) and the first waves of its
 * solve run under the drain of job i's solve (vector-ALU bound, wave slots emptying): "positions in, solutions out" at the rate of
 * the solve kernel alone or better (bench.py `end_to_end`).
 *   inputs_stream  the stream whose earlier work produces the position arrays (NULL: they are ready now); the job waits for it
 *                  on the device.  The arrays must stay untouched until the job's scheduling pass has read them:
 *                  rp_pipeline_stream_wait(job, 0, s) makes stream s wait for exactly that, rp_pipeline_wait(job) the host.
 *   d_out          n rp_solution records in problem order (32-byte aligned; NULL: none -- read the batch through rp_pipeline_batch)
 * (An application note, not something the library does: the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware
 * queues -- 4 by default -- and streams that share a queue serialise.  A process that already owns two or three streams of its own should
 * export GPU_MAX_HW_QUEUES=8 before its first HIP call, or the pipeline's two streams may come to share a queue: 73 against 91 G Newton
 * steps/s on one box, profiles/r6_hw_queues.log.  bench.py does so.)
 *   job            receives the job's number (0, 1, 2, ...); a slot is reused every `depth` jobs, in stream order -- the previous
 *                  job of the slot has finished on the device before the new one touches the batch; its d_out is the caller's to
 *                  have consumed by then. */
typedef struct rp_pipeline rp_pipeline;
RP_API int rp_pipeline_create(rp_pipeline **out, int variant, int dtype, size_t n, int device, int depth, int n_streams);
RP_API int rp_pipeline_destroy(rp_pipeline *p);
RP_API int rp_pipeline_set_params(rp_pipeline *p, const rp_params *params); /* every batch of the pipeline */
/* Where the scheduling pass of a job runs (before the first submit only).  INLINE (the default): on the job's own stream, ahead of its
 * solve.  STREAM / PRIORITY: on one more stream the pipeline owns (PRIORITY: created with the device's highest stream priority), ordered
 * behind the slot's previous job and ahead of the job's solve by events.  Measured equal to INLINE within 1 % where the process owns few
 * streams; every further stream risks sharing a hardware queue with another (the HIP runtime has GPU_MAX_HW_QUEUES = 4 of them by
 * default), and streams that share a queue serialise. */
#define RP_PIPELINE_PREP_INLINE 0
#define RP_PIPELINE_PREP_STREAM 1
#define RP_PIPELINE_PREP_PRIORITY 2
#define RP_PIPELINE_PREP_FAT_KERNELS 16 /* OR-ed in: keep the scheduling pass in its 256-thread form (A/B measurements; same order) */
RP_API int rp_pipeline_set_prep(rp_pipeline *p, int mode);
RP_API int rp_pipeline_submit(rp_pipeline *p, const double *d_pos0, const double *d_pos1, const double *d_pos2, rp_solution *d_out,
                              double gap_tol, int max_iter, void *inputs_stream, int64_t *job);
/* Host waits until job (and everything submitted to its slot before it) has finished; job < 0: everything submitted. */
RP_API int rp_pipeline_wait(rp_pipeline *p, int64_t job);
/* Device-side dependency: `stream` waits until the job's position arrays have been read (what = 0) or its solutions are
 * written (what = 1).  Only while the job is still the last one of its slot. */
RP_API int rp_pipeline_stream_wait(rp_pipeline *p, int64_t job, int what, void *stream);
/* The batch that holds job (while it is still the last one of its slot): every rp_batch_* read-back works on it (rp_batch_reduce,
 * rp_batch_get_iters, rp_batch_sample_device ...), on the job's own stream.  Owned by the pipeline. */
RP_API int rp_pipeline_batch(rp_pipeline *p, int64_t job, rp_batch **batch);

#ifdef __cplusplus
}
#endif
#endif
