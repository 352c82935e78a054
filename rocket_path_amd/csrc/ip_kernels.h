// ip_kernels.h -- host-callable launchers of the HIP kernels (internal; the public surface
// is include/rp_batch.h).  All launchers enqueue on `stream` and return the hipError_t of
// the launch; none synchronises.
#pragma once

#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

namespace rp {

// What the scheduling pass keeps per problem for a batch that has just been given its problems: 32 bytes, one sector.
struct StartRecord {
    double pos0, pos1, pos2;
    long long problem;      // the problem's own index (diagnostic; nothing on the path reads it)
};

// One problem's answer, 32 bytes = one sector (the layout of rp_solution in include/rp_batch.h)
struct Solution {
    double vel1, duration0, duration1;
    int32_t iters;
    uint32_t status;
};

// Device-side view of one batch: `fields` SoA arrays of `n` elements, field f of the problem at POSITION s at
// base + f * stride + s (stride >= n, an odd multiple of 512 elements: 2-4 KiB aligned fields that do not alias in HBM, rp_batch.cpp).
struct BatchView {
    void *base;            // double* or float*
    size_t stride;         // elements between consecutive fields
    size_t n;              // problems
    int variant;           // 3 or 4
    int dtype;             // 0 = f64, 1 = f32, 2 = f32 state with f64 arithmetic (RP_DTYPE_* of rp_batch.h)
    bool zero_end_vel;     // vel0X == vel2X == 0 for every problem (true after every init the reference has; see Prob in ip_core.h)
    int32_t *iters;        // gated Newton steps taken per problem
    uint32_t *status;      // RP_ST_* bits per problem
    uint32_t *slot_of;     // scheduled order (schedule.hip): problem index -> position in the batch ...
    uint32_t *prob_of;     // ... and position -> problem index; both n words, meaningful while `scheduled`
    StartRecord *records;  // per PROBLEM: the three positions set_problems was given, copied by the scheduling pass; what the
                           // feasible start -- or the fused solve -- of a fresh batch gathers through prob_of.  Null until the
                           // first set_problems
    bool scheduled;        // false: the problems lie in problem order (identical problems of initDefault / initStuck)
    Solution *solution;    // bound by rp_batch_bind_solution (null: none): gated solves write each problem's record there, problem order
    int iters_add;         // ungated steps taken since the last init, added to the iteration counts that leave the batch in records
    unsigned long long *counters;   // 128 words: [0,64) shards of "problems still open after the last gated launch", [64,128) shards of gated steps executed
    uint32_t *lists;       // 2 n + 16 words (null until needed): two position lists + the rounds' counts -- the straggler hand-off of the gated solve
};

struct HostParams {
    double accel_limit, mu_divisor, boundary_fraction, backtrack, armijo;
    int max_backtracks;
    int stall_window;
    int mu_mode;                 // 0 = reference centring, 1 = centring by trial (ip_core.h, newton_step)
    double mu_sigma_try[2];      // the two candidates of mode 1
    int handoff_rounds;          // rp_params.handoff_rounds: 0 = automatic, -1 = never, 2..8 = always that many rounds
    int handoff_lanes;           // rp_params.handoff_lanes
};

inline int state_len(int variant) { return variant == 4 ? 12 : 16; }
inline size_t storage_size(int dtype) { return dtype == 0 ? 8 : 4; }      // bytes per state element in HBM
inline int num_constraints(int variant) { return variant == 4 ? 4 : 8; }

// k ungated Newton steps per problem, one launch.
hipError_t launch_steps(const BatchView &b, const HostParams &hp, int k, hipStream_t stream);
// k ungated steps with the per-problem halving totals of both line-search loops (diagnostic twin of launch_steps)
hipError_t launch_steps_counted(const BatchView &b, const HostParams &hp, int k, uint32_t *d_nfeas, uint32_t *d_nresid, hipStream_t stream);
// up to k gated steps per problem (k = max_iter gives the fused solve); zeroes counters[0] first.
hipError_t launch_solve(const BatchView &b, const HostParams &hp, int k, double gap_tol, int max_iter, hipStream_t stream);
// the fused solve (every problem to its gate in one launch).  from_start: the batch holds only its positions (just
// scheduled, schedule.hip) and every problem begins at its feasible start, formed in registers (reference mode, no stall
// detector, zero end velocities only)
hipError_t launch_solve_fused(const BatchView &b, const HostParams &hp, double gap_tol, int max_iter, bool from_start, hipStream_t stream);
// the same for states the batch's order says nothing about (k_solve_chunks<ROUNDS>): a problem whose step leaves its state bit for bit
// unchanged -- a start outside the feasible set -- takes its remaining budget as read (exact), and with rounds > 1 (b.lists must exist) the
// solve runs in ROUNDS: the first launch walks the whole batch and every wave hands its last `lanes` stragglers off -- leaves them open
// and appends their positions to a list -- once they have stepped alone for more than `patience` steps; each further launch walks the
// previous one's list, densely packed, the last one to the end.  Same per-problem arithmetic: same results.
hipError_t launch_solve_rounds(const BatchView &b, const HostParams &hp, double gap_tol, int max_iter, int rounds, int lanes, int patience, hipStream_t stream);
// max ||r||^2, max gap, #converged, gated steps (+ host_steps) -> d_out4 (device); d_partials has 4 * 1024 doubles.
hipError_t launch_reduce(const BatchView &b, const HostParams &hp, double host_steps, double *d_partials,
                         double *d_out4, hipStream_t stream);

// every problem's 32-byte solution record in problem order (walks positions, scatters whole sectors)
hipError_t launch_solution(const BatchView &b, Solution *d_out, hipStream_t stream);

// state movement / initialisation
hipError_t launch_aos_to_soa(const BatchView &b, const double *d_aos, hipStream_t stream);
hipError_t launch_soa_to_aos(const BatchView &b, double *d_aos, hipStream_t stream);
hipError_t launch_restart_feasible(const BatchView &b, const HostParams &hp, hipStream_t stream);      // the feasible start of the positions in the batch's constant fields
// the same from b.records through b.prob_of (a batch that has just been scheduled): positions into the constant fields, the
// feasible start, cleared progress words -- everything k_solve_chunks<START> forms in registers, written out
hipError_t launch_start_from_records(const BatchView &b, const HostParams &hp, hipStream_t stream);
hipError_t launch_init_const(const BatchView &b, const double *host_state /* state_len values */, hipStream_t stream);
hipError_t launch_nudge(const BatchView &b, int field, double delta, hipStream_t stream);
hipError_t launch_clear_progress(const BatchView &b, hipStream_t stream);
// compute the scheduled order (slot_of / prob_of; schedule.hip) from positions given as three strided double arrays in problem
// order, zero the batch's progress counters and -- write_positions -- copy every problem's three positions into b.records;
// d_scratch: schedule_scratch_bytes(n) bytes of device memory
hipError_t schedule_scratch_bytes(size_t n, size_t *bytes);
// one_wave_blocks: the form of the three kernels whose blocks are single waves (slower alone, able to run beside a solve: what
// rp_pipeline uses); the order is the same either way
hipError_t launch_schedule(const BatchView &b, const double *d_pos0, const double *d_pos1, const double *d_pos2, size_t pstride,
                           bool write_positions, void *d_scratch, size_t scratch_bytes, hipStream_t stream, bool one_wave_blocks = false);
// d_dst[problem] = d_src[position of that problem] for per-problem words kept in batch order (requires b.scheduled)
hipError_t launch_gather_u32(const BatchView &b, const uint32_t *d_src, uint32_t *d_dst, hipStream_t stream);

// the rows either side of the hot path
hipError_t launch_move_toward_feasibility(const BatchView &b, const HostParams &hp, hipStream_t stream);
hipError_t launch_sample(const BatchView &b, double *d_pos66, double *d_acc4, hipStream_t stream);
// the same for a whole scheduled batch with zero end velocities whose b.records still hold its positions: through problem-order
// records (solution records written into d_solution_scratch, n of them, first) instead of the gather through slot_of
hipError_t launch_sample_from_records(const BatchView &b, Solution *d_solution_scratch, double *d_pos66, double *d_acc4, hipStream_t stream);

// the same for problems [first, first + count), plus printState's constraint table (1 + 14 m doubles per problem)
hipError_t launch_soa_to_aos_range(const BatchView &b, size_t first, size_t count, double *d_aos, hipStream_t stream);
hipError_t launch_sample_range(const BatchView &b, size_t first, size_t count, double *d_pos66, double *d_acc4, hipStream_t stream);
hipError_t launch_constraint_table(const BatchView &b, const HostParams &hp, size_t first, size_t count, double *d_rows, hipStream_t stream);

}  // namespace rp
