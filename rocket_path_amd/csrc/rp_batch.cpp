// rp_batch.cpp -- implementation of the C ABI in include/rp_batch.h over the HIP kernels.
//
// Owns, per batch: the SoA state in HBM (one allocation, `fields` arrays of `stride`
// elements), the per-problem progress words, a small scratch for reductions and the
// AoS staging buffer used by set_state/get_state.  No CPU fallback exists: without a HIP
// device every entry point that needs one fails with RP_ERR_NO_DEVICE.
#include "../../include/rp_batch.h"

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <new>

#include "ip_kernels.h"

struct rp_batch {
    rp::BatchView view;
    rp::HostParams params;
    int device;
    hipStream_t stream;
    bool own_stream;
    double *d_scratch;        // [0..4095] block partials, [4096..4099] reduction result
    double *d_aos;            // lazily allocated n * state_len doubles
    double *d_pos;            // lazily allocated 3 * n doubles (set_problems staging)
    double *d_range;          // lazily allocated kRangeChunk * kRangeRow doubles: staging of the *_range read-backs
    uint32_t *d_words;        // lazily allocated 2 n words: per-problem words gathered from batch order into problem order
    void *d_sched;            // lazily allocated scratch of the scheduling pass (schedule.hip), sched_bytes bytes
    size_t sched_bytes;
    rp::Solution *d_solscratch;   // lazily allocated n solution records: the plot data of a whole batch reads its state through them
    bool records_current;     // view.records hold the positions the batch's constant fields hold (set_problems; until a set_state, an init,
                              // a nudge of a position or a raw field pointer handed out)
    bool raw_positions_out;   // a raw pointer to a CONSTANT field (a position, an end velocity) has been handed out (rp_batch_field_ptr): the caller may
                              // write positions the records never see, at any later time -- the records path of rp_batch_sample_device stays off
                              // for the life of the batch
    bool sol_stale;           // a solution buffer is bound and something other than a gated solve has touched the state (or the buffer is new): the
                              // records of problems the next gated launch does NOT work on are not current -- that launch seeds the buffer first
    bool raw_state_out;       // a raw pointer to a MUTABLE field (vel1, a duration, a multiplier) has been handed out (rp_batch_field_ptr): the caller
                              // may write state the batch never sees, at any later time -- sticky until the next init / set_problems / set_state
                              // (which invalidate such pointers' meaning): while it is set, a bound solution buffer is seeded before EVERY gated
                              // launch that may skip problems, not only the first one after the hand-out
    bool unpredicted;         // the state has been set, nudged, moved or handed out raw since the last set_problems / init: the batch's internal order
                              // no longer predicts step counts, and the fused gated solve runs in rounds (rp_params.handoff_rounds = 0: automatic)
    bool slim_schedule;       // the scheduling pass runs in its one-wave-per-block form (schedule.hip): set by rp_pipeline for its batches
    bool at_start;            // set_problems has run and nothing else since: the batch holds its scheduled order and its positions; the
                              // feasible start itself (mutable fields, progress words) is NOT materialised yet -- see materialize()
    double ungated_steps;     // per-problem count of ungated steps since the last init
    unsigned long long *h_pinned;   // 72 pinned host words: [0,64) counter shards, [64,68) reduction: read-backs without pageable staging
    hipEvent_t events[8];
    bool event_live[8];
};

namespace {

thread_local char g_err[512] = "";

int fail(int status, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return status;
}

#define RP_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(e_ == hipErrorOutOfMemory ? RP_ERR_NOMEM : RP_ERR_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

#define RP_NEED(b)                                                  \
    do {                                                            \
        if (!(b)) return fail(RP_ERR_INVALID, "null batch handle"); \
        RP_HIP(hipSetDevice((b)->device));                          \
    } while (0)

size_t elem_size(int dtype) { return rp::storage_size(dtype); }

// The *_range calls (a watched problem, a page of a table) go through one small device buffer that lives as long as
// the batch: kRangeChunk problems x the widest row (the F3 constraint table, 1 + 14 * 8 doubles) = 0.9 MB.
constexpr size_t kRangeChunk = 1024, kRangeRow = 113;

void default_params(rp::HostParams &hp)
{
    hp.accel_limit = 100.0;
    hp.mu_divisor = 10.0;
    hp.boundary_fraction = 0.99;
    hp.backtrack = 0.5;
    hp.armijo = 0.01;
    hp.max_backtracks = 100;
    hp.stall_window = 0;
    hp.mu_mode = 0;
    hp.mu_sigma_try[0] = 0.01;
    hp.mu_sigma_try[1] = 0.03;
    hp.handoff_rounds = 0;
    hp.handoff_lanes = 24;
}

int reset_progress(rp_batch *b)
{
    b->ungated_steps = 0.0;
    RP_HIP(rp::launch_clear_progress(b->view, b->stream));
    return RP_OK;
}

// the scheduled order from positions given as three strided double arrays in problem order (device memory), on the batch's
// own stream (three small kernels of ours, schedule.hip; round 2's library sort needed a queue of its own)
int schedule(rp_batch *b, const double *d_pos0, const double *d_pos1, const double *d_pos2, size_t pstride, bool write_positions)
{
    if (!b->d_sched) {
        size_t bytes = 0;
        RP_HIP(rp::schedule_scratch_bytes(b->view.n, &bytes));
        void *p = nullptr;
        RP_HIP(hipMalloc(&p, bytes));
        b->d_sched = p;
        b->sched_bytes = bytes;
    }
    if (write_positions && !b->view.records) {      // 32 B per problem: what the scheduling pass keeps for the feasible start / the fused solve
        void *p = nullptr;
        RP_HIP(hipMalloc(&p, b->view.n * sizeof(rp::StartRecord)));
        b->view.records = (rp::StartRecord *)p;
    }
    RP_HIP(rp::launch_schedule(b->view, d_pos0, d_pos1, d_pos2, pstride, write_positions, b->d_sched, b->sched_bytes, b->stream, b->slim_schedule));
    b->view.scheduled = true;
    return RP_OK;
}

// set_problems leaves the batch "at its start" without writing the start: the fused gated solve forms it in registers
// (k_solve_chunks<START>).  Every other consumer of the state goes through here first: positions from the records into the
// constant fields, their feasible start, cleared progress words -- bit for bit what the fused solve starts from.
int materialize(rp_batch *b)
{
    if (!b->at_start) return RP_OK;
    b->at_start = false;
    RP_HIP(rp::launch_start_from_records(b->view, b->params, b->stream));      // (the progress counters were zeroed by the scheduling pass)
    return RP_OK;
}

// A gated launch writes the bound solution record of every problem it WORKS ON (k_solve_chunks); problems that finished in an
// earlier launch are skipped without touching their state.  Their records are current only if nothing but gated solves has run
// since they were written: otherwise (a new buffer, steps, a nudge, a set_state ... in between) the launch is preceded by one pass
// that writes every record from the state as it is (k_solution, 68 B per problem).  Not on the fresh-batch path: a START launch
// stores every record itself.
int seed_solution(rp_batch *b)
{
    if (!b->view.solution || !(b->sol_stale || b->raw_state_out)) return RP_OK;
    RP_HIP(rp::launch_solution(b->view, b->view.solution, b->stream));
    return RP_OK;      // (sol_stale is cleared by the caller once its gated launch has been enqueued: a failed launch leaves the buffer stale)
}

#define RP_NEED_STATE(b)                     \
    do {                                     \
        RP_NEED(b);                          \
        const int ms_ = materialize(b);      \
        if (ms_ != RP_OK) return ms_;        \
    } while (0)

int need_words(rp_batch *b)
{
    if (!b->d_words) RP_HIP(hipMalloc((void **)&b->d_words, 2 * b->view.n * sizeof(uint32_t)));
    return RP_OK;
}

int need_range(rp_batch *b)
{
    if (!b->d_range) RP_HIP(hipMalloc((void **)&b->d_range, kRangeChunk * kRangeRow * sizeof(double)));
    return RP_OK;
}

int check_range(const rp_batch *b, size_t first, size_t count, const void *out)
{
    if (!out) return fail(RP_ERR_INVALID, "null output");
    if (first > b->view.n || count > b->view.n - first) return fail(RP_ERR_INVALID, "range [%zu, %zu + %zu) outside the batch of %zu", first, first, count, b->view.n);
    return RP_OK;
}

int need_aos(rp_batch *b)
{
    if (!b->d_aos) RP_HIP(hipMalloc((void **)&b->d_aos, b->view.n * rp::state_len(b->view.variant) * sizeof(double)));
    return RP_OK;
}

}  // namespace

extern "C" {

const char *rp_version(void) { return "rocket_path_amd 0.6 (gfx950)"; }
int rp_abi_version(void) { return RP_ABI_VERSION; }
size_t rp_params_size(void) { return sizeof(rp_params); }
const char *rp_last_error(void) { return g_err; }

const char *rp_status_string(int status)
{
    switch (status) {
    case RP_OK: return "ok";
    case RP_ERR_INVALID: return "invalid argument";
    case RP_ERR_DEVICE: return "HIP error";
    case RP_ERR_NOMEM: return "out of memory";
    case RP_ERR_UNSUPPORTED: return "unsupported";
    case RP_ERR_NO_DEVICE: return "no HIP device (this path has no CPU fallback)";
    default: return "unknown status";
    }
}

int rp_device_count(int *count)
{
    if (!count) return fail(RP_ERR_INVALID, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; (void)hipGetLastError(); return fail(RP_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return RP_OK;
}

int rp_device_id(int device, char *out, size_t len)
{
    if (!out || len < 64) return fail(RP_ERR_INVALID, "rp_device_id needs a buffer of at least 64 bytes");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { (void)hipGetLastError(); return fail(RP_ERR_NO_DEVICE, "no HIP device visible"); }
    if (device < 0 || device >= n) return fail(RP_ERR_INVALID, "device %d out of range (%d visible)", device, n);
    char bus[32] = "";
    RP_HIP(hipDeviceGetPCIBusId(bus, (int)sizeof bus, device));
    hipUUID uuid;
    std::memset(&uuid, 0, sizeof uuid);
    char hex[2 * sizeof uuid.bytes + 1] = "";
    if (hipDeviceGetUuid(&uuid, device) == hipSuccess) {
        for (size_t i = 0; i < sizeof uuid.bytes; ++i) snprintf(hex + 2 * i, 3, "%02x", (unsigned)(unsigned char)uuid.bytes[i]);
    } else {
        (void)hipGetLastError();
    }
    snprintf(out, len, "pci %s uuid %s", bus, hex[0] ? hex : "?");
    return RP_OK;
}

void rp_params_default(rp_params *p)
{
    if (!p) return;
    rp::HostParams hp;
    default_params(hp);
    p->accel_limit = hp.accel_limit;
    p->mu_divisor = hp.mu_divisor;
    p->boundary_fraction = hp.boundary_fraction;
    p->backtrack = hp.backtrack;
    p->armijo = hp.armijo;
    p->max_backtracks = hp.max_backtracks;
    p->stall_window = hp.stall_window;
    p->mu_mode = hp.mu_mode;
    p->mu_sigma_try[0] = hp.mu_sigma_try[0];
    p->mu_sigma_try[1] = hp.mu_sigma_try[1];
    p->handoff_rounds = hp.handoff_rounds;
    p->handoff_lanes = hp.handoff_lanes;
}

int rp_batch_create(rp_batch **out, int variant, int dtype, size_t n, int device, void *stream)
{
    if (!out) return fail(RP_ERR_INVALID, "out is null");
    *out = nullptr;
    if (variant != RP_VARIANT_F3 && variant != RP_VARIANT_F4) return fail(RP_ERR_INVALID, "variant %d (want 3 or 4)", variant);
    if (dtype != RP_DTYPE_F64 && dtype != RP_DTYPE_F32 && dtype != RP_DTYPE_F32_STATE)
        return fail(RP_ERR_INVALID, "dtype %d (want 0 = f64, 1 = f32 or 2 = f32 state with f64 arithmetic)", dtype);
    if (n == 0) return fail(RP_ERR_INVALID, "empty batch");
    if (n > 0x7fffffffu) return fail(RP_ERR_INVALID, "batch of %zu problems (at most 2^31 - 1 per batch; shard larger jobs)", n);
    int count = 0;
    int st = rp_device_count(&count);
    if (st != RP_OK || count == 0) return fail(RP_ERR_NO_DEVICE, "no HIP device visible; the interior-point path runs on the GPU only");
    if (device < 0 || device >= count) return fail(RP_ERR_INVALID, "device %d out of range (%d visible)", device, count);
    RP_HIP(hipSetDevice(device));

    rp_batch *b = new (std::nothrow) rp_batch();
    if (!b) return fail(RP_ERR_NOMEM, "host allocation failed");
    std::memset(b, 0, sizeof *b);
    b->device = device;
    default_params(b->params);
    b->view.n = n;
    b->view.variant = variant;
    b->view.dtype = dtype;
    // Field f of problem i lives at base + f * stride + i.  A stride that is a large power of two in bytes (8 MiB at
    // n = 2^20 doubles) puts element i of all 16 fields on the same HBM channel and bank: the 25 streams of a step then
    // fight over one row buffer (measured, profiles/probes/stride_probe.py: 3.7 TB/s at k = 1 against 4.6 TB/s with the
    // fields 1.25 KiB or more out of phase; plateau from there on).  So the stride is an ODD multiple of 512 elements.
    b->view.stride = (n + 511) / 512 * 512;
    if ((b->view.stride / 512) % 2 == 0) b->view.stride += 512;
#ifdef RP_TUNING      // tuning builds only (profiles/probes/stride_probe.py): the shipped library reads nothing from the environment
    if (const char *pad = getenv("RP_STRIDE_PAD")) b->view.stride += (size_t)atoi(pad) / 16 * 16;
#endif
    b->view.zero_end_vel = true;       // the state starts all-zero
    b->view.scheduled = false;         // ... and identical problems lie in problem order
    const size_t fields = (size_t)rp::state_len(variant);

    hipError_t e = hipSuccess;
    if (stream) { b->stream = (hipStream_t)stream; b->own_stream = false; }
    else { e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking); b->own_stream = (e == hipSuccess); }
    if (e == hipSuccess) e = hipMalloc(&b->view.base, fields * b->view.stride * elem_size(dtype));
    if (e == hipSuccess) e = hipMalloc((void **)&b->view.iters, n * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&b->view.status, n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&b->view.slot_of, n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&b->view.prob_of, n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&b->view.counters, 128 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_scratch, (4096 + 4) * sizeof(double));
    if (e == hipSuccess) e = hipHostMalloc((void **)&b->h_pinned, 72 * sizeof(unsigned long long), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMemsetAsync(b->view.base, 0, fields * b->view.stride * elem_size(dtype), b->stream);
    if (e == hipSuccess) e = rp::launch_clear_progress(b->view, b->stream);
    if (e != hipSuccess) {
        const int code = fail(e == hipErrorOutOfMemory ? RP_ERR_NOMEM : RP_ERR_DEVICE, "rp_batch_create: %s", hipGetErrorString(e));
        rp_batch_destroy(b);
        return code;
    }
    *out = b;
    return RP_OK;
}

int rp_batch_destroy(rp_batch *b)
{
    if (!b) return RP_OK;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (int i = 0; i < 8; ++i) if (b->event_live[i]) (void)hipEventDestroy(b->events[i]);
    if (b->view.base) (void)hipFree(b->view.base);
    if (b->view.iters) (void)hipFree(b->view.iters);
    if (b->view.status) (void)hipFree(b->view.status);
    if (b->view.slot_of) (void)hipFree(b->view.slot_of);
    if (b->view.prob_of) (void)hipFree(b->view.prob_of);
    if (b->view.records) (void)hipFree(b->view.records);
    if (b->view.lists) (void)hipFree(b->view.lists);
    if (b->d_words) (void)hipFree(b->d_words);
    if (b->d_sched) (void)hipFree(b->d_sched);
    if (b->view.counters) (void)hipFree(b->view.counters);
    if (b->d_scratch) (void)hipFree(b->d_scratch);
    if (b->h_pinned) (void)hipHostFree(b->h_pinned);
    if (b->d_aos) (void)hipFree(b->d_aos);
    if (b->d_pos) (void)hipFree(b->d_pos);
    if (b->d_range) (void)hipFree(b->d_range);
    if (b->d_solscratch) (void)hipFree(b->d_solscratch);
    if (b->own_stream && b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
    return RP_OK;
}

int rp_batch_set_params(rp_batch *b, const rp_params *p)
{
    if (!b || !p) return fail(RP_ERR_INVALID, "null argument");
    if (!(p->accel_limit > 0) || !(p->mu_divisor > 0) || !(p->boundary_fraction > 0 && p->boundary_fraction <= 1) ||
        !(p->backtrack > 0 && p->backtrack < 1) || !(p->armijo >= 0 && p->armijo < 1) || p->max_backtracks < 0 ||
        p->max_backtracks > 4096 || p->stall_window < 0)      // every device loop must stay short: a runaway kernel takes the GPU with it
        return fail(RP_ERR_INVALID, "parameter out of range");
    if (p->mu_mode != 0 && p->mu_mode != 1) return fail(RP_ERR_INVALID, "mu_mode %d (want 0 = reference or 1 = centring by trial)", p->mu_mode);
    if (!(p->handoff_rounds == 0 || p->handoff_rounds == -1 || (p->handoff_rounds >= 2 && p->handoff_rounds <= 8)))
        return fail(RP_ERR_INVALID, "handoff_rounds %d (want 0 = automatic, -1 = never, or 2..8)", p->handoff_rounds);
    if (p->handoff_lanes < 1 || p->handoff_lanes > 48) return fail(RP_ERR_INVALID, "handoff_lanes %d (want 1..48)", p->handoff_lanes);
    if (p->mu_mode == 1) {
        if (b->view.dtype == RP_DTYPE_F32) return fail(RP_ERR_UNSUPPORTED, "mu_mode 1 needs double arithmetic (RP_DTYPE_F64 or RP_DTYPE_F32_STATE)");
        if (!(p->mu_sigma_try[0] > 0 && p->mu_sigma_try[0] <= p->mu_sigma_try[1] && p->mu_sigma_try[1] < 1))
            return fail(RP_ERR_INVALID, "mu_sigma_try must satisfy 0 < [0] <= [1] < 1");
    }
    if (b->at_start && p->accel_limit != b->params.accel_limit) {
        // the feasible start of a batch that has just been given its problems is formed lazily, from the limit of the moment the
        // problems were set: write it out before the limit changes (set_problems, set_params, solve = the start of the OLD limit,
        // as if set_problems had written it)
        RP_HIP(hipSetDevice(b->device));
        const int ms = materialize(b);
        if (ms != RP_OK) return ms;
    }
    b->params.accel_limit = p->accel_limit;
    b->params.mu_divisor = p->mu_divisor;
    b->params.boundary_fraction = p->boundary_fraction;
    b->params.backtrack = p->backtrack;
    b->params.armijo = p->armijo;
    b->params.max_backtracks = p->max_backtracks;
    b->params.stall_window = p->stall_window;
    b->params.mu_mode = p->mu_mode;
    b->params.mu_sigma_try[0] = p->mu_sigma_try[0];
    b->params.mu_sigma_try[1] = p->mu_sigma_try[1];
    b->params.handoff_rounds = p->handoff_rounds;
    b->params.handoff_lanes = p->handoff_lanes;
    return RP_OK;
}

int rp_batch_get_params(const rp_batch *b, rp_params *p)
{
    if (!b || !p) return fail(RP_ERR_INVALID, "null argument");
    p->accel_limit = b->params.accel_limit;
    p->mu_divisor = b->params.mu_divisor;
    p->boundary_fraction = b->params.boundary_fraction;
    p->backtrack = b->params.backtrack;
    p->armijo = b->params.armijo;
    p->max_backtracks = b->params.max_backtracks;
    p->stall_window = b->params.stall_window;
    p->mu_mode = b->params.mu_mode;
    p->mu_sigma_try[0] = b->params.mu_sigma_try[0];
    p->mu_sigma_try[1] = b->params.mu_sigma_try[1];
    p->handoff_rounds = b->params.handoff_rounds;
    p->handoff_lanes = b->params.handoff_lanes;
    return RP_OK;
}

int rp_batch_size(const rp_batch *b, size_t *n)
{
    if (!b || !n) return fail(RP_ERR_INVALID, "null argument");
    *n = b->view.n;
    return RP_OK;
}

int rp_batch_info(const rp_batch *b, int *variant, int *dtype, int *device)
{
    if (!b) return fail(RP_ERR_INVALID, "null batch handle");
    if (variant) *variant = b->view.variant;
    if (dtype) *dtype = b->view.dtype;
    if (device) *device = b->device;
    return RP_OK;
}

// initDefault: pos (0, 200, 400), zero velocities, durations 3.5, multipliers 1
// (onedpath_ip.cpp:201-228, onedpath2_ip.cpp:164-193).
int rp_batch_init_default(rp_batch *b)
{
    RP_NEED(b);
    double s[16];
    const int m = rp::num_constraints(b->view.variant);
    s[0] = 0.0; s[1] = 3.5; s[2] = 3.5;
    for (int i = 0; i < m; ++i) s[3 + i] = 1.0;
    s[3 + m + 0] = 0.0; s[3 + m + 1] = 0.0; s[3 + m + 2] = 200.0; s[3 + m + 3] = 400.0; s[3 + m + 4] = 0.0;
    RP_HIP(rp::launch_init_const(b->view, s, b->stream));
    b->sol_stale = true;
    b->raw_state_out = false;
    b->unpredicted = false;      // identical problems: identical step counts
    b->view.zero_end_vel = true;
    b->view.scheduled = false;         // identical problems: nothing to schedule
    b->at_start = false;
    b->records_current = false;
    return reset_progress(b);
}

// initStuck, onedpath_ip.cpp:177-199.
int rp_batch_init_stuck(rp_batch *b)
{
    RP_NEED(b);
    if (b->view.variant != RP_VARIANT_F3) return fail(RP_ERR_UNSUPPORTED, "the stuck state exists for F3 only (onedpath_ip.cpp:177-199)");
    const double s[16] = {-9.66825, 4.78149, 4.38968,
                          5.45948e-07, 0.00310769, 3.49109e-08, 0.00281523, 8.39344e-07, 1.76937e-06, 0.0187559, 8.42414e-07,
                          0.0, 0.0, 350.0, 400.0, 0.0};
    RP_HIP(rp::launch_init_const(b->view, s, b->stream));
    b->sol_stale = true;
    b->raw_state_out = false;
    b->unpredicted = false;      // identical problems: identical step counts
    b->view.zero_end_vel = true;
    b->view.scheduled = false;
    b->at_start = false;
    b->records_current = false;
    return reset_progress(b);
}

int rp_batch_set_problems_device(rp_batch *b, const double *d_pos0, const double *d_pos1, const double *d_pos2)
{
    RP_NEED(b);
    if (!d_pos0 || !d_pos1 || !d_pos2) return fail(RP_ERR_INVALID, "null position array");
    // Where each problem goes (scheduled order), its positions copied into its record, progress counters zeroed: three
    // kernels (schedule.hip).  The feasible start itself is not written: a fused gated solve that follows forms it in registers,
    // anything else materialises it first (materialize above).  The position arrays are consumed in stream order, here.
    if (!b->view.zero_end_vel) {       // a set_state / nudge / field_ptr may have left non-zero end velocities: the start rule zeroes them
        const size_t es = elem_size(b->view.dtype), cb = 3 + (size_t)rp::num_constraints(b->view.variant);
        RP_HIP(hipMemsetAsync((char *)b->view.base + (cb + 1) * b->view.stride * es, 0, b->view.n * es, b->stream));
        RP_HIP(hipMemsetAsync((char *)b->view.base + (cb + 4) * b->view.stride * es, 0, b->view.n * es, b->stream));
    }
    int st = schedule(b, d_pos0, d_pos1, d_pos2, 1, true);
    if (st != RP_OK) return st;
    b->view.zero_end_vel = true;       // the feasible-start rule sets vel0 = vel2 = 0
    b->ungated_steps = 0.0;
    b->sol_stale = true;
    b->raw_state_out = false;
    b->unpredicted = false;      // the feasible-start rule: what the scheduled order was fitted to
    b->at_start = true;
    b->records_current = true;
    return RP_OK;
}

int rp_batch_restart(rp_batch *b)
{
    RP_NEED(b);
    b->sol_stale = true;
    if (!b->raw_positions_out) b->unpredicted = false;      // back on the feasible start of the positions the order was computed from
    if (b->at_start) return materialize(b);      // already at the start of its positions: write it out
    RP_HIP(rp::launch_restart_feasible(b->view, b->params, b->stream));
    b->view.zero_end_vel = true;
    b->ungated_steps = 0.0;
    RP_HIP(rp::launch_clear_progress(b->view, b->stream));      // the positions have not changed: the scheduled order stays as it is
    return RP_OK;
}

int rp_batch_set_problems(rp_batch *b, const double *pos0, const double *pos1, const double *pos2)
{
    RP_NEED(b);
    if (!pos0 || !pos1 || !pos2) return fail(RP_ERR_INVALID, "null position array");
    const size_t n = b->view.n;
    if (!b->d_pos) RP_HIP(hipMalloc((void **)&b->d_pos, 3 * n * sizeof(double)));
    RP_HIP(hipMemcpyAsync(b->d_pos, pos0, n * sizeof(double), hipMemcpyHostToDevice, b->stream));
    RP_HIP(hipMemcpyAsync(b->d_pos + n, pos1, n * sizeof(double), hipMemcpyHostToDevice, b->stream));
    RP_HIP(hipMemcpyAsync(b->d_pos + 2 * n, pos2, n * sizeof(double), hipMemcpyHostToDevice, b->stream));
    int st = rp_batch_set_problems_device(b, b->d_pos, b->d_pos + n, b->d_pos + 2 * n);
    if (st != RP_OK) return st;
    RP_HIP(hipStreamSynchronize(b->stream));   // the host arrays may be reused on return
    return RP_OK;
}

int rp_batch_set_state(rp_batch *b, const double *aos)
{
    RP_NEED(b);
    if (!aos) return fail(RP_ERR_INVALID, "null state array");
    int st = need_aos(b);
    if (st != RP_OK) return st;
    const size_t M = (size_t)rp::state_len(b->view.variant), bytes = b->view.n * M * sizeof(double);
    {   // which instantiation the Newton kernels may use: are all end velocities zero?  (NaN counts as non-zero)
        const size_t iv0 = 3 + rp::num_constraints(b->view.variant) + 1, iv2 = iv0 + 3;
        bool zero = true;
        for (size_t i = 0; i < b->view.n && zero; ++i) zero = (aos[i * M + iv0] == 0.0) && (aos[i * M + iv2] == 0.0);
        b->view.zero_end_vel = zero;
    }
    RP_HIP(hipMemcpyAsync(b->d_aos, aos, bytes, hipMemcpyHostToDevice, b->stream));
    {   // schedule by the positions in the rows (columns pos0, pos1, pos2 of the reference's enum), then scatter the rows
        const size_t cb = 3 + (size_t)rp::num_constraints(b->view.variant);
        st = schedule(b, b->d_aos + cb + 0, b->d_aos + cb + 2, b->d_aos + cb + 3, M, false);
        if (st != RP_OK) return st;
        b->at_start = false;      // the rows below are the state
        b->records_current = false;
        b->sol_stale = true;
        b->raw_state_out = false;
        b->unpredicted = true;      // any state: the order (computed from the positions in the rows) predicts nothing about it
    }
    RP_HIP(rp::launch_aos_to_soa(b->view, b->d_aos, b->stream));
    st = reset_progress(b);
    if (st != RP_OK) return st;
    RP_HIP(hipStreamSynchronize(b->stream));
    return RP_OK;
}

int rp_batch_get_state(rp_batch *b, double *aos)
{
    RP_NEED_STATE(b);
    if (!aos) return fail(RP_ERR_INVALID, "null state array");
    int st = need_aos(b);
    if (st != RP_OK) return st;
    const size_t bytes = b->view.n * rp::state_len(b->view.variant) * sizeof(double);
    RP_HIP(rp::launch_soa_to_aos(b->view, b->d_aos, b->stream));
    RP_HIP(hipMemcpyAsync(aos, b->d_aos, bytes, hipMemcpyDeviceToHost, b->stream));
    RP_HIP(hipStreamSynchronize(b->stream));
    return RP_OK;
}

int rp_batch_get_state_range(rp_batch *b, size_t first, size_t count, double *aos)
{
    RP_NEED_STATE(b);
    int st = check_range(b, first, count, aos);
    if (st == RP_OK) st = need_range(b);
    if (st != RP_OK) return st;
    const size_t M = (size_t)rp::state_len(b->view.variant);
    for (size_t done = 0; done < count; done += kRangeChunk) {
        const size_t c = count - done < kRangeChunk ? count - done : kRangeChunk;
        RP_HIP(rp::launch_soa_to_aos_range(b->view, first + done, c, b->d_range, b->stream));
        RP_HIP(hipMemcpyAsync(aos + done * M, b->d_range, c * M * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        RP_HIP(hipStreamSynchronize(b->stream));      // the staging buffer is reused by the next chunk
    }
    return RP_OK;
}

int rp_batch_nudge(rp_batch *b, int var_index, double delta)
{
    RP_NEED_STATE(b);
    if (var_index < 0 || var_index >= rp::state_len(b->view.variant)) return fail(RP_ERR_INVALID, "variable index %d out of range", var_index);
    RP_HIP(rp::launch_nudge(b->view, var_index, delta, b->stream));
    b->sol_stale = true;
    b->unpredicted = true;
    if (var_index >= 3 + rp::num_constraints(b->view.variant)) b->records_current = false;      // a constant moved: the records no longer are the batch's positions
    {
        const int iv0 = 3 + rp::num_constraints(b->view.variant) + 1, iv2 = iv0 + 3;
        if ((var_index == iv0 || var_index == iv2) && delta != 0.0) b->view.zero_end_vel = false;
    }
    return RP_OK;
}

int rp_batch_step(rp_batch *b, int k)
{
    RP_NEED_STATE(b);
    if (k < 0 || k > 1000000) return fail(RP_ERR_INVALID, "step count %d out of range (0..1000000)", k);
    if (k == 0) return RP_OK;
    RP_HIP(rp::launch_steps(b->view, b->params, k, b->stream));
    b->ungated_steps += (double)k;
    b->sol_stale = true;
    return RP_OK;
}

int rp_batch_traffic_probe(rp_batch *b)
{
    RP_NEED_STATE(b);
    RP_HIP(rp::launch_steps(b->view, b->params, 0, b->stream));      // the k = 1 kernel with no steps: its loads and stores, nothing else
    return RP_OK;
}

int rp_batch_step_counted(rp_batch *b, int k, uint32_t *feas_halvings, uint32_t *resid_halvings)
{
    RP_NEED_STATE(b);
    if (k < 0 || k > 1000000) return fail(RP_ERR_INVALID, "step count %d out of range (0..1000000)", k);
    if (!feas_halvings || !resid_halvings) return fail(RP_ERR_INVALID, "null output");
    if (b->params.mu_mode != 0) return fail(RP_ERR_UNSUPPORTED, "the counted step exists for the reference's mu mode only");
    const size_t n = b->view.n;
    uint32_t *d = nullptr;
    RP_HIP(hipMalloc((void **)&d, 4 * n * sizeof(uint32_t)));
    hipError_t e = rp::launch_steps_counted(b->view, b->params, k, d, d + n, b->stream);
    const uint32_t *out_f = d, *out_r = d + n;
    if (b->view.scheduled) {      // the kernel counts per position: bring the counts into problem order
        if (e == hipSuccess) e = rp::launch_gather_u32(b->view, d, d + 2 * n, b->stream);
        if (e == hipSuccess) e = rp::launch_gather_u32(b->view, d + n, d + 3 * n, b->stream);
        out_f = d + 2 * n;
        out_r = d + 3 * n;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(feas_halvings, out_f, n * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(resid_halvings, out_r, n * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RP_ERR_DEVICE, "rp_batch_step_counted: %s", hipGetErrorString(e));
    b->ungated_steps += (double)k;
    b->sol_stale = true;
    return RP_OK;
}

int rp_batch_solve(rp_batch *b, double gap_tol, int max_iter, int steps_per_launch)
{
    RP_NEED(b);
    if (max_iter < 0 || max_iter > 1000000) return fail(RP_ERR_INVALID, "max_iter %d out of range (0..1000000)", max_iter);
    if (!(gap_tol == gap_tol)) return fail(RP_ERR_INVALID, "gap_tol is NaN");
    if (steps_per_launch <= 0) {
        // a batch that has just been given its problems starts from the feasible start formed in registers (reference mode only)
        const bool from_start = b->at_start && b->params.mu_mode == 0 && b->params.stall_window == 0 && b->view.zero_end_vel && max_iter > 0;
        if (from_start) b->at_start = false;
        else { const int ms = materialize(b); if (ms != RP_OK) return ms; }
        b->view.iters_add = (int)b->ungated_steps;
        if (!from_start) { const int ss = seed_solution(b); if (ss != RP_OK) return ss; }      // (the START launch stores every record itself)
        // in rounds (rp_params.handoff_rounds): states the batch's order says nothing about -- reference mode only, and a batch big enough for a second wave
        // A batch whose state has been set, nudged, moved or handed out raw runs the kernel that watches for fixed points (starts outside the
        // feasible set use their budget up at once instead of walking a hundred halvings two hundred times: exact); rounds on request
        const bool plain = from_start || b->params.mu_mode != 0 || b->params.stall_window > 0 || max_iter <= 0;
        int rounds = b->params.handoff_rounds >= 2 ? b->params.handoff_rounds : 1;
        if (b->view.n <= 64) rounds = 1;
        const bool watched = !plain && b->params.handoff_rounds != -1 && (b->unpredicted || rounds > 1);
        if (watched) {
            if (rounds > 1 && !b->view.lists) RP_HIP(hipMalloc((void **)&b->view.lists, (2 * b->view.n + 16) * sizeof(uint32_t)));
            RP_HIP(rp::launch_solve_rounds(b->view, b->params, gap_tol, max_iter, rounds, b->params.handoff_lanes, 1, b->stream));
        } else {
            RP_HIP(rp::launch_solve_fused(b->view, b->params, gap_tol, max_iter, from_start, b->stream));
        }
        b->sol_stale = false;      // only now: a launch that failed has written no record
        return RP_OK;
    }
    { const int ms = materialize(b); if (ms != RP_OK) return ms; }
    b->view.iters_add = (int)b->ungated_steps;
    { const int ss = seed_solution(b); if (ss != RP_OK) return ss; }
    // bounded host loop: every launch either finishes a problem or advances it by >= 1 step
    const int max_launches = max_iter / steps_per_launch + 2;
    for (int l = 0; l < max_launches; ++l) {
        RP_HIP(rp::launch_solve(b->view, b->params, steps_per_launch, gap_tol, max_iter, b->stream));
        b->sol_stale = false;
        RP_HIP(hipMemcpyAsync(b->h_pinned, b->view.counters, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream));
        RP_HIP(hipStreamSynchronize(b->stream));
        unsigned long long open = 0;
        for (int i = 0; i < 64; ++i) open += b->h_pinned[i];
        if (open == 0) break;
    }
    return RP_OK;
}

int rp_batch_solve_launch(rp_batch *b, double gap_tol, int max_iter, int k)
{
    RP_NEED_STATE(b);
    if (max_iter < 0 || max_iter > 1000000) return fail(RP_ERR_INVALID, "max_iter %d out of range (0..1000000)", max_iter);
    if (k < 1 || k > 1000000) return fail(RP_ERR_INVALID, "steps per launch %d out of range (1..1000000)", k);
    if (!(gap_tol == gap_tol)) return fail(RP_ERR_INVALID, "gap_tol is NaN");
    b->view.iters_add = (int)b->ungated_steps;
    { const int ss = seed_solution(b); if (ss != RP_OK) return ss; }
    RP_HIP(rp::launch_solve(b->view, b->params, k, gap_tol, max_iter, b->stream));
    b->sol_stale = false;
    return RP_OK;
}

int rp_batch_move_toward_feasibility(rp_batch *b)
{
    RP_NEED_STATE(b);
    RP_HIP(rp::launch_move_toward_feasibility(b->view, b->params, b->stream));
    b->sol_stale = true;
    b->unpredicted = true;
    return RP_OK;
}

int rp_batch_get_iters(rp_batch *b, int32_t *iters, uint32_t *status)
{
    RP_NEED_STATE(b);
    const size_t n = b->view.n;
    const uint32_t *src_it = reinterpret_cast<const uint32_t *>(b->view.iters), *src_st = b->view.status;
    if (b->view.scheduled) {      // the words lie in batch order
        int st = need_words(b);
        if (st != RP_OK) return st;
        if (iters) RP_HIP(rp::launch_gather_u32(b->view, src_it, b->d_words, b->stream));
        if (status) RP_HIP(rp::launch_gather_u32(b->view, src_st, b->d_words + n, b->stream));
        src_it = b->d_words;
        src_st = b->d_words + n;
    }
    if (iters) RP_HIP(hipMemcpyAsync(iters, src_it, n * sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
    if (status) RP_HIP(hipMemcpyAsync(status, src_st, n * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
    RP_HIP(hipStreamSynchronize(b->stream));
    if (iters && b->ungated_steps > 0) {
        const int32_t add = (int32_t)b->ungated_steps;
        for (size_t i = 0; i < n; ++i) iters[i] += add;
    }
    return RP_OK;
}

int rp_batch_solution_device(rp_batch *b, rp_solution *d_out)
{
    RP_NEED_STATE(b);
    if (!d_out) return fail(RP_ERR_INVALID, "null output");
    if (((uintptr_t)d_out & 31u) != 0) return fail(RP_ERR_INVALID, "solution records must be 32-byte aligned");
    b->view.iters_add = (int)b->ungated_steps;
    RP_HIP(rp::launch_solution(b->view, reinterpret_cast<rp::Solution *>(d_out), b->stream));
    return RP_OK;
}

int rp_batch_bind_solution(rp_batch *b, rp_solution *d_out)
{
    if (!b) return fail(RP_ERR_INVALID, "null batch handle");
    if (((uintptr_t)d_out & 31u) != 0) return fail(RP_ERR_INVALID, "solution records must be 32-byte aligned");
    b->view.solution = reinterpret_cast<rp::Solution *>(d_out);
    b->sol_stale = true;      // nothing in the new buffer is current: the next gated launch that skips finished problems seeds it first
    return RP_OK;
}

int rp_batch_reduce_device(rp_batch *b, double *d_out4)
{
    RP_NEED_STATE(b);
    if (!d_out4) return fail(RP_ERR_INVALID, "null output");
    RP_HIP(rp::launch_reduce(b->view, b->params, b->ungated_steps * (double)b->view.n, b->d_scratch, d_out4, b->stream));
    return RP_OK;
}

int rp_batch_reduce(rp_batch *b, rp_reduction *out)
{
    RP_NEED(b);
    if (!out) return fail(RP_ERR_INVALID, "null output");
    int st = rp_batch_reduce_device(b, b->d_scratch + 4096);
    if (st != RP_OK) return st;
    double *h = reinterpret_cast<double *>(b->h_pinned + 64);
    RP_HIP(hipMemcpyAsync(h, b->d_scratch + 4096, 4 * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    RP_HIP(hipStreamSynchronize(b->stream));
    out->max_residual_sq = h[0];
    out->max_gap = h[1];
    out->n_converged = h[2];
    out->total_steps = h[3];
    return RP_OK;
}

int rp_batch_summary_device(rp_batch *b, double **d_out4)
{
    RP_NEED(b);
    if (!d_out4) return fail(RP_ERR_INVALID, "null output");
    int st = rp_batch_reduce_device(b, b->d_scratch + 4096);
    if (st != RP_OK) return st;
    *d_out4 = b->d_scratch + 4096;
    return RP_OK;
}

int rp_batch_summary_read(rp_batch *b, rp_reduction *out)
{
    RP_NEED(b);
    if (!out) return fail(RP_ERR_INVALID, "null output");
    double *h = reinterpret_cast<double *>(b->h_pinned + 64);
    RP_HIP(hipMemcpyAsync(h, b->d_scratch + 4096, 4 * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    RP_HIP(hipStreamSynchronize(b->stream));
    out->max_residual_sq = h[0];
    out->max_gap = h[1];
    out->n_converged = h[2];
    out->total_steps = h[3];
    return RP_OK;
}

int rp_batch_sample(rp_batch *b, double *pos66, double *acc4)
{
    RP_NEED_STATE(b);
    if (!pos66 || !acc4) return fail(RP_ERR_INVALID, "null output");
    const size_t n = b->view.n;
    double *d = nullptr;
    RP_HIP(hipMalloc((void **)&d, n * 70 * sizeof(double)));
    hipError_t e = rp::launch_sample(b->view, d, d + n * 66, b->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(pos66, d, n * 66 * sizeof(double), hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(acc4, d + n * 66, n * 4 * sizeof(double), hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RP_ERR_DEVICE, "rp_batch_sample: %s", hipGetErrorString(e));
    return RP_OK;
}

int rp_batch_sample_device(rp_batch *b, double *d_pos66, double *d_acc4)
{
    RP_NEED_STATE(b);
    if (!d_pos66 || !d_acc4) return fail(RP_ERR_INVALID, "null output");
    if (((uintptr_t)d_pos66 & 15u) != 0) return fail(RP_ERR_INVALID, "d_pos66 must be 16-byte aligned (the positions are written as 16-byte vectors)");
    if (b->view.scheduled && b->view.zero_end_vel && b->records_current && !b->raw_positions_out && b->view.records) {
        // a whole scheduled batch whose positions are still the ones it was given: through problem-order records (two coalesced
        // sectors per problem) instead of the per-field gather; same arithmetic, same bits
        if (!b->d_solscratch) RP_HIP(hipMalloc((void **)&b->d_solscratch, b->view.n * sizeof(rp::Solution)));
        b->view.iters_add = (int)b->ungated_steps;
        RP_HIP(rp::launch_sample_from_records(b->view, b->d_solscratch, d_pos66, d_acc4, b->stream));
        return RP_OK;
    }
    RP_HIP(rp::launch_sample(b->view, d_pos66, d_acc4, b->stream));
    return RP_OK;
}

int rp_batch_sample_range(rp_batch *b, size_t first, size_t count, double *pos66, double *acc4)
{
    RP_NEED_STATE(b);
    int st = check_range(b, first, count, pos66);
    if (st == RP_OK && !acc4) st = fail(RP_ERR_INVALID, "null output");
    if (st == RP_OK) st = need_range(b);
    if (st != RP_OK) return st;
    for (size_t done = 0; done < count; done += kRangeChunk) {
        const size_t c = count - done < kRangeChunk ? count - done : kRangeChunk;
        double *d_pos = b->d_range, *d_acc = b->d_range + kRangeChunk * 66;
        RP_HIP(rp::launch_sample_range(b->view, first + done, c, d_pos, d_acc, b->stream));
        RP_HIP(hipMemcpyAsync(pos66 + done * 66, d_pos, c * 66 * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        RP_HIP(hipMemcpyAsync(acc4 + done * 4, d_acc, c * 4 * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        RP_HIP(hipStreamSynchronize(b->stream));
    }
    return RP_OK;
}

int rp_batch_constraints_range(rp_batch *b, size_t first, size_t count, double *rows)
{
    RP_NEED_STATE(b);
    int st = check_range(b, first, count, rows);
    if (st == RP_OK) st = need_range(b);
    if (st != RP_OK) return st;
    const size_t row = 1 + 14 * (size_t)rp::num_constraints(b->view.variant);
    for (size_t done = 0; done < count; done += kRangeChunk) {
        const size_t c = count - done < kRangeChunk ? count - done : kRangeChunk;
        RP_HIP(rp::launch_constraint_table(b->view, b->params, first + done, c, b->d_range, b->stream));
        RP_HIP(hipMemcpyAsync(rows + done * row, b->d_range, c * row * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        RP_HIP(hipStreamSynchronize(b->stream));
    }
    return RP_OK;
}

int rp_batch_sync(rp_batch *b)
{
    RP_NEED(b);
    RP_HIP(hipStreamSynchronize(b->stream));
    return RP_OK;
}

int rp_batch_stream(rp_batch *b, void **stream)
{
    if (!b || !stream) return fail(RP_ERR_INVALID, "null argument");
    *stream = (void *)b->stream;
    return RP_OK;
}

int rp_batch_event_record(rp_batch *b, int slot)
{
    RP_NEED(b);
    if (slot < 0 || slot >= 8) return fail(RP_ERR_INVALID, "event slot %d out of range", slot);
    if (!b->event_live[slot]) {
        RP_HIP(hipEventCreate(&b->events[slot]));
        b->event_live[slot] = true;
    }
    RP_HIP(hipEventRecord(b->events[slot], b->stream));
    return RP_OK;
}

int rp_batch_event_elapsed_ms(rp_batch *b, int slot_start, int slot_stop, float *ms)
{
    RP_NEED(b);
    if (!ms || slot_start < 0 || slot_start >= 8 || slot_stop < 0 || slot_stop >= 8 || !b->event_live[slot_start] || !b->event_live[slot_stop])
        return fail(RP_ERR_INVALID, "events not recorded");
    RP_HIP(hipEventSynchronize(b->events[slot_stop]));
    RP_HIP(hipEventElapsedTime(ms, b->events[slot_start], b->events[slot_stop]));
    return RP_OK;
}

int rp_batch_field_ptr(rp_batch *b, int field, void **d_ptr)
{
    if (!b || !d_ptr) return fail(RP_ERR_INVALID, "null argument");
    if (field < 0 || field >= rp::state_len(b->view.variant)) return fail(RP_ERR_INVALID, "field %d out of range", field);
    {   // the caller is about to look at (or write) raw state: it has to exist
        RP_HIP(hipSetDevice(b->device));
        const int ms = materialize(b);
        if (ms != RP_OK) return ms;
    }
    *d_ptr = (char *)b->view.base + (size_t)field * b->view.stride * elem_size(b->view.dtype);
    b->records_current = false;      // the caller may write through the pointer
    b->sol_stale = true;
    if (field >= 3 + rp::num_constraints(b->view.variant)) b->raw_positions_out = true;      // ... now or at any later time: sticky (see the struct)
    else b->raw_state_out = true;                                                            // ... and so may the state: every later gated launch seeds a bound buffer first
    b->unpredicted = true;
    {   // a caller holding a raw pointer to an end-velocity field may write non-zero values the batch never sees: from
        // here on (until the next init / set_problems / set_state) the Newton kernels read vel0X and vel2X
        const int iv0 = 3 + rp::num_constraints(b->view.variant) + 1, iv2 = iv0 + 3;
        if (field == iv0 || field == iv2) b->view.zero_end_vel = false;
    }
    return RP_OK;
}

int rp_batch_slot_map(rp_batch *b, uint32_t *slot_of_problem)
{
    RP_NEED(b);
    if (!slot_of_problem) return fail(RP_ERR_INVALID, "null output");
    const size_t n = b->view.n;
    if (!b->view.scheduled) {
        for (size_t i = 0; i < n; ++i) slot_of_problem[i] = (uint32_t)i;
        return RP_OK;
    }
    RP_HIP(hipMemcpyAsync(slot_of_problem, b->view.slot_of, n * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
    RP_HIP(hipStreamSynchronize(b->stream));
    return RP_OK;
}

// ---- rp_pipeline: positions in -> solutions out, batch after batch, with the batches dealt onto several streams ----
// Nothing but the entry points above, in the order a caller would issue them by hand: what it adds is the arrangement -- `depth` batches
// of n problems, slot j bound to stream j % n_streams at creation, job i in slot i % depth -- under which the scheduling pass of job
// i + 1 (three small memory- and latency-bound kernels) and the head of its solve run while job i's solve, a vector-ALU-bound kernel on
// the other stream, is still draining: wave slots stand empty 18 % of a lone 1 Mi-problem launch at its two ends (profiles/r5_tuning.md).
}  // extern "C"

struct rp_pipeline {
    int device, depth, n_streams;
    size_t n;
    hipStream_t streams[4];
    rp_batch **slots;
    hipEvent_t *done;          // per slot: recorded behind the slot's last job
    hipEvent_t *consumed;      // per slot: recorded behind that job's scheduling pass (its position arrays have been read)
    hipEvent_t inputs_ready;   // scratch: recorded on the caller's stream in rp_pipeline_submit
    hipStream_t prep[4];       // prep_mode != 0: the streams the jobs' scheduling passes run on, job i on prep[i % n_prep] (its solve waits for it through `consumed`)
    int n_prep;
    int prep_mode;             // RP_PIPELINE_PREP_*
    int64_t *job_of;           // per slot: the job it last took (-1: none)
    int64_t next_job;
};

extern "C" {

int rp_pipeline_create(rp_pipeline **out, int variant, int dtype, size_t n, int device, int depth, int n_streams)
{
    if (!out) return fail(RP_ERR_INVALID, "out is null");
    *out = nullptr;
    if (n_streams < 1 || n_streams > 4) return fail(RP_ERR_INVALID, "n_streams %d (want 1..4)", n_streams);
    if (depth < n_streams || depth > 1024 || depth % n_streams != 0)
        return fail(RP_ERR_INVALID, "depth %d (want a multiple of n_streams = %d, at most 1024)", depth, n_streams);
    rp_pipeline *p = new (std::nothrow) rp_pipeline();
    if (!p) return fail(RP_ERR_NOMEM, "host allocation failed");
    std::memset(p, 0, sizeof *p);
    p->device = device; p->depth = depth; p->n_streams = n_streams; p->n = n;
    p->slots = new (std::nothrow) rp_batch *[depth]();
    p->done = new (std::nothrow) hipEvent_t[depth]();
    p->consumed = new (std::nothrow) hipEvent_t[depth]();
    p->job_of = new (std::nothrow) int64_t[depth]();
    if (!p->slots || !p->done || !p->consumed || !p->job_of) { rp_pipeline_destroy(p); return fail(RP_ERR_NOMEM, "host allocation failed"); }
    for (int j = 0; j < depth; ++j) p->job_of[j] = -1;
    int st = RP_OK;
    for (int j = 0; j < depth && st == RP_OK; ++j) {
        // the first batch of a stream creates it (a non-blocking stream of its own); the others of that stream share it
        st = rp_batch_create(&p->slots[j], variant, dtype, n, device, j < n_streams ? nullptr : (void *)p->streams[j % n_streams]);
        if (st == RP_OK && j < n_streams) p->streams[j] = p->slots[j]->stream;
        if (st == RP_OK) p->slots[j]->slim_schedule = n_streams > 1;      // beside a running solve only one-wave blocks get in (schedule.hip)
        if (st == RP_OK && hipEventCreateWithFlags(&p->done[j], hipEventDisableTiming) != hipSuccess) st = fail(RP_ERR_DEVICE, "hipEventCreate failed");
        if (st == RP_OK && hipEventCreateWithFlags(&p->consumed[j], hipEventDisableTiming) != hipSuccess) st = fail(RP_ERR_DEVICE, "hipEventCreate failed");
    }
    if (st == RP_OK && hipEventCreateWithFlags(&p->inputs_ready, hipEventDisableTiming) != hipSuccess) st = fail(RP_ERR_DEVICE, "hipEventCreate failed");
    // (default arrangement: the scheduling pass on the job's own stream.  A stream of its own for it -- rp_pipeline_set_prep -- measured the
    // same to 1 % in a process that owns few streams and WORSE in one that owns many: the HIP runtime multiplexes streams onto a handful of
    // hardware queues, GPU_MAX_HW_QUEUES = 4 by default, and two streams that share a queue serialise: profiles/r6_tuning.md)
    if (st != RP_OK) {
        char keep[sizeof g_err];
        std::memcpy(keep, g_err, sizeof keep);
        rp_pipeline_destroy(p);
        std::memcpy(g_err, keep, sizeof keep);
        return st;
    }
    *out = p;
    return RP_OK;
}

int rp_pipeline_destroy(rp_pipeline *p)
{
    if (!p) return RP_OK;
    (void)hipSetDevice(p->device);
    if (p->slots) {
        // batches that share a stream must go before the batch that owns it (slot j < n_streams owns stream j)
        for (int j = p->depth - 1; j >= 0; --j) if (p->slots[j]) rp_batch_destroy(p->slots[j]);
    }
    for (int j = 0; j < p->depth; ++j) {
        if (p->done && p->done[j]) (void)hipEventDestroy(p->done[j]);
        if (p->consumed && p->consumed[j]) (void)hipEventDestroy(p->consumed[j]);
    }
    if (p->inputs_ready) (void)hipEventDestroy(p->inputs_ready);
    for (int j = 0; j < p->n_prep; ++j) { (void)hipStreamSynchronize(p->prep[j]); (void)hipStreamDestroy(p->prep[j]); }
    delete[] p->slots;
    delete[] p->done;
    delete[] p->consumed;
    delete[] p->job_of;
    delete p;
    return RP_OK;
}

int rp_pipeline_set_prep(rp_pipeline *p, int mode)
{
    if (!p) return fail(RP_ERR_INVALID, "null pipeline handle");
    const bool fat = (mode & RP_PIPELINE_PREP_FAT_KERNELS) != 0;      // A/B: the 256-thread form of the pass whatever the arrangement
    mode &= ~RP_PIPELINE_PREP_FAT_KERNELS;
    if (mode != RP_PIPELINE_PREP_INLINE && mode != RP_PIPELINE_PREP_STREAM && mode != RP_PIPELINE_PREP_PRIORITY)
        return fail(RP_ERR_INVALID, "prep mode %d (want 0 = on the job's stream, 1 = a stream of its own, 2 = ... with the highest priority)", mode);
    if (p->next_job != 0) return fail(RP_ERR_INVALID, "the arrangement is fixed once a job has been submitted");
    RP_HIP(hipSetDevice(p->device));
    for (int j = 0; j < p->n_prep; ++j) (void)hipStreamDestroy(p->prep[j]);
    p->n_prep = 0;
    p->prep_mode = mode;
    for (int j = 0; j < p->depth; ++j) p->slots[j]->slim_schedule = !fat && (p->n_streams > 1 || mode != RP_PIPELINE_PREP_INLINE);
    if (mode == RP_PIPELINE_PREP_INLINE) return RP_OK;
    int least = 0, greatest = 0;
    if (mode == RP_PIPELINE_PREP_PRIORITY) RP_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    // ONE prep stream: beside a running solve a pass takes about as long as the solve itself (three dependent, latency-bound kernels that
    // share every SIMD with four solve waves), which one stream just sustains; a prep stream per solve stream was measured and is no
    // faster -- two passes at once only contend (profiles/r6_pipeline_probe_two_prep_streams.log)
    RP_HIP(hipStreamCreateWithPriority(&p->prep[0], hipStreamNonBlocking, mode == RP_PIPELINE_PREP_PRIORITY ? greatest : 0));
    p->n_prep = 1;
    return RP_OK;
}

int rp_pipeline_set_params(rp_pipeline *p, const rp_params *params)
{
    if (!p || !params) return fail(RP_ERR_INVALID, "null argument");
    for (int j = 0; j < p->depth; ++j) {
        const int st = rp_batch_set_params(p->slots[j], params);
        if (st != RP_OK) return st;
    }
    return RP_OK;
}

int rp_pipeline_submit(rp_pipeline *p, const double *d_pos0, const double *d_pos1, const double *d_pos2, rp_solution *d_out,
                       double gap_tol, int max_iter, void *inputs_stream, int64_t *job)
{
    if (!p) return fail(RP_ERR_INVALID, "null pipeline handle");
    RP_HIP(hipSetDevice(p->device));
    const int64_t id = p->next_job;
    const int slot = (int)(id % p->depth);
    rp_batch *b = p->slots[slot];
    // The scheduling pass runs on the job's own stream or -- prep_mode -- on the pipeline's prep stream, behind the slot's previous job
    // (it overwrites the batch's order and records) and ahead of this job's solve (which waits for `consumed`).
    hipStream_t prep = p->n_prep ? p->prep[id % p->n_prep] : nullptr;
    hipStream_t solve_stream = b->stream, sched_stream = prep ? prep : b->stream;
    if (inputs_stream && (hipStream_t)inputs_stream != sched_stream) {      // the positions are produced by work on the caller's stream: wait for it, on the device
        RP_HIP(hipEventRecord(p->inputs_ready, (hipStream_t)inputs_stream));
        RP_HIP(hipStreamWaitEvent(sched_stream, p->inputs_ready, 0));
    }
    int st = rp_batch_bind_solution(b, d_out);
    if (st != RP_OK) return st;
    if (prep) {
        if (p->job_of[slot] >= 0) RP_HIP(hipStreamWaitEvent(prep, p->done[slot], 0));
        b->stream = prep;
        st = rp_batch_set_problems_device(b, d_pos0, d_pos1, d_pos2);
        b->stream = solve_stream;
    } else {
        st = rp_batch_set_problems_device(b, d_pos0, d_pos1, d_pos2);
    }
    if (st != RP_OK) return st;
    RP_HIP(hipEventRecord(p->consumed[slot], sched_stream));
    if (prep) RP_HIP(hipStreamWaitEvent(solve_stream, p->consumed[slot], 0));
    st = rp_batch_solve(b, gap_tol, max_iter, 0);
    if (st != RP_OK) return st;
    RP_HIP(hipEventRecord(p->done[slot], b->stream));
    p->job_of[slot] = id;
    p->next_job = id + 1;
    if (job) *job = id;
    return RP_OK;
}

static int pipeline_slot_of(rp_pipeline *p, int64_t job, int *slot)
{
    if (!p) return fail(RP_ERR_INVALID, "null pipeline handle");
    if (job < 0 || job >= p->next_job) return fail(RP_ERR_INVALID, "job %lld has not been submitted", (long long)job);
    *slot = (int)(job % p->depth);
    if (p->job_of[*slot] != job) return fail(RP_ERR_INVALID, "job %lld has left the pipeline: its slot holds job %lld", (long long)job, (long long)p->job_of[*slot]);
    return RP_OK;
}

int rp_pipeline_wait(rp_pipeline *p, int64_t job)
{
    if (!p) return fail(RP_ERR_INVALID, "null pipeline handle");
    RP_HIP(hipSetDevice(p->device));
    if (job < 0) {      // everything submitted so far
        for (int j = 0; j < p->n_prep; ++j) RP_HIP(hipStreamSynchronize(p->prep[j]));      // (first: a solve stream's last solve waits on them)
        for (int j = 0; j < p->n_streams; ++j) RP_HIP(hipStreamSynchronize(p->streams[j]));
        return RP_OK;
    }
    if (job >= p->next_job) return fail(RP_ERR_INVALID, "job %lld has not been submitted", (long long)job);
    const int slot = (int)(job % p->depth);
    RP_HIP(hipEventSynchronize(p->done[slot]));      // (the slot's LAST job: a later job of the same slot follows the asked one on one stream)
    return RP_OK;
}

int rp_pipeline_stream_wait(rp_pipeline *p, int64_t job, int what, void *stream)
{
    int slot = 0;
    const int st = pipeline_slot_of(p, job, &slot);
    if (st != RP_OK) return st;
    if (what != 0 && what != 1) return fail(RP_ERR_INVALID, "what = %d (0: the job's positions have been read, 1: its solutions are written)", what);
    RP_HIP(hipSetDevice(p->device));
    RP_HIP(hipStreamWaitEvent((hipStream_t)stream, what == 0 ? p->consumed[slot] : p->done[slot], 0));
    return RP_OK;
}

int rp_pipeline_batch(rp_pipeline *p, int64_t job, rp_batch **batch)
{
    if (!batch) return fail(RP_ERR_INVALID, "null output");
    int slot = 0;
    const int st = pipeline_slot_of(p, job, &slot);
    if (st != RP_OK) return st;
    *batch = p->slots[slot];
    return RP_OK;
}

}  // extern "C"
