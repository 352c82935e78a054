// sharded_problem.cpp -- see sharded_problem.h.
#include "sharded_problem.h"

#include <chrono>
#include <cstdio>

#include <rccl/rccl.h>

namespace {
inline int stateLen(int variant) { return variant == RP_VARIANT_F4 ? 12 : 16; }
inline int numConstraints(int variant) { return variant == RP_VARIANT_F4 ? 4 : 8; }
}  // namespace

ShardedOneDPathIP::ShardedOneDPathIP(size_t nTotal, int nDevices, int variant, int dtype)
    : nTotal_(nTotal), variant_(variant), ok_(false)
{
    int visible = 0;
    if (!check(rp_device_count(&visible), "rp_device_count")) return;
    if (nDevices < 1 || nDevices > visible || nTotal < (size_t)nDevices) {
        fprintf(stderr, "ShardedOneDPathIP: %d device(s) requested, %d visible, %zu problems\n", nDevices, visible, nTotal);
        return;
    }
    const size_t q = nTotal / nDevices, r = nTotal % nDevices;      // shard sizes differ by at most one
    for (int d = 0; d < nDevices; ++d) {
        const size_t first = d * q + ((size_t)d < r ? (size_t)d : r), count = q + ((size_t)d < r ? 1 : 0);
        rp_batch *b = nullptr;
        if (!check(rp_batch_create(&b, variant, dtype, count, d, nullptr), "rp_batch_create")) return;
        shards_.push_back(b);
        first_.push_back(first);
        count_.push_back(count);
    }
    std::vector<int> devs(nDevices);
    std::vector<ncclComm_t> comms(nDevices);
    for (int d = 0; d < nDevices; ++d) devs[d] = d;
    const ncclResult_t rc = ncclCommInitAll(comms.data(), nDevices, devs.data());
    if (rc != ncclSuccess) {
        fprintf(stderr, "ShardedOneDPathIP: ncclCommInitAll failed: %s\n", ncclGetErrorString(rc));
        return;
    }
    for (ncclComm_t c : comms) comms_.push_back((void *)c);
    ok_ = true;
}

ShardedOneDPathIP::~ShardedOneDPathIP()
{
    syncAll();
    for (void *c : comms_) ncclCommDestroy((ncclComm_t)c);
    for (rp_batch *b : shards_) rp_batch_destroy(b);
}

bool ShardedOneDPathIP::check(int status, const char *what)
{
    if (status == RP_OK) return true;
    fprintf(stderr, "ShardedOneDPathIP: %s failed: %s (%s)\n", what, rp_status_string(status), rp_last_error());
    return false;
}

void ShardedOneDPathIP::syncAll()
{
    for (rp_batch *b : shards_) rp_batch_sync(b);
}

void ShardedOneDPathIP::init()
{
    for (rp_batch *b : shards_) check(rp_batch_init_default(b), "rp_batch_init_default");
}

void ShardedOneDPathIP::onActivate()
{
    printf("\n1D path: Interior point, %zu problems sharded over %zu GPU(s), one RCCL all-reduce per summary\n", nTotal_, shards_.size());
}

void ShardedOneDPathIP::onKey(unsigned char key)
{
    if (!ok_) return;
    switch (key) {
    case ' ':
        for (rp_batch *b : shards_) check(rp_batch_move_toward_feasibility(b), "rp_batch_move_toward_feasibility");
        break;
    case 'i':
        init();
        break;
    case 'j':
        if (variant_ == RP_VARIANT_F3)
            for (rp_batch *b : shards_) check(rp_batch_init_stuck(b), "rp_batch_init_stuck");
        break;
    case 'n':
        step(1);
        break;
    case 's': {
        rp_reduction r;
        if (reduce(r))
            printf("Batch: %zu problems on %zu GPU(s), max surrogate gap: %g, max residual^2: %g, converged: %.0f, steps: %.0f\n",
                   nTotal_, shards_.size(), r.max_gap, r.max_residual_sq, r.n_converged, r.total_steps);
        break;
    }
    }
}

void ShardedOneDPathIP::onSpecialKey(int key)
{
    if (!ok_) return;
    const int pos1 = 3 + numConstraints(variant_) + 2;
    int index = -1;
    double delta = 0;
    switch (key) {
    case RP_KEY_END:       index = 1; delta = -0.1; break;
    case RP_KEY_HOME:      index = 1; delta = 0.1; break;
    case RP_KEY_PAGE_DOWN: index = 2; delta = -0.1; break;
    case RP_KEY_PAGE_UP:   index = 2; delta = 0.1; break;
    case RP_KEY_LEFT:      index = 0; delta = -1.0; break;
    case RP_KEY_RIGHT:     index = 0; delta = 1.0; break;
    case RP_KEY_UP:        index = pos1; delta = 10.0; break;
    case RP_KEY_DOWN:      index = pos1; delta = -10.0; break;
    default: return;
    }
    for (rp_batch *b : shards_) check(rp_batch_nudge(b, index, delta), "rp_batch_nudge");
}

void ShardedOneDPathIP::setProblems(const double *pos0, const double *pos1, const double *pos2)
{
    if (!ok_) return;
    for (size_t d = 0; d < shards_.size(); ++d)
        check(rp_batch_set_problems(shards_[d], pos0 + first_[d], pos1 + first_[d], pos2 + first_[d]), "rp_batch_set_problems");
}

void ShardedOneDPathIP::step(int k)
{
    if (!ok_) return;
    for (rp_batch *b : shards_) check(rp_batch_step(b, k), "rp_batch_step");      // every device is launched ...
    syncAll();                                                                    // ... before any is waited for
}

void ShardedOneDPathIP::solve(double gapTol, int maxIter)
{
    if (!ok_) return;
    for (rp_batch *b : shards_) check(rp_batch_solve(b, gapTol, maxIter, 0), "rp_batch_solve");
    syncAll();
}

void ShardedOneDPathIP::restart()
{
    if (!ok_) return;
    for (rp_batch *b : shards_) check(rp_batch_restart(b), "rp_batch_restart");
}

bool ShardedOneDPathIP::bench(int passes, int warmup, double gapTol, int maxIter, BenchResult &out)
{
    if (!ok_ || passes < 1 || warmup < 0) return false;
    rp_reduction r;
    for (int w = 0; w < warmup; ++w) {
        restart();
        for (rp_batch *b : shards_) check(rp_batch_solve(b, gapTol, maxIter, 0), "rp_batch_solve");
    }
    if (!reduce(r)) return false;                 // communicator and summary path warmed up outside the timed region
    syncAll();
    const auto t0 = std::chrono::steady_clock::now();
    for (rp_batch *b : shards_) check(rp_batch_event_record(b, 0), "rp_batch_event_record");
    for (int k = 0; k < passes; ++k) {            // every device gets its whole queue before any is waited for
        for (rp_batch *b : shards_) {
            check(rp_batch_restart(b), "rp_batch_restart");
            check(rp_batch_solve(b, gapTol, maxIter, 0), "rp_batch_solve");
        }
    }
    for (rp_batch *b : shards_) check(rp_batch_event_record(b, 1), "rp_batch_event_record");
    const bool good = reduce(r);                  // the one collective: reduction kernels + the grouped 32-byte all-reduce + read-back
    syncAll();
    out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    out.msPerPass = out.seconds * 1e3 / passes;
    out.deviceMs.clear();
    for (rp_batch *b : shards_) {
        float ms = 0.f;
        check(rp_batch_event_elapsed_ms(b, 0, 1, &ms), "rp_batch_event_elapsed_ms");
        out.deviceMs.push_back(ms / passes);
    }
    // every pass solves the same problems from the same start: steps of the last pass (what the summary counts) x passes
    out.stepsTotal = r.total_steps * passes;
    out.converged = r.n_converged;
    out.maxGap = r.max_gap;
    out.maxResidualSq = r.max_residual_sq;
    return good;
}

bool ShardedOneDPathIP::reduce(rp_reduction &out)
{
    if (!ok_) return false;
    std::vector<double *> slots(shards_.size());
    std::vector<void *> streams(shards_.size());
    for (size_t d = 0; d < shards_.size(); ++d) {
        if (!check(rp_batch_summary_device(shards_[d], &slots[d]), "rp_batch_summary_device")) return false;
        if (!check(rp_batch_stream(shards_[d], &streams[d]), "rp_batch_stream")) return false;
    }
    // the one collective of the path: 32 bytes per device over xGMI
    ncclResult_t rc = ncclGroupStart();
    for (size_t d = 0; d < shards_.size() && rc == ncclSuccess; ++d) {
        rc = ncclAllReduce(slots[d], slots[d], 2, ncclDouble, ncclMax, (ncclComm_t)comms_[d], (hipStream_t)streams[d]);
        if (rc == ncclSuccess)
            rc = ncclAllReduce(slots[d] + 2, slots[d] + 2, 2, ncclDouble, ncclSum, (ncclComm_t)comms_[d], (hipStream_t)streams[d]);
    }
    const ncclResult_t rc2 = ncclGroupEnd();
    if (rc != ncclSuccess || rc2 != ncclSuccess) {
        fprintf(stderr, "ShardedOneDPathIP: all-reduce failed: %s\n", ncclGetErrorString(rc != ncclSuccess ? rc : rc2));
        return false;
    }
    bool good = true;
    rp_reduction each;
    for (size_t d = 0; d < shards_.size(); ++d) {
        good = check(rp_batch_summary_read(shards_[d], &each), "rp_batch_summary_read") && good;
        if (d == 0) out = each;
    }
    return good;
}

bool ShardedOneDPathIP::readState(std::vector<double> &aos)
{
    if (!ok_) return false;
    const size_t M = (size_t)stateLen(variant_);
    aos.resize(nTotal_ * M);
    bool good = true;
    for (size_t d = 0; d < shards_.size(); ++d)
        good = check(rp_batch_get_state(shards_[d], aos.data() + first_[d] * M), "rp_batch_get_state") && good;
    return good;
}
