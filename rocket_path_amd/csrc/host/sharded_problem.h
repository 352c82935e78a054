// sharded_problem.h -- one Problem, N GPUs, one process.
//
// ShardedOneDPathIP is BatchedOneDPathIP spread over the GPUs of a node for hosts that are a single
// process (rocket_path.cpp is one): device d owns the contiguous shard d of the problems as its
// own rp_batch on its own stream; a key press is forwarded to every shard and nothing is exchanged while
// stepping.  The only collective is the batch summary: each shard reduces into its own 4-double device slot
// and RCCL all-reduces the slots in place over xGMI (MAX on {max ||r||^2, max gap}, SUM on {converged,
// steps}), one ncclGroup per call.  (Python hosts use one process per GPU and torch.distributed instead:
// rocket_path_amd/sharding.py.)
#pragma once

#include <cstddef>
#include <vector>

#include "problem_iface.h"
#include "rp_batch.h"

struct ShardedOneDPathIP : public Problem {
    ShardedOneDPathIP(size_t nTotal, int nDevices, int variant = RP_VARIANT_F3, int dtype = RP_DTYPE_F64);
    ~ShardedOneDPathIP() override;

    void init() override;                         // initDefault on every problem of every shard
    void onActivate() override;
    void onKey(unsigned char key) override;       // ' ', 'i', 'j', 'n', 's' as in onedpath_ip.cpp:250-278
    void onSpecialKey(int key) override;          // nudges, onedpath_ip.cpp:280-324
    void onDraw() override {}
    void onMouseMove(int, int) override {}
    void onMouseDown() override {}
    void onMouseUp() override {}

    bool ok() const { return ok_; }
    size_t size() const { return nTotal_; }
    int devices() const { return (int)shards_.size(); }
    void setProblems(const double *pos0, const double *pos1, const double *pos2);   // nTotal values each
    void step(int k);
    void solve(double gapTol = 1e-8, int maxIter = 200);
    bool reduce(rp_reduction &out);               // the all-reduced summary (identical on every device)
    void restart();                               // feasible start again from the positions the shards hold (device side only)
    // K timed passes (restart + fused gated solve on every shard) after W untimed ones, then ONE all-reduced summary; per-device
    // times come from HIP events on each shard's stream, the job's time is the slowest device's.  Fills the numbers bench.py prints.
    struct BenchResult { double seconds, msPerPass, stepsTotal, converged, maxGap, maxResidualSq; std::vector<double> deviceMs; };
    bool bench(int passes, int warmup, double gapTol, int maxIter, BenchResult &out);
    bool readState(std::vector<double> &aos);     // concatenation of the shards, problem order

private:
    bool check(int status, const char *what);
    void syncAll();
    std::vector<rp_batch *> shards_;
    std::vector<size_t> first_, count_;
    std::vector<void *> comms_;                   // ncclComm_t per device
    size_t nTotal_;
    int variant_;
    bool ok_;
};
