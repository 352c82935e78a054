// batched_problem.cpp -- see batched_problem.h.  Like the reference, nothing here returns an
// error to the caller (the interface is all-void); failures of the C ABI are reported on
// stderr with rp_last_error() and the call becomes a no-op.
#include "batched_problem.h"

#include <cstdio>

namespace {
inline int numConstraints(int variant) { return variant == RP_VARIANT_F4 ? 4 : 8; }
inline int stateLen(int variant) { return variant == RP_VARIANT_F4 ? 12 : 16; }
}  // namespace

BatchedOneDPathIP::BatchedOneDPathIP(size_t n, int variant, int dtype, int device)
    : batch_(nullptr), n_(n), watched_(0), variant_(variant)
{
    check(rp_batch_create(&batch_, variant, dtype, n, device, nullptr), "rp_batch_create");
}

BatchedOneDPathIP::~BatchedOneDPathIP()
{
    rp_batch_destroy(batch_);
}

bool BatchedOneDPathIP::check(int status, const char *what)
{
    if (status == RP_OK) return true;
    fprintf(stderr, "BatchedOneDPathIP: %s failed: %s (%s)\n", what, rp_status_string(status), rp_last_error());
    return false;
}

void BatchedOneDPathIP::init()
{
    if (batch_) check(rp_batch_init_default(batch_), "rp_batch_init_default");
}

void BatchedOneDPathIP::onActivate()
{
    printf("\n1D path: Interior point, %s, batch of %zu problems on the GPU\n\n"
           "Space      Move toward feasibility\n"
           "I          Reinitialize\n"
           "%s"
           "N          Take an interior-point step\n"
           "S          Print current state\n"
           "Home/End   Increment/Decrement segment 0 duration\n"
           "PgUp/PgDn  Increment/Decrement segment 1 duration\n"
           "Right/Left Increment/Decrement midpoint velocity\n"
           "Up/Down    Increment/Decrement midpoint position\n",
           variant_ == RP_VARIANT_F4 ? "squared distance constraints" : "lower/upper acceleration constraints", n_,
           variant_ == RP_VARIANT_F4 ? "" : "J          Reinitialize to a stuck state\n");
}

void BatchedOneDPathIP::onKey(unsigned char key)
{
    if (!batch_) return;
    switch (key) {
    case ' ':
        check(rp_batch_move_toward_feasibility(batch_), "rp_batch_move_toward_feasibility");
        break;
    case 'i':
        check(rp_batch_init_default(batch_), "rp_batch_init_default");
        break;
    case 'j':
        if (variant_ == RP_VARIANT_F3) check(rp_batch_init_stuck(batch_), "rp_batch_init_stuck");
        break;
    case 'n':
        step(1);
        break;
    case 's':
        printState();
        break;
    }
}

void BatchedOneDPathIP::onSpecialKey(int key)
{
    if (!batch_) return;
    const int pos1 = 3 + numConstraints(variant_) + 2;   // pos1X in enum V / V2
    switch (key) {
    case RP_KEY_END:       check(rp_batch_nudge(batch_, 1, -0.1), "nudge"); break;   // duration0
    case RP_KEY_HOME:      check(rp_batch_nudge(batch_, 1, 0.1), "nudge"); break;
    case RP_KEY_PAGE_DOWN: check(rp_batch_nudge(batch_, 2, -0.1), "nudge"); break;   // duration1
    case RP_KEY_PAGE_UP:   check(rp_batch_nudge(batch_, 2, 0.1), "nudge"); break;
    case RP_KEY_LEFT:      check(rp_batch_nudge(batch_, 0, -1.0), "nudge"); break;   // vel1X
    case RP_KEY_RIGHT:     check(rp_batch_nudge(batch_, 0, 1.0), "nudge"); break;
    case RP_KEY_UP:        check(rp_batch_nudge(batch_, pos1, 10.0), "nudge"); break;
    case RP_KEY_DOWN:      check(rp_batch_nudge(batch_, pos1, -10.0), "nudge"); break;
    }
}

void BatchedOneDPathIP::onDraw()
{
    if (!batch_) return;
    std::vector<double> pos(n_ * 66), acc(n_ * 4);
    if (!check(rp_batch_sample(batch_, pos.data(), acc.data()), "rp_batch_sample")) return;
    plotPos_.assign(pos.begin() + watched_ * 66, pos.begin() + (watched_ + 1) * 66);
    plotAcc_.assign(acc.begin() + watched_ * 4, acc.begin() + (watched_ + 1) * 4);
}

void BatchedOneDPathIP::setProblems(const double *pos0, const double *pos1, const double *pos2)
{
    if (batch_) check(rp_batch_set_problems(batch_, pos0, pos1, pos2), "rp_batch_set_problems");
}

void BatchedOneDPathIP::step(int k)
{
    if (!batch_) return;
    if (check(rp_batch_step(batch_, k), "rp_batch_step")) check(rp_batch_sync(batch_), "rp_batch_sync");
}

void BatchedOneDPathIP::solve(double gapTol, int maxIter)
{
    if (!batch_) return;
    if (check(rp_batch_solve(batch_, gapTol, maxIter, 0), "rp_batch_solve")) check(rp_batch_sync(batch_), "rp_batch_sync");
}

bool BatchedOneDPathIP::readState(std::vector<double> &aos)
{
    if (!batch_) return false;
    aos.resize(n_ * stateLen(variant_));
    return check(rp_batch_get_state(batch_, aos.data()), "rp_batch_get_state");
}

bool BatchedOneDPathIP::readIters(std::vector<int32_t> &iters, std::vector<uint32_t> &status)
{
    if (!batch_) return false;
    iters.resize(n_);
    status.resize(n_);
    return check(rp_batch_get_iters(batch_, iters.data(), status.data()), "rp_batch_get_iters");
}

bool BatchedOneDPathIP::reduce(rp_reduction &out)
{
    return batch_ && check(rp_batch_reduce(batch_, &out), "rp_batch_reduce");
}

// printState of the watched problem in the reference's format (onedpath_ip.cpp:997-1010),
// followed by the batch summary that replaces the reference's per-step dump.
void BatchedOneDPathIP::printState()
{
    std::vector<double> aos;
    if (!readState(aos)) return;
    const int m = numConstraints(variant_), M = stateLen(variant_);
    const double *v = aos.data() + watched_ * M;
    const double *c = v + 3 + m;   // pos0, vel0, pos1, pos2, vel2
    printf("\nNode 0: pos=%g vel=%g\n", c[0], c[1]);
    printf("Node 1: pos=%g vel=%g\n", c[2], v[0]);
    printf("Node 2: pos=%g vel=%g\n", c[3], c[4]);
    printf("Duration 0: %g\n", v[1]);
    printf("Duration 1: %g\n", v[2]);
    printf("Constraint Multipliers:");
    for (int i = 0; i < m; ++i) printf(" %g", v[3 + i]);
    printf("\n");
    rp_reduction r;
    if (reduce(r)) {
        printf("Batch: %zu problems, max surrogate gap: %g, max residual^2: %g, converged: %.0f, steps: %.0f\n", n_, r.max_gap,
               r.max_residual_sq, r.n_converged, r.total_steps);
    }
    printf("State17:");
    for (int i = 0; i < 3 + m; ++i) printf(" %.17g", v[i]);
    printf("\n");
}
