// batched_problem.cpp -- see batched_problem.h.  Like the reference, nothing here returns an
// error to the caller (the interface is all-void); failures of the C ABI are reported on
// stderr with rp_last_error() and the call becomes a no-op.
#include "batched_problem.h"

#include <cstdio>

// Every handled key of the reference ends in repaint() (onedpath_ip.cpp:256, 261, 266, 271; the special keys 280-324),
// which posts a GLUT redisplay and so brings onDraw() round (rocket_path.cpp:101-106, 178-182).  Inside the reference
// tree that very function is called (declared in draw.h:6; not included here because draw.h needs the GL headers);
// stand-alone hosts register a hook instead.
#ifdef RP_USE_REFERENCE_PROBLEM_H
void repaint();
#endif

namespace {
inline int numConstraints(int variant) { return variant == RP_VARIANT_F4 ? 4 : 8; }
inline int stateLen(int variant) { return variant == RP_VARIANT_F4 ? 12 : 16; }
}  // namespace

BatchedOneDPathIP::BatchedOneDPathIP(size_t n, int variant, int dtype, int device)
    : batch_(nullptr), n_(n), watched_(0), variant_(variant), repaintHook_(nullptr)
{
    // the library this was linked against must speak the header this was compiled against (rp_batch.h, RP_ABI_VERSION)
    if (rp_abi_version() != RP_ABI_VERSION || rp_params_size() != sizeof(rp_params)) {
        fprintf(stderr, "BatchedOneDPathIP: librp_batch.so has ABI revision %d (rp_params %zu B), this host was compiled for %d (%zu B)\n",
                rp_abi_version(), rp_params_size(), RP_ABI_VERSION, sizeof(rp_params));
        return;      // batch_ stays null: every call becomes a no-op, as after any other failure to create
    }
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

void BatchedOneDPathIP::requestRepaint()
{
#ifdef RP_USE_REFERENCE_PROBLEM_H
    repaint();
#else
    if (repaintHook_) repaintHook_();
#endif
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
        requestRepaint();
        break;
    case 'i':
        check(rp_batch_init_default(batch_), "rp_batch_init_default");
        requestRepaint();
        break;
    case 'j':
        if (variant_ == RP_VARIANT_F3) {
            check(rp_batch_init_stuck(batch_), "rp_batch_init_stuck");
            requestRepaint();
        }
        break;
    case 'n':
        step(1);
        requestRepaint();
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
    default: return;      // the reference repaints only after a key it handles (onedpath_ip.cpp:280-324)
    }
    requestRepaint();
}

void BatchedOneDPathIP::onDraw()
{
    if (!batch_) return;
    // the watched problem only: 70 doubles cross PCIe, whatever the batch size
    plotPos_.resize(66);
    plotAcc_.resize(4);
    check(rp_batch_sample_range(batch_, watched_, 1, plotPos_.data(), plotAcc_.data()), "rp_batch_sample_range");
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

// printState of the watched problem, line for line the reference's (onedpath_ip.cpp:997-1010 with printConstraints
// 955-995; F4: onedpath2_ip.cpp:885-898), followed by the batch summary that replaces the reference's per-step dump
// and a full-precision line for tools.  Only the watched problem's rows cross PCIe.
void BatchedOneDPathIP::printState()
{
    if (!batch_) return;
    const int m = numConstraints(variant_), M = stateLen(variant_);
    std::vector<double> row(M), table(1 + 14 * m);
    if (!check(rp_batch_get_state_range(batch_, watched_, 1, row.data()), "rp_batch_get_state_range")) return;
    if (!check(rp_batch_constraints_range(batch_, watched_, 1, table.data()), "rp_batch_constraints_range")) return;
    const double *v = row.data();
    const double *c = v + 3 + m;   // pos0, vel0, pos1, pos2, vel2
    printf("\nNode 0: pos=%g vel=%g\n", c[0], c[1]);
    printf("Node 1: pos=%g vel=%g\n", c[2], v[0]);
    printf("Node 2: pos=%g vel=%g\n", c[3], c[4]);
    printf("Duration 0: %g\n", v[1]);
    printf("Duration 1: %g\n", v[2]);
    printf("Constraint Multipliers:");
    for (int i = 0; i < m; ++i) printf(" %g", v[3 + i]);
    printf("\n");
    printf("Surrogate gap: %g\n", table[0]);
    printf("Constraints:\n");
    if (variant_ == RP_VARIANT_F4) {      // F4's table has a header naming the columns (onedpath2_ip.cpp:848-863)
        static const char *const name[3] = {"v1", "t0", "t1"};
        printf("[%s %s %s] [", name[0], name[1], name[2]);
        for (int i = 0; i < 3; ++i) printf("%s[%s/%s %s/%s %s/%s]", (i > 0) ? " " : "", name[i], name[0], name[i], name[1], name[i], name[2]);
        printf("]\n");
    }
    for (int i = 0; i < m; ++i) {
        const double *t = table.data() + 1 + 14 * i;      // error, deriv[3], second[3][3], dot
        printf("%c%u:", (t[0] > 0) ? '*' : ' ', (unsigned)i);
        printf(" error=%g derivs=[", t[0]);
        for (int j = 0; j < 3; ++j) printf("%s%g", (j > 0) ? " " : "", t[1 + j]);
        printf("] second=[");
        for (int j = 0; j < 3; ++j) {
            printf("%s[", (j > 0) ? " " : "");
            for (int k = 0; k < 3; ++k) printf("%s%g", (k > 0) ? " " : "", t[4 + 3 * j + k]);
            printf("]");
        }
        printf("] dot=%g\n", t[13]);
    }
    rp_reduction r;
    if (reduce(r)) {
        printf("Batch: %zu problems, max surrogate gap: %g, max residual^2: %g, converged: %.0f, steps: %.0f\n", n_, r.max_gap,
               r.max_residual_sq, r.n_converged, r.total_steps);
    }
    printf("State17:");
    for (int i = 0; i < 3 + m; ++i) printf(" %.17g", v[i]);
    printf("\n");
}
