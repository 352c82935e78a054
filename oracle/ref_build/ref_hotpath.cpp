// ref_hotpath.cpp -- harness around the reference's OWN hot-path functions.
//
// TEST INFRASTRUCTURE ONLY.  This TU is ours; the code it wraps is the reference's, taken from where it lies:
// oracle/Makefile pipes the GL-free line ranges of /root/reference/onedpath_ip.cpp (or onedpath2_ip.cpp) into the
// compiler's standard input, and the `#include "/dev/stdin"` below is where they enter this TU -- no reference text is
// written to disk anywhere (not in the repo, not in oracle/_ref/, not in /tmp); only the compiled library
// oracle/_ref/libref_hotpath.so exists afterwards.  The ranges (F3; F4 in brackets):
//     7-167     [7-154]     #undef min/max, <Eigen/Dense>, <algorithm>, enum V, struct Trajectory, tables, prototypes
//     177-228   [164-194]   initStuck, initDefault
//     372-1013  [332-919]   evalAccel* ... evalConstraint* ... moveTowardFeasibility, trajectoryStep,
//                           constraintsSatisfied, residual, residualNorm, surrogateDualityGap, moveInteriorPoint,
//                           printConstraints, printState
// i.e. everything of the file except the Problem subclass's GLUT glue (ctor/dtor, init/onKey/onDraw...: 169-176, 230-370)
// and the two plot functions (1015-1148), which are the only users of <GL/glut.h> and draw.h.  The headers are the REAL ones
// (<cstdio>, <Eigen/Dense> from /root/reference/libs/eigen, ...): nothing is stood in for, no macro touches printf.
// moveInteriorPoint prints its KKT matrix on every call (onedpath_ip.cpp:865-899); the batch entry points below send
// file descriptor 1 to /dev/null while they run.
//
// Built once per variant (-DRP_REF_VARIANT=3 / 4; the two reference files share their global names, so they cannot share
// a TU) and linked into one library; everything the reference defines has internal linkage, the exports are ref3_* / ref4_*.
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include <fcntl.h>
#include <unistd.h>

#include "/dev/stdin"      // <- the reference's own lines, see above

#if RP_REF_VARIANT == 3
typedef Trajectory RefTraj;
enum { kStateLen = M };
#define RP_REF(name) ref3_##name
#else
typedef Trajectory2 RefTraj;
enum { kStateLen = M2 };
#define RP_REF(name) ref4_##name
#endif

namespace {

// fd 1 -> /dev/null for the lifetime of the object (the reference's per-step printf's)
struct Silence {
    int saved;
    Silence()
    {
        fflush(stdout);
        saved = dup(1);
        const int nul = open("/dev/null", O_WRONLY);
        if (nul >= 0) { dup2(nul, 1); close(nul); }
    }
    ~Silence()
    {
        fflush(stdout);
        if (saved >= 0) { dup2(saved, 1); close(saved); }
    }
};

inline RefTraj load(const double *var)
{
    RefTraj t;
    std::memcpy(t.var, var, sizeof t.var);
    return t;
}

}  // namespace

extern "C" {

int RP_REF(state_len)(void) { return (int)kStateLen; }
int RP_REF(num_constraints)(void) { return (int)numConstraints; }

void RP_REF(init_default)(double *var)
{
    RefTraj t;
    initDefault(t);
    std::memcpy(var, t.var, sizeof t.var);
}

#if RP_REF_VARIANT == 3
void ref3_init_stuck(double *var)
{
    RefTraj t;
    initStuck(t);
    std::memcpy(var, t.var, sizeof t.var);
}
#endif

double RP_REF(gap)(const double *var) { return surrogateDualityGap(load(var)); }
double RP_REF(residual_norm)(const double *var, double p) { return residualNorm(load(var), p); }
int RP_REF(constraints_satisfied)(const double *var) { return constraintsSatisfied(load(var)) ? 1 : 0; }

void RP_REF(constraint)(int i, const double *var, double *error, double *deriv3)
{
    Matrix<double, numVars, 1> g;
    (*constraints[i])(load(var), *error, g);
    for (size_t j = 0; j < numVars; ++j) deriv3[j] = g(j);
}

// row-major 3x3
void RP_REF(constraint_hess)(int i, const double *var, double *h9)
{
    Matrix<double, numVars, numVars> h;
    (*constraintSecondDerivs[i])(load(var), h);
    for (size_t r = 0; r < numVars; ++r)
        for (size_t c = 0; c < numVars; ++c) h9[3 * r + c] = h(r, c);
}

// one moveInteriorPoint (the 'n' key), stdout silenced
void RP_REF(step)(double *var)
{
    Silence quiet;
    RefTraj t = load(var);
    moveInteriorPoint(t);
    std::memcpy(var, t.var, sizeof t.var);
}

// the Space key
void RP_REF(move_toward_feasibility)(double *var)
{
    Silence quiet;
    RefTraj t = load(var);
    moveTowardFeasibility(t);
    std::memcpy(var, t.var, sizeof t.var);
}

// printState to the real stdout (callers capture file descriptor 1)
void RP_REF(print_state)(const double *var)
{
    printState(load(var));
    fflush(stdout);
}

// k presses of 'n' on each of n states (AoS rows of state_len doubles)
void RP_REF(batch_steps)(size_t n, double *aos, int k)
{
    Silence quiet;
    for (size_t i = 0; i < n; ++i) {
        RefTraj t = load(aos + i * kStateLen);
        for (int s = 0; s < k; ++s) moveInteriorPoint(t);
        std::memcpy(aos + i * kStateLen, t.var, sizeof t.var);
    }
}

// the gate convention of SURVEY.md appendix A.5; returns the total number of steps
int64_t RP_REF(batch_solve_gated)(size_t n, double *aos, double tol, int max_iter, int32_t *iters)
{
    Silence quiet;
    int64_t total = 0;
    for (size_t i = 0; i < n; ++i) {
        RefTraj t = load(aos + i * kStateLen);
        int it = 0;
        for (; it < max_iter; ++it) {
            if (surrogateDualityGap(t) < tol) break;
            moveInteriorPoint(t);
        }
        std::memcpy(aos + i * kStateLen, t.var, sizeof t.var);
        if (iters) iters[i] = it;
        total += it;
    }
    return total;
}

void RP_REF(batch_move_toward_feasibility)(size_t n, double *aos)
{
    Silence quiet;
    for (size_t i = 0; i < n; ++i) {
        RefTraj t = load(aos + i * kStateLen);
        moveTowardFeasibility(t);
        std::memcpy(aos + i * kStateLen, t.var, sizeof t.var);
    }
}

}  // extern "C"
