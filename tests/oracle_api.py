"""ctypes access to the CPU oracle (oracle/libip_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
`Oracle(eigen=True)` routes the step's linear solve through the reference's own vendored
Eigen QR (oracle/_ref/libeigen_qr_ref.so, built in the container that has /root/reference);
with that solver the restatement reproduces SURVEY.md 8c's known-answer vectors bit for bit.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libip_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libeigen_qr_ref.so")
REF_HOTPATH_SO = os.path.join(ORACLE_DIR, "_ref", "libref_hotpath.so")

_dp = ctypes.POINTER(ctypes.c_double)


class StepInfo(ctypes.Structure):
    _fields_ = [("feas_halvings", ctypes.c_int), ("resid_halvings", ctypes.c_int), ("nonzero_pivots", ctypes.c_int),
                ("step_scale", ctypes.c_double), ("perturbation", ctypes.c_double)]


def build_oracle():
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ORACLE_DIR, "ip_oracle.c")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libip_oracle.so"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


def have_ref():
    return os.path.exists(REF_SO)


def have_ref_hotpath():
    return os.path.exists(REF_HOTPATH_SO)


def _p(a):
    return a.ctypes.data_as(_dp)


class Reference:
    """The reference's OWN hot-path functions (onedpath_ip.cpp / onedpath2_ip.cpp), compiled in the build container from
    the files where they lie (oracle/_ref/libref_hotpath.so; recipe: `make -C oracle ref`, harness
    oracle/ref_build/ref_hotpath.cpp).  This is the pin of the oracle: the restatement is checked against it bit for bit.
    The library travels to the GPU box prebuilt, so the same class also serves as the `"reference"` CPU baseline there."""

    def __init__(self):
        if not have_ref_hotpath():
            raise RuntimeError("oracle/_ref/libref_hotpath.so is not built (make -C oracle ref, needs /root/reference)")
        self.lib = L = ctypes.CDLL(REF_HOTPATH_SO)
        i32p = ctypes.POINTER(ctypes.c_int32)
        for v in (3, 4):
            f = lambda name: getattr(L, "ref%d_%s" % (v, name))      # noqa: E731
            f("gap").restype = ctypes.c_double
            f("gap").argtypes = [_dp]
            f("residual_norm").restype = ctypes.c_double
            f("residual_norm").argtypes = [_dp, ctypes.c_double]
            f("constraints_satisfied").argtypes = [_dp]
            f("constraint").argtypes = [ctypes.c_int, _dp, _dp, _dp]
            f("constraint_hess").argtypes = [ctypes.c_int, _dp, _dp]
            f("init_default").argtypes = [_dp]
            f("step").argtypes = [_dp]
            f("move_toward_feasibility").argtypes = [_dp]
            f("print_state").argtypes = [_dp]
            f("batch_steps").argtypes = [ctypes.c_size_t, _dp, ctypes.c_int]
            f("batch_solve_gated").restype = ctypes.c_int64
            f("batch_solve_gated").argtypes = [ctypes.c_size_t, _dp, ctypes.c_double, ctypes.c_int, i32p]
            f("batch_move_toward_feasibility").argtypes = [ctypes.c_size_t, _dp]
        L.ref3_init_stuck.argtypes = [_dp]
        assert L.ref3_state_len() == 16 and L.ref4_state_len() == 12
        assert L.ref3_num_constraints() == 8 and L.ref4_num_constraints() == 4

    def _f(self, variant, name):
        return getattr(self.lib, "ref%d_%s" % (variant, name))

    def init_default(self, variant=3):
        v = np.zeros(Oracle.state_len(variant))
        self._f(variant, "init_default")(_p(v))
        return v

    def init_stuck(self):
        v = np.zeros(16)
        self.lib.ref3_init_stuck(_p(v))
        return v

    def gap(self, variant, var):
        return self._f(variant, "gap")(_p(np.ascontiguousarray(var, dtype=np.float64)))

    def residual_norm(self, variant, var, p):
        return self._f(variant, "residual_norm")(_p(np.ascontiguousarray(var, dtype=np.float64)), p)

    def satisfied(self, variant, var):
        return bool(self._f(variant, "constraints_satisfied")(_p(np.ascontiguousarray(var, dtype=np.float64))))

    def constraint(self, variant, i, var):
        e = np.zeros(1)
        g = np.zeros(3)
        self._f(variant, "constraint")(i, _p(np.ascontiguousarray(var, dtype=np.float64)), _p(e), _p(g))
        return e[0], g

    def constraint_hess(self, variant, i, var):
        h = np.zeros(9)
        self._f(variant, "constraint_hess")(i, _p(np.ascontiguousarray(var, dtype=np.float64)), _p(h))
        return h

    def step(self, variant, var):
        self._f(variant, "step")(_p(var))

    def move_toward_feasibility(self, variant, var):
        self._f(variant, "move_toward_feasibility")(_p(var))

    def batch_steps(self, variant, aos, k):
        assert aos.flags.c_contiguous and aos.dtype == np.float64
        self._f(variant, "batch_steps")(aos.shape[0], _p(aos), k)
        return aos

    def batch_solve_gated(self, variant, aos, tol=1e-8, max_iter=200):
        assert aos.flags.c_contiguous and aos.dtype == np.float64
        iters = np.zeros(aos.shape[0], dtype=np.int32)
        total = self._f(variant, "batch_solve_gated")(aos.shape[0], _p(aos), tol, max_iter,
                                                       iters.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        return iters, total

    def batch_move_toward_feasibility(self, variant, aos):
        assert aos.flags.c_contiguous and aos.dtype == np.float64
        self._f(variant, "batch_move_toward_feasibility")(aos.shape[0], _p(aos))
        return aos

    def print_state(self, variant, var):
        """The text the reference's printState writes for this state (file descriptor 1 captured)."""
        import tempfile
        import sys
        sys.stdout.flush()
        with tempfile.TemporaryFile() as tmp:
            saved = os.dup(1)
            try:
                os.dup2(tmp.fileno(), 1)
                self._f(variant, "print_state")(_p(np.ascontiguousarray(var, dtype=np.float64)))
            finally:
                os.dup2(saved, 1)
                os.close(saved)
            tmp.seek(0)
            return tmp.read().decode()


class Oracle:
    def __init__(self, eigen=False):
        self.lib = ctypes.CDLL(build_oracle())
        L = self.lib
        L.orc_gap.restype = ctypes.c_double
        L.orc_gap.argtypes = [ctypes.c_int, _dp]
        L.orc_residual_norm.restype = ctypes.c_double
        L.orc_residual_norm.argtypes = [ctypes.c_int, _dp, ctypes.c_double]
        L.orc_constraint.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _dp, _dp]
        L.orc_constraint_hess.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _dp]
        L.orc_constraints_satisfied.argtypes = [ctypes.c_int, _dp]
        L.orc_kkt.argtypes = [ctypes.c_int, _dp, _dp, _dp, _dp]
        L.orc_step_ex.argtypes = [ctypes.c_int, _dp, _dp, ctypes.POINTER(StepInfo), ctypes.c_void_p]
        L.orc_move_toward_feasibility.argtypes = [ctypes.c_int, _dp]
        L.orc_init_default.argtypes = [ctypes.c_int, _dp]
        L.orc_init_stuck_f3.argtypes = [_dp]
        L.orc_init_feasible.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp]
        L.orc_colpiv_qr_solve.argtypes = [ctypes.c_int, _dp, _dp, _dp]
        L.orc_colpiv_qr_solve_dynamic.argtypes = [ctypes.c_int, _dp, _dp, _dp]
        L.orc_batch_init_feasible.argtypes = [ctypes.c_int, ctypes.c_size_t, _dp, _dp, _dp, _dp]
        L.orc_batch_steps.argtypes = [ctypes.c_int, ctypes.c_size_t, _dp, ctypes.c_int, ctypes.c_int]
        L.orc_batch_steps_params.argtypes = [ctypes.c_int, ctypes.c_size_t, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int]
        L.orc_step_params.argtypes = [ctypes.c_int, _dp, ctypes.POINTER(StepInfo), ctypes.c_double, ctypes.c_int]
        L.orc_batch_solve_gated.restype = ctypes.c_int64
        L.orc_batch_solve_gated.argtypes = [ctypes.c_int, ctypes.c_size_t, _dp, ctypes.c_double, ctypes.c_int,
                                            ctypes.POINTER(ctypes.c_int32), ctypes.c_int]
        L.orc_batch_solve_gated_ex.restype = ctypes.c_int64
        L.orc_batch_solve_gated_ex.argtypes = [ctypes.c_int, ctypes.c_size_t, _dp, ctypes.c_double, ctypes.c_int,
                                               ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.c_void_p]
        L.orc_armijo_sides.argtypes = [ctypes.c_int, _dp, ctypes.c_int, ctypes.c_void_p, _dp]
        L.orc_feasibility_margin.argtypes = [ctypes.c_int, _dp, ctypes.c_int, ctypes.c_void_p, _dp]
        L.orc_sample_trajectory.argtypes = [ctypes.c_int, _dp, _dp, _dp]
        L.orc_gen_problems.argtypes = [ctypes.c_uint64, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, _dp, _dp, _dp]
        self.solver = None
        self.ref = None
        if eigen:
            if not have_ref():
                raise RuntimeError("oracle/_ref/libeigen_qr_ref.so is not built (make -C oracle ref, needs /root/reference)")
            self.ref = ctypes.CDLL(REF_SO)
            self.ref.ref_qr_solve.argtypes = [ctypes.c_int, _dp, _dp, _dp, ctypes.c_int]
            self.ref.ref_squared_norm.restype = ctypes.c_double
            self.ref.ref_squared_norm.argtypes = [ctypes.c_int, _dp]
            self.ref.ref_eigen_version.restype = ctypes.c_char_p
            self.solver = ctypes.cast(self.ref.ref_qr_solve, ctypes.c_void_p)

    # ---- scalar pieces ----
    @staticmethod
    def state_len(variant):
        return 12 if variant == 4 else 16

    @staticmethod
    def num_constraints(variant):
        return 4 if variant == 4 else 8

    def init_default(self, variant=3):
        v = np.zeros(self.state_len(variant))
        self.lib.orc_init_default(variant, _p(v))
        return v

    def init_stuck(self):
        v = np.zeros(16)
        self.lib.orc_init_stuck_f3(_p(v))
        return v

    def init_feasible(self, variant, p0, p1, p2):
        v = np.zeros(self.state_len(variant))
        self.lib.orc_init_feasible(variant, p0, p1, p2, _p(v))
        return v

    def gap(self, variant, var):
        return self.lib.orc_gap(variant, _p(np.ascontiguousarray(var, dtype=np.float64)))

    def residual_norm(self, variant, var, p):
        return self.lib.orc_residual_norm(variant, _p(np.ascontiguousarray(var, dtype=np.float64)), p)

    def constraint(self, variant, i, var):
        e = np.zeros(1)
        g = np.zeros(3)
        self.lib.orc_constraint(variant, i, _p(np.ascontiguousarray(var, dtype=np.float64)), _p(e), _p(g))
        return e[0], g

    def constraint_hess(self, variant, i, var):
        h = np.zeros(9)
        self.lib.orc_constraint_hess(variant, i, _p(np.ascontiguousarray(var, dtype=np.float64)), _p(h))
        return h

    def satisfied(self, variant, var):
        return bool(self.lib.orc_constraints_satisfied(variant, _p(np.ascontiguousarray(var, dtype=np.float64))))

    def kkt(self, variant, var):
        c = 3 + self.num_constraints(variant)
        m = np.zeros(c * c)
        r = np.zeros(c)
        p = np.zeros(1)
        self.lib.orc_kkt(variant, _p(np.ascontiguousarray(var, dtype=np.float64)), _p(m), _p(r), _p(p))
        return m.reshape(c, c).T.copy(), r, p[0]   # row-major view of the column-major matrix

    def step(self, variant, var, info=None):
        """One moveInteriorPoint in place on `var` (float64 array of state_len)."""
        d = np.zeros(3 + self.num_constraints(variant))
        self.lib.orc_step_ex(variant, _p(var), _p(d), ctypes.byref(info) if info is not None else None, self.solver)
        return d

    def armijo_sides(self, variant, var, halvings):
        """The residual test of the reference's second backtracking loop (onedpath_ip.cpp:941, onedpath2_ip.cpp:828) at the trial
        made after `halvings` halvings, from the state `var` a step starts from: (|r(trial)|^2, |r(x)|^2 (1 - 0.01 s), s, largest
        change of the first under a one-ulp move of one trial coordinate, largest change of the second under a one-ulp move of one
        coordinate of x, change of the first with a second solver's direction), or None beyond the loop's budget."""
        out = np.zeros(6)
        ok = self.lib.orc_armijo_sides(variant, _p(np.ascontiguousarray(var, dtype=np.float64)), int(halvings), self.solver, _p(out))
        return tuple(out) if ok else None

    def feasibility_margin(self, variant, var, halvings):
        """The feasibility test of the first backtracking loop (onedpath_ip.cpp:919-928) at the trial made after `halvings`
        halvings: (largest constraint value there, s, largest change of a constraint value under a one-ulp move of one variable,
        ... and with a second solver's direction)."""
        out = np.zeros(4)
        ok = self.lib.orc_feasibility_margin(variant, _p(np.ascontiguousarray(var, dtype=np.float64)), int(halvings), self.solver, _p(out))
        return tuple(out) if ok else None

    def solve_gated(self, variant, var, tol=1e-8, max_iter=200):
        it = 0
        while it < max_iter and not (self.gap(variant, var) < tol):
            self.step(variant, var)
            it += 1
        return it

    def move_toward_feasibility(self, variant, var):
        self.lib.orc_move_toward_feasibility(variant, _p(var))

    def sample(self, variant, var):
        pos = np.zeros(66)
        acc = np.zeros(4)
        self.lib.orc_sample_trajectory(variant, _p(np.ascontiguousarray(var, dtype=np.float64)), _p(pos), _p(acc))
        return pos, acc

    def qr_solve(self, A, b, dynamic=False):
        """The oracle's own column-pivoted Householder QR: Eigen's fixed-size order of operations (the Newton step's
        Matrix<double,11,11>) or, dynamic=True, its run-time-sized one (moveTowardFeasibility's MatrixXd)."""
        n = len(b)
        x = np.zeros(n)
        Ac = np.asfortranarray(A, dtype=np.float64)
        f = self.lib.orc_colpiv_qr_solve_dynamic if dynamic else self.lib.orc_colpiv_qr_solve
        nz = f(n, Ac.ctypes.data_as(_dp), _p(np.ascontiguousarray(b, dtype=np.float64)), _p(x))
        return x, nz

    def ref_qr_solve(self, A, b, force_dynamic=False):
        n = len(b)
        x = np.zeros(n)
        Ac = np.asfortranarray(A, dtype=np.float64)
        nz = self.ref.ref_qr_solve(n, Ac.ctypes.data_as(_dp), _p(np.ascontiguousarray(b, dtype=np.float64)), _p(x), int(force_dynamic))
        return x, nz

    # ---- batches (own QR only: the threaded C loops) ----
    def gen_problems(self, seed, first, n, dist):
        p0, p1, p2 = np.zeros(n), np.zeros(n), np.zeros(n)
        self.lib.orc_gen_problems(seed, first, n, dist, _p(p0), _p(p1), _p(p2))
        return p0, p1, p2

    def batch_init_feasible(self, variant, p0, p1, p2):
        n = len(p0)
        aos = np.zeros((n, self.state_len(variant)))
        self.lib.orc_batch_init_feasible(variant, n, _p(np.ascontiguousarray(p0)), _p(np.ascontiguousarray(p1)),
                                         _p(np.ascontiguousarray(p2)), _p(aos))
        return aos

    def batch_steps(self, variant, aos, k, threads=0):
        assert aos.flags.c_contiguous and aos.dtype == np.float64
        if self.solver is None:
            self.lib.orc_batch_steps(variant, aos.shape[0], _p(aos), k, threads)
        else:
            for row in aos:
                for _ in range(k):
                    self.step(variant, row)
        return aos

    def batch_steps_params(self, variant, aos, k, backtrack, max_bt, threads=0):
        """k steps with the line search's backtrack factor and halving budget as parameters ((0.5, 100) = the reference)."""
        assert aos.flags.c_contiguous and aos.dtype == np.float64
        self.lib.orc_batch_steps_params(variant, aos.shape[0], _p(aos), k, threads, backtrack, max_bt)
        return aos

    def step_params(self, variant, var, backtrack, max_bt, info=None):
        self.lib.orc_step_params(variant, _p(var), ctypes.byref(info) if info is not None else None, backtrack, max_bt)

    def batch_solve_gated(self, variant, aos, tol=1e-8, max_iter=200, threads=0):
        assert aos.flags.c_contiguous and aos.dtype == np.float64
        n = aos.shape[0]
        iters = np.zeros(n, dtype=np.int32)
        total = self.lib.orc_batch_solve_gated_ex(variant, n, _p(aos), tol, max_iter,
                                                  iters.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), threads, self.solver)
        return iters, total

    def hw_threads(self):
        return self.lib.orc_hw_threads()
