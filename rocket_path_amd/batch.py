"""Host-side mirror of the batched Problem over the C ABI (include/rp_batch.h).

`Batch` is what the C++ plug-in `BatchedOneDPathIP` (csrc/host/) is to rocket_path.cpp, for
Python callers (tests, bench): the same entry points with the same meaning,
    init_default()  = Problem::init() / onKey('i')      (onedpath_ip.cpp:230-233, 259-262)
    init_stuck()    = onKey('j')                         (onedpath_ip.cpp:264-267)
    step(k)         = k x onKey('n')                     (onedpath_ip.cpp:269-272)
    move_toward_feasibility() = onKey(' ')               (onedpath_ip.cpp:254-257)
    nudge(i, d)     = onSpecialKey                       (onedpath_ip.cpp:280-324)
and, like the reference, no error returns in the happy path: failures raise RpError.
State crosses the boundary in the reference's AoS layout (double var[16] / var[12]).
"""
import ctypes

import numpy as np

from . import capi


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


class Batch:
    def __init__(self, n, variant=capi.VARIANT_F3, dtype=capi.DTYPE_F64, device=0, stream=None, _borrowed=None):
        self._lib = capi.load_library()
        self._h = ctypes.c_void_p()
        self._owned = _borrowed is None
        if _borrowed is not None:          # a batch a pipeline owns (rp_pipeline_batch): every call works, close() leaves it alone
            self._h = ctypes.c_void_p(_borrowed)
        else:
            capi.check(self._lib.rp_batch_create(ctypes.byref(self._h), variant, dtype, n, device,
                                                 ctypes.c_void_p(stream) if stream else None))
        self.n, self.variant, self.dtype, self.device = n, variant, dtype, device
        self.state_len = 12 if variant == capi.VARIANT_F4 else 16
        self.num_constraints = 4 if variant == capi.VARIANT_F4 else 8

    # ---- lifetime ----
    def close(self):
        if self._h:
            if self._owned:
                self._lib.rp_batch_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- parameters ----
    def get_params(self):
        p = capi.Params()
        capi.check(self._lib.rp_batch_get_params(self._h, ctypes.byref(p)))
        return p

    def set_params(self, **kw):
        p = self.get_params()
        for k, v in kw.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        capi.check(self._lib.rp_batch_set_params(self._h, ctypes.byref(p)))

    # ---- init ----
    def init_default(self):
        capi.check(self._lib.rp_batch_init_default(self._h))

    def init_stuck(self):
        capi.check(self._lib.rp_batch_init_stuck(self._h))

    def set_problems(self, pos0, pos1, pos2):
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (pos0, pos1, pos2)]
        for a in arrs:
            if a.shape != (self.n,):
                raise ValueError("position arrays must have shape (%d,)" % self.n)
        capi.check(self._lib.rp_batch_set_problems(self._h, *[_ptr(a) for a in arrs]))

    def set_problems_device(self, d_pos0, d_pos1, d_pos2):
        """Device pointers (ints) to n float64 each, e.g. torch tensors' data_ptr()."""
        capi.check(self._lib.rp_batch_set_problems_device(self._h, *[ctypes.c_void_p(p) for p in (d_pos0, d_pos1, d_pos2)]))

    def restart(self):
        """Feasible start again from the positions already in the batch (no input crosses the boundary)."""
        capi.check(self._lib.rp_batch_restart(self._h))

    def set_state(self, aos):
        a = np.ascontiguousarray(aos, dtype=np.float64)
        if a.shape != (self.n, self.state_len):
            raise ValueError("state must have shape (%d, %d)" % (self.n, self.state_len))
        capi.check(self._lib.rp_batch_set_state(self._h, _ptr(a)))

    def get_state(self):
        a = np.empty((self.n, self.state_len), dtype=np.float64)
        capi.check(self._lib.rp_batch_get_state(self._h, _ptr(a)))
        return a

    def get_state_range(self, first, count):
        a = np.empty((count, self.state_len), dtype=np.float64)
        capi.check(self._lib.rp_batch_get_state_range(self._h, first, count, _ptr(a)))
        return a

    def nudge(self, var_index, delta):
        capi.check(self._lib.rp_batch_nudge(self._h, var_index, float(delta)))

    # ---- hot path ----
    def step(self, k=1):
        capi.check(self._lib.rp_batch_step(self._h, int(k)))

    def step_counted(self, k=1):
        """k steps plus the per-problem totals of (feasibility, residual) halvings over them (diagnostic)."""
        nf = np.empty(self.n, dtype=np.uint32)
        nr = np.empty(self.n, dtype=np.uint32)
        capi.check(self._lib.rp_batch_step_counted(self._h, int(k), _ptr(nf), _ptr(nr)))
        return nf, nr

    def solve(self, gap_tol=1e-8, max_iter=200, steps_per_launch=0):
        capi.check(self._lib.rp_batch_solve(self._h, float(gap_tol), int(max_iter), int(steps_per_launch)))

    def solve_launch(self, gap_tol=1e-8, max_iter=200, k=1):
        """One asynchronous launch of up to k gated steps per open problem (no host polling)."""
        capi.check(self._lib.rp_batch_solve_launch(self._h, float(gap_tol), int(max_iter), int(k)))

    def move_toward_feasibility(self):
        capi.check(self._lib.rp_batch_move_toward_feasibility(self._h))

    # ---- results ----
    def get_iters(self):
        it = np.empty(self.n, dtype=np.int32)
        st = np.empty(self.n, dtype=np.uint32)
        capi.check(self._lib.rp_batch_get_iters(self._h, _ptr(it), _ptr(st)))
        return it, st

    def solution_device(self, d_out):
        """Every problem's 32-byte rp_solution record, in problem order, into device memory (address of n records)."""
        capi.check(self._lib.rp_batch_solution_device(self._h, ctypes.c_void_p(d_out)))

    def bind_solution(self, d_out):
        """Gated solves write each problem's rp_solution record to d_out (device address of n records; None / 0 unbinds)."""
        capi.check(self._lib.rp_batch_bind_solution(self._h, ctypes.c_void_p(d_out) if d_out else None))

    def traffic_probe(self):
        """The k = 1 kernel's loads and stores with no step (bandwidth calibration)."""
        capi.check(self._lib.rp_batch_traffic_probe(self._h))

    def reduce(self):
        r = capi.Reduction()
        capi.check(self._lib.rp_batch_reduce(self._h, ctypes.byref(r)))
        return {"max_residual_sq": r.max_residual_sq, "max_gap": r.max_gap,
                "n_converged": r.n_converged, "total_steps": r.total_steps}

    def reduce_device(self, d_out4):
        capi.check(self._lib.rp_batch_reduce_device(self._h, ctypes.c_void_p(d_out4)))

    def sample(self):
        pos = np.empty((self.n, 66), dtype=np.float64)
        acc = np.empty((self.n, 4), dtype=np.float64)
        capi.check(self._lib.rp_batch_sample(self._h, _ptr(pos), _ptr(acc)))
        return pos, acc

    def sample_device(self, d_pos66, d_acc4):
        """The same into device memory (addresses of n x 66 and n x 4 doubles), asynchronously on the batch stream."""
        capi.check(self._lib.rp_batch_sample_device(self._h, ctypes.c_void_p(d_pos66), ctypes.c_void_p(d_acc4)))

    def sample_range(self, first, count):
        pos = np.empty((count, 66), dtype=np.float64)
        acc = np.empty((count, 4), dtype=np.float64)
        capi.check(self._lib.rp_batch_sample_range(self._h, first, count, _ptr(pos), _ptr(acc)))
        return pos, acc

    def constraints_range(self, first, count):
        """printState's `Surrogate gap` and `Constraints:` table: (gap[count], table[count, m, 14]) with columns
        error, deriv[3], second[9], dot (printConstraints, onedpath_ip.cpp:955-995)."""
        m = self.num_constraints
        rows = np.empty((count, 1 + 14 * m), dtype=np.float64)
        capi.check(self._lib.rp_batch_constraints_range(self._h, first, count, _ptr(rows)))
        return rows[:, 0].copy(), rows[:, 1:].reshape(count, m, 14).copy()

    # ---- plumbing ----
    def sync(self):
        capi.check(self._lib.rp_batch_sync(self._h))

    def stream(self):
        s = ctypes.c_void_p()
        capi.check(self._lib.rp_batch_stream(self._h, ctypes.byref(s)))
        return s.value

    def event_record(self, slot):
        capi.check(self._lib.rp_batch_event_record(self._h, slot))

    def event_elapsed_ms(self, start, stop):
        ms = ctypes.c_float()
        capi.check(self._lib.rp_batch_event_elapsed_ms(self._h, start, stop, ctypes.byref(ms)))
        return ms.value

    def field_ptr(self, field):
        """Device address of one field; its elements are in BATCH order (see slot_map)."""
        p = ctypes.c_void_p()
        capi.check(self._lib.rp_batch_field_ptr(self._h, field, ctypes.byref(p)))
        return p.value

    def slot_map(self):
        """slot_of_problem[i]: where problem i lies inside the batch's field arrays."""
        out = np.empty(self.n, dtype=np.uint32)
        capi.check(self._lib.rp_batch_slot_map(self._h, out.ctypes.data))
        return out


class Pipeline:
    """rp_pipeline: positions in -> solutions out, job after job, the batches dealt onto `n_streams` streams (include/rp_batch.h).
    submit() enqueues one job (scheduling pass + fused gated solve, records into d_out in problem order) and returns its number."""

    PREP_INLINE, PREP_STREAM, PREP_PRIORITY = 0, 1, 2      # RP_PIPELINE_PREP_*: where a job's scheduling pass runs

    def __init__(self, n, variant=capi.VARIANT_F3, dtype=capi.DTYPE_F64, device=0, depth=4, n_streams=2, prep=None):
        self._lib = capi.load_library()
        self._h = ctypes.c_void_p()
        capi.check(self._lib.rp_pipeline_create(ctypes.byref(self._h), variant, dtype, n, device, depth, n_streams))
        self.n, self.variant, self.dtype, self.device, self.depth, self.n_streams = n, variant, dtype, device, depth, n_streams
        if prep is not None:
            capi.check(self._lib.rp_pipeline_set_prep(self._h, int(prep)))

    def close(self):
        if self._h:
            self._lib.rp_pipeline_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_params(self, **kw):
        p = capi.Params()
        self._lib.rp_params_default(ctypes.byref(p))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        capi.check(self._lib.rp_pipeline_set_params(self._h, ctypes.byref(p)))

    def submit(self, d_pos0, d_pos1, d_pos2, d_out=None, gap_tol=1e-8, max_iter=200, inputs_stream=None):
        job = ctypes.c_int64(-1)
        capi.check(self._lib.rp_pipeline_submit(self._h, ctypes.c_void_p(d_pos0), ctypes.c_void_p(d_pos1), ctypes.c_void_p(d_pos2),
                                                ctypes.c_void_p(d_out) if d_out else None, float(gap_tol), int(max_iter),
                                                ctypes.c_void_p(inputs_stream) if inputs_stream else None, ctypes.byref(job)))
        return job.value

    def wait(self, job=-1):
        capi.check(self._lib.rp_pipeline_wait(self._h, int(job)))

    def stream_wait(self, job, what, stream):
        """Make `stream` wait (on the device) until the job's positions have been read (what = 0) / its solutions are written (what = 1)."""
        capi.check(self._lib.rp_pipeline_stream_wait(self._h, int(job), int(what), ctypes.c_void_p(stream)))

    def batch(self, job):
        """The batch that holds `job` (while it is the last job of its slot), as a Batch the pipeline keeps owning."""
        h = ctypes.c_void_p()
        capi.check(self._lib.rp_pipeline_batch(self._h, int(job), ctypes.byref(h)))
        return Batch(self.n, self.variant, self.dtype, self.device, _borrowed=h.value)
