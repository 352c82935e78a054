"""Data-parallel sharding of a batch over the GPUs of a node.

Problems never interact (moveInteriorPoint touches only its own Trajectory,
onedpath_ip.cpp:810-953), so rank r simply owns the contiguous range shard_range(n, r, W) in
its own HBM and no step exchanges anything.  The one collective of the path is the final
summary: max ||r||^2 and max gap (MAX), converged count and Newton steps (SUM) -- 32 bytes,
all-reduced with RCCL over xGMI (torch.distributed backend "nccl") or gloo on CPU in tests.
"""
import torch
import torch.distributed as dist

from .problems import shard_range  # noqa: F401  (re-exported)

SUMMARY_FIELDS = ("max_residual_sq", "max_gap", "n_converged", "total_steps")


def allreduce_summary(local4, group=None, force=False):
    """In-place reduce of the 4-double summary tensor [max_r2, max_gap, n_conv, steps].
    force: run the two collectives even in a group of one rank (the backend's code path end to end: bench.py
    --force-process-group, RCCL on the one GPU there is); the result is then the input."""
    if local4.numel() != 4 or local4.dtype != torch.float64:
        raise ValueError("summary must be 4 float64 values")
    if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
        dist.all_reduce(local4[:2], op=dist.ReduceOp.MAX, group=group)
        dist.all_reduce(local4[2:], op=dist.ReduceOp.SUM, group=group)
    return local4


def summary_dict(t4):
    return {k: float(v) for k, v in zip(SUMMARY_FIELDS, t4.tolist())}


def batch_summary(batch, device=None, group=None):
    """Device-side reduction of `batch` into a torch tensor, then the cross-rank all-reduce.

    The kernel writes straight into the tensor RCCL reduces: no host round trip."""
    dev = device if device is not None else torch.device("cuda", batch.device)
    out = torch.empty(4, dtype=torch.float64, device=dev)
    batch.reduce_device(out.data_ptr())
    batch.sync()          # the batch runs on its own stream; the collective runs on torch's
    return allreduce_summary(out, group)


def solve_with_global_checks(launch, local_summary, n_total, max_iter, steps_per_check, group=None):
    """Gated solve of a sharded batch with a GLOBAL convergence check ("RCCL all-reduce(max) of
    per-GPU max ||r||^2 / max gap each check").

    launch()        -- enqueue up to `steps_per_check` gated steps on this rank's shard (Batch.solve_launch)
    local_summary() -- this rank's 4-double summary tensor (batch_summary's input: reduce_device into a tensor)
    Every check all-reduces the 32-byte summary; the loop ends when every problem of every shard is converged or when
    ceil(max_iter / steps_per_check) rounds have run (problems at their cap stop on their own).  Returns
    (global summary tensor, checks made).  Nothing else crosses between ranks."""
    if steps_per_check < 1:
        raise ValueError("steps_per_check must be positive")
    rounds = max(1, -(-max_iter // steps_per_check))
    g = None
    for check in range(1, rounds + 1):
        launch()
        g = allreduce_summary(local_summary(), group)
        if float(g[2]) >= n_total:
            return g, check
    return g, rounds


def batch_solve_with_global_checks(batch, n_total, gap_tol=1e-8, max_iter=200, steps_per_check=4, device=None, group=None):
    """solve_with_global_checks for a rocket_path_amd.Batch shard (the device writes the summary straight into the tensor
    that is all-reduced)."""
    dev = device if device is not None else torch.device("cuda", batch.device)

    def local():
        out = torch.empty(4, dtype=torch.float64, device=dev)
        batch.reduce_device(out.data_ptr())
        batch.sync()
        return out
    return solve_with_global_checks(lambda: batch.solve_launch(gap_tol, max_iter, steps_per_check), local, n_total, max_iter,
                                    steps_per_check, group)
