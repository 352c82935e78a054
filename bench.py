#!/usr/bin/env python3
"""bench.py -- Newton steps/s of the batched interior-point hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): per GPU a
batch of 1,048,576 F3 problems (monotone synthetic positions, seed 12345, feasible-start rule
of SURVEY.md 8d), fp64, each solved until its surrogate duality gap drops below 1e-8 (cap 200
steps).  One bench "step" = one pass of the hot path over one such batch: the fused gated
kernel solves every problem of the batch from its start state.  K + W batches are laid out in
HBM beforehand (positions -> start states, on the device) so that every timed pass streams a
state it has not touched before, from HBM rather than from the 256 MiB Infinity Cache, and the
timed region contains nothing but the hot path plus the single final reduction / all-reduce.

N > 1: one process per GPU, rank r owns the contiguous shard r of the N x 1,048,576 problems
(weak scaling); no step exchanges anything; the only collective is the final 32-byte summary
all-reduce over RCCL.

Printed by rank 0: ONE JSON line (contract in the task statement), with `roofline` for the
dominant kernel (k_solve_tiled<double,3>), `cpu_baseline` (oracle port, N = 1 only) and three
labelled extras: `per_step_launch` (the same step as one launch per Newton step, which is the
HBM-streaming form: 216 B really cross HBM per step), `fixed50` (configs[1]) and `f4_fp32`
(configs[4]).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PER_GPU = 1 << 20
SEED = 12345
GAP_TOL = 1e-8
MAX_ITER = 200
B_ALG_F3 = 216.0          # bytes per problem per Newton step: read 16, write 11 doubles
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_quota():
    """CPUs this process may really use: affinity mask capped by the cgroup v2 cpu.max quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(sample_n):
    """The oracle (C restatement of the reference's CPU step, own 11x11 QR) on the host cores.
    Reported baseline only -- never the thing measured above."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_api import Oracle
    from rocket_path_amd import problems
    orc = Oracle()
    threads = min(orc.hw_threads(), cpu_quota())
    p0, p1, p2 = problems.generate(SEED, 0, sample_n, problems.DIST_MONOTONE)
    aos = orc.batch_init_feasible(3, p0, p1, p2)
    t0 = time.perf_counter()
    _, total = orc.batch_solve_gated(3, aos, GAP_TOL, MAX_ITER, threads=threads)
    dt = time.perf_counter() - t0
    # single-core rate on a smaller slice, for the per-core figure
    n1 = max(1, sample_n // 32)
    aos1 = orc.batch_init_feasible(3, p0[:n1], p1[:n1], p2[:n1])
    t0 = time.perf_counter()
    _, total1 = orc.batch_solve_gated(3, aos1, GAP_TOL, MAX_ITER, threads=1)
    dt1 = time.perf_counter() - t0
    out = {"value": total / dt, "unit": "Newton steps/s", "cores": threads, "kind": "port",
           "sample": "first %d problems of the same batch, same gate, %d steps in %.2f s on %d threads" % (sample_n, total, dt, threads),
           "single_core_value": total1 / dt1}
    # the same restatement with its 11x11 solve done by the reference's own vendored Eigen 3.3.0 QR (oracle/_ref,
    # prebuilt where /root/reference exists): 78 % of a reference step is that call
    try:
        from oracle_api import have_ref
        if have_ref():
            orc_e = Oracle(eigen=True)
            aos2 = orc_e.batch_init_feasible(3, p0, p1, p2)
            t0 = time.perf_counter()
            _, total2 = orc_e.batch_solve_gated(3, aos2, GAP_TOL, MAX_ITER, threads=threads)
            dt2 = time.perf_counter() - t0
            aos3 = orc_e.batch_init_feasible(3, p0[:n1], p1[:n1], p2[:n1])
            t0 = time.perf_counter()
            _, total3 = orc_e.batch_solve_gated(3, aos3, GAP_TOL, MAX_ITER, threads=1)
            dt3 = time.perf_counter() - t0
            out["with_reference_eigen_qr"] = {"value": total2 / dt2, "single_core_value": total3 / dt3, "cores": threads}
    except Exception as exc:      # the baseline must never take the benchmark down
        out["with_reference_eigen_qr"] = {"error": str(exc)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--problems-per-gpu", type=int, default=N_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="debug: run the N-rank code path with every rank on device 0 and gloo for the collectives "
                         "(RCCL refuses two ranks on one GPU); numbers from such a run are not benchmark results")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import rocket_path_amd as rp
    from rocket_path_amd import problems, sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run "
                             "--nproc-per-node %d bench.py --gpus %d" % (args.gpus, args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available() or rp.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback to time)")
    rehearsal = args.rehearse_on_one_gpu
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    coll_dev = torch.device("cpu") if rehearsal else torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    n_total = args.problems_per_gpu * world
    first, count = problems.shard_range(n_total, rank, world)
    K, W = args.steps, args.warmup

    # ---- inputs resident in HBM before anything is timed ----
    p0, p1, p2 = problems.generate(SEED, first, count, problems.DIST_MONOTONE)
    d_pos = torch.from_numpy(np.stack([p0, p1, p2])).to(torch.device("cuda", local_rank))
    lead = rp.Batch(count, rp.VARIANT_F3, rp.DTYPE_F64, device=local_rank)
    stream = lead.stream()
    # one batch per pass up to POOL of them (136 MB each at 1 Mi problems); longer runs cycle through the pool and
    # pay the re-initialisation of a recycled batch inside the timed region (conservative: ~6 % of a pass)
    POOL = 48
    n_batches = max(1, min(K + W, POOL))
    batches = [lead] + [rp.Batch(count, rp.VARIANT_F3, rp.DTYPE_F64, device=local_rank, stream=stream) for _ in range(n_batches - 1)]
    ptrs = [d_pos[j].data_ptr() for j in range(3)]
    for b in batches:
        b.set_problems_device(*ptrs)
    lead.sync()

    def pass_(index):
        b = batches[index % n_batches]
        if index >= n_batches:
            b.set_problems_device(*ptrs)
        b.solve(GAP_TOL, MAX_ITER, 0)
        return b
    summary = torch.zeros(4, dtype=torch.float64, device=torch.device("cuda", local_rank))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup: W untimed passes ----
    for i in range(W):
        pass_(i)
    if world > 1:
        sharding.allreduce_summary(summary.clone().to(coll_dev))     # RCCL communicator setup outside the timed region
    barrier()

    # ---- timed: exactly K passes, then the final summary reduction (+ all-reduce) ----
    t0 = time.perf_counter()
    lead.event_record(0)
    last = lead
    for i in range(W, W + K):
        last = pass_(i)
    lead.event_record(1)
    last.reduce_device(summary.data_ptr())
    lead.sync()
    summary = sharding.allreduce_summary(summary.to(coll_dev))
    barrier()
    elapsed = time.perf_counter() - t0

    kernel_ms = lead.event_elapsed_ms(0, 1) / max(K, 1)
    # every pass solves the same seeded batch from the same start: steps per pass are those of any solved batch
    one = last.reduce()
    steps_local = one["total_steps"] * K
    conv_local = one["n_converged"] * K
    t = torch.tensor([elapsed, steps_local, conv_local], dtype=torch.float64, device=coll_dev)
    if world > 1:
        tm = t[:1].clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        ts = t[1:].clone()
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        elapsed, steps_all, conv_all = float(tm[0]), float(ts[0]), float(ts[1])
    else:
        steps_all, conv_all = steps_local, conv_local
    g = sharding.summary_dict(summary)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    steps_per_launch = steps_local / max(K, 1)
    achieved = B_ALG_F3 * steps_per_launch / (kernel_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("k_solve_tiled_f3_f64", {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    line = {
        "metric": "interior-point Newton steps/sec (whole node) + achieved HBM GB/s, 1M-problem batch",
        "value": steps_all / elapsed,
        "unit": "Newton steps/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / max(K, 1) * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one GPU, gloo collectives -- not a result)",
        "config": {
            "workload": "BASELINE configs[2] (C3): %d F3 onedpath_ip problems per GPU, convergence-gated "
                        "(surrogate gap < 1e-8 checked before every step, cap 200), fp64, monotone seeded positions, "
                        "feasible-start rule; one fused launch per batch" % args.problems_per_gpu,
            "problems_per_gpu": args.problems_per_gpu,
            "problems_total": n_total,
            "newton_steps_per_pass_per_gpu": steps_per_launch,
            "mean_steps_per_problem": steps_per_launch / count,
            "converged_fraction": conv_all / (n_total * max(K, 1)),
            "final_summary": g,
            "sharding": "contiguous shards, no data-path collective; one 32-byte RCCL all-reduce at the end" if world > 1
                        else "single GPU",
        },
        "roofline": {
            "bound": "hbm", "kernel": "k_solve_tiled<double, F3> (fused gated solve: state stays in VGPRs between steps)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": B_ALG_F3 * steps_per_launch,
            "avg_launch_ms": kernel_ms,
            "note": "algorithmic bytes = 216 B x Newton steps executed in the launch (SURVEY.md 8d); the fused launch "
                    "moves each state across HBM once per solve, so measured traffic is ~1/15 of this and the kernel "
                    "is fp64-ALU bound; see per_step_launch for the form in which 216 B/step really cross HBM",
        },
    }

    # The fused solve is fp64-ALU bound, so next to the (algorithmic) HBM roofline the same launch is priced against
    # the fp64 vector peak: flop per Newton step from the rocprofv3 SQ counters of profiles/r1_sq_counters.json
    # (2 x FMA + MUL + ADD + RCP wave-instructions of the ungated 12-step launch, per lane-step).
    FLOP_PER_STEP, FP64_PEAK_TFLOPS = 643.0, 78.6
    try:
        FLOP_PER_STEP = float(json.load(open(os.path.join(ROOT, "profiles", "r1_sq_counters.json")))["_flop_per_newton_step"])
    except Exception:
        pass
    tflops = FLOP_PER_STEP * steps_per_launch / (kernel_ms * 1e-3) / 1e12
    line["compute_roofline"] = {"bound": "fp64 vector ALU", "flop_per_newton_step": FLOP_PER_STEP, "achieved": tflops,
                                "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_PEAK_TFLOPS,
                                "note": "about 490 VALU instructions per step, 400 of them fp64; measured issue cost 2.1 ns per fp64 "
                                        "wave-instruction per SIMD (profiles/probes/valu_probe.hip); idle lane-steps of the gated "
                                        "solve (~10 %) are not counted as flops"}

    if not args.no_extras:
        # (a) one launch per Newton step: the HBM-streaming form of the same step (216 B really move per step).
        #     cold = every launch on a batch not touched since its init (state comes from HBM);
        #     warm = the same 128 MiB batch stepped again and again (state stays in the 256 MiB Infinity Cache);
        #     probe = the same kernel with zero steps: its 16 loads + 11 stores per problem and nothing else,
        #             i.e. what this access pattern can reach on this box (the kernel's own ceiling).
        spare = batches[:min(len(batches), 20)]

        def reinit():
            for b in spare:
                b.set_problems_device(*ptrs)
            lead.sync()
            torch.cuda.synchronize()

        def sweep(fn):
            lead.event_record(2)
            for b in spare:
                fn(b)
            lead.event_record(3)
            lead.sync()
            return lead.event_elapsed_ms(2, 3) / len(spare)

        reinit()
        ms_cold = sweep(lambda b: b.step(1))
        os.environ["RP_STREAM_PROBE"] = "1"
        reinit()
        ms_probe = sweep(lambda b: b.step(0))
        del os.environ["RP_STREAM_PROBE"]
        lead.set_problems_device(*ptrs)
        lead.step(1)
        lead.event_record(2)
        for _ in range(len(spare)):
            lead.step(1)
        lead.event_record(3)
        lead.sync()
        ms_warm = lead.event_elapsed_ms(2, 3) / len(spare)
        gbs = lambda ms: B_ALG_F3 * count / (ms * 1e-3) / 1e9   # noqa: E731
        line["per_step_launch"] = {
            "kernel": "k_newton_stream<double, F3>, k = 1 (register-prefetched, grid = resident set)",
            "avg_launch_ms": ms_cold, "newton_steps_per_s": count / (ms_cold * 1e-3),
            "achieved_GBps": gbs(ms_cold), "frac_of_hbm_peak": gbs(ms_cold) / HBM_PEAK_GBS,
            "same_access_pattern_without_arithmetic_GBps": gbs(ms_probe),
            "frac_of_that_ceiling": ms_probe / ms_cold,
            "infinity_cache_resident_GBps": gbs(ms_warm),
            "launches": len(spare)}
        # (b) configs[1]: 65,536 problems, exactly 50 steps each, one fused launch
        n2 = min(65536, count)
        with rp.Batch(n2, rp.VARIANT_F3, rp.DTYPE_F64, device=local_rank, stream=stream) as c2:
            ms = []
            for _ in range(4):
                c2.set_problems_device(*ptrs)
                c2.sync()
                c2.event_record(4)
                c2.step(50)
                c2.event_record(5)
                c2.sync()
                ms.append(c2.event_elapsed_ms(4, 5))
        line["fixed50"] = {"workload": "BASELINE configs[1] (C2): 65,536 problems x exactly 50 steps, one fused launch "
                                       "(about 35 of the 50 steps per problem run in the reference's post-convergence regime: "
                                       "~48 residual halvings per step)",
                           "ms": min(ms[1:]), "newton_steps_per_s": n2 * 50 / (min(ms[1:]) * 1e-3)}
        # (c) configs[4]: F4, fp32, 1,048,576 problems x 50 steps (76 B algorithmic per step)
        with rp.Batch(count, rp.VARIANT_F4, rp.DTYPE_F32, device=local_rank, stream=stream) as c5:
            ms50, ms1 = [], []
            for _ in range(3):
                c5.set_problems_device(*ptrs)
                c5.sync()
                c5.event_record(4)
                c5.step(50)
                c5.event_record(5)
                c5.sync()
                ms50.append(c5.event_elapsed_ms(4, 5))
            for _ in range(3):
                c5.set_problems_device(*ptrs)
                c5.sync()
                c5.event_record(4)
                c5.step(1)
                c5.event_record(5)
                c5.sync()
                ms1.append(c5.event_elapsed_ms(4, 5))
        line["f4_fp32"] = {"workload": "BASELINE configs[4] (C5): F4 onedpath2_ip, fp32, %d problems" % count,
                           "fused_50_steps_ms": min(ms50), "newton_steps_per_s": count * 50 / (min(ms50) * 1e-3),
                           "k1_launch_ms": min(ms1), "k1_achieved_GBps": 76.0 * count / (min(ms1) * 1e-3) / 1e9}

    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(min(count, 1 << 19))
    else:
        line["cpu_baseline"] = None

    print(json.dumps(line))
    sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
