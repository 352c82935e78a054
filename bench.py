#!/usr/bin/env python3
"""bench.py -- Newton steps/s of the batched interior-point hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no RANK in the environment launches its own N ranks (the parent, before it has
imported torch or touched HIP, starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process,
relays rank 0's JSON line and returns the child's exit code); under a launcher it is one of the ranks.

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

Printed by rank 0: ONE JSON line (contract in the task statement).  `roofline` prices the timed kernel
(k_solve_chunks<double, F3>, the fused gated solve) against the roof that binds it, the fp64 vector ALU, and
carries the SURVEY 8d algorithmic-HBM figure as the labelled secondary `hbm_algorithmic` (a fused launch moves each
state across HBM once per solve, not once per step).  `cpu_baseline` = the oracle port on the host cores (N = 1
only).  Labelled extras: `per_step_launch` (one launch per Newton step, the form in which the bytes really cross
HBM per step, priced on the bytes that move, next to two ceilings measured in the same run), `fixed50`
(configs[1]) and `f4_fp32` (configs[4], both arithmetic modes, each with its own roofline).  Numbers that are read
from committed rocprofv3 summaries instead of being measured in this run say so in a `source` key.
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
B_MOVED_F3_ZV = 200.0     # what the zero-end-velocity kernels really move per step: read 14, write 11 doubles
B_ALG_F4_F32 = 76.0       # F4 fp32: read 12, write 7 floats; the zero-end-velocity kernels read 10: 68 B
B_MOVED_F4_F32_ZV = 68.0
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_PEAK_TFLOPS = 78.6   # fp64 vector peak: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
FP32_PEAK_TFLOPS = 157.3  # fp32 vector peak, same guide
PROFILE_TAG = "r6"        # profiles/<tag>_*.json hold the rocprofv3 counter summaries the file-sourced numbers come from
# everything that decides what a launch executes and moves: the per-lane code, the kernels, the launch selection (rp_batch.cpp: field
# stride, which kernel a pass runs) and the scheduling pass (chunk order -> idle lane-steps, bytes per launch)
KERNEL_SOURCES = ("rocket_path_amd/csrc/ip_core.h", "rocket_path_amd/csrc/ip_kernels.hip", "rocket_path_amd/csrc/feas_core.h",
                  "rocket_path_amd/csrc/ip_kernels.h", "rocket_path_amd/csrc/rp_batch.cpp", "rocket_path_amd/csrc/schedule.hip")


def source_hashes():
    import hashlib
    return {f: hashlib.sha256(open(os.path.join(ROOT, f), "rb").read()).hexdigest()[:16] for f in KERNEL_SOURCES}


def profile_provenance():
    """Which commit and which kernel sources the file-sourced counter numbers were collected from (profiles/<tag>_sources.json,
    written by profiles/summarize.py at collection time), and whether the kernel sources are still those: a kernel change
    without a re-collection must not silently mis-price a roofline fraction."""
    path = os.path.join(ROOT, "profiles", PROFILE_TAG + "_sources.json")
    try:
        rec = json.load(open(path))
    except Exception:
        return {"file": None, "current": False, "note": "no profiles/%s_sources.json: counter-derived fractions withheld" % PROFILE_TAG}
    now = source_hashes()
    changed = [f for f in KERNEL_SOURCES if rec.get("sha256_16", {}).get(f) != now[f]]
    return {"file": "profiles/%s_sources.json" % PROFILE_TAG, "collected_at_commit": rec.get("commit"), "current": not changed,
            "changed_since_collection": changed}


def profile_number(fname, *keys):
    """A number from a committed rocprofv3 summary under profiles/ (None if absent), with the path it came from."""
    path = os.path.join(ROOT, "profiles", fname)
    try:
        v = json.load(open(path))
        for k in keys:
            v = v[k]
        return float(v), "profiles/" + fname
    except Exception:
        return None, None


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
    # bounded sample: whole passes over the first sample_n problems of the batch until about 10 s of wall time are spent
    total, dt, passes = 0, 0.0, 0
    while dt < 10.0 and passes < 64:
        aos = orc.batch_init_feasible(3, p0, p1, p2)
        t0 = time.perf_counter()
        _, tot = orc.batch_solve_gated(3, aos, GAP_TOL, MAX_ITER, threads=threads)
        dt += time.perf_counter() - t0
        total += tot
        passes += 1
    # single-core rate on a smaller slice, for the per-core figure
    n1 = max(1, sample_n // 32)
    aos1 = orc.batch_init_feasible(3, p0[:n1], p1[:n1], p2[:n1])
    t0 = time.perf_counter()
    _, total1 = orc.batch_solve_gated(3, aos1, GAP_TOL, MAX_ITER, threads=1)
    dt1 = time.perf_counter() - t0
    out = {"value": total / dt, "unit": "Newton steps/s", "cores": threads, "kind": "port",
           "sample": "%d pass(es) over the first %d problems of the same batch, same gate: %d steps in %.2f s on %d threads" % (
               passes, sample_n, total, dt, threads),
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
    # the reference's own functions (oracle/_ref/libref_hotpath.so: onedpath_ip.cpp's hot path compiled in the build
    # container from the reference tree), one thread, a small slice: every moveInteriorPoint call formats its 11 x 11
    # matrix into printf (to /dev/null here), which is most of its time -- reported for completeness, not as the baseline
    try:
        from oracle_api import Reference, have_ref_hotpath
        if have_ref_hotpath():
            ref = Reference()
            n2 = max(1, sample_n // 256)
            aos4 = orc.batch_init_feasible(3, p0[:n2], p1[:n2], p2[:n2])
            t0 = time.perf_counter()
            _, total4 = ref.batch_solve_gated(3, aos4, GAP_TOL, MAX_ITER)
            dt4 = time.perf_counter() - t0
            check = orc.batch_init_feasible(3, p0[:n2], p1[:n2], p2[:n2])
            _, total5 = orc.batch_solve_gated(3, check, GAP_TOL, MAX_ITER, threads=1)
            out["reference_library_with_its_prints"] = {
                "value": total4 / dt4, "cores": 1, "kind": "reference", "steps": int(total4),
                "same_steps_as_the_port": bool(total4 == total5),
                "note": "the reference's moveInteriorPoint prints its KKT matrix at every step (onedpath_ip.cpp:865-899); "
                        "SURVEY section 6 measured 0.45-0.47 M steps/s per core with the prints compiled out"}
    except Exception as exc:
        out["reference_library_with_its_prints"] = {"error": str(exc)}
    return out


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def visible_devices():
    """HIP devices a FRESH child process sees (this process must not touch HIP before it starts the ranks)."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import rocket_path_amd as rp; print(rp.device_count())" % ROOT],
                       capture_output=True, text=True)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def launch_ranks(n, rehearsal=False):
    """`python bench.py --gpus N` as typed: this process has not imported torch nor touched HIP; it starts the N ranks as a
    child (torch.distributed.run, one process per GPU), relays what they print and returns their exit code."""
    import subprocess
    if not rehearsal:
        have = visible_devices()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d needs %d HIP devices, this machine shows %d (checked in a fresh child process): no ranks started. "
                             "One process per GPU is the only form measured; --rehearse-on-one-gpu is the debug form.\n" % (n, n, have))
            return 3
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), RP_BENCH_SELF_LAUNCHED="1")
    sys.stderr.write("bench.py: --gpus %d without RANK in the environment: starting %d ranks: %s\n" % (n, n, " ".join(cmd)))
    sys.stderr.flush()
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--problems-per-gpu", type=int, default=N_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--force-process-group", action="store_true",
                    help="with --gpus 1: still create the torch.distributed process group on the nccl (= RCCL) backend and run the "
                         "summary all-reduce through it on the device tensor -- the N > 1 code path end to end on the one GPU there is")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="debug: run the N-rank code path with every rank on device 0 and gloo for the collectives "
                         "(RCCL refuses two ranks on one GPU); numbers from such a run are not benchmark results")
    ap.add_argument("--condition-launches", type=int, default=None,
                    help="untimed solves right before the W warmup passes (0: none; default: about 30 ms of them, i.e. 160 x 2^20 / problems-per-gpu, "
                         "at most 4096).  From an idle chip the power controller over-reacts for ~20 ms (launches at 0.19 -> 0.23 -> 0.17 ms, "
                         "profiles/r5_transient.log): a timed region of 20 launches that starts 1 ms after idle measures that transient, not the path.  "
                         "The solves run on a small ring of scratch batches (at most 16, at most a tenth of the free device memory), each put back on its start "
                         "before it is solved again; `value_from_idle` in the line is the figure without any of this")
    ap.add_argument("--sustain-seconds", type=float, default=None,
                    help="extra, after the timed region: keep solving batches back to back for at least this many seconds and report the "
                         "steady-state rate with clock / power samples (the device's hwmon files) -- the thermal-steady figure the 4 ms timed region cannot show.  "
                         "Default: 1.5 s on a single-GPU run with extras (it is also what lets an outside sampler see the GPU busy), 0 otherwise")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, rehearsal=args.rehearse_on_one_gpu))

    # The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues, 4 unless the environment says otherwise, and two
    # streams that land on one queue serialise.  This process owns more than that (the timed batches' stream, the pipeline's two, the
    # two-stream extras', torch's): with 4 queues the two streams of `end_to_end`'s rp_pipeline can end up sharing one (measured on one box with
    # four user streams alive: 73 G steps/s against 91 G with 8 queues, profiles/r6_hw_queues.log).  An application's setting, made here by the
    # application before anything initialises HIP -- the library itself neither reads nor writes the environment; the headline (one stream) is
    # not affected either way.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    import torch
    import torch.distributed as dist

    import rocket_path_amd as rp
    from rocket_path_amd import problems, sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available() or rp.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback to time)")
    rehearsal = args.rehearse_on_one_gpu
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    coll_dev = torch.device("cpu") if rehearsal else torch.device("cuda", local_rank)
    grouped = world > 1 or args.force_process_group      # a process group exists and every collective below goes through it
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
    # one batch per pass up to POOL of them (136 MB each at 1 Mi problems: 35 GB of the 288); longer runs cycle through the pool and
    # pay the restart of a recycled batch inside the timed region (conservative: ~10 % of a pass)
    POOL = 256
    n_batches = max(1, min(K + W, POOL))
    batches = [lead] + [rp.Batch(count, rp.VARIANT_F3, rp.DTYPE_F64, device=local_rank, stream=stream) for _ in range(n_batches - 1)]
    ptrs = [d_pos[j].data_ptr() for j in range(3)]
    for b in batches:
        b.set_problems_device(*ptrs)
        b.restart()              # init is outside the timed region -- the feasible start is written out here
    lead.sync()

    def pass_(index):
        b = batches[index % n_batches]
        if index >= n_batches:
            b.restart()      # back to the feasible start of the positions it already holds (device side only)
        b.solve(GAP_TOL, MAX_ITER, 0)
        return b
    summary = torch.zeros(4, dtype=torch.float64, device=torch.device("cuda", local_rank))

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- untimed: communicator set-up, conditioning, then the W warmup passes ----
    if grouped:
        sharding.allreduce_summary(summary.clone().to(coll_dev), force=True)     # RCCL communicator setup outside the timed region ...
        barrier()                                                                # ... and before the conditioning: it idles the GPU for a while
    # Conditioning: the chip is brought to the clocks it HOLDS under this load before anything is timed.  From idle the power
    # controller first over-reacts (a launch takes 0.19 ms, then 0.21-0.23 ms between 2 and 6 ms, then settles at 0.17 ms after ~20 ms:
    # profiles/r5_transient.log; 50 ms of idling bring the whole transient back), so a K = 20 region one millisecond after idle times the
    # controller, not the kernel.  Sized by TIME, not by count: about 30 ms of solves, 160 launches at 2^20 problems and
    # proportionally more of a smaller batch, on a RING of scratch batches -- at most 16, at most a tenth of the free device memory (round 5
    # held one scratch batch per launch: 21.8 GB at the default size, growing with --problems-per-gpu) -- each put back on its feasible start
    # (k_restart_feasible: memory-bound, ~20 us) right before it is solved again.  The scratch batches are not among the timed ones;
    # `value_from_idle` reports the K timed launches' figure with no conditioning at all.
    if args.condition_launches is None:
        cond_launches = int(min(4096, max(1, round(160.0 * N_PER_GPU / max(count, 1)))))
    else:
        cond_launches = max(args.condition_launches, 0)
    free_b, _total_b = torch.cuda.mem_get_info()
    per_batch_b = 200e6 * (count / float(N_PER_GPU)) + 1e6                 # fields + progress words + records + maps, generously
    ring = int(max(1, min(16, cond_launches, 0.1 * free_b / per_batch_b))) if cond_launches > 0 else 0
    scratch = [rp.Batch(count, rp.VARIANT_F3, rp.DTYPE_F64, device=local_rank, stream=stream) for _ in range(ring)]
    for b in scratch:
        b.set_problems_device(*ptrs)
    scratch_fresh = [True]
    conditioning_launches_done = [0]

    def condition():
        for j in range(cond_launches):
            b = scratch[j % ring]
            if not (scratch_fresh[0] and j < ring):
                b.restart()
            b.solve(GAP_TOL, MAX_ITER, 0)      # (the first use forms each start in registers: k_solve_chunks<START>, the same arithmetic)
        scratch_fresh[0] = False
        conditioning_launches_done[0] += cond_launches
    condition()
    for i in range(W):
        pass_(i)
    barrier()

    # ---- timed: exactly K passes, then the final summary reduction (+ all-reduce) ----
    # (the host clock is read at three more points inside the region, so that the line can say where an N-rank run's time went:
    # this rank's launches + reduction done / the summary all-reduce returned / the closing barrier passed)
    t0 = time.perf_counter()
    lead.event_record(0)
    last = lead
    half = K // 2
    for i in range(W, W + K):
        if i == W + half and 0 < half < K:
            lead.event_record(6)         # one extra event: the second half of the launches, timed separately (sustained clocks)
        last = pass_(i)
    lead.event_record(1)
    last.reduce_device(summary.data_ptr())
    lead.sync()
    t_kernels = time.perf_counter()
    summary = sharding.allreduce_summary(summary.to(coll_dev), force=grouped)
    if grouped and not rehearsal:
        torch.cuda.synchronize()         # the all-reduce is asynchronous on the device: its end, not its enqueue
    t_coll = time.perf_counter()
    barrier()
    elapsed = time.perf_counter() - t0
    t_end = t0 + elapsed

    kernel_ms = lead.event_elapsed_ms(0, 1) / max(K, 1)
    sustained_ms = lead.event_elapsed_ms(6, 1) / (K - half) if 0 < half < K else kernel_ms
    # every pass solves the same seeded batch from the same start: steps per pass are those of any solved batch
    one = last.reduce()
    steps_local = one["total_steps"] * K
    conv_local = one["n_converged"] * K
    t = torch.tensor([elapsed, steps_local, conv_local], dtype=torch.float64, device=coll_dev)
    if grouped:
        tm = t[:1].clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        ts = t[1:].clone()
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        elapsed, steps_all, conv_all = float(tm[0]), float(ts[0]), float(ts[1])
    else:
        steps_all, conv_all = steps_local, conv_local
    g = sharding.summary_dict(summary)
    # per rank: launch time from its HIP events, the three host-clock pieces of the timed region, the device it really sat on
    mine = [kernel_ms * K, sustained_ms, (t_kernels - t0) * 1e3, (t_coll - t_kernels) * 1e3, (t_end - t_coll) * 1e3, steps_local]
    my_dev = rp.device_id(local_rank)
    if grouped:
        rows = [torch.zeros(len(mine), dtype=torch.float64, device=coll_dev) for _ in range(world)]
        dist.all_gather(rows, torch.tensor(mine, dtype=torch.float64, device=coll_dev))
        per_rank = [[float(x) for x in r.tolist()] for r in rows]
        devs = [None] * world
        dist.all_gather_object(devs, my_dev)
    else:
        per_rank, devs = [mine], [my_dev]

    if rank != 0:
        if grouped:
            dist.barrier()
            dist.destroy_process_group()
        return

    steps_per_launch = steps_local / max(K, 1)
    alg_gbs = B_ALG_F3 * steps_per_launch / (kernel_ms * 1e-3) / 1e9
    prov = profile_provenance()
    fresh = prov["current"]      # the kernel sources are the ones the committed counters were collected from
    traffic, traffic_src = profile_number(PROFILE_TAG + "_hbm_traffic.json", "k_solve_chunks_f3_f64", "hbm_bytes_per_launch")
    flop_per_step, flop_src = profile_number(PROFILE_TAG + "_sq_counters.json", "_flop_per_gated_newton_step")      # the gated kernel itself
    if flop_per_step is None:
        flop_per_step, flop_src = profile_number(PROFILE_TAG + "_sq_counters.json", "_flop_per_newton_step")
    if flop_per_step is None:
        flop_per_step, flop_src = 560.0, "estimate (no profiles/%s_sq_counters.json)" % PROFILE_TAG
    tflops = flop_per_step * steps_per_launch / (kernel_ms * 1e-3) / 1e12
    winsts, winsts_src = profile_number(PROFILE_TAG + "_sq_counters.json", "_valu_wave_insts_per_gated_launch")
    VALU_ISSUE_PEAK_G = 256 * 4 * 2.4 / 4      # G wave-instructions/s at the fp64 rate
    valu_issue = None
    if winsts is not None and abs(count - N_PER_GPU) == 0 and fresh:
        valu_issue = {"achieved": winsts / (kernel_ms * 1e-3) / 1e9, "peak": VALU_ISSUE_PEAK_G, "unit": "G wave-instructions/s",
                      "frac": winsts / (kernel_ms * 1e-3) / 1e9 / VALU_ISSUE_PEAK_G,
                      "valu_wave_instructions_per_launch": winsts, "source": winsts_src,
                      "note": "SQ_INSTS_VALU of this launch (1,048,576 problems, idle lanes included) over this run's launch time; every "
                              "instruction priced at the fp64 issue cost and the peak at the 2.4 GHz the chip does not hold under this load "
                              "(~2.0 GHz): the remainder is clock, not stalls"}
    line = {
        "metric": "interior-point Newton steps/sec (whole node) + achieved HBM GB/s, 1M-problem batch",
        "value": steps_all / elapsed,
        "unit": "Newton steps/s",
        "n_gpus": world,
        "ranks_in_process_group": dist.get_world_size() if grouped else 1,      # what RCCL (torch.distributed "nccl") saw
        "process_group_backend": (dist.get_backend() if grouped else None),    # "nccl" = RCCL; None: no group was created (N = 1 without --force-process-group)
        "self_launched": bool(os.environ.get("RP_BENCH_SELF_LAUNCHED")),
        "hip_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),      # set to 8 by this program unless the caller's environment had a value (see main)
        "steps": K,
        "warmup": W,
        # everything launched between the last idle moment and the timed region: the conditioning solves + the W warmup passes the driver asked for
        "untimed_launches_before_timed_region": cond_launches + W,
        "ms_per_step": elapsed / max(K, 1) * 1e3,
        # where the timed region's wall clock went (max over ranks is what `value` divides by): every rank's K launches by its own HIP
        # events, then three host-clock pieces -- launches + local reduction done, summary all-reduce done, closing barrier passed.  A
        # rank that finishes early waits in the collective, so collective_ms_min is the collective's own latency and the spread of
        # kernels_ms is the ranks' skew; `devices` must be N different GPUs for an N-GPU figure
        "timed_region_breakdown": {
            "region_ms": elapsed * 1e3,
            "kernels_ms_per_rank": [r[0] for r in per_rank], "kernels_ms_min": min(r[0] for r in per_rank), "kernels_ms_max": max(r[0] for r in per_rank),
            "sustained_ms_per_launch_per_rank": [r[1] for r in per_rank],
            "host_until_local_work_done_ms_per_rank": [r[2] for r in per_rank],
            "collective_ms_per_rank": [r[3] for r in per_rank], "collective_ms_min": min(r[3] for r in per_rank), "collective_ms_max": max(r[3] for r in per_rank),
            "barrier_ms_per_rank": [r[4] for r in per_rank], "barrier_ms_max": max(r[4] for r in per_rank),
            "note": "kernels: HIP events on the rank's stream around its K launches; the other pieces: that rank's host clock"},
        "devices": devs,
        "devices_distinct": len(set(devs)) == len(devs),
        "devices_note": (None if len(set(devs)) == len(devs) else
                         ("REHEARSAL: every rank on one GPU -- not an N-GPU measurement" if rehearsal else "RANKS SHARE A GPU: not an N-GPU measurement")),
        "per_gpu_newton_steps_per_s": [r[5] / (r[0] * 1e-3) if r[0] > 0 else None for r in per_rank],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "conditioning": {"untimed_solves_before_the_warmup": cond_launches, "scratch_batches_in_the_ring": ring,
                         "why": "steady-state clocks: the power controller's transient after idle lasts ~20 ms (profiles/r5_transient.log); "
                                "`value_from_idle` is the same K launches from an idle chip, what rounds 1-4 reported as `value`"},
        "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one GPU, gloo collectives -- not a result)",
        "config": {
            "workload": "BASELINE configs[2] (C3): %d F3 onedpath_ip problems per GPU, convergence-gated "
                        "(surrogate gap < 1e-8 checked before every step, cap 200), fp64, monotone seeded positions, "
                        "feasible-start rule; one fused launch per batch; start states laid out in HBM, in the batch's scheduled "
                        "order (precomputed by set_problems), before the timed region -- `end_to_end` times the same batch "
                        "from bare positions; the chip is under the same load for ~30 ms before the warmup (`conditioning`), "
                        "`value_from_idle` is the figure from idle" % args.problems_per_gpu,
            "problems_per_gpu": args.problems_per_gpu,
            "problems_total": n_total,
            "newton_steps_per_pass_per_gpu": steps_per_launch,
            "mean_steps_per_problem": steps_per_launch / count,
            "converged_fraction": conv_all / (n_total * max(K, 1)),
            "final_summary": g,
            "sharding": "contiguous shards, no data-path collective; one 32-byte RCCL all-reduce at the end" if world > 1
                        else "single GPU",
        },
        # The roof that binds the timed kernel.  The fused solve keeps a problem's state in VGPRs from its first step to
        # its gate and is fp64-VALU bound (VALU busy ~1.0 across the resident waves): flop per Newton step from the
        # rocprofv3 SQ counters (2 x FMA + MUL + ADD + TRANS wave-instructions of an all-lanes-active launch, per lane-step).
        "roofline": {
            "bound": "fp64_valu", "kernel": "k_solve_chunks<double, double, F3> (fused gated solve, one 64-problem chunk of the scheduled order per wave)",
            "achieved": tflops if fresh else None, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_PEAK_TFLOPS if fresh else None,
            "flop_per_newton_step": flop_per_step, "flop_per_newton_step_source": flop_src, "counters_provenance": prov,
            "avg_launch_ms": kernel_ms, "sustained_ms_per_launch": sustained_ms, "newton_steps_per_launch": steps_per_launch,
            "sustained_newton_steps_per_s": steps_per_launch / (sustained_ms * 1e-3),
            "traffic": traffic if fresh else None, "traffic_source": traffic_src,
            "note": "flop actually executed by the timed kernel (SQ counters of the same kernel on identical problems, per lane-step); "
                    "idle lane-steps of the gated solve (~1 % in the scheduled order) are not counted; traffic = HBM bytes per launch from FETCH_SIZE (x2, "
                    "calibrated) + WRITE_SIZE: each state crosses HBM once per solve.  sustained_ms_per_launch = the second half of the K "
                    "launches (the chip drops its clock after ~2 ms of fp64 load, so the first launches of a run are faster); "
                    "avg_launch_ms = all K.  A cheaper step shows as a smaller flop fraction at a higher step rate: the vector ALU is "
                    "issue-saturated either way (valu_issue below is the roof in the unit that binds)",
            # the roof in the unit that binds: wave-level VALU instructions issued per second against one fp64-rate instruction
            # per 4 cycles per SIMD (256 CU x 4 SIMD x 2.4 GHz / 4)
            "valu_issue": valu_issue,
            "same_rate_at_round1_operation_count": {"flop_per_newton_step": 609.4, "achieved": 609.4 * steps_per_launch / (kernel_ms * 1e-3) / 1e12,
                                                    "frac": 609.4 * steps_per_launch / (kernel_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                                    "note": "what the round-1 kernel would have had to sustain for this step rate (for comparison across rounds only)"},
            # SURVEY 8d's definition, kept as a labelled secondary: algorithmic bytes (216 B x Newton steps executed in
            # the launch) / launch time.  Not a fraction of anything for a fused launch -- see per_step_launch.
            "hbm_algorithmic": {
                "bound": "hbm", "achieved": alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "ratio_to_peak": alg_gbs / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": B_ALG_F3 * steps_per_launch,
                "note": "can exceed 1: the fused launch moves ~1/15 of these bytes (traffic above)",
            },
        },
    }

    # ---- opt-in: the steady state (--sustain-seconds S).  The timed region above is ~4 ms, half of it before the power controller has
    # reacted.  Here a LARGE pool of batches (up to 512 x 136 MB: HBM holds them) is solved back to back -- ~0.1 s of uninterrupted
    # fp64 solves per round, HIP-event timed, its second half separately -- then every batch is put back on its start (memory-bound,
    # ~8 % of a round) and the next round follows, for at least S seconds, with the device's hwmon files (clock, power, temperature) sampled beside it ----
    def sustain(seconds):
        import threading
        free_b, _total_b = torch.cuda.mem_get_info()
        per_batch = 200e6 * (count / float(N_PER_GPU))                     # fields + progress words + records + maps, generously
        n_big = int(max(len(batches), min(512, 0.6 * free_b / max(per_batch, 1.0))))
        big = list(batches)
        try:
            while len(big) < n_big:
                nb = rp.Batch(count, rp.VARIANT_F3, rp.DTYPE_F64, device=local_rank, stream=stream)
                nb.set_problems_device(*ptrs)
                big.append(nb)
        except rp.RpError:
            pass                                                           # as many as fit
        samples, stop = [], [False]
        # Clock, power and temperature straight from the amdgpu hwmon files of THIS device (found by the PCI bus id rp_device_id reports).
        # No child process: a process that has initialised the GPU must not fork-and-exec helpers (rocm-smi is a `#!/usr/bin/env python3`
        # script: under rocprofv3 its second exec is exactly the hop the GPU boxes refuse).
        import glob
        pci = (rp.device_id(local_rank).split() + ["", ""])[1].lower()
        hw = sorted(glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % pci))
        fields = (("power_W", "power1_input", 1e-6), ("power_W", "power1_average", 1e-6), ("sclk_MHz", "freq1_input", 1e-6), ("mclk_MHz", "freq2_input", 1e-6),
                  ("junction_C", "temp2_input", 1e-3), ("memory_C", "temp3_input", 1e-3), ("power_cap_W", "power1_cap", 1e-6))

        def poll():
            while not stop[0]:
                d = {}
                for key, fname, scale in fields:
                    if key in d or not hw:
                        continue
                    try:
                        with open(os.path.join(hw[0], fname)) as fh:
                            d[key] = float(fh.read().strip()) * scale
                    except (OSError, ValueError):
                        pass
                samples.append((time.perf_counter(), d))
                time.sleep(0.02)
        th = threading.Thread(target=poll, daemon=True)
        th.start()
        n_big_used = len(big)
        try:      # whatever happens below, the sampler stops and the extra batches are given back
            time.sleep(0.3)                        # a few idle samples first
            rounds = []
            lead.sync()
            t_start = time.perf_counter()
            halfway = len(big) // 2
            while True:
                for b in big:
                    b.restart()
                lead.event_record(2)
                for j, b in enumerate(big):
                    if j == halfway:
                        lead.event_record(7)
                    b.solve(GAP_TOL, MAX_ITER, 0)
                lead.event_record(3)
                lead.sync()
                now = time.perf_counter()
                rounds.append((now - t_start, lead.event_elapsed_ms(2, 3) / len(big), lead.event_elapsed_ms(7, 3) / (len(big) - halfway)))
                if now - t_start >= seconds:
                    break
            t_stop = time.perf_counter()
            time.sleep(0.3)
        finally:
            stop[0] = True
            th.join()
            for nb in big[len(batches):]:
                nb.close()
            del big[len(batches):]
        spl = steps_local / max(K, 1)
        late = [r for r in rounds if r[0] >= 0.5 * rounds[-1][0]] or [rounds[-1]]
        ms_whole, ms_half = float(np.mean([r[1] for r in late])), float(np.mean([r[2] for r in late]))

        def numeric(window):
            cols = {}
            for _, d in window:
                for k, v in d.items():
                    cols.setdefault(k, []).append(float(v))
            return {k: {"min": min(v), "median": float(np.median(v)), "max": max(v)} for k, v in cols.items()}
        under = [x for x in samples if t_start + 0.5 * (t_stop - t_start) <= x[0] <= t_stop]
        idle = [x for x in samples if x[0] < t_start]
        smi = numeric(under)
        power = smi.get("power_W", {}).get("median")
        return {"seconds": t_stop - t_start, "rounds": len(rounds), "batches_per_round": n_big_used,
                "uninterrupted_solve_ms_per_round": ms_whole * n_big_used,
                "ms_per_launch_first_round": rounds[0][1], "ms_per_launch_later_rounds": ms_whole, "ms_per_launch_second_half_of_later_rounds": ms_half,
                "steady_newton_steps_per_s": spl / (ms_half * 1e-3),
                "wall_clock_newton_steps_per_s_including_the_restarts": spl * n_big_used * len(rounds) / (t_stop - t_start),
                "joule_per_newton_step_at_median_power": (power / (spl / (ms_whole * 1e-3))) if power else None,
                "hwmon_idle_before": numeric(idle), "hwmon_second_half_under_load": smi, "hwmon_samples": len(samples),
                "hwmon_source": (hw[0] if hw else None),
                "note": "per round every batch of a large pool is put back on its feasible start (k_restart_feasible, outside the events, ~8 % of the "
                        "round: the chip sees a short memory-bound breather there) and then all are solved back to back (k_solve_chunks, HIP events "
                        "around the solves only; the second half of each solve phase timed separately); steady = second halves of the later rounds"}

    # ---- end to end: bare positions -> solutions, per fresh batch (nothing precomputed), HIP-event timed ----
    def end_to_end(reps):
        use = batches[:min(len(batches), reps)]
        for b in use[:2]:                     # untimed: first-use allocations of the scheduling scratch are long done
            b.set_problems_device(*ptrs)
            b.solve(GAP_TOL, MAX_ITER, 0)
        condition()                           # steady clocks, as for the headline (the loops below follow each other without idling)
        lead.sync()
        lead.event_record(2)
        for b in use:
            b.set_problems_device(*ptrs)      # scheduled order + positions into the batch (three kernels), nothing else
        lead.event_record(3)
        lead.sync()
        sched_ms = lead.event_elapsed_ms(2, 3) / len(use)
        lead.event_record(2)
        for b in use:
            b.set_problems_device(*ptrs)
            b.solve(GAP_TOL, MAX_ITER, 0)     # forms the feasible start in registers, solves, stores
        lead.event_record(3)
        lead.sync()
        ms = lead.event_elapsed_ms(2, 3) / len(use)
        chk = use[-1].reduce()
        # ... and with the answers where a caller can use them: in PROBLEM order, in device memory (the batch keeps its problems in
        # scheduled order; in the reference the answer of problem i is var[] of trajectory i, onedpath_ip.cpp:47-52).  Two forms:
        # (a) a solution buffer bound to the batch -- the solve itself writes each problem's 32-byte record as the problem leaves it;
        # (b) a separate pass after the solve (rp_batch_solution_device: walks positions, scatters whole sectors).
        sol = torch.empty((count, 4), dtype=torch.float64, device=torch.device("cuda", local_rank))      # n x 32 B
        for b in use:
            b.bind_solution(sol.data_ptr())
        use[0].set_problems_device(*ptrs)
        use[0].solve(GAP_TOL, MAX_ITER, 0)
        lead.sync()
        lead.event_record(2)
        for b in use:
            b.set_problems_device(*ptrs)
            b.solve(GAP_TOL, MAX_ITER, 0)
        lead.event_record(3)
        lead.sync()
        ms_bound = lead.event_elapsed_ms(2, 3) / len(use)
        rec_bound = sol.clone()
        for b in use:
            b.bind_solution(None)
        sol.zero_()
        lead.event_record(2)
        for b in use:
            b.set_problems_device(*ptrs)
            b.solve(GAP_TOL, MAX_ITER, 0)
            b.solution_device(sol.data_ptr())
        lead.event_record(3)
        lead.sync()
        ms_gather = lead.event_elapsed_ms(2, 3) / len(use)
        lead.event_record(2)
        for b in use:
            b.solution_device(sol.data_ptr())
        lead.event_record(3)
        lead.sync()
        ms_gather_alone = lead.event_elapsed_ms(2, 3) / len(use)
        same_records = bool(torch.equal(rec_bound.view(torch.int64), sol.view(torch.int64)))
        steps_in_records = int(sol.view(torch.int32)[:, 6].sum().item())      # the iteration counts of the records (word 6 of 8)
        del sol
        # The product's entry point for this: rp_pipeline (include/rp_batch.h, round 6) -- job after job of "positions in, solutions out in
        # problem order", the batches dealt alternately onto two streams, so that job i + 1's scheduling pass (three small memory- and
        # latency-bound kernels) and the head of its solve run under the drain of job i's solve.  Wall clock (HIP events of one stream do
        # not span two) over `jobs` jobs after conditioning; the same pipeline on ONE stream, timed the same way, beside it.
        def run_pipeline(n_streams, jobs):
            outs = [torch.empty((count, 4), dtype=torch.float64, device=torch.device("cuda", local_rank)) for _ in range(4)]
            with rp.Pipeline(count, rp.VARIANT_F3, rp.DTYPE_F64, device=local_rank, depth=4, n_streams=n_streams) as pipe:
                best = None
                for _ in range(2):
                    # untimed: the pipeline's own load for ~30 ms (first-use allocations, and the clocks this mix of kernels settles at --
                    # conditioning with bare solves and then switching load left the first jobs of the burst in another transient)
                    for j in range(max(8, cond_launches)):
                        pipe.submit(*ptrs, d_out=outs[j % 4].data_ptr(), gap_tol=GAP_TOL, max_iter=MAX_ITER)
                    pipe.wait()
                    t_p = time.perf_counter()
                    for j in range(jobs):
                        last_job = pipe.submit(*ptrs, d_out=outs[j % 4].data_ptr(), gap_tol=GAP_TOL, max_iter=MAX_ITER)
                    pipe.wait()
                    dt_p = (time.perf_counter() - t_p) / jobs * 1e3
                    best = dt_p if best is None else min(best, dt_p)
                rec = outs[(jobs - 1) % 4].clone()
                tot = pipe.batch(last_job).reduce()["total_steps"]
            del outs
            return best, rec, tot
        pipe_jobs = 120
        try:
            ms_p2, rec_p2, tot_p2 = run_pipeline(2, pipe_jobs)
            ms_p1, rec_p1, tot_p1 = run_pipeline(1, pipe_jobs)
            pipelined = {"entry_point": "rp_pipeline_submit, depth 4, n_streams 2: per job rp_batch_bind_solution + rp_batch_set_problems_device + rp_batch_solve on "
                                        "stream (job % 4) % 2", "jobs": pipe_jobs,
                         "ms_per_batch": ms_p2, "newton_steps_per_s": tot_p2 / (ms_p2 * 1e-3),
                         "same_pipeline_on_one_stream": {"ms_per_batch": ms_p1, "newton_steps_per_s": tot_p1 / (ms_p1 * 1e-3)},
                         "gain_over_one_stream": ms_p1 / ms_p2,
                         "solutions_bitwise_equal_to_the_one_stream_path": bool(torch.equal(rec_p2.view(torch.int64), rec_bound.view(torch.int64))
                                                                                and torch.equal(rec_p1.view(torch.int64), rec_bound.view(torch.int64))),
                         "ratio_to_the_headline_value": (tot_p2 / (ms_p2 * 1e-3)) / (steps_all / elapsed) if world == 1 else None,
                         "note": "wall clock over %d jobs after conditioning, host enqueue and the final synchronisation included; every job starts from bare "
                                 "positions in device memory and ends with its rp_solution records in problem order" % pipe_jobs}
            del rec_p2, rec_p1
        except Exception as exc:      # an extra: never takes the benchmark down
            pipelined = {"error": str(exc)}
        del rec_bound
        e2e_ms = pipelined.get("ms_per_batch", ms)
        return {"workload": "per fresh batch of %d problems: bare positions in device memory -> the scheduled order (rp_batch_set_problems_device: "
                            "k_sched_count / k_sched_scan / k_sched_scatter) -> the fused gated solve from the feasible start formed in registers "
                            "(k_solve_chunks<START>) -> rp_solution records in problem order; nothing precomputed, no host synchronisation in between.  "
                            "ms_per_batch / newton_steps_per_s: through rp_pipeline on two streams (the product's entry point for this); "
                            "`one_stream_by_hand`: the same calls issued by hand on one stream, HIP-event timed" % count,
                "batches": pipelined.get("jobs", len(use)), "ms_per_batch": e2e_ms, "newton_steps_per_s": chk["total_steps"] / (e2e_ms * 1e-3),
                "pipeline": pipelined,
                "one_stream_by_hand": {"batches": len(use), "ms_per_batch": ms, "newton_steps_per_s": chk["total_steps"] / (ms * 1e-3)},
                "with_solutions_in_problem_order": {
                    "workload": "one stream, by hand, ending with every problem's rp_solution record (vel1, duration0, duration1, iters, status: 32 B) "
                                "in PROBLEM order in device memory",
                    "bound_buffer": {"ms_per_batch": ms_bound, "newton_steps_per_s": chk["total_steps"] / (ms_bound * 1e-3),
                                     "note": "rp_batch_bind_solution: k_solve_chunks writes each record itself (one scattered sector per problem)"},
                    "separate_pass": {"ms_per_batch": ms_gather, "newton_steps_per_s": chk["total_steps"] / (ms_gather * 1e-3),
                                      "solution_pass_alone_ms": ms_gather_alone,
                                      "solution_pass_GBps_on_68_B_per_problem": 68.0 * count / (ms_gather_alone * 1e-3) / 1e9,
                                      "note": "rp_batch_solution_device after the solve (k_solution: 32 B read coalesced + 4 B map + one 32 B sector written per problem)"},
                    "both_forms_bitwise_equal": same_records, "steps_summed_from_the_records": steps_in_records},
                "set_problems_device_ms": sched_ms, "schedule_fraction_of_batch": sched_ms / ms,
                "newton_steps_per_batch": chk["total_steps"], "converged_fraction": chk["n_converged"] / count,
                "headline_for_comparison_ms": kernel_ms,
                "note": "the headline's timed region starts from start states already laid out in scheduled order (SURVEY 8d: init "
                        "excluded); this block is what a caller pays who hands over positions.  Round 2: 0.500 ms per batch (32.6 G "
                        "steps/s): rocPRIM sort 0.188 ms + feasible start 0.067 ms + solve 0.245 ms"}
    line["end_to_end"] = end_to_end(20)

    if not args.no_extras:
        # (a) one launch per Newton step: the HBM-streaming form of the same step (216 B really move per step).
        #     cold = every launch on a batch not touched since its init (state comes from HBM);
        #     warm = the same 128 MiB batch stepped again and again (reported for completeness: with nontemporal accesses ~cold);
        #     probe = the same kernel with zero steps: its 16 loads + 11 stores per problem and nothing else,
        #             i.e. what this access pattern can reach on this box (the kernel's own ceiling).
        spare = batches[:min(len(batches), 20)]

        def reinit():
            for b in spare:
                b.set_problems_device(*ptrs)
                b.restart()          # the start state written out: the timed launches below are steps and nothing else
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
        reinit()
        ms_probe = sweep(lambda b: b.traffic_probe())
        lead.set_problems_device(*ptrs)
        lead.restart()
        lead.step(1)
        lead.event_record(2)
        for _ in range(len(spare)):
            lead.step(1)
        lead.event_record(3)
        lead.sync()
        ms_warm = lead.event_elapsed_ms(2, 3) / len(spare)
        # box ceiling in the same run: a plain 16 B/lane device copy moving the bytes one k = 1 launch moves
        # (100 MiB in, 100 MiB out), every pair of buffers touched once
        nel = int(B_MOVED_F3_ZV * count / 2 / 8)
        pairs = [(torch.empty(nel, dtype=torch.float64, device=torch.device("cuda", local_rank)).fill_(1.0),
                  torch.empty(nel, dtype=torch.float64, device=torch.device("cuda", local_rank))) for _ in range(8)]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for src, dst in pairs:
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        ms_copy = e0.elapsed_time(e1) / len(pairs)
        del pairs
        moved = lambda ms: B_MOVED_F3_ZV * count / (ms * 1e-3) / 1e9   # noqa: E731
        k1_traffic, k1_src = profile_number(PROFILE_TAG + "_hbm_traffic.json", "k_newton_stream16_f3_f64", "hbm_bytes_per_launch")
        line["per_step_launch"] = {
            "kernel": "k_newton_stream16<double, double, F3, zero end velocities>, k = 1: one launch per Newton step, 16 B per lane "
                      "(two problems per lane)",
            "avg_launch_ms": ms_cold, "newton_steps_per_s": count / (ms_cold * 1e-3),
            "roofline": {"bound": "hbm", "bytes_moved_per_step": B_MOVED_F3_ZV, "achieved": moved(ms_cold), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": moved(ms_cold) / HBM_PEAK_GBS,
                         "traffic": k1_traffic, "traffic_source": k1_src,
                         "note": "priced on the 200 B that move (14 fields read, 11 written: the reference's inits leave the two end "
                                 "velocities at zero and the kernel instantiated for that does not read them); on SURVEY 8d's "
                                 "algorithmic 216 B the same launch is %.0f GB/s (%.3f of peak)" % (
                                     B_ALG_F3 * count / (ms_cold * 1e-3) / 1e9, B_ALG_F3 * count / (ms_cold * 1e-3) / 1e9 / HBM_PEAK_GBS)},
            "same_run_ceilings": {
                "same_kernel_without_arithmetic_GBps": moved(ms_probe), "kernel_vs_that": ms_probe / ms_cold,
                "device_copy_16B_per_lane_GBps": moved(ms_copy), "kernel_vs_copy": ms_copy / ms_cold,
                "note": "k = 0 launch of the same kernel (14 loads + 11 stores per problem, nothing else) and a torch device copy of "
                        "the same byte count, both cold, both timed in this run"},
            "same_batch_stepped_repeatedly_GBps": moved(ms_warm),      # nontemporal accesses: the 128 MiB state no longer lingers in the Infinity Cache
            "launches": len(spare)}
        # (b) configs[1]: 65,536 problems, exactly 50 steps each, one fused launch
        n2 = min(65536, count)
        with rp.Batch(n2, rp.VARIANT_F3, rp.DTYPE_F64, device=local_rank, stream=stream) as c2:
            ms = []
            for _ in range(4):
                c2.set_problems_device(*ptrs)
                c2.restart()
                c2.sync()
                c2.event_record(4)
                c2.step(50)
                c2.event_record(5)
                c2.sync()
                ms.append(c2.event_elapsed_ms(4, 5))
        # ... and a gated solve of the same 65,536 problems beside it (the form a caller who wants solutions would run)
        with rp.Batch(n2, rp.VARIANT_F3, rp.DTYPE_F64, device=local_rank, stream=stream) as c2g:
            msg = []
            for _ in range(4):
                c2g.set_problems_device(*ptrs)
                c2g.restart()
                c2g.sync()
                c2g.event_record(4)
                c2g.solve(GAP_TOL, MAX_ITER, 0)
                c2g.event_record(5)
                c2g.sync()
                msg.append(c2g.event_elapsed_ms(4, 5))
            gated_steps = c2g.reduce()["total_steps"]
        t50 = min(ms[1:])
        f50_flop, f50_src = profile_number(PROFILE_TAG + "_sq_counters.json", "_flop_per_fixed50_launch")
        f50_insts, _ = profile_number(PROFILE_TAG + "_sq_counters.json", "_valu_wave_insts_per_fixed50_launch")
        f50_all, _ = profile_number(PROFILE_TAG + "_sq_counters.json", "_wave_insts_per_fixed50_launch")
        f50_roof = None
        try:      # instructions per wave-step of the two regimes, read from the committed counters (never typed into this file: ADVICE r5)
            f50_phase = json.load(open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_sq_counters.json"))).get("_fixed50_per_wave_step") if fresh else None
        except Exception:
            f50_phase = None
        if f50_flop is not None and fresh and n2 == 65536:
            tf = f50_flop / (t50 * 1e-3) / 1e12
            f50_roof = {"bound": "fp64_valu", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS,
                        "flop_per_launch": f50_flop, "source": f50_src,
                        "valu_issue": None if f50_insts is None else {
                            "achieved": f50_insts / (t50 * 1e-3) / 1e9, "peak": VALU_ISSUE_PEAK_G, "unit": "G wave-instructions/s",
                            "frac": f50_insts / (t50 * 1e-3) / 1e9 / VALU_ISSUE_PEAK_G, "valu_wave_instructions_per_launch": f50_insts,
                            "all_wave_instructions_per_launch": f50_all},
                        "per_wave_step": f50_phase,
                        "note": "65,536 problems are 1,024 waves: ONE wave per SIMD; a lone wave issues one instruction of ANY kind (vector, scalar, "
                                "branch) per ~2.3 ns plus its dependency stalls, so the launch is bound by the LENGTH of one wave's instruction stream, "
                                "not by the chip's arithmetic (`per_wave_step`: the SQ counters of this launch shape split by phase, from %s)" % f50_src}
        line["fixed50"] = {"workload": "BASELINE configs[1] (C2): 65,536 problems x exactly 50 steps, one fused launch "
                                       "(about 35 of the 50 steps per problem run in the reference's post-convergence regime: "
                                       "~48 residual halvings per step)",
                           "ms": t50, "newton_steps_per_s": n2 * 50 / (t50 * 1e-3), "roofline": f50_roof,
                           "gated_solve_of_the_same_problems": {"ms": min(msg[1:]), "newton_steps": gated_steps,
                                                                "newton_steps_per_s": gated_steps / (min(msg[1:]) * 1e-3)}}
        # (c) configs[4]: F4, 1,048,576 problems x 50 steps from the feasible start, fp32 state (76 B algorithmic per step), in
        #     both arithmetic modes: fp32 state + fp64 arithmetic (per-problem parity bound, tests/test_gpu_parity.py) and
        #     pure fp32 arithmetic (statistical bound only)
        def f4_mode(dtype, tag, peak_tflops, flop_key):
            with rp.Batch(count, rp.VARIANT_F4, dtype, device=local_rank, stream=stream) as c5:
                ms50, ms1, ms50_idle = [], [], []
                for rep_i in range(5):
                    c5.set_problems_device(*ptrs)
                    c5.restart()
                    if rep_i < 3:
                        condition()          # (the 50-step launch is arithmetic-bound for 1-1.5 ms: steady clocks for it too)
                        c5.sync()
                    else:
                        c5.sync()
                        time.sleep(0.2)      # ... and from an idle chip, as rounds 1-4 measured it (the launch then runs inside the boost window)
                    c5.event_record(4)
                    c5.step(50)
                    c5.event_record(5)
                    c5.sync()
                    (ms50 if rep_i < 3 else ms50_idle).append(c5.event_elapsed_ms(4, 5))
                for _ in range(4):
                    c5.set_problems_device(*ptrs)
                    c5.restart()
                    c5.sync()
                    c5.event_record(4)
                    c5.step(1)
                    c5.event_record(5)
                    c5.sync()
                    ms1.append(c5.event_elapsed_ms(4, 5))
            t50, t1 = min(ms50), min(ms1[1:])
            flop, flop_src = profile_number(PROFILE_TAG + "_sq_counters.json", flop_key)
            k1_traffic, k1_src = profile_number(PROFILE_TAG + "_hbm_traffic.json", "k_newton_stream16_f4_" + tag, "hbm_bytes_per_launch")
            out = {"fused_50_steps_ms": t50, "newton_steps_per_s": count * 50 / (t50 * 1e-3),
                   "from_an_idle_chip": {"fused_50_steps_ms": min(ms50_idle), "newton_steps_per_s": count * 50 / (min(ms50_idle) * 1e-3),
                                         "note": "the same launch 0.2 s after the last work: how rounds 1-4 timed it"},
                   "k1_launch_ms": t1,
                   "k1_roofline": {"bound": "hbm", "bytes_moved_per_step": B_MOVED_F4_F32_ZV, "achieved": B_MOVED_F4_F32_ZV * count / (t1 * 1e-3) / 1e9,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": B_MOVED_F4_F32_ZV * count / (t1 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "traffic": k1_traffic, "traffic_source": k1_src,
                                   "note": "68 B move per step (10 floats read, 7 written); SURVEY 8d's algorithmic 76 B gives %.0f GB/s" % (
                                       B_ALG_F4_F32 * count / (t1 * 1e-3) / 1e9)}}
            if flop is not None and fresh:
                tf = flop * count * 50 / (t50 * 1e-3) / 1e12
                out["fused_roofline"] = {"bound": "fp64_valu" if peak_tflops == FP64_PEAK_TFLOPS else "fp32_valu", "achieved": tf,
                                         "peak": peak_tflops, "unit": "TFLOP/s", "frac": tf / peak_tflops,
                                         "flop_per_newton_step": flop, "flop_per_newton_step_source": flop_src}
            return out
        f4_state = f4_mode(rp.DTYPE_F32_STATE, "f32state", FP64_PEAK_TFLOPS, "_flop_per_f4_step_f32state")
        f4_pure = f4_mode(rp.DTYPE_F32, "f32", FP32_PEAK_TFLOPS, "_flop_per_f4_step_f32")
        f4_pure["parity"] = ("STATISTICAL ONLY: against the fp32-state mode from the same states 0.05 % of the problems take another line-search decision "
                             "(worst one-step difference 2.4e-2); the others stay within 4.6e-4 (tests/test_gpu_parity.py asserts 2e-3 per problem).  SURVEY C5's "
                             "1e-5 per problem is NOT met by pure fp32 arithmetic: cond(K) x 6e-8 on a direction cut by 5-19 halvings, and an Armijo test "
                             "below fp32 resolution")
        f4_state["parity"] = "per problem: one step from an fp32 state within 1e-7 of the fp64 oracle's step from the same state, every problem (tests/test_gpu_parity.py, tests/test_gpu_fullsize.py: all 1,048,576)"
        line["f4_fp32"] = {"workload": "BASELINE configs[4] (C5): F4 onedpath2_ip, fp32 state, %d problems x 50 fused steps" % count,
                           # the configuration's figure: the mode that meets the per-problem tolerance
                           "fused_50_steps_ms": f4_state["fused_50_steps_ms"], "newton_steps_per_s": f4_state["newton_steps_per_s"],
                           "mode_of_the_figure": "fp32_state_fp64_arithmetic (RP_DTYPE_F32_STATE): fp32 state and traffic, fp64 arithmetic in registers -- the parity-clean mode",
                           "fp32_state_fp64_arithmetic": f4_state,
                           "fp32_arithmetic": f4_pure,
                           "note": "50 fused F4 steps from the feasible start spend most of their time in the line search (10-19 "
                                   "feasibility halvings and up to 52 residual halvings per step from step ~6 on: F4 stalls, "
                                   "README.md:34), which is why a fused step costs several times a k = 1 step.  Round 6: problems whose step "
                                   "leaves their fp32 state bit for bit unchanged (the stuck ones: 2.4 % by step 48) sit the rest of a fused launch out -- exact"}

        # (d) the rows either side of the path: the plot data of a solved batch (66 positions + 4 accelerations per
        #     problem into device memory: 48 B of state and positions used, 560 B written) and the feasibility move
        #     (moveTowardFeasibility, the Space key) on the same number of starts pushed out of the feasible set
        with rp.Batch(count, rp.VARIANT_F3, rp.DTYPE_F64, device=local_rank, stream=stream) as c6:
            c6.set_problems_device(*ptrs)
            c6.solve(GAP_TOL, MAX_ITER, 0)
            d_plot = torch.empty((count, 66), dtype=torch.float64, device=torch.device("cuda", local_rank))
            d_acc = torch.empty((count, 4), dtype=torch.float64, device=torch.device("cuda", local_rank))
            ms = []
            for _ in range(5):
                c6.sync()
                c6.event_record(4)
                c6.sample_device(d_plot.data_ptr(), d_acc.data_ptr())
                c6.event_record(5)
                c6.sync()
                ms.append(c6.event_elapsed_ms(4, 5))
            t_s = min(ms[1:])
            del d_plot, d_acc
        b_sample = 6 * 8 + 4 + 70 * 8      # six fields (the end velocities are zero and not read) + the slot map word in, 70 doubles out
        n7 = count
        with rp.Batch(n7, rp.VARIANT_F3, rp.DTYPE_F64, device=local_rank, stream=stream) as c7:
            c7.set_problems_device(*ptrs)
            c7.restart()
            st = c7.get_state()
            st[:, 1] *= 0.7      # both durations too short: all four end accelerations beyond the limit
            st[:, 2] *= 0.7
            ms = []
            for _ in range(4):
                c7.set_state(st)
                c7.sync()
                c7.event_record(4)
                c7.move_toward_feasibility()
                c7.event_record(5)
                c7.sync()
                ms.append(c7.event_elapsed_ms(4, 5))
            t_f = min(ms[1:])
        def feas_roofline(ms_f, nprob):
            fl, fl_src = profile_number(PROFILE_TAG + "_sq_counters.json", "_flop_per_feasibility_move_4_rows")
            vi, _ = profile_number(PROFILE_TAG + "_sq_counters.json", "_valu_insts_per_feasibility_move_4_rows")
            if fl is None or not fresh:
                return None
            tf = fl * nprob / (ms_f * 1e-3) / 1e12
            return {"bound": "fp64_valu", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS,
                    "flop_per_problem": fl, "valu_instructions_per_problem": vi, "source": fl_src,
                    "valu_issue_frac": (vi * nprob / 64.0 / (ms_f * 1e-3) / 1e9 / (256 * 4 * 2.4 / 4)) if vi else None,
                    "note": "SQ counters of k_move_toward_feasibility on 1 Mi starts with four violated rows each (IEEE divisions and square roots "
                            "in Eigen's order: 11-instruction division sequences, few of them multiply-adds): the launch is issue-bound, "
                            "not traffic-bound"}

        line["neighbours"] = {
            "workload": "SURVEY 8f rows 1 and 3: plot data of %d solved problems into device memory (rp_batch_sample_device); feasibility move of %d infeasible "
                        "starts (k_feasibility_move: four violated rows each, the rank-deficient branch of the QR)" % (count, n7),
            "sample": {"ms": t_s, "problems_per_s": count / (t_s * 1e-3),
                       "roofline": {"bound": "hbm", "bytes_moved_per_problem": b_sample, "achieved": b_sample * count / (t_s * 1e-3) / 1e9,
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b_sample * count / (t_s * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "note": "output rows are in problem order, the state in scheduled order: a whole batch whose positions are the ones it was "
                                            "given goes through problem-order records (k_solution: 68 B per problem, then k_sample_records: 64 B read, 560 B "
                                            "written) instead of gathering six 32-byte sectors per problem; on those 692 B the two launches together are "
                                            "%.0f GB/s" % (692.0 * count / (t_s * 1e-3) / 1e9)}},
            "feasibility_move": {"ms": t_f, "problems_per_s": n7 / (t_f * 1e-3),
                                 "hbm_GBps_on_136_B_per_problem": 136.0 * n7 / (t_f * 1e-3) / 1e9,
                                 "roofline": feas_roofline(t_f, n7),
                                 "note": "16 fields read, 3 written per problem; the arithmetic (Gram matrix, Eigen-ordered 4 x 4 column-pivoted "
                                         "Householder QR in double precision, one problem per lane) is what the launch time is made of"}}

    # The headline workload with its batches dealt alternately onto TWO streams: the next launch's first (longest) chunks fill the wave slots
    # the previous launch's last chunks leave empty (wave slots stand empty 18 % of a launch at its two ends).  What a caller gains who keeps
    # two streams busy; wall clock (events of one stream do not span two), steady clocks, the one-stream figure measured the same way beside it.
    if not args.no_extras:
        try:
            m = max(2, min(n_batches, 40) // 2 * 2)
            first = batches[:m // 2]
            other = rp.Batch(count, rp.VARIANT_F3, rp.DTYPE_F64, device=local_rank)                       # its own stream
            second = [other] + [rp.Batch(count, rp.VARIANT_F3, rp.DTYPE_F64, device=local_rank, stream=other.stream()) for _ in range(m // 2 - 1)]
            onestream = [rp.Batch(count, rp.VARIANT_F3, rp.DTYPE_F64, device=local_rank, stream=stream) for _ in range(m // 2)]
            for b in second + onestream:
                b.set_problems_device(*ptrs)

            def burst(bs):
                for b in bs:
                    b.restart()
                condition()
                lead.sync()
                other.sync()
                t_b = time.perf_counter()
                for b in bs:
                    b.solve(GAP_TOL, MAX_ITER, 0)
                lead.sync()
                other.sync()
                return (time.perf_counter() - t_b) / len(bs) * 1e3
            ms_one = min(burst(first + onestream) for _ in range(2))
            ms_two = min(burst([x for pair in zip(first, second) for x in pair]) for _ in range(2))
            line["two_streams"] = {"launches": m, "ms_per_batch": ms_two, "newton_steps_per_s": steps_per_launch / (ms_two * 1e-3),
                                   "one_stream_same_method_ms_per_batch": ms_one, "gain": ms_one / ms_two,
                                   "note": "wall clock over %d launches after conditioning, host enqueue and the final synchronisation included in both "
                                           "figures; kernels of the two streams overlap, so per-kernel durations exceed the time per batch -- which is why "
                                           "the headline stays on one stream (profiles/r5_tuning.md)" % m}
            for b in second + onestream:
                b.close()
        except Exception as exc:      # an extra: never takes the benchmark down
            line["two_streams"] = {"error": str(exc)}

    # the same K launches from an IDLE chip (0.3 s of nothing first): what the timed region measured before it was conditioned (rounds 1-4's
    # `value`).  Always measured, and published as the stable top-level key `value_from_idle` so that rounds stay comparable.
    for j in range(min(K, n_batches)):
        batches[j].restart()
    lead.sync()
    time.sleep(0.3)
    lead.event_record(2)
    for j in range(min(K, n_batches)):
        batches[j].solve(GAP_TOL, MAX_ITER, 0)
    lead.event_record(3)
    lead.sync()
    cold_ms = lead.event_elapsed_ms(2, 3) / min(K, n_batches)
    line["value_from_idle"] = steps_per_launch * world / (cold_ms * 1e-3) if world == 1 else None      # (rank 0's own launches: a whole-job figure only at N = 1)
    line["cold_start"] = {"launches": min(K, n_batches), "ms_per_launch": cold_ms, "newton_steps_per_s": steps_per_launch / (cold_ms * 1e-3),
                          "note": "rank 0's K launches issued after 0.3 s of idling, HIP-event timed: inside the power controller's transient"}

    sustain_s = args.sustain_seconds if args.sustain_seconds is not None else (1.5 if (world == 1 and not args.no_extras) else 0.0)
    if sustain_s > 0:
        try:
            line["sustained"] = sustain(sustain_s)
        except Exception as exc:      # an extra: never takes the benchmark down
            line["sustained"] = {"error": str(exc)}

    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(min(count, 1 << 19))
    else:
        line["cpu_baseline"] = None

    print(json.dumps(line))
    sys.stdout.flush()
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
