#!/usr/bin/env python3
"""Fold rocprofv3 CSV output (gpurun_out/...) into the small summaries committed under profiles/.

    python profiles/summarize.py <tag> <kernel-stats dir> <FETCH_SIZE dir> <WRITE_SIZE dir> [<SQ dir> ...]

kernel-stats / FETCH_SIZE / WRITE_SIZE come from passes over `python3 bench.py` (the driver's default command resp. a
short run of it, see the _method strings); the SQ directories from passes over `python3 profiles/pmc_probe.py`.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
sq_dirs = sys.argv[5:]
N = 1 << 20
PROBE_STEPS = 12      # profiles/pmc_probe.py: F3
PROBE_STEPS_F4 = 50   # ... F4 (all 50 steps of BASELINE configs[4])


def one(d, pat):
    """The newest match: gpurun merges a call's files into gpurun_out/ without removing an earlier call's."""
    return max(glob.glob(os.path.join(d, "*", pat)), key=os.path.getmtime)


if stats_dir != "-":      # "-": the counter summaries only (collect_r4.sh folds them before the kernel-trace pass, which needs them)
    shutil.copy(one(stats_dir, "*_kernel_stats.csv"), os.path.join(ROOT, "profiles", "%s_bench_kernel_stats.csv" % tag))
    # The timed launches themselves, out of the same trace.  The per-kernel statistics above average EVERY launch of a kernel -- for the
    # benchmark's kernel that is the conditioning (which starts inside the power controller's transient), the warmup, the K timed
    # launches, the cold-start extra ...; the K timed ones are the launches of k_solve_chunks<double, double, 3, false, true, 0, false, false> right
    # before the FIRST k_reduce_partial of the run (the final reduction follows them in stream order).
    try:
        rows = sorted(csv.DictReader(open(one(stats_dir, "*_kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
        TIMED = "k_solve_chunks<double, double, 3, false, true, 0, false, false>"      # <S, T, VARIANT, STALL, ZV, MU, START, ROUNDS>: the plain gated kernel
        first_reduce = next(i for i, r in enumerate(rows) if "k_reduce_partial" in r["Kernel_Name"])
        before = [r for r in rows[:first_reduce] if TIMED in r["Kernel_Name"] and int(r["Grid_Size_X"]) == N]
        bench_line = None
        try:
            bench_line = json.loads([l for l in open(os.path.join(os.path.dirname(os.path.normpath(stats_dir)), "bench_under_rocprof.json")) if l.startswith("{")][-1])
        except Exception:
            pass
        K = int(bench_line["steps"]) if bench_line else 20
        W = int(bench_line["warmup"]) if bench_line else 5
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in before]
        timed, warm, cond = dur[-K:], dur[-(K + W):-K], dur[:-(K + W)]
        every = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if TIMED in r["Kernel_Name"] and int(r["Grid_Size_X"]) == N]
        json.dump({"_what": "the K timed launches of `python3 bench.py --gpus 1 --steps %d --warmup %d` under rocprofv3 --kernel-trace, told apart from the other "
                            "launches of the same kernel by their place in the trace (the K right before the first k_reduce_partial)" % (K, W),
                   "kernel": TIMED, "timed_launch_ns": timed, "timed_avg_ns": sum(timed) / max(len(timed), 1),
                   "warmup_avg_ns": (sum(warm) / len(warm)) if warm else None,
                   "conditioning_launches": len(cond), "conditioning_first_20_avg_ns": (sum(cond[:20]) / len(cond[:20])) if cond else None,
                   "conditioning_last_20_avg_ns": (sum(cond[-20:]) / len(cond[-20:])) if cond else None,
                   "all_launches_of_this_kernel": len(every), "all_launches_avg_ns": sum(every) / max(len(every), 1),
                   "bench_line_avg_launch_ms_same_run": (bench_line or {}).get("roofline", {}).get("avg_launch_ms")},
                  open(os.path.join(ROOT, "profiles", "%s_timed_launches.json" % tag), "w"), indent=1)
    except Exception as exc:      # an older trace layout: the statistics file stands alone
        print("timed launches not extracted:", exc)


def counters(d, grid=None):
    """Mean counter values per kernel; grid: only dispatches of that many work-items, else all."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(one(d, "*_counter_collection.csv"))):
        if grid is not None and int(r["Grid_Size"]) != grid:
            continue
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


def short(k):
    k = k.replace("void rp::(anonymous namespace)::", "").replace("rp::(anonymous namespace)::", "")
    return k.split("(")[0].strip()


fetch, write = counters(fetch_dir), counters(write_dir)
out = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --no-cpu-baseline "
                  "--steps 4 --warmup 1` (extras included: the k = 1 launches, the k = 0 probe, configs[1] and configs[4]); "
                  "counters are KiB per dispatch, mean over the dispatches of a kernel; FETCH_SIZE is doubled as "
                  "MI355X_MICROARCH.md prescribes for gfx950 (128-B requests tallied at 64 B).  Calibration in the same passes, "
                  "on launches whose bytes are known exactly: k_reduce_partial reads (16 fields x 8 B + 4 B status) x 1,048,576 "
                  "= 138,412,032 B (8 B per lane); the k = 0 launches of k_newton_stream16 read 14 and write 11 fields "
                  "(16 B per lane): 117,440,512 + 92,274,688 B -- they are part of that kernel's mean"}
for k in fetch:
    name = short(k)
    if not name.startswith("k_"):
        continue
    rd = fetch[k].get("FETCH_SIZE", 0.0) * 1024 * 2
    wr = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
    out[name] = {"FETCH_SIZE_KiB": fetch[k].get("FETCH_SIZE"), "WRITE_SIZE_KiB": write.get(k, {}).get("WRITE_SIZE"),
                 "read_bytes_corrected": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr}


def alias(key, prefix, expected):
    hit = [k for k in out if k.startswith(prefix)]
    if hit:
        out[key] = dict(out[hit[0]], kernel=hit[0], expected=expected)


alias("k_solve_chunks_f3_f64", "k_solve_chunks<double, double, 3, false, true, 0, false, false>",
      "zero-end-velocity instantiation: (14x8 + 4 + 4) B read + (11x8 + 4 + 4) B written per problem = 125.8 + 100.7 MB at n = 1,048,576")
alias("k_newton_stream16_f3_f64", "k_newton_stream16<double, double, 3, true", "14x8 B read + 11x8 B written per problem = 117.4 + 92.3 MB")
alias("k_newton_stream16_f4_f32", "k_newton_stream16<float, float, 4, true", "10x4 B read + 7x4 B written per problem = 41.9 + 29.4 MB")
alias("k_newton_stream16_f4_f32state", "k_newton_stream16<float, double, 4, true", "10x4 B read + 7x4 B written per problem = 41.9 + 29.4 MB")
alias("k_steps_chunks_f4_f32", "k_steps_chunks<float, float, 4, true", "(10x4) B read + 7x4 B written per problem = 41.9 + 29.4 MB per launch")
alias("k_steps_chunks_f4_f32state", "k_steps_chunks<float, double, 4, true", "(10x4) B read + 7x4 B written per problem = 41.9 + 29.4 MB per launch")
alias("k_steps_chunks_f3_f64", "k_steps_chunks<double, double, 3, true", "14x8 B read + 11x8 B written per problem = 117.4 + 92.3 MB per launch")
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_hbm_traffic.json" % tag), "w"), indent=1)

def real_grid(name):
    """Work-items of the 1 Mi-problem launch of a Newton kernel of the probe: one lane per problem in the chunk kernels, two
    problems per lane (16-byte accesses) in the streaming kernel."""
    return N if name.startswith(("k_solve_chunks", "k_steps_chunks", "k_move_toward")) else N // 2


sq = {}
ident = {}
for d in sq_dirs:
    for grid in (N, N // 2):
        for k, cs in counters(d, grid=grid).items():
            name = short(k)
            if name.startswith(("k_newton", "k_solve", "k_steps", "k_move_toward")) and grid == real_grid(name):
                sq.setdefault(name, {}).update(cs)
            elif name.startswith("k_solve_chunks<double, double, 3, false, true, 0, false, false>") and grid == N // 2:
                ident.update(cs)      # the gated kernel on 524,288 identical default problems (pmc_probe.py)
for name, c in sq.items():
    if "SQ_WAVES" in c and "SQ_INSTS_VALU" in c:
        c["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        c["valu_busy_per_wave_cycle"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
# flop per lane-step from the ungated 12-step launches of pmc_probe.py (1 Mi problems, every lane active):
# SQ_INSTS_VALU_* count wave-instructions; x 64 lanes / (12 * 2^20 lane-steps)
top = {}
for name, c in sq.items():
    # k_steps_chunks<S, T, VARIANT, ZV = true, REGBK = false>: the large-batch instantiations for zero end velocities
    if not name.startswith("k_steps_chunks") or not name.endswith(", true, false>"):
        continue
    lane_steps = float(PROBE_STEPS_F4 if ", 4, true, false>" in name else PROBE_STEPS) * N
    f64 = 64.0 * (2 * c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0)
                  + c.get("SQ_INSTS_VALU_TRANS_F64", 0)) / lane_steps
    f32 = 64.0 * (2 * c.get("SQ_INSTS_VALU_FMA_F32", 0) + c.get("SQ_INSTS_VALU_MUL_F32", 0) + c.get("SQ_INSTS_VALU_ADD_F32", 0)
                  + c.get("SQ_INSTS_VALU_TRANS_F32", 0)) / lane_steps
    c["flop_f64_per_lane_step"], c["flop_f32_per_lane_step"] = f64, f32
    if "SQ_INSTS_VALU" in c:
        c["valu_insts_per_lane_step"] = 64.0 * c["SQ_INSTS_VALU"] / lane_steps
    if name.startswith("k_steps_chunks<double, double, 3"):
        top["_flop_per_newton_step"] = f64
    elif name.startswith("k_steps_chunks<float, float, 4"):
        top["_flop_per_f4_step_f32"] = f32 + f64
    elif name.startswith("k_steps_chunks<float, double, 4"):
        top["_flop_per_f4_step_f32state"] = f64
# the gated kernel on identical problems: 15 steps per problem, no idle lanes
GATED = "k_solve_chunks<double, double, 3, false, true, 0, false, false>"
if "SQ_INSTS_VALU_FMA_F64" in ident:
    ls = 15.0 * (N // 2)
    ident["flop_f64_per_lane_step"] = 64.0 * (2 * ident["SQ_INSTS_VALU_FMA_F64"] + ident["SQ_INSTS_VALU_MUL_F64"] + ident["SQ_INSTS_VALU_ADD_F64"]
                                             + ident["SQ_INSTS_VALU_TRANS_F64"]) / ls
    if "SQ_INSTS_VALU" in ident:
        ident["valu_insts_per_lane_step"] = 64.0 * ident["SQ_INSTS_VALU"] / ls
    sq[GATED + " on identical problems"] = ident
    top["_flop_per_gated_newton_step"] = ident["flop_f64_per_lane_step"]
gated_real = sq.get(GATED, {})
if "SQ_INSTS_VALU" in gated_real:
    top["_valu_wave_insts_per_gated_launch"] = gated_real["SQ_INSTS_VALU"]      # the benchmark's launch itself (idle lanes included)
for name, c in sq.items():      # the feasibility move: one launch over 1 Mi starts with four violated rows each (pmc_probe.py)
    if name.startswith("k_move_toward_feasibility<double, double, 3") and "SQ_INSTS_VALU_FMA_F64" in c:
        c["flop_f64_per_problem"] = 64.0 * (2 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_ADD_F64"]
                                           + c["SQ_INSTS_VALU_TRANS_F64"]) / N
        top["_flop_per_feasibility_move_4_rows"] = c["flop_f64_per_problem"]
        if "SQ_INSTS_VALU" in c:
            c["valu_insts_per_problem"] = 64.0 * c["SQ_INSTS_VALU"] / N
            top["_valu_insts_per_feasibility_move_4_rows"] = c["valu_insts_per_problem"]
# BASELINE configs[1] (pmc_probe.py): 65,536 problems x 50 steps = grid 65,536 of the register-column chunk kernel; its first 15 steps on
# 65,472 problems = grid 65,472.  Per launch (not per lane-step: the post-convergence steps are nothing like the first fifteen).
SMALL = "k_steps_chunks<double, double, 3, true, true>"
f50, f15 = {}, {}
for d in sq_dirs:
    for k, cs in counters(d, grid=65536).items():
        if short(k) == SMALL:
            f50.update(cs)
    for k, cs in counters(d, grid=65536 - 64).items():
        if short(k) == SMALL:
            f15.update(cs)


def launch_flop(c):
    return 64.0 * (2 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_TRANS_F64"])


if "SQ_INSTS_VALU_FMA_F64" in f50 and "SQ_INSTS_VALU" in f50:
    sq[SMALL + " configs[1]: 65,536 x 50"] = f50
    top["_flop_per_fixed50_launch"] = launch_flop(f50)
    top["_valu_wave_insts_per_fixed50_launch"] = f50["SQ_INSTS_VALU"]
    top["_wave_insts_per_fixed50_launch"] = f50["SQ_INSTS_VALU"] + f50.get("SQ_INSTS_SALU", 0.0)
    if "SQ_INSTS_VALU" in f15 and "SQ_WAVES" in f15:
        sq[SMALL + " 65,472 x 15"] = f15
        w50, w15 = f50["SQ_WAVES"], f15["SQ_WAVES"]
        per = lambda key: ((f50[key] / w50 - f15[key] / w15) / 35.0, f15[key] / w15 / 15.0)      # noqa: E731
        split = {}
        for key, label in (("SQ_INSTS_VALU", "valu"), ("SQ_INSTS_SALU", "salu"), ("SQ_WAVE_CYCLES", "wave_quad_cycles"), ("SQ_ACTIVE_INST_ANY", "issuing_quad_cycles"),
                           ("SQ_WAIT_ANY", "parked_quad_cycles"), ("SQ_WAIT_INST_ANY", "stalled_quad_cycles")):
            if key in f50 and key in f15:
                late, early = per(key)
                split[label + "_per_wave_step_steps_16_to_50"] = late
                split[label + "_per_wave_step_steps_1_to_15"] = early
        sq["_fixed50_per_wave_step"] = split
sq.update(top)
json.dump({"_method": "rocprofv3 --pmc <SQ counters, <= 8 per pass> -- python3 profiles/pmc_probe.py (1 Mi problems: 12 fused ungated steps "
                      "of F3 f64, 50 of F4 f32 / F4 f32-state, then the fused gated F3 solve and one k = 1 launch); SQ_WAVE_CYCLES / "
                      "SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles; flop = 2 FMA + MUL + ADD + TRANS wave-instructions x 64 lanes "
                      "/ (12 x 2^20 lane-steps; F4: 50 x 2^20, all 50 steps of BASELINE configs[4], executed flop of the whole wave-parallel "
                      "line search included)", **sq},
          open(os.path.join(ROOT, "profiles", "%s_sq_counters.json" % tag), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("k_") or "<" not in k}, indent=1)[:3000])
print(json.dumps(top, indent=1))

# which sources these numbers belong to: bench.py withholds every counter-derived fraction once the kernel sources differ
import hashlib  # noqa: E402
import subprocess  # noqa: E402

KERNEL_SOURCES = ("rocket_path_amd/csrc/ip_core.h", "rocket_path_amd/csrc/ip_kernels.hip", "rocket_path_amd/csrc/feas_core.h",
                  "rocket_path_amd/csrc/ip_kernels.h", "rocket_path_amd/csrc/rp_batch.cpp", "rocket_path_amd/csrc/schedule.hip")      # = bench.py's list
box_hashes = {}      # sha256sum output the collection script wrote next to the counter directories, on the box that ran the kernels
try:
    for line in open(os.path.join(os.path.dirname(os.path.normpath(sq_dirs[0])), "sources.sha256")):
        h, f = line.split()
        if f in KERNEL_SOURCES:
            box_hashes[f] = h[:16]
    if len(box_hashes) != len(KERNEL_SOURCES):
        box_hashes = {}
except Exception:
    box_hashes = {}
# The commit the kernel sources belong to: taken in the build container BEFORE the collection is sent to the GPU box (which has no .git)
# and handed in through the environment by the collection script (profiles/collect_r5.sh: RP_COLLECT_COMMIT / RP_COLLECT_DIRTY); where the
# summaries are folded inside a checkout, from git itself.
commit, dirty = os.environ.get("RP_COLLECT_COMMIT") or None, os.environ.get("RP_COLLECT_DIRTY")
commit_from = "RP_COLLECT_COMMIT: `git rev-parse HEAD` in the build container before the collection ran" if commit else None
dirty = None if dirty is None else dirty not in ("", "0", "false")
if commit is None:
    try:
        commit = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or None
        dirty = bool(subprocess.run(["git", "status", "--porcelain", "--"] + list(KERNEL_SOURCES), cwd=ROOT, capture_output=True, text=True).stdout.strip())
        commit_from = "git, where the summaries were folded" if commit else None
    except Exception:
        commit, dirty = None, None
json.dump({"_what": "kernel sources the %s_* counter summaries were collected from (the .so that ran on the GPU box was built from these files); "
                    "bench.py compares the hashes with the files it finds" % tag,
           "commit": commit, "commit_from": commit_from, "kernel_sources_uncommitted_at_collection": dirty,
           "sha256_16": box_hashes or {f: hashlib.sha256(open(os.path.join(ROOT, f), "rb").read()).hexdigest()[:16] for f in KERNEL_SOURCES},
           "hashes_taken": "on the GPU box by the collection script" if box_hashes else "where the summaries were folded (no sources.sha256 in the collection)"},
          open(os.path.join(ROOT, "profiles", "%s_sources.json" % tag), "w"), indent=1)
