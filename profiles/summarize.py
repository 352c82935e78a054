#!/usr/bin/env python3
"""Fold rocprofv3 CSV output (gpurun_out/...) into the small summaries committed under profiles/.

    python profiles/summarize.py <tag> <kernel-stats dir> <FETCH_SIZE dir> <WRITE_SIZE dir> [<SQ dir> ...]
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


def one(d, pat):
    return glob.glob(os.path.join(d, "*", pat))[0]


shutil.copy(one(stats_dir, "*_kernel_stats.csv"), os.path.join(ROOT, "profiles", "%s_bench_kernel_stats.csv" % tag))


def counters(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(one(d, "*_counter_collection.csv"))):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


def short(k):
    k = k.replace("void rp::(anonymous namespace)::", "").replace("rp::(anonymous namespace)::", "")
    return k.split("(")[0].strip()


fetch, write = counters(fetch_dir), counters(write_dir)
out = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --no-cpu-baseline "
                  "--no-extras --steps 4 --warmup 1`; counters are KiB per dispatch (mean over dispatches); FETCH_SIZE is doubled "
                  "as MI355X_MICROARCH.md prescribes for gfx950 (128-B requests tallied at 64 B) -- calibrated here on "
                  "k_reduce_partial, whose read volume is known exactly: (16 fields x 8 B + 4 B status) x 1,048,576 = 138,412,032 B"}
for k in fetch:
    name = short(k)
    if not name.startswith("k_"):
        continue
    rd = fetch[k].get("FETCH_SIZE", 0.0) * 1024 * 2
    wr = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
    out[name] = {"FETCH_SIZE_KiB": fetch[k].get("FETCH_SIZE"), "WRITE_SIZE_KiB": write.get(k, {}).get("WRITE_SIZE"),
                 "read_bytes_corrected": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr}
gated = [k for k in out if k.startswith("k_solve_tiled<double, 3, true")]
if gated:
    out["k_solve_tiled_f3_f64"] = dict(out[gated[0]], kernel=gated[0],
                                       expected="zero-end-velocity instantiation: (14x8 + 4 + 4 + 2) B read + (11x8 + 4 + 4) B written "
                                                "per problem = 127.9 + 100.7 MB at n = 1,048,576")
json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)       # read by bench.py
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_hbm_traffic.json" % tag), "w"), indent=1)

sq = {}
for d in sq_dirs:
    for k, cs in counters(d).items():
        name = short(k)
        if name.startswith(("k_newton", "k_solve")):
            sq.setdefault(name, {}).update(cs)
for name, c in sq.items():
    if "SQ_WAVES" in c and "SQ_INSTS_VALU" in c:
        c["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        c["valu_busy_per_wave_cycle"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
# flop per lane-step from the ungated 12-step launch of pmc_probe.py (1 Mi problems x 12 steps, every lane active):
# SQ_INSTS_VALU_* count wave-instructions; x 64 lanes / (12 * 2^20 lane-steps)
flop = None
for name, c in sq.items():
    if name.startswith("k_solve_tiled<double, 3, false") and "SQ_INSTS_VALU_FMA_F64" in c:
        lane_steps = 12.0 * (1 << 20)
        c["flop_per_lane_step"] = 64.0 * (2 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_ADD_F64"]
                                          + c["SQ_INSTS_VALU_TRANS_F64"]) / lane_steps
        c["valu_insts_per_wave_step"] = 64.0 * c["SQ_INSTS_VALU"] / lane_steps
        flop = c["flop_per_lane_step"]
if flop is not None:
    sq["_flop_per_newton_step"] = flop
json.dump({"_method": "rocprofv3 --pmc <SQ counters> -- python3 profiles/pmc_probe.py (1 Mi problems: 12 fused ungated steps, "
                      "then one fused gated solve); SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles", **sq},
          open(os.path.join(ROOT, "profiles", "%s_sq_counters.json" % tag), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k.startswith("k_solve") or k.startswith("k_newton")}, indent=1)[:1500])
print(json.dumps(sq, indent=1)[:2500])
