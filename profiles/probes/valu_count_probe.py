#!/usr/bin/env python3
"""Tuning loop for the gated kernel's instruction count.
    python3 profiles/probes/valu_count_probe.py run        the launches the counters are read from (under rocprofv3 --pmc ...)
    python3 profiles/probes/valu_count_probe.py time       HIP-event timings of the same launches
    python3 profiles/probes/valu_count_probe.py read DIR.. per lane-step counts from rocprofv3's counter_collection.csv files
The gated kernel on 524,288 identical default problems executes exactly 15 steps on every lane (no idle lanes): its SQ_INSTS_VALU x 64
/ (15 x 524,288) is the instructions per Newton step; the 1 Mi random batch (16,308,345 steps) gives the launch the benchmark times."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
N = 1 << 20
mode = sys.argv[1] if len(sys.argv) > 1 else "run"
if mode == "read":
    agg = {}
    for d in sys.argv[2:]:
        for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if "k_solve_chunks" not in r["Kernel_Name"]:
                    continue
                agg.setdefault(int(r["Grid_Size"]), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    agg[int(r["Grid_Size"])].setdefault("_ns", []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for grid, steps, tag in ((N // 2, 15.0 * (N // 2), "identical problems"), (N, 16308345.0, "benchmark batch")):
        c = {k: sum(v) / len(v) for k, v in agg.get(grid, {}).items()}
        if not c:
            continue
        per = lambda k: 64.0 * c.get(k, 0.0) / steps      # noqa: E731
        fma, mul, add, tr = per("SQ_INSTS_VALU_FMA_F64"), per("SQ_INSTS_VALU_MUL_F64"), per("SQ_INSTS_VALU_ADD_F64"), per("SQ_INSTS_VALU_TRANS_F64")
        print("%-20s VALU/step %.1f  fma %.1f mul %.1f add %.1f trans %.1f  other %.1f  flop/step %.1f  flop/VALU %.2f  SALU/step %.1f" % (
            tag, per("SQ_INSTS_VALU"), fma, mul, add, tr, per("SQ_INSTS_VALU") - fma - mul - add - tr, 2 * fma + mul + add + tr,
            (2 * fma + mul + add + tr) / max(per("SQ_INSTS_VALU"), 1e-9), per("SQ_INSTS_SALU")))
        if "_ns" in c:
            # GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_ACTIVE_INST_VALU counts quad-cycles over all SIMDs (1,024); SQ_BUSY_CYCLES sums the 32 shader engines
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            print("%-20s under the counters: %.1f us, %.0f cycles per XCD = %.2f GHz; vector ALUs busy %.3f of the launch; SQ busy %.3f; LDS instructions active %.4f, waited on %.4f (of wave-cycles)" % (
                tag, c["_ns"] / 1e3, cyc, cyc / c["_ns"], 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (1024.0 * cyc), c.get("SQ_BUSY_CYCLES", 0.0) / (32.0 * cyc),
                c.get("SQ_ACTIVE_INST_LDS", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0), c.get("SQ_WAIT_INST_LDS", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0)))
    sys.exit(0)

import rocket_path_amd as rp  # noqa: E402

p0, p1, p2 = rp.problems.generate(12345, 0, N, 0)
if mode == "run":
    with rp.Batch(N // 2) as b:
        for _ in range(2):
            b.init_default()
            b.solve(1e-8, 200, 0)
            b.sync()
        assert int(b.reduce()["total_steps"]) == 15 * (N // 2)
    with rp.Batch(N) as b:
        for _ in range(2):
            b.set_problems(p0, p1, p2)
            b.restart()
            b.solve(1e-8, 200, 0)
            b.sync()
        print("steps of the benchmark batch:", int(b.reduce()["total_steps"]))
else:
    lead = rp.Batch(N)
    bs = [lead] + [rp.Batch(N, stream=lead.stream()) for _ in range(11)]
    for rep in range(3):
        for b in bs:
            b.set_problems(p0, p1, p2)
            b.restart()
        lead.sync()
        ms = []
        for b in bs:
            b.event_record(0)
            b.solve(1e-8, 200, 0)
            b.event_record(1)
            b.sync()
            ms.append(b.event_elapsed_ms(0, 1))
        for b in bs:
            b.restart()
        lead.sync()
        lead.event_record(4)
        for b in bs:
            b.solve(1e-8, 200, 0)
        lead.event_record(5)
        lead.sync()
        steps = lead.reduce()["total_steps"]
        print("gated 1 Mi (%d steps): one at a time best %.4f med %.4f ms; 12 back to back %.4f ms each = %.2f G steps/s" % (
            steps, min(ms), sorted(ms)[len(ms) // 2], lead.event_elapsed_ms(4, 5) / len(bs), steps / (lead.event_elapsed_ms(4, 5) / len(bs)) / 1e6))
