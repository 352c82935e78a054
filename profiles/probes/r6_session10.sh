#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r6s10
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f4" > $O/f4_tests.log 2>&1; echo "f4 tests rc $?"; tail -3 $O/f4_tests.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/pmc -- python3 profiles/probes/f4_pmc_valu.py > $O/pmc.out 2>&1
python3 - <<'PY'
import csv, glob, collections
f = max(glob.glob("gpurun_out/r6s10/pmc/**/*_counter_collection.csv", recursive=True))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "k_steps_chunks" in r["Kernel_Name"]: agg[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: sum(x) / len(x) * 64 / (50 * (1 << 20)) for c, x in v.items() if c != "SQ_WAVES"})
PY
for r in 1 2; do timeout -k 10 300 python profiles/probes/f4_speed_probe.py 2>&1 | grep "k = 50"; done
find $O -name "*.csv" -size +1M -delete
