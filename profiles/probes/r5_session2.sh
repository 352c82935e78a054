#!/bin/bash
# Round 5, second GPU session: the new tests with their printed lines, the decision-margin calibration, the bench line with the new keys.
export TMPDIR=/tmp
O=gpurun_out/r5b
mkdir -p $O
python3 tests/checks/decision_margins.py > $O/decision_margins.log 2>&1 || { tail -20 $O/decision_margins.log; exit 1; }
cat $O/decision_margins.log
python3 -m pytest tests -m gpu -q -s -k "certified or decisions or long_sequence or pure_fp32 or bound_late or raw_pointer or rehearsal or forced_rccl" > $O/pytest_new.log 2>&1; echo "pytest new rc $?"
grep -E "passed|failed|F4 |F3 |pure fp32|Error|error|assert" $O/pytest_new.log | head -60
python3 bench.py --sustain-seconds 3 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r5b/bench.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"])
print("breakdown", d["timed_region_breakdown"])
print("devices", d["devices"], d["devices_distinct"])
print("fixed50", d["fixed50"])
print("sustained", json.dumps(d["sustained"])[:1500])
PY
