#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r6s6
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -3 $O/gpu_suite.log
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
l = json.loads([x for x in open("gpurun_out/r6s6/bench.json") if x.startswith("{")][-1])
print("value %.2f G  from idle %.2f G  ms/step %.4f" % (l["value"] / 1e9, (l.get("value_from_idle") or 0) / 1e9, l["ms_per_step"]))
e = l["end_to_end"]; print("end_to_end %.2f G (pipeline: %s)  one stream by hand %.2f G" % (e["newton_steps_per_s"] / 1e9, {k: e["pipeline"].get(k) for k in ("ms_per_batch", "gain_over_one_stream", "solutions_bitwise_equal_to_the_one_stream_path", "ratio_to_the_headline_value", "error")}, e["one_stream_by_hand"]["newton_steps_per_s"] / 1e9))
f = l.get("f4_fp32", {}); print("f4 fp32-state %.2f G  pure %.2f G" % (f.get("newton_steps_per_s", 0) / 1e9, f.get("fp32_arithmetic", {}).get("newton_steps_per_s", 0) / 1e9))
print("fixed50", l.get("fixed50", {}).get("ms"), "two_streams", l.get("two_streams", {}).get("newton_steps_per_s"), "sustained", l.get("sustained", {}).get("steady_newton_steps_per_s"))
PY
