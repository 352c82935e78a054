#!/bin/bash
# Round 6, GPU session 1 (repo root on the GPU box): the suite, the parking A/B (bit for bit + speed), a first bench line.
set -o pipefail
O=gpurun_out/r6s1
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; echo "suite rc $?" | tee -a $O/gpu_suite.log
tail -3 $O/gpu_suite.log
timeout -k 10 600 python tests/checks/fixed_step_ab.py $O/ab_park.npz > $O/ab.log 2>&1 && \
RP_BATCH_LIB=$PWD/profiles/probes/librp_batch_nopark.so timeout -k 10 600 python tests/checks/fixed_step_ab.py $O/ab_nopark.npz >> $O/ab.log 2>&1 && \
python tests/checks/inplace_ab.py cmp $O/ab_park.npz $O/ab_nopark.npz >> $O/ab.log 2>&1; echo "ab rc $?" >> $O/ab.log; tail -3 $O/ab.log
rm -f $O/ab_park.npz $O/ab_nopark.npz
for r in 1 2; do
  timeout -k 10 300 python profiles/probes/f4_speed_probe.py >> $O/f4_speed.log 2>&1
  RP_BATCH_LIB=$PWD/profiles/probes/librp_batch_nopark.so timeout -k 10 300 python profiles/probes/f4_speed_probe.py >> $O/f4_speed.log 2>&1
done
grep "k = 50" $O/f4_speed.log
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
l = json.loads([x for x in open("gpurun_out/r6s1/bench.json") if x.startswith("{")][-1])
print("value %.2f G  from idle %.2f G  ms/step %.4f" % (l["value"] / 1e9, (l.get("value_from_idle") or 0) / 1e9, l["ms_per_step"]))
e = l["end_to_end"]; print("end_to_end %.2f G (pipeline: %s)  one stream by hand %.2f G" % (e["newton_steps_per_s"] / 1e9, {k: e["pipeline"].get(k) for k in ("ms_per_batch", "gain_over_one_stream", "solutions_bitwise_equal_to_the_one_stream_path", "ratio_to_the_headline_value", "error")}, e["one_stream_by_hand"]["newton_steps_per_s"] / 1e9))
f = l.get("f4_fp32", {}); print("f4 fp32-state %.2f G  pure %.2f G" % (f.get("newton_steps_per_s", 0) / 1e9, f.get("fp32_arithmetic", {}).get("newton_steps_per_s", 0) / 1e9))
print("fixed50", l.get("fixed50", {}).get("ms"), "two_streams", l.get("two_streams", {}).get("newton_steps_per_s"), "sustained", l.get("sustained", {}).get("steady_newton_steps_per_s"))
PY
