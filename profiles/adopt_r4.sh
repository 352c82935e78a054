#!/bin/bash
# Build container, repo root, after the GPU collection came back into gpurun_out/: copy the folded summaries and the probe logs into
# profiles/ and stamp the commit the kernel sources belong to.
set -e
O=gpurun_out/r4prof; P=gpurun_out/r4probes
cp $O/r4_sq_counters.json $O/r4_hbm_traffic.json $O/r4_sources.json $O/r4_bench_kernel_stats.csv profiles/
cp $O/bench.json profiles/r4_bench.json
cp $O/bench_under_rocprof.json profiles/r4_bench_under_rocprof.json
cp $O/bench_pipelined.json profiles/r4_bench_pipelined.json
cp $O/bench_5runs.jsonl profiles/r4_bench_5runs.jsonl
for f in gpu_suite valu_diet speed_probe frozen_proof_ab moving_stats k1_ab solution_probe f4_speed fuzz_parity; do cp $P/$f.log profiles/r4_$f.log; done
python3 - <<'PY'
import hashlib, json, subprocess
p = "profiles/r4_sources.json"
d = json.load(open(p))
now = {f: hashlib.sha256(open(f, "rb").read()).hexdigest()[:16] for f in d["sha256_16"]}
assert now == d["sha256_16"], "the kernel sources changed since the collection ran"
d["commit"] = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
d["commit_note"] = "HEAD of the build container when the collection was adopted (the GPU box has no .git); the kernel sources of that commit hash to the values below"
json.dump(d, open(p, "w"), indent=1)
print("adopted; kernel sources match")
PY
