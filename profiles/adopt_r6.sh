#!/bin/bash
# Build container, repo root, after `gpurun -- 'RP_COLLECT_COMMIT=... bash profiles/collect_r6.sh'` came back into gpurun_out/: copy the folded
# summaries into profiles/.  Nothing is stamped here: the commit was handed to the collection (RP_COLLECT_COMMIT) and the hashes were
# taken on the GPU box; this script only refuses to adopt summaries whose kernel sources are not the ones in the tree.
set -e
O=gpurun_out/r6prof
python3 - <<'PY'
import hashlib, json
d = json.load(open("gpurun_out/r6prof/r6_sources.json"))
now = {f: hashlib.sha256(open(f, "rb").read()).hexdigest()[:16] for f in d["sha256_16"]}
assert now == d["sha256_16"], "the kernel sources changed since the collection ran: %s" % [f for f in now if now[f] != d["sha256_16"][f]]
print("kernel sources match the collection (commit %s)" % d["commit"])
PY
cp $O/r6_sq_counters.json $O/r6_hbm_traffic.json $O/r6_sources.json $O/r6_bench_kernel_stats.csv $O/r6_timed_launches.json profiles/
cp $O/bench.json profiles/r6_bench.json
cp $O/bench_under_rocprof.json profiles/r6_bench_under_rocprof.json
cp $O/bench_5runs.jsonl profiles/r6_bench_5runs.jsonl
cp $O/bench_forced_rccl.json profiles/r6_bench_forced_rccl.json

cp $O/pipeline_overlap.log profiles/r6_pipeline_overlap.log
cp $O/pipeline_overlap_one_stream.log profiles/r6_pipeline_overlap_one_stream.log
echo "adopted (round 6 extras)"
