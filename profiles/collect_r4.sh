#!/bin/bash
# Round-4 profile collection (GPU box, from the repo root):  bash profiles/collect_r4.sh
# (gpurun merges outputs into the local gpurun_out/ without deleting an earlier call's files: profiles/summarize.py takes the newest.)
# rocprofv3 gets the program directly after `--` (no env/bash hop); counters and traces in separate passes.
set -e
export TMPDIR=/tmp
O=gpurun_out/r4prof
mkdir -p $O && rm -rf $O/stats $O/fetch $O/write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 > $O/fetch.out 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 > $O/write.out 2> $O/write.err
# (the SQ counter passes over profiles/pmc_probe.py are profiles/collect_r4_counters.sh: they come first after a kernel change)
python3 bench.py > $O/bench.json 2> $O/bench.err
for i in 1 2 3 4 5; do python3 bench.py --no-cpu-baseline >> $O/bench_5runs_new.jsonl 2>> $O/bench.err; done
mv $O/bench_5runs_new.jsonl $O/bench_5runs.jsonl
python3 bench.py --no-cpu-baseline --pipelined > $O/bench_pipelined.json 2>> $O/bench.err
echo collected
