#!/bin/bash
# A/B of the one-wave scheduling kernels: rank by LDS match (in-tree) against the ballot form (librp_batch_sched_ballots.so), same box
set -o pipefail
O=gpurun_out/s17; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_boundary.py -q -m gpu -k "sched or pipeline or order" -x > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for lib in - profiles/probes/librp_batch_sched_ballots.so - profiles/probes/librp_batch_sched_ballots.so; do
  if [ "$lib" = "-" ]; then unset RP_BATCH_LIB; else export RP_BATCH_LIB=$PWD/$lib; fi
  echo "== library: ${RP_BATCH_LIB:-in-tree}" >> $O/probe.log
  timeout -k 10 200 python3 profiles/probes/pipeline_probe.py >> $O/probe.log 2>&1 || exit 1
done
unset RP_BATCH_LIB
grep "==\|best\|round 2" $O/probe.log
