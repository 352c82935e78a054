#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r6s4
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_boundary.py -m gpu -x -q > $O/boundary.log 2>&1; echo "boundary rc $?"; tail -3 $O/boundary.log
rm -f $O/sched_time.log
timeout -k 10 120 python profiles/probes/sched_time.py >> $O/sched_time.log 2>&1
cat $O/sched_time.log
timeout -k 10 600 python profiles/probes/pipeline_probe.py > $O/pipeline_probe.log 2>&1; echo "probe rc $?"; grep -v "solve only" $O/pipeline_probe.log | tail -16
for m in inline prep; do
  MODE=$m rocprofv3 --kernel-trace --output-format csv -d $O/trace_$m -- python3 profiles/probes/pipeline_trace.py > $O/trace_$m.out 2>&1; echo "trace $m rc $?"
  python3 profiles/probes/trace_overlap.py $O/trace_$m 40 > $O/overlap_$m.log 2>&1; cat $O/overlap_$m.log
done
find $O -name "*.csv" -size +1M -delete
