#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r6s3
mkdir -p $O
for m in inline priority; do
  MODE=$m rocprofv3 --kernel-trace --output-format csv -d $O/trace_$m -- python3 profiles/probes/pipeline_trace.py > $O/trace_$m.out 2>&1; echo "trace $m rc $?"
  python3 profiles/probes/trace_overlap.py $O/trace_$m 40 > $O/overlap_$m.log 2>&1; cat $O/overlap_$m.log
done
STREAMS=1 MODE=inline rocprofv3 --kernel-trace --output-format csv -d $O/trace_one -- python3 profiles/probes/pipeline_trace.py > $O/trace_one.out 2>&1
python3 profiles/probes/trace_overlap.py $O/trace_one 40 > $O/overlap_one.log 2>&1; cat $O/overlap_one.log
find $O -name "*.csv" -size +2M -delete
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -4 $O/gpu_suite.log
