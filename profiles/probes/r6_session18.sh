#!/bin/bash
# kernel trace of the pipeline with the LDS-match scatter (side library) against the in-tree ballot form: how long do the pass's kernels take under a solve?
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/s18; mkdir -p $O
for lib in - profiles/probes/librp_batch_sched_lds.so; do
  if [ "$lib" = "-" ]; then unset RP_BATCH_LIB; tag=ballots; else export RP_BATCH_LIB=$PWD/$lib; tag=lds; fi
  MODE=inline rocprofv3 --kernel-trace --output-format csv -d $O/trace_$tag -- python3 profiles/probes/pipeline_trace.py > $O/trace_$tag.out 2>&1 || { tail -5 $O/trace_$tag.out; exit 1; }
  python3 profiles/probes/trace_overlap.py $O/trace_$tag 40 > $O/overlap_$tag.log 2>&1
  head -8 $O/overlap_$tag.log
  timeout -k 10 100 python3 profiles/probes/sched_time.py > $O/sched_time_$tag.log 2>&1; tail -2 $O/sched_time_$tag.log
done
find $O -name "*.csv" -size +1M -delete
