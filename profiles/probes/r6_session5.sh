#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r6s5
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_boundary.py -m gpu -x -q -k "pipeline or scheduled_order" > $O/boundary.log 2>&1; echo "boundary rc $?"; tail -3 $O/boundary.log
timeout -k 10 600 python profiles/probes/pipeline_probe.py > $O/pipeline_probe.log 2>&1; echo "probe rc $?"; grep -v "solve only" $O/pipeline_probe.log | tail -16
MODE=prep rocprofv3 --kernel-trace --output-format csv -d $O/trace_prep -- python3 profiles/probes/pipeline_trace.py > $O/trace_prep.out 2>&1; echo "trace rc $?"
python3 profiles/probes/trace_overlap.py $O/trace_prep 40 > $O/overlap_prep.log 2>&1; head -8 $O/overlap_prep.log
find $O -name "*.csv" -size +1M -delete
