#!/bin/bash
# A/B of fixed-step builds: timing + SQ counters per wave-step.  usage: fixed50_ab.sh <outdir> <lib or "-" for in-tree> ...
set -o pipefail
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset RP_BATCH_LIB; tag=intree; else export RP_BATCH_LIB=$PWD/$lib; tag=$(basename $lib .so); fi
  for r in 1 2; do timeout -k 10 120 python3 profiles/probes/fixed50_time.py >> $O/time.log 2>&1; done
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_$tag -- python3 profiles/probes/fixed50_pmc_probe_r5.py > $O/pmc_$tag.out 2>&1
  python3 profiles/probes/fixed50_pmc_summary.py $O/pmc_$tag $tag >> $O/pmc.log 2>&1
done
unset RP_BATCH_LIB
cat $O/time.log; cat $O/pmc.log
find $O -name "*.csv" -size +1M -delete
