#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r6s13
mkdir -p $O; rm -rf $O/f50
bash profiles/probes/fixed50_ab.sh $O/f50 - $1 > $O/f50.log 2>&1; tail -8 $O/f50.log
timeout -k 10 600 python tests/checks/fixed_step_ab.py $O/ab_new.npz > $O/ab.log 2>&1 && \
RP_BATCH_LIB=$PWD/$1 timeout -k 10 600 python tests/checks/fixed_step_ab.py $O/ab_old.npz >> $O/ab.log 2>&1 && \
python tests/checks/inplace_ab.py cmp $O/ab_new.npz $O/ab_old.npz >> $O/ab.log 2>&1; echo "ab rc $?" >> $O/ab.log; tail -3 $O/ab.log
rm -f $O/ab_new.npz $O/ab_old.npz
