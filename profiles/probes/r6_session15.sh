#!/bin/bash
export TMPDIR=/tmp
for r in 1 2 3; do timeout -k 10 300 python profiles/probes/f4_speed_probe.py 2>&1 | grep "k = 50"; RP_BATCH_LIB=$PWD/profiles/probes/librp_batch_boolflags.so timeout -k 10 300 python profiles/probes/f4_speed_probe.py 2>&1 | grep "k = 50\|library"; done
