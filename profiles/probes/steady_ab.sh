#!/bin/bash
# A/B of library builds at steady clocks: for each library (args: paths) the bench's headline with no extras, alternating, 3 rounds.
# prints avg launch ms (HIP events over the K timed launches) and the value
for r in 1 2 3; do
  for lib in "$@"; do
    RP_BATCH_LIB=$lib python3 bench.py --gpus 1 --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-60s avg %.4f ms second-half %.4f  value %.2f G' % (sys.argv[1][-60:], d['roofline']['avg_launch_ms'], d['roofline']['sustained_ms_per_launch'], d['value']/1e9))" $lib
  done
done
