#!/bin/bash
# Round-3 profile collection (GPU box, from the repo root):  bash profiles/collect_r3.sh
# (gpurun merges outputs into the local gpurun_out/ without deleting an earlier call's files: profiles/summarize.py takes the newest.)
# rocprofv3 gets the program directly after `--` (no env/bash hop); counters and traces in separate passes.
set -e
export TMPDIR=/tmp
O=gpurun_out/r3prof
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 > $O/fetch.out 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 > $O/write.out 2> $O/write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/sq1 -- python3 profiles/pmc_probe.py > $O/sq1.out 2> $O/sq1.err
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python3 profiles/pmc_probe.py > $O/sq2.out 2> $O/sq2.err
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/sq3 -- python3 profiles/pmc_probe.py > $O/sq3.out 2> $O/sq3.err
python3 bench.py > $O/bench.json 2> $O/bench.err
echo collected
