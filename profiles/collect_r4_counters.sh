#!/bin/bash
# Round-4 counter passes only (GPU box, repo root):  bash profiles/collect_r4_counters.sh   -- then profiles/summarize.py, then collect_r4.sh
# (bench.py prices its roofline with the flop-per-step figures of profiles/r4_sq_counters.json: after a kernel change the counters come first.)
set -e
export TMPDIR=/tmp
O=gpurun_out/r4prof
mkdir -p $O
rm -rf $O/sq1 $O/sq2 $O/sq3
sha256sum rocket_path_amd/csrc/ip_core.h rocket_path_amd/csrc/ip_kernels.hip rocket_path_amd/csrc/feas_core.h > $O/sources.sha256
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/sq1 -- python3 profiles/pmc_probe.py > $O/sq1.out 2> $O/sq1.err
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python3 profiles/pmc_probe.py > $O/sq2.out 2> $O/sq2.err
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/sq3 -- python3 profiles/pmc_probe.py > $O/sq3.out 2> $O/sq3.err
echo counters collected
