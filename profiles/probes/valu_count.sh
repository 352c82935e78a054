#!/bin/bash
# bash profiles/probes/valu_count.sh TAG   (GPU box, repo root): counters + timings of the gated kernel -> gpurun_out/valu_TAG.log
export TMPDIR=/tmp
T=${1:-x}
O=gpurun_out/valu_$T
rm -rf $O && mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/a -- python3 profiles/probes/valu_count_probe.py run > $O/a.out 2> $O/a.err || { tail -5 $O/a.err; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $O/b -- python3 profiles/probes/valu_count_probe.py run > $O/b.out 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/c -- python3 profiles/probes/valu_count_probe.py run > $O/c.out 2> $O/c.err || { tail -5 $O/c.err; exit 1; }
python3 profiles/probes/valu_count_probe.py read $O/a $O/b $O/c | tee gpurun_out/valu_$T.log
python3 profiles/probes/valu_count_probe.py time 2>&1 | grep gated | tee -a gpurun_out/valu_$T.log
