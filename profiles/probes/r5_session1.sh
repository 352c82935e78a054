#!/bin/bash
# Round 5, first GPU session: sanity of the suite, the counter list of the box, small-batch grid for the A/B libraries, counters of configs[1].
export TMPDIR=/tmp
O=gpurun_out/r5a
mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -20 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
rocprofv3 -L > $O/counters.txt 2>&1 || true
P=profiles/probes
for lib in tuning small_w1 small_ilp small_w1ilp; do
  RP_BATCH_LIB=$PWD/$P/librp_batch_$lib.so RP_REG_COLUMN_UPTO=300000 python3 $P/small_batch_probe.py >> $O/small_grid.log 2>&1 || { tail -5 $O/small_grid.log; exit 1; }
done
for lanes in 32 16; do
  RP_BATCH_LIB=$PWD/$P/librp_batch_tuning.so RP_REG_COLUMN_UPTO=300000 RP_LANES_PER_WAVE=$lanes SIZES=16384,32768,65536 python3 $P/small_batch_probe.py >> $O/small_grid.log 2>&1 || { tail -5 $O/small_grid.log; exit 1; }
done
cat $O/small_grid.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/f50a -- python3 $P/fixed50_pmc_probe_r5.py > $O/f50a.out 2> $O/f50a.err || { tail -5 $O/f50a.err; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $O/f50b -- python3 $P/fixed50_pmc_probe_r5.py > $O/f50b.out 2> $O/f50b.err || { tail -5 $O/f50b.err; }
echo done
