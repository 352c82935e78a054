#!/bin/bash
# Round-6 profile collection, everything in order (GPU box, repo root):
#     RP_COLLECT_COMMIT=<git rev-parse HEAD, taken in the build container> RP_COLLECT_DIRTY=<0|1> bash profiles/collect_r6.sh
#   1. SQ counter passes over profiles/pmc_probe.py (now with BASELINE configs[1]: 65,536 x 50 and 65,472 x 15), FETCH_SIZE / WRITE_SIZE
#      passes over a short bench.py (separate --pmc passes, nothing else traced, the program directly after `--`)
#   2. profiles/summarize.py folds them into profiles/r6_sq_counters.json, r6_hbm_traffic.json and r6_sources.json (hashes of the kernel
#      sources THIS box ran + the commit handed in) -- bench.py prices its rooflines with these and withholds them when the sources differ
#   3. rocprofv3 --kernel-trace --stats over the driver's command `python3 bench.py --gpus 1 --steps 20 --warmup 5`, then plain runs
# Outputs under gpurun_out/r6prof (the folded summaries are copied there too, so that they travel back).
set -e
export TMPDIR=/tmp
O=gpurun_out/r6prof
mkdir -p $O
rm -rf $O/sq1 $O/sq2 $O/sq3 $O/stats $O/fetch $O/write
sha256sum rocket_path_amd/csrc/ip_core.h rocket_path_amd/csrc/ip_kernels.hip rocket_path_amd/csrc/feas_core.h rocket_path_amd/csrc/ip_kernels.h rocket_path_amd/csrc/rp_batch.cpp rocket_path_amd/csrc/schedule.hip > $O/sources.sha256
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/sq1 -- python3 profiles/pmc_probe.py > $O/sq1.out 2> $O/sq1.err
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python3 profiles/pmc_probe.py > $O/sq2.out 2> $O/sq2.err
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/sq3 -- python3 profiles/pmc_probe.py > $O/sq3.out 2> $O/sq3.err
echo "counters collected"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 --condition-launches 0 > $O/fetch.out 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 --condition-launches 0 > $O/write.out 2> $O/write.err
echo "traffic collected"
python3 profiles/summarize.py r6 - $O/fetch $O/write $O/sq1 $O/sq2 $O/sq3 > $O/summarize.out 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_under_rocprof.json 2> $O/stats.err
python3 profiles/summarize.py r6 $O/stats $O/fetch $O/write $O/sq1 $O/sq2 $O/sq3 >> $O/summarize.out 2>&1
echo "kernel stats collected"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --sustain-seconds 3 > $O/bench.json 2> $O/bench.err
rm -f $O/bench_5runs.jsonl
for i in 1 2 3 4 5; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline >> $O/bench_5runs.jsonl 2>> $O/bench.err; done
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --force-process-group > $O/bench_forced_rccl.json 2>> $O/bench.err
cp profiles/r6_sq_counters.json profiles/r6_hbm_traffic.json profiles/r6_sources.json profiles/r6_bench_kernel_stats.csv profiles/r6_timed_launches.json $O/
# round 6: the pipeline under rocprofv3 --kernel-trace (do k_sched_* and k_solve_chunks of consecutive jobs overlap?), configs[1] per wave-step
for m in inline; do
  MODE=$m rocprofv3 --kernel-trace --output-format csv -d $O/trace_pipeline -- python3 profiles/probes/pipeline_trace.py > $O/trace_pipeline.out 2>&1
  python3 profiles/probes/trace_overlap.py $O/trace_pipeline 40 > $O/pipeline_overlap.log 2>&1
done
STREAMS=1 MODE=inline rocprofv3 --kernel-trace --output-format csv -d $O/trace_pipeline_one -- python3 profiles/probes/pipeline_trace.py > $O/trace_pipeline_one.out 2>&1
python3 profiles/probes/trace_overlap.py $O/trace_pipeline_one 40 > $O/pipeline_overlap_one_stream.log 2>&1
find $O -name "*kernel_trace.csv" -size +1M -delete
echo collected
