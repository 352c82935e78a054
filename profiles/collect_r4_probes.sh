#!/bin/bash
# Round-4 probe logs (GPU box, repo root, after profiles/build_probe_libs.sh in the build container):  bash profiles/collect_r4_probes.sh
# Everything goes to gpurun_out/r4probes; the logs are then copied to profiles/r4_*.log.
export TMPDIR=/tmp
O=gpurun_out/r4probes
mkdir -p $O
P=$PWD/profiles/probes
# the driver-run suite with the lines its tests print (ties, worst errors)
python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" | cut -c1-400 > $O/gpu_suite.log
# instructions per gated step and launch times: shipped kernels, then round 3's
{ echo "== shipped"; bash profiles/probes/valu_count.sh r4final 2>&1 | tail -7; echo "== round 3's kernels (librp_batch_r4base.so)"; RP_BATCH_LIB=$P/librp_batch_r4base.so bash profiles/probes/valu_count.sh r4base 2>&1 | tail -7; } > $O/valu_diet.log
# the gated solve's iterates, bit for bit against round 3's kernels
{ python tests/checks/inplace_ab.py run $O/ab_new.npz && RP_BATCH_LIB=$P/librp_batch_r4base.so python tests/checks/inplace_ab.py run $O/ab_base.npz && python tests/checks/inplace_ab.py cmp $O/ab_new.npz $O/ab_base.npz; } >> $O/valu_diet.log 2>&1
rm -f $O/ab_new.npz $O/ab_base.npz
# launch times of all Newton kernels, alternating
{ for i in 1 2; do python profiles/probes/r4_speed_probe.py; RP_BATCH_LIB=$P/librp_batch_r4base.so python profiles/probes/r4_speed_probe.py; done; } > $O/speed_probe.log 2>&1
# the post-convergence search with and without the closed-form failure count: bit for bit, then speed
{ python tests/checks/fixed_step_ab.py $O/fa.npz && RP_BATCH_LIB=$P/librp_batch_no_frozen_proof.so python tests/checks/fixed_step_ab.py $O/fb.npz && python tests/checks/inplace_ab.py cmp $O/fa.npz $O/fb.npz; RP_BATCH_LIB=$P/librp_batch_no_frozen_proof.so python profiles/probes/r4_speed_probe.py | tail -6; } > $O/frozen_proof_ab.log 2>&1
rm -f $O/fa.npz $O/fb.npz
RP_BATCH_LIB=$P/librp_batch_diag_moving.so python profiles/probes/moving_stats_probe.py > $O/moving_stats.log 2>&1
# k = 1: the shipped streaming kernel, the chunk kernel, the scalar streaming kernel (tuning build)
{ export RP_BATCH_LIB=$P/librp_batch_tuning.so; echo "== k_newton_stream16 (shipped choice)"; python profiles/probes/k1_ab_probe.py | tail -6; echo "== k_steps_chunks from k = 1 (RP_CHUNKS_FROM_K=1)"; RP_CHUNKS_FROM_K=1 python profiles/probes/k1_ab_probe.py | tail -6; echo "== k_newton_stream, 8 B per lane (RP_STREAM_SCALAR=1)"; RP_STREAM_SCALAR=1 python profiles/probes/k1_ab_probe.py | tail -6; unset RP_BATCH_LIB; } > $O/k1_ab.log 2>&1
python profiles/probes/solution_probe.py > $O/solution_probe.log 2>&1
python profiles/probes/f4_speed_probe.py > $O/f4_speed.log 2>&1
# 1.44 M fresh, distinct gated solves (and 12 fixed steps of the same problems) against the oracle
python tests/checks/fuzz_parity.py > $O/fuzz_parity.log 2>&1
echo probes collected
