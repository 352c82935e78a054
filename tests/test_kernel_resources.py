"""Register / scratch budget of the headline kernels, checked at compile time (hipcc cross-compiles gfx950
without a GPU).  Any scratch (spilled VGPRs) makes the fused solve's launch time erratic; the gated solve is built for
FOUR resident waves per SIMD (128 VGPRs: its step works in place, the cold values wait in LDS), and since round 4 so is its
fixed-step sibling (F3: the same in-place step); the k = 1 streaming launch fits three -- all measured, see DESIGN.md's tuning log."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_headline_kernels_fit_their_waves_without_scratch():
    src = os.path.join(ROOT, "rocket_path_amd", "csrc", "ip_kernels.hip")
    r = subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off",
                        "-Rpass-analysis=kernel-resource-usage", "-c", "-o", os.devnull, src],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    usage = {}
    name = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
        if m and name:
            usage[name][m.group(1)] = int(m.group(2))
    # the benchmark's kernel k_solve_chunks<double, double, 3, STALL=false, ZV=true, MU=0, START=false>, its START=true twin (a
    # fresh batch's first solve), its ROUNDS=true twin (round 6: the watched kernel for states that were set or nudged: same budget) and
    # its fixed-step sibling k_steps_chunks<double, double, 3, ZV=true>
    gated = {k: v for k, v in usage.items() if "k_solve_chunksIddLi3ELb0ELb1ELi0E" in k}
    watched = {k: v for k, v in gated.items() if "Li0ELb0ELb1EE" in k}      # <..., MU = 0, START = false, ROUNDS = true>
    assert len(watched) == 1, sorted(gated)
    gated = {k: v for k, v in gated.items() if k not in watched}
    for k, v in watched.items():
        assert v["ScratchSize [bytes/lane]"] == 0 and v["VGPRs Spill"] == 0 and v["VGPRs"] <= 128 and v["LDS Size [bytes/block]"] == 13 * 64 * 8, (k, v)
    fixed = {k: v for k, v in usage.items() if "k_steps_chunksIddLi3ELb1ELb0E" in k}      # ... with the step's start in LDS (large batches)
    small = {k: v for k, v in usage.items() if "k_steps_chunksIddLi3ELb1ELb1E" in k}      # ... in registers (batches below three waves per SIMD)
    assert len(gated) == 2 and len(fixed) == 1 and len(small) == 1, sorted(usage)
    for k, v in list(gated.items()) + list(fixed.items()):
        assert v["ScratchSize [bytes/lane]"] == 0 and v["VGPRs Spill"] == 0, (k, v)
    for k, v in gated.items():
        assert v["VGPRs"] <= 128, (k, v)                              # four waves per SIMD
        assert v["LDS Size [bytes/block]"] == 13 * 64 * 8, (k, v)      # the step's start (11 fields) + the problem's two deltas, per lane
        assert 16 * v["LDS Size [bytes/block]"] <= 160 * 1024, (k, v)  # ... for all 16 single-wave blocks of a CU
    for k, v in fixed.items():      # round 4: F3's fixed-step launches step in place like the gated solve -- same budget, same LDS column
        assert v["VGPRs"] <= 128, (k, v)
        assert v["LDS Size [bytes/block]"] == 13 * 64 * 8, (k, v)
    for k, v in small.items():
        assert v["VGPRs"] <= 168 and v["VGPRs Spill"] == 0 and v["ScratchSize [bytes/lane]"] == 0 and v["LDS Size [bytes/block]"] == 0, (k, v)
    # the k = 1 streaming launch (two problems per lane, 14 x 16 B of fields in registers next to the step) fits three waves per SIMD
    k1 = {k: v for k, v in usage.items() if "k_newton_stream16IddLi3ELb1E" in k}
    assert len(k1) == 1
    for k, v in k1.items():
        assert v["VGPRs"] <= 168 and v["VGPRs Spill"] == 0 and v["ScratchSize [bytes/lane]"] == 0, (k, v)
        assert 3 * v["LDS Size [bytes/block]"] <= 160 * 1024, (k, v)      # three 256-thread blocks per CU
    # nothing on the Newton path may spill in its default build
    checked = 0
    for k, v in usage.items():
        if "k_newton" in k or "k_solve_chunks" in k or "k_steps_chunks" in k:
            assert v.get("VGPRs Spill", 0) == 0, (k, v)
            checked += 1
    assert checked > 60


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_one_wave_scheduling_kernels_fit_where_one_solve_wave_has_left():
    # schedule.hip, slim::: beside a resident solve (16 single-wave blocks of 128 VGPRs and 6.5 KB of LDS per CU) a block gets in only
    # if it needs no more than ONE retiring solve wave leaves behind in registers -- 128 VGPRs -- and its LDS fits beside the other 15
    # columns (160 KB - 15 x 6.5 KB = 62 KB free)
    src = os.path.join(ROOT, "rocket_path_amd", "csrc", "schedule.hip")
    r = subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off",
                        "-Rpass-analysis=kernel-resource-usage", "-c", "-o", os.devnull, src],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "warning" not in r.stderr, r.stderr[-2000:]
    usage, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
        if m and name:
            usage[name][m.group(1)] = int(m.group(2))
    slim = {k: v for k, v in usage.items() if "4slim" in k}
    assert len(slim) == 4, sorted(usage)      # count<records>, count<no records>, scan, scatter
    for k, v in slim.items():
        assert v["VGPRs"] <= 128 and v["VGPRs Spill"] == 0 and v["ScratchSize [bytes/lane]"] == 0, (k, v)
        assert v["LDS Size [bytes/block]"] <= 24 * 1024, (k, v)
