"""Register / scratch budget of the headline kernels, checked at compile time (hipcc cross-compiles gfx950
without a GPU).  Any scratch (spilled VGPRs) makes the fused solve's launch time erratic, and more than 168
VGPRs costs the third resident wave per SIMD -- both were measured, see DESIGN.md's tuning log."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_tiled_kernels_fit_three_waves_without_scratch():
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
    # k_solve_tiled<S=double, T=double, 3, GATED, STALL=false, ZV=true>: the benchmark's kernel and its ungated sibling
    tiled = {k: v for k, v in usage.items() if "k_solve_tiledIddLi3ELb" in k and k.split("k_solve_tiledIddLi3")[1].startswith(("ELb1ELb0ELb1", "ELb0ELb0ELb1"))}
    assert len(tiled) == 2, sorted(usage)
    for k, v in tiled.items():
        assert v["VGPRs"] <= 168, (k, v)
        assert v["ScratchSize [bytes/lane]"] == 0 and v["VGPRs Spill"] == 0, (k, v)
        assert 3 * v["LDS Size [bytes/block]"] <= 160 * 1024, (k, v)
    # nothing on the Newton path may spill in its default build
    for k, v in usage.items():
        if "k_newton" in k or "k_solve_tiled" in k or "k_steps_regrouped" in k:
            assert v.get("VGPRs Spill", 0) == 0, (k, v)
