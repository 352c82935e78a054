"""AddressSanitizer + UBSan over the oracle's C code (CPU build; the GPU pool offers no sanitizers)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_sanitize")
    subprocess.check_call(["gcc", "-g", "-O1", "-std=c99", "-ffp-contract=off", "-fopenmp", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-o", exe,
                           os.path.join(ROOT, "tests", "csrc", "oracle_sanitize.c"), os.path.join(ROOT, "oracle", "ip_oracle.c"), "-lm"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", OMP_NUM_THREADS="3"))
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.strip().endswith("ok")
