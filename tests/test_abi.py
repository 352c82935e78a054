"""The C-ABI library: it loads without a GPU, exports every symbol include/rp_batch.h
declares, and refuses to compute (loudly) when no HIP device is present."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import rocket_path_amd as rp
from rocket_path_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rp_batch.h")).read()
    return sorted(set(re.findall(r"RP_API\s+[\w\s\*]+?\b(rp_\w+)\s*\(", text)))


def test_header_and_binding_agree():
    names = _declared_symbols()
    assert len(names) >= 30
    assert sorted(capi.SIGNATURES) == names


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    # nothing else leaks out of the shared object (kernels and helpers are hidden)
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert {e for e in exported if e.startswith("rp_")} == set(_declared_symbols())
    assert not [e for e in exported if "newton" in e or "oracle" in e or "orc_" in e or e.startswith("ref")]
    # ... and the product library links neither the oracle nor anything built from the reference
    needed = subprocess.run(["readelf", "-d", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "ip_oracle" not in needed and "ref_hotpath" not in needed and "eigen_qr_ref" not in needed


def test_version_status_strings_and_defaults():
    lib = capi.load_library()
    assert b"gfx950" in lib.rp_version()
    assert lib.rp_status_string(capi.RP_OK) == b"ok"
    assert b"no CPU fallback" in lib.rp_status_string(capi.RP_ERR_NO_DEVICE)
    p = capi.Params()
    lib.rp_params_default(ctypes.byref(p))
    # the reference's compile-time constants (onedpath_ip.cpp:54, 812, 915, 927, 941, 919)
    assert (p.accel_limit, p.mu_divisor, p.boundary_fraction, p.backtrack, p.armijo, p.max_backtracks) == \
        (100.0, 10.0, 0.99, 0.5, 0.01, 100)


def test_invalid_arguments_are_rejected_without_a_device():
    lib = capi.load_library()
    h = ctypes.c_void_p()
    assert lib.rp_batch_create(ctypes.byref(h), 5, 0, 10, 0, None) == capi.RP_ERR_INVALID
    assert b"variant" in lib.rp_last_error()
    assert lib.rp_batch_create(ctypes.byref(h), 3, 7, 10, 0, None) == capi.RP_ERR_INVALID
    assert lib.rp_batch_create(ctypes.byref(h), 3, 0, 0, 0, None) == capi.RP_ERR_INVALID
    assert lib.rp_batch_create(None, 3, 0, 1, 0, None) == capi.RP_ERR_INVALID
    assert lib.rp_batch_step(None, 1) == capi.RP_ERR_INVALID
    assert lib.rp_batch_destroy(None) == capi.RP_OK
    n = ctypes.c_size_t()
    assert lib.rp_batch_size(None, ctypes.byref(n)) == capi.RP_ERR_INVALID
    # rp_device_id (ABI revision 5): a short buffer or none is refused before any device is asked
    small = ctypes.create_string_buffer(8)
    assert lib.rp_device_id(0, small, 8) == capi.RP_ERR_INVALID
    assert lib.rp_device_id(0, None, 128) == capi.RP_ERR_INVALID
    if rp.device_count() == 0:
        buf = ctypes.create_string_buffer(128)
        assert lib.rp_device_id(0, buf, 128) == capi.RP_ERR_NO_DEVICE and b"no HIP device" in lib.rp_last_error()
        with pytest.raises(rp.RpError):
            rp.device_id(0)


def test_pipeline_arguments_are_checked_without_a_device():
    # rp_pipeline (ABI revision 6): the arrangement is checked before any device is asked; without a device creation fails as
    # rp_batch_create does (no CPU fallback), and null handles are refused
    lib = capi.load_library()
    h = ctypes.c_void_p()
    assert lib.rp_pipeline_create(None, 3, 0, 64, 0, 4, 2) == capi.RP_ERR_INVALID
    assert lib.rp_pipeline_create(ctypes.byref(h), 3, 0, 64, 0, 4, 0) == capi.RP_ERR_INVALID and b"n_streams" in lib.rp_last_error()
    assert lib.rp_pipeline_create(ctypes.byref(h), 3, 0, 64, 0, 4, 5) == capi.RP_ERR_INVALID
    assert lib.rp_pipeline_create(ctypes.byref(h), 3, 0, 64, 0, 3, 2) == capi.RP_ERR_INVALID and b"multiple" in lib.rp_last_error()
    assert lib.rp_pipeline_create(ctypes.byref(h), 3, 0, 64, 0, 1, 2) == capi.RP_ERR_INVALID
    assert h.value is None
    job = ctypes.c_int64()
    assert lib.rp_pipeline_submit(None, None, None, None, None, 1e-8, 200, None, ctypes.byref(job)) == capi.RP_ERR_INVALID
    assert lib.rp_pipeline_wait(None, -1) == capi.RP_ERR_INVALID
    assert lib.rp_pipeline_stream_wait(None, 0, 0, None) == capi.RP_ERR_INVALID
    assert lib.rp_pipeline_batch(None, 0, ctypes.byref(h)) == capi.RP_ERR_INVALID
    assert lib.rp_pipeline_destroy(None) == capi.RP_OK
    if rp.device_count() == 0:
        assert lib.rp_pipeline_create(ctypes.byref(h), 3, 0, 64, 0, 4, 2) == capi.RP_ERR_NO_DEVICE and h.value is None
        with pytest.raises(rp.RpError) as e:
            rp.Pipeline(64)
        assert e.value.status == capi.RP_ERR_NO_DEVICE


@pytest.mark.skipif(rp.device_count() > 0, reason="a GPU is present: the no-device path cannot be exercised")
def test_no_device_means_loud_failure_not_a_cpu_fallback():
    with pytest.raises(rp.RpError) as e:
        rp.Batch(16)
    assert e.value.status == capi.RP_ERR_NO_DEVICE
    # the product package never imports the oracle
    import sys
    assert not [m for m in sys.modules if "oracle" in m and "rocket_path_amd" in m]
    src = "".join(open(os.path.join(ROOT, "rocket_path_amd", f)).read()
                  for f in os.listdir(os.path.join(ROOT, "rocket_path_amd")) if f.endswith(".py"))
    assert "import oracle" not in src and "oracle_api" not in src and "libip_oracle" not in src


def test_missing_extension_raises(tmp_path):
    with pytest.raises(RuntimeError) as e:
        capi.load_library(str(tmp_path / "nope.so"))
    assert "no CPU fallback" in str(e.value)


def test_headless_shell_without_gpu_fails_cleanly():
    exe = os.path.join(ROOT, "rocket_path_amd", "lib", "rp_headless")
    if not os.path.exists(exe):
        pytest.skip("host layer not built")
    if rp.device_count() > 0:
        pytest.skip("GPU present")
    r = subprocess.run([exe, "--keys", "n s"], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.skipif(not os.path.exists("/root/reference/problem.h"), reason="reference tree not present (GPU box)")
def test_plugin_links_against_the_reference_problem_header_in_static_storage(tmp_path):
    """The drop-in claim of INTEGRATION.md, built for real: csrc/host/static_shell.cpp holds two BatchedOneDPathIP as file-scope
    statics in a `Problem *` table (rocket_path.cpp:33-46); here it is compiled with the REFERENCE's own problem.h
    (RP_USE_REFERENCE_PROBLEM_H: the class derives from the reference's `struct Problem`, problem.h:3-14, and calls the shell's
    repaint()), together with the plug-in's source, linked against librp_batch.so and run.  Without a GPU the statics'
    constructors report the missing device before main starts and main returns 1; with one, tests/test_gpu_boundary.py runs
    the same program through its keys."""
    host = os.path.join(ROOT, "rocket_path_amd", "csrc", "host")
    lib = os.path.join(ROOT, "rocket_path_amd", "lib")
    exe = tmp_path / "dropin"
    r = subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-DRP_USE_REFERENCE_PROBLEM_H", "-DRP_STATIC_N=4",
                        "-I", "/root/reference", "-I", host, "-I", os.path.join(ROOT, "include"),
                        os.path.join(host, "static_shell.cpp"), os.path.join(host, "batched_problem.cpp"),
                        "-L", lib, "-lrp_batch", "-Wl,-rpath," + lib, "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([str(exe), "--keys", "n s"], capture_output=True, text=True, timeout=120)
    if rp.device_count() == 0:
        assert run.returncode == 1 and run.stderr.count("rp_batch_create failed") == 2 and run.stdout == ""
    else:
        assert run.returncode == 0 and run.stderr == "" and "Duration 0:" in run.stdout


def test_static_shell_without_gpu_fails_cleanly():
    exe = os.path.join(ROOT, "rocket_path_amd", "lib", "rp_static")
    if not os.path.exists(exe):
        pytest.skip("host layer not built")
    if rp.device_count() > 0:
        pytest.skip("GPU present")
    r = subprocess.run([exe, "--keys", "n s"], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.count("rp_batch_create failed") == 2 and "no HIP device" in r.stderr      # one line per static problem, printed before main


@pytest.mark.skipif(rp.device_count() > 0, reason="a GPU is present")
def test_bench_refuses_to_run_without_a_gpu():
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no HIP device" in (r.stderr + r.stdout)
    assert "{" not in r.stdout          # no JSON line is ever printed from a CPU run
    # `python bench.py --gpus 2` as typed (no RANK in the environment): the parent -- which has not touched torch or HIP -- asks a fresh
    # child how many HIP devices there are and refuses to start ranks that would have to share GPUs: exit code 3, a message
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 3 and "needs 2 HIP devices, this machine shows 0" in r.stderr and "starting 2 ranks" not in r.stderr
    assert "{\"metric\"" not in r.stdout
    # the debug form (all ranks on one device) skips that check: the parent starts two ranks itself (torch.distributed.run); on a CPU
    # box each rank then stops with "no HIP device" and the parent returns non-zero
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    out = r.stderr + r.stdout
    assert r.returncode != 0
    assert "starting 2 ranks" in out and "torch.distributed.run" in out
    # the children got as far as looking for their GPU (the elastic agent tears the other rank down as soon as the first one has
    # failed, so the second message is not guaranteed: 1 in ~4 runs of this test saw only one)
    assert out.count("no HIP device") >= 1
    assert "{\"metric\"" not in r.stdout


def test_abi_revision_of_header_library_and_binding_agree():
    text = open(os.path.join(ROOT, "include", "rp_batch.h")).read()
    rev = int(re.search(r"#define\s+RP_ABI_VERSION\s+(\d+)", text).group(1))
    lib = capi.load_library()
    assert lib.rp_abi_version() == rev == capi.ABI_VERSION
    assert lib.rp_params_size() == ctypes.sizeof(capi.Params)


def test_python_constants_mirror_the_header():
    text = open(os.path.join(ROOT, "include", "rp_batch.h")).read()
    defs = dict(re.findall(r"#define\s+(RP_\w+)\s+(\d+)u?\b", text))
    assert int(defs["RP_DTYPE_F64"]) == capi.DTYPE_F64 and int(defs["RP_DTYPE_F32"]) == capi.DTYPE_F32
    assert int(defs["RP_DTYPE_F32_STATE"]) == capi.DTYPE_F32_STATE
    for name in ("CONVERGED", "MAXITER", "NONFINITE", "INFEASIBLE", "STALLED", "WRONG_WAY"):
        assert int(defs["RP_ST_" + name]) == getattr(capi, "ST_" + name), name
    # rp_params and its ctypes mirror: same fields, same order
    body = text[text.index("typedef struct {", text.index("Solver constants")):text.index("} rp_params;")]
    fields = re.findall(r"^\s*(?:double|int32_t)\s+(\w+)(?:\[\d+\])?;", body, flags=re.M)
    assert fields == [f[0] for f in capi.Params._fields_]
    lib = capi.load_library()
    p = capi.Params()
    lib.rp_params_default(ctypes.byref(p))
    assert p.mu_mode == 0 and p.stall_window == 0 and tuple(p.mu_sigma_try) == (0.01, 0.03)
