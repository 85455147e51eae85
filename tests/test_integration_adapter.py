"""INTEGRATION.md section B, compiled: integration/hip_optimizer.h (a MultivariateOptimizer
over the C ABI) against the reference's OWN src/multivariate/multivariate.h:132-146 and
include/bbopt_hip.h, linked with libbbopt_hip.so.  Build container only -- the reference's header
is included from where it lies under /root/reference (nothing is copied); skipped elsewhere."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INC = "/root/reference/src/multivariate"


def test_adapter_compiles_links_and_reports_like_the_reference(tmp_path):
    if not os.path.exists(os.path.join(REF_INC, "multivariate.h")):
        pytest.skip("the reference is not on this machine")
    from bboptpy_amd import _ffi
    exe = str(tmp_path / "adapter_check")
    libdir = os.path.dirname(_ffi.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror",
           "-I" + REF_INC, "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "integration"),
           os.path.join(ROOT, "integration", "adapter_check.cpp"),
           "-L" + libdir, "-lbbopt_hip", "-Wl,-rpath," + libdir, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    if _ffi.lib().bbo_device_count() > 0:
        assert run.returncode == 0, run.stdout + run.stderr
        assert "converged: yes" in run.stdout
    else:
        # no GPU: bbo_create -> BBO_ERR_NO_DEVICE -> std::invalid_argument, no CPU fallback
        assert run.returncode == 3, run.stdout + run.stderr
        assert run.stdout.startswith("invalid_argument: ") and "device" in run.stdout.lower()
