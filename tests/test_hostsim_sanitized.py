"""The host build of the kernels' scalar logic (tests/hostsim: logic.h, brent.h, dc_logic.h with a serial reducer) under
AddressSanitizer + UndefinedBehaviorSanitizer: the same checks as test_logic_hostsim.py / test_dc_hostsim.py, run in a child process
that preloads the sanitizer runtimes.  (GPU AddressSanitizer is not available on this pool; the indexing of the device workspaces is
covered by the canary / poison mode, tests/test_gpu_robustness.py.)"""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _runtime(name):
    p = subprocess.check_output(["gcc", "-print-file-name=" + name], text=True).strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_hostsim_under_asan_ubsan(tmp_path):
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("sanitizer runtimes not installed")
    so = str(tmp_path / "libmopt_hostsim_san.so")
    subprocess.check_call(["g++", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-shared", "-fPIC", "-std=c++17", "-o", so, os.path.join(HERE, "hostsim", "hostsim.cpp")])
    env = dict(os.environ, MOPT_HOSTSIM_SO=so, LD_PRELOAD=asan + ":" + ubsan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(HERE, "test_logic_hostsim.py"),
                        os.path.join(HERE, "test_dc_hostsim.py")], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    assert " passed" in r.stdout
