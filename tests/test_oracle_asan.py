"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the
sanitizer runs belong to the CPU build; GPU ASan is not available on this pool).

`make -C oracle asan` builds the same restatement with -fsanitize=address,undefined; the oracle
suites (golden vectors; oracle vs the compiled reference, where that exists) are then run in a
child interpreter with the sanitizer runtime preloaded.  Any report ends the child with a
non-zero status (halt_on_error, -fno-sanitize-recover)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime():
    try:
        p = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    except (OSError, subprocess.CalledProcessError):
        return None
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_suites_clean_under_asan_ubsan():
    rt = _runtime()
    if rt is None:
        pytest.skip("no libasan.so in this toolchain")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    so = os.path.join(ROOT, "oracle", "_build", "libbbo_oracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=rt, BBO_ORACLE_SO=so,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        os.path.join(ROOT, "tests", "test_oracle_vs_reference.py")],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
