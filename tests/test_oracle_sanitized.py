"""CPU: the oracle's own tests once more under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5,
"race / memory-error detection": the reference has none; the oracle is the checker everything else is compared
with, so an out-of-bounds read in it would silently poison parity claims).  oracle/Makefile `liboracle_san.so`,
loaded through $RFD_ORACLE_LIB in a subprocess with libasan preloaded (Python itself is not instrumented);
any sanitizer report aborts the subprocess."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_oracle_tests_under_asan_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle_san.so"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=asan, RFD_ORACLE_LIB=os.path.join(ROOT, "oracle", "liboracle_san.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               OMP_NUM_THREADS="4")
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_ops.py"),
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        os.path.join(ROOT, "tests", "test_mcubes_golden.py")],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    assert " passed" in p.stdout
