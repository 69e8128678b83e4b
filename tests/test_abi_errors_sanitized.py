"""tests/test_abi_errors.py once more against the HOST-ONLY AddressSanitizer + UBSan build of the library
(`vivit_amd/_build.py:build_host_sanitized`: `hipcc --offload-host-only -fsanitize=address,undefined`): the argument
checks, workspace queries and launch planning run under the sanitizers on the CPU box (GPU sanitizers are not
available on the pool).  The refused calls must come back with their status and without a sanitizer report."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_error_paths_under_host_sanitizers():
    from vivit_amd import _build

    rt = _build.sanitizer_runtime()
    if rt is None:
        pytest.skip("clang's shared ASan runtime is not installed")
    lib = _build.build_host_sanitized()
    env = dict(os.environ)
    env.update(VIVIT_HIP_LIB=lib, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    proc = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_abi_errors.py"), "-q", "-x",
                           "-p", "no:cacheprovider"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                          text=True, timeout=900)
    tail = proc.stdout[-3000:]
    assert proc.returncode == 0, tail
    assert "AddressSanitizer" not in proc.stdout and "runtime error:" not in proc.stdout, tail
    assert " passed" in proc.stdout, tail
