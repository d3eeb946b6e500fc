"""The backend's pure-host layer under AddressSanitizer + UBSan ("ASan-instrumented host build"): the ECS column store
(csrc/world.cpp) and the commit hand-off (csrc/telemetry_sink.cpp) built with g++ -fsanitize=address,undefined — no HIP, no GPU — and
driven through growth, error paths and pointer invalidation by csrc/asan_host_test.cpp."""
import shutil
import subprocess
from pathlib import Path

import pytest

CSRC = Path(__file__).resolve().parents[1] / "elodin_amd" / "csrc"


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_host_layer_is_clean_under_asan_and_ubsan():
    build = subprocess.run(["make", "-C", str(CSRC), "asan"], capture_output=True, text=True)
    if build.returncode != 0 and "asan" in build.stderr.lower() and "cannot find" in build.stderr.lower():
        pytest.skip("libasan is not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(CSRC / "build" / "asan_host_test")], capture_output=True, text=True, timeout=120,
                         env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert run.returncode == 0 and "asan_host_test: ok" in run.stdout, (run.stdout[-500:], run.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr
