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
    # the exception barrier (csrc/abi_guard.hpp): injected std::bad_alloc in every allocating entry point came back as
    # SIXDOF_ERR_OUT_OF_MEMORY / a neutral value with the state unchanged — nothing unwound into the caller
    assert "exception barrier" in run.stdout and "came back as statuses" in run.stdout
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr


def test_every_allocating_entry_point_is_behind_the_exception_barrier():
    """Source-level gate: inside the `extern "C"` blocks of the four ABI files, every function with a body of its own is a
    function-try-block closed by SIXDOF_ABI_CATCH / SIXDOF_ABI_CATCH_VALUE; the exceptions are the one-line accessors that touch
    no allocator and the three pure-arithmetic functions.  A new entry point that forgets the barrier fails here."""
    import re
    sig = re.compile(r"^(const char\*|void\*?|int|uint64_t|uint32_t|size_t|double|sixdof_\w+\*) (sixdof_\w+)\(")
    pure = {"sixdof_component_id", "sixdof_quantize_time_step", "sixdof_shard_range",               # integer / float arithmetic only
            "sixdof_gather_block_rows", "sixdof_gather_pack", "sixdof_gather_unpack"}                    # ... and memcpy into caller buffers
    guarded, bare = [], []
    for name in ("sixdof_capi.cpp", "world.cpp", "telemetry_sink.cpp", "campaign_comm.cpp"):
        lines = (CSRC / name).read_text().split("\n")
        inside, i = False, 0
        while i < len(lines):
            ln = lines[i]
            if ln.startswith('extern "C" {'):
                inside = True
            elif ln.startswith('}  // extern "C"'):
                inside = False
            m = sig.match(ln) if inside else None
            if m and not ln.rstrip().endswith(";"):
                j = i
                while not re.search(r"[{};]\s*(//.*)?$", lines[j]):
                    j += 1
                head = " ".join(lines[i:j + 1])
                if re.search(r";\s*$", lines[j]) and "{" not in head:
                    i = j + 1
                    continue                      # a declaration
                one_liner = re.search(r"\{.*\}\s*$", lines[j]) is not None
                if one_liner:
                    assert "std::" not in head.replace("std::nothrow", "") and "new sixdof" not in head, f"{name}: {m.group(2)} allocates on one line without the barrier"
                elif " try {" in head:
                    k = j + 1
                    while not lines[k].startswith("}"):
                        k += 1
                    assert lines[k].startswith("} SIXDOF_ABI_CATCH"), f"{name}: {m.group(2)} opens a try block that no barrier macro closes"
                    guarded.append(m.group(2))
                else:
                    bare.append(m.group(2))
                i = j
            i += 1
    assert set(bare) <= pure, f"entry points without the exception barrier: {sorted(set(bare) - pure)}"
    assert len(guarded) >= 50, len(guarded)
