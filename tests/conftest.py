import os
import sys
from pathlib import Path

import pytest

# Some tests import modules of the reference checkout (read-only by contract): never leave bytecode caches next to them —
# neither from this process nor from the subprocesses it starts.
sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


@pytest.fixture(scope="session", autouse=True)
def _native_libraries_built():
    """A fresh checkout has no binaries (they are git-ignored): build the HIP backend (hipcc cross-compiles for
    gfx950 without a GPU) and the CPU oracle once per session, exactly as __graft_entry__.build() does."""
    import subprocess
    hip_so = ROOT / "elodin_amd" / "libsixdof_hip.so"
    srcs = list((ROOT / "elodin_amd" / "csrc").glob("*.hip")) + list((ROOT / "elodin_amd" / "csrc").glob("*.[ch]pp")) \
        + [ROOT / "include" / "sixdof_hip.h", ROOT / "include" / "sixdof_apollo.h"]
    if not hip_so.exists() or hip_so.stat().st_mtime < max(p.stat().st_mtime for p in srcs):
        subprocess.run(["make", "-C", str(ROOT / "elodin_amd" / "csrc"), "-j4"], check=True, stdout=subprocess.DEVNULL)
    from oracle import oracle as orc
    orc.build()
