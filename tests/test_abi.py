"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/sixdof_hip.h
declares, its host-only helpers agree with the oracle, and it refuses to run without a GPU."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from elodin_amd import _lib as L
from oracle import oracle as orc

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "sixdof_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sixdof_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(str(L.LIB_PATH))
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/sixdof_hip.h but not exported"
    assert set(declared) == set(L.SYMBOLS), set(declared) ^ set(L.SYMBOLS)


def test_host_helpers_match_oracle():
    lib = L.lib()
    assert lib.sixdof_abi_version() == 2
    for name in ("world_pos", "world_vel", "world_accel", "force", "inertia", "tick", "simulation_time_step",
                 "gravity_edge", "seed", "a.world_pos"):
        assert L.component_id(name) == orc.component_id(name)
    for rate in (120.0, 1000.0, 60.0, 24.0, 1.0 / 3600.0, 333.0):
        assert lib.sixdof_quantize_time_step(rate) == orc.quantize_time_step(rate)


def test_struct_sizes_match_header():
    assert C.sizeof(L.EffectorOp) == 64
    assert C.sizeof(L.Desc) == 56
    assert C.sizeof(L.Column) == 56
    assert C.sizeof(L.Timings) == 64      # ABI 2: + graph_launches


def test_product_fails_loudly_without_gpu():
    if L.lib().sixdof_device_count() > 0:
        pytest.skip("GPU present")
    import elodin_amd as ea
    with pytest.raises(ea.BackendError, match="no CPU fallback"):
        ea.HipExec(np.zeros((2, 7)), np.zeros((2, 6)), np.ones((2, 7)))


def test_product_never_imports_oracle():
    """The oracle is the checker: nothing under elodin_amd/ or include/ may reference it."""
    for p in list((ROOT / "elodin_amd").rglob("*")) + list((ROOT / "include").rglob("*")):
        if p.suffix in (".py", ".cpp", ".hip", ".hpp", ".h") and p.is_file():
            text = p.read_text()
            assert "sixdof_oracle" not in text and "from oracle" not in text and "import oracle" not in text, p


def test_stand_alone_edge_fold_fails_loudly_without_gpu():
    if L.lib().sixdof_device_count() > 0:
        pytest.skip("GPU present")
    import elodin_amd as ea
    from elodin_amd import dsl

    @dsl.graph_fold("e", left=("x",), right=("x",), out="x", init=5.0)
    def fold_test(x, a, b):
        return x + a + b
    w = ea.World()
    a, b = w.spawn(ea.C("x", [1.0])), w.spawn(ea.C("x", [2.0]))
    w.spawn(ea.Edge(a, b, component="e"))
    with pytest.raises(ea.BackendError):
        w.build(fold_test)
