"""The generated-code build path: the precompiled preamble (codegen._Hipcc) must change how long a build takes and nothing else,
and an object built for one executor carries only the kernels that executor launches.  hipcc cross-compiles gfx950 without a GPU,
so all of this runs on the CPU."""
import os
import shutil
from pathlib import Path

import numpy as np
import pytest

from elodin_amd import codegen, dsl, isa_check
import elodin_amd.exec as ea

pytestmark = pytest.mark.skipif(shutil.which(codegen.HIPCC) is None and not os.path.exists(codegen.HIPCC), reason="needs hipcc")


def _device_text(so):
    return "\n".join(l for l in isa_check.disassemble(Path(so)).splitlines() if "file format" not in l and not l.startswith("/"))


@dsl.effector(wind=3)
def _drag(wind, vel, force):
    fl = wind - vel.linear()
    return dsl.SpatialForce(linear=force.force() + 0.25 * dsl.np.linalg.norm(fl) * fl)


def _source():
    dsl.Expr.fresh()
    return codegen.generate_source(dsl.pipe(_drag).trace(), "float64", 0)


@pytest.fixture
def jit_dirs(tmp_path, monkeypatch):
    monkeypatch.setattr(codegen, "JIT_DIR", tmp_path / "jit")
    saved = codegen.PCH_DIR[0]
    codegen.PCH_DIR[0] = tmp_path / "pch"
    codegen._PCH_BROKEN.clear()
    yield tmp_path
    codegen.PCH_DIR[0] = saved


def test_preamble_is_the_leading_includes_without_the_comment_that_names_the_program():
    src = _source()
    pre = codegen._preamble(src)
    assert pre.strip() == '#include "step_kernel.hpp"'
    assert codegen._preamble("// only a comment\nint x;\n") == ""
    fast = codegen._preamble('// c\n#define SIXDOF_FAST_MATH 1\n#include "step_kernel.hpp"\n\nnamespace sixdof {\n')
    assert fast == '#define SIXDOF_FAST_MATH 1\n#include "step_kernel.hpp"\n'


def test_build_with_the_precompiled_preamble_emits_the_same_device_code(jit_dirs, monkeypatch):
    src = _source()
    n0 = dict(codegen.build_stats)
    with_pch = codegen._compile(src, "pchcase")
    assert codegen.build_stats.get("pch_builds", 0) == n0.get("pch_builds", 0) + 1
    assert codegen.build_stats.get("pch_uses", 0) == n0.get("pch_uses", 0) + 1
    res_pch = dict(codegen.last_resources)
    assert res_pch["vgprs"] > 0, "the resource remarks must survive the replayed plan"
    text_pch = _device_text(with_pch)
    # second program of the same kind: no second preamble build
    codegen._compile(src + "\n// another program\n", "pchcase")
    assert codegen.build_stats["pch_builds"] == n0.get("pch_builds", 0) + 1 and codegen.build_stats["pch_uses"] == n0.get("pch_uses", 0) + 2
    monkeypatch.setenv("SIXDOF_PCH", "0")
    monkeypatch.setattr(codegen, "JIT_DIR", jit_dirs / "jit_plain")
    plain = codegen._compile(src, "pchcase")
    assert codegen.build_stats["pch_uses"] == n0.get("pch_uses", 0) + 2
    assert _device_text(plain) == text_pch
    assert {k: codegen.last_resources[k] for k in ("vgprs", "agprs", "vgpr_spills", "scratch_bytes_per_lane")} == \
           {k: res_pch[k] for k in ("vgprs", "agprs", "vgpr_spills", "scratch_bytes_per_lane")}
    assert not [f for f in os.listdir("/tmp") if f.startswith("pchcase_")], "the replayed plan leaves no intermediate files behind"


def test_a_preamble_clang_refuses_falls_back_to_plain_hipcc(jit_dirs):
    src = _source()
    codegen._compile(src, "pchbad")
    for f in (jit_dirs / "pch").glob("pch_*.pch"):
        f.write_bytes(b"not a precompiled header")
    n0 = codegen.build_stats.get("pch_fallbacks", 0)
    so = codegen._compile(src + "\n// again\n", "pchbad")
    assert so.exists() and codegen.build_stats["pch_fallbacks"] == n0 + 1
    assert codegen.last_resources["vgprs"] > 0


def test_compile_errors_are_still_reported_with_the_compiler_output(jit_dirs):
    src = _source().replace("namespace sixdof {", "namespace sixdof {\nstatic_assert(sizeof(int) == 3, \"deliberate\");", 1)
    with pytest.raises(RuntimeError, match="deliberate"):
        codegen._compile(src, "pcherr")


def test_old_preambles_are_pruned(jit_dirs, monkeypatch):
    monkeypatch.setattr(codegen, "PCH_KEEP", 1)
    monkeypatch.setattr(codegen, "PCH_PRUNE_MIN_AGE_S", 0.0)
    src = _source()
    codegen._compile(src, "pchprune")
    first = sorted(f.name for f in (jit_dirs / "pch").glob("pch_*"))
    assert len(first) == 3
    monkeypatch.setenv("SIXDOF_JIT_FLAGS", "-ffp-contract=off")          # another flag set: another preamble
    codegen._compile(src, "pchprune")
    second = sorted(f.name for f in (jit_dirs / "pch").glob("pch_*"))
    assert len(second) == 3 and second != first


def _fold():
    @dsl.edge_fold
    def gravity(force, a_pos, a_inertia, b_pos, b_inertia):
        r = a_pos.linear() - b_pos.linear()
        norm = dsl.np.linalg.norm(r)
        return dsl.SpatialForce(linear=force.force() - 6.6743e-11 * b_inertia.mass() * a_inertia.mass() * r / (norm * norm * norm))
    return gravity


def _kernel_names(so):
    return sorted(isa_check.kernels(isa_check.disassemble(Path(so))))


def test_a_pair_object_built_for_one_executor_carries_only_its_kernels(jit_dirs):
    tf = _fold().trace()
    every = _kernel_names(codegen.build_pair(tf))
    small_rk4 = _kernel_names(codegen.build_pair(tf, integrator=0, small=True))
    tick_semi = _kernel_names(codegen.build_pair(tf, integrator=1, small=False))
    assert len(every) == 9         # pack | small x2 | hub chunk x2 | hub reduce x2 | fused fold-and-integrate x2
    # (pair_pack_kernel is a plain __global__ of the header: always there)
    assert len(small_rk4) == 2 and any("pair_small_kernel" in k and "ILi0E" in k for k in small_rk4)
    assert len(tick_semi) == 4 and not any("pair_small_kernel" in k for k in tick_semi), tick_semi
    assert any("pair_tick_fused_kernel" in k for k in tick_semi)
    assert all("ILi1E" in k for k in tick_semi if "ILi" in k), tick_semi       # the semi-implicit stage count / integrator only
    assert set(small_rk4) <= set(every) and set(tick_semi) <= set(every)
    with pytest.raises(ValueError):
        codegen.generate_pair_source(tf, integrator=2)
    src = codegen.generate_pair_source(tf, integrator=0, small=True)
    assert "kOnlyIntegrator = 0, kOnlySmall = 1" in src and "hipErrorInvalidValue" in src
