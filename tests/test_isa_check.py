"""elodin_amd/isa_check.py — the proof SIXDOF_ALLOW_SPILLS=checked asks for — fails CLOSED: a host object, a missing
bundle or a kernel-less text prove nothing, and a scratch store under a narrowed EXEC mask does not define its slot."""
from pathlib import Path

from elodin_amd import _lib, isa_check

ROOT = Path(__file__).resolve().parent.parent


def _findings(text):
    return isa_check.analyse([ln for ln in text.strip().splitlines()])


def test_store_then_load_with_full_exec_is_clean():
    f, st = _findings("""
        scratch_store_dword off, v1, off offset:16
        v_mov_b32 v1, 0
        scratch_load_dword v1, off, off offset:16
        s_endpgm
    """)
    assert f == [] and st["scratch_slots"] == 1 and st["scratch_stores_under_partial_exec"] == 0


def test_load_with_no_store_on_one_path_is_reported():
    f, _ = _findings("""
        s_cbranch_scc0 L1
        scratch_store_dword off, v1, off offset:16
        <L1>:
        scratch_load_dword v1, off, off offset:16
        s_endpgm
    """)
    assert len(f) == 1 and f[0]["slot"] == ("scratch", "off", 16)


def test_a_store_under_a_narrowed_exec_is_not_a_definition():
    """if (cond) { spill }  ...  reload outside the branch: the lanes for which cond was false read garbage."""
    f, st = _findings("""
        v_cmp_gt_f64 vcc, v[2:3], v[4:5]
        s_and_saveexec_b64 s[6:7], vcc
        scratch_store_dword off, v1, off offset:32
        s_or_b64 exec, exec, s[6:7]
        scratch_load_dword v9, off, off offset:32
        s_endpgm
    """)
    assert st["scratch_stores_under_partial_exec"] == 1
    assert [x["slot"] for x in f] == [("scratch", "off", 32)]


def test_exec_restored_from_its_saved_copy_is_full_again():
    f, st = _findings("""
        s_and_saveexec_b64 s[6:7], vcc
        v_add_f64 v[2:3], v[2:3], v[4:5]
        s_or_b64 exec, exec, s[6:7]
        scratch_store_dword off, v1, off offset:8
        scratch_load_dword v9, off, off offset:8
        s_endpgm
    """)
    assert f == [] and st["scratch_stores_under_partial_exec"] == 0


def test_a_clobbered_saved_mask_restores_nothing_provable():
    f, st = _findings("""
        s_and_saveexec_b64 s[6:7], vcc
        s_mov_b64 s[6:7], 0
        s_or_b64 exec, exec, s[6:7]
        scratch_store_dword off, v1, off offset:8
        scratch_load_dword v9, off, off offset:8
        s_endpgm
    """)
    assert st["scratch_stores_under_partial_exec"] == 1 and len(f) == 1


def test_a_divergent_loop_keeps_exec_partial_until_the_saved_mask_comes_back():
    text = """
        s_mov_b64 s[8:9], exec
        <L0>:
        v_cmp_lt_i32 vcc, v0, v1
        s_andn2_b64 exec, exec, vcc
        scratch_store_dword off, v2, off offset:4
        s_cbranch_execnz L0
        s_mov_b64 exec, s[8:9]
        scratch_store_dword off, v3, off offset:12
        scratch_load_dword v4, off, off offset:12
        scratch_load_dword v5, off, off offset:4
        s_endpgm
    """
    f, st = _findings(text)
    assert st["scratch_stores_under_partial_exec"] == 1                     # the one inside the loop
    assert [x["slot"] for x in f] == [("scratch", "off", 4)]                  # offset 12 was stored with the entry mask back


def test_writelane_ignores_exec_and_counts():
    f, _ = _findings("""
        s_and_saveexec_b64 s[6:7], vcc
        v_writelane_b32 v40, s12, 3
        s_or_b64 exec, exec, s[6:7]
        v_readlane_b32 s12, v40, 3
        s_endpgm
    """)
    assert f == []


def test_check_object_fails_closed():
    ok, why, _ = isa_check.check_object(ROOT / "oracle" / "libsixdof_oracle.so", 3)             # a host library: no device code at all
    assert not ok and "no disassembly" in why
    ok, why, _ = isa_check.check_object(ROOT / "does_not_exist.so", 1)
    assert not ok
    ok, why, stats = isa_check.check_object(_lib.LIB_PATH, 0)                                   # the product library: kernels, no spills
    assert ok and len(stats) > 10 and all(s["scratch_slots"] == 0 for s in stats.values()), why
    # the same clean object, had its build reported spills the scan cannot see: inconsistent evidence is not a proof
    ok, why, _ = isa_check.check_object(_lib.LIB_PATH, 100000)
    assert not ok and "reports 100000 VGPR spills" in why
