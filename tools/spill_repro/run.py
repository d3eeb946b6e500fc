#!/usr/bin/env python3
"""The gfx950 spill miscompute of HISTORY.md §4, reproduced as a stand-alone experiment.

tests/test_gpu_fuzz.py's seed-2 program (three random systems, one of them cadenced: every = 2) is generated ONCE into
tools/spill_repro/seed2.hip — the text elodin_amd/codegen.py emits, compilable on its own against elodin_amd/csrc — and built under
the flag sets below, with and without the compiler's spills; each object is installed as it is (dsl.FrozenProgram(prebuilt_so=...))
and its two ticks on 2,048 rows are compared with the numpy walk of the traced program.  Run on an MI355X:

    python tools/spill_repro/run.py            # prints, per build: registers / spills / scratch, the ISA check of
                                               # tools/spill_check.py, and how many values differ from the walker

Build container (no GPU):  `python tools/spill_repro/run.py --build-only` compiles the variants and runs the ISA checks."""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

HERE = Path(__file__).resolve().parent
HIPCC = "/opt/rocm/bin/hipcc"
BUILDS = {          # name -> flags (after the common ones)
    "O3_default (machine-LICM on: spills)": ["-O3"],
    "O3_no_machine_licm (the product's first attempt)": ["-O3", "-mllvm", "-disable-machine-licm"],
    "O1": ["-O1"],
    "O3_default_no_sgpr_to_vgpr_spill": ["-O3", "-mllvm", "-amdgpu-spill-sgpr-to-vgpr=0"],
    "O3_default_no_vgpr_to_agpr_spill": ["-O3", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"],
}


def main():
    from elodin_amd import codegen, dsl
    from tests.fuzz_gen import columns, make_program
    from tests import dsl_numpy
    prog, cols = make_program(2), columns(2, 2048)
    widths = {k: int(v.shape[1]) for k, v in cols.items()}
    dsl.Expr.fresh()
    tp = prog.trace(widths)
    src = codegen.generate_variant(tp, "program", "float64", 2)
    hip_file = HERE / "seed2.hip"
    if not hip_file.exists() or hip_file.read_text() != src:
        hip_file.write_text(src)
    out_dir = ROOT / "gpurun_out" / "spill_repro"
    out_dir.mkdir(parents=True, exist_ok=True)
    report = {}
    for name, flags in BUILDS.items():
        so = out_dir / (name.split(" ")[0] + ".so")
        res = subprocess.run([HIPCC, "--offload-arch=gfx950", *flags, "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                              "-Rpass-analysis=kernel-resource-usage", f"-I{ROOT / 'elodin_amd' / 'csrc'}", str(hip_file), "-o", str(so)],
                             capture_output=True, text=True)
        if res.returncode != 0:
            report[name] = {"build_error": res.stderr[-300:]}
            continue
        used = codegen._resources(res.stderr)
        chk = subprocess.run([sys.executable, str(ROOT / "tools" / "spill_check.py"), str(so), "ELi1ELb0"], capture_output=True, text=True)
        report[name] = {"resources": used, "isa_check": "clean" if chk.returncode == 0 else "READ BEFORE WRITE", "isa_check_detail": chk.stdout.strip().splitlines()[:6]}
    if "--build-only" not in sys.argv:
        import elodin_amd as ea
        from elodin_amd import _lib as L
        from elodin_amd import workloads
        n = 2048
        w = workloads.independent_bodies(n)
        pos, vel, inertia = (np.array(w[k], dtype=np.float64) for k in ("world_pos", "world_vel", "inertia"))
        want = {k: v.copy() for k, v in cols.items()}
        for t in range(1, 3):
            dsl_numpy._run_systems(tp.pre, pos, vel, inertia, want, tp.table, t)
            dsl_numpy._run_systems(tp.post, pos, vel, inertia, want, tp.table, t)
        for name in BUILDS:
            so = out_dir / (name.split(" ")[0] + ".so")
            if "build_error" in report[name]:
                continue
            frozen = dsl.FrozenProgram(None, [(c, wd) for c, wd in tp.columns], prebuilt_so=str(so))
            hip = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], integrator=L.INTEGRATOR_NONE, effectors=frozen,
                             columns={k: v.copy() for k, v in cols.items()})
            hip.run(2)
            bad = {}
            for k in cols:
                got = np.asarray(hip._aux[k], dtype=np.float64)
                err = np.abs(got - want[k]) / np.maximum(np.abs(want[k]), 1.0)
                bad[k] = {"values_off_by_more_than_1e-10": int((err >= 1e-10).sum()), "of": int(err.size), "worst": float(np.nanmax(err)),
                          "rows_affected": int((err >= 1e-10).any(axis=1).sum()), "non_finite": int((~np.isfinite(got)).sum())}
            report[name]["vs_walker"] = bad
            hip.close()
    (out_dir / "report.json").write_text(json.dumps(report, indent=1))
    for name, r in report.items():
        print("==", name)
        print("   ", json.dumps(r.get("resources")), "| ISA check:", r.get("isa_check"))
        for k, b in (r.get("vs_walker") or {}).items():
            print(f"    column {k}: {b['values_off_by_more_than_1e-10']} of {b['of']} values differ (worst {b['worst']:.2e}, rows {b['rows_affected']}, non-finite {b['non_finite']})")


if __name__ == "__main__":
    main()
