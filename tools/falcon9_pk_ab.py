#!/usr/bin/env python3
"""Packed f32 in the Falcon 9 campaign kernel: what LLVM's SLP vectoriser already packs and what more packing buys.
Flies 32,768 rollouts x 20,000 ticks of the fast-math f32 campaign build under different hipcc flag sets (SIXDOF_JIT_FLAGS; the
objects can be built on a machine without a GPU: elodin_amd/_jit travels) and prints us per tick plus the static instruction
mix of each object (v_pk_{add,mul,fma}_f32 vs their scalar forms).
    python tools/falcon9_pk_ab.py [rollouts] [ticks]"""
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np

FLAGS = ["", "-fslp-vectorize", "-fslp-vectorize -mllvm -slp-threshold=12", "-fslp-vectorize -mllvm -slp-threshold=32"]      # "" = the product's build: vectoriser off
REPEATS = 3          # whole passes over FLAGS, interleaved: box drift and clock settling hit every variant alike
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def static_mix(so):
    with tempfile.TemporaryDirectory() as t:
        subprocess.run(["cp", str(so), f"{t}/k.so"], check=True)
        subprocess.run([OBJDUMP, "--offloading", "k.so"], cwd=t, capture_output=True)
        dev = next(Path(t).glob("k.so.0.hipv4*"))
        txt = subprocess.run([OBJDUMP, "-d", str(dev)], capture_output=True, text=True).stdout
    import re
    k = txt[txt.index("ELi0ELb0"):] if "ELi0ELb0" in txt else txt          # the plain-policy instantiation
    k = k[:k.index("\n\n")] if "\n\n" in k else k
    c = lambda pat: len(re.findall(pat, k))
    return {"valu": c(r"\n\s*v_"), "pk_f32": c(r"v_pk_(?:add|mul|fma)_f32"), "scalar_f32": c(r"\sv_(?:add|sub|mul|fma|fmac|fmaak|fmamk)_f32"),
            "moves": c(r"v_mov_b32|v_pk_mov|v_accvgpr|v_mov_b64")}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    from elodin_amd import codegen
    from elodin_amd.models import falcon9 as f9
    params = f9.sample_params(n)
    print(f"{'SIXDOF_JIT_FLAGS':34s} {'us/tick (best of passes)':>26s} {'all passes':>30s} {'static VALU':>11s} {'pk f32':>7s} {'scalar f32':>10s} {'moves':>6s}  resources")
    times, info = {fl: [] for fl in FLAGS}, {}
    for rep in range(REPEATS):
        for fl in FLAGS:
            os.environ["SIXDOF_JIT_FLAGS"] = fl
            f9._PROGRAMS.clear()
            ex = f9.AscentExec(params, dtype=np.float32, fast_math=True)
            if fl not in info:
                so = sorted(codegen.JIT_DIR.glob("pipe_*.so"), key=lambda p: p.stat().st_atime)[-1]
                info[fl] = (dict(codegen.last_resources), static_mix(so) if Path(OBJDUMP).exists() else {})
            ex.hip.invoke_batch(1000)
            t0 = time.perf_counter()
            ex.hip.invoke_batch(ticks)
            times[fl].append((time.perf_counter() - t0) / ticks * 1e6)
            ex.close()
    for fl in FLAGS:
        res, mix = info[fl]
        print(f"{fl or '(product: SLP off)':34s} {min(times[fl]):26.3f} {' '.join(f'{t:.3f}' for t in times[fl]):>30s} {mix.get('valu', 0):11d} {mix.get('pk_f32', 0):7d} "
              f"{mix.get('scalar_f32', 0):10d} {mix.get('moves', 0):6d}  vgprs {res.get('vgprs')} agprs {res.get('agprs')} scratch {res.get('scratch_bytes_per_lane')}")


if __name__ == "__main__":
    main()
