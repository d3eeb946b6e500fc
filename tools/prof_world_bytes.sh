#!/bin/bash
# HBM bytes per launch of the whole-world configs[1] module at one tick per launch (PMC, corrected as the guide prescribes: FETCH_SIZE x 2
# on gfx950, KiB units; separate passes):   gpurun -- 'bash tools/prof_world_bytes.sh'   -> gpurun_out/world_bytes.md
# WORLD_ARITH=relaxed WORLD_ONE_WORLD=1 bash tools/prof_world_bytes.sh   -> gpurun_out/world_bytes_relaxed_one_world.md (stablehlo.world_system(arith="relaxed", one_world=True))
A=${WORLD_ARITH:-reference}; SFX=$([ "$A" = reference ] && echo "" || echo "_$A")$([ "${WORLD_ONE_WORLD:-0}" = 1 ] && echo "_one_world" || echo ""); export WORLD_ARITH=$A
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/world_bytes$SFX; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o k -- python $R/tools/prof_world_bytes.py > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o k -- python $R/tools/prof_world_bytes.py > $O/write.log 2>&1
cd $R
python - $O "$A${WORLD_ONE_WORLD:+, one_world}" <<'PY' > gpurun_out/world_bytes$SFX.md
import csv, glob, sys
O = sys.argv[1]
def avg(which, name):
    v = [float(r["Counter_Value"]) for f in glob.glob(f"{O}/{which}/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f, newline=""))
         if "sixdof_step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name and int(r.get("Grid_Size") or 0) == 65536]
    return sum(v) / len(v), len(v)
f, nf = avg("fetch", "FETCH_SIZE"); w, nw = avg("write", "WRITE_SIZE")
rd, wr = 2 * f * 1024, w * 1024
print(f"whole-world configs[1] module (arith = {sys.argv[2]}), 65,536 entities, one tick per launch ({nf} / {nw} launches counted):")
print(f"FETCH_SIZE {f:.1f} KiB raw -> read {rd/65536:.1f} B per entity-tick; WRITE_SIZE {w:.1f} KiB -> written {wr/65536:.1f} B; total {(rd+wr)/65536:.1f} B per entity-tick")
print("analytic upper bound (every slot read, changed slots written): 296 + 208 = 504 B; the hand-written kernel: 384 B algorithmic, 386 B by PMC")
PY
cat gpurun_out/world_bytes$SFX.md; rm -rf $O/fetch $O/write
