#!/bin/bash
# pair-path kernels under rocprofv3 (kernel trace) + the legs' own numbers:  gpurun -- 'bash tools/r06_pair_prof.sh <tag>'
TAG=${1:-r06pair}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --extras --only-legs nbody,sparse_edges,sparse_edges_hubs --extras-out $O/pair_legs.json > $O/line.json 2> $O/line.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o pair -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --extras --only-legs sparse_edges,sparse_edges_hubs,nbody --extras-out $O/pair_legs_traced.json > $O/traced.json 2> $O/traced.err
cd $R
python - <<'PY' > $O/pair_kernels.md
import csv, glob, sys
from collections import defaultdict
O = sys.argv[1] if len(sys.argv) > 1 else "."
PY
python - $O <<'PY' > $O/pair_kernels.md
import csv, glob, sys
from collections import defaultdict
O = sys.argv[1]
st = defaultdict(list)
for f in glob.glob(f"{O}/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        if any(k in r["Kernel_Name"] for k in ("pair_", "edge_", "allpairs")):
            st[(r["Kernel_Name"][:90], int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("| kernel | grid | launches | avg us | min us | max us |\n|---|---|---|---|---|---|")
for (n, g), d in sorted(st.items(), key=lambda kv: -sum(kv[1])):
    print(f"| `{n}` | {g} | {len(d)} | {sum(d)/len(d)/1e3:.3f} | {min(d)/1e3:.3f} | {max(d)/1e3:.3f} |")
PY
rm -rf $O/trace
cat $O/pair_kernels.md; python -c "
import json; d=json.load(open('$O/pair_legs.json'))
for k,v in d.items(): print(k, {x: v[x] for x in v if x in ('ms_per_tick','launches_per_tick','edge_evals_per_s','pair_evals_per_s','leg_seconds','error')})"
