"""A/B of the hub-source path of the CSR edge fold (pair_kernel.hpp 2c): bench.py's sparse graph (65,536 bodies, 16 lattice
edges each) with 0 / 1 / 8 / 64 all-to-everyone hubs, with the hub path and with SIXDOF_NO_HUBS=1 (one lane per source).
Each case in its own process (the knob is read when the edges are set)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = "import sys, json; sys.path.insert(0, %r); from tools import bench_legs as bench; print(json.dumps(bench.sparse_edges_leg(0, int(sys.argv[1]))))" % ROOT
for hubs in (0, 1, 8, 64):
    row = []
    for no in ("0", "1"):
        o = subprocess.run([sys.executable, "-c", CHILD, str(hubs)], env=dict(os.environ, SIXDOF_NO_HUBS=no), capture_output=True, text=True, timeout=900)
        d = json.loads(o.stdout.strip().splitlines()[-1]) if o.returncode == 0 else {"error": o.stderr[-300:]}
        row.append(d)
    a, b = row
    print(f"hubs={hubs:3d} edges={a.get('edges')}: hub path {a.get('ms_per_tick')} ms/tick ({a.get('launches_per_tick')} launches, {a.get('edge_evals_per_s', 0):.3e} edge-evals/s)"
          f"   one lane per source {b.get('ms_per_tick')} ms/tick ({b.get('launches_per_tick')} launches)")
