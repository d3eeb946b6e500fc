"""SIXDOF_GRAPH_SPLIT (csrc/sixdof_capi.cpp ensure_graph): the replayed graph as S parallel row-block chains.  Off by default — it does
not pay at BASELINE size (profiles/r06_k1_floor.md) — but it stays selectable for A/B runs, so it must stay CORRECT: the same kernel over
row blocks, hence the same bits, whatever S and whether or not the row count divides."""
import hashlib
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu

CHILD = r"""
import hashlib, sys
sys.path.insert(0, %r)
import numpy as np
import bench
n = int(sys.argv[1])
ex, w, eff = bench.make_exec(n, 0, 0, 1, True)
ex.prepare(40)
t = ex.invoke_batch(40)          # one replayed chain of 40 one-tick launches
ex.invoke_batch(3)               # a short batch: eager launches
ex.download()
h = hashlib.sha256()
for f in ("world_pos", "world_vel", "world_accel", "force"):
    h.update(np.ascontiguousarray(getattr(ex, f)).tobytes())
print("SHA", h.hexdigest(), int(t.graph_launches), int(ex.tick))
ex.close()
""" % str(ROOT)


@pytest.mark.parametrize("n", [65536, 70001])
def test_row_block_chains_give_the_same_bits(n):
    got = {}
    for split in ("1", "2", "4"):
        env = dict(os.environ, SIXDOF_GRAPH_SPLIT=split)
        r = subprocess.run([sys.executable, "-c", CHILD, str(n)], capture_output=True, text=True, timeout=300, env=env, cwd=str(ROOT))
        assert r.returncode == 0, r.stderr[-1500:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("SHA")][-1].split()
        got[split] = line[1]
        assert int(line[2]) >= 36 and line[3] == "43"       # replayed from the graph (the first launch after an upload is the eager accel-check kernel), ticks counted once
    assert got["1"] == got["2"] == got["4"], got
