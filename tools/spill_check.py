#!/usr/bin/env python3
"""python tools/spill_check.py object.so [kernel-name-substring]     # exit 1 when a spill slot may be read before it is written
The analysis lives in elodin_amd/isa_check.py (why and how: its docstring)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from elodin_amd.isa_check import analyse, disassemble, kernels  # noqa: E402


def main(argv):
    so = Path(argv[1])
    want = argv[2] if len(argv) > 2 else ""
    text = so.read_text() if so.suffix == ".s" else disassemble(so)
    rc = 0
    for name, lines in kernels(text).items():
        if want not in name:
            continue
        findings, stats = analyse(lines)
        if not (stats["scratch_slots"] or stats["lane_slots"]):
            continue
        print(f"== {name[:100]}: {stats['instructions']} instructions, {stats['blocks']} blocks, {stats['scratch_slots']} scratch slots, "
              f"{stats['lane_slots']} SGPR-in-lane slots in {stats['lane_carriers']}")
        seen = set()
        for f in findings:
            if f["slot"] in seen:
                continue
            seen.add(f["slot"])
            rc = 1
            print(f"   READ BEFORE WRITE on some path: {f['slot']}  read at #{f['read_at']} `{f['read']}` (block {f['block']}); written at "
                  + "; ".join(f"#{w} `{t}`" for w, t in f["writers"]))
        if not findings:
            print("   every spill slot is written on every path before it is read")
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv))
