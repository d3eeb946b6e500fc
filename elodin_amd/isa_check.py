"""ISA-level check of a generated code object: is every SPILL SLOT written on every path before it is read?

codegen.py refuses builds whose register allocation spills VGPRs, because a spilling build of a fuzz-generated program once computed
wrong values on gfx950 (HISTORY.md §4).  This module looks at the machine code for that mechanism: it disassembles a code object
(llvm-objdump), rebuilds the control-flow graph of each kernel and runs a forward MUST-be-initialised analysis over the two places the
register allocator parks values:
  * scratch memory:      scratch_store_dword[xN] off, vA, off offset:K      ...   scratch_load_dword[xN] vB, off, off offset:K
  * lanes of a VGPR:     v_writelane_b32 vS, sX, L   (an SGPR spilled into lane L of vS)   ...   v_readlane_b32 sY, vS, L
A slot that some path reaches a READ of without having passed a WRITE is reported.  Any other definition of a lane-spill carrier
resets its lanes to "unknown" (a reload of the carrier from scratch restores them).  Used by tools/spill_check.py (CLI), by
tools/spill_repro/run.py (profiles/r05_spill_repro.md) and, opt-in, by codegen.py: SIXDOF_ALLOW_SPILLS=checked accepts a spilling
build only when this check is clean."""
import re
import subprocess
import tempfile
from pathlib import Path

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BR = re.compile(r"^(s_branch|s_cbranch_\w+)\s+(\S+)")
END = ("s_endpgm", "s_setpc_b64", "s_trap")


def disassemble(so: Path) -> str:
    with tempfile.TemporaryDirectory() as t:
        subprocess.run(["cp", str(so), f"{t}/k.so"], check=True)
        subprocess.run([OBJDUMP, "--offloading", "k.so"], cwd=t, capture_output=True)
        dev = list(Path(t).glob("k.so.0.hipv4*")) or [Path(t) / "k.so"]
        return subprocess.run([OBJDUMP, "-d", "--symbolize-operands", "--no-show-raw-insn", str(dev[0])], capture_output=True, text=True).stdout


def kernels(text: str):
    """{name: [lines]} of the functions in a disassembly."""
    out, cur = {}, None
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
        if m and not re.match(r"^L\d+$", m.group(1)):
            cur = out.setdefault(m.group(1), [])
        elif m and cur is not None:
            cur.append(f"<{m.group(1)}>:")              # a local label (--symbolize-operands)
        elif cur is not None and ln.strip():
            cur.append(ln)
    return out


def regs(tok: str):
    """'v[4:7]' -> ['v4'..'v7'], 'v12' -> ['v12']"""
    m = re.match(r"^([vsa])\[(\d+):(\d+)\]$", tok)
    if m:
        return [f"{m.group(1)}{k}" for k in range(int(m.group(2)), int(m.group(3)) + 1)]
    return [tok] if re.match(r"^[vsa]\d+$", tok) else []


def analyse(lines):
    """-> (findings, stats) for one function."""
    # ---- instructions and blocks ----
    insts, labels = [], {}
    for ln in lines:
        s = ln.strip()
        m = re.match(r"^<(L\d+)>:$", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        s = s.split("//")[0].strip()
        if s:
            insts.append(s)
    leaders = {0} | set(labels.values())
    for k, s in enumerate(insts):
        if BR.match(s) or s.startswith(END):
            leaders.add(k + 1)
    starts = sorted(x for x in leaders if x < len(insts))
    block_of, blocks = {}, []
    for b, st in enumerate(starts):
        en = starts[b + 1] if b + 1 < len(starts) else len(insts)
        blocks.append((st, en))
        block_of[st] = b
    succ = [[] for _ in blocks]
    for b, (st, en) in enumerate(blocks):
        last = insts[en - 1]
        m = BR.match(last)
        if m:
            tgt = labels.get(m.group(2).strip("<>"))
            if tgt is not None and tgt in block_of:
                succ[b].append(block_of[tgt])
            if m.group(1) != "s_branch" and en in block_of:
                succ[b].append(block_of[en])
        elif not last.startswith(END) and en in block_of:
            succ[b].append(block_of[en])
    pred = [[] for _ in blocks]
    for b, ss in enumerate(succ):
        for t in ss:
            pred[t].append(b)

    # ---- slot events ----
    lane_carriers = set()
    for s in insts:
        m = re.match(r"^v_writelane_b32 (v\d+),", s)
        if m:
            lane_carriers.add(m.group(1))

    def events(s):
        """[(kind, slot)] with kind in {'def', 'use', 'kill-carrier'}"""
        op, _, rest = s.partition(" ")
        args = [a.strip() for a in rest.split(",")]
        ev = []
        m = re.match(r"^scratch_(store|load)_(dword|dwordx2|dwordx3|dwordx4|short|byte|ubyte|sbyte|ushort|sshort)", op)
        if m:
            n = {"dword": 1, "dwordx2": 2, "dwordx3": 3, "dwordx4": 4}.get(m.group(2), 1)
            off = re.search(r"offset:(\d+)", s)
            base = [a for a in args if re.match(r"^s\d+$", a)]
            key = (base[0] if base else "off", int(off.group(1)) if off else 0)
            for k in range(n):
                ev.append(("def" if m.group(1) == "store" else "use", ("scratch", key[0], key[1] + 4 * k)))
            if m.group(1) == "load":                      # a lane-spill carrier coming back from scratch: its lanes are as stored
                for r in regs(args[0]):
                    if r in lane_carriers:
                        ev.append(("kill-carrier", r))
            return ev
        if op == "v_writelane_b32":
            lane = args[2]
            ev.append(("def", ("lane", args[0], lane)))
            return ev
        if op == "v_readlane_b32" and args[1] in lane_carriers:
            ev.append(("use", ("lane", args[1], args[2])))
            return ev
        # any other definition of a lane-spill carrier: its lanes hold whatever that instruction put there
        if args and op.startswith(("v_", "scratch_load", "global_load", "buffer_load", "ds_read", "ds_load")) and not op.startswith(("v_cmp", "v_cmpx")):
            for r in regs(args[0]):
                if r in lane_carriers:
                    ev.append(("kill-carrier", r))
        return ev

    per_inst = [events(s) for s in insts]
    universe = {slot for ev in per_inst for kind, slot in ev if kind in ("def", "use")}
    # a carrier reloaded from scratch as a whole (scratch_load into it) restores the lanes written before its store: treat a
    # scratch_load of a carrier as defining all its lanes IF the matching slot was stored from the same carrier; otherwise unknown.
    # (conservative and simple: reload = all lanes defined only when that scratch slot is initialised, which the scratch analysis checks)
    def transfer(b, state):
        st, en = blocks[b]
        bad = []
        state = set(state)
        for k in range(st, en):
            for kind, slot in per_inst[k]:
                if kind == "use":
                    if slot not in state:
                        bad.append((k, slot))
                elif kind == "def":
                    state.add(slot)
                else:                                   # kill-carrier
                    s = insts[k]
                    if s.startswith("scratch_load"):
                        # reload of a spilled carrier: its lanes come back as they were stored
                        state |= {x for x in universe if x[0] == "lane" and x[1] == slot}
                    else:
                        state -= {x for x in universe if x[0] == "lane" and x[1] == slot}
        return state, bad
    TOP = None
    inn = [TOP] * len(blocks)
    inn[0] = set()
    out = [TOP] * len(blocks)
    changed, rounds = True, 0
    while changed and rounds < 200:
        changed, rounds = False, rounds + 1
        for b in range(len(blocks)):
            if b:
                ps = [out[p] for p in pred[b] if out[p] is not TOP]
                if not ps:
                    continue
                new_in = set.intersection(*ps) if ps else set()
            else:
                new_in = set()
            o, _ = transfer(b, new_in)
            if inn[b] is TOP or new_in != inn[b] or out[b] is TOP or o != out[b]:
                inn[b], out[b], changed = new_in, o, True
    findings = []
    for b in range(len(blocks)):
        if inn[b] is TOP:
            continue
        _, bad = transfer(b, inn[b])
        for k, slot in bad:
            writers = [i for i, ev in enumerate(per_inst) if ("def", slot) in ev]
            findings.append({"slot": slot, "read_at": k, "read": insts[k], "block": b, "writers": [(w, insts[w]) for w in writers[:4]]})
    stats = {"instructions": len(insts), "blocks": len(blocks), "scratch_slots": len({s for s in universe if s[0] == "scratch"}),
             "lane_slots": len({s for s in universe if s[0] == "lane"}), "lane_carriers": sorted(lane_carriers)}
    return findings, stats


