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
build only when check_object() says so.

The check FAILS CLOSED: no device code object extracted, a disassembler that returns an error, no amdgcn kernel in
the text, or fewer spill slots found than the build reports spills — each is "not proven", i.e. dirty.  EXEC masks: a
scratch_store executed while EXEC may be narrower than at kernel entry writes only the active lanes, so it does NOT count as a
definition of the slot (a forward analysis tracks whether EXEC provably equals its entry value through the s_and_saveexec /
s_or_b64 exec pairs of structured control flow; v_writelane_b32 ignores EXEC and always counts).  That refuses some correct
objects — a value spilled and reloaded inside one divergent branch — which is the side to err on."""
import re
import subprocess
import tempfile
from pathlib import Path

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BR = re.compile(r"^(s_branch|s_cbranch_\w+)\s+(\S+)")
END = ("s_endpgm", "s_setpc_b64", "s_trap")


class IsaCheckError(RuntimeError):
    """The machine code could not be obtained: nothing was proven."""


def disassemble(so: Path, strict: bool = True) -> str:
    """Disassembly of the gfx9 code object bundled in a host shared object (or of a bare code object).  strict: raise
    IsaCheckError unless a device ELF was extracted (or `so` is one) and llvm-objdump succeeded — never fall back to the host code."""
    with tempfile.TemporaryDirectory() as t:
        subprocess.run(["cp", str(so), f"{t}/k.so"], check=True)
        ex = subprocess.run([OBJDUMP, "--offloading", "k.so"], cwd=t, capture_output=True, text=True)
        dev = sorted(Path(t).glob("k.so.*.hipv4*"))      # one bundle per translation unit of the library
        if not dev:
            if strict and b"\x7fELF" == Path(t, "k.so").read_bytes()[:4] and Path(t, "k.so").read_bytes()[18:20] != (224).to_bytes(2, "little"):
                raise IsaCheckError(f"{so}: no offload bundle could be extracted (llvm-objdump --offloading rc {ex.returncode}: {ex.stderr.strip()[:200]})")
            dev = [Path(t) / "k.so"]                      # e_machine 224 = EM_AMDGPU: a bare code object
        out = []
        for d in dev:
            r = subprocess.run([OBJDUMP, "-d", "--symbolize-operands", "--no-show-raw-insn", str(d)], capture_output=True, text=True)
            if strict and (r.returncode != 0 or "elf64-amdgpu" not in r.stdout[:400]):
                raise IsaCheckError(f"{so}: llvm-objdump -d rc {r.returncode}, not an amdgpu disassembly: {r.stderr.strip()[:200]}")
            out.append(r.stdout)
        return "\n".join(out)


def check_object(so: Path, reported_vgpr_spills: int = 0):
    """-> (clean, reason, per-kernel stats).  clean = every kernel's spill slots are provably written, with full EXEC or by
    v_writelane, on every path before they are read — and the evidence is consistent with what the build reported."""
    try:
        ks = kernels(disassemble(Path(so), strict=True))
    except (IsaCheckError, OSError, subprocess.SubprocessError) as e:
        return False, f"no disassembly: {e}", {}
    ks = {n: l for n, l in ks.items() if any(x.strip().startswith("s_endpgm") for x in l)}      # functions that are kernels
    if not ks:
        return False, "no amdgcn kernel found in the disassembly", {}
    stats, dirty = {}, []
    for name, lines in ks.items():
        findings, st = analyse(lines)
        stats[name] = st
        dirty += [(name, f) for f in findings]
    slots = sum(st["scratch_slots"] + st["lane_slots"] for st in stats.values())
    if reported_vgpr_spills > 0 and slots < reported_vgpr_spills:
        return False, f"the build reports {reported_vgpr_spills} VGPR spills but only {slots} spill slots were recognised", stats
    if dirty:
        name, f = dirty[0]
        return False, f"{len(dirty)} read(s) of a spill slot not provably written: {f['slot']} at `{f['read']}` in {name[:80]}", stats
    return True, "every spill slot written (full EXEC or writelane) before it is read on every path", stats


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

    # ---- is EXEC provably what it was at kernel entry?  (state: (exec_full, frozenset of SGPR bases holding the entry EXEC)) ----
    def writes(s):
        op, _, rest = s.partition(" ")
        args = [a.strip() for a in rest.split(",")]
        return op, args

    def sregs(tok):
        return set(regs(tok)) if tok.startswith("s") else set()

    def exec_step(s, st):
        full, saved = st
        op, args = writes(s)
        if not args or not args[0]:
            return st
        d = args[0]
        if op in ("s_and_saveexec_b64", "s_or_saveexec_b64", "s_andn2_saveexec_b64", "s_xor_saveexec_b64", "s_and_saveexec_b32",
                  "s_or_saveexec_b32", "s_andn2_saveexec_b32", "s_xor_saveexec_b32", "s_orn2_saveexec_b64", "s_nand_saveexec_b64",
                  "s_nor_saveexec_b64", "s_xnor_saveexec_b64", "s_andn1_saveexec_b64", "s_orn1_saveexec_b64", "s_andn1_wrexec_b64", "s_andn2_wrexec_b64"):
            saved = (saved - sregs(d)) | (sregs(d) if full else set())      # D = old EXEC
            return (False, frozenset(saved))                                # EXEC narrowed (or altered): no longer provably the entry mask
        if d in ("exec", "exec_lo", "exec_hi") or op.startswith("v_cmpx"):
            if op in ("s_mov_b64", "s_mov_b32") and len(args) > 1 and sregs(args[1]) and sregs(args[1]) <= saved and d == "exec":
                return (True, saved)
            if op in ("s_or_b64", "s_or_b32") and d == "exec" and len(args) > 2 and any(sregs(a) and sregs(a) <= saved for a in args[1:3]):
                return (True, saved)                                        # EXEC |= a saved entry mask: every entry lane is back
            return (False, saved)
        if op in ("s_mov_b64", "s_mov_b32") and len(args) > 1 and args[1] == "exec" and d.startswith("s"):
            return (full, frozenset((saved - sregs(d)) | (sregs(d) if full else set())))
        if d.startswith("s") and op.startswith(("s_", "v_readlane", "v_readfirstlane", "v_cmp")):
            hit = sregs(d)
            if op.startswith("v_cmp") and not hit:
                hit = {"vcc"}
            if hit & saved:
                return (full, frozenset(saved - hit))                       # a saved copy overwritten
        return st

    def exec_join(states):
        states = [x for x in states if x is not None]
        if not states:
            return None
        return (all(f for f, _ in states), frozenset(set.intersection(*[set(s_) for _, s_ in states])))

    ex_in = [None] * len(blocks)
    ex_in[0] = (True, frozenset())
    ex_out = [None] * len(blocks)
    changed, rounds = True, 0
    while changed and rounds < 200:
        changed, rounds = False, rounds + 1
        for b, (st_, en_) in enumerate(blocks):
            cur = ex_in[0] if b == 0 else exec_join([ex_out[p] for p in pred[b]])
            if cur is None:
                continue
            start = cur
            for k in range(st_, en_):
                cur = exec_step(insts[k], cur)
            if start != ex_in[b] or cur != ex_out[b]:
                ex_in[b], ex_out[b], changed = start, cur, True
    exec_full_at = [False] * len(insts)
    for b, (st_, en_) in enumerate(blocks):
        cur = ex_in[b]
        for k in range(st_, en_):
            exec_full_at[k] = bool(cur and cur[0])
            cur = exec_step(insts[k], cur) if cur is not None else None
    # a scratch store under a (possibly) narrowed EXEC writes only the active lanes: not a definition of the slot
    partial_stores = 0
    for k, ev in enumerate(per_inst):
        if insts[k].startswith("scratch_store") and not exec_full_at[k]:
            partial_stores += 1
            per_inst[k] = [(kind, slot) for kind, slot in ev if kind != "def"]
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
             "lane_slots": len({s for s in universe if s[0] == "lane"}), "lane_carriers": sorted(lane_carriers),
             "scratch_stores_under_partial_exec": partial_stores}
    return findings, stats


