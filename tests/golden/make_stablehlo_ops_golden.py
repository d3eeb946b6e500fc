#!/usr/bin/env python3
"""Known answers for the StableHLO front-end from the REFERENCE's own op tests.

Build container only (/root/reference):   python tests/golden/make_stablehlo_ops_golden.py

libs/cranelift-mlir/tests/ops.rs holds 215 unit tests of the reference's StableHLO compiler: an inline MLIR module, input buffers
(`f64_buf(&[..])`, `i64_buf`, `i32_buf`, `u32_buf`), one or more `run_mlir(mlir, &[inputs], &[output sizes])` calls and the expected
outputs as assertions (`assert_f64s_close(&read_f64s(&out[0]), &[..])`, `assert_eq!(read_i64s(&out[0])[0], ..)`, Rust float
expressions such as `2.0_f64.sqrt()` included).  This script extracts every test whose module, inputs and expectations are
literal enough to read with regular expressions (tests sharing a module constant, computing expectations in loops or comparing
against another run are left out) into tests/golden/stablehlo_ops.json: {name, mlir, inputs, expected}.  The modules and numbers
are the reference's test DATA, reproduced like its golden CSVs; nothing of its compiler is.  tests/test_stablehlo_ingest.py runs
them through elodin_amd/stablehlo.py on the CPU walker, tests/test_gpu_stablehlo.py through the generated kernel."""
import json
import math
import re
import sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent / "stablehlo_ops.json"
src = (REF / "libs" / "cranelift-mlir" / "tests" / "ops.rs").read_text()

tests = re.split(r'\n#\[test\]\n', src)[1:]
PI = math.pi
def rust_val(s):
    s = s.strip()
    s = re.sub(r'_?(f64|i64|i32|u32|u64)\b', '', s)
    s = s.replace('std::f64::consts::', '').replace('f64::consts::', '').replace('f64::', '')
    s = s.replace('PI', str(math.pi)).replace('E)', str(math.e)+')').replace('NAN','float("nan")').replace('INFINITY','float("inf")')
    s = re.sub(r'(\([^()]*\)|-?[\d\.eE+-]+)\.(sin|cos|tan|exp|sqrt|tanh|abs|floor|ceil|cbrt|asin|acos|atan|sinh|cosh)\(\)', lambda m: f"math.{ {'abs':'fabs'}.get(m.group(2), m.group(2)) }({m.group(1)})", s)
    s = re.sub(r'(\([^()]*\)|-?[\d\.eE+-]+)\.ln\(\)', r'math.log(\1)', s)
    s = re.sub(r'(\([^()]*\)|-?[\d\.eE+-]+)\.ln_1p\(\)', r'math.log1p(\1)', s)
    s = re.sub(r'(\([^()]*\)|-?[\d\.eE+-]+)\.exp_m1\(\)', r'math.expm1(\1)', s)
    s = re.sub(r'(\([^()]*\)|-?[\d\.eE+-]+)\.powf\(([^()]*)\)', r'math.pow(\1, \2)', s)
    s = re.sub(r'(\([^()]*\)|-?[\d\.eE+-]+)\.powi\(([^()]*)\)', r'math.pow(\1, \2)', s)
    s = re.sub(r'(\([^()]*\)|-?[\d\.eE+-]+)\.atan2\(([^()]*)\)', r'math.atan2(\1, \2)', s)
    s = re.sub(r',\s*".*"\s*$', '', s, flags=re.S)
    return eval(s, {"math": math, "float": float})
def rust_list(s):
    s = s.strip()
    if not s: return []
    parts, depth, cur = [], 0, ''
    for ch in s:
        if ch in '([': depth += 1
        if ch in ')]': depth -= 1
        if ch == ',' and depth == 0: parts.append(cur); cur = ''
        else: cur += ch
    if cur.strip(): parts.append(cur)
    return [rust_val(p) for p in parts]
split = []
for t in tests:
    runs = list(re.finditer(r'let (\w+) = run_mlir(_mem)?\(', t))
    if len(runs) <= 1: split.append(t); continue
    m0 = re.match(r'fn (test_\w+)\(\)', t)
    head = t[:runs[0].start()]
    for k, r_ in enumerate(runs):
        # inputs declared between runs belong to the later run too: keep everything before each run, asserts only after it
        end = runs[k+1].start() if k+1 < len(runs) else len(t)
        pre = re.sub(r'assert\w*!?\(.*?\);', '', t[:r_.start()], flags=re.S)
        pre = re.sub(r'let \w+ = run_mlir(_mem)?\(.*?\);', '', pre, flags=re.S)
        split.append(pre.replace(m0.group(0), f"fn {m0.group(1)}__run{k}()", 1) + t[r_.start():end])
tests = split
ok, bad = [], []
for t in tests:
    m = re.match(r'fn (test_\w+)\(\)', t)
    if not m: continue
    name = m.group(1)
    try:
        mm = re.search(r'let mlir = r#"(.*?)"#;', t, re.S)
        if not mm: raise ValueError("no inline mlir")
        mlir = mm.group(1)
        bufs = {}
        for b in re.finditer(r'let (\w+) = (f64|i64|i32|u32)_buf\(&\[(.*?)\]\);', t, re.S):
            bufs[b.group(1)] = (b.group(2), rust_list(b.group(3)))
        runs = list(re.finditer(r'let (\w+) = run_mlir(_mem)?\(\s*mlir,\s*&\[(.*?)\],\s*&\[(.*?)\]\s*\);', t, re.S))
        if not runs: raise ValueError("no run_mlir")
        if len(runs) > 1:
            whole = t
            for k, r_ in enumerate(runs):
                seg = whole[:runs[0].start()] + whole[r_.start(): (runs[k+1].start() if k+1 < len(runs) else len(whole))]
                tests.append("fn %s_run%d() {" % (name, k) + seg[seg.index("{")+1:]) if False else None
            raise ValueError("several runs: split")
        r = runs[0]
        outv = r.group(1)
        ins = [x.strip().lstrip('&') for x in r.group(3).split(',') if x.strip()]
        inputs = [bufs[i] for i in ins]
        sizes = rust_list(r.group(4))
        alias = {}
        for a in re.finditer(r'let (\w+)(?:: [^=]+)? = read_(f64|i64|i32|u32|u64)s\(&%s\[(\d+)\]\);' % outv, t):
            alias[a.group(1)] = (a.group(2), int(a.group(3)))
        exp = {}
        def put(k, ty, idx, vals):
            e = exp.setdefault(k, {"type": ty, "values": {}})
            if idx is None:
                for j, v in enumerate(vals): e["values"][j] = v
            else: e["values"][idx] = vals[0]
        for a in re.finditer(r'assert_f64s_close\(\s*&read_f64s\(&%s\[(\d+)\]\),\s*&\[(.*?)\]\s*,?\s*\);' % outv, t, re.S):
            put(int(a.group(1)), "f64", None, rust_list(a.group(2)))
        for a in re.finditer(r'assert_f64s_close\(\s*&(\w+),\s*&\[(.*?)\]\s*,?\s*\);', t, re.S):
            if a.group(1) in alias: put(alias[a.group(1)][1], "f64", None, rust_list(a.group(2)))
        for a in re.finditer(r'assert_f64_close\(\s*read_f64s\(&%s\[(\d+)\]\)\[(\d+)\],\s*(.*?)\s*\);' % outv, t, re.S):
            put(int(a.group(1)), "f64", int(a.group(2)), [rust_val(a.group(3))])
        for a in re.finditer(r'assert_f64_close\(\s*(\w+)\[(\d+)\],\s*(.*?)\s*\);', t, re.S):
            if a.group(1) in alias: put(alias[a.group(1)][1], "f64", int(a.group(2)), [rust_val(a.group(3))])
        for a in re.finditer(r'assert_eq!\(\s*read_(i64|i32|u32|u64)s\(&%s\[(\d+)\]\)\[(\d+)\],\s*(.*?)\s*\);' % outv, t, re.S):
            put(int(a.group(2)), a.group(1), int(a.group(3)), [rust_val(a.group(4))])
        for a in re.finditer(r'assert_eq!\(\s*read_(i64|i32|u32|u64)s\(&%s\[(\d+)\]\),\s*(?:vec!|&)?\[(.*?)\]\s*\);' % outv, t, re.S):
            put(int(a.group(2)), a.group(1), None, rust_list(a.group(3)))
        for a in re.finditer(r'assert_eq!\(\s*(\w+)\[(\d+)\],\s*(.*?)\s*\);', t, re.S):
            if a.group(1) in alias: put(alias[a.group(1)][1], alias[a.group(1)][0], int(a.group(2)), [rust_val(a.group(3))])
        for a in re.finditer(r'assert_eq!\(\s*(\w+),\s*(?:vec!|&)?\[(.*?)\]\s*\);', t, re.S):
            if a.group(1) in alias: put(alias[a.group(1)][1], alias[a.group(1)][0], None, rust_list(a.group(2)))
        if not exp: raise ValueError("no expectations parsed")
        ok.append(dict(name=name, mlir=mlir, inputs=[{"type": ty, "values": v} for ty, v in inputs], output_bytes=sizes,
                       expected={str(k): {"type": e["type"], "values": {str(j): v for j, v in e["values"].items()}} for k, e in exp.items()}))
    except Exception as e:
        bad.append((name, f"{type(e).__name__}: {e}"[:100]))
# ---- tests written through ops.rs's own helper functions: the helper's module template + the call's literal arguments -------
TEMPLATES = {          # helper -> (result reader, module text with {n} / {op}, number of inputs, element type)
    "mem_binop_test": ("f64", "module @module {{\n  func.func public @main(%arg0: tensor<{n}xf64>, %arg1: tensor<{n}xf64>) -> tensor<{n}xf64> {{\n"
                              "    %0 = stablehlo.{op} %arg0, %arg1 : tensor<{n}xf64>\n    return %0 : tensor<{n}xf64>\n  }}\n}}", 2, "f64"),
    "mem_unop_test": ("f64", "module @module {{\n  func.func public @main(%arg0: tensor<{n}xf64>) -> tensor<{n}xf64> {{\n"
                             "    %0 = stablehlo.{op} %arg0 : tensor<{n}xf64>\n    return %0 : tensor<{n}xf64>\n  }}\n}}", 1, "f64"),
    "i64_binop_mem_test": ("i64", "module @module {{\n  func.func public @main(%arg0: tensor<{n}xi64>, %arg1: tensor<{n}xi64>) -> tensor<{n}xi64> {{\n"
                                  "    %0 = stablehlo.{op} %arg0, %arg1 : tensor<{n}xi64>\n    return %0 : tensor<{n}xi64>\n  }}\n}}", 2, "i64"),
    "chlo_unop_test": ("f64", "module @module {{\n  func.func public @main(%arg0: tensor<{n}xf64>) -> tensor<{n}xf64> {{\n"
                              "    %0 = chlo.{op} %arg0 : tensor<{n}xf64> -> tensor<{n}xf64>\n    return %0 : tensor<{n}xf64>\n  }}\n}}", 1, "f64"),
}
TEMPLATES["chlo_unop_mem_test"] = TEMPLATES["chlo_unop_test"]
for helper, (_, template, _, _) in TEMPLATES.items():      # the templates above must be the helpers' own (ops.rs may move on)
    body = re.search(r"fn %s\(.*?\n}\n" % helper, src, re.S).group(0)
    want = re.search(r'r#"(.*?)"#', body, re.S).group(1)
    assert want == template, helper
by_helper = 0
for t in re.split(r'\n#\[test\]\n', src)[1:]:
    m = re.match(r'fn (test_\w+)\(\) \{\s*(\w+)\(\s*"(\w+)",\s*(.*?)\s*,?\s*\);\s*\}', t, re.S)
    if not m or m.group(2) not in TEMPLATES or any(c["name"] == m.group(1) for c in ok):
        continue
    reader, template, n_in, ety = TEMPLATES[m.group(2)]
    try:
        lists = [rust_list(x) for x in re.findall(r"&\[(.*?)\]", m.group(4), re.S)]
        if len(lists) != n_in + 1:
            raise ValueError("arguments")
        n = len(lists[0])
        ok.append(dict(name=m.group(1), mlir="\n" + template.format(n=n, op=m.group(3)).replace("{{", "{").replace("}}", "}") + "\n",
                       inputs=[{"type": ety, "values": v} for v in lists[:n_in]], output_bytes=[n * 8],
                       expected={"0": {"type": reader, "values": {str(j): v for j, v in enumerate(lists[n_in])}}}))
        bad[:] = [b for b in bad if b[0] != m.group(1)]
        by_helper += 1
    except Exception as e:      # noqa: BLE001
        pass
print(by_helper, "cases through helper templates")
print(len(ok), "cases extracted;", len(bad), "tests left out")
OUT.write_text(json.dumps({"source": "libs/cranelift-mlir/tests/ops.rs (inline modules + asserted outputs)", "cases": ok,
                           "left_out": [{"name": n, "why": w} for n, w in bad]}))
print(OUT, OUT.stat().st_size, "bytes")
