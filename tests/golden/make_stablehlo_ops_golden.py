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


def desugar(t: str) -> str:
    """Spellings the extractor below does not read, rewritten into the one it does (same module text, same numbers):
    `format!(r#"..{n}.."#)` with a literal `let n = K;`; buffers built inside the run_mlir call; ops.rs's int_cmp_test /
    convert_mem_test helpers; byte-level assertions on i1 results."""
    m = re.match(r'fn (test_\w+)\(\) \{\s*int_cmp_test\(\s*"(\w+)",\s*"(\w+)",\s*&(\w+_buf\(&\[.*?\]\)),\s*&(\w+_buf\(&\[.*?\]\)),\s*(\d+),\s*&\[(.*?)\],?\s*\);', t, re.S)
    if m:      # ops.rs int_cmp_test: compare <dir>, SIGNED over tensor<n x ty> -> tensor<n x i1>, expected as bytes
        name, d, ty, a, b, n, exp = m.groups()
        mod = (f"module @module {{\n  func.func public @main(%arg0: tensor<{n}x{ty}>, %arg1: tensor<{n}x{ty}>) -> tensor<{n}xi1> {{\n"
               f"    %0 = stablehlo.compare {d}, %arg0, %arg1, SIGNED : (tensor<{n}x{ty}>, tensor<{n}x{ty}>) -> tensor<{n}xi1>\n    return %0 : tensor<{n}xi1>\n  }}\n}}")
        helper = re.search(r'fn int_cmp_test\(.*?\n}\n', src, re.S).group(0)
        want = re.search(r'r#"(.*?)"#', helper, re.S).group(1).replace("{{", "{").replace("}}", "}")
        assert want.replace("{n}", n).replace("{ty}", ty).replace("{dir}", d) == mod, name
        return (f'fn {name}() {{\n    let mlir = r#"{mod}"#;\n    let in0 = {a};\n    let in1 = {b};\n    let out = run_mlir(mlir, &[&in0, &in1], &[{n}]);\n'
                f'    assert_eq!(read_u8s(&out[0]), vec![{exp}]);\n}}\n')
    m = re.match(r'fn (test_\w+)\(\) \{\s*let out = convert_mem_test\(\s*"(\w+)",\s*"(\w+)",\s*&(.*?),\s*(\d+),\s*(\d+),\s*(\d+),?\s*\);(.*)', t, re.S)
    if m:      # ops.rs convert_mem_test: stablehlo.convert over tensor<n x src> -> tensor<n x dst>
        name, src_ty, dst_ty, inp, _, out_esz, n, rest = m.groups()
        mod = (f"module @module {{\n  func.func public @main(%arg0: tensor<{n}x{src_ty}>) -> tensor<{n}x{dst_ty}> {{\n"
               f"    %0 = stablehlo.convert %arg0 : (tensor<{n}x{src_ty}>) -> tensor<{n}x{dst_ty}>\n    return %0 : tensor<{n}x{dst_ty}>\n  }}\n}}")
        helper = re.search(r'fn convert_mem_test\(.*?\n}\n', src, re.S).group(0)
        want = re.search(r'r#"(.*?)"#', helper, re.S).group(1).replace("{{", "{").replace("}}", "}")
        assert want.replace("{n}", n).replace("{src_ty}", src_ty).replace("{dst_ty}", dst_ty) == mod, name
        inp = inp if "_buf(" in inp else f"u8_buf(&{inp})"
        rest = re.sub(r'assert_eq!\(out,', 'assert_eq!(read_u8s(&out[0]),', rest)
        rest = re.sub(r'read_(\w+)s\(&out\)', r'read_\1s(&out[0])', rest)
        return (f'fn {name}() {{\n    let mlir = r#"{mod}"#;\n    let in0 = {inp};\n    let out = run_mlir(mlir, &[&in0], &[{int(n) * int(out_esz)}]);\n{rest}')
    nm = re.search(r'let n = (\d+);', t)
    fm = re.search(r'let mlir = format!\(\s*r#"(.*?)"#\s*,?\s*\);', t, re.S)
    if fm and (nm or "{n}" not in fm.group(1)) and not re.search(r'\{(?!n\}|\{)[a-z_]+\}', fm.group(1)):
        body = fm.group(1).replace("{n}", nm.group(1) if nm else "").replace("{{", "{").replace("}}", "}")
        t = t[:fm.start()] + 'let mlir = r#"' + body + '"#;' + t[fm.end():]
        t = t.replace("run_mlir(&mlir", "run_mlir(mlir").replace("run_mlir_mem(&mlir", "run_mlir_mem(mlir")
        if nm:
            t = re.sub(r'&\[n \* (\d+)\]', lambda q: f"&[{int(nm.group(1)) * int(q.group(1))}]", t)
    k = [0]
    def hoist(call):
        pre, text = [], call.group(0)
        def one(b):
            pre.append(f"let inl{k[0]} = {b.group(1)};")
            k[0] += 1
            return f"&inl{k[0] - 1}"
        text = re.sub(r'&((?:f64|i64|i32|u32|u8)_buf\(&\[[^\]]*\]\))', one, text)
        return "\n    ".join(pre + [text])
    t = re.sub(r'let \w+ = run_mlir(?:_mem)?\(\s*mlir,\s*&\[.*?\],\s*&\[.*?\]\s*,?\s*\);', hoist, t, flags=re.S)
    # i1 results are bytes: `assert_eq!(out[0][k], v, "..")`, `assert_eq!(out[0], vec![..])`, `assert_eq!(&out[0], &[..])`
    t = re.sub(r'assert_eq!\(\s*(\w+)\[(\d+)\]\[(\d+)\],\s*(\d+)\s*(?:,\s*"[^"]*")?\s*\);', r'assert_eq!(read_u8s(&\1[\2])[\3], \4);', t)
    t = re.sub(r'assert_eq!\(\s*&?(\w+)\[(\d+)\],\s*(?:vec!|&)\[(.*?)\]\s*\);', r'assert_eq!(read_u8s(&\1[\2]), vec![\3]);', t, flags=re.S)
    # `assert!(x[k].abs() < eps)`: zero to within the comparison's tolerance
    t = re.sub(r'assert!\(\s*(\w+)\[(\d+)\]\.abs\(\) < [\d.e-]+\s*\);', r'assert_f64_close(\1[\2], 0.0);', t)
    return t


tests = [desugar(t) for t in tests]
PI = math.pi
def rust_val(s):
    s = s.strip()
    s = s.replace('i64::MIN', '(-9223372036854775808)').replace('i64::MAX', '9223372036854775807').replace('u32::MAX', '4294967295')
    s = s.replace('std::f64::consts::', '').replace('f64::consts::', '').replace('f64::', '')
    s = re.sub(r'_?(f64|i64|i32|u32|u64)\b', '', s)
    for cname, cval in (("FRAC_PI_2", math.pi / 2), ("FRAC_PI_3", math.pi / 3), ("FRAC_PI_4", math.pi / 4), ("FRAC_PI_6", math.pi / 6),
                        ("FRAC_1_SQRT_2", 1 / math.sqrt(2)), ("SQRT_2", math.sqrt(2)), ("LN_2", math.log(2)), ("LN_10", math.log(10))):
        s = re.sub(r'\b%s\b' % cname, "(" + repr(cval) + ")", s)
    s = re.sub(r'\bPI\b', "(" + repr(math.pi) + ")", s)
    s = re.sub(r'\bE\b', "(" + repr(math.e) + ")", s)
    s = s.replace('NEG_INFINITY', '(-float("inf"))').replace('NAN','float("nan")').replace('INFINITY','float("inf")')
    s = re.sub(r'(\d)u8\b', r'\1', s)
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
        for b in re.finditer(r'let (\w+) = (f64|i64|i32|u32|u8)_buf\(&\[(.*?)\]\);', t, re.S):
            bufs[b.group(1)] = (b.group(2), rust_list(b.group(3)))
        runs = list(re.finditer(r'let (\w+) = run_mlir(_mem)?\(\s*mlir,\s*&\[(.*?)\],\s*&\[(.*?)\]\s*,?\s*\);', t, re.S))
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
        for a in re.finditer(r'let (\w+)(?:: [^=]+)? = read_(f64|i64|i32|u32|u64|u8)s\(&%s\[(\d+)\]\);' % outv, t):
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
        for a in re.finditer(r'assert_eq!\(\s*read_(i64|i32|u32|u64|u8)s\(&%s\[(\d+)\]\)\[(\d+)\],\s*(.*?)\s*\);' % outv, t, re.S):
            put(int(a.group(2)), a.group(1), int(a.group(3)), [rust_val(a.group(4))])
        for a in re.finditer(r'assert_eq!\(\s*read_(i64|i32|u32|u64|u8)s\(&%s\[(\d+)\]\),\s*(?:vec!|&)?\[(.*?)\]\s*\);' % outv, t, re.S):
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
# ---- tests whose inputs / expectations are Rust expressions (iterators, byte vectors, loops): transcribed BY HAND, the module text
# still read from ops.rs.  name -> ([(type, values)], {output: (type, values | {index: value})}); line comments = the Rust they restate.
import math as _m
def _mod(name):
    i = src.index(f"fn {name}()")
    return re.search(r'let mlir = r#"(.*?)"#;', src[i:], re.S).group(1)
MANUAL = {
    "test_transpose_3d": ([("f64", list(range(24)))], {0: ("f64", [i * 12 + j * 4 + k for j in range(3) for i in range(2) for k in range(4)])}),   # out[j][i][k] = in[i][j][k]
    "test_gather_row_select": ([("f64", list(range(1, 13))), ("u32", [2, 0])], {0: ("f64", [9, 10, 11, 12, 1, 2, 3, 4])}),
    "test_ssa_shadow_redefine": ([("i64", [10])], {0: ("u32", [42, 99]), 1: ("i64", [17])}),
    "test_select_mem": ([("u8", [1, 0, 1]), ("f64", [10, 20, 30]), ("f64", [100, 200, 300])], {0: ("f64", [10, 200, 30])}),
    "test_divide_ui32_mem": ([("u32", [0, 10, 50, 100]), ("u32", [120] * 4)], {0: ("u32", [60, 65, 85, 110])}),
    "test_not_i1": ([("u8", [1, 0, 1])], {0: ("u8", [0, 1, 0])}),
    "test_dynamic_slice_1d": ([("f64", [10, 20, 30, 40, 50]), ("i64", [2])], {0: ("f64", [30, 40])}),
    "test_dynamic_slice_3d": ([("f64", list(range(24))), ("i64", [1]), ("i64", [0]), ("i64", [0])], {0: ("f64", list(range(12, 24)))}),
    "test_dynamic_update_slice_1d": ([("f64", [1, 2, 3, 4, 5]), ("f64", [99, 100]), ("i64", [1])], {0: ("f64", [1, 99, 100, 4, 5])}),
    "test_iota_2d_dim0": ([], {0: ("i64", [0, 0, 0, 1, 1, 1, 2, 2, 2])}),
    "test_iota_2d_dim1": ([], {0: ("i64", [0, 1, 2, 0, 1, 2, 0, 1, 2])}),
    "test_atan2_mem": ([("f64", [1, -1, 0]), ("f64", [1, 1, -1])], {0: ("f64", [_m.atan2(1, 1), _m.atan2(-1, 1), _m.atan2(0, -1)])}),
    "test_slice_i32_mem": ([("i32", [10, 20, 30, 40, 50])], {0: ("i32", [20, 30, 40])}),
    "test_reduce_and_bool": ([("u8", [1, 0, 1, 0])], {0: ("u8", [1, 0, 1, 0])}),
    "test_broadcast_i32_1d_to_2d_mem": ([("i32", [10, 20, 30])], {0: ("i32", {0: 10, 1: 20, 2: 30})}),
    "test_gather_3d_pivot_permute": ([("f64", [10, 20, 30, 40, 50, 60]), ("i32", [2, 0, 1])], {0: ("f64", [30, 10, 20, 60, 40, 50])}),
    "test_gather_i32_data_mem": ([("i32", list(range(10, 130, 10))), ("i32", [1, 3])], {0: ("i32", [40, 50, 60, 100, 110, 120])}),
    "test_scatter_i32_data_mem": ([("i32", [1, 2, 3, 4, 5]), ("i32", [0, 4]), ("i32", [99, 88])], {0: ("i32", [99, 2, 3, 4, 88])}),
    "test_transpose_nd_i32_mem": ([("i32", [1, 2, 3, 4, 5, 6])], {0: ("i32", [1, 4, 2, 5, 3, 6])}),
    "test_erf_inv": ([("f64", [0.5])], {0: ("f64", [0.4769362762044699])}),
    "test_chlo_erf_inv_function_type_syntax": ([("f64", [0.0, 0.5, -0.5])], {0: ("f64", {0: 0.0, 1: 0.4769362762044699})}),
    "test_batch_norm_inference_mem": ([("f64", [1, 2, 3, 4, 5, 6]), ("f64", [1, 1, 1]), ("f64", [0, 0, 0]), ("f64", [2.5, 3.5, 4.5]), ("f64", [1, 1, 1])],
                                      {0: ("f64", {0: (1.0 - 2.5) / _m.sqrt(1.0 + 1e-5), 3: (4.0 - 2.5) / _m.sqrt(1.0 + 1e-5)})}),
    "test_lapack_syevd_2x2": ([("f64", [2, 1, 1, 3])], {1: ("f64", [(5 - _m.sqrt(5)) / 2, (5 + _m.sqrt(5)) / 2]), 2: ("i32", [0])}),      # eigenvalues + info; the vectors by reconstruction only
    "test_reduce_sum_i64_mem__run1": None, "test_scatter_i32_data_mem__run1": None,     # artefacts of the run splitter: the test has one run
    "test_scatter_i32_data_mem__run0": None,
}
manual = 0
for name, spec in MANUAL.items():
    bad[:] = [b for b in bad if b[0] != name]
    if spec is None or any(c["name"] == name for c in ok):
        continue
    ins, exp = spec
    ok.append(dict(name=name, mlir=_mod(name), inputs=[{"type": t_, "values": list(v)} for t_, v in ins], output_bytes=[],
                   expected={str(k): {"type": t_, "values": ({str(j): x for j, x in v.items()} if isinstance(v, dict) else {str(j): x for j, x in enumerate(v)})}
                             for k, (t_, v) in exp.items()}, transcribed_by_hand=True))
    manual += 1
print(manual, "cases transcribed by hand")
# ---- what stays out, and why (every remaining test must have a reason here) --------------------------------------------------
WHY = {
    "test_lapack_svd_3x3_nontrivial": "asserts U S V^T == A by a Rust loop, no literal outputs (the sign / order of singular vectors is the routine's); dgesdd is read and pinned on test_lapack_svd_* with literal values",
    "test_lapack_qr_3x3": "asserts Q^T Q == I and Q R == A by Rust loops; dgeqrf / dorgqr are read and pinned on LAPACK itself through scipy (tests/test_stablehlo_ingest.py)",
    "test_lapack_qr_orgqr_roundtrip_3x3": "same: a reconstruction property, no literal outputs",
    "test_cholesky_batched_mem": "asserts L L^T == A by a Rust loop over a batch of two matrices; the batched form is covered by the entity-parallel rule (tests/test_stablehlo_world.py)",
    "test_lapack_cholesky_batched": "same: reconstruction property over a batch",
    "test_case_large_branch_splits_into_functions__run0": "module text built by a Rust loop (70 chained adds per branch on tensor<100xf64>): tests the reference's function splitter, not an op",
    "test_case_large_branch_splits_into_functions__run1": "same",
    "test_large_external_constant_8d_dynamic_slice_mem": "module and its 65,536-element hex constant built by Rust code: tests the reference's external-constant arena",
    "test_large_external_constant_8d_dynamic_slice_nested_call_mem": "same",
    "test_dynamic_slice_clamps_out_of_bounds__run2": "artefact of the run splitter (the test has two runs; both are extracted)",
    "test_dynamic_slice_clamps_out_of_bounds__run3": "artefact of the run splitter",
    "test_shift_right_logical_mem__run0": "artefact of the run splitter (a helper-template test; extracted as test_shift_right_logical_mem)",
    "test_shift_right_logical_mem__run1": "artefact of the run splitter",
    "test_transpose_broadcast_multiply_reduce_65x65_mem": "65 x 65 operands (4,225 values each) built by Rust loops: tests the reference's pointer-ABI path for large tensors; a per-lane register program holds columns of <= 64 values",
    "test_65x65_in_pointer_abi_callee_mem": "same: 65 x 65 tensors through the pointer ABI",
    "test_65x65_while_loop_builds_matrix_then_reduces_mem": "same",
    "test_65x65_while_then_transpose_multiply_reduce_mem": "same",
    "test_convert_power_chain_65_mem": "same (65-element chain with expectations computed in a Rust loop)",
    "test_gather_nd_65x65_from_while_loop_callee_mem": "same",
    "test_multi_function_65x65_force_chain_mem": "same",
    "test_full_egm08_chain_65_mem": "same (the EGM08 gravity chain over 65 x 65 coefficient tables)",
    "test_rng_uniform_mem": "stablehlo.rng: asserts only that four values lie in [0, 1] (no literal outputs to extract); the op is read as the reference's deterministic fill and pinned on that rule (tensor_rt.rs:2103-2140) in tests/test_stablehlo_ingest.py",
}
for n_, _ in bad:
    assert n_ in WHY, f"left out without a stated reason: {n_}"
bad[:] = [(n_, WHY[n_]) for n_, _ in bad]
print(by_helper, "cases through helper templates")
print(len(ok), "cases extracted;", len(bad), "tests left out")
OUT.write_text(json.dumps({"source": "libs/cranelift-mlir/tests/ops.rs (inline modules + asserted outputs)", "cases": ok,
                           "left_out": [{"name": n, "why": w} for n, w in bad]}))
print(OUT, OUT.stat().st_size, "bytes")
