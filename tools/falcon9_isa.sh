#!/bin/bash
# Disassemble the Falcon 9 f32 campaign kernel (plain-policy instantiation) into /tmp/f9_isa.s and print its instruction mix.
cd "$(dirname "$0")/.."
SO=$(python - <<'PY'
from elodin_amd.models import falcon9 as f9
from elodin_amd import codegen
cols = f9.initial_columns(f9.default_param_row()[None, :])
widths = {k: v.shape[1] for k, v in cols.items()}
tp = f9.build_program(origin=f9.pad_ecef(), algebraic_geodesy=True).trace(widths)
print(codegen.build(tp, "float32", 1, fast_math=True, column_soa=True, guard_selects=True))
PY
)
T=$(mktemp -d); cp $SO $T/k.so; (cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading k.so > /dev/null 2>&1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d $T/k.so.0.hipv4-amdgcn-amd-amdhsa--gfx950 | awk '/^[0-9a-f]+ <.*ELi0ELb0/{p=1} p{print}' > /tmp/f9_isa.s
rm -rf $T
echo "$SO -> /tmp/f9_isa.s ($(wc -l < /tmp/f9_isa.s) lines)"
for pat in "^\s*v_" "v_accvgpr" "v_mov_b32" "v_cndmask" "v_cmp" "s_nop" "v_readlane\|v_writelane" "v_fma\|v_fmac" "s_waitcnt" "^\s*s_"; do echo "$pat: $(grep -c "$pat" /tmp/f9_isa.s)"; done
