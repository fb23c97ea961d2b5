#!/bin/bash
# Builds tools/bin/skew_asm_bench (the hand-written 4-slot block-step passes against hipcc's run64_skew<4>) and
# tools/bin/mix_bench (issue rates of instruction-class mixes); results: profiles/r06_b_issue_classes.md.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
inc=$(mktemp -d)
cd $root
python3 - "$inc" <<'PY'
import sys
sys.path.insert(0, 'tools')
import gen_skew_asm as G
import gen_skew_asm2 as G2
inc = sys.argv[1]
G.emit(f'{inc}/skew_d8.inc', [(4, 32)], 96, 8)
def v1(path, seq_fn, suffix, base=96, vop3=False, zero_g=True, header=False, C=32):
    G.VOP3_ALL, G.ZERO_G = vop3, zero_g
    L = G.Layout(4, base); seq = seq_fn(G.build(4, C, L))
    if zero_g: assert G.check(4, C, seq, L, trials=8)
    name = f"QE_SKEW_ASM_K4_C{C}{suffix}"; out = []
    if header:
        for nm, regs in (("P", L.P), ("M", L.M), ("A", L.A), ("B", L.B)):
            for k, r in enumerate(regs): out.append(f'#define {name}_{nm}{k} "{{v[{r}:{r + 1}]}}"')
        for nm, r in (("T0", L.T0), ("T1", L.T1), ("HP", L.HP), ("HM", L.HM), ("GP", L.GP), ("GM", L.GM)): out.append(f'#define {name}_{nm} "{{v{r}}}"')
        out.append(f"#define {name}_CLOBBERS " + ", ".join(f'"v{r}"' for r in range(L.first_tmp, L.end)))
    out.append(f"#define {name}_TEXT \\")
    out += [f'    "{G.fmt(x)}\\n\\t" \\' for x in seq] + ['    ""\n']
    open(path, "w").write("\n".join(out)); G.VOP3_ALL, G.ZERO_G = False, True
v1(f'{inc}/skew_prog.inc', lambda p: p, "_PROG")
v1(f'{inc}/skew_d32.inc', lambda p: G.schedule(p, 32, 800), "_D32")
v1(f'{inc}/skew_vop3.inc', lambda p: G.schedule(p, 8), "_VOP3", vop3=True)
v1(f'{inc}/skew_low.inc', lambda p: G.schedule(p, 8), "_LOW", base=16, header=True)
v1(f'{inc}/skew_c8.inc', lambda p: G.schedule(p, 8), "_ACC", zero_g=False, C=8)
G2.emit(f'{inc}/skew2.inc', [(4, 32)], 96, 40, 4)
def v2(path, name, seq):
    open(path, "w").write("\n".join([f"#define {name} \\"] + [f'    "{G2.fmt(x)}\\n\\t" \\' for x in seq] + ['    ""\n']))
L = G2.Layout(4, 96, 40)
v2(f'{inc}/skew2_prog.inc', 'QE_SKEW2_K4_PROG_TEXT', G2.build(4, 32, L))
v2(f'{inc}/skew2_vop3.inc', 'QE_SKEW2_K4_VOP3_TEXT', G2.schedule(G2.build(4, 32, L, vop3_all=True), 4))
PY
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I$inc -Iquicked_amd/csrc -Iinclude tools/skew_asm_bench.hip -o tools/bin/skew_asm_bench
python3 tools/gen_mix_bench.py $inc/mix_bench.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $inc/mix_bench.hip -o tools/bin/mix_bench
rm -rf $inc
