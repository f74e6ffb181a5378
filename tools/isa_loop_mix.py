"""Instruction mix of the hottest (longest) loop of a kernel in hipcc's gfx950 assembly (-save-temps=obj).
usage: python tools/isa_loop_mix.py <file.s> <kernel-name-regex> [--dump]
Classes follow tools/ubench/valu_rate3/4 (profiles/r02_ubench_*): 'full' = the VALU forms that issue at full rate on gfx950."""
import re, sys
from collections import Counter

FULL = re.compile(r"^v_(add_f32|sub_f32|subrev_f32|mul_f32|fma_f32|fmac_f32|fmaak_f32|fmamk_f32|add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|"
                  r"lshrrev_b32|ashrrev_i32|lshlrev_b16|lshrrev_b16|min_u16|max_u16|min_i16|max_i16|mov_b32|add_co_u32|med3_f32)(_e32|_e64)?$")
s = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r'^(_ZN3psd\w+):\s*;.*?\n(.*?)\.amdhsa_kernel \1', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if not re.search(pat, name):
        continue
    lines = body.split('\n')
    labels = {l.split(':')[0]: i for i, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)}
    best = None
    for i, l in enumerate(lines):
        mm = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            span = (labels[mm.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    if best is None:
        print(name, "no loop"); continue
    loop = [l.strip() for l in lines[best[0]:best[1] + 1] if l.startswith('\t') and not l.strip().startswith(';') and not l.strip().startswith('.')]
    ops = [l.split()[0] for l in loop if l]
    c = Counter(ops)
    valu = [o for o in ops if o.startswith('v_')]
    full = [o for o in valu if FULL.match(o)]
    print(f"{name[:90]}\n  loop lines {best}: {len(ops)} instructions; VALU {len(valu)} (full-rate {len(full)}, other {len(valu) - len(full)}); "
          f"ds_ {sum(1 for o in ops if o.startswith('ds_'))}; s_ {sum(1 for o in ops if o.startswith('s_'))}; "
          f"global/buffer {sum(1 for o in ops if o.startswith(('global_', 'buffer_')))}; s_nop {c['s_nop']}; s_waitcnt {c['s_waitcnt']}; s_barrier {c['s_barrier']}")
    print("  top:", ", ".join(f"{k} {v}" for k, v in c.most_common(40)))
    if "--dump" in sys.argv:
        print("\n".join(loop))
