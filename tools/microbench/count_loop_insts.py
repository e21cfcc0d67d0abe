"""Instruction mix of the largest loop of every kernel in a gfx950 assembly file (hipcc --save-temps=obj ... -> *-gfx950.s):
python tools/microbench/count_loop_insts.py file.s   — VALU split into packed / transcendental / v_med3 / other, LDS, barriers, s_nop."""
import re
import sys
from collections import Counter

L = open(sys.argv[1]).read().splitlines()
starts = [i for i, l in enumerate(L) if re.match(r"^_Z\w+:", l)]
ends = [i for i, l in enumerate(L) if l.strip() == "s_endpgm"]
for si in starts:
    ei = min(e for e in ends if e > si)
    body = L[si:ei]
    labels = {re.match(r"^(\.LBB\d+_\d+):", l).group(1): i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = None
    for i, l in enumerate(body):
        mm = re.match(r"\s+s_c?branch\w*\s+(?:\w+,\s*)?(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i and (best is None or i - labels[mm.group(1)] > best[1] - best[0]):
            best = (labels[mm.group(1)], i)
    loop = body[best[0]:best[1] + 1] if best else body
    ins = [l.strip().split()[0] for l in loop if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    pk = sum(v for k, v in c.items() if k.startswith("v_pk_"))
    tr = sum(v for k, v in c.items() if k.startswith(("v_rsq", "v_rcp")))
    med = c.get("v_med3_f32", 0)
    dsr = sum(v for k, v in c.items() if k.startswith(("ds_read", "ds_load")))
    dsw = sum(v for k, v in c.items() if k.startswith(("ds_write", "ds_store")))
    print(f"{L[si][:24]:24s} loop of {len(ins):4d} instructions: VALU {valu:4d} = packed {pk:3d} + transcendental {tr:3d} + v_med3 {med:3d} + other {valu - pk - tr - med:3d}; "
          f"ds_read {dsr:3d}, ds_write {dsw:3d}, s_barrier {c.get('s_barrier', 0)}, s_nop {c.get('s_nop', 0)}")
