#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc --save-temps .s file (static counts; a loop body is the
block(s) between a label and the backward branch to it).  Usage: isa_blocks.py file.s kernel_substring [min_insts]"""
import re, sys
from collections import Counter

path, pat = sys.argv[1], sys.argv[2]
min_insts = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    if re.match(r"^_Z\w*:", l) and pat in l:
        start = i
        break
assert start is not None, "kernel not found"
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:]:
    s = l.strip()
    if s.startswith(".Lfunc_end") or s.startswith("s_endpgm") and False:
        break
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        blocks.append((name, cur)); cur, name = [], m.group(1); continue
    if not s or s.startswith(";") or s.startswith("."):
        continue
    cur.append(s.split(";")[0].strip())
blocks.append((name, cur))

def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_exp") or op.startswith("v_rcp") or op.startswith("v_log") or op.startswith("v_rsq") or op.startswith("v_sqrt"): return "trans"
    if op.startswith("v_mul_lo") or op.startswith("v_mul_hi") or op.startswith("v_mad_u64") or op.startswith("v_mad_i64"): return "imul"
    if op.startswith("v_cvt"): return "cvt"
    if op.startswith("v_cndmask") or op.startswith("v_cmp"): return "cmp/sel"
    if op.startswith("v_accvgpr") : return "acc_mov"
    if op.startswith("v_mov") : return "v_mov"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_nop"): return "nop"
    return "salu"

tot = Counter()
for name, ins in blocks:
    c = Counter(cls(i.split()[0]) for i in ins)
    tot.update(c)
    if len(ins) < min_insts: continue
    tgt = [i.split()[-1] for i in ins if i.startswith("s_cbranch") or i.startswith("s_branch")]
    order = ["mfma", "valu", "cvt", "cmp/sel", "trans", "imul", "v_mov", "acc_mov", "lds", "vmem", "smem", "salu", "waitcnt", "barrier", "nop", "branch"]
    print("%-12s n=%5d  " % (name, len(ins)) + " ".join("%s=%d" % (k, c[k]) for k in order if c[k]) + ("   -> " + ",".join(tgt) if tgt else ""))
print("TOTAL", dict(tot))
