#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a gfx950 assembly listing.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -I../../include peps_amd/csrc/capi.hip -o /tmp/capi.s
    python scripts/isa_loops.py /tmp/capi.s jacobi_rows_grp_kernelILi2ELi16 [n_loops]

For every back edge of the kernel (a branch to an earlier label) the instructions between the label and the branch are counted:
total, VALU, plain register moves (v_mov_b32 / v_mov_b64 that are not DPP moves).  This is how the two findings of round 4 on the
Jacobi tournament kernels were made: a third of the VALU instructions of the hot loops were register moves -- pairs
gathered in front of every packed instruction, copies at branch joins, the turn of the rows in rolled loops -- and every DPP
reduction step took three instructions instead of one.  No GPU needed."""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 6
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^(_Z\w+):", l) and pat in l)
end = start
while not lines[end].startswith(".Lfunc_end"):
    end += 1
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
print(lines[start][:100], "lines", len(body), "loops", len(loops))
for a, b in sorted(loops, key=lambda x: x[0] - x[1])[:top]:
    c = collections.Counter()
    for l in body[a:b + 1]:
        m = re.match(r"^\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|scratch_\w+)(.*)", l)
        if m:
            op = m.group(1)
            if "dpp" in op or "row_" in m.group(2) or "quad_perm" in m.group(2):
                op = "DPP:" + op
            c[op] += 1
    tot = sum(c.values())
    valu = sum(v for o, v in c.items() if o.startswith("v_") or o.startswith("DPP"))
    mov = sum(v for o, v in c.items() if o.startswith("v_mov_b"))
    print(" loop lines %d-%d: %d insts, valu %d, plain v_mov %d (%.1f%% of valu)" % (a, b, tot, valu, mov, 100 * mov / max(valu, 1)))
    print("    ", ", ".join("%s %d" % (o, v) for o, v in c.most_common(14)))
