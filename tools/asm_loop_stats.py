#!/usr/bin/env python3
"""Instruction census of a kernel's hot loop from hipcc -S output.

usage: asm_loop_stats.py file.s <kernel-name-substring> [--dump]
Finds the kernel, splits it into basic blocks at labels, takes the back-edge loop that holds the most
v_mfma instructions and prints counts per instruction class (VALU by opcode, MFMA, LDS, SALU, waits).
"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(key.split()[-1]) or (l.startswith("_Z") and key in l.split(":")[0]))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    # label positions
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    # back edges: branch to a label defined earlier
    best = None
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if not m:
            continue
        tgt = m.group(1) or m.group(2)
        if tgt in labels and labels[tgt] < i:
            seg = body[labels[tgt]:i + 1]
            nm = sum(1 for s in seg if "v_mfma" in s)
            if best is None or nm > best[0]:
                best = (nm, labels[tgt], i)
    nm, a, b = best
    seg = [s.strip() for s in body[a:b + 1]]
    ops = collections.Counter()
    for s in seg:
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        op = s.split()[0]
        ops[op] += 1
    valu = {k: v for k, v in ops.items() if k.startswith("v_") and not k.startswith("v_mfma")}
    print(f"loop lines {a}..{b} of kernel; MFMA {nm}")
    print("VALU total", sum(valu.values()))
    for k, v in sorted(valu.items(), key=lambda kv: -kv[1]):
        print(f"  {k:28s} {v}")
    other = {k: v for k, v in ops.items() if k not in valu and not k.startswith("v_mfma")}
    print("other:")
    for k, v in sorted(other.items(), key=lambda kv: -kv[1]):
        print(f"  {k:28s} {v}")
    if dump:
        print("\n".join(seg))


main()
