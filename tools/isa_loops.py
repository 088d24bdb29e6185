#!/usr/bin/env python3
"""Loops of a gfx950 code object and what sits inside them (no GPU needed): every backward branch of the disassembly is a loop
[target, branch]; per loop the instruction count, scratch (spill) accesses, LDS accesses, global accesses, fp64 VALU and
transcendental instructions.  Where the spills of a kernel are matters more than how many there are: a spill in straight-line
set-up code costs one access per parcel, one inside the node loop one per node.
usage: python tools/isa_loops.py file.co [kernel-name-substring]"""
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def main():
    co = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    txt = subprocess.run([OBJDUMP, "-d", co], capture_output=True, text=True, check=True).stdout
    kernels, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^([0-9a-f]+) <([^>]+)>:", line)
        if m:
            cur = m.group(2)
            kernels[cur] = []
            continue
        m = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and cur is not None:
            kernels[cur].append((int(m.group(3), 16), m.group(1), line))
    for name, ins in kernels.items():
        if want not in name or not ins:
            continue
        addr_idx = {a: i for i, (a, _, _) in enumerate(ins)}
        loops = []
        for i, (a, op, args) in enumerate(ins):
            if op.startswith(("s_cbranch", "s_branch")):
                m = re.search(r"<[^>+]+\+0x([0-9a-f]+)>", args)
                if not m:
                    continue
                # objdump prints the target as <kernel+0xOFF>
                tgt = ins[0][0] + int(m.group(1), 16)
                if tgt in addr_idx and addr_idx[tgt] <= i:
                    loops.append((addr_idx[tgt], i))
        cat = lambda op: ("scratch" if op.startswith("scratch_") else "lds" if op.startswith("ds_") else
                          "global" if op.startswith(("global_", "buffer_", "flat_")) else
                          "trans" if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op) else
                          "f64" if op.endswith("_f64") or "_f64_" in op else "other")
        tot = {}
        for _, op, _ in ins:
            tot[cat(op)] = tot.get(cat(op), 0) + 1
        print(f"{name}: {len(ins)} instructions, " + ", ".join(f"{k} {v}" for k, v in sorted(tot.items())))
        loops.sort(key=lambda ab: (ab[0], -ab[1]))
        for a, b in loops:
            depth = sum(1 for c, d in loops if c <= a and d >= b and (c, d) != (a, b))
            cnt = {}
            for _, op, _ in ins[a:b + 1]:
                cnt[cat(op)] = cnt.get(cat(op), 0) + 1
            ld = sum(1 for _, op, _ in ins[a:b + 1] if op.startswith("scratch_load"))
            st = sum(1 for _, op, _ in ins[a:b + 1] if op.startswith("scratch_store"))
            print(f"  {'  ' * depth}loop [{a:6d}, {b:6d}] {b - a + 1:6d} instr: scratch {ld} ld / {st} st, lds {cnt.get('lds', 0)}, "
                  f"global {cnt.get('global', 0)}, f64 {cnt.get('f64', 0)}, trans {cnt.get('trans', 0)}")


if __name__ == "__main__":
    main()
