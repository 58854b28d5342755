#!/usr/bin/env python3
"""Count scratch (spill) instructions inside loops of a kernel in hipcc's -S output.
  hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only x.hip -o x.s ; python tools/isa_loop_spills.py x.s <substring of kernel name>
A loop = the span between a label and a later branch back to it.  Prints per kernel: total scratch ops, those inside any
loop, and the loops (first line, length, MFMAs, scratch ops) that contain some."""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
    text = open(path).read()
    for m in re.finditer(r"^(\S+):\s*; @\1\n", text, re.M):
        name = m.group(1)
        if pat not in name:
            continue
        end = text.index(".Lfunc_end", m.end())
        lines = [re.sub(r";.*", "", l).strip() for l in text[m.end():end].splitlines()]
        lines = [l for l in lines if l]
        labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
        loops = []
        for i, l in enumerate(lines):
            b = re.match(r"s_c?branch\S*\s+(\S+)", l)
            if b and b.group(1) in labels and labels[b.group(1)] < i:
                loops.append((labels[b.group(1)], i))
        in_loop = set()
        for a, b in loops:
            in_loop.update(range(a, b + 1))
        scr = [i for i, l in enumerate(lines) if l.startswith("scratch_")]
        print(name, "instructions", len(lines), "scratch", len(scr), "in loops", sum(i in in_loop for i in scr))
        for a, b in loops:
            n = sum(1 for i in scr if a <= i <= b)
            if n:
                mf = sum(1 for l in lines[a:b + 1] if "mfma" in l)
                print(f"   loop at {a} len {b - a + 1} mfma {mf} scratch {n}")


if __name__ == "__main__":
    main()
