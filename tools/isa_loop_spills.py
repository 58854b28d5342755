#!/usr/bin/env python3
"""Scratch (spill) instructions inside loops of a kernel in hipcc's -S output.

  hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only x.hip -o x.s ; python tools/isa_loop_spills.py x.s [substring of kernel name]

A loop = the span between a label and a later branch back to it.  Prints per kernel: total scratch ops, those inside any
loop, and the loops (first line, length, MFMAs, scratch ops) that contain some.

Why it matters for the LP attention kernel (csrc/bsattn3.hip): it runs at 256 VGPRs, its LDS-DMA pipeline counts on
`s_waitcnt vmcnt(N)` with N > 0, and a `scratch_load` in the steady state of the unrolled main loop is followed by
`s_waitcnt vmcnt(0)` -- the DMA queue drains three times per 12 steps (measured: -2.8 %).  `main_loop_reloads` returns those
instructions: the scratch ops of the smallest loop holding `mfma_per_iteration` MFMAs that sit BEHIND its first MFMA (the ops
in front of it belong to the rarely taken list-window refill).  tests/test_isa_cpu.py pins them to zero for every product
instantiation."""
import re
import sys


def kernels(text, pat=""):
    """-> {name: [instruction lines]} of the kernels whose mangled name contains `pat`."""
    out = {}
    for m in re.finditer(r"^(\S+):\s*; @\1\n", text, re.M):
        name = m.group(1)
        if pat not in name:
            continue
        end = text.index(".Lfunc_end", m.end())
        lines = [re.sub(r";.*", "", l).strip() for l in text[m.end():end].splitlines()]
        out[name] = [l for l in lines if l]
    return out


def loops_of(lines):
    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    res = []
    far = None      # target of a long jump being assembled (s_getpc / s_add_u32 (.Ltarget-.Lpost_getpc) / s_setpc_b64)
    for i, l in enumerate(lines):
        b = re.match(r"s_c?branch\S*\s+(\S+)", l)
        if b and b.group(1) in labels and labels[b.group(1)] < i:
            res.append((labels[b.group(1)], i))
        f = re.match(r"s_add_u32 \S+ \S+ \((\.LBB\S+)-\.Lpost_getpc\d+\)", l.replace(",", ""))
        if f:
            far = f.group(1)
        if l.startswith("s_setpc_b64") and far is not None:
            if far in labels and labels[far] < i:
                res.append((labels[far], i))
            far = None
    return res


def main_loop_reloads(lines, mfma_per_iteration=192):
    """Scratch instructions in the steady state of the main loop (see the module docstring); None if there is no such loop."""
    best = None
    for a, b in loops_of(lines):
        if sum("mfma" in x for x in lines[a:b + 1]) == mfma_per_iteration and (best is None or b - a < best[1] - best[0]):
            best = (a, b)
    if best is None:
        return None
    a, b = best
    first = next(t for t in range(a, b) if "mfma" in lines[t])
    return [(t - a, lines[t]) for t in range(first, b + 1) if lines[t].startswith("scratch_")]


def hot_path(lines, cold_marker="v_ceil_f32"):
    """The instructions of `lines` (one loop) outside the basic blocks that contain `cold_marker` -- for the attention
    kernels: the exact max-first softmax blocks (the only users of ceilf), entered on a wave-wide ballot."""
    bbs, cur = [], []
    for x in lines:
        if x.endswith(":"):
            if cur:
                bbs.append(cur)
            cur = []
        else:
            cur.append(x)
            if x.startswith("s_cbranch") or x.startswith("s_branch") or x.startswith("s_setpc"):
                bbs.append(cur)
                cur = []
    if cur:
        bbs.append(cur)
    return [x for bb in bbs if not any(cold_marker in y for y in bb) for x in bb]


def main():
    path, pat = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
    for name, lines in kernels(open(path).read(), pat).items():
        loops = loops_of(lines)
        in_loop = set()
        for a, b in loops:
            in_loop.update(range(a, b + 1))
        scr = [i for i, l in enumerate(lines) if l.startswith("scratch_")]
        steady = main_loop_reloads(lines)
        print(name, "instructions", len(lines), "scratch", len(scr), "in loops", sum(i in in_loop for i in scr),
              "main-loop steady state", "n/a" if steady is None else len(steady))
        for a, b in loops:
            n = sum(1 for i in scr if a <= i <= b)
            if n:
                mf = sum(1 for l in lines[a:b + 1] if "mfma" in l)
                print(f"   loop at {a} len {b - a + 1} mfma {mf} scratch {n}")


if __name__ == "__main__":
    main()
