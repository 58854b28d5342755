#!/usr/bin/env python3
"""Static look at a hipcc -save-temps .s file: per basic block, the instruction classes between MFMAs.
  python tools/isa_stream.py file.s [--min-mfma 8]
Prints for every block with >= min-mfma MFMAs: counts by class and the 'gap' histogram (non-MFMA instructions
issued between consecutive MFMAs), which is what the one-wave-per-SIMD budget (<= 5 fillers per gap) is about."""
import re
import sys
from collections import Counter


def klass(op):
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith("v_accvgpr"): return "ACC"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "TRANS"
    if op.startswith("v_"): return "VALU"
    if op.startswith("ds_"): return "DS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "VMEM"
    if op.startswith("s_waitcnt"): return "WAIT"
    if op.startswith("s_nop"): return "NOP"
    if op.startswith("s_barrier"): return "BAR"
    if op.startswith("s_"): return "SALU"
    return "OTHER"


def main():
    path = sys.argv[1]
    min_mfma = int(sys.argv[sys.argv.index("--min-mfma") + 1]) if "--min-mfma" in sys.argv else 8
    dump = "--dump" in sys.argv
    blocks, cur, name = [], [], "entry"
    for line in open(path):
        s = line.strip()
        if not s or s.startswith((";", "//", ".")) and not re.match(r"^\.?LBB\d+_\d+:", s):
            if re.match(r"^\.?LBB\d+_\d+:", s):
                pass
            else:
                continue
        m = re.match(r"^(\.?LBB\d+_\d+|[_A-Za-z0-9]+):", s)
        if m:
            if cur: blocks.append((name, cur))
            name, cur = m.group(1), []
            continue
        op = s.split()[0]
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            cur.append(op)
            blocks.append((name, cur))
            name, cur = name + "'", []
            continue
        cur.append(op)
    if cur: blocks.append((name, cur))
    for name, ins in blocks:
        ks = [klass(o) for o in ins]
        n = ks.count("MFMA")
        if n < min_mfma: continue
        c = Counter(ks)
        gaps, g = [], 0
        seen = False
        for k in ks:
            if k == "MFMA":
                if seen: gaps.append(g)
                seen, g = True, 0
            elif seen:
                g += 1
        print(f"{name}: {len(ins)} instrs  " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
        print("   gaps between MFMAs:", " ".join(map(str, gaps)))
        if dump:
            print("   " + " ".join({"MFMA": "M", "VALU": "v", "TRANS": "t", "DS": "d", "VMEM": "g", "WAIT": "w", "NOP": "n",
                                     "SALU": "s", "ACC": "a", "BAR": "B", "OTHER": "?"}[k] for k in ks))


if __name__ == "__main__":
    main()
