#!/usr/bin/env python3
"""Hazard lint for MFMAs written as inline asm (csrc/lp_core.h, LP_QK_MFMA_ASM), on hipcc's -S output.

hipcc's hazard recogniser does not look into asm statements, so for the QK^T MFMAs of the pair kernel nobody inserts the wait
states the ISA wants around them.  Those MFMAs are recognisable in the listing: their destination is a VGPR tuple (every builtin
MFMA of that kernel is in AGPR form).  Two checks per such MFMA:
  producer:  no VALU / v_accvgpr instruction among the `need` instructions in front of it writes one of its source registers
             (a register-allocator copy right in front of the MFMA -- the bug that made the text rows of query block B wrong in
             the first build: B's Q fragments lived in VGPRs and were copied to AGPRs in front of every use)
  consumer:  the first non-MFMA instruction that reads its destination comes after at least two more MFMAs or `far` other
             instructions (the softmax reads the scores one basic block later; blocks without P.V MFMAs carry explicit s_nops)
python tools/isa_hazards.py x.s [kernel-name substring]"""
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import isa_loop_spills as T  # noqa: E402

_REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def regs(operand):
    out = set()
    for m in _REG.finditer(operand):
        lo = int(m.group(2) if m.group(2) is not None else m.group(3))
        hi = int(m.group(2) if m.group(2) is not None else m.group(4))
        out.update((m.group(1), i) for i in range(lo, hi + 1))
    return out


def split(line):
    op, _, rest = line.partition(" ")
    return op, [x.strip() for x in rest.split(",")] if rest else []


def wait_states(line):
    m = re.match(r"s_nop (\d+)", line)
    return int(m.group(1)) + 1 if m else 1


def check(lines, need=2, far=20):
    """-> list of (index, kind, mfma line, offending line)"""
    bad = []
    ins = [(i, l) for i, l in enumerate(lines) if not l.endswith(":")]
    for n, (i, l) in enumerate(ins):
        op, ops = split(l)
        if "mfma" not in op or not ops or not ops[0].startswith("v"):
            continue
        dst, srcs = regs(ops[0]), set().union(*(regs(o) for o in ops[1:])) if len(ops) > 1 else set()
        # producer check
        ws, k = 0, n - 1
        while k >= 0 and ws < need:
            _, pl = ins[k]
            pop, pops = split(pl)
            if (pop.startswith("v_") and "mfma" not in pop and not pop.startswith("v_cmp") and pops and regs(pops[0]) & srcs):
                bad.append((i, "producer", l, pl))
                break
            ws += wait_states(pl)
            k -= 1
        # consumer check (straight-line scan; stops at the first branch: blocks end with fences / waits anyway)
        mf, ws, k = 0, 0, n + 1
        while k < len(ins) and ws < far and mf < 2:
            _, cl = ins[k]
            cop, cops = split(cl)
            if cop.startswith("s_cbranch") or cop.startswith("s_branch") or cop.startswith("s_setpc") or cop.startswith("s_endpgm"):
                break
            if "mfma" in cop:
                mf += 1
            elif cop.startswith("v_") and cops and any(regs(o) & dst for o in cops[1:]):
                bad.append((i, "consumer", l, cl))
                break
            elif (cop.startswith("global_store") or cop.startswith("ds_write") or cop.startswith("scratch_store")) and \
                    any(regs(o) & dst for o in cops):
                bad.append((i, "consumer", l, cl))
                break
            ws += wait_states(cl)
            k += 1
    return bad


def check_m0(lines):
    """LDS-DMA (global_load_lds_*) takes its LDS base from M0.  With LP_DMA_M0_ONCE (lp_core.h) only the first piece of a
    four-piece stage writes M0; the other three rely on it.  Per straight-line stretch (labels are join points: M0 unknown
    behind them): every global_load_lds must follow an `s_mov_b32 m0, ...` of the same stretch with no other write to M0 in
    between, and at least one wait state behind that s_mov.  -> list of (index, line, reason)"""
    bad, valid, since = [], False, 0
    for i, l in enumerate(lines):
        if l.endswith(":"):
            valid = False
            continue
        op, ops = split(l)
        if op.startswith("global_load_lds") or op.startswith("buffer_load") and "lds" in l:
            if not valid:
                bad.append((i, l, "no M0 write in this straight-line stretch"))
            elif since < 1:
                bad.append((i, l, "no wait state between the M0 write and the LDS-DMA"))
            since += 1
            continue
        if ops and ops[0] == "m0":
            valid, since = op == "s_mov_b32", 0
            continue
        if op.startswith(("s_setpc", "s_swappc", "s_call")):
            valid = False
        since += wait_states(l)
    return bad


def main():
    path, pat = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
    for name, lines in T.kernels(open(path).read(), pat).items():
        bad = check(lines)
        n = sum(1 for l in lines if "mfma" in l and re.match(r"\S+ v", l))
        print(name[-60:], "asm-form MFMAs", n, "violations", len(bad))
        for i, kind, l, o in bad[:12]:
            print("   ", kind, "@", i, "|", l, "<-" if kind == "producer" else "->", o)
        bm = check_m0(lines)
        print("   LDS-DMA pieces", sum(1 for l in lines if l.startswith("global_load_lds")), "M0 violations", len(bm))
        for i, l, why in bm[:8]:
            print("    m0 @", i, "|", l, "|", why)


if __name__ == "__main__":
    main()
