"""CPU (no GPU needed): the register allocation of the LP attention kernel is part of its performance contract.

csrc/bsattn3.hip sits at 256 VGPRs; its LDS-DMA pipeline waits with `s_waitcnt vmcnt(N > 0)`, and a spill reload in the
steady state of the unrolled main loop drains the DMA queue (`scratch_load` + `s_waitcnt vmcnt(0)`; measured -2.8 %).  Code
the compiler can take for a store in FRONT of the main loop -- an atomic, s_sleep, s_memrealtime -- is enough to cause one
("Why the ticket atomics are inline assembly").  This test cross-compiles the file to assembly with the
product flags and pins, for every instantiation the product library launches: no scratch instruction in the main loop's steady
state, and the spill count of the static-mapping kernel at the value the measurements were taken with."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    from jenga_amd import build
    hipcc = build._hipcc()
    flags = dict(build.SOURCES)["bsattn3.hip"]
    out = tmp_path_factory.mktemp("isa") / "bsattn3.s"
    cmd = [hipcc, f"--offload-arch={build.ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-S", "--cuda-device-only",
           os.path.join(ROOT, "jenga_amd", "csrc", "bsattn3.hip"), "-o", str(out)] + flags
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    return out.read_text()


def test_main_loops_of_the_attention_kernel_are_reload_free(asm):
    import isa_loop_spills as T
    ks = T.kernels(asm, "bsattn_lp_kernel")
    variants = {}
    for name, lines in ks.items():
        m = re.search(r"INS_4(BF16|FP16)ELi(\d)E", name)
        variants[(m.group(1), int(m.group(2)))] = lines
    # the product library: static mapping, cross-attention, balanced launch
    assert sorted({v for _, v in variants}) == [0, 2, 4]
    for (dt, v), lines in sorted(variants.items()):
        steady = T.main_loop_reloads(lines)
        if v == 2:
            continue        # (the cross-attention instantiation has no list walk: a shorter loop structure)
        assert steady is not None, (dt, v, "no 192-MFMA main loop found")
        assert steady == [], (dt, v, steady)


def test_static_mapping_kernel_keeps_its_measured_allocation(asm):
    spill = {}
    for m in re.finditer(r"\.name:\s+(\S*bsattn_lp_kernel\S*)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", asm):
        spill[m.group(1)] = int(m.group(2))
    assert spill, "no kernel metadata found"
    for name, n in spill.items():
        if "ELi0E" in name:
            assert n == 22, (name, n)      # 22 spilled VGPRs, all outside the unrolled steady state
        assert n <= 32, (name, n)


@pytest.fixture(scope="module")
def asm_pair(tmp_path_factory):
    from jenga_amd import build
    hipcc = build._hipcc()
    flags = dict(build.SOURCES)["bsattn5.hip"]
    out = tmp_path_factory.mktemp("isa5") / "bsattn5.s"
    cmd = [hipcc, f"--offload-arch={build.ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-S", "--cuda-device-only",
           os.path.join(ROOT, "jenga_amd", "csrc", "bsattn5.hip"), "-o", str(out)] + flags
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    return out.read_text()


def test_pair_kernel_main_loop_stays_inside_the_issue_budget(asm_pair):
    """csrc/bsattn5.hip runs ONE wave per SIMD: nothing covers an instruction that is not hidden in an MFMA's shadow, and a
    32-cycle `v_mfma_f32_32x32x16_bf16` hides about five single-issue instructions (MI355X_MICROARCH.md).  The hot path of
    the unrolled main loop of the shared segment (6 steps x 64 MFMAs; the exact-softmax blocks are cold) is pinned: no
    scratch access, no v_accvgpr move (the first builds had 4 per MFMA: scores, Q fragments and the -m~ operand bounced
    between the two halves of the register file), at most 5.0 instructions per MFMA besides the MFMA, one fragment read
    per two MFMAs, and the QK^T MFMAs in the form the kernel writes them (scores in VGPRs, Q in AGPRs)."""
    import isa_loop_spills as T
    ks = T.kernels(asm_pair, "bsattn_lq_kernel")
    assert len(ks) == 4, list(ks)           # bf16 / fp16 x static / balanced
    for name, lines in ks.items():
        best = None
        for a, b in T.loops_of(lines):
            n = sum("mfma" in x for x in lines[a:b + 1])
            if n >= 384 and (best is None or b - a < best[1] - best[0]):
                best = (a, b)
        assert best is not None, (name, "no 384-MFMA main loop found")
        hot = T.hot_path(lines[best[0]:best[1] + 1])
        ops = [x.split()[0] for x in hot]
        n_mfma = sum("mfma" in o for o in ops)
        assert n_mfma == 384, (name, n_mfma)
        assert not any(o.startswith("scratch_") for o in ops), name
        assert sum(o.startswith("v_accvgpr") for o in ops) <= 8, (name, sum(o.startswith("v_accvgpr") for o in ops))
        assert sum(o.startswith("ds_read_b128") for o in ops) == 192, name
        assert (len(ops) - n_mfma) / n_mfma <= 5.0, (name, (len(ops) - n_mfma) / n_mfma)
        qk = [x for x in hot if "mfma" in x and re.search(r"mfma\S*\s+v\[\d+:\d+\], v\[\d+:\d+\], a\[\d+:\d+\]", x)]
        assert len(qk) == 192, (name, len(qk))      # QK^T: D in VGPRs, A = K fragment in VGPRs, B = Q fragment in AGPRs


def test_pair_kernel_asm_mfmas_have_no_uncovered_hazard(asm_pair):
    """The QK^T MFMAs of csrc/bsattn5.hip are inline asm (lp_core.h, LP_QK_MFMA_ASM): hipcc inserts no wait states around
    them.  tools/isa_hazards.py checks every one of them in every instantiation: no VALU / v_accvgpr write to one of its
    sources among the two instructions in front of it (a register-allocator copy: the first build's wrong text rows), and
    no VALU read of its destination before two more MFMAs or 20 wait states have passed."""
    import isa_hazards as Hz
    import isa_loop_spills as T
    ks = T.kernels(asm_pair, "bsattn_lq_kernel")
    assert len(ks) == 4
    for name, lines in ks.items():
        n = sum(1 for l in lines if "mfma" in l and re.match(r"\S+ v", l))
        assert n >= 400, (name, n)          # the asm form is really there (scores in VGPRs)
        bad = Hz.check(lines)
        assert not bad, (name, bad[:4])


def test_lds_dma_always_follows_an_m0_write_of_its_own_stretch(asm, asm_pair):
    """LDS-DMA takes its LDS base from M0, which the helpers of lp_core.h write in their own asm statements.  The lint
    (tools/isa_hazards.py::check_m0) walks the generated code of both attention kernels: every global_load_lds follows an
    `s_mov_b32 m0` of the same straight-line stretch, nothing else writes M0 in between, and a wait state separates the two.
    (It is what makes the LP_DMA_M0_ONCE form -- one M0 write per four-piece stage, round 6, profiles/r06_attn_pair_diet.json --
    checkable at all; the product build writes M0 per piece and must pass as well.)  A hand-made violation is caught."""
    import isa_hazards as Hz
    import isa_loop_spills as T
    for text, pat in ((asm, "bsattn_lp_kernel"), (asm_pair, "bsattn_lq_kernel")):
        ks = T.kernels(text, pat)
        assert ks
        for name, lines in ks.items():
            assert sum(1 for l in lines if l.startswith("global_load_lds")) >= 16, name
            assert Hz.check_m0(lines) == [], (name, Hz.check_m0(lines)[:3])
    good = ["s_mov_b32 m0, s4", "s_nop 0", "global_load_lds_dwordx4 v1, s[2:3]", "v_add_f32 v0, v0, v1",
            "global_load_lds_dwordx4 v2, s[2:3] offset:1024"]
    assert Hz.check_m0(good) == []
    assert Hz.check_m0(good[:3] + [".LBB0_1:"] + good[4:])                       # a join point in between: M0 unknown
    assert Hz.check_m0(good[:3] + ["s_add_u32 m0, m0, 4"] + good[4:])            # somebody else writes M0
    assert Hz.check_m0(["s_mov_b32 m0, s4", "global_load_lds_dwordx4 v1, s[2:3]"])  # no wait state behind the write


@pytest.mark.parametrize("src", ["rowops.hip", "select.hip", "gilbert.hip", "bsattn.hip", "bsattn3.hip", "bsattn5.hip"])
def test_no_packed_fp32_arithmetic_in_the_device_code(tmp_path, src):
    """Round 5 finding (jenga_amd/build.py, profiles/r05_packed_fp32_under_gpu_sharing.json): kernels with compiler-packed fp32
    arithmetic (v_pk_mul_f32 / v_pk_add_f32 from the SLP vectoriser) returned wrong values on MI355X whenever another process
    kept the same GPU busy.  With the product flags no device source may contain a packed fp32 multiply / add / fma."""
    from jenga_amd import build
    flags = dict(build.SOURCES)[src]
    out = tmp_path / (src + ".s")
    cmd = [build._hipcc(), f"--offload-arch={build.ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-S", "--cuda-device-only",
           os.path.join(ROOT, "jenga_amd", "csrc", src), "-o", str(out)] + flags
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    text = out.read_text()
    hits = re.findall(r"^\s*(v_pk_(?:mul|add|fma)_f32)\b", text, re.M)
    assert not hits, f"{src}: {len(hits)} packed fp32 instructions"
    # stricter, and true today: no packed VALU arithmetic of any type and no operand-half selection at all (whether the 16-bit
    # packed forms share the problem is not known; v_cvt_pk_* conversions are not packed arithmetic)
    other = re.findall(r"^\s*(v_pk_\w+)", text, re.M)
    assert not other, f"{src}: packed VALU instructions {sorted(set(other))}"
    assert "op_sel" not in text, f"{src}: an instruction with op_sel"
