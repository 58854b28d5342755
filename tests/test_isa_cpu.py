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
    # the product library: static mapping, cross-attention, rotated walk, balanced launch, balanced + rotated
    assert sorted({v for _, v in variants}) == [0, 2, 3, 4, 5]
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
