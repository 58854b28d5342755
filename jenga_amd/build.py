"""Build libjenga_amd.so (gfx950) in-tree with hipcc.  `python -m jenga_amd.build` or jenga_amd.build.build()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libjenga_amd.so")
ARCH = "gfx950"

# (source, extra flags).  rowops / select: the reference's eager arithmetic is "fp32 operation, THEN cast to the tensor
# dtype": no fused multiply-adds (-ffp-contract=off) and no v_fma_mix*_f16 -- hipcc otherwise folds `half(float(a) * b)`
# into one mixed-precision instruction that rounds the exact product ONCE, where torch rounds to fp32 first and to fp16
# second (found in round 2: 1e-4 of the fp16 RMSNorm outputs differed by an ulp from the reference for that reason).
EAGER = ["-ffp-contract=off", "-Xclang", "-target-feature", "-Xclang", "-fma-mix-insts"]
SOURCES = [
    ("capi.cpp", []),
    ("gemm.cpp", []),          # host code: hipBLASLt GEMMs with epilogues torch's front-end does not expose
    ("gilbert.hip", []),
    ("rowops.hip", EAGER),
    ("select.hip", EAGER),
    # no NaNs are produced on the attention path (masked logits are -inf, never inf-inf); without this flag every
    # fmaxf on an MFMA result is preceded by a canonicalising v_max.  Infinities stay honoured.
    ("bsattn.hip", ["-fno-honor-nans"]),
    # no SLP vectorisation: v_pk_add_f32 beside MFMAs costs more than the two scalar adds it replaces (guide, per-
    # instruction table); the softmax of the LP kernel is placed instruction by instruction into the MFMA gaps
    # -Wno-inline-asm: the LDS-DMA helpers write M0 and say so in their clobber lists (a compiler-generated M0 user
    # must not assume it survives); clang warns that M0 is a reserved register -- that is the point of declaring it
    ("bsattn3.hip", ["-fno-honor-nans", "-fno-slp-vectorize", "-Wno-inline-asm"]),
]
# measured-and-rejected attention kernels (the pair kernel, the 8-wave LP pair, the ping-pong variant inside
# bsattn.hip): built only into libjenga_amd_exp.so by `python -m jenga_amd.build --experiments` (= every product source
# compiled with -DJENGA_EXPERIMENTS + these); `JENGA_LIB=.../libjenga_amd_exp.so` selects it (tests/test_gpu_pair.py)
EXPERIMENT_SOURCES = [
    ("experiments/bsattn2.hip", ["-fno-honor-nans", "-fno-slp-vectorize", "-Wno-inline-asm"]),
    ("experiments/bsattn4.hip", ["-fno-honor-nans", "-fno-slp-vectorize", "-Wno-inline-asm"]),
]
LIB_EXP = os.path.join(HERE, "libjenga_amd_exp.so")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(d, f) for d, _, fs in os.walk(CSRC) for f in fs] + [
        os.path.join(HERE, "..", "include", "jenga_amd.h"), os.path.abspath(__file__)]   # (the flags live here)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, experiments=False):
    lib = LIB_EXP if experiments else LIB
    if not force and not needs_build(lib):
        return lib
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build", "exp" if experiments else "")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src, extra in SOURCES + (EXPERIMENT_SOURCES if experiments else []):
        obj = os.path.join(objdir, os.path.basename(src).rsplit(".", 1)[0] + ".o")
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c",
               os.path.join(CSRC, src), "-o", obj] + extra + (["-DJENGA_EXPERIMENTS"] if experiments else []) \
            + os.environ.get("JENGA_HIPCC_FLAGS", "").split()
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out:
            print(out.decode(), file=sys.stderr)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib] + objs + ["-lhipblaslt"]
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, experiments="--experiments" in sys.argv))
