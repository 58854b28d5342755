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
# -fno-slp-vectorize (round 5).  On the MI355X boxes of this pool a packed fp32 VALU instruction whose LOW lane selects the HIGH
# half of a source pair (v_pk_add_f32 / v_pk_mul_f32 ... op_sel:[0,1]) returns wrong low-lane results (lanes 48..63) while
# ANOTHER wave on the GPU executes v_mfma_f32_16x16x32_bf16 -- another process or another stream; hipBLASLt's GEMMs are such a
# load.  The SLP vectoriser emits exactly that form for `x - mean` (the mean broadcast from the high half of a pair): ln_modulate,
# rmsnorm_rope and qk_norm_rope_pool were wrong in up to 99 % of their calls beside a GEMM and never alone, so no single-stream
# test saw it.  Stand-alone reproducer: tools/micro/pk_beside_mfma.hip; record: profiles/r05_packed_fp32_under_gpu_sharing.json;
# pinned by tests/test_isa_cpu.py (no packed fp32 in any device source) and tests/test_gpu_shared_device.py (behaviour).
# -packed-fp32-ops on top: no v_pk_*_f32 at all from the sources that do not ask for them explicitly (the vector combiner
# still produced 32 in rmsnorm_rows_kernel without it).  (hipcc repeats "not a recognized feature" for the HOST pass.)
NO_SLP = ["-fno-slp-vectorize"]
NO_PK = NO_SLP + ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
EAGER = ["-ffp-contract=off", "-Xclang", "-target-feature", "-Xclang", "-fma-mix-insts"] + NO_SLP
SOURCES = [
    ("capi.cpp", []),
    ("gemm.cpp", []),          # host code: hipBLASLt GEMMs with epilogues torch's front-end does not expose
    ("gilbert.hip", NO_PK),
    ("rowops.hip", EAGER + NO_PK[1:]),
    ("select.hip", EAGER + NO_PK[1:]),
    # no NaNs are produced on the attention path (masked logits are -inf, never inf-inf); without this flag every
    # fmaxf on an MFMA result is preceded by a canonicalising v_max.  Infinities stay honoured.
    ("bsattn.hip", ["-fno-honor-nans"] + NO_PK),
    # no SLP vectorisation: v_pk_add_f32 beside MFMAs costs more than the two scalar adds it replaces (guide, per-
    # instruction table); the softmax of the LP kernel is placed instruction by instruction into the MFMA gaps
    # -Wno-inline-asm: the LDS-DMA helpers write M0 and say so in their clobber lists (a compiler-generated M0 user
    # must not assume it survives); clang warns that M0 is a reserved register -- that is the point of declaring it
    ("bsattn3.hip", ["-fno-honor-nans", "-fno-slp-vectorize", "-Wno-inline-asm"]),
    # the pair kernel (round 5): same flags; its QK^T MFMAs are inline asm (lp_core.h, LP_QK_MFMA_ASM)
    ("bsattn5.hip", ["-fno-honor-nans", "-fno-slp-vectorize", "-Wno-inline-asm"]),
]

def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _deps():
    return sorted(os.path.join(d, f) for d, _, fs in os.walk(CSRC) for f in fs) + [
        os.path.join(HERE, "..", "include", "jenga_amd.h"), os.path.abspath(__file__)]   # (the flags live here)


def _extra_flags():
    return os.environ.get("JENGA_HIPCC_FLAGS", "").split()


def source_digest():
    """sha256 over everything the library is built from: csrc/, the header, this file (the flags live here) AND the
    effective JENGA_HIPCC_FLAGS -- an elimination build (-DLQ_X_NODMA=1 ...: wrong results by design) must never pass for
    the product library on a later run."""
    import hashlib
    h = hashlib.sha256()
    h.update(("flags:" + " ".join(_extra_flags())).encode())
    for d in _deps():
        h.update(os.path.relpath(d, HERE).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def stamp_path(lib=LIB):
    """written next to the library it describes (git-ignored like it, ships to the GPU box with it)"""
    return lib + ".src.sha256"


STAMP = stamp_path()


def needs_build(lib=LIB):
    """True unless `lib` exists and was built from exactly the sources in the tree with exactly the current extra flags
    (content hash, not mtimes: a snapshot copied to another machine keeps its prebuilt library only if it really matches)."""
    stamp = stamp_path(lib)
    if not os.path.exists(lib) or not os.path.exists(stamp):
        return True
    with open(stamp) as f:
        return f.read().strip() != source_digest()


def build(force=False, verbose=False):
    lib = LIB
    if not force and not needs_build(lib):
        return lib
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src, extra in SOURCES:
        obj = os.path.join(objdir, os.path.basename(src).rsplit(".", 1)[0] + ".o")
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c",
               os.path.join(CSRC, src), "-o", obj] + extra + _extra_flags()
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
    with open(stamp_path(lib), "w") as f:
        f.write(source_digest() + "\n")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
