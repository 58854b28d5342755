"""CPU: the C-ABI library builds, loads, and exports every symbol include/jenga_amd.h declares; host-side
argument validation; no compute (there is no GPU here)."""
import os
import re
import subprocess

import pytest
import torch

from jenga_amd import _capi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_loads():
    build.build()
    L = _capi.lib()
    assert L.jenga_abi_version() == 4


def _declared():
    """symbols declared by include/jenga_amd.h."""
    header = open(os.path.join(ROOT, "include", "jenga_amd.h")).read()
    assert "JENGA_EXPERIMENTS" not in header          # (round 5: no experiments section, no experiments library)
    return set(re.findall(r"\b(jenga_[a-z0-9_]+)\s*\(", header))


def _exported(path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path]).decode()
    return {ln.split()[-1] for ln in out.splitlines() if " T " in ln}


def test_header_symbols_exported():
    declared = _declared()
    exported = _exported(_capi.LIB_PATH)
    assert declared, "no declarations parsed"
    assert declared <= exported, f"missing: {declared - exported}"
    assert declared == set(_capi.SIGNATURES), "ctypes binding out of sync with the header"
    assert not os.path.isdir(os.path.join(ROOT, "jenga_amd", "csrc", "experiments"))


def test_library_is_rebuilt_when_a_source_changes(tmp_path, monkeypatch):
    """build.needs_build compares a content hash of csrc/ + the header + build.py with the stamp written next to the library
    (not mtimes): the prebuilt .so that travels with a snapshot is kept only if it was built from the sources beside it."""
    assert not build.needs_build()
    real = build.source_digest()
    monkeypatch.setattr(build, "source_digest", lambda: "0" * 64)
    assert build.needs_build()
    monkeypatch.setattr(build, "source_digest", lambda: real)
    assert not build.needs_build()


def test_no_cpu_fallback():
    q = torch.zeros(1, 256, 2, 128, dtype=torch.bfloat16)
    with pytest.raises(_capi.JengaError):
        _capi.block_pool(q, 2)
    from jenga_amd.modules.attention_block_sparse import block_sparse_attention
    cu = torch.tensor([0, 200, 256], dtype=torch.int32)
    with pytest.raises(_capi.JengaError):
        block_sparse_attention(q, q, q, top_k=1, cu_seqlens_q=cu, cu_seqlens_kv=cu, text_blocks=1)


def test_argument_errors_mirror_the_reference_contract():
    from jenga_amd.modules.attention_block_sparse import block_sparse_attention
    cu = torch.tensor([0, 200, 256], dtype=torch.int32)
    q32 = torch.zeros(1, 256, 2, 128)
    with pytest.raises(ValueError):  # dtype not in {bf16, fp16} (reference :167-170 would mis-type it silently)
        block_sparse_attention(q32, q32, q32, top_k=1, cu_seqlens_q=cu, cu_seqlens_kv=cu)
    q = torch.zeros(1, 200, 2, 128, dtype=torch.bfloat16)
    with pytest.raises((ValueError, _capi.JengaError)):  # S % 128 != 0: the reference's reshape would throw (:216)
        block_sparse_attention(q, q, q, top_k=1, cu_seqlens_q=cu, cu_seqlens_kv=cu)
    q64 = torch.zeros(1, 256, 2, 64, dtype=torch.bfloat16)
    with pytest.raises((ValueError, _capi.JengaError)):
        block_sparse_attention(q64, q64, q64, top_k=1, cu_seqlens_q=cu, cu_seqlens_kv=cu)


def test_pack_v_bytes():
    assert _capi.lib().jenga_pack_v_bytes(1, 24, 902) == 24 * 902 * 128 * 128 * 2


def test_header_is_c99_and_library_links_from_plain_c(tmp_path):
    """include/jenga_amd.h compiled as C99 (-Wall -Wextra -pedantic) by gcc, linked against libjenga_amd.so and run:
    the ABI is usable without C++ or Python on the calling side."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(root, "jenga_amd")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(root, "include"),
                        os.path.join(root, "tests", "c", "abi_smoke.c"), "-o", exe, "-L" + libdir, "-ljenga_amd",
                        "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok abi="), (r.returncode, r.stdout, r.stderr)


def test_ctypes_signatures_match_the_header_parameter_by_parameter():
    """Every entry of _capi.SIGNATURES against the C declaration in include/jenga_amd.h: same number of parameters and
    the same kind at every position (pointer / int64_t / int / float / size_t) -- a wrong ctypes signature shows up
    here, on the CPU, instead of as an ArgumentError on the GPU box."""
    import ctypes
    header = open(os.path.join(ROOT, "include", "jenga_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    kinds = {ctypes.c_void_p: "ptr", ctypes.c_char_p: "ptr", ctypes.c_int64: "i64", ctypes.c_int: "int",
             ctypes.c_float: "float", ctypes.c_size_t: "size", ctypes.c_double: "double"}

    def kind_of(param):
        param = param.strip()
        if "*" in param:
            return "ptr"
        t = param.rsplit(" ", 1)[0].replace("const", "").strip()
        return {"int64_t": "i64", "int": "int", "float": "float", "size_t": "size", "double": "double"}[t]

    sigs = dict(_capi.SIGNATURES)
    for name, (_res, args) in sigs.items():
        m = re.search(r"\b(?:int64_t|int|size_t|const char\s*\*)\s*" + name + r"\s*\(([^;]*?)\)\s*;", header, re.S)
        assert m, f"{name}: declaration not found"
        params = [p_ for p_ in m.group(1).split(",") if p_.strip() and p_.strip() != "void"]
        assert len(params) == len(args), f"{name}: header has {len(params)} parameters, ctypes {len(args)}"
        for i, (p_, a_) in enumerate(zip(params, args)):
            assert kind_of(p_) == kinds[a_], f"{name}: parameter {i} ({p_.strip()!r}) bound as {a_.__name__}"


def test_attention_flag_constants_match_the_header():
    """The Python flag constants of jenga_bsattn_fwd are the header's #defines (include/jenga_amd.h), and the modules'
    default is XCD remap + balanced launch + LP kernel (+ the Python-side kept-count order)."""
    import os
    import re
    from jenga_amd import _capi
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "jenga_amd.h")).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+JENGA_ATTN_(\w+)\s+(\d+)", hdr)}
    assert defs == {"XCD_REMAP": 1, "BALANCE": 4, "LP": 8}
    for name, val in defs.items():
        assert getattr(_capi, "ATTN_" + name) == val, name
    assert _capi.ATTN_SORTED == 16 and _capi.ATTN_PAIR == 64          # Python-side routing bits: not in the C header
    if "JENGA_ATTN_FLAGS" not in os.environ:
        assert _capi.ATTN_DEFAULT_FLAGS in (_capi.ATTN_LP_FLAGS, _capi.ATTN_PAIR_FLAGS)
    assert _capi.ATTN_LP_FLAGS == _capi.ATTN_XCD_REMAP | _capi.ATTN_BALANCE | _capi.ATTN_LP | _capi.ATTN_SORTED
    assert _capi.ATTN_PAIR_FLAGS == _capi.ATTN_XCD_REMAP | _capi.ATTN_BALANCE | _capi.ATTN_SORTED | _capi.ATTN_PAIR
