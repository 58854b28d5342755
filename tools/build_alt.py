#!/usr/bin/env python3
"""A/B build of the library with extra flags on ONE source: tools/build_alt.py NAME SOURCE "extra flags" -> alt_libs/NAME.so
(the flags of jenga_amd/build.py for that source + the extra ones; the other objects are the tree's: build the library first).
Load it with JENGA_LIB=alt_libs/NAME.so."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jenga_amd import build as B  # noqa: E402

name, src, extra = sys.argv[1], sys.argv[2], sys.argv[3].split() if len(sys.argv) > 3 else []
flags = dict(B.SOURCES)[src]
stem = src.rsplit(".", 1)[0]
os.makedirs(os.path.join(ROOT, "alt_libs"), exist_ok=True)
alt = os.path.join(B.HERE, "build", "alt")
os.makedirs(alt, exist_ok=True)
obj = os.path.join(alt, f"{stem}_{name}.o")
subprocess.check_call([B._hipcc(), f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c",
                       os.path.join(B.CSRC, src), "-o", obj] + flags + extra)
objs = [os.path.join(B.HERE, "build", os.path.basename(s).rsplit(".", 1)[0] + ".o") for s, _ in B.SOURCES if s != src]
out = os.path.join(ROOT, "alt_libs", name + ".so")
subprocess.check_call([B._hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", out] + objs + [obj, "-lhipblaslt"])
print("built", out)
