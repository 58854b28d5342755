#!/bin/bash
# A/B of alternative pair-kernel builds on one box: tools/gpu_ab.sh OUTDIR "bench_attn args" lib1 lib2 ...
# (lib "base" = the tree's library).  Prints one line per (lib) with the attention time.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/$1; ARGS=$2; shift; shift
mkdir -p $O
for lib in "$@"; do
  if [ "$lib" = base ]; then unset JENGA_LIB; else export JENGA_LIB=$PWD/alt_libs/$lib.so; fi
  timeout 300 python tools/bench_attn.py $ARGS > $O/$lib.json 2> $O/$lib.err || echo "$lib FAILED" 
done
python - "$O" "$@" <<'PY'
import json,sys
o=sys.argv[1]
for lib in sys.argv[2:]:
    try:
        d=json.load(open(f"{o}/{lib}.json")); print("%-16s ms=%7.2f TF=%6.0f frac=%.3f" % (lib, d["attn_ms"], d["attn_TFLOPs"], d["attn_frac_of_2.5PF"]))
    except Exception as e: print(lib, "ERR", e)
PY
