#!/bin/bash
# Census of `v_pk_{add,mul,fma}_f32 ... op_sel:[0,1...]` in torch's OWN gfx950 kernels (review item 4, round 6).  CPU only.
#   tools/torch_pk_census.sh [out.txt]                 -> "<count> <mangled kernel>" per kernel that contains the form
#   tools/torch_pk_census.sh --intersect list.txt kernel_stats.csv   -> which kernels of a rocprofv3 --stats trace are on the list
# libtorch_hip.so carries compressed clang offload bundles (CCOB, zstd); clang-offload-bundler unpacks the gfx950 member of each.
set -e
LLVM=/opt/rocm/lib/llvm/bin
if [ "$1" = "--intersect" ]; then
  python3 - "$2" "$3" <<'PY'
import csv, subprocess, sys
lst, stats = sys.argv[1], sys.argv[2]
rows = [l.split(None, 1) for l in open(lst) if l.strip()]
mangled = [r[1].strip().strip("<>:") for r in rows]
dem = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True).stdout.splitlines()
names = {r["Name"]: r for r in csv.DictReader(open(stats))}
hits = []
for n, r in names.items():
    for d, (c, _) in zip(dem, rows):
        if d == n or (len(n) > 400 and d[:400] == n[:400]):
            hits.append((int(c), r.get("Calls"), n[:230])); break
for h in sorted(hits, reverse=True): print(*h)
print(f"# {len(hits)} of {len(names)} kernel names of {stats} contain the form")
PY
  exit 0
fi
OUT=${1:-torch_pk_opsel_kernels.txt}
LIB=$(python3 -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib','libtorch_hip.so'))")
W=$(mktemp -d)
$LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin "$LIB" $W/fatbin.bin
python3 - $W <<'PY'
import struct, sys, os
w = sys.argv[1]; d = open(w + "/fatbin.bin", "rb").read(); os.makedirs(w + "/b"); pos = n = 0
while True:
    i = d.find(b"CCOB", pos)
    if i < 0: break
    ver, meth, fsize, usize = struct.unpack_from("<HHII", d, i + 4)
    open(f"{w}/b/{n:04d}.bundle", "wb").write(d[i:i + fsize]); n += 1; pos = i + fsize
PY
mkdir -p $W/dis
one() { f=$1; n=$(basename $f .bundle)
  $LLVM/clang-offload-bundler --unbundle --type=o --input=$f --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$W/$n.co 2>/dev/null || return 0
  [ -s $W/$n.co ] && $LLVM/llvm-objdump -d $W/$n.co 2>/dev/null | grep -E "^[0-9a-f]+ <|v_pk_(add|mul|fma)_f32.*op_sel:\[0,1" |
    awk '/^[0-9a-f]+ </{k=$2; next} {c[k]++} END{for(x in c) print c[x], x}' > $W/dis/$n.txt
  rm -f $W/$n.co; }
export -f one; export W LLVM
ls $W/b/*.bundle | xargs -P 8 -n 1 bash -c 'one "$0"'
cat $W/dis/*.txt | sort -rn > "$OUT"
echo "$(wc -l < "$OUT") kernels with the form -> $OUT"
rm -rf $W
