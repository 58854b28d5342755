#!/bin/bash
# Census of the operand selections of v_pk_{add,mul,fma}_f32 in torch's OWN gfx950 kernels (review item 4, round 6).  CPU only.
#
#   tools/torch_pk_census.sh OUT.json
#       disassembles every gfx950 code object of libtorch_hip.so and classifies, per kernel and per source operand, the
#       (op_sel, op_sel_hi) pair of every packed fp32 instruction:
#         normal    (0,1)  low result lane reads the low half, high lane the high half
#         bcast_lo  (0,0)  both lanes read the LOW half           (op_sel_hi:[..0..])
#         bcast_hi  (1,1)  both lanes read the HIGH half          (op_sel:[..1..])         <- the round-5 reproducer's failing forms
#         swap      (1,0)  low lane reads the HIGH half, high lane the low half           <- fails in the reproducer too (round 6)
#       OUT.json = {"kernels": {mangled name: {"<insn> src<i> <kind>": count}}, "totals": {...}}; kernels with normal selections
#       only are left out.
#   tools/torch_pk_census.sh --intersect OUT.json kernel_stats.csv
#       which kernels of a rocprofv3 --kernel-trace --stats CSV are in the census, with their selections; the last line counts
#       those with a LOW-LANE-READS-HIGH-HALF selection (bcast_hi or swap) -- every wrong result observed so far, in this
#       library's kernels, the stand-alone reproducer and torch's kernels alike, comes from an instruction with one.
#
# libtorch_hip.so carries compressed clang offload bundles (CCOB, zstd); clang-offload-bundler unpacks the gfx950 member of each.
set -e
LLVM=/opt/rocm/lib/llvm/bin
if [ "$1" = "--intersect" ]; then
  python3 - "$2" "$3" <<'PY'
import csv, gzip, json, subprocess, sys
cen = json.load(gzip.open(sys.argv[1], "rt") if sys.argv[1].endswith(".gz") else open(sys.argv[1]))["kernels"]
mangled = list(cen)
dem = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True).stdout.splitlines()
d2m = dict(zip(dem, mangled))
rows = list(csv.DictReader(open(sys.argv[2])))
hits = risky = 0
for r in rows:
    m = d2m.get(r["Name"])
    if not m:
        continue
    hits += 1
    bad = {k: v for k, v in cen[m].items() if k.endswith(("bcast_hi", "swap"))}
    risky += bool(bad)
    print(("LOW-READS-HIGH " if bad else "bcast_lo only  "), r.get("Calls"), r["Name"][:150], json.dumps(cen[m]))
print(f"# {len(rows)} kernel names in {sys.argv[2]}: {hits} use a non-default packed-fp32 operand selection, {risky} of them one where "
      "a low result lane reads a high source half")
PY
  exit 0
fi
OUT=${1:-torch_pk_selections.json}
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
  [ -s $W/$n.co ] && $LLVM/llvm-objdump -d $W/$n.co 2>/dev/null | grep -E "^[0-9a-f]+ <|v_pk_(add|mul|fma)_f32" | sed 's#//.*##' > $W/dis/$n.txt
  rm -f $W/$n.co; }
export -f one; export W LLVM
ls $W/b/*.bundle | xargs -P 8 -n 1 bash -c 'one "$0"'
python3 - $W "$OUT" <<'PY'
import collections, glob, json, re, sys
w, out = sys.argv[1:3]
def sel(line, nsrc):
    m = re.search(r"op_sel:\[([0-9,]+)\]", line); lo = [int(x) for x in m.group(1).split(",")] if m else [0] * nsrc
    m = re.search(r"op_sel_hi:\[([0-9,]+)\]", line); hi = [int(x) for x in m.group(1).split(",")] if m else [1] * nsrc
    lo += [0] * (nsrc - len(lo)); hi += [1] * (nsrc - len(hi))
    return list(zip(lo[:nsrc], hi[:nsrc]))
KIND = {(0, 1): "normal", (1, 1): "bcast_hi", (0, 0): "bcast_lo", (1, 0): "swap"}
per, tot = collections.defaultdict(collections.Counter), collections.Counter()
for f in glob.glob(w + "/dis/*.txt"):
    k = None
    for l in open(f):
        if re.match(r"^[0-9a-f]+ <", l):
            k = l.split("<", 1)[1].rsplit(">", 1)[0]; continue
        op = l.split()[0]
        for i, s in enumerate(sel(l, 3 if "fma" in op else 2)):
            if KIND[s] != "normal":
                per[k][f"{op} src{i} {KIND[s]}"] += 1; tot[f"{op} src{i} {KIND[s]}"] += 1
risky = sum(1 for v in per.values() if any(k.endswith(("bcast_hi", "swap")) for k in v))
json.dump({"what": "operand selections of v_pk_{add,mul,fma}_f32 in the gfx950 kernels of libtorch_hip.so (tools/torch_pk_census.sh)",
           "kernels_with_a_non_default_selection": len(per), "kernels_with_a_low_lane_reads_high_half_selection": risky,
           "totals": dict(sorted(tot.items())), "kernels": {k: dict(v) for k, v in per.items()}}, open(out, "w"))
print(f"{len(per)} kernels with a non-default selection, {risky} with a low-reads-high one -> {out}")
PY
rm -rf $W
