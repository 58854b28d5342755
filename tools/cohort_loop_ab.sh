#!/bin/bash
# loop A/B of the cohort-on-tickets experiment (experiments library): python bench.py --steps 6 at flags 29 vs 61
cd ${GRAFT_REPO_ROOT:-/root/repo}
export JENGA_LIB=$PWD/jenga_amd/libjenga_amd_exp.so JENGA_COHORT_QUORUM=32 JENGA_COHORT_TIMEOUT_US=60
O=gpurun_out/r04; mkdir -p $O
L="--steps 6 --no-cpu-baseline --no-dense-ref --no-secondary --no-wan-extra --no-rotate-ref"
for t in "29 a" "61 a" "29 b" "61 b"; do set -- $t; JENGA_ATTN_FLAGS=$1 timeout 600 python bench.py $L > $O/AN_loop_$1$2.json 2> $O/AN_loop_$1$2.err; python - $O/AN_loop_$1$2.json $1$2 <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
print("loop", sys.argv[2], d["value"], d["config"]["ms_per_class"], d["roofline"]["frac"])
PY
done
