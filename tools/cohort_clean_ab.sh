#!/bin/bash
# Experiment (round 4, end): the cohort start barrier with a reload-free main loop (opaque counter / clock accesses), on the
# static mapping (flags 57) and on drawn blocks (61 = balanced launch + cohort), against static (25) and balanced (29).
#   python -m jenga_amd.build --experiments ; bash tools/cohort_clean_ab.sh      (on the GPU box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export JENGA_LIB=$PWD/jenga_amd/libjenga_amd_exp.so
one() { tag=$1; fl=$2; shift 2; env "$@" timeout 300 python tools/bench_attn.py --drop 0.7 --iters 40 --attn-only --flags $fl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['flags'], round(d['attn_TFLOPs'],1))"; }
timeout 300 python -m pytest tests/test_gpu_order.py -q -m gpu -k cohort 2>&1 | tail -1
one static 25 A=1
one balanced 29 A=1
one cohort_q32_t60 57 JENGA_COHORT_QUORUM=32 JENGA_COHORT_TIMEOUT_US=60
one balanced_cohort_q32_t60 61 JENGA_COHORT_QUORUM=32 JENGA_COHORT_TIMEOUT_US=60
one balanced_cohort_q16_t60 61 JENGA_COHORT_QUORUM=16 JENGA_COHORT_TIMEOUT_US=60
one balanced_cohort_q48_t100 61 JENGA_COHORT_QUORUM=48 JENGA_COHORT_TIMEOUT_US=100
one balanced_cohort_q64_t300 61 A=1
one balanced2 29 A=1
one balanced_cohort_q32_t60_2 61 JENGA_COHORT_QUORUM=32 JENGA_COHORT_TIMEOUT_US=60
one static2 25 A=1
