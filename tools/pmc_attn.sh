#!/bin/bash
# PMC passes for the attention kernel (run on the GPU box through gpurun).  Counters only, no tracing domains.
set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
ARGS="${@:---iters 2}"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/tools/bench_attn.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/p1 -o p1 -- python $R/tools/bench_attn.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/p2 -o p2 -- python $R/tools/bench_attn.py $ARGS > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/p3 -o p3 -- python $R/tools/bench_attn.py $ARGS > $OUT/p3.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -d $OUT/p4 -o p4 -- python $R/tools/bench_attn.py $ARGS > $OUT/p4.log 2>&1
ls -R $OUT | head -50
