#!/bin/bash
# Collects SQ counters for the render kernel in separate rocprofv3 passes.
# usage: tools/pmc.sh <outdir> -- <python args for quick_bench.py>
out=$1; shift; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {
  name=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $R/$out/$name -- python3 $R/tools/quick_bench.py $ARGS > $R/$out/$name.log 2>&1 || return 1
}
ARGS="$BENCH_ARGS"
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE &&
run p2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA &&
run p3 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS &&
run p4 FETCH_SIZE &&
run p5 WRITE_SIZE &&
run p6 SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES
