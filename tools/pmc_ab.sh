#!/bin/bash
# Vector-instruction counters of the render kernel for two builds of the library on one box.
# usage (on the GPU box): bash tools/pmc_ab.sh <outdir> libA.so libB.so [quick_bench args]
out=$1; A=$2; B=$3; shift; shift; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
for lib in $A $B; do
  timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE \
      --output-format csv -d $R/$out/$lib -- python3 $R/tools/quick_bench.py --lib $lib --reps 1 --modes strict "$@" > $R/$out/$lib.log 2>&1 || exit 1
  echo "== $lib"; tail -1 $R/$out/$lib.log; python3 $R/tools/pmc_read.py $R/$out/$lib
done
