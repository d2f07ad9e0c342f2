#!/bin/bash
# One GPU call of measurement under a tag: the counter passes of tools/pmc.sh for both sphere loops, then bench.py
# and the same under rocprofv3 --kernel-trace --stats.  usage (on the GPU box): bash tools/profile_round.sh <tag>
set -e
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
rm -rf $O/kt $O/kt_workloads $O/pmc_matrix $O/pmc_vector
mkdir -p $O
cd $R
BENCH_ARGS="--spp 500 --reps 1 --modes strict" bash tools/pmc.sh gpurun_out/$tag/pmc_matrix --
BENCH_ARGS="--spp 500 --reps 1 --modes strict --vector-loop" bash tools/pmc.sh gpurun_out/$tag/pmc_vector --
python3 tools/pmc_summary.py gpurun_out/$tag/pmc_matrix "rtiow_final_scene 1920x1080 500spp depth50" strict matrix \
    gpurun_out/$tag/pmc_vector "rtiow_final_scene 1920x1080 500spp depth50" strict vector --out gpurun_out/$tag/$tag > $O/pmc_summary.txt
# bench.py quotes the counters of the tree's own sources: put this run's summary where it looks for it
cp $O/${tag}_pmc_summary.json $O/${tag}_hbm_traffic.json $R/profiles/
cd $R
timeout -k 10 240 python3 bench.py --steps 5 --warmup 1 > $O/bench_strict.json 2> $O/bench_strict.err
timeout -k 10 240 python3 bench.py --steps 5 --warmup 1 --mode fast --no-cpu-baseline > $O/bench_fast.json 2> $O/bench_fast.err
cd /tmp && export TMPDIR=/tmp
# the timed frames and the vector-loop leg alone (the 4K workload runs the same kernel instance as the main frames and would blur its average)
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload-steps 0 > $O/bench_under_rocprof.json 2> $O/kt.err
# the other workloads of the baseline list (one main frame comes with them)
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_workloads -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --vector-steps 0 > $O/workloads_under_rocprof.json 2> $O/kt_workloads.err
cat $O/bench_strict.json
