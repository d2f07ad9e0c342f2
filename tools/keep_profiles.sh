#!/bin/bash
# Copies what tools/profile_round.sh left under gpurun_out/<tag>/ into profiles/ (tracked).  usage: bash tools/keep_profiles.sh <tag>
tag=$1
S=gpurun_out/$tag
cp $S/bench_strict.json profiles/${tag}_bench_strict.json
cp $S/bench_fast.json profiles/${tag}_bench_fast.json
cp $S/bench_under_rocprof.json profiles/${tag}_bench_strict_under_rocprof.json
cp "$(ls -t $S/kt/*/*_kernel_stats.csv | head -1)" profiles/${tag}_bench_strict_kernel_stats.csv
cp "$(ls -t $S/kt_workloads/*/*_kernel_stats.csv | head -1)" profiles/${tag}_workloads_kernel_stats.csv
cp $S/workloads_under_rocprof.json profiles/${tag}_workloads_under_rocprof.json
cp $S/${tag}_pmc_summary.json $S/${tag}_hbm_traffic.json profiles/
{ for loop in matrix vector; do echo "== sphere loop: $loop (tools/pmc.sh, six rocprofv3 --pmc passes, one launch each; sums over the XCDs' records)"; python3 tools/pmc_read.py $S/pmc_$loop; done; } > profiles/${tag}_pmc_render_kernels_strict.txt
ls -la profiles/${tag}_*
