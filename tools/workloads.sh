#!/bin/bash
# The workloads of the baseline list through tools/quick_bench.py (strict, one GPU), each under its own time limit.
# usage (on the GPU box): bash tools/workloads.sh > gpurun_out/<file>
q() { timeout -k 10 170 python3 tools/quick_bench.py --modes strict --reps 2 "$@" | tail -1; }
echo "scene_0 1280x720 100spp:      $(q --scene scene_0 --width 1280 --height 720 --spp 100)" &&
echo "rtiow 1920x1080 62spp:        $(q --spp 62)" &&
echo "rtiow 1920x1080 500spp:       $(q --spp 500)" &&
echo "rtiow 3840x2160 250spp:       $(q --width 3840 --height 2160 --spp 250)" &&
echo "perlin 1920x1080 1000spp:     $(q --scene perlin --spp 1000)" &&
echo "stress 10002 1080p 256spp:    $(q --scene stress:10000 --spp 256)" &&
echo "stress 10002 1080p 64spp vec: $(q --scene stress:10000 --spp 64 --vector-loop)"
