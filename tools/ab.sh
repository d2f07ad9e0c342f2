#!/bin/bash
# Side-by-side timing of several builds of libcaster_hip on one box: kernel times of the benchmark scene at
# three sample counts, the libraries taking turns.  usage: bash tools/ab.sh libA.so libB.so [libC.so ...] [-- quick_bench args]
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
[ "$1" = "--" ] && shift
for round in 1 2; do
  for spp in ${SPPS:-16 62 500}; do
    for lib in "${libs[@]}"; do
      echo "$lib spp=$spp: $(timeout -k 10 120 python3 tools/quick_bench.py --lib $lib --spp $spp --reps 3 --modes ${MODES:-strict} "$@" | tail -2 | sed -e 's/.*kernel \([0-9.]*\) ms.*/\1/' | tr '\n' ' ')"
    done
  done
done
