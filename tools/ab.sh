#!/bin/bash
# A/B of two builds of libcaster_hip on one box: frame times of the benchmark scene at three sample counts,
# alternating between the libraries.  usage: bash tools/ab.sh libA.so libB.so [extra quick_bench args]
A=$1; B=$2; shift; shift
for round in 1 2; do
  for spp in 16 62 500; do
    for lib in $A $B; do
      echo "$lib spp=$spp: $(timeout -k 10 120 python3 tools/quick_bench.py --lib $lib --spp $spp --reps 3 --modes ${MODES:-strict} "$@" | tail -2 | sed -e 's/.*kernel \([0-9.]*\) ms.*/\1/' | tr '\n' ' ')"
    done
  done
done
