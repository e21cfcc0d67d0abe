#!/bin/bash
# planner's geometry against the alternatives on a set of problems (one GPU): bash tools/geometry_sweep.sh > out.txt
for spec in "chr1_500kb 20" "chr1_500kb 10" "chr1_500kb 5" "chr4_1mb 20" "chr21_1mb 20" "chr13_1mb 20" "chr19_500kb 20" "chr10_500kb 20" "chr7_1mb 20" "chr22_1mb 20"; do
  set -- $spec
  python tools/geometry_compare.py $1 $2
  for g in 8x1x4 8x2x4 12x2x4 8x4x4 10x4x4 12x4x4 8x3x4 4x2x4 4x4x4 6x4x4; do
    python tools/geometry_compare.py $1 $2 $g 2>/dev/null | grep -v "path 1" | grep "us/step"
  done
  echo
done
