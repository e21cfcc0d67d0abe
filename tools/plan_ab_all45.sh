#!/bin/bash
# planner A/B over every bundled matrix x 20 replicas: product build (_lib) against a second build (_lib_b); us per step and the geometry chosen
L=chromosome3d_amd
ids=$(ls tests/golden/all45/*_upper.npz | xargs -n1 basename | sed 's/_upper.npz//')
for v in a b; do
  if [ $v = b ]; then mv $L/_lib $L/_lib_a && mv $L/_lib_b $L/_lib; fi
  for c in $ids; do python tools/geometry_compare.py $c 20 2>/dev/null | sed "s/^auto/$v/"; done
  if [ $v = b ]; then mv $L/_lib $L/_lib_b && mv $L/_lib_a $L/_lib; fi
done
