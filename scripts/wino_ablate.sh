#!/bin/bash
# timing experiments: per-layer rates of the Winograd kernel for side builds scratch/libholo_<tag>.so
for n in "$@"; do
  if [ $n = base ]; then unset HOLO_HIP_LIB; else export HOLO_HIP_LIB=$PWD/scratch/libholo_$n.so; fi
  echo "== $n"; python scripts/per_layer.py 2>/dev/null | grep -E "^( 1| 3| 5|11|22) |total"
done
