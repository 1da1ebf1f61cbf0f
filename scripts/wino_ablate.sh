#!/bin/bash
# timing experiments: per-layer rates of the Winograd kernel for the side builds scratch/libholo_abl<n>.so (WINO_ABL=n)
for n in 0 1 2 4 8; do
  if [ $n = 0 ]; then unset HOLO_HIP_LIB; else export HOLO_HIP_LIB=$PWD/scratch/libholo_abl$n.so; fi
  echo "== WINO_ABL=$n"; python scripts/per_layer.py 2>/dev/null | grep -E "^( 1| 3| 5|11|22) |total"
done
