#!/bin/bash
# per-layer timing of the bf16 conv stack at the config-3 image size for every tile variant (A/B on one box)
set -e
B=${1:-64}; S=${2:-512}
for t in ${TILES:-1 8 9}; do
  echo "=== tile $t (batch $B, ${S}x${S}, bf16)"
  python scripts/per_layer.py --batch $B --size $S --dtype bf16 --tile $t --reps 3
done
