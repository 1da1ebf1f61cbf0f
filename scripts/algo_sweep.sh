#!/bin/bash
# holograms/s per convolution algorithm over batch sizes (tunes engine.Plan.WINO_MIN_BLOCKS)
IFS=";" read -ra CFGS <<< "${SWEEP:-1 256;2 256;4 256;8 256;16 256;32 256;5 128;16 128;64 128}"
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  for a in direct winograd auto; do
    python bench.py --batch $1 --size $2 --cpu-sample 0 --conv-algo $a --steps 20 --warmup 5 --no-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b$1 s$2', d['config']['conv_algo'], d['value'], d['ms_per_step'])"
  done
done
