#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats of the default bench, the other BASELINE configurations, the
# bundled-size case per convolution algorithm, PMC traffic (scripts/collect_pmc.py), the propagator alone.
# Outputs under gpurun_out/ (copy what is judged into profiles/).  PART=1 | 2 | all (a gpurun call is limited to 20 min).
R=${1:-r04}
PART=${2:-all}
O=/root/repo/gpurun_out
if [ $PART = 1 ] || [ $PART = all ]; then
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${R}_prof_bench -o bench --output-format csv -- python3 /root/repo/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $O/${R}_prof_bench.json 2> $O/${R}_prof_bench.err
rocprofv3 --kernel-trace --stats -d $O/${R}_prof_asm -o asm --output-format csv -- python3 /root/repo/scripts/bench_asm.py 64 256 50 --eager > $O/${R}_prof_asm.txt 2>&1
cd /root/repo
python scripts/collect_pmc.py --out $O/${R}_pmc_traffic.json > $O/${R}_pmc.log 2>&1
cp $O/${R}_pmc_traffic.json profiles/${R}_pmc_traffic.json      # bench.py reads the traffic figures of THIS library from profiles/
python bench.py > $O/${R}_bench_default.json 2> $O/${R}_bench_default.err
python scripts/per_layer.py > $O/${R}_layers.txt 2>&1
fi
if [ $PART = 3 ] || [ $PART = all ]; then
cd /root/repo
# BASELINE configs[2]: counters of the bf16 convolution kernels (FETCH_SIZE / WRITE_SIZE / SQ in separate passes) and per-layer times
python scripts/collect_pmc.py --out $O/${R}_pmc_config3_bf16_conv.json --scratch $O/pmc_${R}_c3 -- --dtype bf16 --batch 256 --size 512 > $O/${R}_pmc_c3.log 2>&1
python scripts/per_layer.py --batch 64 --size 512 --dtype bf16 --reps 3 > $O/${R}_layers_bf16_b64_s512.txt 2>&1
python scripts/bf16_layer_bench.py --tiles 1,8,9 > $O/${R}_bf16_layer_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${R}_prof_c3 -o bench --output-format csv -- python3 /root/repo/bench.py --dtype bf16 --batch 256 --size 512 --steps 3 --warmup 1 --cpu-sample 0 > $O/${R}_prof_c3.json 2> $O/${R}_prof_c3.err
cd /root/repo
fi
if [ $PART = 2 ] || [ $PART = all ]; then
cd /root/repo
python bench.py --conv-algo direct --cpu-sample 0 > $O/${R}_bench_direct.json 2>/dev/null
python bench.py --dtype bf16 --cpu-sample 0 > $O/${R}_bench_bf16.json 2>/dev/null
python bench.py --dtype bf16 --batch 256 --size 512 --cpu-sample 0 --steps 5 --warmup 2 > $O/${R}_bench_c3.json 2>/dev/null
python bench.py --batch 32 --size 1024 --distances 8 --style red_blood_cell --cpu-sample 0 --steps 3 --warmup 1 > $O/${R}_bench_c5.json 2>/dev/null
python bench.py --batch 64 --size 512 --cpu-sample 0 --steps 5 --warmup 2 > $O/${R}_bench_s512.json 2>/dev/null
python bench.py --total 2048 --cpu-sample 0 --steps 2 --warmup 1 > $O/${R}_bench_c4_total2048_1gpu.json 2>/dev/null
HOLO_DIST_BACKEND=gloo HOLO_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --total 255 --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | grep '^{' > $O/${R}_bench_c4_total255_2rank_gloo_one_card.json
for a in auto direct winograd; do
  python bench.py --batch 5 --size 128 --cpu-sample 0 --conv-algo $a --steps 50 --warmup 10 2>/dev/null > $O/${R}_bench_b5_$a.json
done
python bench.py --batch 5 --size 128 --cpu-sample 0 --steps 200 --warmup 20 --no-events 2>/dev/null > $O/${R}_bench_b5_auto_noevents.json    # the step without a HIP event pair around each of its 45 launches
python bench.py --batch 1 --size 256 --cpu-sample 0 --steps 200 --warmup 20 --no-events 2>/dev/null > $O/${R}_bench_b1_s256_noevents.json
python scripts/per_layer.py --batch 5 --size 128 --reps 20 > $O/${R}_layers_b5.txt 2>&1
# the training step: HIP kernels against stock PyTorch-ROCm autograd, and its kernel statistics
python scripts/train_step.py --batch 16 --size 256 --steps 5 2>/dev/null | tail -1 > $O/${R}_train_step.json
python tests/diagnostics/torch_rocm_train_baseline.py --batch 16 --size 256 --steps 5 2>/dev/null | tail -1 > $O/${R}_torch_rocm_train_baseline.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${R}_prof_train -o train --output-format csv -- python3 /root/repo/scripts/train_step.py --batch 16 --size 256 --steps 3 > $O/${R}_prof_train.json 2> $O/${R}_prof_train.err
cd /root/repo
fi
R_=$R python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('/root/repo/gpurun_out/REPL_bench_*.json'.replace('REPL', os.environ.get('R_','r04')))):
    try:
        d=json.load(open(f)); r=d.get('roofline',{}); p=d.get('roofline_propagator',{})
        print(os.path.basename(f), d['value'], d['ms_per_step'], r.get('achieved'), r.get('frac'), p.get('achieved'), p.get('avg_call_ms'))
    except Exception as e: print(f, 'ERR', e)
PY
