#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats of the default bench, the other BASELINE configurations, the
# bundled-size case per convolution algorithm, PMC traffic (scripts/collect_pmc.py), the propagator alone.
# Outputs under gpurun_out/ (copy what is judged into profiles/).
R=${1:-r02}
O=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${R}_prof_bench -o bench --output-format csv -- python3 /root/repo/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $O/${R}_prof_bench.json 2> $O/${R}_prof_bench.err
rocprofv3 --kernel-trace --stats -d $O/${R}_prof_asm -o asm --output-format csv -- python3 /root/repo/scripts/bench_asm.py 64 256 50 > $O/${R}_prof_asm.txt 2>&1
cd /root/repo
python bench.py > $O/${R}_bench_default.json 2> $O/${R}_bench_default.err
python bench.py --conv-algo direct --cpu-sample 0 > $O/${R}_bench_direct.json 2>/dev/null
python bench.py --dtype bf16 --cpu-sample 0 > $O/${R}_bench_bf16.json 2>/dev/null
python bench.py --dtype bf16 --batch 256 --size 512 --cpu-sample 0 --steps 5 --warmup 2 > $O/${R}_bench_c3.json 2>/dev/null
python bench.py --batch 32 --size 1024 --distances 8 --style red_blood_cell --cpu-sample 0 --steps 3 --warmup 1 > $O/${R}_bench_c5.json 2>/dev/null
python bench.py --batch 64 --size 512 --cpu-sample 0 --steps 5 --warmup 2 > $O/${R}_bench_s512.json 2>/dev/null
for a in auto direct winograd; do
  python bench.py --batch 5 --size 128 --cpu-sample 0 --conv-algo $a --steps 50 --warmup 10 2>/dev/null > $O/${R}_bench_b5_$a.json
done
python scripts/per_layer.py > $O/${R}_layers.txt 2>&1
python scripts/collect_pmc.py --out $O/${R}_pmc_traffic.json > $O/${R}_pmc.log 2>&1
R_=$R python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('/root/repo/gpurun_out/REPL_bench_*.json'.replace('REPL', os.environ.get('R_','r02')))):
    try:
        d=json.load(open(f)); r=d.get('roofline',{}); p=d.get('roofline_propagator',{})
        print(os.path.basename(f), d['value'], d['ms_per_step'], r.get('achieved'), r.get('executed_mfma_frac'), p.get('achieved'), p.get('avg_call_ms'))
    except Exception as e: print(f, 'ERR', e)
PY
