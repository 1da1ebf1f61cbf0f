#!/usr/bin/env python3
"""holograms/s of the field-retrieval hot path on MI355X (BASELINE.json metric).

One step = one pass of the path over one batch of synthetic holograms already resident in HBM:
  encoder -> AdaIN (stored style) -> decoder -> distance head  (utils.functions.field_retrieval)
  -> angular-spectrum back-propagation to focus (Holo_Generator(..., return_field=True)),
i.e. test_field_retrieval_mnist.py:125-126 of the reference without the host-side unwrap.
Default workload = BASELINE.json configs[1]: batch 64 per GPU, 256x256, fp32.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU; every rank runs its own batch (weak scaling, no data-path collective)
and the packed outputs are gathered to rank 0 with one gather per step (RCCL on the GPU node), inside the timed region.
--total N: BASELINE configs[3] as written — a FIXED total of N holograms (2048 there) sharded by
shard.shard_bounds over the ranks, each rank looping micro-batches of --batch over its block, ONE gather of the
rank's packed block per step (= one pass over all N holograms), "scaling": "strong".
Rank 0 prints one JSON line.  `roofline` is the fp32-MFMA convolution kernel (the dominant kernel),
`roofline_propagator` the HBM-bound angular-spectrum propagator; both from HIP events recorded on the
launch stream during the timed steps.  `cpu_baseline` times the CPU oracle (a port of the reference's
torch-CPU path; the reference itself cannot travel to the GPU box) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); keep an
# explicit setting if the launcher made one
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (not the 2:1-sparse headline)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak
FLOP_PER_PIXEL = 1136262       # conv stack per input pixel (+5.24 MFLOP distance head)


def build_network(device, seed=0):
    import copy
    import torch.nn as nn
    import net
    from style_transfer_based_holographic_imaging_amd import synth
    vgg_sd, dec_sd, dist_sd = synth.synthetic_state_dicts(seed, full_vgg=True)
    vgg = copy.deepcopy(net.vgg)
    vgg.load_state_dict(vgg_sd)
    dec = copy.deepcopy(net.decoder)
    dec.load_state_dict(dec_sd)
    dg = net.Distance_G()
    dg.load_state_dict(dist_sd)
    network = net.Net(nn.Sequential(*list(vgg.children())[:31]), dec, None, dg).eval().to(device)
    return network, (vgg_sd, dec_sd, dist_sd)


def cpu_baseline(size, sds, style, sample, seed):
    """The oracle timed on the host cores, on `sample` holograms of the same workload."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import holo_oracle as orc
    from style_transfer_based_holographic_imaging_amd import synth
    hp = orc.HoloParams()
    ph = synth.synthetic_phase_objects(sample, size, seed=seed)
    d_c = synth.synthetic_distances(sample, seed=seed)
    d_s = torch.full((sample, 1, 1, 1), 0.2)
    cores = min(len(os.sched_getaffinity(0)), 16)   # the GPU box exposes every host CPU; one GPU's share is 16
    torch.set_num_threads(cores)
    with torch.no_grad():
        content = torch.sqrt(orc.holo_generator(hp, torch.full_like(ph, 0.6), ph, d_c)).float()
        orc.full_step(content, style, d_s, *sds, hp)  # warm-up
        times = []
        for _ in range(5):
            t0 = time.perf_counter()
            orc.full_step(content, style, d_s, *sds, hp)
            times.append(time.perf_counter() - t0)
    best = sorted(times)[2]
    return {"value": round(sample / best, 3), "unit": "holograms/s", "cores": cores, "kind": "port",
            "sample": f"{sample} holograms {size}x{size}, oracle/holo_oracle.py full_step, median of 5 after 1 warm-up, "
                      f"torch {torch.__version__} CPU"}


PMC_PROFILE = os.path.join(REPO, "profiles", "r04_pmc_traffic.json")


def pmc_traffic(batch, size, dtype, distances, single_rank_step=True):
    """HBM bytes (conv: per launch; propagator: per call) from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    over this same command (scripts/collect_pmc.py -> profiles/r04_pmc_traffic.json; FETCH_SIZE doubled per
    MI355X_MICROARCH.md §HBM).  Counters cannot be read from inside the process, so the numbers are only reported for the
    workload AND the library build they were collected on (the profile stores the library's source hash): anything else
    gives null rather than a stale figure."""
    if not single_rank_step or (batch, size, dtype, distances) != (64, 256, "f32", 1) or not os.path.isfile(PMC_PROFILE):
        return None, None, None
    from style_transfer_based_holographic_imaging_amd import _lib
    with open(PMC_PROFILE) as f:
        prof = json.load(f)
    if prof.get("bench_args") or prof.get("source_hash") != _lib.lib.holo_source_hash().decode():
        return None, None, None
    label = f"profiles/{os.path.basename(PMC_PROFILE)}, collected on library {prof['source_hash'][:12]} (separate rocprofv3 --pmc passes)"
    conv = prof.get("conv_wino_hbm_bytes_per_launch_avg") or prof.get("conv_mfma_hbm_bytes_per_launch_avg")
    asm = prof.get("asm_hbm_bytes_per_call")
    return (round(conv) if conv else None), (round(asm) if asm else None), label


class HoloParams:
    """The five attributes Holo_Generator reads from ``args`` (utils/Forward_model.py:9-13) with the entry script's
    defaults (test_field_retrieval_mnist.py:56-60)."""
    wavelength, pixel_size, distance_normalize, distance_normalize_constant, phase_normalize = 532e-9, 1.5e-6, 1.0, 0.0, 1.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="holograms per GPU per step (with --total: the micro-batch)")
    ap.add_argument("--total", type=int, default=0,
                    help="strong scaling (BASELINE configs[3]): a fixed total of N holograms per step, sharded over the ranks "
                         "(contiguous blocks), processed in micro-batches of --batch, one gather of each rank's block")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dump-block", default="",
                    help="with --total: rank 0 saves the gathered (N, 4*S*S+1) block of one extra pass AFTER the timed region "
                         "(tests compare it across rank counts)")
    ap.add_argument("--cpu-sample", type=int, default=16,
                    help="holograms per pass of the CPU baseline (0 = skip); 16 x (1 warm-up + 5 timed passes) is 10-15 s of host work at 256 x 256")
    ap.add_argument("--no-events", action="store_true", help="do not record per-launch HIP events")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step into a hipGraph (torch.cuda.CUDAGraph) and time K replays; the per-kernel "
                         "roofline figures then come from K extra eager steps after the timed region")
    ap.add_argument("--distances", type=int, default=1,
                    help="propagate every retrieved field to N distances in one sweep (BASELINE configs[4] uses 8)")
    ap.add_argument("--style", default="MNIST", help="style_representation/<name>/style_vector.pt")
    ap.add_argument("--dtype", choices=["f32", "bf16", "f32s"], default="f32",
                    help="conv stack arithmetic: f32 = exact fp32 MFMA (BASELINE configs[1], the parity path); "
                         "bf16 = bf16 MFMA with fp32 accumulation (configs[2]); the propagator is fp32 either way")
    ap.add_argument("--conv-algo", choices=["auto", "direct", "winograd"], default="auto",
                    help="fp32 3x3 layers: auto = Winograd F(2x2,3x3) on the fp32 matrix cores where the layer fills the chip "
                         "(exact fp32 arithmetic, 2.25x fewer multiplications), else the direct implicit-GEMM kernel")
    args = ap.parse_args()

    from style_transfer_based_holographic_imaging_amd import build as _build
    if not os.environ.get("HOLO_HIP_LIB"):
        _build.build(verbose=False)      # no-op when the in-tree library is up to date; locked + atomic rename, so every rank may call it
    from style_transfer_based_holographic_imaging_amd import ops, shard, synth
    # RCCL ("nccl") on the GPU node; HOLO_DIST_BACKEND=gloo + HOLO_SHARE_GPU=1 rehearse the multi-rank control flow
    # with several ranks on ONE card (the one-GPU development box)
    backend = os.environ.get("HOLO_DIST_BACKEND", "nccl")
    rank, world, local = shard.init_from_env(backend)
    dist_on = world > 1 or shard.force_collectives()     # HOLO_FORCE_DIST=1: one rank, but through the collective path all the same
    if os.environ.get("HOLO_SHARE_GPU") == "1":
        local = min(local, torch.cuda.device_count() - 1)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.graph and (dist_on or args.total):
        raise SystemExit("--graph captures one single-rank step: it cannot hold the gather of a multi-rank step or a --total pass")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    from utils.Forward_model import Holo_Generator
    from utils.functions import field_retrieval

    network, sds = build_network(dev)
    network.set_conv_algo(args.conv_algo)
    if args.dtype == "bf16":
        network.set_compute_dtype(torch.bfloat16)
    elif args.dtype == "f32s":   # opt-in: fp32 values as split fp16 halves on the fp16 matrix cores (never the default)
        network.set_compute_dtype("f32s")
    model_forward = Holo_Generator(HoloParams())
    style_cpu = torch.load(os.path.join(REPO, "style_representation", args.style, "style_vector.pt"), map_location="cpu").float()
    style = style_cpu.to(dev)

    # synthetic input, SURVEY.md §8d config 2 / 4: phase objects -> holograms by forward propagation (on the GPU)
    b, s = args.batch, args.size
    if args.total:
        if args.distances != 1:
            raise SystemExit("--total runs the one-distance path (configs[3])")
        # config 4: hologram i of the job is a function of i alone (chunks of 64 seeded 1236 + chunk), whatever the rank count
        lo, hi = shard.shard_bounds(args.total, world, rank)
        seed = 1236
        ph_j, d_j = synth.job_inputs(lo, hi, s, seed=seed)
        chunks = []
        for c0 in range(0, hi - lo, 64):        # forward propagation in pieces of 64 (the generator's own chunking is index-based)
            ph_c, d_c = ph_j[c0:c0 + 64].to(dev), d_j[c0:c0 + 64].to(dev)
            with torch.no_grad():
                chunks.append(model_forward(torch.full_like(ph_c, 0.6), ph_c, d_c))   # hologram INTENSITY: the step takes the sqrt (below)
        content = torch.cat(chunks, 0).contiguous() if chunks else torch.empty((0, 1, s, s), device=dev)
        del chunks, ph_j, d_j
        n_local = hi - lo
        d_style = torch.full((max(b, 1), 1, 1, 1), -0.2, device=dev)     # back-propagation: -distance_style (test_field_retrieval_mnist.py:126)
        # send buffer of the padded block size and (root) receive buffer, both kept for the whole job: a ragged pass
        # copies and allocates nothing
        rows_max = shard.block_rows(args.total, world)
        block = torch.zeros((rows_max, shard.packed_width(s)), dtype=torch.float32, device=dev)
        recv = torch.empty((world, rows_max, shard.packed_width(s)), dtype=torch.float32, device=dev) if (dist_on and rank == 0) else None
    else:
        seed = 1234 + rank
        ph = synth.synthetic_phase_objects(b, s, seed=seed).to(dev)
        d_content = synth.synthetic_distances(b, seed=seed).to(dev)
        d_style = torch.full((b, 1, 1, 1), -0.2, device=dev)             # back-propagation: -distance_style (test_field_retrieval_mnist.py:126)
        with torch.no_grad():
            content = model_forward(torch.full_like(ph, 0.6), ph, d_content).contiguous()   # hologram INTENSITY, as the entry loop loads it
        del ph

    sweep_d = torch.linspace(-0.1, -0.8, max(args.distances, 2), device=dev)[:args.distances]

    def one_batch(x, d_st):
        # test_field_retrieval_mnist.py:119 + 125: sqrt(content_holo) is part of the per-batch work; here the first layer's loader
        # takes it (bit-identical to torch.sqrt first)
        amp, p, d = field_retrieval(network, x, style, 1.0, True, intensity=True)
        if args.distances > 1:   # utils/Data_loader.py:59-62 idiom: one field, a list of distances
            a_f, p_f = model_forward.sweep(amp, p, sweep_d)           # (the entry loop's `ph_field * phase_normalize` is the kernel's phase_scale)
        else:
            a_f, p_f = model_forward(amp, p, d_st, return_field=True, unwrap=False)
        return amp, p, d, a_f, p_f

    def step_weak():
        with torch.no_grad():
            amp, p, d, a_f, p_f = one_batch(content, d_style)
            if dist_on and args.distances == 1:
                # the step's one exchange: started here, waited for one step later, so it rides under the next
                # step's kernels; sync() drains the last one inside the timed region
                pending.append(shard.gather_to_root(shard.pack_outputs(amp, p, a_f, p_f, d), async_op=True))
                if len(pending) > 1:
                    pending.pop(0).wait()
        return amp, p, d, a_f, p_f

    def step_strong():
        # one pass over the job's N holograms: this rank's block in micro-batches, then ONE gather of the packed block
        last = []

        def process(m0, m1, rows):
            out = one_batch(content[m0 - lo:m1 - lo], d_style[:m1 - m0])
            shard.pack_outputs(out[0], out[1], out[3], out[4], out[2], out=rows)
            last[:] = [out]

        with torch.no_grad():
            if pending:                      # the block buffer is reused: the previous pass's gather must have left it
                pending.pop(0).wait()
            pending.append(shard.sharded_pass(args.total, b, process, shard.packed_width(s), dev, async_op=True, block=block, recv=recv))
        return last[0] if last else None

    step = step_strong if args.total else step_weak

    pending = []

    def sync():
        while pending:
            pending.pop(0).wait()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    timer = None if args.no_events else ops.KernelTimer()
    if args.graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            graph_out = step()          # noqa: F841  (keeps the captured outputs alive)
        graph.replay()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            graph.replay()
        sync()
        elapsed = time.perf_counter() - t0
        if timer is not None:           # eager pass for the per-kernel numbers only
            with timer:
                for _ in range(args.steps):
                    step()
            sync()
    else:
        t0 = time.perf_counter()
        if timer is not None:
            with timer:
                for _ in range(args.steps):
                    step()
        else:
            for _ in range(args.steps):
                step()
        sync()
        elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    gather_ms = None
    if dist_on and args.distances == 1:
        # the gather's own time, measured SEPARATELY after the timed region (in the steps it rides under the next
        # step's kernels): three blocking gathers of one step's message between barriers, every rank joins, MAX over ranks
        with torch.no_grad():
            if args.total:
                def one_gather():
                    counts = [shard.shard_bounds(args.total, world, q)[1] - shard.shard_bounds(args.total, world, q)[0] for q in range(world)]
                    shard.gather_to_root(block, counts, recv=recv)
            else:
                amp_g, p_g, d_g, af_g, pf_g = one_batch(content, d_style)
                msg = shard.pack_outputs(amp_g, p_g, af_g, pf_g, d_g)

                def one_gather():
                    shard.gather_to_root(msg)
        one_gather()
        sync()
        tg = time.perf_counter()
        for _ in range(3):
            one_gather()
        sync()
        tgt = torch.tensor([(time.perf_counter() - tg) / 3.0], dtype=torch.float64, device=dev)
        dist.all_reduce(tgt, op=dist.ReduceOp.MAX)
        gather_ms = round(float(tgt.item()) * 1e3, 3)

    if args.total and args.dump_block:
        step()                            # one extra pass, outside the timed region; every rank joins its gather
        gathered = pending.pop(0).wait()
        sync()
        if rank == 0:
            torch.save(gathered.cpu(), args.dump_block)

    if rank == 0:
        total = (args.total if args.total else b * world) * args.steps
        coll = {"nccl": "RCCL"}.get(backend, backend) if dist_on else "none (single rank)"
        # which BASELINE config the run IS (exact match of its defining parameters), else the nearest one named as a variant
        if args.total:
            cfg = "BASELINE configs[3]" if (args.total, s, args.dtype) == (2048, 256, "f32") else "variant of BASELINE configs[3] (fixed total)"
        elif args.distances > 1:
            cfg = "BASELINE configs[4]" if (b, s, args.distances, args.style, args.dtype) == (32, 1024, 8, "red_blood_cell", "f32") else "variant of BASELINE configs[4] (multi-distance sweep)"
        elif args.dtype == "bf16":
            cfg = "BASELINE configs[2]" if (b, s) == (256, 512) else "variant of BASELINE configs[2] (bf16 convs)"
        elif args.dtype == "f32":
            cfg = "BASELINE configs[1]" if (b, s) == (64, 256) else "variant of BASELINE configs[1]"
        else:
            cfg = "opt-in f32s mode (no BASELINE config)"
        per_gpu = f"{args.total} holograms in all, contiguous blocks of {-(-args.total // world)} per GPU in micro-batches of {b}" if args.total else f"batch {b}/GPU"
        out = {
            "metric": "holograms/sec (256x256 complex field retrieval)" if s == 256 else f"holograms/sec ({s}x{s} complex field retrieval)",
            "value": round(total / elapsed, 2), "unit": "holograms/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "strong" if args.total else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{cfg}: {per_gpu}, synthetic {s}x{s} holograms, "
                                   f"{ {'f32': 'fp32', 'bf16': 'bf16 convs / fp32 FFT', 'f32s': 'convs on split-fp16 operands (hi + lo*2^-11, fp32 accumulate; NOT fp32 arithmetic) / fp32 FFT'}[args.dtype] }, VGG-AdaIN-decoder + "
                                   f"distance head + angular-spectrum back-propagation (P={2 * s}), seeded He-normal weights",
                       "batch_per_gpu": b, "size": s, "distances_per_field": args.distances, "style": args.style, "hipgraph": bool(args.graph), "conv_algo": args.conv_algo, "total_holograms": args.total or b * world,
                       "parallelism": f"batch-sharded x{world}, one {coll} gather per step" if dist_on else "single GPU"},
        }
        if gather_ms is not None:
            out["gather_ms"] = gather_ms
            out["gather_note"] = (f"one blocking {coll} gather of a step's packed outputs to rank 0 with a barrier after it, timed "
                                  "alone after the timed region (inside the steps it is asynchronous and overlaps the next step)")
        if timer is not None:
            summ = timer.summary()
            traffic_conv, traffic_asm, traffic_src = pmc_traffic(b, s, args.dtype, args.distances, single_rank_step=(world == 1 and not args.total))
            wino = summ.get("holo_conv3x3_wino_f32")
            # the dominant kernel: the Winograd 3x3 kernel where it runs (fp32, large layers), else the implicit-GEMM kernel
            conv = [wino] if wino else [summ[k] for k in (f"holo_conv3x3_{args.dtype}", f"holo_convt2x2_{args.dtype}") if k in summ]
            # f32s: algorithmic fp32 FLOP against the fp16 matrix peak / 3 (three MFMA products per operand pair)
            peak = {"f32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "f32s": round(PEAK_BF16_MFMA_TFLOPS / 3.0, 1)}[args.dtype]
            if conv:
                fl = sum(r["flops"] for r in conv)
                ms = sum(r["ms"] for r in conv)
                calls = sum(r["calls"] for r in conv)
                alg = fl / (ms * 1e-3) / 1e12
                # Winograd F(2x2,3x3) executes 16 multiplications per 2x2 output tile and channel pair where the direct
                # convolution needs 36: `achieved` / `frac` describe the FLOP the kernel EXECUTES on the matrix pipe (a fraction
                # of peak <= 1); the direct-convolution-equivalent ("algorithmic") rate is reported beside it
                ach = alg * 16.0 / 36.0 if wino else alg
                out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                                   "frac": round(ach / peak, 4), "traffic": traffic_conv, "traffic_source": traffic_src,
                                   "algorithmic_bytes_per_launch": round(sum(r["bytes"] for r in conv) / calls),
                                   "kernel": "conv_wino_f32_kernel (Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32, all large 3x3 layers)" if wino
                                   else ("bf16 3x3 + transposed convolutions, all layers: conv3x3_b16_kernel / conv3x3_b16_wres_kernel / "
                                         "convt2x2_b16_c64_kernel (v_mfma_f32_16x16x32_bf16) and conv_mfma_kernel<bf16> (32x32x16) for the rest"
                                         if args.dtype == "bf16" else f"conv_mfma_kernel<{args.dtype}> (3x3 + transposed, all layers)"),
                                   "launches_per_step": calls // args.steps, "avg_launch_ms": round(ms / calls, 4),
                                   "gflop_per_launch": round(fl * (16.0 / 36.0 if wino else 1.0) / calls / 1e9, 3)}
                if wino:
                    out["roofline"]["direct_conv_equivalent_tflops"] = round(alg, 2)
                    out["roofline"]["direct_conv_equivalent_over_peak"] = round(alg / peak, 4)
                    out["roofline"]["executed_mfma_frac"] = round(ach / peak, 4)
                    out["roofline"]["note"] = ("achieved = executed MFMA FLOP (2*16*Cin*Cout per 2x2 output tile) / time; the same launches "
                                               "counted as a direct convolution (2*36*Cin*Cout) give direct_conv_equivalent_tflops, which "
                                               "may exceed the fp32 MFMA peak — that is the algorithm's gain, not pipe utilisation")
                    rest = [summ[k] for k in (f"holo_conv3x3_{args.dtype}", f"holo_convt2x2_{args.dtype}") if k in summ]
                    if rest:   # the launches that stay on the implicit-GEMM kernel (transposed convs, layers below the Winograd threshold)
                        fl2, ms2, c2 = (sum(r[k] for r in rest) for k in ("flops", "ms", "calls"))
                        out["roofline_conv_direct"] = {"bound": "mfma", "achieved": round(fl2 / (ms2 * 1e-3) / 1e12, 2), "peak": peak,
                                                       "unit": "TFLOP/s", "frac": round(fl2 / (ms2 * 1e-3) / 1e12 / peak, 4),
                                                       "kernel": "conv_mfma_kernel<f32> (transposed convs and 3x3 layers not on Winograd)",
                                                       "launches_per_step": c2 // args.steps, "avg_launch_ms": round(ms2 / c2, 4)}
            if "holo_asm_propagate" in summ:
                r = summ["holo_asm_propagate"]
                gbs = r["bytes"] / (r["ms"] * 1e-3) / 1e9
                out["roofline_propagator"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                              "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": traffic_asm, "traffic_source": traffic_src,
                                              "algorithmic_bytes_per_call": round(r["bytes"] / r["calls"]),
                                              "kernel": "asm_rows_fwd + asm_cols + asm_rows_inv (80*S^2 B per field)",
                                              "avg_call_ms": round(r["ms"] / r["calls"], 4)}
            out["kernel_ms_per_step"] = {k: round(v["ms"] / args.steps, 4) for k, v in sorted(summ.items())}
            out["conv_stack_tflops_overall"] = round(FLOP_PER_PIXEL * s * s * (n_local if args.total else b) * args.steps / 1e12 /
                                                     (sum(v["ms"] for k, v in summ.items() if "conv" in k) * 1e-3), 2)
        if world == 1 and args.distances == 1 and not args.total:
            # the step after the metric's path (test_field_retrieval_mnist.py:126 unwrap=True): timed on its own,
            # outside the holograms/s region, which SURVEY.md §8d defines pre-unwrap.  Single rank only: nothing
            # after the timed region may issue a collective the other ranks do not join.
            from utils.functions import unwrap
            p_f = step()[4]
            unwrap(p_f)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                unwrap(p_f)
            torch.cuda.synchronize()
            out["unwrap_ms_per_batch"] = round((time.perf_counter() - t1) / 5 * 1e3, 3)
        if world == 1 and args.dtype == "f32" and args.distances == 1 and not args.total:
            # informational, after the timed region and never part of `value`: the same step in the opt-in f32s mode
            # (fp32 values as split fp16 halves on the fp16 matrix cores) with its deviation from this run's
            # fp32 outputs
            try:   # informational only: never let it cost the headline line
                ref = step()
                network.set_compute_dtype("f32s")
                alt = step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                dt_alt = (time.perf_counter() - t1) / args.steps
                network.set_compute_dtype(torch.float32)

                def _rel(a, r):
                    return float((a.double() - r.double()).abs().max() / r.double().abs().max())
                out["alt_mode_f32s"] = {"value": round(b / dt_alt, 2), "unit": "holograms/s", "ms_per_step": round(dt_alt * 1e3, 3),
                                        "max_rel_dev_from_fp32_outputs": {"amplitude": _rel(alt[0], ref[0]), "phase": _rel(alt[1], ref[1]),
                                                                          "distance": _rel(alt[2], ref[2]), "focused_amplitude": _rel(alt[3], ref[3])},
                                        "note": "opt-in, not fp32 arithmetic; not the headline"}
            except Exception as e:   # noqa: BLE001
                network.set_compute_dtype(torch.float32)
                out["alt_mode_f32s"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(s, sds, style_cpu, args.cpu_sample, seed)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
