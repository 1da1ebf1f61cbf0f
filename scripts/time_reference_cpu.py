#!/usr/bin/env python3
"""Times the REFERENCE's own modules on the host CPU (build container only: /root/reference does not travel).

One step = ``field_retrieval`` + ``Holo_Generator(return_field=True, unwrap=False)`` under ``torch.no_grad()`` with the seeded
synthetic weights (tests/golden/make_golden.py's import path: the reference's modules from where they lie, nothing copied),
``torch.set_num_threads(all cores)``, 1 warm-up + median of >= 5 runs — the protocol of SURVEY.md 8d / BASELINE.md.
Configs 1, 2 and 5 of BASELINE.json at reduced batch.  Writes profiles/r03_reference_cpu_timing.json."""
import json
import os
import statistics
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
import make_golden as mg  # noqa: E402


def main():
    ref_net, _, Holo_Generator, _, field_retrieval = mg.import_reference()
    from style_transfer_based_holographic_imaging_amd import synth
    cores = len(os.sched_getaffinity(0))
    torch.set_num_threads(cores)
    network = mg.build_network(ref_net, 0)

    class P:
        wavelength, pixel_size, distance_normalize, distance_normalize_constant, phase_normalize = 532e-9, 1.5e-6, 1.0, 0.0, 1.0
    hg = Holo_Generator(P)
    out = {"host": f"{cores} vCPU (build container)", "torch": torch.__version__, "threads": cores,
           "protocol": "1 warm-up + median of 5, torch.no_grad(), reference modules imported from /root/reference", "configs": {}}

    def run(name, content, style, step, n_units, unit):
        with torch.no_grad():
            step(content, style)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                step(content, style)
                ts.append(time.perf_counter() - t0)
        med = statistics.median(ts)
        out["configs"][name] = {"median_s": round(med, 4), "runs_s": [round(t, 4) for t in ts], unit: round(n_units / med, 3),
                                "batch": int(content.shape[0]), "size": int(content.shape[-1])}
        print(name, out["configs"][name], flush=True)

    def load_style(name):
        return torch.load(os.path.join(REPO, "style_representation", name, "style_vector.pt"), map_location="cpu").float()

    def one_distance(content, style):
        amp, ph, d = field_retrieval(network, content, style, 1.0, True)
        return hg(amp, ph, -torch.full((content.shape[0], 1, 1, 1), 0.2), return_field=True, unwrap=False)

    # config 1: one bundled file (5 holograms 128 x 128)
    c1 = torch.sqrt(torch.load(os.path.join(REPO, "test_data", "test_content_holo_0.pt"), map_location="cpu")).float()
    run("config1_bundled_5x128", c1, load_style("MNIST"), one_distance, 5, "holograms_per_s")
    # config 2 at batch 8: synthetic 256 x 256
    ph = synth.synthetic_phase_objects(8, 256, seed=1234)
    c2 = torch.sqrt(hg(torch.full_like(ph, 0.6), ph, synth.synthetic_distances(8, seed=1234))).float()
    run("config2_8x256", c2, load_style("MNIST"), one_distance, 8, "holograms_per_s")
    # config 5 at batch 1: 1024 x 1024, red_blood_cell style, 8 distances with the reference's repeat idiom
    ph5 = synth.synthetic_phase_objects(1, 1024, seed=1237)
    c5 = torch.sqrt(hg(torch.full_like(ph5, 0.6), ph5, synth.synthetic_distances(1, seed=1237))).float()
    dist = torch.linspace(-0.1, -0.8, 8).view(8, 1, 1, 1)

    def sweep(content, style):
        amp, p, d = field_retrieval(network, content, style, 1.0, True)
        return hg(amp.repeat(8, 1, 1, 1), p.repeat(8, 1, 1, 1), dist, return_field=True, unwrap=False)
    run("config5_1x1024_sweep8", c5, load_style("red_blood_cell"), sweep, 1, "frames_per_s")
    with open(os.path.join(REPO, "profiles", "r03_reference_cpu_timing.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
