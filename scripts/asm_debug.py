#!/usr/bin/env python3
"""One propagator call with the dataflow kernel's control block dumped afterwards (diagnostics)."""
import faulthandler
import os
import sys
import time

faulthandler.dump_traceback_later(80, exit=True)
print("import torch ...", flush=True)
import torch  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import HoloParams  # noqa: E402
from style_transfer_based_holographic_imaging_amd import ops  # noqa: E402
from utils.Forward_model import Holo_Generator  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 2
s = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
hg = Holo_Generator(HoloParams())
g = torch.Generator().manual_seed(0)
amp = (torch.rand(b, 1, s, s, generator=g) + 0.2).to(dev)
ph = (torch.rand(b, 1, s, s, generator=g) * 6 - 3).to(dev)
d = torch.full((b, 1, 1, 1), -0.2, device=dev)
torch.cuda.synchronize()
print("inputs ready", flush=True)
ops._ASM_KEEP_WORKSPACE[:] = [None]
for rep in range(2):
    t0 = time.perf_counter()
    a_o, p_o = hg(amp, ph, d, return_field=True)
    print("  launched", flush=True)
    torch.cuda.synchronize()
    print(f"call {rep}: {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
    ws = ops._ASM_KEEP_WORKSPACE[0]
    ctl = ws[3 * 4096 * 8: 3 * 4096 * 8 + 64 + 512 + 4 * 4 * 8 * b].view(torch.int32).cpu().tolist()
    print(f"  next_field {ctl[0]} error {ctl[1]} tickets/XCD {[ctl[16 + 16 * x] for x in range(8)]} end1/XCD {[ctl[16 + 16 * x + 1] for x in range(8)]}")
    base = 144
    for name in ("claim", "done_a", "done_b", "done_c"):
        print("  ", name, [ctl[base + x * b: base + (x + 1) * b] for x in range(8)])
        base += 8 * b
sys.path.insert(0, os.path.join(REPO, "oracle"))
import holo_oracle as orc  # noqa: E402
ref = orc.holo_generator(orc.HoloParams(), amp.cpu(), ph.cpu(), d.cpu(), complex_number=True)
got = torch.polar(a_o, p_o).cpu().to(torch.complex128)
print("max rel err", float((torch.view_as_real(got) - torch.view_as_real(ref)).abs().max() / torch.view_as_real(ref).abs().max()), flush=True)
