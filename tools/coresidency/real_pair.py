#!/usr/bin/env python3
"""The one failing pair of tools/coresidency/run_matrix.py in detail: a one-launch trace (primitives only, depth 1, no AA:
trace_bounce<true,false,false>, 24 workgroups) beside forward passes of the split-fp16 denoiser at 736x1280.
    python tools/coresidency/real_pair.py RUNS [depth] [flags] [mesh 0/1]
    env: AIPT_DEBUG_LAYER_MASK (which conv layers the aggressor launches), PAIR_IMPL (aggressor impl), PAIR_DH/PAIR_DW
Prints, per failing run, the planes, pixels, (workgroup, wave, lane) and got / expected values."""
import os
import sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch  # noqa: E402
from ai_path_tracer_denoiser_amd import api, synth  # noqa: E402
from tests.test_gpu_frame import _mesh_scene  # noqa: E402

RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 1
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 2
mesh = len(sys.argv) > 4 and sys.argv[4] == "1"
W, H = 96, 64
sc, mats, faces, box = _mesh_scene((W, H), depth)
cams = [sc.orbit(phi=sc.phi + 0.1 * k) for k in range(8)]
A = api.Context(0)
A.pathtrace_init(sc.geoms, mats, faces if mesh else faces[:0], box if mesh else None, W, H)
g = torch.zeros(10, H, W, device="cuda")
torch.cuda.synchronize()
ref = []
for c in cams:
    A.pathtrace(c, 1, depth, g, flags); A.sync(); ref.append(g.clone())
B = api.Context(0)
B.load_weights(synth.make_blob(565))
DH, DW = int(os.environ.get("PAIR_DH", 736)), int(os.environ.get("PAIR_DW", 1280))
B.denoise_configure(DH, DW)
B.denoise_set_impl(int(os.environ.get("PAIR_IMPL", 2)))
gb = torch.from_numpy(synth.make_gbuffer(DH, DW, 3, 0)).cuda()
ob = torch.empty(3, DH, DW, device="cuda")
torch.cuda.synchronize()
bad = 0
for r in range(RUNS):
    k = r % 8
    B.denoise(gb, ob, bn_batch=True, carry=False)
    A.pathtrace(cams[k], 1, depth, g, flags)
    A.sync()
    d = (g.view(torch.int32) != ref[k].view(torch.int32))
    n = int(d.sum().item())
    if n:
        bad += 1
        if bad <= 12:
            idx = d.nonzero().cpu().numpy()
            gv, rv = g.cpu().numpy(), ref[k].cpu().numpy()
            per_plane = [int((idx[:, 0] == pl).sum()) for pl in range(10)]
            pix = sorted(set(int(y * W + (W - 1 - x)) for _, y, x in idx))          # un-flipped pixel index = thread index
            runs, s0 = [], pix[0]
            for a, b in zip(pix, pix[1:] + [None]):
                if b != a + 1:
                    runs.append((s0, a)); s0 = b
            print(f"run {r} frame {k}: {n} words, per plane {per_plane}, {len(pix)} pixels; thread runs (first,last | wg wave lane..lane):",
                  [(a, b, a // 256, (a % 256) // 64, a % 64, b % 64) for a, b in runs][:12], flush=True)
            for pl, y, x in idx[:6]:
                print(f"     plane {pl} pixel {y * W + (W - 1 - x)}: got {gv[pl, y, x]!r} expected {rv[pl, y, x]!r}")
    if r % 8 == 7:
        B.sync()
B.sync()
print(f"RESULT depth {depth} flags {flags} mesh {mesh} mask {os.environ.get('AIPT_DEBUG_LAYER_MASK')} impl {os.environ.get('PAIR_IMPL', 2)}: runs {RUNS} bad {bad}", flush=True)
