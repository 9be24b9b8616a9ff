#!/usr/bin/env python3
"""Hashes of the denoiser's outputs on fixed synthetic inputs: run once per library build (AIPT_LIB=...) and compare the lines --
two builds whose conv kernels do the same arithmetic in the same order print the same hashes.
  python tools/bits_probe.py [H W [frames]]            (default 736 1280, 3 frames, hidden carried, batch-stat BN)
Also prints the kernel name of every layer once."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ai_path_tracer_denoiser_amd import api, synth  # noqa: E402


def main():
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 736
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
    nfr = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    ctx = api.Context(0)
    for impl, wseed in ((api.DN_IMPL_MFMA_F16X3, 565), (api.DN_IMPL_MFMA_F16W, 7)):
        ctx.load_weights(synth.make_blob(wseed))
        ctx.denoise_configure(H, W)
        ctx.denoise_set_impl(impl)
        ctx.reset_hidden()
        for j in range(nfr):
            x = torch.from_numpy(synth.make_gbuffer(H, W, 11, j)).cuda()
            y = torch.empty(3, H, W, device="cuda")
            ctx.denoise(x, y, bn_batch=True, carry=j > 0)
            ctx.sync()
            yh = y.cpu().numpy()
            print(f"impl {impl} {H}x{W} frame {j}: sha256 {hashlib.sha256(yh.tobytes()).hexdigest()[:24]}  finite {bool(np.isfinite(yh).all())}  "
                  f"mean {float(yh.mean()):.9g}")
        names = sorted({ctx.layer_info(l)["kernel"] for l in range(28)})
        print("kernels:", ", ".join(names))
    ctx.close()


if __name__ == "__main__":
    main()
