#!/usr/bin/env python3
"""Max abs error of the conv implementations against the CPU oracle run on the TRUE fp32 weights, frame by frame with the
hidden state carried (the recurrence feeds each frame's error into the next)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import oracle
    from ai_path_tracer_denoiser_amd import api, synth
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (192, 320)
    nfr = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 565
    blob = synth.make_blob(seed)
    frames = [synth.make_gbuffer(H, W, 3, k) for k in range(nfr)]
    orc = oracle.DenoiseOracle(blob, H, W)
    refs = [orc.forward(x, True, k > 0) for k, x in enumerate(frames)]
    for name, impl in (("f16x3", api.DN_IMPL_MFMA_F16X3), ("f16w", api.DN_IMPL_MFMA_F16W), ("f32", api.DN_IMPL_MFMA)):
        ctx = api.Context(0)
        ctx.denoise_configure(H, W)
        ctx.load_weights(blob)
        ctx.denoise_set_impl(impl)
        y = torch.empty(3, H, W, device="cuda")
        errs = []
        for k, x in enumerate(frames):
            ctx.denoise(torch.from_numpy(x).cuda(), y, bn_batch=True, carry=k > 0)
            ctx.sync()
            errs.append(float(np.abs(y.cpu().numpy() - refs[k]).max()))
        print(f"{H}x{W} seed {seed} {name:6s} max abs err per frame:", " ".join(f"{e:.1e}" for e in errs), f"(|ref|max {np.abs(refs[-1]).max():.1f})")
        ctx.close()


if __name__ == "__main__":
    main()
