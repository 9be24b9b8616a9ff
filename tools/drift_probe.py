#!/usr/bin/env python3
"""Carried-hidden-state drift against a truth: the denoiser restatement in double throughout (oracle/liboracle64.so) over N
frames of one recurrent sequence, and the max abs error per frame against it of (a) the fp32 CPU oracle (the arithmetic class
of the reference's PyTorch), (b) the split-fp16 MFMA path, (c) the exact-fp32 MFMA path, (d) split-fp16 with fp16 weights
against the truth run on the rounded weights.

    python tools/drift_probe.py [H W [frames [seed [r_minpix]]]]   (GPU box; default 192 320 32 565, library kernel selection)
                                                                 -> JSON on stdout"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(H=192, W=320, nfr=32, seed=565, impls=("f16x3", "f32"), r_minpix=None):
    """r_minpix: AIPT_DN_OPT_R_MINPIX for the GPU runs (0: every level the kernel can take runs on conv3x3_f16x3r, the
    register-staged kernel of the benchmark's big levels; None: the library's default, 200 000 pixels)."""
    import torch
    import oracle
    from ai_path_tracer_denoiser_amd import api, synth
    blob = synth.make_blob(seed)
    frames = [synth.make_gbuffer(H, W, 3, k) for k in range(nfr)]
    o64 = oracle.DenoiseOracle(blob, H, W, fp64=True)
    o32 = oracle.DenoiseOracle(blob, H, W)
    truth = [o64.forward(x, True, k > 0) for k, x in enumerate(frames)]
    ref32 = [o32.forward(x, True, k > 0) for k, x in enumerate(frames)]
    out = {"H": H, "W": W, "frames": nfr, "seed": seed, "bn": "batch", "hidden": "carried",
           "truth": "oracle/denoise_oracle.c -DORC_DN_FP64 (double activations, products, sums, BN)",
           "max_abs_truth": float(np.abs(truth[-1]).max()),
           "oracle_fp32": [float(np.abs(a - b).max()) for a, b in zip(ref32, truth)]}
    codes = {"f16x3": api.DN_IMPL_MFMA_F16X3, "f32": api.DN_IMPL_MFMA}
    for name in impls:
        ctx = api.Context(0)
        ctx.denoise_configure(H, W)
        ctx.load_weights(blob)
        ctx.denoise_set_impl(codes[name])
        if r_minpix is not None:
            ctx.denoise_set_option(api.DN_OPT_R_MINPIX, r_minpix)
        y = torch.empty(3, H, W, device="cuda")
        errs = []
        for k, x in enumerate(frames):
            ctx.denoise(torch.from_numpy(x).cuda(), y, bn_batch=True, carry=k > 0)
            ctx.sync()
            errs.append(float(np.abs(y.cpu().numpy() - truth[k]).max()))
        out["gpu_" + name] = errs
        out["kernels_" + name] = sorted({ctx.layer_info(l)["kernel"] for l in range(28)})
        ctx.close()
    return out


if __name__ == "__main__":
    a = sys.argv[1:]
    H, W = (int(a[0]), int(a[1])) if len(a) > 1 else (192, 320)
    r = run(H, W, int(a[2]) if len(a) > 2 else 32, int(a[3]) if len(a) > 3 else 565, r_minpix=int(a[4]) if len(a) > 4 else None)
    r["r_minpix"] = int(a[4]) if len(a) > 4 else "default"
    print(json.dumps(r))
    for k in ("oracle_fp32", "gpu_f16x3", "gpu_f32"):
        print(f"{k:12s}", " ".join(f"{e:.1e}" for e in r[k]), file=sys.stderr)
