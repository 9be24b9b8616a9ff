#!/usr/bin/env python3
"""Generate golden vectors for the denoiser by importing the REFERENCE model.

Runs only in the build container, where /root/reference exists (it does not travel to the
GPU box).  Imports /root/reference/training/recurrent_autoencoder_model.py unmodified, loads
this repo's deterministic synthetic weights into it (load_state_dict), runs it on this repo's
deterministic synthetic G-buffers and stores the OUTPUTS (data, not source) as .npz fixtures:

  * bn=batch  : model left in train() mode -- the semantics of the reference's shipped
                TorchScript (convert_to_torchscript.py:26-30 never calls .eval())
  * bn=running: model.eval() -- what training/test.py:35 uses
  * hidden reset: model(x, 0) each frame (forward j==0 re-inits hidden, model.py:121-128)
  * hidden carry: model(x_0, 0), model(x_1, 1), ... (test.py:46-48)

Inputs and weights are NOT stored: tests regenerate them bit-identically from
ai_path_tracer_denoiser_amd/synth.py.  Stored per case: every output frame in full, plus for
the 6 hidden states after the last frame per-channel mean/mean-square (fp64) and 512 strided
samples.

Round 5 -- cases at the sizes where the register-staged kernel `conv3x3_f16x3r` runs (selected from 200 000 pixels per level,
include/aiptd.h AIPT_DN_OPT_R_MINPIX): one 384x640 frame stored in full (level 0 on that kernel) and one 736x1280 frame -- the
benchmark size, levels 0 and 1 on it -- stored as 65 536 strided samples + per-channel fp64 mean / mean-square of the output.

Round 6 -- `b_carry_384x640`: two frames with the hidden state carried (model(x0, 0), model(x1, 1)): the reset cases above multiply
the hidden half of every `layer2.0` input (recurrent_autoencoder_model.py:64-67 `torch.cat((out1, self.hidden))`) by zeros, so the
register-staged kernel's hidden-channel weights met reference output only at <= 96x160 through the LDS-tiled kernel.  Stored as
131 072 strided samples over both frames + per-frame per-channel fp64 moments.

Usage: python tests/golden/gen_denoise_goldens.py [case-name ...]      (no names: all cases)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/training")

import recurrent_autoencoder_model as M  # noqa: E402  (the reference, imported as-is)

from ai_path_tracer_denoiser_amd import arch, synth  # noqa: E402

CASES = [
    # name,            H,  W,  wseed, iseed, bn,        frames (carry if >1)
    ("b_reset_64",     64, 64,  565,  0, "batch",   1),
    ("r_reset_64",     64, 64,  565,  0, "running", 1),
    ("b_carry_64",     64, 64,  565,  0, "batch",   3),
    ("r_carry_64",     64, 64,  565,  0, "running", 3),
    ("b_carry_96x160", 96, 160, 566,  1, "batch",   2),
    ("r_reset_96x160", 96, 160, 566,  1, "running", 1),
    ("b_reset_384x640", 384, 640, 567, 2, "batch",  1),
    ("b_reset_736x1280", 736, 1280, 568, 3, "batch", 1),      # stored as samples + moments (SAMPLED below)
    ("b_carry_384x640", 384, 640, 569, 4, "batch",  2),       # [r6] hidden CARRIED into frame 1 at a size conv3x3_f16x3r runs
    ("r_carry_384x640", 384, 640, 570, 5, "running", 2),      # [r6] the same with running-statistics BatchNorm (model.eval())
    ("b_carry_736x1280", 736, 1280, 571, 6, "batch", 2),      # [r6] ... and at the benchmark size (levels 0 and 1 on conv3x3_f16x3r)
]
SAMPLED = {"b_reset_736x1280": 65536, "b_carry_384x640": 131072, "r_carry_384x640": 131072, "b_carry_736x1280": 131072}


def hidden_summary(h):
    h = h.astype(np.float64)
    c = h.shape[0]
    flat = h.reshape(c, -1)
    samples = h.reshape(-1)
    idx = np.linspace(0, samples.size - 1, 512).astype(np.int64)
    return flat.mean(axis=1), (flat * flat).mean(axis=1), samples[idx].astype(np.float32), idx


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    only = set(sys.argv[1:])
    for name, H, W, wseed, iseed, bn, frames in CASES:
        if only and name not in only:
            continue
        params = synth.make_params(wseed)
        model = M.AutoEncoder(10)
        sd = {k: torch.from_numpy(np.array(v)) for k, v in arch.state_dict_from_params(params).items()}
        missing = model.load_state_dict(sd, strict=False)
        # only num_batches_tracked buffers may be missing
        assert all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
        assert not missing.unexpected_keys
        if bn == "running":
            model.eval()
        else:
            model.train()
        outs = []
        with torch.no_grad():
            for j in range(frames):
                x = torch.from_numpy(synth.make_gbuffer(H, W, iseed, j))[None]
                y = model(x, j)          # j==0 resets hidden; j>0 carries (model.py:120-128)
                outs.append(y[0].numpy().copy())
        blocks = [model.encoder1[0], model.encoder2[0], model.encoder3[0], model.encoder4[0],
                  model.encoder5[0], model.bottleneck]
        store = {"out": np.stack(outs).astype(np.float32),
                 "meta": np.array([H, W, wseed, iseed, frames, 1 if bn == "batch" else 0], np.int64)}
        if name in SAMPLED:
            o = store.pop("out")
            flat = o.reshape(-1)
            idx = np.linspace(0, flat.size - 1, SAMPLED[name]).astype(np.int64)
            o64 = o.astype(np.float64).reshape(frames, 3, -1)
            store.update(out_idx=idx, out_samples=flat[idx], out_mean=o64.mean(axis=2), out_msq=(o64 * o64).mean(axis=2),
                         out_absmax=np.abs(o64).max(axis=2))
        for lvl, b in enumerate(blocks):
            m1, m2, smp, idx = hidden_summary(b.hidden[0].numpy())
            store[f"h{lvl}_mean"] = m1
            store[f"h{lvl}_msq"] = m2
            store[f"h{lvl}_samples"] = smp
            store[f"h{lvl}_idx"] = idx
        path = os.path.join(HERE, f"denoise_{name}.npz")
        np.savez_compressed(path, **store)
        o = store["out"] if "out" in store else store["out_samples"]
        print(name, "out range", float(o.min()), float(o.max()), os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
