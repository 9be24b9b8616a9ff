#!/usr/bin/env python3
"""Where a conv3x3_f16x3 workgroup spends its cycles (library built with `make EXTRA=-DAIPT_CONV_PHASES`): wave 0 of every
workgroup stamps s_memtime between the phases of the chunk loop; sums per launch are printed as cycles per tile.

    python tools/conv_phases.py            # layers enc1.l2a, enc1.l2b, enc2.l2a, dec2.c1 at 736x1280
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = ["setup (staging units)", "fetch(0) issue", "barrier 0", "wait loads (vmcnt 0)", "stash (BN, split, LDS write)",
         "barrier 1", "fetch(c+1) issue", "LDS reads + MFMA", "barrier 2", "epilogue",
         "fetch(0) round trip (instr. only)", "BN table (loads + math)", "halo zero fill", "-"]
ORDER = [0, 1, 10, 11, 12, 2, 3, 4, 5, 6, 7, 8, 9]


def main():
    import torch
    from ai_path_tracer_denoiser_amd import api, arch, synth
    H, W = 736, 1280
    ctx = api.Context(0)
    L = api.lib()
    L.aipt_debug_conv_phases.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_uint]
    ctx.denoise_configure(H, W)
    ctx.load_weights(synth.make_blob(1))
    x = torch.from_numpy(synth.make_gbuffer(H, W, 3, 0)).cuda()
    y = torch.empty(3, H, W, device="cuda")
    for _ in range(3):
        ctx.denoise(x, y, bn_batch=True, carry=True)
    ctx.sync()
    pad16 = lambda c: (c + 15) // 16 * 16
    # (name, level, nchunks, cout): nchunks = chunks of a + chunks of b
    layers = [("enc1.l2a 64->32", 0, 4, 32), ("enc1.l2b 32->32", 0, 2, 32), ("enc2.l2a 86->43", 1, 6, 43),
              ("enc2.l2b 43->43", 1, 3, 43), ("dec2.c1 86->32", 1, 6, 32), ("enc3.l2a 114->57", 2, 8, 57),
              ("enc4.l2b 76->76", 3, 5, 76), ("enc5.l2b 101->101", 4, 7, 101), ("bott.l1 101->101", 5, 7, 101), ("dec5.c2 76->76", 4, 5, 76)]
    if len(sys.argv) > 1 and sys.argv[1] == "small":               # the LDS-tiled kernel's own levels only (the big ones run conv3x3_f16x3r)
        layers = layers[5:]
    out = (ctypes.c_ulonglong * 16)()
    for name, lvl, nch, cout in layers:
        key = ((W >> lvl) << 16) | (nch << 8) | cout
        ctx._ck(L.aipt_debug_conv_phases(ctx._h, out, key))
        n = 5
        for _ in range(n):
            ctx.denoise(x, y, bn_batch=True, carry=True)
        ctx._ck(L.aipt_debug_conv_phases(ctx._h, out, 0))
        v = np.array(list(out), dtype=np.float64)
        tiles = v[15]
        if tiles == 0:
            print(f"{name}: no launch matched key {key:#x}")
            continue
        print(f"{name}: {int(tiles / n)} workgroups per launch, {v[14] / tiles:.0f} cycles per workgroup (s_memtime, 100 MHz ticks x ?)")
        for k in ORDER:
            nm = NAMES[k]
            per_chunk = f"  ({v[k] / tiles / nch:7.0f} per chunk)" if 3 <= k <= 8 else ""
            print(f"    {nm:32s} {v[k] / tiles:9.0f}  {100 * v[k] / v[14]:5.1f} %{per_chunk}")
    ctx.close()


if __name__ == "__main__":
    main()
