#!/usr/bin/env python3
"""BVH-walk statistics of the mesh workload, per bounce (needs a library built with `make -C ai_path_tracer_denoiser_amd/csrc
EXTRA=-DAIPT_TRACE_STATS`; rebuild without it afterwards).  Prints lane-level work (node visits, triangle tests per ray) against
wave-level loop trips (what a wave actually executes: the union over its 64 lanes)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from ai_path_tracer_denoiser_amd import api, synth
    ntri = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1           # frames traced together (interleaved batch)
    W, H, depth = 1280, 720, int(os.environ.get("STATS_DEPTH", 8))     # STATS_DEPTH=1: the phase shares of bounce 0 alone
    sc = api.Scene(os.path.join(ROOT, "scenes", "cornell.txt"), res=(W, H), depth=depth)
    stone = api.Material.from_buffer_copy(synth.STONE)
    mats = list(sc.materials) + [stone]
    faces, lb, ub = synth.make_atrium_mesh(ntri, 565, material=len(mats) - 1)
    box = api.AABB(); box.lb[:] = [float(v) for v in lb]; box.ub[:] = [float(v) for v in ub]
    ctx = api.Context(0)
    ctx.pathtrace_init(sc.geoms, mats, faces, box, W, H)
    from ai_path_tracer_denoiser_amd import dist as adist
    cams = [sc.orbit(phi=adist.pan_phi(sc.phi, k)) for k in range(B)]
    if B > 1:
        ctx.trace_configure_batch(W, H, B)
    g = torch.zeros(B, 10, H, W, device="cuda")

    def trace(d):
        if B > 1:
            ctx.pathtrace_batch(cams, 1, d, g)
        else:
            ctx.pathtrace(sc.camera, 1, d, g[0])
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    L = api.lib()
    prev = None
    for d in range(1, depth + 1 if not os.environ.get("PHASES_ONLY") else 1):              # depth d minus depth d-1 = bounce d-1 alone
        L.aipt_debug_trace_stats(ctx._h, out, 1)
        trace(d)
        ctx.sync()
        ctx._ck(L.aipt_debug_trace_stats(ctx._h, out, 0))
        cur = np.array(list(out), np.float64)
        n = ctx.live_counts(d)
        delta = cur - (prev if prev is not None else 0)
        rays = int(n[d - 1])
        waves = (rays + 63) // 64
        print(f"bounce {d - 1}: rays {rays:7d}  node visits/ray {delta[0] / rays:6.1f}  tri tests/ray {delta[2] / rays:5.1f}  "
              f"leaf visits/ray {delta[4] / rays:5.1f} | per wave: node-loop trips {delta[1] / waves:6.1f}  leaf-loop trips {delta[3] / waves:5.1f}  "
              f"SIMD efficiency of node visits {delta[0] / max(1.0, delta[1] * 64):.2f}")
        print(f"          rays that walked the BVH {int(delta[8:15].sum()):7d}; by node visits <=4,8,16,32,64,128,more: "
              f"{[int(v) for v in delta[8:15]]}; max so far {int(cur[5])}; max stack {int(cur[6])}, node visits with sp > 8: {int(delta[7])}, > 12: {int(delta[15])}")
        prev = cur
    # phase cycle sums (one lane per wave): prologue, primitives, BVH walk + winner fetch, shade/scatter/store
    ph = (C.c_ulonglong * 16)()
    L.aipt_debug_trace_stats(ctx._h, ph, 1)
    trace(depth)
    ctx.sync()
    L.aipt_debug_trace_stats(ctx._h, ph, 2)
    v = np.array(list(ph)[:4], np.float64)
    print("phase shares (cycles summed over waves, whole frame): prologue %.2f  primitives %.2f  walk %.2f  shade %.2f  (total %.3g cycles)"
          % tuple(list(v / v.sum()) + [v.sum()]))


if __name__ == "__main__":
    main()
