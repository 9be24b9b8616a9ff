#!/usr/bin/env python3
"""Time the path trace alone (HIP events on the context stream) on the mesh workloads: ms per frame at 1280x720 and at 4x the
pixels (the throughput regime), per-bounce launch times, live counts.  AIPT_LIB selects a kernel-variant build.

    python tools/trace_bench.py [--mesh 262144] [--kind atrium|reflective|living] [--depth 8] [--frames 10]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", type=int, default=262144)
    ap.add_argument("--kind", default="atrium")
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--sizes", default="1280x720,2560x1440")
    ap.add_argument("--flags", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    import torch
    from ai_path_tracer_denoiser_amd import api, synth
    from ai_path_tracer_denoiser_amd import dist as adist
    ctx = api.Context(0)
    for size in args.sizes.split(","):
        W, H = [int(v) for v in size.split("x")]
        sc = api.Scene(os.path.join(ROOT, "scenes", "cornell.txt"), res=(W, H), depth=args.depth)
        mats = list(sc.materials)
        first = len(mats)
        if args.kind == "living":
            faces, lb, ub, recs = synth.make_living_room_mesh(args.mesh, 565, first_material=first)
            mats += [api.Material.from_buffer_copy(r) for r in recs]
        else:
            mats += [api.Material.from_buffer_copy(synth.STONE), api.Material.from_buffer_copy(synth.MIRROR)]
            refl = first + 1 if args.kind == "reflective" else first
            faces, lb, ub = synth.make_atrium_mesh(args.mesh, 565, material=first, floor_material=refl, column_material=refl)
        box = api.AABB(); box.lb[:] = [float(v) for v in lb]; box.ub[:] = [float(v) for v in ub]
        ctx.pathtrace_init(sc.geoms, mats, faces, box, W, H)
        B = args.batch
        if B > 1:
            ctx.trace_configure_batch(W, H, B)
        g = torch.zeros(B, 10, H, W, device="cuda")
        torch.cuda.synchronize()
        cams = [sc.orbit(phi=adist.pan_phi(sc.phi, k)) for k in range((args.frames + 2) * B)]

        def trace(k):
            if B > 1:
                ctx.pathtrace_batch(cams[k * B:(k + 1) * B], 1, args.depth, g, args.flags)
            else:
                ctx.pathtrace(cams[k], 1, args.depth, g[0], args.flags)
        for k in range(2):
            trace(k)
        ctx.sync()
        ctx.trace_profile_begin(args.frames, 1)
        ctx.timer_start()
        for k in range(args.frames):
            trace(2 + k)
        ms = ctx.timer_stop() / args.frames / B
        per, calls = ctx.trace_profile_end(args.depth)
        n = ctx.live_counts(args.depth) // B
        print(f"{size} {args.kind} {args.mesh} tris depth {args.depth} batch {B}: trace {ms:.3f} ms/frame; bounce launches (us): "
              f"{[round(1e3 * v / max(1, calls), 1) for v in per]}; rays {int(n[:-1].sum())} -> {n[:-1].sum() / ms / 1e6:.2f} Grays/s")
    ctx.close()


if __name__ == "__main__":
    main()
