"""CPU launch cost vs GPU time of one frame (is the host the bottleneck?)"""
import sys, time, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from ai_path_tracer_denoiser_amd import api, synth
import ai_path_tracer_denoiser_amd.dist as adist
ROOT="/root/repo"
sc = api.Scene(os.path.join(ROOT, "scenes", "cornell.txt"), res=(1280, 720), depth=8)
ctx = api.Context(0)
ctx.pathtrace_init(sc.geoms, sc.materials, sc.faces, None)
ctx.load_weights(synth.make_blob(565))
ctx.frame_configure(1280, 720)
out = torch.empty(3, 720, 1280, device="cuda")
cam = sc.camera
for k in range(5): ctx.frame(cam, 1, 8, out, carry=k>0)
torch.cuda.synchronize()
for mode in ("frame", "trace-only", "denoise-only"):
    t0=time.perf_counter()
    N=50
    for k in range(N):
        if mode=="frame": ctx.frame(cam, 1, 8, out)
        elif mode=="trace-only": api.lib().aipt_trace(ctx._h, api.C.byref(cam), 1, 8, api.TRACE_DEFAULT, api._P(ctx.gbuffer()[0]), 736, 1280)
        else: api.lib().aipt_denoise(ctx._h, api._P(ctx.gbuffer()[0]), api._P(out.data_ptr()), 3)
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
    print(f"{mode}: cpu enqueue {1e6*(t1-t0)/N:.0f} us/frame, total {1e6*(t2-t0)/N:.0f} us/frame")

# pipelined: trace(k+1) on the side stream during denoise(k)
cams = []
for k in range(2):
    c = api.Camera.from_buffer_copy(bytes(cam))
    api.lib().aipt_camera_orbit(c, sc.zoom, sc.phi + 0.01 * k, sc.theta)
    cams.append(c)
for order in ("frame-then-prefetch", "no-prefetch"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 100
    for k in range(N):
        ctx.frame(cams[k & 1], 1, 8, out)
        if order == "frame-then-prefetch" and k + 1 < N:
            ctx.frame_prefetch(cams[(k + 1) & 1], 1, 8)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{order}: {1e6*(t2-t0)/N:.0f} us/frame")
