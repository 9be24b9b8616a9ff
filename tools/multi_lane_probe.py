"""How many trace lanes (independent sets of bounce launches side by side on their own streams) does a 20-frame call want, with the
split walk (round 5 compared it with the pooled walk, AIPT_TRACE_POOL=0/1, removed in round 6)?  N contexts, one stream each.
    python tools/multi_lane_probe.py [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ai_path_tracer_denoiser_amd import api, synth
from ai_path_tracer_denoiser_amd import dist as adist
W, H, depth = 1280, 720, 8
sc = api.Scene(os.path.join(ROOT, "scenes", "cornell.txt"), res=(W, H), depth=depth)
mats = list(sc.materials) + [api.Material.from_buffer_copy(synth.STONE)]
faces, lb, ub = synth.make_atrium_mesh(262144, 565, material=len(mats) - 1)
box = api.AABB(); box.lb[:] = [float(v) for v in lb]; box.ub[:] = [float(v) for v in ub]
blob = api.scene_pack(sc.geoms, mats, faces, box)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cams = [sc.orbit(phi=adist.pan_phi(sc.phi, k)) for k in range(N)]
def mk(B):
    st = torch.cuda.Stream()
    c = api.Context(0, st.cuda_stream)
    c.pathtrace_init_packed(blob, W, H)
    if B > 1: c.trace_configure_batch(W, H, B)
    return c, torch.zeros(B, 10, H, W, device="cuda")
def run(parts, reps=6):
    ctxs = [mk(len(p)) for p in parts]
    best = 1e9
    for r in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for (c, g), p in zip(ctxs, parts):
            if len(p) > 1: c.pathtrace_batch([cams[k] for k in p], 1, depth, g)
            else: c.pathtrace(cams[p[0]], 1, depth, g[0])
        for c, g in ctxs: c.sync()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    for c, g in ctxs: c.close()
    return best * 1e3 / N
full = list(range(N))
print("AIPT_TRACE_LANES=1 inside every context")
for L in (1, 2, 3, 4, 5, 10, 20):
    if L > N: break
    parts = [full[k::L] for k in range(L)]
    print(f"{L:2d} lanes of {len(parts[0]):2d} frames: {run(parts):.4f} ms/frame", flush=True)
