"""Do two concurrent half-batch traces (two contexts, two streams) beat one full batch?  The per-launch floor of a bounce
(~180 us: the longest ray's chain) would hide behind the other stream's launch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ai_path_tracer_denoiser_amd import api, synth
from ai_path_tracer_denoiser_amd import dist as adist
W, H, depth = 1280, 720, 8
sc = api.Scene(os.path.join(ROOT, "scenes", "cornell.txt"), res=(W, H), depth=depth)
mats = list(sc.materials) + [api.Material.from_buffer_copy(synth.STONE)]
faces, lb, ub = synth.make_atrium_mesh(262144, 565, material=len(mats) - 1)
box = api.AABB(); box.lb[:] = [float(v) for v in lb]; box.ub[:] = [float(v) for v in ub]
blob = api.scene_pack(sc.geoms, mats, faces, box)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
def mk(B):
    st = torch.cuda.Stream()
    c = api.Context(0, st.cuda_stream)
    c.pathtrace_init_packed(blob, W, H)
    c.trace_configure_batch(W, H, B)
    return c, st, torch.zeros(B, 10, H, W, device="cuda")
cams = [sc.orbit(phi=adist.pan_phi(sc.phi, k)) for k in range(N)]
def run(parts, reps=6):
    ctxs = [mk(len(p)) for p in parts]
    best = 1e9
    for r in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for (c, st, g), p in zip(ctxs, parts):
            c.pathtrace_batch([cams[k] for k in p], 1, depth, g)
        for c, st, g in ctxs: c.sync()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    for c, st, g in ctxs: c.close()
    return best * 1e3 / N
full = list(range(N))
print(f"{N} frames, one launch set:            {run([full]):.4f} ms/frame")
print(f"two halves (consecutive frames) beside each other: {run([full[:N // 2], full[N // 2:]]):.4f} ms/frame")
print(f"two halves (even / odd frames) beside each other:  {run([full[0::2], full[1::2]]):.4f} ms/frame")
print(f"three thirds beside each other:        {run([full[0::3], full[1::3], full[2::3]]):.4f} ms/frame")
print(f"two halves one after the other (same context sizes): ", end="")
c, st, g = mk(N // 2)
best = 1e9
for r in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c.pathtrace_batch(cams[:N // 2], 1, depth, g); c.pathtrace_batch(cams[N // 2:], 1, depth, g); c.sync(); torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print(f"{best * 1e3 / N:.4f} ms/frame")
