#!/usr/bin/env python3
"""Known issue probe: a context path-traces the same 8 frames over and over (comparing every G-buffer with its first run) while
a second context on the same GPU denoises / traces / idles in another host thread.

    python tools/concurrency_probe.py N denoise|trace|memset|none [DENOISER_H DENOISER_W]
    PROBE_IMPL=0|1|2 (aggressor conv implementation: f32 MFMA, VALU, split-fp16), PROBE_FLAGS (victim AIPT_TRACE_* bits),
    PROBE_DEPTH, PROBE_NOMESH=1, PROBE_LOCK=1 (host calls of the two threads never overlap), PROBE_COMPARE_ON_GPU=1, PROBE_CU_SPLIT=1

Round 2 (library built WITH packed fp32): beside the split-fp16 conv kernel 3-8 % of the traced frames differed -- runs of
consecutive lanes ending at lane 63 of a wave -- even for a depth-1, no-AA, primitives-only trace (one kernel launch); 0 beside
another trace, a memset loop, the f32-MFMA or the VALU conv kernels.  Round 3 (tools/coresidency/): the victims are the
packed-fp32 VALU instructions hipcc emitted into the bounce kernels, the aggressor any wave that issues fp16 / bf16 MFMAs with
gaps between them; reproduced with two synthetic kernels (pk_f32_mfma_erratum.hip).  The library is now built without packed
fp32 and this probe reports 0 of 6000 without CU masks (gpurun_out/co/orig_probe_nopk.log; AIPT_LIB=<a build with packed fp32>
brings the failures back)."""
import sys, os, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ai_path_tracer_denoiser_amd import api, synth
from tests.test_gpu_frame import _mesh_scene
W, H, depth = 96, 64, int(os.environ.get('PROBE_DEPTH', '4'))
sc, mats, faces, box = _mesh_scene((W, H), depth)
FLAGS = int(os.environ.get('PROBE_FLAGS', '3'))
if os.environ.get('PROBE_NOMESH'):
    import numpy as _np
    faces = faces[:0]
cams = [sc.orbit(phi=sc.phi + 0.1 * k) for k in range(8)]
N = int(sys.argv[1]); what = sys.argv[2]          # what the other thread does: denoise | trace | none | memset
blob = synth.make_blob(565)
def masked_stream(lo, hi):
    """a HIP stream restricted to CUs lo..hi-1 (hipExtStreamCreateWithCUMask), as a raw handle"""
    import ctypes
    import glob
    hip = ctypes.CDLL(glob.glob(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so*'))[0])   # the runtime torch has loaded
    st = ctypes.c_void_p()
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if lo <= w * 32 + b < hi) for w in range(8)])
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return st.value
SPLIT = os.environ.get('PROBE_CU_SPLIT')                 # "1": victim on CUs 0-127, the other context on CUs 128-255
A = api.Context(0, stream=masked_stream(0, 128) if SPLIT else None); A.pathtrace_init(sc.geoms, mats, faces, box, W, H)
B = api.Context(0, stream=masked_stream(128, 256) if SPLIT else None); B.pathtrace_init(sc.geoms, mats, faces, box, W, H); B.load_weights(blob)
DH, DW = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (H, W)
B.denoise_configure(DH, DW)
if os.environ.get('PROBE_IMPL'): B.denoise_set_impl(int(os.environ['PROBE_IMPL']))
BNB = os.environ.get('PROBE_BN', '1') == '1'
g = torch.zeros(10, H, W, device="cuda"); torch.cuda.synchronize()
ref = []
for c in cams:
    A.pathtrace(c, 1, depth, g, FLAGS); A.sync(); ref.append(g.cpu().numpy().copy())
refg = [torch.from_numpy(r).cuda() for r in ref]
ONGPU = bool(os.environ.get('PROBE_COMPARE_ON_GPU'))
LOCK = threading.Lock() if os.environ.get('PROBE_LOCK') else None
import contextlib
def locked():
    return LOCK if LOCK else contextlib.nullcontext()
stop = False
def other():
    gb = torch.from_numpy(synth.make_gbuffer(DH, DW, 3, 0)).cuda(); ob = torch.empty(3, DH, DW, device="cuda"); g2 = torch.zeros(10, H, W, device="cuda")
    torch.cuda.synchronize()
    n = 0
    while not stop:
        if what == "denoise":
            with locked(): B.denoise(gb, ob, bn_batch=BNB, carry=False)
        elif what == "trace": B.pathtrace(cams[n % 8], 1, depth, g2)
        elif what == "memset": ob.zero_()
        else: time.sleep(0.001)
        n += 1
        if n % 8 == 0: B.sync()
    B.sync()
    print("other thread did", n, what)
t = threading.Thread(target=other); t.start()
bad = 0
for it in range(N):
    k = it % 8
    with locked(): A.pathtrace(cams[k], 1, depth, g, FLAGS)
    A.sync()
    if ONGPU:                                           # compare on the device: no device-to-host copy of the G-buffer
        n = int((g.view(torch.int32) != refg[k].view(torch.int32)).sum())
        if n: bad += 1; print("iter", it, "frame", k, "words", n, "(compared on the GPU)", flush=True)
        continue
    r = g.cpu().numpy()
    n = int((r.view(np.uint32) != ref[k].view(np.uint32)).sum())
    if n:
        bad += 1
        w = np.argwhere(r.view(np.uint32) != ref[k].view(np.uint32))
        per_plane = [int((r[pl].view(np.uint32) != ref[k][pl].view(np.uint32)).sum()) for pl in range(10)]
        pix = np.unique(w[:, 1] * W + (W - 1 - w[:, 2]))          # un-flipped pixel index (h-flipped destination)
        blocks = sorted(set((pix // 256).tolist()))
        vals = [(float(r[tuple(i)]), float(ref[k][tuple(i)])) for i in w[:3]]
        prev = ref[(k - 1) % 8]                             # the frame traced into this buffer just before
        stale = sum(1 for i in w if r[tuple(i)] == prev[tuple(i)])
        print("   of the", n, "differing words", stale, "hold the PREVIOUS frame's value at that position", flush=True)
        print("iter", it, "frame", k, "words", n, "per plane", per_plane, "pixels", len(pix), "pixel idx range", int(pix.min()), int(pix.max()), "256-blocks", blocks[:12], "got/ref", vals, flush=True)
stop = True; t.join()
print("other =", what, "frames", N, "bad", bad)
