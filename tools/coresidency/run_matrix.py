#!/usr/bin/env python3
"""Bisecting the co-residency hazard (DESIGN.md "Known issue"): which VICTIM x AGGRESSOR pairs produce wrong values when their
workgroups share CUs?  Victims and aggressors come from three pools:

    synthetic   tools/coresidency/coprobe.hip: one class of operations per kernel (libcoprobe.so, built by `make` here)
    real        libaiptd.so: the bounce kernel (victim) and the denoiser's conv kernels (aggressor; PROBE impl 0/1/2)
    torch       third-party kernels: fp16 / bf16 / fp32 matmul (hipBLASLt / rocBLAS) as aggressors, an elementwise chain as victim

    python tools/coresidency/run_matrix.py RUNS [victims=all] [aggressors=all]      (comma-separated names)

Every victim is compared with its own solo run, bit for bit, on the GPU.  Single host thread: per iteration the aggressor's
launches are queued first (asynchronously, on its own stream), then the victim runs and is checked.
"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ai_path_tracer_denoiser_amd import api, synth  # noqa: E402

RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 500
want_v = sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] != "all" else None
want_a = sys.argv[3].split(",") if len(sys.argv) > 3 and sys.argv[3] != "all" else None

P = C.CDLL(os.path.join(HERE, "libcoprobe.so"))
P.coprobe_victim_name.restype = C.c_char_p
P.coprobe_aggressor_name.restype = C.c_char_p
P.coprobe_victim_run.argtypes = [C.c_int, C.c_int, C.c_void_p]
P.coprobe_aggressor_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
P.coprobe_aggressor_sync.argtypes = [C.c_void_p]
NV_THREADS, V_ITERS = 6144, 400
P.coprobe_init(NV_THREADS)

prop = torch.cuda.get_device_properties(0)
print("device", prop.name, prop.gcnArchName, "CUs", prop.multi_processor_count, flush=True)
try:
    print(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showperflevel", "--showmaxpower"], capture_output=True, text=True,
                         timeout=30).stdout[-1800:], flush=True)
except Exception as e:  # noqa: BLE001
    print("rocm-smi:", e)

# ---------------------------------------------------------------------------------------------------- real kernels (libaiptd.so)
from tests.test_gpu_frame import _mesh_scene  # noqa: E402
W, H, depth = 96, 64, 4
sc, mats, faces, box = _mesh_scene((W, H), depth)
cams = [sc.orbit(phi=sc.phi + 0.1 * k) for k in range(8)]
blob = synth.make_blob(565)


class RealTrace:
    """victim: aipt_trace of a small frame (24 workgroups per bounce launch), G-buffer compared on the GPU"""

    def __init__(self, flags, depth_, mesh):
        self.flags, self.depth = flags, depth_
        self.ctx = api.Context(0)
        self.ctx.pathtrace_init(sc.geoms, mats, faces if mesh else faces[:0], box if mesh else None, W, H)
        self.g = torch.zeros(10, H, W, device="cuda")
        torch.cuda.synchronize()
        self.ref = []
        for c in cams:
            self.ctx.pathtrace(c, 1, self.depth, self.g, self.flags)
            self.ctx.sync()
            self.ref.append(self.g.clone())
        self.k = 0

    def run(self):
        k = self.k % 8
        self.k += 1
        self.ctx.pathtrace(cams[k], 1, self.depth, self.g, self.flags)
        self.ctx.sync()
        return int((self.g.view(torch.int32) != self.ref[k].view(torch.int32)).sum().item())


class RealDenoise:
    """aggressor: forward passes of the denoiser on its own context / stream (impl 2 = split-fp16 conv, 0 = f32 MFMA, 1 = VALU)"""

    def __init__(self, impl, dh=192, dw=320):
        self.ctx = api.Context(0)
        self.ctx.load_weights(blob)
        self.ctx.denoise_configure(dh, dw)
        self.ctx.denoise_set_impl(impl)
        self.gb = torch.from_numpy(synth.make_gbuffer(dh, dw, 3, 0)).cuda()
        self.ob = torch.empty(3, dh, dw, device="cuda")
        torch.cuda.synchronize()

    def launch(self):
        self.ctx.denoise(self.gb, self.ob, bn_batch=True, carry=False)

    def sync(self):
        self.ctx.sync()


class TorchMatmul:
    def __init__(self, dtype, n=4096):
        self.st = torch.cuda.Stream()
        self.a = torch.randn(n, n, device="cuda", dtype=dtype)
        self.b = torch.randn(n, n, device="cuda", dtype=dtype)
        self.c = torch.empty(n, n, device="cuda", dtype=dtype)
        torch.cuda.synchronize()

    def launch(self):
        with torch.cuda.stream(self.st):
            torch.matmul(self.a, self.b, out=self.c)

    def sync(self):
        self.st.synchronize()


class TorchVictim:
    """victim: a chain of torch elementwise kernels with divisions and square roots on 6144 elements"""

    def __init__(self):
        self.st = torch.cuda.Stream()
        self.x = torch.rand(NV_THREADS, device="cuda") + 0.25
        torch.cuda.synchronize()
        self.ref = self._f().clone()

    def _f(self):
        with torch.cuda.stream(self.st):
            y = self.x
            for _ in range(12):
                y = torch.sqrt(y / (self.x + 0.5) + 0.125) / (y + 1.0) + 0.25
        self.st.synchronize()
        return y

    def run(self):
        return int((self._f().view(torch.int32) != self.ref.view(torch.int32)).sum().item())


class SynVictim:
    def __init__(self, kind):
        self.kind = kind
        P.coprobe_victim_reference(kind, V_ITERS)

    def run(self):
        return P.coprobe_victim_run(self.kind, V_ITERS, None)


class SynAggressor:
    def __init__(self, kind):
        self.kind = kind

    def launch(self):
        P.coprobe_aggressor_launch(self.kind, 2048, 64, 3, None)

    def sync(self):
        P.coprobe_aggressor_sync(None)


victims = {}
for k in range(P.coprobe_victim_count()):
    name = "syn:" + P.coprobe_victim_name(k).decode()
    victims[name] = (lambda kk=k: SynVictim(kk))
class KernargVictim:
    """a by-value struct array indexed per lane = vector loads from the kernel-argument segment (what p.cams[fr] compiled to)"""

    def __init__(self, scalar):
        self.scalar, self.seq = scalar, 0
        P.coprobe_kernarg_run.argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]

    def run(self):
        self.seq += 1
        fl = (C.c_int * 2)()
        n = P.coprobe_kernarg_run(self.seq, self.scalar, None, fl)
        if n and not getattr(self, "said", False):
            self.said = True
            print(f"      first hit: {n} threads, {fl[0]}..{fl[1]} (lanes {fl[0] & 63}..{fl[1] & 63})", flush=True)
        return n


victims["syn:kernarg_vload"] = lambda: KernargVictim(0)
victims["syn:kernarg_sload"] = lambda: KernargVictim(1)
victims["real:trace_prims_d1"] = lambda: RealTrace(2, 1, False)          # one launch of trace_bounce<true,false,false>, no AA
victims["real:trace_mesh_d4"] = lambda: RealTrace(3, 4, True)
victims["torch:divsqrt_chain"] = TorchVictim
aggressors = {"none": None}
for k in range(1, P.coprobe_aggressor_count()):
    aggressors["syn:" + P.coprobe_aggressor_name(k).decode()] = (lambda kk=k: SynAggressor(kk))
aggressors["real:conv_f16x3"] = lambda: RealDenoise(2)
aggressors["real:conv_f32mfma"] = lambda: RealDenoise(0)
aggressors["real:conv_f16x3_big"] = lambda: RealDenoise(2, 736, 1280)
aggressors["torch:matmul_f16"] = lambda: TorchMatmul(torch.float16)
aggressors["torch:matmul_bf16"] = lambda: TorchMatmul(torch.bfloat16)
aggressors["torch:matmul_f32"] = lambda: TorchMatmul(torch.float32, 2048)

vn = [v for v in victims if want_v is None or any(w in v for w in want_v)]
an = [a for a in aggressors if want_a is None or any(w in a for w in want_a)]
print(f"{RUNS} victim runs per pair; victims {vn}; aggressors {an}", flush=True)
agg_objs = {a: (aggressors[a]() if aggressors[a] else None) for a in an}
for v in vn:
    vic = victims[v]()
    for a in an:
        ag = agg_objs[a]
        bad_runs = bad_words = 0
        t0 = time.time()
        for r in range(RUNS):
            if ag:
                ag.launch()
            n = vic.run()
            if n:
                bad_runs += 1
                bad_words += n
            if ag and r % 8 == 7:
                ag.sync()
        if ag:
            ag.sync()
        print(f"victim {v:24s} aggressor {a:22s} runs {RUNS} bad {bad_runs} (words {bad_words}) {time.time() - t0:.1f} s", flush=True)
