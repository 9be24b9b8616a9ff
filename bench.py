#!/usr/bin/env python3
"""bench.py -- denoised frames/s @1280x720, 1 spp, depth 8 (BASELINE.json metric) on N MI355X GPUs of one node.

Default workload = BASELINE.json configs[2], the configuration north_star quotes the target on: Cornell walls + the
procedural Sponza-like atrium mesh (262 144 triangles, BVH), 1280x720, 1 spp, depth 8, 300-frame orbit pan, recurrent hidden
state carried, BatchNorm in batch-statistics mode (what the reference's shipped TorchScript computes, SURVEY F4 -- the most
expensive of the four denoiser modes).  --config 0/1/3/4 select the other BASELINE configs (parity-test shapes, not bench
lines).  A "step" is one frame: path trace -> device G-buffer -> denoise (aipt_frame).  Inputs (scene, BVH, weights) are
resident in HBM before the timed region; nothing crosses PCIe per frame.

    python bench.py --gpus N --steps K --warmup W
    N > 1 without WORLD_SIZE in the environment: re-executes itself under torch.distributed.run with N ranks (fails if the
    node has fewer GPUs); under torch.distributed.run: one rank per GPU, frames are sharded in contiguous chunks, rank 0
    packs the scene (BVH built once) and the weights and broadcasts them over RCCL; no per-frame collective -> "scaling": "weak".

Frame batches (aipt_frames; results bit-identical to frame-by-frame rendering, tests/test_gpu_frame.py): a call holds up to 24
consecutive frames (--batch; at most 32).  Their traces share one set of bounce launches per up to 24 frames (a single 1280x720 frame leaves most of the
chip idle in its later bounces; the frames are interleaved pixel by pixel, so that neighbouring lanes walk near-identical rays, and
a wave's idle lanes take over subtrees of its busy lanes' walks).  Their denoiser passes run on two streams, frame n+1 entering an encoder level when frame n has left
it, so the many small launches of one frame's deep levels run beside the full-size layers of the other; the hidden state is
carried through the batch.  --batch 1 is the frame-by-frame sequence.  The trace / denoise split of the JSON line is
measured on one single frame after the timed region.

Prints ONE JSON line on rank 0.  Extra objects: "roofline" for the kernel that dominates THIS workload (HIP events on the
launch stream, on an untimed second pass over the timed region's frames; the runner-up kernel under "roofline_other"), "cpu_baseline" (the CPU oracle timed
on the host cores, N=1 only), "frame" (ms split and the whole-frame HBM fraction).
"""
import argparse
import json
import os
import subprocess
import sys
import time

# idle OpenMP threads sleep instead of spinning: the CPU oracle's pool and torch's CPU pool (cpu_baseline / smoke) share this
# process, and two pools of busy-waiting threads starve each other (tests/conftest.py)
if os.path.exists("/dev/kfd"):
    for _k, _v in (("OMP_WAIT_POLICY", "PASSIVE"), ("GOMP_SPINCOUNT", "0"), ("KMP_BLOCKTIME", "0")):
        os.environ.setdefault(_k, _v)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MI355X_HBM_BPS = 8.0e12          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
MI355X_FP32_MFMA_TFLOPS = 157.3  # f32-input MFMA dense peak = fp32 vector peak
MI355X_FP16_MFMA_TFLOPS = 2500.0  # fp16/bf16 dense MFMA peak (no sparsity)

# BASELINE.json configs[i]: (width, height, depth, mesh kind, triangles, conv impl)
CONFIGS = {
    0: (256, 256, 4, None, 0, "f16x3"),
    1: (1280, 720, 8, None, 0, "f16x3"),
    2: (1280, 720, 8, "atrium", 262144, "f16x3"),
    3: (1280, 720, 8, "reflective", 262144, "f16x3"),
    4: (1920, 1080, 12, "living", 524288, "f16w"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS),
                    help="BASELINE.json configs[i]; 2 (default) = diffuse Sponza-like mesh 1280x720 depth 8, the north-star target")
    ap.add_argument("--scene-only", action="store_true", help="no mesh (= --config 1: Cornell box, 7 primitives)")
    ap.add_argument("--width", type=int)
    ap.add_argument("--height", type=int)
    ap.add_argument("--depth", type=int)
    ap.add_argument("--scene", default=os.path.join(ROOT, "scenes", "cornell.txt"))
    ap.add_argument("--mesh", type=int, metavar="NTRI", help="triangle count of the procedural mesh (0: none)")
    ap.add_argument("--mesh-kind", choices=["atrium", "reflective", "living"])
    ap.add_argument("--bn", choices=["batch", "running"], default="batch")
    ap.add_argument("--hidden", choices=["carry", "reset"], default="carry")
    ap.add_argument("--impl", choices=["f32", "f16x3", "f16w"],
                    help="conv arithmetic: f32-input MFMA (exact fp32 chain), split-fp16 MFMA (default), or split-fp16 activations x fp16 weights")
    ap.add_argument("--batch", type=int, default=None,
                    help="trace this many consecutive frames with one set of launches (aipt_frames; bit-identical frames; "
                         "1 = frame by frame, aipt_frame) and run the denoiser passes of consecutive frames on two streams, "
                         "level by level behind each other.  Default: 32 (traced 16 at a time)")
    ap.add_argument("--prefetch", action="store_true",
                    help="frame by frame (--batch 1) only: trace frame k+1 while frame k is denoised, the two on disjoint halves "
                         "of the CUs (aipt_frame_prefetch; one frame of latency, same bits)")
    ap.add_argument("--trace-flags", type=int, default=None, help="AIPT_TRACE_* bits (default AA | COMPACT)")
    ap.add_argument("--dn-opt", action="append", metavar="OPTION=VALUE", help="aipt_denoise_set_option(OPTION, VALUE): kernel-selection experiments")
    ap.add_argument("--gate-ms", type=float, default=0.0, help="diagnostic: queue the timed region behind a spin of this many ms (profiler timelines)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # test hooks for the multi-rank branch on a ONE-GPU box (tests/test_gpu_bench_path.py): N ranks of this script share GPU 0
    # and talk over gloo, so that init_process_group, the three broadcasts, all_reduce(MAX), the sharding self-check and the
    # gathers run before an 8-GPU node runs them.  Refused when the node has a GPU per rank: the measured path is RCCL.
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help=argparse.SUPPRESS)
    ap.add_argument("--ranks-share-gpu", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-roofline-events", action="store_true")
    ap.add_argument("--layers", action="store_true", help="print the per-conv-layer time table (stderr)")
    args = ap.parse_args(argv)
    if args.scene_only:
        args.config = 1
    W, H, depth, kind, ntri, impl = CONFIGS[args.config]
    args.width = args.width or W
    args.height = args.height or H
    args.depth = args.depth or depth
    args.mesh = ntri if args.mesh is None else args.mesh
    args.mesh_kind = args.mesh_kind or kind or "atrium"
    args.impl = args.impl or impl
    if args.batch is None:
        args.batch = 24                                     # = AIPT_TRACE_BATCH_MAX: a call's traces are ONE set of launches (32: 16 + 16; 851 vs 840 frames/s)
    return args


def respawn_if_needed(args):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: launch N ranks of this script, one per GPU."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and not args.ranks_share_gpu:
        sys.exit(f"bench.py: --gpus {args.gpus} requested but this node exposes {have} GPU(s); refusing to run fewer ranks "
                 f"and report them as {args.gpus}")
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def build_scene(args, api, synth):
    """rank 0: parse the scene file, add the procedural mesh of the chosen config; returns the packed blob and the camera"""
    sc = api.Scene(args.scene, res=(args.width, args.height), depth=args.depth)
    mats = list(sc.materials)
    faces, box, desc = (), None, f"Cornell box ({sc.ngeoms} primitives, no mesh)"
    if args.mesh:
        first = len(mats)
        if args.mesh_kind == "living":
            fnp, lb, ub, recs = synth.make_living_room_mesh(args.mesh, 565, first_material=first)
            mats += [api.Material.from_buffer_copy(r) for r in recs]
            desc = f"Cornell walls + procedural living-room mesh ({args.mesh} triangles: diffuse, reflective and refractive faces, BVH)"
        else:
            mats += [api.Material.from_buffer_copy(synth.STONE), api.Material.from_buffer_copy(synth.MIRROR)]
            refl = first + 1 if args.mesh_kind == "reflective" else first
            fnp, lb, ub = synth.make_atrium_mesh(args.mesh, 565, material=first, floor_material=refl, column_material=refl)
            desc = (f"Cornell walls + procedural Sponza-like atrium mesh ({args.mesh} triangles, "
                    f"{'reflective floor and columns' if args.mesh_kind == 'reflective' else 'all diffuse'}, BVH)")
        faces = fnp
        box = api.AABB()
        box.lb[:] = [float(v) for v in lb]
        box.ub[:] = [float(v) for v in ub]
    import numpy as np
    blob = api.scene_pack(sc.geoms, mats, faces, box)
    cam_bytes = bytes(sc.camera) + np.array([sc.zoom, sc.phi, sc.theta], np.float32).tobytes()
    return blob, cam_bytes, desc


class Workload:
    """Everything bench.py sets up before its timed region, and the calls the timed region makes.  tests/test_gpu_bench_path.py
    builds the same object, so the parity tests check the path -- and the sizes -- that are timed here."""

    def __init__(self, args, rank=0, world=1, local_rank=0, like=None, batch=None, coll_dev=None):
        """like: another Workload whose scene / weight blobs and cameras are reused (no second BVH build, no broadcast);
        batch: override args.batch (1 = a frame-by-frame context: aipt_frame only); coll_dev: device of the broadcast buffers
        (default: this rank's GPU -- RCCL; the CPU under the gloo test hook)"""
        import numpy as np
        import torch
        from ai_path_tracer_denoiser_amd import api, synth
        from ai_path_tracer_denoiser_amd import dist as adist
        self.args, self.rank, self.world = args, rank, world
        self.api, self.adist = api, adist
        self.dev = torch.device("cuda", local_rank)
        self.W, self.H, self.depth = args.width, args.height, args.depth
        self.stream = torch.cuda.Stream(device=self.dev)
        self.ctx = api.Context(local_rank, self.stream.cuda_stream)
        self.trace_flags = api.TRACE_DEFAULT if args.trace_flags is None else args.trace_flags
        # ---- rank 0 parses the scene, builds the BVH and makes the weights; one broadcast each (RCCL over xGMI)
        if like is not None:
            self.scene_blob, self.weight_blob, self.desc = like.scene_blob, like.weight_blob, like.desc
            self.cam0, self.zoom, self.phi0, self.theta = like.cam0, like.zoom, like.phi0, like.theta
        else:
            scene_blob = weight_blob = cam_bytes = desc_b = None
            if rank == 0:
                scene_blob, cam_bytes, desc = build_scene(args, api, synth)
                weight_blob = synth.make_blob(565)
                desc_b = desc.encode()
            cdev = self.dev if coll_dev is None else coll_dev
            self.scene_blob = adist.broadcast_bytes(scene_blob, 0, cdev)
            self.weight_blob = adist.broadcast_bytes(weight_blob, 0, cdev)
            cam_bytes = adist.broadcast_bytes(cam_bytes, 0, cdev)
            self.desc = adist.broadcast_bytes(desc_b, 0, cdev).decode()
            self.cam0 = api.Camera.from_buffer_copy(cam_bytes[:84])
            self.zoom, self.phi0, self.theta = [float(v) for v in np.frombuffer(cam_bytes[84:], np.float32)]
        ctx = self.ctx
        ctx.pathtrace_init_packed(self.scene_blob)   # copies only: the BVH inside the blob was built once, on rank 0
        ctx.load_weights(self.weight_blob)
        ctx.frame_configure(self.W, self.H)
        self.impl = {"f16x3": api.DN_IMPL_MFMA_F16X3, "f16w": api.DN_IMPL_MFMA_F16W, "f32": api.DN_IMPL_MFMA}[args.impl]
        ctx.denoise_set_impl(self.impl)
        for kv in (args.dn_opt or []):                         # tuning experiments: --dn-opt OPTION=VALUE (aipt_denoise_set_option)
            k, v = kv.split("=")
            ctx.denoise_set_option(int(k), int(v))
        self.B = max(1, args.batch if batch is None else batch)
        if self.B > 1:
            ctx.frames_configure(self.B)
        self.outs = [torch.empty(3, self.H, self.W, device=self.dev) for _ in range(self.B)]
        self.bn_batch = args.bn == "batch"
        self.carry = args.hidden == "carry"
        self.cams = []

    def camera_for(self, g):
        """camera of GLOBAL frame g of the orbit pan"""
        cam = self.api.Camera.from_buffer_copy(bytes(self.cam0))
        self.api.lib().aipt_camera_orbit(cam, self.zoom, self.adist.pan_phi(self.phi0, g), self.theta)
        return cam

    def set_frames(self, frames):
        """the global frame indices this rank renders, in order (a contiguous chunk of the sequence)"""
        self.frames = list(frames)
        self.cams = [self.camera_for(g) for g in self.frames]

    def trace_call_sizes(self, k0, k1):
        """frames held by every trace call run_frames(k0, k1) makes (aipt_frames splits a batch evenly over ceil(n/24) calls)"""
        sizes, k = [], k0
        while k < k1:
            nb = min(self.B, k1 - k)
            nc = (nb + 23) // 24
            sizes += [nb // nc + (1 if c < nb % nc else 0) for c in range(nc)]
            k += nb
        return sizes

    def run_frames(self, k0, k1, on_batch=None):
        """frames k0 .. k1-1 of this rank, in order; with --batch B the traces of B consecutive frames share their launches.
        on_batch(k, nb) is called after every call has been queued (tests read the outputs there)."""
        ctx, B, cams, outs, args = self.ctx, self.B, self.cams, self.outs, self.args
        k = k0
        while k < k1:
            nb = min(B, k1 - k)
            if nb > 1:
                ctx.frames(cams[k:k + nb], 1, self.depth, outs[:nb], trace_flags=self.trace_flags, bn_batch=self.bn_batch,
                           carry_first=self.carry and k > 0, carry=self.carry)
            else:
                ctx.frame(cams[k], 1, self.depth, outs[0], trace_flags=self.trace_flags, bn_batch=self.bn_batch,
                          carry=self.carry and k > 0)
                # never across the warmup/timed boundary or past the last frame: the timed region holds exactly K traces
                if args.prefetch and k + 1 < k1:
                    ctx.frame_prefetch(cams[k + 1], 1, self.depth, self.trace_flags)
            if on_batch:
                on_batch(k, nb)
            k += nb

    def run_frame_by_frame(self, cams, out, on_frame=None, prefetch=False):
        """the same sequence as unbatched, unpipelined aipt_frame calls from a zero hidden state (the reference's runCuda loop)"""
        ctx = self.ctx
        ctx.reset_hidden()
        for k, c in enumerate(cams):
            ctx.frame(c, 1, self.depth, out, trace_flags=self.trace_flags, bn_batch=self.bn_batch, carry=self.carry and k > 0)
            if prefetch and k + 1 < len(cams):
                ctx.frame_prefetch(cams[k + 1], 1, self.depth, self.trace_flags)
            if on_frame:
                on_frame(k)


def main():
    args = parse_args()
    respawn_if_needed(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    from ai_path_tracer_denoiser_amd import api, arch, synth
    from ai_path_tracer_denoiser_amd import dist as adist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    shared = False
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.ranks_share_gpu or args.backend != "nccl":
            # the test hook: every rank on GPU 0, collectives over gloo on host tensors (RCCL needs a GPU per rank)
            if not (args.ranks_share_gpu and args.backend == "gloo"):
                sys.exit("bench.py: --ranks-share-gpu and --backend gloo go together (a test hook for one-GPU boxes)")
            if torch.cuda.device_count() >= world:
                sys.exit(f"bench.py: this node has {torch.cuda.device_count()} GPUs for {world} ranks; --ranks-share-gpu is refused "
                         "where every rank can have its own GPU (the measured path is one rank per GPU over RCCL)")
            shared, local_rank = True, 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            if torch.cuda.device_count() <= local_rank:
                sys.exit(f"bench.py: rank {rank} has no GPU {local_rank} ({torch.cuda.device_count()} visible)")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    coll_dev = torch.device("cpu") if shared else torch.device("cuda", local_rank)    # where collective buffers live

    wl = Workload(args, rank, world, local_rank, coll_dev=coll_dev)
    ctx, dev, W, H, depth, B = wl.ctx, wl.dev, wl.W, wl.H, wl.depth, wl.B
    outs, bn_batch, carry, trace_flags = wl.outs, wl.bn_batch, wl.carry, wl.trace_flags
    scene_blob, weight_blob, desc = wl.scene_blob, wl.weight_blob, wl.desc

    per_rank = args.warmup + args.steps
    wl.set_frames(adist.frame_shard(rank, world, per_rank))
    cams = wl.cams
    run_frames = wl.run_frames

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- warmup; on the first warmup frames time every conv layer and every bounce launch to find the dominant kernel
    events = args.warmup > 0 and not args.no_roofline_events
    if events:
        nprof = min(3, args.warmup)
        ctx.profile_stride(1)
        ctx.profile_begin((1 << 28) - 1, nprof)
        ctx.trace_profile_begin(nprof, 1)
    run_frames(0, args.warmup)
    layer_tbl = prof_layers = conv_dominant = None
    if events:
        ms28, ncalls = ctx.profile_end()
        tr_ms, tr_calls = ctx.trace_profile_end(depth)
        layer_tbl = [ctx.layer_info(l) for l in range(28)]
        by_kernel = {}
        for l, info in enumerate(layer_tbl):
            by_kernel.setdefault(info["kernel"], []).append(l)
        conv_dominant = max(by_kernel, key=lambda kn: sum(ms28[l] for l in by_kernel[kn]))
        if args.layers and rank == 0:
            names = [t[0] for t in arch.layer_table()]
            for l, info in enumerate(layer_tbl):
                ms = ms28[l] / max(1, ncalls)
                print(f"{names[l]:9s} {info['cin']:3d}->{info['cout']:3d} {info['h']:4d}x{info['w']:<4d} {info['kernel']:32s} "
                      f"{ms * 1e3:8.1f} us {info['flops'] / ms / 1e9 if ms > 0 else 0:7.1f} TF", file=sys.stderr)
            print(f"conv total {sum(ms28) / max(1, ncalls):.3f} ms; bounce launches (us) "
                  f"{[round(1e3 * v / max(1, tr_calls), 1) for v in tr_ms]}", file=sys.stderr)
        prof_layers = by_kernel[conv_dominant]
    barrier()

    # ---- timed region: exactly K frames, NO events in it and both denoiser streams live.  (Rounds 1-4 recorded the roofline's
    # HIP-event pairs inside the timed region; a forward pass whose launches are timed runs ALONE -- the other denoiser stream
    # drains before it -- which cost `value` about 2 %.)  The pairs are now taken on a second, untimed pass over the same K
    # frames right after the timed region: same calls, same frames per trace launch set.
    timed_trace_calls = wl.trace_call_sizes(args.warmup, per_rank)
    barrier()
    if args.gate_ms > 0:
        # diagnostic (rocprofv3 timelines): the timed call queues up behind a spin on the context's stream, so that the GPU starts it
        # with every launch already enqueued -- under the profiler the host needs ~0.85 ms per frame to enqueue (0.12 ms without)
        # and the streams would otherwise run host-bound.  `value` of such a run includes the spin: not a benchmark figure.
        with torch.cuda.stream(wl.stream):
            torch.cuda._sleep(int(args.gate_ms * 1e-3 * 2.1e9))
    t0 = time.perf_counter()
    run_frames(args.warmup, per_rank)
    t_enq = time.perf_counter()                               # every launch of the timed region is queued (nothing has been waited for)
    torch.cuda.synchronize(dev)
    barrier()
    t1 = time.perf_counter()
    elapsed_local = elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    n_live = ctx.live_counts(depth)            # of the last trace call: totals over all its frames (both trace lanes)
    trace_kernels = [ctx.trace_kernel_name(0), ctx.trace_kernel_name(1)] if depth > 1 else [ctx.trace_kernel_name(0)] * 2
    P = W * H
    Hp, Wp = (H + 31) // 32 * 32, (W + 31) // 32 * 32
    # what the timed region left behind: the last denoised frame and its G-buffer, and the first frame of this rank's chunk
    ctx.sync()
    last_nb = (per_rank - args.warmup - 1) % B + 1 if B > 1 else 1     # frames of the last call of the timed region
    timed_last = outs[last_nb - 1 if B > 1 else 0].clone()
    gptr, grows, gstride = ctx.gbuffer()
    timed_gbuf = np.empty((10, grows, gstride), np.float32)
    assert api.lib().aipt_download(ctx._h, timed_gbuf.ctypes.data, gptr, timed_gbuf.nbytes) == 0
    # ---- roofline pass (untimed): the K frames of the timed region again, with HIP-event pairs on the launch stream around the
    # launches of the dominant conv kernel in EVERY frame (the library runs a forward pass whose launches are timed ALONE, so a
    # pair brackets the kernel and not its overlap with the next frame's launches) and around every bounce launch.
    conv_prof = trace_prof = None
    if prof_layers:
        nrec = min(args.steps, 4096)
        ctx.profile_stride(1)
        ctx.profile_begin(sum(1 << l for l in prof_layers), nrec)
        ctx.trace_profile_begin(nrec if B == 1 else len(timed_trace_calls), 1)
        run_frames(args.warmup, per_rank)
        ctx.sync()
        torch.cuda.synchronize(dev)
        conv_prof = ctx.profile_end()
        fr_calls, ms_calls = ctx.trace_profile_calls(depth)
        ctx.trace_profile_end(depth)
        trace_prof = (fr_calls, ms_calls)

    # ---- validation, outside the timed region: the whole sequence of this rank again, frame by frame (aipt_frame: one trace and
    # one denoiser pass at a time, no batching, no second stream) from a zero hidden state.  The last frame depends on every
    # frame before it through the carried hidden state, so equal bits there vouch for the whole batched / pipelined run.  The
    # same pass gives the frame-by-frame (interactive, no latency) throughput.
    # (a second context, configured for single frames only, on the same stream: what an interactive host would create)
    ctx.sync()
    torch.cuda.synchronize(dev)
    fb = Workload(args, rank, world, local_rank, like=wl, batch=1) if B > 1 else wl
    fb.run_frame_by_frame(cams[:2], fb.outs[0])
    fb.ctx.sync()
    v0 = time.perf_counter()
    fb.run_frame_by_frame(cams, fb.outs[0])
    fb.ctx.sync()
    fbf_fps = per_rank / (time.perf_counter() - v0)
    fbf_last = fb.outs[0].clone()
    fbf_gbuf = np.empty_like(timed_gbuf)
    gptr2, _, _ = fb.ctx.gbuffer()
    assert api.lib().aipt_download(fb.ctx._h, fbf_gbuf.ctypes.data, gptr2, fbf_gbuf.nbytes) == 0
    gbuf_equal = bool(np.array_equal(timed_gbuf.view(np.uint32), fbf_gbuf.view(np.uint32)))
    out_equal = bool(torch.equal(timed_last.view(torch.int32), fbf_last.view(torch.int32)))
    validated = gbuf_equal and out_equal
    # frame by frame with the next frame's trace prefetched on disjoint CUs (one frame of latency); the first pass creates the
    # two CU-masked streams, the second one is timed
    fb.run_frame_by_frame(cams[:3], fb.outs[0], prefetch=True)
    fb.ctx.sync()
    torch.cuda.synchronize(dev)
    v0 = time.perf_counter()
    fb.run_frame_by_frame(cams, fb.outs[0], prefetch=True)
    fb.ctx.sync()
    fbf_pf_fps = per_rank / (time.perf_counter() - v0)
    validated = validated and bool(torch.equal(fb.outs[0].view(torch.int32), fbf_last.view(torch.int32)))

    # ---- multi-GPU self-check: every rank's first and last denoised frame, as 8-byte checksums through ONE
    # all-gather, against rank 0's own frame-by-frame render of those frames (N = 1: rank 0's chunk is the whole sequence)
    local_sums = [adist.checksum64(timed_last)]
    # (the first frame of the chunk was overwritten by later batches: its checksum is taken from a re-run of the first
    # batch-sized call, which is what produced it)
    ctx.reset_hidden()
    first_nb = min(B, args.warmup) if args.warmup > 0 else min(B, args.steps)
    first_sum = {}

    def grab_first(k, nb):
        if k == 0:
            ctx.sync()
            first_sum[0] = adist.checksum64(outs[0])
    run_frames(0, first_nb, on_batch=grab_first)
    local_sums.insert(0, first_sum[0])

    def rerender(r):
        """rank 0 renders rank r's chunk by itself, frame by frame: checksums of its first and last frame"""
        cs = [wl.camera_for(g) for g in adist.frame_shard(r, world, per_rank)]
        got = {}

        def on_frame(k):
            if k == 0 or k == len(cs) - 1:
                fb.ctx.sync()
                got[k] = adist.checksum64(fb.outs[0])
        fb.run_frame_by_frame(cs, fb.outs[0], on_frame=on_frame)
        return [got[0], got[len(cs) - 1]]
    sharded_ok, sharded_per_rank = adist.sharded_equals_single(local_sums, rerender, coll_dev, rank)
    per_rank_fps = adist.gather_int64([int(round(1e3 * args.steps / elapsed_local))], coll_dev)

    # the trace / denoise split: one un-pipelined frame
    ctx.sync()
    ctx.frame_set_timing(True)
    ctx.frame(cams[per_rank - 1], 1, depth, outs[0], trace_flags=trace_flags, bn_batch=bn_batch, carry=carry)
    trace_ms, denoise_ms = ctx.frame_last_times()

    roof = other = None
    if prof_layers:
        # ---- conv kernel
        ms28, ncalls = conv_prof
        launches = ncalls * len(prof_layers)
        tot_ms = float(sum(ms28[l] for l in prof_layers))
        flops_per_frame = sum(layer_tbl[l]["flops"] for l in prof_layers)

        def layer_bytes(l):
            # algorithmic HBM bytes of one conv launch: every input channel read once at its STORED resolution (the
            # decoders' first convs read half-resolution sources through the fused upsample) + every output written once
            info = layer_tbl[l]
            src_px = info["h"] * info["w"] // 4 if (l >= 18 and (l - 18) % 2 == 0) else info["h"] * info["w"]
            return 4.0 * (info["cin"] * src_px + info["cout"] * info["h"] * info["w"])
        bytes_per_frame = sum(layer_bytes(l) for l in prof_layers)
        avg_ms = tot_ms / max(1, launches)
        tflops = flops_per_frame * ncalls / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
        gbps = bytes_per_frame * ncalls / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0
        split = conv_dominant.startswith("conv3x3_f16x3")
        # MFMA ceiling of the kernel's arithmetic: f32-input MFMA 157.3 TFLOP/s; split-fp16 = fp16 dense peak / 3 MFMAs
        mfma_per_product = 2.0 if args.impl == "f16w" else 3.0
        mfma_peak = MI355X_FP16_MFMA_TFLOPS / mfma_per_product if split else MI355X_FP32_MFMA_TFLOPS
        ai = flops_per_frame / bytes_per_frame
        hbm_bound = ai < mfma_peak * 1e12 / MI355X_HBM_BPS
        mfma = {"achieved_tflops_algorithmic": round(tflops, 2), "peak_tflops": round(mfma_peak, 1),
                "frac": round(tflops / mfma_peak, 4),
                "note": (f"fp16 dense MFMA peak 2500 / {int(mfma_per_product)} MFMAs per product" if split else "f32-input MFMA peak")}
        hbm = {"achieved_GBps_algorithmic": round(gbps, 1), "peak_GBps": MI355X_HBM_BPS / 1e9,
               "frac": round(gbps * 1e9 / MI355X_HBM_BPS, 4)}
        # HBM traffic per launch: NOT measured in this run -- rocprofv3 PMC passes of the same command, committed under profiles/
        # (tools/collect_evidence.sh); quoted only when that run's launch mix of the kernel equals this run's
        pm_conv = pmc_entry(conv_dominant)
        conv_traffic = conv_traffic_src = None
        if pm_conv:
            lpf = pm_conv.get("launches_per_frame")
            if lpf is not None and abs(lpf - len(prof_layers)) < 0.01:
                conv_traffic = pm_conv.get("hbm_bytes_per_launch")
                conv_traffic_src = (f"profiles/pmc_dominant.json @{pmc_file_sha()}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                    f"this command (another run; {lpf:g} launches of the kernel per frame there as here), not measured in this run")
            else:
                conv_traffic_src = (f"null: profiles/pmc_dominant.json @{pmc_file_sha()} averaged {lpf} launches of this kernel per frame, "
                                    f"this run has {len(prof_layers)}")
        conv_roof = {"bound": "hbm" if hbm_bound else "mfma",
                     "achieved": round(gbps, 1) if hbm_bound else round(tflops, 3),
                     "peak": MI355X_HBM_BPS / 1e9 if hbm_bound else round(mfma_peak, 1),
                     "unit": "GB/s" if hbm_bound else "TFLOP/s",
                     "frac": hbm["frac"] if hbm_bound else mfma["frac"],
                     "traffic": conv_traffic, "traffic_source": conv_traffic_src,
                     "kernel": conv_dominant, "launches_per_frame": len(prof_layers), "avg_launch_ms": round(avg_ms, 5),
                     "ms_per_frame": round(avg_ms * len(prof_layers), 4),
                     "launches_timed": launches,
                     "timing_note": ("HIP-event pairs on the launch stream, taken on a second UNTIMED pass over the same K frames right "
                                     "after the timed region (the timed region itself records no events).  The kernel ALONE: a "
                                     "forward pass whose launches are timed runs with the other denoiser stream drained; in the "
                                     "timed region two forward passes overlap, and rocprofv3's per-launch durations of the same "
                                     "command include that overlap (profiles/: kernel_stats vs kernel_stats_one_denoiser_stream)"
                                     if B > 1 else "frame by frame: one stream; event pairs on an untimed second pass over the same frames"),
                     "algorithmic_bytes_per_launch": bytes_per_frame / len(prof_layers),
                     "flops_per_launch": flops_per_frame / len(prof_layers),
                     "arithmetic_intensity_flop_per_byte": round(ai, 1), "mfma": mfma, "hbm": hbm,
                     "layers": [arch.layer_table()[l][0] for l in prof_layers]}
        # ---- bounce kernel (bounces 1 .. depth-1 share one instantiation; bounce 0 is its own).  Only the recorded calls that
        # held the most frequent frame count go into the figure (a timed region of 20 frames is traced 10 + 10, one of 70 frames
        # 16 + 16 | 14 + 14 | 6): time per launch, frames per launch and bytes per launch then describe the SAME launches.
        fr_calls, ms_calls = trace_prof
        last_fr = max(1, int(n_live[0]) // P)                                # frames of the last trace call
        nb = [float(v) / last_fr for v in n_live[:depth]]                    # live paths per bounce, per frame
        late = [b for b in range(1, depth) if nb[b] > 0]
        tr_roof = None
        if len(fr_calls) and late:
            vals, counts = np.unique(fr_calls, return_counts=True)
            fpc = int(vals[np.argmax(counts * vals)])                        # the frame count that carries most frames
            sel = fr_calls == fpc
            nsel = int(sel.sum())
            t_late = float(ms_calls[sel][:, late].sum()) / nsel             # ms per such call in the later-bounce launches
            by_late = sum(nb[b] * 160.0 for b in late) * fpc                # N_b x 160 B per bounce
            t_first = float(ms_calls[sel][:, 0].sum()) / nsel
            by_first = (nb[0] * 160.0 + nb[0] * 64.0) * fpc                 # + G-buffer write and image RMW, once per frame
            name = trace_kernels[1]
            g_late = by_late / (t_late * 1e-3) / 1e9 if t_late > 0 else 0.0
            pm = pmc_entry(name)
            traffic = pm.get("hbm_bytes_per_launch") if pm and pm.get("frames_per_launch") == fpc else None
            tr_roof = {"bound": "hbm", "achieved": round(g_late, 1), "peak": MI355X_HBM_BPS / 1e9, "unit": "GB/s",
                       "frac": round(g_late * 1e9 / MI355X_HBM_BPS, 5), "traffic": traffic,
                       "traffic_source": (f"profiles/pmc_dominant.json @{pmc_file_sha()} (rocprofv3 PMC passes of another run with {fpc} "
                                          "frames per launch), not measured in this run") if traffic is not None else None,
                       "traffic_note": (None if traffic is not None else
                                        "the committed PMC pass traced a different number of frames per launch than this run"),
                       "kernel": name, "launches_per_frame": round(len(late) / fpc, 3), "avg_launch_ms": round(t_late / len(late), 5),
                       "ms_per_frame": round(t_late / fpc, 4), "frames_per_launch": fpc, "launches_timed": nsel * len(late),
                       "trace_calls_of_the_timed_region": [int(v) for v in fr_calls],
                       "timing_note": "HIP-event pairs around every bounce launch of a second, untimed pass over the timed region's frames; "
                                      "while they record, a call's frames are traced by ONE set of launches (the kernel alone, as the PMC "
                                      "passes see it); in the timed region aipt_frames traces the two halves of a call side by side on two "
                                      "streams (half the frames per launch each: their launch floors hide behind each other)",
                       "algorithmic_bytes_per_launch": by_late / len(late),
                       "byte_model": "SURVEY 8d: sum over bounces of N_b x 160 B (44 B state read + 44 B write, 36 B hit record "
                                     "write + read of the reference's layout); BVH and triangle fetches are overhead, not algorithmic",
                       "first_bounce": {"kernel": trace_kernels[0], "avg_launch_ms": round(t_first, 5),
                                        "algorithmic_bytes_per_launch": by_first,
                                        "achieved_GBps": round(by_first / (t_first * 1e-3) / 1e9, 1) if t_first > 0 else 0.0},
                       "rays_per_frame": int(sum(nb)), "grays_per_s": round(sum(nb) * fpc / ((t_late + t_first) * 1e-3) / 1e9, 3),
                       "note": "the batched BVH walk is VALU-issue-bound (DESIGN.md 5); the HBM roof is reported because north_star "
                               "asks for it, it is not what bounds this kernel"}
        if tr_roof and tr_roof["ms_per_frame"] > conv_roof["ms_per_frame"]:
            roof, other = tr_roof, conv_roof
        else:
            roof, other = conv_roof, tr_roof

    fps = world * args.steps / elapsed
    last_frames = max(1, int(n_live[0]) // P)
    trace_bytes = float(sum(int(n) for n in n_live[:depth]) * 160 + int(n_live[0]) * 64) / last_frames   # SURVEY 8d byte model, per frame
    dn_bytes = float(arch.activation_bytes(Hp, Wp))
    ms_per_step = elapsed / args.steps * 1e3

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, scene_blob, weight_blob, W, H, Hp, Wp, depth, bn_batch)

    if rank == 0:
        metric_name = "denoised frames/sec @1280×720 1spp depth8; ms/frame trace vs denoise split"
        try:                                                   # BASELINE.json's own wording when the file travels with the repo
            metric_name = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except Exception:
            pass
        line = {
            "metric": metric_name, "value": round(fps, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f16x3": "f32 (split-fp16 MFMA operands, fp32 accumulate)", "f32": "f32",
                      "f16w": "fp16 conv weights, f32 activations (split-fp16 MFMA operands), fp32 accumulate"}[args.impl],
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{args.config}]: {desc} {W}x{H}, 1spp, depth {depth}, orbit pan, "
                                   f"BN {args.bn}-stats, hidden {args.hidden}, conv {args.impl}",
                       "frames_per_gpu": args.steps, "denoiser_input": f"10x{Hp}x{Wp}",
                       "weights": "synthetic Kaiming-variance uniform, seed 565", "parallelism": f"frame-shard x{world}",
                       "pipelining": ((f"throughput mode: a call holds up to {B} consecutive frames; their traces share launches (the two halves "
                                       f"of a call side by side on two streams, at most 24 frames per pair of launch sets) and their denoiser "
                                       f"passes run on two streams, frame n+1 one encoder level behind frame n (aipt_frames)" if B > 1 else
                                      "frame by frame" + ("; the next frame's trace runs beside this frame's denoise on disjoint CUs" if args.prefetch else ""))
                                      + "; frames bit-identical to un-pipelined rendering")},
            # the mode `value` is quoted in: a batch of frames is in flight together, so the first frame of a call is delivered
            # after the whole call; the interactive figures (no batching) are under "frame_by_frame"
            "latency_frames": min(B, args.steps) if B > 1 else (1 if args.prefetch else 0),
            # ... the same in time: the first frame of a call leaves after the whole call (throughput mode), after one frame time
            # otherwise; an interactive host (the reference is a one-frame loop, main.cpp:143-163) reads "frame_by_frame"
            "latency_ms": round((min(B, args.steps) if B > 1 else (2 if args.prefetch else 1)) * ms_per_step, 3),
            "host_enqueue_ms_per_step": round((t_enq - t0) / args.steps * 1e3, 4),
            "frame_by_frame": {"value": round(world * fbf_fps, 3), "unit": "frames/s", "latency_frames": 0,
                               "latency_ms": round(1e3 / fbf_fps, 3) if fbf_fps else None,
                               "prefetch": {"value": round(world * fbf_pf_fps, 3), "latency_frames": 1,
                                            "latency_ms": round(2e3 / fbf_pf_fps, 3) if fbf_pf_fps else None,
                                            "note": "aipt_frame_prefetch: frame k+1 traced beside the denoise of frame k on disjoint CUs"},
                               "note": f"aipt_frame, one call per frame, the {per_rank} frames of this rank's chunk after the timed "
                                       "region (host-synchronised at the end only)"},
            "validated": validated,
            "validation": {"what": "after the timed region the rank's whole sequence is rendered again frame by frame (aipt_frame) from a "
                                   "zero hidden state; the last timed frame's G-buffer and denoised output must equal it bit for bit "
                                   "(the hidden state carries every earlier frame into it); the prefetching frame-by-frame run likewise",
                           "gbuffer_equal": gbuf_equal, "denoised_equal": out_equal},
            "rccl_ranks": 0 if shared else world, "sharded_equals_single": sharded_ok,
            **({"ranks_share_gpu": True, "collectives": "gloo (test hook: every rank on GPU 0; not a scaling measurement)"} if shared else {}),
            "sharding_check": {"per_rank_equal": sharded_per_rank,
                               "per_rank_frames_per_s": [round(v[0] / 1e3, 3) for v in per_rank_fps],
                               "what": "checksums (8 bytes each, one all-gather) of every rank's first and last denoised frame against "
                                       "rank 0's own frame-by-frame render of those frames"},
            "roofline": roof,
            "roofline_other": other,
            # the whole frame against the HBM roof north_star names: algorithmic bytes of one frame (trace byte model + every
            # activation read and written once) over the measured time per frame of the timed region
            "roofline_frame": {"bound": "hbm", "achieved": round((trace_bytes + dn_bytes) / (ms_per_step * 1e-3) / 1e9, 1),
                               "peak": MI355X_HBM_BPS / 1e9, "unit": "GB/s",
                               "frac": round((trace_bytes + dn_bytes) / (ms_per_step * 1e-3) / MI355X_HBM_BPS, 5),
                               "algorithmic_bytes_per_frame": trace_bytes + dn_bytes, "ms_per_frame": round(ms_per_step, 4),
                               "note": "no kernel of the frame is HBM-bound (DESIGN.md 5): the conv launches' parts add up (MFMAs 38 %, epilogue 20 %, per-launch fixed 21 %, loads 12 %: profiles/r06_conv_ablate.txt), the bounce kernel is VALU-issue-bound"},
            "cpu_baseline": cpu,
            "frame": {"ms_trace": round(trace_ms, 4), "ms_denoise": round(denoise_ms, 4),
                      "split_note": "one un-pipelined frame (aipt_frame) after the timed region",
                      "algorithmic_bytes": trace_bytes + dn_bytes, "denoise_gflop": arch.conv_flops(Hp, Wp) / 1e9,
                      "hbm_frac_of_8TBps": round((trace_bytes + dn_bytes) / (ms_per_step * 1e-3) / MI355X_HBM_BPS, 5),
                      "n_live": [int(v) // last_frames for v in n_live], "n_live_note": "per frame (mean over the last batch)" if last_frames > 1 else "last frame"},
        }
        print(json.dumps(line))
    if fb is not wl:
        fb.ctx.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def pmc_entry(kernel):
    """the committed rocprofv3 PMC summary of `kernel` (profiles/pmc_dominant.json), or None"""
    path = os.path.join(ROOT, "profiles", "pmc_dominant.json")
    try:
        return json.load(open(path)).get("kernels", {}).get(kernel)
    except Exception:
        return None


def pmc_file_sha():
    """first 12 hex digits of the sha256 of profiles/pmc_dominant.json (what a quoted traffic figure comes from)"""
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, "profiles", "pmc_dominant.json"), "rb").read()).hexdigest()[:12]
    except Exception:
        return "missing"


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary (profiles/pmc_dominant.json), or None"""
    path = os.path.join(ROOT, "profiles", "pmc_dominant.json")
    try:
        pj = json.load(open(path))
        ent = pj.get("kernels", {}).get(kernel)
        return ent.get("hbm_bytes_per_launch") if ent else None
    except Exception:
        return None


def cpu_baseline(args, scene_blob, weight_blob, W, H, Hp, Wp, depth, bn_batch):
    """The oracle (a port, test infrastructure) on the host cores: 2 warm-up frames, then the median of 5 frames of the same
    workload.  Trace: C/OpenMP restatement through its CPU BVH (same result as the reference's loop over all
    faces, which is also timed on a sub-sampled frame and extrapolated); denoise: the build's PyTorch-CPU restatement."""
    import numpy as np
    import torch
    import oracle
    from oracle.torch_denoise import TorchDenoiser
    from ai_path_tracer_denoiser_amd import dist as adist
    from ai_path_tracer_denoiser_amd import synth

    geoms, mats, faces, box = adist.unpack_scene(scene_blob)
    osc = oracle.OracleScene.parse(args.scene, res=(W, H), depth=depth)
    osc.materials = [oracle.Material.from_buffer_copy(bytes(m)) for m in mats]
    if faces:
        _, _, fnp, _ = adist.scene_bvh(scene_blob)
        osc.set_mesh(np.array(fnp), box.lb, box.ub)
    fl = oracle.TRACE_AA | oracle.TRACE_COMPACT
    t_tr, t_dn = [], []
    td = TorchDenoiser(weight_blob)
    gp = np.zeros((10, Hp, Wp), np.float32)
    for k in range(7):
        osc.set_orbit(osc.zoom, adist.pan_phi(osc.phi, k), osc.theta)
        c0 = time.perf_counter()
        g_ref, _, _ = osc.pathtrace(pad_rows_to=Hp, want_mat0=False, flags=fl | oracle.TRACE_ORACLE_BVH)
        c1 = time.perf_counter()
        gp[:, :, :W] = g_ref
        td.forward(gp, bn_batch, k > 0)
        c2 = time.perf_counter()
        if k >= 2:                                            # 2 warm-ups (oneDNN primitive creation, page faults)
            t_tr.append(c1 - c0)
            t_dn.append(c2 - c1)
    tr, dn = float(np.median(t_tr)), float(np.median(t_dn))
    brute = ""
    if faces:
        # the reference's own algorithm (every face for every ray) on a frame of 1/64 of the pixels, extrapolated x64
        small = oracle.OracleScene.parse(args.scene, res=(max(1, W // 8), max(1, H // 8)), depth=depth)
        small.materials = osc.materials
        small.set_mesh(osc.faces_np, box.lb, box.ub)
        small.set_orbit(small.zoom, adist.pan_phi(small.phi, 0), small.theta)
        c0 = time.perf_counter()
        small.pathtrace(want_mat0=False, flags=fl)
        b = time.perf_counter() - c0
        brute = (f"; the reference's exhaustive face loop on a {W // 8}x{H // 8} frame took {b:.2f} s, i.e. ~{b * 64:.0f} s "
                 f"per full frame extrapolated x64 by pixel count ({1.0 / (b * 64 + dn):.4f} frames/s)")
    orc = oracle.DenoiseOracle(weight_blob, Hp, Wp)
    c0 = time.perf_counter()
    orc.forward(gp, bn_batch, False)
    c_dn = time.perf_counter() - c0
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return {"value": round(1.0 / (tr + dn), 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"median of 5 frames after 2 warm-ups of the same workload ({W}x{H} depth {depth}, {len(faces)} triangles): "
                      f"oracle trace {tr:.3f} s (C/OpenMP restatement, CPU BVH) + PyTorch-CPU denoise {dn:.2f} s (torch "
                      f"{torch.__version__}, {torch.get_num_threads()} threads; one frame of the C/OpenMP restatement of the denoiser: "
                      f"{c_dn:.2f} s){brute}"}


if __name__ == "__main__":
    main()
