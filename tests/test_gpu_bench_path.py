"""-m gpu: parity of the path bench.py TIMES, at the size it times it.

bench.py's timed region calls aipt_frames on batches of up to 32 frames: traces of up to 16 frames per launch set (pixel-
interleaved frames, per-frame live counters, split BVH walks in trace_bounce<false,true>) and the denoiser passes of
consecutive frames pipelined over two streams with the hidden state carried.  These tests build bench.py's own Workload
object (same scene, mesh, weights, cameras, flags and configure calls) and check

  * G-buffers of chosen frames of one 32-frame call against the CPU oracle, bit for bit (BASELINE configs[2], 1280x720,
    262 144 triangles, depth 8; the reference semantics: pathtrace.cu:422-528 through oracle/trace_oracle.c);
  * the denoised first frame against the oracle at the 1e-3 bar of BASELINE.json's north_star;
  * all 32 denoised frames against the frame-by-frame aipt_frame sequence, bit for bit;
  * one batch-mode case each for configs[3] (reflective mesh) and configs[4] (1920x1080, depth 12, fp16 conv weights).
"""
import os

import numpy as np
import pytest

from ai_path_tracer_denoiser_amd import api, synth
from ai_path_tracer_denoiser_amd import dist as adist

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _workload(config, batch, nframes):
    import bench
    args = bench.parse_args(["--config", str(config), "--batch", str(batch), "--no-cpu-baseline"])
    wl = bench.Workload(args)
    wl.set_frames(range(nframes))
    return wl


def _oracle_scene(wl):
    """the CPU oracle's view of bench.py's scene: same primitives, materials, mesh (the packed blob's own arrays)"""
    import oracle
    args = wl.args
    geoms, mats, faces, box = adist.unpack_scene(wl.scene_blob)
    osc = oracle.OracleScene.parse(args.scene, res=(wl.W, wl.H), depth=wl.depth)
    assert [bytes(g) for g in osc.geoms] == [bytes(g) for g in geoms]          # two independent scene front ends agree
    osc.materials = [oracle.Material.from_buffer_copy(bytes(m)) for m in mats]
    if faces:
        _, _, fnp, _ = adist.scene_bvh(wl.scene_blob)
        osc.set_mesh(np.array(fnp), box.lb, box.ub)
    return osc


def _oracle_gbuffer(wl, osc, g, rows, stride):
    import oracle
    osc.set_orbit(osc.zoom, adist.pan_phi(osc.phi, g), osc.theta)
    assert bytes(osc.camera) == bytes(wl.cams[g]), "oracle and product cameras differ"
    g_ref, n_live, _ = osc.pathtrace(pad_rows_to=rows, want_mat0=False,
                                     flags=oracle.TRACE_AA | oracle.TRACE_COMPACT | oracle.TRACE_ORACLE_BVH)
    gp = np.zeros((10, rows, stride), np.float32)
    gp[:, :, :wl.W] = g_ref
    return gp, n_live


def _batch_gbuffer(wl, f):
    ptr, rows, stride = wl.ctx.frames_gbuffer(f)
    host = np.empty((10, rows, stride), np.float32)
    assert api.lib().aipt_download(wl.ctx._h, host.ctypes.data, ptr, host.nbytes) == 0
    return host


def _frame_by_frame(wl, n):
    res = []
    import torch
    out = torch.empty(3, wl.H, wl.W, device="cuda")

    def keep(k):
        wl.ctx.sync()
        res.append(out.cpu().numpy().copy())
    wl.run_frame_by_frame(wl.cams[:n], out, on_frame=keep)
    return res


def test_bench_default_32_frame_call_configs2_full_size():
    """What `python bench.py` times: configs[2], frames_configure(32), one 32-frame aipt_frames call, hidden carried."""
    import oracle
    N = 32
    wl = _workload(2, 32, N)
    assert (wl.W, wl.H, wl.depth) == (1280, 720, 8) and wl.B == 32
    assert wl.trace_call_sizes(0, N) == [16, 16]
    got = []

    def keep(k, nb):
        wl.ctx.sync()
        got.extend(wl.outs[j].cpu().numpy().copy() for j in range(nb))
    wl.run_frames(0, N, on_batch=keep)
    assert len(got) == N
    assert wl.ctx.trace_kernel_name(1) == "trace_bounce<false,true>"           # the batched mesh instantiation is what ran
    # ---- G-buffers of frames of both 16-frame launch sets against the oracle, every bit
    osc = _oracle_scene(wl)
    rows, stride = (wl.H + 31) // 32 * 32, (wl.W + 31) // 32 * 32
    gp0 = None
    for f in (0, 7, 15, 16, 31):
        gp, n_ref = _oracle_gbuffer(wl, osc, f, rows, stride)
        host = _batch_gbuffer(wl, f)
        assert np.array_equal(host.view(np.uint32), gp.view(np.uint32)), f"frame {f}: G-buffer differs from the oracle"
        if f >= 16:                                                            # live counts of the last launch set (frames 16..31)
            assert wl.ctx.live_counts_frame(f - 16, wl.depth)[:len(n_ref)].tolist() == n_ref.tolist(), f
        if f == 0:
            gp0 = gp
    # ---- denoised frame 0 (zero hidden state) against the oracle: north_star's 1e-3 max abs per channel
    orc = oracle.DenoiseOracle(wl.weight_blob, rows, stride)
    y_ref = orc.forward(gp0, True, False)[:, :wl.H, :wl.W]
    err = float(np.abs(got[0] - y_ref).max())
    assert err <= 1e-3, err
    # ---- all 32 denoised frames against the frame-by-frame sequence, every bit
    ref = _frame_by_frame(wl, N)
    for k in range(N):
        assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), f"denoised frame {k} differs from aipt_frame"
    wl.ctx.close()


def test_driver_command_20_frames_after_5_warmup_configs2():
    """`bench.py --steps 20 --warmup 5` (the driver's command): a 5-frame call, then a 20-frame call traced by ONE set of launches
    (round 4: up to 24 frames per set; rounds 2-3 traced it 10 + 10) with the hidden state carried across the calls; every
    denoised frame equals the frame-by-frame sequence and G-buffers across the 20-frame launch set equal the oracle."""
    N = 25
    wl = _workload(2, 24, N)                                                   # bench.py's default --batch
    assert wl.trace_call_sizes(5, 25) == [20] and wl.trace_call_sizes(0, 5) == [5]
    got = []

    def keep(k, nb):
        wl.ctx.sync()
        got.extend(wl.outs[j].cpu().numpy().copy() for j in range(nb))
    wl.run_frames(0, 5, on_batch=keep)
    wl.run_frames(5, 25, on_batch=keep)
    osc = _oracle_scene(wl)
    rows, stride = (wl.H + 31) // 32 * 32, (wl.W + 31) // 32 * 32
    for j in (0, 9, 10, 19):                                                   # frames of the 20-frame call
        gp, _ = _oracle_gbuffer(wl, osc, 5 + j, rows, stride)
        assert np.array_equal(_batch_gbuffer(wl, j).view(np.uint32), gp.view(np.uint32)), f"frame {5 + j}"
    ref = _frame_by_frame(wl, N)
    for k in range(N):
        assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), f"denoised frame {k}"
    wl.ctx.close()


@pytest.mark.parametrize("config,nframes", [(3, 16), (4, 16)])
def test_batch_mode_configs3_and_4_full_size(config, nframes):
    """configs[3] (reflective floor and columns, 1280x720 depth 8) and configs[4] (living room, 1920x1080, depth 12, fp16
    conv weights) through one 16-frame aipt_frames call: first / middle / last G-buffer against the oracle, the denoised
    frames against the frame-by-frame sequence."""
    wl = _workload(config, 16, nframes)
    got = []

    def keep(k, nb):
        wl.ctx.sync()
        got.extend(wl.outs[j].cpu().numpy().copy() for j in range(nb))
    wl.run_frames(0, nframes, on_batch=keep)
    assert wl.ctx.trace_kernel_name(1) == "trace_bounce<false,true>"
    osc = _oracle_scene(wl)
    rows, stride = (wl.H + 31) // 32 * 32, (wl.W + 31) // 32 * 32
    for f in (0, nframes // 2, nframes - 1):
        gp, n_ref = _oracle_gbuffer(wl, osc, f, rows, stride)
        assert np.array_equal(_batch_gbuffer(wl, f).view(np.uint32), gp.view(np.uint32)), f"configs[{config}] frame {f}"
        assert wl.ctx.live_counts_frame(f, wl.depth)[:len(n_ref)].tolist() == n_ref.tolist()
    ref = _frame_by_frame(wl, nframes)
    for k in range(nframes):
        assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), f"configs[{config}] denoised frame {k}"
    wl.ctx.close()


def _run_bench(extra, nproc=None, timeout=900):
    import json
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    bench = os.path.join(ROOT, "bench.py")
    if nproc:
        import socket
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), bench] + extra
    else:
        cmd = [sys.executable, bench] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                        # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_multi_rank_branch_on_one_gpu():
    """[r5] VERDICT r4 missing 4: bench.py's `world > 1` branch -- init_process_group, the broadcasts of the packed scene, the
    weights and the cameras from rank 0, the barrier + all_reduce(MAX) around the timed region, the sharding self-check (every
    rank's first and last frame against rank 0's own render of that chunk) and the gathers -- executed before an 8-GPU node
    executes it: two ranks of the driver's launch line share GPU 0 and talk over gloo (a test hook bench.py refuses on a node with a
    GPU per rank).  What differs from the measured path is the transport of three small broadcasts; frames, sharding and
    reporting are the same code."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("a GPU per rank is available: the RCCL path itself runs (tests/test_gpu_comm.py, bench.py --gpus 2)")
    line = _run_bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--config", "0", "--backend", "gloo", "--ranks-share-gpu",
                       "--no-cpu-baseline"], nproc=2)
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["warmup"] == 2 and line["scaling"] == "weak"
    assert line["ranks_share_gpu"] is True and line["rccl_ranks"] == 0
    assert line["validated"] is True
    assert line["sharded_equals_single"] is True and line["sharding_check"]["per_rank_equal"] == [True, True]
    assert len(line["sharding_check"]["per_rank_frames_per_s"]) == 2
    assert line["value"] > 0 and abs(line["value"] - 2 * 6 / (line["ms_per_step"] * 6e-3)) < 1e-2 * line["value"]
    assert line["cpu_baseline"] is None and line["roofline"] is not None


def test_bench_eight_rank_launch_line_on_one_gpu():
    """[r6] VERDICT r5 item 7: the driver's EIGHT-rank launch line (`torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 ...`)
    before an 8-GPU node runs it: eight processes, eight contexts on GPU 0, eight chunks of the frame sequence, rank 0 re-rendering
    every chunk's first and last frame -- `per_rank_equal == [True] * 8` -- and ONE JSON line whose value is the eight ranks' frames
    over the slowest rank's time."""
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("a GPU per rank is available: bench.py --gpus 8 runs the RCCL path itself")
    line = _run_bench(["--gpus", "8", "--steps", "4", "--warmup", "2", "--config", "0", "--backend", "gloo", "--ranks-share-gpu",
                       "--no-cpu-baseline"], nproc=8, timeout=900)
    assert line["n_gpus"] == 8 and line["steps"] == 4 and line["warmup"] == 2 and line["scaling"] == "weak"
    assert line["ranks_share_gpu"] is True and line["rccl_ranks"] == 0
    assert line["validated"] is True
    assert line["sharded_equals_single"] is True and line["sharding_check"]["per_rank_equal"] == [True] * 8
    assert len(line["sharding_check"]["per_rank_frames_per_s"]) == 8
    assert line["value"] > 0 and abs(line["value"] - 8 * 4 / (line["ms_per_step"] * 4e-3)) < 1e-2 * line["value"]
    assert line["config"]["parallelism"] == "frame-shard x8"


def test_bench_line_of_the_driver_command_small():
    """the one-rank line at a small config: every key the contract names, the timed region validated, the roofline pass after it"""
    line = _run_bench(["--steps", "6", "--warmup", "2", "--config", "0", "--no-cpu-baseline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["validated"] is True and line["vs_baseline"] is None
    r = line["roofline"]
    assert r["launches_timed"] >= 6 and "untimed" in r["timing_note"].lower() and 0 < r["frac"] < 1
