"""-m gpu: the per-frame pipeline aipt_frame (trace -> device G-buffer -> denoise -> crop) vs the oracle."""
import numpy as np
import pytest

from ai_path_tracer_denoiser_amd import api, synth
from tests.gpu_util import CORNELL, to_api_scene

pytestmark = pytest.mark.gpu


def test_frame_sequence_matches_oracle():
    import torch
    import oracle
    W, H, depth = 80, 48, 4                # pads to 96 x 64
    sc = oracle.OracleScene.parse(CORNELL, res=(W, H), depth=depth)
    blob = synth.make_blob(565)
    ctx = api.Context(0)
    geoms, mats, faces, box, cam = to_api_scene(sc)
    ctx.pathtrace_init(geoms, mats, faces, box)
    ctx.load_weights(blob)
    ctx.frame_configure(W, H)
    ptr, rows, stride = ctx.gbuffer()
    assert (rows, stride) == (64, 96)
    orc = oracle.DenoiseOracle(blob, rows, stride)
    out = torch.empty(3, H, W, device="cuda")
    for k in range(3):                      # 3-frame orbit pan, hidden state carried (BASELINE configs[1] semantics)
        sc.set_orbit(sc.zoom, sc.phi + 0.1 * k, sc.theta)
        cam = api.Camera.from_buffer_copy(bytes(sc.camera))
        ctx.frame(cam, 1, depth, out, bn_batch=True, carry=True)
        ctx.sync()
        host = np.empty((10, rows, stride), np.float32)
        assert api.lib().aipt_download(ctx._h, host.ctypes.data, ptr, host.nbytes) == 0
        g_ref, _, _ = sc.pathtrace(pad_rows_to=rows)
        gp = np.zeros((10, rows, stride), np.float32)
        gp[:, :, :W] = g_ref
        assert np.array_equal(host, gp), f"frame {k}: G-buffer differs"
        y_ref = orc.forward(gp, True, k > 0)[:, :H, :W]
        assert np.abs(out.cpu().numpy() - y_ref).max() <= 1e-3
    ctx.close()


@pytest.mark.gpu
def test_prefetch_pipeline_is_bit_identical():
    """aipt_frame_prefetch (trace k+1 on the side stream during denoise k) must not change any output bit."""
    import torch
    W, H, depth = 80, 48, 4                # pads to 96 x 64: the crop path is pipelined too
    sc = api.Scene(CORNELL, res=(W, H), depth=depth)
    blob = synth.make_blob(7)

    def cams():
        out = []
        for k in range(4):
            c = api.Camera.from_buffer_copy(bytes(sc.camera))
            api.lib().aipt_camera_orbit(c, sc.zoom, sc.phi + 0.1 * k, sc.theta)
            out.append(c)
        return out

    def run(pipelined):
        ctx = api.Context(0)
        ctx.pathtrace_init(sc.geoms, sc.materials, sc.faces, None)
        ctx.load_weights(blob)
        ctx.frame_configure(W, H)
        out = torch.empty(3, H, W, device="cuda")
        res = []
        cs = cams()
        for k, c in enumerate(cs):
            ctx.frame(c, 1, depth, out, bn_batch=True, carry=k > 0)
            if pipelined and k + 1 < len(cs):
                ctx.frame_prefetch(cs[k + 1], 1, depth)
            ctx.sync()
            res.append(out.cpu().numpy().copy())
        return res

    a, b = run(False), run(True)
    for k, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x, y), f"frame {k} differs under prefetch"


def test_frame_sharding_is_equivalent_to_one_rank_with_resets():
    """rank r renders a contiguous chunk of the frame sequence and starts it from a zero hidden state.  Two
    'ranks' (two contexts on this GPU, chunks of 3 frames, cameras from dist.frame_shard / pan_phi) must produce the frames
    one context produces when it resets the hidden state at the chunk boundary -- byte for byte."""
    import torch
    from ai_path_tracer_denoiser_amd import dist as adist
    W, H, depth, per = 96, 64, 4, 3
    sc = api.Scene(CORNELL, res=(W, H), depth=depth)
    blob = synth.make_blob(11)

    def cam(g):
        c = api.Camera.from_buffer_copy(bytes(sc.camera))
        api.lib().aipt_camera_orbit(c, sc.zoom, adist.pan_phi(sc.phi, g), sc.theta)
        return c

    def make_ctx():
        ctx = api.Context(0)
        ctx.pathtrace_init(sc.geoms, sc.materials, sc.faces, None)
        ctx.load_weights(blob)
        ctx.frame_configure(W, H)
        return ctx

    out = torch.empty(3, H, W, device="cuda")
    sharded = {}
    for rank in range(2):
        ctx = make_ctx()
        for k, g in enumerate(adist.frame_shard(rank, 2, per)):
            ctx.frame(cam(g), 1, depth, out, bn_batch=True, carry=k > 0)
            ctx.sync()
            sharded[g] = out.cpu().numpy().copy()
        ctx.close()
    ctx = make_ctx()
    for g in range(2 * per):
        ctx.frame(cam(g), 1, depth, out, bn_batch=True, carry=g % per != 0)
        ctx.sync()
        assert np.array_equal(out.cpu().numpy(), sharded[g]), f"frame {g}"
    ctx.close()
