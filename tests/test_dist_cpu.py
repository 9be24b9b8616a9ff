"""N>1 path on CPU: world_size-2 gloo run of the one-time scene/weight broadcast and the frame sharding used by
bench.py --gpus N (no data-path collective)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ai_path_tracer_denoiser_amd import api, synth
from ai_path_tracer_denoiser_amd import dist as adist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORNELL = os.path.join(ROOT, "scenes", "cornell.txt")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    scene_blob = weights = None
    if rank == 0:
        sc = api.Scene(CORNELL, res=(64, 48))
        scene_blob = adist.pack_scene(sc.geoms, sc.materials, sc.faces, None)
        weights = synth.make_blob(565)
    scene_blob = adist.broadcast_bytes(scene_blob, 0, dev)
    weights = adist.broadcast_bytes(weights, 0, dev)
    geoms, mats, faces, box = adist.unpack_scene(scene_blob)
    frames = list(adist.frame_shard(rank, world, 5))
    import hashlib
    q.put((rank, hashlib.sha256(scene_blob).hexdigest(), hashlib.sha256(weights).hexdigest(), len(geoms), len(mats),
           frames))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_shard_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, w0, g0, m0, f0), (r1, s1, w1, g1, m1, f1) = res
    assert (s0, w0) == (s1, w1) and (g0, m0) == (7, 5) == (g1, m1)
    assert f0 == [0, 1, 2, 3, 4] and f1 == [5, 6, 7, 8, 9]          # contiguous chunks, disjoint, covering


def test_scene_blob_roundtrip_and_pan():
    sc = api.Scene(CORNELL)
    blob = adist.pack_scene(sc.geoms, sc.materials, sc.faces, None)
    geoms, mats, faces, box = adist.unpack_scene(blob)
    assert [bytes(g) for g in geoms] == [bytes(g) for g in sc.geoms]
    assert [bytes(m) for m in mats] == [bytes(m) for m in sc.materials] and faces == []
    assert adist.pan_phi(0.0, 0) == 0.0 and abs(adist.pan_phi(0.0, 75) - 0.35) < 1e-6
    assert adist.pan_phi(0.0, 300) == pytest.approx(0.0, abs=1e-6)
    # single-process: broadcast is the identity
    assert adist.broadcast_bytes(b"abc", 0, torch.device("cpu")) == b"abc"


# ---- the multi-GPU run checks itself: checksums of every rank's first / last frame, one all-gather, rank 0 re-renders ----------
def _fake_frame(g):
    """stands in for the denoised frame of global frame g (the CPU tests have no GPU to render one)"""
    rng = np.random.default_rng(1000 + g)
    return torch.from_numpy(rng.standard_normal((3, 24, 32)).astype(np.float32))


def _check_worker(rank, world, port, q, corrupt_rank):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    per = 6
    frames = list(adist.frame_shard(rank, world, per))
    first, last = _fake_frame(frames[0]), _fake_frame(frames[-1])
    if rank == corrupt_rank:
        last[1, 2, 3] += 1e-6                                   # one bit pattern off in one rank's last frame
    local = [adist.checksum64(first), adist.checksum64(last)]

    def rerender(r):
        fr = list(adist.frame_shard(r, world, per))
        return [adist.checksum64(_fake_frame(fr[0])), adist.checksum64(_fake_frame(fr[-1]))]
    ok, per_rank = adist.sharded_equals_single(local, rerender, dev, rank)
    fps = adist.gather_int64([1000 + rank], dev)
    q.put((rank, ok, per_rank, fps))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("corrupt_rank", [-1, 1])
def test_sharded_equals_single_world2(corrupt_rank):
    """bench.py --gpus N / SURVEY 8e: rank 0 learns through ONE all-gather of 8-byte checksums whether every rank's first and
    last frame equal its own render of them; a single flipped value on one rank is reported for that rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_check_worker, args=(r, 2, port, q, corrupt_rank)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, ok0, per0, fps0), (_, ok1, per1, fps1) = res
    assert ok1 is None and per1 is None                         # only rank 0 holds the verdict
    assert fps0 == fps1 == [[1000], [1001]]
    if corrupt_rank < 0:
        assert ok0 is True and per0 == [True, True]
    else:
        assert ok0 is False and per0 == [True, False]


def test_checksum64_is_position_sensitive_and_order_free():
    a = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    b = a.clone()
    b[0, 0], b[0, 1] = a[0, 1], a[0, 0]                         # same multiset of values, different positions
    assert adist.checksum64(a) == adist.checksum64(a.clone()) != adist.checksum64(b)
    assert adist.checksum64(torch.tensor([0.0])) != adist.checksum64(torch.tensor([-0.0]))    # bit patterns, not values
    # not periodic in the position: an exchange between positions 65521 apart (the period of the round-3 weights) is seen
    t = torch.arange(200000, dtype=torch.float32)
    u = t.clone()
    u[5], u[5 + 65521] = t[5 + 65521], t[5]
    assert adist.checksum64(t) != adist.checksum64(u)
