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
