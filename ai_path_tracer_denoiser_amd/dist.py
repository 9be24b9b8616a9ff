"""Multi-GPU plumbing: frame sharding + the one-time broadcast of scene and weights.

The reference is single-GPU.  The hot path shards at FRAME granularity: a frame depends only
on the scene, the weights and its camera, so rank r renders a contiguous chunk of the frame sequence (contiguous so
that a carried recurrent hidden state is valid inside the chunk; the first frame of a chunk starts from a zero hidden
state).  The only collective is a one-time broadcast of the packed scene blob and the weight blob from rank 0
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  Nothing is exchanged
per frame.
"""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

from . import api

SCENE_MAGIC = b"AIPTSC01"


def pack_scene(geoms, materials, faces, mesh_box) -> bytes:
    """Scene blob: magic, counts, then the raw POD arrays (layouts of include/aiptd.h)."""
    out = [SCENE_MAGIC, struct.pack("<III", len(geoms), len(materials), len(faces))]
    out += [bytes(g) for g in geoms]
    out += [bytes(m) for m in materials]
    out += [bytes(f) for f in faces]
    out.append(bytes(mesh_box) if mesh_box is not None else bytes(C.sizeof(api.AABB)))
    return b"".join(out)


def unpack_scene(blob: bytes):
    assert blob[:8] == SCENE_MAGIC, "bad scene blob"
    ng, nm, nf = struct.unpack_from("<III", blob, 8)
    off = 20
    def take(cls, n):
        nonlocal off
        sz = C.sizeof(cls)
        items = [cls.from_buffer_copy(blob[off + i * sz: off + (i + 1) * sz]) for i in range(n)]
        off += n * sz
        return items
    geoms, mats, faces = take(api.Geom, ng), take(api.Material, nm), take(api.Face, nf)
    box = take(api.AABB, 1)[0]
    assert off == len(blob)
    return geoms, mats, faces, box


def broadcast_bytes(payload, src: int, device):
    """Broadcast a bytes object from rank `src` to every rank (two collectives: length, then data)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return payload
    rank = dist.get_rank()
    n = torch.tensor([len(payload) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    if rank == src:
        buf = torch.from_numpy(np.frombuffer(payload, np.uint8).copy()).to(device)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src)
    return bytes(buf.cpu().numpy().tobytes())


def frame_shard(rank: int, world: int, frames_per_rank: int) -> range:
    """Global frame indices of rank `rank`: a contiguous chunk (weak scaling: frames_per_rank is fixed)."""
    return range(rank * frames_per_rank, (rank + 1) * frames_per_rank)


def pan_phi(phi0: float, frame: int, period: int = 300, amplitude: float = 0.35) -> float:
    """Build-defined orbit pan (the reference's lives on its absent data_gen branch):
    phi_k = phi0 + 0.35 * sin(2*pi*k/300), rounded to fp32 like the reference's float phi."""
    return float(np.float32(phi0 + amplitude * np.sin(2.0 * np.pi * frame / period)))
