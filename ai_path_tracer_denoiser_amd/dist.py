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

SCENE_MAGIC = b"AIPTSB02"
_HDR = struct.Struct("<8s4Ii3I6fQ")          # PackHeader of csrc/trace.hip (72 bytes)
NODE_DTYPE = np.dtype([("p", np.float32, 3), ("exps", np.uint32), ("qlo", np.uint32, 3), ("qhi", np.uint32, 3),
                       ("ref", np.int32, 4), ("pad", np.uint32, 2)])                          # Bvh4Node, 64 B
TRI_DTYPE = np.dtype([("v0", np.float32, 3), ("e1", np.float32, 3), ("e2", np.float32, 3), ("face", np.int32),
                      ("pad", np.int32, 2)])                                                  # TriRec, 48 B
assert NODE_DTYPE.itemsize == 64 and TRI_DTYPE.itemsize == 48 and _HDR.size == 72


def pack_scene(geoms, materials, faces, mesh_box) -> bytes:
    """The packed scene blob of aipt_scene_pack: validated scene + the mesh BVH, built once on the host (rank 0)."""
    return api.scene_pack(geoms, materials, faces, mesh_box)


def _align16(v):
    return (v + 15) & ~15


def scene_sections(blob: bytes):
    """Header fields and section offsets of a packed scene (layout: csrc/trace.hip PackHeader)."""
    magic, ng, nm, nf, nn, need, _, _, _, *rest = _HDR.unpack_from(blob, 0)
    assert magic == SCENE_MAGIC, "bad scene blob"
    box, total = rest[:6], rest[6]
    assert total == len(blob)
    off = {}
    o = _align16(_HDR.size)
    for name, sz in (("geoms", 248 * ng), ("materials", 44 * nm), ("faces", 76 * nf), ("nodes", 64 * nn), ("tris", 48 * nf)):
        off[name] = o
        o = _align16(o + sz)
    assert o == len(blob)
    return dict(ngeoms=ng, nmaterials=nm, nfaces=nf, nnodes=nn, stack_need=need, box=box), off


def unpack_scene(blob: bytes):
    """-> (geoms, materials, faces, mesh_box) as api structs (inspection and tests; contexts take the blob as it is)."""
    h, off = scene_sections(blob)

    def take(cls, o, n):
        sz = C.sizeof(cls)
        return [cls.from_buffer_copy(blob[o + i * sz: o + (i + 1) * sz]) for i in range(n)]
    geoms = take(api.Geom, off["geoms"], h["ngeoms"])
    mats = take(api.Material, off["materials"], h["nmaterials"])
    faces = take(api.Face, off["faces"], h["nfaces"])
    box = api.AABB()
    box.lb[:] = h["box"][:3]
    box.ub[:] = h["box"][3:]
    return geoms, mats, faces, box


def scene_bvh(blob: bytes):
    """-> (nodes[NODE_DTYPE], tris[TRI_DTYPE], faces[synth.FACE_DTYPE], stack_need) views of a packed scene"""
    from . import synth
    h, off = scene_sections(blob)
    nodes = np.frombuffer(blob, NODE_DTYPE, h["nnodes"], off["nodes"])
    tris = np.frombuffer(blob, TRI_DTYPE, h["nfaces"], off["tris"])
    faces = np.frombuffer(blob, synth.FACE_DTYPE, h["nfaces"], off["faces"])
    return nodes, tris, faces, h["stack_need"]


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


# ---- the multi-GPU run checks itself ("8-GPU result == 1-GPU result per frame") ---------------------------------
def checksum64(t) -> int:
    """Position-weighted 64-bit checksum of the bit patterns of a float32 torch tensor (any device).  Integer arithmetic
    modulo 2^64: the value does not depend on the order a reduction adds in."""
    import torch
    bits = t.contiguous().view(torch.int32).flatten().to(torch.int64) & 0xFFFFFFFF
    # weight of position i: an odd 64-bit multiplicative hash of i (Knuth's golden-ratio constant, arithmetic modulo 2^64) -- not
    # periodic in i, so two elements swapped or a block shifted anywhere in the frame changes the sum (the earlier
    # (i mod 65521) + 1 missed exchanges between positions congruent modulo 65521)
    i = torch.arange(bits.numel(), device=bits.device, dtype=torch.int64)
    w = ((i + 1) * -7046029254386353131) | 1                 # 0x9E3779B97F4A7C15 as int64; int64 products wrap modulo 2^64
    return int((bits * w).sum().item())


def gather_int64(values, device):
    """all-gather of a short list of int64 per rank -> [world][len(values)] (identity without a process group)"""
    import torch
    import torch.distributed as dist
    mine = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [mine.cpu().tolist()]
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [o.cpu().tolist() for o in out]


def sharded_equals_single(local_sums, rerender, device, rank=0):
    """Every rank contributes the checksums of the first and the last denoised frame of its chunk (an 8-byte word each, one
    all-gather); rank 0 renders those frames again by itself -- rerender(r) -> the same checksums for rank r's chunk, computed
    frame by frame on rank 0's GPU -- and compares.  Returns (all equal, [per-rank equal]) on rank 0, (None, None) elsewhere."""
    gathered = gather_int64(local_sums, device)
    if rank != 0:
        return None, None
    per_rank = [[int(v) for v in rerender(r)] == [int(v) for v in gathered[r]] for r in range(len(gathered))]
    return all(per_rank), per_rank
