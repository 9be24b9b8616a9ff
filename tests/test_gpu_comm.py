"""-m gpu: the one-time broadcast between contexts (aipt_comm_*, csrc/comm.cpp): RCCL when every context has its own GPU
(exercised here with the GPUs this box has), the in-process shim when contexts share one."""
import ctypes as C

import numpy as np
import pytest

from ai_path_tracer_denoiser_amd import api, synth

pytestmark = pytest.mark.gpu


def _bcast(ctxs, payload, force_shim):
    L = api.lib()
    n = len(ctxs)
    hs = (api._P * n)(*[c._h for c in ctxs])
    comm = api._P()
    rc = L.aipt_comm_create(hs, n, 1 if force_shim else 0, C.byref(comm))
    assert rc == 0, L.aipt_last_error(ctxs[0]._h)
    is_rccl = L.aipt_comm_is_rccl(comm)
    bufs = (api._P * n)()
    for r, c in enumerate(ctxs):
        p = api._P()
        assert L.aipt_malloc(c._h, len(payload), C.byref(p)) == 0
        bufs[r] = p
        fill = payload if r == 0 else bytes(len(payload))
        assert L.aipt_upload(c._h, p, fill, len(payload)) == 0
    rc = L.aipt_comm_broadcast(comm, bufs, len(payload), 0)
    assert rc == 0, L.aipt_last_error(ctxs[0]._h)
    got = []
    for r, c in enumerate(ctxs):
        out = C.create_string_buffer(len(payload))
        assert L.aipt_download(c._h, out, bufs[r], len(payload)) == 0
        got.append(out.raw)
        L.aipt_free(c._h, bufs[r])
    L.aipt_comm_destroy(comm)
    return is_rccl, got


def test_shim_broadcast_between_contexts_on_one_gpu():
    ctxs = [api.Context(0) for _ in range(3)]
    payload = synth.make_blob(565)                       # the 6 MB weight blob
    is_rccl, got = _bcast(ctxs, payload, force_shim=False)
    assert not is_rccl and all(g == payload for g in got)
    for c in ctxs:
        c.close()


def test_rccl_broadcast_over_the_distinct_gpus_present():
    import torch
    n = torch.cuda.device_count()
    ctxs = [api.Context(d) for d in range(n)]            # one context per GPU: the RCCL transport (n = 1 on a one-GPU box)
    payload = np.random.default_rng(1).integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
    is_rccl, got = _bcast(ctxs, payload, force_shim=False)
    assert is_rccl and all(g == payload for g in got)
    for c in ctxs:
        c.close()
