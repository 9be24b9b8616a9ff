"""Weight interchange: TorchScript archives and {'net': state_dict} checkpoints -> the flat blob.

The archive reader is checked three ways: (a) on an archive scripted from a module tree that carries the reference's parameter
and buffer names (runs everywhere), (b) on the archive torch.jit.trace(model.forward) produces from the REFERENCE model object
itself, exactly as convert_to_torchscript.py:26-30 does (only where /root/reference exists), (c) -m gpu: the blob exported from
an archive reproduces the reference golden outputs through the HIP denoiser."""
import os
import subprocess
import sys

import numpy as np
import pytest

from ai_path_tracer_denoiser_amd import arch, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/training"


def _named_module(sd):
    """a torch module tree whose state_dict() has exactly the keys of sd (names like encoder1.0.layer1.0.weight)"""
    import torch

    class Node(torch.nn.Module):
        pass
    root = Node()
    for key, val in sd.items():
        parts = key.split(".")
        m = root
        for p in parts[:-1]:
            if p not in m._modules:
                m.add_module(p, Node())
            m = m._modules[p]
        t = torch.from_numpy(np.asarray(val, np.float32).copy())
        if parts[-1] in ("running_mean", "running_var"):
            m.register_buffer(parts[-1], t)
        else:
            m.register_parameter(parts[-1], torch.nn.Parameter(t))
    return root


def _archive(tmp_path, seed):
    import torch
    sd = arch.state_dict_from_params(synth.make_params(seed))
    path = str(tmp_path / "model_ts.pt")
    torch.jit.script(_named_module(sd)).save(path)
    return path, sd


def test_blob_from_torchscript_archive_and_checkpoint(tmp_path):
    import torch
    path, sd = _archive(tmp_path, 3)
    assert arch.blob_from_file(path) == synth.make_blob(3)
    ck = str(tmp_path / "ck.pt")
    torch.save({"net": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}}, ck)
    assert arch.blob_from_file(ck) == synth.make_blob(3)
    out = tmp_path / "w.aiptw"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "export_weights.py"), path, str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == synth.make_blob(3)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference only exists in the build container")
def test_blob_from_the_reference_models_own_torchscript_archive(tmp_path):
    import warnings
    import torch
    sys.path.insert(0, REF)
    try:
        import recurrent_autoencoder_model as M
    finally:
        sys.path.remove(REF)
    model = M.AutoEncoder(10)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in arch.state_dict_from_params(synth.make_params(5)).items()}
    model.load_state_dict(sd, strict=False)                   # num_batches_tracked stays
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        traced = torch.jit.trace(model.forward, torch.zeros(1, 10, 64, 64))     # convert_to_torchscript.py:29
    path = str(tmp_path / "cpp_autoencoder.pt")
    traced.save(path)                                                             # :30
    blob = arch.blob_from_file(path)
    got = arch.unpack_blob(blob)
    want = synth.make_params(5)
    for name in want:
        for k in ("w", "b", "gamma", "beta"):
            assert np.array_equal(got[name][k], want[name][k]), (name, k)
        # the trace ran one forward in train mode: running statistics moved by momentum 0.1 from the loaded values
        assert got[name]["mean"].shape == want[name]["mean"].shape and np.isfinite(got[name]["var"]).all()


@pytest.mark.gpu
def test_exported_archive_blob_reproduces_the_reference_goldens(tmp_path, golden_dir):
    import torch
    from ai_path_tracer_denoiser_amd import api
    g = np.load(os.path.join(golden_dir, "denoise_b_reset_64.npz"))
    H, W, wseed, iseed, nfr, batch = [int(v) for v in g["meta"]]
    path, _ = _archive(tmp_path, wseed)
    blob = arch.blob_from_file(path)
    ctx = api.Context(0)
    ctx.load_weights(blob)
    ctx.denoise_configure(H, W)
    y = torch.empty(3, H, W, device="cuda")
    ctx.denoise(torch.from_numpy(synth.make_gbuffer(H, W, iseed, 0)).cuda(), y, bn_batch=bool(batch), carry=False)
    ctx.sync()
    assert np.abs(y.cpu().numpy() - g["out"][0]).max() <= 1e-3
    ctx.close()
