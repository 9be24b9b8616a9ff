"""Pin the CPU oracle of the denoiser against golden vectors produced by the REFERENCE model
(tests/golden/gen_denoise_goldens.py imports training/recurrent_autoencoder_model.py).
Tolerance: fp32 re-association noise only (the oracle and torch sum in different orders)."""
import glob
import os

import numpy as np
import pytest

from ai_path_tracer_denoiser_amd import arch, synth
from oracle import DenoiseOracle

TOL = 2e-4   # max abs on O(1..10) activations after 28 fp32 layers; typical observed ~1e-5


def _cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "denoise_*.npz")))


def test_goldens_present(golden_dir):
    assert len(_cases(golden_dir)) >= 6


@pytest.mark.parametrize("name", ["b_reset_64", "r_reset_64", "b_carry_64", "r_carry_64",
                                  "b_carry_96x160", "r_reset_96x160"])
def test_oracle_matches_reference_goldens(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"denoise_{name}.npz"))
    H, W, wseed, iseed, frames, batch = [int(v) for v in g["meta"]]
    orc = DenoiseOracle(synth.make_blob(wseed), H, W)
    for j in range(frames):
        x = synth.make_gbuffer(H, W, iseed, j)
        y = orc.forward(x, bn_batch=bool(batch), carry=(j > 0))
        ref = g["out"][j]
        err = np.abs(y - ref).max()
        assert err <= 1e-3, (name, j, err)                       # north_star's bar, absolute (measured: 2e-5 .. 3.6e-4)
    for lvl, shp in enumerate(arch.hidden_shapes(H, W)):
        h = orc.hidden(lvl).astype(np.float64)
        flat = h.reshape(shp[0], -1)
        np.testing.assert_allclose(flat.mean(axis=1), g[f"h{lvl}_mean"], atol=2e-4, rtol=1e-4)
        np.testing.assert_allclose((flat * flat).mean(axis=1), g[f"h{lvl}_msq"], atol=2e-4, rtol=1e-3)
        np.testing.assert_allclose(h.reshape(-1)[g[f"h{lvl}_idx"]], g[f"h{lvl}_samples"],
                                   atol=2e-4 * max(1.0, np.abs(g[f"h{lvl}_samples"]).max()))


def test_oracle_matches_the_reference_model_where_the_register_staged_kernel_runs(golden_dir):
    """[r5] denoise_b_reset_384x640.npz: one full output frame of the imported reference model at a size whose level 0 the GPU
    runs on conv3x3_f16x3r (>= 200 000 pixels).  (The 736x1280 case -- samples + moments -- is checked on the GPU box, where the
    oracle's 140 GFLOP take a second instead of a minute: tests/test_gpu_denoise_kernels.py.)"""
    g = np.load(os.path.join(golden_dir, "denoise_b_reset_384x640.npz"))
    H, W, wseed, iseed, frames, batch = [int(v) for v in g["meta"]]
    assert (H, W, frames, batch) == (384, 640, 1, 1)
    y = DenoiseOracle(synth.make_blob(wseed), H, W).forward(synth.make_gbuffer(H, W, iseed, 0), bn_batch=True, carry=False)
    err = np.abs(y - g["out"][0]).max()
    assert err <= 1e-3, err


@pytest.mark.parametrize("name", ["b_carry_384x640", "r_carry_384x640"])
def test_oracle_matches_the_reference_model_with_a_carried_hidden_state_at_384x640(golden_dir, name):
    """[r6] denoise_b_carry_384x640.npz: model(x0, 0), model(x1, 1) of the imported reference model (131 072 strided samples over
    both frames + moments): the oracle's hidden-channel path at a size whose level 0 the GPU runs on conv3x3_f16x3r."""
    g = np.load(os.path.join(golden_dir, f"denoise_{name}.npz"))
    H, W, wseed, iseed, frames, batch = [int(v) for v in g["meta"]]
    assert (H, W, frames, batch) == (384, 640, 2, 1 if name.startswith("b_") else 0)
    orc = DenoiseOracle(synth.make_blob(wseed), H, W)
    y = np.stack([orc.forward(synth.make_gbuffer(H, W, iseed, j), bn_batch=bool(batch), carry=j > 0) for j in range(frames)])
    tol = 1e-3
    err = np.abs(y.reshape(-1)[g["out_idx"]] - g["out_samples"]).max()
    assert err <= tol, err
    for j in range(frames):
        y64 = y[j].astype(np.float64).reshape(3, -1)
        np.testing.assert_allclose(y64.mean(axis=1), g["out_mean"][j], atol=2e-5, rtol=1e-5)


def test_reset_equals_first_carry_frame():
    """forward(x, j=0) semantics: a carry call with no valid hidden state starts from zeros."""
    blob = synth.make_blob(7)
    o1 = DenoiseOracle(blob, 32, 64)
    o2 = DenoiseOracle(blob, 32, 64)
    x = synth.make_gbuffer(32, 64, 3, 0)
    a = o1.forward(x, True, carry=False)
    b = o2.forward(x, True, carry=True)
    assert np.array_equal(a, b)


def test_rejects_non_multiple_of_32():
    with pytest.raises(ValueError):
        DenoiseOracle(synth.make_blob(1), 720, 1280)   # the reference model fails here too


def test_blob_roundtrip_and_param_count():
    assert arch.n_parameters() == 1508181          # SURVEY Appendix A.3 [probed]
    p = synth.make_params(3)
    q = arch.unpack_blob(arch.pack_blob(p))
    for k in p:
        for f in p[k]:
            assert np.array_equal(p[k][f], q[k][f])
    assert abs(arch.conv_flops(736, 1280) / 1e9 - 139.87) < 0.01


def test_state_dict_exporter_roundtrip():
    """Weight interchange: the reference's {'net': state_dict} key names (Appendix A.3) -> flat blob."""
    import torch
    p = synth.make_params(9)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in arch.state_dict_from_params(p).items()}
    sd["encoder1.0.layer1.1.num_batches_tracked"] = torch.tensor(3)        # extra BN buffers are ignored
    assert arch.blob_from_state_dict(sd) == arch.pack_blob(p)
    names = [t[1] for t in arch.layer_table()]
    assert names[0] == "encoder1.0.layer1.0" and names[4] == "encoder2.0.layer2.0" and names[-1] == "decoder1.layer1.4"


def test_torch_restatement_agrees_with_the_c_oracle():
    """oracle/torch_denoise.py (the PyTorch-CPU leg of bench.py's cpu_baseline) against the C restatement: two independent
    restatements of recurrent_autoencoder_model.py, all BN / hidden modes, two frames each."""
    import oracle
    from oracle.torch_denoise import TorchDenoiser
    from ai_path_tracer_denoiser_amd import synth
    blob = synth.make_blob(3)
    H, W = 64, 96
    rng = np.random.default_rng(0)
    orc = oracle.DenoiseOracle(blob, H, W)
    td = TorchDenoiser(blob)
    for bn_batch, carry in [(True, False), (True, True), (False, True), (False, False)]:
        orc.reset_hidden()
        td.hidden = None
        for k in range(2):
            x = rng.random((10, H, W), dtype=np.float32)
            a = orc.forward(x, bn_batch, carry and k > 0)
            b = td.forward(x, bn_batch, carry and k > 0)
            assert np.abs(a - b).max() < 1e-3, (bn_batch, carry, k, float(np.abs(a - b).max()))


def test_fp64_build_of_the_oracle_is_the_same_network():
    """oracle/liboracle64.so (-DORC_DN_FP64: double activations, products, sums, BatchNorm) is the truth of the drift study
    (tools/drift_probe.py): one frame of it and of the fp32 build differ by fp32 rounding only, in both BN modes, and its
    hidden state carries."""
    blob = synth.make_blob(565)
    H, W = 64, 96
    x0, x1 = synth.make_gbuffer(H, W, 3, 0), synth.make_gbuffer(H, W, 3, 1)
    for bn in (True, False):
        o32, o64 = DenoiseOracle(blob, H, W), DenoiseOracle(blob, H, W, fp64=True)
        a, b = o32.forward(x0, bn, False), o64.forward(x0, bn, False)
        assert np.isfinite(b).all() and 0 < np.abs(a - b).max() < 5e-4
        a1, b1 = o32.forward(x1, bn, True), o64.forward(x1, bn, True)
        assert np.abs(a1 - b1).max() < 2e-3
        assert np.abs(o32.hidden(0) - o64.hidden(0)).max() < 2e-3
        assert np.abs(o64.forward(x1, bn, False) - b1).max() > 1e-3      # the carried state matters
