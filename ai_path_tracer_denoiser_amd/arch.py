"""Denoiser architecture table and the flat weight-blob format.

The network is the reference's recurrent auto-encoder
(reference: training/recurrent_autoencoder_model.py:93-142).  It is 28
(conv3x3 + BatchNorm) pairs.  This module is pure data: the layer list, the
mapping from the reference's ``state_dict`` keys to blob order, and
pack/unpack of the flat blob that ``aipt_denoise_load_weights`` consumes
(include/aiptd.h).

Blob layout (little endian):
    char[8]  magic  = b"AIPTDW01"
    u32      n_layers (= 28)
    u32      reserved (= 0)
    n_layers x { u32 cin, u32 cout }
    then per layer, fp32:  W[cout][cin][3][3], bias[cout],
                           gamma[cout], beta[cout], running_mean[cout], running_var[cout]
"""
from __future__ import annotations

import struct
from collections import OrderedDict

import numpy as np

MAGIC = b"AIPTDW01"
IN_CHANNELS = 10
OUT_CHANNELS = 3
ENC_CH = [32, 43, 57, 76, 101]          # recurrent_autoencoder_model.py:98-107
BOTT_CH = 101                           # :109
DEC_CH = {5: 76, 4: 57, 3: 43, 2: 32, 1: 3}   # :111-115
BN_EPS = 1e-5
LRELU_SLOPE = 0.1


def layer_table():
    """[(name, conv_key, bn_key, cin, cout)] in network (and blob) order.

    conv_key / bn_key are the reference ``state_dict`` prefixes
    (SURVEY.md Appendix A.3)."""
    L = []
    cin = IN_CHANNELS
    for i, c in enumerate(ENC_CH, start=1):
        p = f"encoder{i}.0."
        L.append((f"enc{i}.l1", p + "layer1.0", p + "layer1.1", cin, c))
        L.append((f"enc{i}.l2a", p + "layer2.0", p + "layer2.2", 2 * c, c))
        L.append((f"enc{i}.l2b", p + "layer2.3", p + "layer2.4", c, c))
        cin = c
    p = "bottleneck."
    L.append(("bott.l1", p + "layer1.0", p + "layer1.1", cin, BOTT_CH))
    L.append(("bott.l2a", p + "layer2.0", p + "layer2.1", 2 * BOTT_CH, BOTT_CH))
    L.append(("bott.l2b", p + "layer2.3", p + "layer2.4", BOTT_CH, BOTT_CH))
    prev = BOTT_CH
    for k in (5, 4, 3, 2, 1):
        p = f"decoder{k}.layer1."
        skip = ENC_CH[k - 1]
        assert prev == skip                 # cat(prev, skip) = 2*in_channel (:41)
        L.append((f"dec{k}.c1", p + "1", p + "2", prev + skip, DEC_CH[k]))
        L.append((f"dec{k}.c2", p + "4", p + "5", DEC_CH[k], DEC_CH[k]))
        prev = DEC_CH[k]
    assert len(L) == 28
    return L


def n_parameters():
    return sum(9 * cin * cout + cout + 2 * cout for _, _, _, cin, cout in layer_table())


def hidden_shapes(H, W):
    """6 recurrent hidden states (model.py:121-128): (C, H/f, W/f)."""
    chans = ENC_CH + [BOTT_CH]
    facs = [1, 2, 4, 8, 16, 32]
    return [(c, H // f, W // f) for c, f in zip(chans, facs)]


def conv_flops(H, W):
    """2*9*Cin*Cout*h*w summed over the 28 convs."""
    res = {}
    for i in range(1, 6):
        res[f"enc{i}"] = (H >> (i - 1), W >> (i - 1))
    res["bott"] = (H >> 5, W >> 5)
    for k in (5, 4, 3, 2, 1):
        res[f"dec{k}"] = (H >> (k - 1), W >> (k - 1))
    tot = 0
    for name, _, _, cin, cout in layer_table():
        h, w = res[name.split(".")[0]]
        tot += 2 * 9 * cin * cout * h * w
    return tot


def activation_bytes(H, W, elem=4):
    """Algorithmic HBM bytes of one forward: every tensor written once and
    read once per consumer at its stored resolution, pool/upsample/concat/BN/LReLU fused."""
    def sz(c, lvl):
        return c * (H >> lvl) * (W >> lvl) * elem
    tot = sz(IN_CHANNELS, 0)                       # read input
    for i, c in enumerate(ENC_CH):
        tot += 2 * sz(c, i)                        # out1 write + read
        tot += sz(c, i)                            # hidden read
        tot += 2 * sz(c, i)                        # conv2 out write + read
        tot += sz(c, i)                            # out2 write (= new hidden)
        tot += 2 * sz(c, i)                        # out2 read by pool consumers (next enc + decoder skip)
    c = BOTT_CH
    tot += 2 * sz(c, 5) + sz(c, 5) + 2 * sz(c, 5) + 2 * sz(c, 5)
    for k in (5, 4, 3, 2, 1):
        tot += 2 * sz(DEC_CH[k], k - 1)            # c1 out write + read
        tot += sz(DEC_CH[k], k - 1)                # c2 out write
        if k > 1:
            tot += sz(DEC_CH[k], k - 1)            # read by next decoder
    return tot


def pack_blob(params: "OrderedDict[str, dict]") -> bytes:
    """params[name] = dict(w, b, gamma, beta, mean, var) as float32 arrays."""
    tbl = layer_table()
    out = [MAGIC, struct.pack("<II", len(tbl), 0)]
    for name, _, _, cin, cout in tbl:
        out.append(struct.pack("<II", cin, cout))
    for name, _, _, cin, cout in tbl:
        p = params[name]
        w = np.ascontiguousarray(p["w"], dtype=np.float32)
        assert w.shape == (cout, cin, 3, 3), (name, w.shape)
        out.append(w.tobytes())
        for k in ("b", "gamma", "beta", "mean", "var"):
            v = np.ascontiguousarray(p[k], dtype=np.float32)
            assert v.shape == (cout,), (name, k, v.shape)
            out.append(v.tobytes())
    return b"".join(out)


def unpack_blob(blob: bytes) -> "OrderedDict[str, dict]":
    assert blob[:8] == MAGIC, "bad weight blob magic"
    n, _ = struct.unpack_from("<II", blob, 8)
    tbl = layer_table()
    assert n == len(tbl)
    off = 16
    for name, _, _, cin, cout in tbl:
        ci, co = struct.unpack_from("<II", blob, off)
        assert (ci, co) == (cin, cout), (name, ci, co)
        off += 8
    params = OrderedDict()
    for name, _, _, cin, cout in tbl:
        p = {}
        nw = cout * cin * 9
        p["w"] = np.frombuffer(blob, np.float32, nw, off).reshape(cout, cin, 3, 3).copy()
        off += 4 * nw
        for k in ("b", "gamma", "beta", "mean", "var"):
            p[k] = np.frombuffer(blob, np.float32, cout, off).copy()
            off += 4 * cout
        params[name] = p
    assert off == len(blob), (off, len(blob))
    return params


def blob_from_state_dict(sd) -> bytes:
    """Weight interchange: reference ``{'net': state_dict}`` -> flat blob.
    ``sd`` maps the reference key names (A.3) to array-likes (torch tensors or numpy)."""
    def arr(x):
        return np.asarray(x.detach().cpu().numpy() if hasattr(x, "detach") else x, dtype=np.float32)
    params = OrderedDict()
    for name, ck, bk, cin, cout in layer_table():
        params[name] = dict(w=arr(sd[ck + ".weight"]), b=arr(sd[ck + ".bias"]),
                            gamma=arr(sd[bk + ".weight"]), beta=arr(sd[bk + ".bias"]),
                            mean=arr(sd[bk + ".running_mean"]), var=arr(sd[bk + ".running_var"]))
    return pack_blob(params)


def state_dict_from_file(path: str):
    """The reference's two weight containers -> state_dict:
    * a TorchScript archive as training/convert_to_torchscript.py:29-30 saves it (torch.jit.trace(model.forward).save(...),
      the file the C++ host loads with torch::jit::load, main.cpp:107): the traced module keeps the model's parameter and
      buffer names, so torch.jit.load(path).state_dict() is the state_dict;
    * a training checkpoint ``{'net': state_dict}`` (training/train.py:109-112), or a bare state_dict."""
    import torch
    try:
        return torch.jit.load(path, map_location="cpu").state_dict()
    except RuntimeError:
        pass
    ck = torch.load(path, map_location="cpu")
    if isinstance(ck, dict) and "net" in ck:
        ck = ck["net"]
    return ck


def blob_from_file(path: str) -> bytes:
    """TorchScript archive or ``{'net': ...}`` checkpoint -> flat weight blob (tools/export_weights.py)."""
    return blob_from_state_dict(state_dict_from_file(path))


def state_dict_from_params(params):
    """Inverse mapping (numpy arrays keyed by the reference's state_dict names); used only
    by the golden-vector generator to load our synthetic weights into the reference model."""
    sd = OrderedDict()
    for name, ck, bk, cin, cout in layer_table():
        p = params[name]
        sd[ck + ".weight"] = p["w"]
        sd[ck + ".bias"] = p["b"]
        sd[bk + ".weight"] = p["gamma"]
        sd[bk + ".bias"] = p["beta"]
        sd[bk + ".running_mean"] = p["mean"]
        sd[bk + ".running_var"] = p["var"]
    return sd
