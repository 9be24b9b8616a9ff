"""Deterministic synthetic weights and G-buffer inputs.

The reference ships no trained weights and no dataset, so tests,
golden-vector generation and the bench all draw from this generator.  Only integer
hashing and exactly-rounded fp32 operations (mul/add/div/sqrt) are used, so every
machine regenerates bit-identical arrays; nothing large needs committing.

Weight statistics follow the reference's init (training/train.py:32-38): conv weights
with Kaiming fan-in variance 2/fan_in (uniform here instead of normal so that no libm
call is involved), bias 0.01.  BatchNorm parameters are *randomised* (the reference
init is gamma=1, beta=0, mean=0, var=1) so that parity tests exercise every term,
including negative gamma.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from . import arch

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix64(x):
    x = x.astype(np.uint64, copy=True)
    x ^= x >> np.uint64(30)
    x *= _M1
    x ^= x >> np.uint64(27)
    x *= _M2
    x ^= x >> np.uint64(31)
    return x


def uniform01(n, seed, stream):
    """n fp32 values in [0,1), a pure function of (seed, stream, index)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        key = _mix64(np.array([np.uint64(seed) * _GOLD + np.uint64(stream)], dtype=np.uint64))[0]
        h = _mix64(idx * _GOLD + key)
    return ((h >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def make_params(seed=565, bn_random=True):
    """Synthetic parameters for the 28 (conv, BN) pairs, keyed by arch.layer_table() names."""
    params = OrderedDict()
    for li, (name, _, _, cin, cout) in enumerate(arch.layer_table()):
        fan_in = cin * 9
        bound = np.float32(np.sqrt(np.float32(6.0) / np.float32(fan_in)))   # uniform with var 2/fan_in
        u = uniform01(cout * cin * 9, seed, 16 * li + 0)
        w = ((u - np.float32(0.5)) * np.float32(2.0) * bound).astype(np.float32).reshape(cout, cin, 3, 3)
        b = np.full(cout, 0.01, np.float32)
        if bn_random:
            g = np.float32(0.5) + uniform01(cout, seed, 16 * li + 1)            # 0.5 .. 1.5
            sgn = np.where(uniform01(cout, seed, 16 * li + 2) < np.float32(0.1), np.float32(-1), np.float32(1))
            gamma = (g * sgn).astype(np.float32)
            beta = ((uniform01(cout, seed, 16 * li + 3) - np.float32(0.5)) * np.float32(0.4)).astype(np.float32)
            mean = ((uniform01(cout, seed, 16 * li + 4) - np.float32(0.5)) * np.float32(0.2)).astype(np.float32)
            var = (np.float32(0.5) + uniform01(cout, seed, 16 * li + 5)).astype(np.float32)
        else:
            gamma = np.ones(cout, np.float32)
            beta = np.zeros(cout, np.float32)
            mean = np.zeros(cout, np.float32)
            var = np.ones(cout, np.float32)
        params[name] = dict(w=w, b=b, gamma=gamma, beta=beta, mean=mean, var=var)
    return params


def make_blob(seed=565, bn_random=True) -> bytes:
    return arch.pack_blob(make_params(seed, bn_random))


_PALETTE = np.array([[.98, .98, .98], [.85, .35, .35], [.35, .85, .35], [5., 5., 5.], [.75, .7, .6]],
                    dtype=np.float32)   # Cornell materials (scenes/Scenes/cornell.txt) + light + Sponza stone


def make_gbuffer(H, W, seed=0, frame=0):
    """A G-buffer-like [10,H,W] fp32 input: noisy RGB in [0,1] with 2% light
    pixels at 5.0, unit normals, depth in [0,25), palette albedo, 10% all-zero miss pixels."""
    n = H * W
    s = 1000 + 97 * frame
    g = np.zeros((10, n), np.float32)
    for c in range(3):
        g[c] = uniform01(n, seed, s + c)
    light = uniform01(n, seed, s + 3) < np.float32(0.02)
    g[0:3, light] = np.float32(5.0)
    v = np.stack([uniform01(n, seed, s + 4 + c) * np.float32(2) - np.float32(1) for c in range(3)])
    nrm = np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]).astype(np.float32)
    nrm = np.where(nrm == 0, np.float32(1), nrm)
    g[3:6] = (v / nrm).astype(np.float32)
    g[6] = uniform01(n, seed, s + 7) * np.float32(25.0)
    pal = (uniform01(n, seed, s + 8) * np.float32(len(_PALETTE))).astype(np.int64)
    pal = np.minimum(pal, len(_PALETTE) - 1)
    g[7:10] = _PALETTE[pal].T
    miss = uniform01(n, seed, s + 9) < np.float32(0.10)
    g[:, miss] = 0
    return g.reshape(10, H, W)
