"""PyTorch-CPU restatement of the denoiser forward pass -- TEST / BASELINE INFRASTRUCTURE, never the product path.

north_star asks for "PyTorch-CPU denoise timed on the same box's host cores": the reference's own model file cannot travel
to the GPU box, so this is the build's restatement of training/recurrent_autoencoder_model.py:8-142 with torch.nn.functional
(encoder conv -> LReLU -> BN order on layer2, bottleneck conv -> BN -> LReLU, decoders cat -> nearest
upsample x2 -> conv BN LReLU x2, hidden <- out2, MaxPool2d(2) after each encoder).  Only bench.py's cpu_baseline leg and
tests import it; tests/test_oracle_denoise.py checks it against the C oracle and, through the goldens, against the
reference."""
import numpy as np

ENC = ["enc1", "enc2", "enc3", "enc4", "enc5"]


class TorchDenoiser:
    def __init__(self, blob: bytes, threads=None):
        import torch
        from ai_path_tracer_denoiser_amd import arch
        self.torch = torch
        if threads:
            torch.set_num_threads(int(threads))
        self.p = {k: {n: torch.from_numpy(v) for n, v in d.items()} for k, d in arch.unpack_blob(blob).items()}
        self.hidden = None

    def _cbn(self, x, name, bn_batch, order):
        """order 'cba': conv, BN, LReLU; 'cab': conv, LReLU, BN (encoder layer2.0-2.2)"""
        F = self.torch.nn.functional
        p = self.p[name]
        y = F.conv2d(x, p["w"], p["b"], padding=1)
        def bn(t):
            return F.batch_norm(t, None if bn_batch else p["mean"], None if bn_batch else p["var"], p["gamma"], p["beta"],
                                training=bn_batch, eps=1e-5)
        if order == "cab":
            return bn(F.leaky_relu(y, 0.1))
        return F.leaky_relu(bn(y), 0.1)

    def forward(self, x10: np.ndarray, bn_batch: bool, carry: bool) -> np.ndarray:
        torch = self.torch
        F = torch.nn.functional
        with torch.no_grad():
            x = torch.from_numpy(np.ascontiguousarray(x10, np.float32))[None]
            hid = self.hidden if (carry and self.hidden is not None) else [None] * 6
            new_hid, skips = [], []
            for i, e in enumerate(ENC):
                o1 = self._cbn(x, e + ".l1", bn_batch, "cba")
                h = hid[i] if hid[i] is not None else torch.zeros_like(o1)
                z = self._cbn(torch.cat([o1, h], 1), e + ".l2a", bn_batch, "cab")
                o2 = self._cbn(z, e + ".l2b", bn_batch, "cba")
                new_hid.append(o2)
                x = F.max_pool2d(o2, 2)
                skips.append(x)
            o1 = self._cbn(x, "bott.l1", bn_batch, "cba")
            h = hid[5] if hid[5] is not None else torch.zeros_like(o1)
            z = self._cbn(torch.cat([o1, h], 1), "bott.l2a", bn_batch, "cba")
            o2 = self._cbn(z, "bott.l2b", bn_batch, "cba")
            new_hid.append(o2)
            prev = o2
            for k in (5, 4, 3, 2, 1):
                u = F.interpolate(torch.cat([prev, skips[k - 1]], 1), scale_factor=2, mode="nearest")
                prev = self._cbn(self._cbn(u, f"dec{k}.c1", bn_batch, "cba"), f"dec{k}.c2", bn_batch, "cba")
            self.hidden = new_hid
            return prev[0].numpy()
