#!/usr/bin/env python3
"""Do two denoiser passes on two streams overlap on one MI355X?  Two contexts (each its own stream and activation set)
denoise their own frame sequences, launched alternately from one host thread; compared with one context doing all frames."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from ai_path_tracer_denoiser_amd import api, synth
    H, W = 736, 1280
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    blob = synth.make_blob(1)
    nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    opts = [kv.split("=") for kv in sys.argv[3:]]              # aipt_denoise_set_option experiments: OPTION=VALUE ...
    ctxs = [api.Context(0) for _ in range(nctx)]
    xs, ys = [], []
    for c in ctxs:
        c.denoise_configure(H, W)
        c.load_weights(blob)
        for k, v in opts:
            c.denoise_set_option(int(k), int(v))
        xs.append(torch.from_numpy(synth.make_gbuffer(H, W, 3, 0)).cuda())
        ys.append(torch.empty(3, H, W, device="cuda"))

    def run(which, frames):
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(frames):
            for w in which:
                ctxs[w].denoise(xs[w], ys[w], bn_batch=True, carry=True)
        for c in ctxs:
            c.sync()
        return (time.perf_counter() - t0) / (frames * len(which))

    run(list(range(nctx)), 10)
    one = run([0], n)
    two = run(list(range(nctx)), n)
    print(f"one stream: {one * 1e3:.3f} ms per denoise; {nctx} streams interleaved: {two * 1e3:.3f} ms per denoise "
          f"({one / two:.2f}x throughput)")


if __name__ == "__main__":
    main()
