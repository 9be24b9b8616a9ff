"""How far is the fp16-weight mode (AIPT_DN_IMPL_MFMA_F16W) from the reference goldens (fp32 weights)?  Prints the max abs
output error per golden file for both split-fp16 implementations.  python tools/f16w_error.py (on the GPU box)"""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ai_path_tracer_denoiser_amd import api, synth

ctx = api.Context(0)
for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "denoise_*.npz"))):
    g = np.load(path)
    H, W, wseed, iseed, nfr, batch = [int(v) for v in g["meta"]]
    frames = [synth.make_gbuffer(H, W, iseed, j) for j in range(nfr)]
    row = []
    for impl in (api.DN_IMPL_MFMA_F16X3, api.DN_IMPL_MFMA_F16W):
        ctx.load_weights(synth.make_blob(wseed)); ctx.denoise_configure(H, W); ctx.denoise_set_impl(impl); ctx.reset_hidden()
        err = 0.0
        for j, x in enumerate(frames):
            y = torch.empty(3, H, W, device="cuda")
            xd = torch.from_numpy(x).cuda()
            ctx.denoise(xd, y, bn_batch=bool(batch), carry=j > 0)
            ctx.sync()                                   # the context runs on its own stream: keep xd alive until done
            err = max(err, float(np.abs(y.cpu().numpy() - g["out"][j]).max()))
        row.append(err)
    print(f"{os.path.basename(path):32s} ref max {float(np.abs(g['out']).max()):7.3f}   f16x3 err {row[0]:.2e}   f16w err {row[1]:.2e}")
