import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ai_path_tracer_denoiser_amd import api, synth, arch
H, W = 384, 640
ctx = api.Context(0)
blob = synth.make_blob(565)
ctx.load_weights(blob); ctx.denoise_configure(H, W); ctx.denoise_set_impl(api.DN_IMPL_MFMA_F16X3)
for mx in (1.0e6, 1.04e6, 1.06e6, 1.2e6, 3e6):
    for rmin in (200000, 10**9):
        ctx.denoise_set_option(api.DN_OPT_R_MINPIX, rmin)
        x = synth.make_gbuffer(H, W, 3, 0)
        x[6] *= np.float32(mx / float(np.abs(x[6]).max()))
        y = torch.empty(3, H, W, device="cuda")
        ctx.reset_hidden()
        ctx.denoise(torch.from_numpy(x).cuda(), y, bn_batch=True, carry=False)
        ctx.sync()
        yy = y.cpu().numpy()
        hid = []
        for lvl, shp in enumerate(arch.hidden_shapes(H, W)):
            h = torch.empty(*shp, device="cuda"); ctx.get_hidden(lvl, h); ctx.sync()
            hid.append(bool(torch.isfinite(h).all()))
        print(mx, rmin, ctx.layer_info(0)["kernel"], "finite:", bool(np.isfinite(yy).all()), "nan frac", float(np.isnan(yy).mean()), "hidden finite", hid)
