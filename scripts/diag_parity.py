"""Diagnostic (not a test): error of both HIP MLP kernels and of the fp32 oracle against an fp64 evaluation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oracle
from intrinsicnerf_amd import _capi, kernels, packing
dev = torch.device("cuda:0")
n = 512
g = torch.Generator().manual_seed(0)
o = torch.tensor([[2.5, 1.5, 2.0]]).expand(n, 3)
d = -o / o.norm(dim=-1, keepdim=True) + 0.2 * torch.randn(n, 3, generator=g)
rays = torch.cat([o, d, 2 * torch.ones(n, 1), 6 * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1)
t_vals, u = torch.linspace(0., 1., 64), torch.linspace(0., 1., 128)
for name in ("1/f calibrated", "default-init"):
    if name == "default-init":
        sd_c, sd_f = oracle.make_state_dict("object", seed=0), oracle.make_state_dict("object", seed=1)
    else:
        sd_c, _ = oracle.calibrated_lcg_weights("object", 0, 30, rays[:128]); sd_f, _ = oracle.calibrated_lcg_weights("object", 0, 31, rays[:128])
    cfg = oracle.RenderConfig(variant="object", white_bkgd=True)
    with torch.no_grad():
        w32 = oracle.render_rays(rays, sd_c, sd_f, cfg, t_vals=t_vals, u=u, stages=True)
        w64 = oracle.render_rays(rays.double(), {k: v.double() for k, v in sd_c.items()}, {k: v.double() for k, v in sd_f.items()},
                                 cfg, t_vals=t_vals.double(), u=u.double(), stages=True)
    print(f"--- weights: {name}")
    def rms(a, b):
        a, b = a.double(), b.double()
        m = ~(torch.isnan(a) | torch.isnan(b))
        return float(((a - b)[m] ** 2).mean().sqrt()), float((a - b)[m].abs().max())
    res = {"oracle fp32": w32}
    for prec, tag in ((_capi.PREC_F32, "hip f32"), (_capi.PREC_F16X3, "hip f16x3")):
        desc = _capi.net_desc(0, 0, 10, 4, 1.0, prec)
        res[tag] = {k: v.cpu() for k, v in kernels.render_rays_fused(
            desc, packing.pack_state_dict(desc, sd_c).to(dev), packing.pack_state_dict(desc, sd_f).to(dev), rays.to(dev), 64, 128,
            t_vals.to(dev), u.to(dev), white_bkgd=True, want_stages=True, want_raw_coarse=True, want_raw_fine=True).items()}
    for k in ("raw_coarse", "weights_coarse", "rgb_coarse", "z_samples", "rgb_fine", "acc_fine", "depth_fine"):
        print(f"{k:15s} vs fp64 (rms / max):  " + "   ".join(f"{tag}: {rms(r[k], w64[k])[0]:.2e} / {rms(r[k], w64[k])[1]:.2e}" for tag, r in res.items()))
