"""Diagnostic (not a test): error of the HIP path and of the fp32 oracle against an fp64 evaluation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oracle
from intrinsicnerf_amd import _capi, kernels, packing
dev = torch.device("cuda:0")
torch.manual_seed(0)
n = 512
g = torch.Generator().manual_seed(0)
o = torch.tensor([[2.5, 1.5, 2.0]]).expand(n, 3)
d = -o / o.norm(dim=-1, keepdim=True) + 0.2 * torch.randn(n, 3, generator=g)
rays = torch.cat([o, d, 2 * torch.ones(n, 1), 6 * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1)
for name, kw in [("strong", dict(sigma_gain_log2=5, sigma_bias=-6.0, weight_gain_log2=1)), ("default", None)]:
    if kw is None:
        sd_c, sd_f = oracle.make_state_dict("object", seed=0), oracle.make_state_dict("object", seed=1)
    else:
        sd_c, sd_f = oracle.lcg_state_dict("object", seed=30, **kw), oracle.lcg_state_dict("object", seed=31, **kw)
    cfg = oracle.RenderConfig(variant="object", white_bkgd=True)
    t_vals, u = torch.linspace(0., 1., 64), torch.linspace(0., 1., 128)
    with torch.no_grad():
        w32 = oracle.render_rays(rays, sd_c, sd_f, cfg, t_vals=t_vals, u=u, stages=True)
        sd_c64 = {k: v.double() for k, v in sd_c.items()}; sd_f64 = {k: v.double() for k, v in sd_f.items()}
        w64 = oracle.render_rays(rays.double(), sd_c64, sd_f64, cfg, t_vals=t_vals.double(), u=u.double(), stages=True)
    desc = _capi.net_desc(0, 0, 10, 4, 1.0)
    got = kernels.render_rays_fused(desc, packing.pack_state_dict(desc, sd_c).to(dev), packing.pack_state_dict(desc, sd_f).to(dev),
                                    rays.to(dev), 64, 128, t_vals.to(dev), u.to(dev), white_bkgd=True, want_stages=True,
                                    want_raw_coarse=True, want_raw_fine=True)
    torch.cuda.synchronize()
    print(f"--- weights: {name}")
    def rel(a, b):
        a, b = a.double(), b.double()
        m = ~(torch.isnan(a) | torch.isnan(b))
        return float(((a - b).abs()[m] / (1e-1 + b.abs()[m])).max())
    for k in ("raw_coarse", "weights_coarse", "rgb_coarse", "z_samples", "raw_fine", "rgb_fine", "albedo_fine", "shading_fine",
              "residual_fine", "acc_fine", "depth_fine", "disp_fine", "z_std"):
        print(f"{k:16s} hip-vs-o32 {rel(got[k].cpu(), w32[k]):.2e}   hip-vs-o64 {rel(got[k].cpu(), w64[k]):.2e}   o32-vs-o64 {rel(w32[k], w64[k]):.2e}")
