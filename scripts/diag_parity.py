"""Diagnostic (not a test): error of every HIP MLP kernel form and of the fp32 oracle against an fp64 evaluation of the
reference arithmetic.  Regenerates profiles/rNN_accuracy_vs_fp64.txt:

    python scripts/diag_parity.py > gpurun_out/accuracy_vs_fp64.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

import oracle  # noqa: E402
from oracle import calibration as cal  # noqa: E402
from intrinsicnerf_amd import _capi, kernels, packing  # noqa: E402

dev = torch.device("cuda:0")
n = 512
g = torch.Generator().manual_seed(0)
o = torch.tensor([[2.5, 1.5, 2.0]]).expand(n, 3)
d = -o / o.norm(dim=-1, keepdim=True) + 0.2 * torch.randn(n, 3, generator=g)
synthetic = torch.cat([o, d, 2 * torch.ones(n, 1), 6 * torch.ones(n, 1), d / d.norm(dim=-1, keepdim=True)], -1)
import test_unfiltered_parity as tup  # noqa: E402  (the benchmark frame's strided rays)
chair = tup.chair_rays(n)
t_vals, u = torch.linspace(0., 1., 64), torch.linspace(0., 1., 128)
print("# python scripts/diag_parity.py   (MI355X; 512 rays x (64+128) samples, object-level networks)")
print("# error of the fp32-class evaluations of the same path against an fp64 evaluation of the reference arithmetic:")
print("#   oracle fp32 = PyTorch-CPU restatement (bit-identical to the reference), hip f32 = exact-fp32 MFMA kernel,")
print("#   hip f16x3 2wg = k_encode_mlp_f16x3_dual (the DEFAULT kernel: two workgroups per CU, truncation split),")
print("#   hip f16x3 1wg = k_encode_mlp_f16x3 (INERF_F16_KERNEL=single).  Entries are rms / max absolute error.")
cases = (("1/f calibrated (the golden fixtures' weights), synthetic rays", synthetic,
          lambda r: (oracle.calibrated_lcg_weights("object", 0, 30, r[:128])[0], oracle.calibrated_lcg_weights("object", 0, 31, r[:128])[0])),
         ("default init, density head calibrated (bench.py's network), every 1251st ray of the 800x800 chair frame", chair,
          lambda r: (cal.calibrated_default_init("object", 0, 0, r), cal.calibrated_default_init("object", 0, 1, r))),
         ("default init, plain seeds 0/1 (sigma < 0 everywhere: acc = 0, every map is exactly the background)", synthetic,
          lambda r: (oracle.make_state_dict("object", seed=0), oracle.make_state_dict("object", seed=1))))
for name, rays, mk in cases:
    sd_c, sd_f = mk(rays)
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
    for prec, form, tag in ((_capi.PREC_F32, None, "hip f32"), (_capi.PREC_F16X3, None, "hip f16x3 2wg"), (_capi.PREC_F16X3, "single", "hip f16x3 1wg")):
        if form:
            os.environ["INERF_F16_KERNEL"] = form
        else:
            os.environ.pop("INERF_F16_KERNEL", None)
        desc = _capi.net_desc(0, 0, 10, 4, 1.0, prec)
        res[tag] = {k: v.cpu() for k, v in kernels.render_rays_fused(
            desc, packing.pack_state_dict(desc, sd_c).to(dev), packing.pack_state_dict(desc, sd_f).to(dev), rays.to(dev), 64, 128,
            t_vals.to(dev), u.to(dev), white_bkgd=True, want_stages=True, want_raw_coarse=True, want_raw_fine=True).items()}
        torch.cuda.synchronize()
    os.environ.pop("INERF_F16_KERNEL", None)
    for k in ("raw_coarse", "weights_coarse", "rgb_coarse", "z_samples", "rgb_fine", "acc_fine", "depth_fine"):
        print(f"{k:15s} vs fp64 (rms / max):  " + "   ".join(f"{tag}: {rms(r[k], w64[k])[0]:.2e} / {rms(r[k], w64[k])[1]:.2e}" for tag, r in res.items()))
    # raw at the REFERENCE's fine depths (the end-to-end raw_fine sits behind sample_pdf): the kernels alone
    z_f = w32["z_fine"]
    with torch.no_grad():
        pts32 = rays[:, None, 0:3] + rays[:, None, 3:6] * z_f[:, :, None]            # the fp32 sample positions ...
        raw64 = oracle.query_network({k: v.double() for k, v in sd_f.items()}, pts32.double(), rays[:, 8:11].double(), cfg)   # ... in fp64
        raw32 = oracle.query_network(sd_f, pts32, rays[:, 8:11], cfg)
    line = f"fine network at the reference's fine depths, raw vs fp64 of the same positions:  oracle fp32: {rms(raw32, raw64)[0]:.2e} / {rms(raw32, raw64)[1]:.2e}"
    for prec, form, tag in ((_capi.PREC_F32, None, "hip f32"), (_capi.PREC_F16X3, None, "hip f16x3 2wg"), (_capi.PREC_F16X3, "single", "hip f16x3 1wg")):
        if form:
            os.environ["INERF_F16_KERNEL"] = form
        else:
            os.environ.pop("INERF_F16_KERNEL", None)
        desc = _capi.net_desc(0, 0, 10, 4, 1.0, prec)
        raw = kernels.encode_mlp(desc, packing.pack_state_dict(desc, sd_f).to(dev), rays.to(dev), z_f.to(dev)).cpu()
        line += f"   {tag}: {rms(raw, raw64)[0]:.2e} / {rms(raw, raw64)[1]:.2e}"
    os.environ.pop("INERF_F16_KERNEL", None)
    print(line)
