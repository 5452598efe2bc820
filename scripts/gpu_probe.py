#!/usr/bin/env python3
"""First-contact probe for the GPU box: stage-by-stage errors vs the oracle + rough timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import emap_amd
from emap_amd import synthetic, _lib
from oracle import emap_oracle as O
from conftest import load_golden, t, net_state

dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0))
torch.manual_seed(0)

def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

def mk(name, scale=1.0, precision="bf16x3"):
    kw, state = net_state(name)
    net = emap_amd.UDFNetwork(scale=scale, precision=precision, **kw)
    net.load_state_dict(state)
    return net.to(dev), state, O.UDFConfig(d_in=3, d_out=1, d_hidden=kw["d_hidden"], n_layers=kw["n_layers"], skip_in=(4,), multires=kw["multires"], scale=scale)

g2 = load_golden("g2_mlp")
x = t(g2["x"])
for name in ["d8w256L10", "d8w256L6", "d4w128L10", "d8w256L10_init"]:
    for prec in ["f16x3", "bf16x3", "f16", "bf16"]:
        net, state, cfg = mk(name, precision=prec)
        with torch.no_grad():
            u, g = net.hip_udf(x.to(dev), with_grad=True)
            u2, _ = net.hip_udf(x.to(dev), with_grad=False)
        torch.cuda.synchronize()
        ur, gr = t(g2[f"{name}.udf"]), t(g2[f"{name}.grad"]).reshape(-1, 3)
        print(f"{name:16s} {prec:7s} udf(grad-kernel) {rel(u, ur):.2e}  udf(value-kernel) {rel(u2, ur):.2e}  grad {rel(g, gr):.2e}")

# PE
g1 = load_golden("g1_pe")
fn, d = emap_amd.get_embedder(10)
print("embed L10", rel(fn(t(g1["x"]).to(dev)), t(g1["pe_L10"])))

# sample_pdf
g3 = load_golden("g3_sample_pdf")
L = _lib.lib()
for m in (10, 16):
    b, w = t(g3["bins"]).to(dev), t(g3["weights"]).to(dev)
    s = torch.empty(b.shape[0], m, device=dev); inds = torch.empty(b.shape[0], m, device=dev, dtype=torch.int64)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(L.emap_sample_pdf(_lib.ptr(b), _lib.ptr(w), b.shape[0], b.shape[1], m, _lib.ptr(s), _lib.ptr(inds), _lib.ptr(err), _lib.stream_ptr()))
    torch.cuda.synchronize()
    print(f"sample_pdf m={m}: inds mismatches {(inds.cpu() != t(g3[f'inds_m{m}'])).sum().item()}  samples bit-mismatches {(s.cpu() != t(g3[f'samples_m{m}'])).sum().item()} maxdiff {(s.cpu()-t(g3[f'samples_m{m}'])).abs().max().item():.2e} err {err.item()}")

# upsample + merge
g4 = load_golden("g4_upsample_step")
ro, rd = t(g4["rays_o"]).to(dev), t(g4["rays_d"]).to(dev)
z, udf = t(g4["z_vals"]).to(dev), t(g4["udf"]).to(dev)
sd = torch.tensor([float(g4["sample_dist"])], device=dev)
for i in range(2):
    inv_s, beta, gamma = [float(v) for v in g4[f"step{i}.params"]]
    N, n = z.shape
    zn = torch.empty(N, 16, device=dev); inds = torch.empty(N, 16, device=dev, dtype=torch.int64)
    _lib.check(L.emap_upsample_step(_lib.ptr(ro), _lib.ptr(rd), _lib.ptr(z), _lib.ptr(udf), N, n, 16, _lib.ptr(sd), inv_s, beta, gamma, _lib.ptr(zn), _lib.ptr(inds), None, _lib.stream_ptr()))
    zref = t(g4[f"step{i}.z_new"])
    print(f"upsample step{i}: inds mismatches {(inds.cpu() != t(g4[f'step{i}.inds'])).sum().item()}/{inds.numel()}  z_new maxdiff {(zn.cpu()-zref).abs().max().item():.2e}")
    zo = torch.empty(N, n + 16, device=dev); uo = torch.empty(N, n + 16, device=dev); perm = torch.empty(N, n + 16, device=dev, dtype=torch.int64)
    zn_ref = zref.to(dev)
    un = t(g4[f"step{i}.udf_out"]).to(dev)  # placeholder for new udf: derive from golden gather
    idx = t(g4[f"step{i}.sort_index"]).to(dev)
    cat_u = torch.empty(N, n + 16, device=dev); cat_u.scatter_(1, idx, un)
    _lib.check(L.emap_merge_sorted(_lib.ptr(z), _lib.ptr(zn_ref), _lib.ptr(udf), _lib.ptr(cat_u[:, n:].contiguous()), N, n, 16, _lib.ptr(zo), _lib.ptr(uo), _lib.ptr(perm), _lib.stream_ptr()))
    print(f"merge step{i}: perm mismatches {(perm.cpu() != t(g4[f'step{i}.sort_index'])).sum().item()}  z mism {(zo.cpu() != t(g4[f'step{i}.z_out'])).sum().item()} udf mism {(uo.cpu() != t(g4[f'step{i}.udf_out'])).sum().item()}")
    z, udf = t(g4[f"step{i}.z_out"]).to(dev), t(g4[f"step{i}.udf_out"]).to(dev)

# full render vs goldens
G5 = {"c64_50_5": "d8w256L10", "c64_64_4": "d8w256L10", "c32_32_4_small": "d4w128L10", "c64_64_4_L6": "d8w256L6"}
for case, name in G5.items():
    g = load_golden("g5_render_" + case)
    ns, ni, steps = [int(v) for v in g["cfg"]]
    for prec in ["f16x3", "bf16x3", "f16", "bf16"]:
        net, state, cfg = mk(name, precision=prec)
        devn = emap_amd.SingleVarianceNetwork(0.3).to(dev); bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(dev)
        r = emap_amd.UDFRendererBlending(None, net, devn, bet, ns, ni, 0, steps, 1.0, device=dev)
        a = [t(g[k]).to(dev) for k in ("rays_o", "rays_d", "near", "far", "depth_scale")]
        with torch.no_grad():
            out = r.render(*a, cos_anneal_ratio=1.0, perturb_overwrite=0, flip_saturation=0.9)
        torch.cuda.synchronize()
        zref = t(g[f"z_after_step{steps-1}"])
        zm = ((out["z_vals"].cpu() - zref).abs().max(dim=1)[0] > 1e-5).float().mean().item()
        s = " ".join(f"{k}:{rel(out[k], t(g['out.'+k]).reshape(out[k].shape)):.1e}" for k in ["edge", "depth", "weights", "normals", "gradient_error", "gradient_error_near_surface", "udf", "gradients", "mid_z_vals", "dists", "inside_sphere", "gradient_mag", "variance", "beta", "gamma"])
        print(f"render {case} {prec}: rays with moved samples {zm:.3f} | {s} | err {r.error_flags()}")

print("PROBE DONE")
