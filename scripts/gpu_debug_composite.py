import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch.nn.functional as F
from conftest import load_golden, t, net_state
import test_gpu_parity as T
from oracle import emap_oracle as O
case = "c64_64_4"
g = load_golden("g5_render_" + case)
ns, ni, steps = [int(v) for v in g["cfg"]]
net, _, _ = T.mk(T.G5[case], "f16x3")
r = T.mk_renderer(net, ns, ni, steps)
z = t(g[f"z_after_step{steps - 1}"])
out = T._render_core_on_z(net, r, g, z, 1.0, 0.9)
ref_w = t(g["out.weights"]); w = out["weights"].cpu()
d = (w - ref_w).abs(); i = int(d.argmax()); ray, s = divmod(i, w.shape[1])
print("worst weights diff", float(d.max()), "at ray", ray, "sample", s, "ours", float(w[ray, s]), "ref", float(ref_w[ray, s]), "max ref", float(ref_w.max()))
# torch (CPU) composite from OUR udf/grad
udf = out["udf"].cpu(); grad = out["gradients"].cpu()
ro, rd, near, far, ds = [t(g[k]) for k in ("rays_o", "rays_d", "near", "far", "depth_scale")]
N, S = z.shape
sd = ((far - near) / ns).mean().item()
dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((N, 1), sd)], -1)
dirs = rd[:, None, :].expand(N, S, 3)
true_cos = (dirs * grad).sum(-1)
inv_s = torch.exp(torch.tensor(0.3) * 10).clip(1e-6, 1e6); beta = torch.exp(torch.tensor(0.5) * 10).clip(0, 20000.); gamma = torch.exp(torch.tensor(0.3) * 10)
raw_occ = O.udf2logistic(udf, beta, 1.0, 1.0)
alpha_occ = 1.0 - torch.exp(-F.relu(raw_occ) * gamma * dists)
vm = (true_cos < 0.01).float(); vm = torch.cat([vm[:, 1:], torch.ones(N, 1)], -1)
vp = torch.cumprod(torch.cat([torch.ones(N, 1), (1.0 - alpha_occ + 0.9 * vm).clip(0, 1) + 1e-7], -1), -1)[:, :-1].clip(0, 1)
ap = O.sdf2alpha(udf, -true_cos.abs(), dists, inv_s, 1.0); am = O.sdf2alpha(-udf, -true_cos.abs(), dists, inv_s, 1.0)
alpha = ap * vp + am * (1 - vp)
wt = alpha * torch.cumprod(torch.cat([torch.ones(N, 1), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
print("kernel vs torch-composite-on-our-udf: weights", float((w - wt).abs().max()), "alpha", float((out["alpha"].cpu() - alpha).abs().max()))
print("torch-composite-on-our-udf vs reference weights", float((wt - ref_w).abs().max()))
print("at worst: udf ours/ref", float(udf[ray, s]), float(t(g["out.udf"])[ray, s]), "true_cos ours", float(true_cos[ray, s]), "ref grad·d", float((dirs * t(g["out.gradients"])).sum(-1)[ray, s]))
print("vis_mask ours vs ref flips:", int(((true_cos < 0.01) != ((dirs * t(g["out.gradients"])).sum(-1) < 0.01)).sum()))
print("vp around", vp[ray, max(0, s - 2):s + 3].tolist(), "alpha", alpha[ray, max(0, s - 2):s + 3].tolist(), "ref alpha? n/a")
print("alpha_occ around", alpha_occ[ray, max(0, s - 3):s + 2].tolist(), "vm", vm[ray, max(0, s - 3):s + 2].tolist(), "tc", true_cos[ray, max(0, s - 3):s + 3].tolist())
