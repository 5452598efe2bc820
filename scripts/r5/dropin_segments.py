#!/usr/bin/env python3
"""Host time of the segments of the patched drop-in training step (bench._RunnerLoop under dropin.train_wrapper), perf_counter stamps."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from emap_amd import dropin

dev = torch.device("cuda", 0)
loop = bench._RunnerLoop(dev, "f16x3", 512, False)
T = {}
def seg(name, t0):
    t1 = time.perf_counter(); T.setdefault(name, []).append(t1 - t0); return t1

def step(self=loop):
    t = time.perf_counter()
    it = self.iter_step
    for g_, base in zip(self.optimizer.param_groups, (self.learning_rate_geo, self.learning_rate, self.learning_rate)):
        g_["lr"] = base * min(1.0, (it + 1) / 1000.0)
    smp = self.sampler.gen_random_rays_patches_at(self.image_perm[it % len(self.image_perm)], self.batch_size, importance_sample=True)
    data = smp["rays"]
    rays_o, rays_d, true_edge = data["rays_o"], data["rays_v"], data["edge"]
    mask = torch.ones_like(true_edge).float()
    mask_sum = mask.sum() + 1e-5
    t = seg("1 lr + sampler + mask", t)
    EV.append({"A": ev()})
    render_out = self.renderer.render(rays_o, rays_d, self.near, self.far, depth_scale=smp["depth_scale"], flip_saturation=0.9, pose=None,
                                      fx=None, fy=None, img_index=None, cos_anneal_ratio=1.0)
    EV[-1]["B"] = ev()
    t = seg("2 render()", t)
    udf, edge = render_out["udf"], render_out["edge"]
    variance, beta = render_out["variance"], render_out["beta"]
    gradient_error, gradient_error_near_surface = render_out["gradient_error"], render_out["gradient_error_near_surface"]
    udf_min = udf.min(dim=1)[0][mask[:, 0] > 0.5].mean()
    edge_loss = self.edge_loss_func(edge, true_edge) * self.edge_weight
    psnr = 20.0 * torch.log10(1.0 / (((edge - true_edge) ** 2 * mask).sum() / mask_sum).sqrt())
    gradient_error_loss = gradient_error
    t = seg("3 udf_min, edge_loss, psnr", t)
    if (variance.mean() < 2 * beta.item() and variance.mean() < 0.01 and self.beta_flag and self.variance_network_fine.variance.requires_grad):
        self.beta_network.set_beta_trainable(); self.beta_flag = False
    if self.variance_network_fine.variance.requires_grad is False and self.iter_step > 20000:
        self.variance_network_fine.set_trainable()
    loss = edge_loss + gradient_error_near_surface * self.igr_ns_weight + gradient_error_loss * self.igr_weight
    t = seg("4 variance/beta checks + loss", t)
    self.last_loss = "PSNR: {:.2f}, Loss: {:.2f}".format(psnr, loss.item())
    t = seg("5 format(psnr), loss.item()  [WAIT for the forward]", t)
    self.optimizer.zero_grad()
    t = seg("6 zero_grad", t)
    loss.backward()
    if BK.get("_rb_start"):
        T.setdefault("7a   backward() call -> RenderFn.backward entered (engine thread, 6 small nodes)", []).append(BK["_rb_start"][-1] - t)
        T.setdefault("7c   RenderFn.backward returned -> backward() returns", []).append(time.perf_counter() - BK["_rb_start"][-1] - BK["RenderFn.backward total"][-1])
    t = seg("7 loss.backward()", t)
    self.optimizer.step()
    EV[-1]["E"] = ev()
    t = seg("8 optimizer.step()", t)
    self.iter_step += 1
    w = self.writer
    w.add_scalar("Loss/loss", loss, self.iter_step); w.add_scalar("Loss/edge_loss", edge_loss, self.iter_step)
    w.add_scalar("Loss/gradient_error_loss", gradient_error_loss * self.igr_weight, self.iter_step)
    w.add_scalar("Loss/gradient_error_near_surface", gradient_error_near_surface * self.igr_ns_weight, self.iter_step)
    w.add_scalar("Sta/variance", variance.mean(), self.iter_step); w.add_scalar("Sta/beta", beta.item(), self.iter_step)
    w.add_scalar("Sta/psnr", psnr, self.iter_step)
    t = seg("9 writer", t)

loop.step = step
# inside loss.backward(): when does the engine reach RenderFn.backward's C call?
import emap_amd.backward as BW
_orig_bi = loop.renderer.backward_into
_orig_rb = BW.RenderFn.backward
BK = {}
EV = []
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
def bi(*a, **k):
    EV[-1]["C"] = ev()
    t0 = time.perf_counter(); r = _orig_bi(*a, **k); t1 = time.perf_counter()
    EV[-1]["D"] = ev()
    BK.setdefault("backward_into (host prep + C call launching the kernels)", []).append(t1 - t0); BK["_bi_end"] = t1; return r
loop.renderer.backward_into = bi
def rb(ctx, *g):
    t0 = time.perf_counter(); BK.setdefault("_rb_start", []).append(t0); r = _orig_rb(ctx, *g); BK.setdefault("RenderFn.backward total", []).append(time.perf_counter() - t0); return r
BW.RenderFn.backward = staticmethod(rb)
_orig_step = step
def step2(self=loop):
    BK["_call"] = None
    return _orig_step(self)
def run(st):
    for _ in range(30): st()
    T.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): st()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 100
patched = "--unpatched" not in sys.argv
train = bench._RunnerLoop.train_udf
if patched:
    train = dropin.train_wrapper(train, sys.modules["bench"])
dt = train(loop, run)
print("patched" if patched else "unpatched", "ms/step %.3f" % (dt * 1e3))
for k in sorted(T):
    v = sorted(T[k]); print("  %-55s median %.3f ms  mean %.3f" % (k, v[len(v)//2] * 1e3, sum(v) / len(v) * 1e3))
for k in sorted(BK):
    if not k.startswith("_"):
        v = sorted(BK[k][-100:]); print("  7b   %-50s median %.3f ms" % (k, v[len(v)//2] * 1e3))
torch.cuda.synchronize()
E = EV[-100:]
def med(xs): xs = sorted(xs); return xs[len(xs)//2]
print("  GPU timeline (HIP events on the stream, median over 100 steps):")
print("    A->B  forward render kernels                          %.3f ms" % med([e["A"].elapsed_time(e["B"]) for e in E]))
print("    B->C  forward done -> backward kernels launched        %.3f ms   (the runner's loss kernels + GPU IDLE while the host gets there)" % med([e["B"].elapsed_time(e["C"]) for e in E]))
print("    C->D  backward kernels                                 %.3f ms" % med([e["C"].elapsed_time(e["D"]) for e in E]))
print("    D->E  gradient hand-over + Adam                        %.3f ms" % med([e["D"].elapsed_time(e["E"]) for e in E]))
print("    E->A' Adam done -> next forward launched               %.3f ms   (GPU IDLE unless the host was ahead)" % med([E[i]["E"].elapsed_time(E[i+1]["A"]) for i in range(len(E)-1)]))
print("  sum of medians %.3f ms" % sum(sorted(v)[len(v)//2] * 1e3 for v in T.values()))
