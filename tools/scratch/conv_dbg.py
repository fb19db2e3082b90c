import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_conv as T
from oracle import conv_vae_oracle as CO
from vae_training_amd.conv_vae import ConvVAE
size, widths, L, B, tdv = 64, (32, 64, 128, 256), 32, 2, True
cfg = CO.ConvConfig(size, widths, L, -1.5, tdv)
p, x, z1, z2 = T._conv_problem(cfg, B)
loss, g = CO.loss_and_grad(cfg, p, x, z1, z2)
eloss, eg = T._bf16_emulation_grads(cfg, p, x, z1, z2)
net = ConvVAE(B, size, widths, L, -1.5, tdv)
params, grads = net.new_flat(), net.new_flat()
for name in net.leaves:
    net.view(params, name).copy_(T._dev(p[name]))
out4 = net.loss_and_grad(params, grads, T._dev(x), T._dev(z1), T._dev(z2)).cpu().numpy()
print(out4[0], eloss, loss)
for name in net.leaves:
    got = net.view(grads, name).cpu().numpy().astype(np.float64)
    emu, want = eg[name], g[name]
    print(f"{name:28s} vs emu {np.max(np.abs(got-emu))/(np.max(np.abs(emu))+1e-30):.2e}  vs f64 rms {np.sqrt(np.mean((got-want)**2))/(np.sqrt(np.mean(want**2))+1e-30):.2e}  emu vs f64 {np.sqrt(np.mean((emu-want)**2))/(np.sqrt(np.mean(want**2))+1e-30):.2e}")


import torch.nn.functional as F
import vae_training_amd.conv_vae as CV
rb = lambda t: t.to(torch.bfloat16).to(torch.float64)
ident = lambda t: t
o_f, o_t, o_w = CV.conv2d_forward, CV.conv2d_transpose_forward, CV.conv2d_weight_grad
def rel(a, b): return float((a - b).abs().max() / (b.abs().max() + 1e-30))
def f(x, w, bias=None, relu=False, mask=None, out=None):
    y = o_f(x, w, bias, relu, mask, out)
    thin = x.shape[3] == 1 and 256 % w.shape[3] == 0
    r = ident if thin else rb
    ref = F.conv2d(r(x.double().cpu().permute(0,3,1,2)), r(w.double().cpu().permute(3,2,0,1)), None if bias is None else bias.double().cpu(), stride=2, padding=1).permute(0,2,3,1)
    if relu: ref = torch.relu(ref)
    if mask is not None: ref = ref * (mask.cpu() > 0)
    print("fwd ", tuple(x.shape), tuple(w.shape), "thin" if thin else "", rel(y.double().cpu(), ref))
    return y
def t(y, w, bias=None, relu=False, mask=None):
    out = o_t(y, w, bias, relu, mask)
    thin = w.shape[2] == 1 and w.shape[3] % 4 == 0
    r = ident if thin else rb
    ref = F.conv_transpose2d(r(y.double().cpu().permute(0,3,1,2)), r(w.double().cpu().permute(3,2,0,1)), None if bias is None else bias.double().cpu(), stride=2, padding=1).permute(0,2,3,1)
    if relu: ref = torch.relu(ref)
    if mask is not None: ref = ref * (mask.cpu() > 0)
    print("convt", tuple(y.shape), tuple(w.shape), "thin" if thin else "", rel(out.double().cpu(), ref))
    return out
def wg(x, dy, want_bias=True, dw=None, db=None):
    dw, db = o_w(x, dy, want_bias, dw, db)
    thin = x.shape[3] == 1 and 256 % dy.shape[3] == 0
    r = ident if thin else rb
    ref = torch.nn.grad.conv2d_weight(r(x.double().cpu().permute(0,3,1,2)), (dy.shape[3], x.shape[3], 4, 4), r(dy.double().cpu().permute(0,3,1,2)), stride=2, padding=1).permute(2,3,1,0)
    print("wgrad", tuple(x.shape), tuple(dy.shape), "thin" if thin else "", rel(dw.double().cpu().reshape(ref.shape), ref))
    return dw, db
CV.conv2d_forward, CV.conv2d_transpose_forward, CV.conv2d_weight_grad = f, t, wg
net.loss_and_grad(params, grads, T._dev(x), T._dev(z1), T._dev(z2))
