"""Debug aid (GPU box): plain-PyTorch fp32 restatement of the train-mode forward with the same injected masks, autograd
gradients per parameter AND per block output, compared with the HIP backward."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "subspace-reg_amd"), os.path.join(REPO, "tests")]
from test_hip_train import _train_net
from oracle.resnet_ref import MaskSource, dropblock_gamma
from subreg_hip import synthetic as syn
torch.backends.cudnn.enabled = False
hw, B = int(sys.argv[1]) if len(sys.argv) > 1 else 84, int(sys.argv[2]) if len(sys.argv) > 2 else 6
net = _train_net("f32")
x = torch.from_numpy(syn.make_images(72, B, hw)).cuda()
y = torch.from_numpy(np.random.RandomState(73).randint(0, 60, B)).cuda()
sd = {k: v.detach().clone().double().requires_grad_(v.dtype.is_floating_point) for k, v in net.state_dict().items()}
masks = MaskSource(74)
outs = []
t2s = []
t1s = []
def bn(t, p):
    return F.batch_norm(t, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, 1e-5)
cur = x.double()
for name, cin, cout, stride, ds, db in syn.backbone_blocks():
    o = F.leaky_relu(bn(F.conv2d(cur, sd[name + ".conv1.weight"], padding=1), name + ".bn1"), 0.1)
    o.retain_grad(); t1s.append(o)
    o = F.leaky_relu(bn(F.conv2d(o, sd[name + ".conv2.weight"], padding=1), name + ".bn2"), 0.1)
    o.retain_grad(); t2s.append(o)
    o = bn(F.conv2d(o, sd[name + ".conv3.weight"], padding=1), name + ".bn3")
    r = bn(F.conv2d(cur, sd[name + ".downsample.0.weight"]), name + ".downsample.1") if ds else cur
    o = F.leaky_relu(o + r, 0.1)
    if stride == 2: o = F.max_pool2d(o, 2)
    Bq, C, H, W = o.shape
    if db:
        s = masks.bernoulli((Bq, C, H, W), dropblock_gamma(1, H, 1)); bm = 1.0 - s
        o = o * torch.from_numpy(bm).cuda().double() * (bm.size / bm.sum())
    else:
        o = o * torch.from_numpy(masks.dropout_keep((Bq, C, H, W), 0.1)).cuda().double() / 0.9
    o.retain_grad(); outs.append(o); cur = o
feat = cur.mean(dim=(2, 3))
loss = F.cross_entropy(feat @ sd["classifier.weight"].t(), y)
loss.backward()
net.train()
import ctypes as C
logits = net(x)
stash = net.hip_backbone()._train_stash
dumps = [torch.zeros_like(o, dtype=torch.float32).permute(0, 2, 3, 1).contiguous() for o in outs]
arr = (C.c_void_p * len(dumps))(*[d.data_ptr() for d in dumps])
stash.desc.grad_out_dump = C.cast(arr, C.POINTER(C.c_void_p))
l2 = torch.nn.CrossEntropyLoss()(logits, y); l2.backward()
torch.cuda.synchronize()
for i, (o, d) in enumerate(zip(outs, dumps)):
    w = o.grad.permute(0, 2, 3, 1).float()
    e = (d - w).abs()
    idx = torch.nonzero(e > 1e-3 * w.abs().max())
    print("gout block %d: max rel %.2e, n bad %d of %d%s" % (i, (e.max() / w.abs().max()).item(), idx.shape[0], e.numel(),
          "" if idx.shape[0] == 0 else "  bad (b,h,w) %s ch %s" % (sorted(set((int(a), int(b), int(c)) for a, b, c, _ in idx[:2000].tolist()))[:10], sorted(set(int(q[3]) for q in idx[:2000].tolist()))[:10])))
print("loss torch64 %.7f hip %.7f" % (loss.item(), l2.item()))
rows = []
for n, p in net.named_parameters():
    w = sd[n].grad.float().cpu().numpy(); g = p.grad.cpu().numpy()
    rows.append((n, float(np.abs(g - w).max() / max(np.abs(w).max(), 1e-20)), float(np.linalg.norm(g - w) / max(np.linalg.norm(w), 1e-20))))
blk = {}
for n, e, e2 in rows: blk.setdefault(".".join(n.split(".")[:2]), []).append(e2)
print({k: "%.1e" % max(v) for k, v in sorted(blk.items())})
rows.sort(key=lambda r: -r[2])
for r in rows[:8]: print("  %-36s maxrel %.3e  l2rel %.3e" % r)

# ---- replay block 3 (layer3.1) tail -> bn3 bwd -> dX(conv3) on the stash with the exact gout, compare d(act2)
from subreg_hip import _lib
lib = _lib.load(); hb = net.hip_backbone(); nm = stash.named; bi = 3
Bq, C, H, W = outs[bi].shape
sp = _lib.stream_ptr
gout = outs[bi].grad.permute(0, 2, 3, 1).float().contiguous()
dv, dr, dt = [torch.zeros(Bq * H * W * C, device="cuda") for _ in range(3)]
keep = hb._keep[bi]
xin = nm[(bi - 1, "out")]
_lib.check(lib.subreg_block_tail_bwd(_lib.ptr(gout), _lib.ptr(keep), hb.mask_scale(bi), None, _lib.ptr(nm[(bi, "conv3", "raw")]),
           _lib.ptr(nm[(bi, "conv3", "bscale")]), _lib.ptr(nm[(bi, "conv3", "bshift")]), _lib.ptr(xin), None, None, _lib.ptr(dv), Bq, H, W, C, 0, 0, sp()))
part = torch.zeros(lib.subreg_bn_bwd_slices(Bq * H * W) * C * 2, dtype=torch.float64, device="cuda")
dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
gam = dict(net.named_parameters())["layer3.1.bn3.weight"]
_lib.check(lib.subreg_bn_bwd(_lib.ptr(dv), None, _lib.ptr(nm[(bi, "conv3", "raw")]), _lib.ptr(nm[(bi, "conv3", "mean")]), _lib.ptr(nm[(bi, "conv3", "invstd")]),
           _lib.ptr(gam), _lib.ptr(part), _lib.ptr(dg), _lib.ptr(db), _lib.ptr(dr), Bq * H * W, C, 0, sp()))
zero = torch.zeros(C, device="cuda")
_lib.check(lib.subreg_conv_fwd(_lib.ptr(dr), _lib.ptr(nm[(bi, "conv3", "w_dgrad")]), _lib.ptr(dt), None, _lib.ptr(zero), None, None, None, None, 0,
           Bq, H, W, C, C, 3, 0, 0, sp()))
torch.cuda.synchronize()
want = t2s[bi].grad.permute(0, 2, 3, 1).float().reshape(-1)
e = (dt - want).abs()
print("replayed d(act2) of layer3.1: max rel %.2e, n bad %d" % ((e.max() / want.abs().max()).item(), int((e > 1e-3 * want.abs().max()).sum())))
bad = torch.nonzero(e.reshape(Bq * H * W, C) > 1e-3 * want.abs().max())
if bad.shape[0]:
    rows = sorted(set(int(r) for r, _ in bad.tolist())); chs = sorted(set(int(c) for _, c in bad.tolist()))
    print("  bad rows", rows[:40], "n rows", len(rows), " channels", chs[:20], "n ch", len(chs))
# same conv through torch with the original weight, to make sure the packed dgrad weights are what they should be
w3 = dict(net.named_parameters())["layer3.1.conv3.weight"].detach()
ref = F.conv_transpose2d(dr.reshape(Bq, H, W, C).permute(0, 3, 1, 2), w3, padding=1).permute(0, 2, 3, 1).reshape(-1)
print("HIP dX vs torch conv_transpose2d on the same dr: max rel %.2e" % ((dt - ref).abs().max() / ref.abs().max()).item())

# ---- continue the replay: bn2 backward (+lrelu'), dX(conv2), bn1 backward
def relerr(a, b): return ((a - b).abs().max() / b.abs().max()).item()
P = dict(net.named_parameters())
dg2, db2, dr2_ = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(Bq * H * W * C, device="cuda")
_lib.check(lib.subreg_bn_bwd(_lib.ptr(dt), _lib.ptr(nm[(bi, "conv2", "act")]), _lib.ptr(nm[(bi, "conv2", "raw")]), _lib.ptr(nm[(bi, "conv2", "mean")]),
           _lib.ptr(nm[(bi, "conv2", "invstd")]), _lib.ptr(P["layer3.1.bn2.weight"]), _lib.ptr(part), _lib.ptr(dg2), _lib.ptr(db2), _lib.ptr(dr2_), Bq * H * W, C, 0, sp()))
torch.cuda.synchronize()
print("replay bn2: dgamma rel %.2e dbeta rel %.2e   | real-run grads: dgamma rel %.2e dbeta rel %.2e" % (
      relerr(dg2, sd["layer3.1.bn2.weight"].grad.float()), relerr(db2, sd["layer3.1.bn2.bias"].grad.float()),
      relerr(P["layer3.1.bn2.weight"].grad, sd["layer3.1.bn2.weight"].grad.float()), relerr(P["layer3.1.bn2.bias"].grad, sd["layer3.1.bn2.bias"].grad.float())))
# is the stashed act2 / raw2 what torch computed?
a2 = t2s[bi].detach().permute(0, 2, 3, 1).float().reshape(-1)
print("stash act2 vs torch: %.2e" % relerr(nm[(bi, "conv2", "act")], a2))
wb = sd["layer3.1.bn2.bias"].grad.float()
eb = (db2 - wb).abs() / wb.abs().max()
print("dbeta err by channel: bad idx", torch.nonzero(eb > 1e-3).flatten().tolist()[:40], "n", int((eb > 1e-3).sum()))
# brute-force the same sums on the stash tensors with torch ops
A = nm[(bi, "conv2", "act")].reshape(-1, C); G = dt.reshape(-1, C)
slope = torch.where(A > 0, torch.ones_like(A), torch.full_like(A, 0.1))
print("torch-on-stash dbeta vs autograd: %.2e ; kernel vs torch-on-stash: %.2e" % (relerr((G * slope).double().sum(0).float(), wb), relerr(db2, (G * slope).double().sum(0).float())))
y2 = nm[(bi, "conv2", "raw")].reshape(-1, C) * nm[(bi, "conv2", "bscale")] + nm[(bi, "conv2", "bshift")]
print("sign(act) != sign(y): %d ; act==0: %d ; |y|<1e-6: %d" % (int(((A > 0) != (y2 > 0)).sum()), int((A == 0).sum()), int((y2.abs() < 1e-6).sum())))
