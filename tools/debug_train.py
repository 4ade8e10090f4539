import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "subspace-reg_amd"), os.path.join(REPO, "tests")]
from test_hip_train import _train_net
from subreg_hip import synthetic as syn
g = np.load(os.path.join(REPO, "tests/golden/train_step.npz"))
for hw, dtype in ((84, "f32"), (32, "bf16"), (84, "bf16")):
    key = "hw%d" % hw
    net = _train_net(dtype)
    x = torch.from_numpy(syn.make_images(72, int(g[key + ".B"]), hw)).cuda()
    y = torch.from_numpy(g[key + ".labels"]).cuda()
    net.train()
    loss = torch.nn.CrossEntropyLoss()(net(x), y)
    loss.backward()
    print(hw, dtype, "loss", loss.item(), float(g[key + ".loss"]))
    grads = {n: p.grad.detach().cpu().numpy() for n, p in net.named_parameters()}
    rows = []
    for k in g.files:
        if k.startswith(key + ".gnorm."):
            name = k[len(key) + 7:]
            rows.append((name, "norm", abs(np.linalg.norm(grads[name].astype(np.float64)) - float(g[k])) / float(g[k])))
        elif k.startswith(key + ".grad."):
            name = k[len(key) + 6:]
            want = g[k]; got = grads[name][:want.shape[0]]
            rows.append((name, "tensor", float(np.abs(got - want).max() / np.abs(want).max())))
    rows.sort(key=lambda r: -r[2])
    for r in rows[:12]: print("   %-36s %-6s rel err %.3e" % r)
    byblock = {}
    for n, kind, e in rows:
        if kind == "norm": byblock.setdefault(n.split(".")[0] + "." + n.split(".")[1], []).append(e)
    print("   max norm err per block:", {k: "%.1e" % max(v) for k, v in sorted(byblock.items())})
