#!/usr/bin/env python3
"""Probe (round 6): how far do two pretraining epochs from the SAME state, batches and mask seeds drift apart - eager against eager, graph
against graph, eager against graph (subreg_hip.pretrain.train, f32, 8-image batches of 32x32; argv[1] = steps)?  The float atomics of the
f32 dW kernels leave last bits run-dependent; chained steps carry that across LeakyReLU sides and MaxPool argmaxes.  Measured: eager
against eager 8e-4 (3 steps) / 7e-4 (4) / 2e-2 (7) on layer1.0.conv1.weight, 3e-7 / 9e-6 / 2e-4 on layer4.1.conv3.weight; eager against
graph no larger.  (tests/test_hip_train.py::test_pretrain_epoch_with_the_step_as_one_hipgraph_equals_the_eager_epoch gates on that.)"""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "subspace-reg_amd"), os.path.join(REPO, "tests")]
import numpy as np, torch
from types import SimpleNamespace
from subreg_hip import synthetic as syn, pretrain as pt
from subreg_hip.train import SGD
from subreg_hip.resnet_language import create_model

def plain_net():
    from test_hip_loop import make_opt
    net = create_model("resnet18", 60, make_opt(hip_dtype="f32", no_dropblock=True))
    sd = syn.make_state_dict(71)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return net.cuda()

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 7
batches = [(torch.from_numpy(syn.make_images(500 + i, 8, 32)), torch.from_numpy(np.random.RandomState(600 + i).randint(0, 60, 8))) for i in range(nb)]
res = []
for mode in ("eager", "eager", "graph", "graph"):
    net = plain_net().train()
    opt = SimpleNamespace(print_freq=1000, hip_graph=(mode == "graph"), label_pull=None)
    sgd = SGD(net.parameters(), lr=0.002, momentum=0.9, weight_decay=5e-4)
    torch.manual_seed(77)
    acc, loss = pt.train(1, batches, net, None, sgd, opt, log=lambda *_a: None)
    torch.cuda.synchronize()
    res.append((mode, acc, loss, {k: v.detach().clone().double() for k, v in net.state_dict().items()}, torch.get_rng_state().clone()))
def rel(a, b, k): return float((a[k] - b[k]).norm() / a[k].norm().clamp_min(1e-30))
for i, j in ((0, 1), (2, 3), (0, 2)):
    a, b = res[i], res[j]
    print("%s vs %s: loss %.6f / %.6f, rng equal %s, conv1 %.2e, layer4.1.conv3 %.2e, classifier %.2e" % (a[0], b[0], a[2], b[2], torch.equal(a[4], b[4]),
          rel(a[3], b[3], "layer1.0.conv1.weight"), rel(a[3], b[3], "layer4.1.conv3.weight"), rel(a[3], b[3], "classifier.weight")))
