"""Checkpoint files on both sides of the hot path, in the reference's own on-disk format.

Writer: /root/reference/train_supervised.py:181-202 (`ckpt_epoch_N.pth` = {'epoch', 'model'} and `<model>_last.pth` =
{'opt', 'model'}, each + {'training_classes', 'label2human'} when `opt.continual`).  Reader: eval_incremental.py:86-110
(`torch.load(opt.model_path)`, the linear-bias rule, `create_model` + `load_state_dict`, `ckpt['training_classes']`) and
eval/language_eval.py:139-142.  `model` is the 133-key state_dict of models/resnet_language.py::ResNet, which
`subreg_hip.resnet_language.ResNet` reproduces key for key, so files written by either side load in the other.
"""
import os
from collections import OrderedDict

import torch

from .resnet_language import create_model


def _cpu_state_dict(model):
    sd = model.module.state_dict() if hasattr(model, "module") else model.state_dict()      # train_supervised.py:185,196
    return OrderedDict((k, v.detach().to("cpu").clone()) for k, v in sd.items())


def save_checkpoint(path, model, opt=None, epoch=None, training_classes=None, label2human=None):
    """Write what train_supervised.py writes.  `epoch` given -> the periodic file ({'epoch','model'}), else the last-model
    file ({'opt','model'}).  `training_classes` (dataset.basec_map) / `label2human` go in when given (opt.continual)."""
    state = {"epoch": epoch} if epoch is not None else {"opt": opt}
    state["model"] = _cpu_state_dict(model)
    if training_classes is not None:
        state["training_classes"] = dict(training_classes)
    if label2human is not None:
        state["label2human"] = list(label2human)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(state, path)
    return path


def load_checkpoint(path, map_location="cpu"):
    """eval_incremental.py:86: the whole dict.  The file pickles `opt` (an argparse.Namespace), so this is a full unpickle
    like the reference's `torch.load`: only open checkpoints you trust."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    if "model" not in ckpt:
        raise KeyError("checkpoint %s has no 'model' entry (keys: %s)" % (path, sorted(ckpt.keys())))
    return ckpt


def infer_linear_bias(ckpt):
    """eval_incremental.py:96-103: the classifier has a bias iff the checkpoint holds one (None there is an error)."""
    if "classifier.bias" in ckpt["model"].keys():
        if ckpt["model"]["classifier.bias"] is None:
            raise ValueError()
        return True
    return False


def base_class_maps(ckpt):
    """eval_incremental.py:118-123: (basec_map, basec_map_rev) of a continual checkpoint."""
    basec_map = ckpt["training_classes"]
    return basec_map, {v: k for k, v in basec_map.items()}


def model_from_checkpoint(ckpt, name, n_cls, opt, vocab=None, dataset="miniImageNet"):
    """eval_incremental.py:96-108: set opt.linear_bias from the file, build the model, load the weights (strict)."""
    opt.linear_bias = infer_linear_bias(ckpt)
    model = create_model(name, n_cls, opt, vocab=vocab, dataset=dataset)
    model.load_state_dict(ckpt["model"])
    return model
