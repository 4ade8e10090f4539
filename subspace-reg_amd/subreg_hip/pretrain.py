"""Pretraining driver around the HIP train step: the routine of /root/reference/train_supervised.py.

  adjust_learning_rate   util.py:45-51          step decay at opt.lr_decay_epochs
  cosine_lr              train_supervised.py:146-157   CosineAnnealingLR(optimizer, opt.epochs, eta_min, -1), stepped BEFORE each epoch
  train                  train_supervised.py:205-268   one epoch: forward, CE (+ label-pull penalty), meters, backward, SGD step
  validate               eval/util.py:185-232          one pass in eval mode: loss, top-1, top-5
  fit                    train_supervised.py:150-202   epochs, periodic + last checkpoint in the reference's format
  GradientSync           replaces nn.DataParallel (:139-140): one process per GPU, every rank takes its share of the batch and
                         back-propagates loss * n_local / n_global; the backbone's flat gradient buffer is SUM-all-reduced
                         stage by stage, overlapped with the backward (RCCL over xGMI), the classifier's at the end
The model is `subreg_hip.resnet_language.ResNet`; its train-mode forward/backward run on the kernels of csrc/backward.hip.
"""
import math
import os
import sys
import time

import numpy as np
import torch

from . import checkpoint as ck
from . import functional as HF
from .train import SGD, Adam, GraphedStep


class AverageMeter(object):
    """eval/util.py:9-24."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def adjust_learning_rate(epoch, opt, optimizer):
    """util.py:45-51: initial LR decayed by decay rate at every passed milestone (strictly greater)."""
    steps = np.sum(epoch > np.asarray(opt.lr_decay_epochs))
    if steps > 0:
        new_lr = opt.learning_rate * (opt.lr_decay_rate ** steps)
        for param_group in optimizer.param_groups:
            param_group["lr"] = new_lr


def cosine_lr(opt, n_steps):
    """LR after `n_steps` calls of CosineAnnealingLR(optimizer, T_max=opt.epochs, eta_min=lr*decay^3).step()
    (train_supervised.py:146-157: the scheduler is stepped at the top of every epoch, so epoch e trains at n_steps=e)."""
    eta_min = opt.learning_rate * (opt.lr_decay_rate ** 3)
    return eta_min + (opt.learning_rate - eta_min) * (1.0 + math.cos(math.pi * n_steps / opt.epochs)) / 2.0


def set_epoch_lr(epoch, opt, optimizer):
    if getattr(opt, "cosine", False):
        for g in optimizer.param_groups:
            g["lr"] = cosine_lr(opt, epoch)
    else:
        adjust_learning_rate(epoch, opt, optimizer)


class GradientSync:
    """Data-parallel gradient reduction, one process per GPU (replaces nn.DataParallel, train_supervised.py:141-142).

    Every rank back-propagates its local mean loss times n_local / n_global (`train` below), so the SUM over ranks is exactly
    the gradient of the global-batch mean loss - also for uneven shards - and nothing is divided afterwards.
    The backbone's gradients are views of one flat buffer; `stage_ready` (hooked into the staged backward,
    train.BackboneTrainFn.backward) starts the all-reduce of a stage's range asynchronously while the backward of the earlier
    blocks is still running (RCCL over xGMI: the collective runs on its own stream behind an event of the compute stream).
    `finish` reduces every gradient storage not covered that way (the classifier) and waits for all collectives."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.calls = 0
        self.pending, self.covered = [], []

    def stage_ready(self, flat_slice):
        if self.world == 1:
            return
        self.pending.append(self.dist.all_reduce(flat_slice, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.covered.append((flat_slice.data_ptr(), flat_slice.data_ptr() + flat_slice.numel() * flat_slice.element_size()))
        self.calls += 1

    def finish(self, params):
        if self.world == 1:
            return
        todo, partly = {}, set()
        for p in params:
            if p.grad is None:
                continue
            a = p.grad.data_ptr()
            base = p.grad._base if p.grad._base is not None else p.grad
            if any(lo <= a < hi for lo, hi in self.covered):
                partly.add(id(base))                           # part of a stage range already in flight
                continue
            todo.setdefault(id(base), (base, []))[1].append(p.grad)
        for key, (base, views) in todo.items():
            # a storage none of whose views went out with a stage: ONE collective for the whole storage; otherwise view by view
            for t in ([base] if key not in partly else [v.contiguous() if not v.is_contiguous() else v for v in views]):
                self.pending.append(self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
                self.calls += 1
        try:
            for w in self.pending:
                w.wait()
        finally:                                                # never carry handles / ranges of a failed step into the next
            self.pending, self.covered = [], []

    __call__ = finish


def shard_sizes(n, world):
    """Balanced split of a batch of n over `world` ranks (sizes differ by at most one; DataParallel scatters in chunks too)."""
    base, extra = divmod(n, world)
    return [base + (1 if r < extra else 0) for r in range(world)]


def shard_batch(input, target, rank, world):
    """This rank's contiguous share of a global batch.  Every rank must get at least one sample (all ranks see the same
    n and world, so all of them raise together otherwise - nobody is left waiting in a collective)."""
    if world == 1:
        return input, target
    n = input.shape[0]
    if n < world:
        raise ValueError("batch of %d samples cannot be sharded over %d ranks" % (n, world))
    sizes = shard_sizes(n, world)
    lo = sum(sizes[:rank])
    return input[lo:lo + sizes[rank]], target[lo:lo + sizes[rank]]


def _batch_metrics(output, target, counters):
    """loss (autograd) + top-1/top-5 % of one batch through one kernel launch (eval/util.py:26-40)."""
    counters.zero_()
    loss = HF.cross_entropy(output, target, counters, 5)
    c = counters.tolist()
    n = float(target.shape[0])
    return loss, 100.0 * c[0] / n, 100.0 * c[1] / n


def train(epoch, train_loader, model, criterion, optimizer, opt, lang_puller=None, grad_sync=None, rank=0, world=1, log=print):
    """train_supervised.py:205-268.  `criterion` is accepted for signature compatibility: CrossEntropyLoss and the accuracy
    counters are one fused kernel.  Returns (top1.avg, losses.avg)."""
    model.train()
    batch_time, data_time, losses, top1, top5 = (AverageMeter() for _ in range(5))
    dev = next(model.parameters()).device
    counters = torch.zeros(2, dtype=torch.int32, device=dev)
    # One process, one GPU, opt.hip_graph = True: the whole step (forward, loss + accuracy counters, backward, optimiser step) as ONE
    # hipGraph per batch shape and learning rate (train.GraphedStep).  Opt-in: the step is bound by the GPU (3.9 ms eager, 4.06 ms
    # replayed at 64 images, profiles/r06_train_step.txt) - the graph frees 2.5 ms of host time per step, it does not shorten the step.
    # The stepper lives on the optimiser, so the graphs survive from epoch to epoch.
    stepper = None
    if world == 1 and grad_sync is None and getattr(opt, "hip_graph", False):
        stepper = getattr(optimizer, "_subreg_graphed_step", None)
        if stepper is None or stepper.model is not model:
            def loss_fn(input, target, _model=model, _opt=opt, _lp=lang_puller):
                output = _model(input)
                stepper.counters.zero_()
                loss = HF.cross_entropy(output, target, stepper.counters, 5)
                if getattr(_opt, "label_pull", None) is not None and _lp is not None:   # :232-236
                    loss = loss + _lp.loss1(_opt.label_pull, _lp(_model.classifier.weight), _model.classifier.weight)
                return loss
            stepper = optimizer._subreg_graphed_step = GraphedStep(model, optimizer, loss_fn)
            stepper.counters = counters
    end = time.time()
    for idx, (input, target, *_rest) in enumerate(train_loader):
        data_time.update(time.time() - end)
        n_global = input.shape[0]
        input, target = shard_batch(input.float(), target, rank, world)
        input, target = input.to(dev), target.to(dev).long()
        if stepper is not None:
            loss = stepper(input, target)
            c, n = stepper.counters.tolist(), float(target.shape[0])
            acc1, acc5 = 100.0 * c[0] / n, 100.0 * c[1] / n
            losses.update(loss.item(), input.size(0))
            top1.update(acc1, input.size(0))
            top5.update(acc5, input.size(0))
        else:
            output = model(input)
            loss, acc1, acc5 = _batch_metrics(output, target, counters)
            if getattr(opt, "label_pull", None) is not None and lang_puller is not None:       # :232-236
                loss = loss + lang_puller.loss1(opt.label_pull, lang_puller(model.classifier.weight), model.classifier.weight)
            losses.update(loss.item(), input.size(0))
            top1.update(acc1, input.size(0))
            top5.update(acc5, input.size(0))
            optimizer.zero_grad()
            if world > 1:
                # SUM over ranks of d(local mean * n_local / n_global) = d(global mean): DataParallel's gradient, no division after
                (loss * (float(input.shape[0]) / float(n_global))).backward()
            else:
                loss.backward()
            if grad_sync is not None:
                grad_sync.finish(optimizer.param_groups[0]["params"])
            optimizer.step()
        batch_time.update(time.time() - end)
        end = time.time()
        if idx % opt.print_freq == 0:
            log("Epoch: [{0}][{1}/{2}]\t"
                "Time {batch_time.val:.3f} ({batch_time.avg:.3f})\t"
                "Data {data_time.val:.3f} ({data_time.avg:.3f})\t"
                "Loss {loss.val:.4f} ({loss.avg:.4f})\t"
                "Acc@1 {top1.val:.3f} ({top1.avg:.3f})\t"
                "Acc@5 {top5.val:.3f} ({top5.avg:.3f})".format(epoch, idx, len(train_loader), batch_time=batch_time,
                                                               data_time=data_time, loss=losses, top1=top1, top5=top5))
            sys.stdout.flush()
    log(" * Acc@1 {top1.avg:.3f} Acc@5 {top5.avg:.3f}".format(top1=top1, top5=top5))
    return top1.avg, losses.avg


def validate(val_loader, model, criterion, opt, log=print):
    """eval/util.py:185-232.  Returns (top1.avg, top5.avg, losses.avg)."""
    batch_time, losses, top1, top5 = (AverageMeter() for _ in range(4))
    model.eval()
    dev = next(model.parameters()).device
    counters = torch.zeros(2, dtype=torch.int32, device=dev)
    with torch.no_grad():
        end = time.time()
        for idx, (input, target, *_rest) in enumerate(val_loader):
            input, target = input.float().to(dev), target.to(dev).long()
            output = model(input)
            if getattr(opt, "dataset", "") == "tieredImageNet" and getattr(opt, "augment_pretrain_wtrainb", False):
                output = output[:, :200].contiguous()
            loss, acc1, acc5 = _batch_metrics(output, target, counters)
            losses.update(loss.item(), input.size(0))
            top1.update(acc1, input.size(0))
            top5.update(acc5, input.size(0))
            batch_time.update(time.time() - end)
            end = time.time()
            if idx % opt.print_freq == 0:
                log("Test: [{0}/{1}]\tLoss {loss.val:.4f} ({loss.avg:.4f})\tAcc@1 {top1.val:.3f} ({top1.avg:.3f})\t"
                    "Acc@5 {top5.val:.3f} ({top5.avg:.3f})".format(idx, len(val_loader), loss=losses, top1=top1, top5=top5))
        log(" * Acc@1 {top1.avg:.3f} Acc@5 {top5.avg:.3f}".format(top1=top1, top5=top5))
    return top1.avg, top5.avg, losses.avg


def fit(model, opt, train_loader, val_loader=None, lang_puller=None, rank=0, world=1, group=None, log=print):
    """train_supervised.py:122-202 without the tensorboard logger: optimizer, LR schedule, epochs, periodic and final
    checkpoints (rank 0 writes).  Returns the per-epoch history."""
    if getattr(opt, "adam", False):                                                              # :128-131
        optimizer = Adam(model.parameters(), lr=opt.learning_rate, weight_decay=0.0005)
    else:                                                                                        # :133-136
        optimizer = SGD(model.parameters(), lr=opt.learning_rate, momentum=opt.momentum, weight_decay=opt.weight_decay)
    sync = None
    if world > 1:
        from . import sweep
        sweep.broadcast_module(model, 0, group)                 # every replica starts from rank 0's parameters and buffers
        sync = GradientSync(group)
        model.hip_backbone().grad_stage_hook = sync.stage_ready   # overlap the all-reduce with the backward, stage by stage
    history = []
    try:
        _fit_epochs(model, opt, train_loader, val_loader, lang_puller, rank, world, log, optimizer, sync, history)
    finally:
        if sync is not None:                                    # a backward outside fit() must not start collectives nobody waits on
            model.hip_backbone().grad_stage_hook = None
            sync.pending, sync.covered = [], []
    if rank == 0:                                                                                 # :193-202
        ck.save_checkpoint(os.path.join(opt.model_path, "{}_last.pth".format(opt.model)), model, opt=opt,
                           **_continual_extra(opt, train_loader))
    return history


def _fit_epochs(model, opt, train_loader, val_loader, lang_puller, rank, world, log, optimizer, sync, history):
    for epoch in range(1, opt.epochs + 1):
        set_epoch_lr(epoch, opt, optimizer)
        rec = {"epoch": epoch, "lr": optimizer.param_groups[0]["lr"]}
        if not getattr(opt, "eval_only", False):
            t0 = time.time()
            rec["train_acc"], rec["train_loss"] = train(epoch, train_loader, model, None, optimizer, opt, lang_puller, sync,
                                                        rank, world, log)
            log("epoch {}, total time {:.2f}".format(epoch, time.time() - t0))
        if val_loader is not None:
            rec["test_acc"], rec["test_acc_top5"], rec["test_loss"] = validate(val_loader, model, None, opt, log)
        history.append(rec)
        if rank == 0 and epoch % opt.save_freq == 0:                                              # :181-191
            log("==> Saving...")
            extra = _continual_extra(opt, train_loader)
            ck.save_checkpoint(os.path.join(opt.model_path, "ckpt_epoch_{epoch}.pth".format(epoch=epoch)), model, epoch=epoch,
                               **extra)


def _continual_extra(opt, train_loader):
    if not getattr(opt, "continual", False):
        return {}
    ds = train_loader.dataset
    return {"training_classes": ds.basec_map, "label2human": ds.label2human}
