"""Index-level episode sampler of the continual miniImageNet pipeline (which images feed the hot path, in which order).

Follows /root/reference/dataset/mini_imagenet.py with numpy's legacy RNG call for call, so that for the same `set_seed` and
label list the same images land in the same sessions as in the reference:
  continual_split        ImageNet.__init__ :31,66-107     seed -> shuffle(arange(100)) -> 60 sorted base classes + 40 novel,
                                                            base images shuffled and cut 500/50/rest per class count
  BaseSplit              ImageNet (split='train', phase)    the base test/val/train loaders (labels remapped to 0..59)
  base_support_episode   MetaImageNet.__getitem__ :286-312  replay memory: per base class n_base_support_samples images
  NovelSessions          MetaImageNet (split='val', fix_seed, disjoint_classes) :273-277, :314-350
                                                            disjoint 5-class sessions, n_shots support (tiled x n_aug), queries
It returns INDICES into the dataset arrays (and the label vectors the loop sees); decoding, augmentation and normalisation
of the pixels are the data pipeline's business (torchvision in the reference) and out of scope.
"""
import numpy as np


def continual_split(labels, set_seed, n_classes=100, n_base=60):
    """-> dict(basec, basec_map, valc, train, val, test): class split and the base-image index split of one seed."""
    rs = np.random.RandomState(set_seed)                       # np.random.seed(args.set_seed), :31
    all_classes = np.arange(n_classes)
    rs.shuffle(all_classes)                                    # :67
    basec = np.sort(all_classes[:n_base])
    basec_map = dict(zip(basec.tolist(), range(len(basec))))
    valc = all_classes[n_base:]
    base_set = set(basec.tolist())
    base_samples = [i for i, e in enumerate(labels) if e in base_set]
    rs.shuffle(base_samples)                                   # :78 (only the split='train' datasets draw this)
    nbc = len(basec)
    return dict(basec=basec, basec_map=basec_map, valc=valc,
                train=np.array(base_samples[:500 * nbc]), val=np.array(base_samples[500 * nbc:550 * nbc]),
                test=np.array(base_samples[550 * nbc:]))


class BaseSplit:
    """ImageNet(args, split='train', phase=...) in continual mode: `indices` into the full arrays, `labels` remapped."""

    def __init__(self, labels, set_seed, phase):
        sp = continual_split(labels, set_seed)
        if phase not in ("train", "val", "test"):
            raise ValueError(f"Phase {phase} is unrecognized for split train.")
        self.indices = sp[phase]
        self.labels = [sp["basec_map"][int(labels[i])] for i in self.indices]
        self.basec_map = sp["basec_map"]

    def __len__(self):
        return len(self.labels)

    def item(self, i):
        """(index into the full image array, target, item) - ImageNet.__getitem__ :168-172."""
        return int(self.indices[i]), self.labels[i] - min(self.labels), i


def _by_class(labels):
    """self.data of MetaImageNet (:263-268): class -> positions, classes in first-appearance order."""
    data = {}
    for pos, lab in enumerate(labels):
        data.setdefault(lab, []).append(pos)
    return data


def base_support_episode(split_labels, item, n_base_support_samples, n_base_aug_support_samples=0):
    """MetaImageNet(split='train', phase='train', fix_seed=True).__getitem__(item), :286-312.
    `split_labels` = BaseSplit(..., 'train').labels.  -> (positions within the split [tiled], support_ys)."""
    data = _by_class(split_labels)
    classes = list(data.keys())
    rs = np.random.RandomState(item)                           # np.random.seed(item), :293
    cls_sampled = rs.choice(classes, len(classes), False)
    pos, ys = [], []
    for cls in np.sort(cls_sampled):
        ids = rs.choice(range(len(data[cls])), n_base_support_samples, False)
        pos.append([data[cls][i] for i in ids])
        ys.append([cls] * n_base_support_samples)
    pos, ys = np.array(pos).reshape(-1), np.array(ys)
    if n_base_aug_support_samples > 1:
        pos = np.tile(pos, n_base_aug_support_samples)
        ys = np.tile(ys.reshape((-1,)), n_base_aug_support_samples)
    return pos, ys


class NovelSessions:
    """MetaImageNet(args, split='val', fix_seed=True, disjoint_classes=True): session `item` takes the next n_ways classes
    of the seed-shuffled class list (the reference MUTATES the list per __getitem__, so items must be drawn in order, once).
    Positions index the novel subset `indices` (all images of the 40 novel classes, in dataset order)."""

    def __init__(self, labels, set_seed, n_ways=5, n_shots=5, n_queries=25, n_aug_support_samples=5,
                 eval_mode="few-shot-incremental-fine-tune"):
        sp = continual_split(labels, set_seed)
        valc = set(sp["valc"].tolist())
        self.indices = np.array([i for i, e in enumerate(labels) if e in valc])      # :95-98 (labels keep their ids)
        self.labels = [int(labels[i]) for i in self.indices]
        self.data = _by_class(self.labels)
        self.classes = list(self.data.keys())
        rs = np.random.RandomState(set_seed)                   # :274-276 (fix_seed)
        rs.shuffle(self.classes)
        self.n_ways, self.n_shots, self.n_queries, self.n_aug = n_ways, n_shots, n_queries, n_aug_support_samples
        self.eval_mode = eval_mode

    def __len__(self):
        return len(self.classes) // self.n_ways

    def next_session(self, item):
        """-> (support positions [tiled n_aug times], support_ys, query positions, query_ys), :314-350."""
        rs = np.random.RandomState(item)                       # np.random.seed(item), :315
        cls_sampled = self.classes[:self.n_ways]               # :318-319
        self.classes = self.classes[self.n_ways:]
        s_pos, s_ys, q_pos, q_ys = [], [], [], []
        for idx, cls in enumerate(np.sort(cls_sampled)):
            n = len(self.data[cls])
            sup = rs.choice(range(n), self.n_shots, False)
            lbl = cls if self.eval_mode in ["few-shot-incremental-fine-tune"] else idx
            s_pos.append([self.data[cls][i] for i in sup])
            s_ys.append([lbl] * self.n_shots)
            rest = np.setxor1d(np.arange(n), sup)
            qry = rs.choice(rest, self.n_queries, False)
            q_pos.append([self.data[cls][i] for i in qry])
            q_ys.append([lbl] * qry.shape[0])
        s_pos, s_ys = np.array(s_pos).reshape(-1), np.array(s_ys)
        q_pos, q_ys = np.array(q_pos).reshape(-1), np.array(q_ys).reshape(-1)
        if self.n_aug > 1:
            s_pos = np.tile(s_pos, self.n_aug)
            s_ys = np.tile(s_ys.reshape((-1,)), self.n_aug)
        return s_pos, s_ys, q_pos, q_ys
