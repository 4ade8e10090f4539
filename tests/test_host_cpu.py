"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/subreg_hip.h declares, the
ctypes structs match the C layout, the host-side tiling/index logic of the conv kernel is right, and the product
package never imports the oracle."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import REPO
from subreg_hip import _lib, synthetic as syn


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "subreg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(subreg_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return ctypes.CDLL(_lib.LIB_PATH)


def test_header_symbols_exported_and_bound(built):
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(built, name), "libsubreg_hip.so does not export %s" % name
    assert sorted(_lib.SIGNATURES) == declared, set(declared) ^ set(_lib.SIGNATURES)


def test_abi_version_and_strerror(built):
    built.subreg_abi_version.restype = ctypes.c_int
    assert built.subreg_abi_version() == _lib.ABI_VERSION
    built.subreg_strerror.restype = ctypes.c_char_p
    assert b"invalid" in built.subreg_strerror(-1)


def test_bad_arguments_return_einval_without_gpu(built):
    # argument validation happens before any HIP call, so it is testable on a CPU-only box
    built.subreg_conv_fwd.restype = ctypes.c_int
    rc = built.subreg_conv_fwd(None, None, None, None, None, None, None, None, None, 0, 1, 8, 8, 32, 32, 3, 0, 0, None)
    assert rc == -1
    built.subreg_subspace_basis.restype = ctypes.c_int
    assert built.subreg_subspace_basis(None, None, None, 60, 640, None, None) == -1


def test_struct_layouts_match_c():
    # sizes computed by hand from include/subreg_hip.h on LP64
    assert ctypes.sizeof(_lib.ConvDesc) == 9 * 8 + 5 * 4 + 4
    assert ctypes.sizeof(_lib.LoopState) == 20
    assert ctypes.sizeof(_lib.MaskParam) == 16 and _lib.MaskParam.p_drop.offset == 8     # subreg_mask_param
    assert ctypes.sizeof(_lib.BlockDesc) == 4 * ctypes.sizeof(_lib.ConvDesc) + 8 + 8 + 8 + 8 + 8 + 8     # ... + mask_scale_dev
    assert _lib.StepDesc.weight.offset == 32 and _lib.StepDesc.n_base.offset == 72
    # offsetof(subreg_step_desc, exp_avg_sq / bias / bias_base), sizeof: gcc on include/subreg_hip.h
    assert (_lib.StepDesc.exp_avg_sq.offset, _lib.StepDesc.bias.offset, _lib.StepDesc.bias_base.offset) == (232, 240, 264)
    assert ctypes.sizeof(_lib.StepDesc) == 272
    # offsetof(subreg_train_desc, side_stream / dr_alt / stats_side), sizeof
    assert (_lib.TrainDesc.side_stream.offset, _lib.TrainDesc.dr_alt.offset, _lib.TrainDesc.stats_side.offset) == (96, 152, 168)
    assert ctypes.sizeof(_lib.TrainDesc) == 200 and _lib.TrainDesc.splitk_ws.offset == 176 and _lib.TrainDesc.eval_mode.offset == 192


def test_conv_tiling_index_emulation():
    exe = os.path.join(REPO, "tests", "csrc", "conv_index_test")
    subprocess.run(["g++", "-O2", "-std=c++17", exe + ".cpp", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-2000:]


def test_layer1_plane_swizzle_emulation():
    """Host emulation of the layer-1 fused kernel's LDS plane layout (csrc/conv64_resident.hip::conv64_fused_first_kernel, W = 84, tiles of
    3 image rows): conv1's ds_write_b64 of (pixel, 8-byte half, logical slot q) and conv2's ds_read_b128 of (row tile, tap, k-step) with
    the tap's dy as an IMMEDIATE offset of dy * P rows.  (1) with P = 96 every fragment read lands on the slot conv1 wrote for that pixel
    and channel group; with P = 100 it does not (swz(row + P) != swz(row)) - which is why the pitch that would keep a tile crossing an
    image-row boundary conflict-free cannot be used.  (2) LDS-array cycles of a tile's A-fragment reads under MI355X_MICROARCH.md's
    ds_read_b128 lane groups: 1.31 x the conflict-free count at P = 96, all of it in the two row tiles that cross a row boundary."""
    W, R, ROWB = 84, 3, 64
    swz = lambda row: (row >> 2) & 3
    g = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    groups = g + [[l + 32 for l in x] for x in g]

    def plane(P):
        """byte offset -> (pixel row block, column, channel) for every byte conv1 writes: lane (pixel, lh) writes channels 8 q + 4 lh .. + 3"""
        mem = {}
        for rb in range(R + 2):
            for x in range(W):
                row = rb * P + 1 + x
                for q in range(4):
                    for lh in range(2):
                        base = row * ROWB + 8 * lh + 16 * (q ^ swz(row))
                        for e in range(4):
                            mem[base + 2 * e] = (rb, x, 8 * q + 4 * lh + e)
        return mem

    def reads(P):
        """(lane addresses, wanted (block, column, first channel)) of every A-fragment read of a tile"""
        out = []
        for tile in range(8):
            for dx in range(3):
                for s in range(2):
                    for dy in range(3):
                        addr, want = [], []
                        for lane in range(64):
                            lr, lh = lane & 31, lane >> 5
                            j = tile * 32 + lr
                            jv = j if j < R * W else 0
                            ir, w = divmod(jv, W)
                            row = ir * P + w + dx                       # the dy = -1 row: column (w + 1) + (dx - 1) of block ir
                            a = (row * ROWB + 16 * (lh ^ swz(row))) ^ (32 * s)
                            addr.append(a + dy * P * ROWB)
                            want.append((ir + dy, w + dx - 1, 16 * s + 8 * lh))
                        out.append((tile, addr, want))
        return out

    def consistent(P):
        mem = plane(P)
        for _tile, addr, want in reads(P):
            for a, (rb, x, c0) in zip(addr, want):
                if 0 <= x < W:                                          # (pad columns are zeros wherever they are read from)
                    if mem.get(a) != (rb, x, c0) or mem.get(a + 14) != (rb, x, c0 + 7):
                        return False
        return True

    assert consistent(96) and not consistent(100)
    cyc, ideal, by_tile = 0, 0, {}
    for tile, addr, _want in reads(96):
        c = 0
        for grp in groups:
            banks = {}
            for l in grp:
                for d in range(4):
                    dw = addr[l] // 4 + d
                    banks.setdefault(dw % 64, set()).add(dw)
            c += max(len(v) for v in banks.values())
        cyc, ideal = cyc + c, ideal + 4
        by_tile[tile] = by_tile.get(tile, 0) + c - 4
    assert 1.25 < cyc / ideal < 1.35, cyc / ideal
    assert sorted(t for t, extra in by_tile.items() if extra > 0) == [2, 5, 7], by_tile   # pixels 84 and 168 fall in tiles 2 and 5 (7: its padding rows)


def test_m0_is_touched_only_by_the_lds_dma_statements(built):
    """The LDS-DMA helpers declare M0 clobbered instead of saving / restoring it (csrc/subreg_common.h): sound only while the
    compiler never keeps a value of its own in M0.  The Makefile checks every kernel file's ISA on every build
    (tools/check_isa.py over the .s files -save-temps leaves in build/); this test re-runs that check on the library the suite
    loads, and proves the checker itself fires on an offending line."""
    import subprocess
    import sys
    import tempfile
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bdir = os.path.join(repo, "subspace-reg_amd", "build")
    import glob
    if not glob.glob(os.path.join(bdir, "*-gfx950.s")):
        _lib.build()                                         # (a library built elsewhere: rebuild here to get the ISA files)
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "check_isa.py"), bdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "x-hip-amdgcn-amd-amdhsa-gfx950.s"), "w") as f:
            f.write("\ts_mov_b32 m0, s4\n\t;;#ASMSTART\n\ts_mov_b32 m0, s5\n\t;;#ASMEND\n\tv_movrels_b32 v1, v2\n")
        r = subprocess.run([sys.executable, os.path.join(repo, "tools", "check_isa.py"), td], capture_output=True, text=True)
        assert r.returncode == 1 and "2 line(s)" in r.stdout, r.stdout


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "subspace-reg_amd", "subreg_hip")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn


def test_compute_entry_points_fail_loudly_without_gpu():
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from subreg_hip.resnet_language import create_model
    from types import SimpleNamespace
    net = create_model("resnet18", 60, SimpleNamespace(no_dropblock=True, linear_bias=False))
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 84, 84))


def test_module_state_dict_keys_match_reference():
    torch = pytest.importorskip("torch")
    from subreg_hip.resnet_language import create_model
    from types import SimpleNamespace
    net = create_model("resnet18", 60, SimpleNamespace(no_dropblock=True, linear_bias=False))
    sd = syn.make_state_dict(1)
    assert sorted(net.state_dict().keys()) == sorted(sd.keys()) and len(sd) == 133
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    net.augment_base_classifier_(5)
    assert net.classifier.weight.shape == (65, 640) and net.num_classes == 60
    names = [n for n, _ in net.named_parameters() if n.startswith("classifier")]
    assert names == ["classifier.weight"]


def test_synthetic_episode_layout():
    sy, qy = syn.session_labels(2)
    assert sy.shape == (125,) and qy.shape == (125,) and set(sy) == set(range(70, 75))
    assert list(sy[:10]) == [70] * 5 + [71] * 5 and list(qy[:26]) == [70] * 25 + [71]
    from oracle.resnet_ref import conv_flops_per_image
    assert abs(conv_flops_per_image(84) / 8.1219e9 - 1) < 1e-3


def test_use_synonyms_fails_like_the_reference(tmp_path):
    """models/resnet_language.py:35-45: `--use_synonyms` opens <dataset>_dim<dim>_base_synonyms.pickle (absent from the reference
    repository), builds a Python list from it and calls .float() on the list - FileNotFoundError, then AttributeError."""
    import pickle
    from types import SimpleNamespace
    from subreg_hip.resnet_language import LangPuller
    opt = SimpleNamespace(use_synonyms=True, word_embed_path=str(tmp_path), dataset="miniImageNet", word_embed_size=500, temperature=1)
    with pytest.raises(FileNotFoundError):
        LangPuller(opt, ["a", "b"], ["c"])
    with open(os.path.join(str(tmp_path), "miniImageNet_dim500_base_synonyms.pickle"), "wb") as f:
        pickle.dump({"a": np.zeros(500), "b": np.ones(500)}, f)
    with pytest.raises(AttributeError, match="'list' object has no attribute 'float'"):
        LangPuller(opt, ["a", "b"], ["c"])
    with pytest.raises(KeyError):
        LangPuller(opt, ["a", "zzz"], ["c"])


def test_checkpoint_round_trip_in_reference_format(tmp_path):
    """train_supervised.py:181-202 writer / eval_incremental.py:86-123 reader: same dict keys, 133 model keys, bias rule."""
    import argparse
    import torch
    from subreg_hip import checkpoint as ck
    from subreg_hip.resnet_language import create_model
    opt = argparse.Namespace(no_dropblock=True, linear_bias=False, model="resnet18", continual=True)
    net = create_model("resnet18", 60, opt)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in syn.make_state_dict(3).items()}
    net.load_state_dict(sd)
    basec = {int(c): i for i, c in enumerate(range(5, 65))}
    names = ["class%d" % i for i in range(60)] + [""] * 40
    last = ck.save_checkpoint(str(tmp_path / "m" / "resnet18_last.pth"), net, opt=opt, training_classes=basec, label2human=names)
    per = ck.save_checkpoint(str(tmp_path / "m" / "ckpt_epoch_10.pth"), net, epoch=10)
    c1, c2 = ck.load_checkpoint(last), ck.load_checkpoint(per)
    assert sorted(c1.keys()) == ["label2human", "model", "opt", "training_classes"] and sorted(c2.keys()) == ["epoch", "model"]
    assert c2["epoch"] == 10 and c1["opt"].model == "resnet18" and len(c1["model"]) == 133
    assert ck.infer_linear_bias(c1) is False
    fwd, rev = ck.base_class_maps(c1)
    assert fwd == basec and all(fwd[rev[i]] == i for i in range(60))
    opt2 = argparse.Namespace(no_dropblock=True, linear_bias=True)
    net2 = ck.model_from_checkpoint(c1, "resnet18", 60, opt2)
    assert opt2.linear_bias is False
    for k, v in net2.state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k
    # a plain torch state_dict written the way the reference writes it loads too (no HIP-side keys leak into the file)
    torch.save({"opt": opt, "model": sd}, str(tmp_path / "ref_style.pth"))
    net3 = ck.model_from_checkpoint(ck.load_checkpoint(str(tmp_path / "ref_style.pth")), "resnet18", 60, opt2)
    assert torch.equal(net3.state_dict()["layer4.1.bn3.running_var"].cpu(), sd["layer4.1.bn3.running_var"])
    with pytest.raises(KeyError):
        torch.save({"opt": opt}, str(tmp_path / "bad.pth"))
        ck.load_checkpoint(str(tmp_path / "bad.pth"))


def test_pretrain_lr_schedules_match_torch_and_reference_rule():
    """util.py:45-51 step decay and train_supervised.py:146-157 cosine schedule (torch's CosineAnnealingLR, stepped first)."""
    import torch
    from types import SimpleNamespace
    from subreg_hip import pretrain as pt
    opt = SimpleNamespace(learning_rate=0.05, lr_decay_rate=0.1, lr_decay_epochs=[60, 80], epochs=100, cosine=False)
    o = SimpleNamespace(param_groups=[{"lr": 0.05}])
    lrs = []
    for e in range(1, 101):
        pt.set_epoch_lr(e, opt, o)
        lrs.append(o.param_groups[0]["lr"])
    assert lrs[59] == 0.05 and abs(lrs[60] - 0.005) < 1e-12 and abs(lrs[79] - 0.005) < 1e-12 and abs(lrs[80] - 0.0005) < 1e-12
    opt.cosine = True
    p = torch.nn.Parameter(torch.zeros(1))
    ref_opt = torch.optim.SGD([p], lr=opt.learning_rate)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(ref_opt, opt.epochs, opt.learning_rate * opt.lr_decay_rate ** 3, -1)
    for e in range(1, 101):
        ref_opt.step()
        sched.step()
        pt.set_epoch_lr(e, opt, o)
        assert abs(o.param_groups[0]["lr"] - ref_opt.param_groups[0]["lr"]) < 1e-9, e
    m = pt.AverageMeter()
    m.update(2.0, 3)
    m.update(4.0, 1)
    assert m.avg == 2.5 and m.val == 4.0 and m.count == 4
    x, y = torch.arange(10)[:, None], torch.arange(10)
    assert [pt.shard_batch(x, y, r, 4)[1].tolist() for r in range(4)] == [[0, 1, 2], [3, 4, 5], [6, 7], [8, 9]]   # balanced: sizes differ by at most one


def test_episode_sampler_matches_reference_dataset_classes():
    """subreg_hip.episodes vs the reference's ImageNet / MetaImageNet run on a synthetic all.pickle (tests/golden/episodes.npz):
    class split, base test split, replay-memory episode, eight disjoint novel sessions with x5 tiled support."""
    from conftest import GOLDEN
    from subreg_hip import episodes as ep
    g = np.load(os.path.join(GOLDEN, "episodes.npz"))
    labels = g["labels"].tolist()
    for seed in (1, 7):
        key = "seed%d" % seed
        sp = ep.continual_split(labels, seed)
        assert np.array_equal(sp["basec"], g[key + ".basec"]) and len(sp["valc"]) == 40
        assert int(g[key + ".label2human_nonempty"]) == 60
        bt = ep.BaseSplit(labels, seed, "test")
        assert len(bt) == int(g[key + ".base_test_len"]) == 60 * 50
        got = np.array([bt.item(i) for i in range(0, len(bt), 37)])
        assert np.array_equal(got, g[key + ".base_test_items"])
        btr = ep.BaseSplit(labels, seed, "train")
        assert len(btr) == 60 * 500 and len(ep.BaseSplit(labels, seed, "val")) == 60 * 50
        for item in (0, 3):
            pos, ys = ep.base_support_episode(btr.labels, item, 1, 0)
            assert np.array_equal(btr.indices[pos], g["%s.base_support%d.pos" % (key, item)])
            assert np.array_equal(ys, g["%s.base_support%d.ys" % (key, item)])
        ns = ep.NovelSessions(labels, seed)
        assert len(ns) == 8
        seen = set()
        for item in range(8):
            s_pos, s_ys, q_pos, q_ys = ns.next_session(item)
            assert np.array_equal(ns.indices[s_pos], g["%s.s%d.sup" % (key, item)])
            assert np.array_equal(s_ys, g["%s.s%d.sup_ys" % (key, item)])
            assert np.array_equal(ns.indices[q_pos], g["%s.s%d.qry" % (key, item)])
            assert np.array_equal(q_ys, g["%s.s%d.qry_ys" % (key, item)])
            assert s_pos.shape == (125,) and q_pos.shape == (125,) and not set(s_pos) & set(q_pos)
            cls = set(np.unique(q_ys).tolist())
            assert len(cls) == 5 and not cls & seen and cls <= set(sp["valc"].tolist())      # disjoint sessions of novel classes
            seen |= cls
    with pytest.raises(ValueError):
        ep.BaseSplit(labels, 1, "nope")
