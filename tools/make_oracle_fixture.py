#!/usr/bin/env python3
"""Expected outputs of the 351-base-class incremental case at 84x84 (BASELINE.json configs[4], incremental leg), computed ONCE in
the build container by the build's own NumPy oracle (oracle/loop_ref.py, pinned at 60 classes by the reference-generated
goldens of tests/test_oracle_golden.py) and committed as data: tests/golden/oracle_loop351_hw84.npz.

NOT a reference golden: the reference cannot run tieredImageNet (eval_incremental.py:82-83 raises), so this fixture only saves
the GPU box the ~20 minutes of NumPy the case costs.  Inputs are regenerated from the seeds below by the test
(tests/test_hip_loop.py::test_fused_loop_351_base_classes_hw84_against_cached_oracle); only expected outputs are stored.

  python tools/make_oracle_fixture.py            (about 20 minutes on 8 cores)
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "subspace-reg_amd"), os.path.join(REPO, "tests")]
import numpy as np   # noqa: E402

CASE = dict(NB=351, seed=9, signal=3.0, hw=84, ns=2, n_epochs=3, sd_seed=40, mask_seed=77, n_base_batch=64)


def case_inputs():
    from subreg_hip import synthetic as syn
    c = CASE
    sd = syn.make_state_dict(c["sd_seed"], n_cls=c["NB"])
    sessions = syn.make_sessions(c["seed"], c["ns"], c["hw"], class_signal=c["signal"], first_novel=c["NB"])
    bx, by = syn.make_base_batch(c["seed"], c["n_base_batch"], c["hw"], n_base=c["NB"], class_signal=c["signal"])
    sx, sy = syn.make_base_support(c["seed"], c["hw"], n_base=c["NB"], class_signal=c["signal"])
    inits = syn.make_novel_inits(c["seed"], c["ns"])
    picks = [np.array([1]), np.array([3])][:c["ns"]]
    return sd, sessions, (bx, by), (sx, sy), inits, picks


def main():
    from oracle import loop_ref
    from oracle.resnet_ref import MaskSource, ResNetRef, copy_state_dict
    from test_hip_loop import make_opt
    c = CASE
    sd, sessions, base, bsup, inits, picks = case_inputs()
    opt = make_opt(set_seed=c["seed"], neval_episodes=c["ns"], memory_replay=1, hip_dtype="f32", max_novel_epochs=c["n_epochs"],
                   dataset="tieredImageNet", avg_weights_follow_n_base=True)
    t0 = time.time()
    want = loop_ref.run_incremental(ResNetRef(copy_state_dict(sd)), sessions, base, opt, inits, base_support=bsup,
                                    masks=MaskSource(c["mask_seed"]), memory_picks=picks, n_base=c["NB"])
    print("oracle run: %.0f s" % (time.time() - t0))
    out = {"case." + k: np.asarray(v) for k, v in c.items()}
    for k in ("epochs", "weighted_avg", "acc_base", "base_vals", "novel_vals", "novel_acc"):
        out[k] = np.asarray(want[k], np.float64)
    out["classifier_weight"] = np.asarray(want["classifier_weight"], np.float32)
    for s in range(c["ns"]):
        out["loss.%d" % s] = np.asarray(want["loss"][s], np.float64)
        out["test_acc.%d" % s] = np.asarray(want["test_acc"][s], np.float64)
        out["test_acc_top5.%d" % s] = np.asarray(want["test_acc_top5"][s], np.float64)
    path = os.path.join(REPO, "tests", "golden", "oracle_loop351_hw84.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
