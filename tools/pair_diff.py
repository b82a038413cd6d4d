"""Dump the probabilities and gradients of one FCDenseNet103 training step at 2x64x96 (the parity test's inputs) to an
.npz, or compare two such dumps: used to tell a summation-order perturbation from a bug when a kernel change moves the
ill-conditioned gradient tensors of that test.  usage: pair_diff.py dump out.npz | pair_diff.py cmp a.npz b.npz"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def dump(path, variant="103"):
    import torch
    from oracle import fcdensenet_oracle as O
    from sim2real_lane_segment_amd.models.FCDenseNet import tiramisu as T
    from tests.golden.common import synth_batch
    from tests.test_gpu_parity import make_engine
    down, up, bott, growth = T._VARIANTS[variant]
    cfg = O.NetConfig(down_blocks=down, up_blocks=up, bottleneck_layers=bott, growth_rate=growth, n_classes=4)
    st = O.init_state(cfg, 21)
    x, y = synth_batch(2, 64, 96, 4, 22)
    if os.environ.get("PERTURB"):  # conditioning probe: relative input perturbation
        x = x * (1.0 + float(os.environ["PERTURB"]))
    scales = O.make_drop_scales(cfg, 2, 23)
    if os.environ.get("ORACLE"):  # the CPU oracle's own gradients on the same inputs
        ts = O.TrainState({k: v.clone() for k, v in st.items()})
        loss, acc, grads, probs_ref = O.train_step(ts, x, y, cfg, scales, apply_update=False)
        np.savez(path, probs=probs_ref.numpy(), **{"g:" + k: v.numpy() for k, v in grads.items()})
        return
    eng = make_engine(cfg, st)
    probs, _ = eng.forward(x.cuda(), training=True, with_backward=True, drop_scales=eng.pack_drop_scales(scales))
    eng.loss(probs, y.cuda(), weighted=True)
    eng.backward(1.0)
    torch.cuda.synchronize()
    out = {"probs": probs.cpu().numpy()}
    for k, v in eng.grad_views.items():
        out["g:" + k] = v.cpu().numpy()
    np.savez(path, **out)


def cmp(a, b):
    A, B = np.load(a), np.load(b)
    rows = []
    for k in A.files:
        x, y = A[k].astype(np.float64), B[k].astype(np.float64)
        rows.append((float(np.linalg.norm(x - y) / max(np.linalg.norm(x), 1e-30)), k))
    for r in rows:
        if r[0] > 3e-3 and "convTrans.bias" not in r[1]:
            print(f"   {r[0]:.3e}  {r[1]}")
    rows.sort(reverse=True)
    print("probs rel diff", [r for r in rows if r[1] == "probs"][0][0])
    for r in rows[:5]:
        print(f"{r[0]:.3e}  {r[1]}")
    print("median", np.median([r[0] for r in rows]))


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2])
    else:
        cmp(sys.argv[2], sys.argv[3])
