"""Per-tensor gradient error of the HIP path against (a) the REFERENCE's sampled gradients of fixture
fcd67_train_120x160 (what tests/test_gpu_parity.py::test_fcd67_train_steps_vs_golden asserts) and (b) the CPU oracle run
live on the same inputs with full tensors, for several arithmetic modes.  GPU box: python tools/grad_ref_probe.py"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fcdensenet_oracle as O
from tests.golden.common import synth_batch, sample_idx, cfg_from_arrays
from sim2real_lane_segment_amd.engine import Engine, NetSpec, parse_dense_arith

z = np.load("tests/golden/fcd67_train_120x160.npz")
noise = json.load(open("tests/golden/grad_noise_floor.json"))["fcd67_2x120x160"]["per_tensor"]
cfg = cfg_from_arrays(z, O.NetConfig)
n, h, w, seed = int(z["n"]), int(z["h"]), int(z["w"]), int(z["seed"])
st = O.init_state(cfg, seed)
x, y = synth_batch(n, h, w, 4, seed + 1)
y[0][y[0] == 3] = 0
scales = O.make_drop_scales(cfg, n, seed + 2)
torch.set_num_threads(16)
ts = O.TrainState({k: v.clone() for k, v in st.items()})
loss, acc, grads, _ = O.train_step(ts, x, y, cfg, scales, apply_update=False)
modes = sys.argv[1:] or ["fp32,fp32", "f16x2,bf16x2"]
for mode in modes:
    spec = NetSpec(in_channels=cfg.in_channels, down_blocks=cfg.down_blocks, up_blocks=cfg.up_blocks,
                   bottleneck_layers=cfg.bottleneck_layers, growth_rate=cfg.growth_rate,
                   out_chans_first_conv=cfg.out_chans_first_conv, n_classes=cfg.n_classes)
    eng = Engine(spec, device="cuda", dense_arith=parse_dense_arith(mode))
    eng.load_state(st)
    probs, _ = eng.forward(x.cuda(), training=True, with_backward=True, drop_scales=eng.pack_drop_scales(scales))
    out, _, _ = eng.loss(probs, y.cuda(), weighted=True)
    eng.backward(1.0)
    torch.cuda.synchronize()
    rows = []
    for k, g in grads.items():
        got = eng.grad_views[k].cpu()
        floor = 1e-6 * g.numel() ** 0.5
        full = float((got - g).norm()) / max(float(g.norm()), floor)
        idx = sample_idx(g.numel(), 1024, 1234)
        ref = z["gradsamp/" + k].astype(np.float64)
        nrm = float(z["gradnorm/" + k])
        d2 = float(((got.reshape(-1)[idx].numpy().astype(np.float64) - ref) ** 2).sum()) * g.numel() / len(idx)
        samp = np.sqrt(d2) / max(nrm, floor)
        # the oracle itself against the reference samples (thread-count / summation-order noise of the CPU operators)
        o2 = float(((g.reshape(-1)[idx].numpy().astype(np.float64) - ref) ** 2).sum()) * g.numel() / len(idx)
        rows.append((samp, full, np.sqrt(o2) / max(nrm, floor), noise.get(k, 0.0), g.numel(), k))
    rows.sort(reverse=True)
    s = np.array([r[0] for r in rows]); f = np.array([r[1] for r in rows])
    print(f"== mode {mode}: vs reference samples median {np.median(s):.2e} p90 {np.quantile(s, .9):.2e} max {s.max():.2e} | "
          f"vs live oracle (full tensors) median {np.median(f):.2e} p90 {np.quantile(f, .9):.2e} max {f.max():.2e}")
    print("   sampled-vs-ref  full-vs-oracle  oracle-vs-ref  noise-floor  numel  tensor")
    for r in rows[:25]:
        print(f"   {r[0]:.2e}        {r[1]:.2e}        {r[2]:.2e}       {r[3]:.2e}   {r[4]:7d}  {r[5]}")
