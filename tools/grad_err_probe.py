"""Per-tensor gradient error of the HIP path against the CPU oracle (fp32) for several dense-arithmetic modes, next to the
fp32-vs-fp64 noise floor of tests/golden/grad_noise_floor.json.  GPU box: python tools/grad_err_probe.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fcdensenet_oracle as O
from tests.golden.common import synth_batch
from sim2real_lane_segment_amd.engine import Engine, NetSpec, parse_dense_arith
from sim2real_lane_segment_amd.models.FCDenseNet import tiramisu as T

noise = json.load(open("tests/golden/grad_noise_floor.json"))

def run(name, cfg, n, h, w, seeds, modes):
    st = O.init_state(cfg, seeds[0])
    x, y = synth_batch(n, h, w, cfg.n_classes, seeds[1])
    scales = O.make_drop_scales(cfg, n, seeds[2])
    torch.set_num_threads(16)
    ts = O.TrainState({k: v.clone() for k, v in st.items()})
    loss, acc, grads, probs_ref = O.train_step(ts, x, y, cfg, scales, apply_update=False)
    nf = noise[name]["per_tensor"]
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
        errs, ratios = [], []
        for k, g in grads.items():
            got = eng.grad_views[k].cpu()
            floor = 1e-5 * g.numel() ** 0.5
            e = float((got - g).norm()) / max(float(g.norm()), floor)
            errs.append((e, k))
            ratios.append(e / max(nf[k], 1e-5))
        errs.sort()
        ratios.sort()
        q = lambda v, f: v[int(f * (len(v) - 1))]
        print(f"{name} mode {mode:16s} loss diff {abs(float(out[0]) - float(loss)):.2e} probs max {float((probs.cpu() - probs_ref).abs().max()):.2e} | "
              f"L2 err median {q(errs, .5)[0]:.2e} p90 {q(errs, .9)[0]:.2e} max {errs[-1][0]:.2e} ({errs[-1][1]}) | "
              f"err/noise median {q(ratios, .5):.1f} p90 {q(ratios, .9):.1f} max {ratios[-1]:.1f}", flush=True)

modes = ["fp32,fp32", "f16x2,bf16x2", "bf16x3,bf16x3", "f16x2,bf16x3", "bf16x2,bf16x2"]
run("fcd67_2x120x160", O.fcdensenet67_config(4), 2, 120, 160, (700, 701, 702), modes)
for v in ("57", "103"):
    down, up, bott, growth = T._VARIANTS[v]
    cfg = O.NetConfig(down_blocks=down, up_blocks=up, bottleneck_layers=bott, growth_rate=growth, n_classes=4)
    run(f"fcd{v}_2x64x96", cfg, 2, 64, 96, (21, 22, 23), modes[:3])
