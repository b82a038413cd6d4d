"""Sweep of input geometries (batch, height, width) through the engine in three modes -- exact-fp32 kernels, the default
split arithmetic, bf16 storage -- one eval forward + one training step each; prints the agreement with the exact family
and any call that fails.  usage: python tools/geom_fuzz.py [n,h,w ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import fcdensenet_oracle as O  # noqa: E402  (inputs / initialiser only)
from sim2real_lane_segment_amd.engine import Engine, NetSpec, parse_dense_arith  # noqa: E402

DEFAULT = [(1, 32, 32), (5, 32, 40), (7, 40, 56), (1, 64, 64), (3, 72, 104), (2, 100, 140), (1, 33, 47), (4, 120, 160),
           (1, 120, 200), (2, 160, 120), (9, 48, 80), (1, 200, 320), (2, 35, 45), (6, 64, 80), (1, 360, 480)]


def run(n, h, w):
    cfg = O.NetConfig()
    st = O.init_state(cfg, 3)
    g = torch.Generator().manual_seed(5 + h + w)
    x = torch.randn(n, 3, h, w, generator=g).cuda()
    y = torch.randint(0, 4, (n, h, w), generator=g).cuda()
    scales = O.make_drop_scales(cfg, n, 11)
    res = {}
    for mode in ("exact", "default", "bf16"):
        try:
            if mode == "exact":
                eng = Engine(NetSpec(n_classes=4), device="cuda", dense_arith=parse_dense_arith("fp32,fp32"))
            else:
                eng = Engine(NetSpec(n_classes=4), device="cuda")
                if mode == "bf16":
                    eng.set_storage("bf16")
            eng.load_state(st)
            probs, _ = eng.forward(x, training=False)
            pe = probs.float().cpu()
            pt, _ = eng.forward(x, training=True, with_backward=True, drop_scales=eng.pack_drop_scales(scales))
            out, _, _ = eng.loss(pt, y, weighted=True)
            eng.backward(1.0)
            torch.cuda.synchronize()
            if not eng.check_ws_guard():  # (RLN_WS_GUARD=<bytes>: a kernel wrote past the workspace)
                print(f"  {n}x{h}x{w} {mode}: WORKSPACE GUARD BROKEN")
            res[mode] = (pe, float(out[0]), eng.grads.clone().cpu())
        except Exception as e:  # noqa: BLE001
            print(f"  {n}x{h}x{w} {mode}: FAILED {str(e)[:200]}")
    if "exact" not in res:
        return
    pe, le, ge = res["exact"]
    for mode in ("default", "bf16"):
        if mode not in res:
            continue
        p, l, gr = res[mode]
        ok = bool(torch.isfinite(gr).all())
        print(f"  {n}x{h}x{w} {mode:8s} masks {float((p.argmax(1) == pe.argmax(1)).float().mean()):.4f} max|dp| "
              f"{float((p - pe).abs().max()):.1e} |dloss| {abs(l - le):.1e} grad arena {float((gr - ge).norm() / ge.norm()):.1e}"
              f"{'' if ok else '  NON-FINITE GRADS'}")


if __name__ == "__main__":
    geoms = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or DEFAULT
    for n, h, w in geoms:
        run(n, h, w)
