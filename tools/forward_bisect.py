"""Which buffer diverges first when a training forward is not bitwise repeatable: the forward is repeated on one workspace;
on a mismatch the whole workspace is compared with the reference run's copy and the differing byte ranges are reported
per 64 MiB window together with the first differing offset (the level stacks come first in the carve: S[0], S[1], ...)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import fcdensenet_oracle as O  # noqa: E402
from sim2real_lane_segment_amd.engine import Engine, NetSpec  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 600
cfg = O.fcdensenet67_config(4)
st = O.init_state(cfg, 21)
arith = [a for a in sys.argv[1:] if "," in a]
if arith:
    from sim2real_lane_segment_amd.engine import parse_dense_arith
    eng = Engine(NetSpec(n_classes=4), device="cuda", dense_arith=parse_dense_arith(arith[0]))
else:
    eng = Engine(NetSpec(n_classes=4), device="cuda")
eng.load_state(st)
EVAL = "eval" in sys.argv
g = torch.Generator().manual_seed(5)
N, H, W = 64, 120, 160
x = torch.randn(N, 3, H, W, generator=g).cuda()


def fwd():
    eng.load_state(st)
    p = eng.forward(x, training=not EVAL, with_backward=not EVAL, seed=77)[0]
    torch.cuda.synchronize()
    return p


ref = fwd().clone()
ws_ref = eng._ws.clone()
found = 0
for it in range(iters):
    p = fwd()
    if torch.equal(p, ref):
        continue
    found += 1
    ws = eng._ws
    base = (ws.data_ptr() + 255) // 256 * 256 - ws.data_ptr()
    plane = H * W * 4
    win = 64 << 20
    total, first, wins, planes = 0, None, [], []
    for o in range(0, ws.numel(), win):
        a, b = ws[o:o + win], ws_ref[o:o + win]
        if torch.equal(a, b):
            continue
        idx = torch.nonzero(a != b).flatten() + (o - base)
        total += int(idx.numel())
        wins.append((o // win, int(idx.numel())))
        if first is None:
            first = int(idx[0])
        if len(planes) < 48:
            planes += torch.unique(idx // plane).tolist()
    print(f"iteration {it}: {total} workspace bytes differ; first at carve offset {first} (= plane {first // plane}, row "
          f"{first % plane // (W * 4)}, byte-in-row {first % (W * 4)})")
    print("   64 MiB windows with differences:", wins[:20], "..." if len(wins) > 20 else "")
    print("   first differing planes:", planes[:48])
    # per level stack: which channels differ (any sample): the down path fills a level's channels in layer order
    cl_ = [288, 368, 448, 528, 608, 528]
    hw_ = [19200, 4800, 1200, 300, 70, 15]
    off_ = base
    for L_ in range(6):
        size = N * cl_[L_] * hw_[L_] * 4
        a_ = ws[off_:off_ + size].view(torch.float32).view(N, cl_[L_], hw_[L_])
        b_ = ws_ref[off_:off_ + size].view(torch.float32).view(N, cl_[L_], hw_[L_])
        ch = torch.nonzero((a_ != b_).any(dim=2).any(dim=0)).flatten().tolist()
        same = sorted(set(range(cl_[L_])) - set(ch))
        print(f"   level {L_} stack: {len(ch)} of {cl_[L_]} channels differ; first {ch[:12]}, last {ch[-4:]}; identical: {same[:60]}")
        off_ += size
    # statistics arrays behind the stacks (carve order: S[0..5], G[0..5], mean, var, invstd, stdv, ...; FCDenseNet67, N = 64)
    cl = [288, 368, 448, 528, 608, 528]
    hw = [19200, 4800, 1200, 300, 70, 15]
    stacks = sum(N * c_ * p_ * 4 for c_, p_ in zip(cl, hw))
    mean0 = base + (1 if EVAL else 2) * stacks
    nchan = sum(cl)
    seg = (nchan * 4 + 255) // 256 * 256
    for name, k in (("mean", 0), ("var", 1), ("invstd", 2)):
        a = ws[mean0 + k * seg: mean0 + k * seg + nchan * 4].view(torch.float32)
        b = ws_ref[mean0 + k * seg: mean0 + k * seg + nchan * 4].view(torch.float32)
        dd = torch.nonzero(a != b).flatten().tolist()
        print(f"   {name}: {len(dd)} of {nchan} channels differ; first {dd[:24]}"
              f" (level starts at {[sum(cl[:i]) for i in range(6)]})")
    if found >= 2:
        break
print("bisect: mismatches found", found, "of", it + 1)
