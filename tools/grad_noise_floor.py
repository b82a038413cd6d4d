"""fp32-vs-fp64 noise floor of the path's gradients, measured on the CPU oracle (same algorithm, same inputs, same
Dropout2d masks; only the arithmetic width differs).  A 60-100 layer pre-activation DenseNet flips individual ReLU /
max-pool decisions under fp32 rounding, so two CORRECT fp32 implementations with different summation orders differ by
about this much; the GPU parity tests hold every gradient tensor to max(bar, FACTOR x its own noise floor) and read this
script's stored output (tests/golden/grad_noise_floor.json).

usage: python tools/grad_noise_floor.py   (CPU, ~2 minutes; writes tests/golden/grad_noise_floor.json)"""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import fcdensenet_oracle as O  # noqa: E402
from tests.golden.common import synth_batch  # noqa: E402


def one(cfg, n, h, w, seed_state, seed_x, seed_drop):
    st = O.init_state(cfg, seed_state)
    x, y = synth_batch(n, h, w, cfg.n_classes, seed_x)
    scales = O.make_drop_scales(cfg, n, seed_drop)
    _, _, g32, p32 = O.train_step(O.TrainState({k: v.clone() for k, v in st.items()}), x, y, cfg, scales,
                                  apply_update=False)
    st64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in st.items()}
    _, _, g64, p64 = O.train_step(O.TrainState(st64), x.double(), y, cfg, [s.double() for s in scales],
                                  apply_update=False)
    out = {}
    for k in g32:
        ref = g64[k]
        floor = 1e-5 * ref.numel() ** 0.5
        out[k] = float((g32[k].double() - ref).norm()) / max(float(ref.norm()), floor)
    vals = sorted(out.values())
    summary = {"tensors": len(vals), "median": vals[len(vals) // 2], "p90": vals[int(0.9 * len(vals))], "max": vals[-1],
               "probs_max_abs": float((p32.double() - p64).abs().max())}
    return out, summary


def main():
    from sim2real_lane_segment_amd.models.FCDenseNet import tiramisu as T
    res = {}
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    # FCDenseNet67, the golden training fixture's geometry (tests/golden/gen_golden.py: fcd67_train_120x160)
    cfg = O.fcdensenet67_config(4)
    per, summ = one(cfg, 2, 120, 160, 700, 701, 702)
    res["fcd67_2x120x160"] = {"summary": summ, "per_tensor": per}
    print("fcd67", summ, flush=True)
    for variant in ("57", "103"):  # tests/test_gpu_parity.py::test_named_variants_train_step_vs_oracle
        down, up, bott, growth = T._VARIANTS[variant]
        cfg = O.NetConfig(down_blocks=down, up_blocks=up, bottleneck_layers=bott, growth_rate=growth, n_classes=4)
        per, summ = one(cfg, 2, 64, 96, 21, 22, 23)
        res[f"fcd{variant}_2x64x96"] = {"summary": summ, "per_tensor": per}
        print(variant, summ, flush=True)
    with open(os.path.join(REPO, "tests", "golden", "grad_noise_floor.json"), "w") as f:
        json.dump(res, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
