"""Legacy EncDecNet: oracle vs golden vectors from the reference class (CPU), API surface (CPU),
HIP forward vs golden (-m gpu)."""
import os

import numpy as np
import pytest
import torch

from oracle import encdecnet_oracle as E
from tests.golden.common import synth_batch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["encdec_k3_relu_24x40", "encdec_k7_leaky_40x56", "encdec_k3_prelu_30x34"]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def state_of(z):
    return {k[6:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith("state/")}


@pytest.mark.parametrize("name", CASES)
def test_oracle_vs_reference(name):
    z = load(name)
    st = state_of(z)
    n, h, w, seed = int(z["n"]), int(z["h"]), int(z["w"]), int(z["seed"])
    nf, nl, k, lin = int(z["n_feat"]), int(z["n_levels"]), int(z["k"]), str(z["n_lin"])
    x, _ = synth_batch(n, h, w, 2, seed + 2)
    with torch.no_grad():
        out = E.forward(st, x, nl, k, lin)
        np.testing.assert_allclose(out.numpy(), z["eval_out"], atol=2e-6)
        masks = E.make_masks(E.mask_shapes(n, h, w, nf, nl, k), 0.3, seed + 3)
        new = {}
        out_t = E.forward(st, x, nl, k, lin, training=True, masks=masks, new_stats=new)
    np.testing.assert_allclose(out_t.numpy(), z["train_out"], atol=2e-6)
    for key, v in new.items():
        np.testing.assert_allclose(v.numpy(), z["buf1/" + key], rtol=1e-5, atol=1e-6)


def test_api_surface_cpu():
    from sim2real_lane_segment_amd.models.EncDecNet import Conv, EncDecNet, activationTypes
    assert activationTypes == ['relu', 'prelu', 'leakyRelu', 'sigmoid', 'tanh', 'none']
    net = EncDecNet(64, 3, 7)
    assert net.getNParams() == 7237570          # the number the reference's __main__ prints
    assert EncDecNet(64, 3, 3).getNParams() == 1331650
    z = load("encdec_k3_prelu_30x34")
    small = EncDecNet(int(z["n_feat"]), int(z["n_levels"]), int(z["k"]), str(z["n_lin"]))
    assert sorted(small.state_dict().keys()) == sorted(k[6:] for k in z.files if k.startswith("state/"))
    small.load_state_dict(state_of(z))
    for bad in [dict(nFeat=0, nLevels=1), dict(nFeat=4, nLevels=0), dict(nFeat=4, nLevels=1, nLinType="gelu")]:
        with pytest.raises(ValueError):
            EncDecNet(**bad)
    with pytest.raises(ValueError):
        Conv(3, 4, dropOut=1.0)
    with pytest.raises(RuntimeError, match="GPU only"):
        small(torch.zeros(1, 3, 32, 32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_forward_vs_reference(name):
    from sim2real_lane_segment_amd.models.EncDecNet import EncDecNet
    z = load(name)
    n, h, w, seed = int(z["n"]), int(z["h"]), int(z["w"]), int(z["seed"])
    nf, nl, k, lin = int(z["n_feat"]), int(z["n_levels"]), int(z["k"]), str(z["n_lin"])
    net = EncDecNet(nf, nl, k, lin)
    net.load_state_dict(state_of(z))
    net = net.cuda()
    x, _ = synth_batch(n, h, w, 2, seed + 2)
    net.eval()
    out = net(x.cuda())
    np.testing.assert_allclose(out.cpu().numpy(), z["eval_out"], atol=1e-4)
    assert np.array_equal(out.argmax(1).cpu().numpy()[np.abs(z["eval_out"][:, 0] - 0.5) > 1e-3],
                          z["eval_out"].argmax(1)[np.abs(z["eval_out"][:, 0] - 0.5) > 1e-3])
    masks = E.make_masks(E.mask_shapes(n, h, w, nf, nl, k), 0.3, seed + 3)
    net.train()
    out_t = net(x.cuda(), drop_masks=masks)
    np.testing.assert_allclose(out_t.cpu().numpy(), z["train_out"], atol=1e-4)
    sd = net.state_dict()
    for key in z.files:
        if key.startswith("buf1/"):
            np.testing.assert_allclose(sd[key[5:]].cpu().numpy(), z[key], rtol=1e-4, atol=1e-5, err_msg=key)
    # device-generated masks: runs, output is a distribution over the 2 classes
    out_r = net(x.cuda())
    assert torch.allclose(out_r.sum(1), torch.ones_like(out_r[:, 0]), atol=1e-5)


# ---- the sizes the reference itself builds (models/EncDecNet.py:119-130) ---------------------------------
REF_CASES = [("encdec_ref_64_3_3_120x160", 3), ("encdec_ref_64_3_7_120x160", 7)]


def _ref_inputs(z):
    from tests.golden.common import synth_batch
    x, _ = synth_batch(1, 120, 160, 2, int(z["seed"]) + 2)
    return {"ones": torch.ones(1, 3, 120, 160), "rand": x}


def _check_ref_output(z, tag, y, atol):
    from tests.golden.common import sample_idx, unpack_masks
    idx = sample_idx(y.numel(), 4096, 77)
    np.testing.assert_allclose(y.reshape(-1)[idx].numpy(), z[tag + "_samp"], atol=atol)
    mask = y.argmax(1).reshape(-1)
    ref = unpack_masks(z[tag + "_mask"], mask.numel())
    near = set(int(i) for i in z[tag + "_near"])  # pixels whose two class probabilities differ by < 1e-4
    diff = torch.nonzero(mask != ref).reshape(-1).tolist()
    assert all(d in near for d in diff), f"{len(diff)} argmax flips, some outside the near-tie list"
    return len(diff)


@pytest.mark.parametrize("name,k", REF_CASES)
def test_oracle_vs_reference_at_reference_size(name, k):
    from tests.golden.common import det_state
    from sim2real_lane_segment_amd.models.EncDecNet import EncDecNet
    z = load(name)
    net = EncDecNet(64, 3, k)  # parameter container only (CPU): shapes for the deterministic weights
    assert net.getNParams() == (7237570 if k == 7 else 1331650)
    st = det_state({key: v.shape for key, v in net.state_dict().items()}, int(z["seed"]))
    with torch.no_grad():
        for tag, x in _ref_inputs(z).items():
            _check_ref_output(z, tag, E.forward(st, x, 3, k, "relu"), 5e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name,k", REF_CASES)
def test_hip_forward_at_reference_size(name, k):
    from tests.golden.common import det_state
    from sim2real_lane_segment_amd.models.EncDecNet import EncDecNet
    z = load(name)
    net = EncDecNet(64, 3, k)
    net.load_state_dict(det_state({key: v.shape for key, v in net.state_dict().items()}, int(z["seed"])))
    net = net.cuda().eval()
    for tag, x in _ref_inputs(z).items():
        out = net(x.cuda())
        assert out.shape == (1, 2, 120, 160)
        flips = _check_ref_output(z, tag, out.cpu(), 1e-4)
        print(f"[{name}/{tag}] argmax flips vs the reference: {flips}")
