"""CPU-only checks of the C ABI and the host logic: the library loads, exports every symbol that
include/rln.h declares, and describes the reference's state_dict exactly (no compute calls: no GPU here)."""
import ctypes
import os
import re

import pytest
import torch

from oracle import fcdensenet_oracle as O
from sim2real_lane_segment_amd import _lib
from sim2real_lane_segment_amd.engine import Engine, NetSpec

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(REPO, "include", "rln.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rln_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rln.h but not exported by librln.so"
    for n in _lib.EXPORTED_SYMBOLS:
        assert n in names, f"{n} bound by ctypes but not declared in include/rln.h"
    assert lib.rln_version() >= 1


def spec_of(cfg):
    return NetSpec(in_channels=cfg.in_channels, down_blocks=cfg.down_blocks, up_blocks=cfg.up_blocks,
                   bottleneck_layers=cfg.bottleneck_layers, growth_rate=cfg.growth_rate,
                   out_chans_first_conv=cfg.out_chans_first_conv, n_classes=cfg.n_classes)


@pytest.mark.parametrize("cfg", [O.fcdensenet67_config(4),
                                 O.NetConfig(down_blocks=(4, 4, 4, 4, 4), up_blocks=(4, 4, 4, 4, 4), bottleneck_layers=4,
                                             growth_rate=12, n_classes=2),
                                 O.NetConfig(down_blocks=(4, 5, 7, 10, 12), up_blocks=(12, 10, 7, 5, 4),
                                             bottleneck_layers=15, n_classes=12),
                                 O.NetConfig(down_blocks=(2, 2), up_blocks=(2, 2), bottleneck_layers=2, growth_rate=4,
                                             out_chans_first_conv=8, n_classes=4)])
def test_plan_matches_reference_state_dict(cfg):
    eng = Engine(spec_of(cfg), device="cpu")
    spec = O.state_spec(cfg)
    got = {m.name: m for m in eng.metas}
    assert set(got) == {k for k, _ in spec}
    for k, shape in spec:
        assert tuple(got[k].shape) == tuple(shape), k
    n_param = sum(int(torch.tensor(s).prod()) for k, s in spec if O.is_param(k))
    assert eng.n_param == n_param
    # parameter arena: dense, non-overlapping, in forward execution order (first conv first, classifier last)
    params = sorted((m for m in eng.metas if m.kind == _lib.T_PARAM), key=lambda m: m.offset)
    off = 0
    for m in params:
        assert m.offset == off, m.name
        off += m.numel
    assert off == eng.n_param
    assert params[0].name == "featureExtractor.firstconv.weight"
    assert params[-1].name == "classifier.finalConv.bias"
    assert eng.feature_channels == O.feature_channels(cfg)
    assert eng.drop_channels == O.dropout_channels(cfg)
    # backward segments tile the arena back to front without gaps
    assert eng.seg_ranges[0][1] == eng.n_param
    for (b0, e0), (b1, e1) in zip(eng.seg_ranges[:-1], eng.seg_ranges[1:]):
        assert e1 == b0 and b1 < e1
    assert eng.seg_ranges[-1][0] == 0


def test_fcd67_counts():
    eng = Engine(spec_of(O.fcdensenet67_config(4)), device="cpu")
    assert len(eng.metas) == 434 and eng.n_param == 3461220 and eng.n_bnstat == 39680 and eng.n_nbt == 60


def test_error_conventions():
    lib = _lib.lib()
    cfg = _lib.make_config(3, (5, 5), (5,), 5, 16, 48, 4)  # n_down != n_up
    h = ctypes.c_void_p()
    assert lib.rln_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"n_down" in lib.rln_last_error()
    eng = Engine(spec_of(O.fcdensenet67_config(4)), device="cpu")
    need = lib.rln_workspace_bytes(eng.ctx, 2, 120, 160, 1)
    assert need > 2 * 288 * 120 * 160 * 4 * 2
    assert lib.rln_workspace_bytes(eng.ctx, 2, 120, 160, 0) < need
    # five floor-poolings do not fit into 16x16 ("Output size is too small" in the reference)
    assert lib.rln_workspace_bytes(eng.ctx, 1, 16, 16, 0) == 0
    assert lib.rln_set_workspace(eng.ctx, ctypes.c_void_p(256), 1 << 40, 1, 16, 16, 0) == -1
    assert lib.rln_set_workspace(eng.ctx, None, 0, 2, 120, 160, 1) == -3
    with pytest.raises(RuntimeError):
        eng.forward(torch.zeros(1, 3, 64, 64), training=False)  # CPU engine: no fallback, loud failure
