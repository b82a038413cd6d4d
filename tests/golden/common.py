"""Seeded input builders and fixture helpers shared by gen_golden.py and the tests
(so both sides of every parity check see bit-identical inputs)."""
import numpy as np
import torch


def synth_batch(n, h, w, n_classes, seed):
    """x ~ N(0,1) fp32 [n,3,h,w]; labels uniform int64 [n,h,w]."""
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, h, w, generator=gen)
    y = torch.randint(0, n_classes, (n, h, w), generator=gen)
    return x, y


def sample_idx(numel, k, seed):
    gen = torch.Generator().manual_seed(seed)
    if numel <= k:
        return torch.arange(numel)
    return torch.randperm(numel, generator=gen)[:k]


def pack_masks(mask, n_classes):
    """2 bits per pixel, 4 pixels per byte."""
    assert n_classes <= 4
    m = mask.reshape(-1).to(torch.uint8).numpy()
    pad = (-len(m)) % 4
    m = np.concatenate([m, np.zeros(pad, np.uint8)])
    m = m.reshape(-1, 4)
    return (m[:, 0] | (m[:, 1] << 2) | (m[:, 2] << 4) | (m[:, 3] << 6)).astype(np.uint8)


def unpack_masks(packed, numel):
    p = np.asarray(packed, np.uint8)
    m = np.stack([p & 3, (p >> 2) & 3, (p >> 4) & 3, (p >> 6) & 3], 1).reshape(-1)
    return torch.from_numpy(m[:numel].astype(np.int64))


def cfg_to_arrays(cfg):
    return dict(cfg_down=np.array(cfg.down_blocks), cfg_up=np.array(cfg.up_blocks),
                cfg_misc=np.array([cfg.in_channels, cfg.bottleneck_layers, cfg.growth_rate,
                                   cfg.out_chans_first_conv, cfg.n_classes]))


def cfg_from_arrays(z, cfg_cls):
    misc = [int(v) for v in z["cfg_misc"]]
    return cfg_cls(in_channels=misc[0], down_blocks=tuple(int(v) for v in z["cfg_down"]),
                   up_blocks=tuple(int(v) for v in z["cfg_up"]), bottleneck_layers=misc[1],
                   growth_rate=misc[2], out_chans_first_conv=misc[3], n_classes=misc[4])
