"""CPU oracle of the legacy EncDecNet forward (TEST INFRASTRUCTURE ONLY; see fcdensenet_oracle.py for the rules).

Restates rightLaneNetwork/models/EncDecNet.py with plain PyTorch CPU operators on a state_dict with the reference's
keys; Dropout masks are explicit inputs.  Pinned by tests/golden/encdec_*.npz generated from the reference class."""
import torch
import torch.nn.functional as F

ACTS = {
    'relu': lambda x, st: F.relu(x),
    'prelu': lambda x, st: F.prelu(x, st['activation.weight']),
    'leakyRelu': lambda x, st: F.leaky_relu(x, 0.01),
    'sigmoid': lambda x, st: torch.sigmoid(x),
    'tanh': lambda x, st: torch.tanh(x),
    'none': lambda x, st: x,
}


def _conv_block(st, prefix, x, k, act, training, mask, new_stats, bnorm=True):
    """Conv.forward (EncDecNet.py:28-37): conv -> activation -> batch norm -> dropout."""
    x = F.conv2d(x, st[prefix + '.conv.weight'], st[prefix + '.conv.bias'], padding=k // 2)
    x = act(x, st)
    if bnorm:
        rm, rv = st[prefix + '.bn.running_mean'].clone(), st[prefix + '.bn.running_var'].clone()
        x = F.batch_norm(x, rm, rv, st[prefix + '.bn.weight'], st[prefix + '.bn.bias'], training, 0.1, 1e-5)
        if training and new_stats is not None:
            new_stats[prefix + '.bn.running_mean'] = rm
            new_stats[prefix + '.bn.running_var'] = rv
    if training and mask is not None:
        x = x * mask
    return x


def forward(st, x, n_levels, kernel_size, n_lin='relu', bnorm=True, training=False, masks=None, new_stats=None):
    """EncDecNet.forward (EncDecNet.py:100-112)."""
    act = ACTS[n_lin]
    i = 0
    for l in range(n_levels):
        x = _conv_block(st, f'encoders.{l}', x, kernel_size, act, training, masks[i] if masks else None, new_stats, bnorm)
        x = F.max_pool2d(x, kernel_size, stride=2, padding=kernel_size // 2)
        i += 1
    for l in range(n_levels):
        x = _conv_block(st, f'decoders.{l}', x, kernel_size, act, training, masks[i] if masks else None, new_stats, bnorm)
        x = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
        i += 1
    x = F.conv2d(x, st['classifier.conv.weight'], st['classifier.conv.bias'])
    return F.softmax(x, dim=-3)


def make_masks(shapes, p, seed):
    gen = torch.Generator().manual_seed(seed)
    return [(torch.rand(s, generator=gen) >= p).float() / (1.0 - p) for s in shapes]


def mask_shapes(n, h, w, n_feat, n_levels, k):
    shapes, c = [], n_feat
    for _ in range(n_levels):
        shapes.append((n, c, h, w))
        h = (h + 2 * (k // 2) - k) // 2 + 1
        w = (w + 2 * (k // 2) - k) // 2 + 1
        c *= 2
    c //= 2
    for _ in range(n_levels):
        shapes.append((n, c, h, w))
        h, w, c = 2 * h, 2 * w, c // 2
    return shapes
