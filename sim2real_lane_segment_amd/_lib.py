"""ctypes binding of the C ABI in include/rln.h (librln.so, hand-written HIP for gfx950).

The library is built in-tree by ``sim2real_lane_segment_amd/csrc/build.sh`` (driven by
``__graft_entry__.build()``).  There is NO fallback: if the shared object is missing or a call
fails, a ``RuntimeError`` is raised -- the product path never routes through PyTorch operators
or the CPU oracle.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, byref, c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_uint8,
                    c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librln.so")
RLN_MAX_BLOCKS = 8

T_PARAM, T_RUNNING_MEAN, T_RUNNING_VAR, T_NUM_BATCHES = 0, 1, 2, 3


class RlnConfig(Structure):
    _fields_ = [("in_channels", c_int), ("n_down", c_int), ("down_blocks", c_int * RLN_MAX_BLOCKS),
                ("n_up", c_int), ("up_blocks", c_int * RLN_MAX_BLOCKS), ("bottleneck_layers", c_int),
                ("growth_rate", c_int), ("first_conv_channels", c_int), ("n_classes", c_int),
                ("temperature", c_float), ("bn_eps", c_float), ("bn_momentum", c_float), ("drop_p", c_float)]


_lib = None

_PROTOS = {
    "rln_last_error": (c_char_p, []),
    "rln_version": (c_int, []),
    "rln_create": (c_int, [POINTER(RlnConfig), POINTER(c_void_p)]),
    "rln_destroy": (None, [c_void_p]),
    "rln_num_tensors": (c_int, [c_void_p]),
    "rln_param_count": (c_int64, [c_void_p]),
    "rln_bnstat_count": (c_int64, [c_void_p]),
    "rln_nbt_count": (c_int64, [c_void_p]),
    "rln_tensor_info": (c_int, [c_void_p, c_int, c_char_p, c_int, POINTER(c_int), POINTER(c_int64), POINTER(c_int),
                                POINTER(c_int64)]),
    "rln_feature_channels": (c_int, [c_void_p]),
    "rln_num_dropouts": (c_int, [c_void_p]),
    "rln_dropout_channels": (c_int64, [c_void_p, POINTER(c_int)]),
    "rln_bind_params": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rln_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int, c_int]),
    "rln_set_workspace": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int]),
    "rln_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_uint64, c_void_p, c_void_p,
                            c_int, c_void_p]),
    "rln_classifier_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "rln_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                         c_void_p]),
    "rln_entropy_loss": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "rln_set_output_grad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int]),
    "rln_sgd_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_int, c_float,
                             c_void_p]),
    "rln_backward_segments": (c_int, [c_void_p]),
    "rln_backward_segment_range": (c_int, [c_void_p, c_int, POINTER(c_int64), POINTER(c_int64)]),
    "rln_backward": (c_int, [c_void_p, c_float, c_int, c_int, c_void_p]),
    "rln_backward_scaled": (c_int, [c_void_p, c_float, c_void_p, c_int, c_int, c_void_p]),
    "rln_bind_grads": (c_int, [c_void_p, c_void_p]),
    "rln_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                               c_float, c_int, c_float, c_void_p]),
    "rln_op_conv_bnrelu": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                   c_void_p, c_void_p, c_size_t, c_void_p]),
    "rln_op_dense3_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p,
                                  c_size_t, c_void_p]),
    "rln_op_dense3_fwd_pair": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int] + [c_void_p] * 12 +
                               [c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rln_op_td_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                              c_void_p, c_size_t, c_void_p]),
    "rln_op_tu_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                              c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "rln_op_tu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                              c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "rln_op_td_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                              c_void_p, c_size_t, c_void_p]),
    "rln_op_fc_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                              c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "rln_op_fc_wgrad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p,
                                c_size_t, c_void_p]),
    "rln_set_dense_arith": (c_int, [c_void_p, c_int, c_int, c_int, c_int]),
    "rln_set_wgrad_parts": (c_int, [c_void_p, c_int]),
    "rln_get_wgrad_parts": (c_int, [c_void_p]),
    "rln_set_storage": (c_int, [c_void_p, c_int]),
    "rln_get_storage": (c_int, [c_void_p]),
    "rln_set_eval_cache": (c_int, [c_void_p, c_int]),
    "rln_op_convt": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                             c_int, c_int, c_void_p]),
    "rln_profile_enable": (c_int, [c_void_p, c_int]),
    "rln_profile_num_classes": (c_int, []),
    "rln_profile_class_name": (c_char_p, [c_int]),
    "rln_profile_entries": (ctypes.c_int64, [c_void_p, POINTER(c_int), POINTER(ctypes.c_double), POINTER(ctypes.c_double),
                            POINTER(ctypes.c_double), c_int64]),
    "rln_profile_read": (c_int, [c_void_p, POINTER(ctypes.c_double), POINTER(ctypes.c_double),
                                 POINTER(ctypes.c_double), POINTER(c_int64)]),
    "rln_debug_read_stamps": (c_int, [POINTER(ctypes.c_uint64)]),
    "rln_op_conv_act": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rln_op_bn_affine": (c_int, [c_void_p, c_int, ctypes.c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "rln_op_bn_drop_maxpool": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                       c_void_p, c_void_p]),
    "rln_op_bn_drop_upsample2": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p]),
    "rln_op_softmax_channels": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rln_op_dropout_mask": (c_int, [c_void_p, c_int64, c_float, c_uint64, c_void_p]),
    "rln_op_scaled_softmax": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p]),
    "rln_preprocess_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, POINTER(c_float),
                                  POINTER(c_float), c_void_p, c_void_p, c_void_p]),
    "rln_augment_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, POINTER(c_float),
                               POINTER(c_float), c_void_p, c_void_p, c_void_p, c_void_p]),
    "rln_overlay_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, ctypes.c_uint,
                               c_void_p, c_void_p, c_void_p]),
    "rln_op_classifier": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p, c_int,
                                  c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS.keys())


def lib():
    """Loads librln.so once; raises RuntimeError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"HIP library {LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or sim2real_lane_segment_amd/csrc/build.sh). There is no CPU/PyTorch fallback for this path.")
    handle = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(handle, name)  # AttributeError here = header/ABI drift, must be loud
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


def check(code, what=""):
    if code != 0:
        msg = lib().rln_last_error()
        raise RuntimeError(f"rln call failed ({code}) {what}: {msg.decode() if msg else ''}")


def make_config(in_channels, down_blocks, up_blocks, bottleneck_layers, growth_rate, first_conv, n_classes,
                temperature=0.05, bn_eps=1e-5, bn_momentum=0.1, drop_p=0.2):
    if len(down_blocks) > RLN_MAX_BLOCKS or len(up_blocks) > RLN_MAX_BLOCKS:
        raise ValueError(f"at most {RLN_MAX_BLOCKS} blocks per path")
    cfg = RlnConfig()
    cfg.in_channels = in_channels
    cfg.n_down = len(down_blocks)
    cfg.n_up = len(up_blocks)
    for i, v in enumerate(down_blocks):
        cfg.down_blocks[i] = int(v)
    for i, v in enumerate(up_blocks):
        cfg.up_blocks[i] = int(v)
    cfg.bottleneck_layers = bottleneck_layers
    cfg.growth_rate = growth_rate
    cfg.first_conv_channels = first_conv
    cfg.n_classes = n_classes
    cfg.temperature = temperature
    cfg.bn_eps = bn_eps
    cfg.bn_momentum = bn_momentum
    cfg.drop_p = drop_p
    return cfg
