"""ctypes binding of libpfn_hip.so (the C ABI declared in include/pfn_hip.h).

There is NO CPU fallback: if the library is missing or a call fails, a RuntimeError is raised.
PyTorch is used only for device memory, streams and autograd plumbing; every tensor crosses the
boundary as a raw device pointer + sizes.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpfn_hip.so")

# every symbol include/pfn_hip.h declares (checked by tests/test_abi.py without a GPU)
SYMBOLS = (
    "pfn_abi_version", "pfn_last_error", "pfn_padded_ld",
    "pfn_graph_workspace_bytes", "pfn_graph_build", "pfn_graph_info", "pfn_graph_segments", "pfn_graph_segments_async", "pfn_graph_poison_if_bad", "pfn_graph_export_edges",
    "pfn_mpn_num_params", "pfn_mpn_workspace_bytes", "pfn_mpn_forward", "pfn_mpn_backward", "pfn_mpn_mse_tail_ok", "pfn_mpn_backward_mse", "pfn_mpn_backward_masked_l2", "pfn_mpn_export_gates",
    "pfn_edge_aggr_workspace_bytes", "pfn_edge_aggr_forward", "pfn_edge_aggr_backward",
    "pfn_tag_conv_workspace_bytes", "pfn_tag_conv_forward", "pfn_tag_conv_backward",
    "pfn_scatter_add", "pfn_pad_rows", "pfn_mse_loss", "pfn_masked_l2_loss", "pfn_power_imbalance", "pfn_dropout_mask", "pfn_adamw_step", "pfn_adamw_step_dev", "pfn_adamw_step_guarded",
    "pfn_profile_enable", "pfn_profile_report",
)


class MpnConfig(C.Structure):
    """struct pfn_mpn_config."""
    _fields_ = [("nfeature_dim", C.c_int32), ("efeature_dim", C.c_int32), ("output_dim", C.c_int32),
                ("hidden_dim", C.c_int32), ("n_gnn_layers", C.c_int32), ("K", C.c_int32),
                ("dropout_rate", C.c_float), ("training", C.c_int32), ("need_backward", C.c_int32)]


_lib = None
ABI_VERSION = 8


def load() -> C.CDLL:
    """Load (once) and type the library.  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            f"or make -C poweflownet_amd/csrc).  poweflownet_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    p, i64, i32, sz, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_size_t, C.c_float
    cfgp = C.POINTER(MpnConfig)
    sig = {
        "pfn_abi_version": (C.c_int, []),
        "pfn_last_error": (C.c_char_p, []),
        "pfn_padded_ld": (i64, [i64]),
        "pfn_graph_workspace_bytes": (sz, [i64, i64]),
        "pfn_graph_build": (C.c_int, [p, i64, i64, i32, p, sz, p]),
        "pfn_graph_info": (C.c_int, [p, i64, i64, C.POINTER(C.c_int32), C.POINTER(C.c_int64), p]),
        "pfn_graph_segments": (C.c_int, [p, i64, i64, i64, C.POINTER(C.c_int32), p]),
        "pfn_graph_segments_async": (C.c_int, [p, i64, i64, i64, p]),
        "pfn_graph_poison_if_bad": (C.c_int, [p, i64, i64, p, i64, p]),
        "pfn_graph_export_edges": (C.c_int, [p, i64, i64, p, p]),
        "pfn_mpn_num_params": (C.c_int, [cfgp]),
        "pfn_mpn_workspace_bytes": (sz, [cfgp, i64, i64]),
        "pfn_mpn_forward": (C.c_int, [cfgp, p, i64, i64, p, p, p, i32, p, p, p, sz, p, i64, p]),
        "pfn_mpn_backward": (C.c_int, [cfgp, p, i64, i64, p, p, p, p, i32, p, p, p, p, p, sz, i64, p]),
        "pfn_mpn_mse_tail_ok": (C.c_int, [cfgp, i64, i64, i64]),
        "pfn_mpn_backward_mse": (C.c_int, [cfgp, p, i64, i64, p, p, p, p, p, p, p, p, p, p, sz, p, sz, i64, p]),
        "pfn_mpn_backward_masked_l2": (C.c_int, [cfgp, p, i64, i64, p, p, p, p, p, i32, f32, p, p, p, p, p, sz, p, sz, i64, p]),
        "pfn_mpn_export_gates": (C.c_int, [cfgp, p, i64, i64, p, p, p, sz, i64, C.c_int32, C.c_int32, p, p]),
        "pfn_edge_aggr_workspace_bytes": (sz, [i64, i64, i32, i32, i32, i32]),
        "pfn_edge_aggr_forward": (C.c_int, [p, i64, i64, i32, i32, i32, i32, p, i64, p, p, p, p, p, p, i64, p, sz, p]),
        "pfn_edge_aggr_backward": (C.c_int, [p, i64, i64, i32, i32, i32, i32, p, i64, p, p, p, p, p, p, i64, p, i64, p,
                                             p, p, p, p, p, sz, p]),
        "pfn_tag_conv_workspace_bytes": (sz, [i64, i64, i32, i32, i32]),
        "pfn_tag_conv_forward": (C.c_int, [p, i64, i64, i32, i32, i32, p, i64, p, p, p, i64, p, sz, i64, p]),
        "pfn_tag_conv_backward": (C.c_int, [p, i64, i64, i32, i32, i32, p, i64, p, p, i64, p, i64, p, p, p, sz, i64, p]),
        "pfn_scatter_add": (C.c_int, [p, i64, i64, p, p, i64, p]),
        "pfn_pad_rows": (C.c_int, [p, i64, p, i64, i64, i64, p]),
        "pfn_mse_loss": (C.c_int, [p, p, i64, p, p, p, sz, p]),
        "pfn_masked_l2_loss": (C.c_int, [p, p, p, C.c_int, i64, C.c_int, C.c_float, p, p, p, sz, p]),
        "pfn_power_imbalance": (C.c_int, [p, i64, i64, p, p, p, p, p, p, p, sz, p]),
        "pfn_dropout_mask": (C.c_int, [p, C.c_int32, i64, i64, f32, p, p]),
        "pfn_adamw_step": (C.c_int, [p, p, p, p, i64, f32, f32, f32, f32, f32, p, p]),
        "pfn_adamw_step_dev": (C.c_int, [p, p, p, p, i64, p, p, p]),
        "pfn_adamw_step_guarded": (C.c_int, [p, p, p, p, i64, p, p, p, p]),
        "pfn_profile_enable": (C.c_int, [i32]),
        "pfn_profile_report": (C.c_int, [C.c_char_p, sz, i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    if lib.pfn_abi_version() != ABI_VERSION:
        raise RuntimeError("libpfn_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().pfn_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def stream_ptr() -> int:
    """hipStream_t of torch's current stream on the current device."""
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int | None:
    return None if t is None else t.data_ptr()


def ptr_table(tensors):
    """Host array of device pointers (const float* const*)."""
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def require_device(*tensors, what="tensor") -> torch.device:
    """The reference surfaces bad inputs as RuntimeError (SURVEY 8b 'errors'); so do we -- and a CPU tensor is
    one of them, because this package has no CPU path."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(f"poweflownet_amd: {what} must live on a HIP device (got {t.device}); "
                               f"there is no CPU fallback -- the CPU oracle lives under oracle/ for tests only")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"poweflownet_amd: tensors on different devices ({dev} vs {t.device})")
    return dev


def f32c(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise RuntimeError(f"poweflownet_amd: {what} must be float32 (got {t.dtype})")
    return t if t.is_contiguous() else t.contiguous()


def profile_enable(on: bool) -> None:
    check(load().pfn_profile_enable(1 if on else 0), "pfn_profile_enable")


def profile_report(reset: bool = True) -> dict:
    """{kernel class: {count, ms, bytes, flops}} from the HIP-event brackets (synchronises)."""
    import json
    buf = C.create_string_buffer(1 << 16)
    check(load().pfn_profile_report(buf, len(buf), 1 if reset else 0), "pfn_profile_report")
    return json.loads(buf.value.decode())
