"""ctypes binding of libindm_hip.so (the C ABI in include/indm_hip.h).

This is the ONLY compute backend of the package: every function here takes torch tensors that
live in MI355X HBM, hands raw device pointers + the current HIP stream to a hand-written gfx950
kernel and returns torch tensors.  There is no CPU / eager fallback -- a missing library or a
non-GPU tensor raises immediately (tests/ may monkeypatch this module with a CPU emulation to
exercise host logic; nothing in the package does).

PyTorch's role is plumbing: device memory (torch.empty), streams, autograd bookkeeping.
"""
import ctypes
import os

import torch

_LIB_PATH = os.environ.get("INDM_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libindm_hip.so")
_lib = None

c_f = ctypes.c_float
c_i = ctypes.c_int
c_l = ctypes.c_long
c_p = ctypes.c_void_p
c_u64 = ctypes.c_ulonglong

ERRORS = {-1: "INDM_E_SHAPE", -2: "INDM_E_DTYPE", -3: "INDM_E_LAUNCH", -4: "INDM_E_ARG"}

# name -> argtypes (restype is int unless listed in _LONG_RET)
_SIGS = {
    "indm_gemm_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_i, c_l, c_l, c_l, c_f, c_p, c_p,
                      c_i, c_p, c_l, c_l, c_f, c_p, c_l, c_p],
    "indm_conv2d_fwd_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p,
                            c_p, c_p, c_f, c_p, c_l, c_p],
    "indm_conv2d_dgrad_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_f,
                              c_p, c_l, c_p],
    "indm_conv2d_wgrad_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_f,
                              c_p, c_l, c_p],
    "indm_gemm_bf16": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_i, c_l, c_l, c_l, c_f, c_p, c_p,
                       c_i, c_p, c_l, c_l, c_f, c_p, c_l, c_p],
    "indm_conv2d_fwd_bf16": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p,
                             c_p, c_p, c_f, c_p, c_l, c_p],
    "indm_conv2d_dgrad_bf16": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_f,
                               c_p, c_l, c_p],
    "indm_conv2d_wgrad_bf16": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_f,
                               c_p, c_l, c_p],
    "indm_groupnorm_ws_floats": [c_i, c_i, c_i, c_i],
    "indm_groupnorm_fused": [c_i, c_i, c_i],
    "indm_groupnorm_fwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_u64, c_p, c_p, c_p],
    "indm_groupnorm_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_u64,
                               c_p, c_p, c_p],
    "indm_softmax_fwd_f32": [c_p, c_p, c_l, c_i, c_f, c_p],
    "indm_softmax_bwd_f32": [c_p, c_p, c_p, c_l, c_i, c_f, c_p],
    "indm_colreduce_ws_floats": [c_l, c_i],
    "indm_colreduce_f32": [c_p, c_p, c_p, c_l, c_i, c_i, c_f, c_f, c_p, c_p],
    "indm_segreduce_ws_floats": [c_l, c_l, c_i],
    "indm_segreduce_f32": [c_p, c_p, c_l, c_l, c_i, c_f, c_p, c_p],
    "indm_igemm_last_variant": [],
    "indm_version": [],
    "indm_conv3x3_thin_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_f, c_p],
    "indm_conv3x3_thin_bf16": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_f, c_p],
    "indm_gemm_ires_bwd_act_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_l, c_l, c_l, c_p, c_p],
    "indm_gemm_gated_ires_bwd_act_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_l, c_l, c_l, c_p, c_p],
    "indm_gemm_sin_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_l, c_l, c_l, c_p, c_p, c_i, c_p],
    "indm_conv3x3_thin_act_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p],
    "indm_conv3x3_thin_act_bf16": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p],
    "indm_sincos_act_fwd_f32": [c_p, c_p, c_p, c_l, c_p],
    "indm_sincos_act_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_l, c_p],
    "indm_attn_fwd_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p],
    "indm_attn_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p],
    "indm_ires_chain_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "indm_ires_chain_bf16": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "indm_ires_bwd_act_ws_floats": [c_l, c_i],
    "indm_ires_bwd_act_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_l, c_p, c_p],
    "indm_act_fwd_f32": [c_p, c_p, c_l, c_i, c_p],
    "indm_act_bwd_f32": [c_p, c_p, c_p, c_l, c_i, c_p],
    "indm_axpby_f32": [c_p, c_p, c_p, c_f, c_f, c_f, c_l, c_p],
    "indm_mul_f32": [c_p, c_p, c_p, c_f, c_l, c_p],
    "indm_mul2_f32": [c_p, c_p, c_p, c_p, c_p, c_l, c_p],
    "indm_scale_rows_f32": [c_p, c_p, c_p, c_l, c_l, c_p],
    "indm_rowreduce_ws_floats": [c_l, c_l],
    "indm_rowreduce_f32": [c_p, c_p, c_p, c_l, c_l, c_i, c_f, c_p, c_p],
    "indm_upsample2x_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p],
    "indm_downsample2x_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p],
    "indm_squeeze2x_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "indm_unsqueeze2x_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "indm_nchw_nhwc_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "indm_split_cols_f32": [c_p, c_p, c_p, c_i, c_l, c_i, c_i, c_p],
    "indm_ires_chain_bf16w": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "indm_weightnorm_fwd_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_p],
    "indm_weightnorm_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p],
    "indm_lop_normalize_fwd_f32": [c_p, c_p, c_p, c_i, c_i, c_f, c_p],
    "indm_lop_normalize_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_p],
    "indm_multi_copy_f32": [c_p, c_i, c_l, c_p],
    "indm_lop_normalize_multi_fwd_f32": [c_p, c_i, c_i, c_l, c_f, c_p],
    "indm_lop_normalize_multi_bwd_f32": [c_p, c_p, c_i, c_i, c_l, c_f, c_p],
    "indm_philox_randn_f32": [c_p, c_l, c_p, c_p],
    "indm_coupling_fwd_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "indm_coupling_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "indm_sde_perturb_f32": [c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_p],
    "indm_dsm_ws_floats": [c_l, c_l],
    "indm_dsm_loss_fwd_f32": [c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_i, c_i, c_i, c_p, c_p],
    "indm_dsm_loss_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_i, c_i, c_i, c_p],
    "indm_affine_update_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_p],
    "indm_langevin_step_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_p],
    "indm_pc_step_em_f32": [c_p, c_p, c_p, c_p, c_p, ctypes.c_double, c_i, c_p, c_p, c_p, c_p, c_l, c_l, c_p],
    "indm_pc_step_rd_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_l, c_l, c_p],
    "indm_pc_step_langevin_f32": [c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_p],
    "indm_hutch_div_f32": [c_p, c_p, c_p, c_l, c_l, c_p, c_p],
    "indm_pc_step_ws_floats": [c_l, c_l],
    "indm_hutch_div_ws_floats": [c_l, c_l],
    "indm_sqnorm_ws_floats": [],
    "indm_sqnorm_f32": [c_p, c_l, c_p, c_p, c_p],
    "indm_adamw_ema_f32": [c_p, c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_f, c_i, c_f, c_p, c_f, c_f, c_p],
    "indm_tap_gather_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p],
    "indm_tap_spread_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p],
    "indm_tap_gather_acc_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_p],
    "indm_fixed_point_check_f32": [c_p, c_p, c_p, c_l, c_f, c_f, c_p, c_p],
    "indm_ema_update_f32": [c_p, c_p, c_l, c_f, c_p],
    "indm_slogdet_inv_f32": [c_p, c_i, c_p, c_p, c_p],
    "indm_bn_stats_f32": [c_p, c_p, c_p, c_p, c_l, c_i, c_f, c_p, c_p],
    "indm_bn_apply_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p],
    "indm_bn_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_p],
    "indm_upfirdn2d_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                           c_p],
    "indm_upfirdn2d_bf16": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                            c_p],
    "indm_fused_bias_act_f32": [c_p, c_p, c_p, c_p, c_l, c_l, c_i, c_i, c_i, c_f, c_f, c_p],
    "indm_iresblock_ws_floats": [c_i, c_i, c_i, c_i, c_i, c_i],
    "indm_iresblock_fwd_logdet": [c_p, c_p],
    "indm_iresblock_bwd": [c_p, c_p],
    "indm_iresblock_inverse": [c_p, c_f, c_f, c_p, c_p],
    "indm_f16_pack_f32": [c_p, c_p, c_l, c_i, c_p],
    "indm_f16_unpack_f32": [c_p, c_p, c_l, c_i, c_p],
    "indm_f16_thin_conv": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "indm_f16_tap_spread": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "indm_f16_chain_conv": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "indm_f16_conv_gemm": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "indm_f16_gnet": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "indm_f16_rows": [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "indm_f16_bwd_act_ok": [c_l, c_l],
    "indm_f16_bwd_act": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_p],
    "indm_f16_wgrad_ws_floats": [c_l, c_i],
    "indm_f16_thin_wgrad": [c_p, c_p, c_p, c_p, c_l, c_i, c_f, c_i, c_l, c_p, c_l, c_p],
    "indm_f16_wgrad_gemm": [c_p, c_p, c_p, c_p, c_l, c_f, c_i, c_p, c_l, c_p],
    "indm_h16_pack_weights": [c_p, c_i, c_i, c_p],
    "indm_h16_conv2d": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_f, c_i,
                        c_p, c_l, c_p, c_i, c_p],
    "indm_h16_conv2d_wgrad": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_l, c_p, c_i,
                              c_p],
    "indm_h16_attn_fwd": [c_p, c_p, c_p, c_l, c_p, c_p, c_i, c_i, c_i, c_f, c_p],
    "indm_h16_attn_bwd": [c_p, c_p, c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_f, c_p],
    "indm_h16_rowdot": [c_p, c_p, c_p, c_l, c_i, c_p],
    "indm_h16_groupnorm_ok": [c_i, c_i, c_i],
    "indm_h16_groupnorm_ws_floats": [c_i, c_i],
    "indm_h16_groupnorm_fwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_u64, c_p, c_p],
    "indm_h16_groupnorm_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_u64, c_p, c_p, c_f,
                               c_p, c_p],
    "indm_h16_segreduce_ws_floats": [c_l, c_l, c_i],
    "indm_h16_segreduce": [c_p, c_p, c_l, c_l, c_i, c_f, c_p, c_p],
    "indm_h16_scale": [c_p, c_p, c_l, c_f, c_p],
    "indm_h16_axpby_mixed": [c_p, c_p, c_p, c_l, c_f, c_f, c_p],
    "indm_h16_upsample2x": [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p],
    "indm_h16_downsample2x": [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p],
    "indm_h16_from_f32": [c_p, c_p, c_l, c_p],
    "indm_h16_to_f32": [c_p, c_p, c_l, c_f, c_p],
    "indm_timer_create": [c_i, c_p],
    "indm_timer_destroy": [c_p],
    "indm_timer_reset": [c_p, c_i],
    "indm_timer_read": [c_p, c_p, c_p, c_p],
}
_LONG_RET = {"indm_groupnorm_ws_floats", "indm_colreduce_ws_floats", "indm_segreduce_ws_floats", "indm_rowreduce_ws_floats",
             "indm_dsm_ws_floats", "indm_sqnorm_ws_floats", "indm_ires_bwd_act_ws_floats", "indm_iresblock_ws_floats",
             "indm_pc_step_ws_floats", "indm_hutch_div_ws_floats", "indm_f16_bwd_act_ok", "indm_f16_wgrad_ws_floats",
             "indm_h16_groupnorm_ws_floats", "indm_h16_segreduce_ws_floats"}

EXPORTED_SYMBOLS = tuple(sorted(_SIGS))


def lib():
    """Load libindm_hip.so once; fail loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"indm_amd: {_LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C indm_amd/csrc`. There is no CPU fallback.")
        L = ctypes.CDLL(_LIB_PATH)
        partial = os.environ.get("INDM_HIP_LIB_PARTIAL") == "1"   # tools: A/B against an older build that lacks newer entry points
        for name, args in _SIGS.items():
            if partial and not hasattr(L, name):
                continue
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = c_l if name in _LONG_RET else (ctypes.c_char_p if name in ("indm_igemm_last_variant", "indm_version") else c_i)
        _lib = L
    return _lib


def available():
    return os.path.exists(_LIB_PATH)


def _chk(status, name):
    if status != 0:
        raise RuntimeError(f"{name} failed: {ERRORS.get(status, status)}")


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("indm_amd.hip: tensor is not on the GPU; the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"indm_amd.hip: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError("indm_amd.hip: tensor must be contiguous")
    return t.data_ptr()


if hasattr(torch._C, "_cuda_getCurrentRawStream") and hasattr(torch._C, "_cuda_getDevice"):
    def _stream():
        # the raw handle of torch's CURRENT stream (side streams of graph capture included); torch.cuda.current_stream()
        # builds a Stream object through ~10 Python frames (8.6 us per launch = 10 ms of host time per training step)
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
else:  # a torch build without the private accessors
    def _stream():
        return torch.cuda.current_stream().cuda_stream


def to_device(t, device):
    """Small host tensor ([B] schedule vectors, tables) -> device through pinned memory, non-blocking: a pageable
    `.to(device)` stages synchronously (0.4 ms each, nine per training step).  The caching host allocator keeps the
    pinned block alive until the copy has run."""
    device = torch.device(device)
    if device.type != "cuda" or t.is_cuda:
        return t.to(device)
    return t.contiguous().pin_memory().to(device, non_blocking=True)


_ws_cache = {}


POISON_WS = {"on": os.environ.get("INDM_POISON_WS", "0") == "1"}


def _ws(nfloats, device):
    """Grow-only scratch buffer per device (stream-ordered reuse).  Every kernel pair that goes through it (partials +
    reduce) must write what it reads: the buffer holds whatever the previous op left.  POISON_WS (env INDM_POISON_WS=1)
    fills the buffer with NaN on every request -- a debugging mode in which a read of stale scratch turns
    the result into NaN instead of a run-to-run difference in the last bits (tests/test_properties_gpu.py)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nfloats:
        buf = torch.empty(max(int(nfloats), 1 << 22), dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    if POISON_WS["on"]:
        buf.fill_(float("nan"))   # the whole buffer: a kernel reading past the extent it asked for is a bug as well
    return buf


SPLITK_WS_FLOATS = 1 << 25  # 128 MiB of fp32 partial slabs for weight-grad split-K


def version():
    """indm_version() of the loaded library, e.g. '6.6 gfx950'"""
    return lib().indm_version().decode()


def debug_state():
    """Fingerprint of every piece of process-wide state a launch could depend on (scratch buffer, caches, mode
    switches, allocator totals): tests/test_properties_gpu.py prints it next to a reproducibility failure so that two
    runs of one test can be compared for what the process history left behind."""
    ws = {k: (v.data_ptr(), v.numel()) for k, v in _ws_cache.items()}
    out = dict(ws=ws, packed=len(_packed), w2t_packed=len(_w2t_packed), tables=len(_tables), col_offs=len(_col_offs),
               ires_ws=len(_ires_ws_sizes), dtype=_MFMA["dtype"], block_drivers=BLOCK_DRIVERS["on"],
               poison=POISON_WS["on"], profile=_Profile.enabled, timer=_active_timer[0] is not None)
    if torch.cuda.is_available():
        out["allocated"], out["reserved"] = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
    return out


_MFMA = {"dtype": "f32"}  # "f32" | "bf16": see set_compute_dtype()

# --------------------------------------------------------------------------- live per-kernel timing (bench.py)
class _Profile:
    """HIP-event timing of the dense (igemm) launches on the stream they are launched on.  Keys name the kernel
    template instantiation, e.g. 'igemm<A_ROW,B_ROW,128x128>'.  Used by bench.py for the roofline line."""
    enabled = False
    only = None
    rec = {}
    shapes = bool(os.environ.get("INDM_PROFILE_SHAPES"))


def profile_start(only=None):
    _Profile.enabled, _Profile.only, _Profile.rec = True, only, {}


def profile_stop():
    """-> {key: dict(launches, flops, ms)} (synchronises)."""
    _Profile.enabled = False
    torch.cuda.synchronize()
    out = {}
    for k, (flops, evs) in _Profile.rec.items():  # flops = [flops, algorithmic bytes]
        out[k] = dict(launches=len(evs), flops=flops[0], bytes=flops[1], ms=sum(a.elapsed_time(b) for a, b in evs))
    _Profile.rec = {}
    return out


class _NoTiming:
    """what _timed() hands out while no profile is being recorded (or the launch cannot belong to the one class being
    recorded): a shared do-nothing context.  Attribute writes (`tm.variant = ...`, `tm.flops += ...`) are dropped, so
    the process-wide singleton never accumulates state from the main or the autograd thread."""
    __slots__ = ()
    variant, flops = None, 0.0

    def __setattr__(self, name, value):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NO_TIMING = _NoTiming()
_HOST_NAMED = ("ires_chain", "attn_fused")   # variants named by this module (every other one is chosen by the C side)


def _timed(amode, bmode, *a, variant=None, **k):
    """variant: the class name when the caller knows it before the launch.  With profile_start(only=key) a launch is
    bracketed by HIP events only if it can belong to `key`: in bench.py's timed region that is the ~60 launches per
    step of the dominant kernel instead of all ~1 500 dense launches."""
    if not _Profile.enabled:
        return _NO_TIMING
    only = _Profile.only
    if only is not None:
        if variant is not None:
            if only != f"igemm<{amode},{bmode},{variant}>":
                return _NO_TIMING
        elif only.split(",", 2)[-1].startswith(_HOST_NAMED):
            return _NO_TIMING
    t = _Timed(amode, bmode, *a, **k)
    t.variant = variant
    return t


class _Timed:
    """HIP-event bracket around one dense launch.  The class key names the kernel variant the C side actually
    dispatched to (indm_igemm_last_variant) -- no Python mirror of the dispatch rules."""

    def __init__(self, amode, bmode, M, N, K, batch=1, splitk=False, nbytes=None, gated=False):
        self.amode, self.bmode, self.shape = amode, bmode, (M, N, K, batch)
        self.flops = 2.0 * M * N * K * batch
        # algorithmic bytes: every operand read once, the result written once (conv callers pass the image size)
        self.bytes = float(nbytes) if nbytes is not None else 4.0 * batch * (M * K + K * N + M * N)
        self.on = _Profile.enabled
        self.variant = None

    def __enter__(self):
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if self.on:
            self.e1.record()
            key = f"igemm<{self.amode},{self.bmode},{self.variant or lib().indm_igemm_last_variant().decode()}>"
            if _Profile.shapes:  # INDM_PROFILE_SHAPES=1: one class per problem shape (tools / bench stderr table only)
                key += " M=%d N=%d K=%d b=%d" % self.shape
            if _Profile.only is None or _Profile.only == key:
                r = _Profile.rec.setdefault(key, [[0.0, 0.0], []])
                r[0][0] += self.flops
                r[0][1] += self.bytes
                r[1].append((self.e0, self.e1))
        return False


# --------------------------------------------------------------------------- GEMM / conv
def set_compute_dtype(name):
    """'f32' (default: exact fp32 MFMA) or 'bf16' (MFMA inputs rounded to bf16, fp32 accumulate -- the
    "bf16 compute / fp32 accum" mode of BASELINE configs #3-#5).  Affects every dense contraction; tensors stay fp32."""
    if name not in ("f32", "fp32", "bf16"):
        raise ValueError(f"compute dtype {name} unknown")
    _MFMA["dtype"] = "bf16" if name == "bf16" else "f32"


def compute_dtype():
    return _MFMA["dtype"]


F16_STORAGE = {"on": os.environ.get("INDM_F16_STORAGE", "1") != "0"}   # tools / tests: bf16 compute on fp32 tensors


def storage_dtype():
    """'bf16' when the flow's 512-wide tensors live in HBM as bf16 (bf16 compute mode on the block drivers: F16 fragment
    order, csrc/flow_f16.hip), else 'f32'.
    THREE switches decide it: set_compute_dtype('bf16'), f16_storage(True) (default) and block_drivers(True) (default) -- the
    op-level composition has no F16 form, so block_drivers(False) silently falls back to bf16 MFMA inputs on fp32 tensors
    (bench.py's profiled warm-up step and tools/ rely on that).  It applies to the training AND the eval forward (the
    NELBO's power-series log-det); the fixed-point inverse always runs on fp32 tensors (bf16 MFMA inputs in bf16 mode):
    tests/test_block_drivers_gpu.py::test_bf16_storage_eval_logdet_and_inverse_tolerances states the resulting tolerances."""
    return "bf16" if (_MFMA["dtype"] == "bf16" and F16_STORAGE["on"] and BLOCK_DRIVERS["on"]) else "f32"


def f16_storage(on=None):
    if on is not None:
        F16_STORAGE["on"] = bool(on)
    return F16_STORAGE["on"]


def _dense(name):
    return getattr(lib(), f"indm_{name}_bf16" if _MFMA["dtype"] == "bf16" else f"indm_{name}_f32")


def gemm(a, b, trans_a=False, trans_b=False, alpha=1.0, bias=None, bias2=None, rows_per_group=1, res=None,
         beta=1.0, splitk=False, gate=None):
    """C[(batch),M,N] = alpha * (op(a * gate) @ op(b) + bias[N] + bias2[M//rows_per_group, N]) + beta*res.
    gate (optional) has a's shape and multiplies it elementwise inside the operand loader.

    a: [..,M,K] (or stored [..,K,M] when trans_a); b: [..,K,N] (or stored [..,N,K] when trans_b).
    2-D operands broadcast over the batch of the other one.
    """
    ab = a.shape[0] if a.dim() == 3 else 1
    bb = b.shape[0] if b.dim() == 3 else 1
    batch = max(ab, bb)
    am, ak = (a.shape[-1], a.shape[-2]) if trans_a else (a.shape[-2], a.shape[-1])
    bk, bn = (b.shape[-1], b.shape[-2]) if trans_b else (b.shape[-2], b.shape[-1])
    if ak != bk:
        raise RuntimeError(f"gemm: inner dims differ {tuple(a.shape)} x {tuple(b.shape)}")
    M, N, K = am, bn, ak
    shape = (batch, M, N) if (a.dim() == 3 or b.dim() == 3) else (M, N)
    out = torch.empty(shape, dtype=torch.float32, device=a.device)
    sA = a.shape[-1] * a.shape[-2] if a.dim() == 3 else 0
    sB = b.shape[-1] * b.shape[-2] if b.dim() == 3 else 0
    ws = _ws(SPLITK_WS_FLOATS, a.device)  # split-K is decided inside (only when the tile count cannot fill the chip)
    nbytes = 4 * (a.numel() * (2 if gate is not None else 1) + b.numel() + out.numel()
                  + (res.numel() if res is not None else 0))
    with _timed("A_COL" if trans_a else "A_ROW", "B_COL" if trans_b else "B_ROW", M, N, K, batch, splitk, nbytes,
                gated=gate is not None):
        st = _dense("gemm")(_ptr(a), _ptr(gate), _ptr(b), _ptr(out), M, N, K, int(trans_a), int(trans_b), a.shape[-1],
                                 b.shape[-1], N, batch, sA, sB, M * N, alpha, _ptr(bias), _ptr(bias2), rows_per_group,
                                 _ptr(res), N, M * N if res is not None and res.dim() == 3 else 0, beta,
                                 _ptr(ws), SPLITK_WS_FLOATS, _stream())
    _chk(st, "indm_gemm_f32")
    return out


_packed = {}  # id(weight) -> (weakref(weight), version, kind, packed copy)


def _packed_weight(w, kind):
    """Re-laid-out copies of a (small) conv weight used by the skinny-conv GEMM forms, cached per weight OBJECT: the
    residual flow multiplies by the same normalised weights n+4 times per call, and each call used to re-permute them.
    An entry is valid only while the very same tensor object is alive and unmodified (weakref + version counter)."""
    import weakref
    # never cache a Parameter: the fused optimiser rewrites parameter memory through the flat buffer, which no version
    # counter sees.  The flow's normalised weights are fresh non-parameter tensors on every flow call.
    # ... nor an ALIAS of one (graphed.py captures the score network on detached aliases of its parameters: a hit here during
    # the capture would freeze the warm-up pass's packed copy into the graph -- the weights of every later replay)
    cacheable = not isinstance(w, torch.nn.Parameter) and not getattr(w, "_indm_param_alias", False)
    key = (id(w), kind)
    hit = _packed.get(key) if cacheable else None
    if hit is not None and hit[0]() is w and hit[1] == w._version:
        return hit[2]
    KH, KW, Ci, Co = w.shape
    if kind == "ci_tapco":      # [Ci, (tap, co)] : 3x3 conv with a skinny OUTPUT as one GEMM with N = 9*Co
        out = w.permute(2, 0, 1, 3).reshape(Ci, KH * KW * Co).contiguous()
    else:                       # "tapco_ci": [(tap, co), Ci] : transposed conv with a skinny INPUT
        out = w.permute(0, 1, 3, 2).reshape(KH * KW * Co, Ci).contiguous()
    if len(_packed) > 1024:
        for k in [k for k, v in _packed.items() if v[0]() is None]:
            del _packed[k]
    if cacheable:
        _packed[key] = (weakref.ref(w), w._version, out)
    return out


def prepack_weights(ws):
    """Fill the _packed_weight cache for a list of same-shape RSCK weights with TWO launches per layout (stack +
    permuted copy) instead of one small copy per weight and layout: the flow does this once per call for the skinny
    3x3 weights of a whole scale.  Non-parameter tensors only (see _packed_weight)."""
    import weakref
    ws = [w for w in ws if not isinstance(w, torch.nn.Parameter)]
    if not ws:
        return
    KH, KW, Ci, Co = ws[0].shape
    for k in [k for k, v in _packed.items() if v[0]() is None]:   # entries of weights that no longer exist (they pin
        del _packed[k]                                            # the stacked copies of earlier flow calls)
    with torch.no_grad():
        st = torch.stack([w.detach() for w in ws])                                          # [n, KH, KW, Ci, Co]
        a = st.permute(0, 3, 1, 2, 4).reshape(len(ws), Ci, KH * KW * Co).contiguous()       # "ci_tapco"
        b = st.permute(0, 1, 2, 4, 3).reshape(len(ws), KH * KW * Co, Ci).contiguous()       # "tapco_ci"
    for i, w in enumerate(ws):
        _packed[(id(w), "ci_tapco")] = (weakref.ref(w), w._version, a[i])
        _packed[(id(w), "tapco_ci")] = (weakref.ref(w), w._version, b[i])


def _skinny(kh, kw, stride, pad, c_small, c_wide):
    return kh == 3 and kw == 3 and stride == 1 and pad == 1 and c_small <= 16 and c_wide >= 64


def gemm_ires_bwd_act(x, b_nk, du, d, a, t, want_dt=False, rows_per_seg=None):
    """ires_bwd_act(gemm(x, b_nk, trans_b=True), du, d, a, t, ...) in ONE launch (fp32): the GEMM result is never written and the
    elementwise pass runs as the GEMM's epilogue.  Returns (dp, dt, seg) bit-identical to the two calls, or None when the
    direct-to-LDS kernel does not cover the GEMM: the caller runs the two steps."""
    M, K = x.shape
    N = b_nk.shape[0]
    dp = torch.empty((M, N), dtype=torch.float32, device=x.device)
    dt = torch.empty_like(dp) if want_dt else None
    seg = ws = None
    if rows_per_seg is not None:
        if rows_per_seg % 64 or M % rows_per_seg:
            raise RuntimeError("gemm_ires_bwd_act: rows_per_seg must be a multiple of 64 that divides the row count")
        seg = torch.empty((M // rows_per_seg, N), dtype=torch.float32, device=x.device)
        ws = _ws(lib().indm_ires_bwd_act_ws_floats(M, N), x.device)
    st = lib().indm_gemm_ires_bwd_act_f32(_ptr(x), _ptr(b_nk), _ptr(du), _ptr(d), _ptr(a), _ptr(t), _ptr(dp), _ptr(dt), _ptr(seg),
                                          M, N, K, x.stride(0), b_nk.stride(0), rows_per_seg or 0, _ptr(ws), _stream())
    if st == 1:
        return None
    _chk(st, "indm_gemm_ires_bwd_act_f32")
    return dp, dt, seg


def gemm_gated_ires_bwd_act(x, gate, b_kn, da, d, a, t, rows_per_seg=None):
    """du = gemm(x, b_kn, gate=gate) AND ires_bwd_act(da, du, d, a, t, False, rows_per_seg) in ONE launch (fp32).  Returns
    (du, dp, seg) bit-identical to the two calls, or None when the gated direct-to-LDS kernel does not cover the GEMM."""
    M, K = x.shape
    N = b_kn.shape[1]
    du = torch.empty((M, N), dtype=torch.float32, device=x.device)
    dp = torch.empty_like(du)
    seg = ws = None
    if rows_per_seg is not None:
        if rows_per_seg % 64 or M % rows_per_seg:
            raise RuntimeError("gemm_gated_ires_bwd_act: rows_per_seg must be a multiple of 64 that divides the row count")
        seg = torch.empty((M // rows_per_seg, N), dtype=torch.float32, device=x.device)
        ws = _ws(lib().indm_ires_bwd_act_ws_floats(M, N), x.device)
    st = lib().indm_gemm_gated_ires_bwd_act_f32(_ptr(x), _ptr(gate), _ptr(b_kn), _ptr(da), _ptr(d), _ptr(a), _ptr(t), _ptr(du),
                                                _ptr(dp), _ptr(seg), M, N, K, x.stride(0), b_kn.stride(0), rows_per_seg or 0,
                                                _ptr(ws), _stream())
    if st == 1:
        return None
    _chk(st, "indm_gemm_gated_ires_bwd_act_f32")
    return du, dp, seg


def gemm_sin(a, b_nk, bias=None, bias2=None, rows_per_group=1):
    """(s, c) = Sin(a @ b_nk^T + bias + bias2[row // rows_per_group]) in ONE launch (fp32; the pre-activation is never written):
    the second level of the flow's g-net.  a [M, K], b_nk [N, K].  Bit-identical to gemm(trans_b=True) followed by sincos_fwd.
    None when the direct-to-LDS kernel does not cover the problem: the caller runs the two steps."""
    M, K = a.shape
    N = b_nk.shape[0]
    s = torch.empty((M, N), dtype=torch.float32, device=a.device)
    c = torch.empty_like(s)
    st = lib().indm_gemm_sin_f32(_ptr(a), _ptr(b_nk), _ptr(s), _ptr(c), M, N, K, a.stride(0), b_nk.stride(0), N, _ptr(bias),
                                 _ptr(bias2), rows_per_group, _stream())
    if st == 1:
        return None
    _chk(st, "indm_gemm_sin_f32")
    return s, c


def conv3x3_thin_act(x, w2d, sign=-1, bias=None, bias2=None):
    """(a, d) = Sin(conv3x3_thin(x, w2d, sign) + bias + bias2) in ONE launch (the pre-activation is never written): the first
    level of the flow's g-net.  Bit-identical to conv3x3_thin followed by sincos_fwd.  None when the row-fragment kernel does
    not cover the geometry: the caller runs the two steps."""
    N, H, W, C = x.shape
    Co = w2d.shape[1]
    a = torch.empty((N, H, W, Co), dtype=torch.float32, device=x.device)
    d = torch.empty_like(a)
    with _timed("A_CONV3", "B_ROW", N * H * W, Co, 9 * C, nbytes=4 * (x.numel() + w2d.numel() + 2 * a.numel())):
        st = _dense("conv3x3_thin_act")(_ptr(x), _ptr(w2d), _ptr(a), _ptr(d), N, H, W, C, Co, sign, _ptr(bias), _ptr(bias2),
                                        _stream())
    if st == 1:
        return None
    _chk(st, "indm_conv3x3_thin_act_f32")
    return a, d


def tap_gather(P, bias, sign, alpha=1.0):
    """y[n,h,w,c] = alpha*(sum_taps P[n, h+s(kh-1), w+s(kw-1), tap*C+c] + bias[c]);  P: [N,H,W,9C]."""
    N, H, W, C9 = P.shape
    C = C9 // 9
    y = torch.empty((N, H, W, C), dtype=torch.float32, device=P.device)
    _chk(lib().indm_tap_gather_f32(_ptr(P), _ptr(bias), _ptr(y), N, H, W, C, sign, alpha, _stream()),
         "indm_tap_gather_f32")
    return y


def tap_gather_acc(P, sign, mulp=None, acc_in=None, coeff=0.0, alpha=1.0):
    """tap_gather with the series bookkeeping fused: y = alpha*gather(P) * mulp; acc = acc_in + coeff*y.
    Returns (y, acc) (acc None without acc_in)."""
    N, H, W, C9 = P.shape
    C = C9 // 9
    y = torch.empty((N, H, W, C), dtype=torch.float32, device=P.device)
    acc = torch.empty_like(y) if acc_in is not None else None
    _chk(lib().indm_tap_gather_acc_f32(_ptr(P), None, _ptr(mulp), _ptr(y), _ptr(acc_in), _ptr(acc), coeff, N, H, W, C,
                                       sign, alpha, _stream()), "indm_tap_gather_acc_f32")
    return y, acc


def tap_spread(y, sign, alpha=1.0):
    """Adjoint of tap_gather: P[n,h,w,tap*C+c] = alpha*y[n, h-s(kh-1), w-s(kw-1), c]."""
    N, H, W, C = y.shape
    P = torch.empty((N, H, W, 9 * C), dtype=torch.float32, device=y.device)
    _chk(lib().indm_tap_spread_f32(_ptr(y), _ptr(P), N, H, W, C, sign, alpha, _stream()), "indm_tap_spread_f32")
    return P


def conv3x3_thin(x, w2d, sign, alpha=1.0, bias=None, bias2=None, res=None, beta=1.0):
    """3x3 / pad 1 / stride 1 conv of a skinny NHWC input x [N,H,W,C] with w2d [9C, Co] (rows (tap, c)); the 9-tap gather
    happens inside the thin-K GEMM's operand loader.  sign -1: conv forward, +1: transposed conv (w2d pre-packed).
    Returns None when the kernel does not cover the geometry: the caller falls back."""
    N, H, W, C = x.shape
    Co = w2d.shape[1]
    y = torch.empty((N, H, W, Co), dtype=torch.float32, device=x.device)
    nbytes = 4 * (x.numel() + w2d.numel() + y.numel() + (res.numel() if res is not None else 0))
    with _timed("A_CONV3", "B_ROW", N * H * W, Co, 9 * C, nbytes=nbytes):
        st = _dense("conv3x3_thin")(_ptr(x), _ptr(w2d), _ptr(y), N, H, W, C, Co, sign, alpha, _ptr(bias),
                                    _ptr(bias2), _ptr(res), beta, _stream())
    if st == 1:
        return None
    _chk(st, "indm_conv3x3_thin_f32")
    return y


def conv_out_size(h, k, stride, pad):
    return (h + 2 * pad - k) // stride + 1


def conv2d(x, w, stride=1, pad=1, alpha=1.0, bias=None, bias2=None, res=None, beta=1.0):
    """NHWC cross-correlation. x [N,H,W,Ci], w [KH,KW,Ci,Co] -> alpha*(conv + bias[Co] + bias2[N,Co]) + beta*res."""
    N, H, W, Ci = x.shape
    KH, KW, Ci2, Co = w.shape
    if Ci != Ci2:
        raise RuntimeError("conv2d: channel mismatch")
    if _skinny(KH, KW, stride, pad, Co, Ci) and bias2 is None and res is None:
        # 3x3 conv with a skinny output (flow edges 512 -> 3 / 12): one MFMA GEMM with N = 9*Co + a tap gather
        P = gemm(x.reshape(-1, Ci), _packed_weight(w, "ci_tapco"))
        return tap_gather(P.view(N, H, W, 9 * Co), bias, 1, alpha)
    if _skinny(KH, KW, stride, pad, Ci, Co):
        # skinny INPUT (flow edges 3 / 12 -> 512): spread x over the 9 taps (a [pixels, 9*Ci] panel, a few MB) and
        # run one plain MFMA GEMM; keeps every epilogue fusion of the GEMM
        y = conv3x3_thin(x, w.reshape(9 * Ci, Co), -1, alpha, bias, bias2, res, beta)
        if y is not None:
            return y
        xcol = tap_spread(x, -1)
        r = res.reshape(-1, Co) if res is not None else None
        y2 = gemm(xcol.reshape(-1, 9 * Ci), w.reshape(9 * Ci, Co), alpha=alpha, bias=bias, bias2=bias2,
                  rows_per_group=H * W, res=r, beta=beta)
        return y2.view(N, H, W, Co)
    OH, OW = conv_out_size(H, KH, stride, pad), conv_out_size(W, KW, stride, pad)
    y = torch.empty((N, OH, OW, Co), dtype=torch.float32, device=x.device)
    with _timed("A_CONV", "B_ROW", N * OH * OW, Co, KH * KW * Ci, nbytes=4 * (x.numel() + w.numel() + y.numel())):
        ws = _ws(SPLITK_WS_FLOATS, x.device)
        st = _dense("conv2d_fwd")(_ptr(x), _ptr(w), _ptr(y), N, H, W, Ci, Co, KH, KW, stride, pad, 1, OH, OW,
                                       alpha, _ptr(bias), _ptr(bias2), _ptr(res), beta, _ptr(ws), SPLITK_WS_FLOATS,
                                       _stream())
    _chk(st, "indm_conv2d_fwd_f32")
    return y


def conv2d_dgrad(dy, w, in_hw, stride=1, pad=1, alpha=1.0, res=None, beta=1.0, gate=None):
    """Gradient of conv2d w.r.t. its input (= transposed convolution). dy [N,OH,OW,Co], w [KH,KW,Ci,Co]
    -> dx [N,H,W,Ci] with (H,W) = in_hw.  gate (optional, dy's shape): computes with dy*gate."""
    N, OH, OW, Co = dy.shape
    if gate is not None and not (_skinny(w.shape[0], w.shape[1], stride, pad, w.shape[2], Co) and res is None):
        dy, gate = mul(dy, gate), None
    KH, KW, Ci, Co2 = w.shape
    if Co != Co2:
        raise RuntimeError("conv2d_dgrad: channel mismatch")
    H, W = in_hw
    if _skinny(KH, KW, stride, pad, Ci, Co) and res is None:
        # transposed 3x3 conv with a skinny output: P = dy @ w[(tap,ci), co]^T, then gather with flipped shifts
        P = gemm(dy.reshape(-1, Co), w.reshape(9 * Ci, Co), trans_b=True,
                 gate=None if gate is None else gate.reshape(-1, Co))
        return tap_gather(P.view(N, H, W, 9 * Ci), None, -1, alpha)
    if _skinny(KH, KW, stride, pad, Co, Ci):
        # transposed conv with a skinny INPUT dy: spread dy (+1), then [pixels, 9*Co] x [(tap,co), ci]
        dx = conv3x3_thin(dy, _packed_weight(w, "tapco_ci"), 1, alpha, None, None, res, beta)
        if dx is not None:
            return dx
        dycol = tap_spread(dy, 1)
        r = res.reshape(-1, Ci) if res is not None else None
        return gemm(dycol.reshape(-1, 9 * Co), _packed_weight(w, "tapco_ci"), alpha=alpha, res=r,
                    beta=beta).view(N, H, W, Ci)
    dx = torch.empty((N, H, W, Ci), dtype=torch.float32, device=dy.device)
    with _timed("A_CONV", "B_DGRAD", N * H * W, Ci, KH * KW * Co, nbytes=4 * (dy.numel() + w.numel() + dx.numel())):
        ws = _ws(SPLITK_WS_FLOATS, dy.device)
        st = _dense("conv2d_dgrad")(_ptr(dy), _ptr(w), _ptr(dx), N, OH, OW, Ci, Co, KH, KW, KH - 1 - pad, stride,
                                         H, W, alpha, _ptr(res), beta, _ptr(ws), SPLITK_WS_FLOATS, _stream())
    _chk(st, "indm_conv2d_dgrad_f32")
    return dx


def conv2d_wgrad(x, dy, kh, kw, stride=1, pad=1, alpha=1.0, dy_gate=None, x_gate=None):
    """Gradient of conv2d w.r.t. its RSCK weight. x [N,H,W,Ci], dy [N,OH,OW,Co] -> [KH,KW,Ci,Co].
    dy_gate (optional, dy's shape): computes with dy .* dy_gate; on the skinny-x form the gate rides on the GEMM's
    operand loader (the wide tensor is read once, no gated copy is written).  x_gate: the same for x on the
    skinny-dy form."""
    N, H, W, Ci = x.shape
    _, OH, OW, Co = dy.shape
    if x_gate is not None:
        if dy_gate is None and _skinny(kh, kw, stride, pad, Co, Ci):
            dycol = tap_spread(dy, 1, alpha)
            q = gemm(x.reshape(-1, Ci), dycol.reshape(-1, 9 * Co), trans_a=True, splitk=True,
                     gate=x_gate.reshape(-1, Ci))
            return q.view(Ci, 3, 3, Co).permute(1, 2, 0, 3).contiguous()
        x = mul(x, x_gate)
    if dy_gate is not None:
        if _skinny(kh, kw, stride, pad, Ci, Co):
            # (dy .* g)^T [Co x pixels] x xcol [pixels x 9*Ci]  ->  [Co, 9*Ci], transposed back to RSCK (tiny)
            xcol = tap_spread(x, -1)
            q = gemm(dy.reshape(-1, Co), xcol.reshape(-1, 9 * Ci), trans_a=True, alpha=alpha, splitk=True,
                     gate=dy_gate.reshape(-1, Co))
            return q.t().contiguous().view(3, 3, Ci, Co)
        dy = mul(dy, dy_gate)
    if _skinny(kh, kw, stride, pad, Co, Ci):
        # weight grad with a skinny dy: spread dy over the 9 taps, then one [Ci x pixels] x [pixels x 9*Co] GEMM
        dycol = tap_spread(dy, 1, alpha)
        q = gemm(x.reshape(-1, Ci), dycol.reshape(-1, 9 * Co), trans_a=True, splitk=True)
        return q.view(Ci, 3, 3, Co).permute(1, 2, 0, 3).contiguous()
    if _skinny(kh, kw, stride, pad, Ci, Co):
        # weight grad with a skinny x: xcol^T [9*Ci x pixels] x dy [pixels x Co] is already RSCK
        xcol = tap_spread(x, -1)
        return gemm(xcol.reshape(-1, 9 * Ci), dy.reshape(-1, Co), trans_a=True, alpha=alpha,
                    splitk=True).view(3, 3, Ci, Co)
    dw = torch.empty((kh, kw, Ci, Co), dtype=torch.float32, device=x.device)
    ws = _ws(SPLITK_WS_FLOATS, x.device)
    with _timed("A_WGRAD", "B_ROW", kh * kw * Ci, Co, N * OH * OW, nbytes=4 * (x.numel() + dy.numel() + dw.numel())):
        st = _dense("conv2d_wgrad")(_ptr(x), _ptr(dy), _ptr(dw), N, H, W, Ci, Co, kh, kw, stride, pad, OH, OW,
                                         alpha, None, 0.0, _ptr(ws), SPLITK_WS_FLOATS, _stream())
    _chk(st, "indm_conv2d_wgrad_f32")
    return dw


# --------------------------------------------------------------------------- norms / softmax
SEED_DEV = [None]   # a [1] int64 device tensor while a hipGraph of the score network is captured (indm_amd/graphed.py), else None


def _seed_ptr(seed_dev):
    return None if seed_dev is None else seed_dev.data_ptr()


def groupnorm_fwd(x, gamma, beta, G, eps, act=0, drop_p=0.0, seed=0, seed_dev=None):
    """x [N,..spatial..,C] -> y, mean[N,G], rstd[N,G]. act: 0 none, 1 SiLU. dropout via Philox(seed + *seed_dev)."""
    N, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * C)
    y = torch.empty_like(x)
    mean = torch.empty((N, G), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    ws = _ws(lib().indm_groupnorm_ws_floats(N, HW, C, G), x.device)
    st = lib().indm_groupnorm_fwd_f32(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd), N, HW, C, G,
                                      eps, act, drop_p, seed, _seed_ptr(seed_dev), _ptr(ws), _stream())
    _chk(st, "indm_groupnorm_fwd_f32")
    return y, mean, rstd


def groupnorm_bwd(dy, x, gamma, beta, mean, rstd, G, act=0, drop_p=0.0, seed=0, seed_dev=None):
    N, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * C)
    dx = torch.empty_like(x)
    dgamma = torch.empty_like(gamma)
    dbeta = torch.empty_like(beta)
    ws = _ws(lib().indm_groupnorm_ws_floats(N, HW, C, G), x.device)
    st = lib().indm_groupnorm_bwd_f32(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd), _ptr(dx),
                                      _ptr(dgamma), _ptr(dbeta), N, HW, C, G, act, drop_p, seed, _seed_ptr(seed_dev), _ptr(ws),
                                      _stream())
    _chk(st, "indm_groupnorm_bwd_f32")
    return dx, dgamma, dbeta


def softmax_fwd(x, scale=1.0):
    y = torch.empty_like(x)
    T = x.shape[-1]
    _chk(lib().indm_softmax_fwd_f32(_ptr(x), _ptr(y), x.numel() // T, T, scale, _stream()), "indm_softmax_fwd_f32")
    return y


def softmax_bwd(dy, y, scale=1.0):
    dx = torch.empty_like(y)
    T = y.shape[-1]
    _chk(lib().indm_softmax_bwd_f32(_ptr(dy), _ptr(y), _ptr(dx), y.numel() // T, T, scale, _stream()),
         "indm_softmax_bwd_f32")
    return dx


def colreduce(x, y=None, mode=0, alpha=1.0):
    """out[c] = alpha * sum over all leading dims of f(x, y); mode 0: x, 1: x*y, 2: x*x."""
    C = x.shape[-1]
    R = x.numel() // C
    out = torch.empty((C,), dtype=torch.float32, device=x.device)
    ws = _ws(lib().indm_colreduce_ws_floats(R, C), x.device)
    _chk(lib().indm_colreduce_f32(_ptr(x), _ptr(y), _ptr(out), R, C, mode, alpha, 0.0, _ptr(ws), _stream()),
         "indm_colreduce_f32")
    return out


def segreduce(x, rows_per_seg, alpha=1.0):
    """out[s, c] = alpha * sum of the s-th run of rows_per_seg rows of x (viewed as [rows, C]): per-image bias grads."""
    C = x.shape[-1]
    R = x.numel() // C
    if R % rows_per_seg:
        raise RuntimeError("segreduce: rows do not divide into segments")
    nseg = R // rows_per_seg
    out = torch.empty((nseg, C), dtype=torch.float32, device=x.device)
    ws = _ws(lib().indm_segreduce_ws_floats(nseg, rows_per_seg, C), x.device)
    _chk(lib().indm_segreduce_f32(_ptr(x), _ptr(out), nseg, rows_per_seg, C, alpha, _ptr(ws), _stream()),
         "indm_segreduce_f32")
    return out


def bn_stats(x, eps):
    """Per-channel batch statistics over all leading dims of NHWC x: mean, biased var, rstd."""
    C = x.shape[-1]
    R = x.numel() // C
    mean = torch.empty((C,), dtype=torch.float32, device=x.device)
    var = torch.empty_like(mean)
    rstd = torch.empty_like(mean)
    ws = _ws(lib().indm_colreduce_ws_floats(R, C) + 2 * C, x.device)
    _chk(lib().indm_bn_stats_f32(_ptr(x), _ptr(mean), _ptr(var), _ptr(rstd), R, C, eps, _ptr(ws), _stream()),
         "indm_bn_stats_f32")
    return mean, var, rstd


def bn_apply(x, mean, rstd, gamma, beta, act=0):
    y = torch.empty_like(x)
    C = x.shape[-1]
    _chk(lib().indm_bn_apply_f32(_ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(y), x.numel() // C, C,
                                 act, _stream()), "indm_bn_apply_f32")
    return y


def bn_bwd(dy, x, mean, rstd, gamma, beta, act=0):
    """Backward of act(BN(x)) with batch statistics -> dx, dgamma, dbeta."""
    C = x.shape[-1]
    R = x.numel() // C
    dx = torch.empty_like(x)
    dgamma = torch.empty_like(gamma)
    dbeta = torch.empty_like(gamma)
    ws = _ws(lib().indm_colreduce_ws_floats(R, C) + R * C + 2 * C, x.device)
    _chk(lib().indm_bn_bwd_f32(_ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(dx),
                               _ptr(dgamma), _ptr(dbeta), R, C, act, _ptr(ws), _stream()), "indm_bn_bwd_f32")
    return dx, dgamma, dbeta


# --------------------------------------------------------------------------- elementwise
def sincos_fwd(p, want_d=True):
    a = torch.empty_like(p)
    d = torch.empty_like(p) if want_d else None
    _chk(lib().indm_sincos_act_fwd_f32(_ptr(p), _ptr(a), _ptr(d), p.numel(), _stream()), "indm_sincos_act_fwd_f32")
    return a, d


def sincos_bwd(da, dd, a, d):
    dp = torch.empty_like(a)
    _chk(lib().indm_sincos_act_bwd_f32(_ptr(da), _ptr(dd), _ptr(a), _ptr(d), _ptr(dp), a.numel(), _stream()),
         "indm_sincos_act_bwd_f32")
    return dp


def attn_fwd(q, k, v, scale):
    """Fused attention forward: (h, lse) or None when the kernel does not cover the geometry (or in bf16 mode)."""
    if _MFMA["dtype"] != "f32":
        return None
    B, T, C = q.shape
    h = torch.empty_like(q)
    lse = torch.empty((B, T), dtype=torch.float32, device=q.device)
    with _timed("ATTN", "FWD", T, C, T, B, variant="attn_fused") as tm:
        st = lib().indm_attn_fwd_f32(_ptr(q), _ptr(k), _ptr(v), _ptr(h), _ptr(lse), B, T, C, scale, _stream())
        tm.flops = 4.0 * B * T * T * C
    if st == 1:
        return None
    _chk(st, "indm_attn_fwd_f32")
    return h, lse


def attn_bwd(q, k, v, h, dh, lse, scale):
    """Fused attention backward -> (dq, dk, dv); logits / probabilities are recomputed in registers."""
    B, T, C = q.shape
    Dv = rowreduce(dh.view(B * T, C), h.view(B * T, C), 0).view(B, T)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    with _timed("ATTN", "BWD", T, C, T, B, variant="attn_fused") as tm:
        st = lib().indm_attn_bwd_f32(_ptr(q), _ptr(k), _ptr(v), _ptr(dh), _ptr(lse), _ptr(Dv), _ptr(dq), _ptr(dk),
                                     _ptr(dv), B, T, C, scale, _stream())
        tm.flops = 10.0 * B * T * T * C
    _chk(st, "indm_attn_bwd_f32")
    return dq, dk, dv


_w2t_packed = {}  # id(w2t) -> (weakref, version, packed bf16 copy)


def pack_w2t_bf16(w2ts):
    """bf16 MFMA-fragment-order copies of a list of same-shape [K, N] fp32 weights (two launches for the whole list; N is
    zero-padded to a multiple of 32), registered for the bf16 chain / F16 kernels:
    out[s][b][lane][e] = bf16(w[16 s + 8 (e >> 2) + 4 (lane >> 5) + (e & 3)][32 b + (lane & 31)])."""
    import weakref
    if not w2ts:
        return
    K, N = w2ts[0].shape
    if K % 16:
        return
    with torch.no_grad():
        st = torch.stack([w.detach() for w in w2ts])
        if N % 32:
            st = torch.nn.functional.pad(st, (0, 32 - N % 32))
            N = st.shape[-1]
        st = st.to(torch.bfloat16)                                                            # [n, K, N]
        pk = st.view(len(w2ts), K // 16, 2, 2, 4, N // 32, 32).permute(0, 1, 5, 3, 6, 2, 4).contiguous()
    for k in [k for k, v in _w2t_packed.items() if v[0]() is None]:   # drop the copies of weights that no longer exist
        del _w2t_packed[k]
    for i, w in enumerate(w2ts):
        # never register a Parameter: the fused optimiser rewrites parameter memory through the flat buffer, which no
        # version counter sees (the same rule as _packed_weight; the flow's normalised weights are fresh tensors per call)
        if not isinstance(w, torch.nn.Parameter):
            _w2t_packed[id(w)] = (weakref.ref(w), w._version, pk[i])
    return pk


def packed_bf16(w):
    """the registered packed copy of w (pack_w2t_bf16), packing it now if the flow's prepare() did not"""
    hit = _w2t_packed.get(id(w))
    if hit is None or hit[0]() is not w or hit[1] != w._version:
        pk = pack_w2t_bf16([w])
        if pk is None:
            raise RuntimeError(f"packed_bf16: a [{w.shape[0]}, {w.shape[1]}] weight has no fragment-order form (K % 16 != 0)")
        return pk[0]
    return hit[2]


def ires_chain(t3, d2, w2t, d1, w1p, J, want_t2=False):
    """P[M, J] = ((t3 .* d2) @ w2t .* d1) @ w1p[:, :J] in ONE launch with the 512-wide intermediate kept on chip (the
    back-to-back contraction of the VJP chain).  t3, d2: [M, K]; w2t: [K, 512]; d1: [M, 512]; w1p: [512, 32 if J <= 32 else 128] (zero-padded).
    want_t2: also return t2 = (t3 .* d2) @ w2t ([M, 512], stored from the accumulators): -> (P, t2).
    Returns None when the kernel does not cover the geometry: the caller runs the two GEMMs."""
    M, K = t3.shape
    N = w2t.shape[1]
    P = torch.empty((M, J), dtype=torch.float32, device=t3.device)
    t2 = torch.empty((M, N), dtype=torch.float32, device=t3.device) if want_t2 else None
    nbytes = 4 * (2 * t3.numel() + d1.numel() + w2t.numel() + P.numel() + (t2.numel() if want_t2 else 0))
    # the launch that also stores the 512-wide intermediate (training's last chain link) is its own class: same MFMA
    # work plus a 4*M*N-byte store.  J <= 32 and J <= 128 are two instantiations of the kernel -- JB = 1 / 4 panel blocks
    # in its second stage -- and two rows of a rocprofv3 summary: one class each, so that a class average is ONE kernel's
    variant = (("ires_chain64x512x32,gated" if J <= 32 else "ires_chain64x512x128,gated")
               + (",t2out" if want_t2 else "") + (",bf16" if _MFMA["dtype"] == "bf16" else ""))
    with _timed("A_ROW", "B_ROW", M, N, K, nbytes=nbytes, variant=variant) as tm:
        hit = _w2t_packed.get(id(w2t)) if _MFMA["dtype"] == "bf16" else None
        if hit is not None and hit[0]() is w2t and hit[1] == w2t._version:
            st = lib().indm_ires_chain_bf16w(_ptr(t3), _ptr(d2), hit[2].data_ptr(), _ptr(d1), _ptr(w1p), _ptr(P),
                                             _ptr(t2), M, K, N, J, _stream())
        else:
            st = _dense("ires_chain")(_ptr(t3), _ptr(d2), _ptr(w2t), _ptr(d1), _ptr(w1p), _ptr(P), _ptr(t2), M, K, N, J,
                                      _stream())
        tm.flops += 2.0 * M * N * J   # algorithmic: the panel's J live columns, not its zero padding
    if st == 1:
        return None
    _chk(st, "indm_ires_chain_f32")
    return (P, t2) if want_t2 else P


def ires_bwd_act(da, du, d, a, t, want_dt=False, rows_per_seg=None):
    """One Sin level of the iResBlock's hand-written backward: dp = da*d - 4 pi^2 a (du*t); optionally dt = du*d and the
    per-segment column sums of dp ([rows / rows_per_seg, C]: the per-image bias gradient).  -> (dp, dt, seg)."""
    C = a.shape[-1]
    R = a.numel() // C
    dp = torch.empty_like(a)
    dt = torch.empty_like(a) if want_dt else None
    seg = ws = None
    if rows_per_seg is not None:
        if rows_per_seg % 64 or R % rows_per_seg:
            raise RuntimeError("ires_bwd_act: rows_per_seg must be a multiple of 64 that divides the row count")
        seg = torch.empty((R // rows_per_seg, C), dtype=torch.float32, device=a.device)
        ws = _ws(lib().indm_ires_bwd_act_ws_floats(R, C), a.device)
    _chk(lib().indm_ires_bwd_act_f32(_ptr(da), _ptr(du), _ptr(d), _ptr(a), _ptr(t), _ptr(dp), _ptr(dt), _ptr(seg), R, C,
                                     rows_per_seg or 0, _ptr(ws), _stream()), "indm_ires_bwd_act_f32")
    return dp, dt, seg


ACT_SILU, ACT_ELU = 1, 2


def act_fwd(x, act):
    y = torch.empty_like(x)
    _chk(lib().indm_act_fwd_f32(_ptr(x), _ptr(y), x.numel(), act, _stream()), "indm_act_fwd_f32")
    return y


def act_bwd(dy, x, act):
    dx = torch.empty_like(x)
    _chk(lib().indm_act_bwd_f32(_ptr(dy), _ptr(x), _ptr(dx), x.numel(), act, _stream()), "indm_act_bwd_f32")
    return dx


def axpby(x, y, a, b, c=0.0):
    """a*x + b*y + c (y may be None)."""
    out = torch.empty_like(x)
    _chk(lib().indm_axpby_f32(_ptr(x), _ptr(y), _ptr(out), a, b, c, x.numel(), _stream()), "indm_axpby_f32")
    return out


def mul(x, y, alpha=1.0):
    out = torch.empty_like(x)
    _chk(lib().indm_mul_f32(_ptr(x), _ptr(y), _ptr(out), alpha, x.numel(), _stream()), "indm_mul_f32")
    return out


def mul2(x, y1, y2):
    """(x*y1, x*y2) in one pass over x."""
    o1, o2 = torch.empty_like(x), torch.empty_like(x)
    _chk(lib().indm_mul2_f32(_ptr(x), _ptr(y1), _ptr(y2), _ptr(o1), _ptr(o2), x.numel(), _stream()), "indm_mul2_f32")
    return o1, o2


def scale_rows(x, s):
    """out[b, ...] = x[b, ...] * s[b]."""
    out = torch.empty_like(x)
    B = x.shape[0]
    _chk(lib().indm_scale_rows_f32(_ptr(x), _ptr(s), _ptr(out), B, x.numel() // B, _stream()), "indm_scale_rows_f32")
    return out


def rowreduce(x, y=None, mode=0, alpha=1.0):
    """out[b] = alpha * sum_i f(x[b,i], y[b,i]); mode 0: x*y, 1: x*x, 2: x."""
    B = x.shape[0]
    D = x.numel() // B
    out = torch.empty((B,), dtype=torch.float32, device=x.device)
    ws = _ws(lib().indm_rowreduce_ws_floats(B, D), x.device)
    _chk(lib().indm_rowreduce_f32(_ptr(x), _ptr(y), _ptr(out), B, D, mode, alpha, _ptr(ws), _stream()),
         "indm_rowreduce_f32")
    return out


def upsample2x(x, alpha=1.0):
    N, H, W, C = x.shape
    y = torch.empty((N, 2 * H, 2 * W, C), dtype=torch.float32, device=x.device)
    _chk(lib().indm_upsample2x_f32(_ptr(x), _ptr(y), N, H, W, C, alpha, _stream()), "indm_upsample2x_f32")
    return y


def downsample2x(x, alpha=0.25):
    N, H, W, C = x.shape
    y = torch.empty((N, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
    _chk(lib().indm_downsample2x_f32(_ptr(x), _ptr(y), N, H, W, C, alpha, _stream()), "indm_downsample2x_f32")
    return y


_col_offs = {}


def split_cols(x, sizes, inverse_of=None):
    """x [R, sum(sizes)] -> flat buffer of sum(sizes)*R floats in which block b is the contiguous [R, sizes[b]]
    matrix at offset R * sum(sizes[:b]) (one launch instead of one strided copy per block).
    inverse_of=R: x is such a flat buffer, the result is the [R, sum(sizes)] matrix."""
    sizes = tuple(int(v) for v in sizes)
    key = (sizes, x.device)
    offs = _col_offs.get(key)
    if offs is None:
        acc = [0]
        for v in sizes:
            acc.append(acc[-1] + v)
        offs = _col_offs[key] = torch.tensor(acc, dtype=torch.int32).to(x.device)
    tot = sum(sizes)
    R = x.shape[0] if inverse_of is None else int(inverse_of)
    y = torch.empty((R * tot,) if inverse_of is None else (R, tot), dtype=torch.float32, device=x.device)
    _chk(lib().indm_split_cols_f32(_ptr(x), _ptr(y), offs.data_ptr(), len(sizes), R, tot, 0 if inverse_of is None else 1,
                                   _stream()), "indm_split_cols_f32")
    return y


def weightnorm_fwd(v, g):
    """v [O, I], g [O] (or [O, 1]) -> (w = v * g / ||v||_row, ||v||_row [O])"""
    O, I = v.shape
    w, n = torch.empty_like(v), torch.empty((O,), dtype=torch.float32, device=v.device)
    _chk(lib().indm_weightnorm_fwd_f32(_ptr(v), _ptr(g), _ptr(w), _ptr(n), O, I, _stream()), "indm_weightnorm_fwd_f32")
    return w, n


def weightnorm_bwd(dw, v, g, n):
    """-> (dv [O, I], dg [O])"""
    O, I = v.shape
    dv, dg = torch.empty_like(v), torch.empty((O,), dtype=torch.float32, device=v.device)
    _chk(lib().indm_weightnorm_bwd_f32(_ptr(dw), _ptr(v), _ptr(g), _ptr(n), _ptr(dv), _ptr(dg), O, I, _stream()),
         "indm_weightnorm_bwd_f32")
    return dv, dg


def squeeze2x(x):
    N, H, W, C = x.shape
    y = torch.empty((N, H // 2, W // 2, 4 * C), dtype=torch.float32, device=x.device)
    _chk(lib().indm_squeeze2x_f32(_ptr(x), _ptr(y), N, H, W, C, _stream()), "indm_squeeze2x_f32")
    return y


def unsqueeze2x(x):
    N, H, W, C4 = x.shape
    y = torch.empty((N, 2 * H, 2 * W, C4 // 4), dtype=torch.float32, device=x.device)
    _chk(lib().indm_unsqueeze2x_f32(_ptr(x), _ptr(y), N, H, W, C4, _stream()), "indm_unsqueeze2x_f32")
    return y


def nchw_to_nhwc(x):
    N, C, H, W = x.shape
    y = torch.empty((N, H, W, C), dtype=torch.float32, device=x.device)
    _chk(lib().indm_nchw_nhwc_f32(_ptr(x), _ptr(y), N, C, H * W, 0, _stream()), "indm_nchw_nhwc_f32")
    return y


def nhwc_to_nchw(x):
    N, H, W, C = x.shape
    y = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
    _chk(lib().indm_nchw_nhwc_f32(_ptr(x), _ptr(y), N, C, H * W, 1, _stream()), "indm_nchw_nhwc_f32")
    return y


def lop_normalize_fwd(w, coeff):
    """w: RSCK weight (any leading dims, Cout last) -> wn, l1[Cout]."""
    Co = w.shape[-1]
    R = w.numel() // Co
    wn = torch.empty_like(w)
    l1 = torch.empty((Co,), dtype=torch.float32, device=w.device)
    _chk(lib().indm_lop_normalize_fwd_f32(_ptr(w), _ptr(wn), _ptr(l1), R, Co, coeff, _stream()),
         "indm_lop_normalize_fwd_f32")
    return wn, l1


def lop_normalize_bwd(dwn, w, l1, coeff):
    Co = w.shape[-1]
    R = w.numel() // Co
    dw = torch.empty_like(w)
    dot = torch.empty_like(l1)
    _chk(lib().indm_lop_normalize_bwd_f32(_ptr(dwn), _ptr(w), _ptr(l1), _ptr(dw), _ptr(dot), R, Co, coeff, _stream()),
         "indm_lop_normalize_bwd_f32")
    return dw


def multi_copy(srcs, dsts):
    """dsts[i][...] = srcs[i] (contiguous fp32, same numel) for all i in ONE launch."""
    if not srcs:
        return
    rows = [[_ptr(s_), _ptr(d), d.numel()] for s_, d in zip(srcs, dsts)]
    table = _lop_table(rows, dsts[0].device)
    _chk(lib().indm_multi_copy_f32(table.data_ptr(), len(rows), max(r[2] for r in rows), _stream()),
         "indm_multi_copy_f32")


def multi_copy_table(srcs, dsts):
    """a PERSISTENT device table for multi_copy of fixed (src, dst) pairs -> handle for multi_copy_run (indm_amd/graphed.py:
    the static gradient buffers of a captured backward graph -> their slices of the flat gradient buffer)"""
    if not srcs:
        return None
    rows = [[_ptr(s_), _ptr(d), d.numel()] for s_, d in zip(srcs, dsts)]
    tab = torch.tensor(rows, dtype=torch.int64).to(dsts[0].device)
    return (tab, len(rows), max(r[2] for r in rows))


def multi_copy_run(handle):
    if handle is not None:
        _chk(lib().indm_multi_copy_f32(handle[0].data_ptr(), handle[1], handle[2], _stream()), "indm_multi_copy_f32")


_tables = {}  # (device, rows) -> (pinned host staging tensor, device table): both stay alive with the entry


def _lop_table(rows, device):
    """Device-side pointer table of a multi-tensor launch.  Staged through a PINNED host tensor that outlives the
    asynchronous copy (a dropped pageable temporary would rely on the runtime staging it before returning) and cached
    by content: the flow's weights / flat-buffer slices keep their addresses from step to step, and the caching
    allocator hands autograd the same gradient blocks again, so most steps re-use the resident table."""
    flat = tuple(r if isinstance(r, int) else tuple(r) for r in rows)
    key = (device.type, device.index, flat)
    hit = _tables.get(key)
    if hit is not None:
        return hit[1]
    host = torch.tensor(rows, dtype=torch.int64)
    if device.type == "cuda":
        host = host.pin_memory()
    dev = host.to(device, non_blocking=True)
    if len(_tables) >= 256:
        _tables.pop(next(iter(_tables)))
    _tables[key] = (host, dev)
    return dev


def lop_normalize_multi_fwd(ws, coeff):
    """Lop-normalise a list of RSCK weights in two launches -> (list of wn, list of l1[Cout])."""
    dev = ws[0].device
    sizes = [w.numel() for w in ws]
    cos = [w.shape[-1] for w in ws]
    wn_all = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    l1_all = torch.empty(sum(cos), dtype=torch.float32, device=dev)
    wns, l1s, rows, o, q = [], [], [], 0, 0
    for w, n, co in zip(ws, sizes, cos):
        wn, l1 = wn_all[o:o + n].view(w.shape), l1_all[q:q + co]
        wns.append(wn); l1s.append(l1)
        rows.append([_ptr(w), 0, wn.data_ptr(), l1.data_ptr(), n // co, co])
        o += n; q += co
    table = _lop_table(rows, dev)
    _chk(lib().indm_lop_normalize_multi_fwd_f32(table.data_ptr(), len(ws), max(cos), max(sizes), coeff, _stream()),
         "indm_lop_normalize_multi_fwd_f32")
    return wns, l1s


def lop_normalize_multi_bwd(dwns, ws, l1s, coeff):
    dev = ws[0].device
    sizes = [w.numel() for w in ws]
    cos = [w.shape[-1] for w in ws]
    dw_all = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    dot_all = torch.empty(sum(cos), dtype=torch.float32, device=dev)
    dws, rows, o, q = [], [], 0, 0
    for dwn, w, l1, n, co in zip(dwns, ws, l1s, sizes, cos):
        dw = dw_all[o:o + n].view(w.shape)
        dws.append(dw)
        rows.append([_ptr(w), _ptr(dwn), dw.data_ptr(), dot_all[q:q + co].data_ptr(), n // co, co])
        o += n; q += co
    table = _lop_table(rows, dev)
    l1tab = _lop_table([l1.data_ptr() for l1 in l1s], dev)
    _chk(lib().indm_lop_normalize_multi_bwd_f32(table.data_ptr(), l1tab.data_ptr(), len(ws), max(cos), max(sizes),
                                                coeff, _stream()), "indm_lop_normalize_multi_bwd_f32")
    return dws


# --------------------------------------------------------------------------- SDE / loss / sampler / optimiser
def coupling_fwd(x, prm, split, up, inverse):
    """Affine coupling of the prior flow on [B, 2F] rows -> (y, logdet[B]); split 0 continuous / 1 skip."""
    B, D = x.shape
    y = torch.empty_like(x)
    ld = torch.empty((B,), dtype=torch.float32, device=x.device)
    _chk(lib().indm_coupling_fwd_f32(_ptr(x), _ptr(prm), _ptr(y), _ptr(ld), B, D // 2, int(split), int(up), int(inverse),
                                     _stream()), "indm_coupling_fwd_f32")
    return y, ld


def coupling_bwd(dy, dld, x, prm, split, up, inverse):
    B, D = x.shape
    dx, dprm = torch.empty_like(x), torch.empty_like(prm)
    _chk(lib().indm_coupling_bwd_f32(_ptr(dy), _ptr(dld), _ptr(x), _ptr(prm), _ptr(dx), _ptr(dprm), B, D // 2, int(split),
                                     int(up), int(inverse), _stream()), "indm_coupling_bwd_f32")
    return dx, dprm


def philox_state(seed, device):
    """{seed, offset = 0} as an int64[2] device tensor (the generator state of philox_randn)."""
    return torch.tensor([int(seed) & ((1 << 63) - 1), 0], dtype=torch.int64).to(device)


def philox_randn(shape, state):
    """Standard normals of `shape` drawn on the device from the Philox state tensor (advanced in stream order)."""
    if not state.is_cuda or state.dtype != torch.int64:
        raise RuntimeError("philox_randn: state must be an int64[2] tensor on the GPU")
    out = torch.empty(shape, dtype=torch.float32, device=state.device)
    _chk(lib().indm_philox_randn_f32(_ptr(out), out.numel(), state.data_ptr(), _stream()), "indm_philox_randn_f32")
    return out


def sde_perturb(x, noise, mc, sd):
    out = torch.empty_like(x)
    B = x.shape[0]
    _chk(lib().indm_sde_perturb_f32(_ptr(x), _ptr(noise), _ptr(mc), _ptr(sd), _ptr(out), B, x.numel() // B, _stream()),
         "indm_sde_perturb_f32")
    return out


def dsm_loss_fwd(out, z, std, w, kind, mode, reduce_mean):
    B = out.shape[0]
    D = out.numel() // B
    loss = torch.empty((B,), dtype=torch.float32, device=out.device)
    ws = _ws(lib().indm_dsm_ws_floats(B, D), out.device)
    _chk(lib().indm_dsm_loss_fwd_f32(_ptr(out), _ptr(z), _ptr(std), _ptr(w), _ptr(loss), B, D, kind, mode,
                                     int(reduce_mean), _ptr(ws), _stream()), "indm_dsm_loss_fwd_f32")
    return loss


def dsm_loss_bwd(gl, out, z, std, w, kind, mode, reduce_mean):
    B = out.shape[0]
    D = out.numel() // B
    dout = torch.empty_like(out)
    _chk(lib().indm_dsm_loss_bwd_f32(_ptr(gl), _ptr(out), _ptr(z), _ptr(std), _ptr(w), _ptr(dout), B, D, kind, mode,
                                     int(reduce_mean), _stream()), "indm_dsm_loss_bwd_f32")
    return dout


def affine_update(x, s, z, a, b, c, want_mean=True):
    """x_mean = a[b]*x + b[b]*s ; x_new = x_mean + c[b]*z."""
    B = x.shape[0]
    x_new = torch.empty_like(x)
    x_mean = torch.empty_like(x) if want_mean else None
    _chk(lib().indm_affine_update_f32(_ptr(x), _ptr(s), _ptr(z), _ptr(a), _ptr(b), _ptr(c), _ptr(x_new), _ptr(x_mean),
                                      B, x.numel() // B, _stream()), "indm_affine_update_f32")
    return x_new, x_mean


def langevin_step(grad_sq, noise_sq, alpha, snr):
    step = torch.empty_like(grad_sq)
    ncoef = torch.empty_like(grad_sq)
    _chk(lib().indm_langevin_step_f32(_ptr(grad_sq), _ptr(noise_sq), _ptr(alpha), _ptr(step), _ptr(ncoef),
                                      grad_sq.numel(), snr, _stream()), "indm_langevin_step_f32")
    return step, ncoef


def _noise_args(z, philox):
    """(z pointer, Philox state pointer) of a sampler update: exactly one noise source (None, None = no noise)"""
    if z is not None and philox is not None:
        raise RuntimeError("pc_step: pass z or philox, not both")
    if philox is not None and (not philox.is_cuda or philox.dtype != torch.int64):
        raise RuntimeError("pc_step: philox must be an int64[2] tensor on the GPU")
    return _ptr(z), (philox.data_ptr() if philox is not None else None)


def pc_step_em(x, net_out, f_lin, g, k, dt, probability_flow=False, z=None, philox=None, want_mean=True):
    """EulerMaruyamaPredictor.update_fn (sampling.py:191-197 with RSDE.sde sde_lib.py:96-103) -> (x_new, x_mean).
    f_lin, g, k: [B] device vectors (forward drift f = f_lin x, diffusion, score = k net_out); noise from z or drawn
    inside the kernel from the device Philox state."""
    B = x.shape[0]
    x_new, x_mean = torch.empty_like(x), (torch.empty_like(x) if want_mean else None)
    zp, pp = _noise_args(z, philox)
    _chk(lib().indm_pc_step_em_f32(_ptr(x), _ptr(net_out), _ptr(f_lin), _ptr(g), _ptr(k), float(dt),
                                   int(bool(probability_flow)), zp, pp, _ptr(x_new), _ptr(x_mean), B, x.numel() // B,
                                   _stream()), "indm_pc_step_em_f32")
    return x_new, x_mean


def pc_step_rd(x, net_out, f_lin, G, k, probability_flow=False, z=None, philox=None, want_mean=True):
    """ReverseDiffusionPredictor.update_fn (sampling.py:205-210 with RSDE.discretize sde_lib.py:105-118)"""
    B = x.shape[0]
    x_new, x_mean = torch.empty_like(x), (torch.empty_like(x) if want_mean else None)
    zp, pp = _noise_args(z, philox)
    _chk(lib().indm_pc_step_rd_f32(_ptr(x), _ptr(net_out), _ptr(f_lin), _ptr(G), _ptr(k), int(bool(probability_flow)),
                                   zp, pp, _ptr(x_new), _ptr(x_mean), B, x.numel() // B, _stream()),
         "indm_pc_step_rd_f32")
    return x_new, x_mean


def pc_step_langevin(x, net_out, k, alpha, snr, z=None, philox=None):
    """One LangevinCorrector iteration (sampling.py:279-290) -> (x_new, x_mean)"""
    B = x.shape[0]
    D = x.numel() // B
    x_new, x_mean = torch.empty_like(x), torch.empty_like(x)
    ws = _ws(lib().indm_pc_step_ws_floats(B, D), x.device)
    zp, pp = _noise_args(z, philox)
    _chk(lib().indm_pc_step_langevin_f32(_ptr(x), _ptr(net_out), _ptr(k), _ptr(alpha), float(snr), zp, pp, _ptr(x_new),
                                         _ptr(x_mean), _ptr(ws), B, D, _stream()), "indm_pc_step_langevin_f32")
    return x_new, x_mean


def hutch_div(grad_fn_eps, eps):
    """Hutchinson divergence estimate (likelihood.py:27-38): [B] = sum over the image of grad_fn_eps * eps"""
    B = eps.shape[0]
    D = eps.numel() // B
    div = torch.empty((B,), dtype=torch.float32, device=eps.device)
    ws = _ws(lib().indm_hutch_div_ws_floats(B, D), eps.device)
    _chk(lib().indm_hutch_div_f32(_ptr(grad_fn_eps), _ptr(eps), _ptr(div), B, D, _ptr(ws), _stream()),
         "indm_hutch_div_f32")
    return div


def sqnorm(g):
    out = torch.empty((1,), dtype=torch.float32, device=g.device)
    ws = _ws(lib().indm_sqnorm_ws_floats(), g.device)
    _chk(lib().indm_sqnorm_f32(_ptr(g), g.numel(), _ptr(out), _ptr(ws), _stream()), "indm_sqnorm_f32")
    return out


def adamw_ema(p, g, m, v, ema, lr, beta1, beta2, eps, wd, step, clip, gnorm_sq, gscale, ema_one_minus_decay):
    _chk(lib().indm_adamw_ema_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(ema), p.numel(), lr, beta1, beta2, eps, wd,
                                  step, clip, _ptr(gnorm_sq), gscale, ema_one_minus_decay, _stream()),
         "indm_adamw_ema_f32")


def fixed_point_check(x, xp, y, atol, rtol):
    """[1] tensor: count of elements with (x-xp)^2/(atol+|y|rtol) >= 1 (0 => converged)."""
    out = torch.empty((1,), dtype=torch.float32, device=x.device)
    _chk(lib().indm_fixed_point_check_f32(_ptr(x), _ptr(xp), _ptr(y), x.numel(), atol, rtol, _ptr(out), _stream()),
         "indm_fixed_point_check_f32")
    return out


def ema_update(s, p, one_minus_decay):
    _chk(lib().indm_ema_update_f32(_ptr(s), _ptr(p), s.numel(), one_minus_decay, _stream()), "indm_ema_update_f32")


def slogdet_inv(w):
    """log|det w| (shape [1]) and w^-1 for a square matrix with n <= 64."""
    n = w.shape[0]
    ld = torch.empty((1,), dtype=torch.float32, device=w.device)
    inv = torch.empty_like(w)
    _chk(lib().indm_slogdet_inv_f32(_ptr(w), n, _ptr(ld), _ptr(inv), _stream()), "indm_slogdet_inv_f32")
    return ld, inv


# --------------------------------------------------------------------------- reference native ops
def upfirdn2d(x, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, layout):
    """layout 0: x is [N,C,H,W]; layout 1: x is [N,H,W,C]."""
    if layout == 0:
        N, C, H, W = x.shape
    else:
        N, H, W, C = x.shape
    kh, kw = kernel.shape
    oh = (H * up_y + pad_y0 + pad_y1 - kh) // down_y + 1
    ow = (W * up_x + pad_x0 + pad_x1 - kw) // down_x + 1
    shape = (N, C, oh, ow) if layout == 0 else (N, oh, ow, C)
    y = torch.empty(shape, dtype=x.dtype, device=x.device)
    if x.dtype == torch.bfloat16:   # bf16 in, bf16 out (fp32 taps and accumulation)
        _chk(lib().indm_upfirdn2d_bf16(_ptr16(x), _ptr(kernel), y.data_ptr(), N, C, H, W, kh, kw, up_x, up_y, down_x, down_y,
                                       pad_x0, pad_x1, pad_y0, pad_y1, layout, _stream()), "indm_upfirdn2d_bf16")
        return y
    _chk(lib().indm_upfirdn2d_f32(_ptr(x), _ptr(kernel), _ptr(y), N, C, H, W, kh, kw, up_x, up_y, down_x, down_y,
                                  pad_x0, pad_x1, pad_y0, pad_y1, layout, _stream()), "indm_upfirdn2d_f32")
    return y


def fused_bias_act(x, bias, ref, act, grad, alpha, scale):
    """Same contract as the reference's fused.fused_bias_act (op/fused_bias_act.cpp:11-21), x is [N,C,...]."""
    y = torch.empty_like(x)
    size_b = bias.numel() if bias is not None and bias.numel() > 0 else 0
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    _chk(lib().indm_fused_bias_act_f32(_ptr(x), _ptr(bias) if size_b else None,
                                       _ptr(ref) if ref is not None and ref.numel() > 0 else None, _ptr(y), x.numel(),
                                       step_b, size_b, act, grad, alpha, scale, _stream()), "indm_fused_bias_act_f32")
    return y


# --------------------------------------------------------------------------- bf16 storage of the flow (csrc/flow_f16.hip)
def _ptr16(t):
    """raw pointer of an F16 / packed-weight tensor (torch.bfloat16, contiguous, on the GPU)"""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("indm_amd.hip: tensor is not on the GPU; the HIP path has no CPU fallback")
    if t.dtype != torch.bfloat16 or not t.is_contiguous():
        raise RuntimeError(f"indm_amd.hip: expected a contiguous bfloat16 tensor, got {t.dtype}")
    return t.data_ptr()


def f16_pack(x):
    """fp32 [M, N] row-major -> F16 (bf16, MFMA-fragment order; include/indm_hip.h) as a flat bfloat16 tensor"""
    M, N = x.shape
    y = torch.empty(M * N, dtype=torch.bfloat16, device=x.device)
    _chk(lib().indm_f16_pack_f32(_ptr(x), y.data_ptr(), M, N, _stream()), "indm_f16_pack_f32")
    return y


def f16_unpack(y, M, N):
    x = torch.empty((M, N), dtype=torch.float32, device=y.device)
    _chk(lib().indm_f16_unpack_f32(_ptr16(y), _ptr(x), M, N, _stream()), "indm_f16_unpack_f32")
    return x


def pack_weight_bf16(w):
    """row-major fp32 W[K, Nw] (K % 16 == 0; Nw zero-padded to a multiple of 32) -> the packed bf16 fragment order every F16
    kernel multiplies by: Wp[ks][nblk][lane][e] = W[16 ks + 8 (e >> 2) + 4 (lane >> 5) + (e & 3)][32 nblk + (lane & 31)]"""
    K, N = w.shape
    if N % 32:
        w = torch.nn.functional.pad(w, (0, 32 - N % 32))
        N = w.shape[1]
    with torch.no_grad():
        return (w.detach().to(torch.bfloat16).view(K // 16, 2, 2, 4, N // 32, 32).permute(0, 4, 2, 5, 1, 3).contiguous().view(-1))


F16_PLAIN, F16_ACT, F16_GATE = 0, 1, 2
F16_GEMM, F16_GEMM_ACT, F16_CHAIN, F16_THINN, F16_GEMM_GATE, F16_GEMM_ACT_THINN = 0, 1, 2, 3, 4, 5


def f16_thin_conv(x, w2d, sign, epilogue=F16_PLAIN, bias=None, gate=None, want_raw=False):
    """3x3 conv of a skinny NHWC fp32 x with w2d [9C, Cout] -> F16 [pixels, Cout] tensors (see indm_f16_thin_conv).
    Returns None if the geometry is not covered, else out1 | (a, cos) | (gated, raw or None)."""
    N, H, W, C = x.shape
    Co = w2d.shape[1]
    new = lambda: torch.empty(N * H * W * Co, dtype=torch.bfloat16, device=x.device)  # noqa: E731
    o1 = new()
    o2 = new() if (epilogue == F16_ACT or (epilogue == F16_GATE and want_raw)) else None
    st = lib().indm_f16_thin_conv(_ptr(x), _ptr(w2d), _ptr(bias), _ptr16(gate), o1.data_ptr(),
                                  o2.data_ptr() if o2 is not None else None, N, H, W, C, Co, sign, epilogue, _stream())
    if st == 1:
        return None
    _chk(st, "indm_f16_thin_conv")
    return o1 if epilogue == F16_PLAIN else (o1, o2)


def f16_rows(mode, A, M, Wp=None, D1=None, Pp=None, J=0, bias=None, bias2=None, rows_per_group=1, want_t2=False):
    """indm_f16_rows: mode F16_GEMM -> C; F16_GEMM_ACT -> (a, cos); F16_CHAIN -> (P, t2 or None); F16_THINN -> P;
    F16_GEMM_GATE -> (C .* D1, C).  None if not covered."""
    dev = A.device
    new16 = lambda: torch.empty(M * 512, dtype=torch.bfloat16, device=dev)  # noqa: E731
    o1 = new16() if (mode in (F16_GEMM, F16_GEMM_ACT, F16_GEMM_GATE, F16_GEMM_ACT_THINN) or (mode == F16_CHAIN and want_t2)) else None
    o2 = new16() if mode in (F16_GEMM_ACT, F16_GEMM_GATE, F16_GEMM_ACT_THINN) else None
    P = torch.empty((M, J), dtype=torch.float32, device=dev) if mode in (F16_CHAIN, F16_THINN, F16_GEMM_ACT_THINN) else None
    st = lib().indm_f16_rows(mode, _ptr16(A), _ptr16(Wp), _ptr16(D1), _ptr16(Pp), _ptr(P),
                             o1.data_ptr() if o1 is not None else None, o2.data_ptr() if o2 is not None else None,
                             _ptr(bias), _ptr(bias2), M, J, rows_per_group, _stream())
    if st == 1:
        return None
    _chk(st, "indm_f16_rows")
    if mode == F16_GEMM:
        return o1
    if mode == F16_GEMM_ACT_THINN:
        return o1, o2, P
    if mode in (F16_GEMM_ACT, F16_GEMM_GATE):
        return o1, o2
    if mode == F16_CHAIN:
        return P, o1
    return P


def pack_thin_conv_bf16(w2d):
    """[9C, 512] fp32 -> indm_f16_chain_conv's Wc (rows zero-padded to 16 S; k = 16 s + 8 (lane >> 5) + e)"""
    K, N = w2d.shape
    S = (K + 15) // 16
    m = torch.nn.functional.pad(w2d, (0, 0, 0, 16 * S - K)).to(torch.bfloat16)
    return m.view(S, 2, 8, N // 32, 32).permute(0, 3, 1, 4, 2).contiguous()


def f16_chain_conv(x, Wc, gate, W2Tp, D1, Pp, sign=1, want_t=False):
    """indm_f16_chain_conv -> P fp32 [N H W, 9 C] (want_t: (P, t3, t2) with the two saved F16 tensors); None if not covered"""
    N, H, W, C = x.shape
    M = N * H * W
    P = torch.empty((M, 9 * C), dtype=torch.float32, device=x.device)
    t3 = torch.empty(M * 512, dtype=torch.bfloat16, device=x.device) if want_t else None
    t2 = torch.empty(M * 512, dtype=torch.bfloat16, device=x.device) if want_t else None
    st = lib().indm_f16_chain_conv(_ptr(x), _ptr16(Wc), _ptr16(gate), _ptr16(W2Tp), _ptr16(D1), _ptr16(Pp), _ptr(P), _ptr16(t3),
                                   _ptr16(t2), N, H, W, C, sign, 9 * C, _stream())
    if st == 1:
        return None
    _chk(st, "indm_f16_chain_conv")
    return (P, t3, t2) if want_t else P


def f16_gnet(x, W1c, b1, W2p, b2, cb, rows_per_group, W3p):
    """indm_f16_gnet -> (a1, d1, a2, d2, P) (None if not covered)"""
    N, H, W, C = x.shape
    M = N * H * W
    a1, d1, a2, d2 = (torch.empty(M * 512, dtype=torch.bfloat16, device=x.device) for _ in range(4))
    P = torch.empty((M, 9 * C), dtype=torch.float32, device=x.device)
    st = lib().indm_f16_gnet(_ptr(x), _ptr16(W1c), _ptr(b1), a1.data_ptr(), d1.data_ptr(), _ptr16(W2p), _ptr(b2), _ptr(cb), rows_per_group,
                             a2.data_ptr(), d2.data_ptr(), _ptr16(W3p), _ptr(P), N, H, W, C, _stream())
    if st == 1:
        return None
    _chk(st, "indm_f16_gnet")
    return a1, d1, a2, d2, P


def f16_conv_gemm(x, Wc, gate, Wp, sign=-1):
    """indm_f16_conv_gemm -> (out, raw, gated) F16 [N H W, 512] (None if not covered)"""
    N, H, W, C = x.shape
    M = N * H * W
    o, r, g = (torch.empty(M * 512, dtype=torch.bfloat16, device=x.device) for _ in range(3))
    st = lib().indm_f16_conv_gemm(_ptr(x), _ptr16(Wc), _ptr16(gate), _ptr16(Wp), o.data_ptr(), r.data_ptr(), g.data_ptr(), N, H, W, C,
                                  sign, _stream())
    if st == 1:
        return None
    _chk(st, "indm_f16_conv_gemm")
    return o, r, g


def f16_tap_spread(x, sign):
    """indm_f16_tap_spread: NHWC fp32 x [N,H,W,C] -> F16 [N H W, 512] im2col operand (None if not covered)"""
    N, H, W, C = x.shape
    out = torch.empty(N * H * W * 512, dtype=torch.bfloat16, device=x.device)
    st = lib().indm_f16_tap_spread(_ptr(x), out.data_ptr(), N, H, W, C, sign, _stream())
    if st == 1:
        return None
    _chk(st, "indm_f16_tap_spread")
    return out


def f16_bwd_act(da, du, d, a, t, R, rows_per_seg=None):
    """indm_f16_bwd_act -> (dp F16, seg fp32 [R / rows_per_seg, 512] or None); None if not covered"""
    dp = torch.empty(R * 512, dtype=torch.bfloat16, device=a.device)
    seg = torch.empty((R // rows_per_seg, 512), dtype=torch.float32, device=a.device) if rows_per_seg else None
    st = lib().indm_f16_bwd_act(_ptr16(da), _ptr16(du), _ptr16(d), _ptr16(a), _ptr16(t), dp.data_ptr(), _ptr(seg), R,
                                rows_per_seg or R, _stream())
    if st == 1:
        return None
    _chk(st, "indm_f16_bwd_act")
    return dp, seg


def f16_thin_wgrad(thin, wide, M, gate=None, alpha=1.0, transpose=False):
    """out[t][n] = alpha sum_m thin[m][t] (wide .* gate)[m][n] (transpose: out[n][t]); thin fp32 [M, T], wide / gate F16"""
    T = thin.shape[1]
    out = torch.empty((512, T) if transpose else (T, 512), dtype=torch.float32, device=thin.device)
    ws = _ws(SPLITK_WS_FLOATS, thin.device)
    st = lib().indm_f16_thin_wgrad(_ptr(thin), _ptr16(wide), _ptr16(gate), _ptr(out), M, T, alpha, int(transpose),
                                   T if transpose else 512, _ptr(ws), SPLITK_WS_FLOATS, _stream())
    if st == 1:
        return None
    _chk(st, "indm_f16_thin_wgrad")
    return out


def f16_wgrad_gemm(X, Y, M, gate=None, alpha=1.0, transpose=False):
    """out[n1][n2] = alpha sum_m (X .* gate)[m][n1] Y[m][n2]; X, gate, Y F16 [M, 512]"""
    out = torch.empty((512, 512), dtype=torch.float32, device=X.device)
    ws = _ws(SPLITK_WS_FLOATS, X.device)
    st = lib().indm_f16_wgrad_gemm(_ptr16(X), _ptr16(gate), _ptr16(Y), _ptr(out), M, alpha, int(transpose), _ptr(ws),
                                   SPLITK_WS_FLOATS, _stream())
    if st == 1:
        return None
    _chk(st, "indm_f16_wgrad_gemm")
    return out


# --------------------------------------------------------------------------- bf16 storage of the score network (csrc/unet_h16.hip)
H16 = {"on": os.environ.get("INDM_H16", "1") != "0"}   # tools / tests: bf16 compute on fp32 activations (rounds 2-5)
_H16_STATS = {"conv": 0, "conv_fallback": 0, "wgrad": 0, "wgrad_fallback": 0}


def h16(on=None):
    """bf16 STORAGE of the score network's activations in bf16 compute mode (default on; INDM_H16=0 or h16(False) keeps
    the U-Net on fp32 tensors with bf16 MFMA inputs, the round 2-5 behaviour)."""
    if on is not None:
        H16["on"] = bool(on)
    return H16["on"] and _MFMA["dtype"] == "bf16"


def h16_stats(reset=False):
    """launch counts of the H16 kernels vs their fp32-storage fallbacks since the last reset (tests assert coverage)"""
    out = dict(_H16_STATS)
    if reset:
        for k in _H16_STATS:
            _H16_STATS[k] = 0
    return out


def to_bf16(x):
    """fp32 tensor -> bf16 (RNE), same shape"""
    x = x if x.is_contiguous() else x.contiguous()
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    n = x.numel()
    if n % 2 or x.data_ptr() % 16:
        return x.to(torch.bfloat16)   # (odd sizes never occur on the NHWC activations; ATen's converter rounds the same way)
    _chk(lib().indm_h16_from_f32(_ptr(x), y.data_ptr(), n, _stream()), "indm_h16_from_f32")
    return y


def to_f32(x, alpha=1.0):
    """bf16 tensor -> fp32 (exact), scaled by alpha"""
    x = x if x.is_contiguous() else x.contiguous()
    n = x.numel()
    if n % 2 or x.data_ptr() % 16:
        return x.float() * alpha if alpha != 1.0 else x.float()
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _chk(lib().indm_h16_to_f32(_ptr16(x), _ptr(y), n, alpha, _stream()), "indm_h16_to_f32")
    return y


class H16Packer:
    """Packed bf16 copies of a fixed list of fp32 weights (RSCK [KH,KW,Cin,Cout] or [Cin,Cout]): persistent destination
    buffers and a persistent device job table, so that re-packing is ONE kernel launch with no host->device traffic (and
    can sit inside a captured hipGraph).  pack() returns [(fwd [Cout, T*Cin], dg [Cin, T*Cout] taps flipped)], views of the
    persistent buffer: every forward call re-packs (the fused optimiser rewrites parameter memory with no version counter
    to watch), a backward reads the pack of its own forward -- valid as long as no parameter update sits between a forward
    and its backward."""

    def __init__(self, weights, cat_groups=()):
        """cat_groups: [(first index, count)] -- runs of same-shape 2-D weights [Cin, Cout] whose packs are used side by side
        (AttnBlockpp's q | k | v projections): group_views[first] = (fwd [count*Cout, Cin], dg [Cin, count*Cout])"""
        import numpy as np
        self.key = tuple(w.data_ptr() for w in weights)
        dev = weights[0].device
        shapes = []
        for w in weights:
            if w.dim() == 4:
                shapes.append((w.shape[0] * w.shape[1], w.shape[2], w.shape[3]))
            else:
                shapes.append((1, w.shape[0], w.shape[1]))
        total = sum(T * Ci * Co for T, Ci, Co in shapes)
        self.flat = torch.empty(2 * total, dtype=torch.bfloat16, device=dev)
        job = np.zeros(len(weights), dtype=np.dtype([("src", "<u8"), ("fwd", "<u8"), ("dg", "<u8"), ("T", "<i4"), ("Cin", "<i4"),
                                                     ("Cout", "<i4"), ("tile0", "<i4"), ("dg_ld", "<i4"), ("pad", "<i4")]))
        member = {}
        for first, cnt in cat_groups:
            for j in range(cnt):
                member[first + j] = (first, cnt, j)
        self.views, self.group_views, off, tiles, offs = [], {}, 0, 0, []
        base = self.flat.data_ptr()
        for i, (w, (T, Ci, Co)) in enumerate(zip(weights, shapes)):
            n = T * Ci * Co
            offs.append(off)
            if i in member:
                first, cnt, j = member[i]
                g0 = offs[first]
                job[i] = (_ptr(w), base + 2 * off, base + 2 * (total + g0 + j * Co), T, Ci, Co, tiles, cnt * Co, 0)
                self.views.append(None)
                if j == cnt - 1:
                    self.group_views[first] = (self.flat[g0:g0 + cnt * n].view(cnt * Co, Ci),
                                               self.flat[total + g0:total + g0 + cnt * n].view(Ci, cnt * Co))
            else:
                self.views.append((self.flat[off:off + n].view(Co, T * Ci), self.flat[total + off:total + off + n].view(Ci, T * Co)))
                job[i] = (_ptr(w), base + 2 * off, base + 2 * (total + off), T, Ci, Co, tiles, 0, 0)
            tiles += T * ((Ci + 31) // 32) * ((Co + 31) // 32)
            off += n
        self.tab = torch.from_numpy(job.view(np.uint8).copy()).to(dev)
        self.n, self.tiles = len(weights), tiles

    def valid_for(self, weights):
        return self.key == tuple(w.data_ptr() for w in weights)

    def pack(self):
        _chk(lib().indm_h16_pack_weights(self.tab.data_ptr(), self.n, self.tiles, _stream()), "indm_h16_pack_weights")
        return self.views


def h16_pack_weights(weights):
    """one-off pack of a list of fp32 weights -> [(fwd, dg)] (tests / weights that belong to no H16Packer)"""
    if not weights:
        return []
    pk = H16Packer([w.detach() for w in weights])
    out = pk.pack()
    out[0][0]._h16_keep = pk    # the job table must outlive the (asynchronous) launch
    return out


_h16_cnt = {}
H16_COUNTERS = 8192


def _h16_counters(device):
    """per-device tile-arrival counters of the in-kernel split-K combine: zero at rest (the kernels restore them); one array per
    device is enough because every H16 launch of a process goes to the current stream in program order"""
    key = device.index if device.index is not None else torch.cuda.current_device()
    t = _h16_cnt.get(key)
    if t is None:
        t = _h16_cnt[key] = torch.zeros(H16_COUNTERS, dtype=torch.int32, device=device)
    return t


def h16_conv2d(x, wp, KH, KW, stride=1, pad=1, alpha=1.0, bias=None, bias2=None, res=None, beta=1.0, out_f32=False):
    """bf16 NHWC x [N,H,W,Cin], wp packed bf16 [Cout, KH*KW*Cin] -> y [N,OH,OW,Cout] (bf16, or fp32 with out_f32);
    None when the H16 kernel does not cover the geometry (the caller converts and takes the fp32-storage path)."""
    N, H, W, Ci = x.shape
    Co = wp.shape[0]
    if wp.shape[1] != KH * KW * Ci:
        raise RuntimeError("h16_conv2d: packed weight does not match the input")
    OH, OW = conv_out_size(H, KH, stride, pad), conv_out_size(W, KW, stride, pad)
    y = torch.empty((N, OH, OW, Co), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    ws = _ws(SPLITK_WS_FLOATS, x.device)
    st = lib().indm_h16_conv2d(_ptr16(x), _ptr16(wp), y.data_ptr(), N, H, W, Ci, Co, KH, KW, stride, pad, OH, OW, alpha,
                               _ptr(bias), _ptr(bias2), _ptr16(res), beta, int(out_f32), _ptr(ws), SPLITK_WS_FLOATS,
                               _h16_counters(x.device).data_ptr(), H16_COUNTERS, _stream())
    if st == 1:
        _H16_STATS["conv_fallback"] += 1
        return None
    _chk(st, "indm_h16_conv2d")
    _H16_STATS["conv"] += 1
    return y


def h16_conv2d_wgrad(x, dy, KH, KW, stride=1, pad=1, alpha=1.0):
    """bf16 x [N,H,W,Cin], dy [N,OH,OW,Cout] -> fp32 dw [KH,KW,Cin,Cout]; None when not covered"""
    N, H, W, Ci = x.shape
    _, OH, OW, Co = dy.shape
    dw = torch.empty((KH, KW, Ci, Co), dtype=torch.float32, device=x.device)
    ws = _ws(SPLITK_WS_FLOATS, x.device)
    st = lib().indm_h16_conv2d_wgrad(_ptr16(x), _ptr16(dy), _ptr(dw), N, H, W, Ci, Co, KH, KW, stride, pad, OH, OW, alpha,
                                     _ptr(ws), SPLITK_WS_FLOATS, _h16_counters(x.device).data_ptr(), H16_COUNTERS, _stream())
    if st == 1:
        _H16_STATS["wgrad_fallback"] += 1
        return None
    _chk(st, "indm_h16_conv2d_wgrad")
    _H16_STATS["wgrad"] += 1
    return dw


def h16_attn_fwd(qkv, B, T, C, scale):
    """qkv: bf16 [B*T, 3C] (q | k | v column blocks) -> (h bf16 [B*T, C], lse fp32 [B, T]); None when not covered"""
    h = torch.empty((B * T, C), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B, T), dtype=torch.float32, device=qkv.device)
    base = _ptr16(qkv)
    st = lib().indm_h16_attn_fwd(base, base + 2 * C, base + 4 * C, 3 * C, h.data_ptr(), _ptr(lse), B, T, C, scale, _stream())
    if st == 1:
        return None
    _chk(st, "indm_h16_attn_fwd")
    return h, lse


def h16_attn_bwd(qkv, h, dh, lse, B, T, C, scale):
    """-> dqkv bf16 [B*T, 3C] (dq | dk | dv)"""
    D = torch.empty((B, T), dtype=torch.float32, device=qkv.device)
    _chk(lib().indm_h16_rowdot(_ptr16(dh), _ptr16(h), _ptr(D), B * T, C, _stream()), "indm_h16_rowdot")
    dqkv = torch.empty_like(qkv)
    base, gb = _ptr16(qkv), dqkv.data_ptr()
    st = lib().indm_h16_attn_bwd(base, base + 2 * C, base + 4 * C, 3 * C, _ptr16(dh), _ptr(lse), _ptr(D), gb, gb + 2 * C,
                                 gb + 4 * C, 3 * C, B, T, C, scale, _stream())
    _chk(st, "indm_h16_attn_bwd")
    return dqkv


def h16_groupnorm_ok(HW, C, G):
    return bool(lib().indm_h16_groupnorm_ok(HW, C, G))


def h16_groupnorm_fwd(x, gamma, beta, G, eps, act=0, drop_p=0.0, seed=0, seed_dev=None):
    """bf16 x [N,..,C] -> (y bf16, mean, rstd fp32 [N,G]); None when the geometry is not covered"""
    N, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * C)
    if not h16_groupnorm_ok(HW, C, G):
        return None
    y = torch.empty_like(x)
    mean = torch.empty((N, G), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    st = lib().indm_h16_groupnorm_fwd(_ptr16(x), _ptr(gamma), _ptr(beta), y.data_ptr(), _ptr(mean), _ptr(rstd), N, HW, C, G,
                                      eps, act, drop_p, seed, _seed_ptr(seed_dev), _stream())
    _chk(st, "indm_h16_groupnorm_fwd")
    return y, mean, rstd


def h16_groupnorm_bwd(dy, x, gamma, beta, mean, rstd, G, act=0, drop_p=0.0, seed=0, seed_dev=None, add=None, add_scale=1.0):
    """add (optional, bf16, x's shape): dx += add_scale * add inside the same pass"""
    N, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * C)
    dx = torch.empty_like(x)
    dgamma = torch.empty_like(gamma)
    dbeta = torch.empty_like(beta)
    ws = _ws(lib().indm_h16_groupnorm_ws_floats(N, C), x.device)
    st = lib().indm_h16_groupnorm_bwd(_ptr16(dy), _ptr16(x), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd), dx.data_ptr(),
                                      _ptr(dgamma), _ptr(dbeta), N, HW, C, G, act, drop_p, seed, _seed_ptr(seed_dev),
                                      _ptr16(add), add_scale, _ptr(ws), _stream())
    _chk(st, "indm_h16_groupnorm_bwd")
    return dx, dgamma, dbeta


def h16_segreduce(x, rows_per_seg, alpha=1.0):
    """bf16 x viewed [rows, C] -> fp32 [rows / rows_per_seg, C] column sums per segment (C % 8 == 0)"""
    C = x.shape[-1]
    R = x.numel() // C
    if R % rows_per_seg:
        raise RuntimeError("h16_segreduce: rows do not divide into segments")
    nseg = R // rows_per_seg
    out = torch.empty((nseg, C), dtype=torch.float32, device=x.device)
    ws = _ws(lib().indm_h16_segreduce_ws_floats(nseg, rows_per_seg, C), x.device)
    _chk(lib().indm_h16_segreduce(_ptr16(x), _ptr(out), nseg, rows_per_seg, C, alpha, _ptr(ws), _stream()), "indm_h16_segreduce")
    return out


def h16_scale(x, a):
    if x.numel() % 8:
        return (x.float() * a).to(torch.bfloat16)
    y = torch.empty_like(x)
    _chk(lib().indm_h16_scale(_ptr16(x), y.data_ptr(), x.numel(), a, _stream()), "indm_h16_scale")
    return y


def h16_axpby_mixed(x32, y16, a, b):
    """bf16(a * x32 + b * y16)"""
    if x32.numel() % 8:
        return (a * x32 + b * y16.float()).to(torch.bfloat16)
    out = torch.empty_like(y16)
    _chk(lib().indm_h16_axpby_mixed(_ptr(x32), _ptr16(y16), out.data_ptr(), x32.numel(), a, b, _stream()), "indm_h16_axpby_mixed")
    return out


def h16_upsample2x(x, alpha=1.0):
    N, H, W, C = x.shape
    y = torch.empty((N, 2 * H, 2 * W, C), dtype=torch.bfloat16, device=x.device)
    _chk(lib().indm_h16_upsample2x(_ptr16(x), y.data_ptr(), N, H, W, C, alpha, _stream()), "indm_h16_upsample2x")
    return y


def h16_downsample2x(x, alpha=0.25):
    N, H, W, C = x.shape
    y = torch.empty((N, H // 2, W // 2, C), dtype=torch.bfloat16, device=x.device)
    _chk(lib().indm_h16_downsample2x(_ptr16(x), y.data_ptr(), N, H, W, C, alpha, _stream()), "indm_h16_downsample2x")
    return y


# --------------------------------------------------------------------------- block-level drivers (csrc/iresblock.hip)
class _IResFwdArgs(ctypes.Structure):
    """indm_iresblock_fwd_args (include/indm_hip.h)"""
    _fields_ = ([(n, c_i) for n in ("N", "H", "W", "C", "idim", "preact", "training", "bf16")]
                + [(n, c_p) for n in ("x", "cb", "W1", "W2T", "W2T_bf16w", "W3_ci_tapco", "W3_tapco_ci", "panel", "b1", "b2",
                                      "b3", "eps", "coeffs")]
                + [("ncoeff", c_i)]
                + [(n, c_p) for n in ("y", "logdet", "a0", "d0", "a1", "d1", "a2", "d2", "t3", "t2", "c1", "w")]
                + [("ws", c_p), ("ws_floats", c_l), ("splitk_ws", c_p), ("splitk_ws_floats", c_l), ("timer", c_p)]
                + [(n, c_p) for n in ("W2_bf16w", "panel_bf16w", "W3_bf16w", "W1_bf16w", "W3t_bf16w", "W3t_thin_bf16w", "W1_thin_bf16w")])


class _IResBwdArgs(ctypes.Structure):
    """indm_iresblock_bwd_args (include/indm_hip.h)"""
    _fields_ = ([(n, c_i) for n in ("N", "H", "W", "C", "idim", "preact", "bf16")]
                + [(n, c_p) for n in ("gy", "gld", "a0", "d0", "a1", "d1", "a2", "d2", "t3", "t2", "c1", "w", "eps", "W1", "W2",
                                      "W2T", "W3_tapco_ci", "dx", "dcb", "dW1", "dW2", "dW3", "dW2T", "db1", "db2", "db3")]
                + [("ws", c_p), ("ws_floats", c_l), ("splitk_ws", c_p), ("splitk_ws_floats", c_l)]
                + [(n, c_p) for n in ("W2_bf16w", "W2T_bf16w", "panel_bf16w", "W1_bf16w", "W3t_bf16w")]
                + [("side_stream", c_p), ("side_splitk_ws", c_p), ("side_splitk_ws_floats", c_l), ("W1_thin_bf16w", c_p)])


BLOCK_DRIVERS = {"on": os.environ.get("INDM_BLOCK_DRIVERS", "1") != "0"}   # tools / tests switch to the op-level composition


def block_drivers(on=None):
    """query / set whether flow.py runs an iResBlock through ONE C-ABI call per direction (indm_iresblock_*) or composes
    the op-level calls in Python (same kernels and results; ~55 host calls per block instead of 2)"""
    if on is not None:
        BLOCK_DRIVERS["on"] = bool(on)
    return BLOCK_DRIVERS["on"]


_ires_ws_sizes = {}


def _ires_ws(N, H, W, C, idim, which, device):
    key = (N, H, W, C, idim, which)
    n = _ires_ws_sizes.get(key)
    if n is None:
        n = _ires_ws_sizes[key] = int(lib().indm_iresblock_ws_floats(N, H, W, C, idim, which))
    # per-call scratch from torch's caching allocator (stream-ordered reuse; the block's working set, not the
    # process-wide split-K buffer: two blocks of different scales are in flight on the GPU back to back)
    return torch.empty(n, dtype=torch.float32, device=device), n


class ChainTimer:
    """HIP events around the chain-kernel launches the block drivers enqueue (bench.py's roofline line)."""
    CLASSES = {"ires_chain64x512x32,gated": 1, "ires_chain64x512x32,gated,t2out": 2,
               "ires_chain64x512x128,gated": 4, "ires_chain64x512x128,gated,t2out": 8}

    def __init__(self, capacity=4096):
        self.h = ctypes.c_void_p()
        _chk(lib().indm_timer_create(int(capacity), ctypes.byref(self.h)), "indm_timer_create")

    def reset(self, variant):
        """variant: an ires_chain class name (bf16 suffix ignored) -> True if the drivers can time it"""
        mask = self.CLASSES.get(variant.replace(",bf16", ""), 0)
        _chk(lib().indm_timer_reset(self.h, mask), "indm_timer_reset")
        return mask != 0

    def read(self):
        ms, n, fl = ctypes.c_float(), ctypes.c_int(), ctypes.c_double()
        _chk(lib().indm_timer_read(self.h, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)), "indm_timer_read")
        return dict(ms=ms.value, launches=n.value, flops=fl.value)

    def __del__(self):
        try:
            lib().indm_timer_destroy(self.h)
        except Exception:  # noqa: BLE001
            pass


_active_timer = [None]


def set_chain_timer(t):
    _active_timer[0] = t


def _w2t_bf16w(w2t):
    hit = _w2t_packed.get(id(w2t)) if _MFMA["dtype"] == "bf16" else None
    if hit is not None and hit[0]() is w2t and hit[1] == w2t._version:
        return hit[2].data_ptr()
    return None


def _f16_geometry(N, H, W, C, idim):
    M = N * H * W
    return (idim == 512 and (C <= 3 or (C <= 48 and C % 4 == 0)) and M >= 1024 and M % 128 == 0 and (H * W) % 64 == 0
            and M * 9 * C < (1 << 31))


def f16_wide(C):
    """12 < C <= 48 (C % 4 == 0): a flow scale the F16 drivers run on the im2col operand (indm_f16_tap_spread) instead of the
    gathering thin kernels -- the 48-channel scale of the 64 x 64 flows"""
    return 12 < C <= 48 and C % 4 == 0


# the series iterations of the F16 forward driver as one launch each (indm_f16_chain_conv); INDM_F16_CHAIN_CONV=0: thin conv + chain
CHAIN_CONV = {"on": os.environ.get("INDM_F16_CHAIN_CONV", "1") != "0"}
_kpad_packed = {}  # (id(w), kind) -> (weakref, version, packed bf16 copy of w as [9C -> 512, idim])


def pack_kpad_bf16(ws, kind):
    """F16 wide scales: the skinny 3x3 weights of a whole scale as K-padded packed GEMM operands, registered per weight object.
    kind 'w1': w [3,3,C,idim] -> [(tap,c), idim]; 'w3t': w [3,3,idim,C] -> [(tap,co), ci].  Rows zero-padded to 512.
    kinds 'w3t_thin' / 'w1_thin' (C <= 12): the 'w3t' / 'w1' matrix with its rows padded to 16 S only, in indm_f16_chain_conv's Wc order
    (k = 16 s + 8 (lane >> 5) + e: the thin convolution kernel's own fragment order)."""
    import weakref
    with torch.no_grad():
        st = torch.stack([w.detach() for w in ws])
        n, KH, KW, A, B = st.shape
        m = st.reshape(n, KH * KW * A, B) if kind in ("w1", "w1_thin") else st.permute(0, 1, 2, 4, 3).reshape(n, KH * KW * B, A)
        if kind in ("w3t_thin", "w1_thin"):
            S = (m.shape[1] + 15) // 16
            m = torch.nn.functional.pad(m, (0, 0, 0, 16 * S - m.shape[1])).to(torch.bfloat16)
            pk = m.view(n, S, 2, 8, m.shape[2] // 32, 32).permute(0, 1, 4, 2, 5, 3).contiguous()   # [n, s, nb, lhi, l31, e]
        else:
            m = torch.nn.functional.pad(m, (0, 0, 0, 512 - m.shape[1]))
    if kind not in ("w3t_thin", "w1_thin"):
        pk = pack_w2t_bf16(list(m.unbind(0)))
    for k in [k for k, v in _kpad_packed.items() if v[0]() is None]:
        del _kpad_packed[k]
    for i, w in enumerate(ws):
        if not isinstance(w, torch.nn.Parameter):
            _kpad_packed[(id(w), kind)] = (weakref.ref(w), w._version, pk[i])
    return pk


def packed_kpad_bf16(w, kind):
    hit = _kpad_packed.get((id(w), kind))
    if hit is None or hit[0]() is not w or hit[1] != w._version:
        return pack_kpad_bf16([w], kind)[0]
    return hit[2]


_F16_STATS = {}


def f16_stats(reset=False):
    """{(C, H, W): [blocks run on F16 (bf16-storage) kernels, blocks run on fp32 tensors]} of the block-driver forwards since
    the last reset -- which scales of a flow the bf16-storage path actually covers (tests assert on it)."""
    out = {k: list(v) for k, v in _F16_STATS.items()}
    if reset:
        _F16_STATS.clear()
    return out


def iresblock_fwd(x, cb, W1, W2T, W3, panel, b1, b2, b3, eps, coeffs, preact, training, W2=None):
    """One iResBlock forward + log-det estimate through indm_iresblock_fwd_logdet.
    -> None if the geometry is not covered, else (y, logdet, saved) with saved = the tensors the backward needs
    (training) or None (eval).  With bf16 storage (storage_dtype() == 'bf16', W2 given, geometry covered) the saved
    [N,H,W,idim] tensors are F16 buffers (flat bfloat16 tensors in MFMA-fragment order)."""
    N, H, W, C = x.shape
    idim = W2T.shape[0]
    if panel is None:
        return None
    dev = x.device
    M = N * H * W
    f16 = storage_dtype() == "bf16" and W2 is not None and _f16_geometry(N, H, W, C, idim)
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
    a = _IResFwdArgs()
    a.N, a.H, a.W, a.C, a.idim, a.preact, a.training = N, H, W, C, idim, int(preact), int(training)
    a.bf16 = 2 if f16 else int(_MFMA["dtype"] == "bf16")
    a.x, a.cb, a.W1, a.W2T = _ptr(x), _ptr(cb), _ptr(W1), _ptr(W2T)
    a.W2T_bf16w = _w2t_bf16w(W2T)
    w3ct = _packed_weight(W3, "ci_tapco")
    a.W3_ci_tapco, a.W3_tapco_ci = _ptr(w3ct), _ptr(_packed_weight(W3, "tapco_ci"))
    if f16:
        keep = [packed_bf16(W2T), packed_bf16(W2), packed_bf16(panel), packed_bf16(w3ct)]   # alive until the call returns
        a.W2T_bf16w, a.W2_bf16w, a.panel_bf16w, a.W3_bf16w = (t.data_ptr() for t in keep)
        if f16_wide(C):
            keep += [packed_kpad_bf16(W1, "w1"), packed_kpad_bf16(W3, "w3t")]
            a.W1_bf16w, a.W3t_bf16w = keep[-2].data_ptr(), keep[-1].data_ptr()
        elif CHAIN_CONV["on"]:
            keep += [packed_kpad_bf16(W3, "w3t_thin"), packed_kpad_bf16(W1, "w1_thin")]
            a.W3t_thin_bf16w, a.W1_thin_bf16w = keep[-2].data_ptr(), keep[-1].data_ptr()
    a.panel, a.b1, a.b2, a.b3, a.eps = _ptr(panel), _ptr(b1), _ptr(b2), _ptr(b3), _ptr(eps)
    cf = (ctypes.c_float * max(1, len(coeffs)))(*coeffs)
    a.coeffs, a.ncoeff = ctypes.cast(cf, c_p), len(coeffs)
    y, logdet = new(N, H, W, C), new(N)
    a.y, a.logdet = y.data_ptr(), logdet.data_ptr()
    saved = None
    if training:
        a0, d0, c1 = (new(N, H, W, C), new(N, H, W, C), new(N, H, W, C)) if preact else (x, None, None)
        if f16:
            a1, d1, a2, d2, t3, t2 = (torch.empty(M * idim, dtype=torch.bfloat16, device=dev) for _ in range(6))
        else:
            a1, d1, a2, d2, t3, t2 = (new(N, H, W, idim) for _ in range(6))
        w = new(N, H, W, C)
        if preact:
            a.a0, a.d0, a.c1 = a0.data_ptr(), d0.data_ptr(), c1.data_ptr()
        a.a1, a.d1, a.a2, a.d2, a.t3, a.t2, a.w = (t.data_ptr() for t in (a1, d1, a2, d2, t3, t2, w))
        saved = (a0, d0, a1, d1, a2, d2, t3, t2, c1, w)
    ws, n = _ires_ws(N, H, W, C, idim, (0 if training else 1) + (16 if f16 else 0), dev)
    a.ws, a.ws_floats = ws.data_ptr(), n
    sk = _ws(SPLITK_WS_FLOATS, dev)
    a.splitk_ws, a.splitk_ws_floats = sk.data_ptr(), SPLITK_WS_FLOATS
    t = _active_timer[0]
    a.timer = t.h if t is not None else None
    st = lib().indm_iresblock_fwd_logdet(ctypes.byref(a), _stream())
    if st == 1:
        return None
    _chk(st, "indm_iresblock_fwd_logdet")
    _F16_STATS.setdefault((C, H, W), [0, 0])[0 if f16 else 1] += 1    # counted only when the driver RAN the block
    return y, logdet, saved


# INDM_BWD_STREAMS=1: the fp32 / bf16-compute backward driver enqueues the block's weight gradients on a second stream
# (include/indm_hip.h, indm_iresblock_bwd_args.side_stream).  OFF by default: measured (round 6, profiles/r06_two_stream_ab.txt)
# 35 ms of the step's kernels do run concurrently and the step is 0.3 % SLOWER -- 85 % of the data path are MFMA-bound GEMMs, so
# the side stream's GEMMs mostly share the MFMA pipe with them, and its HBM-bound launches take LDS / CU slots from them.
BWD_STREAMS = {"on": os.environ.get("INDM_BWD_STREAMS", "0") == "1"}
# the same for the bf16-storage (F16) backward driver, whose weight-gradient launches are short and latency-bound at the 16 x 16
# scale and at the per-rank shards: INDM_BWD_STREAMS_F16
BWD_STREAMS_F16 = {"on": os.environ.get("INDM_BWD_STREAMS_F16", "0") == "1"}
_side = {}


def bwd_streams(on=None):
    if on is not None:
        BWD_STREAMS["on"] = bool(on)
    return BWD_STREAMS["on"]


def _side_stream(dev):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    hit = _side.get(key)
    if hit is None:
        hit = _side[key] = (torch.cuda.Stream(device=dev), torch.empty(SPLITK_WS_FLOATS, dtype=torch.float32, device=dev))
    return hit


def iresblock_bwd(gy, gld, saved, eps, W1, W2, W2T, W3, preact, panel=None):
    """Backward of the training-mode block through indm_iresblock_bwd -> (dx, dcb, dW1, dW2, dW3, dW2T, db1, db2, db3)
    or None if not covered."""
    a0, d0, a1, d1, a2, d2, t3, t2, c1, w = saved
    N, H, W, C = eps.shape
    f16 = a1.dtype == torch.bfloat16        # F16 buffers saved by the bf16-storage forward
    idim = W2.shape[0]
    dev = eps.device
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
    a = _IResBwdArgs()
    a.N, a.H, a.W, a.C, a.idim, a.preact = N, H, W, C, idim, int(preact)
    a.bf16 = 2 if f16 else int(_MFMA["dtype"] == "bf16")
    a.gy, a.gld = _ptr(gy), _ptr(gld)
    a.a0, a.d0, a.c1 = _ptr(a0), _ptr(d0), _ptr(c1)
    if f16:
        a.a1, a.d1, a.a2, a.d2, a.t3, a.t2 = (_ptr16(t) for t in (a1, d1, a2, d2, t3, t2))
        a.w, a.eps = _ptr(w), _ptr(eps)
        if panel is None:
            return None
        keep = [packed_bf16(W2), packed_bf16(W2T), packed_bf16(panel)]
        a.W2_bf16w, a.W2T_bf16w, a.panel_bf16w = (t.data_ptr() for t in keep)
        if f16_wide(C):
            keep += [packed_kpad_bf16(W1, "w1"), packed_kpad_bf16(W3, "w3t")]
            a.W1_bf16w, a.W3t_bf16w = keep[-2].data_ptr(), keep[-1].data_ptr()
        elif CHAIN_CONV["on"]:
            keep.append(packed_kpad_bf16(W1, "w1_thin"))
            a.W1_thin_bf16w = keep[-1].data_ptr()
    else:
        a.a1, a.d1, a.a2, a.d2, a.t3, a.t2, a.w, a.eps = (_ptr(t) for t in (a1, d1, a2, d2, t3, t2, w, eps))
    a.W1, a.W2, a.W2T, a.W3_tapco_ci = _ptr(W1), _ptr(W2), _ptr(W2T), _ptr(_packed_weight(W3, "tapco_ci"))
    out = (new(N, H, W, C), new(N, idim), new(3, 3, C, idim), new(idim, idim), new(3, 3, idim, C), new(idim, idim),
           new(idim), new(idim), new(C))
    a.dx, a.dcb, a.dW1, a.dW2, a.dW3, a.dW2T, a.db1, a.db2, a.db3 = (t.data_ptr() for t in out)
    ws, n = _ires_ws(N, H, W, C, idim, 2 + (16 if f16 else 0), dev)
    a.ws, a.ws_floats = ws.data_ptr(), n
    sk = _ws(SPLITK_WS_FLOATS, dev)
    a.splitk_ws, a.splitk_ws_floats = sk.data_ptr(), SPLITK_WS_FLOATS
    if (BWD_STREAMS["on"] or (f16 and BWD_STREAMS_F16["on"])) and not torch.cuda.is_current_stream_capturing():
        side = _side_stream(dev)
        a.side_stream, a.side_splitk_ws, a.side_splitk_ws_floats = side[0].cuda_stream, side[1].data_ptr(), SPLITK_WS_FLOATS
    st = lib().indm_iresblock_bwd(ctypes.byref(a), _stream())
    if st == 1:
        return None
    _chk(st, "indm_iresblock_bwd")
    return out


def iresblock_inverse(y, cb, W1, W2T, W3, b1, b2, b3, preact, atol, rtol):
    """Fixed-point inverse of one block through indm_iresblock_inverse -> (x, iterations) or None if not covered."""
    N, H, W, C = y.shape
    idim = W2T.shape[0]
    dev = y.device
    a = _IResFwdArgs()
    a.N, a.H, a.W, a.C, a.idim, a.preact, a.training = N, H, W, C, idim, int(preact), 0
    a.bf16 = int(_MFMA["dtype"] == "bf16")
    a.x, a.cb, a.W1, a.W2T = _ptr(y), _ptr(cb), _ptr(W1), _ptr(W2T)
    a.W3_ci_tapco = _ptr(_packed_weight(W3, "ci_tapco"))
    a.b1, a.b2, a.b3 = _ptr(b1), _ptr(b2), _ptr(b3)
    x = torch.empty_like(y)
    a.y = x.data_ptr()
    ws, n = _ires_ws(N, H, W, C, idim, 3, dev)
    a.ws, a.ws_floats = ws.data_ptr(), n
    sk = _ws(SPLITK_WS_FLOATS, dev)
    a.splitk_ws, a.splitk_ws_floats = sk.data_ptr(), SPLITK_WS_FLOATS
    it = ctypes.c_int(0)
    st = lib().indm_iresblock_inverse(ctypes.byref(a), atol, rtol, ctypes.byref(it), _stream())
    if st == 1:
        return None
    _chk(st, "indm_iresblock_inverse")
    return x, it.value
