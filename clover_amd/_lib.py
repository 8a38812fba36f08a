"""ctypes binding of libclover_hip.so — the C ABI declared in include/clover_hip.h.

The product path has NO CPU fallback: every op in ``clover_amd.ops`` goes through this
library, and a missing/unloadable library raises at first use (``lib()``).  Build it with
``python __graft_entry__.py`` (or ``make -C clover_amd/csrc``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The 16-bit element type of activations, weight shadows and gradients.  Default (round 6): IEEE fp16 — the reference's own
# arithmetic type (configs/exp_local/pretrain_webvid_cc3m.py:21 fp16 = dict(loss_scale='dynamic'),
# mmaction/core/hooks/fp16_utils.py:215-259): three more significand bits than bf16 at the same MFMA rate, which puts the
# step's losses within ~2e-3 of the fp32 reference (bf16: 1.8e-2) at the same step time; the engine applies a static loss
# scale.  CLOVER_HALF=bf16 selects the bf16 build of the same kernels (libclover_hip.so; no loss scale needed).  The same
# sources are compiled twice (csrc/Makefile: -DCLV_HALF_F16); HALF / half_dtype() is the torch dtype every module stores in.
HALF_F16 = os.environ.get('CLOVER_HALF', 'f16').lower() in ('f16', 'fp16', 'float16', 'half')
LIB_PATH = os.environ.get('CLOVER_LIB_PATH') or os.path.join(_HERE, 'libclover_hip_f16.so' if HALF_F16 else 'libclover_hip.so')


# Static loss scale of the 16-bit backward (mmcv_Fp16OptimizerHook.py:96-149; the reference's fp16 = dict(loss_scale='dynamic')):
# fp16 activation gradients of 1e-7 .. 1e-4 would sit in the subnormal range, so the recognizer's _parse_losses multiplies
# the gradient at the ROOT of every backward by this factor (a power of two: exact) and whoever consumes parameter gradients
# divides it out — the engine inside its norm / AdamW kernels (their grad_scale factor), plain autograd users through a
# per-parameter gradient hook the recognizer registers (so `loss.backward()` leaves true-scale .grad tensors either way).
# bf16 has fp32's exponent range and needs none.
# CLOVER_LOSS_SCALE: a number (static) or 'dynamic' (the reference's LossScaler(mode='dynamic') — engine only: it lives on
# the device next to the optimizer's scalars; plain autograd then uses the static default).
_scale_env = os.environ.get('CLOVER_LOSS_SCALE', '1024' if HALF_F16 else '1')
LOSS_SCALE_SPEC = 'dynamic' if _scale_env.strip().lower() == 'dynamic' else float(_scale_env)
LOSS_SCALE = (1024.0 if HALF_F16 else 1.0) if LOSS_SCALE_SPEC == 'dynamic' else LOSS_SCALE_SPEC


def half_dtype():
    import torch
    return torch.float16 if HALF_F16 else torch.bfloat16

ABI_VERSION = 17

ERRORS = {-1: 'CLV_ERR_ARG (bad argument)', -2: 'CLV_ERR_UNSUPPORTED (shape not supported by the kernels)',
          -3: 'CLV_ERR_LAUNCH (HIP launch failed)'}


class ClvAttnGeom(C.Structure):
    """Mirror of ``struct ClvAttnGeom`` (include/clover_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in
                ('mode', 'groups', 'N', 'nH', 'hd', 'D', 'H', 'W', 'wd', 'wh', 'ww', 'sd', 'sh', 'sw',
                 'ldq', 'ldk', 'ldv', 'ldo', 'bwd', 'bwh', 'bww')] + [('scale', C.c_float), ('dropout_p', C.c_float),
                                                                      ('dbias_index', C.c_void_p), ('work', C.c_void_p)]


_p, _i32, _i64, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class ClvFoldEntry(C.Structure):
    _fields_ = [('partial', _p), ('dw', _p), ('db', _p), ('nk', _i64), ('e2', _i64), ('splits', _i32),
                ('sg_shift', _i32), ('block_begin', _i32), ('overwrite', _i32)]


FOLD_MAX = 64


class ClvWgradEntry(C.Structure):
    _fields_ = [('dy', _p), ('x', _p), ('work', _p), ('dw', _p), ('db', _p), ('M', _i64), ('work_floats', _i64),
                ('N', _i32), ('K', _i32),
                ('ldy', _i32), ('ldx', _i32), ('want_bias', _i32), ('splits', _i32), ('overwrite', _i32), ('pad', _i32)]


WGRAD_GROUP_MAX = min(80, max(1, int(os.environ.get('CLOVER_WGRAD_GROUP_MAX', '80'))))     # problems per grouped launch


class ClvDbiasGather(C.Structure):
    """Mirror of ``struct ClvDbiasGather`` (include/clover_hip.h): one pending table-gradient gather."""
    _fields_ = [('partial', _p), ('dtable', _p), ('index', _p), ('split_stride', _i64), ('nkt', _i32), ('nH', _i32),
                ('N', _i32), ('nsplit', _i32), ('slot0', _i32), ('nslots', _i32), ('block_begin', _i32), ('pad', _i32)]


DBIAS_GATHER_MAX = 32


class ClvLnReduceEntry(C.Structure):
    _fields_ = [('partial', _p), ('dgamma', _p), ('dbeta', _p), ('nblk', _i32), ('C', _i32), ('block_begin', _i32),
                ('pad', _i32)]


class ClvLnExtra(C.Structure):
    """Mirror of ``struct ClvLnExtra`` (include/clover_hip.h)."""
    _fields_ = [('xscale', _p), ('rows_per_sample', _i32), ('drop_p', _f), ('seed', _p), ('dy2', _p), ('dres', _p),
                ('x_is_sum', _i32), ('gather_c', _i32), ('gather_h2', _i32), ('gather_w2', _i32), ('no_reduce', _i32),
                ('q8', _p), ('qscale', _p)]


# name -> (restype, argtypes); must list EVERY symbol include/clover_hip.h declares
SIGNATURES = {
    'clv_abi_version': (C.c_int, []),
    'clv_half_type': (C.c_int, []),
    'clv_attn_fwd': (C.c_int, [_p] * 9 + [C.POINTER(ClvAttnGeom), _p]),
    'clv_attn_seq_work_bytes': (C.c_int64, [C.POINTER(ClvAttnGeom)]),
    'clv_attn_seq_max_keys': (C.c_int, []),
    'clv_attn_bwd_work_bytes': (C.c_int64, [C.POINTER(ClvAttnGeom)]),
    'clv_attn_bwd_one_kernel': (C.c_int, [C.POINTER(ClvAttnGeom)]),
    'clv_attn_dbias_index_count': (C.c_int64, [C.POINTER(ClvAttnGeom)]),
    'clv_attn_dbias_index': (C.c_int, [C.POINTER(ClvAttnGeom), _p, _p]),
    'clv_attn_dbias_gather_entry': (C.c_int, [C.POINTER(ClvAttnGeom), _p, _p, C.POINTER(ClvDbiasGather)]),
    'clv_attn_dbias_partial_bytes': (C.c_int64, [_p]),
    'clv_attn_dbias_gather_batch': (C.c_int, [C.POINTER(ClvDbiasGather), _i32, _p]),
    'clv_attn_bwd': (C.c_int, [_p] * 16 + [_i32, C.POINTER(ClvAttnGeom), _p]),
    'clv_softmax_rows_fwd': (C.c_int, [_p] * 5 + [_i64, _i32, _i32, _i32, _f, _f, _p]),
    'clv_softmax_rows_bwd': (C.c_int, [_p] * 4 + [_i64, _i32, _i32, _f, _f, _p]),
    'clv_layernorm_fwd': (C.c_int, [_p] * 8 + [_i64, _i32, _f, _i32, C.POINTER(ClvLnExtra), _p]),
    'clv_layernorm_bwd_blocks': (C.c_int, [_i64, _i32]),
    'clv_layernorm_bwd': (C.c_int, [_p] * 11 + [_i64, _i32, _i32, C.POINTER(ClvLnExtra), _p]),
    'clv_ln_fold_fwd': (C.c_int, [_p] * 6 + [_i32, _i32, _p]),
    'clv_ln_fold_bwd': (C.c_int, [_p] * 9 + [_i32, _i32, _p]),
    'clv_batchnorm1d_fwd': (C.c_int, [_p] * 8 + [_i32, _i32, _f, _f, _i32, _p]),
    'clv_batchnorm1d_bwd': (C.c_int, [_p] * 8 + [_i32, _i32, _i32, _p]),
    'clv_gelu_fwd': (C.c_int, [_p, _p, _i64, _i32, _p]),
    'clv_gelu_bwd': (C.c_int, [_p, _p, _p, _i64, _i32, _p]),
    'clv_patch_embed_fwd': (C.c_int, [_p] * 12 + [_i32] * 7 + [_f, _p]),
    'clv_patch_embed_fwd_fp8': (C.c_int, [_p] * 13 + [_i32] * 7 + [_f, _p]),
    'clv_patch_embed_blend_bwd': (C.c_int, [_p] * 5 + [_i32] * 7 + [_p]),
    'clv_im2col_patches': (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _p]),
    'clv_linear_wgrad_work_floats': (C.c_int64, [_i64, _i32, _i32]),
    'clv_linear_wgrad': (C.c_int, [_p] * 5 + [_i64, _i32, _i32, _i32, _i32, _p, _p, _i32, _p]),
    'clv_rowgemm_supported': (C.c_int, [_i32, _i32, _i32]),
    'clv_rowgemm': (C.c_int, [_p] * 11 + [_i64, _i32, _i32, _i32, _i32, _i32, _i32, _f, _p]),
    'clv_rowgemm_xs': (C.c_int, [_p] * 11 + [_i64, _i32, _i32, _i32, _i32, _i32, _i32, _f, _p, _i32, _p]),
    'clv_mlp_fused_supported': (C.c_int, [_i32, _i32]),
    'clv_mlp_fused_fwd': (C.c_int, [_p] * 10 + [_i64, _i32, _i32, _f, _p, _i32, _p]),
    'clv_mlp_fused_bwd': (C.c_int, [_p] * 13 + [_i64, _i32, _i32, _p, _i32, _p]),
    'clv_colsum': (C.c_int, [_p, _p, _i64, _i32, _i32, _p]),
    'clv_focal_ce_fwd': (C.c_int, [_p, _i32, _p, _p, _p, _p, _p, _i64, _i32, _f, _p]),
    'clv_focal_ce_bwd': (C.c_int, [_p, _i32, _p, _p, _p, _p, _p, _p, _i64, _i32, _f, _p]),
    'clv_focal_ce_fwd_ld': (C.c_int, [_p, _i32, _p, _p, _p, _p, _p, _i64, _i32, _i64, _f, _p]),
    'clv_focal_ce_bwd_ld': (C.c_int, [_p, _i32, _p, _p, _p, _p, _p, _p, _i64, _i32, _i64, _f, _p]),
    'clv_infonce_work_floats': (C.c_int64, [_i32, _i32]),
    'clv_infonce_fwd': (C.c_int, [_p] * 6 + [_i32, _i32, _i32, _f, _f, _p]),
    'clv_infonce_bwd': (C.c_int, [_p] * 10 + [_i32, _i32, _i32, _f, _f, _p]),
    'clv_infonce_pair_fwd': (C.c_int, [_p] * 4 + [_i32, _i32, _i32, _f, _f, _p]),
    'clv_infonce_pair_bwd': (C.c_int, [_p] * 4 + [_i32, _i32, _i32, _f, _f, _p]),
    'clv_normsoftmax_work_floats': (C.c_int64, [_i32, _i32]),
    'clv_normsoftmax_fwd': (C.c_int, [_p] * 5 + [_i32, _i32, _f, _f, _p]),
    'clv_normsoftmax_bwd': (C.c_int, [_p] * 6 + [_i32, _i32, _f, _p]),
    'clv_sumsq': (C.c_int, [_p, _p, _i64, _p]),
    'clv_adamw_step': (C.c_int, [_p] * 6 + [_i64] + [_f] * 9 + [_p]),
    'clv_gemm_nt_supported': (C.c_int, [_i64, _i32, _i32]),
    'clv_gemm_nt': (C.c_int, [_p] * 6 + [_i64, _i32, _i32, _i64, _i64, _i64, _i32, _p]),
    'clv_gemm_nt_work_bytes': (C.c_int64, [_i64, _i32, _i32]),
    'clv_gemm_nt_ex': (C.c_int, [_p] * 6 + [_i64, _i32, _i32, _i64, _i64, _i64, _i32, _p, _i64, _p]),
    'clv_transpose_batch': (C.c_int, [_p, _p, _p, _i32, _i32, _p]),
    'clv_layernorm_bwd_needs_reduce': (C.c_int, [_i64, _i32]),
    'clv_ln_reduce_batch': (C.c_int, [_p, _i32, _p]),
    'clv_linear_wgrad_in_place': (C.c_int, [C.c_int64, C.c_int32, C.c_int32]),
    'clv_linear_wgrad_class': (C.c_int, [C.c_int64, C.c_int32, C.c_int32]),
    'clv_linear_wgrad_batch_plan': (C.c_int, [_p, _i32]),
    'clv_linear_wgrad_batch': (C.c_int, [_p, _i32, _p]),
    'clv_linear_wgrad_batch_ss': (C.c_int, [_p, _i32, _p, _p]),
    'clv_linear_wgrad_splits': (C.c_int, [_i64, _i32, _i32]),
    'clv_wgrad_fold_batch': (C.c_int, [_p, _i32, _p]),
    'clv_wgrad_fold_batch_ss': (C.c_int, [_p, _i32, _p, _p]),
    'clv_optim_prep_slots': (C.c_int, [_p, _p, _p] + [_f] * 4 + [_p]),
    'clv_sumsq_ranges': (C.c_int, [_p, _p, _i32, _p, _p]),
    'clv_optim_prep': (C.c_int, [_p, _p] + [_f] * 4 + [_p]),
    'clv_adamw_step_dev': (C.c_int, [_p] * 6 + [_i64] + [_f] * 5 + [_p]),
    'clv_quant_fp8_rows': (C.c_int, [_p, _p, _p, _i64, _i32, _i64, _i64, _p]),
    'clv_gemm_nt_fp8_supported': (C.c_int, [_i64, _i32, _i32]),
    'clv_gemm_nt_fp8': (C.c_int, [_p] * 7 + [_i64, _i32, _i32, _i64, _i64, _i64, _i32, _p]),
    'clv_pack_bf16': (C.c_int, [_p, _p, _i64, _p]),
    'clv_sumsq_bf16': (C.c_int, [_p, _p, _i64, _p]),
    'clv_adamw_step_dev_bf16g': (C.c_int, [_p] * 6 + [_i64] + [_f] * 5 + [_p]),
    'clv_sgemm_strided': (C.c_int, [_p] * 4 + [_i64, _i32, _i32, _i64, _i64, _i64, _i64, _i64, _i32, _p]),
    'clv_sgemm_strided_rowsum': (C.c_int, [_p] * 4 + [_i64, _i32, _i32, _i64, _i64, _i64, _i64, _i64, _i32, _p]),
    'clv_sgemm_nt': (C.c_int, [_p] * 4 + [_i64, _i32, _i32, _i64, _i64, _i64, _p]),
    'clv_attn_f32_fwd': (C.c_int, [_p] * 7 + [C.POINTER(ClvAttnGeom), _i32, _p]),
    'clv_attn_f32_bwd_work_floats': (C.c_int64, [C.POINTER(ClvAttnGeom)]),
    'clv_attn_f32_bwd': (C.c_int, [_p] * 12 + [C.POINTER(ClvAttnGeom), _p]),
}

_lib = None


def lib():
    """The loaded library; raises (loudly) when it is missing — there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'clover_amd: {LIB_PATH} is missing — the HIP extension is not built. '
                'Run `python __graft_entry__.py` (hipcc --offload-arch=gfx950). There is no CPU fallback.')
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)           # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        v = L.clv_abi_version()
        if v != ABI_VERSION:
            raise RuntimeError(f'clover_amd: ABI version mismatch (library {v}, binding {ABI_VERSION}); rebuild')
        if L.clv_half_type() != int(HALF_F16):
            raise RuntimeError(f'clover_amd: {LIB_PATH} computes in {"fp16" if L.clv_half_type() else "bf16"} but '
                               f'CLOVER_HALF asks for {"fp16" if HALF_F16 else "bf16"}')
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f'clover_amd: {what} failed: {ERRORS.get(rc, rc)}')
