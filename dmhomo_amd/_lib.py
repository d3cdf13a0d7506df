"""ctypes binding of libdmhomo_hip.so (C ABI declared in include/dmhomo_hip.h).

The product path has no CPU or PyTorch-eager fallback: if the shared library is
missing or a symbol cannot be resolved, importing / calling fails loudly.
Tensors cross the boundary as raw device pointers (``tensor.data_ptr()``) plus
sizes; the HIP stream is torch's current stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DMH_LIB_PATH: development knob for A/B runs of two builds in one process environment (same GPU, same clocks)
LIB_PATH = os.environ.get('DMH_LIB_PATH') or os.path.join(_HERE, 'libdmhomo_hip.so')

c_f32p = C.c_void_p
c_i64 = C.c_int64
c_int = C.c_int
c_float = C.c_float


ABI_VERSION = 500          # include/dmhomo_hip.h: DMH_ABI_VERSION


class DmhConv(C.Structure):
    _fields_ = [('struct_size', C.c_uint64)] + \
               [(n, C.c_void_p) for n in ('src0', 'src1', 'wpack', 'bias', 'in_coef', 'res', 'res_coef', 'out',
                                          'stats')] + \
               [(n, C.c_int32) for n in ('B', 'Hin', 'Win', 'C0', 'C1', 'Cout', 'KH', 'KW', 'stride', 'upsample2')] + \
               [('in_bound', C.c_void_p), ('in_bound_n', C.c_int32), ('fin_n', C.c_int32), ('fin_w', C.c_void_p),
                ('fin_b', C.c_void_p), ('fin_out', C.c_void_p), ('pix_stats', C.c_void_p), ('pix_eps', C.c_float),
                ('rows', C.c_void_p)]


class DmhPackJob(C.Structure):
    _fields_ = [('src', C.c_void_p), ('ws', C.c_void_p), ('wpack', C.c_void_p), ('Cout', C.c_int32), ('C0', C.c_int32),
                ('C1', C.c_int32), ('KH', C.c_int32), ('transposed', C.c_int32)]


class DmhStep(C.Structure):
    _fields_ = [('objective', C.c_int32), ('clip', C.c_int32), ('mode', C.c_int32), ('cond_scale', C.c_float),
                ('sqrt_recip_ac', C.c_float), ('sqrt_recipm1_ac', C.c_float), ('sqrt_ac', C.c_float),
                ('sqrt_1m_ac', C.c_float), ('c0', C.c_float), ('c1', C.c_float), ('c2', C.c_float)]


# name -> (restype, argtypes); every symbol include/dmhomo_hip.h declares
SIGNATURES = {
    'dmh_last_error': (C.c_char_p, []),
    'dmh_version': (c_int, []),
    'dmh_ws_standardize': (c_int, [c_f32p, c_f32p, c_int, c_int, c_float, C.c_void_p]),
    'dmh_conv_pack_floats': (c_i64, [c_int, c_int, c_int, c_int, c_int]),
    'dmh_pack_conv_weight': (c_int, [c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, C.c_void_p]),
    'dmh_pack_conv_weights_multi': (c_int, [C.c_void_p, c_int, c_float, C.c_void_p]),
    'dmh_conv_up2_pack_floats': (c_i64, [c_int, c_int]),
    'dmh_pack_conv_weight_up2': (c_int, [c_f32p, c_f32p, c_int, c_int, C.c_void_p]),
    'dmh_conv_tiles': (c_int, [c_int, c_int, c_int, c_int]),
    'dmh_conv2d': (c_int, [C.POINTER(DmhConv), C.c_void_p]),
    'dmh_rows_from_keep': (c_int, [C.c_void_p, c_int, c_int, C.c_void_p, C.c_void_p]),
    'dmh_gn_finalize': (c_int, [c_f32p, c_int, c_f32p, c_f32p, c_f32p, c_i64, c_f32p, c_int, c_int, c_int, c_int,
                                c_float, C.c_void_p, C.c_void_p]),
    'dmh_gn_finalize_bound': (c_int, [c_f32p, c_int, c_f32p, c_f32p, c_f32p, c_i64, c_f32p, c_f32p, c_int, c_int, c_int,
                                      c_int, c_float, C.c_void_p, C.c_void_p]),
    'dmh_gn_silu_residual': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, C.c_void_p, C.c_void_p]),
    'dmh_gn_silu_residual_stats': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_float, C.c_void_p,
                                           C.c_void_p]),
    'dmh_chan_layernorm': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_int, c_float, C.c_void_p, c_i64, C.c_void_p]),
    'dmh_linattn_splits': (c_int, [c_int]),
    'dmh_linattn_partial_floats': (c_i64, [c_int, c_int]),
    'dmh_linattn_context': (c_int, [c_f32p, c_f32p, c_int, c_int, C.c_void_p, C.c_void_p]),
    'dmh_linattn_merge': (c_int, [c_f32p, c_f32p, c_int, c_int, C.c_void_p, C.c_void_p]),
    'dmh_linattn_apply': (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_float, C.c_void_p, C.c_void_p]),
    'dmh_conv_wgrad_workspace_floats': (c_i64, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'dmh_conv_wgrad': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_int, C.c_void_p]),
    'dmh_s2d_shift': (c_int, [c_f32p, c_f32p, c_int, c_int, c_int, c_int, C.c_void_p]),
    'dmh_d2s': (c_int, [c_f32p, c_f32p, c_int, c_int, c_int, c_int, C.c_void_p]),
    'dmh_sumpool2': (c_int, [c_f32p, c_f32p, c_int, c_int, c_int, c_int, C.c_void_p]),
    'dmh_gn_finalize_train': (c_int, [c_f32p, c_int, c_f32p, c_f32p, c_f32p, c_i64, c_f32p, c_f32p, c_int, c_int, c_int,
                                      c_int, c_float, C.c_void_p]),
    'dmh_gn_bwd_chunks': (c_int, [c_int]),
    'dmh_gn_silu_backward': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_f32p, c_f32p, c_f32p,
                                     c_f32p, c_int, c_int, c_int, c_int, C.c_void_p]),
    'dmh_sum_over_batch': (c_int, [c_f32p, c_f32p, c_int, c_i64, C.c_void_p]),
    'dmh_ws_backward': (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_float, C.c_void_p]),
    'dmh_lnb_blocks': (c_int, []),
    'dmh_chan_layernorm_backward': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_int, c_float, C.c_void_p]),
    'dmh_linattn_merge_ms': (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, C.c_void_p]),
    'dmh_linattn_bwd_workspace_floats': (c_i64, [c_int, c_int]),
    'dmh_linattn_backward': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_float, C.c_void_p]),
    'dmh_bgemm': (c_int, [c_f32p, C.POINTER(c_i64), c_f32p, C.POINTER(c_i64), c_f32p, C.POINTER(c_i64), c_int, c_int, c_int,
                          c_int, c_int, c_float, C.c_void_p]),
    'dmh_softmax_rows': (c_int, [c_f32p, c_f32p, c_i64, c_int, C.c_void_p]),
    'dmh_softmax_rows_backward': (c_int, [c_f32p, c_f32p, c_i64, c_int, C.c_void_p]),
    'dmh_act': (c_int, [c_f32p, c_f32p, c_f32p, c_i64, c_int, C.c_void_p]),
    'dmh_class_embed_backward': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, C.c_void_p]),
    'dmh_loss_backward': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int,
                                  C.c_void_p]),
    'dmh_sumsq_blocks': (c_int, []),
    'dmh_sumsq': (c_int, [c_f32p, c_i64, c_f32p, C.c_void_p]),
    'dmh_gradnorm_finalize': (c_int, [c_f32p, c_int, c_float, c_f32p, C.c_void_p]),
    'dmh_adam': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_float, c_float, c_float, c_float, c_int,
                         C.c_void_p]),
    'dmh_ema': (c_int, [c_f32p, c_f32p, c_i64, c_float, C.c_void_p]),
    'dmh_multi_blocks': (c_i64, [C.c_void_p, c_int]),
    'dmh_sumsq_multi': (c_int, [C.c_void_p, C.c_void_p, c_int, c_f32p, C.c_void_p]),
    'dmh_adam_multi': (c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_int, c_f32p, c_float, c_float,
                               c_float, c_float, c_int, C.c_void_p]),
    'dmh_resize_bilinear_u8': (c_int, [C.c_void_p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_i64, c_float,
                                       C.c_void_p]),
    'dmh_mask_open_nearest': (c_int, [c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_i64, C.c_void_p]),
    'dmh_pixel_stats': (c_int, [c_f32p, c_f32p, c_i64, c_int, c_float, C.c_void_p, c_i64, C.c_void_p]),
    'dmh_linattn_fused_pack_floats': (c_i64, [c_int]),
    'dmh_linattn_fused_pack': (c_int, [c_f32p, c_f32p, c_int, C.c_void_p]),
    'dmh_linattn_fused_splits': (c_int, [c_int, c_int]),
    'dmh_linattn_fused_context': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, C.c_void_p, C.c_void_p]),
    'dmh_linattn_out_pack_floats': (c_i64, []),
    'dmh_linattn_out_pack': (c_int, [c_f32p, c_f32p, C.c_void_p]),
    'dmh_linattn_fused_apply_out': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int,
                                            c_int, c_int, c_float, c_float, C.c_void_p, C.c_void_p]),
    'dmh_linattn_merge_n': (c_int, [c_f32p, c_f32p, c_int, c_int, c_int, C.c_void_p, C.c_void_p]),
    'dmh_linattn_fused_apply': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_float,
                                        C.c_void_p, C.c_void_p]),
    'dmh_attention': (c_int, [c_f32p, c_f32p, c_int, c_int, c_float, C.c_void_p, C.c_void_p]),
    'dmh_sinusoidal_embed': (c_int, [C.c_void_p, c_f32p, c_f32p, c_int, c_int, C.c_void_p]),
    'dmh_fourier_embed': (c_int, [C.c_void_p, c_f32p, c_f32p, c_int, c_int, C.c_void_p]),
    'dmh_class_embed': (c_int, [C.c_void_p, C.c_void_p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, C.c_void_p]),
    'dmh_linear': (c_int, [c_f32p, c_i64, c_f32p, c_f32p, c_f32p, c_i64, c_int, c_int, c_int, c_int, c_int,
                           C.c_void_p]),
    'dmh_ss_gather': (c_int, [c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_void_p, C.c_void_p, c_int, c_f32p, c_int, c_int, c_int,
                              C.c_void_p]),
    'dmh_assemble_input': (c_int, [c_f32p, c_int, c_f32p, c_int, c_f32p, c_f32p, c_int, c_int, c_int, c_int,
                                   C.c_void_p]),
    'dmh_final_conv_nchw': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, C.c_void_p]),
    'dmh_sampler_step': (c_int, [C.POINTER(DmhStep), c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64,
                                 C.c_void_p, c_i64, C.c_void_p]),
    'dmh_sampler_step_dev': (c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64, C.c_void_p,
                                     c_i64, C.c_void_p]),
    'dmh_sampler_seek': (c_int, [C.c_void_p, c_int, C.c_void_p, C.c_void_p, c_int, C.c_void_p, C.c_void_p, c_int,
                                 C.c_void_p]),
    'dmh_rng_indexed': (c_int, [c_f32p, c_int, c_i64, C.c_void_p, C.c_void_p, c_int, C.c_void_p]),
    'dmh_rng_keep_mask': (c_int, [C.c_void_p, c_int, C.c_void_p, C.c_void_p, c_float, C.c_void_p]),
    'dmh_rows_lincomb': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_i64, c_int, C.c_void_p]),
    'dmh_pixel_grid': (c_int, [c_f32p, c_int, c_int, c_int, c_float, C.c_void_p]),
    'dmh_norm_grid': (c_int, [c_f32p, c_f32p, c_int, c_int, c_int, C.c_void_p]),
    'dmh_homography_flow_points': (c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_int, c_int, c_int, c_int, c_int, C.c_void_p]),
    'dmh_dlt_points': (c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_int, c_int, C.c_void_p]),
    'dmh_affine': (c_int, [c_f32p, c_f32p, c_float, c_float, c_i64, C.c_void_p]),
    'dmh_affine_tail': (c_int, [c_f32p, c_int, c_int, c_int, c_int, c_float, c_float, C.c_void_p]),
    'dmh_q_sample': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_i64, C.c_void_p]),
    'dmh_diff_mean': (c_int, [c_f32p, c_f32p, c_f32p, c_int, C.c_void_p, c_f32p, c_int, c_int, c_int, C.c_void_p]),
    'dmh_loss_combine': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, C.c_void_p]),
    'dmh_to_uint8': (c_int, [c_f32p, C.c_void_p, c_i64, C.c_void_p]),
    'dmh_homography_flow': (c_int, [C.c_void_p, c_f32p, c_f32p, c_int, c_int, c_int, c_float, C.c_void_p]),
    'dmh_flow_to_image': (c_int, [c_f32p, c_f32p, c_int, c_int, c_float, C.c_void_p]),
    'dmh_flow_warp': (c_int, [c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                              C.c_void_p]),
    'dmh_dlt_homography': (c_int, [c_f32p, C.c_void_p, C.c_void_p, c_int, c_int, c_int, C.c_void_p]),
}

DLT_BLOCKS = 64

_lib = None


class DmhError(RuntimeError):
    pass


def lib():
    """the loaded shared library; raises if it was not built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DmhError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           f'or `make -C dmhomo_amd/csrc` — dmhomo_amd has no CPU / eager fallback')
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        got = int(h.dmh_version())
        if got != ABI_VERSION:
            raise DmhError(f'{LIB_PATH} reports ABI version {got}, this binding was written against {ABI_VERSION} '
                           f'(include/dmhomo_hip.h): rebuild the library with `make -C dmhomo_amd/csrc`')
        _lib = h
    return _lib


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name, *args):
    """invoke an int-returning entry point on torch's current stream; raise on a DMH_E* code."""
    h = lib()
    rc = getattr(h, name)(*args, stream())
    if rc != 0:
        raise DmhError(f'{name} failed ({rc}): {h.dmh_last_error().decode()}')


def ptr(t, dtype=torch.float32):
    """device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise DmhError('dmhomo_amd kernels need tensors on the GPU (got a CPU tensor); there is no CPU path')
    if dtype is not None and t.dtype != dtype:
        raise DmhError(f'expected dtype {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise DmhError('expected a contiguous tensor')
    return C.c_void_p(t.data_ptr())
