"""ctypes binding of libcut3r_hip.so (the C ABI declared in include/cut3r_hip.h).

The product path has NO fallback: if the shared library is missing or a kernel returns non-zero, we raise.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcut3r_hip.so")

c_void_p, c_int, c_float, c_ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong


class GemmDesc(C.Structure):
    """mirror of `cut3r_gemm_desc` (include/cut3r_hip.h)"""
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("bias", c_void_p), ("res1", c_void_p), ("res2", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int), ("lda", c_int), ("ldb", c_int), ("ldc", c_int), ("ldr1", c_int),
        ("ldr2", c_int),
        ("act", c_int), ("out_f16", c_int), ("res1_f16", c_int), ("res2_f16", c_int),
        ("batch", c_int),
        ("strideA", c_ll), ("strideB", c_ll), ("strideC", c_ll), ("strideBias", c_ll), ("strideR1", c_ll),
        ("strideR2", c_ll),
        ("conv_k", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int), ("conv_stride", c_int), ("Ho", c_int),
        ("Wo", c_int), ("relu_in", c_int),
        ("shuf", c_int), ("shuf_cout", c_int), ("shuf_Hin", c_int), ("shuf_Win", c_int),
        ("tile", c_int), ("stages", c_int),
        ("rope_pos", c_void_p), ("rope_table", c_void_p), ("rope_cols", c_int), ("rope_pmin", c_int), ("rope_npos", c_int),
        ("rope_d", c_int),
        ("ln_stats", c_void_p), ("ln_colsum", c_void_p), ("ln_nslab", c_int), ("ln_eps", c_float),
        ("stats_out", c_void_p), ("out16", c_void_p), ("ld16", c_int),
    ]


class LcTerm(C.Structure):
    """mirror of `cut3r_lc_term` (include/cut3r_hip.h)"""
    _fields_ = [("a", c_void_p), ("c", c_void_p), ("mask", c_void_p), ("ia", c_int), ("ic", c_int), ("w", c_float), ("pad", c_int)]


# name -> argtypes ; every function returns int (0 ok / 1 bad argument / 2 launch failure)
SIGNATURES = {
    "cut3r_abi_version": [],
    "cut3r_rope2d": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_ll, c_ll, c_ll, c_float, c_float, c_void_p],
    "cut3r_rope2d_qk": [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_ll, c_ll, c_ll, c_ll, c_float, c_float,
                        c_void_p],
    "cut3r_rope2d_tab": [c_void_p, c_void_p, c_ll, c_ll, c_void_p, c_void_p, c_ll, c_ll, c_int, c_int, c_void_p, c_int, c_int, c_float,
                         c_float, c_void_p],
    "cut3r_layernorm": [c_void_p, c_int, c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_int, c_void_p, c_int,
                        c_void_p, c_void_p, c_void_p],
    "cut3r_layernorm_dual": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_float, c_int,
                             c_int, c_void_p],
    "cut3r_gemm_f16": [C.POINTER(GemmDesc), c_void_p],
    "cut3r_gemm_tile_for": [C.POINTER(GemmDesc)],
    "cut3r_gemm_f16_pair": [C.POINTER(GemmDesc), C.POINTER(GemmDesc), c_void_p],
    "cut3r_rope2d_table": [c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p],
    "cut3r_gemv_f16w": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                        c_void_p, c_int, c_int, c_void_p],
    "cut3r_attention_variant": [c_int],
    "cut3r_attention_f16": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                            c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_float, c_void_p],
    "cut3r_im2col_patch": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "cut3r_cast_f32_f16": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p],
    "cut3r_colmean": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "cut3r_colmean_batched": [c_void_p, c_int, c_ll, c_int, c_int, c_int, c_void_p, c_ll, c_void_p],
    "cut3r_upsample2x_nhwc": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "cut3r_dpt_final": [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p],
    "cut3r_postprocess_pts": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "cut3r_postprocess_pose": [c_void_p, c_int, c_void_p, c_void_p],
    "cut3r_patch_overlap": [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p],
    "cut3r_patch_overlap_chain": [c_void_p, c_void_p, c_int, c_int, c_int, c_float, C.c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p],
    "cut3r_overlap_fwd": [c_void_p, c_int, C.POINTER(c_float), c_float, c_void_p, c_int, c_float, c_float, c_float, c_float,
                          c_int, c_int, c_int, c_void_p, c_void_p],
    "cut3r_overlap_bwd": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_float, c_float, c_float, c_float, c_int, c_int,
                          c_void_p, c_void_p],
    "cut3r_window_update": [c_void_p, c_void_p, c_int, c_int, c_int, C.POINTER(c_float), c_float, c_int, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_int, c_int, c_void_p, C.POINTER(c_float), c_int, c_int, c_float, c_float, c_float, c_float,
                            c_void_p, c_int, c_void_p, c_void_p],
    "cut3r_logdepth_accum": [c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "cut3r_resize_linear_u8": [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p],
    "cut3r_mfma_probe": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "cut3r_remap_linear_u8": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "cut3r_align_view": [c_void_p, c_void_p, c_int, c_int, C.POINTER(c_float), c_float, c_int, c_void_p, c_void_p,
                         c_void_p, c_void_p],
    "cut3r_logdepth_sum": [c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "cut3r_lie_unary": [c_int, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "cut3r_lie_unary_bwd": [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "cut3r_lie_mul": [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "cut3r_lie_mul_bwd": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "cut3r_lie_act": [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "cut3r_lie_act_bwd": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "cut3r_lie_adj": [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "cut3r_lc_workspace_floats": [c_int, c_int],
    "cut3r_lc_optimize": [c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_int, c_int, c_ll, c_int, c_float,
                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "cut3r_lc_optimize_terms": [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p],
    "cut3r_transform_submaps": [c_void_p, c_void_p, c_int, c_ll, c_void_p],
    "cut3r_corr_index_forward": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "cut3r_corr_index_backward": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "cut3r_ba_workspace_floats": [c_int, c_int, c_int, c_int, c_int, c_int],
    "cut3r_ba_assemble": [c_void_p] * 12 + [c_int] * 7 + [c_void_p] * 5,
    "cut3r_ba_solve": [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p],
    "cut3r_ba_backsub": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "cut3r_ba_proj_trans": [c_void_p] * 10 + [c_int] * 5 + [c_void_p] * 4,
    "cut3r_bi_inter": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "cut3r_schur_mono_prior_workspace_floats": [c_int, c_ll],
    "cut3r_schur_mono_prior": [c_void_p] * 5 + [c_int, c_ll, c_float, c_float] + [c_void_p] * 6,
    "cut3r_jdsa_blocks": [c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "cut3r_depth_filter": [c_void_p] * 5 + [c_int] * 4 + [c_void_p, c_void_p],
    "cut3r_altcorr_forward": [c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p, c_void_p],
    "cut3r_altcorr_backward": [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p, c_void_p, c_void_p],
    "cut3r_ba_step": [c_void_p] * 12 + [c_int] * 6 + [c_float, c_float] + [c_void_p] * 5,
    "cut3r_gs_preprocess": [c_int] + [c_void_p] * 5 + [c_int, c_int] + [c_void_p] * 4 + [c_int, c_int] + [c_float] * 4 + [c_void_p] * 5
                           + [c_ll, c_void_p],
    "cut3r_gs_workspace_bytes": [c_int, c_ll],
    "cut3r_gs_bin": [c_int, c_void_p, c_void_p, c_ll, c_int, c_int] + [c_void_p] * 6 + [c_ll, c_void_p, c_void_p],
    "cut3r_gs_render_forward": [c_void_p] * 3 + [c_int, c_int, c_float, c_float] + [c_void_p] * 11,
    "cut3r_pixel_loss_forward": [c_void_p] * 5 + [c_int, c_int] + [c_float] * 4 + [c_void_p, c_void_p],
    "cut3r_pixel_loss_backward": [c_void_p] * 5 + [c_int, c_int] + [c_float] * 4 + [c_void_p] * 4,
    "cut3r_normal_agree_forward": [c_void_p, c_void_p, c_int, c_int] + [c_float] * 4 + [c_void_p, c_void_p],
    "cut3r_normal_agree_backward": [c_void_p, c_void_p, c_int, c_int] + [c_float] * 5 + [c_void_p, c_void_p, c_void_p],
    "cut3r_gs_densify_stats": [c_int] + [c_void_p] * 7,
    "cut3r_refine_loss_forward": [c_void_p] * 5 + [c_float, c_int, c_int, c_void_p, c_void_p],
    "cut3r_refine_loss_backward": [c_void_p] * 5 + [c_float, c_int, c_int] + [c_void_p] * 4,
    "cut3r_ssim_forward": [c_void_p, c_void_p, c_int, c_int, c_int] + [c_void_p] * 5,
    "cut3r_ssim_backward": [c_void_p] * 5 + [c_int, c_int, c_int] + [c_void_p] * 3,
    "cut3r_gs_activate": [c_int] + [c_void_p] * 8,
    "cut3r_gs_activate_backward": [c_int] + [c_void_p] * 8 + [c_float] + [c_void_p] * 4,
    "cut3r_gs_pose_step": [c_void_p, c_void_p, c_float, c_void_p, c_float, c_float, c_int, c_void_p],
    "cut3r_gs_adam": [c_ll] + [c_void_p] * 5 + [c_float] * 5 + [c_void_p],
    "cut3r_gs_map_coef": [c_void_p, c_float, c_float, c_float, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "cut3r_gs_refine_coef": [c_void_p, c_float, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "cut3r_knn3_chunks": [c_int],
    "cut3r_knn3_mean_dist2": [c_void_p, c_int, c_void_p, c_void_p, c_void_p],
    "cut3r_knn3_grid_workspace_bytes": [c_int],
    "cut3r_knn3_grid_mean_dist2": [c_void_p, c_int, c_void_p, c_void_p, c_ll, c_void_p],
    "cut3r_gs_render_backward": [c_void_p] * 3 + [c_int, c_int, c_int, c_float, c_float] + [c_void_p] * 16,
    "cut3r_gs_preprocess_backward": [c_int] + [c_void_p] * 5 + [c_int, c_int] + [c_void_p] * 3 + [c_int, c_int] + [c_float] * 4 + [c_void_p] * 10,
}
RESTYPES = {"cut3r_ba_workspace_floats": c_ll, "cut3r_gs_workspace_bytes": c_ll, "cut3r_schur_mono_prior_workspace_floats": c_ll,
            "cut3r_knn3_grid_workspace_bytes": c_ll}

_lib = None


class Cut3rHipError(RuntimeError):
    pass


def load():
    """dlopen the in-tree library and bind every declared symbol (raises if the .so or a symbol is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise Cut3rHipError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.argtypes = argtypes
        fn.restype = RESTYPES.get(name, c_int)
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        raise Cut3rHipError(f"{what} failed with code {rc} ({'bad argument' if rc == 1 else 'launch failure'})")
