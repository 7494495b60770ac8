"""ctypes binding of libposehip.so (the C ABI declared in include/posehip.h).

The HIP library is the product: if it cannot be loaded this module raises -- there is no
CPU or PyTorch fallback anywhere in ``sleap_nn_amd``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libposehip.so")

PH_OK = 0
PH_E_INVALID, PH_E_HIP, PH_E_CAPACITY, PH_E_INFEASIBLE, PH_E_WORKSPACE = -1, -2, -3, -4, -5

OP_INPUT_CONV, OP_CONV, OP_POOL, OP_UPSAMPLE, OP_CONVT, OP_HEAD, OP_STEM = 1, 2, 3, 4, 5, 6, 7
OP_PATCH_STEM, OP_DWCONV, OP_LAYERNORM, OP_LINEAR, OP_PATCH_CONV, OP_GELU, OP_SCALE_ADD, OP_GLOBAL_MAXPOOL = 8, 9, 10, 11, 12, 13, 14, 15
FLAG_RELU, FLAG_SIGMOID, FLAG_GELU, FLAG_SCALE_RESIDUAL, FLAG_SOFTMAX, FLAG_SILU = 1, 2, 4, 8, 16, 32
# PH_KV_*: kernel family an op of the last forward ran (ph_model_last_kernels) and the share of its direct-convolution
# FLOPs that family puts through the matrix cores
KV_NONE, KV_DIRECT, KV_WINO1D, KV_WINO2D, KV_W16, KV_C16, KV_ROWGEMM, KV_WINO4, KV_F16, KV_STEM, KV_WINO2D_KS, KV_FUSED, KV_SMALLMAP, KV_F16_ROWS, KV_F16_BLOCK, KV_MLP = range(16)
KV_NAMES = {KV_NONE: "none", KV_DIRECT: "conv3x3_mfma_dma_persist_kernel (direct)", KV_WINO1D: "conv3x3_wino_persist_kernel (Winograd F(2,3) along x)",
            KV_WINO2D: "conv3x3_wino2d_kernel<64> (Winograd F(2x2,3x3))", KV_W16: "conv3x3_w16_kernel (wave-private Winograd F(2x2,3x3), 16x16x4 MFMA)",
            KV_C16: "conv3x3_c16_kernel (direct)", KV_ROWGEMM: "gemm_mfma_dma_kernel (row GEMM, direct)", KV_WINO4: "conv3x3_wino4_kernel (Winograd F(4x4,3x3))",
            KV_F16: "conv3x3_f16_persist_kernel (direct, fp16 matrix pipe)", KV_STEM: "stem_fused_kernel",
            KV_WINO2D_KS: "conv3x3_wino2d_kernel<64, split K> + splitk_reduce_kernel", KV_FUSED: "fused into its producer's epilogue",
            KV_SMALLMAP: "conv3x3_sm_kernel (Winograd F(2x2,3x3), small maps)",
            KV_F16_ROWS: "conv3x3_f16_rows_kernel (direct, plain fp16 on v_mfma_f32_16x16x32_f16: row tiles, loader waves, folded bilinear x2)",
            KV_F16_BLOCK: "block2_c32_f16_kernel (two convs of a 32-channel encoder block in one launch, plain fp16)",
            KV_MLP: "cnblock_mlp_kernel (CNBlock MLP: Linear + GELU + Linear + layer scale + residual in one launch, chained MFMA products)"}
KV_MFMA_SHARE = {KV_NONE: 0.0, KV_DIRECT: 1.0, KV_WINO1D: 2.0 / 3.0, KV_WINO2D: 4.0 / 9.0, KV_W16: 4.0 / 9.0, KV_C16: 1.0, KV_ROWGEMM: 1.0, KV_WINO4: 0.25, KV_F16: 1.0, KV_STEM: 4.0 / 9.0, KV_WINO2D_KS: 4.0 / 9.0, KV_FUSED: 0.0, KV_SMALLMAP: 4.0 / 9.0, KV_F16_ROWS: 1.0, KV_F16_BLOCK: 1.0, KV_MLP: 1.0}


class OpDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kind", "src0", "src1", "dst", "cin0", "cin1", "cout", "ksize", "flags", "weight", "bias", "out_index", "dst2", "weight2", "bias2", "cmid")]


class PosehipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libposehip error {code}: {msg}")
        self.code = code


_lib = None

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes); mirrors include/posehip.h one to one
SIGNATURES = {
    "ph_last_error": (C.c_char_p, []),
    "ph_version": (C.c_int, []),
    "ph_local_peaks_scratch_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "ph_op_desc_size": (_i32, []),
    "ph_model_set_option": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "ph_model_get_option": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_double)]),
    "ph_model_create": (_vp, [C.POINTER(OpDesc), _i32, C.POINTER(_vp), C.POINTER(_i64), _i32, _i32, _i32]),
    "ph_model_destroy": (None, [_vp]),
    "ph_model_workspace_bytes": (_i64, [_vp, _i32, _i32, _i32]),
    "ph_model_output_shape": (C.c_int, [_vp, _i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "ph_model_forward": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i64, C.POINTER(_vp), _vp]),
    "ph_model_num_params": (_i64, [_vp]),
    "ph_model_set_params": (C.c_int, [_vp, _vp, _vp]),
    "ph_model_grad_bucket_split": (_i64, [_vp]),
    "ph_model_set_bucket_event": (C.c_int, [_vp, _vp]),
    "ph_comm_available": (C.c_int, []),
    "ph_comm_unique_id": (C.c_int, [_vp]),
    "ph_comm_create": (_vp, [_vp, _i32, _i32]),
    "ph_comm_destroy": (None, [_vp]),
    "ph_comm_world": (_i32, [_vp]),
    "ph_allreduce": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "ph_model_set_comm": (C.c_int, [_vp, _vp, _vp]),
    "ph_model_backward_workspace_bytes": (_i64, [_vp, _i32, _i32, _i32]),
    "ph_model_backward": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i64, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_f32), _vp, _i32, _f32,
                                    _i32, _i32, _f32, _vp, _vp, _vp]),
    "ph_debug_split_plan": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_int64)]),
    "ph_debug_gemm_bench": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_float)]),
    "ph_debug_row_wgrad_bench": (C.c_int, [_i32, _i32, _i32, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "ph_render_confmaps": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    "ph_render_pafs": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    "ph_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _i32, _f32, _vp]),
    "ph_adamw_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _vp]),
    "ph_model_set_profiling": (C.c_int, [_vp, _i32]),
    "ph_model_profile_read": (C.c_int, [_vp, C.POINTER(C.c_double), _i32, C.POINTER(_i32)]),
    "ph_model_last_kernels": (C.c_int, [_vp, C.POINTER(_i32), _i32]),
    "ph_model_set_clock_probe": (C.c_int, [_vp, _vp]),
    "ph_model_read_slot": (C.c_int, [_vp, _i32, _vp, _i64, _vp]),
    "ph_local_peaks": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _vp, _i64, _vp]),
    "ph_global_peaks": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _vp, _vp, _vp]),
    "ph_paf_score": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _vp]),
    "ph_crop_bboxes": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "ph_resize_bilinear_aa": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp]),
    "ph_sample_class_maps": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    "ph_group_class_peaks": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ph_lsap": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "ph_toposort_edges": (C.c_int, [_vp, _i32, _vp]),
    "ph_centroid_select": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ph_topdown_scatter": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "ph_group_packed": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _f32, C.c_double, _i32, _i32, _i32, _f32, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ph_group_batch": (C.c_int, [_i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, C.c_double, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
}


def lib():
    """Load (once) and return the bound library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m sleap_nn_amd.build` "
            "(hipcc --offload-arch=gfx950). sleap_nn_amd has no CPU fallback."
        )
    # libposehip must share ONE HIP runtime with torch (torch bundles its own libamdhip64 with the
    # same SONAME as /opt/rocm's): load torch's copy first, globally, so the loader binds
    # libposehip's libamdhip64.so.7 dependency to it whatever the import order was.
    import torch

    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        C.CDLL(bundled, mode=C.RTLD_GLOBAL)
    l = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(l, name)
        fn.restype = res
        fn.argtypes = args
    if l.ph_op_desc_size() != C.sizeof(OpDesc):
        raise ImportError(f"{LIB_PATH}: struct ph_op_desc is {l.ph_op_desc_size()} bytes, this binding's OpDesc {C.sizeof(OpDesc)}: rebuild the library")
    _lib = l
    return l


def check(rc: int) -> int:
    if rc is not None and rc < 0:
        raise PosehipError(int(rc), lib().ph_last_error().decode("utf-8", "replace"))
    return rc


def current_stream_ptr():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(t, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); sleap_nn_amd runs on MI355X only")
    return t
