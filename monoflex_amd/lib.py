"""ctypes binding of libmonoflex_hip.so (the C ABI declared in include/monoflex_hip.h).

The product path has no CPU fallback: if the library is missing, or an entry point returns an
error code, a RuntimeError is raised with the library's own message.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MFX_LIB_PATH") or os.path.join(HERE, "csrc", "libmonoflex_hip.so")      # (MFX_LIB_PATH: another BUILD of the same library, for A/B timing of kernel revisions: tools/*_bench.py)

MFX_F32, MFX_BF16, MFX_F16, MFX_F16X2 = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_DCN_OFFMASK = 0, 1, 2, 3
MAX_SEG = 9

c_int, c_void_p, c_float, c_size_t = ctypes.c_int32, ctypes.c_void_p, ctypes.c_float, ctypes.c_size_t


class ConvDesc(ctypes.Structure):
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("w_frag", c_void_p), ("scale", c_void_p), ("shift", c_void_p), ("res", c_void_p),
                ("y", c_void_p), ("rowmap", c_void_p),
                ("B", c_int), ("H", c_int), ("W", c_int), ("x_pixstride", c_int), ("Ck", c_int),
                ("kh", c_int), ("kw", c_int), ("stride", c_int), ("pad_h", c_int), ("pad_w", c_int), ("dil_w", c_int),
                ("Ho", c_int), ("Wo", c_int), ("M", c_int), ("Cout", c_int), ("Cout_pad", c_int), ("K_pad", c_int),
                ("ldy", c_int), ("ldres", c_int), ("act", c_int), ("dtype", c_int), ("out_dtype", c_int),
                ("workspace", c_void_p), ("workspace_bytes", ctypes.c_int64),
                ("stats", c_void_p), ("stats_ncopy", c_int), ("stats_done", ctypes.POINTER(c_int)), ("w_frag_pair", c_void_p)]


class CatDesc(ctypes.Structure):
    _fields_ = [("src", c_void_p * MAX_SEG), ("stride", c_int * MAX_SEG), ("off", c_int * MAX_SEG),
                ("nseg", c_int), ("Cseg", c_int),
                ("w", c_void_p), ("scale", c_void_p), ("shift", c_void_p), ("res", c_void_p), ("y", c_void_p),
                ("M", c_int), ("Cout", c_int), ("Cout_pad", c_int), ("K_pad", c_int), ("ldy", c_int),
                ("ldres", c_int), ("act", c_int), ("dtype", c_int)]


class DcnDesc(ctypes.Structure):
    _fields_ = [("x", c_void_p), ("offmask", c_void_p), ("w", c_void_p), ("scale", c_void_p), ("shift", c_void_p),
                ("y", c_void_p), ("w_frag", c_void_p),
                ("B", c_int), ("H", c_int), ("W", c_int), ("C", c_int),
                ("kh", c_int), ("kw", c_int), ("stride", c_int), ("pad", c_int), ("dil", c_int),
                ("Ho", c_int), ("Wo", c_int), ("Cout", c_int), ("Cout_pad", c_int), ("K_pad", c_int),
                ("ldy", c_int), ("act", c_int), ("dtype", c_int), ("w_frag_f16", c_void_p),
                ("workspace", c_void_p), ("workspace_bytes", ctypes.c_int64),
                ("off_w_frag_f16", c_void_p), ("off_shift", c_void_p), ("offmask_out", c_void_p),
                ("nonsquare", c_int), ("stride_w", c_int), ("pad_w", c_int), ("dil_w", c_int), ("w_pair_f16", c_void_p)]


class HeadsDesc(ctypes.Structure):
    _fields_ = [("x", c_void_p), ("w1", c_void_p), ("scale1", c_void_p), ("shift1", c_void_p), ("w2", c_void_p),
                ("bias2", c_void_p), ("out", c_void_p), ("planar", c_void_p),
                ("B", c_int), ("H", c_int), ("W", c_int), ("nbranch", c_int), ("K_pad", c_int), ("ld_out", c_int),
                ("dtype", c_int), ("planar_c", c_int), ("ch_off", c_int * 16), ("c_out", c_int * 16), ("w2_scale", ctypes.c_float * 16),
                ("w1_32", c_void_p), ("w2_32", c_void_p)]


class KittiDesc(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "records", "n_obj", "P", "img_wh", "flip", "hm", "cls_ids", "target_centers", "keypoints", "keypoints_depth_mask",
        "dimensions", "locations", "reg_mask", "reg_weight", "offset_3D", "bboxes", "gt_bboxes", "rotys", "trunc_mask", "alphas",
        "orientations", "occlusions", "truncations", "pad_size", "edge_indices", "edge_len", "P_out", "heat_radius", "status")] + \
        [(n, c_int) for n in ("B", "max_objs", "in_w", "in_h", "down", "num_classes")] + \
        [(n, ctypes.c_double) for n in ("filter_trunc", "filter_size", "edge_ratio")]


class KittiEvalDesc(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("gt", "dt", "gt_off", "dt_off", "pair_off", "classes", "min_overlaps", "overlaps", "tp_scores",
                                        "num_valid_gt", "thresholds", "num_thresholds", "pr")] + \
        [(n, c_int) for n in ("B", "n_gt", "n_dt", "num_classes", "num_k", "compute_aos")] + [("n_pairs", ctypes.c_int64)]


class AdamWDesc(ctypes.Structure):                              # mfx_adamw_desc
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("step", c_void_p), ("numel", ctypes.c_longlong),
                ("group", c_int), ("pad_", c_int)]


class AdamWGroup(ctypes.Structure):                             # mfx_adamw_group
    _fields_ = [("lr", c_void_p), ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("weight_decay", c_float)]


class PackDesc(ctypes.Structure):                               # mfx_pack_desc
    _fields_ = [("w", c_void_p), ("packed", c_void_p), ("frag", c_void_p)] + \
        [(n, c_int) for n in ("Cout", "Cin", "kh", "kw", "mode", "rows_pad", "K_pad", "ck")]


HEAD_MAX_BRANCH = 8


class HeadSparseDesc(ctypes.Structure):                         # mfx_head_sparse_desc
    _fields_ = [(n, c_int) for n in ("nbranch", "N", "B", "H", "W", "C", "dtype", "ld_out")] + [("rows", c_void_p)] + \
        [(n, c_void_p * HEAD_MAX_BRANCH) for n in ("y", "mean", "rstd", "gamma", "beta", "w2", "b2")] + \
        [("k", c_int * HEAD_MAX_BRANCH), ("out_off", c_int * HEAD_MAX_BRANCH), ("out", c_void_p), ("dout", c_void_p), ("g", c_void_p)] + \
        [(n, c_void_p * HEAD_MAX_BRANCH) for n in ("sums", "dw2", "db2", "dx")] + [("arena", c_void_p), ("arena_bytes", c_size_t)]


class GramDesc(ctypes.Structure):                               # mfx_gram_desc (csrc/gram_heads.hip)
    _fields_ = [("x", c_void_p), ("rows", c_void_p), ("extra_rows", c_void_p)] + \
        [(n, c_int) for n in ("B", "H", "W", "C", "N", "Ne", "F", "nbranch", "extra_branch", "ld_out", "dtype", "nring", "ring_width", "arena_bytes")] + \
        [("momentum", c_float), ("Mt", c_float)] + \
        [(n, c_void_p * HEAD_MAX_BRANCH) for n in ("wk", "gamma", "beta", "w2", "b2", "run_mean", "run_var", "nbt", "dgamma", "dbeta", "dw2", "db2", "dwt")] + \
        [("eps", c_float * HEAD_MAX_BRANCH), ("k", c_int * HEAD_MAX_BRANCH), ("off", c_int * HEAD_MAX_BRANCH)] + \
        [(n, c_void_p) for n in ("A", "Wkc", "WkT", "R5", "S0", "P", "csA", "G", "m", "Tm", "sums", "stat", "Y", "Ye", "act", "act_e", "out",
                                 "dout", "dact_e", "dYh", "dYl", "dYeh", "dYel", "dsum", "ds", "scal", "Dw16", "dm", "dGs", "Kx", "Gn16", "cscale", "cshift",
                                 "dAf", "dArows", "dx", "ring_inv", "ring_idx", "dwo", "dwe")]


OBJ_ROW, OBJ_TERMS, OBJ_VALUES = 72, 10, 24                    # MFX_OBJ_ROW / MFX_OBJ_TERMS / MFX_OBJ_VALUES


class ObjectLossCfg(ctypes.Structure):                          # mfx_object_loss_cfg
    _fields_ = [("w", c_float * OBJ_TERMS), ("dim_mean", c_float * 9), ("dim_std", c_float * 9), ("dim_weight", c_float * 3),
                ("depth_ref", c_float * 2), ("depth_range", c_float * 2)] + \
        [(n, c_float) for n in ("unc_lo", "unc_hi", "down_ratio", "eps")] + \
        [(n, c_int) for n in ("depth_mode", "has_depth_range", "dim_exp", "dim_use_std", "iou_type", "corner_depth_mode",
                              "separate_trunc", "trunc_log", "modify_invalid")] + [("ch", c_int * 9)]


# every symbol include/monoflex_hip.h declares: name -> (restype, argtypes)
_P, _I, _F, _S = c_void_p, c_int, c_float, c_size_t
SYMBOLS = {
    "mfx_abi_version": (_I, []),
    "mfx_last_error": (ctypes.c_char_p, []),
    "mfx_stem_conv7x7_nchw": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mfx_f1_fused": (_I, [_P] * 11 + [_I, _I, _I, _I, _I, _P]),
    "mfx_set_option": (_I, [ctypes.c_char_p, _I]),
    "mfx_reset_options": (_I, []),
    "mfx_commit_options": (_I, []),
    "mfx_get_counter": (ctypes.c_long, [ctypes.c_char_p]),
    "mfx_f16x2_range_check": (ctypes.c_int, [ctypes.c_int]),
    "mfx_dcn_v2_workspace_bytes": (_S, [_I] * 14),
    "mfx_dcn_v2_workspace_bytes_g": (_S, [_I] * 15),
    "mfx_dcn_v2_forward": (_I, [_P] * 6 + [_I] * 14 + [_P, _S, _P]),
    "mfx_dcn_v2_backward": (_I, [_P] * 11 + [_I] * 14 + [_P, _S, _P]),
    "mfx_conv2d_nhwc": (_I, [ctypes.POINTER(ConvDesc), _P]),
    "mfx_cat_conv1x1_nhwc": (_I, [ctypes.POINTER(CatDesc), _P]),
    "mfx_dcn_nhwc": (_I, [ctypes.POINTER(DcnDesc), _P]),
    "mfx_dcn_sample_nhwc": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mfx_project_nhwc": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mfx_maxpool2x2_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "mfx_upsample_add_nhwc": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mfx_nchw_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mfx_nhwc_to_nchw": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mfx_pack_image_nhwc4": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mfx_heads_fused": (_I, [ctypes.POINTER(HeadsDesc), _P]),
    "mfx_dcn_fuses_offset_conv": (_I, [ctypes.POINTER(DcnDesc)]),
    "mfx_edge_scatter_add": (_I, [_P, _I, _I, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P, _P]),
    "mfx_decode_topk_workspace_bytes": (_S, [_I, _I, _I]),
    "mfx_decode_topk": (_I, [_P, ctypes.c_long, ctypes.c_long, ctypes.c_long, _I, _I, _I, _I, _I, _P, _P, _P, _S, _P]),
    "mfx_conv_wgrad_nhwc": (_I, [_P, _P, _P] + [_I] * 15 + [_P]),
    "mfx_conv_wgrad_nhwc_dil": (_I, [_P, _P, _P] + [_I] * 16 + [_P]),
    "mfx_conv_wgrad_oihw": (_I, [_P, _P, _P] + [_I] * 17 + [_P, _S, _P]),
    "mfx_pack_conv_weight": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _P]),
    "mfx_adamw_chunk_elems": (_I, []),
    "mfx_adamw_multi": (_I, [_P, _P, _I, ctypes.c_longlong, _P, _P, _P]),
    "mfx_colsum": (_I, [_P, _P, ctypes.c_long, _I, _I, _I, _P]),
    "mfx_colsum_add": (_I, [_P, _P, ctypes.c_long, _I, _I, _I, _P]),
    "mfx_bn_stats": (_I, [_P, _P, _P, ctypes.c_long, _I, _I, _P]),
    "mfx_bn_finalize": (_I, [_P, _P, _P, _P, _P, _P, _F, _F, ctypes.c_long, _P, _P, _P, _P, _I, _P]),
    "mfx_bn_act_fwd": (_I, [_P, _P, _P, _P, _P, ctypes.c_long, _I, _I, _I, _P]),
    "mfx_bn_act_bwd": (_I, [_P] * 10 + [ctypes.c_long, _I, _I, _I, _P]),
    "mfx_stem_wgrad_bf16": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _S, _P]),
    "mfx_stem_wgrad_16": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _S, _P]),
    "mfx_pack_chunk_elems": (_I, []),
    "mfx_pack_conv_weights_batched": (_I, [_P, _P, _I, ctypes.c_longlong, _I, _P]),
    "mfx_head_sparse_fwd": (_I, [ctypes.POINTER(HeadSparseDesc), _P]),
    "mfx_head_sparse_bwd": (_I, [ctypes.POINTER(HeadSparseDesc), _P]),
    "mfx_gram_heads": (_I, [ctypes.POINTER(GramDesc), _I, _P]),
    "mfx_bn_train_stats": (_I, [_P] * 6 + [_F, _F, ctypes.c_long, _I, _I, _P, _P, _P, _I, _P]),
    "mfx_bn_scratch_bytes": (_S, []),
    "mfx_bn_ncopy": (_I, [_I]),
    "mfx_bn_train_fwd": (_I, [_P] * 8 + [_F, _F, ctypes.c_long, _I, _I, _I, _P, _P, _P, _I, _P]),
    "mfx_bn_train_bwd": (_I, [_P] * 11 + [ctypes.c_long, _I, _I, _I, _P, _P]),
    "mfx_bn_bwd_reduce": (_I, [_P] * 7 + [ctypes.c_long, _I, _I, _I, _P]),
    "mfx_bn_bwd_apply": (_I, [_P] * 10 + [ctypes.c_long, ctypes.c_long, _I, _I, _I, _P]),
    "mfx_maxpool2x2_bwd_nhwc": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mfx_bn_onepass_stuck": (_I, [_I]),
    "mfx_upsample_bwd_workspace_bytes": (_S, [_I, _I, _I, _I]),
    "mfx_upsample_bwd_nhwc": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _S, _P]),
    "mfx_upsample_bwd_nhwc_oihw": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _S, _P]),
    "mfx_zero_insert2_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mfx_dcn_backward_nhwc_workspace_bytes": (_S, [_I] * 10),
    "mfx_dcn_backward_nhwc": (_I, [_P] * 8 + [_I] * 10 + [_P, _S, _P]),
    "mfx_dcn_backward_nhwc_bf16": (_I, [_P] * 8 + [_I] * 10 + [_P, _S, _P]),
    "mfx_dcn_backward_v2_workspace_bytes": (_S, [_I] * 6),
    "mfx_dcn_backward_v2": (_I, [_P] * 8 + [_I] * 6 + [_P, _S, _P]),
    "mfx_dcn_backward_v2_rt": (_I, [_P] * 6 + [_I] + [_P] * 2 + [_I] * 6 + [_P, _S, _P]),
    "mfx_focal_loss": (_I, [_P, _P, _I, _I, _I, _I, _F, _F, _P, _P, _P]),
    "mfx_object_loss": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, ctypes.POINTER(ObjectLossCfg), _P, _P, _P]),
    "mfx_object_loss_backward": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _I, _I, _P]),
    "mfx_kitti_encode_targets": (_I, [ctypes.POINTER(KittiDesc), _P]),
    "mfx_kitti_preprocess_u8": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, ctypes.POINTER(c_float), ctypes.POINTER(c_float), _P]),
    "mfx_kitti_eval_overlaps": (_I, [ctypes.POINTER(KittiEvalDesc), _P]),
    "mfx_kitti_eval_match_pass1": (_I, [ctypes.POINTER(KittiEvalDesc), _P]),
    "mfx_kitti_eval_thresholds": (_I, [ctypes.POINTER(KittiEvalDesc), _P, _P]),
    "mfx_kitti_eval_match_pass2": (_I, [ctypes.POINTER(KittiEvalDesc), _P]),
    "mfx_decode_boxes": (_I, [_P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _F, _P, _P, _P, _P]),
    "mfx_decode_boxes_mode": (_I, [_P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _F, _I, _P, _P, _P, _P]),
}

# mfx_decode_boxes_mode's depth_mode (include/monoflex_hip.h MFX_DEPTH_*), by the reference's `output_depth` names (detector_infer.py:149-198)
DEPTH_MODES = {"soft": 0, "hard": 1, "mean": 2, "direct": 3, "keypoints_avg": 4, "keypoints_center": 5, "keypoints_02": 6, "keypoints_13": 7}

_lib = None


def load():
    """Load the shared library (building it is __graft_entry__.build()'s / build.py's job)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libmonoflex_hip.so is missing (%s): run `python -m monoflex_amd.build`; "
                           "the MonoFlex HIP path has no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)            # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.mfx_abi_version() != 3:
        raise RuntimeError("libmonoflex_hip.so ABI version mismatch")
    _lib = lib
    # MFX_OPTIONS="name=value,name=value": library tuning switches for this process (mfx_set_option; A/B sweeps of an unmodified bench.py / training script)
    for kv in filter(None, os.environ.get("MFX_OPTIONS", "").split(",")):
        k, _, v = kv.partition("=")
        check(lib.mfx_set_option(k.strip().encode(), int(v)), "MFX_OPTIONS: %s" % kv)
    check(lib.mfx_commit_options(), "mfx_commit_options")      # mfx_reset_options() now restores the values AFTER the environment's switches
    return lib


def f16x2_range_ok(reset=True):
    """Split-precision mode (MODEL.COMPUTE_DTYPE fp16x2): True if every activation turned into an fp16 (hi, lo) operand pair since the last reset was
    finite and inside fp16's range (|x| <= 65504) -- the mode's one numerical precondition.  Synchronises the device."""
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    rc = load().mfx_f16x2_range_check(1 if reset else 0)
    if rc < 0:
        check(rc, "mfx_f16x2_range_check")
    return rc == 0


def bn_onepass_ok(reset=True):
    """False if a one-launch BatchNorm (csrc/train_kernels.hip bn_*_onepass_kernel) gave up waiting at its grid barrier since the last reset: its
    workgroups did not all become resident, which needs another PROCESS computing on the same device (or two such launches in flight at once).
    The outputs of that launch are wrong; run with MFX_OPTIONS=bn_onepass=0 in such a setup.  Synchronises the device."""
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    rc = load().mfx_bn_onepass_stuck(1 if reset else 0)
    if rc < 0:
        check(rc, "mfx_bn_onepass_stuck")
    return rc == 0


def set_deterministic(on=True):
    """Bit-reproducible training steps: the library's reductions run in a fixed order (option "deterministic": single-writer
    partial sums instead of fp32 atomics, fixed-point accumulation of the DCN input gradient) and torch's own scatter-type ops
    (index_add_ behind the edge-fusion gather) take their sorted forms.  Slower (roughly 2-3x on a full-size step); meant for
    tests and debugging -- two runs, eager or replayed from a hipGraph, then agree bit for bit."""
    import torch
    check(load().mfx_set_option(b"deterministic", 1 if on else 0), "set_option(deterministic)")
    torch.use_deterministic_algorithms(bool(on), warn_only=True)
    if on:
        torch.utils.deterministic.fill_uninitialized_memory = False        # (torch.empty need not be NaN-filled: every buffer is written)


def check(rc, what):
    if rc != 0:
        msg = load().mfx_last_error().decode(errors="replace")
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, msg))
