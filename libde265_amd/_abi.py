"""ctypes mirrors of the POD structs declared in include/de265_hip.h.

Field order and types must match the header exactly; tests/test_abi.py checks
sizeof() of every struct against the values the C library reports.
"""
import ctypes as C

MAX_DPB_SLOTS = 20
SCALING_BLOB_BYTES = 6 * 16 + 6 * 64 + 6 * 256 + 2 * 1024

OK = 0
ERROR_OUT_OF_MEMORY = 7
ERROR_PARAMETER_OUT_OF_RANGE = 8
ERROR_INIT_FAILED = 11
ERROR_DECODING = 18
ERROR_NOT_IMPLEMENTED = 502

STAGE_PREFILTER, STAGE_DEBLOCKED, STAGE_FINAL = 0, 1, 2

TU_INTRA, TU_CBF, TU_TSKIP, TU_BYPASS = 1, 2, 4, 8
TU_EXPLICIT_RDPCM, TU_EXPLICIT_RDPCM_VERT = 0x10, 0x20
BLK_INTRA, BLK_NONZERO, BLK_PCM, BLK_BYPASS = 1, 2, 4, 8
BLK_EDGE_TU_V, BLK_EDGE_TU_H, BLK_EDGE_PB_V, BLK_EDGE_PB_H = 0x10, 0x20, 0x40, 0x80

K_NAMES = ["mc", "resid", "intra", "bs", "deblock_v", "deblock_h", "sao", "pcm", "intra_front"]


class PicParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("bit_depth_luma", C.c_int32), ("bit_depth_chroma", C.c_int32),
        ("chroma_format_idc", C.c_int32),
        ("log2_ctb_size", C.c_int32), ("log2_min_cb_size", C.c_int32),
        ("log2_min_tb_size", C.c_int32),
        ("pcm_loop_filter_disable_flag", C.c_int32),
        ("strong_intra_smoothing_enable_flag", C.c_int32),
        ("constrained_intra_pred_flag", C.c_int32),
        ("sample_adaptive_offset_enabled_flag", C.c_int32),
        ("scaling_list_enable_flag", C.c_int32),
        ("weighted_pred_flag", C.c_int32), ("weighted_bipred_flag", C.c_int32),
        ("pic_cb_qp_offset", C.c_int32), ("pic_cr_qp_offset", C.c_int32),
        ("loop_filter_across_tiles_enabled_flag", C.c_int32),
        ("num_tile_columns", C.c_int32), ("num_tile_rows", C.c_int32),
        ("col_bd", C.c_uint16 * 24), ("row_bd", C.c_uint16 * 24),
        ("disable_deblocking", C.c_int32), ("disable_sao", C.c_int32),
        ("implicit_rdpcm_enabled_flag", C.c_int32), ("transform_skip_rotation_enabled_flag", C.c_int32),
        ("intra_smoothing_disabled_flag", C.c_int32), ("cross_component_prediction_enabled_flag", C.c_int32),
        ("extended_precision_processing_flag", C.c_int32), ("high_precision_offsets_enabled_flag", C.c_int32),
    ]


class SliceParams(C.Structure):
    _fields_ = [
        ("slice_type", C.c_int32), ("slice_addr_rs", C.c_int32),
        ("slice_deblocking_filter_disabled_flag", C.c_int32),
        ("slice_beta_offset", C.c_int32), ("slice_tc_offset", C.c_int32),
        ("slice_loop_filter_across_slices_enabled_flag", C.c_int32),
        ("slice_sao_luma_flag", C.c_int32), ("slice_sao_chroma_flag", C.c_int32),
        ("luma_log2_weight_denom", C.c_int32), ("chroma_log2_weight_denom", C.c_int32),
        ("luma_weight", (C.c_int16 * 16) * 2), ("luma_offset", (C.c_int16 * 16) * 2),
        ("chroma_weight", ((C.c_int16 * 2) * 16) * 2),
        ("chroma_offset", ((C.c_int16 * 2) * 16) * 2),
        ("ref_pic_list", (C.c_int8 * 16) * 2),
    ]


class CtbInfo(C.Structure):
    _fields_ = [
        ("slice_addr_rs", C.c_uint16), ("slice_idx", C.c_uint16),
        ("sao_type_idx", C.c_uint8), ("sao_eo_class", C.c_uint8),
        ("sao_band_position", C.c_uint8 * 3),
        ("sao_offset_val", (C.c_int8 * 4) * 3),
        ("pad", C.c_uint8 * 3),
    ]


class TU(C.Structure):
    _fields_ = [
        ("x0", C.c_uint16), ("y0", C.c_uint16),
        ("log2_size", C.c_uint8), ("c_idx", C.c_uint8), ("flags", C.c_uint8),
        ("intra_mode", C.c_uint8), ("qp", C.c_int8), ("res_scale_val", C.c_int8),
        ("n_coeff", C.c_uint16), ("coeff_offset", C.c_uint32),
    ]


class PU(C.Structure):
    _fields_ = [
        ("x", C.c_uint16), ("y", C.c_uint16), ("w", C.c_uint8), ("h", C.c_uint8),
        ("pred_flag", C.c_uint8), ("pad", C.c_uint8), ("slice_idx", C.c_uint16),
        ("ref_idx", C.c_int8 * 2), ("mv", (C.c_int16 * 2) * 2),
    ]


class PCM(C.Structure):
    _fields_ = [
        ("x0", C.c_uint16), ("y0", C.c_uint16), ("log2_cb_size", C.c_uint8),
        ("pad", C.c_uint8 * 3), ("sample_offset", C.c_uint32),
    ]


class Motion(C.Structure):
    _fields_ = [
        ("mv", (C.c_int16 * 2) * 2), ("ref_slot", C.c_int8 * 2), ("pad", C.c_uint8 * 2),
    ]


class PictureDesc(C.Structure):
    _fields_ = [
        ("params", PicParams),
        ("scaling_factors", C.POINTER(C.c_uint8)),
        ("n_slices", C.c_int32), ("slices", C.POINTER(SliceParams)),
        ("n_ctbs", C.c_int32), ("ctbs", C.POINTER(CtbInfo)),
        ("n_tus", C.c_int32), ("tus", C.POINTER(TU)),
        ("n_coeffs", C.c_int32), ("coeff_val", C.POINTER(C.c_int16)),
        ("coeff_pos", C.POINTER(C.c_uint16)),
        ("n_pus", C.c_int32), ("pus", C.POINTER(PU)),
        ("n_pcms", C.c_int32), ("pcms", C.POINTER(PCM)),
        ("n_pcm_samples", C.c_int32), ("pcm_samples", C.POINTER(C.c_uint16)),
        ("blk_flags", C.POINTER(C.c_uint8)),
        ("blk_qp_y", C.POINTER(C.c_int8)),
        ("blk_motion", C.POINTER(Motion)),
    ]


class PictureStats(C.Structure):
    _fields_ = [
        ("n_levels", C.c_int32), ("n_tu_tasks", C.c_int32), ("n_mc_tasks", C.c_int32),
        ("n_runs", C.c_int32), ("n_run_levels", C.c_int32), ("n_in_run_levels", C.c_int32),
        ("device_bytes", C.c_int64),
        ("alg_bytes_mc", C.c_int64), ("alg_bytes_resid", C.c_int64),
        ("alg_bytes_intra", C.c_int64), ("alg_bytes_deblock", C.c_int64),
        ("alg_bytes_sao", C.c_int64),
        ("alg_bytes_intra_front", C.c_int64), ("n_front_runs", C.c_int32), ("pad", C.c_int32),
    ]
