"""ctypes binding of the synthetic command-buffer generator (tools/libsynth.so).

Test/bench infrastructure: produces de265hip_picture_desc structures that play
the role of libde265's host parser output.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from libde265_amd import _abi  # noqa: E402


class SynthConfig(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("bit_depth", C.c_int32),
        ("log2_ctb_size", C.c_int32), ("log2_min_tb_size", C.c_int32), ("log2_max_tb_size", C.c_int32),
        ("seed", C.c_uint64),
        ("slice_type", C.c_int32), ("intra_pct", C.c_int32),
        ("n_ref_slots", C.c_int32), ("ref_slots", C.c_int8 * 16),
        ("bi_pct", C.c_int32), ("mv_sigma_qpel", C.c_int32), ("weighted_pred", C.c_int32),
        ("n_slices", C.c_int32), ("tile_cols", C.c_int32), ("tile_rows", C.c_int32),
        ("slice_per_tile", C.c_int32),
        ("cbf_pct", C.c_int32), ("tskip_pct", C.c_int32), ("bypass_pct", C.c_int32),
        ("pcm_pct", C.c_int32), ("pcm_loop_filter_disable", C.c_int32),
        ("scaling_list", C.c_int32), ("constrained_intra_pred", C.c_int32),
        ("strong_intra_smoothing", C.c_int32), ("deblocking", C.c_int32), ("sao", C.c_int32),
        ("lf_across_slices_pct", C.c_int32), ("lf_across_tiles", C.c_int32),
        ("big_coeff_pct", C.c_int32), ("qp_min", C.c_int32), ("qp_max", C.c_int32),
        ("amp", C.c_int32), ("split_bias", C.c_int32),
        ("chroma_format", C.c_int32), ("cross_component_pct", C.c_int32), ("implicit_rdpcm", C.c_int32),
        ("explicit_rdpcm_pct", C.c_int32), ("rotation", C.c_int32), ("intra_smoothing_disabled", C.c_int32),
        ("log2_max_tskip_size", C.c_int32), ("high_precision_offsets", C.c_int32), ("monochrome", C.c_int32),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libsynth.so")
    srcs = [os.path.join(_HERE, f) for f in ("synth.c", "synth.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "de265_hip.h"))
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["gcc", "-O2", "-Wall", "-fPIC", "-shared", "-o", so,
                               os.path.join(_HERE, "synth.c"), "-lm"])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.synth_default_config.restype = None
        L.synth_default_config.argtypes = [C.POINTER(SynthConfig), C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_uint64]
        L.synth_generate.restype = C.c_void_p
        L.synth_generate.argtypes = [C.POINTER(SynthConfig)]
        L.synth_free.restype = None
        L.synth_free.argtypes = [C.c_void_p]
        L.synth_desc.restype = C.POINTER(_abi.PictureDesc)
        L.synth_desc.argtypes = [C.c_void_p]
        L.synth_order.restype = C.POINTER(C.c_uint32)
        L.synth_order.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        for n in ("synth_cb_log2_size", "synth_cb_part_mode", "synth_tu_split", "synth_blk_flags_noedge"):
            getattr(L, n).restype = C.POINTER(C.c_uint8)
            getattr(L, n).argtypes = [C.c_void_p]
        L.synth_fill_plane.restype = None
        L.synth_fill_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64]
        _lib = L
    return _lib


def default_config(width, height, bit_depth=8, slice_type=0, seed=1, **over):
    cfg = SynthConfig()
    lib().synth_default_config(C.byref(cfg), width, height, bit_depth, slice_type, seed)
    for k, v in over.items():
        if k == "ref_slots":
            for i, s in enumerate(v):
                cfg.ref_slots[i] = s
            cfg.n_ref_slots = len(v)
        else:
            if not hasattr(cfg, k):
                raise KeyError(k)
            setattr(cfg, k, v)
    return cfg


class SynthPicture:
    """Owns one generated picture; .desc is a POINTER(PictureDesc) valid while alive."""

    def __init__(self, cfg):
        self.cfg = cfg
        self._h = lib().synth_generate(C.byref(cfg))
        if not self._h:
            raise MemoryError("synth_generate failed")
        self.desc = lib().synth_desc(self._h)
        n = C.c_int32()
        p = lib().synth_order(self._h, C.byref(n))
        self.order = np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint32)

    @property
    def d(self):
        return self.desc.contents

    def _arr(self, fn, n):
        return np.ctypeslib.as_array(getattr(lib(), fn)(self._h), shape=(n,)).copy()

    def structure(self):
        """(cb_log2_size, cb_part_mode, tu_split, blk_flags_noedge) numpy copies."""
        P = self.d.params
        cbs = ((P.width + 7) // 8) * ((P.height + 7) // 8)
        ctb = 1 << P.log2_ctb_size
        cw, ch = (P.width + ctb - 1) // ctb, (P.height + ctb - 1) // ctb
        sh = P.log2_ctb_size - P.log2_min_tb_size
        tbs = (cw << sh) * (ch << sh)
        w4, h4 = (P.width + 3) // 4, (P.height + 3) // 4
        return (self._arr("synth_cb_log2_size", cbs), self._arr("synth_cb_part_mode", cbs),
                self._arr("synth_tu_split", tbs), self._arr("synth_blk_flags_noedge", w4 * h4))

    def blk_flags(self):
        P = self.d.params
        w4, h4 = (P.width + 3) // 4, (P.height + 3) // 4
        return np.ctypeslib.as_array(self.d.blk_flags, shape=(h4, w4)).copy()

    def close(self):
        if self._h:
            lib().synth_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def chroma_dims(width, height, chroma_format=1):
    """(w, h) of a chroma plane: SubWidthC / SubHeightC of sps.cc:540-552"""
    if chroma_format == 0:
        return (0, 0)                                   # monochrome: no chroma planes
    return (width // (2 if chroma_format in (1, 2) else 1), height // (2 if chroma_format == 1 else 1))


def fill_planes(width, height, bit_depth, seed, chroma_format=1):
    """Seeded reference picture [y, cb, cr] (noise + gradient)."""
    dt = np.uint16 if bit_depth > 8 else np.uint8
    out = []
    cw, ch = chroma_dims(width, height, chroma_format)
    for c, (w, h) in enumerate([(width, height), (cw, ch), (cw, ch)]):
        a = np.zeros((h, w), dt)
        lib().synth_fill_plane(a.ctypes.data, w, w, h, bit_depth, seed * 3 + c)
        out.append(a)
    return out
