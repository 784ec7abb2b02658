"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Never imported by libde265_amd.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from libde265_amd import _abi  # noqa: E402  (POD struct layouts of the boundary)

ORD_PU, ORD_PCM, ORD_TU = 1 << 28, 2 << 28, 3 << 28


class OracleImage(C.Structure):
    _fields_ = [("plane", C.c_void_p * 3), ("stride", C.c_int32 * 3)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("hevc_oracle.c", "oracle_px.inc", "hevc_oracle.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "de265_hip.h"))
    if force or not os.path.exists(so) or any(
            os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["gcc", "-O2", "-Wall", "-fPIC", "-shared", "-o", so,
                               os.path.join(_HERE, "hevc_oracle.c")])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.oracle_reconstruct.restype = C.c_int
        L.oracle_reconstruct.argtypes = [C.POINTER(_abi.PictureDesc), C.POINTER(C.c_uint32), C.c_int,
                                         C.POINTER(OracleImage), C.POINTER(OracleImage),
                                         C.POINTER(OracleImage), C.c_int]
        for name in ("oracle_stage_mc",):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [C.POINTER(_abi.PictureDesc), C.POINTER(OracleImage),
                                         C.POINTER(OracleImage)]
        for name in ("oracle_stage_pcm", "oracle_stage_tus", "oracle_stage_deblock"):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [C.POINTER(_abi.PictureDesc), C.POINTER(OracleImage)]
        L.oracle_stage_sao.restype = C.c_int
        L.oracle_stage_sao.argtypes = [C.POINTER(_abi.PictureDesc), C.POINTER(OracleImage),
                                       C.POINTER(OracleImage)]
        L.oracle_dequant.restype = None
        L.oracle_dequant.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_transform_add.restype = None
        L.oracle_transform_add.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p]
        for name in ("oracle_transform_skip_add", "oracle_transform_bypass_add"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p]
        for name in ("oracle_put_qpel", "oracle_put_epel"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t,
                                         C.c_int, C.c_int, C.c_int, C.c_int]
        L.oracle_put_pred.restype = None
        L.oracle_put_pred.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_void_p,
                                      C.c_ssize_t] + [C.c_int] * 7
        L.oracle_intra_predict.restype = None
        L.oracle_intra_predict.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p]
        L.oracle_dct_coeff.restype = C.c_int
        L.oracle_dct_coeff.argtypes = [C.c_int, C.c_int]
        L.oracle_table.restype = C.c_int
        L.oracle_table.argtypes = [C.c_char_p, C.c_int]
        L.oracle_derive_edge_flags.restype = C.c_int
        L.oracle_derive_edge_flags.argtypes = [C.POINTER(_abi.PicParams), C.POINTER(_abi.SliceParams),
                                               C.c_int, C.POINTER(_abi.CtbInfo), C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_derive_bs.restype = None
        L.oracle_derive_bs.argtypes = [C.POINTER(_abi.PictureDesc), C.c_int, C.c_void_p]
        _lib = L
    return _lib


def make_image(planes):
    """planes: list of 3 C-contiguous 2-D numpy arrays (uint8 or uint16)."""
    img = OracleImage()
    for i, p in enumerate(planes):
        assert p.flags["C_CONTIGUOUS"] and p.ndim == 2
        img.plane[i] = p.ctypes.data
        img.stride[i] = p.shape[1]
    return img


def alloc_planes(width, height, bit_depth, fill=None, chroma_format=1):
    dt = np.uint16 if bit_depth > 8 else np.uint8
    cw, ch = width // (1 if chroma_format == 3 else 2), height // (2 if chroma_format == 1 else 1)
    if chroma_format == 0:
        cw, ch = 0, 0                                  # monochrome: empty chroma planes
    shapes = [(height, width), (ch, cw), (ch, cw)]
    if fill is None:
        return [np.zeros(s, dt) for s in shapes]
    return [np.full(s, fill, dt) for s in shapes]


def reconstruct(desc, order, dpb_planes, out_planes, last_stage=_abi.STAGE_FINAL):
    """desc: _abi.PictureDesc (pointer or struct); dpb_planes: {slot: [y,cb,cr]};
    out_planes: [y,cb,cr] modified in place."""
    L = lib()
    dpb = (OracleImage * _abi.MAX_DPB_SLOTS)()
    for slot, pl in (dpb_planes or {}).items():
        dpb[slot] = make_image(pl)
    img = make_image(out_planes)
    scratch_planes = [np.empty_like(p) for p in out_planes]
    scratch = make_image(scratch_planes)
    if order is not None:
        order = np.ascontiguousarray(order, dtype=np.uint32)
        optr, n = order.ctypes.data_as(C.POINTER(C.c_uint32)), len(order)
    else:
        optr, n = None, 0
    dptr = desc if isinstance(desc, C.POINTER(_abi.PictureDesc)) else C.pointer(desc)
    rc = L.oracle_reconstruct(dptr, optr, n, dpb, C.byref(img), C.byref(scratch), last_stage)
    if rc:
        raise RuntimeError("oracle_reconstruct failed: %d" % rc)
    return out_planes
