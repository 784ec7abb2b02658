"""ctypes binding of the COMPILED REFERENCE (oracle/_ref/libde265_ref.so = libde265's own decoder
sources + oracle/ref_shim.cc, built by oracle/Makefile in the build container).

TEST INFRASTRUCTURE ONLY.  Same call shapes as pyoracle so that a test can run restatement,
reference and HIP path on one input.  available() is False where neither the built .so nor
/root/reference exists; tests then fall back to the committed reference-generated fixtures in
tests/golden/ (tools/make_ref_golden.py).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from libde265_amd import _abi  # noqa: E402
import pyoracle  # noqa: E402  (OracleImage / make_image only)

REF_ROOT = os.environ.get("DE265_REFERENCE", "/root/reference")
SO = os.environ.get("DE265_REF_SO", os.path.join(_HERE, "_ref", "libde265_ref.so"))


def can_build():
    return os.path.isdir(os.path.join(REF_ROOT, "libde265"))


def build(force=False):
    """make -C oracle ref (only where the reference sources exist); returns the .so path or None."""
    if can_build():
        srcs = [os.path.join(_HERE, f) for f in ("ref_shim.cc", "hevc_oracle.h", "Makefile")]
        srcs.append(os.path.join(_HERE, "..", "include", "de265_hip.h"))
        if force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs):
            subprocess.check_call(["make", "-s", "-j8", "-C", _HERE, "ref", "REF=" + REF_ROOT])
    return SO if os.path.exists(SO) else None


def build_f1():
    """make -C oracle f1 f2: the patched (recording / offloading) decoder (reference + f1_recorder.patch + f1_recorder.cc), the
    reference's encoder CLI and the synthetic bitstream writer (f2_writer.cc), only where the reference sources exist.
    Returns the decoder's path or None."""
    exe = os.path.join(_HERE, "_ref", "f1_dec")
    if can_build():
        subprocess.check_call(["make", "-s", "-j8", "-C", _HERE, "f1", "f2", "REF=" + REF_ROOT])
    return exe if os.path.exists(exe) else None


def available():
    return os.path.exists(SO) or can_build()


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = build()
        if so is None:
            raise RuntimeError("compiled reference not available (no oracle/_ref/libde265_ref.so and no %s)" % REF_ROOT)
        L = C.CDLL(so)
        L.ref_version.restype = C.c_char_p
        u8p = C.c_void_p
        L.ref_reconstruct.restype = C.c_int
        L.ref_reconstruct.argtypes = [C.POINTER(_abi.PictureDesc), C.POINTER(C.c_uint32), C.c_int,
                                      C.POINTER(pyoracle.OracleImage), C.POINTER(pyoracle.OracleImage), C.c_int,
                                      u8p, u8p, u8p, u8p]
        L.ref_derive_edge_flags.restype = C.c_int
        L.ref_derive_edge_flags.argtypes = [C.POINTER(_abi.PictureDesc), u8p, u8p, u8p, u8p]
        L.ref_derive_bs.restype = C.c_int
        L.ref_derive_bs.argtypes = [C.POINTER(_abi.PictureDesc), u8p, u8p, u8p, C.c_int, u8p]
        L.ref_transform_add.restype = None
        L.ref_transform_add.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p]
        for name in ("ref_transform_skip_add", "ref_transform_bypass_add"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p]
        L.ref_transform_residual.restype = None
        L.ref_transform_residual.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        for name in ("ref_put_qpel", "ref_put_epel"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t,
                                         C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_put_pred.restype = None
        L.ref_put_pred.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_void_p,
                                   C.c_ssize_t] + [C.c_int] * 7
        L.ref_vtable_compare.restype = C.c_int
        L.ref_vtable_compare.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_int)]
        L.ref_intra_tu.restype = C.c_int
        L.ref_intra_tu.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_ssize_t,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib = L
    return _lib


def _dptr(desc):
    return desc if isinstance(desc, C.POINTER(_abi.PictureDesc)) else C.pointer(desc)


def _w4h4(desc):
    P = _dptr(desc).contents.params
    return (P.width + 3) // 4, (P.height + 3) // 4


def reconstruct(desc, order, dpb_planes, out_planes, structure, last_stage=_abi.STAGE_FINAL, want_deblk=False):
    """Like pyoracle.reconstruct; structure = SynthPicture.structure() (cb_log2_size, cb_part_mode,
    tu_split, ...).  out_planes are modified in place."""
    L = lib()
    dpb = (pyoracle.OracleImage * _abi.MAX_DPB_SLOTS)()
    for slot, pl in (dpb_planes or {}).items():
        dpb[slot] = pyoracle.make_image(pl)
    img = pyoracle.make_image(out_planes)
    if order is not None:
        order = np.ascontiguousarray(order, dtype=np.uint32)
        optr, n = order.ctypes.data_as(C.POINTER(C.c_uint32)), len(order)
    else:
        optr, n = None, 0
    cb_log2, cb_part, tu_split = [np.ascontiguousarray(a, np.uint8) for a in structure[:3]]
    deblk = None
    if want_deblk:
        w4, h4 = _w4h4(desc)
        deblk = np.zeros((h4, w4), np.uint8)
    rc = L.ref_reconstruct(_dptr(desc), optr, n, dpb, C.byref(img), last_stage,
                           cb_log2.ctypes.data, cb_part.ctypes.data, tu_split.ctypes.data,
                           deblk.ctypes.data if deblk is not None else None)
    if rc:
        raise RuntimeError("ref_reconstruct failed: %d" % rc)
    return (out_planes, deblk) if want_deblk else out_planes


def derive_edge_flags(desc, structure):
    """blk_flags_noedge | edge bits as the reference's derive_edgeFlags marks them."""
    w4, h4 = _w4h4(desc)
    cb_log2, cb_part, tu_split, noedge = [np.ascontiguousarray(a, np.uint8) for a in structure[:4]]
    out = noedge.copy().reshape(h4, w4)
    rc = lib().ref_derive_edge_flags(_dptr(desc), cb_log2.ctypes.data, cb_part.ctypes.data,
                                     tu_split.ctypes.data, out.ctypes.data)
    if rc:
        raise RuntimeError("ref_derive_edge_flags failed: %d" % rc)
    return out


def derive_bs(desc, structure, vertical):
    w4, h4 = _w4h4(desc)
    cb_log2, cb_part, tu_split = [np.ascontiguousarray(a, np.uint8) for a in structure[:3]]
    out = np.zeros((h4, w4), np.uint8)
    rc = lib().ref_derive_bs(_dptr(desc), cb_log2.ctypes.data, cb_part.ctypes.data, tu_split.ctypes.data,
                             1 if vertical else 0, out.ctypes.data)
    if rc:
        raise RuntimeError("ref_derive_bs failed: %d" % rc)
    return out
