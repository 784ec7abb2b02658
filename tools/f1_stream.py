"""Reader of the picture dumps the RECORDING reference decoder writes (oracle/f1_recorder.cc) and of
the compact fixtures made from them (tests/golden/stream_*.npz, tools/make_stream_golden.py).  Test infrastructure.

A RecordedPicture holds numpy copies of every array of a de265hip_picture_desc plus the CU/TU structure arrays and
builds the ctypes struct on demand; `order` is None because a real decoder's hooks fire in decode order already and
the oracle's default order (PUs, PCM, TUs) is equivalent for it (tests/test_oracle_picture.py: phase order == decode
order)... EXCEPT that intra TUs may read inter neighbours, which the phase order also provides.  See to_desc().
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
from libde265_amd import _abi  # noqa: E402

SECTIONS = ["scaling", "slices", "ctbs", "tus", "coeff_val", "coeff_pos", "pus", "pcms", "pcm_samples",
            "blk_flags", "blk_qp_y", "blk_motion", "cb_log2", "cb_part", "tu_split", "edges"]
_STRUCT = {"slices": _abi.SliceParams, "ctbs": _abi.CtbInfo, "tus": _abi.TU, "pus": _abi.PU, "pcms": _abi.PCM,
           "blk_motion": _abi.Motion}
_DTYPE = {"coeff_val": np.int16, "coeff_pos": np.uint16, "pcm_samples": np.uint16, "blk_qp_y": np.int8}


class RecordedPicture:
    def __init__(self, params_bytes, meta, arrays):
        # (fixtures recorded before the range-extension fields were appended to de265hip_pic_params hold the shorter struct:
        #  the new fields are all 0 for Main / Main10 streams)
        self.params_bytes = bytes(params_bytes).ljust(C.sizeof(_abi.PicParams), b"\0")
        self.meta = dict(meta)                 # dst_slot, poc, counts
        self.a = {k: np.ascontiguousarray(v) for k, v in arrays.items()}     # raw uint8 for struct sections
        self._keep = None

    @property
    def params(self):
        return _abi.PicParams.from_buffer_copy(self.params_bytes)

    def _ptr(self, name, ctype):
        arr = self.a[name]
        return C.cast(arr.ctypes.data, C.POINTER(ctype)) if arr.size else None

    def to_desc(self):
        """ctypes de265hip_picture_desc pointing into this object's arrays (keep the object alive)."""
        d = _abi.PictureDesc()
        d.params = self.params
        m = self.meta
        d.scaling_factors = self._ptr("scaling", C.c_uint8) if self.a["scaling"].size else None
        d.n_slices = m["n_slices"]; d.slices = self._ptr("slices", _abi.SliceParams)
        d.n_ctbs = m["n_ctbs"]; d.ctbs = self._ptr("ctbs", _abi.CtbInfo)
        d.n_tus = m["n_tus"]; d.tus = self._ptr("tus", _abi.TU)
        d.n_coeffs = m["n_coeffs"]; d.coeff_val = self._ptr("coeff_val", C.c_int16); d.coeff_pos = self._ptr("coeff_pos", C.c_uint16)
        d.n_pus = m["n_pus"]; d.pus = self._ptr("pus", _abi.PU)
        d.n_pcms = m["n_pcms"]; d.pcms = self._ptr("pcms", _abi.PCM)
        d.n_pcm_samples = m["n_pcm_samples"]; d.pcm_samples = self._ptr("pcm_samples", C.c_uint16)
        # the product takes the edge bits inside blk_flags; here as the REFERENCE's derive_edgeFlags marked them
        self._flags = (self.a["blk_flags"] | self.a["edges"]).astype(np.uint8)
        d.blk_flags = C.cast(self._flags.ctypes.data, C.POINTER(C.c_uint8))
        d.blk_qp_y = self._ptr("blk_qp_y", C.c_int8)
        d.blk_motion = self._ptr("blk_motion", _abi.Motion)
        self._keep = d
        return d

    def structure(self):
        return self.a["cb_log2"], self.a["cb_part"], self.a["tu_split"], self.a["blk_flags"]


def _plane_shapes(P):
    dt = np.uint16 if P.bit_depth_luma > 8 else np.uint8
    cw, ch = P.width // (1 if P.chroma_format_idc == 3 else 2), P.height // (2 if P.chroma_format_idc == 1 else 1)
    return [((P.height, P.width), dt), ((ch, cw), dt), ((ch, cw), dt)]


def load_dump(path):
    """-> RecordedPicture, prefilter planes, final planes (as the reference decoder produced them)."""
    b = open(path, "rb").read()
    assert b[:8] == b"F1DESC02", path
    off = 8
    psz = C.sizeof(_abi.PicParams)
    params_bytes = b[off:off + psz]; off += psz
    names = ["n_slices", "n_ctbs", "n_tus", "n_coeffs", "n_pus", "n_pcms", "n_pcm_samples", "w4", "h4", "n_cb", "n_tb",
             "dst_slot", "poc", "has_scaling"]
    vals = np.frombuffer(b, np.int32, len(names), off); off += 4 * len(names)
    m = dict(zip(names, (int(v) for v in vals)))
    P = _abi.PicParams.from_buffer_copy(params_bytes)

    def take(nbytes):
        nonlocal off
        a = np.frombuffer(b, np.uint8, nbytes, off).copy(); off += nbytes
        return a

    arrays = {}
    arrays["scaling"] = take(_abi.SCALING_BLOB_BYTES if m["has_scaling"] else 0)
    arrays["slices"] = take(m["n_slices"] * C.sizeof(_abi.SliceParams))
    arrays["ctbs"] = take(m["n_ctbs"] * C.sizeof(_abi.CtbInfo))
    arrays["tus"] = take(m["n_tus"] * C.sizeof(_abi.TU))
    arrays["coeff_val"] = take(m["n_coeffs"] * 2).view(np.int16)
    arrays["coeff_pos"] = take(m["n_coeffs"] * 2).view(np.uint16)
    arrays["pus"] = take(m["n_pus"] * C.sizeof(_abi.PU))
    arrays["pcms"] = take(m["n_pcms"] * C.sizeof(_abi.PCM))
    arrays["pcm_samples"] = take(m["n_pcm_samples"] * 2).view(np.uint16)
    nblk = m["w4"] * m["h4"]
    arrays["blk_flags"] = take(nblk)
    arrays["blk_qp_y"] = take(nblk).view(np.int8)
    arrays["blk_motion"] = take(nblk * C.sizeof(_abi.Motion))
    arrays["cb_log2"] = take(m["n_cb"]); arrays["cb_part"] = take(m["n_cb"]); arrays["tu_split"] = take(m["n_tb"])

    def planes():
        out = []
        for shape, dt in _plane_shapes(P):
            n = shape[0] * shape[1] * np.dtype(dt).itemsize
            out.append(take(n).view(dt).reshape(shape))
        return out

    pre = planes()
    arrays["edges"] = take(nblk)
    fin = planes()
    assert off == len(b), (off, len(b))
    return RecordedPicture(params_bytes, m, arrays), pre, fin


def save_fixture(path, pics):
    """pics: list of (RecordedPicture, digests dict) -> one compressed npz (no sample planes, only digests)."""
    z = {}
    for i, (rp, dg) in enumerate(pics):
        z["%d/params" % i] = np.frombuffer(rp.params_bytes, np.uint8)
        z["%d/meta" % i] = np.array([rp.meta[k] for k in sorted(rp.meta)], np.int64)
        z["%d/meta_keys" % i] = np.array(sorted(rp.meta))
        for k in SECTIONS:
            z["%d/%s" % (i, k)] = rp.a[k].view(np.uint8) if rp.a[k].dtype != np.uint8 else rp.a[k]
        z["%d/digests" % i] = np.array([dg["prefilter"], dg["final"]])
    z["n"] = np.array([len(pics)])
    np.savez_compressed(path, **z)


def load_fixture(path):
    """-> list of (RecordedPicture, {'prefilter': md5, 'final': md5})"""
    z = np.load(path, allow_pickle=False)
    out = []
    for i in range(int(z["n"][0])):
        meta = dict(zip((str(k) for k in z["%d/meta_keys" % i]), (int(v) for v in z["%d/meta" % i])))
        arrays = {}
        for k in SECTIONS:
            a = z["%d/%s" % (i, k)]
            arrays[k] = a.view(_DTYPE[k]) if k in _DTYPE else a
        dg = z["%d/digests" % i]
        out.append((RecordedPicture(z["%d/params" % i].tobytes(), meta, arrays), {"prefilter": str(dg[0]), "final": str(dg[1])}))
    return out
