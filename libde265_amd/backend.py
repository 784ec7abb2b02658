"""Host-side Python mirror of the C ABI in include/de265_hip.h.

This is plumbing only: every call goes straight into libde265_hip.so (hand
written HIP for gfx950).  There is no CPU fallback: if the shared library is
missing or no GPU is present the calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("DE265HIP_SO") or os.path.join(_HERE, "libde265_hip.so")   # (override: A/B of builds)

# every symbol include/de265_hip.h declares (tests/test_abi.py checks the .so exports them all)
EXPORTS = [
    "de265hip_version", "de265hip_device_count",
    "de265hip_decoder_new", "de265hip_decoder_free", "de265hip_decoder_set_lanes",
    "de265hip_dpb_alloc", "de265hip_dpb_alloc_ex", "de265hip_dpb_chroma_format", "de265hip_dpb_upload", "de265hip_dpb_fill", "de265hip_dpb_download", "de265hip_dpb_plane", "de265hip_dpb_info",
    "de265hip_dpb_copy", "de265hip_dpb_download_async", "de265hip_dpb_download_planes_async", "de265hip_dpb_wait_copy_out", "de265hip_dpb_wait", "de265hip_host_alloc", "de265hip_host_free",
    "de265hip_pipeline_new", "de265hip_pipeline_submit", "de265hip_pipeline_submit_desc", "de265hip_pipeline_wait", "de265hip_pipeline_drain", "de265hip_pipeline_free",
    "de265hip_debug_build_host_only", "de265hip_debug_last_build_hash", "de265hip_debug_build_host_only_ex",
    "de265hip_debug_fault_injection", "de265hip_debug_picture_layout", "de265hip_debug_picture_read",
    "de265hip_picture_build", "de265hip_picture_build_host", "de265hip_picture_enqueue", "de265hip_picture_enqueue_batch", "de265hip_picture_ready", "de265hip_picture_run", "de265hip_decoder_sync", "de265hip_picture_free",
    "de265hip_decode_picture", "de265hip_picture_get_stats",
    "de265hip_set_profiling", "de265hip_get_kernel_times", "de265hip_derive_edge_flags", "de265hip_intra_used_units",
    "de265hip_recorder_new", "de265hip_recorder_free", "de265hip_record_tu", "de265hip_record_pu",
    "de265hip_record_pcm", "de265hip_record_slice", "de265hip_record_ctb", "de265hip_record_blk_planes",
    "de265hip_recorder_desc", "de265hip_recorder_submit",
    "de265hip_fn_transform_add", "de265hip_fn_transform_skip_add", "de265hip_fn_transform_bypass_add",
    "de265hip_fn_put_qpel", "de265hip_fn_put_epel", "de265hip_fn_put_pred",
    "init_acceleration_functions_hip",                  # include/de265_hip_vtable.h
]


class De265HipError(RuntimeError):
    def __init__(self, code, where):
        super().__init__("%s failed with de265_error %d" % (where, code))
        self.code = code


_lib = None


# de265hip_prepare_fn: int (*)(void* user, de265hip_recorder** out)
PREPARE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p))


def lib():
    """Loads libde265_hip.so; raises (never falls back) if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError("%s not found: run `python -m libde265_amd.build` (hipcc, gfx950)" % SO_PATH)
    L = C.CDLL(SO_PATH)
    vp, i32, pp = C.c_void_p, C.c_int, C.POINTER
    L.de265hip_version.restype = C.c_char_p
    L.de265hip_device_count.restype = i32
    L.de265hip_decoder_new.argtypes = [pp(vp), i32]
    L.de265hip_decoder_free.argtypes = [vp]
    L.de265hip_decoder_free.restype = None
    L.de265hip_dpb_alloc.argtypes = [vp, i32, i32, i32, i32, i32]
    L.de265hip_dpb_alloc_ex.argtypes = [vp, i32, i32, i32, i32, i32, i32]
    L.de265hip_dpb_chroma_format.argtypes = [vp, i32]
    L.de265hip_dpb_upload.argtypes = [vp, i32, i32, vp, C.c_ssize_t]
    L.de265hip_dpb_fill.argtypes = [vp, i32, i32, i32, i32]
    L.de265hip_dpb_download.argtypes = [vp, i32, i32, vp, C.c_ssize_t]
    L.de265hip_dpb_plane.argtypes = [vp, i32, i32, pp(vp), pp(C.c_ssize_t)]
    L.de265hip_dpb_info.argtypes = [vp, i32, pp(i32), pp(i32), pp(i32), pp(i32)]
    L.de265hip_dpb_copy.argtypes = [vp, i32, vp, i32]
    L.de265hip_dpb_download_async.argtypes = [vp, i32, i32, vp, C.c_ssize_t]
    L.de265hip_dpb_download_planes_async.argtypes = [vp, i32, C.POINTER(C.c_void_p), C.POINTER(C.c_ssize_t), C.POINTER(C.c_uint64)]
    L.de265hip_dpb_wait_copy_out.argtypes = [vp, i32, C.c_uint64]
    L.de265hip_dpb_wait.argtypes = [vp, i32]
    L.de265hip_host_alloc.argtypes = [C.c_size_t]
    L.de265hip_host_alloc.restype = vp
    L.de265hip_host_free.argtypes = [vp]
    L.de265hip_host_free.restype = None
    L.de265hip_pipeline_new.argtypes = [pp(vp), vp, i32]
    L.de265hip_pipeline_submit.argtypes = [vp, i32, PREPARE_FN, vp, pp(vp), pp(C.c_ssize_t), pp(C.c_uint64)]
    L.de265hip_pipeline_submit_desc.argtypes = [vp, i32, C.POINTER(_abi.PictureDesc), pp(vp), pp(C.c_ssize_t), pp(C.c_uint64)]
    L.de265hip_pipeline_wait.argtypes = [vp, C.c_uint64]
    L.de265hip_pipeline_drain.argtypes = [vp]
    L.de265hip_pipeline_free.argtypes = [vp]
    L.de265hip_pipeline_free.restype = None
    L.de265hip_debug_build_host_only.argtypes = [pp(_abi.PictureDesc), i32]
    L.de265hip_debug_build_host_only_ex.argtypes = [pp(_abi.PictureDesc), i32, i32, pp(vp)]
    L.de265hip_debug_fault_injection.argtypes = [vp, i32, C.c_uint32]
    L.de265hip_debug_picture_layout.argtypes = [vp, pp(C.c_int64)]
    L.de265hip_debug_picture_read.argtypes = [vp, C.c_int64, C.c_int64, vp]
    L.de265hip_debug_last_build_hash.restype = C.c_uint64
    L.de265hip_debug_last_build_hash.argtypes = []
    L.de265hip_picture_build.argtypes = [vp, i32, pp(_abi.PictureDesc), pp(vp)]
    L.de265hip_picture_build_host.argtypes = [vp, i32, pp(_abi.PictureDesc), pp(vp)]
    L.de265hip_picture_enqueue.argtypes = [vp]
    L.de265hip_picture_ready.argtypes = [vp]
    L.de265hip_picture_enqueue_batch.argtypes = [pp(vp), i32]
    L.de265hip_picture_run.argtypes = [vp, vp, i32]
    L.de265hip_decoder_sync.argtypes = [vp]
    L.de265hip_decoder_set_lanes.argtypes = [vp, i32]
    L.de265hip_picture_free.argtypes = [vp]
    L.de265hip_picture_free.restype = None
    L.de265hip_decode_picture.argtypes = [vp, i32, pp(_abi.PictureDesc)]
    L.de265hip_picture_get_stats.argtypes = [vp, pp(_abi.PictureStats)]
    L.de265hip_set_profiling.argtypes = [vp, i32]
    L.de265hip_get_kernel_times.argtypes = [vp, pp(C.c_double), pp(C.c_int64), i32]
    L.de265hip_intra_used_units.argtypes = [i32, i32, i32, pp(C.c_uint64)]
    L.de265hip_derive_edge_flags.argtypes = [pp(_abi.PicParams), pp(_abi.SliceParams), i32, pp(_abi.CtbInfo),
                                             vp, vp, vp, vp]
    L.de265hip_recorder_new.argtypes = [pp(vp), pp(_abi.PicParams), vp]
    L.de265hip_recorder_free.argtypes = [vp]
    L.de265hip_recorder_free.restype = None
    L.de265hip_record_tu.argtypes = [vp, pp(_abi.TU), vp, vp]
    L.de265hip_record_pu.argtypes = [vp, pp(_abi.PU)]
    L.de265hip_record_pcm.argtypes = [vp, i32, i32, i32, vp]
    L.de265hip_record_slice.argtypes = [vp, pp(_abi.SliceParams)]
    L.de265hip_record_ctb.argtypes = [vp, i32, pp(_abi.CtbInfo)]
    L.de265hip_record_blk_planes.argtypes = [vp, vp, vp, vp]
    L.de265hip_recorder_desc.argtypes = [vp]
    L.de265hip_recorder_desc.restype = pp(_abi.PictureDesc)
    L.de265hip_recorder_submit.argtypes = [vp, i32, vp, pp(vp)]
    for n in ("de265hip_fn_transform_add",):
        getattr(L, n).argtypes = [i32, i32, i32, vp, C.c_ssize_t, i32, i32, vp, vp]
    for n in ("de265hip_fn_transform_skip_add", "de265hip_fn_transform_bypass_add"):
        getattr(L, n).argtypes = [i32, i32, vp, C.c_ssize_t, i32, i32, vp, vp]
    for n in ("de265hip_fn_put_qpel", "de265hip_fn_put_epel"):
        getattr(L, n).argtypes = [i32, vp, C.c_ssize_t, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    L.de265hip_fn_put_pred.argtypes = [i32, i32, vp, C.c_ssize_t, i32, i32, i32, i32, vp, vp, vp] + [i32] * 5
    _lib = L
    return L


def _chk(rc, where):
    if rc != 0:
        raise De265HipError(rc, where)


def device_count():
    return lib().de265hip_device_count()


class Picture:
    """Device-resident command buffers of one picture (de265hip_picture)."""

    def __init__(self, dec, handle):
        self.dec, self._h = dec, handle

    def stats(self):
        s = _abi.PictureStats()
        _chk(lib().de265hip_picture_get_stats(self._h, C.byref(s)), "picture_get_stats")
        return s

    def free(self):
        if self._h:
            lib().de265hip_picture_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Pipeline:
    """the C ABI's picture pipeline (de265hip_pipeline_*): pictures are prepared + built on worker threads and
    launched in submission order; submit(slot, make_recorder, planes) -> ticket, wait(ticket)."""

    def __init__(self, dec, n_workers=2):
        self._dec, self._h, self._keep = dec, C.c_void_p(), {}
        _chk(lib().de265hip_pipeline_new(C.byref(self._h), dec._h, n_workers), "pipeline_new")

    def submit(self, slot, make_recorder, pinned=None):
        """make_recorder() -> backend.Recorder, called on a pipeline worker thread; pinned: _PinnedPlanes or None"""
        def cb(user, out):
            try:
                rec = make_recorder()
                out[0] = rec._h.value
                rec._h = C.c_void_p()                 # the pipeline frees it
                return 0
            except Exception:                          # noqa: BLE001 - reported through the C error path (pipeline_wait of this ticket)
                import traceback
                traceback.print_exc()
                return 18
        fn = PREPARE_FN(cb)
        planes = (C.c_void_p * 3)(*(pinned.ptrs if pinned else [None] * 3))
        strides = (C.c_ssize_t * 3)(*(pinned.strides if pinned else [0] * 3))
        t = C.c_uint64()
        _chk(lib().de265hip_pipeline_submit(self._h, slot, fn, None, planes, strides, C.byref(t)), "pipeline_submit")
        self._keep[t.value] = (fn, planes, strides)   # the callback object must outlive the call on the worker thread
        return t.value

    def submit_desc(self, slot, desc, pinned=None):
        """a ready-made description (POINTER(PictureDesc), kept alive by the caller until the ticket is waited for)"""
        planes = (C.c_void_p * 3)(*(pinned.ptrs if pinned else [None] * 3))
        strides = (C.c_ssize_t * 3)(*(pinned.strides if pinned else [0] * 3))
        t = C.c_uint64()
        dptr = desc if isinstance(desc, C.POINTER(_abi.PictureDesc)) else C.pointer(desc)
        _chk(lib().de265hip_pipeline_submit_desc(self._h, slot, dptr, planes, strides, C.byref(t)), "pipeline_submit_desc")
        self._keep[t.value] = (dptr, planes, strides)
        return t.value

    def wait(self, ticket):
        _chk(lib().de265hip_pipeline_wait(self._h, ticket), "pipeline_wait")
        self._keep.pop(ticket, None)

    def drain(self):
        _chk(lib().de265hip_pipeline_drain(self._h), "pipeline_drain")
        self._keep.clear()

    def close(self):
        if self._h:
            lib().de265hip_pipeline_free(self._h)
            self._h = C.c_void_p()
            self._keep.clear()


class PinnedPlanes:
    """three planes in de265hip_host_alloc memory (numpy views in .planes), for asynchronous copy-outs"""

    def __init__(self, width, height, bit_depth, chroma_format=1):
        dt = np.uint16 if bit_depth > 8 else np.uint8
        self.ptrs, self.planes, self.strides = [], [], []
        cw = 0 if chroma_format == 0 else (width if chroma_format == 3 else width // 2)      # SubWidthC / SubHeightC (sps.cc:540-552)
        ch = 0 if chroma_format == 0 else (height // 2 if chroma_format == 1 else height)
        for sh in [(height, width), (ch, cw), (ch, cw)]:
            nbytes = sh[0] * sh[1] * np.dtype(dt).itemsize
            ptr = lib().de265hip_host_alloc(nbytes)
            if not ptr:
                raise MemoryError("de265hip_host_alloc(%d)" % nbytes)
            self.ptrs.append(ptr)
            self.strides.append(sh[1] * np.dtype(dt).itemsize)
            self.planes.append(np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dt).reshape(sh))

    def free(self):
        self.planes = []
        for ptr in self.ptrs:
            lib().de265hip_host_free(ptr)
        self.ptrs = []


class _PendingDownload:
    def __init__(self, dec, slot, shapes, dt):
        self._dec, self._slot, self._ptrs, self.planes = dec, slot, [], []
        for c, sh in enumerate(shapes):
            nbytes = sh[0] * sh[1] * np.dtype(dt).itemsize
            ptr = lib().de265hip_host_alloc(nbytes)
            if not ptr:
                raise MemoryError("de265hip_host_alloc(%d)" % nbytes)
            self._ptrs.append(ptr)
            self.planes.append(np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dt).reshape(sh))
            _chk(lib().de265hip_dpb_download_async(dec._h, slot, c, ptr, sh[1] * np.dtype(dt).itemsize), "dpb_download_async")

    def wait(self):
        _chk(lib().de265hip_dpb_wait(self._dec._h, self._slot), "dpb_wait")
        return self.planes

    def free(self):
        self.planes = []
        for ptr in self._ptrs:
            lib().de265hip_host_free(ptr)
        self._ptrs = []


class Decoder:
    """Mirror of de265hip_decoder: one HIP stream + device-resident DPB."""

    def __init__(self, device=-1):
        h = C.c_void_p()
        _chk(lib().de265hip_decoder_new(C.byref(h), device), "decoder_new")
        self._h = h

    def close(self):
        if self._h:
            lib().de265hip_decoder_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- DPB ---
    def dpb_alloc(self, slot, width, height, bit_depth_luma, bit_depth_chroma=None, chroma_format=1):
        bdc = bit_depth_luma if bit_depth_chroma is None else bit_depth_chroma
        _chk(lib().de265hip_dpb_alloc_ex(self._h, slot, width, height, bit_depth_luma, bdc, chroma_format), "dpb_alloc")

    def _chroma_dims(self, slot, w, h):
        cf = lib().de265hip_dpb_chroma_format(self._h, slot)
        if cf == 0:
            return (0, 0)                      # monochrome: empty chroma planes
        return (w if cf == 3 else w // 2, h // 2 if cf == 1 else h)

    def dpb_info(self, slot):
        """(width, height, bit_depth_luma, bit_depth_chroma) of the picture the slot holds."""
        w, h, by, bc = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _chk(lib().de265hip_dpb_info(self._h, slot, C.byref(w), C.byref(h), C.byref(by), C.byref(bc)), "dpb_info")
        return w.value, h.value, by.value, bc.value

    def _check_planes(self, slot, shapes_dtypes, what):
        w, h, by, bc = self.dpb_info(slot)
        cw, ch = self._chroma_dims(slot, w, h)
        for c, (shape, dt) in enumerate(shapes_dtypes):
            want = (h, w) if c == 0 else (ch, cw)
            wdt = np.uint16 if (by if c == 0 else bc) > 8 else np.uint8
            if tuple(shape) != want or np.dtype(dt) != np.dtype(wdt):
                raise ValueError("%s: plane %d is %s %s, DPB slot %d holds %s %s" % (what, c, tuple(shape), np.dtype(dt), slot, want, np.dtype(wdt)))

    def upload(self, slot, planes):
        # the C entry point copies the slot's geometry from `src` (like memcpy, it trusts the caller): check here
        self._check_planes(slot, [(p.shape, p.dtype) for p in planes], "upload")
        for c, p in enumerate(planes):
            p = np.ascontiguousarray(p)
            _chk(lib().de265hip_dpb_upload(self._h, slot, c, p.ctypes.data, p.strides[0]), "dpb_upload")

    def fill(self, slot, y, cb, cr):
        """every sample of the slot's planes set to one value per component (an unavailable reference picture, decctx.cc:1408-1434)"""
        _chk(lib().de265hip_dpb_fill(self._h, slot, int(y), int(cb), int(cr)), "dpb_fill")

    def download(self, slot, width, height, bit_depth):
        dt = np.uint16 if bit_depth > 8 else np.uint8
        cw, ch = self._chroma_dims(slot, width, height)
        self._check_planes(slot, [((height, width), dt), ((ch, cw), dt), ((ch, cw), dt)], "download")
        out = []
        for c, (w, h) in enumerate([(width, height), (cw, ch), (cw, ch)]):
            a = np.empty((h, w), dt)
            _chk(lib().de265hip_dpb_download(self._h, slot, c, a.ctypes.data, a.strides[0]), "dpb_download")
            out.append(a)
        return out

    def download_async(self, slot, width, height, bit_depth):
        """enqueue the copy-out of a decoded picture into pinned planes without waiting; returns a handle whose
        wait() blocks until the planes have landed and hands them out (numpy views of the pinned memory, valid until free())."""
        dt = np.uint16 if bit_depth > 8 else np.uint8
        cw, ch = self._chroma_dims(slot, width, height)
        shapes = [(height, width), (ch, cw), (ch, cw)]
        self._check_planes(slot, [(sh, dt) for sh in shapes], "download_async")
        return _PendingDownload(self, slot, shapes, dt)

    def download_planes_async(self, slot, ptrs, strides):
        """de265hip_dpb_download_planes_async: all planes of the slot's picture in one call (ptrs[c] None skips plane c)."""
        planes = (C.c_void_p * 3)(*ptrs)
        st = (C.c_ssize_t * 3)(*strides)
        cid = C.c_uint64()
        _chk(lib().de265hip_dpb_download_planes_async(self._h, slot, planes, st, C.byref(cid)), "dpb_download_planes_async")
        return cid.value

    def wait_slot(self, slot, copy_out_id=None):
        if copy_out_id is None:
            _chk(lib().de265hip_dpb_wait(self._h, slot), "dpb_wait")
        else:
            _chk(lib().de265hip_dpb_wait_copy_out(self._h, slot, copy_out_id), "dpb_wait_copy_out")

    def plane(self, slot, c_idx):
        p, s = C.c_void_p(), C.c_ssize_t()
        _chk(lib().de265hip_dpb_plane(self._h, slot, c_idx, C.byref(p), C.byref(s)), "dpb_plane")
        return p.value, s.value

    def copy_slot_to(self, slot, other, other_slot):
        """Device-to-device hand-over of a finished reference picture to another decoder."""
        _chk(lib().de265hip_dpb_copy(self._h, slot, other._h, other_slot), "dpb_copy")

    # --- pictures ---
    def build(self, dst_slot, desc):
        dptr = desc if isinstance(desc, C.POINTER(_abi.PictureDesc)) else C.pointer(desc)
        h = C.c_void_p()
        _chk(lib().de265hip_picture_build(self._h, dst_slot, dptr, C.byref(h)), "picture_build")
        return Picture(self, h)

    def run(self, pic, last_stage=_abi.STAGE_FINAL):
        _chk(lib().de265hip_picture_run(self._h, pic._h, last_stage), "picture_run")

    def set_lanes(self, n_lanes):
        """Picture-level concurrency inside this decoder: independent pictures on up to 4 HIP streams (de265_hip.h)."""
        _chk(lib().de265hip_decoder_set_lanes(self._h, n_lanes), "decoder_set_lanes")

    def sync(self):
        _chk(lib().de265hip_decoder_sync(self._h), "decoder_sync")

    def decode_picture(self, dst_slot, desc):
        dptr = desc if isinstance(desc, C.POINTER(_abi.PictureDesc)) else C.pointer(desc)
        _chk(lib().de265hip_decode_picture(self._h, dst_slot, dptr), "decode_picture")

    # --- profiling ---
    def set_profiling(self, on, only=None):
        """on: time every kernel's launches with hipEvents; only=[kernel names]: just those (each timed launch costs
        two event records on the stream)."""
        v = int(bool(on))
        if on and only is not None:
            v = 0
            for name in only:
                v |= 2 << _abi.K_NAMES.index(name)
        _chk(lib().de265hip_set_profiling(self._h, v), "set_profiling")

    def kernel_times(self, reset=True):
        ms = (C.c_double * len(_abi.K_NAMES))()
        n = (C.c_int64 * len(_abi.K_NAMES))()
        _chk(lib().de265hip_get_kernel_times(self._h, ms, n, int(reset)), "get_kernel_times")
        return {k: (ms[i], n[i]) for i, k in enumerate(_abi.K_NAMES)}


class Recorder:
    """Incremental form of the frame-level interface (de265hip_recorder_*): what a host parser calls
    per TU / PU / PCM block; submit() == Decoder.build() on the accumulated description."""

    def __init__(self, params, scaling_factors=None):
        h = C.c_void_p()
        sf = scaling_factors.ctypes.data if scaling_factors is not None else None
        _chk(lib().de265hip_recorder_new(C.byref(h), C.byref(params), sf), "recorder_new")
        self._h = h

    def record_desc(self, d):
        """Replays an existing description call by call (used by tests)."""
        L = lib()
        cv = C.cast(d.coeff_val, C.c_void_p).value or 0
        cp = C.cast(d.coeff_pos, C.c_void_p).value or 0
        for i in range(d.n_slices):
            _chk(L.de265hip_record_slice(self._h, C.byref(d.slices[i])), "record_slice")
        for i in range(d.n_ctbs):
            _chk(L.de265hip_record_ctb(self._h, i, C.byref(d.ctbs[i])), "record_ctb")
        for i in range(d.n_tus):
            t = d.tus[i]
            _chk(L.de265hip_record_tu(self._h, C.byref(t), cv + 2 * t.coeff_offset, cp + 2 * t.coeff_offset), "record_tu")
        for i in range(d.n_pus):
            _chk(L.de265hip_record_pu(self._h, C.byref(d.pus[i])), "record_pu")
        ps = C.cast(d.pcm_samples, C.c_void_p).value or 0
        for i in range(d.n_pcms):
            p = d.pcms[i]
            _chk(L.de265hip_record_pcm(self._h, p.x0, p.y0, p.log2_cb_size, ps + 2 * p.sample_offset), "record_pcm")
        _chk(L.de265hip_record_blk_planes(self._h, C.cast(d.blk_flags, C.c_void_p), C.cast(d.blk_qp_y, C.c_void_p),
                                          C.cast(d.blk_motion, C.c_void_p)), "record_blk_planes")

    @property
    def desc(self):
        return lib().de265hip_recorder_desc(self._h)

    def submit(self, dec, dst_slot):
        h = C.c_void_p()
        _chk(lib().de265hip_recorder_submit(dec._h, dst_slot, self._h, C.byref(h)), "recorder_submit")
        return Picture(dec, h)

    def free(self):
        if self._h:
            lib().de265hip_recorder_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def derive_edge_flags(params, slices, n_slices, ctbs, cb_log2_size, cb_part_mode, tu_split, blk_flags):
    """a11 host helper; ORs the edge bits into blk_flags (numpy uint8, in place)."""
    _chk(lib().de265hip_derive_edge_flags(C.byref(params), slices, n_slices, ctbs,
                                          cb_log2_size.ctypes.data, cb_part_mode.ctypes.data,
                                          tu_split.ctypes.data, blk_flags.ctypes.data), "derive_edge_flags")
    return blk_flags


# ---------------- function-level (vtable-shaped) interface ----------------
def _xy(blocks):
    return np.ascontiguousarray(np.asarray(blocks, dtype=np.int32).reshape(-1, 2))


def fn_transform_add(plane, bit_depth, log2_size, blocks, coeffs, dst=False):
    xy = _xy(blocks)
    coeffs = np.ascontiguousarray(coeffs, dtype=np.int16)
    _chk(lib().de265hip_fn_transform_add(log2_size, int(dst), bit_depth, plane.ctypes.data, plane.shape[1],
                                         plane.shape[0], len(xy), xy.ctypes.data, coeffs.ctypes.data),
         "fn_transform_add")


def fn_transform_skip_add(plane, bit_depth, log2_size, blocks, coeffs):
    xy = _xy(blocks)
    coeffs = np.ascontiguousarray(coeffs, dtype=np.int16)
    _chk(lib().de265hip_fn_transform_skip_add(log2_size, bit_depth, plane.ctypes.data, plane.shape[1],
                                              plane.shape[0], len(xy), xy.ctypes.data, coeffs.ctypes.data),
         "fn_transform_skip_add")


def fn_transform_bypass_add(plane, bit_depth, log2_size, blocks, coeffs):
    xy = _xy(blocks)
    coeffs = np.ascontiguousarray(coeffs, dtype=np.int16)
    _chk(lib().de265hip_fn_transform_bypass_add(log2_size, bit_depth, plane.ctypes.data, plane.shape[1],
                                                plane.shape[0], len(xy), xy.ctypes.data, coeffs.ctypes.data),
         "fn_transform_bypass_add")


def fn_put_qpel(plane, bit_depth, w, h, dx, dy, blocks):
    xy = _xy(blocks)
    out = np.empty((len(xy), h, w), np.int16)
    _chk(lib().de265hip_fn_put_qpel(bit_depth, plane.ctypes.data, plane.shape[1], plane.shape[1], plane.shape[0],
                                    w, h, dx, dy, len(xy), xy.ctypes.data, out.ctypes.data), "fn_put_qpel")
    return out


def fn_put_epel(plane, bit_depth, w, h, mx, my, blocks):
    xy = _xy(blocks)
    out = np.empty((len(xy), h, w), np.int16)
    _chk(lib().de265hip_fn_put_epel(bit_depth, plane.ctypes.data, plane.shape[1], plane.shape[1], plane.shape[0],
                                    w, h, mx, my, len(xy), xy.ctypes.data, out.ctypes.data), "fn_put_epel")
    return out


def fn_put_pred(plane, bit_depth, mode, blocks, src0, src1=None, w0=0, o0=0, w1=0, o1=0, log2wd=1):
    xy = _xy(blocks)
    src0 = np.ascontiguousarray(src0, dtype=np.int16)
    h, w = src0.shape[1], src0.shape[2]
    s1 = None
    if src1 is not None:
        src1 = np.ascontiguousarray(src1, dtype=np.int16)
        s1 = src1.ctypes.data
    _chk(lib().de265hip_fn_put_pred(mode, bit_depth, plane.ctypes.data, plane.shape[1], plane.shape[0], w, h,
                                    len(xy), xy.ctypes.data, src0.ctypes.data, s1, w0, o0, w1, o1, log2wd),
         "fn_put_pred")
