"""Canonical view of a picture's run-side structures (runs, their TU lists, producers, mailbox segments, level-0 task lists,
ticket slots), read back from its arena through de265hip_debug_picture_layout / _read.

Two scans build these structures: the round-3 host scan (host.hip) and the round-4 passes of scan_core.h (on the device, or -
the CPU rehearsal - on the host).  They lay them out differently (sorted run records against sparse ids, lists in another
order), so the comparison is by content: a run is named by its colour component and the smallest (y, x) of its TUs, a
producer by its run's name, a residual-only task by the run its block belongs to.  Test infrastructure.
"""
import bisect
import ctypes as C

import numpy as np

from libde265_amd import backend

RUN = np.dtype([("x0", "<u2"), ("y0", "<u2"), ("x1", "<u2"), ("y1", "<u2"), ("wx1", "<u2"), ("wy1", "<u2"), ("c_idx", "u1"),
                ("micro", "u1"), ("n_tus", "<u2"), ("first_tu", "<u4"), ("dep_offset", "<u4"), ("n_deps", "<u2"), ("n_lvls", "<u2"),
                ("res_offset", "<u4"), ("n_samples", "<u4"), ("wave_end", "<u2", (4,))])
TU = np.dtype([("x0", "<u2"), ("y0", "<u2"), ("log2", "u1"), ("c_idx", "u1"), ("flags", "u1"), ("mode", "u1"), ("qp", "i1"),
               ("run_level", "u1"), ("n_coeff", "<u2"), ("coeff_offset", "<u4"), ("avail", "<u8"), ("resid_offset", "<u4"),
               ("angle", "i1"), ("pad3", "u1"), ("inv_angle", "<i2")])
assert RUN.itemsize == 44 and TU.itemsize == 32
RESID_ONLY = 0x80


def build_dry(desc, mode):
    """de265hip_debug_build_host_only_ex: mode 0 the round-3 host scan, 2 the passes of scan_core.h on the CPU; returns the
    picture handle (free with backend.lib().de265hip_picture_free)."""
    L = backend.lib()
    h = C.c_void_p()
    rc = L.de265hip_debug_build_host_only_ex(desc, 1, mode, C.byref(h))
    if rc:
        raise backend.De265HipError(rc, "debug_build_host_only_ex(mode %d)" % mode)
    return h


def _read(h, off, n, dt):
    a = np.zeros(n, dt)
    if n:
        rc = backend.lib().de265hip_debug_picture_read(h, int(off), int(a.nbytes), a.ctypes.data)
        if rc:
            raise backend.De265HipError(rc, "debug_picture_read")
    return a


def canon(h):
    """-> dict(runs={name: record}, l0=[multiset per size class], l0x=multiset, counts={...}); raises AssertionError on a
    structure that is inconsistent in itself (a run without a ticket, a producer behind its reader ...)."""
    L = backend.lib()
    lay = (C.c_int64 * 32)()
    rc = L.de265hip_debug_picture_layout(h, lay)
    if rc:
        raise backend.De265HipError(rc, "debug_picture_layout")
    (o_runs, o_rtus, o_deps, o_slots, o_l0, o_l0x, o_mbx, o_segs, o_front, o_ntus, o_nall, o_level, o_counts) = lay[0:13]
    dev, n_runs, n_front, n_batches = lay[13], lay[14], lay[15], lay[16]
    n_l0 = list(lay[17:21])
    n_l0x, n_rec, arena_bytes = lay[21], lay[22], lay[23]
    runs = _read(h, o_runs, n_rec, RUN)
    if dev:
        valid = np.nonzero(_read(h, o_ntus, n_rec, np.uint8))[0]
    else:
        valid = np.arange(n_runs)
    assert len(valid) == n_runs, (len(valid), n_runs)
    # the TU array and the producer pool: as far as the valid runs reach
    rv = runs[valid]
    n_tu_all = int((rv["first_tu"].astype(np.int64) + rv["n_tus"]).max()) if n_runs else 0
    tus = _read(h, o_rtus, n_tu_all, TU)
    n_dep_all = int((rv["dep_offset"].astype(np.int64) + 65536).max()) if n_runs else 0
    n_dep_all = min(n_dep_all, (arena_bytes - o_deps) // 4)
    deps = _read(h, o_deps, n_dep_all, np.uint32) if n_runs else np.zeros(0, np.uint32)
    mbx = _read(h, o_mbx, 3 * n_rec, np.uint32) if o_mbx >= 0 else None
    any_mb = bool(n_runs) and bool((rv["micro"] & 12).any())
    segs = _read(h, o_segs, (arena_bytes - o_segs) // 4 if dev else 0, np.uint32) if False else None
    name_of = {}
    tu_of = {}
    for r in valid:
        R = runs[r]
        t = tus[R["first_tu"]:R["first_tu"] + R["n_tus"]]
        assert len(t) == R["n_tus"] and (t["c_idx"] == R["c_idx"]).all()
        key = (int(R["c_idx"]), int((t["y0"].astype(np.int64) << 16 | t["x0"]).min()))
        assert key not in tu_of, ("two runs with the same first TU", key)
        name_of[int(r)] = key
        tu_of[key] = t
    front = set()
    if dev:
        for r in valid:
            if runs[r]["micro"] & 16:
                front.add(int(r))
        fi = _read(h, o_front, n_front, np.uint32)
        assert sorted(int(x) for x in fi) == sorted(front), "front_idx does not list the front runs"
    else:
        front = set(range(n_front))
    assert len(front) == n_front
    # mailbox id -> owner
    mb_owner = {}
    if mbx is not None:
        for r in valid:
            if runs[r]["micro"] & 8:
                mb_owner[int(mbx[3 * r])] = int(r)
    seg_cache = {}

    def seg_words(at, n):
        if (at, n) not in seg_cache:
            seg_cache[(at, n)] = _read(h, o_segs + 4 * int(at), n, np.uint32)
        return seg_cache[(at, n)]

    out = {}
    res_starts, res_names = [], []
    for r in valid:
        R = runs[r]
        key = name_of[int(r)]
        t = tu_of[key]
        nd = int(R["n_deps"])
        dl = [int(x) for x in deps[R["dep_offset"]:R["dep_offset"] + nd]]
        assert len(set(dl)) == nd and all(x in name_of for x in dl), ("producer list", key, dl)
        assert not any(x in front for x in dl), ("a front run in a producer list", key)
        rec = {
            "box": tuple(int(R[k]) for k in ("x0", "y0", "x1", "y1", "wx1", "wy1")), "micro": int(R["micro"]) & 15,
            "front": int(r) in front, "n_lvls": int(R["n_lvls"]), "wave_end": tuple(int(x) for x in R["wave_end"]),
            "n_samples": int(R["n_samples"]),
            "tus": [(int(a["x0"]), int(a["y0"]), int(a["log2"]), int(a["flags"]), int(a["mode"]), int(a["qp"]), int(a["run_level"]),
                     int(a["n_coeff"]), int(a["coeff_offset"]), int(a["avail"]), int(a["resid_offset"]) - int(R["res_offset"]),
                     int(a["angle"]), int(a["pad3"]), int(a["inv_angle"])) for a in t],
            "deps": sorted(name_of[x] for x in dl),
        }
        if R["micro"] & 8:
            at = int(mbx[3 * r + 2])
            rec["ready"] = None if at == 0xFFFFFFFF else tuple(int(x) for x in seg_words(at, 17))
        if R["micro"] & 4:
            at = int(mbx[3 * r + 1])
            hd = seg_words(at, 3)
            nseg, ngroups = int(hd[0]) & 0xFF, (int(hd[0]) >> 8) & 0xFF
            ends = [(int(hd[1]) >> (8 * g)) & 0xFF for g in range(4)]
            body = seg_words(at, 3 + 2 * nseg)[3:]
            samples, acc = set(), 0
            for q in range(nseg):
                a, b = int(body[2 * q]), int(body[2 * q + 1])
                cnt, col = ((a >> 24) & 63) + 1, a >> 31
                owner = mb_owner[a & 0xFFFFFF]
                for off in range(cnt):
                    s_ = acc + off
                    grp = 0 if ngroups <= 1 else sum(1 for g in range(ngroups - 1) if s_ >= ends[g])
                    samples.add((name_of[owner], col, (b & 63) + off, (b >> 8) + off * (112 if col else 1), grp))
                acc += cnt
            rec["reads"] = (ngroups, tuple(ends) if ngroups > 1 else None, int(hd[2]) if ngroups > 1 else None, frozenset(samples))
        out[key] = rec
        res_starts.append(int(R["res_offset"])); res_names.append((key, int(R["n_samples"])))
    order = np.argsort(res_starts, kind="stable")
    res_starts = [res_starts[i] for i in order]; res_names = [res_names[i] for i in order]
    for i in range(len(res_starts) - 1):
        assert res_starts[i] + res_names[i][1] <= res_starts[i + 1], "residual ranges of two runs overlap"

    def task_key(a):
        ro = None
        if a["flags"] & RESID_ONLY:
            i = bisect.bisect_right(res_starts, int(a["resid_offset"])) - 1
            assert i >= 0 and int(a["resid_offset"]) < res_starts[i] + res_names[i][1], "a residual-only task outside every run"
            ro = (res_names[i][0], int(a["resid_offset"]) - res_starts[i])
        return (int(a["x0"]), int(a["y0"]), int(a["log2"]), int(a["c_idx"]), int(a["flags"]), int(a["mode"]), int(a["qp"]),
                int(a["run_level"]), int(a["n_coeff"]), int(a["coeff_offset"]), int(a["avail"]), ro if ro else int(a["resid_offset"]),
                int(a["angle"]), int(a["pad3"]), int(a["inv_angle"]))
    l0 = _read(h, o_l0, sum(n_l0), TU)
    at = 0
    l0c = [None] * 4
    for k in (3, 2, 1, 0):                                 # sorted [32x32 | 16x16 | 8x8 | 4x4]
        part = l0[at:at + n_l0[k]]
        assert (part["log2"] == k + 2).all(), "level-0 list not sorted by size"
        l0c[k] = sorted(task_key(a) for a in part)
        at += n_l0[k]
    l0x = sorted(task_key(a) for a in _read(h, o_l0x, n_l0x, TU))
    # tickets: every run that is not a front run exactly once, producers in earlier tickets
    slots = _read(h, o_slots, 8 * n_batches, np.uint32)
    ticket = {}
    for i, v in enumerate(slots):
        v = int(v)
        if v == 0xFFFFFFFF:
            continue
        r = v & 0x7FFFFFFF
        assert r in name_of and r not in front and r not in ticket, ("slot", i, v)
        assert bool(v >> 31) == bool(runs[r]["micro"] & 1), "micro bit of a slot"
        if not (v >> 31):
            assert i % 8 == 0 and all(int(x) == 0xFFFFFFFF for x in slots[i + 1:i + 8]), "an ordinary run shares its ticket"
        ticket[r] = (i // 8, i % 8)
    missing = [name_of[int(r)] for r in valid if int(r) not in front and int(r) not in ticket]
    for r in valid:
        r = int(r)
        if r in ticket:
            R = runs[r]
            for x in deps[R["dep_offset"]:R["dep_offset"] + int(R["n_deps"])]:
                if int(x) in ticket:                        # (a producer without a ticket: the fault injection's victim)
                    assert ticket[int(x)] < ticket[r], ("a producer behind its reader", name_of[r])
    return {"runs": out, "l0": l0c, "l0x": l0x, "no_ticket": sorted(missing),
            "counts": {"n_runs": int(n_runs), "n_front": int(n_front), "n_l0": n_l0, "n_l0x": int(n_l0x)}}


def diff(a, b):
    """first difference between two canonical views, as text (None: equal)"""
    if a["counts"] != b["counts"]:
        return "counts %s vs %s" % (a["counts"], b["counts"])
    if set(a["runs"]) != set(b["runs"]):
        only = sorted(set(a["runs"]) ^ set(b["runs"]))[:5]
        return "run sets differ, e.g. %s" % (only,)
    for k in sorted(a["runs"]):
        ra, rb = a["runs"][k], b["runs"][k]
        for f in ra:
            if ra[f] != rb.get(f):
                return "run %s field %s: %r vs %r" % (k, f, ra[f] if f != "tus" else ra[f][:4], rb.get(f) if f != "tus" else rb[f][:4])
        if set(ra) != set(rb):
            return "run %s fields %s vs %s" % (k, sorted(ra), sorted(rb))
    for k in range(4):
        if a["l0"][k] != b["l0"][k]:
            return "level-0 tasks of size class %d differ" % k
    if a["l0x"] != b["l0x"]:
        return "k_resid_rext tasks differ"
    if a["no_ticket"] != b["no_ticket"]:
        return "runs without a ticket: %s vs %s" % (a["no_ticket"][:4], b["no_ticket"][:4])
    return None
