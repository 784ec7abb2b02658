"""Descriptor fuzzer for the host stage of de265hip_picture_build (no GPU: de265hip_debug_build_host_only).  Random corruptions of
a valid picture description must come back as an error code or build normally - never read or write out of bounds.  Meant for an
address-sanitizer build of the host code (host-only instrumentation; the device code is not touched):

    hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-gpu-sanitize -c host.hip pipeline.hip vtable.hip ...
    LD_PRELOAD=<libclang_rt.asan-x86_64.so> ASAN_OPTIONS=detect_leaks=0 DE265HIP_SO=<asan .so> python tools/fuzz_desc.py <seed> <n>
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import pysynth  # noqa: E402
from libde265_amd import backend, _abi  # noqa: E402

MODE = int(os.environ.get("FUZZ_MODE", "0"))


def run(seed, N):
    """-> {return code: count} over N corrupted descriptions"""
    L = backend.lib()
    rng = np.random.default_rng(seed)
    codes = {}
    for it in range(N):
        w, h = int(rng.integers(2, 40)) * 8, int(rng.integers(2, 30)) * 8
        bd = int(rng.choice([8, 10, 12])); st = int(rng.choice([0, 1, 2]))
        cf = int(rng.choice([1, 1, 2, 3]))
        over = dict(chroma_format=cf, tskip_pct=20, pcm_pct=int(rng.choice([0, 10])), n_slices=int(rng.integers(1, 4)), log2_ctb_size=int(rng.choice([4, 5, 6])),
                    cross_component_pct=30 if cf == 3 else 0, implicit_rdpcm=int(rng.integers(0, 2)))
        if over["log2_ctb_size"] == 4:
            over["log2_max_tb_size"] = 4
        sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, st, seed=int(rng.integers(1 << 30)), **over))
        d = sp.d
        n_tus, n_pus, n_pcms, n_ctbs, n_slices = d.n_tus, d.n_pus, d.n_pcms, d.n_ctbs, d.n_slices      # (the arrays' real sizes)
        mild = it % 2 == 1                                             # every other picture: values that pass the range checks (the deep paths)
        for _ in range(int(rng.integers(1, 6))):                      # a few corruptions per picture
            kind = int(rng.integers(0, 12))
            if mild and n_tus:
                t = d.tus[int(rng.integers(n_tus))]
                f = int(rng.integers(0, 7))
                cw = w if t.c_idx == 0 else (w if cf == 3 else w // 2)
                ch = h if t.c_idx == 0 else (h // 2 if cf == 1 else h)
                if f == 0: t.log2_size = int(rng.integers(2, 6)); t.n_coeff = min(t.n_coeff, 1 << (2 * t.log2_size))
                elif f == 1: t.x0 = 4 * int(rng.integers(0, max(1, (cw - (1 << t.log2_size)) // 4 + 1)))
                elif f == 2: t.y0 = 4 * int(rng.integers(0, max(1, (ch - (1 << t.log2_size)) // 4 + 1)))
                elif f == 3: t.intra_mode = int(rng.integers(0, 35))
                elif f == 4: t.flags = int(rng.integers(0, 256))
                elif f == 5: t.c_idx = int(rng.integers(0, 3))
                else:                                                  # a duplicate of another TU (overlap)
                    o2 = d.tus[int(rng.integers(n_tus))]
                    t.x0, t.y0, t.log2_size, t.c_idx = o2.x0, o2.y0, o2.log2_size, o2.c_idx; t.n_coeff = min(t.n_coeff, 1 << (2 * t.log2_size))
                continue
            if kind <= 4 and n_tus:
                t = d.tus[int(rng.integers(n_tus))]
                f = int(rng.integers(0, 9))
                if f == 0: t.x0 = int(rng.integers(0, 65536))
                elif f == 1: t.y0 = int(rng.integers(0, 65536))
                elif f == 2: t.log2_size = int(rng.integers(0, 9))
                elif f == 3: t.c_idx = int(rng.integers(0, 5))
                elif f == 4: t.flags = int(rng.integers(0, 256))
                elif f == 5: t.intra_mode = int(rng.integers(0, 256))
                elif f == 6: t.n_coeff = int(rng.integers(0, 65536))
                elif f == 7: t.coeff_offset = int(rng.integers(0, 1 << 32))
                else: t.res_scale_val = int(rng.integers(-128, 128))
            elif kind <= 6 and n_pus:
                q = d.pus[int(rng.integers(n_pus))]
                f = int(rng.integers(0, 7))
                if f == 0: q.x = int(rng.integers(0, 65536))
                elif f == 1: q.y = int(rng.integers(0, 65536))
                elif f == 2: q.w = int(rng.integers(0, 256))
                elif f == 3: q.h = int(rng.integers(0, 256))
                elif f == 4: q.pred_flag = int(rng.integers(0, 256))
                elif f == 5: q.slice_idx = int(rng.integers(0, 65536))
                else: q.ref_idx[int(rng.integers(2))] = int(rng.integers(-128, 128))
            elif kind == 7 and n_pcms:
                q = d.pcms[int(rng.integers(n_pcms))]
                f = int(rng.integers(0, 4))
                if f == 0: q.x0 = int(rng.integers(0, 65536))
                elif f == 1: q.y0 = int(rng.integers(0, 65536))
                elif f == 2: q.log2_cb_size = int(rng.integers(0, 9))
                else: q.sample_offset = int(rng.integers(0, 1 << 32))
            elif kind == 8:
                c = d.ctbs[int(rng.integers(n_ctbs))]
                c.slice_idx = int(rng.integers(0, 65536)) if rng.integers(2) else c.slice_idx
                c.slice_addr_rs = int(rng.integers(0, 1 << 31)) if rng.integers(2) else c.slice_addr_rs
            elif kind == 9:
                s = d.slices[int(rng.integers(n_slices))]
                s.slice_type = int(rng.integers(0, 5))
                s.ref_pic_list[int(rng.integers(2))][int(rng.integers(16))] = int(rng.integers(-128, 128))
            elif kind == 10:
                P = d.params
                f = int(rng.integers(0, 6))
                if f == 0: P.num_tile_columns = int(rng.integers(0, 30))
                elif f == 1: P.num_tile_rows = int(rng.integers(0, 30))
                elif f == 2: P.col_bd[int(rng.integers(0, 4))] = int(rng.integers(0, 100))
                elif f == 3: P.log2_min_tb_size = int(rng.integers(0, 8))
                elif f == 4: P.chroma_format_idc = int(rng.integers(0, 5))
                else: P.log2_ctb_size = int(rng.integers(0, 9))
            else:
                f = int(rng.integers(0, 4))
                if f == 0: d.n_coeffs = int(rng.integers(0, max(1, d.n_coeffs)))        # (shorter than what the TUs refer to)
                elif f == 1: d.n_pcm_samples = int(rng.integers(0, max(1, d.n_pcm_samples)))
                elif f == 2: d.n_ctbs = int(rng.integers(0, n_ctbs + 1))
                else: d.n_slices = int(rng.integers(0, n_slices + 1))                # (fewer than the CTBs / PUs refer to)
        rc = L.de265hip_debug_build_host_only_ex(sp.desc, 1, MODE, None)      # FUZZ_MODE=2: the passes of scan_core.h (CPU rehearsal)
        codes[rc] = codes.get(rc, 0) + 1
        sp.close()
    return codes


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    print("fuzzed %d descriptors; return codes:" % n, dict(sorted(run(seed, n).items())))
