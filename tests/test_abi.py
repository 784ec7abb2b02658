"""C-ABI checks that need no GPU: the shared library loads, exports every symbol
include/de265_hip.h declares, struct layouts match the ctypes mirrors, and the
host-only entry points validate their arguments."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

from libde265_amd import _abi, backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "de265_hip.h")


def declared_functions():
    """every function include/*.h declares: de265hip_* of de265_hip.h + the vtable entry point of de265_hip_vtable.h"""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(de265hip_\w+)\s*\(", txt))
    vt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "de265_hip_vtable.h")).read(), flags=re.S)
    names |= set(re.findall(r"\bvoid\s+(init_acceleration_functions_\w+)\s*\(", vt))
    return sorted(names)


def test_library_is_built_and_loads():
    assert os.path.exists(backend.SO_PATH), "run python -m libde265_amd.build"
    assert b"gfx950" in backend.lib().de265hip_version()


def test_every_declared_symbol_is_exported():
    L = backend.lib()
    decl = declared_functions()
    assert len(decl) >= 20
    for name in decl:
        assert hasattr(L, name), "missing export " + name
    assert sorted(backend.EXPORTS) == decl, "backend.EXPORTS out of sync with include/de265_hip.h"


def test_code_object_is_gfx950():
    out = subprocess.run(["strings", backend.SO_PATH], capture_output=True, text=True).stdout
    assert "amdgcn-amd-amdhsa--gfx950" in out


def test_struct_layouts_match_header():
    names = {"de265hip_pic_params": _abi.PicParams, "de265hip_slice_params": _abi.SliceParams,
             "de265hip_ctb_info": _abi.CtbInfo, "de265hip_tu": _abi.TU, "de265hip_pu": _abi.PU,
             "de265hip_pcm": _abi.PCM, "de265hip_motion": _abi.Motion,
             "de265hip_picture_desc": _abi.PictureDesc, "de265hip_picture_stats": _abi.PictureStats}
    src = '#include <stdio.h>\n#include "de265_hip.h"\nint main(){\n'
    for n in names:
        src += 'printf("%s %%zu\\n", sizeof(%s));\n' % (n, n)
    src += 'printf("scaling %d\\n", DE265HIP_SCALING_BLOB_BYTES); printf("slots %d\\n", DE265HIP_MAX_DPB_SLOTS);'
    src += "return 0;}\n"
    with tempfile.TemporaryDirectory() as td:
        cfile, exe = os.path.join(td, "s.c"), os.path.join(td, "s")
        open(cfile, "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), cfile, "-o", exe])
        out = dict(l.split() for l in subprocess.check_output([exe], text=True).splitlines())
    for n, cls in names.items():
        assert int(out[n]) == C.sizeof(cls), n
    assert int(out["scaling"]) == _abi.SCALING_BLOB_BYTES
    assert int(out["slots"]) == _abi.MAX_DPB_SLOTS


def test_error_codes_follow_de265_error():
    # de265.h:82-139 numeric values
    assert (_abi.OK, _abi.ERROR_OUT_OF_MEMORY, _abi.ERROR_PARAMETER_OUT_OF_RANGE) == (0, 7, 8)
    assert (_abi.ERROR_INIT_FAILED, _abi.ERROR_DECODING, _abi.ERROR_NOT_IMPLEMENTED) == (11, 18, 502)


def test_null_arguments_are_rejected_without_a_gpu():
    L = backend.lib()
    assert L.de265hip_decoder_new(None, 0) == _abi.ERROR_PARAMETER_OUT_OF_RANGE
    assert L.de265hip_dpb_alloc(None, 0, 64, 64, 8, 8) == _abi.ERROR_PARAMETER_OUT_OF_RANGE
    assert L.de265hip_picture_run(None, None, 0) == _abi.ERROR_PARAMETER_OUT_OF_RANGE
    assert L.de265hip_decoder_sync(None) == _abi.ERROR_PARAMETER_OUT_OF_RANGE
    L.de265hip_decoder_free(None)
    L.de265hip_picture_free(None)


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a GPU the decoder cannot be created."""
    if backend.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(backend.De265HipError):
        backend.Decoder()
    plane = np.zeros((16, 16), np.uint8)
    with pytest.raises(backend.De265HipError):
        backend.fn_transform_add(plane, 8, 2, [(0, 0)], np.zeros((1, 4, 4), np.int16))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "libde265_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".inc")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in txt and "hevc_oracle" not in txt and "liboracle" not in txt, f


def test_bench_touches_the_oracle_only_in_its_cpu_baseline_leg():
    """oracle/ (the restatement, the compiled reference, the patched decoder, the bitstream writer) is test infrastructure:
    bench.py may use it as the CPU baseline / checker and nowhere else (the end-to-end stream rates come from tools/exp/)."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    allowed = [(n.lineno, n.end_lineno) for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "cpu_baseline"]
    assert allowed
    for i, line in enumerate(src.splitlines(), 1):
        code = line.split("#", 1)[0]
        if any(w in code for w in ("pyoracle", "pyref", "f1_dec", "f2_writer", "liboracle", "_ref/", "\"_ref\"")):
            assert any(a <= i <= b for a, b in allowed), "bench.py:%d uses the oracle outside cpu_baseline: %s" % (i, line.strip())
