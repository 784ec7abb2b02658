"""Interface 1 of the boundary: init_acceleration_functions_hip(struct acceleration_functions*),
include/de265_hip_vtable.h, libde265_amd/csrc/vtable.hip.

CPU, build container:  a translation unit that includes the REFERENCE's acceleration.h proves the layout mirror
                       (size, every offset) and the slot types (assignment both ways) and that the entry point has the
                       shape of init_acceleration_functions_fallback (fallback.cc:26).
GPU + compiled reference: the product fills a real struct acceleration_functions behind the fallback, as
                       decctx.cc:430-449 would, and every decoder slot is called through it next to the fallback's slot
                       on seeded inputs (oracle/ref_shim.cc ref_vtable_compare): 0 mismatches, n = 1 per call."""
import ctypes as C
import os
import subprocess

import pytest

import pyref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILIES = ["put_*_pred", "put_hevc_epel*", "put_hevc_qpel", "transform_add / dst_add", "int32-residual family",
            "add_residual", "rotate_coefficients", "transform_skip_rdpcm_*_8", "slots that keep / lose the fallback pointer"]


@pytest.mark.skipif(not pyref.can_build(), reason="needs the reference's headers (/root/reference)")
def test_layout_mirror_and_slot_types_against_the_reference_header():
    ref = pyref.REF_ROOT
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-I%s/libde265" % ref, "-I%s" % ref, "-idirafter", "%s/extra" % ref,
                        "-I%s/include" % ROOT, os.path.join(ROOT, "tests", "native", "vtable_layout_check.cc")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_product_exports_the_entry_point():
    from libde265_amd import backend
    assert hasattr(backend.lib(), "init_acceleration_functions_hip")


@pytest.mark.gpu
@pytest.mark.skipif(not pyref.available(), reason="compiled reference (oracle/_ref) not present here")
def test_every_decoder_slot_matches_the_fallback_through_the_reference_struct():
    from libde265_amd import backend
    assert backend.device_count() > 0
    init = C.cast(backend.lib().init_acceleration_functions_hip, C.c_void_p)
    counts = (C.c_int * 9)()
    calls = pyref.lib().ref_vtable_compare(init, 20261004, 64, counts)
    assert calls > 1000
    bad = {FAMILIES[k]: counts[k] for k in range(9) if counts[k]}
    assert not bad, bad
