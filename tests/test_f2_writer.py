"""the synthetic bitstream writer (oracle/f2_writer.cc) against the reference's parser, in the build container
(the writer links the compiled reference; `make -C oracle f1 f2`).  For a few seeds per configuration: the stream decodes
without a warning (wrong entry points, a failed SEI picture hash or a CTB outside the picture are warnings), the decoder parsed
exactly the PUs / PCM blocks / coefficients the writer coded (any mis-binarised or mis-contexted bin desynchronises CABAC and
changes those), and the CPU restatement replays libde265's pictures.  tools/f2_check.py is the same check for long sweeps."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF = os.path.join(ROOT, "oracle", "_ref")
pytestmark = pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "f2_writer")) and os.path.exists(os.path.join(REF, "f1_dec"))),
                                reason="oracle/_ref/f2_writer + f1_dec are built in the build container only (make -C oracle f1 f2)")

CONFIGS = {
    "intra_all_tools": "gop=I pics=2 w=200 h=136 log2ctb=4 log2maxtb=4 sdh=1 tskip=1 tqbypass=1",
    "p_lists_mod": "gop=P pics=3 w=176 h=144 nref=3 lists_mod=1",
    "hier_b_10bit_weighted_slices": "gop=B pics=5 w=192 h=128 bits=10 wp=1 slices=3 log2ctb=6",
    "ldb_wpp_scaling_lists": "gop=LDB pics=3 w=256 h=192 wpp=1 slices=2 scaling=2",
    "p_tiles": "gop=P pics=3 w=256 h=192 log2ctb=4 log2maxtb=4 tile_cols=3 tile_rows=2 tile_uniform=0 lf_tiles=0 slices=4",
    "p_wpp_dependent_segments": "gop=P pics=3 w=256 h=192 dep=60 wpp=1 slices=2",
    "i_12bit_pcm": "gop=I pics=1 w=136 h=104 bits=12 pcm_bits=9 pcm_lf_off=1 cip=1",
}


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_writer_and_reference_parser_agree_and_oracle_replays(name):
    import f2_check
    f2_check.REPLAY = True
    for seed in (1, 2):
        err, want, got = f2_check.run(CONFIGS[name].split(), seed)
        assert err is None, "%s seed %d: %s" % (name, seed, err)
        assert len(got) == len(want) and sum(w["coeffs"] for w in want) > 0


def test_writer_rejects_inconsistent_settings():
    import subprocess
    r = subprocess.run([os.path.join(REF, "f2_writer"), "out=/dev/null", "w=100", "h=64"], capture_output=True, text=True)
    assert r.returncode == 2 and "multiples" in r.stderr
