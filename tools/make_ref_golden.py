#!/usr/bin/env python3
"""Writes the REFERENCE-GENERATED fixtures under tests/golden/ (build container only: needs the compiled
reference, oracle/_ref/libde265_ref.so = make -C oracle ref):

  ref_pictures.json         per picture case of tools/ref_cases.py: MD5 of the planes after each stage
                            (prefilter / deblocked / final), of the derived edge flags and of both
                            boundary-strength passes, as libde265's own functions produce them
  ref_functions.json        per function case: MD5 of the outputs of the fallback vtable slots
  ref_small_pictures.npz    three small pictures' final planes in full

    python tools/make_ref_golden.py

The tests (tests/test_ref_golden.py) check that the CPU restatement and the HIP path reproduce these, and,
where the compiled reference is present, that it still does."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import pyref  # noqa: E402
import ref_cases  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    ver = pyref.lib().ref_version().decode()
    pics, full = {}, {}
    for c in ref_cases.PICTURE_CASES:
        keep = c["name"] in ref_cases.FULL_PICTURE_CASES
        r = ref_cases.picture_record(c, "ref", keep_final=keep)
        if keep:
            r, final = r
            for i, p in enumerate(final):
                full["%s/%d" % (c["name"], i)] = p
        pics[c["name"]] = r
        print("picture", c["name"], r["final"])
    json.dump({"_generated_by": "tools/make_ref_golden.py from libde265 %s compiled by oracle/Makefile" % ver,
               "cases": pics}, open(os.path.join(GOLD, "ref_pictures.json"), "w"), indent=1, sort_keys=True)
    fns = {}
    for case in ref_cases.function_cases():
        fns[case["key"]] = ref_cases.digest([ref_cases.run_function_case(case, "ref")])
    json.dump({"_generated_by": "tools/make_ref_golden.py from libde265 %s compiled by oracle/Makefile" % ver,
               "cases": fns}, open(os.path.join(GOLD, "ref_functions.json"), "w"), indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(GOLD, "ref_small_pictures.npz"), **full)
    print("wrote %d picture cases, %d function cases, %d full planes" % (len(pics), len(fns), len(full)))


if __name__ == "__main__":
    main()
