#!/usr/bin/env python3
"""Remove the coded slice NAL units of picture number k (decode order, 0-based) from an Annex-B HEVC bitstream: a LOST picture.
libde265 then finds a reference missing from its DPB and synthesises one (generate_unavailable_reference_picture,
decctx.cc:1408-1434) - the case the back end's de265hip_dpb_fill exists for.  Test infrastructure.
    python tools/drop_picture.py in.bin out.bin k"""
import sys


def nal_units(data):
    """-> list of (start_of_start_code, start_of_payload, end)"""
    out, i, n = [], 0, len(data)
    starts = []
    while True:
        j = data.find(b"\x00\x00\x01", i)
        if j < 0:
            break
        starts.append(j)
        i = j + 3
    for a, s in enumerate(starts):
        end = starts[a + 1] if a + 1 < len(starts) else n
        sc = s - 1 if s > 0 and data[s - 1] == 0 else s          # a four-byte start code
        if a + 1 < len(starts) and end > 0 and data[end - 1] == 0:
            end -= 1                                             # the leading zero of the next start code
        out.append((sc, s + 3, end))
    return out


def drop_picture(data, k):
    keep, pic = [], -1
    for sc, p, e in nal_units(data):
        nal_type = (data[p] >> 1) & 0x3F
        if nal_type < 32:                                        # VCL: a slice segment
            if data[p + 2] & 0x80:                               # first_slice_segment_in_pic_flag
                pic += 1
            if pic == k:
                continue
        keep.append(data[sc:e])
    return b"".join(keep), pic + 1


if __name__ == "__main__":
    src, dst, k = sys.argv[1], sys.argv[2], int(sys.argv[3])
    out, n = drop_picture(open(src, "rb").read(), k)
    open(dst, "wb").write(out)
    print("%d pictures, picture %d dropped" % (n, k))
