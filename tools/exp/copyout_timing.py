"""How long does ENQUEUEING the asynchronous copy-out of a 4K Main10 picture take on the host (de265hip_dpb_download_async into
de265hip_host_alloc memory), and how long until it has landed?  GPU box."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401  (first: the wheel's ROCm runtime)
from libde265_amd import backend
dec = backend.Decoder()
w, h, bd = 3840, 2160, 10
dec.dpb_alloc(0, w, h, bd)
dec.upload(0, [np.full((h, w), 512, np.uint16), np.full((h // 2, w // 2), 300, np.uint16), np.full((h // 2, w // 2), 700, np.uint16)])
for rep in range(5):
    t0 = time.perf_counter()
    pend = dec.download_async(0, w, h, bd)       # includes 3 x host_alloc
    t1 = time.perf_counter()
    planes = pend.wait()
    t2 = time.perf_counter()
    assert int(planes[0][7, 9]) == 512 and int(planes[2][5, 5]) == 700
    pend.free()
    print("alloc+enqueue %.3f ms, wait %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
# enqueue only, memory allocated beforehand
L = backend.lib()
ptrs = [L.de265hip_host_alloc(n) for n in (w * h * 2, w * h // 2, w * h // 2)]
for rep in range(5):
    t0 = time.perf_counter()
    for c, (p, ww) in enumerate(zip(ptrs, (w, w // 2, w // 2))):
        rc = L.de265hip_dpb_download_async(dec._h, 0, c, p, ww * 2)
        assert rc == 0
    t1 = time.perf_counter()
    L.de265hip_dpb_wait(dec._h, 0)
    t2 = time.perf_counter()
    print("enqueue only %.3f ms, wait %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
dec.close()

# the same behind a picture whose kernels have just been enqueued (what the pipelined decoder does)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import pysynth
dec = backend.Decoder()
import pyoracle
for s in (0, 1, 2):
    dec.dpb_alloc(s, w, h, bd); dec.upload(s, pysynth.fill_planes(w, h, bd, s + 1))
sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, 2, seed=5))
ptrs = [L.de265hip_host_alloc(n) for n in (w * h * 2, w * h // 2, w * h // 2)]
for rep in range(6):
    pic = dec.build(2, sp.desc)
    dec.sync()
    t0 = time.perf_counter()
    dec.run(pic, 2)
    t1 = time.perf_counter()
    for c, (p, ww) in enumerate(zip(ptrs, (w, w // 2, w // 2))):
        assert L.de265hip_dpb_download_async(dec._h, 2, c, p, ww * 2) == 0
    t2 = time.perf_counter()
    L.de265hip_dpb_wait(dec._h, 2)
    t3 = time.perf_counter()
    pic.free()
    print("run %.3f ms, enqueue copy-out behind it %.3f ms, wait %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
dec.close()
