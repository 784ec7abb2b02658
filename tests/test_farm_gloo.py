"""N>1 path on CPU: two gloo ranks farm independent closed GOPs (no data-path
collective), time with barrier + MAX over ranks, and exercise the open-GOP
point-to-point hand-off of a finished reference picture."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import pyoracle
    import pysynth
    from libde265_amd import farm
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, BD, GOP = 128, 64, 8, 3
    units = farm.shard(list(range(4)), rank, world)                # 4 GOPs over 2 ranks
    timer = farm.RankTimer(dist)
    timer.start()
    digests, last = [], None
    for g in units:
        planes = {}
        for k, (st, refs) in enumerate(farm.gop_plan(GOP)):
            over = dict(ref_slots=refs) if refs else {}
            sp = pysynth.SynthPicture(pysynth.default_config(W, H, BD, st, seed=farm.gop_seed(4, 0, g) + k,
                                                             log2_ctb_size=5, **over))
            out = pyoracle.alloc_planes(W, H, BD)
            pyoracle.reconstruct(sp.desc, sp.order, planes, out)
            planes[k] = out
        last = planes[GOP - 1]
        digests.append((g, int(sum(int(p.astype(np.uint64).sum()) for p in last))))
    elapsed = timer.stop()
    total = farm.total_units(dist, len(units) * GOP)
    # open-GOP hand-off: rank 0 sends its last picture to rank 1, which uses it as reference
    tens = [torch.from_numpy(p.copy()) for p in last]
    farm.send_reference_picture(dist, tens, 0, 1, rank)
    recv_sum = int(sum(int(t.numpy().astype(np.uint64).sum()) for t in tens))
    sp = pysynth.SynthPicture(pysynth.default_config(W, H, BD, 1, seed=777, log2_ctb_size=5, ref_slots=[0]))
    out = pyoracle.alloc_planes(W, H, BD)
    pyoracle.reconstruct(sp.desc, sp.order, {0: [t.numpy() for t in tens]}, out)
    dep_sum = int(sum(int(p.astype(np.uint64).sum()) for p in out))
    q.put((rank, units, digests, elapsed, total, recv_sum, dep_sum))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gop_farm():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, u0, d0, e0, t0, s0, dep0), (r1, u1, d1, e1, t1, s1, dep1) = res
    assert u0 == [0, 2] and u1 == [1, 3]                         # disjoint shards, all units covered
    assert e0 == e1 > 0                                          # MAX over ranks is the same everywhere
    assert t0 == t1 == 12                                        # 4 GOPs x 3 pictures
    assert len({g for g, _ in d0 + d1}) == 4 and len({s for _, s in d0 + d1}) == 4   # different GOPs, different content
    assert s1 == s0 == d0[-1][1]                                 # rank 1 received rank 0's last picture
    assert dep0 == dep1                                          # dependent picture decodes identically on both


class _StubDecoder:
    """What exchange_reference_picture_host needs of a decoder, on numpy planes (no GPU): download, upload, _chroma_dims."""

    def __init__(self, chroma_format):
        self.cf, self.slots = chroma_format, {}

    def _chroma_dims(self, slot, w, h):
        if self.cf == 0:
            return (0, 0)
        return (w if self.cf == 3 else w // 2, h // 2 if self.cf == 1 else h)

    def download(self, slot, w, h, bd):
        return self.slots[slot]

    def upload(self, slot, planes):
        self.slots[slot] = [p.copy() for p in planes]


def _handoff_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from libde265_amd import farm
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    W, H = 64, 48
    for cf, bd in ((1, 8), (2, 10), (3, 10), (3, 8), (0, 8)):
        dec = _StubDecoder(cf)
        shapes = farm.plane_shapes(W, H, cf)
        dt = np.uint16 if bd > 8 else np.uint8
        rng = np.random.default_rng(100 * cf + bd)           # both ranks draw the same picture; only rank 0 keeps it
        planes = [rng.integers(0, 1 << bd, sh, dtype=dt) for sh in shapes]
        if rank == 0:
            dec.slots[3] = planes
        farm.exchange_reference_picture_host(dist, dec, 3, 5, 0, 1, rank, W, H, bd)
        if rank == 1:
            got = dec.slots[5]
            ok = ok and [g.shape for g in got] == [tuple(sh) for sh in shapes]
            ok = ok and all(np.array_equal(g, e) for g, e in zip(got, planes))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_handoff_of_every_chroma_format():
    """The open-GOP hand-over takes its plane geometry from the slot's chroma format (4:2:2 and 4:4:4 chroma planes are as
    high as the luma plane; a monochrome slot has none): every format arrives whole on the receiving rank."""
    import torch.multiprocessing as mp
    from libde265_amd import farm
    assert farm.plane_shapes(64, 48, 1) == [(48, 64), (24, 32), (24, 32)]
    assert farm.plane_shapes(64, 48, 2) == [(48, 64), (48, 32), (48, 32)]
    assert farm.plane_shapes(64, 48, 3) == [(48, 64), (48, 64), (48, 64)]
    assert farm.plane_shapes(64, 48, 0) == [(48, 64), (0, 0), (0, 0)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_handoff_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]
