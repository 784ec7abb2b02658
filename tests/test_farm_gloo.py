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
