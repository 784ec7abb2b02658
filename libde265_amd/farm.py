"""Frame-parallel farming of independent closed GOPs over the GPUs of one node
(SURVEY.md 8e): one process per GPU, every rank decodes its own GOPs, no
collective on the data path.  torch.distributed is used only for the timing
barrier and the max-over-ranks reduction (and, optionally, to hand a finished
reference picture to another rank -- the open-GOP case).
"""
import time

BASE_SEED = 0xDE265000


def gop_seed(config_id, rank, gop_index=0):
    """Seeds of SURVEY 8d (0xDE265000 + config id), spread so that ranks and GOPs never collide."""
    return BASE_SEED + config_id + 1000 * rank + 100000 * gop_index


def gop_plan(gop_len):
    """Closed GOP: picture k -> DPB slot k; k=0 is intra, k>0 is a B picture that
    references the two previously decoded pictures.  Returns [(slice_type, [ref slots])]."""
    plan = [(2, [])]
    for k in range(1, gop_len):
        plan.append((0, [k - 1, max(k - 2, 0)]))
    return plan


def shard(units, rank, world):
    """Units (GOPs / independent pictures) owned by `rank`: round-robin, no exchange needed."""
    return [u for i, u in enumerate(units) if i % world == rank]


class RankTimer:
    """barrier + sync on both sides of the timed region, MAX over ranks (bench.py contract).
    `dist` is torch.distributed or None; `sync` synchronises the local device (or is a no-op on CPU)."""

    def __init__(self, dist=None, sync=lambda: None, device="cpu"):
        self.dist, self.sync, self.device = dist, sync, device
        self.t0 = None

    def _barrier(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.barrier()

    def start(self):
        self.sync()
        self._barrier()
        self.t0 = time.perf_counter()

    def stop(self):
        self.sync()
        self._barrier()
        elapsed = time.perf_counter() - self.t0
        if self.dist is not None and self.dist.is_initialized():
            import torch
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed


def total_units(dist, local_units, device="cpu"):
    """Sum over ranks of the units each rank processed."""
    if dist is None or not dist.is_initialized():
        return local_units
    import torch
    t = torch.tensor([local_units], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def send_reference_picture(dist, planes, src, dst, rank, device="cpu"):
    """Open-GOP exchange step: rank `src` hands a finished reference picture
    (list of 3 torch tensors) to rank `dst` point-to-point; other ranks do nothing."""
    if rank == src:
        for p in planes:
            dist.send(p, dst=dst)
    elif rank == dst:
        for p in planes:
            dist.recv(p, src=src)
    return planes


# ---- the same exchange on DEVICE-RESIDENT DPB planes (the product's picture store) -----------------------------
class _DevPlane:
    """__cuda_array_interface__ holder: lets torch view a DPB plane's device memory without copying."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def plane_rows(dec, slot):
    """Rows of the three planes of a DPB slot: the chroma planes follow the slot's chroma_format_idc
    (de265hip_dpb_chroma_format; SubHeightC is 2 for 4:2:0 only, a monochrome slot has empty chroma planes)."""
    w, h, _, _ = dec.dpb_info(slot)
    _, ch = dec._chroma_dims(slot, w, h)
    return [h, ch, ch]


def dpb_plane_tensors(dec, slot):
    """Zero-copy torch uint8 views (pitch x rows, padding included) of the planes of a decoder's DPB slot
    (de265hip_dpb_plane); empty planes (monochrome chroma) are left out.  The caller orders access: dec.sync() before
    another stream reads them, and torch.cuda.synchronize() (or an event) before the decoder's kernels read what torch wrote."""
    import torch          # (import torch before the first libde265_hip.so call in such a process: tests/conftest.py)
    views = []
    for c, rows in enumerate(plane_rows(dec, slot)):
        ptr, stride_bytes = dec.plane(slot, c)
        if rows:
            views.append(torch.as_tensor(_DevPlane(ptr, stride_bytes * rows), device="cuda"))
    return views


def dpb_slot_tensor(dec, slot):
    """The whole slot as ONE zero-copy uint8 view: a slot is one device allocation, luma first, so a picture
    crosses xGMI in one message instead of three.  Two slots of the same geometry (size, bit depths, chroma format) have the
    same layout, whichever decoder or rank owns them."""
    import torch
    rows = plane_rows(dec, slot)
    p0, _ = dec.plane(slot, 0)
    last = max(c for c in range(3) if rows[c])
    pl, sl = dec.plane(slot, last)
    return torch.as_tensor(_DevPlane(p0, pl + sl * rows[last] - p0), device="cuda")


def send_reference_picture_dpb(dist, dec, slot, src, dst, rank):
    """Open-GOP exchange between ranks on the planes where they live: rank `src` sends DPB slot `slot` of its decoder
    to rank `dst` (RCCL point-to-point over xGMI, one message per picture: 24.9 MB at 4K Main10 4:2:0), which receives into
    the same slot of its own decoder (already dpb_alloc'ed to the same geometry, chroma format included).  No host staging,
    no collective for the other ranks."""
    import torch
    if rank not in (src, dst):
        return
    # both sides: the sender's picture is finished before it leaves; on the receiver, everything its decoder has already
    # enqueued (pictures that still read the slot's old content as a reference, a copy-out of it) is done before RCCL,
    # which runs on torch's stream, overwrites the slot
    dec.sync()
    t = dpb_slot_tensor(dec, slot)
    if rank == src:
        dist.send(t, dst=dst)
    else:
        dist.recv(t, src=src)
    torch.cuda.synchronize()                         # received planes are in place before the decoder's stream reads them


def broadcast_reference_picture_dpb(dist, dec, slot, src, rank, group=None):
    """The same for several consumers: RCCL broadcast of the three planes from rank `src` (per-link bound on xGMI:
    prefer send_reference_picture_dpb to the actual consumers when they are few)."""
    import torch
    dec.sync()                                       # (sender: picture finished; receivers: nothing queued still uses the slot)
    dist.broadcast(dpb_slot_tensor(dec, slot), src=src, group=group)
    torch.cuda.synchronize()


def exchange_reference_picture_dpb(dist, dec, src_slot, dst_slot, src, dst, rank):
    """send_reference_picture_dpb with different slots on the two sides: rank `src` sends its DPB slot `src_slot`, rank `dst`
    receives into its slot `dst_slot` (already dpb_alloc'ed to the same geometry).  RCCL point-to-point on the planes where they live."""
    import torch
    if rank not in (src, dst):
        return
    dec.sync()                                       # sender: the picture is finished; receiver: nothing queued still uses the slot
    t = dpb_slot_tensor(dec, src_slot if rank == src else dst_slot)
    if rank == src:
        dist.send(t, dst=dst)
    else:
        dist.recv(t, src=src)
    torch.cuda.synchronize()


def plane_shapes(width, height, chroma_format=1):
    """(rows, columns) of the three planes of a picture: SubWidthC / SubHeightC by chroma_format_idc (sps.cc:540-552)."""
    if chroma_format == 0:
        return [(height, width), (0, 0), (0, 0)]
    cw = width if chroma_format == 3 else width // 2
    ch = height // 2 if chroma_format == 1 else height
    return [(height, width), (ch, cw), (ch, cw)]


def exchange_reference_picture_host(dist, dec, src_slot, dst_slot, src, dst, rank, width, height, bit_depth):
    """The same hand-off staged through host memory (gloo sends CPU tensors only): a REHEARSAL of the open-GOP step on boxes
    without RCCL peers, not a transport anybody would deploy."""
    import numpy as np
    import torch
    if rank == src:
        for p in dec.download(src_slot, width, height, bit_depth):
            if p.size:                                   # (monochrome: the empty chroma planes are not sent)
                dist.send(torch.from_numpy(np.ascontiguousarray(p).view(np.uint8).reshape(-1)), dst=dst)
    elif rank == dst:
        dt = np.uint16 if bit_depth > 8 else np.uint8
        planes = []
        cw, ch = dec._chroma_dims(dst_slot, width, height)      # (the receiving slot was dpb_alloc'ed with the sender's chroma format)
        for (h, w) in ((height, width), (ch, cw), (ch, cw)):
            buf = torch.empty(h * w * np.dtype(dt).itemsize, dtype=torch.uint8)
            if h * w:
                dist.recv(buf, src=src)
            planes.append(buf.numpy().view(dt).reshape(h, w))
        dec.upload(dst_slot, planes)


def check_handoff(dist, dec, src_slot, dst_slot, rank, world, width, height, bit_depth, device="cpu"):
    """After the chain of hand-offs r -> r + 1: what rank r + 1 holds in `dst_slot` must be what rank r holds in `src_slot`
    (CRC32 of the three planes, gathered over all ranks)."""
    import zlib
    import torch

    def crc(slot):
        v = 0
        for p in dec.download(slot, width, height, bit_depth):
            v = zlib.crc32(p.tobytes(), v)
        return v
    mine = torch.tensor([crc(src_slot), crc(dst_slot) if rank > 0 else 0], dtype=torch.int64, device=device)
    allv = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(allv, mine)
    return all(int(allv[r + 1][1]) == int(allv[r][0]) for r in range(world - 1))
