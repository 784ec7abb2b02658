#!/usr/bin/env python3
"""bench.py -- decoded frames/s of the MI355X HEVC reconstruction back end.

Workload (BASELINE.json configs[3] / SURVEY.md 8d config 4): closed GOPs of
3840x2160 10-bit 4:2:0 pictures (1 I + 15 B, each B referencing the two
previously decoded pictures), synthetic command buffers (seed 0xDE265000+4),
all stages on the device: MC -> residual -> intra -> deblock -> SAO.
One "step" = one pass over --streams independent GOPs per GPU (default 3, each
on its own decoder / HIP stream: the intra dependency chain of one GOP's I
picture is latency-bound, so independent GOPs are kept in flight to fill the
GPU); all command buffers and reference pictures are resident in HBM before
the timed region starts.  value = pictures/s over all ranks (weak scaling:
every rank decodes its own independent GOPs, no data-path collective).

`value` is the rate of the PRODUCT PATH: every picture goes build -> run -> free through the C ABI inside the timed
region (de265hip_pipeline_*: the library's worker threads run the host stage + pinned asynchronous upload of later
pictures while the device reconstructs earlier ones; thread and core counts in `config`).  `device_replay` next to it
is what the device alone sustains: the same pictures, command buffers built and uploaded before the timed region and
re-run every step.  `roofline` is about kernel device time (hipEvents on the decoder's stream inside the timed region).

    python bench.py [--gpus N] [--steps K] [--warmup W]         N > 1: starts N ranks itself (one per GPU)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import collections
import json
import os
import socket
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

# The ROCm runtime maps HIP streams onto a few hardware queues PER PRIORITY (4 by default), the least used one of a stream's
# priority; streams that share a queue run their kernels one after the other.  The library creates its streams in pools of the
# process, in an order that puts the decoders' kernel streams and its scan streams on different dispatch pipes:
# that order needs a queue of its own per stream - eight per priority leave room ("host placement").  Must be
# set before the runtime initialises.
os.environ.setdefault("DE265HIP_TUNING", "1")      # (the library reads its DE265HIP_* switches only in a process that sets this: csrc/env.h)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# a fourth scan stream on the scan pipe (libde265_amd/csrc/host.hip, DeviceStreams): sixteen hardware queues in the process - only
# where no collective library adds queues of its own (a single rank)
if int(os.environ.get("WORLD_SIZE", "1")) == 1 and "--gpus" not in " ".join(sys.argv[1:]).replace("--gpus 1", ""):
    os.environ.setdefault("DE265HIP_SCAN_STREAMS", "4")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

PMC_FILE = os.path.join(ROOT, "profiles", os.environ.get("DE265HIP_PMC_FILE", "r04_pmc_traffic.json"))
PMC_FALLBACK = os.path.join(ROOT, "profiles", "r03_i_pmc_traffic.json")
# kernel id of the C ABI (de265hip_get_kernel_times) -> the kernels rocprofv3 names under it (tests/test_bench_launcher.py checks
# every name against the kernels the library really holds)
PMC_KERNEL = {"intra": ["k_run<unsigned short, 64>"], "intra_front": ["k_intra_front<unsigned short>"],
              "mc": ["k_mc_all<unsigned short>"], "sao": ["k_sao_ctb<unsigned short>"],
              "deblock_v": ["k_deblock_fused<unsigned short>"],   # both directions in one kernel, reported under deblock_v
              "resid": ["k_resid_big<unsigned short>"]}     # all sizes in one launch
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
CONFIG_ID = 4                  # SURVEY 8d config 4 -> seed 0xDE265000 + 4

ALG_KEY = {"mc": "alg_bytes_mc", "resid": "alg_bytes_resid", "intra": "alg_bytes_intra", "intra_front": "alg_bytes_intra_front",
           "deblock_v": "alg_bytes_deblock", "deblock_h": "alg_bytes_deblock", "sao": "alg_bytes_sao"}


def make_gop(pysynth, farm, width, height, bit_depth, gop, seed, chroma_format=1):
    """Picture k is decoded into DPB slot k; B pictures reference slots k-1 and k-2 (farm.gop_plan)."""
    pics = []
    for k, (slice_type, refs) in enumerate(farm.gop_plan(gop)):
        over = dict(ref_slots=refs, weighted_pred=1 if (k % 10) == 5 else 0) if refs else {}
        if chroma_format != 1:
            over["chroma_format"] = chroma_format
        pics.append(pysynth.SynthPicture(pysynth.default_config(width, height, bit_depth, slice_type,
                                                                seed=seed + k, **over)))
    return pics


def hand_over_motion_plane(gops, yes):
    """The frame-level interface takes the per-4x4 motion plane of a picture (what the deblocking's boundary strengths read) either
    ready-made (de265hip_picture_desc::blk_motion, 6.2 MB per 4K picture, as libde265 keeps it) or not at all (NULL): the
    device then makes it from the PU records, which are handed over anyway.  The bench hands no plane over (--motion-plane: it
    does); the CPU baseline, which reads the plane, gets it back."""
    import ctypes as C
    from libde265_amd import _abi
    for g in gops:
        for sp in g:
            if not hasattr(sp, "_motion_ptr"):
                sp._motion_ptr = C.cast(sp.d.blk_motion, C.c_void_p).value
            sp.d.blk_motion = C.cast(sp._motion_ptr, C.POINTER(_abi.Motion)) if yes else None


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) BEFORE this process
    touches the GPU, rank 0's stdout is ours (the one JSON line), exit with the worst return code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:           # a rank died: the others would wait in the barrier for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


class _stdout_to_stderr:
    """C-level stdout -> stderr for the duration (gloo prints a connection banner on stdout while the process group comes up;
    rank 0's stdout carries the ONE JSON line and nothing else)."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def host_cores(world=1, pinned=False):
    """CPUs THIS RANK may keep busy: its affinity mask (its own share of it when the ranks are not pinned apart), capped by its
    share of the CPU bandwidth quota of the job's cgroup (a container with 256 visible CPUs and cpu.max = "1600000 100000" has
    16: threads beyond the quota are throttled, all of them at once - and all ranks of a node share that quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if not pinned:
        n = max(1, n // max(1, world))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]           # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period)) // max(1, world)))
    except (OSError, ValueError):
        try:                                                                        # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, (quota // period) // max(1, world)))
        except (OSError, ValueError):
            pass
    return n


def pin_rank_to_its_cores(local_rank, world):
    """Before the first GPU call of a rank process: restrict it (and every thread it starts later: the library's pipeline workers,
    the submitters) to its share of the CPUs this job may use, so that N ranks on one node do not migrate over each other's
    cores and caches.  The share comes from the NUMA node of the rank's GPU where /sys tells (the ranks whose GPUs hang off one
    node split that node's CPUs), else from an even split of the affinity mask.  Returns (cpus, numa_node or None)."""
    if not hasattr(os, "sched_setaffinity"):
        return None, None
    allowed = sorted(os.sched_getaffinity(0))
    if world <= 1 or len(allowed) < 2 * world:
        return None, None                                   # (nothing to split: the rank keeps the whole mask)
    node = None
    try:                                                    # GPU -> PCI device -> NUMA node (no HIP call: sysfs only)
        cards = sorted(d for d in os.listdir("/sys/class/drm") if d.startswith("card") and d[4:].isdigit()
                       and os.path.exists("/sys/class/drm/%s/device/numa_node" % d))
        if local_rank < len(cards):
            node = int(open("/sys/class/drm/%s/device/numa_node" % cards[local_rank]).read())
            nodes = [int(open("/sys/class/drm/%s/device/numa_node" % c).read()) for c in cards[:world]]
            if node >= 0:
                cpus = set()
                for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
                    a, _, b = part.partition("-")
                    cpus.update(range(int(a), int(b or a) + 1))
                mine = [c for c in allowed if c in cpus]
                peers = [r for r in range(world) if nodes[r] == node]
                if len(mine) >= 2 * len(peers):
                    k = peers.index(local_rank)
                    share = mine[k * len(mine) // len(peers):(k + 1) * len(mine) // len(peers)]
                    os.sched_setaffinity(0, share)
                    return share, node
            node = None
    except (OSError, ValueError, IndexError):
        node = None
    share = allowed[local_rank * len(allowed) // world:(local_rank + 1) * len(allowed) // world]
    os.sched_setaffinity(0, share)
    return share, node


def product_pass(pipes, gops, stagger, steps=1, pinned=None):
    """`steps` steps through the product path: every picture of every stream is SUBMITTED to its decoder's pipeline
    (de265hip_pipeline_submit_desc: the library's own worker threads build it - host stage + pinned asynchronous upload -,
    launch it in decode order and free it).  submit() blocks while 4 n_workers + 4 pictures of that decoder are between
    parser and device, so every stream has its own submitting thread, as every stream of a server has its own parser
    (one thread feeding all streams would stall them all behind the one whose window is full)."""
    S, GOP = len(pipes), len(gops[0])

    def feed(s_i):
        # with_copy_out: every picture leaves through de265hip_dpb_download_async into pinned host planes (a ring per stream: the
        # ticket that used a buffer before is waited for before it is handed out again - what a player's output queue does)
        ring = pinned[s_i] if pinned else None
        tickets = [None] * (len(ring) if ring else 0)
        n = 0
        for _ in range(steps):
            for j in range(GOP):
                k = (j + s_i * (GOP // S)) % GOP if stagger else j
                if ring:
                    b = n % len(ring)
                    if tickets[b] is not None:
                        pipes[s_i].wait(tickets[b])
                    tickets[b] = pipes[s_i].submit_desc(k, gops[s_i][k].desc, ring[b])
                    n += 1
                else:
                    pipes[s_i].submit_desc(k, gops[s_i][k].desc)
        for t in tickets:
            if t is not None:
                pipes[s_i].wait(t)

    if S == 1:
        feed(0)
        return
    with ThreadPoolExecutor(S) as pool:
        list(pool.map(feed, range(S)))


def cpu_baseline(gops, decs, W, H, BD, GOP, full=True, CF=1):
    """libde265's own pixel-reconstruction path on this box's host cores, next to the GPU number (a reported baseline, not
    the target).  kind "reference": the compiled reference (oracle/_ref/libde265_ref.so: libde265's decoder sources built
    with plain g++, scalar fallback DSP -- what the reference itself runs for 10-bit, x86/sse.cc:67-100 overrides 8-bit
    slots only) reconstructs the same command buffers through oracle/ref_shim.cc; kind "port": the C restatement, only
    where the compiled reference did not travel.  One GOP per host thread (the path shards by GOP; the reference's own
    WPP/tile threads live in its CABAC front end, which is not on this path), the first GOP alone for the 1-core figure.
    Doubles as the bit-exactness check of stream 0's last picture."""
    import numpy as np
    import pyoracle
    import pyref
    kind = "reference" if pyref.available() else "port"
    cores = host_cores()

    def decode_gop(g):
        planes = {}
        for k in range(GOP):
            out = pyoracle.alloc_planes(W, H, BD, chroma_format=CF)
            if kind == "reference":
                pyref.reconstruct(g[k].desc, g[k].order, planes, out, g[k].structure())
            else:
                pyoracle.reconstruct(g[k].desc, g[k].order, planes, out)
            planes[k] = out
            if k >= 2:
                planes.pop(k - 3, None)
        return planes[GOP - 1]

    if kind == "reference":
        pyref.lib()
    t0 = time.perf_counter()
    last = decode_gop(gops[0])
    t1 = time.perf_counter()
    one = GOP / (t1 - t0)
    got = decs[0].download(GOP - 1, W, H, BD)
    parity = "bit-exact" if all(np.array_equal(g, e) for g, e in zip(got, last)) else "MISMATCH"
    if not full:                                  # N > 1: the bit-exactness check only (the baseline is reported at N = 1)
        return None, parity
    # all cores: `cores` GOPs at once, one per thread (ctypes releases the GIL), cycling through the streams' GOPs
    n = max(1, min(cores, 32))
    with ThreadPoolExecutor(n) as pool:
        t0 = time.perf_counter()
        list(pool.map(decode_gop, [gops[i % len(gops)] for i in range(n)]))
        t1 = time.perf_counter()
    allc = n * GOP / (t1 - t0)
    cpu = {"value": round(allc, 3), "unit": "frames/s", "cores": n, "kind": kind, "value_1_core": round(one, 3),
           "sample": "%d GOP(s) of the same workload (1 I + %d B, %dx%d %d-bit each), one per host thread, one pass; "
                     "1-core figure: the first GOP alone" % (n, GOP - 1, W, H, BD)}
    return cpu, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--bit-depth", type=int, default=10)
    ap.add_argument("--chroma-format", type=int, default=1, choices=[1, 2, 3],
                    help="chroma_format_idc of the workload: 1 = 4:2:0 (the headline), 2 = 4:2:2, 3 = 4:4:4 (range extensions, SURVEY 8 f4)")
    ap.add_argument("--motion-plane", action="store_true",
                    help="hand the flattened per-4x4 motion plane over with every picture (default: the device makes it from the PU records)")
    ap.add_argument("--copy-out-ring", type=int, default=int(os.environ.get("DE265HIP_BENCH_RING", "8")),
                    help="with_copy_out: pinned output pictures per stream (an application's output queue; a picture's buffer is "
                         "reused once its ticket has been waited for).  4 holds the pipeline to four pictures in flight per decoder")
    ap.add_argument("--no-copy-out", action="store_true",
                    help="skip the with_copy_out leg (product path with every picture copied out to pinned host memory)")
    ap.add_argument("--no-affinity", action="store_true", help="N > 1: do not pin the rank to its share of the host's CPUs")
    ap.add_argument("--gop", type=int, default=16)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DE265HIP_BENCH_STREAMS", "3")),
                    help="independent closed GOPs decoded concurrently per GPU (one decoder/HIP stream each)")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("DE265HIP_BENCH_LANES", "1")),
                    help="picture-level concurrency inside each decoder (de265hip_decoder_set_lanes): independent pictures of ONE "
                         "stream on up to 4 HIP streams (the closed GOPs of a stream overlap at their boundaries)")
    ap.add_argument("--no-stagger", dest="stagger", action="store_false",
                    help="start all GOP streams at their I picture in lockstep instead of phase-shifted")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--events", choices=["dominant", "all"], default="dominant",
                    help="launches bracketed by hipEvents inside the timed region: only the dominant kernel's (the one "
                         "the roofline is reported for; found by a profiled pass before the timed region), or every kernel's")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL on ROCm; gloo for rehearsals)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (a 1-GPU box cannot give each rank its own GPU)")
    ap.add_argument("--host-threads", type=int, default=0,
                    help="host threads building pictures in the product-path region, over all decoders' pipelines "
                         "(0: the cores this process may use divided by the ranks, at most 16)")
    ap.add_argument("--no-host-inclusive", action="store_true",
                    help="profiling runs: skip the product-path region; value is then the device replay rate (labelled so)")
    ap.add_argument("--open-gop", action="store_true",
                    help="N > 1: also time the hand-off of the last picture of a GOP from rank r to rank r + 1 (open_gop object)")
    ap.add_argument("--no-e2e", action="store_true", help=argparse.SUPPRESS)   # accepted and ignored (older command lines): the
    # real-bitstream end-to-end measurement drives the product through the PATCHED REFERENCE DECODER, which is test infrastructure
    # (oracle/f1_recorder.cc) - it lives in tools/exp/e2e_stream_bench.sh, not in the bench (profiles/r02_e2e_*.txt)
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher rehearsal without a GPU (tests/test_bench_launcher.py): ranks rendezvous over gloo, time an empty "
                         "region with the barrier + MAX-over-ranks timer, rank 0 prints the JSON line with value null")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    # host placement BEFORE anything touches the GPU (and before the threads of torch / the library exist): this rank's cores
    cpus, numa = (None, None) if args.no_affinity else pin_rank_to_its_cores(rank if args.single_device else local_rank, world)
    if args.dry_run:
        if os.environ.get("DE265HIP_BENCH_FAIL_RANK") == str(rank):     # (launcher test: a rank that dies early)
            raise SystemExit(3)
        import torch.distributed as dist
        from libde265_amd import farm
        if world > 1:
            with _stdout_to_stderr():
                dist.init_process_group("gloo")
                dist.barrier()
        timer = farm.RankTimer(dist if world > 1 else None)
        timer.start()
        time.sleep(0.01 * (rank + 1))
        elapsed = timer.stop()
        units = farm.total_units(dist if world > 1 else None, args.gop * max(1, args.streams) * args.steps)
        open_gop = None
        if args.open_gop and world > 1:
            # the open-GOP leg without a GPU: the same chain of hand-offs r -> r + 1 (farm.send_reference_picture) on CPU planes
            import zlib
            import torch
            g = torch.Generator().manual_seed(1234 + rank)
            mine = [torch.randint(0, 255, (n,), dtype=torch.uint8, generator=g) for n in (4096, 1024, 1024)]
            got = [torch.zeros_like(t) for t in mine]
            tm = farm.RankTimer(dist)
            tm.start()
            for r in range(world - 1):
                farm.send_reference_picture(dist, mine if rank == r else got, r, r + 1, rank)
            t_x = tm.stop()

            def crc(planes):
                v = 0
                for t in planes:
                    v = zlib.crc32(t.numpy().tobytes(), v)
                return v
            me = torch.tensor([crc(mine), crc(got)], dtype=torch.int64)
            allv = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(allv, me)
            open_gop = {"handoffs": world - 1, "ms_per_handoff": round(1e3 * t_x / (world - 1), 3),
                        "checksum_ok": all(int(allv[r + 1][1]) == int(allv[r][0]) for r in range(world - 1))}
        # every rank's share of the host, gathered: a SCALE record can then be read as host-bound or not
        cores = host_cores(world, pinned=cpus is not None)
        nthr = args.host_threads if args.host_threads > 0 else max(1, min(16, cores))
        shares = [(len(cpus) if cpus else None, nthr)]
        if world > 1:
            import torch
            t_ = torch.tensor([len(cpus) if cpus else -1, nthr], dtype=torch.int64)
            allv = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(allv, t_)
            shares = [(int(v[0]) if int(v[0]) >= 0 else None, int(v[1])) for v in allv]
        disjoint = None
        if world > 1 and cpus:
            import torch
            mask = torch.zeros(4096, dtype=torch.int64)
            mask[torch.tensor(cpus)] = 1
            dist.all_reduce(mask)
            disjoint = bool(int(mask.max()) <= 1)
        if rank == 0:
            print(json.dumps({"metric": "decoded frames/sec (4K Main10)", "value": None, "unit": "frames/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "dry_run": True, "units_all_ranks": units,
                              "config": {"chroma_format": args.chroma_format, "host_threads_per_rank": [s_[1] for s_ in shares],
                                         "rank_cpus": [s_[0] for s_ in shares], "rank_cpu_sets_disjoint": disjoint,
                                         "device_replay_over_value": None},
                              "with_copy_out": None if args.no_copy_out else {"value": None},
                              "open_gop": open_gop, "ms_per_step": round(1e3 * elapsed / max(1, args.steps), 3)}))
        if world > 1:
            dist.destroy_process_group()
        return

    import torch
    import pysynth
    from libde265_amd import backend, farm, _abi

    if not torch.cuda.is_available() or backend.device_count() == 0:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    # DE265HIP_BENCH_FORCE_RCCL=1: a communicator also for a single rank (rehearsal on the one-GPU box of what a rank of an N-GPU
    # run has in its process: RCCL's streams and hardware queues next to the library's, the timer's all-reduce on the device)
    force_dist = world == 1 and os.environ.get("DE265HIP_BENCH_FORCE_RCCL") == "1"
    if force_dist:
        for k_, v_ in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
            os.environ.setdefault(k_, v_)
    if world > 1 or force_dist:
        import torch.distributed as dist
        with _stdout_to_stderr():
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(args.backend)
                dist.barrier()              # (gloo connects lazily: its banner comes with the first collective)
    red_dev = "cuda" if args.backend == "nccl" else "cpu"

    W, H, BD, GOP, S, CF = args.width, args.height, args.bit_depth, args.gop, max(1, args.streams), args.chroma_format
    if args.open_gop and GOP >= backend._abi.MAX_DPB_SLOTS:
        raise SystemExit("bench.py: --open-gop needs a free DPB slot beside the GOP's %d (at most %d slots)" % (GOP, backend._abi.MAX_DPB_SLOTS))
    gops, decs, pics = [], [], []
    for s_i in range(S):
        g = make_gop(pysynth, farm, W, H, BD, GOP, farm.gop_seed(CONFIG_ID, rank, s_i), CF)
        d = backend.Decoder(device=local_rank)
        if args.lanes > 1:
            d.set_lanes(args.lanes)
        for k in range(GOP):
            d.dpb_alloc(k, W, H, BD, chroma_format=CF)
        hand_over_motion_plane([g], args.motion_plane)
        gops.append(g); decs.append(d)
        pics.append([d.build(k, g[k].desc) for k in range(GOP)])  # inputs now resident in HBM
    gop, dec = gops[0], decs[0]
    stats = [p.stats() for ps in pics for p in ps]

    def step():
        # Every step decodes every picture of every GOP exactly once, in GOP order per stream.  The streams
        # are phase-shifted by GOP/S pictures (stream s starts its pass at picture s*GOP/S and wraps; its DPB
        # holds the identical pictures of the previous pass), the way unrelated video streams are in a server:
        # one stream's latency-bound I picture then overlaps the other streams' B pictures.
        for j in range(GOP):
            for s_i in range(S):
                k = (j + s_i * (GOP // S)) % GOP if args.stagger else j
                decs[s_i].run(pics[s_i][k], _abi.STAGE_FINAL)

    def sync():
        for d in decs:
            d.sync()
        torch.cuda.synchronize()

    # ---- (1) device replay: what the device alone sustains on prebuilt pictures
    for _ in range(args.warmup):
        step()
    sync()
    # One profiled pass, every kernel's launches bracketed by hipEvents: the per-kernel breakdown (`kernels`) and the
    # dominant kernel.  Inside the timed regions only the dominant kernel is timed (--events all: every kernel): each
    # timed launch costs two event records on its stream.
    def collect():
        kt = {}
        for d in decs:
            for kname, (ms, n) in d.kernel_times(reset=True).items():
                a = kt.get(kname, (0.0, 0))
                kt[kname] = (a[0] + ms, a[1] + n)
        return kt

    for d in decs:
        d.set_profiling(True)
        d.kernel_times(reset=True)
    step()
    sync()
    ktimes_all = collect()
    dom = max(ktimes_all, key=lambda k: ktimes_all[k][0])
    for d in decs:
        d.set_profiling(True, only=None if args.events == "all" else [dom])
    # (experiment, DE265HIP_BENCH_NOISE=n: n host threads launch tiny torch kernels on streams of their own during the replay -
    #  how much do kernel boundaries of OTHER streams, with their cache write-backs and invalidates, cost the reconstruction?)
    noise_n, noise_stop, noise_threads, noise_count = int(os.environ.get("DE265HIP_BENCH_NOISE", "0")), [False], [], [0]
    if noise_n:
        import threading

        def noise():
            st = torch.cuda.Stream()
            x = torch.zeros(int(os.environ.get("DE265HIP_BENCH_NOISE_ELEMS", "64")), device="cuda")
            with torch.cuda.stream(st):
                while not noise_stop[0]:
                    for _ in range(64):
                        x.add_(1.0)
                    noise_count[0] += 64
                    st.synchronize()
        noise_threads = [threading.Thread(target=noise) for _ in range(noise_n)]
        for t_ in noise_threads:
            t_.start()
    timer = farm.RankTimer(dist, sync, device=red_dev)
    timer.start()                       # barrier + synchronize
    t_host = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_host = time.perf_counter() - t_host   # host time spent enqueueing (the device runs behind it)
    replay_elapsed = timer.stop()       # synchronize + barrier, MAX over ranks
    if noise_n:
        noise_stop[0] = True
        for t_ in noise_threads:
            t_.join()
        sys.stderr.write("noise: %d tiny launches during the replay (%.0f per second)\n" % (noise_count[0], noise_count[0] / replay_elapsed))
    ktimes_replay = collect()           # the dominant kernel's launches of that region (all kernels' with --events all)
    for d in decs:
        d.set_profiling(False)
    device_replay = {"value": round(world * args.steps * GOP * S / replay_elapsed, 2), "unit": "frames/s", "steps": args.steps,
                     "ms_per_step": round(1e3 * replay_elapsed / args.steps, 3),
                     "host_enqueue_ms_per_step": round(1e3 * t_host / args.steps, 3),
                     "what": "device replay of prebuilt pictures: command buffers built and uploaded before this region, every "
                             "picture's kernels re-enqueued each step (not the product's rate: no host stage, no upload)"}

    # ---- (2) the product path, THE TIMED REGION of this bench: every picture built, launched and freed through the C ABI
    elapsed, ktimes, product = replay_elapsed, ktimes_replay, None
    cores = host_cores(world, pinned=cpus is not None)   # this rank's share
    if not args.no_host_inclusive:
        # threads of this rank: per GOP stream one submitter (this script's: the "parser"), one launcher (the library's: every HIP
        # call of the decoder) and `per` workers (the library's: the host stage of the builds); --host-threads counts the workers
        nthr = args.host_threads if args.host_threads > 0 else max(S, min(9, cores - 2 * S - 1))
        per = max(1, min(16, nthr // S))                 # worker threads of each decoder's pipeline
        for ps in pics:                                  # the prebuilt pictures go back to the pools
            for p_ in ps:
                p_.free()
        sync()
        pipes = [backend.Pipeline(d, per) for d in decs]

        def drain():
            for pp_ in pipes:
                pp_.drain()

        for _ in range(max(1, args.warmup)):             # fills the arena / staging pools and the workers' scratch
            product_pass(pipes, gops, args.stagger)
        drain()
        for d in decs:
            d.set_profiling(True, only=None if args.events == "all" else [dom])
            d.kernel_times(reset=True)
        timer = farm.RankTimer(dist, sync, device=red_dev)
        timer.start()
        product_pass(pipes, gops, args.stagger, steps=args.steps)
        drain()                                          # every picture launched, finished, freed
        elapsed = timer.stop()
        ktimes = collect()
        for d in decs:
            d.set_profiling(False)
        # ---- (2b) the same product path with every decoded picture DELIVERED: copied out through de265hip_dpb_download_async
        # into pinned host planes inside the timed region (what de265_get_image_plane hands an application), PCIe included
        copy_out = None
        if not args.no_copy_out:
            ring_n = max(1, args.copy_out_ring)
            rings = [[backend.PinnedPlanes(W, H, BD, chroma_format=CF) for _ in range(ring_n)] for _ in range(S)]
            product_pass(pipes, gops, args.stagger, pinned=rings)
            drain()
            tm = farm.RankTimer(dist, sync, device=red_dev)
            tm.start()
            product_pass(pipes, gops, args.stagger, steps=args.steps, pinned=rings)
            drain()
            t_co = tm.stop()
            pic_bytes = sum(p_.nbytes for p_ in rings[0][0].planes)
            copy_out = {"value": round(world * args.steps * GOP * S / t_co, 2), "unit": "frames/s",
                        "ms_per_step": round(1e3 * t_co / args.steps, 3), "bytes_per_picture": int(pic_bytes),
                        "d2h_GBs": round(world * args.steps * GOP * S * pic_bytes / 1e9 / t_co, 2),
                        "what": "product path + de265hip_dpb_download_async of every picture into pinned host planes (ring of %d per "
                                "stream, the ticket that held a buffer is waited for before its reuse), inside the timed region" % ring_n}
            # the last picture each stream delivered must be the one the device holds
            import numpy as _np
            ok = True
            for s_i in range(S):
                k_last = ((GOP - 1) + s_i * (GOP // S)) % GOP if args.stagger else GOP - 1
                got = decs[s_i].download(k_last, W, H, BD)
                b = (args.steps * GOP - 1) % ring_n
                ok = ok and all(_np.array_equal(g_, e_) for g_, e_ in zip(got, rings[s_i][b].planes))
            copy_out["delivered_equals_dpb"] = bool(ok)
            for r_ in rings:
                for pp2 in r_:
                    pp2.free()
        for pp_ in pipes:
            pp_.close()
        # one host thread: stream 0 alone through a one-worker pipeline
        one = backend.Pipeline(decs[0], 1)
        product_pass([one], gops[:1], False); one.drain()
        t1 = time.perf_counter()
        product_pass([one], gops[:1], False); one.drain()
        t1 = time.perf_counter() - t1
        one.close()
        product = {"host_threads": per * S + S, "workers_per_decoder": per, "launcher_threads": S, "host_cores_available": cores,
                   "host_threads_per_rank": per * S + S, "rank_cpus": len(cpus) if cpus else None, "rank_numa_node": numa,
                   "value_1_host_thread": round(GOP / t1, 2), "with_copy_out": copy_out,
                   "what": "de265hip_pipeline_submit_desc per picture: the library's worker threads run the host stage of the build (validation, "
                           "MC tasks, staging, pinned upload), its launcher thread enqueues the device-side scan of the TU records, launches in "
                           "decode order (de265hip_picture_run) and frees; decoded pictures stay in the device-resident DPB (no copy-out in "
                           "the timed region)"}
        pics = [[d.build(k, g[k].desc) for k in range(GOP)] for d, g in zip(decs, gops)]   # (for the isolated pass below)
        stats = [p.stats() for ps in pics for p in ps]

    # ---- (3) open-GOP hand-off, timed separately from `value`: the last picture of a GOP of rank r
    # goes to rank r + 1 (DPB slot GOP, beside the receiver's own GOP).  RCCL point-to-point on the DPB planes with
    # --backend nccl; a gloo rehearsal stages through the host (gloo sends CPU tensors only).
    open_gop = None
    if args.open_gop and world > 1:
        slot_in = GOP                                     # (a free slot beside the GOP's: checked at start-up)
        decs[0].dpb_alloc(slot_in, W, H, BD, chroma_format=CF)
        sync()
        tm = farm.RankTimer(dist, sync, device=red_dev)
        tm.start()
        for r in range(world - 1):
            if args.backend == "nccl":
                farm.exchange_reference_picture_dpb(dist, decs[0], GOP - 1, slot_in, r, r + 1, rank)
            else:
                farm.exchange_reference_picture_host(dist, decs[0], GOP - 1, slot_in, r, r + 1, rank, W, H, BD)
        t_x = tm.stop()
        nbytes = sum(r_ * c_ for r_, c_ in farm.plane_shapes(W, H, CF)) * (2 if BD > 8 else 1)
        open_gop = {"handoffs": world - 1, "ms_per_handoff": round(1e3 * t_x / (world - 1), 3), "bytes_per_picture": nbytes,
                    "GBs": round(nbytes / 1e9 / (t_x / (world - 1)), 2), "transport": "rccl p2p on DPB planes" if args.backend == "nccl" else "gloo via host (rehearsal)",
                    "checksum_ok": farm.check_handoff(dist, decs[0], GOP - 1, slot_in, rank, world, W, H, BD, device=red_dev)}

    # one more, untimed, pass of stream 0 alone: per-kernel device times without the other streams'
    # kernels competing for the GPU (reported as kernels_isolated; value/roofline come from the timed region)
    iso = {}
    if rank == 0:
        decs[0].set_profiling(True)
        decs[0].kernel_times(reset=True)
        for _ in range(2):
            for p in pics[0]:
                decs[0].run(p, _abi.STAGE_FINAL)
        iso = decs[0].kernel_times(reset=True)
        decs[0].set_profiling(False)
    sync()
    if dist is not None:
        dist.barrier()                  # every rank has finished its GPU work before rank 0 loads the host cores with the CPU baseline

    if rank == 0:
        frames = world * args.steps * GOP * S
        fps = frames / elapsed
        # ---- roofline of the dominant kernel (device time from hipEvents on the decoder's stream, inside the timed region)
        dom_ms, dom_launches = ktimes[dom]
        alg_total = sum(getattr(s, ALG_KEY[dom]) for s in stats) * args.steps if dom in ALG_KEY else 0
        two_pass = ktimes_all.get("deblock_h", (0, 0))[1] > 0      # DE265HIP_TWO_PASS_DEBLOCK: SURVEY 8d's 2P is for both passes together
        if dom in ("deblock_v", "deblock_h") and two_pass:
            alg_total //= 2
        achieved = (alg_total / 1e9) / (dom_ms / 1e3) if dom_ms > 0 else 0.0
        # HBM-side traffic of that kernel: rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) committed under
        # profiles/ (bench.py cannot run the profiler on itself); FETCH_SIZE doubled as the guide prescribes
        # for gfx950 (MI355X_MICROARCH.md, HBM).  Only valid for the default 4K 10-bit workload.
        traffic = None
        try:
            if (W, H, BD, GOP, CF) == (3840, 2160, 10, 16, 1) and dom in PMC_KERNEL:
                pk = json.load(open(PMC_FILE if os.path.exists(PMC_FILE) else PMC_FALLBACK))["kernels"]
                traffic = int(sum((2 * pk[k]["fetch_kb_per_launch"] + pk[k]["write_kb_per_launch"]) * 1024
                                  for k in PMC_KERNEL[dom]))
        except Exception:
            traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "launches": int(dom_launches), "avg_launch_us": round(1e3 * dom_ms / max(dom_launches, 1), 3),
                    "alg_bytes_per_launch": int(alg_total / max(dom_launches, 1)),
                    "note": ("intra (k_run) is bound by the z-scan dependency chain (single-wavefront latency), not by HBM; "
                             "the streaming kernels' algorithmic GB/s are under kernels / kernels_isolated")
                    if dom == "intra" else ""}
        # per-kernel breakdown of one step with all streams in flight: the profiled device-replay pass
        kernels = {k: {"ms_per_step": round(v[0], 4), "launches_per_step": v[1],
                       "alg_GBs": round((sum(getattr(s, ALG_KEY[k]) for s in stats) / (2 if k.startswith("deblock") and two_pass else 1)
                                         / 1e9) / (v[0] / 1e3), 1) if k in ALG_KEY and v[0] > 0 else None}
                   for k, v in ktimes_all.items()}

        # all kernels together: algorithmic bytes of a step (every stage of every picture) over the step's wall time,
        # in the timed region and in the device replay
        alg_step = sum(getattr(s_, a) for s_ in stats for a in ("alg_bytes_mc", "alg_bytes_resid", "alg_bytes_intra", "alg_bytes_intra_front", "alg_bytes_deblock", "alg_bytes_sao"))

        def agg(el):
            gbs = (alg_step / 1e9) / (el / args.steps)
            return {"achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
        aggregate = dict(agg(elapsed), alg_bytes_per_step=int(alg_step), unit="GB/s", peak=HBM_PEAK_GBS, device_replay=agg(replay_elapsed),
                         note="sum of every stage's algorithmic bytes of one step / wall time of the step (all streams, all kernels)")
        st0 = [p.stats() for p in pics[0]]
        kernels_iso = {k: {"us_per_picture": round(1e3 * v[0] / max(v[1], 1), 1) if k != "resid" else
                           round(1e3 * v[0] / (2 * GOP), 1),
                           "alg_bytes_per_picture": int(sum(getattr(x, ALG_KEY[k]) for x in st0) / (2 if k.startswith("deblock") and two_pass else 1)
                                                        / GOP) if k in ALG_KEY else None,
                           "alg_GBs": round((sum(getattr(x, ALG_KEY[k]) for x in st0) / (2 if k.startswith("deblock") and two_pass else 1)
                                             / 1e9) / (v[0] / 2 / 1e3), 1) if k in ALG_KEY and v[0] > 0 else None}
                       for k, v in iso.items() if v[1]}
        cpu = None
        parity = "not checked"
        if not args.no_cpu_baseline:
            hand_over_motion_plane(gops, True)                # (the reference reads the plane)
            cpu, parity = cpu_baseline(gops, decs, W, H, BD, GOP, full=(world == 1), CF=CF)
        if product is not None:
            region = ("product path: every picture of the step goes build -> run -> free through the C ABI inside the timed region "
                      "(%d host threads in the library's pipelines); device_replay is the device alone" % product["host_threads"])
        else:
            region = "--no-host-inclusive: device replay of prebuilt pictures only (NOT the product's rate; profiling runs)"
        line = {
            "metric": "decoded frames/sec (4K Main10)", "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u16" if BD > 8 else "u8", "data": "synthetic",
            "config": {"workload": "%dx%d %d-bit %s random-access closed GOPs of %d pictures (1 I + %d B, 2 refs), "
                                   "%d independent GOP(s) in flight per GPU, all stages on device"
                                   % (W, H, BD, {1: "4:2:0", 2: "4:2:2", 3: "4:4:4"}[CF], GOP, GOP - 1, S),
                       "timed_region": region,
                       "motion_plane": "handed over (6.2 MB per 4K picture)" if args.motion_plane else "made on the device from the PU records (blk_motion = NULL)",
                       "gop": GOP, "streams_per_gpu": S, "lanes_per_decoder": args.lanes, "pictures_per_step": GOP * S,
                       "host_threads": product["host_threads"] if product else 0, "host_cores_available": cores,
                       "host_threads_per_rank": product["host_threads_per_rank"] if product else 0,
                       "device_replay_over_value": round(device_replay["value"] / fps, 2) if fps > 0 else None,
                       "bound": ("not the reconstruction kernels (replayed alone they sustain %.1fx this rate): the build of a picture - host "
                                 "stage on the worker threads, then the device-side scan of its records, which shares the device with the "
                                 "reconstruction" % (device_replay["value"] / fps)
                                 if device_replay["value"] > 1.15 * fps else "device") if product else "device replay",
                       "events_in_timed_region": "every kernel" if args.events == "all" else "dominant kernel (%s) only" % dom,
                       "parallelism": "%d gop stream(s) x %d gpu(s)" % (S, world)},
            "roofline": roofline, "roofline_aggregate": aggregate, "cpu_baseline": cpu, "device_replay": device_replay,
            "product_path": product, "with_copy_out": product["with_copy_out"] if product else None, "open_gop": open_gop, "parity_vs_reference": parity, "kernels": kernels,
            "kernels_isolated": kernels_iso,
        }
        print(json.dumps(line))
        sys.stdout.flush()
        if parity == "MISMATCH":
            if dist is not None:
                dist.barrier()
                dist.destroy_process_group()
            sys.exit(2)
    if dist is not None:
        dist.barrier()                  # the other ranks stay until rank 0 is through with the CPU baseline
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
