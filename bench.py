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

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

PMC_FILE = os.path.join(ROOT, "profiles", "r01_o_pmc_traffic.json")
PMC_KERNEL = {"intra": ["k_run<unsigned short, 64>"], "mc": ["k_mc<unsigned short>"], "sao": ["k_sao<unsigned short>"],
              "deblock_v": ["k_deblock_fused<unsigned short>"],   # both directions in one kernel, reported under deblock_v
              "resid": ["k_resid_big<unsigned short>"]}     # all sizes in one launch
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
CONFIG_ID = 4                  # SURVEY 8d config 4 -> seed 0xDE265000 + 4

ALG_KEY = {"mc": "alg_bytes_mc", "resid": "alg_bytes_resid", "intra": "alg_bytes_intra",
           "deblock_v": "alg_bytes_deblock", "deblock_h": "alg_bytes_deblock", "sao": "alg_bytes_sao"}


def make_gop(pysynth, farm, width, height, bit_depth, gop, seed):
    """Picture k is decoded into DPB slot k; B pictures reference slots k-1 and k-2 (farm.gop_plan)."""
    pics = []
    for k, (slice_type, refs) in enumerate(farm.gop_plan(gop)):
        over = dict(ref_slots=refs, weighted_pred=1 if (k % 10) == 5 else 0) if refs else {}
        pics.append(pysynth.SynthPicture(pysynth.default_config(width, height, bit_depth, slice_type,
                                                                seed=seed + k, **over)))
    return pics


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--bit-depth", type=int, default=10)
    ap.add_argument("--gop", type=int, default=16)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DE265HIP_BENCH_STREAMS", "3")),
                    help="independent closed GOPs decoded concurrently per GPU (one decoder/HIP stream each)")
    ap.add_argument("--no-stagger", dest="stagger", action="store_false",
                    help="start all GOP streams at their I picture in lockstep instead of phase-shifted")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--events", choices=["dominant", "all"], default="dominant",
                    help="launches bracketed by hipEvents inside the timed region: only the dominant kernel's (the one "
                         "the roofline is reported for; found by a profiled pass before the timed region), or every kernel's")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL on ROCm; gloo for rehearsals)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (a 1-GPU box cannot give each rank its own GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import pysynth
    from libde265_amd import backend, farm, _abi

    if not torch.cuda.is_available() or backend.device_count() == 0:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    red_dev = "cuda" if args.backend == "nccl" else "cpu"

    W, H, BD, GOP, S = args.width, args.height, args.bit_depth, args.gop, max(1, args.streams)
    gops, decs, pics = [], [], []
    for s_i in range(S):
        g = make_gop(pysynth, farm, W, H, BD, GOP, farm.gop_seed(CONFIG_ID, rank, s_i))
        d = backend.Decoder(device=local_rank)
        for k in range(GOP):
            d.dpb_alloc(k, W, H, BD)
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

    for _ in range(args.warmup):
        step()
    sync()
    # One profiled pass ahead of the timed region, every kernel's launches bracketed by hipEvents: the per-kernel
    # breakdown (`kernels`) and the dominant kernel.  Inside the timed region only the dominant kernel is timed
    # (--events all: every kernel): each timed launch costs two event records on its stream.
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
    timer = farm.RankTimer(dist, sync, device=red_dev)
    timer.start()                       # barrier + synchronize
    t_host = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_host = time.perf_counter() - t_host   # host time spent enqueueing (the device runs behind it)
    elapsed = timer.stop()              # synchronize + barrier, MAX over ranks
    ktimes = collect()                  # the dominant kernel's launches of the timed region (all kernels' with --events all)
    for d in decs:
        d.set_profiling(False)

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

    if rank == 0:
        frames = world * args.steps * GOP * S
        fps = frames / elapsed
        # ---- roofline of the dominant kernel (device time from hipEvents on the decoder's stream)
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
            if (W, H, BD, GOP) == (3840, 2160, 10, 16) and dom in PMC_KERNEL:
                pk = json.load(open(PMC_FILE))["kernels"]
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
        # per-kernel breakdown of one step with all streams in flight: the profiled pass ahead of the timed region
        kernels = {k: {"ms_per_step": round(v[0], 4), "launches_per_step": v[1],
                       "alg_GBs": round((sum(getattr(s, ALG_KEY[k]) for s in stats) / (2 if k.startswith("deblock") and two_pass else 1)
                                         / 1e9) / (v[0] / 1e3), 1) if k in ALG_KEY and v[0] > 0 else None}
                   for k, v in ktimes_all.items()}

        st0 = [p.stats() for p in pics[0]]
        kernels_iso = {k: {"us_per_picture": round(1e3 * v[0] / max(v[1], 1), 1) if k != "resid" else
                           round(1e3 * v[0] / (2 * GOP), 1),
                           "alg_GBs": round((sum(getattr(x, ALG_KEY[k]) for x in st0) / (2 if k.startswith("deblock") and two_pass else 1)
                                             / 1e9) / (v[0] / 2 / 1e3), 1) if k in ALG_KEY and v[0] > 0 else None}
                       for k, v in iso.items() if v[1]}
        cpu = None
        parity = "not checked"
        if not args.no_cpu_baseline:
            import numpy as np
            import pyoracle
            planes = {}
            tc0 = time.perf_counter()
            for k in range(GOP):
                out = pyoracle.alloc_planes(W, H, BD)
                pyoracle.reconstruct(gop[k].desc, gop[k].order, planes, out)
                planes[k] = out
                if k >= 2:
                    planes.pop(k - 3, None)
            tc1 = time.perf_counter()
            cpu = {"value": round(GOP / (tc1 - tc0), 3), "unit": "frames/s", "cores": 1, "kind": "port",
                   "sample": "the same %d-picture GOP (1 I + %d B, %dx%d %d-bit), one pass, scalar C oracle"
                             % (GOP, GOP - 1, W, H, BD)}
            got = dec.download(GOP - 1, W, H, BD)
            parity = "bit-exact" if all(np.array_equal(g, e) for g, e in zip(got, planes[GOP - 1])) else "MISMATCH"

        line = {
            "metric": "decoded frames/sec (4K Main10)", "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u16" if BD > 8 else "u8", "data": "synthetic",
            "config": {"workload": "%dx%d %d-bit 4:2:0 random-access closed GOPs of %d pictures (1 I + %d B, 2 refs), "
                                   "%d independent GOP(s) in flight per GPU, all stages on device"
                                   % (W, H, BD, GOP, GOP - 1, S),
                       "gop": GOP, "streams_per_gpu": S, "pictures_per_step": GOP * S,
                       "host_enqueue_ms_per_step": round(1e3 * t_host / args.steps, 3),
                       "events_in_timed_region": "every kernel" if args.events == "all" else "dominant kernel (%s) only" % dom,
                       "parallelism": "%d gop stream(s) x %d gpu(s)" % (S, world)},
            "roofline": roofline, "cpu_baseline": cpu, "parity_vs_oracle": parity, "kernels": kernels,
            "kernels_isolated": kernels_iso,
        }
        print(json.dumps(line))
        if parity == "MISMATCH":
            sys.exit(2)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
