"""bench.py --gpus N must be self-contained: without a launcher it starts N rank processes itself,
with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, before touching the GPU; rank 0 prints the single JSON line with
n_gpus == N; under an external launcher WORLD_SIZE must equal --gpus.  Rehearsed on the CPU with --dry-run (gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def test_gpus_n_starts_n_ranks_and_prints_one_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--dry-run"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()              # rank 0's stdout carries the one JSON line and NOTHING else (gloo's banner goes to stderr)
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["open_gop"] is None
    assert out["units_all_ranks"] == 2 * 16 * 3 * 3          # both ranks' pictures: world x GOP x streams x steps
    assert out["ms_per_step"] >= 20.0 / 3 - 1e-6              # MAX over ranks: rank 1 slept 20 ms


def test_single_rank_needs_no_launcher():
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--dry-run"], env=_env(), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["n_gpus"] == 1


def test_world_size_must_match_gpus():
    e = _env()
    e.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_a_failing_rank_fails_the_launch():
    """A rank that dies must not leave the others waiting in a barrier: the launcher ends them and reports failure."""
    e = _env()
    e["DE265HIP_BENCH_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_open_gop_leg_hands_the_last_picture_to_the_next_rank():
    """--open-gop: after the timed region rank r hands a reference picture to rank r + 1; rehearsed with
    three gloo ranks on CPU planes, every receiver must hold what its sender sent."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--steps", "2", "--dry-run", "--open-gop"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.splitlines()[-1])
    assert out["n_gpus"] == 3 and out["open_gop"]["handoffs"] == 2 and out["open_gop"]["checksum_ok"] is True


def test_new_switches_and_per_rank_host_placement():
    """--chroma-format, the with_copy_out leg and the per-rank host placement: the line names the
    chroma format, carries a with_copy_out object unless --no-copy-out, and says how many host threads and CPUs every rank has;
    ranks that were pinned apart hold disjoint CPU sets."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--dry-run", "--chroma-format", "3"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.splitlines()[-1])
    cfg = out["config"]
    assert cfg["chroma_format"] == 3 and "device_replay_over_value" in cfg
    assert len(cfg["host_threads_per_rank"]) == 2 and all(t >= 1 for t in cfg["host_threads_per_rank"])
    assert out["with_copy_out"] is not None
    if len(os.sched_getaffinity(0)) >= 4:                     # enough CPUs to split between two ranks
        assert cfg["rank_cpu_sets_disjoint"] is True and all(c and c >= 2 for c in cfg["rank_cpus"])
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--dry-run", "--no-copy-out", "--no-affinity"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.splitlines()[-1])
    assert out["with_copy_out"] is None and out["config"]["chroma_format"] == 1 and out["config"]["rank_cpus"] == [None]


def test_pmc_kernel_names_are_kernels_of_the_library():
    """bench.py maps kernel ids to the names rocprofv3 prints (roofline.traffic is looked up under them): every name must be a
    kernel the built library really holds (round 2 and 3 each had one stale name here)."""
    import re
    sys.path.insert(0, ROOT)
    import bench
    from libde265_amd import backend
    blob = open(backend.SO_PATH, "rb").read()
    for kid, names in bench.PMC_KERNEL.items():
        for n in names:
            base = re.match(r"(k_[a-z0-9_]+)<", n).group(1)
            assert re.search(rb"_ZN4d265" + str(len(base)).encode() + base.encode() + rb"I", blob), (kid, n)
    from libde265_amd import _abi
    assert set(bench.PMC_KERNEL) <= set(_abi.K_NAMES)
