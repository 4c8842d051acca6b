"""The N > 1 path of bench.py on CPU: world_size 2 over gloo (one process per rank, rendezvous on 127.0.0.1)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import time
    import tc2li_loader
    tc2li_loader.load()
    from tc2li_slam_amd import dist_util
    dist = dist_util.init("gloo", rank, world)
    # every rank owns its own sequences (no overlap, nothing lost)
    mine = dist_util.shard_units(10, rank, world)
    dist_util.barrier(dist)
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))      # rank 1 is the slow one
    dist_util.barrier(dist)
    elapsed = time.perf_counter() - t0 - (0.0 if rank else 0.0)
    local = 0.05 * (rank + 1)
    mx = dist_util.max_elapsed(dist, local)
    out[rank] = (mine, mx, dist_util.job_throughput(32, 5, world, mx), elapsed)
    dist.destroy_process_group()


def test_two_ranks_over_gloo():
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + os.getpid() % 400
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    (m0, mx0, thr0, e0), (m1, mx1, thr1, e1) = out[0], out[1]
    assert sorted(m0 + m1) == list(range(10)) and not set(m0) & set(m1)
    assert mx0 == mx1 == pytest.approx(0.10)                 # MAX over ranks, identical everywhere
    assert thr0 == thr1 == pytest.approx(32 * 5 * 2 / 0.10)   # whole-job units/s
    assert e0 >= 0.09 and e1 >= 0.09                          # the closing barrier holds the fast rank back


def test_single_rank_needs_no_process_group():
    sys.path.insert(0, ROOT)
    import tc2li_loader
    tc2li_loader.load()
    from tc2li_slam_amd import dist_util
    assert dist_util.init("gloo", 0, 1) is None
    assert dist_util.max_elapsed(None, 1.5) == 1.5
    assert dist_util.shard_units(5, 0, 1) == [0, 1, 2, 3, 4]


def _select_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch
    import torch.distributed as dist
    import tc2li_loader
    pkg = tc2li_loader.load()
    from tc2li_slam_amd import synthetic
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = synthetic.ba_window(2, n_opt=4, n_fix=4, n_points=301)
    e = pkg.pack_ba_edges(w["edges"])
    lo, eo = pkg.capi.ba_shard_select(e, len(w["points"]), rank, world)
    # what the sharded window's collectives do with the ranks' parts, on host tensors: the ownership masks sum to all-ones
    t = torch.from_numpy(np.concatenate([lo, eo]).astype(np.float64))
    dist.all_reduce(t)
    out[rank] = (int(lo.sum()), int(eo.sum()), bool((t == 1).all()), bool(lo[rank::world].all()), bool(np.array_equal(eo, lo[e["point"]])),
                 len(w["points"]), len(e))
    dist.destroy_process_group()


def test_landmark_partition_of_a_sharded_window_over_gloo():
    """tc2li_ba_shard_select (host logic of tc2li_local_lv_bundle_adjustment_sharded): every landmark and edge has exactly one owner."""
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29950 + os.getpid() % 40
    mp.spawn(_select_worker, args=(world, port, out), nprocs=world, join=True)
    n_points, n_edges = out[0][5], out[0][6]
    assert n_points > 100 and out[0][0] + out[1][0] == n_points and out[0][0] == (n_points + 1) // 2
    assert out[0][1] + out[1][1] == n_edges
    assert out[0][2] and out[1][2] and out[0][3] and out[1][3] and out[0][4] and out[1][4]
    assert out[0][1] > 0 and out[1][1] > 0


def _bench(*argv, env=None):
    import subprocess
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, timeout=300, env=e)


def test_bench_spawns_its_ranks_world_2_gloo():
    """`python bench.py --gpus 2` outside torch.distributed.run starts two ranks itself (fresh child processes, rendezvous on 127.0.0.1);
    --rehearse runs the rank protocol without device work: the line must say 2 GPUs / 2 ranks seen, and the strong-scaling list of 11
    sequences must be dealt completely (6 + 5)."""
    import json
    r = _bench("--gpus", "2", "--rehearse", "--steps", "3", "--warmup", "1", "--scaling", "strong", "--sequences", "11")
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["scaling"] == "strong"
    assert line["config"]["sequences_total"] == 11 and line["config"]["sequences_of_rank0"] == 6
    assert line["steps"] == 3 and line["value"] > 0
    # the host side of a rank: rank 0 keeps to its half of the cores this test may run on (pinned before any thread of
    # its own exists) and the library sizes its pools from that share -- every pool at least one thread, none beyond the share
    hb = line["config"]["host_threads_gpu_path"]
    cores = len(os.sched_getaffinity(0))
    assert hb["ranks_on_node"] == 2 and hb["pinned"] and hb["cores_of_this_rank"] == max(1, cores // 2)
    assert line["config"]["affinity_of_rank0"] == hb["cores_of_this_rank"]
    # the thread budget is the rank's share of what the cgroup really grants (min(affinity, quota) / ranks), the pools at most eight threads per CPU
    assert 1 <= hb["thread_budget"] <= hb["cores_of_this_rank"] and hb["thread_budget"] <= max(1, hb["cpus_effective"] // 2)
    assert all(1 <= v <= 8 * hb["thread_budget"] for v in hb["library_pools"].values())
    assert len(r.stdout.strip().splitlines()[-1]) <= 4096  # the line the driver parses stays small


def test_host_thread_budget_of_eight_ranks_fits_the_node():
    """What apply_host_budget gives the library on the driver's 8-GPU node (256 cores): 32 cores per rank, pools that add up -- with the
    five stage threads and three lock-step BA groups -- to at most three threads per core (round 6: the per-window host steps between the phases,
    which wanted eight, are gone with the device-side LM loop; rounds 1-3: ~130 threads per rank whatever the rank count); on a node whose
    cgroup grants 16 CPUs to 8 ranks (2 each) every pool is one thread."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r)\nimport tc2li_loader\npkg = tc2li_loader.load()\npkg.capi.set_host_thread_budget(256 // 8)\n"
            "h = pkg.capi.host_threads()\nprint(5 + h['extractor_pool'] + h['tracking_pool'] + h['lidar_pool'] + 3 * h['ba_group_pool'])" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert int(r.stdout.split()[-1]) <= 3 * 32
    r = subprocess.run([sys.executable, "-c", code.replace("256 // 8", "16 // 8")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert int(r.stdout.split()[-1]) == 5 + 1 + 1 + 1 + 3


def test_the_printed_line_keeps_to_its_byte_budget():
    """the driver parses ONE line and round 4's 20 KB line was not parsed.  compact_line() cuts any report -- here one fatter
    than round 4's -- to the contract's keys + one-object summaries within LINE_BUDGET, the roofline's `traffic` as a number."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    table = {"k_kernel_%02d" % i: {"launches_per_step": 1.5, "ms_per_step": 0.123, "avg_launch_us": 12.3, "share_of_kernel_time": 0.04, "achieved_GBps": 123.4,
                                    "frac_hbm": 0.0123} for i in range(40)}
    rf = {"kernel": "k_ba_linearize_b", "bound": "hbm", "achieved": 459.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.057, "avg_launch_ms": 0.352,
          "algorithmic_bytes_per_launch": 161600000, "traffic": {"bytes_per_launch": 177700000, "source": "profiles/r05_pmc_traffic.json", "note": "x" * 300},
          "all_kernels": table, "measured_in": "y" * 900, "peaks_measured": {"note": "z" * 500}}
    cpu = {"value": 60.0, "unit": "frames/s", "cores": 16, "kind": "port", "sample": "s" * 700, "single_sequence": {"value": 24.5, "cores": 4, "sample": "t" * 200}}
    full = {"metric": bench.METRIC, "value": 19900.0, "unit": "frames/s", "n_gpus": 1, "steps": 20, "warmup": 3, "ms_per_step": 25.7, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8, f32, f64", "data": "synthetic",
            "config": {"workload": "w" * 900, "stage_threads": "v" * 900, "sequences_total": 512, "host_threads_gpu_path": {"cpus_effective": 16, "cgroup_cpu_quota": 16.0,
                       "cpu_s_per_wall_s_by_thread_name": {"t%d" % i: 0.5 for i in range(60)}}},
            "roofline": rf, "cpu_baseline": cpu, "single_sequence": {"value": 846.0, "unit": "frames/s", "ms_per_frame": 1.18, "workload": "u" * 500},
            "host_fed": {"value": 8914.0, "ms_per_step": 57.4, "GB_per_step": 2.61, "link_GBps": 45.5, "workload": "q" * 600},
            "inertial_config": {"value": 12000.0, "unit": "frames/s", "ms_per_step": 42.0, "workload": "r" * 800, "roofline": dict(rf), "cpu_baseline": dict(cpu),
                                "single_sequence": {"value": 460.5, "ms_per_frame": 2.17}},
            "mfma_config": {"workload": "m" * 400, "windows_per_s": 2000.0, "roofline": dict(rf, bound="mfma", unit="TFLOP/s")},
            "sequences_per_gpu_sweep": {"512": 19576.0, "256": 17800.0, "128": 16300.0, "64": 14521.0, "unit": "n" * 400}}
    assert len(json.dumps(full)) > 15000
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    txt = json.dumps(line)
    assert len(txt) <= bench.LINE_BUDGET, len(txt)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line, k
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"]) and line["roofline"]["traffic"] == 177700000
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"]) and line["cpu_baseline"]["cores"] == 16
    assert len(line["config"]["workload"]) <= 200 and "all_kernels" not in txt
    assert line["inertial_config"]["roofline"]["kernel"] and line["mfma_config"]["roofline"]["bound"] == "mfma"


def test_bench_weak_scaling_line_under_torchrun_env():
    """The driver's launch: RANK / WORLD_SIZE already set by torch.distributed.run -> no spawning, world 1 here."""
    import json
    r = _bench("--gpus", "1", "--rehearse", "--scaling", "weak", "--frames", "7", "--steps", "2",
               env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode == 0, r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["config"]["sequences_total"] == 7


def test_bench_refuses_more_ranks_than_gpus():
    """On a node with fewer GPUs than --gpus the bench fails loudly instead of printing n_gpus: 1."""
    r = _bench("--gpus", "2", "--steps", "1", env={"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert r.returncode != 0 and "refusing" in r.stderr and "{" not in r.stdout


def test_bench_world_size_must_match_gpus():
    r = _bench("--gpus", "1", "--rehearse", env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
