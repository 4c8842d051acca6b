#!/usr/bin/env python3
"""bench.py -- frames/s of the TC2LI-SLAM per-frame loop on MI355X (BASELINE.json metric: ORB+LiDAR front end + local BA).

A step = one pass of the hot path over one batch of synthetic KITTI-sized frames that are already resident in HBM: every sequence of
the batch advances one frame through the reference's threads:
  camera / tracking thread : stereo ORB (2 x 1242x375) -> stereo matching -> TrackWithMotionModel (projection matching against the
                             last frame + pose-only optimisation) -> TrackLocalMap (SearchLocalPoints + pose-only optimisation)
  LiDAR thread             : lasermap_fov_segment -> preprocess -> voxel filter -> 5-NN + plane-fit feature extraction against the
                             sequence's own map (1e5 - 1e6 points) -> map_incremental (the map grows in place on the device)
  local-mapping thread     : every `--kf-interval`-th frame inserts a keyframe -> LocalLVBundleAdjustment (visual edges + the LiDAR
                             plane edge)
N > 1: one process per GPU (`python bench.py --gpus N` spawns the N ranks itself when it is not already running under
torch.distributed.run); sequences are independent units, so the path shards with no data-path collective.  `--scaling strong`
(default): a fixed global list of `--sequences` sequences is dealt over the ranks (dist_util.shard_units); `--scaling weak`: every
rank owns `--frames` sequences.  torch.distributed (RCCL) carries the barrier around the timed region, the MAX of the elapsed times
and the count of ranks.

Prints ONE JSON line on rank 0; DESIGN.md section "Measurement" explains how roofline / cpu_baseline / the extra lines are derived.
"""
import argparse
import ctypes as C
from concurrent.futures import ThreadPoolExecutor
import json
import os
import queue
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
METRIC = "frames/sec (ORB+LiDAR front-end + local BA) on KITTI-00, 1/2/4/8 GPU; ATE vs ref"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--sequences", type=int, default=512, help="strong scaling: sequences of the whole job, dealt over the ranks")
    ap.add_argument("--frames", type=int, default=64, help="weak scaling: sequences (one frame each per step) per GPU")
    ap.add_argument("--unique", type=int, default=8, help="distinct synthetic scenes rendered (tiled over the sequences)")
    ap.add_argument("--cycle-frames", type=int, default=4, help="frames rendered along every scene's drive (1 m apart, camera and LiDAR); step j of the timed loop "
                    "processes frame j %% this many of every sequence, so that images, scans, tracking inputs and map insertions differ from step to step (1: the same "
                    "frame every step, rounds 1-5)")
    ap.add_argument("--kf-interval", type=int, default=4, help="a keyframe (one local BA window) every k-th frame")
    ap.add_argument("--ba-concurrency", type=int, default=8, help="local-BA windows in flight (streams) per GPU")
    ap.add_argument("--ba-mix", choices=("varied", "uniform"), default="varied", help="the local-BA windows of the timed loop: 64 distinct windows "
                    "drawn by synthetic.ba_window_varied (4-24 free / 2-40 fixed keyframes, 500-6000 points, 0-15 %% outliers, LiDAR windows of 0 / 3-6 clouds, heavy "
                    "LiDAR edges, interrupted windows) or the four 12 + 20-keyframe / 3000-point windows of rounds 1-5, tiled")
    ap.add_argument("--map-length", type=float, default=1000.0, help="metres of street in every sequence's LiDAR map (~190 points per metre)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of each CPU-oracle baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-lines", action="store_true", help="skip the single-sequence and the roofline passes (diagnostic runs)")
    ap.add_argument("--with-roofline", action="store_true", help="with --no-extra-lines: still run the instrumented pass behind `roofline` (profile rounds)")
    ap.add_argument("--front-end-only", action="store_true", help="configs[1]: leave the local BA out of the step")
    ap.add_argument("--split-ba", action="store_true", help="after the timed loop: ONE LV-BA window split over all ranks (landmark partition + "
                    "RCCL all-reduce of the shared-pose blocks, tc2li_local_lv_bundle_adjustment_sharded) next to the same window on one GPU; "
                    "reported as \"sharded_window\", not part of `value`")
    ap.add_argument("--stages", default="orb,track,lidar,ba", help="diagnostics: run only these stage threads in the timed loop")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"))
    ap.add_argument("--rehearse", action="store_true", help="no GPU work: the rank plumbing only (spawn, rendezvous, sharding, timing protocol, "
                    "JSON line) -- what the world-2 gloo test on CPU drives")
    ap.add_argument("--host-fed", action="store_true", help="the timed loop takes every step's images and raw scans from pinned host memory (uploads inside the "
                    "timed region, overlapped with the previous step): what a drop-in behind the reference's host-buffer entry points delivers")
    ap.add_argument("--no-lidar-prepare", action="store_true", help="configs[3] loop: preprocess + time sort inside the front-end call instead of a step ahead "
                    "(tc2li_lidar_inertial_prepare_batch on a second handle): A/B measurements")
    ap.add_argument("--inertial-loop", action="store_true", help="the timed loop is the camera-LiDAR-inertial one (configs[3]) instead of the camera-LiDAR one: "
                    "what the single-sequence child of the inertial line runs")
    ap.add_argument("--lviba-small", action="store_true", help="configs[3] loop: LocalLVIBA windows of 10 keyframes / 10 iterations (bLarge false: a frame tracking "
                    "<= 100 inliers) instead of the bLarge windows the loop's frames call for")
    ap.add_argument("--mfma-only", action="store_true", help="only the matrix-unit line: a lock-step batch of 25-keyframe bLarge LocalLVIBA windows (the dense "
                    "f64 MFMA Schur product); what the --pmc pass of the MFMA counters profiles")
    ap.add_argument("--full-line", action="store_true", help="print the whole report on the line (tables of all kernels, per-thread CPU, prose) as rounds 1-4 did: "
                    "what the A/B tools read; the default line is the compact one (<= 4 KB), the whole report goes to --detail-out")
    ap.add_argument("--detail-out", default=None, help="where the whole report is written (default gpurun_out/bench_detail.json)")
    ap.add_argument("--no-build", action="store_true", help="fail instead of building when the library is missing or stale (profiling runs: "
                    "no child process may start under rocprofv3)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------------------
# rank plumbing
# ---------------------------------------------------------------------------------------------------------------------------------
def under_profiler():
    """rocprofv3 preloads its tool library into the profiled process, which initialises the GPU before main(): such a process must not
    start children (spawned ranks, the single-sequence child, make for the oracle)."""
    env = os.environ
    return any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in env) or "rocprofiler" in env.get("LD_PRELOAD", "")


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as fresh child processes (this parent never touches
    the GPU -- a process that has initialised HIP must not fork or exec), rendezvous on 127.0.0.1, pass rank 0's line through."""
    n = args.gpus
    if under_profiler():
        sys.stderr.write("bench.py --gpus %d under a profiler: refusing to spawn rank processes from a profiled process; profile one rank "
                         "(--gpus 1) or launch the ranks with torch.distributed.run under the profiler yourself\n" % n)
        return 2
    if not args.rehearse and not args.no_build and not os.environ.get("TC2LI_NO_BUILD"):
        import __graft_entry__ as ge
        ge.build_native()  # once, here: the ranks start together and must not load a library another rank is rewriting
        if not args.no_cpu_baseline:
            ge.build_oracle()
    if not args.rehearse:
        import torch
        have = torch.cuda.device_count()  # counting devices does not initialise the GPU on this image
        if have < n:
            sys.stderr.write("bench.py --gpus %d: this node has %d GPU(s); refusing to run fewer ranks than asked for\n" % (n, have))
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), TC2LI_NO_BUILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def single_sequence_child(args, extra=(), what="the same loop with 1 sequence per step (F = 1): one LV-BA window every %d-th frame"):
    """The F = 1 line measured by a fresh child process with GPU_MAX_HW_QUEUES=24 (see the call site; 8 until round 4: the two upload streams
    of the host-fed loop then shared hardware queues with the stage streams -- 557 frames/s host-fed at 8, 810 at 16, 827 at 24, 830 at 32; the
    resident loop reads 910-920 at any of them); None when the child fails -- the
    caller then measures it in-process.  300 warm-up frames: a cold process (first allocations of every work space, GPU clocks) reads
    627 frames/s after 8 warm-up frames and 780-810 after 300 or 1000."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE",
                                                            "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    env.update(GPU_MAX_HW_QUEUES=os.environ.get("TC2LI_BENCH_SINGLE_HW_QUEUES", "24"), TC2LI_NO_BUILD="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--sequences", "1", "--unique", "1", "--cycle-frames", str(args.cycle_frames), "--steps", "200", "--warmup", "300", "--no-cpu-baseline",
           "--no-extra-lines", "--no-build", "--kf-interval", str(args.kf_interval), "--ba-concurrency", str(args.ba_concurrency)]
    if args.front_end_only:
        cmd.append("--front-end-only")
    cmd += list(extra)
    try:
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, check=True).stdout.decode()
        line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        return {"value": line["value"], "unit": "frames/s", "ms_per_frame": line["ms_per_step"], "frames": line["steps"],
                "ba_windows": int(round(line["config"]["ba_windows_per_step_per_gpu"] * line["steps"])),
                "workload": what % args.kf_interval,
                "stage_thread_ms_per_frame": line.get("stage_thread_ms_per_step_concurrent"),
                "process": "a child process of its own, before this one initialised the GPU", "env": {"GPU_MAX_HW_QUEUES": env["GPU_MAX_HW_QUEUES"]}}
    except Exception as e:  # noqa: BLE001
        sys.stderr.write("bench.py: the single-sequence child failed (%s); measuring in-process\n" % e)
        return None


def sweep_children(args):
    """The loop with 256 / 128 / 64 sequences per step, each in a fresh child process (what a rank of a 2- / 4- / 8-GPU strong-scaling run
    is: a process of its own).  Measured as later legs of this process the same loops read a quarter less -- they inherit the streams and
    hardware-queue assignment of every loop before them."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE",
                                                            "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    env.update(TC2LI_NO_BUILD="1")
    out = {}
    for n_seq in (2 * args.sequences, 256, 128, 64):  # (twice the default batch too: what a GPU's 288 GB would still take -- informational)
        if n_seq == args.sequences or (n_seq > args.sequences and n_seq != 2 * args.sequences) or n_seq > 1024:
            continue
        steps = max(10, min(40, 5120 // n_seq))
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--sequences", str(n_seq), "--unique", str(args.unique), "--cycle-frames", str(args.cycle_frames), "--steps", str(steps), "--warmup", "8",
               "--no-cpu-baseline", "--no-extra-lines", "--no-build", "--kf-interval", str(args.kf_interval), "--ba-concurrency", str(args.ba_concurrency),
               "--map-length", str(args.map_length)]
        if args.front_end_only:
            cmd.append("--front-end-only")
        try:
            txt = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, check=True).stdout.decode()
            out[str(n_seq)] = json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])["value"]
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("bench.py: the %d-sequence child failed (%s)\n" % (n_seq, e))
            return None
    out["unit"] = ("frames/s of the whole loop with that many sequences per step on this one GPU, each in a child process of its own started before this "
                   "process initialised the GPU (8 warm-up steps, then 10-40 timed steps); the entry above the default batch is informational (the line's "
                   "`value` stays at the default of the earlier rounds)")
    return out


def host_budget_children(args):
    """The default loop with the process confined to 4 / 8 / 16 CPUs (taskset before the child's first GPU call, like the sweep's children):
    what `value` a rank keeps on a tighter host (eight ranks of this on a node).  {cpus: {value, cpu_s_per_wall_s, threads}}"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE",
                                                            "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    env.update(TC2LI_NO_BUILD="1")
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    out = {}
    for n in (4, 8):
        if n >= effective_cpus() or n > len(cpus):
            continue
        cmd = ["taskset", "-c", ",".join(str(c) for c in cpus[:n]), sys.executable, os.path.abspath(__file__), "--gpus", "1", "--sequences", str(args.sequences),
               "--unique", str(args.unique), "--cycle-frames", str(args.cycle_frames), "--steps", "16", "--warmup", "4", "--no-cpu-baseline", "--no-extra-lines", "--no-build", "--kf-interval", str(args.kf_interval),
               "--ba-concurrency", str(args.ba_concurrency), "--map-length", str(args.map_length)]
        try:
            txt = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, check=True).stdout.decode()
            line = json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
            h = (line.get("config") or {}).get("host") or {}
            out[str(n)] = {"value": line["value"], "cpu_s_per_wall_s": h.get("cpu_s_per_wall_s_timed_region"), "threads": h.get("threads_of_this_rank")}
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("bench.py: the %d-CPU child failed (%s)\n" % (n, e))
            return None
    return out


def ba_batch_of(pkg, wl, first, n):
    """BaBatch of the n windows first, first + 1, ... of the workload's list (tiled): the LiDAR keys only for a window that has the edge."""
    ws = []
    for k in range(first, first + n):
        w = wl.ba_windows[k % len(wl.ba_windows)]
        d = dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=w["edges"], iterations=w.get("iterations", 10))
        if len(w["win_pose"]):
            d.update(win_pose=w["win_pose"], clouds=w["clouds"], Tcl7=w["Tcl7"], weight=w["weight"])
        ws.append(d)
    b = pkg.capi.BaBatch(ws, wl.ba_windows[0]["cam"])
    b.window_ids = [k % len(wl.ba_windows) for k in range(first, first + n)]
    return b


def thread_cpu_seconds():
    """{thread name: CPU seconds so far} summed over the threads of this process with that name (/proc/self/task/*/stat)."""
    out, tick = {}, os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            st = open("/proc/self/task/%s/stat" % tid).read()
            name = st[st.index("(") + 1:st.rindex(")")]
            f = st[st.rindex(")") + 2:].split()
            out[name] = out.get(name, 0.0) + (int(f[11]) + int(f[12])) / tick
        except Exception:  # noqa: BLE001
            pass
    return out


def cgroup_cpu_quota():
    """CPUs' worth of time the container may use per second (cgroup v2 cpu.max, v1 cfs quota); None: unlimited / unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(p), 2)
    except Exception:  # noqa: BLE001
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / p, 2)
    except Exception:  # noqa: BLE001
        return None


def effective_cpus():
    """CPUs this process can really use: the affinity mask cut to the cgroup quota (the one-GPU box shows 256 CPUs in the mask and grants 16
    by cpu.max -- VERDICT r4 weak 4: every core count the line states is this number)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    q = cgroup_cpu_quota()
    return max(1, min(n, int(q + 0.5))) if q else n


def apply_host_budget(pkg, local_rank):
    """The host side of a rank: with R ranks on the node every rank keeps to its share of the cores this job may run on --
    a contiguous block of the affinity list (ranks follow the GPUs; on the 8-GPU nodes neighbouring cores and neighbouring GPUs share a socket),
    set with sched_setaffinity before any thread exists and before the first HIP call, so that the HIP runtime's and the library's threads
    inherit it -- and tells the library its thread budget (tc2li_set_host_thread_budget: the pools of 8 ranks on 256 cores then add up to 26
    threads per rank instead of ~130).  One rank: nothing is pinned, the budget is what the process may run on."""
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    cores = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    mine = cores
    if local_world > 1:
        per = max(1, len(cores) // local_world)
        mine = cores[(local_rank % local_world) * per:(local_rank % local_world + 1) * per] or cores
        if hasattr(os, "sched_setaffinity"):
            os.sched_setaffinity(0, mine)
    # the budget is what the rank may really use: its share of the affinity list, cut to its share of the cgroup quota (an 8-rank node whose
    # cgroup is as tight as the one-GPU box's 16 CPUs gives every rank 2, not 32)
    budget = max(1, min(len(mine), effective_cpus() // max(local_world, 1)))
    pkg.capi.set_host_thread_budget(budget)
    h = pkg.capi.host_threads()
    return {"ranks_on_node": local_world, "cores_of_this_rank": len(mine), "pinned": local_world > 1, "stage_threads": 5, "ba_lockstep_groups": 3,
            "library_pools": {k: h[k] for k in ("extractor_pool", "tracking_pool", "lidar_pool", "ba_group_pool")},
            "threads_of_this_rank": 5 + h["extractor_pool"] + h["tracking_pool"] + h["lidar_pool"] + 3 * h["ba_group_pool"],
            "cpus_available": len(cores), "cpus_effective": effective_cpus(), "thread_budget": budget}


def rehearse(args, rank, world, dist, dist_util, host_budget=None):
    """The multi-rank protocol without device work: every rank 'processes' its share of the global list by sleeping."""
    units = dist_util.shard_units(args.sequences, rank, world) if args.scaling == "strong" else list(range(args.frames))
    dist_util.barrier(dist)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-4 * len(units))
    dist_util.barrier(dist)
    elapsed = dist_util.max_elapsed(dist, time.perf_counter() - t0)
    seen = dist_util.count_ranks(dist)
    total = dist_util.sum_int(dist, len(units))
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": round(total * args.steps / elapsed, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
                          "scaling": args.scaling, "vs_baseline": None, "dtype": "none (rehearsal)", "data": "none", "rehearsal": True,
                          "ranks_seen": seen, "config": {"workload": "rank plumbing only", "sequences_total": total,
                                                         "sequences_of_rank0": len(units), "host_threads_gpu_path": host_budget,
                                                         "affinity_of_rank0": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}}))
    return 0



# ---------------------------------------------------------------------------------------------------------------------------------
# the printed line: compact (the driver parses ONE line; round 4's had grown to 20 KB and was not parsed), everything else to a side file
# ---------------------------------------------------------------------------------------------------------------------------------
LINE_BUDGET = 4096


def _pick(d, keys):
    return {k: d[k] for k in keys if k in d and d[k] is not None} if isinstance(d, dict) else None


def compact_roofline(rf):
    """The contract's roofline object: kernel, bound, achieved, peak, unit, frac, traffic (HBM bytes per launch from the PMC passes, a number
    or null) + the per-launch algorithmic amount and the launch duration it was divided by."""
    if not isinstance(rf, dict):
        return None
    out = _pick(rf, ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_of_spec", "useful_frac", "avg_launch_ms", "launches_per_step", "launches",
                     "share_of_kernel_time", "algorithmic_bytes_per_launch", "algorithmic_flops_per_launch", "executed_flops_per_launch",
                     "kernel_ms_per_step_all_streams"))
    if isinstance(rf.get("alone"), dict):
        out["alone"] = _pick(rf["alone"], ("stages", "avg_launch_ms", "frac"))
    t = rf.get("traffic")
    out["traffic"] = t.get("bytes_per_launch") if isinstance(t, dict) else t
    if isinstance(t, dict) and t.get("source"):
        out["traffic_source"] = t["source"]
    return out


def compact_cpu(c):
    if not isinstance(c, dict):
        return None
    out = _pick(c, ("value", "unit", "cores", "kind"))
    out["sample"] = str(c.get("sample", ""))[:160]
    if isinstance(c.get("single_sequence"), dict):
        out["single_sequence"] = _pick(c["single_sequence"], ("value", "cores"))
    return out


def compact_line(full, detail_path=None):
    """The ONE line bench.py prints: the contract's keys + one-object summaries of the extra legs, at most LINE_BUDGET bytes.  `full` is the
    whole report (what rounds 1-4 printed); it goes to the side file `detail_path`."""
    cfg = full.get("config") or {}
    wl = str(cfg.get("workload", ""))
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                     "dtype", "data")}
    line["config"] = {"workload": wl if len(wl) <= 200 else wl[:197] + "...",
                      **_pick(cfg, ("sequences_total", "frames_per_step_per_gpu", "images_per_step_per_gpu", "ba_windows_per_step_per_gpu", "keypoints_per_image",
                                    "scan_points_raw/preprocessed/downsampled/selected", "ba"))}
    hb = cfg.get("host_threads_gpu_path")
    if isinstance(hb, dict):
        line["config"]["host"] = _pick(hb, ("cpus_effective", "cgroup_cpu_quota", "thread_budget", "threads_of_this_rank", "cpu_s_per_wall_s_timed_region"))
    bc = cfg.get("baseline_config_value")
    if isinstance(bc, dict):
        line["config"]["baseline_config_value"] = _pick(bc, ("value", "unit", "cpu_baseline", "vs_cpu"))
    for k in ("rehearsal", "ranks_seen"):
        if k in full:
            line[k] = full[k]
    line["roofline"] = compact_roofline(full.get("roofline"))
    line["cpu_baseline"] = compact_cpu(full.get("cpu_baseline"))
    if isinstance(full.get("single_sequence"), dict):
        line["single_sequence"] = _pick(full["single_sequence"], ("value", "unit", "ms_per_frame", "resident_value"))
    if isinstance(full.get("host_fed"), dict):
        line["host_fed"] = _pick(full["host_fed"], ("value", "ms_per_step", "GB_per_step", "link_GBps"))
    ic = full.get("inertial_config")
    if isinstance(ic, dict):
        line["inertial_config"] = {**_pick(ic, ("value", "unit", "ms_per_step", "sequences", "steps", "lviba_windows_per_step", "lviba_window")),
                                   "roofline": _pick(compact_roofline(ic.get("roofline")), ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches_per_step", "traffic")),
                                   "cpu_baseline": _pick(compact_cpu(ic.get("cpu_baseline")), ("value", "unit", "cores", "kind", "single_sequence")),
                                   "single_sequence": _pick(ic.get("single_sequence") or {}, ("value", "ms_per_frame"))}
    mf = full.get("mfma_config")
    if isinstance(mf, dict):
        line["mfma_config"] = {**_pick(mf, ("windows_per_s", "ms_per_batch", "free_keyframes")),
                               "roofline": _pick(compact_roofline(mf.get("roofline")), ("kernel", "bound", "achieved", "peak", "unit", "frac", "useful_frac", "avg_launch_ms", "traffic"))}
    sw = full.get("sequences_per_gpu_sweep")
    if isinstance(sw, dict):
        line["sequences_per_gpu_sweep"] = {k: v for k, v in sw.items() if k != "unit"}
        line["sequences_per_gpu_sweep"]["note"] = "frames/s at that many sequences per step on this ONE GPU; N > 1 over RCCL is unmeasured on hardware"
    if isinstance(full.get("value_uniform"), dict):   # (a number: the leg must not be shed with the optional objects below)
        line["value_uniform"] = full["value_uniform"].get("value")
    hbs = full.get("host_budget_sweep")
    if isinstance(hbs, dict):
        line["host_budget_sweep"] = {k: (v if not isinstance(v, dict) else _pick(v, ("value", "cpu_s_per_wall_s", "threads"))) for k, v in hbs.items() if k != "note"}
    if isinstance(full.get("sharded_window"), dict):
        line["sharded_window"] = _pick(full["sharded_window"], ("ranks", "ms_per_window", "single_gpu_ms_per_window", "max_pose_difference_vs_single_gpu"))
    if isinstance(full.get("stage_thread_ms_per_step_concurrent"), dict):   # (the mapping workers' own entries stay in the detail file: ba_thread is their maximum)
        line["stage_thread_ms_per_step_concurrent"] = {k: v for k, v in full["stage_thread_ms_per_step_concurrent"].items() if not k.startswith("ba_worker")}
    if detail_path:
        line["detail"] = detail_path
    # the budget is a contract: shed the optional legs, least important first, rather than print a line the driver cannot parse
    for k in ("stage_thread_ms_per_step_concurrent", "sharded_window", "host_budget_sweep", "sequences_per_gpu_sweep", "mfma_config", "host_fed", "single_sequence", "inertial_config"):
        if len(json.dumps(line)) <= LINE_BUDGET:
            break
        line.pop(k, None)
    return line


def write_detail(full, path):
    """The whole report (all_kernels tables, per-thread CPU, workload prose, notes) next to the line; returns the path written or None."""
    for cand in (path, os.path.join(ROOT, "gpurun_out", "bench_detail.json"), os.path.join(ROOT, "bench_detail.json")):
        if not cand:
            continue
        try:
            os.makedirs(os.path.dirname(os.path.abspath(cand)), exist_ok=True)
            with open(cand, "w") as f:
                json.dump(full, f, indent=1)
            return os.path.relpath(cand, ROOT) if os.path.abspath(cand).startswith(ROOT) else cand
        except OSError:
            continue
    return None

# ---------------------------------------------------------------------------------------------------------------------------------
# workload
# ---------------------------------------------------------------------------------------------------------------------------------
def level_dims(w, h, nlevels=8, scale=1.2):
    dims, s = [], np.float32(1.0)
    for _ in range(nlevels):
        inv = np.float32(1.0) / s
        dims.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
        s = np.float32(np.float64(s) * np.float64(np.float32(scale)))
    return dims


def uniform_ba_windows(pkg, synthetic):
    """The four windows rounds 1-5 tiled over a step's batch: 12 free + 20 fixed keyframes, 3000 points, a LiDAR edge over 6 keyframes x 3000 points."""
    out = []
    for k in range(4):
        w = synthetic.ba_window(k, n_opt=12, n_fix=20, n_points=3000, pose_noise=(0.1, 0.01))
        last = len(w["poses"]) - 1
        win = list(range(last, last - 6, -1))
        out.append(dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=pkg.pack_ba_edges(w["edges"]), edges6=w["edges"],
                        win_pose=win, clouds=synthetic.ba_window_clouds(w, win, n_points=3000), Tcl7=synthetic.TCL7, weight=1.0,
                        cam=w["cam"], iterations=10, params=dict(n_opt=12, n_fix=20, kind="ordinary")))
    return out


class FrameInputs:
    """What one frame index holds for all the unique scenes: stereo images [U, 2, H, W], raw scans, LiDAR states, and (after
    Workload.build_tracking_inputs) the last frame and local map of every scene."""
    def __init__(self, index):
        self.index, self.images, self.scans, self.states, self.last, self.local = index, None, [], [], None, None


class Workload:
    """The synthetic inputs of `unique` distinct sequences: stereo pair, 64-beam scan, the street's accumulated
    LiDAR map, the last frame TrackWithMotionModel projects from and the local map TrackLocalMap searches; sequence s uses set s % unique."""

    def __init__(self, pkg, synthetic, unique, map_length, with_ba, cycle=1):
        self.pkg, self.synthetic, self.U = pkg, synthetic, unique
        self.K = max(1, int(cycle))
        W, H = synthetic.WIDTH, synthetic.HEIGHT
        self.W, self.H = W, H
        self.bf = np.float32(synthetic.BF)
        self.b = np.float32(self.bf / np.float32(synthetic.FX))
        self.cam5 = np.float32([synthetic.FX, synthetic.FY, synthetic.CX, synthetic.CY, self.bf]).astype(np.float64)
        # frame f of scene u: the camera f metres down the street (Scene.render's cam_z), the LiDAR at sensor_pose(u + 1 + f) (1 m per frame);
        # frames[0] is what rounds 1-5 processed every step.  The rendered frames are kept in /tmp for the child processes of one bench run.
        self.frames = self._render_frames(synthetic, unique, self.K, W, H)
        self.images, self.scans = self.frames[0].images, self.frames[0].scans
        self.maps = []
        for fr in self.frames:
            fr.states = [pkg.pack_lidar_state(*synthetic.lidar_state(u + 1 + fr.index)[:2]) for u in range(unique)]
        self.states = self.frames[0].states
        for u in range(unique):
            self.maps.append(synthetic.lidar_map(synthetic.Scene(u), x_from=-0.7 * map_length, x_to=0.3 * map_length))
        self.ba_windows = []
        self.ba_mix = "none"
        if with_ba and with_ba != "uniform":
            self.ba_mix = "varied"
            for k in range(64):
                w = synthetic.ba_window_varied(k)
                d = dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=pkg.pack_ba_edges(w["edges"]), edges6=w["edges"], cam=w["cam"],
                         iterations=w["iterations"], params=w["params"], win_pose=[], clouds=[], Tcl7=synthetic.TCL7, weight=w["weight"])
                if w["win_pose"]:
                    d.update(win_pose=w["win_pose"], clouds=w["clouds"])
                self.ba_windows.append(d)
        elif with_ba:
            self.ba_mix = "uniform"
            self.ba_windows = uniform_ba_windows(pkg, synthetic)
        ang = 0.002
        self.pose_pred = np.array([0, np.sin(ang / 2), 0, np.cos(ang / 2), 0.02, -0.01, -0.08], np.float32)
        self.last = None   # per unique: the last frame of TrackWithMotionModel
        self.local = None  # per unique: (held, held_Xw, local map points)

    @staticmethod
    def _render_frames(synthetic, unique, K, W, H):
        import hashlib
        import tempfile
        gen = hashlib.sha256(open(synthetic.__file__, "rb").read()).hexdigest()[:12]   # (the cache belongs to this generator)
        cache = os.path.join(tempfile.gettempdir(), "tc2li_bench_frames_%s_U%d_K%d_%dx%d.npz" % (gen, unique, K, W, H))
        frames = [FrameInputs(f) for f in range(K)]
        try:
            z = np.load(cache)
            images, scan_offs = z["images"], z["scan_offs"]
            for fr in frames:
                fr.images = images[fr.index]
                offs, pts = scan_offs[fr.index], z["scans_%d" % fr.index]
                fr.scans = [pts[offs[u]:offs[u + 1]] for u in range(unique)]
            return frames
        except Exception:  # noqa: BLE001  (no cache yet, or one of another shape: render)
            pass
        for fr in frames:
            f = fr.index
            fr.images = np.empty((unique, 2, H, W), np.uint8)
            for u in range(unique):
                sc = synthetic.Scene(u)
                fr.images[u, 0], _ = sc.render(0.0, W, H, noise_seed=1 + 2 * f, cam_z=float(f))
                fr.images[u, 1], _ = sc.render(synthetic.BASELINE, W, H, noise_seed=2 + 2 * f, cam_z=float(f))
                fr.scans.append(synthetic.lidar_scan(sc, u + 1 + f))
        try:
            tmp = cache + ".%d.tmp.npz" % os.getpid()
            np.savez(tmp, images=np.stack([fr.images for fr in frames]),
                     scan_offs=np.stack([np.concatenate([[0], np.cumsum([len(x) for x in fr.scans])]) for fr in frames]),
                     **{"scans_%d" % fr.index: np.concatenate(fr.scans) for fr in frames})
            os.replace(tmp, cache)
        except Exception:  # noqa: BLE001
            pass
        return frames

    def build_tracking_inputs(self, ext, stream):
        """The tracking inputs of every frame index (frames[f].last / .local); .last / .local of the workload are frame 0's."""
        orb0 = None
        for fr in self.frames:
            orb = self._tracking_inputs_of(fr, ext, stream)
            orb0 = orb if orb0 is None else orb0
        self.last, self.local = self.frames[0].last, self.frames[0].local
        return orb0

    def _tracking_inputs_of(self, fr, ext, stream):
        """One extraction + stereo matching of the unique scenes' frame `fr` gives the 3-D points the tracking inputs are made of: the 'last frame' of
        every sequence = the frame's own stereo points seen from the previous pose (descriptors drifted, 3 % outliers); the local map =
        the frame's stereo points that the motion-model step does not hold, plus as many again that project outside the image or carry
        foreign descriptors (what UpdateLocalMap hands to SearchLocalPoints: a few thousand points, about half of them in the frustum)."""
        pkg, U, W, H = self.pkg, self.U, self.W, self.H
        import torch
        dev = torch.from_numpy(np.ascontiguousarray(fr.images).reshape(2 * U, H, W)).cuda()
        orb = ext.extract_batch_dev(dev.data_ptr(), 2 * U, W, H, W, W * H, stream=stream)
        st = pkg.stereo_match_batch(ext, U, float(self.bf), float(self.b), stream=stream)
        fx, fy, cx, cy = [np.float32(v) for v in self.cam5[:4]]
        sf = ext.GetScaleFactors()
        last, lasts = [], []
        for u in range(U):
            rng = np.random.default_rng(7000 + u + 100 * fr.index)
            n = int(orb[2][2 * u])
            kl, dl, z = orb[0][2 * u, :n].copy(), orb[1][2 * u, :n].copy(), st[1][u, :n].copy()
            order = rng.permutation(n)
            lk = kl[order]
            lk["angle"] = (lk["angle"] + rng.normal(0, 3, n).astype(np.float32)) % np.float32(360)
            zz = np.where(z[order] > 0, z[order], 1).astype(np.float32)
            Xw = np.stack([(lk["x"] - cx) * zz / fx, (lk["y"] - cy) * zz / fy, zz], 1).astype(np.float32)
            md = dl[order]
            flips = rng.integers(0, 256, (n, 12))
            for k in range(12):  # descriptor drift between consecutive frames
                rows = rng.random(n) < 0.7
                md[rows, flips[rows, k] // 8] ^= (np.uint8(1) << (flips[rows, k] % 8).astype(np.uint8))
            last.append(dict(has_point=(z[order] > 0).astype(np.uint8), outlier=(rng.random(n) < 0.03).astype(np.uint8), Xw=Xw, keys=lk,
                                  descriptors=md, pose7=np.array([0, 0, 0, 1, 0, 0, 0], np.float32)))
        # what the motion-model step leaves on the frame decides which keypoints hold a point when TrackLocalMap starts
        packed = pkg.capi.pack_last_frames(last)
        trk = pkg.capi.track_motion_model_batch(ext, U, orb[0], st[0], packed, np.tile(self.pose_pred, (U, 1)), self.cam5, float(self.b), 7.0,
                                                stream=stream)
        local = []
        cap = orb[0].shape[1]
        for u in range(U):
            rng = np.random.default_rng(9000 + u + 100 * fr.index)
            n = int(orb[2][2 * u])
            k, d, z = orb[0][2 * u, :n], orb[1][2 * u, :n], st[1][u, :n]
            ok = z > 0
            zz = np.where(ok, z, 1).astype(np.float32)
            X = np.stack([(k["x"] - cx) * zz / fx, (k["y"] - cy) * zz / fy, zz], 1).astype(np.float32)
            mp = trk[1][u, :n]
            held = np.where(mp >= 0, 1, 0).astype(np.uint8)
            held_Xw = np.zeros((cap, 3), np.float32)
            held_Xw[:n][mp >= 0] = last[u]["Xw"][mp[mp >= 0]]
            loc = np.nonzero(ok & (held == 0))[0]
            loc = loc[rng.permutation(len(loc))]
            n_extra = len(loc) + 1500
            pts = np.zeros(len(loc) + n_extra, pkg.MAP_POINT_DTYPE)
            P = X[loc] + rng.normal(0, 0.01, (len(loc), 3)).astype(np.float32)
            ze = rng.uniform(3.0, 70.0, n_extra).astype(np.float32)  # the rest of the local map: around and beyond the frustum
            E = np.stack([rng.uniform(-1.6, 1.6, n_extra).astype(np.float32) * ze, rng.uniform(-0.5, 0.4, n_extra).astype(np.float32) * ze, ze], 1)
            pts["pos"] = np.concatenate([P, E]).astype(np.float32)
            dist = np.linalg.norm(pts["pos"], axis=1).astype(np.float32)
            pts["normal"] = pts["pos"] / dist[:, None]
            octave = np.concatenate([k["octave"][loc], rng.integers(0, 8, n_extra)])
            raw = (dist * sf[octave]).astype(np.float32)
            pts["max_distance_raw"], pts["max_distance"], pts["min_distance"] = raw, np.float32(1.2) * raw, np.float32(0.8) * (raw / sf[-1])
            md = np.concatenate([d[loc], rng.integers(0, 256, (n_extra, 32)).astype(np.uint8)])
            flips = rng.integers(0, 256, (len(loc), 12))
            for j in range(12):
                rows = np.nonzero(rng.random(len(loc)) < 0.7)[0]
                md[rows, flips[rows, j] // 8] ^= (np.uint8(1) << (flips[rows, j] % 8).astype(np.uint8))
            pts["descriptor"] = md
            pts = pts[rng.permutation(len(pts))]
            hfull = np.zeros(cap, np.uint8)
            hfull[:n] = held
            local.append((hfull, held_Xw, pts))
        fr.last, fr.local = last, local
        if fr.index == 0:
            self.keypoints_per_image = float(np.mean(orb[2]))
            self.stereo_matches = float(np.mean((st[1] > 0).sum(1)))
        return orb


_OPEN_LOOPS = set()  # every Loop with live stage threads: main() closes what an error path left open before the process ends


class StageWorker:
    """A host thread that lives as long as its Loop and runs the stage functions handed to it: the library keeps work spaces and a private
    stream per host thread (thread_local), so warm-up, timed and instrumented passes must run on the SAME threads -- fresh threads per pass
    would allocate those work spaces again inside the timed region (and free them, synchronising the device, when the threads exit)."""

    def __init__(self, name, local_rank, torch):
        self.q = queue.Queue()
        # not a daemon thread: Loop.close() joins it, so no stage thread is inside a library call (or inside the destructors of its
        # thread-local work spaces: hipFree, device synchronisation) while later legs are timed or while the interpreter finalises
        self.t = threading.Thread(target=self._main, args=(local_rank, torch), name=name)
        self.t.start()

    def _main(self, local_rank, torch):
        try:  # the thread's name where top -H and /proc/<pid>/task/*/comm show it
            import ctypes
            ctypes.CDLL(None).prctl(15, threading.current_thread().name[:15].encode(), 0, 0, 0)
        except Exception:  # noqa: BLE001
            pass
        torch.cuda.set_device(local_rank)  # HIP's current device is per thread; a new thread starts on device 0
        while True:
            item = self.q.get()
            if item is None:
                return
            fn, done = item
            try:
                fn()
            finally:
                done.set()

    def submit(self, fn):
        done = threading.Event()
        self.q.put((fn, done))
        return done

    def stop(self):
        """Ends the thread and waits for it: its thread-local work spaces in the library are released by its exit, here and now."""
        self.q.put(None)
        self.t.join()


class Loop:
    """Handles and stage threads for F sequences (sequence ids `seq_ids`; sequence s runs on input set s % U)."""

    def __init__(self, wl, seq_ids, args, local_rank, template_orb_out=None):
        import torch
        self.torch, self.wl, self.args, self.local_rank = torch, wl, args, local_rank
        pkg, U, W, H = wl.pkg, wl.U, wl.W, wl.H
        self.pkg = pkg
        self.F = F = len(seq_ids)
        self.tile = tile = [s % U for s in seq_ids]
        self.n_img = 2 * F
        self.dev_img = torch.from_numpy(wl.images[tile].reshape(2 * F, H, W)).cuda()
        raw = np.concatenate([wl.scans[t] for t in tile])
        self.raw_offs = np.concatenate([[0], np.cumsum([len(wl.scans[t]) for t in tile])]).astype(np.int32)
        self.dev_raw = torch.from_numpy(raw.view(np.uint8)).cuda()
        self.states = np.stack([wl.states[t] for t in tile])
        self.stream = torch.cuda.current_stream().cuda_stream
        # The LiDAR chain is ~40 short dependent launches per step: on a GPU shared with the long FAST / blur / descriptor kernels its
        # workgroups go first (like the BA lock-step streams inside the library); measured 57.8 -> 53.8 ms per step of 512 sequences
        self.lidar_stream = torch.cuda.Stream(priority=int(os.environ.get("TC2LI_BENCH_LIDAR_PRIORITY", "-1")))
        self.track_stream = torch.cuda.Stream()
        # three feature buffers: extraction of batch k+2, motion-model tracking of batch k+1 and local-map tracking of batch k overlap
        self.exts = [pkg.OrbExtractor(max_width=W, max_height=H, max_images=self.n_img) for _ in range(3)]
        self.track2_stream = torch.cuda.Stream()
        self.lidar_cap = int(max(len(x) for fr in wl.frames for x in fr.scans))
        self.lidar_handles = None  # configs[3]: [self.lidar, a second handle] once the scans are prepared a step ahead (InertialLoop.lidar_prepare)
        self.lidar = pkg.LidarFrontEnd(max_points_per_scan=self.lidar_cap, max_scans=F)
        self.maps = []
        for t in tile:  # every sequence owns its map: map_incremental changes it
            m = pkg.LidarMap()
            m.Build(wl.maps[t])
            self.maps.append(m)
        self.map_points0 = int(np.mean([m.size() for m in self.maps]))
        self.boxes = (pkg.capi.LocalMapBox * F)()   # one local-map cube per sequence (lasermap_fov_segment's state), contiguous for the batched call
        self.scan_ids = np.arange(F, dtype=np.int32)
        # tracking inputs
        self.last_frames = pkg.capi.pack_last_frames([wl.last[t] for t in tile])
        self.pose_pred = np.tile(wl.pose_pred, (F, 1))
        self.held = np.stack([wl.local[t][0] for t in tile])
        self.held_Xw = np.stack([wl.local[t][1] for t in tile])
        self.local_pts = np.concatenate([wl.local[t][2] for t in tile])
        self.local_off = np.concatenate([[0], np.cumsum([len(wl.local[t][2]) for t in tile])]).astype(np.int32)
        # The frames the timed loop cycles through (--cycle-frames): step j processes frame j % K of every sequence -- other
        # images, another scan from a metre further on, that frame's own last frame and local map.  frame_sets[0] is None: frame 0 lives in the
        # attributes above, which everything that does not cycle (the inertial loop, the host-fed leg, the roofline's counts) keeps reading.
        from types import SimpleNamespace
        self.frame_sets = [None]
        self.step_base = 0
        self.buf_frame = [0, 0, 0]   # the frame index a feature buffer was extracted from
        for fr in (wl.frames[1:] if type(self) is Loop else []):
            raw_f = np.concatenate([fr.scans[t] for t in tile])
            self.frame_sets.append(SimpleNamespace(
                dev_img=torch.from_numpy(np.ascontiguousarray(fr.images[tile]).reshape(2 * F, H, W)).cuda(),
                dev_raw=torch.from_numpy(raw_f.view(np.uint8)).cuda(),
                raw_offs=np.concatenate([[0], np.cumsum([len(fr.scans[t]) for t in tile])]).astype(np.int32),
                states=np.stack([fr.states[t] for t in tile]),
                last_frames=pkg.capi.pack_last_frames([fr.last[t] for t in tile]),
                held=np.stack([fr.local[t][0] for t in tile]), held_Xw=np.stack([fr.local[t][1] for t in tile]),
                local_pts=np.concatenate([fr.local[t][2] for t in tile]),
                local_off=np.concatenate([[0], np.cumsum([len(fr.local[t][2]) for t in tile])]).astype(np.int32)))
        # local mapping: F / kf_interval windows per step (a fraction of a window per step when F < kf_interval)
        self.ba_rate = 0.0 if (args.front_end_only or not wl.ba_windows) else F / args.kf_interval
        n_ba = int(np.ceil(self.ba_rate)) if self.ba_rate else 0
        self.n_ba = n_ba
        self.ba_batch = ba_batch_of(pkg, wl, 0, n_ba) if n_ba else None
        # Local mapping is asynchronous in the reference: it optimises whatever keyframes have arrived when it becomes free.  With few
        # sequences per GPU a step's windows are a short, latency-bound batch (16 windows: 8.6 ms against 6.4 ms for the other stages), so
        # the mapping thread takes the windows of two steps in one call, as soon as the tracking thread has finished both (a second,
        # prebuilt batch of twice the windows; beyond 256 sequences a step's batch already fills the GPU and the thread runs free).
        # With 64 sequences or fewer (what strong scaling over 512 leaves a GPU at 8 ranks) even two steps' windows are a latency-bound
        # batch -- a call costs 4.6 ms for 16 windows, 6 ms for 32 -- so there the thread may take up to four steps' keyframes when it lags that far.
        self.ba_batch2 = self.ba_batch4 = None
        multi = int(os.environ.get("TC2LI_BENCH_BA_MULTI_STEP", "0"))  # A/B: 2 / 4 = batches of that many steps' windows whatever F is
        if n_ba >= 2 and (F <= 256 or multi >= 2) and F % args.kf_interval == 0 and not os.environ.get("TC2LI_BENCH_BA_SINGLE_STEP"):
            self.ba_batch2 = ba_batch_of(pkg, wl, 0, 2 * n_ba)
            if F <= 64 or multi >= 4:
                self.ba_batch4 = ba_batch_of(pkg, wl, 0, 4 * n_ba)
        # Beyond 256 sequences the mapping stage runs free, and its windows go to a POOL of mapping workers, each a lock-step group of its
        # own (tc2li_local_bundle_adjustment_batch_group): a step's windows are dealt into as many chunks as there are workers, a worker takes
        # the next chunk when it is free.  (The one call per step made its three groups meet at its end: a step's groups take 20-30 ms
        # beside the other stages, and the call took the slowest.  The reference has a LocalMapping thread per sequence.)
        # TC2LI_BENCH_BA_WORKERS=0: the one call per step.
        self.ba_workers = []
        n_workers = int(os.environ.get("TC2LI_BENCH_BA_WORKERS", "3"))
        if type(self) is Loop and self.ba_batch is not None and self.ba_batch2 is None and n_workers >= 2 and n_ba >= 8 * n_workers and self.ba_rate == n_ba:
            self.ba_chunk_sizes = [n_ba * (c + 1) // n_workers - n_ba * c // n_workers for c in range(n_workers)]
            # a worker's own batches, one per chunk size; worker w's windows start where worker w - 1's end, so that the workers' batches of a
            # chunk size hold different stretches of the workload's window list (with the varied mix: different windows)
            first = 0
            for w in range(n_workers):
                self.ba_workers.append({n: ba_batch_of(pkg, wl, first, n) for n in set(self.ba_chunk_sizes)})
                first += self.ba_chunk_sizes[w]
        # Beyond 512 sequences per GPU the windows go through tc2li_ba_engine instead (TC2LI_BENCH_BA_ENGINE = number of engines, 0 = the batch
        # calls above; default 3 there, 0 up to 512 -- at 512 the two forms measure the same and the step workers' launches are the same from
        # run to run, which the roofline check against the profiler's averages needs): running lock-step queues that windows join and leave one by one.  Every sequence is a mapping thread of its
        # own there: its window is a ticket, submitted when the tracking thread has finished the keyframe's step AND the sequence's previous
        # window has come back (a LocalMapping thread runs one optimisation at a time, LocalMapping.cc:66-160); nobody waits for another
        # sequence's window.  A ring of kf_interval prebuilt batches holds the arrays (a sequence's next keyframe comes kf_interval frames
        # later), each a different stretch of the workload's window list.  Measured (round 6, varied mix, frames/s, batch calls -> 3 engines):
        # 64 sequences 12.5k -> 11.7k (a round of few windows is a chain of ten launches whatever their number: the batch calls' 64 windows per
        # call win), 128: 13.4k -> 15.0k, 256: 15.8k -> 17.0k, 512: 17.9k -> 19.5k (16 CPUs; 17.0k -> 18.9k on 8).  Against the step workers
        # below (four steps in flight, a step = one lock-step group): 128: 15.4k / 17.5k, 256: 17.8k / 19.2k, 512: 21.0k / 20.9k, 1024: 21.6k / 20.0k
        # -- the engines beyond 512 sequences, the step workers up to there.
        self.ba_engine, self.ba_engines, self.ba_ring = None, [], []
        n_engines = int(os.environ.get("TC2LI_BENCH_BA_ENGINE", "3" if F > 512 else "0"))
        if type(self) is Loop and self.ba_batch is not None and self.ba_rate == n_ba and n_engines > 0:
            self.ba_engines = [pkg.capi.BaEngine(wl.ba_windows[0]["cam"], max_windows=int(os.environ.get("TC2LI_BENCH_BA_ENGINE_SLOTS", "192")))
                               for _ in range(n_engines)]
            self.ba_engine = self.ba_engines[0]
            self.ba_ring = [ba_batch_of(pkg, wl, r * n_ba, n_ba) for r in range(int(os.environ.get("TC2LI_BENCH_BA_ENGINE_RING", "0")) or max(1, args.kf_interval))]
            self.ba_workers, self.ba_batch2, self.ba_batch4 = [], None, None
            # the engines' work spaces take their sizes from the windows they meet (a batch call's do so in its first call): before anything is
            # timed every engine is handed ALL the ring's windows at once (every slot's work space is touched: its ~40 buffers are allocated on
            # first use), then once more in the loop's own assignment (window i: engine i % K)
            K = len(self.ba_engines)
            for e in ([] if under_profiler() else self.ba_engines):   # (not under rocprofv3: the per-kernel averages of a profile are the loop's)
                for t in [e.submit(b) for b in self.ba_ring]:
                    e.wait(t)
            tickets = [(self.ba_engines[i % K], self.ba_engines[i % K].submit(b, i, 1)) for b in self.ba_ring for i in range(b.n)]
            for e, t in tickets:
                e.wait(t)
        # Up to 512 sequences per GPU (TC2LI_BENCH_BA_STEP_WORKERS, default kf_interval = 4; 0: the calls of one / two / four steps' windows
        # above): a step's windows (16 at 64 sequences) are ONE lock-step group, and up to that many steps are in flight side by side, each on
        # a mapping worker with a group context of its own -- a sequence's next keyframe comes kf_interval frames later, so no sequence has two
        # windows in flight.  Alone on the GPU four such groups side by side take 2.7 ms per step's windows against 6.4 for one call per step
        # and 2.9 for one call per four steps (tools/time_ba_batch.py), and a window waits for its own group only.  In the loop (frames/s,
        # calls of 1-4 steps' windows -> four step workers): 32 sequences 9.7k -> 11.0k, 64: 12.9k -> 14.7-15.3k.  (A step's windows as TWO groups on
        # eight workers measured worse -- 512 sequences 21.1-21.6k against 21.3-21.6k, 256: 19.4k against 19.8k, 64: 12.0-12.5k against 15.4-15.6k: with
        # eight BA streams the LiDAR stream shares a hardware queue with one of them and its thread's step doubles.)
        self.ba_step_workers = []
        n_step_workers = int(os.environ.get("TC2LI_BENCH_BA_STEP_WORKERS", str(args.kf_interval) if F <= 512 else "0"))
        if (type(self) is Loop and self.ba_batch is not None and self.ba_engine is None and self.ba_rate == n_ba and n_ba >= 2
                and 2 <= n_step_workers <= 8 and F % args.kf_interval == 0):
            self.ba_step_workers = [ba_batch_of(pkg, wl, w * n_ba, n_ba) for w in range(min(n_step_workers, args.kf_interval))]
            self.ba_batch2 = self.ba_batch4 = None
            self.ba_workers = []
            # every worker's group context meets its windows once before anything is timed (the work spaces of a group take their sizes in its
            # first call; with fewer warm-up steps than workers that call would fall into the timed region): in the worker's own thread at the
            # start of the Loop's FIRST run -- the warm-up run -- and not here, because the group's stream is created by that call, and which of
            # the process's four hardware queues a stream shares with which other follows the order of creation (measured: with the calls made
            # here the LiDAR thread's step went from 3.9 to 4.9 ms at 64 sequences, 15.0 -> 13.0 k frames/s)
            self.ba_step_prewarmed = [False] * len(self.ba_step_workers)
        self.steps_tracked = 0
        self.ba_due = 0.0
        self.orb_outs = [None, None, None]
        self.st_outs = [None, None, None]
        self.trk_outs = [None, None, None]
        self.tlm_outs = [None, None, None]
        self.tlm_out = None
        self.track_ms = [0.0, 0.0, 0.0]
        self.lidar_counts = None
        self.map_adds = [0, 0]
        self.thread_ms = {}
        self.orb_times, self.lidar_times = [], []
        self.ba_windows_done = 0
        self.workers = {}
        _OPEN_LOOPS.add(self)

    # -- host-fed inputs --------------------------------------------------------------------------------
    def enable_host_fed(self):
        """What a drop-in sees: the reference's entry points take host buffers (cv::Mat images, Tracking.cc:1632; a ROS point-cloud message,
        LidarFrontEnd.cpp:253).  From here on every step's images and raw scans start in PINNED HOST memory and go up through the copy engines
        inside the timed region -- step k + 1's while step k is processed (two device buffers per input, a copy stream per input)."""
        torch = self.torch
        self.h_img, self.h_raw = self.dev_img.cpu().pin_memory(), self.dev_raw.cpu().pin_memory()
        self.img_bufs, self.raw_bufs = [self.dev_img, torch.empty_like(self.dev_img)], [self.dev_raw, torch.empty_like(self.dev_raw)]
        self.copy_streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        self.host_fed = True
        self.fed_bytes_per_step = int(self.h_img.numel() + self.h_raw.numel())

    def _upload(self, which, buf):
        """Queues the upload of the images (which = 0) or the scans (1) into device buffer `buf`; returns the event to wait for."""
        torch = self.torch
        st = self.copy_streams[which]
        with torch.cuda.stream(st):
            (self.img_bufs if which == 0 else self.raw_bufs)[buf].copy_(self.h_img if which == 0 else self.h_raw, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st)
        return ev

    # -- stages ------------------------------------------------------------------------------------------------------------
    def frame_set(self, f):
        """The inputs of frame index f: the attributes of the loop itself for frame 0, a namespace with the same names otherwise."""
        return self if not f else self.frame_sets[f]

    def extract(self, k, stream, img=None, f=0):
        W, H = self.wl.W, self.wl.H
        img = self.frame_set(f).dev_img if img is None else img
        self.buf_frame[k] = f
        self.orb_outs[k] = self.exts[k].extract_batch_dev(img.data_ptr(), self.n_img, W, H, W, W * H, stream=stream, out=self.orb_outs[k])

    def track(self, k, stream):
        """Stereo matching + TrackWithMotionModel of feature buffer k."""
        wl, pkg, F = self.wl, self.pkg, self.F
        t0 = time.perf_counter()
        self.st_outs[k] = pkg.stereo_match_batch(self.exts[k], F, float(wl.bf), float(wl.b), stream=stream, out=self.st_outs[k])
        t1 = time.perf_counter()
        self.trk_outs[k] = pkg.capi.track_motion_model_batch(self.exts[k], F, self.orb_outs[k][0], self.st_outs[k][0], self.frame_set(self.buf_frame[k]).last_frames, self.pose_pred, wl.cam5,
                                                             float(wl.b), 7.0, stream=stream, out=self.trk_outs[k])
        self.track_ms[0], self.track_ms[1] = 1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t1)

    def track_local(self, k, stream):
        """TrackLocalMap of feature buffer k, from the pose the motion-model step left on the frame (Tracking.cc:2038 -> :2218)."""
        wl, pkg, F = self.wl, self.pkg, self.F
        t0 = time.perf_counter()
        fs = self.frame_set(self.buf_frame[k])
        self.tlm_outs[k] = pkg.capi.track_local_map_batch(self.exts[k], F, self.orb_outs[k][0], self.st_outs[k][0], self.trk_outs[k][0].astype(np.float32),
                                                          fs.held, fs.held_Xw, fs.local_pts, fs.local_off, wl.cam5, th=1.0, stream=stream,
                                                          out=self.tlm_outs[k])
        self.tlm_out = self.tlm_outs[k]
        self.track_ms[2] = 1e3 * (time.perf_counter() - t0)

    def fov_positions(self, f=0):
        """the LiDAR positions lasermap_fov_segment looks at, [F, 3]"""
        return np.ascontiguousarray(self.frame_set(f).states[:, 9:12])

    def lidar_step(self, raw=None, f=0):
        pkg, F = self.pkg, self.F
        fs = self.frame_set(f)
        raw = fs.dev_raw if raw is None else raw
        # lasermap_fov_segment (host logic) + the box deletions it asks for, per sequence
        # (one call for all sequences: 512 calls through ctypes were 2.5 ms of interpreter time per step on the stage thread, more under the
        # other stage threads' contention for the interpreter lock)
        boxes_all, n_boxes = pkg.capi.lidar_fov_segment_batch(self.boxes, self.fov_positions(f), cube_len=1000.0, det_range=100.0)
        todo = np.nonzero(n_boxes)[0]
        todo_maps, todo_boxes = [self.maps[s] for s in todo], [boxes_all[s, :n_boxes[s]].copy() for s in todo]
        if todo_maps:
            pkg.capi.delete_point_boxes_batch(todo_maps, todo_boxes, stream=self.lidar_stream.cuda_stream)
        self.lidar_counts = self.lidar.frontend_batch(raw.data_ptr(), fs.raw_offs, self.maps, fs.states, stream=self.lidar_stream.cuda_stream,
                                                      want_points=False)[0]
        self.lidar_times.append(self.lidar.last_timings().astype(float))
        # UpdateMap -> map_incremental (Tracking.cc:1602-1603) for every sequence's map: one batched call
        na, nn, _ = pkg.capi.map_incremental_batch(self.lidar, self.scan_ids, self.maps, fs.states, stream=self.lidar_stream.cuda_stream)
        self.map_adds = [int(na.sum()), int(nn.sum())]

    def ba_stats_batch(self):
        """The batch whose statistics describe the windows the loop has optimised."""
        if self.ba_ring:
            return self.ba_ring[0]
        if self.ba_step_workers:
            return self.ba_step_workers[0]
        if not self.ba_workers:
            return self.ba_batch
        ran = [b for w in self.ba_workers for b in w.values() if b.stats[0].iterations > 0]  # (a worker may not have met every chunk size)
        return ran[0] if ran else next(iter(self.ba_workers[0].values()))

    def ba_run_batches(self):
        """Every BaBatch of the loop that has run at least once."""
        all_b = [self.ba_batch, self.ba_batch2, self.ba_batch4] + [b for w in self.ba_workers for b in w.values()] + list(self.ba_ring) + list(self.ba_step_workers)
        return [b for b in all_b if b is not None and any(b.stats[i].iterations > 0 or b.results[i] > 0 for i in range(b.n))]

    def ba_mix_summary(self, wl):
        """What the optimised windows were and did: per distinct window of the workload's list (as last run) its sizes, iterations and trials;
        the sums the roofline's byte / FLOP counts are made of (weighted by the linearisations a window ran)."""
        seen = {}
        for b in self.ba_run_batches():
            if not hasattr(b, "window_ids"):   # (the inertial loop's LocalLVIBA batches: its windows are described by algorithmic_work_inertial)
                continue
            for i, wid in enumerate(b.window_ids):
                st, ls = b.stats[i], b.lstats[i]
                seen[wid] = (int(st.iterations), int(st.trials), int(st.n_free_poses), int(ls.n_planes))
        if not seen:
            return None
        rows = []
        for wid, (it, tr, nf, planes) in sorted(seen.items()):
            w = wl.ba_windows[wid]
            e6 = np.asarray(w["edges6"])
            free_edge = np.asarray(w["fixed"])[e6[:, 1].astype(int)] == 0
            f_l = np.bincount(e6[free_edge, 0].astype(int), minlength=len(w["points"]))
            rows.append(dict(edges=len(e6), points=len(w["points"]), free=nf, planes=planes, win=len(w["win_pose"]), cloud_points=int(sum(len(c) for c in w["clouds"])),
                             free_edges=int(free_edge.sum()), pose_pairs=int((f_l * (f_l + 1) // 2).sum()), it=it, tr=tr))
        it = np.array([r["it"] for r in rows], float)
        tr = np.array([r["tr"] for r in rows], float)
        wgt = np.maximum(it, 1e-9) / max(it.sum(), 1e-9)

        def mean_w(key):
            return float(np.sum(wgt * np.array([r[key] for r in rows], float)))
        agg = {k: mean_w(k) for k in ("edges", "points", "free", "planes", "win", "cloud_points", "free_edges", "pose_pairs")}
        for k in ("edges", "points", "free_edges", "pose_pairs", "planes", "cloud_points"):
            agg[k] = int(round(agg[k]))
        agg["free"] = max(1, int(round(agg["free"]))); agg["win"] = max(1, int(round(agg["win"])))
        agg.update(linearisations=float(it.mean()), trials=float(tr.mean()), distinct_windows=len(rows),
                   iterations_min_mean_max=[int(it.min()), round(float(it.mean()), 2), int(it.max())], trials_min_mean_max=[int(tr.min()), round(float(tr.mean()), 2), int(tr.max())],
                   free_keyframes_min_max=[int(min(r["free"] for r in rows)), int(max(r["free"] for r in rows))],
                   points_min_max=[int(min(r["points"] for r in rows)), int(max(r["points"] for r in rows))],
                   edges_min_max=[int(min(r["edges"] for r in rows)), int(max(r["edges"] for r in rows))],
                   windows_without_lidar_edge=int(sum(r["win"] == 0 for r in rows)), windows_with_rejected_steps=int(sum(r["tr"] > r["it"] for r in rows)),
                   windows_interrupted=int(sum(r["it"] < 10 for r in rows)))
        return agg

    def ba_step(self, m=1):
        """The local-mapping work of m (1, 2 or 4) steps."""
        self.ba_due += m * self.ba_rate
        k = int(self.ba_due + 1e-9)
        if k < 1:
            return
        self.ba_due -= k
        batch, n = {1: (self.ba_batch, self.n_ba), 2: (self.ba_batch2, 2 * self.n_ba), 4: (self.ba_batch4, 4 * self.n_ba)}[m]
        if batch.run(self.args.ba_concurrency) != n:
            raise RuntimeError("a local BA window failed")
        self.ba_windows_done += n

    # -- the loop ------------------------------------------------------------------------------------------------------------
    def run(self, n_steps, stages=("orb", "track", "lidar", "ba")):
        """Every stage thread processes n_steps batches; ORB extraction and tracking form a two-deep pipeline over two feature buffers.
        A failure in any stage stops all of them (no thread is left blocked on a queue) and is re-raised."""
        torch = self.torch
        self.steps_tracked = 0
        free, ready, ready2 = queue.Queue(), queue.Queue(), queue.Queue()
        for k in range(3):
            free.put(k)
        failed = threading.Event()
        errors = []

        def get(q):
            while not failed.is_set():
                try:
                    return q.get(timeout=0.2)
                except queue.Empty:
                    continue
            return None

        fed = getattr(self, "host_fed", False)
        K_frames, base = (1 if fed else len(self.frame_sets)), self.step_base   # (the host-fed leg feeds frame 0: its pinned copies are of that frame)
        self.step_base += n_steps

        def orb_thread():
            ev = self._upload(0, 0) if fed else None
            for i in range(n_steps):
                k = get(free)
                if k is None:
                    return
                img = None
                if fed:  # this step's images have been on their way since the step before; the next step's start now, into the other buffer
                    ev.synchronize()
                    img = self.img_bufs[i % 2]
                    if i + 1 < n_steps:
                        ev = self._upload(0, (i + 1) % 2)
                self.extract(k, self.stream, img, (base + i) % K_frames)
                self.orb_times.append(self.exts[k].last_timings().astype(float))
                ready.put(k)

        def track_thread():   # the tracking thread, first half: stereo matching + TrackWithMotionModel
            for _ in range(n_steps):
                k = get(ready)
                if k is None:
                    return
                self.track(k, self.track_stream.cuda_stream)
                ready2.put(k)

        def track2_thread():  # the tracking thread, second half: TrackLocalMap (a pipeline stage of its own when sequences are batched)
            for _ in range(n_steps):
                k = get(ready2)
                if k is None:
                    return
                self.track_local(k, self.track2_stream.cuda_stream)
                self.steps_tracked += 1
                free.put(k)

        # configs[3]: the scans of step k + 1 are preprocessed and their time-sort order found (a second front-end handle, an own stream) while
        # step k's iterated filter runs -- the reference preprocesses in the scan callback, ahead of the thread that consumes lidar_buffer
        piped = hasattr(self, "lidar_prepare") and not getattr(self.args, "no_lidar_prepare", False)
        lfree, lready = queue.Queue(), queue.Queue()
        for h in range(2):
            lfree.put(h)

        def lidar_prep_thread():
            for _ in range(n_steps):
                h = get(lfree)
                if h is None:
                    return
                self.lidar_prepare(h)
                lready.put(h)

        def lidar_thread():
            ev = self._upload(1, 0) if fed else None
            for i in range(n_steps):
                if failed.is_set():
                    return
                if piped:
                    h = get(lready)
                    if h is None:
                        return
                    self.lidar_step(handle=h)
                    lfree.put(h)
                    continue
                raw = None
                if fed:
                    ev.synchronize()
                    raw = self.raw_bufs[i % 2]
                    if i + 1 < n_steps:
                        ev = self._upload(1, (i + 1) % 2)
                if raw is not None:
                    self.lidar_step(raw)
                elif K_frames > 1:
                    self.lidar_step(f=(base + i) % K_frames)
                else:
                    self.lidar_step()

        def ba_thread():
            done = 0  # steps whose windows have been optimised; a step's keyframes exist once the tracking thread has finished it
            # (with a full GPU batch per step -- no second batch -- the thread runs free as before: nothing to gain from waiting)
            follows_tracking = "track" in set(stages) and self.ba_batch2 is not None
            while done < n_steps and not failed.is_set():
                ready = (self.steps_tracked if follows_tracking else n_steps) - done
                # the keyframes of the steps tracked so far, at most two steps' per call: one step's when the thread has caught up (so that the
                # run does not end on a double batch that could only start when the last frame had been tracked), two when it lags
                if ready < 1:
                    time.sleep(0.0002)
                    continue
                m = (4 if ready >= 4 and self.ba_batch4 is not None else min(2, ready)) if follows_tracking else 1
                self.ba_step(m)
                done += m

        def ba_engine_thread():
            """Every window (r, i) of the ring is a sequence's mapping thread: its window of step j = r, r + R, ... is submitted (a ticket of its own)
            when the tracking thread has finished step j AND the sequence's previous window has come back -- a LocalMapping thread runs one
            optimisation at a time; nobody waits for another sequence's window."""
            follows_tracking = "track" in set(stages) and F_follow
            R, K, SENT = len(self.ba_ring), len(self.ba_engines), -(1 << 30)
            nxt = [np.full(b.n, r, np.int64) for r, b in enumerate(self.ba_ring)]      # the step of the window's next submission
            ticket = [np.zeros(b.n, np.int64) for b in self.ba_ring]                   # > 0: in flight
            left = sum(len(range(r, n_steps, R)) * b.n for r, b in enumerate(self.ba_ring))
            dbg = [] if os.environ.get("TC2LI_BENCH_BA_ENGINE_DEBUG") else None
            t_sub = [np.zeros(b.n) for b in self.ba_ring]
            t_tracked = {}
            while left > 0 and not failed.is_set():
                tracked = min(n_steps, self.steps_tracked if follows_tracking else n_steps)
                if dbg is not None and tracked not in t_tracked:
                    t_tracked[tracked] = time.perf_counter()
                progressed = False
                for r, b in enumerate(self.ba_ring):
                    back = np.nonzero((ticket[r] > 0) & (b.results != SENT))[0]
                    for i in back:
                        if self.ba_engines[i % K].wait(int(ticket[r][i])) != 1:
                            raise RuntimeError("a local BA window failed")
                        ticket[r][i] = 0
                        self.ba_windows_done += 1
                        left -= 1
                        if dbg is not None:
                            dbg.append((time.perf_counter() - t_sub[r][i], int(b.stats[i].trials), int(nxt[r][i] - R)))
                    for i in np.nonzero((ticket[r] == 0) & (nxt[r] < tracked))[0]:
                        b.results[i] = SENT
                        ticket[r][i] = self.ba_engines[i % K].submit(b, int(i), 1)
                        t_sub[r][i] = time.perf_counter()
                        nxt[r][i] += R
                    progressed |= len(back) > 0
                if not progressed:
                    time.sleep(0.0002)
            if dbg:
                lat = np.array([d[0] for d in dbg]) * 1e3
                tr = np.array([d[1] for d in dbg])
                last = max(t_tracked.values())
                print("engine debug: %d windows, latency ms mean %.2f p50 %.2f p90 %.2f max %.2f; per trial %.3f; end lag after last tracked step %.2f ms" % (
                    len(lat), lat.mean(), np.median(lat), np.percentile(lat, 90), lat.max(), (lat / np.maximum(tr, 1)).mean(), 1e3 * (time.perf_counter() - last)), file=sys.stderr)
        ba_engine_thread.__name__ = "ba_thread"
        F_follow = self.F <= 256

        ba_jobs = {"next": 0}
        ba_lock = threading.Lock()

        def make_ba_worker(w):
            def fn():
                n_chunks = len(self.ba_chunk_sizes)
                while not failed.is_set():
                    with ba_lock:
                        j = ba_jobs["next"]
                        ba_jobs["next"] += 1
                    if j >= n_steps * n_chunks:
                        return
                    b = self.ba_workers[w][self.ba_chunk_sizes[j % n_chunks]]
                    if b.run_group(w) != b.n:
                        raise RuntimeError("a local BA window failed")
                    with ba_lock:
                        self.ba_windows_done += b.n
            fn.__name__ = "ba_worker%d_thread" % w
            return fn

        def make_ba_step_worker(w):
            def fn():
                follows_tracking = "track" in set(stages)
                if not self.ba_step_prewarmed[w]:
                    self.ba_step_prewarmed[w] = True
                    if self.ba_step_workers[w].run_group(w) != self.ba_step_workers[w].n:
                        raise RuntimeError("a local BA window failed")
                while not failed.is_set():
                    with ba_lock:
                        j = ba_jobs["next"]
                        ba_jobs["next"] += 1
                    if j >= n_steps:
                        return
                    while follows_tracking and self.steps_tracked <= j and not failed.is_set():
                        time.sleep(0.0002)
                    b = self.ba_step_workers[w]
                    if b.run_group(w) != b.n:
                        raise RuntimeError("a local BA window failed")
                    with ba_lock:
                        self.ba_windows_done += b.n
            fn.__name__ = "ba_worker%d_thread" % w
            return fn

        want = set(stages)
        if "track" in want:
            want.add("orb")  # tracking consumes what the extraction produces
        fns = [f for f in (orb_thread, track_thread, track2_thread, lidar_thread, ba_thread) + ((lidar_prep_thread,) if piped else ())
               if f.__name__.split("_")[0].rstrip("2") in want]
        if not self.ba_batch:
            fns = [f for f in fns if f is not ba_thread]
        elif self.ba_engine is not None and ba_thread in fns:
            fns = [f for f in fns if f is not ba_thread] + [ba_engine_thread]
        elif self.ba_step_workers and ba_thread in fns:
            fns = [f for f in fns if f is not ba_thread] + [make_ba_step_worker(w) for w in range(len(self.ba_step_workers))]
        elif self.ba_workers and ba_thread in fns:
            fns = [f for f in fns if f is not ba_thread] + [make_ba_worker(w) for w in range(len(self.ba_workers))]
        if "track" not in want and "orb" in want:  # nobody returns the feature buffers: the extraction thread recycles them itself
            for _ in range(n_steps):
                free.put(0)

        def wrap(fn):
            def wrapped():
                try:
                    t = time.perf_counter()
                    fn()
                    self.thread_ms[fn.__name__] = 1e3 * (time.perf_counter() - t) / max(n_steps, 1)
                except BaseException as e:  # noqa: BLE001
                    errors.append(e)
                    failed.set()
            return wrapped
        # the stage threads persist over the Loop's lifetime (one per stage function name)
        done = []
        for fn in fns:
            if fn.__name__ not in self.workers:
                self.workers[fn.__name__] = StageWorker(fn.__name__, self.local_rank, torch)
            done.append(self.workers[fn.__name__].submit(wrap(fn)))
        for d in done:
            d.wait()
        if errors:
            raise errors[0]
        if (self.ba_workers or self.ba_step_workers) and any(k.startswith("ba_worker") for k in self.thread_ms):
            self.thread_ms["ba_thread"] = max(v for k, v in self.thread_ms.items() if k.startswith("ba_worker"))  # local mapping is done when its last worker is

    def close(self):
        for w in self.workers.values():
            w.q.put(None)
        for w in self.workers.values():
            w.t.join()
        self.workers = {}
        if self.ba_engine is not None:
            for e in self.ba_engines:
                e.close()
            self.ba_engine, self.ba_engines = None, []
        _OPEN_LOOPS.discard(self)


# ---------------------------------------------------------------------------------------------------------------------------------
# configs[3]: the camera-LiDAR-inertial loop (tc2li): IMU pre-integration, pose-inertial optimisation in TrackLocalMap, scan motion
# compensation + iterated ESKF in the LiDAR thread, LocalLVIBA in local mapping
# ---------------------------------------------------------------------------------------------------------------------------------
class InertialLoop(Loop):
    """F sequences of the inertial configuration (BASELINE configs[3]) on the stage threads of Loop.  Camera / tracking thread: ORB extraction and stereo
    matching as in the main loop; with the IMU initialised TrackWithMotionModel is the IMU prediction alone and TrackLocalMap is SearchLocalPoints
    (tc2li_search_local_points_batch), the IMU pre-integration between the frames (tc2li_imu_preintegrate_frames, host) and
    PoseInertialOptimizationLastFrame for all frames (tc2li_pose_inertial_optimization_batch) -- Tracking.cc:2746-2752, 2857-2878.  LiDAR thread: lasermap_fov_segment + box deletions, then LidarInertialProcess for all scans in one call
    (tc2li_lidar_inertial_frontend_batch: preprocess, forward propagation, UndistortPcl with its time sort, voxel filter, iterated ESKF in lock
    step) and map_incremental for all maps.  Local mapping: F / kf_interval LocalLVIBA windows per step in lock step
    (tc2li_local_lvi_bundle_adjustment_batch: 10 optimisable keyframes + the fixed one, LiDAR edge over 6 keyframes x 2400 points)."""

    def __init__(self, wl, seq_ids, args, local_rank):
        from scipy.spatial.transform import Rotation
        super().__init__(wl, seq_ids, argparse.Namespace(**dict(vars(args), front_end_only=True)), local_rank)
        self.args = args
        pkg, synthetic, F, U = wl.pkg, wl.synthetic, self.F, wl.U
        # pose-inertial problems (previous-frame form): the held map points of every frame; U distinct ones, tiled over the sequences
        self.uniq_pi = [synthetic.pose_inertial_problem(50 + u, n_points=2000, last_frame=True) for u in range(U)]
        packed = [dict(q, edges=pkg.pack_ba_edges(q["edges"])) for q in self.uniq_pi]
        self.pi = pkg.capi.PoseInertialBatch([packed[t] for t in self.tile], self.uniq_pi[0]["calib24"], self.uniq_pi[0]["cam"], synthetic.IMU_NOISE)
        # LiDAR-inertial: per sequence the IMU samples over the sweep, the filter state at the last scan end and its covariance
        self.uniq_li = []
        for u in range(U):
            R, p = synthetic.sensor_pose(u + 1)
            rng = np.random.default_rng(300 + u)
            beg, end = 10.0, 10.1
            k0, k1 = int(np.floor((beg - 0.012) * 100)), int(np.ceil((end + 0.004) * 100))
            ts = np.arange(k0, k1 + 1) / 100.0
            imu = np.zeros((len(ts), 7))
            imu[:, 0] = ts
            imu[:, 1:4] = np.array([0.0, 0.0, 9.81]) + rng.normal(0, 0.02, (len(ts), 3))
            imu[:, 4:7] = rng.normal(0, 0.002, (len(ts), 3))
            x = np.concatenate([p + rng.normal(0, 0.02, 3), (R @ Rotation.from_rotvec(rng.normal(0, 0.002, 3)).as_matrix()).ravel(), [10.0, 0.0, 0.0], np.zeros(3),
                                np.zeros(3), [0, 0, -9.81], np.eye(3).ravel(), np.zeros(3)])
            A = rng.normal(0, 1, (23, 23))
            P = A @ A.T * 1e-6 + np.diag([1e-3] * 3 + [1e-4] * 3 + [1e-5] * 6 + [1e-2] * 3 + [1e-5] * 6 + [1e-6] * 2)
            self.uniq_li.append(dict(imu=imu, x=x, P=P, times=[beg, end, beg - 0.001, 1.0]))
        self.cov12 = np.array([0.1] * 3 + [0.1] * 3 + [1e-4] * 3 + [1e-4] * 3)
        li = [self.uniq_li[t] for t in self.tile]
        self.li = pkg.capi.LidarInertialBatch(self.lidar, self.raw_offs, self.maps, np.stack([q["x"] for q in li]), np.stack([q["P"] for q in li]),
                                              [q["imu"] for q in li], np.array([q["times"] for q in li]), self.cov12, max_iter=3)
        # local mapping: LVIBA windows.  LocalMapping.cc:156-161: bLarge = GetMatchesInliers() > 100 for a stereo rig -- the loop's frames track ~1000
        # inliers, so the reference calls LocalLVIBA with bLarge: 25 optimisable keyframes, 4 iterations from lambda 1e-2
        # (OptimizerWithLidar.cc:493-500, 618-623).  --lviba-small: the 10-keyframe / 10-iteration / lambda 1 window of a weakly tracked frame
        # (what rounds 3-4 benched here).
        self.lvi_large = not getattr(args, "lviba_small", False)
        self.lvi_iterations, self.lvi_lambda = (4, 1e-2) if self.lvi_large else (10, 1.0)
        self.uniq_lvi = []
        for k in range(4):
            w = synthetic.inertial_window(100 + k, n_opt=25, n_points=1500) if self.lvi_large else synthetic.inertial_window(k, n_opt=10, n_points=900)
            pre = []
            for smp, t1, t2 in w["samples"]:
                q = pkg.capi.Preintegrated(w["bias6"], *synthetic.IMU_NOISE)
                q.preintegrate(smp, t1, t2)
                pre.append(q)
            K = len(w["kf33"])
            win = list(range(K - 1, K - 7, -1))
            self.uniq_lvi.append(dict(w, pre=pre, win_kf=win, clouds=synthetic.inertial_window_clouds(w, win, n_points=2400, seed=k), packed=pkg.pack_ba_edges(w["edges"]),
                                      Tcl7=synthetic.TCL7, Tbl7=synthetic.tbl7(), weight=1.0))
        self.ba_rate = 0.0 if args.front_end_only else F / args.kf_interval
        self.n_ba = int(np.ceil(self.ba_rate)) if self.ba_rate else 0
        if self.n_ba:
            wins = [self.uniq_lvi[k % 4] for k in range(self.n_ba)]
            self.ba_batch = pkg.capi.LviBatch([dict(kf33=w["kf33"], fixed=w["fixed"], has_imu=w["has_imu"], points=w["points"], edges=w["packed"], link4=w["link4"],
                                                    pre=w["pre"], win_kf=w["win_kf"], clouds=w["clouds"], Tcl7=w["Tcl7"], Tbl7=w["Tbl7"], weight=1.0,
                                                    iterations=self.lvi_iterations, lambda_init=self.lvi_lambda) for w in wins],
                                              self.uniq_lvi[0]["calib24"], self.uniq_lvi[0]["cam"])
        self.ba_batch2 = None
        self.slp_outs = [None, None, None]
        # mapping workers as in the main loop (TC2LI_BENCH_BA_WORKERS): each a lock-step group of its own on tc2li_local_lvi_bundle_adjustment_batch_group
        n_workers = int(os.environ.get("TC2LI_BENCH_BA_WORKERS", "3"))
        if self.n_ba and F > 256 and n_workers >= 2 and self.n_ba >= 8 * n_workers and self.ba_rate == self.n_ba:
            self.ba_chunk_sizes = [self.n_ba * (c + 1) // n_workers - self.n_ba * c // n_workers for c in range(n_workers)]

            def lvi_batch(n):
                return pkg.capi.LviBatch([dict(kf33=w["kf33"], fixed=w["fixed"], has_imu=w["has_imu"], points=w["points"], edges=w["packed"], link4=w["link4"],
                                               pre=w["pre"], win_kf=w["win_kf"], clouds=w["clouds"], Tcl7=w["Tcl7"], Tbl7=w["Tbl7"], weight=1.0,
                                               iterations=self.lvi_iterations, lambda_init=self.lvi_lambda) for w in [self.uniq_lvi[k % 4] for k in range(n)]],
                                         self.uniq_lvi[0]["calib24"], self.uniq_lvi[0]["cam"])
            for w_ in range(n_workers):
                self.ba_workers.append({n: lvi_batch(n) for n in set(self.ba_chunk_sizes)})
            # (step workers as in the main loop -- a step's windows one lock-step group, four steps in flight -- measured here: 15.3 k against 16.0-16.3 k
            # frames/s; this loop waits for its LiDAR thread, 31-33 ms per step, whichever way local mapping runs)

    # The camera path with the IMU initialised (the steady state of configs[3]): TrackWithMotionModel is PredictStateIMU() and nothing else
    # (Tracking.cc:2746-2752: no search against the last frame, no PoseOptimization), and TrackLocalMap optimises with
    # PoseInertialOptimizationLastFrame / LastKeyFrame INSTEAD of PoseOptimization (Tracking.cc:2857-2878).  Rounds 3-4 and the first part of
    # round 5 ran the main loop's TrackWithMotionModel and TrackLocalMap (both with their visual PoseOptimization) and the pose-inertial
    # optimisation on top: more than the reference does per frame.
    def track(self, k, stream):
        """Stereo matching of feature buffer k; the motion-model step is the IMU prediction (host, from the pre-integration since the last frame)."""
        wl, pkg, F = self.wl, self.pkg, self.F
        t0 = time.perf_counter()
        self.st_outs[k] = pkg.stereo_match_batch(self.exts[k], F, float(wl.bf), float(wl.b), stream=stream, out=self.st_outs[k])
        self.track_ms[0], self.track_ms[1] = 1e3 * (time.perf_counter() - t0), 0.0

    def track_local(self, k, stream):
        """TrackLocalMap: SearchLocalPoints from the predicted pose, IMU pre-integration between the frames, PoseInertialOptimizationLastFrame."""
        wl, pkg, F = self.wl, self.pkg, self.F
        t0 = time.perf_counter()
        self.slp_outs[k] = pkg.capi.search_local_points_batch(self.exts[k], F, self.orb_outs[k][0], self.st_outs[k][0], self.pose_pred, self.held, self.held_Xw,
                                                              self.local_pts, self.local_off, wl.cam5, th=1.0, stream=stream, out=self.slp_outs[k])
        self.pi.preintegrate()
        self.pi.run(stream)
        self.track_ms[2] = 1e3 * (time.perf_counter() - t0)

    def lidar_prepare(self, handle):
        """Preprocess::process + the order of UndistortPcl's time sort of the next step's scans, into front-end handle `handle` (0 / 1)."""
        if self.lidar_handles is None:
            self.lidar_handles = [self.lidar, self.pkg.LidarFrontEnd(max_points_per_scan=self.lidar_cap, max_scans=self.F)]
            self.lidar_prep_stream = self.torch.cuda.Stream()
        self.li.prepare(self.dev_raw.data_ptr(), stream=self.lidar_prep_stream.cuda_stream, fe=self.lidar_handles[handle])

    def lidar_step(self, handle=None):
        pkg, F = self.pkg, self.F
        # (one call for all sequences: 512 calls through ctypes were 2.5 ms of interpreter time per step on the stage thread, more under the
        # other stage threads' contention for the interpreter lock)
        boxes_all, n_boxes = pkg.capi.lidar_fov_segment_batch(self.boxes, self.fov_positions(), cube_len=1000.0, det_range=100.0)
        todo = np.nonzero(n_boxes)[0]
        todo_maps, todo_boxes = [self.maps[s] for s in todo], [boxes_all[s, :n_boxes[s]].copy() for s in todo]
        if todo_maps:
            pkg.capi.delete_point_boxes_batch(todo_maps, todo_boxes, stream=self.lidar_stream.cuda_stream)
        fe = self.lidar if handle is None else self.lidar_handles[handle]
        self.li.run(None if handle is not None else self.dev_raw.data_ptr(), stream=self.lidar_stream.cuda_stream, fe=fe)
        x = self.li.states36()
        st24 = np.concatenate([x[:, 3:12], x[:, 0:3], x[:, 24:33], x[:, 33:36]], 1)
        na, nn, _ = pkg.capi.map_incremental_batch(fe, self.scan_ids, self.maps, st24, stream=self.lidar_stream.cuda_stream)
        self.map_adds = [int(na.sum()), int(nn.sum())]

    def inertial_stats(self):
        ls = np.array(self.li.stats())
        return {"scan_points_preprocessed/downsampled/eskf_features/h_share_model_calls": [int(ls[:, 3].mean()), int(ls[:, 4].mean()), int(ls[:, 2].mean()), int(ls[:, 0].mean())],
                "pose_inertial_edges/inliers": [int(np.mean([len(q["edges"]) for q in self.uniq_pi])), int(self.pi.inliers().mean())],
                "lviba_iterations/planes": [int(self.ba_stats_batch().stats[0].iterations), int(self.ba_stats_batch().lstats[0].n_planes)] if self.n_ba else None,
                "map_points_added_per_step": self.map_adds}


# ---------------------------------------------------------------------------------------------------------------------------------
# roofline: algorithmic bytes (or FLOPs) of one step per kernel (SURVEY.md section 8d figures x the units a step processes)
# ---------------------------------------------------------------------------------------------------------------------------------
def algorithmic_work(wl, loop, nkp, lid, ba):
    """{kernel name: (bytes or flops per STEP of `loop`, 'B' | 'FLOP')}.  lid = mean (raw, preprocessed, down-sampled, selected) points per scan;
    ba = dict(edges, points, free, planes, win, windows, linearisations, trials) per step."""
    px = [a * b for a, b in level_dims(wl.W, wl.H)]
    n_img, F = loop.n_img, loop.F
    raw, pre, down, sel = lid
    mp = loop.map_points0
    nloc = float(np.mean(np.diff(loop.local_off))) if getattr(loop, "local_off", None) is not None else 0.0
    w = {
        "k_resize_linear": (n_img * (sum(px[:-1]) + sum(px[1:])), "B"),                # read levels 0..6, write levels 1..7
        "k_resize_strips": (n_img * (sum(px[:-1]) + sum(px[1:])), "B"),                # the same resize as column strips (round 3)
        "k_resize_tail": (n_img * (sum(px[2:-1]) + sum(px[3:])), "B"),                  # round 4: levels 3..7 in one launch, a workgroup per image
        "k_fast_cells": (n_img * (sum(px) + 4 * 15000), "B"),                          # every level once + the candidate list
        "k_blur7_strips": (n_img * 2 * sum(px), "B"),
        "k_orient_describe": (n_img * nkp * (709 + 512 + 64 + 16), "B"),                # patch gathers + descriptor / angle / key out
        "k_compact_cells": (n_img * 120e3, "B"),
        "k_stereo_match": (F * nkp * 40 * 64, "B"),                                     # ~40 candidate pairs per left key, 64 B per pair
        "k_pre_stream": (F * (raw * 32 + pre * 48), "B"),                               # round 4: one pass, the raw scan read once, the kept points written
        "k_pre_count": (F * raw * 32, "B"), "k_pre_scatter": (F * (raw * 32 + pre * 48), "B"),   # SURVEY 8d counts the raw scan once: 7.3 MB per scan in all
        "k_voxel_bbox": (F * pre * 48, "B"), "k_voxel_insert": (F * pre * 48, "B"), "k_voxel_fill": (F * pre * 8, "B"),
        "k_voxel_rank": (F * pre * (48 + 32), "B"), "k_voxel_centroid": (F * (pre * 32 + down * 48), "B"),
        "k_knn_plane": (F * down * (48 + 48 + 48 + 5 * 8 + 27 * 16), "B"),
        "k_knn_hard": (F * down * 0.02 * (48 + 48 + 48 + 5 * 8 + 125 * 16), "B"),
        "k_sel_scatter": (F * (down + sel * 4 * 48), "B"),
        "k_map_keep_scatter": (F * mp * 96, "B"), "k_map_count": (F * mp * 52, "B"), "k_map_scatter": (F * mp * (48 + 16 + 8), "B"),
        "k_map_keep_count": (F * mp, "B"),
        # keypoint distribution: every candidate once (4 B) + the picks; the rounds of the walk re-read keys that stay in L2
        "k_quadtree": (n_img * (4 * 15000 + 8 * nkp), "B"),
        "k_quadtree_sorted": (n_img * (4 * 15000 + 8 * nkp), "B"),                      # round 3: keys sorted once by their path, everything in LDS
        "k_quadtree_sorted_list": (n_img * (4 * 15000 + 8 * nkp), "B"),                 # round 4: the same jobs from per-class lists, resident workgroups
        # voxel filter, sorted form: xyz in, (voxel, point) out; the gather reads a point and writes its 32-byte record; the sums read the records
        "k_voxel_sort_points": (F * pre * (16 + 8), "B"), "k_voxel_gather_sorted": (F * pre * (4 + 48 + 32), "B"),
        "k_voxel_centroid_sorted": (F * (pre * 32 + down * 48), "B"), "k_voxel_centroid_fused": (F * (pre * (4 + 48) + down * 48), "B"),
        "k_stereo_rows": (F * nkp * (12 + 2 * 7), "B"),                                 # right keys in, ~7 row entries of 2 B each out
        # tracking: TrackWithMotionModel + TrackLocalMap per frame -- queries (64 B out, source point in), candidate windows, edges
        "k_track_queries_last": (F * nkp * (53 + 64), "B"), "k_track_queries_local": (F * nloc * (68 + 64), "B"),
        "k_match_candidates": (F * (nkp + nloc) * (64 + 20 * (32 + 12 + 4)), "B"),      # the query + ~20 window candidates (descriptor, key, pool entry)
        "k_match_resolve": (F * (nkp + nloc) * (8 + 20 * 4), "B"), "k_match_grid": (2 * F * nkp * (12 + 2), "B"),
        "k_track_edges_last": (F * nkp * (4 + 12 + 4 + 12 + 40 + 24 + 4), "B"), "k_track_edges_local": (F * nkp * (4 + 12 + 4 + 12 + 40 + 24 + 4), "B"),
        # Optimizer::PoseOptimization, two calls per frame: every correspondence read once (edge 40 B + point 24 B), outlier flag and chi2 out;
        # the 4 x 10 Gauss-Newton passes over them are meant to stay on chip
        "k_pose_optimization": (2 * F * nkp * 0.5 * (40 + 24 + 1 + 8), "B"),
        "k_pose_optimization_lds": (2 * F * nkp * 0.5 * (40 + 24 + 1 + 8), "B"),       # the same with the correspondences staged in LDS (round 3)
        "k_map_holes": (F * mp * (1 + 4), "B"),                                          # compaction in place: every point's deleted flag in, its hole / mover rank out
        # round 5: the compaction from the step's deletion list (~1 600 per map): the list in, a flag back per entry, the moved points (48 B in and out)
        "k_map_compact_list": (F * 1600 * (4 + 1 + 0.5 * 96), "B"),
        "k_mapinc_apply": (F * sel * (48 + 16 + 48), "B"),
    }
    if ba:
        E, P, lin, tr, nw = ba["edges"], ba["points"], ba["linearisations"], ba["trials"], ba["windows"]
        Ef, pairs, nf = ba["free_edges"], ba["pose_pairs"], ba["free"]    # edges with a free pose; sum over landmarks of f (f + 1) / 2
        blocks, slices = -(-Ef // 256), max(-(-Ef // 256), -(-P // 64))   # 256-edge blocks of the pose role; slices of the Schur kernel
        lower = 6 * nf * (6 * nf + 1) // 2
        w.update({
            # all edges once: edge 40 B + its place in the landmark-major list 4 B in, chi2 / rho 16 B out; per landmark the point in, Hll / b_l out
            # (round 4: one launch for both roles) + free-pose edges: edge + slot in, W (144 B) out; one 27-vector per (block, pose) out
            "k_ba_linearize_b": (nw * lin * (E * 60 + P * (24 + 80) + Ef * (40 + 4 + 144) + blocks * nf * 224), "B"),
            "k_ba_linearize_imu_b": (nw * lin * (E * 60 + P * (24 + 80) + Ef * (40 + 4 + 144) + blocks * nf * 224), "B"),  # the same over ImuCamPose vertices (configs[3])
            "k_ba_reduce_all_b": (nw * lin * (blocks * nf * 224 + nf * 216), "B"),
            # S -= W D^-1 W^T: SURVEY 8d prices it per landmark with n (free-pose) observations at n (n + 1) / 2 x (6x3 . 3x3 + 6x3 . 3x6) =
            # n (n + 1) / 2 x 324 FLOP.  (The kernel runs it as a zero-padded dense f64 MFMA product per chunk of landmarks: about nine times
            # these FLOPs at this covisibility, 55 % of the measured matrix peak when it has the GPU to itself -- DESIGN.md section 4.)
            # the same products in 128-slot slices (the default since round 4).  Priced in BYTES since round 6: per slot the W block (144 B) and its
            # three indices, per landmark Hll + b_l (72 B), per part of 8 slices the lower triangle of S and the coefficient row out -- at ~5 FLOP
            # per byte the kernel sits below the f64 vector unit's ridge (66 TFLOP/s over 6.3 TB/s = 10.5): HBM is the roof that bounds it
            "k_ba_schur_lean_b": (nw * tr * (Ef * (144 + 12) + P * 72 + max(1, -(-slices // 8)) * (lower + 6 * nf) * 8), "B"),
            "k_ba_schur_finish_b": (nw * tr * (slices + 1) * lower * 8, "B"),
            # round 6, the reduced system's LDL^T on the device (one workgroup per window and trial): the lower triangles of S and of the LiDAR
            # block and the two right-hand sides in, the step out -- the three variants (5 / 8 / 9 tile rows) read the same bytes
            "k_ba_solve5_b": (nw * tr * (2 * lower + 3 * 6 * nf) * 8, "B"), "k_ba_solve_b": (nw * tr * (2 * lower + 3 * 6 * nf) * 8, "B"),
            "k_ba_solve9_b": (nw * tr * (2 * lower + 3 * 6 * nf) * 8, "B"),
            # the LM bookkeeping on the device: per linearisation the plane term's (6W)^2 Hessian + gradient in and its camera-se3 form out (twice:
            # the kept copy and the window's (6K)^2 block entries), per trial the step and b_p in, the 128-byte state in and out (+ its host mirror)
            "k_ba_lm_begin_b": (nw * lin * ((36 * ba["win"] ** 2 + 18 * ba["win"]) * 8 * 3 + 256), "B"),
            "k_ba_lm_decide_b": (nw * tr * (3 * 6 * nf * 8 + 3 * 128), "B"),
            "k_ba_trial_update_b": (nw * tr * (Ef * (144 + 4) + P * (48 + 24 + 24 + 24)), "B"),
            "k_ba_errors_b": (nw * tr * E * 112, "B"),
            "k_ba_errors_reduce_b": (nw * tr * E * 112, "B"),  # the same pass; a window's last workgroup adds the window's ~110 partial sums
            # round 5 (off by default): the whole trial as one launch over the landmark groups
            "k_ba_trial_fused_b": (nw * tr * (E * 112 + Ef * (144 + 4) + P * (48 + 24 + 24 + 24)), "B"),
            "k_ba_trial_fused_imu_b": (nw * tr * (E * 112 + Ef * (144 + 4) + P * (48 + 24 + 24 + 24)), "B"),
            # dense windows (> 21 free keyframes): W D^-1 per slot, then the block-sparse MFMA product (priced in mfma_line by its executed FLOPs)
            "k_ba_schur_coef_b": (nw * tr * Ef * (144 + 144 + 48 + 72), "B"),
            "k_ba_schur_full_b": (nw * tr * pairs * 324.0, "FLOP"), "k_ba_schur_units_b": (nw * tr * pairs * 324.0, "FLOP"),
            # the reduced system of an inertial window on the device (round 5): the multiply-adds of the LDL^T inside its envelope -- per velocity / bias
            # column (9 per free keyframe, band 18) the band below it, the pose rows that reach it (6 per keyframe up to the next one) and their
            # pairs; then the dense pose block -- x 2 FLOP
            "k_lvi_solve_b": (nw * tr * 2.0 * (sum(18 * 9 + 18 * min(6 * nf, 6 * (c // 9 + 2)) + min(6 * nf, 6 * (c // 9 + 2)) ** 2 / 2.0 for c in range(9 * nf)) + (6 * nf) ** 3 / 6.0), "FLOP"),
            "k_lvi_solve": (nw * tr * 2.0 * (sum(18 * 9 + 18 * min(6 * nf, 6 * (c // 9 + 2)) + min(6 * nf, 6 * (c // 9 + 2)) ** 2 / 2.0 for c in range(9 * nf)) + (6 * nf) ** 3 / 6.0), "FLOP"),
            # the LiDAR term's Hessian / gradient from the chunks' partial sums: (21 pair blocks x 36 + 36 + 1) doubles per chunk of 8 planes in, the
            # (6W)^2 + 6W + 1 doubles and the W poses out
            "k_balm_combine_b": (nw * lin * (((ba["planes"] + 7) // 8) * (ba["win"] * (ba["win"] + 1) // 2 * 36 + 6 * ba["win"] + 1) + 36 * ba["win"] ** 2 + 18 * ba["win"]) * 8, "B"),
            "k_balm_hessian_b": (nw * lin * ba["planes"] * ba["win"] * 80, "B"),
            "k_balm_hessian_lean3_b": (nw * lin * ba["planes"] * ba["win"] * 80, "B"),  # the same body held to three / four wavefronts per SIMD
            "k_balm_hessian_lean4_b": (nw * lin * ba["planes"] * ba["win"] * 80, "B"),
            # plane extraction on the device (round 4), per window of N cloud points: the point in (12 B), common-frame point + octants + table
            # slot out; the three orders (a 4 B index per point and layer) from a key per point; the plane test reads every point's common-frame
            # coordinates once per layer; the walk reads a flag and a key per cell (at most a cell per point and layer); the clusters read the
            # planes' points once more and write 80 B per (plane, keyframe)
            "k_balm_cut_points": (nw * ba.get("cloud_points", 0) * (12 + 24 + 1 + 4 + 12), "B"),
            "k_balm_cut_sort": (nw * ba.get("cloud_points", 0) * (5 + 3 * (4 + 4)), "B"),
            "k_balm_cut_judge": (nw * ba.get("cloud_points", 0) * 3 * (4 + 24), "B"),
            "k_balm_cut_walk": (nw * ba.get("cloud_points", 0) * 3 * 5, "B"),
            "k_balm_cut_clusters": (nw * (ba.get("cloud_points", 0) * (4 + 12) + ba["planes"] * ba["win"] * 80), "B"),
            "k_balm_residual_total_b": (nw * (lin + tr) * ba["planes"] * ba["win"] * 80, "B"),
            # the windows' graphs over the bus and their results back, read + written once each (an estimate from the windows' sizes: edges 40 B,
            # CSR / slot index arrays ~24 B per edge, points 24 B up and 25 B down, chi2 8 B per edge down, plane clusters 80 B per plane and
            # keyframe).  A copy is priced against HBM like everything byte-bound; what bounds it is the host link (~55 GB/s), DESIGN.md section 4
            "k_copy_tasks": (2 * nw * (E * (40 + 24 + 8) + P * 49 + ba["planes"] * ba["win"] * 80), "B"),
        })
    return w


def algorithmic_work_inertial(wl, il, nkp, windows_per_step):
    """The table of algorithmic_work for the loop of configs[3]: the camera kernels as before, the LiDAR kernels with the counts of the
    inertial chain, the BA kernels at the LVIBA windows' sizes."""
    ls = np.array(il.li.stats(), np.float64)  # calls, searches, effct, pre, down per scan
    calls, searches, pre, down = ls[:, 0].mean(), ls[:, 1].mean(), ls[:, 3].mean(), ls[:, 4].mean()
    raw = float(np.mean(np.diff(il.raw_offs)))
    F = il.F
    ba = None
    if il.n_ba:
        w0 = il.uniq_lvi[0]
        s0 = il.ba_stats_batch().stats[0]
        fixed = np.asarray(w0["fixed"])
        e6 = np.asarray(w0["edges"])
        free_edge = fixed[e6[:, 1].astype(int)] == 0
        f_l = np.bincount(e6[free_edge, 0].astype(int), minlength=len(w0["points"]))
        ba = {"edges": len(e6), "points": len(w0["points"]), "free": int(s0.n_free_poses), "planes": int(il.ba_stats_batch().lstats[0].n_planes), "win": 6,
              "cloud_points": int(sum(len(c) for c in w0["clouds"])),
              "free_edges": int(free_edge.sum()), "pose_pairs": int((f_l * (f_l + 1) // 2).sum()), "windows": windows_per_step,
              "linearisations": int(s0.iterations), "trials": int(s0.trials)}
    w = algorithmic_work(wl, il, nkp, (raw, pre, down, 0.0), ba)
    mp = il.map_points0
    w.update({
        # the time sort: every point's time stamp in, its place out (what a permutation needs); the compensation: point in and out + the place
        "k_time_sort": (F * pre * 8, "B"), "k_time_sort_lds": (F * pre * (8 + 4), "B"), "k_undistort_batch": (F * pre * (48 + 4 + 48), "B"),
        # the neighbour search runs once per converged iteration, the re-evaluation of the kept neighbours otherwise, the rows every time
        "k_knn_plane": (F * searches * down * (48 + 48 + 48 + 5 * 8 + 27 * 16), "B"),
        "k_knn_hard": (F * searches * down * 0.02 * (48 + 48 + 48 + 5 * 8 + 125 * 16), "B"),
        "k_eskf_refit_b": (F * (calls - searches) * down * (48 + 20 + 5 * 48 + 48 + 48 + 1), "B"),
        "k_eskf_normal_b": (F * calls * down * (48 + 48 + 1), "B"),
        "k_map_keep_scatter": (F * mp * 96, "B"), "k_map_count": (F * mp * 52, "B"), "k_map_scatter": (F * mp * (48 + 16 + 8), "B"),
        # PoseInertialOptimizationLastFrame: every correspondence read once (edge 40 B + point 24 B + close flag), outlier flag out
        "k_pose_inertial": (F * float(np.mean([len(q["edges"]) for q in il.uniq_pi])) * (40 + 24 + 1 + 1), "B"),
        "k_pose_inertial_batch": (F * float(np.mean([len(q["edges"]) for q in il.uniq_pi])) * (40 + 24 + 1 + 1), "B"),  # the same body, two workgroups per CU (> 256 frames)
    })
    for k in ("k_sel_scatter", "k_sel_count"):
        w.pop(k, None)
    return w


def mfma_line(pkg, synthetic, peaks, n_windows=96, repeats=3):
    """the path that DOES use the matrix unit, benched.  The 25-keyframe `bLarge` LocalLVIBA window (Optimizer.cc:1516-1523,
    OptimizerWithLidar.cc:493-500: opt_it 4, lambda 1e-2): 25 free keyframes put the reduced system beyond the block-by-block Schur kernel, so
    S -= (W D^-1) W^T runs as the dense k-major f64 MFMA product (k_ba_schur_gemm*).  A lock-step batch of such windows, timed, then once more
    with every launch timed (tc2li_profile_*): the product's average launch against the measured f64 MFMA peak.  FLOPs = what the kernel executes:
    round 5's k_ba_schur_full_b (k_ba_schur_units_b for wider windows) multiplies only the (chunk of 16 landmarks, 16 x 16 tile) pairs the window's observations touch, counted here
    from the window exactly as the host's chunk masks count them (round 4's line counted all tiles^2 of a dense product whose kernel formed
    the lower triangle: 1.82x too many); `useful_frac` = the FLOPs of the 6x3 . 3x3 . 3x6 products g2o forms / the executed ones."""
    uniq = []
    for k in range(4):
        w = synthetic.inertial_window(100 + k, n_opt=25, n_points=1500)
        pre = []
        for smp, t1, t2 in w["samples"]:
            q = pkg.capi.Preintegrated(w["bias6"], *synthetic.IMU_NOISE)
            q.preintegrate(smp, t1, t2)
            pre.append(q)
        K = len(w["kf33"])
        win = list(range(K - 1, K - 7, -1))
        uniq.append(dict(kf33=w["kf33"], fixed=w["fixed"], has_imu=w["has_imu"], points=w["points"], edges=pkg.pack_ba_edges(w["edges"]), edges6=np.asarray(w["edges"]),
                         link4=w["link4"], pre=pre, win_kf=win, clouds=synthetic.inertial_window_clouds(w, win, n_points=2400, seed=100 + k), Tcl7=synthetic.TCL7,
                         Tbl7=synthetic.tbl7(), weight=1.0, iterations=4, lambda_init=1e-2, calib24=w["calib24"], cam=w["cam"]))
    batch = pkg.capi.LviBatch([uniq[k % 4] for k in range(n_windows)], uniq[0]["calib24"], uniq[0]["cam"])
    if batch.run(8) != n_windows:
        raise RuntimeError("a bLarge LocalLVIBA window failed")
    t0 = time.perf_counter()
    for _ in range(repeats):
        batch.run(8)
    dt = (time.perf_counter() - t0) / repeats
    pkg.capi.profile_enable(True)
    batch.run(8)
    pkg.capi.profile_enable(False)
    report = pkg.capi.profile_report()
    s0 = batch.stats[0]
    nf, P = int(s0.n_free_poses), len(uniq[0]["points"])
    np_pad = max(16, (6 * nf + 15) // 16 * 16)
    tiles = np_pad // 16
    e6, fixed = uniq[0]["edges6"], np.asarray(uniq[0]["fixed"])
    free_edge = fixed[e6[:, 1].astype(int)] == 0
    f_l = np.bincount(e6[free_edge, 0].astype(int), minlength=P)
    useful = float((f_l * (f_l + 1) // 2).sum()) * 324.0
    # what k_ba_schur_full_b / k_ba_schur_units_b EXECUTE for one window and trial: per chunk of 16 landmarks and 16 x 16 tile on or below the diagonal whose rows
    # and whose columns the chunk touches (the host's chunk_mask, rebuilt here from the window), 12 MFMAs of 16 x 16 x 4 x 2 = 2048 FLOP
    pose_var = np.cumsum(fixed == 0) - 1
    fi, fl = pose_var[e6[free_edge, 1].astype(int)], e6[free_edge, 0].astype(int)
    first, last = np.full(P, 1 << 30), np.full(P, -1)
    np.minimum.at(first, fl, fi); np.maximum.at(last, fl, fi)
    seen = np.nonzero(last >= 0)[0]                                  # the host's order: landmarks by (first, last) free pose, stable
    order = seen[np.lexsort((last[seen], first[seen]))]
    rank = np.full(P, -1); rank[order] = np.arange(len(order))
    masks = np.zeros((len(order) + 15) // 16, np.int64)
    c0 = 6 * fi
    np.bitwise_or.at(masks, rank[fl] // 16, (1 << (c0 // 16)) | (1 << ((c0 + 5) // 16)))
    n_mfma = 0
    for ti in range(tiles):
        for tj in range(ti + 1):
            n_mfma += 12 * int((((masks >> ti) & 1) & ((masks >> tj) & 1)).sum())
    flop_window = n_mfma * 2048.0
    dense_tiles = tiles * (tiles + 1) // 2 * 12 * len(masks)
    gemm = {k: v for k, v in report.items() if base_name(k).startswith(("k_ba_schur_full", "k_ba_schur_units"))}
    out = {"workload": "%d bLarge LocalLVIBA windows in lock step (25 free keyframes + the fixed one, %d points, %d stereo edges, LiDAR edge over 6 keyframes x 2400 "
                       "points, 4 iterations at lambda 1e-2): the MFMA Schur path" % (n_windows, P, len(e6)),
           "windows_per_s": round(n_windows / dt, 1), "ms_per_batch": round(1e3 * dt, 3), "iterations/trials": [int(s0.iterations), int(s0.trials)],
           "free_keyframes": nf, "reduced_system": "(6 + 9) x %d" % nf}
    if gemm:
        name = max(gemm, key=lambda k: gemm[k][1])
        calls, ms = gemm[name]
        third = n_windows // 3 if n_windows >= 6 else n_windows  # three lock-step groups: a launch covers a third of the windows
        per_launch = flop_window * n_windows * int(s0.trials) / max(calls, 1)
        rate = per_launch / (ms / max(calls, 1) * 1e-3)
        out["roofline"] = {"kernel": base_name(name), "bound": "mfma", "achieved": round(rate / 1e12, 3), "peak": round(peaks["mfma_f64_tflops"], 2), "unit": "TFLOP/s",
                           "frac": round(rate / 1e12 / max(peaks["mfma_f64_tflops"], 1e-9), 5), "frac_of_spec": round(rate / 1e12 / 78.6, 5),
                           "launches": int(calls), "avg_launch_ms": round(ms / max(calls, 1), 6), "executed_flops_per_launch": int(per_launch),
                           "useful_frac": round(useful / max(flop_window, 1.0), 4), "windows_per_launch": third,
                           "mfma_per_window_and_trial": n_mfma, "skipped_frac_of_dense_lower_triangle": round(1.0 - n_mfma / max(dense_tiles, 1), 4),
                           "tiles": "%d lower-triangle tiles of 16 x 16, %d chunks of 16 landmarks (48 operand rows)" % (tiles * (tiles + 1) // 2, len(masks)),
                           "peak_source": "measured: back-to-back v_mfma_f64_16x16x4_f64 (tc2li_diag_peaks); data sheet 78.6 TFLOP/s (frac_of_spec)",
                           "counters": "profiles/*_pmc_mfma.json: SQ_VALU_MFMA_BUSY_CYCLES of this kernel / 64 x 2048 FLOP / its launch duration reproduces `achieved` "
                                       "(tools/profile_round.sh, own --pmc pass of `bench.py --mfma-only`)"}
    return out


def cpu_baseline_inertial(wl, il, args, n_seq_gpu):
    """configs[3] on the CPU oracle with the reference's threads: per frame the tracking thread (left / right ORB on two threads, stereo matching,
    the IMU prediction in place of TrackWithMotionModel, SearchLocalPoints, IMU pre-integration, PoseInertialOptimizationLastFrame -- the camera path with
    the IMU initialised, Tracking.cc:2746-2752, 2857-2878) beside the LiDAR thread (preprocess, forward
    propagation, UndistortPcl, voxel filter, iterated ESKF against the sequence's ikd-Tree, Add_Points), LocalLVIBA of every kf_interval-th
    frame on a local-mapping thread."""
    from oracle import pyoracle
    cores = effective_cpus()
    synthetic = wl.synthetic
    lasts = [dict(pose7=l["pose7"], has_point=l["has_point"], outlier=l["outlier"], Xw=l["Xw"], keys6=pyoracle._kps_to_floats(l["keys"]),
                  descriptors=l["descriptors"]) for l in wl.last]
    no_scan = wl.scans[0][:0].copy()
    noise = synthetic.IMU_NOISE
    tbl = synthetic.tbl7()
    lvi_pre = []  # the keyframes' pre-integrations exist when local mapping starts: not part of the timed work
    for w in il.uniq_lvi:
        lvi_pre.append(np.stack([pyoracle.pack_preintegrated(pyoracle.imu_preintegrate(smp, t1, t2, w["bias6"], *noise)[1], w["bias6"]) for smp, t1, t2 in w["samples"]]))

    def make_seq(t):
        return dict(cam=pyoracle.Sequence(wl.maps[t][:8]), tree=pyoracle.KdTree(wl.maps[t]))

    def st24(x):
        return np.concatenate([x[3:12], x[0:3], x[24:33], x[33:36]])

    def lidar_thread(seq, t):
        q = il.uniq_li[t]
        pts = pyoracle.lidar_preprocess(wl.scans[t])
        x, P, poses = pyoracle.imu_propagate_cov(q["x"], q["P"], il.cov12, q["imu"], *q["times"], np.zeros(6))[:3]
        down = pyoracle.voxel_grid(pyoracle.undistort(pts, poses, st24(x)))
        x2, _, _ = pyoracle.eskf_update(x, P, seq["tree"], down, max_iter=3)
        # map_incremental (LidarFrontEnd.cpp:760): the scan in the world frame goes into the tree with its down-sampling rule
        Rw, pw, Rl, tl = x2[3:12].reshape(3, 3), x2[0:3], x2[24:33].reshape(3, 3), x2[33:36]
        body = np.stack([down["x"], down["y"], down["z"]], 1).astype(np.float64)
        world = down.copy()
        xyz = ((body @ Rl.T + tl) @ Rw.T + pw).astype(np.float32)
        world["x"], world["y"], world["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
        pyoracle.kdtree_add_points(seq["tree"], world, True, 0.5)

    def frame(seq, t, pool):
        lid = pool.submit(lidar_thread, seq, t)
        held, held_Xw, pts = wl.local[t]
        seq["cam"].frame(wl.images[t, 0], wl.images[t, 1], float(wl.bf), float(wl.b), no_scan, wl.states[t], wl.pose_pred, lasts[t], wl.cam5, 7.0,
                         held, held_Xw, pts, th_local=1.0, imu_mode=True)
        q = il.uniq_pi[t]
        pre298 = pyoracle.pack_preintegrated(pyoracle.imu_preintegrate(q["samples"], q["t1"], q["t2"], q["bias6"], *noise)[1], q["bias6"])
        pyoracle.pose_inertial(q["cur33"], q["other33"], True, q["prior246"], q["calib24"], pre298, pre298, q["Xw"], q["edges"], q["close"], q["cam"])
        lid.result()

    def cpu_lviba(k):
        w = il.uniq_lvi[k % len(il.uniq_lvi)]
        return pyoracle.local_lviba(w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"], w["edges"], w["link4"], lvi_pre[k % len(lvi_pre)], w["cam"],
                                    w["win_kf"], w["clouds"], synthetic.TCL7, tbl, 1.0, iterations=il.lvi_iterations, lambda_init=il.lvi_lambda)[4]

    def run_sequences(n_seq, n_workers, budget):
        seqs = [make_seq(s % wl.U) for s in range(n_seq)]
        lidar_pool = ThreadPoolExecutor(max_workers=n_workers)
        ba_pool = ThreadPoolExecutor(max_workers=n_workers)
        ba_futs, done = [], [0] * n_seq
        t0 = time.perf_counter()

        def worker(j):
            while True:
                for s in range(j, n_seq, n_workers):
                    frame(seqs[s], s % wl.U, lidar_pool)
                    done[s] += 1
                    if il.n_ba and done[s] % args.kf_interval == 0:
                        ba_futs.append(ba_pool.submit(cpu_lviba, done[s] // args.kf_interval + s))
                if time.perf_counter() - t0 > budget and min(done[j::n_workers]) >= 2:
                    return
        ts = [threading.Thread(target=worker, args=(j,)) for j in range(n_workers)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for f in ba_futs:
            f.result()
        return sum(done), time.perf_counter() - t0, len(ba_futs)

    budget = 0.5 * args.cpu_seconds
    frames1, dt1, nba1 = run_sequences(1, 1, budget)
    # a sequence in flight = the reference's 4 threads, busy about a quarter of the time each on average: one sequence in flight per granted
    # CPU keeps the CPUs busy (4 in flight on 16 CPUs read 126-136 frames/s, round 4's 256 workers on the same 16 CPUs 204)
    n_workers = max(1, cores)
    n_many = max(2, min(n_seq_gpu, 2 * n_workers))
    framesN, dtN, nbaN = run_sequences(n_many, n_workers, budget)
    return {"value": round(framesN / dtN, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d sequences, %d in flight (the reference's 4 threads each) on the %d CPUs granted (affinity %d, cgroup quota %s): "
                      "%d frames in %.1f s, %d LocalLVIBA windows" % (n_many, n_workers, cores, len(os.sched_getaffinity(0)), cgroup_cpu_quota(), framesN, dtN, nbaN),
            "single_sequence": {"value": round(frames1 / dt1, 3), "unit": "frames/s", "cores": 4,
                                "sample": "1 sequence, %d frames in %.1f s (%d LocalLVIBA windows)" % (frames1, dt1, nba1)}}


def base_name(name):
    return name.split("<")[0]


def roofline_from_profile(report, work, peaks, n_steps):
    """The kernel with the largest total device time of the profiled pass and what it achieves against its bound."""
    by_kernel = {}
    for name, (calls, ms) in report.items():
        b = base_name(name)
        c0, m0 = by_kernel.get(b, (0, 0.0))
        by_kernel[b] = (c0 + calls, m0 + ms)
    total_ms = sum(m for _, m in by_kernel.values())
    ranked = sorted(by_kernel.items(), key=lambda kv: -kv[1][1])
    table = {}
    for name, (calls, ms) in ranked[:24]:
        row = {"launches_per_step": round(calls / n_steps, 2), "ms_per_step": round(ms / n_steps, 4), "avg_launch_us": round(1e3 * ms / max(calls, 1), 2),
               "share_of_kernel_time": round(ms / max(total_ms, 1e-9), 4)}
        if name in work:
            amount, unit = work[name]
            rate = amount * n_steps / (ms * 1e-3)
            if unit == "B":
                row.update({"achieved_GBps": round(rate / 1e9, 1), "frac_hbm": round(rate / 1e9 / 8000.0, 5)})
            elif unit == "FLOP_VALU":
                row.update({"achieved_TFLOPs": round(rate / 1e12, 3), "frac_valu_f64_measured": round(rate / 1e12 / max(peaks["fma_f64_tflops"], 1e-9), 5)})
            else:
                row.update({"achieved_TFLOPs": round(rate / 1e12, 3), "frac_mfma_f64_measured": round(rate / 1e12 / max(peaks["mfma_f64_tflops"], 1e-9), 5)})
        table[name] = row
    dom, (calls, ms) = ranked[0]
    # copy / fill engines' blit kernels are not ours: the dominant kernel is the first one the library launches
    for name, (c, m) in ranked:
        if name.startswith("k_"):
            dom, calls, ms = name, c, m
            break
    out = {"kernel": dom, "launches_per_step": round(calls / n_steps, 2), "avg_launch_ms": round(ms / max(calls, 1), 6),
           "share_of_kernel_time": round(ms / max(total_ms, 1e-9), 4)}
    if dom in work:
        amount, unit = work[dom]
        per_launch = amount * n_steps / max(calls, 1)
        rate = per_launch / (ms / max(calls, 1) * 1e-3)
        if unit == "B":
            out.update({"bound": "hbm", "achieved": round(rate / 1e9, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(rate / 1e9 / 8000.0, 5),
                        "algorithmic_bytes_per_launch": int(per_launch)})
        elif unit == "FLOP_VALU":
            # a kernel of the f64 VECTOR unit (k_ba_schur_blocks_b issues no matrix instruction): against the vector FMA rate this GPU reaches
            # (tc2li_diag_peaks) and the data sheet's f64 vector figure -- the contract's vocabulary has "hbm" and "mfma" only, neither fits
            out.update({"bound": "valu_f64", "achieved": round(rate / 1e12, 3), "peak": round(peaks["fma_f64_tflops"], 2), "unit": "TFLOP/s",
                        "frac": round(rate / 1e12 / max(peaks["fma_f64_tflops"], 1e-9), 5), "algorithmic_flops_per_launch": int(per_launch),
                        "peak_source": "measured: dependent-free f64 v_fma chains on every SIMD (tc2li_diag_peaks); data sheet 78.6 TFLOP/s f64 vector (frac_of_spec)",
                        "frac_of_spec": round(rate / 1e12 / 78.6, 5)})
        else:
            out.update({"bound": "mfma", "achieved": round(rate / 1e12, 3), "peak": round(peaks["mfma_f64_tflops"], 2), "unit": "TFLOP/s",
                        "frac": round(rate / 1e12 / max(peaks["mfma_f64_tflops"], 1e-9), 5), "algorithmic_flops_per_launch": int(per_launch),
                        "peak_source": "measured: back-to-back v_mfma_f64_16x16x4_f64 (tc2li_diag_peaks); the part's f64 matrix spec is 78.6 TFLOP/s (frac_of_spec), "
                                       "the measured f64 vector FMA peak peaks_measured.fma_f64_tflops -- k_ba_schur_blocks_b runs on the vector unit",
                        "frac_of_spec": round(rate / 1e12 / 78.6, 5), "frac_of_measured_vector_fma": round(rate / 1e12 / max(peaks["fma_f64_tflops"], 1e-9), 5)})
    else:
        out.update({"bound": "hbm", "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None})
    return out, table, total_ms / n_steps


def source_hash():
    """sha256 over the library's sources, its header and the synthetic workload generator (round 5: no longer this file -- a change of the
    report's wording must not orphan the counters measured on the same kernels): what tools/summarize_profile.py stamps into profiles/*_pmc_traffic.json, so that a
    traffic figure is only quoted for the code it was measured on (the GPU box has no git history to ask)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "tc2li-slam_amd", "csrc")
    for f in sorted(os.listdir(csrc)) + ["../../include/tc2li_hip.h", "../synthetic.py"]:
        path = os.path.join(csrc, f)
        if os.path.isfile(path) and not os.path.basename(f).startswith("."):
            h.update(f.encode()); h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, pattern="*_pmc_traffic.json"):
    """HBM bytes per launch of `kernel` from a committed PMC summary (tools/profile_round.sh: FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc
    passes of this bench command; the counters cannot be read inside this process) -- only from a summary made on THIS source tree
    (`source_hash`); None otherwise."""
    import glob
    here = source_hash()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        pmc = json.load(open(f))
        if pmc.get("_source_hash") != here:
            continue
        hit = [v for k, v in pmc.items() if not k.startswith("_") and base_name(k.split("::")[-1]) == kernel and v.get("hbm_bytes_per_launch")]
        if hit:
            n = sum(v["launches"] for v in hit)
            return {"bytes_per_launch": int(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hit) / max(n, 1)), "source": "profiles/" + os.path.basename(f),
                    "source_hash": here,
                    "note": "FETCH_SIZE x 2 + WRITE_SIZE per launch from separate rocprofv3 --pmc passes of this bench command on the same sources"}
    return None


# ---------------------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle; rank 0 at N = 1 only)
# ---------------------------------------------------------------------------------------------------------------------------------
def cpu_baseline(wl, args, n_seq_gpu, with_ba):
    from oracle import pyoracle  # built by main() before the GPU was touched
    cores = effective_cpus()
    W, H = wl.W, wl.H
    lasts = [dict(pose7=l["pose7"], has_point=l["has_point"], outlier=l["outlier"], Xw=l["Xw"], keys6=pyoracle._kps_to_floats(l["keys"]),
                  descriptors=l["descriptors"]) for l in wl.last]

    def make_seq(t):
        return pyoracle.Sequence(wl.maps[t])

    def frame(seq, t):
        n = len(wl.last[t]["keys"])  # noqa: F841
        held, held_Xw, pts = wl.local[t]
        return seq.frame(wl.images[t, 0], wl.images[t, 1], float(wl.bf), float(wl.b), wl.scans[t], wl.states[t], wl.pose_pred, lasts[t], wl.cam5, 7.0,
                         held, held_Xw, pts, th_local=1.0)

    def cpu_ba(k):
        w = wl.ba_windows[k % len(wl.ba_windows)]
        if not len(w["win_pose"]):
            return pyoracle.local_ba(w["poses"], w["fixed"], w["points"], w["edges6"], w["cam"], iterations=w.get("iterations", 10))[4]
        return pyoracle.local_ba_lidar(w["poses"], w["fixed"], w["points"], w["edges6"], w["cam"], w["win_pose"], w["clouds"], w["Tcl7"], w["weight"],
                                       iterations=w.get("iterations", 10))[4]

    def run_sequences(n_seq, n_workers, budget):
        """n_seq sequences advanced frame by frame by n_workers concurrent tracking threads (each frame call spawns the reference's ORB and LiDAR
        threads itself); one local-mapping thread per sequence slot runs the LV-BA of every kf_interval-th frame."""
        seqs = [make_seq(s % wl.U) for s in range(n_seq)]
        ba_pool = ThreadPoolExecutor(max_workers=n_workers)
        ba_futs, done = [], [0] * n_seq
        stop = threading.Event()
        t0 = time.perf_counter()

        def worker(j):
            while not stop.is_set():
                for s in range(j, n_seq, n_workers):
                    frame(seqs[s], s % wl.U)
                    done[s] += 1
                    if with_ba and done[s] % args.kf_interval == 0:
                        ba_futs.append(ba_pool.submit(cpu_ba, done[s] // args.kf_interval + s))
                if time.perf_counter() - t0 > budget and min(done[j::n_workers]) >= 2:
                    return
        ts = [threading.Thread(target=worker, args=(j,)) for j in range(n_workers)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for f in ba_futs:
            f.result()  # the loop is not finished before local mapping has caught up
        dt = time.perf_counter() - t0
        return sum(done), dt, len(ba_futs), seqs[0].map_size()

    frames1, dt1, nba1, msize = run_sequences(1, 1, args.cpu_seconds)
    # a sequence in flight = the reference's 4 threads (tracking + second ORB image + LiDAR front end + local mapping), busy about a quarter of
    # the time each on average: one sequence in flight per granted CPU keeps the CPUs busy
    n_workers = max(1, cores)
    n_many = max(2, min(n_seq_gpu, 2 * n_workers))
    framesN, dtN, nbaN, _ = run_sequences(n_many, n_workers, args.cpu_seconds)
    return {"value": round(framesN / dtN, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d sequences, %d in flight (the reference's 4 threads each) on the %d CPUs granted (affinity %d, cgroup quota %s): "
                      "%d frames in %.1f s, %d LV-BA windows" % (n_many, n_workers, cores, len(os.sched_getaffinity(0)), cgroup_cpu_quota(), framesN, dtN, nbaN),
            "single_sequence": {"value": round(frames1 / dt1, 3), "unit": "frames/s", "cores": 4 if with_ba else 3,
                                "sample": "1 sequence, %d frames in %.1f s (%d local LV-BA windows), the reference's 4 threads" % (frames1, dt1, nba1)},
            "map_points": msize, "host_cpus": os.cpu_count(), "cpus_granted": cores}


# ---------------------------------------------------------------------------------------------------------------------------------
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, argv)

    import tc2li_loader
    pkg = tc2li_loader.load()
    from tc2li_slam_amd import synthetic, dist_util
    rank, local_rank, world = dist_util.rank_info()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE = %d" % (args.gpus, world))
    host_budget = apply_host_budget(pkg, local_rank)  # before the first HIP call and before this process has other threads
    if args.rehearse:
        dist = dist_util.init("gloo", rank, world)
        rc = rehearse(args, rank, world, dist, dist_util, host_budget)
        if dist is not None:
            dist.destroy_process_group()
        return rc

    import torch
    import __graft_entry__ as ge
    if args.no_build or os.environ.get("TC2LI_NO_BUILD"):
        if not ge.native_is_fresh():
            raise SystemExit("bench.py --no-build: %s is missing or older than its sources; run `python __graft_entry__.py` first" % ge.LIB)
    elif rank == 0:
        if under_profiler() and not ge.native_is_fresh():
            raise SystemExit("bench.py under a profiler with a stale library: build first (python __graft_entry__.py); a profiled process must not start make")
        ge.build_native()
    else:
        # under torch.distributed.run the ranks start together: only rank 0 builds, the others wait until the library is newer than its sources
        t_wait = time.time()
        while not ge.native_is_fresh():
            if time.time() - t_wait > 900:
                raise SystemExit("bench.py: rank %d waited 15 minutes for rank 0 to build %s" % (rank, ge.LIB))
            time.sleep(0.5)
    # the CPU oracle (the cpu_baseline leg's checker) is built before this process touches the GPU: a process that has initialised HIP does not
    # start children
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.rehearse:
        if under_profiler():
            raise SystemExit("bench.py under a profiler: pass --no-cpu-baseline (the CPU leg builds and runs the oracle; nothing to profile there)")
        ge.build_oracle()
    # ---- the single-sequence line runs FIRST, in a child process of its own (this process has not touched the GPU yet: a process that has
    # may not start children): the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default), and the ten
    # streams of the five stage threads then queue behind each other -- with one sequence per step, where every kernel is tiny, that false
    # ordering is most of a frame's time (479 frames/s with 4 queues, 761 with 8; the batched lines lose 1-9 % with 8 and keep the default)
    single_child = single_fed_child = single_inertial_child = None
    if rank == 0 and not args.no_extra_lines and not args.rehearse and not os.environ.get("TC2LI_BENCH_SINGLE_INPROC") and not under_profiler():
        single_child = single_sequence_child(args)
        # the same with every frame's images and scan taken from pinned host memory (what a drop-in delivers), and the inertial configuration
        single_fed_child = single_sequence_child(args, ("--host-fed",))
        if not args.front_end_only:
            single_inertial_child = single_sequence_child(args, ("--inertial-loop",) + (("--lviba-small",) if args.lviba_small else ()), "configs[3] with 1 sequence per step: one LocalLVIBA window every %d-th frame")
    sweep_child = budget_child = None
    if (rank == 0 and world == 1 and not args.no_extra_lines and not args.rehearse and not args.front_end_only and not under_profiler()
            and args.scaling == "strong" and set(args.stages.split(",")) == {"orb", "track", "lidar", "ba"}):
        sweep_child = sweep_children(args)
        budget_child = host_budget_children(args)
    if not torch.cuda.is_available() or pkg.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU (the node shows %d)" % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if args.mfma_only:  # the matrix-unit line alone (what the MFMA counters' --pmc pass profiles)
        pk = pkg.capi.diag_peaks()
        print(json.dumps({"mfma_config": mfma_line(pkg, synthetic, {"mfma_f64_tflops": pk[0], "fma_f64_tflops": pk[1]})}))
        sys.stdout.flush()
        return 0
    dist = dist_util.init(args.backend, rank, world, device=torch.device("cuda", local_rank) if args.backend == "nccl" else None)

    if args.scaling == "strong":
        seq_ids = dist_util.shard_units(args.sequences, rank, world)
        total_sequences = args.sequences
    else:
        seq_ids = list(range(rank * args.frames, (rank + 1) * args.frames))
        total_sequences = args.frames * world
    if not seq_ids:
        raise SystemExit("bench.py: rank %d owns no sequence (%d sequences over %d ranks)" % (rank, args.sequences, world))
    U = min(args.unique, max(len(seq_ids), 1) if args.scaling == "weak" else args.unique)
    wl = Workload(pkg, synthetic, U, args.map_length, with_ba=False if args.front_end_only else args.ba_mix,
                  cycle=1 if (args.inertial_loop or args.mfma_only) else args.cycle_frames)
    stream = torch.cuda.current_stream().cuda_stream
    ext0 = pkg.OrbExtractor(max_width=wl.W, max_height=wl.H, max_images=2 * U)
    wl.build_tracking_inputs(ext0, stream)
    ext0.close()
    loop = InertialLoop(wl, seq_ids, args, local_rank) if args.inertial_loop else Loop(wl, seq_ids, args, local_rank)
    if args.host_fed:
        loop.enable_host_fed()
    F = loop.F

    def barrier():
        dist_util.barrier(dist, torch.cuda.synchronize)

    stages = tuple(args.stages.split(","))
    if args.warmup:
        loop.run(args.warmup, stages)
    barrier()
    loop.orb_times.clear(); loop.lidar_times.clear()
    ba0 = loop.ba_windows_done
    tcpu0 = thread_cpu_seconds()
    cpu0 = os.times()
    t0 = time.perf_counter()
    loop.run(args.steps, stages)
    barrier()
    local_elapsed = time.perf_counter() - t0
    cpu1 = os.times()
    host_budget["cpu_s_per_wall_s_timed_region"] = round(((cpu1.user - cpu0.user) + (cpu1.system - cpu0.system)) / max(local_elapsed, 1e-9), 2)
    host_budget["cgroup_cpu_quota"] = cgroup_cpu_quota()
    if loop.ba_engine is not None:
        host_budget["ba_engines"] = len(loop.ba_engines)
    if loop.ba_step_workers:
        host_budget["mapping_workers"] = len(loop.ba_step_workers)
        host_budget["stage_threads"] = 4 + len(loop.ba_step_workers)
    if loop.ba_workers:  # local mapping as a pool of mapping workers: that many stage threads instead of one, each driving its lock-step group itself
        host_budget["mapping_workers"] = len(loop.ba_workers)
        host_budget["stage_threads"] = 4 + len(loop.ba_workers)
    tcpu1 = thread_cpu_seconds()
    host_budget["cpu_s_per_wall_s_by_thread_name"] = {k: round((v - tcpu0.get(k, 0.0)) / max(local_elapsed, 1e-9), 2)
                                                       for k, v in sorted(tcpu1.items(), key=lambda kv: -(kv[1] - tcpu0.get(kv[0], 0.0)))
                                                       if v - tcpu0.get(k, 0.0) > 0.02 * local_elapsed}
    elapsed = dist_util.max_elapsed(dist, local_elapsed, device="cuda" if args.backend == "nccl" else "cpu")
    ranks_seen = dist_util.count_ranks(dist, device="cuda" if args.backend == "nccl" else "cpu")
    if set(stages) != {"orb", "track", "lidar", "ba"}:  # diagnostics: which stages slow each other down
        if rank == 0:
            print(json.dumps({"diagnostic_stages": args.stages, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "frames_per_step": F,
                              "track_calls_ms_last_step": getattr(loop, "track_ms", None),
                              "stage_thread_ms_per_step_concurrent": {k: round(v, 3) for k, v in loop.thread_ms.items()}}))
        if dist is not None:
            dist.destroy_process_group()
        return 0
    thread_ms = dict(loop.thread_ms)
    ba_windows_timed = loop.ba_windows_done - ba0
    orb_out, st_out, trk_out, tlm_out = loop.orb_outs[0], loop.st_outs[0], loop.trk_outs[0], loop.tlm_out
    nkp = float(np.mean(orb_out[2]))
    if loop.lidar_counts is not None:
        lid_mean = [int(np.mean(np.diff(loop.raw_offs)))] + [int(v) for v in np.mean(loop.lidar_counts, 1)]
    else:  # the inertial loop (--inertial-loop): preprocessed, down-sampled, ESKF features from the LiDAR-inertial call's own counters
        ls_ = np.array(loop.li.stats())
        lid_mean = [int(np.mean(np.diff(loop.raw_offs))), int(ls_[:, 3].mean()), int(ls_[:, 4].mean()), int(ls_[:, 2].mean())]
    map_points_end = int(np.mean([m.size() for m in loop.maps]))

    # ---- stage wall times of one more step, every stage alone (host clock, each stage synchronises at its end) ----
    wall = {}
    if rank == 0 and not args.no_extra_lines:
        t_a = time.perf_counter(); loop.extract(0, stream); wall["orb_extract_batch"] = time.perf_counter() - t_a
        t_a = time.perf_counter(); loop.track(0, stream); wall["stereo + TrackWithMotionModel"] = time.perf_counter() - t_a
        t_a = time.perf_counter(); loop.track_local(0, stream); wall["TrackLocalMap"] = time.perf_counter() - t_a
        t_a = time.perf_counter(); loop.lidar_step(); wall["lidar_frontend_batch + map_incremental"] = time.perf_counter() - t_a
        if loop.ba_batch:
            t_a = time.perf_counter(); loop.ba_batch.run(args.ba_concurrency); wall["local_lv_ba_batch(%d windows)" % loop.n_ba] = time.perf_counter() - t_a

    # ---- roofline: a second pass of the same loop with an event pair around every kernel launch (the first, timed pass is
    # un-instrumented: `value` comes from it); the dominant kernel = the one with the largest total device time over ALL kernels ----
    roofline, kernel_table, kernel_ms_per_step, peaks = None, None, None, None
    if rank == 0 and (not args.no_extra_lines or args.with_roofline):
        pk = pkg.capi.diag_peaks()
        ck = pkg.capi.diag_clocks()
        peaks = {"mfma_f64_tflops": round(pk[0], 2), "fma_f64_tflops": round(pk[1], 2), "hbm_copy_GBps": round(pk[2], 1),
                 "clock_ghz_in_mfma_loop": round(ck[0], 3), "clock_ghz_in_fma_loop": round(ck[1], 3),
                 "mfma_f64_spec_tflops": 78.6,
                 "mfma_note": "the matrix loop holds its clock (s_memtime against the 100 MHz counter): the gap to the 78.6 TFLOP/s of the data sheet is "
                              "issue cadence (~105 cycles per v_mfma_f64_16x16x4_f64 and SIMD where 64 would give the data-sheet rate), not DVFS",
                 "note": "measured on this GPU by tc2li_diag_peaks: back-to-back v_mfma_f64_16x16x4_f64, f64 vector FMA, 1 GiB float4 copy (read + write)"}
        n_prof = max(4, min(args.steps, 10))
        pkg.capi.profile_enable(True)
        ba1 = loop.ba_windows_done
        loop.run(n_prof, stages)
        torch.cuda.synchronize()
        pkg.capi.profile_enable(False)
        report = pkg.capi.profile_report()
        ba = None
        if loop.ba_batch:
            # the windows' sizes as means over the mix the loop ran, weighted by the linearisations a window took (the products windows x
            # linearisations x size of the table below are then the sums over the windows; with the uniform mix: the one window's own numbers)
            ba = dict(loop.ba_mix_summary(wl), windows=(loop.ba_windows_done - ba1) / n_prof)
        roofline, kernel_table, kernel_ms_per_step = roofline_from_profile(report, algorithmic_work(wl, loop, nkp, lid_mean, ba), peaks, n_prof)
        roofline["traffic"] = pmc_traffic(roofline["kernel"])
        roofline["measured_in"] = ("a second pass of %d steps of the same concurrent loop in which every kernel is launched with a start and a stop "
                                   "event of its own dispatch (hipExtLaunchKernelGGL, on the stream the kernel is launched on; tc2li_profile_*): the "
                                   "durations are those of kernels sharing the GPU with the other stage threads, as in the rocprofv3 --kernel-trace "
                                   "--stats summary of this command that tools/profile_round.sh commits under profiles/" % n_prof)
        roofline["all_kernels"] = kernel_table
        roofline["kernel_ms_per_step_all_streams"] = round(kernel_ms_per_step, 3)
        roofline["peaks_measured"] = peaks
        # The same kernel with its STAGE alone on the GPU (the other stage threads idle): in the loop a launch shares the vector units with
        # three other stages' kernels and takes two to four times as long, which says how full the GPU is, not how good the kernel is.
        # Same recipe: the stage's loop for a few steps with per-dispatch events, the step's algorithmic amount over the kernel's launches.
        dom = roofline.get("kernel") or ""
        alone_stages = (("ba",) if dom.startswith(("k_ba_", "k_balm_", "k_copy_")) else ("lidar",) if dom.startswith(("k_knn", "k_voxel", "k_pre_", "k_map", "k_sel_", "k_seg_")) else
                        ("orb", "track") if dom.startswith(("k_stereo", "k_match", "k_pose_opt", "k_track", "k_project")) else ("orb",))
        if set(alone_stages) <= set(stages) and not under_profiler():
            n_alone = 4 if "ba" not in alone_stages else max(4, args.kf_interval)
            loop.run(2, alone_stages)
            torch.cuda.synchronize()
            pkg.capi.profile_enable(True)
            loop.run(n_alone, alone_stages)
            torch.cuda.synchronize()
            pkg.capi.profile_enable(False)
            rep_a = {k: v for k, v in pkg.capi.profile_report().items() if base_name(k) == dom}
            calls_a, ms_a = sum(v[0] for v in rep_a.values()), sum(v[1] for v in rep_a.values())
            amount = roofline.get("algorithmic_bytes_per_launch") or roofline.get("algorithmic_flops_per_launch")
            if calls_a and ms_a > 0 and amount and roofline.get("launches_per_step") and roofline.get("peak"):
                per_launch = amount * roofline["launches_per_step"] * n_alone / calls_a   # the same work per step over this pass's launches
                rate = per_launch / (ms_a / calls_a * 1e-3) / (1e9 if roofline.get("unit") == "GB/s" else 1e12)
                roofline["alone"] = {"stages": list(alone_stages), "avg_launch_ms": round(ms_a / calls_a, 6), "achieved": round(rate, 2), "frac": round(rate / roofline["peak"], 5)}

    # ---- host-fed inputs: the same loop with every step's images and raw scans taken from pinned host memory (what a drop-in delivers) ----
    host_fed = None
    if rank == 0 and not args.no_extra_lines and not args.host_fed and set(stages) == {"orb", "track", "lidar", "ba"}:
        loop.enable_host_fed()
        loop.run(3, stages)
        torch.cuda.synchronize()
        n_hf = max(6, min(args.steps, 12))
        t1 = time.perf_counter()
        loop.run(n_hf, stages)
        torch.cuda.synchronize()
        dt_hf = time.perf_counter() - t1
        gb = loop.fed_bytes_per_step / 1e9
        host_fed = {"value": round(F * n_hf / dt_hf, 2), "unit": "frames/s", "ms_per_step": round(1e3 * dt_hf / n_hf, 3), "sequences": F, "steps": n_hf,
                    "GB_per_step": round(gb, 3), "link_GBps": round(gb * n_hf / dt_hf, 2),
                    "link_spec_GBps": 63.0, "stage_thread_ms_per_step_concurrent": {k: round(v, 3) for k, v in loop.thread_ms.items()},
                    "workload": "the timed loop of `value` with the inputs of every step -- %d images and %d raw scans, %.2f GB -- starting in pinned host memory: "
                                "uploads through the copy engines inside the timed region, step k + 1's under step k's processing (two device buffers per "
                                "input); `link_GBps` = bytes uploaded / elapsed, i.e. what the loop drew from the host link (PCIe Gen5 x16: 63 GB/s by the "
                                "data sheet)" % (loop.n_img, F, gb)}
        loop.host_fed = False

    # ---- the same loop with the windows rounds 1-5 timed: four 12 + 20-keyframe windows tiled over the step's batch ----
    value_uniform = None
    if rank == 0 and world == 1 and not args.no_extra_lines and wl.ba_mix == "varied" and loop.ba_batch and set(stages) == {"orb", "track", "lidar", "ba"} \
            and not args.inertial_loop:
        import copy
        wl_u = copy.copy(wl)
        wl_u.ba_windows, wl_u.ba_mix = uniform_ba_windows(pkg, synthetic), "uniform"
        lu = Loop(wl_u, seq_ids, args, local_rank)
        lu.run(max(args.warmup, 3), stages)
        torch.cuda.synchronize()
        n_u = max(6, min(args.steps, 16))
        t1 = time.perf_counter()
        lu.run(n_u, stages)
        torch.cuda.synchronize()
        dt_u = time.perf_counter() - t1
        su = lu.ba_mix_summary(wl_u) or {}
        value_uniform = {"value": round(F * n_u / dt_u, 2), "unit": "frames/s", "ms_per_step": round(1e3 * dt_u / n_u, 3), "steps": n_u,
                         "ba": {k: su.get(k) for k in ("iterations_min_mean_max", "trials_min_mean_max", "distinct_windows")},
                         "workload": "the loop of `value` with local mapping's windows as rounds 1-5 timed them: four windows of 12 free + 20 fixed keyframes, "
                                     "3000 points and a LiDAR edge over 6 keyframes, tiled over the step's batch -- every window of a lock-step group takes the same "
                                     "Levenberg-Marquardt path"}
        lu.close()
        del lu

    # ---- the same loop for ONE sequence (F = 1): what a single KITTI-00 run sees ----
    single = single_child
    if single is not None and single_fed_child is not None:
        # what a drop-in delivers for one sequence is the host-fed number: that is the one quoted; the HBM-resident one beside it
        single = dict(single_fed_child, resident_value=single_child["value"], resident_ms_per_frame=single_child["ms_per_frame"],
                      workload=single_fed_child["workload"] + "; every frame's two images and its scan (5.1 MB) start in pinned host memory (--host-fed)")
    if rank == 0 and not args.no_extra_lines and single is None:
        one = Loop(wl, [0], args, local_rank)
        one.run(8, stages)
        torch.cuda.synchronize()
        n1 = 48
        t1 = time.perf_counter()
        one.run(n1, stages)
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        single = {"value": round(n1 / dt1, 2), "unit": "frames/s", "ms_per_frame": round(1e3 * dt1 / n1, 3), "frames": n1,
                  "ba_windows": one.ba_windows_done, "workload": "the same loop with 1 sequence per step (F = 1): one LV-BA window every %d-th frame" % args.kf_interval,
                  "stage_thread_ms_per_frame": {k: round(v, 3) for k, v in one.thread_ms.items()}}
        one.close()

    # ---- the same loop on a tighter host: 4 / 8 CPUs (children) beside this process's own figure at the full grant ----
    host_budget_sweep = None
    if budget_child is not None:
        host_budget_sweep = dict(budget_child)
        host_budget_sweep[str(host_budget.get("cpus_effective"))] = {"value": round(total_sequences * args.steps / elapsed, 1),
                                                                      "cpu_s_per_wall_s": host_budget.get("cpu_s_per_wall_s_timed_region"),
                                                                      "threads": host_budget.get("threads_of_this_rank")}
        host_budget_sweep["note"] = ("frames/s of the default loop with the process confined to that many CPUs (taskset before the first GPU call); CPU-seconds "
                                     "per wall second and library + stage threads of the rank beside it")
    # ---- sequences per GPU: what strong scaling over the fixed list turns into at 8 / 4 / 2 ranks, timed here on one GPU ----
    sweep = None
    if sweep_child is not None:
        sweep = {str(F): round(total_sequences * args.steps / elapsed, 1), **sweep_child}
    elif rank == 0 and world == 1 and not args.no_extra_lines and set(stages) == {"orb", "track", "lidar", "ba"}:
        sweep = {str(F): round(total_sequences * args.steps / elapsed, 1)}
        for n_seq in (256, 128, 64):
            if n_seq >= F:
                continue
            sub = Loop(wl, seq_ids[:n_seq], args, local_rank)
            sub.run(8, stages)  # the first steps at a new batch size size the work spaces (hipMalloc) and start the mapping thread's two-step rhythm
            torch.cuda.synchronize()
            n_sw = max(10, min(40, 5120 // n_seq))
            t1 = time.perf_counter()
            sub.run(n_sw, stages)
            torch.cuda.synchronize()
            sweep[str(n_seq)] = round(n_seq * n_sw / (time.perf_counter() - t1), 1)
            sub.close()
            del sub
        sweep["unit"] = "frames/s of the whole loop with that many sequences per step on this one GPU (8 warm-up steps, then 20 / 40 / 40 timed steps for 256 / 128 / 64 sequences)"

    # ---- configs[3]: the inertial configuration (IMU pre-integration, pose-inertial optimisation, UndistortPcl + ESKF, LocalLVIBA), the same
    # sequences batched the same way, with its own roofline pass, single-sequence line and CPU leg ----
    inertial = None
    if rank == 0 and not args.no_extra_lines and not args.front_end_only:
        il = InertialLoop(wl, seq_ids, args, local_rank)
        lvi_large = il.lvi_large
        il.run(2, stages)
        torch.cuda.synchronize()
        n_i = max(4, min(args.steps, 10))
        ba_i0 = il.ba_windows_done
        t1 = time.perf_counter()
        il.run(n_i, stages)
        torch.cuda.synchronize()
        dti = time.perf_counter() - t1
        thread_ms_i = {k: round(v, 3) for k, v in il.thread_ms.items()}
        ba_windows_i = il.ba_windows_done - ba_i0
        stats_i = il.inertial_stats()
        n_pi = 4
        pkg.capi.profile_enable(True)
        ba_i1 = il.ba_windows_done
        il.run(n_pi, stages)
        torch.cuda.synchronize()
        pkg.capi.profile_enable(False)
        report_i = pkg.capi.profile_report()
        rf_i, table_i, kms_i = roofline_from_profile(report_i, algorithmic_work_inertial(wl, il, nkp, (il.ba_windows_done - ba_i1) / n_pi), peaks, n_pi)
        rf_i["all_kernels"] = table_i
        rf_i["kernel_ms_per_step_all_streams"] = round(kms_i, 3)
        rf_i["traffic"] = pmc_traffic(rf_i["kernel"], "*_pmc_traffic_inertial.json")
        single_i = single_inertial_child
        if single_i is None:  # no child (under a profiler, or it failed): in this process, after the batched loops
            one_i = InertialLoop(wl, [seq_ids[0]], args, local_rank)
            one_i.run(8, stages)
            torch.cuda.synchronize()
            n1 = 48
            t1 = time.perf_counter()
            one_i.run(n1, stages)
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            single_i = {"value": round(n1 / dt1, 2), "unit": "frames/s", "ms_per_frame": round(1e3 * dt1 / n1, 3), "frames": n1,
                        "stage_thread_ms_per_frame": {k: round(v, 3) for k, v in one_i.thread_ms.items()},
                        "process": "in this process (4 hardware queues; the camera-LiDAR single-sequence line runs in a child with 24)"}
            one_i.close()
            del one_i
        cpu_i = None
        if world == 1 and not args.no_cpu_baseline:
            cpu_i = cpu_baseline_inertial(wl, il, args, F)
        inertial = {"value": round(F * n_i / dti, 2), "unit": "frames/s", "ms_per_step": round(1e3 * dti / n_i, 3), "sequences": F, "steps": n_i,
                    "lviba_windows_per_step": round(ba_windows_i / n_i, 2), "stage_thread_ms_per_step_concurrent": thread_ms_i,
                    "lviba_window": "bLarge: 25 keyframes x 4 iterations (LocalMapping.cc:156); IMU-initialised camera path" if lvi_large else "10 keyframes x 10 iterations",
                    "workload": "configs[3], camera-LiDAR-inertial, %d batched sequences, one frame of every sequence per step: ORB + stereo matching, "
                                "the IMU prediction in place of TrackWithMotionModel, SearchLocalPoints + IMU pre-integration + PoseInertialOptimizationLastFrame (batched; the "
                                "camera path with the IMU initialised, Tracking.cc:2746, 2857); LidarInertialProcess for all scans in one call "
                                "(preprocess, forward propagation on the host, UndistortPcl with its time sort on the device, voxel filter, iterated ESKF "
                                "in lock step, max 3 iterations) + map_incremental; LocalLVIBA in lock step (%s, LiDAR edge over "
                                "6 keyframes x 2400 points) every %d-th frame" % (F, "bLarge as LocalMapping.cc:156 decides at > 100 tracked inliers: 25 + 1 keyframes, "
                                "~1240 points, 4 iterations from lambda 1e-2" if lvi_large else "10 + 1 keyframes, ~900 points, 10 iterations", args.kf_interval),
                    "roofline": rf_i, "cpu_baseline": cpu_i,
                    "single_sequence": single_i,
                    **stats_i}
        il.close()
        del il

    # ---- the matrix unit where it is used: 25-keyframe bLarge LocalLVIBA windows (dense f64 MFMA Schur product) ----
    mfma = None
    if rank == 0 and not args.no_extra_lines and not args.front_end_only and peaks is not None:
        mfma = mfma_line(pkg, synthetic, peaks)

    # ---- CPU baseline: the oracle (a port) with the reference's threading ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # a reported baseline, timed at N = 1 only
        cpu = cpu_baseline(wl, args, F, with_ba=bool(loop.ba_batch))

    # ---- optional: one window over all ranks (BASELINE configs[4]); every rank passes the same window ----
    sharded_window = None
    if args.split_ba:
        w = synthetic.ba_window(4242, n_opt=12, n_fix=20, n_points=3000, pose_noise=(0.1, 0.01))
        last = len(w["poses"]) - 1
        win = list(range(last, last - 6, -1))
        clouds = synthetic.ba_window_clouds(w, win, n_points=3000)
        e = pkg.pack_ba_edges(w["edges"])
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(pkg.capi.RcclComm.unique_id()), dtype=torch.uint8))
        if dist is not None:
            dist.broadcast(uid, 0)
        comm = pkg.capi.RcclComm(uid.cpu().numpy().tobytes(), rank, world)
        shard = comm.shard()
        t_sh, t_one = [], []
        for _ in range(4):
            barrier()
            t_a = time.perf_counter()
            got = pkg.capi.local_lv_bundle_adjustment_sharded(shard, w["poses"], w["fixed"], w["points"], e, w["cam"], win_pose=win, clouds=clouds,
                                                              Tcl7=synthetic.TCL7, weight=1.0)
            t_sh.append(time.perf_counter() - t_a)
            t_a = time.perf_counter()
            one_gpu = pkg.capi.local_lv_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"], win, clouds, synthetic.TCL7, 1.0)
            t_one.append(time.perf_counter() - t_a)
        comm.close()
        sharded_window = {"ranks": world, "ms_per_window": round(1e3 * min(t_sh[1:]), 3), "single_gpu_ms_per_window": round(1e3 * min(t_one[1:]), 3),
                          "iterations/trials": [int(got[4].iterations), int(got[4].trials)], "allreduces_per_window": 2 * int(got[4].trials) + int(got[4].iterations) + 3,
                          "max_pose_difference_vs_single_gpu": float(np.abs(got[0] - one_gpu[0]).max()),
                          "note": "landmarks l % ranks; per LM trial one sum of [S | b_schur | b_p | status] and one of [scale, chi2, stop, status]"}

    if rank == 0:
        line = {
            "metric": METRIC, "value": round(total_sequences * args.steps / elapsed, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u8 (ORB, matching), f32 (LiDAR), f64 (optimisation)", "data": "synthetic", "ranks_seen": ranks_seen,
            "config": {
                "workload": ("configs[1]: KITTI-00 camera-LiDAR front end" if args.front_end_only else
                             "configs[1]+[2]: KITTI camera-LiDAR loop, front end + HIP local LV-BA every %d-th frame" % args.kf_interval) +
                            ", %d batched sequences in all (%s scaling: %s), one frame of every sequence per step (step j: frame j %% %d of the sequence's drive) -- stereo ORB (2 x 1242x375, 2000 "
                            "features, 8 levels, FAST 20/7), stereo matching, TrackWithMotionModel + TrackLocalMap (projection matching + pose "
                            "optimisation, twice), LiDAR fov_segment / preprocess / voxel 0.5 m / 5-NN plane features against the sequence's own "
                            "map / map_incremental (64-beam scan, ~130k returns)" % (
                                total_sequences, args.scaling, "the list is dealt over the ranks" if args.scaling == "strong" else "%d per rank" % args.frames, len(loop.frame_sets)) +
                            ("" if args.front_end_only else
                             ", LocalLVBundleAdjustment over 64 distinct windows (4-24 free / 2-40 fixed keyframes, 500-6000 points, 0-15 % outliers, LiDAR edge over "
                             "0 / 3-6 keyframes, heavy LiDAR edges with rejected steps, interrupted windows: config.ba)" if wl.ba_mix == "varied" else
                             ", LocalLVBundleAdjustment (12 free + 20 fixed keyframes, ~2500 points, ~26k stereo "
                             "edges, LiDAR plane edge over 6 keyframes x 3000 points)"),
                "stage_threads": "ORB extraction | stereo matching + TrackWithMotionModel | TrackLocalMap | LiDAR front end + map maintenance | local "
                                 "mapping, each on its own host thread and HIP stream (the reference's tracking / LiDAR / local-mapping threads; with batched "
                                 "sequences the tracking thread's two halves are pipeline stages over three feature buffers); a step = every stage has "
                                 "processed one batch" + ("; local mapping follows the tracking thread and takes the keyframes of the steps tracked since its last call, at most %d "
                                                          "steps' per call" % (4 if loop.ba_batch4 is not None else 2) if loop.ba_batch2 is not None else "; the LiDAR stream has high priority") +
                                 ("; local mapping = %d bundle-adjustment engines (tc2li_ba_engine: running lock-step queues), every sequence a ticket stream of its "
                                  "own: a window is submitted when its keyframe's step has been tracked and the sequence's previous window has come back" % len(loop.ba_engines)
                                  if loop.ba_engine is not None else "") +
                                 ("; local mapping = %d mapping workers, each taking the next tracked step's windows as ONE lock-step group "
                                  "(tc2li_local_bundle_adjustment_batch_group): up to that many steps in flight, no sequence with two windows in flight" % len(loop.ba_step_workers)
                                  if loop.ba_step_workers else "") +
                                 ("; local mapping = %d mapping workers, each a lock-step group of its own (tc2li_local_bundle_adjustment_batch_group), taking the steps' "
                                  "windows chunk by chunk (%s windows)" % (len(loop.ba_workers), "/".join(str(n) for n in loop.ba_chunk_sizes)) if loop.ba_workers else ""),
                "sequences_total": total_sequences, "frames_per_step_per_gpu": F, "images_per_step_per_gpu": loop.n_img,
                "frames_cycled_per_sequence": len(loop.frame_sets),
                "ba_windows_per_step_per_gpu": round(ba_windows_timed / args.steps, 3), "host_threads_gpu_path": host_budget,
                "keypoints_per_image": round(nkp, 1), "stereo_matches_per_frame": round(float(np.mean((st_out[1] > 0).sum(1))), 1),
                # (the inertial loop's camera path: the IMU prediction in place of TrackWithMotionModel, SearchLocalPoints in place of TrackLocalMap's visual half)
                "motion_model_matches/inliers_per_frame": None if trk_out is None else [round(float(np.mean(trk_out[2])), 1), round(float(np.mean(trk_out[3])), 1)],
                "local_map_points/matches/inliers_per_frame": [int(np.mean(np.diff(loop.local_off))), round(float(np.mean(loop.slp_outs[0][1])), 1), None]
                if tlm_out is None else [int(np.mean(np.diff(loop.local_off))), round(float(np.mean(tlm_out[3])), 1), round(float(np.mean(tlm_out[4])), 1)],
                "scan_points_raw/preprocessed/downsampled/selected": lid_mean,
                "map_points_per_sequence_start/end": [loop.map_points0, map_points_end], "map_incremental_to_add/no_need_last_step": loop.map_adds,
                "ba_options": pkg.capi.ba_options(),
                "ba": None if not loop.ba_batch else dict(mix=wl.ba_mix, **{k: v for k, v in (loop.ba_mix_summary(wl) or {}).items()
                                                                             if k.endswith("min_max") or k.endswith("mean_max") or k.startswith("windows_") or k == "distinct_windows"})},
            "roofline": roofline, "cpu_baseline": cpu, "single_sequence": single, "host_fed": host_fed, "inertial_config": inertial, "mfma_config": mfma, "sequences_per_gpu_sweep": sweep,
            "host_budget_sweep": host_budget_sweep, "value_uniform": value_uniform, **({"sharded_window": sharded_window} if sharded_window else {}),
            "stage_thread_ms_per_step_concurrent": {k: round(v, 3) for k, v in thread_ms.items()},
            "stage_wall_ms_alone": {k: round(1e3 * v, 3) for k, v in wall.items()},
            "track_calls_ms_last_step": dict(zip(("stereo_match_batch", "track_motion_model_batch", "track_local_map_batch"), [round(v, 3) for v in loop.track_ms])),
        }
        # BASELINE.json's 1-GPU configs describe ONE sequence: that number and its CPU leg side by side, next to the batched `value`
        if single is not None:
            c1 = (cpu or {}).get("single_sequence")
            line["config"]["baseline_config_value"] = {
                "workload": "configs[1]+[2] as BASELINE.json states them: one KITTI-shaped sequence (F = 1), front end + local LV-BA every %d-th frame" % args.kf_interval,
                "value": single["value"], "unit": "frames/s", "cpu_baseline": None if not c1 else c1["value"],
                "vs_cpu": None if not c1 else round(single["value"] / c1["value"], 2),
                "note": "`value` of this line is the same loop for %d sequences per step (multi-sequence operation, configs[4]'s workload on one GPU)" % total_sequences}
        if args.full_line:
            print(json.dumps(line))
        else:
            # the child legs (--no-extra-lines: single sequence, sweep) have nothing worth a side file
            detail = None if args.no_extra_lines and not args.detail_out else write_detail(line, args.detail_out)
            print(json.dumps(compact_line(line, detail)))
        sys.stdout.flush()
    loop.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0


def run_and_leave():
    """main(), then an orderly end: every stage thread joined, the library's pools joined and its process-wide work spaces released
    (tc2li_shutdown) while the HIP runtime is alive, then a plain sys.exit -- the same path under rocprofv3, whose output is written by
    exit handlers."""
    rc = 1
    try:
        rc = main()
    finally:
        for loop in list(_OPEN_LOOPS):
            try:
                loop.close()
            except Exception:  # noqa: BLE001
                pass
        pkg = sys.modules.get("tc2li_slam_amd")
        if pkg is not None and getattr(pkg, "capi", None) is not None and pkg.capi.lib_loaded():
            pkg.capi.shutdown()
        sys.stdout.flush()
        sys.stderr.flush()
    return rc


if __name__ == "__main__":
    sys.exit(run_and_leave())
