#!/usr/bin/env python3
"""bench.py -- frames/s of the TC2LI-SLAM per-frame front end on MI355X (BASELINE.json metric, configs[1]).

A step = one pass of the hot path over one batch of `--frames` synthetic KITTI-00-sized frames that are already
resident in HBM: stereo ORB extraction (2 x 1242x375) -> stereo matching -> LiDAR preprocess -> voxel filter ->
5-NN + plane-fit feature extraction (one 64-beam scan per frame against a resident map).
N > 1: one process per GPU; torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of the
elapsed time -- frames are independent units, so the path shards with no data-path collective ("weak": every rank
processes its own `--frames` frames per step).

Prints ONE JSON line on rank 0; DESIGN.md section "Measurement" explains how roofline / cpu_baseline are derived.
"""
import argparse
import ctypes as C
from concurrent.futures import ThreadPoolExecutor
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def level_dims(w, h, nlevels=8, scale=1.2):
    dims, s = [], np.float32(1.0)
    for _ in range(nlevels):
        inv = np.float32(1.0) / s
        dims.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
        s = np.float32(np.float64(s) * np.float64(np.float32(scale)))
    return dims


def algorithmic_bytes(w, h, nkp):
    """Compulsory HBM bytes per image of each ORB kernel group."""
    px = [a * b for a, b in level_dims(w, h)]
    return {
        "pyramid": sum(px[:-1]) + sum(px[1:]),           # read levels 0..6, write levels 1..7
        "fast": sum(px) + 4 * 15000,                      # read every level once, write the candidate list
        "blur": 2 * sum(px),                              # read + write every level
        "orient_describe": nkp * (709 + 512 + 64 + 16),  # patch gathers + descriptor/angle/key out
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=32, help="frames (stereo pair + scan) per step per GPU")
    ap.add_argument("--unique", type=int, default=8, help="distinct synthetic frames rendered (tiled to --frames)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU-oracle baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    ge.build_native()
    import tc2li_loader
    pkg = tc2li_loader.load()
    from tc2li_slam_amd import synthetic

    if not torch.cuda.is_available() or pkg.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    W, H = synthetic.WIDTH, synthetic.HEIGHT
    F, U = args.frames, min(args.unique, args.frames)
    bf = np.float32(synthetic.BF)
    b = np.float32(bf / np.float32(synthetic.FX))

    # ---- synthetic workload: U distinct frames (scene seed per rank), tiled to F --------------------------------
    scene = synthetic.Scene(1000 * rank)
    uniq_img = np.empty((U, 2, H, W), np.uint8)
    scans = []
    for f in range(U):
        sc = synthetic.Scene(1000 * rank + f)
        uniq_img[f, 0], _ = sc.render(0.0, W, H, noise_seed=1)
        uniq_img[f, 1], _ = sc.render(synthetic.BASELINE, W, H, noise_seed=2)
        scans.append(synthetic.lidar_scan(scene, f + 1))
    tile = [f % U for f in range(F)]
    frames = uniq_img[tile]
    n_img = 2 * F
    dev_img = torch.from_numpy(frames.reshape(n_img, H, W)).cuda()
    raw = np.concatenate([scans[t] for t in tile])
    raw_offs = np.concatenate([[0], np.cumsum([len(scans[t]) for t in tile])]).astype(np.int32)
    dev_raw = torch.from_numpy(raw.view(np.uint8)).cuda()
    states = np.stack([pkg.pack_lidar_state(*synthetic.lidar_state(t + 1)[:2]) for t in tile])
    stream = torch.cuda.current_stream().cuda_stream

    ext = pkg.OrbExtractor(max_width=W, max_height=H, max_images=n_img)
    lidar = pkg.LidarFrontEnd(max_points_per_scan=int(max(len(s) for s in scans)), max_scans=F)
    # resident map: the world-frame down-sampled scan of frame 0 (what ikdtree.Build gets, LidarFrontEnd.cpp:918-931)
    boot = pkg.LidarMap()
    scan0 = synthetic.lidar_scan(scene, 0)
    down0 = lidar.voxel_filter(lidar.process(scan0))
    boot.Build(down0[:8])
    world0 = lidar.feature_extraction(boot, down0, pkg.pack_lidar_state(*synthetic.lidar_state(0)[:2]))["world"]
    lmap = pkg.LidarMap()
    lmap.Build(world0)
    maps = [lmap] * F
    orb_out = st_out = None
    lidar_counts = None

    # Camera path and LiDAR path are independent until SyncWithLidar (SF/src/Tracking.cc:1565): like the reference's
    # camera thread and LiDAR thread (src/examples/camera_lidar.cc:84) they run concurrently, each on its own stream.
    lidar_stream = torch.cuda.Stream()
    pool = ThreadPoolExecutor(max_workers=1)

    def lidar_path():
        return lidar.frontend_batch(dev_raw.data_ptr(), raw_offs, maps, states, stream=lidar_stream.cuda_stream,
                                    want_points=False)[0]

    def step():
        nonlocal orb_out, st_out, lidar_counts
        fut = pool.submit(lidar_path)
        orb_out = ext.extract_batch_dev(dev_img.data_ptr(), n_img, W, H, W, W * H, stream=stream, out=orb_out)
        st_out = pkg.stereo_match_batch(ext, F, float(bf), float(b), stream=stream, out=st_out)
        lidar_counts = fut.result()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stage_ms = ext.last_timings().astype(float)
    nkp = float(np.mean(orb_out[2]))
    n_match = float(np.mean((st_out[1] > 0).sum(1)))

    # stage wall times of one more step (host clock, each stage synchronises at its end)
    t_a = time.perf_counter()
    ext.extract_batch_dev(dev_img.data_ptr(), n_img, W, H, W, W * H, stream=stream, out=orb_out)
    t_b = time.perf_counter()
    pkg.stereo_match_batch(ext, F, float(bf), float(b), stream=stream, out=st_out)
    t_c = time.perf_counter()
    lidar.frontend_batch(dev_raw.data_ptr(), raw_offs, maps, states, stream=stream, want_points=False)
    t_d = time.perf_counter()

    # ---- roofline of the dominant kernel: per-kernel HIP-event durations, kernels serialised on one stream ----
    ext.set_profiling(True)
    prof = []
    for _ in range(3):
        ext.extract_batch_dev(dev_img.data_ptr(), n_img, W, H, W, W * H, stream=stream, out=orb_out)
        prof.append(ext.last_timings().astype(float))
    ext.set_profiling(False)
    prof = np.mean(prof, 0)
    alg = algorithmic_bytes(W, H, nkp)
    kern_ms = {"pyramid": prof[0], "fast": prof[1], "blur": prof[3], "orient_describe": prof[4]}
    dom = max(kern_ms, key=lambda k: kern_ms[k])
    launches = {"pyramid": 7, "fast": 1, "blur": 8, "orient_describe": 1}[dom]
    bytes_per_launch = alg[dom] * n_img / launches
    achieved = bytes_per_launch / (kern_ms[dom] / launches * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": {"fast": "k_fast_cells", "blur": "k_blur7", "pyramid": "k_resize_linear",
                                           "orient_describe": "k_orient_describe"}[dom],
                "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 5),
                "traffic": None, "avg_launch_ms": round(kern_ms[dom] / launches, 5),
                "algorithmic_bytes_per_launch": int(bytes_per_launch),
                "all_kernels_ms": {k: round(v, 4) for k, v in kern_ms.items()},
                "all_kernels_GBps": {k: round(alg[k] * n_img / (kern_ms[k] * 1e-3) / 1e9, 2) for k in kern_ms if kern_ms[k] > 0}}

    # ---- CPU baseline: the oracle (a port) with the reference's threading -----------------------------------------
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import pyoracle
        pyoracle.build()
        L = pyoracle.lib()
        ol, orr = pyoracle.OrbOracle(), pyoracle.OrbOracle()
        tree = pyoracle.KdTree(world0)
        nsel = C.c_int(0)
        done, tcpu0 = 0, time.perf_counter()
        while done < 4 or (time.perf_counter() - tcpu0 < args.cpu_seconds and done < 400):
            t = tile[done % F]
            L.oracle_frontend_frame(ol._h, orr._h, uniq_img[t, 0].ctypes.data, uniq_img[t, 1].ctypes.data, W, H, float(bf),
                                    float(b), scans[t].ctypes.data, len(scans[t]), tree._h, states[done % F].ctypes.data,
                                    C.byref(nsel))
            done += 1
        tcpu = time.perf_counter() - tcpu0
        cpu = {"value": round(done / tcpu, 3), "unit": "frames/s", "cores": 3, "kind": "port",
               "sample": "%d synthetic frames of the same workload in %.1f s; threads as in the reference: left/right ORB on 2 "
                         "threads + stereo match, LiDAR front end on a 3rd thread" % (done, tcpu),
               "host_cpus": os.cpu_count()}

    if rank == 0:
        total_frames = F * args.steps * world
        line = {
            "metric": "frames/sec (ORB+LiDAR front-end + local BA) on KITTI-00, 1/2/4/8 GPU; ATE vs ref",
            "value": round(total_frames / elapsed, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "configs[1]: KITTI-00 camera-LiDAR front end on 1xMI355X per rank -- HIP stereo ORB "
                                   "(2 x 1242x375, 2000 features, 8 levels, FAST 20/7) + stereo matching + LiDAR "
                                   "preprocess/voxel 0.5 m/5-NN plane features (64-beam scan, ~130k returns); local BA on CPU "
                                   "is not part of this configuration",
                       "frames_per_step_per_gpu": F, "images_per_step_per_gpu": n_img,
                       "keypoints_per_image": round(nkp, 1), "stereo_matches_per_frame": round(n_match, 1),
                       "scan_points_raw/preprocessed/downsampled/selected": [int(np.mean(np.diff(raw_offs)))] +
                       [int(v) for v in np.mean(lidar_counts, 1)],
                       "map_points": int(lmap.size())},
            "roofline": roofline, "cpu_baseline": cpu,
            "stage_wall_ms_per_step": {"orb_extract_batch": round(1e3 * (t_b - t_a), 3), "stereo_match_batch": round(1e3 * (t_c - t_b), 3),
                                       "lidar_frontend_batch": round(1e3 * (t_d - t_c), 3)},
            "orb_stage_ms_last_step": {"pyramid": round(stage_ms[0], 4), "fast": round(stage_ms[1], 4),
                                       "compact": round(stage_ms[2], 4), "blur": round(stage_ms[3], 4),
                                       "orient_describe": round(stage_ms[4], 4), "host_quadtree": round(stage_ms[5], 4),
                                       "host_until_quadtree": round(stage_ms[6], 4), "call_total": round(stage_ms[7], 4)},
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
