#!/usr/bin/env python3
"""bench.py -- frames/s of the TC2LI-SLAM per-frame front end on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of `--frames` synthetic KITTI-00-sized stereo frames that
are already resident in HBM.  N > 1: one process per GPU (torch.distributed over RCCL is used only for the
barrier and the max-over-ranks of the elapsed time -- frames are independent, so the path shards with no
data-path collective: weak scaling, every rank processes its own `--frames` frames per step).

Prints ONE JSON line on rank 0; see DESIGN.md section "Measurement" for how roofline/cpu_baseline are derived.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def level_dims(w, h, nlevels=8, scale=1.2):
    dims, s = [], np.float32(1.0)
    for l in range(nlevels):
        inv = np.float32(1.0) / s
        dims.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
        s = np.float32(np.float64(s) * np.float64(np.float32(scale)))
    return dims


def algorithmic_bytes(w, h, nkp):
    """Compulsory HBM bytes per image of each ORB kernel group."""
    d = level_dims(w, h)
    px = [a * b for a, b in d]
    return {
        "pyramid": sum(px[:-1]) + sum(px[1:]),          # read levels 0..6, write levels 1..7
        "fast": sum(px) + 4 * 8000,                      # read every level once, write ~candidates
        "blur": 2 * sum(px),                             # read + write every level
        "orient_describe": nkp * (709 + 512 + 32 + 12),  # patch gathers + descriptor/angle out
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=32, help="stereo frames per step per GPU")
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
    F = args.frames
    uniq = synthetic.stereo_batch(min(args.unique, F), seed=1000 * rank)  # [U, 2, H, W]
    reps = (F + len(uniq) - 1) // len(uniq)
    frames = np.concatenate([uniq] * reps, 0)[:F]
    n_img = 2 * F
    dev = torch.from_numpy(frames.reshape(n_img, H, W)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    ext = pkg.OrbExtractor(max_width=W, max_height=H, max_images=n_img)
    out = None

    def step():
        nonlocal out
        out = ext.extract_batch_dev(dev.data_ptr(), n_img, W, H, W, W * H, stream=stream, out=out)

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
    kps, desc, counts, mono = out
    nkp = float(np.mean(counts))

    # ---- roofline of the dominant kernel: per-kernel HIP-event durations, kernels serialised on one stream ----
    ext.set_profiling(True)
    prof = []
    for _ in range(3):
        step()
        prof.append(ext.last_timings().astype(float))
    ext.set_profiling(False)
    prof = np.mean(prof, 0)
    alg = algorithmic_bytes(W, H, nkp)
    kern_ms = {"pyramid": prof[0], "fast": prof[1], "blur": prof[3], "orient_describe": prof[4]}
    dom = max(kern_ms, key=lambda k: kern_ms[k])
    launches = {"pyramid": 7, "fast": 1, "blur": 8, "orient_describe": 1}[dom]
    bytes_per_launch = alg[dom] * n_img / launches
    achieved = bytes_per_launch / (kern_ms[dom] / launches * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 5), "traffic": None,
                "avg_launch_ms": round(kern_ms[dom] / launches, 5),
                "all_kernels_ms": {k: round(v, 4) for k, v in kern_ms.items()},
                "all_kernels_GBps": {k: round(alg[k] * n_img / (kern_ms[k] * 1e-3) / 1e9, 2) for k in kern_ms if kern_ms[k] > 0}}

    # ---- CPU baseline: the oracle (a port), left/right on two threads like SF/src/Frame.cc:139-142 -------------
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        import ctypes as C
        from oracle import pyoracle
        pyoracle.build()
        L = pyoracle.lib()
        ol, orr = pyoracle.OrbOracle(), pyoracle.OrbOracle()
        cap = ol.cap
        kl = np.zeros((cap, 6), np.float32); kr = np.zeros((cap, 6), np.float32)
        dl = np.zeros((cap, 32), np.uint8); dr = np.zeros((cap, 32), np.uint8)
        nl, nr = C.c_int(0), C.c_int(0)
        done, tcpu0 = 0, time.perf_counter()
        while done < 4 or (time.perf_counter() - tcpu0 < args.cpu_seconds and done < 400):
            f = frames[done % F]
            L.oracle_orb_extract_pair(ol._h, orr._h, f[0].ctypes.data, f[1].ctypes.data, W, H, W, kl.ctypes.data,
                                      dl.ctypes.data, kr.ctypes.data, dr.ctypes.data, cap, C.byref(nl), C.byref(nr))
            done += 1
        tcpu = time.perf_counter() - tcpu0
        cpu = {"value": round(done / tcpu, 3), "unit": "frames/s", "cores": 2, "kind": "port",
               "sample": "%d synthetic stereo frames (stereo ORB extraction only), L/R on 2 threads, %.1f s" % (done, tcpu),
               "host_cpus": os.cpu_count()}

    if rank == 0:
        total_frames = F * args.steps * world
        line = {
            "metric": "frames/sec (ORB+LiDAR front-end + local BA) on KITTI-00, 1/2/4/8 GPU; ATE vs ref",
            "value": round(total_frames / elapsed, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "configs[1]: KITTI-00-sized stereo front end on 1xMI355X per rank -- HIP stereo ORB "
                                   "extraction (2 x 1242x375, 2000 features, 8 levels, FAST 20/7)",
                       "frames_per_step_per_gpu": F, "images_per_step_per_gpu": n_img,
                       "keypoints_per_image": round(nkp, 1), "stages_in_step": ["orb_left", "orb_right"]},
            "roofline": roofline, "cpu_baseline": cpu,
            "stage_ms_last_step": {"pyramid": round(stage_ms[0], 4), "fast": round(stage_ms[1], 4),
                                   "compact": round(stage_ms[2], 4), "blur": round(stage_ms[3], 4),
                                   "orient_describe": round(stage_ms[4], 4), "host_quadtree": round(stage_ms[5], 4),
                                   "host_until_quadtree": round(stage_ms[6], 4), "call_total": round(stage_ms[7], 4)},
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
