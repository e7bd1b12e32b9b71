#!/usr/bin/env python3
"""Benchmark of the rendering hot path: whole frames of a synthetic scene on N MI355X GPUs.

A step = one complete frame of the workload (camera rays -> BVH traversal -> shading -> HDR
framebuffer) with the scene already resident in HBM.  With N > 1 (one process per GPU under
torch.distributed / RCCL) the frame is sharded by 32x32 tiles over the ranks and gathered on rank 0
inside the timed region (strong scaling: the frame is fixed as N grows).

Prints ONE JSON line on rank 0:
  value     = rays (primary + shadow + secondary) of all ranks / max-over-ranks wall time, in Mrays/s
  roofline  = algorithmic traversal+shading bytes of the render kernels / their device time vs HBM peak
  cpu_baseline = the reference renderer (oracle/_ref/raytracer, built from the reference sources)
              or, when absent, the in-repo CPU restatement, timed on this host on a bounded sample
"""
import argparse
import ctypes as C
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak

WORKLOADS = {
    # config 4: 1M+ triangle mesh, path traced, 4K, 256 spp
    "pt1m_4k_256spp": dict(kind="pt", n=710, width=3840, height=2160, spp=256, max_depth=4,
                           cpu=dict(width=480, height=270, spp=16)),
    # config 2: ~70K triangle mesh, Blinn-Phong + shadows, 1080p, 1 spp
    "mesh70k_1080p": dict(kind="mesh", n=188, width=1920, height=1080, cpu=dict(width=1920, height=1080)),
}


def make_scene(scenegen, out_dir, w, name, cpu=False):
    dims = dict(w)
    if cpu:
        dims.update(w["cpu"])
    if w["kind"] == "pt":
        return scenegen.write_path_traced_scene(out_dir, name=name, n=w["n"], width=dims["width"], height=dims["height"],
                                                spp=dims["spp"], max_depth=w["max_depth"])
    return scenegen.write_mesh_scene(out_dir, name=name, n=w["n"], width=dims["width"], height=dims["height"])


def cpu_baseline(R, scenegen, w, scene_dir, rays_per_unit_hint):
    """Times the CPU renderer on a bounded sample of the workload; returns the JSON object."""
    threads = max(1, min(16, os.cpu_count() or 1))
    path = make_scene(scenegen, scene_dir, w, "cpu_sample", cpu=True)
    dims = dict(w)
    dims.update(w["cpu"])
    sample = f"{dims['width']}x{dims['height']}" + (f"x{dims['spp']}spp" if "spp" in dims else "x1spp") + " of the same scene"
    sc = R.Scene(path)
    cam = sc.camera(0)
    # ray count of the sample: from the device counters of the same sample (the reference has no ray counter)
    ren = R.Renderer(sc)
    _, st = ren.render(cam)
    rays = st.total_rays
    ren.close()
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "raytracer")
    if os.path.exists(ref_bin):
        with tempfile.TemporaryDirectory() as cwd:
            try:
                out = subprocess.run([ref_bin, path, "0", str(threads)], cwd=cwd, capture_output=True, text=True, timeout=900)
                m = re.search(r"rendering done: (\d+) ms", out.stdout)
                if m and int(m.group(1)) > 0:
                    ms = float(m.group(1))
                    return {"value": rays / ms / 1e3, "unit": "Mrays/s", "cores": threads, "kind": "reference",
                            "sample": sample, "ms": ms}
            except (subprocess.SubprocessError, OSError):
                pass
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    orc = Oracle(sc)
    t0 = time.perf_counter()
    _, so = orc.render(cam, threads=threads)
    ms = (time.perf_counter() - t0) * 1e3
    return {"value": so.total_rays / ms / 1e3, "unit": "Mrays/s", "cores": threads, "kind": "port", "sample": sample, "ms": ms}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("RT_BENCH_WORKLOAD", "pt1m_4k_256spp"), choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import raytracer_amd as R
    from raytracer_amd import scenegen
    from raytracer_amd.sharding import gather_frame

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rendering core has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    w = WORKLOADS[args.workload]
    scene_dir = os.path.join(tempfile.gettempdir(), f"rt_bench_scenes_{os.getuid()}_{rank}")
    path = make_scene(scenegen, scene_dir, w, args.workload)
    sc = R.Scene(path)
    cam = sc.camera(0)
    ren = R.Renderer(sc)   # scene resident in HBM from here on

    # full-frame device framebuffer (torch is used for device memory and the collective only)
    fb = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.float32, device="cuda")
    gathered = None

    def step():
        _, st = ren.render(cam, out=fb.data_ptr(), rank=rank, world=world)
        gather_frame(fb, world)   # every rank wrote only its own tiles into a zeroed frame
        return st

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        if world > 1:
            fb.zero_()
        step()
    sync()
    kernel_ms = 0.0
    rays = 0
    alg_bytes = 0
    t0 = time.perf_counter()
    stats = None
    for _ in range(args.steps):
        if world > 1:
            fb.zero_()
        stats = step()
        kernel_ms += stats.kernel_ms
        rays += stats.total_rays
    sync()
    elapsed = time.perf_counter() - t0

    # one extra, untimed frame with traversal counters on: algorithmic bytes per frame for the roofline
    ren.set_counters(True)
    _, sc_stats = ren.render(cam, out=fb.data_ptr(), rank=rank, world=world)
    ren.set_counters(False)
    alg_bytes = sc_stats.algorithmic_bytes

    t = torch.tensor([elapsed, float(rays), kernel_ms, float(alg_bytes)], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rays = float(tsum[1])
    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        kernel_ms_per_launch = kernel_ms / args.steps
        achieved = alg_bytes / (kernel_ms_per_launch * 1e-3) / 1e9 if kernel_ms_per_launch > 0 else 0.0
        line = {
            "metric": "Mrays/s (primary+shadow+secondary), whole job", "value": rays / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "resolution": f"{cam.resx}x{cam.resy}", "spp": cam.num_samples,
                       "triangles": sc.desc.n_triangles, "bvh_nodes": sc.desc.n_bvh_nodes,
                       "renderer": "path tracing" if cam.path_tracing else "whitted", "sharding": f"tiles32/{world}"},
            "rays_per_frame": rays / args.steps,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel_ms_per_launch": kernel_ms_per_launch, "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(R, scenegen, w, scene_dir, None)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
