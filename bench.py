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
    # config 3: textured multi-object scene, two area lights, glossy mirror, depth of field, 64 spp (Whitted shading)
    "textured_1080p_64spp": dict(kind="textured", n=188, width=1920, height=1080, spp=64, cpu=dict(width=480, height=270, spp=16)),
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
    if w["kind"] == "textured":
        return scenegen.write_textured_scene(out_dir, name=name, n=w["n"], width=dims["width"], height=dims["height"], spp=dims["spp"])
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
    ren.set_counters(True)   # counted mode: rays exactly as the reference's algorithm casts them
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


def kernel_bytes(st, timed):
    """Algorithmic bytes per kernel class for one frame.  `st`: the frame's work counters (counted frame, every light
    sample evaluated and traced on its own), `timed`: the statistics of a timed frame (sphere emitters deferred: one
    shadow ray per shading point, samples evaluated where it arrives).  Traversal: 32 B per node visit, 36 B per
    triangle test, 64 B per sphere test, 24 B per plane test, 32 B ray read + 16 B hit write per ray.  Shading: 64 B
    material + 12 B framebuffer per shaded hit, 16 B per texel fetch.  Queue records at their stored size: path 52 B,
    vertex 112 B, shadow ray 32 B + 4 B visibility."""
    camera = st.primary_rays
    paths = st.primary_rays + st.secondary_rays
    rays = timed.shadow_rays_traced
    return {
        "generate": 64 * camera,
        "trace_closest": st.traversal_bytes(False),
        "shade": 60 * paths + 76 * st.shaded_hits + 16 * st.texel_fetches + 112 * st.shaded_hits + 52 * st.secondary_rays,
        # deferred emitters: position in, ray out; the gather reads 32 B of every vertex, the visibility, the whole
        # vertex where the emitter is visible (bounded here by every vertex), and adds 12 B to the slot
        "light_setup": 16 * st.shaded_hits + 32 * rays,
        "trace_any": (st.traversal_bytes(True) - 12 * st.shadow_rays_traced) * (rays / max(1, st.shadow_rays_traced)),
        "light_gather": (32 + 4 + 112 + 12) * st.shaded_hits,
        "resolve": 12 * camera + 12 * camera,
    }


PMC_KERNEL_NAMES = {"generate": "generate_kernel", "trace_closest": "trace_kernel<false, false, true>", "trace_any": "trace_kernel<true, false, true>", "shade": "shade_kernel<true>",
                    "light_setup": "light_setup_kernel", "light_gather": "light_gather_kernel", "resolve": "resolve_kernel"}


def pmc_traffic(workload, kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 counter passes over this benchmark
    (tools/pmc_traffic_bench.sh: FETCH_SIZE and WRITE_SIZE in separate passes, reads doubled as gfx950 requires).
    None when there is no pass for this workload or kernel (a counter pass cannot run inside the timed program)."""
    path = os.path.join(ROOT, "profiles", "r1_pmc_traffic_bench.json")
    try:
        doc = json.load(open(path))
    except (OSError, ValueError):
        return None
    if doc.get("workload") != workload or kernel not in PMC_KERNEL_NAMES:
        return None
    entry = doc.get("kernels", {}).get(PMC_KERNEL_NAMES[kernel])
    return entry.get("traffic_bytes_per_launch") if entry else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("RT_BENCH_WORKLOAD", "pt1m_4k_256spp"), choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-frames-only", action="store_true",
                    help="profiling runs: skip the counted frame and the frame without ray sharing (their launches would mix "
                         "into the per-kernel averages); the line then carries the timing only")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --one-device rehearses the multi-rank path on a single GPU")
    ap.add_argument("--one-device", action="store_true", help="every rank uses GPU 0 (rehearsal only)")
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
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

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
    ren.set_profiling(True)   # HIP events around every kernel launch, on the library's own stream
    traced = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if world > 1:
            fb.zero_()
        stats = step()
        traced += stats.primary_rays + stats.secondary_rays + stats.shadow_rays_traced
    sync()
    elapsed = time.perf_counter() - t0
    ktimes = ren.kernel_times()
    ren.set_profiling(False)

    if args.timed_frames_only:
        if rank == 0:
            print(json.dumps({"note": "timed frames only (profiling run)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": elapsed * 1e3 / args.steps, "config": {"workload": args.workload},
                              "kernel_ms_per_frame": {k: v[1] / args.steps for k, v in ktimes.items()},
                              "launches_per_frame": {k: v[0] / args.steps for k, v in ktimes.items()}}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    # one extra, untimed frame with the work counters on: algorithmic bytes for the roofline.  In this mode every
    # light sample's shadow ray is traversed on its own, which is also the frame time without ray sharing.
    ren.set_counters(True)
    _, cs = ren.render(cam, out=fb.data_ptr(), rank=rank, world=world)
    ren.set_counters(False)
    # rays of a frame as the reference's algorithm casts them: the counted frame's (same camera, same seed, same tiles);
    # the timed frames do not evaluate the light samples of shading points that a sphere emitter's shadow ray misses
    rays = cs.total_rays * args.steps
    unshared_ms = None
    if world == 1:
        os.environ["RT_SHARE_SHADOW_RAYS"] = "0"
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ren.render(cam, out=fb.data_ptr())
        torch.cuda.synchronize()
        unshared_ms = (time.perf_counter() - t1) * 1e3
        del os.environ["RT_SHARE_SHADOW_RAYS"]

    t = torch.tensor([elapsed, float(rays), float(traced)], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rays = float(tsum[1])
        traced = float(tsum[2])
    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        # dominant kernel = the kernel class with the largest device time in the timed steps
        name = max(ktimes, key=lambda k: ktimes[k][1])
        launches, total_ms = ktimes[name]
        per_frame = kernel_bytes(cs, stats)   # algorithmic bytes of each kernel class in one frame
        launches_per_frame = launches / args.steps
        bytes_per_launch = per_frame[name] / launches_per_frame
        ms_per_launch = total_ms / launches
        achieved = bytes_per_launch / (ms_per_launch * 1e-3) / 1e9
        traffic = pmc_traffic(args.workload, name) if world == 1 else None
        line = {
            "metric": "Mrays/s (primary+shadow+secondary), whole job", "value": rays / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "resolution": f"{cam.resx}x{cam.resy}", "spp": cam.num_samples,
                       "triangles": sc.desc.n_triangles, "bvh_nodes": sc.desc.n_bvh_nodes,
                       "renderer": "path tracing" if cam.path_tracing else "whitted", "sharding": f"tiles32/{world}"},
            # rays as the reference's algorithm casts them (the unit the CPU baseline is counted in) and the
            # traversals actually run: light samples of a shading point with bit-identical shadow rays share one
            "ray_accounting": "rays as the reference's algorithm casts them (oracle counters), the unit of cpu_baseline; "
                              "light samples of a shading point with bit-identical shadow rays share one traversal: "
                              "traversals_per_frame / value_traversals count traversals run, without_ray_sharing times "
                              "a frame with sharing off",
            "rays_per_frame": rays / args.steps, "traversals_per_frame": traced / args.steps,
            "value_traversals": traced / elapsed / 1e6,
            "roofline": {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # the algorithmic bytes are served largely by LDS, L1 and L2, so their rate may pass the HBM peak;
                         # the HBM-side rate is the measured traffic over the same launch time
                         "traffic_rate": traffic / (ms_per_launch * 1e-3) / 1e9 if traffic else None,
                         "traffic_frac": traffic / (ms_per_launch * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None,
                         "kernel_ms_per_launch": ms_per_launch, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "launches_per_frame": launches_per_frame,
                         "kernel_ms_per_frame": {k: v[1] / args.steps for k, v in ktimes.items()}},
        }
        if unshared_ms:
            line["without_ray_sharing"] = {"ms_per_step": unshared_ms, "value": cs.total_rays / unshared_ms / 1e3}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(R, scenegen, w, scene_dir, None)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
