#!/usr/bin/env python3
"""Benchmark of the rendering hot path: whole frames of a synthetic scene on N MI355X GPUs.

A step = one complete frame of the workload (camera rays -> BVH traversal -> shading -> HDR
framebuffer) with the scene already resident in HBM.  With N > 1 (one process per GPU under
torch.distributed / RCCL) the frame is sharded by 32x32 tiles over the ranks and gathered on rank 0
inside the timed region (strong scaling: the frame is fixed as N grows).

Prints ONE JSON line on rank 0:
  value     = BVH traversals actually run (camera + bounce paths + shadow rays traced) of all ranks / max-over-ranks
              wall time, in Mrays/s.  value_reference_equivalent counts the rays the reference's algorithm casts for the
              same image (every light sample its own shadow ray) over the same time: the unit of cpu_baseline.
  roofline  = bytes the dominant kernel moves behind the L2 per launch, measured in this run by two rocprofv3 counter passes
              over one frame (FETCH_SIZE, WRITE_SIZE; child processes, before the timed frames), over its launch time from
              HIP events in the timed frames, against the HBM peak: `frac`.  FETCH_SIZE counts a gathered 64-byte record
              as 64 bytes and a coalesced stream as half its bytes (calibrated on known byte counts), so the trace kernels'
              reads enter as 1 x FETCH_SIZE and the streaming kernels' as 2 x.  `request_slot_frac` prices every read
              request as the 128-byte line it occupies the memory side for; `frac_of_gather_ceiling` is the kernel's request
              rate over that of independent 64-byte gathers measured in this run (tools/calibrate_fetch).  Next to them the
              fixed per-ray byte model, a normaliser
  cpu_baseline = the reference renderer (oracle/_ref/raytracer, built from the reference sources)
              or, when absent, the in-repo CPU restatement, timed on this host on a bounded sample with every CPU this
              process may use
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
L2_PEAK_GBS = 34500.0   # aggregate L2 bandwidth of the eight XCDs

WORKLOADS = {
    # config 4: 1M+ triangle mesh, path traced, 4K, 256 spp
    "pt1m_4k_256spp": dict(kind="pt", n=710, width=3840, height=2160, spp=256, max_depth=4,
                           cpu=dict(width=960, height=540, spp=16)),
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


def host_threads():
    """Threads the CPU leg runs: every CPU this process may use -- its affinity mask, cut to the container's CPU quota and
    to the share the box announces through OMP_NUM_THREADS where those are set (a GPU box shows the whole machine's CPUs
    to a job that owns a fraction of them).  Returns (threads, description of how the number came about)."""
    host = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = host
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, -(-int(q) // int(period)))
    except (OSError, ValueError):
        pass
    share = None
    try:
        share = int(os.environ.get("OMP_NUM_THREADS", ""))
    except ValueError:
        pass
    threads = affinity
    if quota:
        threads = min(threads, quota)
    if share and share > 0:
        threads = min(threads, share)
    return max(1, threads), {"host_cpus": host, "affinity": affinity, "cgroup_quota": quota, "announced_share": share}


REFERENCE_BUILD = ("g++ -std=c++20 -O3 -march=x86-64-v3 -ffast-math -ftree-vectorize -msse4.1 -flto=8 -DNDEBUG: the reference Makefile's "
                   "flags with its -march=native replaced by x86-64-v3 (AVX2 + FMA), because the binary is built in another container "
                   "than it runs in")


def cpu_baseline(R, scenegen, w, scene_dir, rays_per_unit_hint):
    """Times the CPU renderer on a bounded sample of the workload; returns the JSON object."""
    threads, how = host_threads()
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
                            "sample": sample, "ms": ms, "threads": how, "reference_build": REFERENCE_BUILD,
                            "scheduler": "row_dynamic (threading_type 0)"}
            except (subprocess.SubprocessError, OSError):
                pass
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    orc = Oracle(sc)
    t0 = time.perf_counter()
    _, so = orc.render(cam, threads=threads)
    ms = (time.perf_counter() - t0) * 1e3
    return {"value": so.total_rays / ms / 1e3, "unit": "Mrays/s", "cores": threads, "kind": "port", "sample": sample, "ms": ms,
            "threads": how}


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


def pmc_kernel_class(name):
    """Kernel class (the keys of rt_kernel_times) of a kernel name as rocprofv3 prints it."""
    if "trace_kernel" in name:
        return "trace_any" if re.search(r"trace_kernel<\s*true", name) else "trace_closest"
    for k in ("generate", "shade", "light_setup", "light_gather", "resolve"):
        if k + "_kernel" in name:
            return k
    return None


def pmc_pass(counter, workload, out_dir):
    """One rocprofv3 counter pass over one frame of `workload` in a child process (this script with
    --timed-frames-only); returns {kernel class: (sum of the counter over the frame's launches, launches)} or None."""
    import csv
    import shutil
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    shutil.rmtree(out_dir, ignore_errors=True)
    cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", out_dir, "-o", "p", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--timed-frames-only", "--steps", "1", "--warmup", "0", "--workload", workload]
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        child = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            rc = child.wait(timeout=240)
        except subprocess.TimeoutExpired:
            os.killpg(child.pid, 9)   # the group started above: the profiler and its child
            child.wait()
            return None
        if rc != 0:
            return None
        acc, n = {}, {}
        with open(os.path.join(out_dir, "p_counter_collection.csv")) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                k = pmc_kernel_class(row["Kernel_Name"])
                if k:
                    acc[k] = acc.get(k, 0.0) + float(row["Counter_Value"])
                    n[k] = n.get(k, 0) + 1
        return {k: (acc[k], n[k]) for k in acc}
    except (OSError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def measure_traffic(workload):
    """Memory-side traffic per kernel class and launch, from two counter passes of this run (FETCH_SIZE and WRITE_SIZE
    cannot share a pass).  FETCH_SIZE tallies 64 bytes per read request of the L2 to the fabric, whatever the request
    moves; measured on this GPU with known access patterns (tools/calibrate_fetch.sh, profiles/r2_fetch_calibration.json):
    a gather of 64-byte records, of 16-byte pieces and of whole 128-byte lines all run at the same REQUEST rate (50 G/s
    from a 2 GiB table, 60 G/s from a 64 MiB one = 8 TB/s of 128-byte lines), so a request occupies the memory side like a
    128-byte line; a coalesced stream reads 128 bytes per request.  Line-slot bytes = 2 x FETCH_SIZE + WRITE_SIZE;
    requested bytes count a gathered 64-byte record as 64 (1 x FETCH_SIZE).  Returns None when the profiler is not
    available or a pass fails."""
    tmp = os.path.join(tempfile.gettempdir(), f"rt_bench_pmc_{os.getuid()}")
    fetch = pmc_pass("FETCH_SIZE", workload, tmp + "_f")
    if fetch is None:
        return None
    write = pmc_pass("WRITE_SIZE", workload, tmp + "_w")
    if write is None:
        return None
    out = {}
    for k, (kb, launches) in fetch.items():
        wkb = write.get(k, (0.0, launches))[0]
        out[k] = {"launches": launches, "read_requests": kb * 1024.0 / 64.0 / launches, "fetch_size_bytes": kb * 1024.0 / launches,
                  "write_size_bytes": wkb * 1024.0 / launches}
    return out


def fetch_factor(kernel_class):
    """What FETCH_SIZE has to be multiplied by to give bytes, per access class (tools/calibrate_fetch.sh): the trace
    kernels gather 64-byte records, which the counter reports as they are; every other kernel streams its queue
    records, which it reports at half (the microarchitecture guide's gfx950 correction)."""
    return 1.0 if kernel_class.startswith("trace") else 2.0


def gather_ceiling():
    """Rate of independent 64-byte record gathers from a 128 MiB table (the size class of the benchmark's tree: beyond the
    L2s, inside the Infinity Cache), measured now, on this GPU, by tools/calibrate_fetch (a child process, before this
    process holds device memory): requests per second, or None when the binary is not there or fails."""
    exe = os.path.join(ROOT, "tools", "calibrate_fetch")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "128"], capture_output=True, text=True, timeout=120)
        if out.returncode != 0:
            return None
        rates = [float(f[8]) for f in (line.split() for line in out.stdout.splitlines()) if len(f) >= 9 and f[0] == "gather64" and f[7] == "GBps"]
        return max(rates) * 1e9 / 64.0 if rates else None
    except (subprocess.SubprocessError, OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("RT_BENCH_WORKLOAD", "pt1m_4k_256spp"), choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 counter passes (roofline.traffic is then null)")
    ap.add_argument("--timed-frames-only", action="store_true",
                    help="profiling runs: skip the counted frame and the frame without ray sharing (their launches would mix "
                         "into the per-kernel averages); the line then carries the timing only")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier and the timing reductions (the frame's gather is the library's own RCCL)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import raytracer_amd as R
    from raytracer_amd import scenegen
    from raytracer_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rendering core has no CPU fallback")
    # (modulo: a rehearsal of several ranks on a box with fewer GPUs, --backend gloo, shares devices; a node has one per rank)
    local_rank %= torch.cuda.device_count()
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
    # counter passes first, while this process holds no device memory (the children render the same frame)
    traffic_by_kernel = None
    ceiling = None
    if world == 1 and not args.timed_frames_only and not args.no_pmc:
        traffic_by_kernel = measure_traffic(args.workload)
        ceiling = gather_ceiling()
    sc = R.Scene(path)
    cam = sc.camera(0)
    ren = R.Renderer(sc)   # scene resident in HBM from here on (a full replica on every rank)

    # full-frame device framebuffer on rank 0 (torch is used for device memory, the barrier and the timing reductions only)
    fb = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.float32, device="cuda")
    comm = None
    gather = "none"
    if world > 1:   # the library's communicator: tiles are rendered per rank and gathered on rank 0 over RCCL
        gather = "packed tiles to rank 0, library RCCL send/recv"
        where = "cuda" if args.backend == "nccl" else "cpu"
        # making the communicator is a collective of its own (every rank must enter it): first agree that every rank can
        g = R.gpu_lib()
        g.rt_comm_available.restype = C.c_int32
        able = torch.tensor([int(g.rt_comm_available())], device=where)
        dist.all_reduce(able, op=dist.ReduceOp.MIN)
        if int(able[0]) == 1:
            try:
                comm = R.Comm.init_rank(sharding.exchange_unique_id(rank, world), rank, world)
            except Exception as e:   # said aloud, and in the line: the frames are then summed by torch.distributed instead
                sys.stderr.write(f"bench.py rank {rank}: library communicator unavailable ({e}); gathering with torch.distributed\n")
        elif rank == 0:
            sys.stderr.write("bench.py: RCCL cannot be opened by the library on every rank; gathering with torch.distributed\n")
        ok = torch.tensor([1 if comm is not None else 0], device=where)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok[0]) == 0:
            if comm is not None:
                comm.close()
            comm = None
            gather = "full frames summed on rank 0, torch.distributed reduce (library communicator unavailable)"

    def step():
        if world == 1:
            _, st = ren.render(cam, out=fb.data_ptr())
        elif comm is not None:
            _, st = ren.render_gather(comm, cam, out=fb.data_ptr())
        else:   # every rank's tiles at their frame positions in a zeroed frame; the sum on rank 0 is the image
            fb.zero_()
            torch.cuda.synchronize()
            _, st = ren.render(cam, out=fb.data_ptr(), rank=rank, world=world)
            if args.backend == "nccl":
                dist.reduce(fb, dst=0, op=dist.ReduceOp.SUM)
            else:
                host = fb.cpu()
                dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
                if rank == 0:
                    fb.copy_(host)
        return st

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    ren.set_profiling(True)   # HIP events around every kernel launch, on the library's own stream
    traced = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
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
        if comm is not None:
            comm.close()
        if world > 1:
            dist.destroy_process_group()
        return

    # one extra, untimed frame with the work counters on: algorithmic bytes for the roofline.  In this mode every
    # light sample's shadow ray is traversed on its own, which is also the frame time without ray sharing.
    ren.set_counters(True)
    _, cs = ren.render(cam, out=fb.data_ptr(), rank=rank, world=world)   # (rt_render_shard: the rank's tiles at their frame positions)
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
        per_frame = kernel_bytes(cs, stats)   # fixed per-ray byte model of each kernel class, one frame
        launches_per_frame = launches / args.steps
        model_bytes_per_launch = per_frame[name] / launches_per_frame
        ms_per_launch = total_ms / launches
        model_rate = model_bytes_per_launch / (ms_per_launch * 1e-3) / 1e9
        t = traffic_by_kernel.get(name) if traffic_by_kernel else None
        roof = {"bound": "hbm", "kernel": name, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "kernel_ms_per_launch": ms_per_launch, "launches_per_frame": launches_per_frame,
                "kernel_ms_per_frame": {k: v[1] / args.steps for k, v in ktimes.items()},
                "kernel_ms_note": "event times per kernel class; in path-traced frames the gather of a generation runs on a second stream "
                                  "beside the next generation's shading, so the times of shade and light_gather overlap (their sum exceeds "
                                  "their share of the frame; RT_PAIR_GATHER=0 runs them in turn)",
                # the fixed accounting of the reference's two-child walk (32 B per node visit, 36 B per triangle test, ...):
                # comparable across layouts, but LDS, L1 and L2 serve most of these bytes, so it is no bound of this kernel
                "byte_model": {"bytes_per_launch": model_bytes_per_launch, "rate_GBps": model_rate,
                               "ratio_to_hbm_peak": model_rate / HBM_PEAK_GBS, "ratio_to_l2_peak": model_rate / L2_PEAK_GBS,
                               "note": "normaliser only: bytes of the reference's node-by-node walk, served mostly from LDS/L1/L2"}}
        if t:
            sec = ms_per_launch * 1e-3
            factor = fetch_factor(name)
            true_bytes = factor * t["fetch_size_bytes"] + t["write_size_bytes"]
            slot_bytes = 2.0 * t["fetch_size_bytes"] + t["write_size_bytes"]
            roof.update({
                "traffic": true_bytes, "achieved": true_bytes / sec / 1e9, "frac": true_bytes / sec / 1e9 / HBM_PEAK_GBS,
                "fetch_size_factor": factor,
                "traffic_definition": "bytes moved behind the L2 per launch, measured in this run: FETCH_SIZE x the factor calibrated for the "
                                      "kernel's access class (1 for the trace kernels' 64-byte record gathers, 2 for coalesced streams: "
                                      "tools/calibrate_fetch.sh, profiles/r2_fetch_calibration.json) + WRITE_SIZE",
                "request_slot_bytes": slot_bytes, "request_slot_frac": slot_bytes / sec / 1e9 / HBM_PEAK_GBS,
                "request_slot_definition": "128 B per L2 read request + WRITE_SIZE: a request occupies the memory side like a whole line "
                                           "whatever it moves (60 G requests/s x 128 B = the 8 TB/s peak); how busy the request path is, not bytes",
                "read_requests_per_launch": t["read_requests"], "read_requests_per_s": t["read_requests"] / sec})
            if ceiling and name.startswith("trace"):
                roof["gather_ceiling_requests_per_s"] = ceiling
                roof["gather_ceiling_definition"] = "independent 64-byte record gathers from a 128 MiB table, measured in this run (tools/calibrate_fetch)"
                roof["frac_of_gather_ceiling"] = t["read_requests"] / sec / ceiling
        if traffic_by_kernel:
            roof["traffic_per_frame_by_kernel"] = {k: (fetch_factor(k) * v["fetch_size_bytes"] + v["write_size_bytes"]) * v["launches"]
                                                   for k, v in traffic_by_kernel.items()}
        line = {
            "metric": "Mrays/s (primary+shadow+secondary), whole job", "value": traced / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "resolution": f"{cam.resx}x{cam.resy}", "spp": cam.num_samples,
                       "triangles": sc.desc.n_triangles, "bvh_nodes": sc.desc.n_bvh_nodes,
                       "renderer": "path tracing" if cam.path_tracing else "whitted", "sharding": f"tiles32/{world}", "gather": gather},
            "ray_accounting": "value counts the BVH traversals run in the timed frames (camera and bounce paths + shadow rays traced; light "
                              "samples of a shading point whose shadow rays are identical bit for bit share one).  "
                              "value_reference_equivalent counts the rays the reference's algorithm casts for the same image (oracle "
                              "counters: every light sample its own shadow ray) over the same time; it is the unit of cpu_baseline and "
                              "not a device ray rate.  without_ray_sharing times a frame with sharing off",
            "traversals_per_frame": traced / args.steps, "reference_rays_per_frame": rays / args.steps,
            "value_reference_equivalent": rays / elapsed / 1e6,
            "roofline": roof,
        }
        if unshared_ms:
            line["without_ray_sharing"] = {"ms_per_step": unshared_ms, "value": cs.total_rays / unshared_ms / 1e3}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(R, scenegen, w, scene_dir, None)
            # same image on both sides: the reference's rays over each side's time
            line["vs_cpu_baseline"] = line["value_reference_equivalent"] / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
