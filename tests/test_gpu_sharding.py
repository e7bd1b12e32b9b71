"""GPU: image-space shards reassemble to the single-rank frame byte for byte, and size-independent properties of
the full-size workloads (the oracle is too slow to check 4K frames pixel by pixel)."""
import numpy as np
import pytest

import raytracer_amd as R
from raytracer_amd import scenegen, sharding
from golden import scenes as GS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,world", [("pathtrace", 2), ("distributed", 3), ("materials", 8)])
def test_shards_reassemble_byte_identical(name, world, scene_dir):
    sc = R.Scene(GS.SCENES[name][0](scene_dir))
    cam = sc.camera(0)
    ren = R.Renderer(sc)
    full, st = ren.render(cam)
    total = np.zeros_like(full)
    rays = 0
    for rank in range(world):
        part = np.zeros_like(full)
        _, sr = ren.render(cam, out=part, rank=rank, world=world)
        mask = sharding.pixel_mask(cam.resx, cam.resy, rank, world)
        assert not part[~mask].any()                      # nothing outside the rank's own tiles
        total += part
        rays += sr.total_rays
    # float atomics into a sample's accumulator may re-associate between runs; a frame is reproducible only to that
    err = np.abs(total - full) / np.maximum(np.abs(full), 1 / 255)
    assert err.max() <= 1e-5
    assert rays == st.total_rays                           # same paths, whoever renders them


def test_exact_scene_shards_are_bitwise(scene_dir):
    """1 spp, point lights: one add per light in a fixed order, so sharded and whole frames are bit-identical."""
    path = scenegen.write_mesh_scene(scene_dir, n=188, width=480, height=270)
    sc = R.Scene(path)
    cam = sc.camera(0)
    ren = R.Renderer(sc)
    full, _ = ren.render(cam)
    total = np.zeros_like(full)
    for rank in range(2):
        part = np.zeros_like(full)
        ren.render(cam, out=part, rank=rank, world=2)
        total += part
    assert total.tobytes() == full.tobytes()
    again, _ = ren.render(cam)
    assert again.tobytes() == full.tobytes()


def gathered_frame(sc, cam, world, frames=1):
    """`frames` frames of `cam` through rt_render_gather with `world` ranks that are threads of this process, all on
    device 0 (each with its own replica of the scene): what the command line does with one GPU per rank."""
    import threading
    comms = R.Comm.init_all([0] * world)
    rens = [R.Renderer(sc) for _ in range(world)]
    out, stats, errors = [None] * frames, [[None] * world for _ in range(frames)], []

    def work(rank):
        try:
            for f in range(frames):
                img, st = rens[rank].render_gather(comms[rank], cam)
                stats[f][rank] = st
                if rank == 0:
                    out[f] = img
        except Exception as e:   # a failed rank must not leave the others waiting unnoticed
            errors.append(e)

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads), "a rank is stuck in the gather"
    for c in comms:
        c.close()
    return out, stats


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_gathered_frame_equals_single_render_bitwise(world, scene_dir):
    """The tile-sharded, gathered frame against rt_render, byte for byte: 1 spp with point lights gives every pixel one
    add per light in a fixed order, so nothing depends on who renders a tile.  Ragged frame (15 x 9 tiles, the last
    column and row cut), two frames in a row through the same communicator."""
    sc = R.Scene(scenegen.write_mesh_scene(scene_dir, n=188, width=470, height=265))
    cam = sc.camera(0)
    full, st = R.Renderer(sc).render(cam)
    frames, stats = gathered_frame(sc, cam, world, frames=2)
    for img, per_rank in zip(frames, stats):
        assert img.tobytes() == full.tobytes()
        assert sum(s.total_rays for s in per_rank) == st.total_rays
        counts, _ = sharding.layout(cam.resx, cam.resy, world)
        assert [s.primary_rays > 0 for s in per_rank] == [c > 0 for c in counts]


@pytest.mark.parametrize("name,world", [("pathtrace", 2), ("textured", 3), ("materials", 8)])
def test_gathered_frame_equals_single_render(name, world, scene_dir):
    """Branching and stochastic scenes: the same paths whoever renders them; sums inside a sample may re-associate
    between runs (float atomics), which bounds the difference at 1e-5."""
    sc = R.Scene(GS.SCENES[name][0](scene_dir))
    cam = sc.camera(0)
    full, st = R.Renderer(sc).render(cam)
    (img,), (per_rank,) = gathered_frame(sc, cam, world)
    err = np.abs(img - full) / np.maximum(np.abs(full), 1 / 255)
    assert err.max() <= 1e-5
    assert sum(s.total_rays for s in per_rank) == st.total_rays


def test_single_rank_communicator_over_rccl(scene_dir):
    """One process per GPU, as under torchrun: the id comes from RCCL (ncclGetUniqueId), the communicator from
    ncclCommInitRank -- here with a single rank, which is all one GPU allows -- and the frame through rt_render_gather."""
    sc = R.Scene(scenegen.write_mesh_scene(scene_dir, n=188, width=470, height=265))
    cam = sc.camera(0)
    ren = R.Renderer(sc)
    full, st = ren.render(cam)
    comm = R.Comm.init_rank(sharding.exchange_unique_id(0, 1), 0, 1)
    img, sg = ren.render_gather(comm, cam)
    assert img.tobytes() == full.tobytes() and sg.total_rays == st.total_rays
    comm.close()
    plain = R.Comm.init_rank(None, 0, 1)               # and without an id: no RCCL at all
    img2, _ = ren.render_gather(plain, cam)
    assert img2.tobytes() == full.tobytes()
    plain.close()


def run_rccl_ranks(scene, world, failing, repo_root):
    import json
    import os
    import subprocess
    import sys
    lib = os.path.join(repo_root, "tests", "tools", "libfake_rccl.so")
    assert os.path.exists(lib), "tests/tools/libfake_rccl.so missing: run __graft_entry__.build()"
    r = subprocess.run([sys.executable, os.path.join(repo_root, "tests", "tools", "rccl_ranks.py"), scene, str(world), str(failing)],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, RT_RCCL_LIB=lib))
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("world", [2, 3, 8])
def test_rccl_branch_with_several_ranks(world, scene_dir, repo_root):
    """The exchange as it runs with one process per GPU -- communicators from rt_comm_init_rank, the status all-reduce,
    one ncclSend per rank and the root's grouped ncclRecv at their offsets -- with 2, 3 and 8 ranks on this box's one GPU:
    a stand-in for librccl.so (tests/tools/fake_rccl.cpp, named through RT_RCCL_LIB, in a child process so that this
    process keeps the real library) moves the data between threads.  Three frames through the same communicators, each
    byte for byte rt_render's."""
    scene = scenegen.write_mesh_scene(scene_dir, n=188, width=470, height=265)
    res = run_rccl_ranks(scene, world, -1, repo_root)
    assert not res["stuck"] and res["frame_equal"] == [True] * 3 and res["rays_equal"] == [True] * 3
    assert all(e is None for per in res["errors"] for e in per)


@pytest.mark.parametrize("world,failing", [(3, 2), (3, 0), (8, 5)])
def test_failed_rank_fails_the_frame_and_strands_nobody(world, failing, scene_dir, repo_root):
    """One rank's render fails in the second of three frames (a camera the renderer refuses): that rank returns its own
    error, every other rank returns "a peer rank failed" instead of waiting in the exchange, and the third frame goes
    through the same communicators as if nothing had happened."""
    scene = scenegen.write_mesh_scene(scene_dir, n=188, width=470, height=265)
    res = run_rccl_ranks(scene, world, failing, repo_root)
    assert not res["stuck"]
    assert res["frame_equal"] == [True, False, True] and res["rays_equal"][0] and res["rays_equal"][2]
    bad = res["errors"][1]
    assert all(e is not None for e in bad), bad
    assert "sample count" in bad[failing]
    assert all("peer rank failed" in e for r, e in enumerate(bad) if r != failing), bad
    assert all(e is None for e in res["errors"][0] + res["errors"][2])


@pytest.mark.parametrize("failing", [0, 1])
def test_failed_rank_on_the_shared_device_transport(failing, scene_dir):
    """The same with ranks that share a device (rt_comm_init_all, device-to-device copies): every rank reaches the meeting
    point whatever went wrong on it."""
    import threading
    sc = R.Scene(scenegen.write_mesh_scene(scene_dir, n=188, width=470, height=265))
    cam = sc.camera(0)
    full, _ = R.Renderer(sc).render(cam)
    world = 3
    comms = R.Comm.init_all([0] * world)
    rens = [R.Renderer(sc) for _ in range(world)]
    errors = [[None] * world for _ in range(3)]
    out = [None] * 3

    def work(rank):
        for f in range(3):
            c = sc.camera(0, num_samples=0) if (f == 1 and rank == failing) else cam
            try:
                img, _ = rens[rank].render_gather(comms[rank], c)
                if rank == 0:
                    out[f] = img
            except R.RaytracerError as e:
                errors[f][rank] = str(e)

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a rank is stuck at the meeting point"
    assert all(e is None for e in errors[0] + errors[2]) and out[0].tobytes() == full.tobytes() and out[2].tobytes() == full.tobytes()
    assert "sample count" in errors[1][failing]
    assert all("peer rank failed" in e for r, e in enumerate(errors[1]) if r != failing), errors[1]
    for c in comms:
        c.close()


def test_command_line_team_of_ranks(tmp_path, scene_dir, repo_root):
    """./raytracer with a team of three ranks (all on GPU 0: RT_GPUS_DEVICES) splits each camera's frame by tiles and
    gathers it; the PNG equals the one a single GPU writes."""
    import os
    import subprocess
    from test_host import png_pixels
    exe = os.path.join(repo_root, "raytracer")
    path = scenegen.write_mesh_scene(scene_dir, name="team", n=188, width=470, height=265)
    for sub, env in (("one", {}), ("team", {"RT_GPUS_DEVICES": "0,0,0"})):
        cwd = tmp_path / sub
        cwd.mkdir()
        r = subprocess.run([exe, path], cwd=cwd, capture_output=True, text=True, timeout=120, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout + r.stderr
        assert "team.png rendering done:" in r.stdout
    assert np.array_equal(png_pixels(str(tmp_path / "one" / "outputs" / "team.png")), png_pixels(str(tmp_path / "team" / "outputs" / "team.png")))
    assert png_pixels(str(tmp_path / "one" / "outputs" / "team.png")).std() > 5


def test_full_size_path_traced_frame_properties(scene_dir):
    """1M-triangle scene at 3840x2160 (4 spp to keep the test short; per-sample work is what 256 spp repeats)."""
    path = scenegen.write_path_traced_scene(scene_dir, name="pt1m_props", n=710, width=3840, height=2160, spp=4, max_depth=4)
    sc = R.Scene(path)
    assert sc.desc.n_triangles >= 1_000_000
    cam = sc.camera(0)
    ren = R.Renderer(sc)
    img, st = ren.render(cam)
    assert st.primary_rays == 3840 * 2160 * 4
    assert st.secondary_rays <= st.primary_rays * 4 and st.shadow_rays > st.primary_rays
    assert np.isfinite(img).all() and (img >= 0).all() and img.mean() > 0
    # primary visibility does not depend on how the frame is batched or which seed drives the bounces
    s0, i0, t0, _ = ren.trace_primary(sc.camera(0, num_samples=1))
    s1, i1, t1, _ = ren.trace_primary(sc.camera(0, num_samples=1, seed=7))
    assert np.array_equal(s0, s1) and np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
    assert 0.2 < (t0 < 1e29).mean() <= 1.0
    # a second render with the same seed: same rays, image equal up to atomic re-association
    img2, st2 = ren.render(cam)
    assert st2.total_rays == st.total_rays
    assert np.abs(img2 - img).max() <= 1e-5 * max(1.0, float(img.max()))
    # a quarter-resolution render of the same camera sees the same scene: mean radiance agrees to Monte-Carlo noise
    small, _ = ren.render(sc.camera(0, resx=960, resy=540))
    assert abs(small.mean() - img.mean()) <= 0.05 * img.mean()


@pytest.mark.parametrize("name", ["emitters", "pathtrace", "distributed"])
def test_shared_shadow_rays_change_nothing_but_the_work(name, scene_dir):
    """Light samples of a shading point whose shadow rays are identical share one traversal (sphere emitters:
    the reference always aims at the sphere's centre), and a sphere emitter's samples are evaluated after that
    traversal, only where it found the light.  Against a render with the work counters on, where every sample is
    evaluated and its ray traversed as in the reference: the same image, less work."""
    sc = R.Scene(GS.SCENES[name][0](scene_dir))
    cam = sc.camera(0)
    ren = R.Renderer(sc)
    shared, st_shared = ren.render(cam)
    ren.set_counters(True)
    each, st_each = ren.render(cam)
    assert (st_shared.primary_rays, st_shared.secondary_rays) == (st_each.primary_rays, st_each.secondary_rays)
    assert st_each.shadow_rays_traced == st_each.shadow_rays
    assert st_shared.shadow_rays <= st_each.shadow_rays            # samples of a light the point cannot see are not evaluated
    assert st_shared.shadow_rays_traced <= st_each.shadow_rays
    if name == "emitters":
        assert st_shared.shadow_rays_traced < st_each.shadow_rays and st_shared.shadow_rays < st_each.shadow_rays
    err = np.abs(shared - each) / np.maximum(np.abs(each), 1 / 255)
    assert err.max() <= 1e-5


def many_lights_scene(out_dir, n_emitters=5):
    """More than 64 light samples per shading point (two passes of the lighting kernels), kinds mixed: a point light,
    an area light, a mesh emitter and sphere emitters with 16 samples each, so that one emitter's samples straddle the
    pass boundary whatever order the lights are numbered in."""
    mats = [GS.MAT_DIFFUSE, {"_id": "2", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0 0 0", "SpecularReflectance": "0 0 0"},
            {"_id": "3", "AmbientReflectance": "1 1 1", "DiffuseReflectance": "0.3 0.6 0.8", "SpecularReflectance": "0.4 0.4 0.4", "PhongExponent": "25"}]
    centres = [(-2.0 + 1.0 * k, 2.2 + 0.3 * (k % 2), 0.5 + 0.4 * k) for k in range(n_emitters)]
    verts = GS.GROUND + "  -1 0.6 0.3  1.1 0.5 -0.2  -2.5 2.5 -1  -1.5 2.5 -1  -1.5 2.5 0  -2.5 2.5 0  " + GS.verts_str(centres)
    objects = {"Mesh": {"_id": "1", "Material": "1", "Faces": {"_data": GS.GROUND_FACES}},
               "LightMesh": {"_id": "2", "Material": "2", "Radiance": "30 24 18", "Faces": {"_data": "7 9 8  7 10 9"}},
               "Sphere": [{"_id": "1", "Material": "3", "Center": "5", "Radius": "0.6"},
                          {"_id": "2", "Material": "1", "Center": "6", "Radius": "0.5"}],
               "LightSphere": [{"_id": str(k + 1), "Material": "2", "Center": str(11 + k), "Radius": "0.25",
                                "Radiance": f"{20 + 5 * k} {22 + 3 * k} {30 - 2 * k}"} for k in range(n_emitters)]}
    lights = {"AmbientLight": "5 5 5", "PointLight": {"_id": "1", "Position": "3 5 4", "Intensity": "2000 2000 2000"},
              "AreaLight": {"_id": "1", "Position": "1 4 2", "Normal": "0 -1 0", "Size": "1.5", "Radiance": "15 15 15"}}
    sc = GS.scene(GS.camera("many_lights.png", res=(64, 48), NumSamples="4"), lights, mats, verts, objects)
    return GS.dump(out_dir, "many_lights", sc)


def test_many_lights_two_passes_all_paths_agree(scene_dir):
    """The deferred-emitter path, the counted path (every sample on its own) and the CPU oracle give the same image for
    a scene whose light samples need two passes and mix deferred and recorded lights in one shading point."""
    from oracle_lib import Oracle
    sc = R.Scene(many_lights_scene(scene_dir))
    assert sum(sc.desc.lights[i].samples for i in range(sc.desc.n_lights)) > 128     # three passes
    cam = sc.camera(0)
    ren = R.Renderer(sc)
    fast, st_fast = ren.render(cam)
    ren.set_counters(True)
    counted, st_counted = ren.render(cam)
    want, st_oracle = Oracle(sc).render(cam)
    assert st_counted.shadow_rays == st_oracle.shadow_rays and st_counted.shadow_rays > 40 * st_counted.primary_rays
    assert st_fast.shadow_rays_traced < st_counted.shadow_rays_traced
    floor = 1 / 255
    assert (np.abs(counted - want) / np.maximum(np.abs(want), floor)).max() <= 1e-5
    assert (np.abs(fast - counted) / np.maximum(np.abs(counted), floor)).max() <= 1e-5
