"""Synthetic benchmark / test scenes in the reference's JSON + PLY dialect.

The reference ships no scene files; the configurations named by the benchmark are instantiated
here from a procedural mesh (a displaced UV sphere written as a binary little-endian PLY) so that
they can be regenerated anywhere, bit for bit, without external assets.
"""
import json
import os

import numpy as np


def write_displaced_sphere_ply(path, n):
    """(n+1)^2 vertices, 2 n^2 triangles: r = 1 + 0.08 sin(7u) sin(5v)."""
    u = np.linspace(0.0, 2.0 * np.pi, n + 1)
    v = np.linspace(0.05, np.pi - 0.05, n + 1)
    vv, uu = np.meshgrid(v, u, indexing="ij")
    r = 1.0 + 0.08 * np.sin(7.0 * uu) * np.sin(5.0 * vv)
    pts = np.stack([r * np.sin(vv) * np.cos(uu), r * np.cos(vv), r * np.sin(vv) * np.sin(uu)], axis=-1)
    pts = pts.reshape(-1, 3).astype("<f4")
    idx = np.arange((n + 1) * (n + 1), dtype=np.int32).reshape(n + 1, n + 1)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    tris = np.concatenate([np.stack([a, c, b], axis=1), np.stack([b, c, d], axis=1)], axis=0).astype("<i4")
    header = ("ply\nformat binary_little_endian 1.0\n"
              f"element vertex {len(pts)}\nproperty float x\nproperty float y\nproperty float z\n"
              f"element face {len(tris)}\nproperty list uchar int vertex_indices\nend_header\n")
    faces = np.empty(len(tris), dtype=[("n", "u1"), ("i", "<i4", 3)])
    faces["n"] = 3
    faces["i"] = tris
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(pts.tobytes())
        f.write(faces.tobytes())
    return len(pts), len(tris)


def _scene(cameras, lights, materials, vertex_data, objects, **scene_keys):
    s = {"BackgroundColor": "0 0 0", "ShadowRayEpsilon": "1e-3", "IntersectionTestEpsilon": "1e-6",
         "MaxRecursionDepth": "6"}
    s.update(scene_keys)
    s.update({"Cameras": {"Camera": cameras}, "Lights": lights, "Materials": {"Material": materials},
              "VertexData": {"_data": vertex_data}, "Objects": objects})
    return {"Scene": s}


def write_mesh_scene(out_dir, name="bunny_like", n=188, width=1920, height=1080):
    """Config 2: ~70K-triangle mesh + ground quad, two point lights, Blinn-Phong + shadows, 1 spp."""
    os.makedirs(out_dir, exist_ok=True)
    ply = f"m{n}.ply"
    if not os.path.exists(os.path.join(out_dir, ply)):
        write_displaced_sphere_ply(os.path.join(out_dir, ply), n)
    cam = {"_id": "1", "_type": "lookAt", "Position": "0 0.5 3.2", "GazePoint": "0 0 0", "Up": "0 1 0", "FovY": "45",
           "NearDistance": "1", "ImageResolution": f"{width} {height}", "ImageName": f"{name}.png"}
    lights = {"AmbientLight": "25 25 25",
              "PointLight": [{"_id": "1", "Position": "3 4 4", "Intensity": "6000 6000 6000"},
                             {"_id": "2", "Position": "-4 2 3", "Intensity": "3000 3000 3000"}]}
    mats = [{"_id": "1", "AmbientReflectance": "1 1 1", "DiffuseReflectance": "0.8 0.7 0.6",
             "SpecularReflectance": "0.5 0.5 0.5", "PhongExponent": "30"}]
    verts = "-4 -1.2 -4  4 -1.2 -4  4 -1.2 4  -4 -1.2 4"
    objects = {"Mesh": [{"_id": "1", "Material": "1", "Faces": {"_plyFile": ply}},
                        {"_id": "2", "Material": "1", "Faces": {"_data": "1 4 3  1 3 2"}}]}
    path = os.path.join(out_dir, f"{name}.json")
    with open(path, "w") as f:
        json.dump(_scene(cam, lights, mats, verts, objects), f, indent=1)
    return path


def write_path_traced_scene(out_dir, name="pt1m", n=710, width=3840, height=2160, spp=256, max_depth=4):
    """Config 4: 1M+ triangle mesh, path tracing (importance sampling + NEE + Russian roulette),
    one spherical emitter, EXR output + photographic tone map."""
    os.makedirs(out_dir, exist_ok=True)
    ply = f"m{n}.ply"
    if not os.path.exists(os.path.join(out_dir, ply)):
        write_displaced_sphere_ply(os.path.join(out_dir, ply), n)
    cam = {"_id": "1", "_type": "lookAt", "Position": "0 0.5 3.2", "GazePoint": "0 0 0", "Up": "0 1 0", "FovY": "45",
           "NearDistance": "1", "ImageResolution": f"{width} {height}", "ImageName": f"{name}.exr",
           "NumSamples": str(spp), "Renderer": "PathTracing",
           "RendererParams": "ImportanceSampling NextEventEstimation RussianRoulette", "MinRecursionDepth": "2",
           "Tonemap": {"TMO": "Photographic", "TMOOptions": "0.18 1", "Saturation": "1.0", "Gamma": "2.2",
                       "Extension": "_phot.png"}}
    lights = {"AmbientLight": "0 0 0"}
    mats = [{"_id": "1", "AmbientReflectance": "1 1 1", "DiffuseReflectance": "0.8 0.7 0.6",
             "SpecularReflectance": "0.5 0.5 0.5", "PhongExponent": "30"},
            {"_id": "2", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0 0 0", "SpecularReflectance": "0 0 0"}]
    verts = "-4 -1.2 -4  4 -1.2 -4  4 -1.2 4  -4 -1.2 4  0 4 2"
    objects = {"Mesh": [{"_id": "1", "Material": "1", "Faces": {"_plyFile": ply}},
                        {"_id": "2", "Material": "1", "Faces": {"_data": "1 4 3  1 3 2"}}],
               "LightSphere": {"_id": "1", "Material": "2", "Center": "5", "Radius": "0.6", "Radiance": "40 40 40"}}
    path = os.path.join(out_dir, f"{name}.json")
    with open(path, "w") as f:
        json.dump(_scene(cam, lights, mats, verts, objects, MaxRecursionDepth=str(max_depth)), f, indent=1)
    return path
