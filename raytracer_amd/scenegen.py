"""Synthetic benchmark / test scenes in the reference's JSON + PLY dialect.

The reference ships no scene files; the configurations named by the benchmark are instantiated
here from a procedural mesh (a displaced UV sphere written as a binary little-endian PLY) so that
they can be regenerated anywhere, bit for bit, without external assets.
"""
import json
import os
import struct
import zlib

import numpy as np


def write_displaced_sphere_ply(path, n, outward=False):
    """(n+1)^2 vertices, 2 n^2 triangles: r = 1 + 0.08 sin(7u) sin(5v).  The face winding [a, c, b] / [b, c, d] of the
    benchmark meshes gives face normals that point into the solid (the reference shades with the face normal as it
    is, src/shapes.cpp:152-162, so such a mesh receives ambient light only); `outward` swaps two indices per face."""
    u = np.linspace(0.0, 2.0 * np.pi, n + 1)
    v = np.linspace(0.05, np.pi - 0.05, n + 1)
    vv, uu = np.meshgrid(v, u, indexing="ij")
    r = 1.0 + 0.08 * np.sin(7.0 * uu) * np.sin(5.0 * vv)
    pts = np.stack([r * np.sin(vv) * np.cos(uu), r * np.cos(vv), r * np.sin(vv) * np.sin(uu)], axis=-1)
    pts = pts.reshape(-1, 3).astype("<f4")
    idx = np.arange((n + 1) * (n + 1), dtype=np.int32).reshape(n + 1, n + 1)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    tris = np.concatenate([np.stack([a, c, b], axis=1), np.stack([b, c, d], axis=1)], axis=0).astype("<i4")
    if outward:
        tris = np.ascontiguousarray(tris[:, [0, 2, 1]])
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


def write_png_rgb8(path, rgb):
    """8-bit RGB PNG (one IDAT chunk, filter 0 on every row), top row first."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def _tiles_image(n=128):
    """Coloured tiles with a soft gradient: the wall texture of the textured scene."""
    y, x = np.mgrid[0:n, 0:n]
    cell = ((x * 8 // n) + (y * 8 // n)) % 3
    img = np.zeros((n, n, 3), dtype=np.uint8)
    img[..., 0] = (60 + 80 * cell).astype(np.uint8)
    img[..., 1] = (200 - 60 * cell).astype(np.uint8)
    img[..., 2] = (40 + x * 200 // (n - 1)).astype(np.uint8)
    return img


def _marble_image(n=64):
    """Smooth bands: the texture of the image-mapped sphere."""
    y, x = np.mgrid[0:n, 0:n].astype(np.float64)
    v = 0.5 + 0.5 * np.sin((x + 0.6 * y) * 2 * np.pi * 4 / n)
    img = np.stack([90 + 150 * v, 70 + 120 * v, 200 - 100 * v], axis=-1)
    return img.astype(np.uint8)


def write_textured_scene(out_dir, name="textured", n=188, width=1920, height=1080, spp=64):
    """Config 3: a textured multi-object scene with distribution effects -- checkerboard ground, Perlin-textured mesh
    (2 n^2 triangles), image-textured wall (bilinear) and sphere (nearest), a glossy mirror sphere (roughness 0.1), two
    area lights (16 samples each), thin-lens depth of field, `spp` jittered samples per pixel; Whitted shading."""
    os.makedirs(out_dir, exist_ok=True)
    ply = f"m{n}_outward.ply"
    if not os.path.exists(os.path.join(out_dir, ply)):
        write_displaced_sphere_ply(os.path.join(out_dir, ply), n, outward=True)
    write_png_rgb8(os.path.join(out_dir, "tiles.png"), _tiles_image())
    write_png_rgb8(os.path.join(out_dir, "marble.png"), _marble_image())
    cam = {"_id": "1", "_type": "lookAt", "Position": "0 0.9 5.4", "GazePoint": "0 -0.1 0", "Up": "0 1 0", "FovY": "40",
           "NearDistance": "1", "ImageResolution": f"{width} {height}", "ImageName": f"{name}.png",
           "NumSamples": str(spp), "FocusDistance": "5.2", "ApertureSize": "0.08"}
    lights = {"AmbientLight": "12 12 12",
              "AreaLight": [{"_id": "1", "Position": "1.5 4.5 2.5", "Normal": "0 -1 0", "Size": "1.6", "Radiance": "2600 2500 2300"},
                            {"_id": "2", "Position": "-3.5 3 3", "Normal": "0.6 -0.6 -0.5", "Size": "1.2", "Radiance": "2400 2000 1600"}]}
    mats = [{"_id": "1", "AmbientReflectance": "1 1 1", "DiffuseReflectance": "0.8 0.7 0.6",
             "SpecularReflectance": "0.4 0.4 0.4", "PhongExponent": "30"},
            {"_id": "2", "_type": "mirror", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0.1 0.1 0.1",
             "SpecularReflectance": "0.2 0.2 0.2", "MirrorReflectance": "0.8 0.8 0.8", "Roughness": "0.1", "PhongExponent": "50"},
            {"_id": "3", "AmbientReflectance": "1 1 1", "DiffuseReflectance": "0.7 0.7 0.7",
             "SpecularReflectance": "0.1 0.1 0.1", "PhongExponent": "10"}]
    textures = {"Images": {"Image": [{"_id": "1", "_data": "tiles.png"}, {"_id": "2", "_data": "marble.png"}]},
                "TextureMap": [
                    {"_id": "1", "_type": "checkerboard", "DecalMode": "replace_kd", "BlackColor": "0.08 0.08 0.1",
                     "WhiteColor": "0.9 0.88 0.8", "Scale": "0.8", "Offset": "0.01"},
                    {"_id": "2", "_type": "perlin", "DecalMode": "replace_kd", "NoiseConversion": "absval", "NoiseScale": "4",
                     "NumOctaves": "3"},
                    {"_id": "3", "_type": "image", "ImageId": "1", "DecalMode": "replace_kd", "Interpolation": "bilinear"},
                    {"_id": "4", "_type": "image", "ImageId": "2", "DecalMode": "blend_kd", "Interpolation": "nearest"}]}
    #        ground (1-4)                                   wall (5-8)                                       sphere centres (9, 10)
    verts = "-6 -1.2 -6  6 -1.2 -6  6 -1.2 6  -6 -1.2 6   -5 -1.2 -3.5  5 -1.2 -3.5  5 3.8 -3.5  -5 3.8 -3.5   -2.0 -0.55 0.9  2.0 -0.6 0.7"
    uvs = "0 0  1 0  1 1  0 1   0 1  2 1  2 0  0 0   0 0  0 0"
    objects = {"Mesh": [{"_id": "1", "Material": "1", "Textures": "2", "Faces": {"_plyFile": ply}},
                        {"_id": "2", "Material": "3", "Textures": "1", "Faces": {"_data": "1 4 3  1 3 2"}},
                        {"_id": "3", "Material": "3", "Textures": "3", "Faces": {"_data": "5 6 7  5 7 8"}}],
               "Sphere": [{"_id": "1", "Material": "2", "Center": "9", "Radius": "0.65"},
                          {"_id": "2", "Material": "1", "Textures": "4", "Center": "10", "Radius": "0.6"}]}
    doc = _scene(cam, lights, mats, verts, objects, BackgroundColor="0.05 0.08 0.12")
    doc["Scene"]["Textures"] = textures
    doc["Scene"]["TexCoordData"] = {"_data": uvs}
    path = os.path.join(out_dir, f"{name}.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
    return path
