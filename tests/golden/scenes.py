"""Small feature scenes (reference JSON dialect) used to pin the CPU oracle against the compiled
reference and the GPU core against the oracle.  Everything is generated procedurally — the reference
ships no scenes — and deterministically, so the files can be recreated wherever the tests run.

Each entry of SCENES: name -> (builder(out_dir) -> json path, kind) with kind
  "exact"       deterministic image (1 spp, no random-number use that changes the image)
  "stochastic"  image depends on random numbers; compared statistically against the reference
"""
import json
import os
import struct
import zlib

import numpy as np


# ---- asset writers ----------------------------------------------------------------------------------
def write_png(path, rgb):
    """8-bit RGB PNG, top row first."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_exr(path, rgb):
    """Uncompressed 32-bit float RGB scan-line EXR, top row first."""
    rgb = np.ascontiguousarray(rgb, dtype="<f4")
    h, w, _ = rgb.shape
    out = bytearray(b"\x76\x2f\x31\x01\x02\x00\x00\x00")

    def attr(name, typ, data):
        out.extend(name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data)

    ch = b""
    for c in "BGR":
        ch += c.encode() + b"\0" + struct.pack("<iBBBBii", 2, 0, 0, 0, 0, 1, 1)
    attr("channels", "chlist", ch + b"\0")
    attr("compression", "compression", b"\0")
    attr("dataWindow", "box2i", struct.pack("<iiii", 0, 0, w - 1, h - 1))
    attr("displayWindow", "box2i", struct.pack("<iiii", 0, 0, w - 1, h - 1))
    attr("lineOrder", "lineOrder", b"\0")
    attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0))
    attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    out.append(0)
    table = len(out)
    line = 8 + w * 12
    out.extend(b"\0" * (h * 8))
    for y in range(h):
        struct.pack_into("<Q", out, table + y * 8, table + h * 8 + y * line)
        out.extend(struct.pack("<ii", y, w * 12))
        for c in (2, 1, 0):
            out.extend(rgb[y, :, c].tobytes())
    with open(path, "wb") as f:
        f.write(bytes(out))


def write_ascii_ply(path, verts, faces, normals=None, uvs=None):
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\n")
        f.write(f"element vertex {len(verts)}\nproperty float x\nproperty float y\nproperty float z\n")
        if normals is not None:
            f.write("property float nx\nproperty float ny\nproperty float nz\n")
        if uvs is not None:
            f.write("property float u\nproperty float v\n")
        f.write(f"element face {len(faces)}\nproperty list uchar int vertex_indices\nend_header\n")
        for i, v in enumerate(verts):
            row = list(v)
            if normals is not None:
                row += list(normals[i])
            if uvs is not None:
                row += list(uvs[i])
            f.write(" ".join(repr(float(x)) for x in row) + "\n")
        for face in faces:
            f.write(f"{len(face)} " + " ".join(str(int(i)) for i in face) + "\n")


def checker_image(n=32, cells=4):
    y, x = np.mgrid[0:n, 0:n]
    c = ((x * cells // n + y * cells // n) % 2).astype(np.uint8)
    img = np.zeros((n, n, 3), dtype=np.uint8)
    img[..., 0] = 40 + 200 * c
    img[..., 1] = 220 - 150 * c
    img[..., 2] = (x * 255 // (n - 1)).astype(np.uint8)
    return img


def bumps_image(n=64):
    y, x = np.mgrid[0:n, 0:n].astype(np.float64)
    v = 0.5 + 0.5 * np.sin(x * 2 * np.pi * 3 / n) * np.cos(y * 2 * np.pi * 2 / n)
    g = (v * 255).astype(np.uint8)
    return np.stack([g, g, g], axis=-1)


def normalmap_image(n=64):
    y, x = np.mgrid[0:n, 0:n].astype(np.float64)
    nx = 0.4 * np.sin(x * 2 * np.pi * 2 / n)
    ny = 0.4 * np.cos(y * 2 * np.pi * 2 / n)
    nz = np.sqrt(np.maximum(0.0, 1 - nx * nx - ny * ny))
    img = np.stack([(nx * 0.5 + 0.5) * 255, (ny * 0.5 + 0.5) * 255, (nz * 0.5 + 0.5) * 255], axis=-1)
    return img.astype(np.uint8)


def sky_image(w=64, h=32):
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, 3), dtype=np.float32)
    img[..., 0] = 0.2 + 0.7 * (x / w) ** 2
    img[..., 1] = 0.4 + 0.4 * (1 - y / h)
    img[..., 2] = 0.9 - 0.5 * (y / h)
    sun = np.exp(-(((x - w * 0.7) / 3) ** 2 + ((y - h * 0.25) / 2) ** 2))
    img += (6.0 * sun)[..., None].astype(np.float32)
    return img


# ---- scene assembly -----------------------------------------------------------------------------------
def camera(name, res=(96, 96), pos="0 1.2 4.5", gaze_point="0 0.6 0", fovy="40", **extra):
    c = {"_id": "1", "_type": "lookAt", "Position": pos, "GazePoint": gaze_point, "Up": "0 1 0", "FovY": fovy,
         "NearDistance": "1", "ImageResolution": f"{res[0]} {res[1]}", "ImageName": name}
    c.update(extra)
    return c


def scene(cam, lights, materials, vertex_data, objects, **keys):
    s = {"BackgroundColor": "0.05 0.08 0.12", "ShadowRayEpsilon": "1e-3", "IntersectionTestEpsilon": "1e-6",
         "MaxRecursionDepth": "6"}
    s.update(keys)
    s["Cameras"] = {"Camera": cam}
    s["Lights"] = lights
    s["Materials"] = {"Material": materials}
    s["VertexData"] = {"_data": vertex_data}
    s["Objects"] = objects
    return {"Scene": s}


def dump(out_dir, name, sc):
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, name + ".json")
    with open(path, "w") as f:
        json.dump(sc, f, indent=1)
    return path


GROUND = "-5 0 -5  5 0 -5  5 0 5  -5 0 5"   # vertices 1..4
GROUND_FACES = "1 4 3  1 3 2"
MAT_DIFFUSE = {"_id": "1", "AmbientReflectance": "1 1 1", "DiffuseReflectance": "0.7 0.7 0.7", "SpecularReflectance": "0.3 0.3 0.3",
               "PhongExponent": "20"}
TWO_LIGHTS = {"AmbientLight": "20 20 20", "PointLight": [{"_id": "1", "Position": "3 5 4", "Intensity": "9000 9000 9000"},
                                                          {"_id": "2", "Position": "-4 3 2", "Intensity": "3000 2500 2000"}]}

# cube 0..1 (8 vertices) as 12 triangles, local indices
CUBE_V = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
CUBE_F = [(0, 2, 1), (0, 3, 2), (4, 5, 6), (4, 6, 7), (0, 1, 5), (0, 5, 4), (3, 7, 6), (3, 6, 2), (0, 4, 7), (0, 7, 3), (1, 2, 6), (1, 6, 5)]


def verts_str(vs):
    return "  ".join(f"{x} {y} {z}" for x, y, z in vs)


def faces_str(fs, base):
    return "  ".join(" ".join(str(i + base + 1) for i in f) for f in fs)


def materials_scene(out_dir):
    """Mirror, conductor, dielectric (with absorption) and diffuse spheres over a ground quad and a plane."""
    mats = [MAT_DIFFUSE,
            {"_id": "2", "_type": "mirror", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0.1 0.1 0.1",
             "SpecularReflectance": "0.2 0.2 0.2", "MirrorReflectance": "0.9 0.9 0.9", "PhongExponent": "50"},
            {"_id": "3", "_type": "conductor", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0.1 0.08 0.02",
             "SpecularReflectance": "0.3 0.3 0.1", "MirrorReflectance": "0.9 0.8 0.4", "RefractionIndex": "0.35",
             "AbsorptionIndex": "2.45", "PhongExponent": "30"},
            {"_id": "4", "_type": "dielectric", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0 0 0",
             "SpecularReflectance": "0.1 0.1 0.1", "MirrorReflectance": "1 1 1", "AbsorptionCoefficient": "0.3 0.05 0.2",
             "RefractionIndex": "1.5", "PhongExponent": "100"},
            {"_id": "5", "AmbientReflectance": "1 1 1", "DiffuseReflectance": "0.2 0.5 0.8", "SpecularReflectance": "0 0 0"}]
    verts = GROUND + "  -1.6 0.6 0  0 0.6 0.6  1.6 0.6 0  0 1.9 -1.2  0 0 -4"
    objects = {"Mesh": {"_id": "1", "Material": "1", "Faces": {"_data": GROUND_FACES}},
               "Sphere": [{"_id": "1", "Material": "2", "Center": "5", "Radius": "0.6"},
                          {"_id": "2", "Material": "4", "Center": "6", "Radius": "0.6"},
                          {"_id": "3", "Material": "3", "Center": "7", "Radius": "0.6"},
                          {"_id": "4", "Material": "5", "Center": "8", "Radius": "0.5"}],
               "Plane": {"_id": "1", "Material": "5", "Point": "9", "Normal": "0 0 1"}}
    return dump(out_dir, "materials", scene(camera("materials.png"), TWO_LIGHTS, mats, verts, objects))


def textures_scene(out_dir):
    """Image (nearest + bilinear), checker, Perlin, normal-map, bump-map (image + procedural) and replace_all textures."""
    os.makedirs(out_dir, exist_ok=True)
    write_png(os.path.join(out_dir, "checker.png"), checker_image())
    write_png(os.path.join(out_dir, "bumps.png"), bumps_image())
    write_png(os.path.join(out_dir, "normals.png"), normalmap_image())
    tex = {"Images": {"Image": [{"_id": "1", "_data": "checker.png"}, {"_id": "2", "_data": "bumps.png"},
                                {"_id": "3", "_data": "normals.png"}]},
           "TextureMap": [
               {"_id": "1", "_type": "image", "ImageId": "1", "DecalMode": "replace_kd", "Interpolation": "bilinear"},
               {"_id": "2", "_type": "image", "ImageId": "1", "DecalMode": "blend_kd", "Interpolation": "nearest"},
               {"_id": "3", "_type": "checkerboard", "DecalMode": "replace_kd", "BlackColor": "0.1 0.1 0.1", "WhiteColor": "0.9 0.9 0.8",
                "Scale": "0.4", "Offset": "0.01"},
               {"_id": "4", "_type": "perlin", "DecalMode": "replace_kd", "NoiseConversion": "absval", "NoiseScale": "3", "NumOctaves": "3"},
               {"_id": "5", "_type": "image", "ImageId": "3", "DecalMode": "replace_normal", "Interpolation": "bilinear"},
               {"_id": "6", "_type": "image", "ImageId": "2", "DecalMode": "bump_normal", "Interpolation": "bilinear", "BumpFactor": "0.3"},
               {"_id": "7", "_type": "perlin", "DecalMode": "bump_normal", "NoiseConversion": "linear", "NoiseScale": "4", "NumOctaves": "2",
                "BumpFactor": "0.02"},
               {"_id": "8", "_type": "image", "ImageId": "1", "DecalMode": "replace_all", "Interpolation": "nearest"},
               {"_id": "9", "_type": "image", "ImageId": "2", "DecalMode": "replace_ks", "Interpolation": "bilinear"}]}
    mats = [MAT_DIFFUSE]
    verts = GROUND + "  -2 0.7 0  -0.7 0.7 0.5  0.7 0.7 0.5  2 0.7 0  -1.2 2.0 -1  0 2.1 -1  1.2 2.0 -1" \
                     "  -3 0 -2.5  3 0 -2.5  3 3.5 -2.5  -3 3.5 -2.5"
    uvs = "0 0  4 0  4 4  0 4  0 0 0 0 0 0 0 0 0 0 0 0 0 0  0 1  1 1  1 0  0 0"
    objects = {"Mesh": [{"_id": "1", "Material": "1", "Textures": "3", "Faces": {"_data": GROUND_FACES}},
                        {"_id": "2", "Material": "1", "Textures": "1 9", "Faces": {"_data": "12 13 14  12 14 15"}}],
               "Sphere": [{"_id": "1", "Material": "1", "Textures": "1 5", "Center": "5", "Radius": "0.6"},
                          {"_id": "2", "Material": "1", "Textures": "4 6", "Center": "6", "Radius": "0.6"},
                          {"_id": "3", "Material": "1", "Textures": "2 7", "Center": "7", "Radius": "0.6"},
                          {"_id": "4", "Material": "1", "Textures": "8", "Center": "8", "Radius": "0.6"},
                          {"_id": "5", "Material": "1", "Textures": "7", "Center": "9", "Radius": "0.45"},
                          {"_id": "6", "Material": "1", "Textures": "4", "Center": "10", "Radius": "0.45"},
                          {"_id": "7", "Material": "1", "Textures": "6", "Center": "11", "Radius": "0.45"}]}
    sc = scene(camera("textures.png", res=(128, 96)), TWO_LIGHTS, mats, verts, objects)
    sc["Scene"]["Textures"] = tex
    sc["Scene"]["TexCoordData"] = {"_data": uvs}
    return dump(out_dir, "textures", sc)


def brdf_scene(out_dir):
    """All five BRDF types, normalized / kdfresnel variants, degamma."""
    brdfs = {"OriginalBlinnPhong": {"_id": "1", "Exponent": "40"}, "OriginalPhong": {"_id": "2", "Exponent": "40"},
             "ModifiedBlinnPhong": [{"_id": "3", "Exponent": "40"}, {"_id": "4", "_normalized": "true", "Exponent": "40"}],
             "ModifiedPhong": [{"_id": "5", "Exponent": "40"}, {"_id": "6", "_normalized": "true", "Exponent": "40"}],
             "TorranceSparrow": [{"_id": "7", "Exponent": "40"}, {"_id": "8", "_kdfresnel": "true", "Exponent": "40"}]}
    mats = [MAT_DIFFUSE]
    for i in range(8):
        mats.append({"_id": str(i + 2), "_BRDF": str(i + 1), "_degamma": "true" if i % 3 == 0 else "false",
                     "AmbientReflectance": "1 1 1", "DiffuseReflectance": "0.6 0.3 0.2", "SpecularReflectance": "0.6 0.6 0.6"})
    centers = [(-2.25 + 1.5 * (i % 4), 0.6 + 1.3 * (i // 4), 0.0) for i in range(8)]
    verts = GROUND + "  " + verts_str(centers)
    objects = {"Mesh": {"_id": "1", "Material": "1", "Faces": {"_data": GROUND_FACES}},
               "Sphere": [{"_id": str(i + 1), "Material": str(i + 2), "Center": str(5 + i), "Radius": "0.55"} for i in range(8)]}
    lights = {"AmbientLight": "10 10 10", "PointLight": {"_id": "1", "Position": "1 4 5", "Intensity": "8000 8000 8000"},
              "DirectionalLight": {"_id": "1", "Direction": "-1 -1 -0.5", "Radiance": "1.5 1.4 1.2"},
              "SpotLight": {"_id": "1", "Position": "0 6 2", "Direction": "0 -1 -0.3", "Intensity": "20000 20000 24000",
                            "CoverageAngle": "50", "FalloffAngle": "30"}}
    sc = scene(camera("brdfs.png", res=(128, 96), pos="0 1.6 6"), lights, mats, verts, objects)
    sc["Scene"]["BRDFs"] = brdfs
    return dump(out_dir, "brdfs", sc)


def transforms_scene(out_dir):
    """Translated / rotated / scaled / composite meshes, mesh instances (one with reset), a transformed
    sphere (ellipsoid), smooth shading, an ASCII PLY with normals+uvs, a transformed camera and light."""
    os.makedirs(out_dir, exist_ok=True)
    # small smooth-shaded octahedron as ASCII PLY with explicit normals and uvs
    ov = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    of = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    write_ascii_ply(os.path.join(out_dir, "octa.ply"), [tuple(0.5 * c for c in v) for v in ov], of, normals=ov,
                    uvs=[(0.5 + 0.5 * v[0], 0.5 + 0.5 * v[1]) for v in ov])
    write_ascii_ply(os.path.join(out_dir, "quadfan.ply"), [(0, 0, 0), (1, 0, 0), (1.3, 0.8, 0), (0.5, 1.4, 0), (-0.3, 0.8, 0)], [(0, 1, 2, 3, 4)])
    tr = {"Translation": [{"_id": "1", "_data": "-1.8 0 0"}, {"_id": "2", "_data": "1.5 0.4 -0.5"}, {"_id": "3", "_data": "0 1.5 -1.5"}],
          "Rotation": [{"_id": "1", "_data": "35 0 1 0"}, {"_id": "2", "_data": "-25 1 0 1"}],
          "Scaling": [{"_id": "1", "_data": "0.8 1.6 0.8"}, {"_id": "2", "_data": "1.4 0.6 1.0"}],
          "Composite": {"_id": "1", "_data": "0.9 0 0.2 0.3  0 1.1 0 0.2  -0.2 0 0.9 1.2  0 0 0 1"}}
    mats = [MAT_DIFFUSE, {"_id": "2", "AmbientReflectance": "1 1 1", "DiffuseReflectance": "0.8 0.3 0.3", "SpecularReflectance": "0.5 0.5 0.5",
                          "PhongExponent": "60"},
            {"_id": "3", "AmbientReflectance": "1 1 1", "DiffuseReflectance": "0.3 0.7 0.4", "SpecularReflectance": "0.2 0.2 0.2", "PhongExponent": "10"}]
    base = 4
    verts = GROUND + "  " + verts_str(CUBE_V) + "  0 0.8 0.5"
    objects = {
        "Mesh": [{"_id": "1", "Material": "1", "Faces": {"_data": GROUND_FACES}},
                 {"_id": "2", "Material": "2", "Transformations": "s1 r1 t1", "Faces": {"_data": faces_str(CUBE_F, base)}},
                 {"_id": "3", "Material": "3", "_shadingMode": "smooth", "Transformations": "t3", "Faces": {"_plyFile": "octa.ply"}},
                 {"_id": "4", "Material": "2", "_shadingMode": "smooth", "Transformations": "c1", "Faces": {"_data": faces_str(CUBE_F, base)}},
                 {"_id": "7", "Material": "3", "Transformations": "t2 t3", "Faces": {"_plyFile": "quadfan.ply"}}],
        "MeshInstance": [{"_id": "5", "_baseMeshId": "2", "Material": "3", "Transformations": "t2 r2"},
                         {"_id": "6", "_baseMeshId": "3", "_resetTransform": "true", "Transformations": "s2 t2"}],
        "Triangle": {"_id": "1", "Material": "2", "Transformations": "r1", "Indices": "5 7 8"},
        "Sphere": {"_id": "1", "Material": "3", "Center": "13", "Radius": "0.5", "Transformations": "s2 r2"}}
    lights = {"AmbientLight": "20 20 20", "PointLight": {"_id": "1", "Position": "3 5 4", "Intensity": "12000 12000 12000", "Transformations": "t1"}}
    cam = camera("transforms.png", res=(128, 96), pos="0 2 6", Transformations="r1")
    sc = scene(cam, lights, mats, verts, objects)
    sc["Scene"]["Transformations"] = tr
    return dump(out_dir, "transforms", sc)


def envmap_scene(out_dir):
    """EXR output with a lat-long environment map as background and (sampled) light; mirror sphere."""
    os.makedirs(out_dir, exist_ok=True)
    write_exr(os.path.join(out_dir, "sky.exr"), sky_image())
    mats = [MAT_DIFFUSE, {"_id": "2", "_type": "mirror", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0 0 0",
                          "SpecularReflectance": "0 0 0", "MirrorReflectance": "0.95 0.95 0.95"}]
    verts = GROUND + "  0 0.7 0"
    objects = {"Mesh": {"_id": "1", "Material": "1", "Faces": {"_data": GROUND_FACES}},
               "Sphere": {"_id": "1", "Material": "2", "Center": "5", "Radius": "0.7"}}
    lights = {"AmbientLight": "0 0 0", "SphericalDirectionalLight": {"_id": "1", "_type": "latlong", "ImageId": "1", "Sampler": "cosine"}}
    sc = scene(camera("envmap.exr", res=(96, 64), NumSamples="16"), lights, mats, verts, objects)
    sc["Scene"]["Textures"] = {"Images": {"Image": {"_id": "1", "_data": "sky.exr"}}}
    return dump(out_dir, "envmap", sc)


def distributed_scene(out_dir):
    """Distribution effects: area lights, glossy mirror, depth of field, motion blur, 16 spp."""
    mats = [MAT_DIFFUSE, {"_id": "2", "_type": "mirror", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0.1 0.1 0.1",
                          "SpecularReflectance": "0.2 0.2 0.2", "MirrorReflectance": "0.8 0.8 0.8", "Roughness": "0.15"},
            {"_id": "3", "AmbientReflectance": "1 1 1", "DiffuseReflectance": "0.8 0.2 0.2", "SpecularReflectance": "0.3 0.3 0.3", "PhongExponent": "30"}]
    verts = GROUND + "  -1 0.6 0  1 0.6 -1  " + verts_str([(x + 0.2, y, z + 1.2) for x, y, z in [(0, 0, 0), (0.8, 0, 0), (0.4, 0.9, 0)]])
    objects = {"Mesh": [{"_id": "1", "Material": "1", "Faces": {"_data": GROUND_FACES}},
                        {"_id": "2", "Material": "3", "MotionBlur": "0.5 0 0", "Faces": {"_data": "7 8 9"}}],
               "Sphere": [{"_id": "1", "Material": "2", "Center": "5", "Radius": "0.6"},
                          {"_id": "2", "Material": "3", "Center": "6", "Radius": "0.6"}]}
    lights = {"AmbientLight": "10 10 10",
              "AreaLight": [{"_id": "1", "Position": "1 4 2", "Normal": "0 -1 0", "Size": "1.5", "Radiance": "25 25 25"},
                            {"_id": "2", "Position": "-3 3 3", "Normal": "0.6 -0.6 -0.5", "Size": "1", "Radiance": "30 25 20"}]}
    cam = camera("distributed.png", res=(96, 72), NumSamples="16", FocusDistance="4.6", ApertureSize="0.12")
    return dump(out_dir, "distributed", scene(cam, lights, mats, verts, objects))


def pathtrace_scene(out_dir):
    """Path tracing with NEE + importance sampling + Russian roulette: emissive sphere and emissive mesh quad,
    diffuse / mirror / glass objects, sample clamp, splitting factor 2."""
    mats = [MAT_DIFFUSE, {"_id": "2", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0 0 0", "SpecularReflectance": "0 0 0"},
            {"_id": "3", "_type": "mirror", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0 0 0", "SpecularReflectance": "0 0 0",
             "MirrorReflectance": "0.9 0.9 0.9"},
            {"_id": "4", "_type": "dielectric", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0 0 0", "SpecularReflectance": "0 0 0",
             "MirrorReflectance": "1 1 1", "AbsorptionCoefficient": "0.1 0.3 0.1", "RefractionIndex": "1.5"},
            {"_id": "5", "AmbientReflectance": "1 1 1", "DiffuseReflectance": "0.75 0.25 0.25", "SpecularReflectance": "0 0 0"}]
    verts = GROUND + "  0 3.5 1  -1.3 0.6 0  1.3 0.6 0.3  0 0.5 1.5  -2.5 0 -2  2.5 0 -2  2.5 3 -2  -2.5 3 -2  -0.6 3.2 -1  0.6 3.2 -1  0.6 3.2 -0.2  -0.6 3.2 -0.2"
    objects = {"Mesh": [{"_id": "1", "Material": "1", "Faces": {"_data": GROUND_FACES}},
                        {"_id": "2", "Material": "5", "Faces": {"_data": "9 10 11  9 11 12"}}],
               "LightMesh": {"_id": "3", "Material": "2", "Radiance": "12 11 9", "Faces": {"_data": "13 15 14  13 16 15"}},
               "Sphere": [{"_id": "1", "Material": "3", "Center": "6", "Radius": "0.6"},
                          {"_id": "2", "Material": "4", "Center": "7", "Radius": "0.6"},
                          {"_id": "3", "Material": "5", "Center": "8", "Radius": "0.5"}],
               "LightSphere": {"_id": "1", "Material": "2", "Center": "5", "Radius": "0.4", "Radiance": "30 30 30"}}
    cam = camera("pathtrace.exr", res=(64, 48), NumSamples="64", Renderer="PathTracing",
                 RendererParams="ImportanceSampling NextEventEstimation RussianRoulette", MinRecursionDepth="2", SplittingFactor="2",
                 SampleMaxVal="50")
    return dump(out_dir, "pathtrace", scene(cam, {"AmbientLight": "0 0 0"}, mats, verts, objects, MaxRecursionDepth="5"))


def pathtrace_uniform_scene(out_dir):
    """Path tracing without NEE / importance sampling / roulette (uniform hemisphere sampling)."""
    mats = [MAT_DIFFUSE, {"_id": "2", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0 0 0", "SpecularReflectance": "0 0 0"}]
    verts = GROUND + "  0 2.5 0.5  0 0.6 0"
    objects = {"Mesh": {"_id": "1", "Material": "1", "Faces": {"_data": GROUND_FACES}},
               "Sphere": {"_id": "1", "Material": "1", "Center": "6", "Radius": "0.6"},
               "LightSphere": {"_id": "1", "Material": "2", "Center": "5", "Radius": "0.8", "Radiance": "8 8 8"}}
    cam = camera("pt_uniform.exr", res=(48, 36), NumSamples="64", Renderer="PathTracing")
    return dump(out_dir, "pt_uniform", scene(cam, {"AmbientLight": "0 0 0"}, mats, verts, objects, MaxRecursionDepth="3"))


def emitters_scene(out_dir):
    """Whitted shading lit only by emitters: a sphere light (transformed) and a mesh light (two triangle lights)."""
    mats = [MAT_DIFFUSE, {"_id": "2", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0 0 0", "SpecularReflectance": "0 0 0"},
            {"_id": "3", "AmbientReflectance": "1 1 1", "DiffuseReflectance": "0.3 0.6 0.8", "SpecularReflectance": "0.4 0.4 0.4", "PhongExponent": "25"}]
    verts = GROUND + "  0 0 0  -1 0.6 0.3  1.1 0.5 -0.2  -2.5 2.5 -1  -1.5 2.5 -1  -1.5 2.5 0  -2.5 2.5 0"
    objects = {"Mesh": {"_id": "1", "Material": "1", "Faces": {"_data": GROUND_FACES}},
               "LightMesh": {"_id": "2", "Material": "2", "Radiance": "30 24 18", "Faces": {"_data": "8 10 9  8 11 10"}},
               "Sphere": [{"_id": "1", "Material": "3", "Center": "6", "Radius": "0.6"},
                          {"_id": "2", "Material": "1", "Center": "7", "Radius": "0.5"}],
               "LightSphere": {"_id": "1", "Material": "2", "Center": "5", "Radius": "0.3", "Radiance": "60 60 70", "Transformations": "t1"}}
    sc = scene(camera("emitters.png", res=(96, 72), NumSamples="4"), {"AmbientLight": "5 5 5"}, mats, verts, objects)
    sc["Scene"]["Transformations"] = {"Translation": {"_id": "1", "_data": "1.5 3 1.5"}}
    return dump(out_dir, "emitters", sc)


def cornell_scene(out_dir):
    """A closed box with diffuse walls lit by nothing but a mesh emitter in the ceiling (two emitter triangles), path
    traced with next-event estimation: every direct-light sample is a shadow ray that ends ON an emitter triangle, the
    case in which the reference's test `t < distance - epsilon` (src/ray_trace.cpp:232-241, src/lights.cpp:340-386)
    decides between light and shadow at rounding level."""
    white = {"_id": "1", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0.73 0.73 0.73", "SpecularReflectance": "0 0 0"}
    red = {"_id": "2", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0.65 0.05 0.05", "SpecularReflectance": "0 0 0"}
    green = {"_id": "3", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0.12 0.45 0.15", "SpecularReflectance": "0 0 0"}
    black = {"_id": "4", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0 0 0", "SpecularReflectance": "0 0 0"}
    box = [(-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1), (-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1)]
    lamp = [(-0.3, 1.995, -0.3), (0.3, 1.995, -0.3), (0.3, 1.995, 0.3), (-0.3, 1.995, 0.3)]
    block = [(x * 0.3 + 0.35, y * 0.6, z * 0.3 - 0.2) for x, y, z in CUBE_V]
    verts = verts_str(box + lamp + block + [(-0.45, 0.35, 0.25)])
    objects = {"Mesh": [{"_id": "1", "Material": "1", "Faces": {"_data": "1 4 3  1 3 2  5 6 7  5 7 8  1 2 6  1 6 5"}},   # floor, ceiling, back wall
                        {"_id": "2", "Material": "2", "Faces": {"_data": "1 5 8  1 8 4"}},                             # left wall
                        {"_id": "3", "Material": "3", "Faces": {"_data": "2 3 7  2 7 6"}},                             # right wall
                        {"_id": "4", "Material": "1", "Faces": {"_data": faces_str(CUBE_F, 12)}}],
               "LightMesh": {"_id": "5", "Material": "4", "Radiance": "18 15 10", "Faces": {"_data": "9 10 11  9 11 12"}},
               "Sphere": {"_id": "1", "Material": "1", "Center": "21", "Radius": "0.35"}}
    cam = camera("cornell.exr", res=(64, 64), pos="0 1 3.4", gaze_point="0 1 0", fovy="40", NumSamples="64", Renderer="PathTracing",
                 RendererParams="ImportanceSampling NextEventEstimation RussianRoulette", MinRecursionDepth="2")
    return dump(out_dir, "cornell", scene(cam, {"AmbientLight": "0 0 0"}, [white, red, green, black], verts, objects, MaxRecursionDepth="4"))


def meshlight_scene(out_dir):
    """The same box under recursive ray tracing: no emission term, no indirect light, so every lit pixel is the sum of
    its 2 x 16 shadow rays towards samples ON the emitter's triangles -- the image is the direct measure of how often
    such a ray counts as blocked by the emitter itself (src/ray_trace.cpp:232-241)."""
    path = cornell_scene(out_dir)
    doc = json.load(open(path))
    cam = doc["Scene"]["Cameras"]["Camera"]
    for k in ("Renderer", "RendererParams", "MinRecursionDepth"):
        cam.pop(k)
    cam["NumSamples"] = "16"
    cam["ImageName"] = "meshlight.exr"
    return dump(out_dir, "meshlight", doc)


def pathtrace_env_scene(out_dir):
    """Path tracing under an environment map (cosine-sampled light + background), NEE on."""
    os.makedirs(out_dir, exist_ok=True)
    write_exr(os.path.join(out_dir, "sky.exr"), sky_image())
    mats = [MAT_DIFFUSE, {"_id": "2", "_type": "mirror", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0 0 0",
                          "SpecularReflectance": "0 0 0", "MirrorReflectance": "0.9 0.9 0.9"}]
    verts = GROUND + "  -0.8 0.6 0  0.9 0.5 0.4"
    objects = {"Mesh": {"_id": "1", "Material": "1", "Faces": {"_data": GROUND_FACES}},
               "Sphere": [{"_id": "1", "Material": "2", "Center": "5", "Radius": "0.6"},
                          {"_id": "2", "Material": "1", "Center": "6", "Radius": "0.5"}]}
    lights = {"AmbientLight": "0 0 0", "SphericalDirectionalLight": {"_id": "1", "_type": "latlong", "ImageId": "1", "Sampler": "cosine"}}
    cam = camera("pt_env.exr", res=(48, 36), NumSamples="64", Renderer="PathTracing",
                 RendererParams="ImportanceSampling NextEventEstimation")
    sc = scene(cam, lights, mats, verts, objects, MaxRecursionDepth="3")
    sc["Scene"]["Textures"] = {"Images": {"Image": {"_id": "1", "_data": "sky.exr"}}}
    return dump(out_dir, "pt_env", sc)


def env_probe_scene(out_dir):
    """Angular-probe environment map (get_probe_uv, src/ray_trace.cpp:24-39) as background and cosine-sampled light."""
    os.makedirs(out_dir, exist_ok=True)
    write_exr(os.path.join(out_dir, "probe.exr"), sky_image(48, 48))
    mats = [MAT_DIFFUSE, {"_id": "2", "_type": "mirror", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0 0 0",
                          "SpecularReflectance": "0 0 0", "MirrorReflectance": "0.95 0.95 0.95"}]
    verts = GROUND + "  0 0.7 0"
    objects = {"Mesh": {"_id": "1", "Material": "1", "Faces": {"_data": GROUND_FACES}},
               "Sphere": {"_id": "1", "Material": "2", "Center": "5", "Radius": "0.7"}}
    lights = {"AmbientLight": "0 0 0", "SphericalDirectionalLight": {"_id": "1", "_type": "probe", "ImageId": "1", "Sampler": "cosine"}}
    sc = scene(camera("env_probe.exr", res=(96, 64), NumSamples="16"), lights, mats, verts, objects)
    sc["Scene"]["Textures"] = {"Images": {"Image": {"_id": "1", "_data": "probe.exr"}}}
    return dump(out_dir, "env_probe", sc)


def env_uniform_scene(out_dir):
    """Lat-long environment light sampled uniformly over the hemisphere (the else-branch of
    SphericalDirectionalLight::get_sample, src/lights.cpp:235)."""
    os.makedirs(out_dir, exist_ok=True)
    write_exr(os.path.join(out_dir, "sky.exr"), sky_image())
    mats = [MAT_DIFFUSE, {"_id": "2", "AmbientReflectance": "1 1 1", "DiffuseReflectance": "0.3 0.5 0.8", "SpecularReflectance": "0.4 0.4 0.4",
                          "PhongExponent": "15"}]
    verts = GROUND + "  -0.7 0.6 0  0.9 0.45 0.6"
    objects = {"Mesh": {"_id": "1", "Material": "1", "Faces": {"_data": GROUND_FACES}},
               "Sphere": [{"_id": "1", "Material": "2", "Center": "5", "Radius": "0.6"},
                          {"_id": "2", "Material": "1", "Center": "6", "Radius": "0.45"}]}
    lights = {"AmbientLight": "0 0 0", "SphericalDirectionalLight": {"_id": "1", "_type": "latlong", "ImageId": "1", "Sampler": "uniform"}}
    sc = scene(camera("env_uniform.exr", res=(96, 64), NumSamples="16"), lights, mats, verts, objects)
    sc["Scene"]["Textures"] = {"Images": {"Image": {"_id": "1", "_data": "sky.exr"}}}
    return dump(out_dir, "env_uniform", sc)


def background_scene(out_dir):
    """An 8-bit background image (DecalMode replace_background) looked up by pixel position for every ray that leaves
    the scene, reflected ones included (src/main.cpp:380-388,455-456, src/ray_trace.cpp:691-721), and a mesh whose
    face indices are shifted by _vertexOffset (src/parser.cpp:749-752)."""
    os.makedirs(out_dir, exist_ok=True)
    write_png(os.path.join(out_dir, "backdrop.png"), checker_image(48, 6))
    mats = [MAT_DIFFUSE, {"_id": "2", "_type": "mirror", "AmbientReflectance": "0 0 0", "DiffuseReflectance": "0.05 0.05 0.05",
                          "SpecularReflectance": "0.1 0.1 0.1", "MirrorReflectance": "0.9 0.9 0.9"},
            {"_id": "3", "AmbientReflectance": "1 1 1", "DiffuseReflectance": "0.8 0.5 0.2", "SpecularReflectance": "0.2 0.2 0.2", "PhongExponent": "12"}]
    verts = "-2.5 0 -2.5  2.5 0 -2.5  2.5 0 2.5  -2.5 0 2.5  " + verts_str([(x - 1.9, y, z - 0.3) for x, y, z in CUBE_V]) + "  0.6 0.7 0.2"
    objects = {"Mesh": [{"_id": "1", "Material": "1", "Faces": {"_data": GROUND_FACES}},
                        {"_id": "2", "Material": "3", "Faces": {"_data": faces_str(CUBE_F, 0), "_vertexOffset": "4"}}],
               "Sphere": {"_id": "1", "Material": "2", "Center": "13", "Radius": "0.7"}}
    sc = scene(camera("background.png", res=(112, 80)), TWO_LIGHTS, mats, verts, objects)
    sc["Scene"]["Textures"] = {"Images": {"Image": {"_id": "1", "_data": "backdrop.png"}},
                               "TextureMap": {"_id": "1", "_type": "image", "ImageId": "1", "DecalMode": "replace_background",
                                              "Interpolation": "bilinear"}}
    return dump(out_dir, "background", sc)


def textured_scene(out_dir):
    """The benchmark's textured scene with distribution effects in small: checkerboard ground, Perlin mesh, image-mapped
    wall and sphere, glossy mirror (roughness 0.1), two area lights, depth of field, 64 samples per pixel."""
    from raytracer_amd import scenegen
    return scenegen.write_textured_scene(out_dir, name="textured", n=24, width=96, height=72, spp=64)


SCENES = {
    "materials": (materials_scene, "exact"),
    "textures": (textures_scene, "exact"),
    "brdfs": (brdf_scene, "exact"),
    "transforms": (transforms_scene, "exact"),
    "envmap": (envmap_scene, "stochastic"),
    "distributed": (distributed_scene, "stochastic"),
    "pathtrace": (pathtrace_scene, "stochastic"),
    "pt_uniform": (pathtrace_uniform_scene, "stochastic"),
    "emitters": (emitters_scene, "stochastic"),
    "cornell": (cornell_scene, "stochastic"),
    "meshlight": (meshlight_scene, "stochastic"),
    "pt_env": (pathtrace_env_scene, "stochastic"),
    "env_probe": (env_probe_scene, "stochastic"),
    "env_uniform": (env_uniform_scene, "stochastic"),
    "background": (background_scene, "exact"),
    "textured": (textured_scene, "stochastic"),
}
