"""Asset layer: OBJ + MTL + image textures -> the arrays the C-ABI takes, and a
frame writer in place of the display pass.

  load_obj_scene   ResourceManager::get_model / get_texture (src/resource/resource_manager.rs:43-91) with
                   ResourceParser::parse_model / parse_material_lib (resource_parser.rs:13-109): the model's
                   material ids are in first-`usemtl` order (model.rs:89-97); material i's map_Kd image
                   becomes albedo slot i.  The fused shader has modelTextures[2] (ray_trace_and_shade.frag:21):
                   without a material table triangles of further materials sample slot (mat_idx & 1), so more
                   than two textured materials are reported; with `shading_materials` handed to
                   HipRayTracer.upload_materials every material has its own slot (16) and its MTL Ka/Kd/Ks/Ns.
  frame_to_rgb8 /  util/display.frag:9 writes the colour target to an 8-bit unorm back buffer: clamp to
  write_ppm /      [0,1], x255, round to nearest; frames are stored bottom row first (GL), images top row
  write_png        first.

Everything here is host-side marshalling (numpy / PIL); no computation of the hot path.
"""
import os

import numpy as np

from . import host


def load_image_rgb8(path):
    """Decode an image file to (h, w, 3) uint8, first row = top row, as image::open(..).to_rgb8() does
    (src/resource/resource.rs:30-33).  Binary PPM (P6) is read without PIL."""
    with open(path, "rb") as f:
        head = f.read(2)
    if head == b"P6":
        return _read_ppm(path)
    from PIL import Image   # BMP / PNG decoding is lossless: identical texels to the `image` crate
    return np.asarray(Image.open(path).convert("RGB")).copy()


def _read_ppm(path):
    with open(path, "rb") as f:
        data = f.read()
    fields, pos = [], 0
    while len(fields) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        fields.append(data[pos:end])
        pos = end
    pos += 1
    if fields[0] != b"P6" or int(fields[3]) != 255:
        raise ValueError(f"{path}: only 8-bit binary PPM (P6) is supported")
    w, h = int(fields[1]), int(fields[2])
    return np.frombuffer(data, np.uint8, w * h * 3, pos).reshape(h, w, 3).copy()


def load_obj_scene(obj_path, texture_dirs=(), bvh_threads=1):
    """Returns dict(model, scene, materials, textures, shading_materials, warnings).
    `scene` is ready for HipRayTracer.upload_scene; textures[i] (or None) is material i's map_Kd image for
    upload_texture(i, ...), i < 16.  Slots 0 and 1 are what the reference shader samples (modelTextures[2]);
    `shading_materials` is the table for upload_materials (Ka / Kd / Ks / Ns of material.rs:5-16 and
    texture slot i, or -1), with which materials 2..15 get their own textures and every material its MTL colours."""
    from .renderer import MAX_TEXTURE_SLOTS, SHADING_MATERIAL_DTYPE
    base = os.path.dirname(os.path.abspath(obj_path))
    with open(obj_path, "r", encoding="utf-8", errors="replace") as f:
        model = host.ResourceParser.parse_model(f.read())
    model.build_bvh(bvh_threads)
    libs = {}
    for lib in model.material_libs():
        with open(os.path.join(base, lib), "r", encoding="utf-8", errors="replace") as f:
            for name, mat in host.ResourceParser.parse_material_lib(f.read()):
                libs.setdefault(name, mat)
    names = model.materials()
    warnings, materials, textures = [], [], [None] * max(2, min(len(names), MAX_TEXTURE_SLOTS))
    table = np.zeros(len(names), SHADING_MATERIAL_DTYPE)
    search = [base] + [os.path.abspath(d) for d in texture_dirs]
    for i, name in enumerate(names):
        mat = libs.get(name)
        if mat is None:
            warnings.append(f"material '{name}' (id {i}) is not defined in any mtllib: default material")
            mat = host.ResourceParser.parse_material_lib("newmtl _default\n")[0][1]
        materials.append((name, mat))
        table[i] = (mat["ambient_color"], mat["diffuse_color"], mat["specular_color"], mat["specular_exp"], -1)
        tex = mat["diffuse_tex"]
        if tex is None:
            continue
        if i >= MAX_TEXTURE_SLOTS:
            warnings.append(f"material '{name}' (id {i}) has map_Kd {tex}, but there are {MAX_TEXTURE_SLOTS} texture slots: untextured")
            continue
        if i >= 2:
            warnings.append(f"material '{name}' (id {i}) has map_Kd {tex}: the reference shader has modelTextures[2] (its triangles "
                            f"would sample slot {i & 1}); slot {i} is used only with upload_materials(shading_materials)")
        for d in search:
            p = os.path.join(d, tex)
            if os.path.exists(p):
                textures[i] = load_image_rgb8(p)
                table[i]["texture_slot"] = i
                break
        else:
            warnings.append(f"texture {tex} of material '{name}' not found in {search}: slot {i} stays white")
    return {"model": model, "scene": model.scene_arrays(), "materials": materials, "textures": textures,
            "shading_materials": table, "warnings": warnings}


def frame_to_rgb8(rgba, width, height):
    """(h*w, 4) or (h, w, 4) float frame, row 0 = bottom -> (h, w, 3) uint8, row 0 = top."""
    a = np.asarray(rgba, dtype=np.float32).reshape(height, width, -1)[::-1, :, :3]
    a = np.nan_to_num(a, nan=0.0, posinf=1.0, neginf=0.0)
    return np.rint(np.clip(a, 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)


def write_ppm(path, rgba, width, height):
    img = frame_to_rgb8(rgba, width, height)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (width, height))
        f.write(np.ascontiguousarray(img).tobytes())


def write_png(path, rgba, width, height):
    from PIL import Image
    Image.fromarray(frame_to_rgb8(rgba, width, height), "RGB").save(path)
