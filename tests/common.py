"""Shared helpers for the tests: golden fixture loading and comparison metrics."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_SCENES = ("cube", "f16", "teapot", "bunny")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"golden_{name}.npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    g["scene"] = {"nodes": g["nodes"], "triangles": g["triangles"], "positions": g["positions"],
                  "tex_coords": g.get("tex_coords"), "normals": g.get("normals")}
    return g


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def mixed_material_mask(hit_id, mat_idx, h, w, block=4):
    """Pixels whose 4x4 block holds hit pixels of more than one material.  llvmpipe
    lacks GL_EXT_nonuniform_qualifier, so the reference shader's
    modelTextures[triangles[i].matIdx] (ray_trace_and_shade.frag:221) is indexed with a
    non-dynamically-uniform value there — undefined by GLSL 4.60; llvmpipe uses one
    lane's index for the whole 4x4 fragment block.  Such pixels are excluded from
    colour comparison against llvmpipe (and counted)."""
    hit = hit_id != 0xFFFFFFFF
    mat = np.where(hit, mat_idx[np.minimum(hit_id, len(mat_idx) - 1)], 255).reshape(h, w)
    mask = np.zeros((h, w), bool)
    for y0 in range(0, h, block):
        for x0 in range(0, w, block):
            blk = mat[y0:y0 + block, x0:x0 + block]
            if len(set(np.unique(blk)) - {255}) > 1:
                mask[y0:y0 + block, x0:x0 + block] = True
    return mask.reshape(-1)


def rgb_err_255(a, b):
    """max |clamp(a) - clamp(b)| per pixel in units of 1/255 (the RGBA32F target is
    unclamped; the display path clamps)."""
    a = np.clip(np.asarray(a, np.float64)[..., :3], 0, 1)
    b = np.clip(np.asarray(b, np.float64)[..., :3], 0, 1)
    return np.abs(a - b).max(-1) * 255.0
