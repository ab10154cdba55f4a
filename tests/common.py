"""Shared helpers for the tests: golden fixture loading and comparison metrics."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# every model under the reference's res/models: the first four from round 1, the other four from round 2
# (make_golden.py: FIRST_CASES / MORE_CASES; "testobj" = testObj.obj, five materials, F16t.bmp in both sampler slots)
GOLDEN_SCENES = ("cube", "f16", "teapot", "bunny", "armadillo", "dome", "sphere", "testobj")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"golden_{name}.npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    if "nodes" not in g:
        return g
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
    lane's index for the whole 4x4 fragment block.  Kept to DOCUMENT that behaviour
    (test_oracle_golden.py): the parity assertions use reference_colour(), which is
    defined on every pixel."""
    hit = hit_id != 0xFFFFFFFF
    mat = np.where(hit, mat_idx[np.minimum(hit_id, len(mat_idx) - 1)], 255).reshape(h, w)
    mask = np.zeros((h, w), bool)
    for y0 in range(0, h, block):
        for x0 in range(0, w, block):
            blk = mat[y0:y0 + block, x0:x0 + block]
            if len(set(np.unique(blk)) - {255}) > 1:
                mask[y0:y0 + block, x0:x0 + block] = True
    return mask.reshape(-1)


def reference_colour(g):
    """The reference colour for EVERY pixel of a fixture: b_fused_uniform where the scene has two textures (the
    unmodified shader run on llvmpipe once per image with that image bound to both sampler slots, composited by
    the hit triangle's matIdx — make_golden.py:fused_uniform; the shader's non-uniform sampler index is then
    immaterial, so mixed-material blocks are defined too), else b_fused."""
    return (g["b_fused_uniform"] if "b_fused_uniform" in g else g["b_fused"]).reshape(-1, 4)


def rgb_err_255(a, b):
    """max |clamp(a) - clamp(b)| per pixel in units of 1/255 (the RGBA32F target is
    unclamped; the display path clamps)."""
    a = np.clip(np.asarray(a, np.float64)[..., :3], 0, 1)
    b = np.clip(np.asarray(b, np.float64)[..., :3], 0, 1)
    return np.abs(a - b).max(-1) * 255.0


DISPATCH_KEYS = ("org", "shadow_dir", "reflect_dir", "ambient_dir")
# ambientDir goes through a LINEAR texture fetch whose weights depend on the rasteriser's
# interpolated fragPos (not bit-reproducible); everything else must match llvmpipe bit for bit


def check_dispatch(got, g):
    """got: dict of (h, w, 3) arrays; g: golden_dispatch fixture (llvmpipe outputs b_*)."""
    surface = g["normal_mat"][..., 3] != np.float32(1e30)
    assert 2000 < surface.sum() < surface.size
    for k in DISPATCH_KEYS:
        assert np.all(got[k][~surface] == 0), k                      # discarded fragments keep NO_RAY
        assert np.all(g["b_" + k][~surface] == 0), k
    # all four targets bit for bit.  ambientDir too since round 2: llvmpipe converts the noise texels as x * fl(1/255), and the
    # fixtures fetch at texel centres as src/main.rs:161-165 sets the offsets up, where LINEAR returns the texel itself
    for k in DISPATCH_KEYS:
        assert np.array_equal(bits(got[k]), bits(g["b_" + k])), k
    n = np.linalg.norm(got["ambient_dir"][surface].astype(np.float64), axis=-1)
    assert np.abs(n - 1).max() < 1e-6


def atrium_golden():
    """The headline scene rebuilt (procedural + host BVH builder, deterministic) with the llvmpipe outputs of
    tests/golden/golden_atrium.npz: dict(scene, dir, org, tex0, tex1, b_trace, b_fused, w, h).  The digest
    check guards the rebuild: fixtures and scene must belong together."""
    import hashlib
    import math
    from raytracer_amd import host, scenes
    g = load_golden("atrium")
    w, h = int(g["width"]), int(g["height"])
    tri, pos, uv, nrm, cam = scenes.atrium()
    m = host.Model.from_arrays(tri, pos, uv, nrm)
    m.build_bvh()
    sc = m.scene_arrays()
    digest = hashlib.sha256(b"".join(np.ascontiguousarray(sc[k]).tobytes() for k in ("nodes", "triangles", "positions", "tex_coords", "normals"))).hexdigest()
    assert digest == str(g["scene_sha256"]), "the rebuilt atrium differs from the one the fixture was made with"
    d = np.asarray(cam["direction"], np.float32)
    d /= np.linalg.norm(d)
    camera = host.Camera(16 / 9, fov=math.radians(cam["fov_deg"]), position=cam["position"], direction=d)
    dirs, orgs = camera.primary_rays(w, h)
    return {"scene": sc, "dir": dirs, "org": orgs, "tex0": scenes.checker_texture(), "tex1": scenes.noise_texture(),
            "b_trace": g["b_trace"].reshape(-1, 4), "b_fused": g["b_fused"].reshape(-1, 4),
            "b_fused_uniform": g["b_fused_uniform"].reshape(-1, 4), "w": w, "h": h}


def soup_golden():
    """BASELINE config 4's kind of scene at fixture size (tests/golden/golden_soup.npz, make_golden.py:soup):
    dict(scene, dir, org, rand_org, rand_dir, b_trace, b_fused, b_trace_rand, w, h)."""
    import hashlib
    import math
    from raytracer_amd import host, scenes
    g = load_golden("soup")
    w, h = int(g["width"]), int(g["height"])
    tri, pos, cam = scenes.triangle_soup(100_000, seed=3, extent=12.0, size=0.5)
    m = host.Model.from_arrays(tri, pos)
    m.build_bvh()
    sc = m.scene_arrays()
    digest = hashlib.sha256(b"".join(np.ascontiguousarray(sc[k]).tobytes() for k in ("nodes", "triangles", "positions"))).hexdigest()
    assert digest == str(g["scene_sha256"]), "the rebuilt soup differs from the one the fixture was made with"
    camera = host.Camera(w / h, fov=math.radians(cam["fov_deg"]), position=cam["position"], direction=cam["direction"])
    dirs, orgs = camera.primary_rays(w, h)
    return {"scene": sc, "dir": dirs, "org": orgs, "rand_org": g["rand_org"].reshape(-1, 3), "rand_dir": g["rand_dir"].reshape(-1, 3),
            "b_trace": g["b_trace"].reshape(-1, 4), "b_fused": g["b_fused"].reshape(-1, 4), "b_trace_rand": g["b_trace_rand"].reshape(-1, 4), "w": w, "h": h}


def check_raycreate(generate):
    """generate(w, h, inv_proj_view16, near, far) -> (h*w, 3) directions, checked against the llvmpipe runs of
    ray_trace/ray_create.frag in tests/golden/golden_raycreate.npz: the shader uses the rasteriser's interpolated
    fragPos, so 95+ % of the components are bit-equal and the rest within 2e-5 (3e-4 for the camera far from the origin,
    whose unprojection cancels more).  The shader's arithmetic itself is pinned bit for bit on llvmpipe's own fragPos values:
    tests/test_oracle_golden.py:test_ray_create_arithmetic_bit_exact_on_the_rasterisers_frag_pos."""
    g = load_golden("raycreate")
    near, far = (float(x) for x in g["near_far"])
    for k in range(int(g["count"])):
        w, h = (int(x) for x in g[f"size{k}"])
        want = g[f"b_dir{k}"].reshape(-1, 3)
        got = np.asarray(generate(w, h, g[f"inv_proj_view{k}"], near, far), np.float32).reshape(-1, 3)
        assert np.abs(got - want).max() <= (2e-5 if k < 3 else 3e-4), (k, float(np.abs(got - want).max()))
        assert (bits(got) == bits(want)).mean() >= 0.95, k


def check_fullsize_rows(name, scene, hits, w, h):
    """hits: the (t, u, v, idx-bits) records of the whole frame of bench.py workload `name`, row 0 = bottom, against the digests of
    ray_trace.frag on llvmpipe AT THE CONFIG'S STATED SIZE (tests/golden/golden_fullsize.npz, make_golden.py:fullsize): every
    pixel row must hash to llvmpipe's row, i.e. the full frame is the reference's bit for bit."""
    import hashlib
    g = load_golden("fullsize")
    assert tuple(int(x) for x in g[f"{name}_size"]) == (w, h)
    digest = hashlib.sha256(b"".join(np.ascontiguousarray(scene[k]).tobytes() for k in ("nodes", "triangles", "positions"))).hexdigest()
    assert digest == str(g[f"{name}_scene_sha256"]), "the rebuilt scene differs from the one the fixture was made with"
    rec = np.ascontiguousarray(hits, dtype=np.float32).reshape(h, w, 4)
    assert np.array_equal((rec[..., 0] < 1e30).sum(axis=1), g[f"{name}_row_hits"]), "hit counts per row"
    rows = np.stack([np.frombuffer(hashlib.sha256(rec[y].tobytes()).digest(), np.uint8) for y in range(h)])
    wrong = np.where((rows != g[f"{name}_row_sha256"]).any(axis=1))[0]
    assert len(wrong) == 0, f"{len(wrong)} of {h} rows differ from llvmpipe's, first row {wrong[0]}"


def block_means(rgb, block=16):
    """(h, w, 3) -> (ceil(h / block), ceil(w / block), 3): the mean over each block's pixels, accumulated in float64."""
    h, w = rgb.shape[:2]
    ys, xs = np.arange(h) // block, np.arange(w) // block
    nb_y, nb_x = ys[-1] + 1, xs[-1] + 1
    flat = (ys[:, None] * nb_x + xs[None, :]).reshape(-1)
    out = np.stack([np.bincount(flat, weights=rgb[..., c].reshape(-1).astype(np.float64), minlength=nb_y * nb_x) for c in range(3)], axis=-1)
    return (out / np.bincount(flat, minlength=nb_y * nb_x)[:, None]).reshape(nb_y, nb_x, 3)


def check_fullsize_colour(name, rgba, w, h, tol_255):
    """rgba: the fused shader's whole frame of bench.py workload `name` (row 0 = bottom) against llvmpipe's frame at the config's
    stated size, kept as the mean RGB of every 16 x 16 pixel block (golden_fullsize.npz): every block within tol_255 / 255."""
    g = load_golden("fullsize")
    got = block_means(np.asarray(rgba, np.float32).reshape(h, w, -1)[..., :3])
    err = np.abs(got - g[f"{name}_block_mean_rgb"]).max() * 255
    assert err <= tol_255, f"block means differ by {err:.4f}/255"


def group_digests(a, rows=16):
    """(h, w, c) float32 -> one SHA-256 per group of `rows` pixel rows (tests/golden/make_golden.py:split4k)."""
    import hashlib
    a = np.ascontiguousarray(a, dtype=np.float32)
    return np.stack([np.frombuffer(hashlib.sha256(a[y:y + rows].tobytes()).digest(), np.uint8) for y in range(0, a.shape[0], rows)])


def check_group_digests(g, key, a):
    wrong = np.where((group_digests(a) != g[key]).any(axis=1))[0]
    assert len(wrong) == 0, f"{key}: {len(wrong)} of {len(g[key])} row groups differ from llvmpipe's, first group {wrong[0]} (rows {16 * wrong[0]}..)"
