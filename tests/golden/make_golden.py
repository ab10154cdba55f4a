#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ — BUILD CONTAINER ONLY.

Runs the UNMODIFIED reference shaders (read from /root/reference/res/shaders at
run time, never copied) on Mesa llvmpipe through oracle/glsl_ref.c and stores
inputs + outputs as compressed .npz files:

  inputs   nodes / triangles / positions / tex_coords / normals  (the five SSBO
           slices, built from the reference's res/models/*.obj by the host layer),
           dir / org ray arrays (h, w, 3; row 0 = bottom), tex0 / tex1 RGB8
  outputs  b_trace  (h, w, 4)  ray_trace/ray_trace.frag   -> (t, u, v, bits(idx))
           b_fused  (h, w, 4)  ray_trace_and_shade.frag   -> fragCol
           b_fused_uniform (h, w, 4)  the same shader with one image bound to both sampler slots, once per image,
                    composited per pixel by the hit triangle's matIdx: defined colour for every pixel (f16, atrium)
           b_trace_rand (n, 1, 4) same shader on incoherent rays (secondary-ray stand-in)
  golden_dispatch.npz: ray_trace/ray_dispatcher.frag on a G-buffer (positions, flat normals,
           material bits / 1e30 "no surface") derived from the bunny frame, a seeded 64x64
           unit-vector noise texture -> b_org, b_shadow_dir, b_reflect_dir, b_ambient_dir

  golden_atrium.npz: the two shaders' outputs for the procedural headline scene (262 944 triangles, 240x135);
           the scene itself is rebuilt by the test (deterministic) and pinned by a digest

  golden_raycreate.npz: ray_trace/ray_create.frag for three cameras (inputs + RGB32F direction target)
  golden_soup.npz: the same as golden_atrium for a 100 000-triangle seeded soup (camera frame + 8192 random rays)

The fixtures are data only.  Usage:  python tests/golden/make_golden.py [--only-atrium | --only-soup | --only-raycreate | --only-more-models | --only-dispatch2 | --add-uniform]
"""
import math
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from common import bits, block_means, group_digests  # noqa: E402
from oracles import OracleA, OracleB  # noqa: E402
from raytracer_amd import host  # noqa: E402

REF = "/root/reference/res"
W, H = 160, 120


def load_rgb(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB")).copy()


def camera_for(pos, dist, offset=(1.0, 0.4, 0.6), fov_deg=120.0):
    lo, hi = pos.min(0), pos.max(0)
    c = (lo + hi) / 2
    r = np.linalg.norm(hi - lo) / 2
    off = np.asarray(offset, np.float64)
    eye = c + dist * r * off / np.linalg.norm(off)
    d = c - eye
    d /= np.linalg.norm(d)
    return host.Camera(W / H, fov=math.radians(fov_deg), position=eye.astype(np.float32), direction=d.astype(np.float32))


def random_rays(pos, n, seed):
    rng = np.random.default_rng(seed)
    lo, hi = pos.min(0), pos.max(0)
    c, e = (lo + hi) / 2, (hi - lo)
    org = (c + rng.uniform(-1.0, 1.0, (n, 3)) * e).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    d[::97] = 0.0  # NO_RAY pixels (ray_trace.frag:6,133)
    return org.reshape(n, 1, 3), d.reshape(n, 1, 3)


def fused_uniform(B, sc, dirs, orgs, has_tc, has_n, tex0, tex1, b_trace):
    """Reference colour for EVERY pixel, mixed-material blocks included.  llvmpipe has no
    GL_EXT_nonuniform_qualifier, so modelTextures[triangles[i].matIdx] (ray_trace_and_shade.frag:221) takes one
    lane's index for a whole 4x4 fragment block — undefined behaviour by GLSL 4.60, visible where two materials
    meet.  With the SAME image bound to both sampler slots the index no longer matters and the unmodified shader's
    output is defined everywhere: one pass with (tex0, tex0), one with (tex1, tex1), and each pixel is taken from
    the pass that bound its own material's image (matIdx of the triangle the pixel hit; sky pixels agree in both)."""
    pass0, _ = B.fused(sc, dirs, orgs, has_tc, has_n, tex0, tex0)
    pass1, _ = B.fused(sc, dirs, orgs, has_tc, has_n, tex1, tex1)
    idx = np.ascontiguousarray(b_trace[..., 3]).view(np.uint32)
    hit = b_trace[..., 0] < 1000.0                                   # the fused shader's hit test, :206
    mat = np.where(hit, sc["triangles"]["mat_idx"][np.minimum(idx, len(sc["triangles"]) - 1)] & 1, 0)
    assert np.array_equal(pass0[~hit], pass1[~hit])
    return np.where((mat == 1)[..., None], pass1, pass0).astype(np.float32)


def add_uniform_to_existing():
    """Adds b_fused_uniform to the committed golden_f16.npz / golden_atrium.npz without touching their other arrays."""
    from common import atrium_golden
    B = OracleB()
    path = os.path.join(HERE, "golden_f16.npz")
    g = dict(np.load(path, allow_pickle=False))
    sc = {"nodes": g["nodes"], "triangles": g["triangles"], "positions": g["positions"], "tex_coords": g["tex_coords"], "normals": g["normals"]}
    again, _ = B.fused(sc, g["dir"], g["org"], bool(g["has_tex_coords"]), bool(g["has_normals"]), g["tex0"], g["tex1"])
    assert np.array_equal(again.view(np.uint32), g["b_fused"].view(np.uint32)), "llvmpipe no longer reproduces the committed b_fused"
    g["b_fused_uniform"] = fused_uniform(B, sc, g["dir"], g["org"], bool(g["has_tex_coords"]), bool(g["has_normals"]), g["tex0"], g["tex1"], g["b_trace"])
    np.savez_compressed(path, **g)
    diff = np.abs(np.clip(g["b_fused_uniform"], 0, 1) - np.clip(g["b_fused"], 0, 1))[..., :3].max(-1) * 255 > 1
    print(f"f16: b_fused_uniform differs from b_fused (> 1/255) on {int(diff.sum())} of {diff.size} pixels")
    a = atrium_golden()
    path = os.path.join(HERE, "golden_atrium.npz")
    g = dict(np.load(path, allow_pickle=False))
    h, w = a["h"], a["w"]
    g["b_fused_uniform"] = fused_uniform(B, a["scene"], a["dir"], a["org"], True, True, a["tex0"], a["tex1"], g["b_trace"].reshape(h, w, 4))
    np.savez_compressed(path, **g)
    diff = np.abs(np.clip(g["b_fused_uniform"], 0, 1) - np.clip(g["b_fused"].reshape(h, w, 4), 0, 1))[..., :3].max(-1) * 255 > 1
    print(f"atrium: b_fused_uniform differs from b_fused (> 1/255) on {int(diff.sum())} of {diff.size} pixels")


FIRST_CASES = [
    # name, obj, camera distance (in bbox radii), textures
    ("cube", "cube.obj", 1.6, None),
    ("f16", "f16.obj", 0.55, ("F16s.bmp", "F16t.bmp")),
    ("teapot", "teapot.obj", 0.8, None),
    ("bunny", "bunny.obj", 0.7, None),
]
# the rest of res/models (round 2).  testObj.obj has texture coordinates and FIVE materials; the fused shader declares
# modelTextures[2] (:21), so an index above 1 is out of the array's bounds and its colour undefined — except when every slot
# holds the same image: F16t.bmp is bound to both, and the fixture pins the colour for that binding only.  (uv_debug.png and
# F16s.bmp were tried first: the backdrop minifies them to noise, llvmpipe filters RGBA8 texels with 8-bit weights, and
# between full-contrast neighbours that quantisation alone reaches 1.2/255 on 1-6 pixels of the frame — the fixture would
# test the rasteriser's filter precision, not the shader.  With F16t.bmp the largest difference is 0.89/255, as on f16.)
MORE_CASES = [
    ("armadillo", "armadillo_lowres.obj", 0.6, None),
    ("dome", "dome.obj", 0.8, None),
    ("sphere", "sphere.obj", 0.62, None),
    # camera aimed at the small objects at the origin (materials 1, 2, 4) in front of the backdrop (0) under the light panel (3)
    ("testobj", "testObj.obj", 0.6, ("F16t.bmp", "F16t.bmp"), (1, 2, 4), (-0.7, 0.5, 1.0)),
]


def main(cases=FIRST_CASES, with_extras=True):
    B = OracleB()
    print("oracle B:", B.renderer)
    for name, obj, dist, texs, *focus in cases:
        model = host.ResourceParser.parse_model(open(os.path.join(REF, "models", obj)).read())
        model.build_bvh()
        sc = model.scene_arrays()
        aim = sc["positions"]
        if focus:     # frame only the triangles of these materials
            t = sc["triangles"][np.isin(sc["triangles"]["mat_idx"], focus[0])]
            aim = sc["positions"][np.unique(np.concatenate([t["p0"], t["p1"], t["p2"]]))]
        cam = camera_for(aim, dist, *focus[1:2])
        dirs, orgs = cam.primary_rays(W, H)
        tex0 = tex1 = None
        if texs:
            tex0, tex1 = (load_rgb(os.path.join(REF, "textures", t)) for t in texs)
        b_trace, _ = B.trace(sc, dirs, orgs)
        b_fused, _ = B.fused(sc, dirs, orgs, model.has_tex_coords(), model.has_normals(), tex0, tex1)
        ro, rd = random_rays(sc["positions"], 4096, seed=1234)
        b_rand, _ = B.trace(sc, rd, ro)
        out = {
            "nodes": sc["nodes"], "triangles": sc["triangles"], "positions": sc["positions"],
            "dir": dirs, "org": orgs, "rand_dir": rd, "rand_org": ro,
            "has_tex_coords": np.uint8(model.has_tex_coords()), "has_normals": np.uint8(model.has_normals()),
            "b_trace": b_trace, "b_fused": b_fused, "b_trace_rand": b_rand,
            "renderer": np.array(B.renderer),
        }
        if sc["tex_coords"] is not None:
            out["tex_coords"] = sc["tex_coords"]
        if sc["normals"] is not None:
            out["normals"] = sc["normals"]
        if tex0 is not None:
            out["tex0"], out["tex1"] = tex0, tex1
            out["b_fused_uniform"] = fused_uniform(B, sc, dirs, orgs, model.has_tex_coords(), model.has_normals(), tex0, tex1, b_trace)
        path = os.path.join(HERE, f"golden_{name}.npz")
        np.savez_compressed(path, **out)
        hits = int((b_trace[..., 0] < 1e30).sum())
        print(f"{name}: {len(sc['triangles'])} tris, {len(sc['nodes'])} nodes, {hits}/{W * H} hit pixels, "
              f"{int((b_rand[..., 0] < 1e30).sum())}/4096 random hits -> {os.path.getsize(path) / 1e6:.2f} MB")

    if not with_extras:
        return
    # secondary-ray dispatcher on a G-buffer derived from the bunny frame just written
    g = dict(np.load(os.path.join(HERE, "golden_bunny.npz")))
    tr = g["b_trace"]
    t = tr[..., 0]
    hit = t < 1e30
    idx = np.ascontiguousarray(tr[..., 3]).view(np.uint32)
    gpos = (g["org"] + g["dir"] * t[..., None]).astype(np.float32)
    gpos[~hit] = 0
    tri, P = g["triangles"], g["positions"]
    p0, p1, p2 = P[tri["p0"][idx]], P[tri["p1"][idx]], P[tri["p2"][idx]]
    nrm = np.cross(p1 - p0, p2 - p0)
    nrm = (nrm / np.maximum(1e-30, np.linalg.norm(nrm, axis=-1, keepdims=True))).astype(np.float32)
    nm = np.zeros((H, W, 4), np.float32)
    nm[..., :3] = nrm
    mat_bits = (idx & 1).astype(np.uint32).view(np.float32)          # floatBitsToInt(normalMat.w), :31
    nm[..., 3] = np.where(hit, mat_bits, np.float32(1e30))           # 1e30 = "no material", main.rs:130
    rng = np.random.default_rng(7)
    v = rng.normal(size=(64, 64, 3))
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    noise = np.clip(np.round((v * 0.5 + 0.5) * 255), 0, 255).astype(np.uint8)   # "is unit length vector", :20
    light = np.array([3.0, 20.0, -4.0], np.float32)
    cam_pos = g["org"][0, 0].astype(np.float32)
    nos = np.array([5 / 64, 9 / 64, W / 64, H / 64], np.float32)    # offset on the texel grid, scale = frame / noise (main.rs:161-166)
    d = B.dispatch(gpos, nm, noise, light, cam_pos, nos)
    np.savez_compressed(os.path.join(HERE, "golden_dispatch.npz"), position=gpos, normal_mat=nm, noise=noise, light_pos=light,
                        camera_pos=cam_pos, noise_offset_scale=nos, b_org=d["org"], b_shadow_dir=d["shadow_dir"],
                        b_reflect_dir=d["reflect_dir"], b_ambient_dir=d["ambient_dir"], renderer=np.array(B.renderer))
    print("dispatch:", int(hit.sum()), "surface pixels of", W * H, "->", os.path.getsize(os.path.join(HERE, "golden_dispatch.npz")) / 1e6, "MB")

    # special values: axis-parallel rays (1/0 = inf, 0*inf = NaN in the slab test),
    # origins on box planes, denormal coordinates (llvmpipe runs FTZ/DAZ)
    pos = np.array([[0, 0, 0], [2, 0, 0], [0, 2, 0]], np.float32)
    tri = np.array([[0, 1, 2, 0]], np.uint32).view(host.TRIANGLE_DTYPE).reshape(-1)
    nodes = np.zeros(1, host.NODE_DTYPE)
    nodes["bounds"][0] = [0, 0, 0, 1, 2, 0]
    nodes["is_leaf"] = 1
    nodes["b"] = 1
    sc = {"nodes": nodes, "triangles": tri, "positions": pos, "tex_coords": None, "normals": None}
    hh = ww = 8
    orgs = np.zeros((hh, ww, 3), np.float32)
    dirs = np.zeros((hh, ww, 3), np.float32)
    orgs[..., 0] = np.array([0, 1, 0.5, 1, 0, 1, 0, 0.5], np.float32)[None, :]
    dirs[..., 0] = np.array([0, 0, 0, -0.0, -0.0, 1e-30, 1e-30, -0.0], np.float32)[None, :]
    orgs[..., 1] = np.array([0.25, 0, 2, 0.25, 1e-40, -1e-40, 0.25, 0.25], np.float32)[:, None]
    orgs[..., 2] = 1
    dirs[..., 2] = -1
    orgs[6, :, 0] = np.array([-1e-40, -1e-39, -1e-38, -2e-38, 1e-40, 0, -0.0, 1e-45], np.float32)
    b_trace, _ = B.trace(sc, dirs, orgs)
    np.savez_compressed(os.path.join(HERE, "golden_special.npz"), nodes=nodes, triangles=tri, positions=pos, dir=dirs, org=orgs,
                        b_trace=b_trace, renderer=np.array(B.renderer))
    print("special:", int((b_trace[..., 0] < 1e30).sum()), "hits of", hh * ww)


def dispatch2():
    """A second dispatcher fixture (round 2): ray_dispatcher.frag on the G-buffer of the f16 frame — interpolated vertex normals
    and two materials, made from the committed b_trace records by the oracle's restatement of the rasteriser pass — with the
    reference's own 512 x 512 blue-noise image (res/textures/blue_noise.png), a light below the model, the camera of the frame and
    a texel-grid offset in both axes.  Stored like golden_dispatch.npz; the noise travels as a 128 x 128 crop."""
    from oracles import OracleA
    B, A = OracleB(), OracleA()
    g = dict(np.load(os.path.join(HERE, "golden_f16.npz")))
    sc = {"nodes": g["nodes"], "triangles": g["triangles"], "positions": g["positions"], "tex_coords": g["tex_coords"], "normals": g["normals"]}
    h, w = g["dir"].shape[:2]
    pos, nm = A.surface_from_hits(sc, g["org"], g["dir"], g["b_trace"], 1)
    gpos, gnm = pos.reshape(h, w, 3), nm.reshape(h, w, 4)
    noise = load_rgb(os.path.join(REF, "textures", "blue_noise.png"))[192:320, 64:192].copy()
    light = np.array([-6.0, -9.0, 2.5], np.float32)
    cam_pos = g["org"][0, 0].astype(np.float32)
    nos = np.array([37 / 128, 101 / 128, w / 128, h / 128], np.float32)
    d = B.dispatch(gpos, gnm, noise, light, cam_pos, nos)
    path = os.path.join(HERE, "golden_dispatch2.npz")
    np.savez_compressed(path, position=gpos, normal_mat=gnm, noise=noise, light_pos=light, camera_pos=cam_pos, noise_offset_scale=nos,
                        b_org=d["org"], b_shadow_dir=d["shadow_dir"], b_reflect_dir=d["reflect_dir"], b_ambient_dir=d["ambient_dir"], renderer=np.array(B.renderer))
    surface = gnm[..., 3] != np.float32(1e30)
    print("dispatch2:", int(surface.sum()), "surface pixels of", w * h, "materials", np.unique(gnm[..., 3][surface].view(np.uint32)), "->", os.path.getsize(path) / 1e6, "MB")


def fullsize(names=("f16_640x480", "bunny_1080p", "atrium_1080p", "soup_10m_4k")):
    """BASELINE configs 1-4 AT THEIR STATED SIZES (round 2): ray_trace.frag on llvmpipe over the primary rays of bench.py's workloads,
    kept as one SHA-256 per pixel row of the (t, u, v, idx) records (a full frame of records is 33-133 MB; hit records are bit-exact, so
    a digest loses nothing) plus the number of hits per row and a digest of the rebuilt scene; for the colour of the fused shader, which
    is not bit-exact, the mean RGB of every 16 x 16 pixel block.  tests/test_gpu_api.py compares the library's whole frames with these."""
    import hashlib
    sys.path.insert(0, REPO)
    import bench
    B = OracleB()
    path = os.path.join(HERE, "golden_fullsize.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    for name in names:
        model, camera, w, h, tex = bench.build_workload(name)
        sc = model.scene_arrays()
        digest = hashlib.sha256(b"".join(np.ascontiguousarray(sc[k]).tobytes() for k in ("nodes", "triangles", "positions"))).hexdigest()
        dirs, orgs = camera.primary_rays(w, h)
        t0 = time.time()
        hits, _ = B.trace(sc, dirs.reshape(h, w, 3), orgs.reshape(h, w, 3))
        hits = np.ascontiguousarray(hits.reshape(h, w, 4), dtype=np.float32)
        out[f"{name}_size"] = np.int32([w, h])
        out[f"{name}_scene_sha256"] = np.array(digest)
        out[f"{name}_row_sha256"] = np.stack([np.frombuffer(hashlib.sha256(hits[y].tobytes()).digest(), np.uint8) for y in range(h)])
        out[f"{name}_row_hits"] = (hits[..., 0] < 1e30).sum(axis=1).astype(np.int32)
        # colour is not bit-exact, so no digest: the mean of the fused shader's RGB over every 16 x 16 pixel block instead
        # (float64 sums; each material's image bound to both slots, see fused_uniform)
        white = np.full((1, 1, 3), 255, np.uint8)
        tex0, tex1 = (tex[0] if tex[0] is not None else white), (tex[1] if tex[1] is not None else white)
        colour = fused_uniform(B, sc, dirs.reshape(h, w, 3), orgs.reshape(h, w, 3), model.has_tex_coords(), model.has_normals(), tex0, tex1, hits)
        out[f"{name}_block_mean_rgb"] = block_means(colour[..., :3]).astype(np.float32)
        print(f"fullsize {name}: {w}x{h}, {int((hits[..., 0] < 1e30).sum())} hits, llvmpipe {time.time() - t0:.1f} s")
    out["renderer"] = np.array(B.renderer)
    np.savez_compressed(path, **out)
    print("fullsize ->", os.path.getsize(path) / 1e3, "kB")


def split4k():
    """The split pipeline of src/main.rs:156-203 at config 5's frame size (round 2): the atrium at 3840 x 2160 -- primary hit records
    (ray_trace.frag on llvmpipe), the G-buffer made from them (oracle A's restatement of the rasteriser pass, which the reference
    fills by rasterising), ray_dispatcher.frag on llvmpipe with the 128 x 128 crop of the reference's blue-noise image at a texel-grid
    offset and scale (frame / image size, as main.rs:161-165), then ray_trace.frag three times on (org, shadowDir | reflectDir |
    ambientDir).  Kept: one SHA-256 per 16 pixel rows of each of the nine buffers.  tests/test_gpu_api.py runs the same chain through
    the library."""
    sys.path.insert(0, REPO)
    import bench
    from oracles import OracleA
    A, B = OracleA(), OracleB()
    model, camera, _, _, _ = bench.build_workload("atrium_1080p")
    w, h = 3840, 2160
    sc = model.scene_arrays()
    dirs, orgs = camera.primary_rays(w, h)
    dirs, orgs = dirs.reshape(h, w, 3), orgs.reshape(h, w, 3)
    out = {"size": np.int32([w, h])}
    primary, _ = B.trace(sc, dirs, orgs)
    out["primary"] = group_digests(primary.reshape(h, w, 4))
    pos, nm = A.surface_from_hits(sc, orgs.reshape(-1, 3), dirs.reshape(-1, 3), primary.reshape(-1, 4), 1)
    gpos, gnm = pos.reshape(h, w, 3), nm.reshape(h, w, 4)
    out["position"] = group_digests(gpos); out["normal_mat"] = group_digests(gnm)
    noise = np.load(os.path.join(HERE, "golden_dispatch2.npz"))["noise"]
    light = np.array([4.0, 9.0, -3.0], np.float32)
    cam_pos = np.asarray(camera.position, np.float32)
    nos = np.array([37 / 128, 101 / 128, w / 128, h / 128], np.float32)
    d = B.dispatch(gpos, gnm, noise, light, cam_pos, nos)
    for k in ("org", "shadow_dir", "reflect_dir", "ambient_dir"):
        out[k] = group_digests(d[k])
    for k in ("shadow_dir", "reflect_dir", "ambient_dir"):
        hits, _ = B.trace(sc, d[k], d["org"])
        out["hits_" + k] = group_digests(hits.reshape(h, w, 4))
        out["n_hits_" + k] = np.int64((hits.reshape(-1, 4)[:, 0] < 1e30).sum())
    out["light_pos"] = light; out["camera_pos"] = cam_pos; out["noise_offset_scale"] = nos; out["renderer"] = np.array(B.renderer)
    path = os.path.join(HERE, "golden_split4k.npz")
    np.savez_compressed(path, **out)
    print("split4k:", {k: int(out[k]) for k in out if k.startswith("n_hits")}, "->", os.path.getsize(path) / 1e3, "kB")


def sampler():
    """llvmpipe's RGBA8 LINEAR / REPEAT sampler, probed (round 2): ray_dispatcher.frag with `ambientDir = normalize(...)` replaced by
    `ambientDir = random` in a temporary copy returns texture(blueNoise, fragPos * zw + xy) itself (and reflectDir the coordinate).  Stored: the images, the
    noiseOffsetScale uniforms and llvmpipe's samples * 255 (integral: it filters in 8 bits) for power-of-two and other sizes,
    coordinates far outside [0, 1] and exact ties of the 8-bit weight.  Pins oa_set_texture_filter_8bit (tests/test_oracle_golden.py)."""
    import shutil
    import tempfile
    from oracles import REFERENCE_SHADERS
    root = tempfile.mkdtemp(prefix="rtamd_shaders_")
    shutil.rmtree(root)
    shutil.copytree(REFERENCE_SHADERS, root)
    frag = os.path.join(root, "ray_trace", "ray_dispatcher.frag")
    text = open(frag).read()
    statement = "ambientDir = normalize(normal + random * 2 - 1);"
    assert text.count(statement) == 1
    statement2 = "reflectDir = normalize(reflect(position - cameraPos, normal));"
    assert text.count(statement2) == 1
    open(frag, "w").write(text.replace(statement, "ambientDir = random;").replace(statement2, "reflectDir = vec3(fragPos * noiseOffsetScale.zw + noiseOffsetScale.xy, 0);"))
    B = OracleB(shader_root=root)
    rng = np.random.default_rng(8)
    out = {}
    w, h = 96, 80
    cases = [((8, 8), [0.137, -1.29, 2.7, 3.3]), ((7, 12), [0.137, -1.29, 2.7, 3.3]), ((64, 64), [3.7, -4.2, 9.0, 8.5]), ((17, 64), [-40.3, 77.1, 3.0, 2.2]),
             ((3, 64), [-2.0, -3.35, 0.001, 0.01]), ((1, 4), [0.0, 0.0, 1.0, 1.0]), ((17, 20), [0.5, 0.25, 1.7, 1.3]),
             ((2, 2), [3.5 / 2, 0.25, 1.0 / 2 * 96 / 256, 0.0]), ((2, 2), [-6.5 / 2, 0.25, 1.0 / 2 * 96 / 256, 0.0])]      # the last two: weights at exact ties
    for k, ((th, tw), nos) in enumerate(cases):
        img = rng.integers(0, 256, (th, tw, 3), dtype=np.uint8)
        pos = np.zeros((h, w, 3), np.float32)
        nm = np.zeros((h, w, 4), np.float32)
        nm[..., 0] = 1
        both = B.dispatch(pos, nm, img, np.float32([1, 1, 1]), np.float32([2, 2, 2]), np.float32(nos))
        r = both["ambient_dir"]
        out[f"b_coord{k}"] = np.ascontiguousarray(both["reflect_dir"][..., :2])      # the coordinate as the shader computed it
        k255 = np.rint(r * 255)
        assert np.abs(r * 255 - k255).max() < 1e-3
        out[f"image{k}"] = img; out[f"noise_offset_scale{k}"] = np.float32(nos); out[f"b_sample{k}"] = k255.astype(np.uint8)
        out[f"b_float{k}"] = r
    out["count"] = np.int32(len(cases)); out["size"] = np.int32([w, h]); out["renderer"] = np.array(B.renderer)
    assert B.L.glslref_init(REFERENCE_SHADERS.encode()) == 0
    path = os.path.join(HERE, "golden_sampler.npz")
    np.savez_compressed(path, **out)
    print("sampler:", len(cases), "images x", w * h, "samples ->", os.path.getsize(path) / 1e3, "kB")


def edge():
    """Rays aimed at the edges and vertices of triangles (round 2): the decisions of intersectTriangle (:72, :78, :83) within an ulp
    of the boundary.  On the edge u + v = 1 the reference's compiler (Mesa NIR) evaluates f*du + f*dv as f*(du + dv); the fixture
    holds rays on which the two forms decide differently (counted below with the oracle's switch), so it pins that.  No ray passes
    through the coordinate origin (the scene is shifted): see llvmpipe_origin_artifact.py."""
    A = OracleA()
    B = OracleB()
    rng = np.random.default_rng(83)
    n_tri = 48
    tri_pos = (rng.uniform(-3, 3, (n_tri, 3, 3)) + np.array([7.0, 5.0, 3.0])).astype(np.float32)
    tri = np.concatenate([np.arange(3 * n_tri, dtype=np.uint32).reshape(n_tri, 3), rng.integers(0, 2, (n_tri, 1)).astype(np.uint32)], axis=1)
    m = host.Model.from_arrays(tri, tri_pos.reshape(-1, 3))
    m.build_bvh()
    sc = m.scene_arrays()

    def aimed(n):
        t = tri_pos[rng.integers(0, n_tri, n)].astype(np.float64)
        w = rng.dirichlet([1, 1, 1], n)
        kind = rng.random(n)
        on_hyp = kind < 0.7                                         # p1..p2: u + v = 1
        w[on_hyp, 0] = 0.0
        on_u0 = (kind >= 0.7) & (kind < 0.8)                        # p0..p2: u = 0
        w[on_u0, 1] = 0.0
        on_v0 = (kind >= 0.8) & (kind < 0.9)                        # p0..p1: v = 0
        w[on_v0, 2] = 0.0
        vert = kind >= 0.9
        w[vert] = np.eye(3)[rng.integers(0, 3, int(vert.sum()))]
        w /= w.sum(1, keepdims=True)
        target = (t * w[:, :, None]).sum(1)
        org = target + rng.normal(0, 1, (n, 3)) * rng.uniform(0.5, 6, (n, 1))
        d = target - org
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        return org.astype(np.float32), d.astype(np.float32)

    # many candidates; keep every ray on which the two forms of :83 decide differently, fill up with the rest
    org, d = aimed(300000)
    as_compiled = A.trace(sc, org, d)["hits"]
    A.set_edge_sum_as_written(1)
    as_written = A.trace(sc, org, d)["hits"]
    A.set_edge_sum_as_written(0)
    differ = np.where((bits(as_compiled) != bits(as_written)).any(axis=1))[0]
    hh = ww = 64
    differ = differ[:hh * ww // 2]
    rest = np.setdiff1d(np.arange(len(org)), differ)[:hh * ww - len(differ)]
    pick = np.concatenate([differ, rest])
    rng.shuffle(pick)
    orgs, dirs = org[pick].reshape(hh, ww, 3), d[pick].reshape(hh, ww, 3)
    b_trace, _ = B.trace(sc, dirs, orgs)
    b_fused, _ = B.fused(sc, dirs, orgs, False, False, None, None)
    path = os.path.join(HERE, "golden_edge.npz")
    np.savez_compressed(path, nodes=sc["nodes"], triangles=sc["triangles"], positions=sc["positions"], dir=dirs, org=orgs,
                        b_trace=b_trace, b_fused=b_fused, n_discriminating=np.int32(len(differ)), renderer=np.array(B.renderer))
    got = A.trace(sc, orgs, dirs)["hits"]
    print("edge:", len(differ), "rays of", hh * ww, "on which u + v as written decides differently;",
          "oracle A == llvmpipe:", np.array_equal(bits(got), bits(b_trace.reshape(-1, 4))), "->", os.path.getsize(path) / 1e3, "kB")


def atrium():
    """The headline scene (bench.py's atrium_1080p, BASELINE config 3) at 240x135: the scene is procedural and
    deterministic (raytracer_amd/scenes.py + the host BVH builder), so only the shader outputs are stored; the
    test rebuilds the scene and also checks a digest of its arrays."""
    import hashlib
    from raytracer_amd import scenes
    B = OracleB()
    w, h = 240, 135
    tri, pos, uv, nrm, cam = scenes.atrium()
    m = host.Model.from_arrays(tri, pos, uv, nrm)
    m.build_bvh()
    sc = m.scene_arrays()
    d = np.asarray(cam["direction"], np.float32)
    d /= np.linalg.norm(d)
    camera = host.Camera(16 / 9, fov=math.radians(cam["fov_deg"]), position=cam["position"], direction=d)
    dirs, orgs = camera.primary_rays(w, h)
    tex0, tex1 = scenes.checker_texture(), scenes.noise_texture()
    b_trace, _ = B.trace(sc, dirs, orgs)
    b_fused, _ = B.fused(sc, dirs, orgs, True, True, tex0, tex1)
    digest = hashlib.sha256(b"".join(np.ascontiguousarray(sc[k]).tobytes() for k in ("nodes", "triangles", "positions", "tex_coords", "normals"))).hexdigest()
    path = os.path.join(HERE, "golden_atrium.npz")
    b_uniform = fused_uniform(B, sc, dirs, orgs, True, True, tex0, tex1, b_trace)
    np.savez_compressed(path, width=np.int32(w), height=np.int32(h), b_trace=b_trace, b_fused=b_fused, b_fused_uniform=b_uniform,
                        scene_sha256=np.array(digest), renderer=np.array(B.renderer))
    print(f"atrium: {len(sc['triangles'])} tris, {int((b_trace[..., 0] < 1e30).sum())}/{w * h} hit pixels -> {os.path.getsize(path) / 1e6:.2f} MB")


def soup():
    """BASELINE config 4's kind of scene at fixture size: 100 000-triangle seeded soup (dense enough that most
    rays hit: extent 12, size 0.5), camera frame 200x150 plus 8192 random rays with NO_RAY entries; outputs only."""
    import hashlib
    from raytracer_amd import scenes
    B = OracleB()
    w, h = 200, 150
    tri, pos, cam = scenes.triangle_soup(100_000, seed=3, extent=12.0, size=0.5)
    m = host.Model.from_arrays(tri, pos)
    m.build_bvh()
    sc = m.scene_arrays()
    camera = host.Camera(w / h, fov=math.radians(cam["fov_deg"]), position=cam["position"], direction=cam["direction"])
    dirs, orgs = camera.primary_rays(w, h)
    b_trace, _ = B.trace(sc, dirs, orgs)
    b_fused, _ = B.fused(sc, dirs, orgs, False, False, None, None)
    ro, rd = random_rays(sc["positions"], 8192, seed=77)
    b_rand, _ = B.trace(sc, rd, ro)
    digest = hashlib.sha256(b"".join(np.ascontiguousarray(sc[k]).tobytes() for k in ("nodes", "triangles", "positions"))).hexdigest()
    path = os.path.join(HERE, "golden_soup.npz")
    np.savez_compressed(path, width=np.int32(w), height=np.int32(h), b_trace=b_trace, b_fused=b_fused, rand_org=ro, rand_dir=rd, b_trace_rand=b_rand,
                        scene_sha256=np.array(digest), renderer=np.array(B.renderer))
    print(f"soup: {len(sc['triangles'])} tris, {int((b_trace[..., 0] < 1e30).sum())}/{w * h} hit pixels, "
          f"{int((b_rand[..., 0] < 1e30).sum())}/8192 random hits -> {os.path.getsize(path) / 1e6:.2f} MB")


def raycreate():
    """ray_trace/ray_create.frag on llvmpipe for four cameras: inputs (inverse proj*view, near, far) and the RGB32F
    direction target.  The shader works on the rasteriser's interpolated fragPos, ours on pixel centres computed in
    float: 96-100 % of the components are bit-equal, the rest within 1.5e-5 for the first three cameras and 2.1e-4 for the
    fourth (far from the origin: the unprojection cancels more).  Round 2 also stores llvmpipe's OWN values of the varying
    (b_fragpos: the same draw with the shader's one statement replaced by `dir = vec3(fragPos, 0)`, patched in a temporary
    copy), so that the shader's arithmetic is pinned bit for bit on every pixel: oracle(b_fragpos) == b_dir."""
    import shutil
    import tempfile
    from oracles import REFERENCE_SHADERS
    B = OracleB()
    root = tempfile.mkdtemp(prefix="rtamd_shaders_")
    shutil.rmtree(root)
    shutil.copytree(REFERENCE_SHADERS, root)
    frag = os.path.join(root, "ray_trace", "ray_create.frag")
    text = open(frag).read()
    statement = "dir = normalize((invProjView * (vec4(fragPos, 1, 1) * far - vec4(fragPos, -1, 1) * near)).xyz);"
    assert text.count(statement) == 1
    open(frag, "w").write(text.replace(statement, "dir = vec3(fragPos, 0);"))
    out = {}
    cams = [(240, 135, (0.5, 1.5, -2.0), (0.2, -0.1, 0.97), 75.0), (160, 120, (3.0, 4.0, 5.0), (-0.5, -0.4, -0.6), 120.0), (64, 64, (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 30.0),
            (320, 180, (31.0, -42.0, 25.5), (-0.55, 0.62, -0.56), 95.0)]
    for k, (w, h, pos, d, fov) in enumerate(cams):
        dd = np.asarray(d, np.float32)
        dd /= np.linalg.norm(dd)
        cam = host.Camera(w / h, fov=math.radians(fov), position=pos, direction=dd)
        inv = np.asarray(cam.inv_proj_view(), np.float32).reshape(16)
        assert B.L.glslref_init(REFERENCE_SHADERS.encode()) == 0
        b = np.asarray(B.raycreate(w, h, inv, host.NEAR, host.FAR), np.float32).reshape(h, w, -1)[..., :3]
        assert B.L.glslref_init(root.encode()) == 0
        fp = np.asarray(B.raycreate(w, h, inv, host.NEAR, host.FAR), np.float32).reshape(h, w, -1)[..., :2]
        out[f"size{k}"] = np.int32([w, h]); out[f"inv_proj_view{k}"] = inv; out[f"b_dir{k}"] = np.ascontiguousarray(b)
        out[f"b_fragpos{k}"] = np.ascontiguousarray(fp)
    assert B.L.glslref_init(REFERENCE_SHADERS.encode()) == 0
    out["near_far"] = np.float32([host.NEAR, host.FAR]); out["count"] = np.int32(len(cams)); out["renderer"] = np.array(B.renderer)
    path = os.path.join(HERE, "golden_raycreate.npz")
    np.savez_compressed(path, **out)
    print("raycreate:", len(cams), "cameras ->", os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    if "--add-uniform" in sys.argv:
        add_uniform_to_existing()
    elif "--only-dispatch2" in sys.argv:
        dispatch2()
    elif "--only-more-models" in sys.argv:
        main(MORE_CASES, with_extras=False)
    elif "--only-fullsize" in sys.argv:
        extra = [a for a in sys.argv[1:] if not a.startswith("--")]
        fullsize(tuple(extra)) if extra else fullsize()
    elif "--only-split4k" in sys.argv:
        split4k()
    elif "--only-sampler" in sys.argv:
        sampler()
    elif "--only-edge" in sys.argv:
        edge()
    elif "--only-raycreate" in sys.argv:
        raycreate()
    elif "--only-atrium" in sys.argv:
        atrium()
    elif "--only-soup" in sys.argv:
        soup()
    else:
        main()
        atrium()
        soup()
        raycreate()
