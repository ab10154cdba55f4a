#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference and oracle/_ref): oracle A against the reference itself beyond the committed
fixtures.  The adversarial scene / ray generators of tests/test_gpu_fuzz.py (exact ties, flat boxes, duplicated and degenerate
triangles, 1e-30 .. 1e6 coordinates, denormals, zero direction components, origins on box faces, NO_RAY entries) feed the
UNMODIFIED ray_trace.frag and ray_trace_and_shade.frag on Mesa llvmpipe and oracle/oracle_a.c: hit records must agree bit for
bit, colours within 0.01/255 (textured scenes with llvmpipe's 8-bit texture filter restated in the oracle).

One class of colour differences is the executor's, not the shader's, and is counted apart ("llvmpipe and
rays through the coordinate origin"; tests/golden/llvmpipe_origin_artifact.py shows it without any oracle): when a primary ray
passes EXACTLY through (0, 0, 0) -- its slab distances to that point are equal on all three axes -- the fused shader's second
traversal on llvmpipe ends after its first node, for that pixel and for pixels that share its SIMD group, so a shadowed pixel
comes out lit.  A pixel whose colour differs is therefore rendered once more alone in a 1x1 draw: it must then agree, or its
own ray must pass through the origin.  Anything else fails the soak.
Usage: python tests/golden/soak_oracle_vs_llvmpipe.py [scenes per kind] [seed]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from common import bits, rgb_err_255  # noqa: E402
from oracles import OracleA, OracleB  # noqa: E402
from test_gpu_fuzz import _rays, _scene  # noqa: E402



def through_origin(org, d):
    """The slab test of ray_trace_and_shade.frag:53-61 against a box of zeros with t = 1000000, in float32; rays with a zero
    direction component (0 * inf) count as through the origin when the other axes allow it."""
    with np.errstate(all="ignore"):
        tiny = np.float32(1.1754944e-38)                    # llvmpipe runs FTZ / DAZ: denormal components are zeros
        org = np.where(np.abs(org) < tiny, np.copysign(np.float32(0), org), org)
        d = np.where(np.abs(d) < tiny, np.copysign(np.float32(0), d), d)
        tx = ((np.float32(0) - org) * (np.float32(1) / d)).astype(np.float32)
        tmin, tmax = np.nanmax(tx, axis=1), np.nanmin(tx, axis=1)
        return np.isnan(tx).all(axis=1) | ((tmax >= tmin) & (tmax > 0))


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
A, B = OracleA(), OracleB()
A.set_flush_denormals(1)
print("reference:", B.renderer)
bad = n_origin = n_group = 0
worst_unshadowed = 0.0      # over pixels the oracle does not put in shadow: the origin artefact cannot hide below the tolerance there
def check_colour(kind, rep, sc, org, d2, a, fused, args, tol=1.0):
    """Shared by the plain and the textured part: colour within tol/255 (a number or one per pixel), the origin artefact counted apart."""
    global bad, n_origin, n_group
    fused = fused.reshape(-1, 4)
    tol = np.broadcast_to(np.asarray(tol, np.float64), (len(fused),))
    ok = np.isfinite(fused).all(axis=1) & np.isfinite(a["rgba"]).all(axis=1)
    err = np.where(ok, rgb_err_255(a["rgba"], np.where(ok[:, None], fused, a["rgba"])), 0.0)
    origin = through_origin(org, d2)
    for i in np.where(err > tol)[0]:
        alone, _ = B.fused(sc, d2[i].reshape(1, 1, 3), org[i].reshape(1, 1, 3), *args)
        e1 = rgb_err_255(a["rgba"][i:i + 1], alone.reshape(1, 4))[0]
        if e1 <= tol[i] and a["shadow"][i] == 1:
            n_group += 1
            err[i] = e1
        elif origin[i] and a["shadow"][i] == 1:
            n_origin += 1
            err[i] = 0.0
    if (err > tol).any() or not np.array_equal(np.isfinite(fused).all(axis=1), np.isfinite(a["rgba"]).all(axis=1)):
        bad += 1
        k = int((err - tol).argmax())
        print(f"MISMATCH colour {kind} rep {rep}: {err[k]:.3f}/255 (allowed {tol[k]:.3f}) at ray {k}: oracle {a['rgba'][k]} reference {fused[k]} org {org[k]} dir {d2[k]}")
    global worst_unshadowed
    worst_unshadowed = max(worst_unshadowed, float(err[a["shadow"] != 1].max(initial=0.0)))
    return float(err.max(initial=0.0))


for kind in ("grid", "dup", "degenerate", "scales", "soup"):
    rng = np.random.default_rng({"grid": 11, "dup": 12, "degenerate": 13, "scales": 14, "soup": 15}[kind] + 1000 * seed)
    n_rays = n_hit = 0
    worst = 0.0
    for rep in range(reps):
        sc, tri_pos = _scene(rng, kind)
        n = int(rng.integers(1, 3000))
        org, d = _rays(rng, tri_pos, n)
        o3, d3 = org.reshape(n, 1, 3), d.reshape(n, 1, 3)
        ref, _ = B.trace(sc, d3, o3)
        got = A.trace(sc, org, d)["hits"]
        ref = ref.reshape(-1, 4)
        if not np.array_equal(bits(got), bits(ref)):
            bad += 1
            k = np.where((bits(got) != bits(ref)).any(axis=1))[0]
            print(f"MISMATCH trace {kind} rep {rep}: {len(k)} of {n} rays, first {k[0]}: oracle {got[k[0]]} reference {ref[k[0]]} org {org[k[0]]} dir {d[k[0]]}")
        d2 = d.copy()
        d2[~d.any(axis=1)] = np.float32([0, 0, 1])
        fused, _ = B.fused(sc, d2.reshape(n, 1, 3), o3, False, False, None, None)
        a = A.render(sc, d2, org, 0, 0)
        fused = fused.reshape(-1, 4)
        worst = max(worst, check_colour(kind, rep, sc, org, d2, a, fused, (False, False, None, None), 0.01))
        n_rays += n
        n_hit += int((ref[:, 0] < 1e30).sum())
    print(f"{kind}: {reps} scenes, {n_rays} rays, {n_hit} hits, worst colour difference {worst:.4f}/255")


# textured scenes with interpolated normals (tests/test_gpu_fuzz.py:test_random_textured_scenes_match_oracle).  Texture rows are
# multiples of 4 bytes: the reference uploads tightly packed RGB rows with GL's default UNPACK_ALIGNMENT of 4
# (src/gl_wrapper/texture.rs:11-26), so other widths reach GL skewed and read past the image -- nothing to pin there.
from raytracer_amd import host  # noqa: E402
rng = np.random.default_rng(16 + 1000 * seed)
worst = 0.0
n_rays = 0
for rep in range(reps):
    n = int(rng.integers(1, 300))
    tri_pos = rng.uniform(-3, 3, (n, 3, 3)).astype(np.float32)
    tri = np.concatenate([np.arange(3 * n, dtype=np.uint32).reshape(n, 3), rng.integers(0, 2, (n, 1)).astype(np.uint32)], axis=1)
    uv = rng.uniform(-4, 5, (3 * n, 2)).astype(np.float32)
    snap = rng.random(3 * n) < 0.3
    uv[snap] = np.round(uv[snap] * 8) / 8
    nrm = rng.normal(size=(3 * n, 3)).astype(np.float32)
    nrm[rng.random(3 * n) < 0.5] *= np.float32(0.01)
    m = host.Model.from_arrays(tri, tri_pos.reshape(-1, 3), uv, nrm)
    m.build_bvh()
    sc = m.scene_arrays()
    # llvmpipe filters RGBA8 in 8-bit fixed point and returns 8-bit values: up to 2/255 from float bilinear between texels of
    # full contrast, times up to 1.7 of light.  The oracle therefore runs with llvmpipe's filter restated
    # (oa_set_texture_filter_8bit: 100 % of 1.8 M probed samples equal), which leaves 0.01/255 for everything else.
    tw, th = int(rng.choice([16, 32, 64])), int(rng.choice([1, 16, 64]))
    yy, xx = np.meshgrid(np.arange(th), np.arange(tw), indexing="ij")
    smooth = np.stack([128 + 90 * np.sin(2 * np.pi * (xx / tw + k / 3)) * np.cos(2 * np.pi * yy / th) for k in range(3)], axis=-1).astype(np.uint8)
    noise = rng.integers(0, 256, (int(rng.choice([1, 2, 3, 5, 8, 17, 64])), int(rng.choice([4, 8, 12, 16, 64])), 3), dtype=np.uint8)
    nr = int(rng.integers(1, 3000))
    org, d = _rays(rng, tri_pos, nr)
    d[~d.any(axis=1)] = np.float32([0, 0, 1])
    # one material image in both slots: llvmpipe's sampler index is per 4x4 block ("Colour is pinned")
    A.set_texture_filter_8bit(1)
    has_tc, has_n = [(True, True), (True, False), (False, True)][rep % 3]       # hasTexCoords / hasNormals apart as well
    for t in (smooth, noise):
        args = (has_tc, has_n, t, t)
        fused, _ = B.fused(sc, d.reshape(nr, 1, 3), org.reshape(nr, 1, 3), *args)
        a = A.render(sc, d, org, int(has_tc), int(has_n), t, t)
        worst = max(worst, check_colour("textured", rep, sc, org, d, a, fused, args, 0.01))
    A.set_texture_filter_8bit(0)
    n_rays += 2 * nr
print(f"textured + interpolated normals: {reps} scenes, {n_rays} rays, worst colour difference {worst:.4f}/255")

# ray_dispatcher.frag on random G-buffers (tests/test_gpu_fuzz.py:test_random_gbuffers_dispatch_match_oracle), the noise texture
# sampled at texel centres (as src/main.rs:161-165 sets it up) in every other case and anywhere between texels in the rest, the
# oracle with llvmpipe's filter restated and on llvmpipe's own values of the varying fragPos (read back from a second draw: between
# texels an ulp of the coordinate can move an 8-bit weight): all four targets bit for bit, NaN and inf in the same places.
import shutil  # noqa: E402
import tempfile  # noqa: E402
from oracles import REFERENCE_SHADERS  # noqa: E402
droot = tempfile.mkdtemp(prefix="rtamd_shaders_")          # a copy whose dispatcher writes the varying itself: reflectDir = fragPos
shutil.rmtree(droot)
shutil.copytree(REFERENCE_SHADERS, droot)
frag = os.path.join(droot, "ray_trace", "ray_dispatcher.frag")
text = open(frag).read()
statement = "reflectDir = normalize(reflect(position - cameraPos, normal));"
assert text.count(statement) == 1
open(frag, "w").write(text.replace(statement, "reflectDir = vec3(fragPos, 0);"))
rng = np.random.default_rng(17 + 1000 * seed)
n_px = 0
worst_ambient = 0.0
for rep in range(reps):
    w, h = int(rng.integers(1, 300)), int(rng.integers(1, 40))
    nw, nh = int(rng.choice([4, 8, 12, 16, 20, 64])), int(rng.choice([1, 2, 3, 7, 16, 64]))
    noise = rng.integers(0, 256, (nh, nw, 3), dtype=np.uint8)
    pos = (rng.uniform(-1, 1, (h, w, 3)) * 10.0 ** rng.uniform(-3, 4, (h, w, 1))).astype(np.float32)
    nm = np.zeros((h, w, 4), np.float32)
    nm[..., :3] = rng.normal(size=(h, w, 3)) * 10.0 ** rng.uniform(-2, 1, (h, w, 1))
    nm[..., 3] = np.where(rng.random((h, w)) < 0.2, np.float32(1e30), rng.integers(0, 2, (h, w)).astype(np.uint32).view(np.float32))
    light = rng.uniform(-20, 20, 3).astype(np.float32)
    cam = rng.uniform(-20, 20, 3).astype(np.float32)
    pos[rng.random((h, w)) < 0.02] = light
    pos[rng.random((h, w)) < 0.02] = cam
    if rep % 2:     # texel centres, as src/main.rs:161-165 sets the uniform up
        nos = np.float32([np.float32(rng.integers(0, nw)) / np.float32(nw), np.float32(rng.integers(0, nh)) / np.float32(nh),
                          np.float32(w) / np.float32(nw), np.float32(h) / np.float32(nh)])
    else:           # anywhere between texels
        nos = np.float32([rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(0.1, 9), rng.uniform(0.1, 9)])
    b = B.dispatch(pos, nm, noise, light, cam, nos)
    assert B.L.glslref_init(droot.encode()) == 0
    frag_pos = B.dispatch(pos, nm, noise, light, cam, nos)["reflect_dir"][..., :2]     # llvmpipe's own values of the varying
    assert B.L.glslref_init(REFERENCE_SHADERS.encode()) == 0
    A.set_texture_filter_8bit(1)
    a = A.dispatch(pos, nm, noise, light, cam, nos, frag_pos=frag_pos)
    A.set_texture_filter_8bit(0)
    n_px += w * h
    for k in ("org", "shadow_dir", "reflect_dir", "ambient_dir"):
        fa, fb = np.isfinite(a[k]), np.isfinite(b[k])
        if k == "ambient_dir":
            worst_ambient = max(worst_ambient, float(np.abs(np.where(fa & fb, a[k], 0) - np.where(fa & fb, b[k], 0)).max(initial=0.0)))
        wrong = not np.array_equal(fa, fb) or not np.array_equal(bits(a[k])[fa], bits(b[k])[fa])
        if wrong:
            bad += 1
            print(f"MISMATCH dispatcher rep {rep}: {k}")
print(f"dispatcher: {reps} G-buffers, {n_px} pixels, worst ambientDir difference {worst_ambient:.2e}")

# ray_create.frag with random cameras and frame sizes.  The varying fragPos is the rasteriser's (llvmpipe evaluates a plane
# equation; most values differ from the correctly rounded pixel centre by some 1e-7), so llvmpipe's own values are read back
# from the same draw with a patched copy of the shader: on them the restatement must give llvmpipe's directions bit for bit.
import math  # noqa: E402
import shutil  # noqa: E402
import tempfile  # noqa: E402
from oracles import REFERENCE_SHADERS  # noqa: E402
root = tempfile.mkdtemp(prefix="rtamd_shaders_")
shutil.rmtree(root)
shutil.copytree(REFERENCE_SHADERS, root)
frag = os.path.join(root, "ray_trace", "ray_create.frag")
text = open(frag).read()
statement = "dir = normalize((invProjView * (vec4(fragPos, 1, 1) * far - vec4(fragPos, -1, 1) * near)).xyz);"
assert text.count(statement) == 1
open(frag, "w").write(text.replace(statement, "dir = vec3(fragPos, 0);"))
rng = np.random.default_rng(18 + 1000 * seed)
n_px = n_off = 0
worst_centre = 0.0
for rep in range(reps):
    w, h = (int(rng.integers(1, 400)), int(rng.integers(1, 300))) if rep % 4 else [(640, 480), (1920, 1080), (240, 135)][(rep // 4) % 3]
    direction = rng.normal(size=3)
    direction /= np.linalg.norm(direction)
    if abs(direction[1]) > 0.99:
        direction = np.array([0.6, 0.0, 0.8])
    cam = host.Camera(w / h, fov=math.radians(rng.uniform(5, 170)), position=rng.uniform(-50, 50, 3).astype(np.float32), direction=direction.astype(np.float32))
    inv = np.asarray(cam.inv_proj_view(), np.float32).reshape(16)
    assert B.L.glslref_init(REFERENCE_SHADERS.encode()) == 0
    b = np.asarray(B.raycreate(w, h, inv, host.NEAR, host.FAR), np.float32).reshape(h, w, -1)[..., :3]
    assert B.L.glslref_init(root.encode()) == 0
    fp = np.asarray(B.raycreate(w, h, inv, host.NEAR, host.FAR), np.float32).reshape(h, w, -1)[..., :2]
    a = A.raycreate_points(fp, inv, host.NEAR, host.FAR)
    centre = A.raycreate(w, h, inv, host.NEAR, host.FAR)
    n_px += w * h
    n_off += int((bits(centre) != bits(b)).any(axis=-1).sum())
    worst_centre = max(worst_centre, float(np.abs(centre - b).max()))
    if not np.array_equal(bits(a), bits(b)):
        bad += 1
        k = np.argwhere((bits(a) != bits(b)).any(axis=-1))
        print(f"MISMATCH ray_create rep {rep}: {len(k)} of {w * h} pixels, first {k[0]}: oracle {a[tuple(k[0])]} reference {b[tuple(k[0])]}")
assert B.L.glslref_init(REFERENCE_SHADERS.encode()) == 0
print(f"(from correctly rounded pixel centres instead of llvmpipe's fragPos: {n_off} pixels differ, by at most {worst_centre:.2e})")
print(f"ray_create: {reps} cameras, {n_px} pixels")

print(f"worst colour difference over all pixels that are not in shadow (sky, lit surfaces; every part above): {worst_unshadowed:.4f}/255")
print(f"llvmpipe origin artefact (not counted): {n_origin} pixels whose own ray passes through (0,0,0), {n_group} pixels that agree when drawn alone")
print("FAILED" if bad else "ok", bad)
sys.exit(1 if bad else 0)
