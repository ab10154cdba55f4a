"""Randomised parity: many small scenes built by the host BVH builder from adversarial geometry, traced
with adversarial rays, GPU (both kernel variants) against oracle A, bit for bit.

What the generators aim at: exact ties (duplicated triangles, shared edges and vertices hit exactly),
zero-size boxes (axis-aligned geometry), rays with zero direction components (infinite reciprocals:
the general slab path instead of the octant path), origins on box faces and inside the scene,
degenerate (zero-area) triangles, huge and tiny coordinates, denormal inputs (flushed), NO_RAY
entries, and every work-item raggedness the queue sees (ray counts that are not multiples of 64)."""
import numpy as np
import pytest

from common import bits, rgb_err_255
from raytracer_amd import host

pytestmark = pytest.mark.gpu


def _scene(rng, kind):
    n = int(rng.integers(1, 400))
    if kind == "grid":          # axis-aligned quads on an integer lattice: flat boxes, shared edges, exact ties
        c = rng.integers(-4, 5, (n, 3)).astype(np.float32)
        axis = rng.integers(0, 3, n)
        u = np.eye(3, dtype=np.float32)[(axis + 1) % 3]
        v = np.eye(3, dtype=np.float32)[(axis + 2) % 3]
        tri_pos = np.stack([c, c + u, c + v], axis=1)
    elif kind == "dup":         # few distinct triangles, each repeated: ties resolved by traversal order only
        base = rng.uniform(-3, 3, (max(1, n // 8), 3, 3)).astype(np.float32)
        tri_pos = base[rng.integers(0, len(base), n)]
    elif kind == "degenerate":  # zero-area and needle triangles among ordinary ones
        tri_pos = rng.uniform(-3, 3, (n, 3, 3)).astype(np.float32)
        k = rng.random(n)
        tri_pos[k < 0.2, 2] = tri_pos[k < 0.2, 1]
        tri_pos[(k >= 0.2) & (k < 0.3), 1] = tri_pos[(k >= 0.2) & (k < 0.3), 0]
    elif kind == "scales":      # 1e-30 .. 1e6 coordinates, denormal vertices
        tri_pos = (rng.uniform(-1, 1, (n, 3, 3)) * 10.0 ** rng.uniform(-30, 6, (n, 1, 1))).astype(np.float32)
        tri_pos[rng.random(n) < 0.1] *= np.float32(1e-40 / 1e-30)
    else:                       # plain soup
        tri_pos = rng.uniform(-3, 3, (n, 3, 3)).astype(np.float32)
    pos = tri_pos.reshape(-1, 3)
    tri = np.concatenate([np.arange(3 * n, dtype=np.uint32).reshape(n, 3), rng.integers(0, 2, (n, 1)).astype(np.uint32)], axis=1)
    m = host.Model.from_arrays(tri, pos)
    m.build_bvh()
    return m.scene_arrays(), tri_pos


def _rays(rng, tri_pos, n):
    k = len(tri_pos)
    # aim at exact vertices / edge points / interior points of random triangles
    t = tri_pos[rng.integers(0, k, n)].astype(np.float64)
    w = rng.dirichlet([1, 1, 1], n)
    sel = rng.random(n)
    w[sel < 0.15] = np.eye(3)[rng.integers(0, 3, int((sel < 0.15).sum()))]                 # a vertex
    e = (sel >= 0.15) & (sel < 0.3)
    w[e, 2] = 0.0; w[e, :2] /= np.maximum(w[e, :2].sum(1, keepdims=True), 1e-30)         # on an edge
    target = (t * w[:, :, None]).sum(1)
    org = rng.uniform(-6, 6, (n, 3))
    inside = rng.random(n) < 0.2
    org[inside] = target[inside] + rng.uniform(-0.5, 0.5, (int(inside.sum()), 3))           # origins inside the scene's boxes
    lattice = rng.random(n) < 0.3
    org[lattice] = np.round(org[lattice])                                                   # on lattice planes = on box faces of the grid scenes
    d = target - org
    axis_aligned = rng.random(n) < 0.25                                                     # zero components: infinite reciprocals
    a = rng.integers(0, 3, n)
    for i in np.where(axis_aligned)[0]:
        keep = d[i, a[i]] if d[i, a[i]] != 0 else 1.0
        d[i] = 0.0; d[i, a[i]] = keep
    planar = (~axis_aligned) & (rng.random(n) < 0.15)
    d[planar, rng.integers(0, 3)] = 0.0
    nrm = np.linalg.norm(d, axis=1, keepdims=True)
    d = np.where(nrm > 0, d / np.maximum(nrm, 1e-300), [[0.0, 0.0, 1.0]])
    unnorm = rng.random(n) < 0.1
    d[unnorm] *= rng.uniform(1e-3, 1e3, (int(unnorm.sum()), 1))                            # the shaders do not require unit directions
    org = org.astype(np.float32); d = d.astype(np.float32)
    d[rng.random(n) < 0.02] = 0.0                                                           # NO_RAY (ray_trace.frag:133)
    neg0 = rng.random(n) < 0.05
    d[neg0] = np.where(d[neg0] == 0, np.float32(-0.0), d[neg0])
    return org, d


@pytest.mark.parametrize("kind", ["grid", "dup", "degenerate", "scales", "soup"])
def test_random_scenes_trace_and_fused_match_oracle(tracer, oracle_a, kind):
    # RTAMD_FUZZ_SEED / RTAMD_FUZZ_REPS: a soak run with other seeds and more scenes (the committed run: seed 0, 12 scenes per kind)
    import os
    seed, reps = int(os.environ.get("RTAMD_FUZZ_SEED", "0")), int(os.environ.get("RTAMD_FUZZ_REPS", "12"))
    rng = np.random.default_rng({"grid": 11, "dup": 12, "degenerate": 13, "scales": 14, "soup": 15}[kind] + 1000 * seed)
    white = np.full((1, 1, 3), 255, np.uint8)
    tracer.upload_texture(0, white); tracer.upload_texture(1, white)
    try:
        for rep in range(reps):
            sc, tri_pos = _scene(rng, kind)
            n = int(rng.integers(1, 3000))
            org, d = _rays(rng, tri_pos, n)
            tracer.upload_scene(sc)
            want = oracle_a.trace(sc, org, d)["hits"]
            lit = d.any(axis=1)                                  # the fused shader has no NO_RAY convention: give those a direction
            d2 = d.copy(); d2[~lit] = np.float32([0, 0, 1])
            a = oracle_a.render(sc, d2, org, 0, 0)
            for variant in (0, 1):
                tracer.set_kernel_variant(variant)
                got = tracer.trace(org, d)
                assert np.array_equal(bits(got), bits(want)), (kind, rep, variant, int((bits(got) != bits(want)).any(axis=1).sum()))
                r = tracer.render(n, 1, dir=d2, org=org, has_tex_coords=0, has_normals=0)
                assert np.array_equal(a["hit_id"], r["hit_id"]) and np.array_equal(bits(a["hit_t"]), bits(r["hit_t"])), (kind, rep, variant)
                ok = np.isfinite(a["rgba"]).all(axis=1) & np.isfinite(r["rgba"]).all(axis=1)
                assert np.array_equal(np.isfinite(a["rgba"]), np.isfinite(r["rgba"]))
                assert rgb_err_255(a["rgba"][ok], r["rgba"][ok]).max(initial=0.0) <= 0.05, (kind, rep, variant)
    finally:
        tracer.set_kernel_variant(0)


def test_random_textured_scenes_match_oracle(tracer, oracle_a):
    """The shading inputs the other kinds leave out: per-vertex texture coordinates far outside [0, 1] (REPEAT wrap of negative and
    large coordinates, texel centres hit exactly), interpolated normals that are not unit length or point away from the viewer,
    two materials, textures of odd sizes down to 1 x 1 and 1 x n.  Hit buffers bit for bit, colour within 0.05/255 of oracle A
    (the same float bilinear on both sides), both kernel variants."""
    import os
    seed, reps = int(os.environ.get("RTAMD_FUZZ_SEED", "0")), int(os.environ.get("RTAMD_FUZZ_REPS", "12"))
    rng = np.random.default_rng(16 + 1000 * seed)
    try:
        for rep in range(reps):
            n = int(rng.integers(1, 300))
            tri_pos = rng.uniform(-3, 3, (n, 3, 3)).astype(np.float32)
            pos = tri_pos.reshape(-1, 3)
            tri = np.concatenate([np.arange(3 * n, dtype=np.uint32).reshape(n, 3), rng.integers(0, 2, (n, 1)).astype(np.uint32)], axis=1)
            uv = rng.uniform(-4, 5, (3 * n, 2)).astype(np.float32)
            snap = rng.random(3 * n) < 0.3
            uv[snap] = np.round(uv[snap] * 8) / 8                              # texel centres / edges of power-of-two textures, exactly
            nrm = rng.normal(size=(3 * n, 3)).astype(np.float32)
            nrm[rng.random(3 * n) < 0.5] *= np.float32(0.01)                    # far from unit length
            m = host.Model.from_arrays(tri, pos, uv, nrm)
            m.build_bvh()
            sc = m.scene_arrays()
            tex = []
            for _ in range(2):
                w, h = [int(x) for x in rng.choice([1, 2, 3, 5, 8, 17, 64], 2)]
                tex.append(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
            nr = int(rng.integers(1, 3000))
            org, d = _rays(rng, tri_pos, nr)
            d[~d.any(axis=1)] = np.float32([0, 0, 1])
            tracer.upload_scene(sc)
            tracer.upload_texture(0, tex[0]); tracer.upload_texture(1, tex[1])
            for filter8 in (0, 1):      # float bilinear, and the reference executor's 8-bit filter (option "texture_filter_8bit")
                oracle_a.set_texture_filter_8bit(filter8)
                tracer.set_option("texture_filter_8bit", filter8)
                a = oracle_a.render(sc, d, org, 1, 1, tex[0], tex[1])
                for variant in (0, 1):
                    tracer.set_kernel_variant(variant)
                    r = tracer.render(nr, 1, dir=d, org=org, has_tex_coords=1, has_normals=1)
                    assert np.array_equal(a["hit_id"], r["hit_id"]) and np.array_equal(bits(a["hit_t"]), bits(r["hit_t"])), (rep, variant)
                    assert np.array_equal(np.isfinite(a["rgba"]), np.isfinite(r["rgba"])), (rep, variant)
                    ok = np.isfinite(a["rgba"]).all(axis=1)
                    assert rgb_err_255(a["rgba"][ok], r["rgba"][ok]).max(initial=0.0) <= 0.05, (rep, variant, filter8)
    finally:
        tracer.set_kernel_variant(0)
        tracer.set_option("texture_filter_8bit", 0)
        oracle_a.set_texture_filter_8bit(0)


def test_random_gbuffers_dispatch_match_oracle(tracer, oracle_a):
    """ray_dispatcher.frag's restatements against each other on random G-buffers: frame sizes around the kernel's 256-pixel blocks,
    noise textures of odd sizes, offsets far outside [0, 1], normals of any length, surfaces at the light or the camera (zero-length
    vectors to normalise), "no surface" pixels anywhere: the four ray arrays bit for bit wherever both are finite, NaN / inf in the same places."""
    import os
    seed, reps = int(os.environ.get("RTAMD_FUZZ_SEED", "0")), int(os.environ.get("RTAMD_FUZZ_REPS", "12"))
    rng = np.random.default_rng(17 + 1000 * seed)
    for rep in range(reps):
        w, h = int(rng.integers(1, 300)), int(rng.integers(1, 40))
        nw, nh = [int(x) for x in rng.choice([1, 2, 3, 7, 16, 64], 2)]
        noise = rng.integers(0, 256, (nh, nw, 3), dtype=np.uint8)
        pos = (rng.uniform(-1, 1, (h, w, 3)) * 10.0 ** rng.uniform(-3, 4, (h, w, 1))).astype(np.float32)
        nm = np.zeros((h, w, 4), np.float32)
        nm[..., :3] = rng.normal(size=(h, w, 3)) * 10.0 ** rng.uniform(-2, 1, (h, w, 1))
        nm[..., 3] = np.where(rng.random((h, w)) < 0.2, np.float32(1e30), rng.integers(0, 2, (h, w)).astype(np.uint32).view(np.float32))
        light = rng.uniform(-20, 20, 3).astype(np.float32)
        cam = rng.uniform(-20, 20, 3).astype(np.float32)
        pos[rng.random((h, w)) < 0.02] = light                                  # lightPos - position = 0
        pos[rng.random((h, w)) < 0.02] = cam                                    # position - cameraPos = 0
        nos = np.float32([rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(0.1, 9), rng.uniform(0.1, 9)])
        tracer.upload_noise_texture(noise)
        for filter8 in (0, 1):          # float bilinear, and the reference executor's 8-bit filter (option "texture_filter_8bit")
            tracer.set_option("texture_filter_8bit", filter8)
            oracle_a.set_texture_filter_8bit(filter8)
            try:
                r = tracer.dispatch_rays(pos, nm, light, cam, nos)
                a = oracle_a.dispatch(pos, nm, noise, light, cam, nos)
            finally:
                tracer.set_option("texture_filter_8bit", 0)
                oracle_a.set_texture_filter_8bit(0)
            for k in ("org", "shadow_dir", "reflect_dir", "ambient_dir"):
                fa, fr = np.isfinite(a[k]), np.isfinite(r[k])
                assert np.array_equal(fa, fr), (rep, k, filter8)
                assert np.array_equal(bits(a[k])[fa], bits(r[k])[fa]), (rep, k, w, h, filter8)


def test_odd_boxes_match_oracle(tracer, oracle_a):
    """intersectAABB (ray_trace_and_shade.frag:53-61) is defined for any six floats, and the C-ABI takes any caller's nodes:
    boxes with min > max on an axis, NaN bounds and infinite bounds, which the reference's builder never emits.  The
    direction-octant slab test equals the general one only for finite boxes with min <= max, so such a scene is routed
    through the general test (rtamd_scene_info.odd_box_count > 0): trace and fused hit buffers equal oracle A bit for bit,
    both kernel variants.  Rays are general-position ones, i.e. the ones that WOULD take the octant path."""
    import os
    seed, reps = int(os.environ.get("RTAMD_FUZZ_SEED", "0")), int(os.environ.get("RTAMD_FUZZ_REPS", "12"))
    rng = np.random.default_rng(18 + 1000 * seed)
    try:
        for rep in range(reps):
            sc, tri_pos = _scene(rng, "soup")
            tracer.upload_scene(sc)
            assert tracer.scene_info().odd_box_count == 0
            nodes = sc["nodes"].copy()
            cand = np.array([i for i in range(len(nodes)) if i != 1])          # node 1 is the builder's unreachable dummy
            picks = rng.choice(cand, size=max(1, len(cand) // 6), replace=False)
            for i in picks:
                b = nodes["bounds"][i]
                kind, ax = rng.integers(0, 4), rng.integers(0, 3)
                if kind == 0:
                    b[ax], b[3 + ax] = b[3 + ax], b[ax]                        # inverted on one axis
                    if b[ax] == b[3 + ax]:
                        b[ax] += np.float32(1.0)
                elif kind == 1:
                    b[rng.integers(0, 6)] = np.float32(np.nan)
                elif kind == 2:
                    b[ax], b[3 + ax] = np.float32(-np.inf), np.float32(np.inf)  # an infinite slab
                else:
                    b[3 + ax] = np.float32(-np.inf) if rng.random() < 0.5 else np.float32(np.inf)
                nodes["bounds"][i] = b
            sc2 = dict(sc, nodes=nodes)
            n = int(rng.integers(64, 3000))
            org, d = _rays(rng, tri_pos, n)
            d[~d.any(axis=1)] = np.float32([0.3, -0.5, 0.8])
            tracer.upload_scene(sc2)
            assert tracer.scene_info().odd_box_count == len(picks)
            want = oracle_a.trace(sc2, org, d)["hits"]
            a = oracle_a.render(sc2, d, org, 0, 0)
            for variant in (0, 1):
                tracer.set_kernel_variant(variant)
                got = tracer.trace(org, d)
                assert np.array_equal(bits(got), bits(want)), (rep, variant, int((bits(got) != bits(want)).any(axis=1).sum()))
                r = tracer.render(n, 1, dir=d, org=org, has_tex_coords=0, has_normals=0)
                assert np.array_equal(a["hit_id"], r["hit_id"]) and np.array_equal(bits(a["hit_t"]), bits(r["hit_t"])), (rep, variant)
    finally:
        tracer.set_kernel_variant(0)
