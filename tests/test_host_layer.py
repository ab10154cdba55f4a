"""Host scene layer (C++, raytracer_amd/csrc/host) — OBJ parsing quirks, vertex
unification, BVH builder vs the independent numpy restatement and structural
invariants, camera / ray generation.  CPU only."""
import math

import numpy as np
import pytest

from common import GOLDEN_SCENES, load_golden
from raytracer_amd import host, scenes

CUBE_OBJ = "\n".join(
    ["v -1.0 -1.0 1.0", "v 1.0 -1.0 1.0", "v 1.0 1.0 1.0", "v -1.0 1.0 1.0",
     "v -1.0 -1.0 -1.0", "v 1.0 -1.0 -1.0", "v 1.0 1.0 -1.0", "v -1.0 1.0 -1.0",
     "f 1 2 3", "f 3 4 1", "f 2 6 7", "f 7 3 2", "f 8 7 6", "f 6 5 8",
     "f 5 1 4", "f 4 8 5", "f 5 6 2", "f 2 1 5", "f 4 3 7", "f 7 8 4"]) + "\n"


def check_invariants(nodes, tris, pos):
    """SURVEY §8c pins: root 0, dummy 1, adjacent sibling pairs, first child's subtree
    numbered before the second's, leaf ranges partition [0, T), boxes tight and nested."""
    n_t = len(tris)
    assert nodes["is_leaf"][1] == 1 and nodes["a"][1] == 0 and nodes["b"][1] == 0 and not nodes["bounds"][1].any()
    covered = np.zeros(n_t, np.int32)
    seen = np.zeros(len(nodes), bool)
    stack = [0]
    while stack:
        i = stack.pop()
        assert not seen[i]
        seen[i] = True
        nd = nodes[i]
        if nd["is_leaf"]:
            a, b = int(nd["a"]), int(nd["b"])
            assert b >= 1 and a + b <= n_t
            covered[a:a + b] += 1
            p = pos[np.stack([tris["p0"][a:a + b], tris["p1"][a:a + b], tris["p2"][a:a + b]], 1).reshape(-1)]
            assert np.array_equal(p.min(0), nd["bounds"][:3]) and np.array_equal(p.max(0), nd["bounds"][3:])
        else:
            a, b = int(nd["a"]), int(nd["b"])
            assert b == a + 1 and a > i and a >= 2
            for c in (a, b):
                assert np.all(nodes["bounds"][c][:3] >= nd["bounds"][:3]) and np.all(nodes["bounds"][c][3:] <= nd["bounds"][3:])
            stack += [b, a]
    assert np.all(covered == 1)
    assert seen.sum() == len(nodes) - 1 and not seen[1]


def test_parse_cube_and_bvh_matches_numpy_restatement():
    from oracle.bvh_oracle import build_bvh
    m = host.ResourceParser.parse_model(CUBE_OBJ)
    info = m.info()
    assert (info.triangle_count, info.vertex_count, info.has_tex_coords, info.has_normals) == (12, 8, 0, 0)
    tri0, pos = m.triangles(), m.positions()
    assert tri0["p0"][0] == 0 and tri0["p1"][0] == 1 and tri0["p2"][0] == 2 and np.all(tri0["mat_idx"] == 0)
    m.build_bvh()
    nodes, tris = m.bvh_nodes(), m.triangles()
    check_invariants(nodes, tris, pos)
    want_nodes, want_tris = build_bvh(np.stack([tri0["p0"], tri0["p1"], tri0["p2"], tri0["mat_idx"]], 1), pos)
    assert nodes.tobytes() == want_nodes.tobytes()
    assert np.array_equal(np.stack([tris["p0"], tris["p1"], tris["p2"], tris["mat_idx"]], 1), want_tris)


@pytest.mark.parametrize("kind", ["soup", "sphere", "degenerate"])
def test_bvh_bit_identical_to_numpy_restatement(kind):
    from oracle.bvh_oracle import build_bvh
    rng = np.random.default_rng(5)
    if kind == "soup":
        tri, pos, _ = scenes.triangle_soup(300, seed=3, extent=10.0, size=1.5)
    elif kind == "sphere":
        m = scenes._Mesh()
        scenes._sphere(m, (0.3, -0.2, 0.1), 2.0, 16, 12, 0)
        tri, pos, _, _ = m.arrays()
    else:  # duplicates, coplanar and zero-area triangles: exercises the min == max axis skip and one-sided splits
        pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [2, 2, 0]], np.float32)
        tri = np.array([[0, 1, 2, 0]] * 5 + [[1, 3, 2, 0]] * 3 + [[4, 4, 4, 1]] * 2 + [[0, 3, 4, 1]], np.uint32)
    m = host.Model.from_arrays(tri, pos)
    m.build_bvh()
    nodes, tris = m.bvh_nodes(), m.triangles()
    check_invariants(nodes, tris, m.positions())
    want_nodes, want_tris = build_bvh(tri, pos)
    assert nodes.tobytes() == want_nodes.tobytes()
    assert np.array_equal(np.stack([tris["p0"], tris["p1"], tris["p2"], tris["mat_idx"]], 1), want_tris)
    del rng


@pytest.mark.parametrize("name", GOLDEN_SCENES)
def test_golden_scene_trees_satisfy_invariants(name):
    g = load_golden(name)
    check_invariants(g["nodes"], g["triangles"], g["positions"])


def test_bvh_build_is_deterministic_and_rebuild_is_a_fixed_point():
    tri, pos, _ = scenes.triangle_soup(2000, seed=9, extent=20.0)
    a = host.Model.from_arrays(tri, pos); a.build_bvh()
    b = host.Model.from_arrays(tri, pos); b.build_bvh()
    assert a.bvh_nodes().tobytes() == b.bvh_nodes().tobytes() and a.triangles().tobytes() == b.triangles().tobytes()


@pytest.mark.parametrize("case", ["atrium", "soup200k", "signed_zero_grid", "tiny", "clustered"])
def test_parallel_bvh_build_is_bit_identical_to_the_serial_build(case):
    """Row f3: BVHBuilder::build_parallel must reproduce build() bit for bit — nodes (boxes, numbering,
    leaf ranges) and the triangle permutation — for any thread count, including inputs where the
    order of min/max reductions could show (signed zeros, duplicates) and unbalanced trees."""
    rng = np.random.default_rng(11)
    uv = nrm = None
    if case == "atrium":
        tri, pos, uv, nrm, _ = scenes.atrium()
    elif case == "soup200k":
        tri, pos, _ = scenes.triangle_soup(200_000, seed=4, extent=50.0)
    elif case == "signed_zero_grid":
        # 300 x 300 quads on the z = +-0 plane with x / y coordinates that hit +0 and -0: every box has zero bounds
        k = 300
        xs = (np.arange(k + 1, dtype=np.float32) - k // 2)
        gx, gy = np.meshgrid(xs, xs)
        pos = np.stack([gx, gy, np.zeros_like(gx)], -1).reshape(-1, 3).astype(np.float32)
        pos[::3, 2] = -0.0
        pos[pos[:, 0] == 0, 0] = np.where(np.arange((pos[:, 0] == 0).sum()) % 2 == 0, np.float32(0.0), np.float32(-0.0))
        idx = np.arange(k * (k + 1)).reshape(k, k + 1)[:, :k].reshape(-1)
        tri = np.concatenate([np.stack([idx, idx + 1, idx + k + 1, np.zeros_like(idx)], 1),
                              np.stack([idx + 1, idx + k + 2, idx + k + 1, np.ones_like(idx)], 1)]).astype(np.uint32)
    elif case == "tiny":
        tri, pos, _ = scenes.triangle_soup(7, seed=1, extent=2.0)
    else:  # 95 % of the triangles in a tiny cluster: deep, unbalanced tree; plus exact duplicates
        tri, pos, _ = scenes.triangle_soup(60_000, seed=8, extent=40.0)
        pos = pos.copy()
        pos[: 3 * 57_000] = pos[: 3 * 57_000] * np.float32(1e-3) + np.float32(5.0)
        pos[3 * 1000: 3 * 2000] = pos[: 3 * 1000]
        del rng
    ref = host.Model.from_arrays(tri, pos, uv, nrm)
    ref.build_bvh()
    for threads in (2, 5, 8):
        m = host.Model.from_arrays(tri, pos, uv, nrm)
        m.build_bvh(threads)
        assert m.bvh_nodes().tobytes() == ref.bvh_nodes().tobytes(), (case, threads)
        assert m.triangles().tobytes() == ref.triangles().tobytes(), (case, threads)


def test_scene_cache_round_trip(tmp_path):
    tri, pos, uv, nrm, _ = scenes.atrium()
    m = host.Model.from_arrays(tri[:5000], pos, uv, nrm)
    m.build_bvh(4)
    sc = m.scene_arrays()
    path = str(tmp_path / "scene.npz")
    host.save_scene(path, sc)
    back = host.load_scene(path)
    for k in ("nodes", "triangles", "positions", "tex_coords", "normals"):
        assert back[k].dtype == sc[k].dtype and back[k].tobytes() == sc[k].tobytes(), k
    np.savez(str(tmp_path / "other.npz"), nodes=sc["nodes"])
    with pytest.raises(ValueError):
        host.load_scene(str(tmp_path / "other.npz"))


def test_obj_quirks():
    text = "\n".join([
        "# comment", "mtllib a.mtl", "v 0 0 0", "v 1 0 0", "v 1 1 0", "v 0 1 0", "  v 9 9 9",   # indented line is ignored
        "vt 0 0", "vt 1 0 0.5", "vt 1 1", "vn 0 0 1",
        "usemtl red", "f 1/1/1 2/2/1 3/3/1 4//1",      # quad -> fan (0,1,2) (0,2,3); missing vt -> index 0
        "usemtl blue", "f -4/-3/-1 -3\\2\\1 -2/3/1",   # negative indices, backslash separators
        "usemtl red", "f 1/1/1 2/2/1 3/3/1", ""])
    m = host.ResourceParser.parse_model(text)
    assert m.materials() == ["red", "blue"]
    t = m.triangles()
    assert list(t["mat_idx"]) == [0, 0, 1, 0]
    assert m.info().vertex_count == 4            # (pos,tex,nor) bundles unified in first-use order
    assert list(t[0]) == [0, 1, 2, 0] and list(t[1]) == [0, 2, 3, 0]
    assert list(t[2])[:3] == [0, 1, 2] and list(t[3])[:3] == [0, 1, 2]
    assert np.array_equal(m.tex_coords(), np.array([[0, 0], [1, 0], [1, 1], [0, 0]], np.float32))
    assert np.array_equal(m.positions()[3], [0, 1, 0])


@pytest.mark.parametrize("text,kind", [
    ("v 1 2\n", "InvalidLineArgCount"), ("v 1 2 3 4\n", "InvalidLineArgCount"), ("f 1 2\n", "InvalidLineArgCount"),
    ("v 0 0 0\nf 1 x 1\n", "ParseIntError"), ("v 0 0 zero\n", "ParseFloatError"), ("usemtl a b\n", "InvalidLineArgCount"),
    ("v 0 0 0\nf 1 2 3\n", "IndexOutOfRange")])
def test_obj_errors(text, kind):
    with pytest.raises(host.ResourceParseError, match=kind):
        host.ResourceParser.parse_model(text)


def test_obj_error_reports_line_number():
    with pytest.raises(host.ResourceParseError, match="line 3"):
        host.ResourceParser.parse_model("v 0 0 0\nv 1 0 0\nv 1 1\n")


def test_camera_defaults_and_matrices():
    cam = host.Camera.new_default(1000 / 800)            # camera.rs:34-42, window 1000x800 (main.rs:23)
    view, proj, inv = cam.view_proj_matrices()
    f = 1.0 / math.tan(math.radians(120.0) / 2)
    assert proj[5] == pytest.approx(f, rel=1e-6) and proj[0] == pytest.approx(f / 1.25, rel=1e-6)
    assert proj[11] == -1.0 and proj[10] == pytest.approx((1000 + 0.01) / (0.01 - 1000), rel=1e-6)
    pv = proj.reshape(4, 4).T @ view.reshape(4, 4).T
    assert np.allclose(pv @ inv.reshape(4, 4).T, np.eye(4), atol=2e-3)
    d, o = cam.primary_rays(8, 6)
    assert np.allclose(np.linalg.norm(d, axis=-1), 1, atol=1e-6) and not o.any()
    centre = d[2:4, 3:5].mean((0, 1))
    assert centre[0] > 0.9 and abs(centre[1]) < 1e-6 and abs(centre[2]) < 1e-6   # looks down +x
    assert d[-1, 0, 1] > 0 and d[0, 0, 1] < 0                                      # row 0 = bottom


def test_set_rotation_clamps_pitch():
    cam = host.Camera.new_default(1.0)
    cam.set_rotation(0.0, 10.0)
    assert cam.direction[1] == pytest.approx(math.sin(math.pi / 2.1), rel=1e-6)


def test_host_ray_generation_matches_oracle_restatement(oracle_a):
    cam = host.Camera(16 / 9, fov=math.radians(70), position=(1, 2, 3), direction=(0.3, -0.2, 0.9))
    inv = cam.inv_proj_view()
    d, _ = cam.primary_rays(32, 18)
    assert np.array_equal(d.view(np.uint32), oracle_a.raycreate(32, 18, inv, host.NEAR, host.FAR).view(np.uint32))


def test_subdivide_counts():
    g = load_golden("bunny")
    t = g["triangles"]
    tri = np.stack([t["p0"], t["p1"], t["p2"], t["mat_idx"]], 1)
    t2, p2, n2 = scenes.subdivide(tri, g["positions"], g["normals"], levels=2)
    assert len(t2) == 16 * len(tri) == 80016 and len(p2) == len(n2)
    assert np.allclose(np.linalg.norm(n2, axis=1), 1, atol=1e-5)


def test_config1_workload_is_the_f16_fixture_at_full_size():
    """bench.py's f16_640x480 (BASELINE config 1) takes scene, textures and camera from the committed llvmpipe fixture:
    its camera at the fixture's 160x120 gives the fixture's rays bit for bit, so the 640x480 frame is the frame the
    reference shader was pinned on, at four times the resolution."""
    import bench
    from common import bits, load_golden
    model, camera, w, h, tex = bench.build_workload("f16_640x480")
    g = load_golden("f16")
    assert (w, h) == (640, 480) and model.has_tex_coords() and model.has_normals()
    d, o = camera.primary_rays(160, 120)
    assert np.array_equal(bits(d), bits(g["dir"])) and np.array_equal(bits(o), bits(g["org"]))
    sc = model.scene_arrays()
    assert len(sc["triangles"]) == 4056 and np.array_equal(sc["nodes"], g["nodes"])
    assert tex[0].shape == (1024, 1024, 3) and tex[1].shape == (512, 256, 3)     # F16s.bmp, F16t.bmp (256 wide, 512 high)
