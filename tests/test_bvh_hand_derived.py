"""BVH node tables derived BY HAND from the reference's builder (src/raytracing/bvh.rs:150-242, types.rs:59-99) — the one pin
for the tree that is neither the C++ builder nor its numpy restatement (no rustc here, no node dumps in the reference).  The
expected tables are data (tests/golden/cube_bvh_expected.json); this is how they were worked out.  h = fl(1/3) = 0.33333334.

cube.obj (res/models/cube.obj, 12 triangles T0..T11 in file order; ModelBuilder unifies the 8 positions in first-use order).
Centroids (v0+v1+v2)*THIRD: T0 (h,-h,1) T1 (-h,h,1) T2 (1,-h,-h) T3 (1,h,h) T4 (h,h,-1) T5 (-h,-h,-1) T6 (-1,-h,h) T7 (-1,h,-h)
T8 (h,-1,-h) T9 (-h,-1,h) T10 (h,1,-h) T11 (-h,1,h).  "face" = a 2x2x0 box (half-area 4), "cube" = the full box (half-area 12).

 node 0, 12 triangles, parent cost 12*12 = 144.  Centroid bounds [-1,1]^3, factor 25: c=-1 -> bin 0, -h -> 16, h -> 33, 1 -> 49.
   x: bin 0 {T6,T7} = the x=-1 face; bins 16 and 33 each hold a z=+1 and a z=-1 triangle = cube; bin 49 {T2,T3} = the x=+1 face.
      planes 0..15: 2*4 + 10*12 = 128; 16..32: 6*12 + 6*12 = 144; 33..48: 10*12 + 2*4 = 128.  y and z: the same numbers (128 is
      not < 128).  Best: x, plane 0, split -1 + 0.04 = -0.96.  Swap-with-last partition (bvh.rs:134-145) of T0..T11 gives
      [T7 T6 | T3 T4 T5 T2 T1 T8 T9 T10 T11 T0]  ->  node 2 = leaf(0,2), node 3 = leaf(2,10).
 node 2 {T7,T6}: centroid x both -1: axis skipped (min == max, :159); y: one triangle per end bin, 1*4 + 1*4 = 8; parent cost
      2*4 = 8 <= 8: stays a leaf (:221).
 node 3, 10 triangles, parent 120.  x: bounds [-h,1], factor 37.5: -h -> bin 0 {T5,T1,T9,T11} (cube), h -> bin 25 {T4,T8,T10,T0}
      (cube), 1 -> bin 49 {T3,T2} (face).  planes 0..24: 4*12 + 6*12 = 120; 25..48: 8*12 + 2*4 = 104.  y: 2*4 + 8*12 = 104 at best,
      z likewise: not < 104.  Split x at -h + 26*step (about 0.36): [T0 T4 T5 T11 T1 T8 T9 T10 | T2 T3] -> node 4 = leaf(2,8), node 5 = leaf(10,2).
 node 4, 8 triangles, parent 96.  x: two bins, both cube: 96.  y: factor 25: bin 0 {T8,T9} = the y=-1 face: planes 0..15:
      2*4 + 6*12 = 80.  z: 80, not < 80.  Split y at -0.96: [T9 T8 | T11 T1 T5 T4 T10 T0] -> node 6 = leaf(2,2), node 7 = leaf(4,6).
 node 6 {T9,T8}: y skipped; x: 4 + 4 = 8, parent 2*4 = 8: leaf.
 node 7, 6 triangles, parent 72.  x: 3*12 + 3*12 = 72.  y: bounds [-h,1], factor 37.5: bin 0 {T5,T0} cube, bin 25 {T1,T4} cube,
      bin 49 {T11,T10} = the y=+1 face: planes 25..48: 4*12 + 2*4 = 56.  z: 56 at best, not < 56.  Split y at about 0.36:
      [T0 T1 T5 T4 | T10 T11] -> node 8 = leaf(4,4), node 9 = leaf(8,2).
 node 8 {T0,T1,T5,T4}, parent 48.  x: 48; y: 48; z: bin 0 {T5,T4} = the z=-1 face, bin 49 {T0,T1} = the z=+1 face: 2*4 + 2*4 = 16.
      Split z at -0.96: [T4 T5 | T1 T0] -> node 10 = leaf(4,2), node 11 = leaf(6,2).
 nodes 10, 11, 9, 5: two coplanar triangles each, best cost 8, parent cost 8: leaves.
 Final triangle order: T7 T6 T9 T8 T4 T5 T1 T0 T10 T11 T2 T3.

Four triangles that reach BOTH early exits.  All in the plane z = 0 with centroid y = 1 (axes y and z: min == max, :159), two
(needles) at x = P = 2^24 and two at x = P+2.  f32 has spacing 2 there: the centroids come out as P and P+4, so the x bins are 0
and 49, every plane costs 2*0 + 2*0 = 0 < parent cost 4*4 = 16, plane 0 wins, and its split position P + 4/50 rounds to P:
no centroid is < P, the partition is one-sided and the node stays a leaf (:232) — but the partition's swaps have already
rotated the triangles to [T1 T2 T3 T0], which build() writes back to the model (:74)."""
import json
import os

import numpy as np

from raytracer_amd import host
from test_host_layer import CUBE_OBJ

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXPECTED = json.load(open(os.path.join(REPO, "tests", "golden", "cube_bvh_expected.json")))


def _check(exp, tri0, pos, nodes, tris, who):
    want_tris = np.asarray(tri0, np.uint32)[exp["triangle_order"]]
    assert np.array_equal(np.stack([tris["p0"], tris["p1"], tris["p2"]], 1), want_tris), who
    assert len(nodes) == len(exp["nodes"]), who
    for i, e in enumerate(exp["nodes"]):
        assert np.array_equal(nodes["bounds"][i], np.float32(e["bounds"])), (who, i)
        assert (int(nodes["is_leaf"][i]), int(nodes["a"][i]), int(nodes["b"][i])) == (e["is_leaf"], e["a"], e["b"]), (who, i)


def _numpy_builder(tri4, pos):
    from oracle.bvh_oracle import build_bvh
    nodes, tris = build_bvh(tri4, pos)
    return nodes, {"p0": tris[:, 0], "p1": tris[:, 1], "p2": tris[:, 2]}


def test_cube_obj_tree_equals_the_hand_derived_table():
    exp = EXPECTED["cube_obj"]
    m = host.ResourceParser.parse_model(CUBE_OBJ)
    tri0, pos = m.triangles().copy(), m.positions().copy()
    # the derivation's starting point: ModelBuilder's unified vertices and triangles (model.rs:101-149), first-use order
    assert np.array_equal(pos, np.float32(exp["unified_positions"]))
    assert np.array_equal(np.stack([tri0["p0"], tri0["p1"], tri0["p2"]], 1), np.uint32(exp["unified_triangles_before_build"]))
    tri4 = np.stack([tri0["p0"], tri0["p1"], tri0["p2"], tri0["mat_idx"]], 1)
    for threads in (1, 4):
        mm = host.ResourceParser.parse_model(CUBE_OBJ)
        mm.build_bvh(threads)
        _check(exp, exp["unified_triangles_before_build"], pos, mm.bvh_nodes(), mm.triangles(), f"C++ builder, {threads} thread(s)")
    nodes, tris = _numpy_builder(tri4, pos)
    _check(exp, exp["unified_triangles_before_build"], pos, nodes, tris, "oracle/bvh_oracle.py")


def test_axis_skip_and_one_sided_partition_stop_equal_the_hand_derived_table():
    exp = EXPECTED["four_triangles_axis_skip_and_one_sided_stop"]
    pos = np.float32(exp["positions"])
    assert pos[6, 0] == np.float32(16777218.0)                     # 2^24 + 2 is a float
    tri4 = np.concatenate([np.uint32(exp["triangles_before_build"]), np.zeros((4, 1), np.uint32)], 1)
    for threads in (1, 4):
        m = host.Model.from_arrays(tri4, pos)
        m.build_bvh(threads)
        _check(exp, exp["triangles_before_build"], pos, m.bvh_nodes(), m.triangles(), f"C++ builder, {threads} thread(s)")
    nodes, tris = _numpy_builder(tri4, pos)
    _check(exp, exp["triangles_before_build"], pos, nodes, tris, "oracle/bvh_oracle.py")
