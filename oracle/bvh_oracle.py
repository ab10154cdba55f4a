"""bvh_oracle.py — TEST INFRASTRUCTURE ONLY.

Independent restatement (numpy float32 scalars, plain Python loops — small
meshes only) of the reference's host-side tree builder, used to cross-check the
C++ host layer (raytracer_amd/csrc/host/bvh.cpp).  It follows
src/raytracing/bvh.rs:6-243 and src/raytracing/types.rs:17-145 line by line:

  centroid            bvh.rs:16-21     ((v0 + v1) + v2) * (1/3 as f32)
  aabb_from_*         bvh.rs:101-119   AABBBuilder (Option<AABB>), types.rs:59-99
  find_best_split     bvh.rs:150-204   50 bins, prefix/suffix sweeps, strict <
  split_triangles     bvh.rs:134-145   swap-with-last partition
  split_leaf_node_sah bvh.rs:208-242   early outs at :214, :221, :232
  build               bvh.rs:69-79     root, dummy node at index 1, recursion

Parity status: "parity unpinned" against the Rust builder itself — no rustc in the
build image and the reference ships no node dumps; this file and the C++ builder
are two independent readings of the same source that must agree bit for bit.
"""
import sys

import numpy as np

f32 = np.float32
BINS = 50


class _Box:
    """AABBBuilder: None until the first include (types.rs:59-99)."""

    def __init__(self):
        self.mn = None
        self.mx = None

    def include(self, p):
        if self.mn is None:
            self.mn = p.copy()
            self.mx = p.copy()
        else:
            self.mn = np.fmin(self.mn, p)   # f32::min ignores NaN like fmin
            self.mx = np.fmax(self.mx, p)

    def include_other(self, o):
        if o.mn is not None:
            self.include(o.mn)
            self.include(o.mx)

    def build(self):
        if self.mn is None:
            z = np.zeros(3, f32)
            return z, z.copy()
        return self.mn, self.mx

    def area(self):
        if self.mn is None:
            return f32(0.0)
        e = self.mx - self.mn
        return f32(f32(f32(e[0] * e[1]) + f32(e[1] * e[2])) + f32(e[2] * e[0]))


def _area(mn, mx):
    e = mx - mn
    return f32(f32(f32(e[0] * e[1]) + f32(e[1] * e[2])) + f32(e[2] * e[0]))


def build_bvh(triangles, positions):
    """triangles: (n, 4) uint32 (p0, p1, p2, mat); positions: (v, 3) float32.
    Returns (nodes structured array of 36-byte records, reordered triangles)."""
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(100000)
    try:
        with np.errstate(all="ignore"):
            return _build(np.asarray(triangles, np.uint32).reshape(-1, 4).copy(), np.asarray(positions, f32).reshape(-1, 3))
    finally:
        sys.setrecursionlimit(old)


def _build(tris, pos):
    third = f32(1.0) / f32(3.0)
    n = len(tris)
    cent = np.zeros((n, 3), f32)
    for i in range(n):
        cent[i] = ((pos[tris[i, 0]] + pos[tris[i, 1]]) + pos[tris[i, 2]]) * third
    nodes = []  # [minx..maxz(6), is_leaf, a, b]

    def tri_box(first, count):
        b = _Box()
        for i in range(first, first + count):
            b.include(pos[tris[i, 0]]); b.include(pos[tris[i, 1]]); b.include(pos[tris[i, 2]])
        return b.build()

    def make_leaf(first, count):
        mn, mx = tri_box(first, count)
        nodes.append([mn.copy(), mx.copy(), 1, first, count])

    def best_split(first, count):
        best_axis, best_pos, lowest = 0, f32(0.0), f32(1e30)
        cb = _Box()
        for i in range(first, first + count):
            cb.include(cent[i])
        cmn, cmx = cb.build()
        for axis in range(3):
            mn, mx = cmn[axis], cmx[axis]
            if mn == mx:
                continue
            bins = [[_Box(), 0] for _ in range(BINS)]
            factor = f32(BINS) / f32(mx - mn)
            for i in range(first, first + count):
                v = f32(f32(cent[i, axis] - mn) * factor)
                # Rust `as usize`: truncate, saturate, NaN -> 0; then min(BINS-1, .)
                if v >= f32(BINS - 1):
                    bi = BINS - 1
                elif v > 0:
                    bi = int(v)
                else:
                    bi = 0
                bins[bi][1] += 1
                bins[bi][0].include(pos[tris[i, 0]]); bins[bi][0].include(pos[tris[i, 1]]); bins[bi][0].include(pos[tris[i, 2]])
            la = [f32(0)] * (BINS - 1); ra = [f32(0)] * (BINS - 1)
            lc = [0] * (BINS - 1); rc = [0] * (BINS - 1)
            lb, rb = _Box(), _Box()
            ls = rs = 0
            for i in range(BINS - 1):
                ls += bins[i][1]; lc[i] = ls
                lb.include_other(bins[i][0]); la[i] = lb.area()
                rs += bins[BINS - 1 - i][1]; rc[BINS - 2 - i] = rs
                rb.include_other(bins[BINS - 1 - i][0]); ra[BINS - 2 - i] = rb.area()
            step = f32(f32(mx - mn) / f32(BINS))
            for i in range(BINS - 1):
                cost = f32(f32(f32(lc[i]) * la[i]) + f32(f32(rc[i]) * ra[i]))
                if cost < lowest:
                    best_axis, best_pos, lowest = axis, f32(mn + f32(step * f32(i + 1))), cost
        return best_axis, best_pos, lowest

    def partition(first, count, axis, position):
        last = first + count
        while first < last:
            if cent[first, axis] < position:
                first += 1
            else:
                tris[[first, last - 1]] = tris[[last - 1, first]]
                cent[[first, last - 1]] = cent[[last - 1, first]]
                last -= 1
        return first

    def split(node_idx):
        node_count = len(nodes)
        first, count = nodes[node_idx][3], nodes[node_idx][4]
        if count <= 1:
            return
        axis, position, cost = best_split(first, count)
        parent_cost = f32(f32(count) * _area(nodes[node_idx][0], nodes[node_idx][1]))
        if parent_cost <= cost:
            return
        middle = partition(first, count, axis, position)
        left = middle - first
        if left == 0 or left == count:
            return
        nodes[node_idx][2], nodes[node_idx][3], nodes[node_idx][4] = 0, node_count, node_count + 1
        make_leaf(first, left)
        make_leaf(middle, count - left)
        split(node_count)
        split(node_count + 1)

    make_leaf(0, n)
    z = np.zeros(3, f32)
    nodes.append([z, z.copy(), 1, 0, 0])  # dummy, bvh.rs:72
    split(0)
    out = np.zeros(len(nodes), dtype=[("bounds", "<f4", (6,)), ("is_leaf", "<u4"), ("a", "<u4"), ("b", "<u4")])
    for i, nd in enumerate(nodes):
        out["bounds"][i, :3] = nd[0]; out["bounds"][i, 3:] = nd[1]
        out["is_leaf"][i], out["a"][i], out["b"][i] = nd[2], nd[3], nd[4]
    return out, tris
