// BVHBuilder — binned-SAH top-down BVH build on the host, API and output kept
// identical to the reference (src/raytracing/bvh.rs:54-243): same f32 operation
// order, same node numbering (root 0, dummy leaf 1, sibling pairs adjacent, the
// first child's subtree numbered before the second's), same in-place triangle
// permutation, and the model's triangles/indices rewritten into leaf order.
//
// Parity status: the Rust builder cannot be executed in the build container (no
// rustc), so bit-identity with it is pinned only by an independent numpy
// restatement (oracle/bvh_oracle.py), a hand-checked cube.obj tree and structural
// invariants (tests/test_host_bvh.py) — "parity unpinned by execution".
#pragma once
#include <vector>

#include "model.h"
#include "types.h"

namespace rtamd {
namespace host {

class BVHBuilder {
public:
    explicit BVHBuilder(Model &model);  // bvh.rs:61-67
    BVH build();                        // bvh.rs:69-79 (consumes the builder)
    // Same tree, same node numbering, same triangle permutation as build(), bit for bit, on
    // `threads` host threads (no counterpart in the reference, whose build is
    // single-threaded).  Split decisions depend only on a node's triangle range, so the top of
    // the tree is split with chunk-parallel binning (min / max / count reductions merged in chunk
    // order are exact), the subtrees below are built by independent tasks with local numbering,
    // and one depth-first pass that replays build()'s allocation order relocates them.
    BVH build_parallel(unsigned threads);

    static constexpr size_t BINS = 50;  // bvh.rs:151

private:
    struct BVHTriangle {  // bvh.rs:6-38
        size_t p0, p1, p2;
        uint32_t mat_idx;
        float centroid[3];
    };
    struct Split { size_t axis; float pos; float cost; };

    AABB aabb_from_triangles(size_t first, size_t count) const;   // bvh.rs:101-110
    AABB aabb_from_centroids(size_t first, size_t count) const;   // bvh.rs:112-119
    void create_leaf_node_from_triangles(size_t first, size_t count);  // bvh.rs:121-127
    size_t split_triangles_along_plane(size_t first, size_t count, size_t axis, float position);  // bvh.rs:134-145
    Split find_best_split_binned(size_t first_tri, size_t tri_count) const;  // bvh.rs:150-204
    // bvh.rs:208-242; returns true and appends two children when the node was split
    bool split_leaf_node_sah(size_t node_idx);

    // pieces of build() usable on any node vector / from any thread (disjoint triangle ranges)
    Split find_best_split_chunked(size_t first_tri, size_t tri_count, unsigned threads) const;
    AABB aabb_from_triangles_chunked(size_t first, size_t count, unsigned threads) const;
    void build_subtree(std::vector<BVHNode> &out, size_t first, size_t count, const AABB &bounds);
    BVH finish(std::vector<BVHNode> nodes);

    Model &model_;
    std::vector<BVHTriangle> triangles_;
    std::vector<BVHNode> nodes_;
};

}  // namespace host
}  // namespace rtamd
