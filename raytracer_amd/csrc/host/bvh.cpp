#include "bvh.h"

#include <utility>

namespace rtamd {
namespace host {

BVHBuilder::BVHBuilder(Model &model) : model_(model) {
    const float THIRD = 1.0f / 3.0f;  // bvh.rs:16
    const std::vector<Vec3> &v = model.positions();
    triangles_.reserve(model.triangles().size());
    for (const Triangle &t : model.triangles()) {
        Vec3 c = ((v[t.p0] + v[t.p1]) + v[t.p2]) * THIRD;  // bvh.rs:17-21
        triangles_.push_back(BVHTriangle{t.p0, t.p1, t.p2, t.mat_idx, {c.x, c.y, c.z}});
    }
    nodes_.reserve(model.triangles().size() * 2);
}

BVH BVHBuilder::build() {
    create_leaf_node_from_triangles(0, triangles_.size());
    nodes_.push_back(BVHNode::new_dummy());  // bvh.rs:72: index 1 is never referenced
    // The reference recurses (split child a's whole subtree, then child b's).  The
    // same visiting order with an explicit LIFO keeps deep trees off the C stack;
    // node indices are taken from nodes_.size() at the moment a node is processed,
    // exactly as `node_count` is at bvh.rs:209.
    std::vector<size_t> todo;
    todo.push_back(0);
    while (!todo.empty()) {
        size_t node_idx = todo.back();
        todo.pop_back();
        size_t first_child = nodes_.size();
        if (split_leaf_node_sah(node_idx)) {
            todo.push_back(first_child + 1);
            todo.push_back(first_child);
        }
    }
    std::vector<Triangle> tris;
    std::vector<uint32_t> idx;
    tris.reserve(triangles_.size());
    idx.reserve(triangles_.size() * 3);
    for (const BVHTriangle &t : triangles_) {  // bvh.rs:74-77
        tris.push_back(Triangle{(uint32_t)t.p0, (uint32_t)t.p1, (uint32_t)t.p2, t.mat_idx});
        idx.push_back((uint32_t)t.p0); idx.push_back((uint32_t)t.p1); idx.push_back((uint32_t)t.p2);
    }
    model_.set_triangles(std::move(tris));
    model_.set_indices(std::move(idx));
    return BVH(std::move(nodes_));
}

AABB BVHBuilder::aabb_from_triangles(size_t first, size_t count) const {
    const std::vector<Vec3> &v = model_.positions();
    AABBBuilder b;
    for (size_t i = first; i < first + count; i++) {
        const BVHTriangle &t = triangles_[i];
        b.include(v[t.p0]); b.include(v[t.p1]); b.include(v[t.p2]);
    }
    return b.build();
}

AABB BVHBuilder::aabb_from_centroids(size_t first, size_t count) const {
    AABBBuilder b;
    for (size_t i = first; i < first + count; i++) {
        const BVHTriangle &t = triangles_[i];
        b.include(Vec3{t.centroid[0], t.centroid[1], t.centroid[2]});
    }
    return b.build();
}

void BVHBuilder::create_leaf_node_from_triangles(size_t first, size_t count) {
    nodes_.push_back(BVHNode::new_leaf(aabb_from_triangles(first, count), (uint32_t)first, (uint32_t)count));
}

size_t BVHBuilder::split_triangles_along_plane(size_t first, size_t count, size_t axis, float position) {
    size_t last = first + count;
    while (first < last) {
        if (triangles_[first].centroid[axis] < position) {
            first += 1;
        } else {
            std::swap(triangles_[first], triangles_[last - 1]);
            last -= 1;
        }
    }
    return first;
}

// Rust `f32 as usize` then usize::min(BINS-1, ..): truncation toward zero,
// saturating, NaN -> 0 (bvh.rs:165).
static inline size_t bin_index(float v, size_t bins) {
    if (v >= (float)(bins - 1)) return bins - 1;
    if (v > 0.0f) return (size_t)v;
    return 0;
}

BVHBuilder::Split BVHBuilder::find_best_split_binned(size_t first_tri, size_t tri_count) const {
    const std::vector<Vec3> &v = model_.positions();
    size_t best_axis = 0;
    float best_split_pos = 0.0f;
    float lowest_cost = 1e30f;
    AABB bounds = aabb_from_centroids(first_tri, tri_count);
    for (size_t axis = 0; axis < 3; axis++) {
        float mn = bounds.min[(int)axis];
        float mx = bounds.max[(int)axis];
        if (mn == mx) continue;

        Bin bins[BINS];
        float factor = (float)BINS / (mx - mn);
        for (size_t i = first_tri; i < first_tri + tri_count; i++) {
            const BVHTriangle &t = triangles_[i];
            size_t bin_idx = bin_index((t.centroid[axis] - mn) * factor, BINS);
            bins[bin_idx].tri_count += 1;
            bins[bin_idx].bounds.include(v[t.p0]);
            bins[bin_idx].bounds.include(v[t.p1]);
            bins[bin_idx].bounds.include(v[t.p2]);
        }

        float left_area[BINS - 1], right_area[BINS - 1];
        uint32_t left_count[BINS - 1], right_count[BINS - 1];
        AABBBuilder left_box, right_box;
        uint32_t left_sum = 0, right_sum = 0;
        for (size_t i = 0; i < BINS - 1; i++) {
            left_sum += bins[i].tri_count;
            left_count[i] = left_sum;
            left_box.include_other(bins[i].bounds);
            left_area[i] = left_box.area();

            right_sum += bins[BINS - 1 - i].tri_count;
            right_count[BINS - 2 - i] = right_sum;
            right_box.include_other(bins[BINS - 1 - i].bounds);
            right_area[BINS - 2 - i] = right_box.area();
        }

        float step_size = (mx - mn) / (float)BINS;
        for (size_t i = 0; i < BINS - 1; i++) {
            float cost = (float)left_count[i] * left_area[i] + (float)right_count[i] * right_area[i];
            if (cost < lowest_cost) {
                best_axis = axis;
                best_split_pos = mn + step_size * (float)(i + 1);
                lowest_cost = cost;
            }
        }
    }
    return Split{best_axis, best_split_pos, lowest_cost};
}

bool BVHBuilder::split_leaf_node_sah(size_t node_idx) {
    size_t node_count = nodes_.size();
    size_t first_triangle = nodes_[node_idx].first_triangle();
    size_t triangle_count = nodes_[node_idx].triangle_count();

    if (triangle_count <= 1) return false;  // bvh.rs:214

    Split s = find_best_split_binned(first_triangle, triangle_count);

    float parent_cost = (float)triangle_count * nodes_[node_idx].bounds.area();
    if (parent_cost <= s.cost) return false;  // bvh.rs:220-221

    size_t middle = split_triangles_along_plane(first_triangle, triangle_count, s.axis, s.pos);

    size_t left_count = middle - first_triangle;
    size_t right_count = triangle_count - left_count;
    if (left_count == 0 || left_count == triangle_count) return false;  // bvh.rs:232

    nodes_[node_idx].convert_to_node((uint32_t)node_count, (uint32_t)(node_count + 1));
    create_leaf_node_from_triangles(first_triangle, left_count);
    create_leaf_node_from_triangles(middle, right_count);
    return true;
}

}  // namespace host
}  // namespace rtamd
