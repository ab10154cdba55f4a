#include "bvh.h"

#include <algorithm>
#include <atomic>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <utility>
#include <chrono>
#include <cstdio>
#include <cstdlib>

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
    return finish(std::move(nodes_));
}

BVH BVHBuilder::finish(std::vector<BVHNode> nodes) {
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
    return BVH(std::move(nodes));
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

// ---------------------------------------------------------------------------
// build_parallel (row f3): identical output on several threads
// ---------------------------------------------------------------------------
namespace {

// Small persistent pool for the chunked passes of the top phase (a thread spawn per chunk
// costs more than binning 8 192 triangles).  run(n, fn) executes fn(0..n-1) and returns when
// all are done; the calling thread takes part.
class Pool {
public:
    explicit Pool(unsigned threads) {
        for (unsigned i = 1; i < threads; i++) workers_.emplace_back([this]() { loop(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; gen_++; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void run(unsigned n, const std::function<void(unsigned)> &fn) {
        if (n == 0) return;
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn; n_ = n; next_ = 0; pending_ = n; gen_++;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this]() { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void work() {
        for (;;) {
            unsigned k;
            const std::function<void(unsigned)> *fn;
            {
                std::lock_guard<std::mutex> g(m_);
                if (!fn_ || next_ >= n_) return;
                k = next_++; fn = fn_;
            }
            (*fn)(k);
            std::lock_guard<std::mutex> g(m_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&]() { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
            }
            work();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(unsigned)> *fn_ = nullptr;
    unsigned n_ = 0, next_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

Pool *g_pool = nullptr;   // set for the duration of one build_parallel call (builds are not re-entrant per process thread)

// run fn(chunk, begin, end) over [first, first+count) cut into `chunks` contiguous pieces
void for_chunks(size_t first, size_t count, unsigned chunks, const std::function<void(unsigned, size_t, size_t)> &fn) {
    if (chunks <= 1 || !g_pool) { for (unsigned c = 0; c < std::max(1u, chunks); c++) fn(c, first + count * c / std::max(1u, chunks), first + count * (c + 1) / std::max(1u, chunks)); return; }
    g_pool->run(chunks, [&](unsigned c) { fn(c, first + count * c / chunks, first + count * (c + 1) / chunks); });
}

// Merge of two AABBBuilder states where every point of `later` was included after every
// point of `acc` in the sequential order: component-wise f32_min/f32_max with the later
// operand second, which is exactly what the sequential chain of include() calls computes
// (also for NaN and signed-zero operands, whose result depends on operand order).
void merge_later(AABBBuilder &acc, const AABBBuilder &later) {
    if (!later.some) return;
    if (!acc.some) { acc = later; return; }
    acc.aabb.min.x = f32_min(acc.aabb.min.x, later.aabb.min.x); acc.aabb.min.y = f32_min(acc.aabb.min.y, later.aabb.min.y);
    acc.aabb.min.z = f32_min(acc.aabb.min.z, later.aabb.min.z);
    acc.aabb.max.x = f32_max(acc.aabb.max.x, later.aabb.max.x); acc.aabb.max.y = f32_max(acc.aabb.max.y, later.aabb.max.y);
    acc.aabb.max.z = f32_max(acc.aabb.max.z, later.aabb.max.z);
}

}  // namespace

AABB BVHBuilder::aabb_from_triangles_chunked(size_t first, size_t count, unsigned threads) const {
    const unsigned chunks = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, count / 8192));
    if (chunks <= 1) return aabb_from_triangles(first, count);
    const std::vector<Vec3> &v = model_.positions();
    std::vector<AABBBuilder> part(chunks);
    for_chunks(first, count, chunks, [&](unsigned c, size_t b, size_t e) {
        AABBBuilder bb;
        for (size_t i = b; i < e; i++) { const BVHTriangle &t = triangles_[i]; bb.include(v[t.p0]); bb.include(v[t.p1]); bb.include(v[t.p2]); }
        part[c] = bb;
    });
    AABBBuilder acc;
    for (const AABBBuilder &pb : part) merge_later(acc, pb);
    return acc.build();
}

// find_best_split_binned with the two O(n) passes (centroid bounds, binning) cut into chunks.
BVHBuilder::Split BVHBuilder::find_best_split_chunked(size_t first_tri, size_t tri_count, unsigned threads) const {
    const unsigned chunks = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, tri_count / 8192));
    if (chunks <= 1) return find_best_split_binned(first_tri, tri_count);
    const std::vector<Vec3> &v = model_.positions();
    AABBBuilder cb;
    {
        std::vector<AABBBuilder> part(chunks);
        for_chunks(first_tri, tri_count, chunks, [&](unsigned c, size_t b, size_t e) {
            AABBBuilder bb;
            for (size_t i = b; i < e; i++) { const BVHTriangle &t = triangles_[i]; bb.include(Vec3{t.centroid[0], t.centroid[1], t.centroid[2]}); }
            part[c] = bb;
        });
        for (const AABBBuilder &pb : part) merge_later(cb, pb);
    }
    const AABB bounds = cb.build();
    // all three axes binned in one pass per chunk
    struct AxisBins { Bin bins[3][BINS]; };
    std::vector<AxisBins> part(chunks);
    float mn[3], factor[3];
    bool use[3];
    for (int a = 0; a < 3; a++) {
        mn[a] = bounds.min[a];
        use[a] = !(bounds.min[a] == bounds.max[a]);
        factor[a] = use[a] ? (float)BINS / (bounds.max[a] - bounds.min[a]) : 0.0f;
    }
    for_chunks(first_tri, tri_count, chunks, [&](unsigned c, size_t b, size_t e) {
        AxisBins &ab = part[c];
        for (size_t i = b; i < e; i++) {
            const BVHTriangle &t = triangles_[i];
            for (int a = 0; a < 3; a++) {
                if (!use[a]) continue;
                Bin &bin = ab.bins[a][bin_index((t.centroid[a] - mn[a]) * factor[a], BINS)];
                bin.tri_count += 1;
                bin.bounds.include(v[t.p0]); bin.bounds.include(v[t.p1]); bin.bounds.include(v[t.p2]);
            }
        }
    });
    size_t best_axis = 0;
    float best_split_pos = 0.0f;
    float lowest_cost = 1e30f;
    for (size_t axis = 0; axis < 3; axis++) {
        if (!use[axis]) continue;
        const float lo = bounds.min[(int)axis], hi = bounds.max[(int)axis];
        Bin bins[BINS];
        for (unsigned c = 0; c < chunks; c++)
            for (size_t k = 0; k < BINS; k++) {
                bins[k].tri_count += part[c].bins[axis][k].tri_count;
                merge_later(bins[k].bounds, part[c].bins[axis][k].bounds);
            }
        // from here on: the reference's sweep, verbatim from find_best_split_binned
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
        float step_size = (hi - lo) / (float)BINS;
        for (size_t i = 0; i < BINS - 1; i++) {
            float cost = (float)left_count[i] * left_area[i] + (float)right_count[i] * right_area[i];
            if (cost < lowest_cost) {
                best_axis = axis;
                best_split_pos = lo + step_size * (float)(i + 1);
                lowest_cost = cost;
            }
        }
    }
    return Split{best_axis, best_split_pos, lowest_cost};
}

// build() restricted to one subtree, into `out` with local numbering: out[0] is the subtree
// root, its descendants follow in build()'s allocation order (pairs, first child's subtree
// first).  Touches only triangles_[first, first+count).
void BVHBuilder::build_subtree(std::vector<BVHNode> &out, size_t first, size_t count, const AABB &bounds) {
    out.clear();
    out.push_back(BVHNode::new_leaf(bounds, (uint32_t)first, (uint32_t)count));
    std::vector<size_t> todo{0};
    while (!todo.empty()) {
        const size_t ni = todo.back();
        todo.pop_back();
        const size_t f = out[ni].first_triangle(), n = out[ni].triangle_count();
        if (n <= 1) continue;
        const Split s = find_best_split_binned(f, n);
        if ((float)n * out[ni].bounds.area() <= s.cost) continue;
        const size_t middle = split_triangles_along_plane(f, n, s.axis, s.pos);
        const size_t lc = middle - f;
        if (lc == 0 || lc == n) continue;
        const size_t k = out.size();
        out[ni].convert_to_node((uint32_t)k, (uint32_t)(k + 1));
        out.push_back(BVHNode::new_leaf(aabb_from_triangles(f, lc), (uint32_t)f, (uint32_t)lc));
        out.push_back(BVHNode::new_leaf(aabb_from_triangles(middle, n - lc), (uint32_t)middle, (uint32_t)(n - lc)));
        todo.push_back(k + 1);
        todo.push_back(k);
    }
}

BVH BVHBuilder::build_parallel(unsigned threads) {
    if (threads < 1) threads = 1;
    static std::mutex build_mutex;   // one parallel build at a time per process (g_pool)
    std::lock_guard<std::mutex> build_guard(build_mutex);
    Pool pool(threads);
    g_pool = &pool;
    struct PoolReset { ~PoolReset() { g_pool = nullptr; } } pool_reset;
    const size_t n_tris = triangles_.size();
    // ---- 1. top of the tree: split every node above the task size, big ranges chunk-parallel
    struct Top { AABB bounds; size_t first, count; int a = -1, b = -1; int task = -1; };
    std::vector<Top> top;
    top.push_back(Top{aabb_from_triangles_chunked(0, n_tris, threads), 0, n_tris});
    const size_t task_size = std::max<size_t>(2048, n_tris / ((size_t)threads * 32));
    const double t_begin = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    std::vector<int> tasks;
    // level by level (any order would do: a node's split depends on its range only): big nodes one
    // after the other with chunk-parallel passes, the smaller ones of a level in parallel with each other
    struct SplitResult { bool split = false; size_t middle = 0; AABB ba{}, bb{}; };
    const size_t big = (size_t)8192 * threads;
    auto try_split = [&](const Top &t, unsigned th) {
        SplitResult r;
        const size_t f = t.first, n = t.count;
        const Split s = th > 1 ? find_best_split_chunked(f, n, th) : find_best_split_binned(f, n);
        if ((float)n * t.bounds.area() <= s.cost) return r;
        const size_t middle = split_triangles_along_plane(f, n, s.axis, s.pos);
        const size_t lc = middle - f;
        if (lc == 0 || lc == n) return r;
        r.split = true; r.middle = middle;
        r.ba = th > 1 ? aabb_from_triangles_chunked(f, lc, th) : aabb_from_triangles(f, lc);
        r.bb = th > 1 ? aabb_from_triangles_chunked(middle, n - lc, th) : aabb_from_triangles(middle, n - lc);
        return r;
    };
    std::vector<size_t> frontier{0};
    while (!frontier.empty()) {
        std::vector<SplitResult> res(frontier.size());
        std::vector<unsigned> small;
        for (size_t i = 0; i < frontier.size(); i++) {
            Top &t = top[frontier[i]];
            if (t.count <= task_size) { t.task = (int)tasks.size(); tasks.push_back((int)frontier[i]); }
            else if (t.count >= big) res[i] = try_split(t, threads);
            else small.push_back((unsigned)i);
        }
        pool.run((unsigned)small.size(), [&](unsigned j) { res[small[j]] = try_split(top[frontier[small[j]]], 1); });
        std::vector<size_t> next_frontier;
        for (size_t i = 0; i < frontier.size(); i++) {
            if (!res[i].split) continue;
            const size_t ti = frontier[i];
            const size_t f = top[ti].first, n = top[ti].count, lc = res[i].middle - f;
            top[ti].a = (int)top.size();
            top[ti].b = (int)top.size() + 1;
            top.push_back(Top{res[i].ba, f, lc});
            top.push_back(Top{res[i].bb, res[i].middle, n - lc});
            next_frontier.push_back(top.size() - 2);
            next_frontier.push_back(top.size() - 1);
        }
        frontier.swap(next_frontier);
    }
    const bool timing = std::getenv("RTAMD_BVH_TIMING") != nullptr;
    if (timing) std::fprintf(stderr, "[bvh] top phase %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_begin);
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_top = now();
    // ---- 2. subtrees: independent tasks, largest first
    std::vector<std::vector<BVHNode>> sub(tasks.size());
    std::vector<size_t> order(tasks.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return top[tasks[x]].count > top[tasks[y]].count; });
    std::vector<double> task_s(order.size(), 0.0);
    pool.run((unsigned)order.size(), [&](unsigned k) {
        const Top &t = top[tasks[order[k]]];
        const double t0 = timing ? now() : 0.0;
        build_subtree(sub[order[k]], t.first, t.count, t.bounds);
        if (timing) task_s[k] = now() - t0;
    });
    if (timing) {
        double sum = 0, mx = 0;
        for (double x : task_s) { sum += x; mx = std::max(mx, x); }
        std::fprintf(stderr, "[bvh] task seconds: sum %.3f max %.3f\n", sum, mx);
    }
    const double t_sub = now();
    if (timing) {
        size_t biggest = 0;
        for (int ti : tasks) biggest = std::max(biggest, top[ti].count);
        std::fprintf(stderr, "[bvh] top nodes %zu, tasks %zu (largest %zu tris, task size %zu), subtrees %.3f s\n", top.size(), tasks.size(), biggest, task_size, t_sub - t_top);
    }
    // ---- 3. replay build()'s allocation order: root 0, dummy 1, children pairs taken from the
    // running node count when their parent is processed, first child's subtree before the second's
    size_t total = 2;
    for (const Top &t : top) if (t.a >= 0) total += 2;
    for (const auto &sv : sub) total += sv.size() - 1;
    std::vector<BVHNode> nodes(total);
    nodes[0] = BVHNode::new_leaf(top[0].bounds, (uint32_t)top[0].first, (uint32_t)top[0].count);
    nodes[1] = BVHNode::new_dummy();
    size_t counter = 2;
    std::vector<std::pair<int, size_t>> todo{{0, 0}};   // (top index, global index)
    while (!todo.empty()) {
        const auto [ti, gi] = todo.back();
        todo.pop_back();
        const Top &t = top[ti];
        if (t.task >= 0) {
            // local index l >= 1 lands at counter + l - 1; local 0 is the node at gi
            const std::vector<BVHNode> &sv = sub[t.task];
            const size_t shift = counter - 1;
            BVHNode root = sv[0];
            if (!root.is_leaf) { root.a += (uint32_t)shift; root.b += (uint32_t)shift; }
            nodes[gi] = root;
            for (size_t l = 1; l < sv.size(); l++) {
                BVHNode nd = sv[l];
                if (!nd.is_leaf) { nd.a += (uint32_t)shift; nd.b += (uint32_t)shift; }
                nodes[shift + l] = nd;
            }
            counter += sv.size() - 1;
        } else if (t.a >= 0) {
            const size_t k = counter;
            counter += 2;
            nodes[gi].convert_to_node((uint32_t)k, (uint32_t)(k + 1));
            nodes[k] = BVHNode::new_leaf(top[t.a].bounds, (uint32_t)top[t.a].first, (uint32_t)top[t.a].count);
            nodes[k + 1] = BVHNode::new_leaf(top[t.b].bounds, (uint32_t)top[t.b].first, (uint32_t)top[t.b].count);
            todo.push_back({t.b, k + 1});
            todo.push_back({t.a, k});
        }
    }
    return finish(std::move(nodes));
}

}  // namespace host
}  // namespace rtamd
