#include "model.h"

#include <algorithm>

#include "bvh.h"

namespace rtamd {
namespace host {

void Model::build_bvh() { bvh_ = std::make_unique<BVH>(BVHBuilder(*this).build()); }
void Model::build_bvh_parallel(unsigned threads) { bvh_ = std::make_unique<BVH>(BVHBuilder(*this).build_parallel(threads)); }

Model Model::from_arrays(std::vector<Triangle> triangles, std::vector<Vec3> positions,
                         const std::vector<Vec2> *tex_coords, const std::vector<Vec3> *normals) {
    Model m;
    m.indices_.reserve(triangles.size() * 3);
    for (const Triangle &t : triangles) { m.indices_.push_back(t.p0); m.indices_.push_back(t.p1); m.indices_.push_back(t.p2); }
    m.triangles_ = std::move(triangles);
    m.positions_ = std::move(positions);
    if (tex_coords) { m.tex_coords_ = *tex_coords; m.has_tex_coords_ = true; }
    if (normals) { m.normals_ = *normals; m.has_normals_ = true; }
    return m;
}

void ModelBuilder::add_material_lib(const std::string &lib) {
    if (std::find(material_libs_.begin(), material_libs_.end(), lib) == material_libs_.end()) material_libs_.push_back(lib);
}

void ModelBuilder::add_material(const std::string &mat) {
    auto it = materials_.find(mat);
    if (it != materials_.end()) {
        current_mat_ = it->second;
    } else {
        uint32_t next_id = (uint32_t)materials_.size();
        materials_.emplace(mat, next_id);
        current_mat_ = next_id;
    }
}

namespace {
struct BundleHash {
    size_t operator()(const IndexBundle &b) const {
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (uint32_t v : {(uint32_t)b.pos_idx, (uint32_t)b.tex_idx, (uint32_t)b.nor_idx}) {
            h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        }
        return (size_t)h;
    }
};
}  // namespace

Model ModelBuilder::build(ParseError *err) const {
    // Vertex unification: one output vertex per distinct (pos,tex,nor) bundle,
    // numbered in first-use order (model.rs:116-134).  Only lookups/inserts touch
    // the map, so its iteration order never matters.
    std::unordered_map<IndexBundle, uint32_t, BundleHash> bundle_map;
    Model m;
    const bool has_tex = !tex_coords_.empty();
    const bool has_nor = !normals_.empty();
    const int32_t pos_len = (int32_t)positions_.size();
    const int32_t tex_len = (int32_t)tex_coords_.size();
    const int32_t nor_len = (int32_t)normals_.size();
    uint32_t new_idx = 0;
    m.triangles_.reserve(indices_.size());
    m.indices_.reserve(indices_.size() * 3);
    for (const IBTriangle &ib_tri : indices_) {
        uint32_t idx[3];
        for (int k = 0; k < 3; k++) {
            IndexBundle ib = ib_tri.index_bundles[k];
            ib.normalize(pos_len, tex_len, nor_len);
            auto it = bundle_map.find(ib);
            if (it != bundle_map.end()) { idx[k] = it->second; continue; }
            bool bad = ib.pos_idx < 0 || ib.pos_idx >= pos_len ||
                       (has_tex && (ib.tex_idx < 0 || ib.tex_idx >= tex_len)) ||
                       (has_nor && (ib.nor_idx < 0 || ib.nor_idx >= nor_len));
            if (bad) {
                if (err) { err->kind = ParseErrorKind::IndexOutOfRange; err->detail = "face references a vertex attribute that does not exist"; }
                return Model();
            }
            m.positions_.push_back(positions_[(size_t)ib.pos_idx]);
            if (has_tex) m.tex_coords_.push_back(tex_coords_[(size_t)ib.tex_idx]);
            if (has_nor) m.normals_.push_back(normals_[(size_t)ib.nor_idx]);
            bundle_map.emplace(ib, new_idx);
            idx[k] = new_idx++;
        }
        m.triangles_.push_back(Triangle{idx[0], idx[1], idx[2], ib_tri.mat_idx});
        m.indices_.push_back(idx[0]); m.indices_.push_back(idx[1]); m.indices_.push_back(idx[2]);
    }
    m.has_tex_coords_ = has_tex;
    m.has_normals_ = has_nor;
    m.material_libs_ = material_libs_;
    m.materials_.assign(materials_.size(), std::string());
    for (const auto &kv : materials_) m.materials_[kv.second] = kv.first;
    return m;
}

}  // namespace host
}  // namespace rtamd
