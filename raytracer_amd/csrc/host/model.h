// Model + ModelBuilder — host-side scene container, API kept from the reference.
//   Model         src/rendering/model.rs:6-42
//   ModelBuilder  src/rendering/model.rs:55-149
// The accessors (triangles/positions/tex_coords/normals/get_bvh) return exactly
// the slices the reference uploads at src/main.rs:98-102 and that
// rtamd_upload_scene() (include/rtamd.h) takes.
#pragma once
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "types.h"

namespace rtamd {
namespace host {

class BVH {  // src/raytracing/bvh.rs:40-52
public:
    BVH() = default;
    explicit BVH(std::vector<BVHNode> nodes) : nodes_(std::move(nodes)) {}
    const std::vector<BVHNode> &data() const { return nodes_; }

private:
    std::vector<BVHNode> nodes_;
};

class Model {
public:
    const std::vector<Triangle> &triangles() const { return triangles_; }
    const std::vector<uint32_t> &indices() const { return indices_; }
    const std::vector<Vec3> &positions() const { return positions_; }
    // Option<Vec<..>>: nullptr == None
    const std::vector<Vec2> *tex_coords() const { return has_tex_coords_ ? &tex_coords_ : nullptr; }
    const std::vector<Vec3> *normals() const { return has_normals_ ? &normals_ : nullptr; }
    bool has_tex_coords() const { return has_tex_coords_; }
    bool has_normals() const { return has_normals_; }
    void set_triangles(std::vector<Triangle> t) { triangles_ = std::move(t); }
    void set_indices(std::vector<uint32_t> i) { indices_ = std::move(i); }
    void build_bvh();  // model.rs:36
    void build_bvh_parallel(unsigned threads);  // same BVH bit for bit (BVHBuilder::build_parallel)
    const std::vector<std::string> &get_material_libs() const { return material_libs_; }
    const std::vector<std::string> &get_materials() const { return materials_; }
    const BVH *get_bvh() const { return bvh_ ? bvh_.get() : nullptr; }

    // Direct construction from arrays (synthetic scenes); not in the reference,
    // which only builds models through ModelBuilder.
    static Model from_arrays(std::vector<Triangle> triangles, std::vector<Vec3> positions,
                             const std::vector<Vec2> *tex_coords, const std::vector<Vec3> *normals);

private:
    friend class ModelBuilder;
    std::vector<Triangle> triangles_;
    std::vector<uint32_t> indices_;
    std::vector<Vec3> positions_;
    std::vector<Vec2> tex_coords_;
    std::vector<Vec3> normals_;
    bool has_tex_coords_ = false, has_normals_ = false;
    std::vector<std::string> material_libs_;
    std::vector<std::string> materials_;
    std::unique_ptr<BVH> bvh_;
};

class ModelBuilder {
public:
    void add_indices(const IndexBundle &i0, const IndexBundle &i1, const IndexBundle &i2, uint32_t mat_idx) {
        indices_.push_back({{i0, i1, i2}, mat_idx});
    }
    void add_position(Vec3 p) { positions_.push_back(p); }
    void add_tex_coord(Vec2 t) { tex_coords_.push_back(t); }
    void add_normal(Vec3 n) { normals_.push_back(n); }
    void add_material_lib(const std::string &lib);
    void add_material(const std::string &mat);  // model.rs:89-97: ids in first-use order
    uint32_t get_current_mat() const { return current_mat_; }
    // model.rs:101-149.  The reference panics on an out-of-range index; here that
    // is reported through `err` (kind IndexOutOfRange) and an empty model returned.
    Model build(ParseError *err) const;

private:
    struct IBTriangle { IndexBundle index_bundles[3]; uint32_t mat_idx; };
    std::vector<IBTriangle> indices_;
    std::vector<Vec3> positions_;
    std::vector<Vec2> tex_coords_;
    std::vector<Vec3> normals_;
    std::vector<std::string> material_libs_;
    std::unordered_map<std::string, uint32_t> materials_;
    uint32_t current_mat_ = 0;
};

}  // namespace host
}  // namespace rtamd
