// POD scene types shared by the host layer and the device library.
//
// Mirrors the reference's wire structs one-to-one (same field order, same sizes),
// because the arrays built from them are what crosses the C-ABI boundary
// (include/rtamd.h: rtamd_scene_desc):
//   Triangle     src/raytracing/types.rs:4-15     16 bytes
//   AABB         src/raytracing/types.rs:17-42    24 bytes
//   AABBBuilder  src/raytracing/types.rs:59-99    Option<AABB>
//   Bin          src/raytracing/types.rs:44-57
//   BVHNode      src/raytracing/types.rs:101-145  36 bytes
//   IndexBundle  src/raytracing/types.rs:147-181
// All float arithmetic here is plain IEEE binary32 with no contraction
// (the library is built with -ffp-contract=off), as rustc emits it.
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

namespace rtamd {
namespace host {

struct Vec2 { float x, y; };
struct Vec3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }

// f32::min / f32::max (Rust): IEEE minNum/maxNum — a NaN operand is ignored.
inline float f32_min(float a, float b) { return std::fmin(a, b); }
inline float f32_max(float a, float b) { return std::fmax(a, b); }

struct Triangle {
    uint32_t p0, p1, p2, mat_idx;
};
static_assert(sizeof(Triangle) == 16, "wire format");

struct AABB {
    Vec3 min, max;
    void include(const Vec3 &p) {  // types.rs:28-36
        min.x = f32_min(min.x, p.x); min.y = f32_min(min.y, p.y); min.z = f32_min(min.z, p.z);
        max.x = f32_max(max.x, p.x); max.y = f32_max(max.y, p.y); max.z = f32_max(max.z, p.z);
    }
    float area() const {  // types.rs:38-41 (half surface area, left-to-right sums)
        Vec3 e = max - min;
        return e.x * e.y + e.y * e.z + e.z * e.x;
    }
};
static_assert(sizeof(AABB) == 24, "wire format");

struct AABBBuilder {  // types.rs:59-99
    bool some = false;
    AABB aabb{{0, 0, 0}, {0, 0, 0}};
    void include(const Vec3 &p) {
        if (!some) { aabb = AABB{p, p}; some = true; }
        else aabb.include(p);
    }
    void include_other(const AABBBuilder &o) {  // includes the other box's two corner points
        if (o.some) { include(o.aabb.min); include(o.aabb.max); }
    }
    AABB build() const { return some ? aabb : AABB{{0, 0, 0}, {0, 0, 0}}; }
    float area() const { return some ? aabb.area() : 0.0f; }
};

struct Bin {  // types.rs:44-57
    AABBBuilder bounds;
    uint32_t tri_count = 0;
};

struct BVHNode {  // types.rs:101-145
    AABB bounds;
    uint32_t is_leaf;
    uint32_t a;  // inner: first child ("right_node")   leaf: first_triangle
    uint32_t b;  // inner: second child ("left_node")   leaf: triangle_count
    static BVHNode new_leaf(const AABB &bounds, uint32_t first, uint32_t count) { return {bounds, 1u, first, count}; }
    static BVHNode new_dummy() { return {AABB{{0, 0, 0}, {0, 0, 0}}, 1u, 0u, 0u}; }
    void convert_to_node(uint32_t right_node, uint32_t left_node) { is_leaf = 0; a = right_node; b = left_node; }
    uint32_t first_triangle() const { return a; }
    uint32_t triangle_count() const { return b; }
};
static_assert(sizeof(BVHNode) == 36, "wire format");

struct IndexBundle {  // types.rs:147-181
    int32_t pos_idx = 0, tex_idx = 0, nor_idx = 0;
    void normalize(int32_t pos_len, int32_t tex_len, int32_t nor_len) {
        if (pos_idx < 0) pos_idx += pos_len;
        if (tex_idx < 0) tex_idx += tex_len;
        if (nor_idx < 0) nor_idx += nor_len;
    }
    bool operator==(const IndexBundle &o) const { return pos_idx == o.pos_idx && tex_idx == o.tex_idx && nor_idx == o.nor_idx; }
};

// Error convention: mirrors util/error.rs:17-33 as (kind, message, line).
enum class ParseErrorKind { None = 0, ParseIntError, ParseFloatError, InvalidLineArgCount, IndexOutOfRange, NoMaterialNamed };
struct ParseError {
    ParseErrorKind kind = ParseErrorKind::None;
    std::string detail;
    uint32_t line = 0;  // 1-based line in the file, as resource_parser.rs:16 counts
    explicit operator bool() const { return kind != ParseErrorKind::None; }
};

}  // namespace host
}  // namespace rtamd
