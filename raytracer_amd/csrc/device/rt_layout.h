// Device-side scene layout (HBM) — built once per rtamd_upload_scene() from the
// reference-layout slices (36-byte nodes, 16-byte index triangles, packed float
// positions; include/rtamd.h) and read by every traversal kernel.
//
// WideNode (64 B, 16-B aligned, one per INNER node of the reference tree)
//   The reference visits an inner node by loading it (36 B) and then both of its
//   children (2 x 36 B, each 9 scalar loads at a 36-byte stride) just to test
//   their boxes (ray_trace_and_shade.frag:120,139-140).  Here the parent record
//   carries both child boxes and both child references, so one visit is four
//   coalescable 16-byte loads from one 64-byte line, and leaf children never need
//   a record of their own.  The tree stays binary, so the visit order — hence the
//   result, including ties — is exactly the reference's.
//     q0 = (a.minx, a.miny, a.minz, a.maxx)
//     q1 = (a.maxy, a.maxz, b.minx, b.miny)
//     q2 = (b.minz, b.maxx, b.maxy, b.maxz)
//     q3 = (ref_a, ref_b, 0, 0)
//
// Child reference (32 bit)
//   bit 31 = 0: inner node, bits 30..0 = WideNode index
//   bit 31 = 1: leaf, bits 30..27 = triangle count 1..15, bits 26..0 = first triangle;
//               count field 0 = "big leaf": bits 26..0 index BigLeaf {first, count}
//
// TriRecord (48 B, 16-B aligned, one per triangle, in the reference's leaf order so
//   index i here IS the reference's triangle index i)
//   The reference reaches a triangle's vertices through two dependent hops
//   (triangles[i] -> 3 indices -> 9 scalar position loads).  The record holds
//   p0 and the two edges e1 = p1 - p0, e2 = p2 - p0 (the very subtractions the
//   shader performs first, ray_trace_and_shade.frag:66-67, so precomputing them
//   is bit-safe): three 16-byte loads, no indirection.
//     r0 = (p0.x, p0.y, p0.z, e1.x)
//     r1 = (e1.y, e1.z, e2.x, e2.y)
//     r2 = (e2.z, 0, 0, 0)
#pragma once
#include <stdint.h>

namespace rtamd {

struct alignas(16) WideNode {
    float a_min[3];
    float a_max[3];
    float b_min[3];
    float b_max[3];
    uint32_t ref_a, ref_b;
    uint32_t pad[2];
};
static_assert(sizeof(WideNode) == 64, "WideNode must be one 64-byte record");

struct alignas(16) TriRecord {
    float p0[3];
    float e1[3];
    float e2[3];
    float pad[3];
};
static_assert(sizeof(TriRecord) == 48, "TriRecord must be 48 bytes");

struct BigLeaf { uint32_t first, count; };

constexpr uint32_t REF_LEAF_BIT = 0x80000000u;
constexpr uint32_t REF_COUNT_SHIFT = 27;
constexpr uint32_t REF_FIRST_MASK = (1u << REF_COUNT_SHIFT) - 1u;
constexpr uint32_t REF_INLINE_MAX_COUNT = 15;
constexpr uint32_t REF_NONE = 0x7FFFFFFFu;  // "no node": an inner index that cannot exist
constexpr uint32_t MAX_TRIANGLES = 1u << REF_COUNT_SHIFT;

// Traversal stack: LDS_STACK_ENTRIES per lane in LDS, the rest spills (k_static: to
// scratch; k_persistent: to a per-wave-slot HBM buffer owned by the context).  The
// reference's shader has 100 entries in all (ray_trace_and_shade.frag:12); upload
// rejects trees whose deferred-sibling depth exceeds MAX_STACK_ENTRIES.
#ifndef RTAMD_LDS_STACK
#define RTAMD_LDS_STACK 16
#endif
constexpr int LDS_STACK_ENTRIES = RTAMD_LDS_STACK;
constexpr int MAX_STACK_ENTRIES = 128;
constexpr int SPILL_STACK_ENTRIES = MAX_STACK_ENTRIES - LDS_STACK_ENTRIES;

constexpr uint32_t RTAMD_TILE_PIXELS = 1024u;  // 32 x 32 (RTAMD_TILE_DIM in include/rtamd.h)

// 36-byte reference node / 16-byte reference triangle as uploaded.
struct RefNode { float bounds[6]; uint32_t is_leaf, a, b; };
struct RefTriangle { uint32_t p0, p1, p2, mat_idx; };
static_assert(sizeof(RefNode) == 36 && sizeof(RefTriangle) == 16, "wire format");

}  // namespace rtamd
