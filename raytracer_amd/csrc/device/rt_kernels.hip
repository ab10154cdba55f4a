// Hand-written gfx950 (CDNA4, wave64) kernels for the ray-tracing hot path.
//
// What is implemented (reference file:line each device function follows):
//   aabb_dist()         intersectAABB       res/shaders/ray_trace_and_shade.frag:53-61
//   tri_test()          intersectTriangle   ray_trace_and_shade.frag:63-87
//   traversal loops     traverseBVH         ray_trace_and_shade.frag:112-151
//                                           (= ray_trace/ray_trace.frag:104-129)
//   shade_primary()     main (hit branch)   ray_trace_and_shade.frag:206-231
//   skybox()            skybox + constants  ray_trace_and_shade.frag:180-193
//   gen_primary_dir()   ray_create.frag:12 with util/quad-11.vert:8
//   sample_albedo()     texture() on RGBA8/LINEAR/REPEAT (src/gl_wrapper/texture.rs:11-42)
//
// Numerical contract (bit-exact hit records): the file is compiled with
// -ffp-contract=off (no FMA), correctly rounded divide/sqrt and f32 denormals
// flushed, and dot products are summed z,y,x — the arithmetic Mesa llvmpipe
// executes for the unmodified shader (the pinned reference).
// The traversal keeps the reference's binary tree and visiting order (near child
// first, child b first on ties, current-t culling, ascending triangles, strict <).
//
// Execution model: one wave64 = 64 independent rays.  k_persistent is the
// production kernel: waves stay resident, pull 8x8-pixel work items from atomic
// queues (one queue per XCD-sized slice of the item list, with stealing) and
// re-fill idle lanes (ballot + mbcnt compaction of the pending-pixel pool) when
// fewer than `dyn_thresh` lanes are still traversing.  k_static (one lane per
// pixel, no re-fill) is kept as the A/B baseline.  Per-lane traversal stacks keep
// their top LDS_STACK_ENTRIES entries in an LDS ring ([entry][thread], conflict-free
// for ds_read/write_b32); older entries move to per-lane HBM columns (k_static:
// LDS + private scratch).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "rt_layout.h"

namespace rtamd {

constexpr float MISS = 1e30f;          // ray_trace_and_shade.frag:10
constexpr float EPSILON = 0.000001f;   // :11
constexpr float T_PRIMARY0 = 1000000.0f;  // :199,211
constexpr float T_HIT_LIMIT = 1000.0f;    // :206,219
#ifndef RTAMD_BLOCK
#define RTAMD_BLOCK 256
#endif
constexpr int BLOCK = RTAMD_BLOCK;   // threads per workgroup of the traversal kernels (multiple of 64)
// k_persistent is latency-bound: occupancy matters.  8 waves per SIMD (<= 64 VGPRs; the
// register spills this forces sit in the shading code and in the rare stack-overflow
// branches, not in the visit loop) measured, with four frames in flight, 3 % faster than 7
// (<= 72 VGPRs) and 10 % faster than 6 (<= 80 VGPRs) on coherent rays, 15 % faster on shuffled
// soup rays.
#ifndef RTAMD_WAVES_PER_EU
#define RTAMD_WAVES_PER_EU 8
#endif
#define RTAMD_PERSISTENT_ATTR __attribute__((amdgpu_waves_per_eu(RTAMD_WAVES_PER_EU, RTAMD_WAVES_PER_EU)))

// Per-material shading parameters: the MTL quantities of src/rendering/material.rs:5-16 the fused
// shader has a use for (it hard-codes AMBIENT / DIFFUSE / SPECULAR / SPEC_POW, ray_trace_and_shade.frag:184-187).
struct MaterialDev {
    float ka[3], kd[3], ks[3];    // ambient_color, diffuse_color, specular_color
    float ns;                     // specular_exp
    uint32_t tex_slot;            // albedo texture slot (diffuse_tex); RTAMD_MAX_TEXTURE_SLOTS = the 1x1 white texel
    uint32_t pad;
};
static_assert(sizeof(MaterialDev) == 48, "MaterialDev must be 48 bytes");
struct TexSlotDev { const uint32_t *texels; int32_t w, h; };

struct SceneDev {
    const WideNode *nodes;
    const TriRecord *tris;        // == (char*)nodes + tri_byte_offset (same allocation)
    uint32_t tri_byte_offset;
    const BigLeaf *big;
    const RefTriangle *tri_idx;   // reference 16-byte records (vertex indices + mat_idx)
    const float *normals;         // 3 per vertex or nullptr
    const float *texcoords;       // 2 per vertex or nullptr
    const uint32_t *tex[2];       // RGBA8 texels, row-major, first row first
    int32_t tex_w[2], tex_h[2];
    // optional per-material shading (rtamd_upload_materials): nullptr = the
    // shader's hard-coded constants and modelTextures[2]
    const MaterialDev *materials;
    const TexSlotDev *tex_slots;  // RTAMD_MAX_TEXTURE_SLOTS + 1 entries; a slot never uploaded, and the last, = the 1x1 white texel
    float root_box[6];
    uint32_t root_ref;
    // 1: every reachable box has finite bounds with min <= max on every axis, the precondition under which slab_sel<OCT> is
    // bit-equal to slab (rtamd_upload_scene checks; the reference builder never emits another kind, a C-ABI caller may).
    // 0: every visit takes the general slab test.
    uint32_t octant_ok;
};

// One frame of a batch (rtamd_render_batch_device): its rays or camera in, its pixels out.
constexpr int MAX_BATCH_FRAMES = 16;          // RTAMD_MAX_BATCH_FRAMES
constexpr uint32_t BATCH_FRAME_SHIFT = 28;    // a lane's out_index carries its frame in the top bits: pixels per frame < 2^28
struct FrameIO {
    const float *dir, *org;       // nullptr dir => generate from the camera below
    float *rgba;
    uint32_t *hit_id;
    float *hit_t;
    uint32_t *counts;
    float ipv[16];
    float near_, far_;
    float cam[3];
    uint32_t n_rays;              // trace batch: rays of this job (dir / org = its ray list, rgba = its hit records); else 0
};
static_assert(sizeof(FrameIO) == 136, "FrameIO layout");
static_assert(MAX_BATCH_FRAMES <= (1 << (32 - BATCH_FRAME_SHIFT)), "the frame index must fit above BATCH_FRAME_SHIFT");

struct JobParams {
    // fused render
    uint32_t width, height;
    const float *dir, *org;       // nullptr dir => generate from camera
    float ipv[16];
    float near_, far_;
    float cam[3];
    uint32_t has_tc, has_n;
    uint32_t rank, world;
    uint32_t tiles_x;
    float *rgba;
    uint32_t *hit_id;
    float *hit_t;
    uint32_t *counts;
    // trace
    uint64_t n_rays;
    float *hits;
    // scheduling
    uint32_t n_items;             // work items of 64 pixels / rays
    uint32_t *queues;             // n_queues counters, 16 dwords apart
    uint32_t n_queues;
    uint32_t dyn_thresh;
    uint32_t leaf_thresh;         // lanes waiting at a leaf that release the leaf phase, while the wave runs without lane re-fill (coherent rays) ...
    uint32_t leaf_thresh_refill;  // ... and while it re-fills (incoherent rays)
    uint32_t shadow_anyhit;       // fused mode: shadow rays stop at the first leaf that leaves t < T_HIT_LIMIT
    uint32_t tex_filter8;         // 1: filter textures as the pinned reference executor does (8-bit weights and lerps), 0: in float
    uint32_t queue_stride;        // u32 words between queue counters
    // coherence feedback (dyn_thresh auto): {wave-uniform visits, visits, active lanes, threshold used} over a sample of the waves;
    // stats_prev = what the previous launch of this mode on this stream measured (nullptr: none yet)
    const uint32_t *stats_prev;
    uint32_t *stats_cur;
    uint32_t dyn_auto;            // 1: pick the lane re-fill threshold from stats_prev (coherent: 1, incoherent: dyn_incoherent); batches of ray lists: per wave and list
    uint32_t dyn_incoherent;
    uint32_t queue_block;         // consecutive schedule positions a queue owns per turn
    uint32_t *spill;              // k_persistent: [wave slot][spill_wave_words / 64 entries][64] stack overflow area
    uint32_t spill_wave_words;    // u32 words per wave slot of `spill` (SPILL_STACK_ENTRIES * 64 for the context's own areas)
    // batch of frames in ONE launch (k_persistent, fused mode): item i belongs to frame i / items_per_frame; frame f's rays,
    // camera and outputs are frames[f] (the fields above are not used then).  n_frames == 1: the fields above, as ever.
    // batch of ray lists in ONE launch (trace mode): job j owns items job_item_begin[j] .. job_item_begin[j + 1] - 1; its ray list
    // and hit records are frames[j].{org, dir, rgba}, its length frames[j].n_rays (< 2^28: a lane carries the job like a frame)
    uint32_t n_frames;
    uint32_t items_per_frame;
    uint32_t job_item_begin[MAX_BATCH_FRAMES + 1];
    FrameIO frames[MAX_BATCH_FRAMES];
#ifdef RTAMD_EXP_STAMPS
    unsigned long long *stamps;   // diagnostic build only: per wave {start, end, items, xcc} in 100 MHz ticks
#endif
};

// ---------------------------------------------------------------------------
// arithmetic helpers (evaluation order fixed, see header comment)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float dot_zyx(float ax, float ay, float az, float bx, float by, float bz) {
    return (az * bz + ay * by) + ax * bx;
}

struct Ray { float ox, oy, oz, dx, dy, dz, rx, ry, rz; };

__device__ __forceinline__ void set_ray(Ray &r, float ox, float oy, float oz, float dx, float dy, float dz) {
    r.ox = ox; r.oy = oy; r.oz = oz;
    r.dx = dx; r.dy = dy; r.dz = dz;
    r.rx = 1.0f / dx; r.ry = 1.0f / dy; r.rz = 1.0f / dz;  // `1 / dir`, :204
}

// intersectAABB: tmin, or MISS when !(tmax >= tmin && tmin < t && tmax > 0)
__device__ __forceinline__ float aabb_dist(const Ray &r, float minx, float miny, float minz, float maxx, float maxy,
                                           float maxz, float t) {
    float tx1 = (minx - r.ox) * r.rx, tx2 = (maxx - r.ox) * r.rx;
    float tmin = fminf(tx1, tx2), tmax = fmaxf(tx1, tx2);
    float ty1 = (miny - r.oy) * r.ry, ty2 = (maxy - r.oy) * r.ry;
    tmin = fmaxf(tmin, fminf(ty1, ty2)); tmax = fminf(tmax, fmaxf(ty1, ty2));
    float tz1 = (minz - r.oz) * r.rz, tz2 = (maxz - r.oz) * r.rz;
    tmin = fmaxf(tmin, fminf(tz1, tz2)); tmax = fminf(tmax, fmaxf(tz1, tz2));
    return (tmax >= tmin && tmin < t && tmax > 0.0f) ? tmin : MISS;
}

// intersectTriangle on a precomputed-edge record, evaluated branch-free: every
// reject test of :72,78,83 and the caller's accept test :131 are folded into one
// predicate, so lanes of a wave never diverge inside a triangle test.
__device__ __forceinline__ void tri_test(const Ray &r, const TriRecord *__restrict__ rec, uint32_t i, float &best_t,
                                         uint32_t &best_idx, float &best_u, float &best_v) {
    const char *rp = reinterpret_cast<const char *>(rec) + i * (uint32_t)sizeof(TriRecord);   // 32-bit offset, as for the nodes
    const float4 r0 = *reinterpret_cast<const float4 *>(rp);
    const float4 r1 = *reinterpret_cast<const float4 *>(rp + 16);
    const float r2 = *reinterpret_cast<const float *>(rp + 32);
    const float p0x = r0.x, p0y = r0.y, p0z = r0.z;
    const float e1x = r0.w, e1y = r1.x, e1z = r1.y;
    const float e2x = r1.z, e2y = r1.w, e2z = r2;
    // h = cross(dir, edge2)
    float hx = r.dy * e2z - e2y * r.dz;
    float hy = r.dz * e2x - e2z * r.dx;
    float hz = r.dx * e2y - e2x * r.dy;
    float a = dot_zyx(e1x, e1y, e1z, hx, hy, hz);
    bool ok = !(fabsf(a) < EPSILON);
    float f = 1.0f / a;
    float sx = r.ox - p0x, sy = r.oy - p0y, sz = r.oz - p0z;
    float du = dot_zyx(sx, sy, sz, hx, hy, hz);
    float u = f * du;
    ok = ok && !(u < 0.0f || u > 1.0f);
    // q = cross(s, edge1)
    float qx = sy * e1z - e1y * sz;
    float qy = sz * e1x - e1z * sx;
    float qz = sx * e1y - e1x * sy;
    float dv = dot_zyx(r.dx, r.dy, r.dz, qx, qy, qz);
    float v = f * dv;
    // :83 as the reference's compiler evaluates it: f*du + f*dv is factored to f*(du + dv), which decides
    // differently within an ulp of the edge u + v = 1 (pinned by tests/golden/golden_edge.npz)
    ok = ok && !(v < 0.0f || f * (du + dv) > 1.0f);
    float t = f * dot_zyx(e2x, e2y, e2z, qx, qy, qz);
    ok = ok && (t > EPSILON && t < best_t);
    if (ok) { best_t = t; best_idx = i; best_u = u; best_v = v; }
}

// ---------------------------------------------------------------------------
// per-lane traversal state
// ---------------------------------------------------------------------------
struct Trav {
    Ray ray;
    float t;
    uint32_t idx;
    float u, v;
    uint32_t cur;   // node reference being visited, REF_NONE when the traversal is finished
    uint32_t sp;    // deferred references on this lane's stack (Stack policy's representation)
    uint32_t sb;    // StackGlobal: entries moved out to HBM (same representation); else unused
};

// Stack policies.  Entry e of thread tid lives at lds[e*BLOCK + tid] (ds_read/write_b32:
// bank = tid % 32, conflict-free whatever the depths of the lanes).  Deeper entries
// spill: StackScratch to a private (scratch) array, StackGlobal to the wave slot's
// slice of an HBM buffer laid out [entry][lane] so that a spill is one coalesced line.
// `sp` / `sb` are opaque to the traversal code.  StackScratch counts entries (sb unused).
// StackGlobal keeps the TOP LDS_STACK_ENTRIES entries of a lane's stack in LDS, as a ring: sp is
// depth*STEP + tid*4, whose low bits (sp & RING_MASK) are the ds_read/ds_write address of the next
// free slot; sb = (number of bottom entries moved out to HBM)*STEP.  A push into a full ring first
// moves the oldest ring entry to the lane's HBM column; a pop from an empty ring reads the HBM
// column directly.  Deep trees (triangle soup: 30+ levels) thus keep their hot entries in LDS —
// with a fixed bottom-16 layout the 1 M-triangle soup was 17 % slower than with 24 LDS entries.
struct StackScratch {
    uint32_t *lds;
    uint32_t spill[SPILL_STACK_ENTRIES];
    __device__ __forceinline__ uint32_t sp_init(uint32_t) const { return 0u; }
    __device__ __forceinline__ bool empty(uint32_t sp) const { return sp == 0u; }
    __device__ __forceinline__ void push(uint32_t tid, uint32_t &sp, uint32_t &, uint32_t v) {
        if (__builtin_expect(sp < (uint32_t)LDS_STACK_ENTRIES, 1)) lds[sp * BLOCK + tid] = v;
        else if (sp < (uint32_t)MAX_STACK_ENTRIES) spill[sp - LDS_STACK_ENTRIES] = v;
        sp++;
    }
    __device__ __forceinline__ uint32_t pop(uint32_t tid, uint32_t &sp, uint32_t &) const {
        sp--;
        if (__builtin_expect(sp < (uint32_t)LDS_STACK_ENTRIES, 1)) return lds[sp * BLOCK + tid];
        if (sp < (uint32_t)MAX_STACK_ENTRIES) return spill[sp - LDS_STACK_ENTRIES];
        return REF_NONE;
    }
};
struct StackGlobal {
    uint32_t *lds;
    // HBM columns: entry e of lane l of the workgroup's wave w sits at spill_base[(w * wave_words) + e * 64 + l].
    // Only wave-uniform values are kept: the lane's own address is rebuilt from sp's low bits (tid * 4) in the rare
    // branches that need it, so the column costs no VGPR in the visit loop (a per-lane 64-bit pointer cost two).
    uint32_t *spill_base;     // the workgroup's first wave slot
    uint32_t wave_words;
    __device__ __forceinline__ uint32_t *column(uint32_t sp) const {
        const uint32_t t = (sp & (STEP - 1u)) >> 2;     // threadIdx.x
        return spill_base + (size_t)(t >> 6) * wave_words + (t & 63u);
    }
    static constexpr uint32_t STEP = BLOCK * 4u;
    static constexpr uint32_t RING = LDS_STACK_ENTRIES * STEP;
    static_assert((LDS_STACK_ENTRIES & (LDS_STACK_ENTRIES - 1)) == 0, "ring addressing needs a power-of-two entry count");
    __device__ __forceinline__ uint32_t lds_word(uint32_t off) const { return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lds) + (off & (RING - 1u))); }
    __device__ __forceinline__ void lds_word(uint32_t off, uint32_t v) const { *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(lds) + (off & (RING - 1u))) = v; }
    __device__ __forceinline__ uint32_t sp_init(uint32_t tid) const { return tid * 4u; }
    __device__ __forceinline__ bool empty(uint32_t sp) const { return sp < STEP; }
    __device__ __forceinline__ void push(uint32_t, uint32_t &sp, uint32_t &sb, uint32_t v) {
        if (__builtin_expect(sp - sb >= RING, 0)) {   // ring full: its oldest entry goes to HBM
            const uint32_t e = sb / STEP;
            if (e < (wave_words >> 6)) column(sp)[e * 64u] = lds_word(sp);   // slot (sp & mask) holds entry sb; the column has wave_words / 64 entries
            sb += STEP;
        }
        lds_word(sp, v);
        sp += STEP;
    }
    __device__ __forceinline__ uint32_t pop(uint32_t, uint32_t &sp, uint32_t &sb) const {
        sp -= STEP;
        if (__builtin_expect(sp >= sb, 1)) return lds_word(sp);   // sp - sb carries + tid*4, so >= means "entry sp is in the ring"
        sb -= STEP;                                               // ring empty: the top entry is the last one moved out
        const uint32_t e = sb / STEP;
        return e < (wave_words >> 6) ? column(sp)[e * 64u] : REF_NONE;
    }
};

// Start a traversal: the root box is tested once (:115-117).
template <bool COUNT>
__device__ __forceinline__ void trav_begin(const SceneDev &s, Trav &tr, float t0, uint32_t sp0, uint32_t &nbox) {
    tr.t = t0; tr.idx = 0; tr.u = 0.0f; tr.v = 0.0f; tr.sp = sp0; tr.sb = 0;
    if (COUNT) nbox++;
    float d = aabb_dist(tr.ray, s.root_box[0], s.root_box[1], s.root_box[2], s.root_box[3], s.root_box[4], s.root_box[5], tr.t);
    tr.cur = (d == MISS) ? REF_NONE : s.root_ref;
}

// Slab test of intersectAABB (:53-61) split into its interval and its accept
// predicate `tmax >= tmin && tmin < t && tmax > 0`; the shader's "return MISS"
// encoding of a rejected box is only ever compared (!= 1e30, dist0 < dist1), which
// trav_inner evaluates on the predicates directly.
__device__ __forceinline__ bool slab(const Ray &r, float minx, float miny, float minz, float maxx, float maxy, float maxz, float t, float &tmin) {
    float tx1 = (minx - r.ox) * r.rx, tx2 = (maxx - r.ox) * r.rx;
    float lo = fminf(tx1, tx2), hi = fmaxf(tx1, tx2);
    float ty1 = (miny - r.oy) * r.ry, ty2 = (maxy - r.oy) * r.ry;
    lo = fmaxf(lo, fminf(ty1, ty2)); hi = fminf(hi, fmaxf(ty1, ty2));
    float tz1 = (minz - r.oz) * r.rz, tz2 = (maxz - r.oz) * r.rz;
    lo = fmaxf(lo, fminf(tz1, tz2)); hi = fminf(hi, fmaxf(tz1, tz2));
    tmin = lo;
    return hi >= lo && lo < t && hi > 0.0f;
}

// Slab test for a ray whose direction octant is known: OCT bit k set = component k is
// negative; all three inverses finite and non-zero.  (lo - o) * r <= (hi - o) * r holds in
// float for r > 0 (subtraction and multiplication by a positive number are monotonic) and
// the reverse for r < 0, so min(t1,t2) / max(t1,t2) of `slab` are known operands, bit for
// bit, and the six min/max per box disappear -- for a box with finite bounds and min <= max on every axis
// (SceneDev::octant_ok; an inverted or NaN box decides differently under the two forms).  OCT < 0: the general test.
template <int OCT>
__device__ __forceinline__ bool slab_sel(const Ray &r, float minx, float miny, float minz, float maxx, float maxy, float maxz, float t, float &tmin) {
    if (OCT < 0) return slab(r, minx, miny, minz, maxx, maxy, maxz, t, tmin);
    const float nx = (OCT & 1) ? maxx : minx, fx = (OCT & 1) ? minx : maxx;
    const float ny = (OCT & 2) ? maxy : miny, fy = (OCT & 2) ? miny : maxy;
    const float nz = (OCT & 4) ? maxz : minz, fz = (OCT & 4) ? minz : maxz;
    float lo = fmaxf(fmaxf((nx - r.ox) * r.rx, (ny - r.oy) * r.ry), (nz - r.oz) * r.rz);
    float hi = fminf(fminf((fx - r.ox) * r.rx, (fy - r.oy) * r.ry), (fz - r.oz) * r.rz);
    tmin = lo;
    return hi >= lo && lo < t && hi > 0.0f;
}

// One inner-node visit (:138-148): test both children with the current t, descend
// into the nearer hit child, defer the other.  With dist = hit ? tmin : 1e30 the
// shader's `dist0 < dist1` is  hit0 && (!hit1 || tmin0 < tmin1)  (a hit has
// tmin < t <= 1e30), so equal distances visit child b first, exactly as :142-148.
// Child selection + stack update shared by the two fetch paths of trav_inner.
template <class Stack>
__device__ __forceinline__ void trav_select(Trav &tr, Stack &st, uint32_t tid, bool hit0, bool hit1, float tmin0, float tmin1,
                                            uint32_t ref_a, uint32_t ref_b) {
    const bool a_first = hit0 && (!hit1 || tmin0 < tmin1);
    const uint32_t near_ref = a_first ? ref_a : ref_b;
    const uint32_t far_ref = a_first ? ref_b : ref_a;
    if (hit0 && hit1) st.push(tid, tr.sp, tr.sb, far_ref);
    if (hit0 || hit1) tr.cur = near_ref;
    else tr.cur = st.empty(tr.sp) ? REF_NONE : st.pop(tid, tr.sp, tr.sb);
}

// (Touch-prefetch — one dword load from each child's record while the box tests run — was measured and
// removed: the two extra divergent loads per visit cost more L1 accesses than the latency they hide,
// k_static 0.59 -> 0.90 ms, k_persistent 0.55 -> 0.61 ms; re-measured later: 18 % slower.)

// Vector fetch: every lane loads its own 64-byte record (4 x global_load_dwordx4).
template <bool COUNT, int OCT = -1, class Stack>
__device__ __forceinline__ void trav_inner(const SceneDev &s, Trav &tr, Stack &st, uint32_t tid, uint32_t &nbox) {
    // 32-bit byte offset from the (scalar) base: the loads take the base in SGPRs and one VGPR offset,
    // instead of a 64-bit per-lane address (upload checks that nodes + triangle records stay below 4 GiB)
    const float4 *np = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s.nodes) + (tr.cur << 6));
    const float4 q0 = np[0], q1 = np[1], q2 = np[2];
    const uint2 refs = *reinterpret_cast<const uint2 *>(&np[3]);
    if (COUNT) nbox += 2;
    float tmin0, tmin1;
    const bool hit0 = slab_sel<OCT>(tr.ray, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tr.t, tmin0);
    const bool hit1 = slab_sel<OCT>(tr.ray, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tr.t, tmin1);
    trav_select(tr, st, tid, hit0, hit1, tmin0, tmin1, refs.x, refs.y);
}

// Scalar fetch: all active lanes of the wave visit the SAME node (coherent rays near
// the top of the tree: about half of all visits of a camera-ray wave).  The record
// is read once through the scalar cache (s_load_dwordx16) into SGPRs, which the slab
// arithmetic takes as scalar operands: no texture-address/L1 traffic, no VGPRs.
typedef const __attribute__((address_space(4))) uint32_t *const_u32_ptr;
template <bool COUNT, int OCT = -1, class Stack>
__device__ __forceinline__ void trav_inner_uniform(const SceneDev &s, Trav &tr, Stack &st, uint32_t tid, uint32_t node, uint32_t &nbox) {
    const_u32_ptr q = (const_u32_ptr)(uintptr_t)(s.nodes + node);
    uint32_t w[14];
#pragma unroll
    for (int k = 0; k < 14; k++) w[k] = q[k];
    if (COUNT) nbox += 2;
    float tmin0, tmin1;
    const bool hit0 = slab_sel<OCT>(tr.ray, __uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3]),
                                    __uint_as_float(w[4]), __uint_as_float(w[5]), tr.t, tmin0);
    const bool hit1 = slab_sel<OCT>(tr.ray, __uint_as_float(w[6]), __uint_as_float(w[7]), __uint_as_float(w[8]), __uint_as_float(w[9]),
                                    __uint_as_float(w[10]), __uint_as_float(w[11]), tr.t, tmin1);
    trav_select(tr, st, tid, hit0, hit1, tmin0, tmin1, w[12], w[13]);
}

// One inner step for the lanes in `in` (ballot `m_in`): scalar path when they agree.
template <bool COUNT, int OCT = -1, class Stack>
__device__ __forceinline__ bool trav_inner_step(const SceneDev &s, Trav &tr, Stack &st, uint32_t tid, bool in, unsigned long long m_in, uint32_t &nbox) {
    const uint32_t c0 = __builtin_amdgcn_readlane(tr.cur, (int)__builtin_ctzll(m_in));   // node of the first lane in m_in
    const bool uniform = __builtin_amdgcn_ballot_w64(tr.cur == c0) == m_in;   // c0 is an inner ref: equal lanes are in m_in
    if (uniform) { if (in) trav_inner_uniform<COUNT, OCT>(s, tr, st, tid, c0, nbox); }
    else if (in) trav_inner<COUNT, OCT>(s, tr, st, tid, nbox);
    return uniform;
}

// One leaf visit (:122-136): all triangles in ascending order, then pop.
// `occlusion`: this lane's ray is a shadow ray whose caller consumes only the predicate
// `t < T_HIT_LIMIT` (:219); once a leaf leaves t below the limit the rest of the traversal
// can only lower t further, so the lane stops there.  The predicate, hence every output,
// is the one the full traversal gives; only the box/triangle test counters differ.
template <bool COUNT, class Stack>
__device__ __forceinline__ void trav_leaf(const SceneDev &s, Trav &tr, const Stack &st, uint32_t tid, uint32_t &ntri, bool occlusion = false) {
    uint32_t ref = tr.cur;
    uint32_t count = (ref >> REF_COUNT_SHIFT) & 15u;
    uint32_t first = ref & REF_FIRST_MASK;
    if (count == 0) {
        BigLeaf bl = s.big[first];
        first = bl.first; count = bl.count;
    }
    if (COUNT) ntri += count;
    for (uint32_t i = first; i < first + count; i++) tri_test(tr.ray, s.tris, i, tr.t, tr.idx, tr.u, tr.v);
    if (occlusion && tr.t < T_HIT_LIMIT) { tr.cur = REF_NONE; return; }
    tr.cur = st.empty(tr.sp) ? REF_NONE : st.pop(tid, tr.sp, tr.sb);
}

__device__ __forceinline__ bool ref_is_inner(uint32_t ref) { return (ref & REF_LEAF_BIT) == 0 && ref != REF_NONE; }

// Run this lane's traversal to completion (static kernel).
template <bool COUNT, class Stack>
__device__ __forceinline__ void trav_run(const SceneDev &s, Trav &tr, Stack &st, uint32_t tid, uint32_t &nbox, uint32_t &ntri, bool occlusion = false) {
    while (tr.cur != REF_NONE) {
        while (ref_is_inner(tr.cur)) trav_inner<COUNT>(s, tr, st, tid, nbox);
        if (tr.cur != REF_NONE) trav_leaf<COUNT>(s, tr, st, tid, ntri, occlusion);
    }
}

// ---------------------------------------------------------------------------
// shading
// ---------------------------------------------------------------------------
struct V3 { float x, y, z; };

__device__ __forceinline__ V3 sun_dir() {
    // normalize(vec3(1,2,1)) as the GLSL compiler folds it: v * (1/sqrt(6))
    const float inv = 1.0f / sqrtf(6.0f);
    return V3{1.0f * inv, 2.0f * inv, 1.0f * inv};
}

__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
// pow(x,y) is undefined in GLSL for x < 0; the pinned reference yields NaN which the
// following clamp turns into 0 — restated as "negative base => 0".
// Only the colour (tolerance 1/255) depends on it, so pow is exp2(y*log2(x)) on the
// hardware transcendental units (v_log_f32 / v_exp_f32, ~1 ulp each).
__device__ __forceinline__ float pow_clamp01(float x, float y) {
    return (x >= 0.0f) ? clamp01(__builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x))) : 0.0f;
}

__device__ __forceinline__ V3 skybox(float dx, float dy, float dz) {
    const V3 sd = sun_dir();
    float sun_fac = pow_clamp01(dot_zyx(dx, dy, dz, sd.x, sd.y, sd.z), 200.0f);
    float sky_fac = (dy + 1.0f) * 0.5f;
    float inv_fac = 1.0f - sky_fac;
    return V3{(0.5f * sky_fac + 0.2f * inv_fac) + 1.0f * sun_fac,
              (0.7f * sky_fac + 0.5f * inv_fac) + 0.97f * sun_fac,
              (0.9f * sky_fac + 0.8f * inv_fac) + 0.86f * sun_fac};
}

__device__ __forceinline__ int wrap_repeat(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }

// unorm8 -> float as the pinned reference executor converts it: x * fl(1/255), one multiplication (not the correctly rounded
// quotient, from which it differs in the last bit for about half of the 256 values; tests/test_oracle_golden.py
// test_dispatcher_ambient_directions_bit_exact pins it through ray_dispatcher.frag's noise fetch).
__device__ __forceinline__ float unorm8_to_float(uint32_t x) { return (float)x * (1.0f / 255.0f); }

// LINEAR min/mag, REPEAT, single level, RGBA8 unorm (GL 4.6 spec 8.14.2: texel
// centres at i+0.5, weights frac(u*W-0.5)).  f8 = 0: float bilinear.  f8 = 1 (option "texture_filter_8bit"): the filter as the
// pinned reference executor (Mesa llvmpipe) runs it, bit for bit on its 69 120 probed samples (tests/golden/golden_sampler.npz):
// coordinates wrapped before scaling unless both sizes are powers of two, weights rounded half to even to 8 bits, three 8-bit
// lerps v0 + (((v1 - v0) * w + 128) >> 8) (x on both rows, then y), an 8-bit result.  GL allows both; float is the more exact one.
__device__ __forceinline__ V3 sample_albedo(const uint32_t *__restrict__ tex, int W, int H, float s, float t, bool f8) {
    if (f8 && (((W & (W - 1)) | (H & (H - 1))) != 0)) {
        s = s - floorf(s); t = t - floorf(t);
        if (s >= 1.0f) s = 0.99999994f;
        if (t >= 1.0f) t = 0.99999994f;
    }
    float u = s * (float)W - 0.5f, v = t * (float)H - 0.5f;
    float fu = floorf(u), fv = floorf(v);
    float a = u - fu, b = v - fv;
    int i0 = wrap_repeat((int)fu, W), j0 = wrap_repeat((int)fv, H);
    int i1 = i0 + 1 == W ? 0 : i0 + 1, j1 = j0 + 1 == H ? 0 : j0 + 1;
    uint32_t t00 = tex[(size_t)j0 * W + i0], t10 = tex[(size_t)j0 * W + i1];
    uint32_t t01 = tex[(size_t)j1 * W + i0], t11 = tex[(size_t)j1 * W + i1];
    float c[3];
    if (f8) {
        const int wx = (int)__builtin_rintf(a * 256.0f), wy = (int)__builtin_rintf(b * 256.0f);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int c00 = (int)((t00 >> (8 * k)) & 255u), c10 = (int)((t10 >> (8 * k)) & 255u);
            const int c01 = (int)((t01 >> (8 * k)) & 255u), c11 = (int)((t11 >> (8 * k)) & 255u);
            const int top = c00 + (((c10 - c00) * wx + 128) >> 8);
            const int bot = c01 + (((c11 - c01) * wx + 128) >> 8);
            c[k] = unorm8_to_float((uint32_t)(top + (((bot - top) * wy + 128) >> 8)));
        }
        return V3{c[0], c[1], c[2]};
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float c00 = unorm8_to_float((t00 >> (8 * k)) & 255u), c10 = unorm8_to_float((t10 >> (8 * k)) & 255u);
        float c01 = unorm8_to_float((t01 >> (8 * k)) & 255u), c11 = unorm8_to_float((t11 >> (8 * k)) & 255u);
        float top = c00 + a * (c10 - c00);
        float bot = c01 + a * (c11 - c01);
        c[k] = top + b * (bot - top);
    }
    return V3{c[0], c[1], c[2]};
}

// Hit branch of main (:207-229) up to the shadow traversal: computes the shading
// normal, the two candidate colours (lit / in shadow) and the shadow ray.
// F8: how texture() filters (option "texture_filter_8bit"): 0 float bilinear, 1 the pinned executor's 8-bit filter, -1 read
// p.tex_filter8 at run time (k_static).  The production kernel takes it as a template argument, so that the default
// instantiation carries none of the 8-bit filter's code or registers.
template <bool MAT = false, int F8 = -1>
__device__ __forceinline__ void shade_primary(const SceneDev &s, const JobParams &p, const Trav &tr, V3 &col_shadow, V3 &col_lit, Ray &shadow_ray) {
    const bool f8 = F8 < 0 ? p.tex_filter8 != 0 : F8 != 0;
    const RefTriangle tri = s.tri_idx[tr.idx];
    const float u = tr.u, v = tr.v;
    const float w = 1.0f - u - v;
    float nx, ny, nz;
    if (p.has_n) {
        const float *n0 = s.normals + 3 * (size_t)tri.p0, *n1 = s.normals + 3 * (size_t)tri.p1, *n2 = s.normals + 3 * (size_t)tri.p2;
        nx = (n1[0] * u + n2[0] * v) + n0[0] * w;
        ny = (n1[1] * u + n2[1] * v) + n0[1] * w;
        nz = (n1[2] * u + n2[2] * v) + n0[2] * w;
    } else {
        const TriRecord &rec = s.tris[tr.idx];   // cross(v1 - v0, v2 - v0), :176
        nx = rec.e1[1] * rec.e2[2] - rec.e2[1] * rec.e1[2];
        ny = rec.e1[2] * rec.e2[0] - rec.e2[2] * rec.e1[0];
        nz = rec.e1[0] * rec.e2[1] - rec.e2[0] * rec.e1[1];
    }
    {   // normalize(v) = v * (1 / sqrt(dot(v,v)))
        float inv = 1.0f / sqrtf(dot_zyx(nx, ny, nz, nx, ny, nz));
        nx *= inv; ny *= inv; nz *= inv;
    }
    float tcx = 0.0f, tcy = 0.0f;
    if (p.has_tc) {
        const float *t0 = s.texcoords + 2 * (size_t)tri.p0, *t1 = s.texcoords + 2 * (size_t)tri.p1, *t2 = s.texcoords + 2 * (size_t)tri.p2;
        tcx = (t1[0] * u + t2[0] * v) + t0[0] * w;
        tcy = (t1[1] * u + t2[1] * v) + t0[1] * w;
    }
    const Ray &r = tr.ray;
    // reflect(I, N) = I - 2 * dot(N, I) * N
    float two_d = 2.0f * dot_zyx(nx, ny, nz, r.dx, r.dy, r.dz);
    float rfx = r.dx - nx * two_d, rfy = r.dy - ny * two_d, rfz = r.dz - nz * two_d;
    const V3 sd = sun_dir();
    // shadow ray (:214-217): org + dir * t + normal * 0.0001
    set_ray(shadow_ray, (r.ox + r.dx * tr.t) + nx * 0.0001f, (r.oy + r.dy * tr.t) + ny * 0.0001f, (r.oz + r.dz * tr.t) + nz * 0.0001f,
            sd.x, sd.y, sd.z);
    // AMBIENT, DIFFUSE, SPECULAR, SPEC_POW (:184-187) — or, with a material table, the triangle's
    // material: the same expressions with Ka / Kd / Ks / Ns in the constants' places, so a table
    // holding (.5, .8, .4, 30) reproduces the reference bit for bit
    float ka[3] = {0.5f, 0.5f, 0.5f}, kd[3] = {0.8f, 0.8f, 0.8f}, ks[3] = {0.4f, 0.4f, 0.4f}, ns = 30.0f;
    V3 albedo;
    if (MAT) {
        const MaterialDev &md = s.materials[tri.mat_idx];   // upload checked mat_idx < table size
#pragma unroll
        for (int k = 0; k < 3; k++) { ka[k] = md.ka[k]; kd[k] = md.kd[k]; ks[k] = md.ks[k]; }
        ns = md.ns;
        const TexSlotDev ts = s.tex_slots[md.tex_slot];     // "no texture" is the slot of the 1x1 white texel
        albedo = sample_albedo(ts.texels, ts.w, ts.h, tcx, -tcy, f8);
    } else {
        const uint32_t m = tri.mat_idx & 1u;
        albedo = sample_albedo(s.tex[m], s.tex_w[m], s.tex_h[m], tcx, -tcy, f8);
    }
    V3 sky = skybox(nx, ny, nz);
    float ndl = clamp01(dot_zyx(nx, ny, nz, sd.x, sd.y, sd.z));
    float spec = pow_clamp01(dot_zyx(rfx, rfy, rfz, sd.x, sd.y, sd.z), ns);
    V3 ambient{(albedo.x * sky.x) * ka[0], (albedo.y * sky.y) * ka[1], (albedo.z * sky.z) * ka[2]};
    V3 diffuse{((albedo.x * 1.0f) * ndl) * kd[0], ((albedo.y * 0.97f) * ndl) * kd[1], ((albedo.z * 0.86f) * ndl) * kd[2]};
    V3 specular{(1.0f * spec) * ks[0], (0.97f * spec) * ks[1], (0.86f * spec) * ks[2]};
    col_lit = V3{(ambient.x + diffuse.x) + specular.x, (ambient.y + diffuse.y) + specular.y, (ambient.z + diffuse.z) + specular.z};
    // in shadow: diffuse *= 0, specular *= 0 (:226-229)
    col_shadow = V3{(ambient.x + diffuse.x * 0.0f) + specular.x * 0.0f, (ambient.y + diffuse.y * 0.0f) + specular.y * 0.0f,
                    (ambient.z + diffuse.z * 0.0f) + specular.z * 0.0f};
}

// ray_create.frag:12 at the centre of pixel (px,py); row 0 = bottom.
// (off_x, off_y) is the sample position inside the pixel; (0.5, 0.5) = the pixel centre the reference rasterises.
__device__ __forceinline__ void gen_primary_dir(const float *__restrict__ m, float near_, float far_, uint32_t width, uint32_t height,
                                                uint32_t px, uint32_t py, float &dx, float &dy, float &dz, float off_x = 0.5f, float off_y = 0.5f) {
    float x = ((float)px + off_x) / (float)width * 2.0f - 1.0f;
    float y = ((float)py + off_y) / (float)height * 2.0f - 1.0f;
    float v0 = x * far_ - x * near_;
    float v1 = y * far_ - y * near_;
    float v2 = 1.0f * far_ - (-1.0f) * near_;
    float v3 = 1.0f * far_ - 1.0f * near_;
    float d0 = ((m[0] * v0 + m[4] * v1) + m[8] * v2) + m[12] * v3;
    float d1 = ((m[1] * v0 + m[5] * v1) + m[9] * v2) + m[13] * v3;
    float d2 = ((m[2] * v0 + m[6] * v1) + m[10] * v2) + m[14] * v3;
    float inv = 1.0f / sqrtf(dot_zyx(d0, d1, d2, d0, d1, d2));
    dx = d0 * inv; dy = d1 * inv; dz = d2 * inv;
}

// ---------------------------------------------------------------------------
// work-item geometry: item -> 8x8 pixel block of a 32x32 tile owned by this rank
// ---------------------------------------------------------------------------
struct PixelRef { uint32_t ray_index; uint32_t out_index; bool valid; uint32_t px, py; };

__device__ __forceinline__ PixelRef item_pixel(const JobParams &p, uint32_t item, uint32_t l) {
    uint32_t j = item >> 4, sub = item & 15u;
    uint32_t g = j * p.world + p.rank;
    uint32_t tx = g % p.tiles_x, ty = g / p.tiles_x;
    uint32_t x = (tx * 4u + (sub & 3u)) * 8u + (l & 7u);
    uint32_t y = (ty * 4u + (sub >> 2)) * 8u + (l >> 3);
    PixelRef r;
    r.px = x; r.py = y;
    r.valid = x < p.width && y < p.height;
    r.ray_index = y * p.width + x;
    r.out_index = (p.world == 1u) ? r.ray_index : j * (RTAMD_TILE_PIXELS) + ((y & 31u) * 32u + (x & 31u));
    return r;
}

// Pixel of entry i of a rank's buffer: scan-line order for world <= 1, else the compact tile-major layout of
// include/rtamd.h (local tile j = i / 1024 is global tile j * world + rank, row-major 32 x 32 inside the tile).
// false: the entry lies in the part of an edge tile that is outside the frame.
__device__ __forceinline__ bool shard_pixel(uint32_t i, uint32_t width, uint32_t height, uint32_t rank, uint32_t world, uint32_t tiles_x,
                                            uint32_t &px, uint32_t &py) {
    if (world <= 1u) { px = i % width; py = i / width; return true; }
    const uint32_t j = i >> 10, within = i & 1023u;
    const uint32_t g = j * world + rank;
    px = (g % tiles_x) * 32u + (within & 31u);
    py = (g / tiles_x) * 32u + (within >> 5);
    return px < width && py < height;
}

// ---------------------------------------------------------------------------
// queues: n_queues counters (queue_stride words apart) over ONE schedule, the items in
// their natural (tile) order.  A wave starts on the counter of its XCD and steals from the
// others when that one is exhausted; which wave renders which item never changes a result
// bit.  (Ordering the items by the cost the previous frame counted for them — LPT or
// heavy-first — was measured slower and removed)
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v & 7u;
}

struct QueueCursor { uint32_t tried, k, v_next, v_end; };

// Exhausted queues are found by one failing atomic each.  (Reading all counters with one agent-scope
// load per lane when a queue runs dry, to skip the failing atomics, was measured 19 % SLOWER:
// the loads contend with the atomics on the same lines.)
__device__ __forceinline__ uint32_t fetch_item(const JobParams &p, uint32_t home, QueueCursor &q, uint32_t lane) {
    for (;;) {
        if (q.v_next < q.v_end) {
            // queue k serves schedule positions k, k + nq, k + 2nq, ...: every queue sees the
            // same profile of the schedule
            // with queue_block B > 1 a queue owns runs of B consecutive positions (B items = a compact
            // screen region) instead of single positions: the waves of one XCD then share more of their nodes
            const uint32_t pos = p.queue_block == 1u ? q.v_next * p.n_queues + q.k
                                                     : ((q.v_next / p.queue_block) * p.n_queues + q.k) * p.queue_block + q.v_next % p.queue_block;
            q.v_next++;
            if (pos < p.n_items) return pos;
            q.v_next = q.v_end;   // this queue is exhausted
            q.tried++;
        }
        if (q.tried >= p.n_queues) return 0xFFFFFFFFu;
        uint32_t k = home + q.tried;
        if (k >= p.n_queues) k -= p.n_queues;
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(&p.queues[k * p.queue_stride], 1u);   // one position per atomic: 2 / 4 were 6 % / 50 % slower
        v = __builtin_amdgcn_readfirstlane(v);
        q.k = k; q.v_next = v; q.v_end = v + 1u;
    }
}

// fragCol = vec4(colour, 1) (:231,233) as one 16-byte store.  The alpha is materialised right here: as a literal the
// compiler hoists it out of the kernel's main loop as the 4th register of a tuple and then spills and reloads that
// whole tuple around the traversal (16 bytes of scratch traffic per lane and outer iteration).
__device__ __forceinline__ void store_frag_col(float *__restrict__ rgba, uint32_t index, float r, float g, float b) {
    float one;
    asm volatile("v_mov_b32 %0, 1.0" : "=v"(one));
    *reinterpret_cast<float4 *>(rgba + 4 * (size_t)index) = make_float4(r, g, b, one);
}

enum : uint32_t { PH_IDLE = 0, PH_PRIMARY = 1, PH_SHADOW = 2 };   // bits 2..5: (8 | direction octant) << 2 when the octant test applies
enum : int { MODE_FUSED = 0, MODE_TRACE = 1 };
__device__ __forceinline__ uint32_t ray_pos_flag(const Ray &r, uint32_t octant_ok) {
    // octant code for slab_sel<OCT>: all three inverse components finite and non-zero, in a scene whose boxes allow it
    const float inf = __builtin_inff();
    const bool ok = octant_ok != 0u && fabsf(r.rx) < inf && fabsf(r.ry) < inf && fabsf(r.rz) < inf && r.rx != 0.0f && r.ry != 0.0f && r.rz != 0.0f;
    const uint32_t oct = (r.rx < 0.0f ? 1u : 0u) | (r.ry < 0.0f ? 2u : 0u) | (r.rz < 0.0f ? 4u : 0u);
    return ok ? ((8u | oct) << 2) : 0u;
}

// ---------------------------------------------------------------------------
// k_persistent — production kernel
// ---------------------------------------------------------------------------
// SINGLE (fused mode only): the launch is known to hold one frame, and the kernel carries none of the batch's frame look-ups.  Batches (and
// trace mode) use SINGLE = false, which tests JobParams::n_frames at run time and serves both.  Measured (profiles/r03_ab_batch_template.txt): the
// single-frame instantiation renders a frame alone 0.8 % (atrium) to 4 % (bunny) faster; a batch-only instantiation is smaller (59 VGPRs, 42
// instead of 56 spilled SGPRs) and 1.4 % SLOWER on the atrium's batches, so batches keep the kernel that serves both.
template <int MODE, bool COUNT, bool MAT = false, bool F8 = false, bool SINGLE = false>
__global__ void __launch_bounds__(BLOCK) RTAMD_PERSISTENT_ATTR k_persistent(const SceneDev s, const JobParams p) {
#define IS_BATCH (!SINGLE && p.n_frames > 1u)
    __shared__ uint32_t lds_stack[LDS_STACK_ENTRIES * BLOCK];
    // fused shader: the in-shadow colour of a lane's pixel waits here ([channel][thread]) while its shadow ray is traced
    __shared__ float lds_shadow_col[MODE == MODE_FUSED ? 3 * BLOCK : 1];
    // batch of frames: a lane finds its frame's output buffers here (filled from the kernel arguments below)
    __shared__ uint64_t lds_frame_out[MAX_BATCH_FRAMES][MODE == MODE_FUSED ? 4 : 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (MODE == MODE_TRACE && p.n_frames > 1u) {
        for (uint32_t f = 0; f < p.n_frames; f++)
            if (tid == f) lds_frame_out[f][0] = (uint64_t)(uintptr_t)p.frames[f].rgba;     // the job's hit records
        __syncthreads();
    }
    if (MODE == MODE_FUSED && IS_BATCH) {
        for (uint32_t f = 0; f < p.n_frames; f++)      // f is wave-uniform: scalar loads of the arguments, one lane stores
            if (tid == f) {
                lds_frame_out[f][0] = (uint64_t)(uintptr_t)p.frames[f].rgba;  lds_frame_out[f][1] = (uint64_t)(uintptr_t)p.frames[f].hit_id;
                lds_frame_out[f][2] = (uint64_t)(uintptr_t)p.frames[f].hit_t; lds_frame_out[f][3] = (uint64_t)(uintptr_t)p.frames[f].counts;
            }
        __syncthreads();
    }
    StackGlobal st;
    st.lds = lds_stack;
    st.spill_base = p.spill + (size_t)(blockIdx.x * (BLOCK / 64)) * p.spill_wave_words;
    st.wave_words = p.spill_wave_words;

    // wave-uniform pool of pending pixels / rays: item `pool_item`, lanes [pool_next, 64)
    uint32_t pool_item = 0, pool_next = 64;
    bool drained = false;
    QueueCursor qc{0, 0, 0, 0};
    // home queue: the wave's XCD, and with more than 8 queues one of that XCD's nq/8 counters
    const uint32_t home = p.n_queues > 8 ? (xcc_id() + 8u * ((blockIdx.x >> 3) % (p.n_queues >> 3))) % p.n_queues
                                         : (p.n_queues > 1 ? xcc_id() % p.n_queues : 0u);
#ifdef RTAMD_EXP_STAMPS
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long stamp_items = 0, stamp_last = stamp_t0, stamp_worst = 0, stamp_worst_item = 0;
    unsigned long long stamp_inner = 0, stamp_inner_lanes = 0, stamp_leaf = 0, stamp_leaf_lanes = 0, stamp_uniform = 0;
    unsigned long long tm_inner = 0, tm_leaf = 0, snap_tmi = 0, snap_tml = 0;
    const unsigned long long tm_start = __builtin_amdgcn_s_memtime();
    unsigned long long snap_inner = 0, snap_leaf = 0, snap_uniform = 0, snap_lanes = 0, w_inner = 0, w_leaf = 0, w_uniform = 0, w_lanes = 0;
#endif

    // Lane re-fill threshold, chosen from what the previous launch of this kind on this stream measured
    // (stats: {wave-uniform visits, visits, active lanes summed over visits, threshold | age << 8}).
    // Without re-fill (1) the lanes of an 8x8 block walk the tree in lock step — best for camera and
    // sun-shadow rays (63-81 % of the lanes active per visit) — unless the rays of a block end at very
    // different times (triangle soups, shuffled rays: 19-37 %), where re-filling below dyn_incoherent busy
    // lanes is 6-30 % faster.  The measure is only meaningful without re-fill, so a launch that re-fills
    // keeps doing so for 63 launches and then one launch probes again without.
    // A batch of ray lists is different: coherent and incoherent lists sit side by side in ONE launch (config 5: shadow rays
    // next to ambient bounces), and the launch before it held other kinds of lists.  There every wave decides for itself,
    // list by list: it starts a list without re-fill and switches re-fill on for the rest of that list as soon as its own
    // visits in it (at least 32) had under half of the lanes active.
    uint32_t dyn_thresh = p.dyn_thresh, dyn_age = 0;
    const bool per_list = MODE == MODE_TRACE && p.dyn_auto && (p.n_frames > 1u || p.width != 0u || p.tiles_x != 0u);
    uint32_t probe_job = 0xFFFFFFFFu, probe_inner = 0, probe_lanes = 0;
    if (per_list) dyn_thresh = 1u;
    else if (p.dyn_auto && p.stats_prev) {
        const uint32_t n = p.stats_prev[1], l = p.stats_prev[2], was = p.stats_prev[3] & 255u, age = p.stats_prev[3] >> 8;
        if (n >= 4096u && was != 0u) {
            if (was == 1u) dyn_thresh = (2ull * l < 64ull * n) ? p.dyn_incoherent : 1u;   // under 50 % of the lanes active per visit
            else if (age < 63u) { dyn_thresh = was; dyn_age = age + 1u; }
            else dyn_thresh = 1u;
        }
    }
    uint32_t stat_inner = 0, stat_uniform = 0, stat_lanes = 0;
    uint32_t phase = PH_IDLE;
    Trav tr;
    tr.cur = REF_NONE; tr.sp = st.sp_init(tid); tr.sb = 0; tr.t = 0; tr.idx = 0; tr.u = tr.v = 0;
    tr.ray = Ray{0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t out_index = 0;
    uint32_t nbox = 0, ntri = 0;

    for (;;) {
        // ---- re-fill idle lanes from the pool (ballot + mbcnt compaction) ----
        unsigned long long idle = __ballot(phase == PH_IDLE);
        while (idle != 0ull && !drained) {
            if (pool_next >= 64u) {
                uint32_t item = fetch_item(p, home, qc, lane);
#ifdef RTAMD_EXP_STAMPS
                {   // with dyn_thresh == 1 a fetch happens only when the previous item is completely done
                    unsigned long long now = __builtin_amdgcn_s_memrealtime();
                    if (stamp_items && lane == 0 && p.stamps && pool_item < 65536u) {   // the item records hold 65536 items
                        unsigned long long *o = p.stamps + 1048576ull + 4ull * pool_item;
                        o[0] = stamp_last; o[1] = now; o[2] = (stamp_inner - snap_inner) | ((stamp_leaf - snap_leaf) << 32); o[3] = ((tm_inner - snap_tmi) >> 4) | (((tm_leaf - snap_tml) >> 4) << 32);
                    }
                    if (stamp_items && now - stamp_last > stamp_worst) { stamp_worst = now - stamp_last; stamp_worst_item = pool_item;
                        w_inner = stamp_inner - snap_inner; w_leaf = stamp_leaf - snap_leaf; w_uniform = stamp_uniform - snap_uniform; w_lanes = stamp_inner_lanes - snap_lanes; }
                    if (item != 0xFFFFFFFFu) {
                        stamp_last = now; snap_inner = stamp_inner; snap_leaf = stamp_leaf; snap_uniform = stamp_uniform; snap_lanes = stamp_inner_lanes;
                        snap_tmi = tm_inner; snap_tml = tm_leaf;
                        stamp_items++;
                    }
                }
#endif
                if (item == 0xFFFFFFFFu) { drained = true; break; }
                pool_item = item; pool_next = 0;
                if (MODE == MODE_TRACE && per_list) {
                    uint32_t job = 0;
                    for (uint32_t k = 1; k < p.n_frames; k++) if (item >= p.job_item_begin[k]) job = k;
                    if (job != probe_job) { probe_job = job; dyn_thresh = 1u; probe_inner = stat_inner; probe_lanes = stat_lanes; }
                    else if (dyn_thresh == 1u && stat_inner - probe_inner >= 32u && 2u * (stat_lanes - probe_lanes) < 64u * (stat_inner - probe_inner))
                        dyn_thresh = p.dyn_incoherent;
                }
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const uint32_t avail = 64u - pool_next;
            if (phase == PH_IDLE && rank < avail) {
                const uint32_t l = pool_next + rank;
                if (MODE == MODE_FUSED) {
                    // batch: item -> (frame, item of that frame); everything here is wave-uniform
                    uint32_t frame = 0, item_in_frame = pool_item;
                    const float *f_dir = p.dir, *f_org = p.org;
                    if (IS_BATCH) {
                        for (uint32_t k = 1; k < p.n_frames; k++) if (pool_item >= k * p.items_per_frame) frame = k;
                        item_in_frame = pool_item - frame * p.items_per_frame;
                        f_dir = p.frames[frame].dir; f_org = p.frames[frame].org;
                    }
                    PixelRef pr = item_pixel(p, item_in_frame, l);
                    if (pr.valid) {
                        float dx, dy, dz, ox, oy, oz;
                        if (f_dir) {
                            const float *d = f_dir + 3 * (size_t)pr.ray_index, *o = f_org + 3 * (size_t)pr.ray_index;
                            dx = d[0]; dy = d[1]; dz = d[2]; ox = o[0]; oy = o[1]; oz = o[2];
                        } else if (IS_BATCH) {
                            const FrameIO &fr = p.frames[frame];
                            gen_primary_dir(fr.ipv, fr.near_, fr.far_, p.width, p.height, pr.px, pr.py, dx, dy, dz);
                            ox = fr.cam[0]; oy = fr.cam[1]; oz = fr.cam[2];
                        } else {
                            gen_primary_dir(p.ipv, p.near_, p.far_, p.width, p.height, pr.px, pr.py, dx, dy, dz);
                            ox = p.cam[0]; oy = p.cam[1]; oz = p.cam[2];
                        }
                        set_ray(tr.ray, ox, oy, oz, dx, dy, dz);
                        out_index = pr.out_index | (frame << BATCH_FRAME_SHIFT);   // frame is 0 unless this is a batch
                        nbox = 0; ntri = 0;
                        trav_begin<COUNT>(s, tr, T_PRIMARY0, st.sp_init(tid), nbox);
                        phase = PH_PRIMARY | ray_pos_flag(tr.ray, s.octant_ok);
                    }
                } else {
                    // batch of ray lists: item -> (job, item of that job); everything here is wave-uniform
                    uint32_t job = 0, item_in_job = pool_item;
                    const float *j_dir = p.dir, *j_org = p.org;
                    float *j_hits = p.hits;
                    uint64_t j_rays = p.n_rays;
                    if (p.n_frames > 1u) {
                        for (uint32_t k = 1; k < p.n_frames; k++) if (pool_item >= p.job_item_begin[k]) job = k;
                        item_in_job = pool_item - p.job_item_begin[job];
                        j_dir = p.frames[job].dir; j_org = p.frames[job].org; j_hits = p.frames[job].rgba; j_rays = p.frames[job].n_rays;
                    }
                    // Which ray of the list this lane takes.  A plain list: 64 consecutive rays per item.  A list that is an image in
                    // scan-line order (p.width) or in the compact tile-major layout (p.tiles_x): an 8x8 pixel block per item, the fused
                    // kernel's item shape -- the same rays, grouped so that a wave's rays stay together longer (config 5's lists at
                    // 3840x2160: -9 % over a sample's twelve lists, -20 % on the coherent ones, profiles/r03_ab_block_order.txt).
                    // An item is eight runs of eight rays: first ray `base`, `stride` rays from run to run, `cols` rays per run that exist
                    // (all wave-uniform); 64 consecutive rays are the case stride = 8.
                    uint32_t base = item_in_job * 64u, stride = 8u, cols = 8u;
                    if (p.tiles_x) {
                        const uint32_t sub = item_in_job & 15u;
                        base = (item_in_job >> 4) * RTAMD_TILE_PIXELS + (sub >> 2) * 256u + (sub & 3u) * 8u; stride = 32u;
                    } else if (p.width) {
                        // by = item / blocks per row, as a multiplication (p.height carries ceil(2^32 / blocks per row); the host checked its range)
                        const uint32_t bpr = (p.width + 7u) >> 3, by = __umulhi(item_in_job, p.height), bx = item_in_job - by * bpr;
                        base = by * 8u * p.width + bx * 8u; stride = p.width;
                        cols = p.width - bx * 8u; cols = cols < 8u ? cols : 8u;
                    }
                    const uint64_t ri = (uint64_t)base + (l >> 3) * stride + (l & 7u);
                    const bool has_ray = (l & 7u) < cols;
                    if (has_ray && ri < j_rays) {
                        const float *d = j_dir + 3 * ri, *o = j_org + 3 * ri;
                        out_index = (uint32_t)ri | (job << BATCH_FRAME_SHIFT);      // job is 0 unless this is a batch
                        nbox = 0; ntri = 0;
                        if (d[0] == 0.0f && d[1] == 0.0f && d[2] == 0.0f) {
                            // NO_RAY: the fragment is discarded, the target keeps its clear value
                            *reinterpret_cast<float4 *>(j_hits + 4 * ri) = make_float4(MISS, 0.0f, 0.0f, 0.0f);
                            if (COUNT && p.counts) { p.counts[2 * ri] = 0; p.counts[2 * ri + 1] = 0; }
                        } else {
                            set_ray(tr.ray, o[0], o[1], o[2], d[0], d[1], d[2]);
                            trav_begin<COUNT>(s, tr, MISS, st.sp_init(tid), nbox);
                            phase = PH_PRIMARY | ray_pos_flag(tr.ray, s.octant_ok);
                        }
                    }
                }
            }
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            pool_next += n_idle < avail ? n_idle : avail;
            idle = __ballot(phase == PH_IDLE);
        }
        if (__ballot(phase != PH_IDLE) == 0ull) break;
        // ---- traverse until too few lanes are still busy ----
        const uint32_t thresh = drained ? 1u : dyn_thresh;
        for (;;) {
            if ((uint32_t)__popcll(__builtin_amdgcn_ballot_w64(tr.cur != REF_NONE)) < thresh) break;
#ifdef RTAMD_EXP_STAMPS
            const unsigned long long tm0 = __builtin_amdgcn_s_memtime();
#endif
            // while-while with early leaf release: lanes that reached a leaf wait for the
            // others to finish their inner-node steps, but only until `leaf_thresh` lanes
            // are waiting (64 = classic while-while, 1 = if-if).  Per-lane order is
            // untouched, so results are too; only the wave's lane utilisation changes.
            // Octant fast path: when every busy lane's ray lies in the same direction octant (always
            // for the sun shadow rays, mostly for the camera rays of an 8x8 block) the visits use the
            // octant's slab_sel: same bits, 12 fewer VALU per visit.  phase bits 2..5 = 8 | octant.
            const uint32_t my_oct = (phase >> 2) & 15u;
            const unsigned long long m_busy = __builtin_amdgcn_ballot_w64(tr.cur != REF_NONE);
            // This round's leaf-release count: leaf_thresh, or half of the busy lanes when that is fewer -- a wave with few rays left must
            // not make a lane at a leaf wait for a count the others can hardly reach.  (The form matters to the register allocator: the
            // same count from the loop head's ballot gave 61 instead of 64 VGPRs, equally spill-free, and 12-23 % slower incoherent
            // rays; a run-time fraction or shift instead of the constant one 1.4 % slower batches: profiles/r03_ab_leaf_release_rule.txt.)
            uint32_t leaf_eff = (uint32_t)__popcll(m_busy) >> 1;
            leaf_eff = leaf_eff < 1u ? 1u : leaf_eff;
            const uint32_t leaf_cap = dyn_thresh == 1u ? p.leaf_thresh : p.leaf_thresh_refill;
            leaf_eff = leaf_eff < leaf_cap ? leaf_eff : leaf_cap;
            const uint32_t oct0 = m_busy ? __builtin_amdgcn_readlane(my_oct, (int)__builtin_ctzll(m_busy)) : 0u;
            const bool oct_wave = (oct0 & 8u) && __builtin_amdgcn_ballot_w64(tr.cur != REF_NONE && my_oct != oct0) == 0ull;
            auto inner_loop = [&](auto pos_tag) __attribute__((always_inline)) {
                constexpr int OCT = decltype(pos_tag)::value;
                for (;;) {
                    // reference classes by one compare each: inner < REF_NONE (unsigned), leaf < 0 (signed)
                    const bool in = tr.cur < REF_NONE;
                    const unsigned long long m_in = __builtin_amdgcn_ballot_w64(in);
                    if (!m_in) break;
                    const unsigned long long m_leaf = __builtin_amdgcn_ballot_w64((int32_t)tr.cur < 0);
                    if ((uint32_t)__popcll(m_leaf) >= leaf_eff) break;
#ifdef RTAMD_EXP_STAMPS
                    stamp_inner++; stamp_inner_lanes += __popcll(m_in);
                    {
                        const uint32_t c0 = __builtin_amdgcn_readfirstlane(tr.cur);
                        if (__ballot(in && tr.cur == c0) == m_in) stamp_uniform++;
                    }
#endif
                    stat_uniform += trav_inner_step<COUNT, OCT>(s, tr, st, tid, in, m_in, nbox) ? 1u : 0u;
                    stat_inner += 1u;
                    stat_lanes += (uint32_t)__popcll(m_in);
                }
            };
            if (!oct_wave) inner_loop(std::integral_constant<int, -1>{});
            else switch (oct0 & 7u) {
                case 0: inner_loop(std::integral_constant<int, 0>{}); break;
                case 1: inner_loop(std::integral_constant<int, 1>{}); break;
                case 2: inner_loop(std::integral_constant<int, 2>{}); break;
                case 3: inner_loop(std::integral_constant<int, 3>{}); break;
                case 4: inner_loop(std::integral_constant<int, 4>{}); break;
                case 5: inner_loop(std::integral_constant<int, 5>{}); break;
                case 6: inner_loop(std::integral_constant<int, 6>{}); break;
                default: inner_loop(std::integral_constant<int, 7>{}); break;
            }
            {
                const bool lf = (int32_t)tr.cur < 0;
#ifdef RTAMD_EXP_STAMPS
                const unsigned long long tm1 = __builtin_amdgcn_s_memtime();
                tm_inner += tm1 - tm0;
                const unsigned long long m = __ballot(lf);
                if (m) { stamp_leaf++; stamp_leaf_lanes += __popcll(m); }
#endif
                if (lf) trav_leaf<COUNT>(s, tr, st, tid, ntri, MODE == MODE_FUSED && p.shadow_anyhit && (phase & PH_SHADOW));
#ifdef RTAMD_EXP_STAMPS
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                tm_leaf += __builtin_amdgcn_s_memtime() - tm1;
#endif
            }
        }

        // ---- retire finished traversals ----
        if (phase != PH_IDLE && tr.cur == REF_NONE) {
            if (MODE == MODE_TRACE) {
                // ray_trace.frag:139-142
                float *o_hits = p.hits;
                uint32_t out_i = out_index;
                if (p.n_frames > 1u) {      // batch: the lane's job sits above BATCH_FRAME_SHIFT, its hit records in the LDS table
                    o_hits = (float *)(uintptr_t)lds_frame_out[out_index >> BATCH_FRAME_SHIFT][0];
                    out_i = out_index & ((1u << BATCH_FRAME_SHIFT) - 1u);
                }
                *reinterpret_cast<float4 *>(o_hits + 4 * (size_t)out_i) = make_float4(tr.t, tr.u, tr.v, __uint_as_float(tr.idx));
                if (COUNT && p.counts) { p.counts[2 * (size_t)out_i] = nbox; p.counts[2 * (size_t)out_i + 1] = ntri; }
                phase = PH_IDLE;
            } else {
              // this lane's output buffers: the kernel arguments, or for a batch its frame's entry of the LDS table
              float *o_rgba = p.rgba, *o_t = p.hit_t;
              uint32_t *o_id = p.hit_id, *o_cnt = p.counts;
              uint32_t out_i = out_index;
              if (IS_BATCH) {
                  const uint32_t f = out_index >> BATCH_FRAME_SHIFT;
                  out_i = out_index & ((1u << BATCH_FRAME_SHIFT) - 1u);
                  o_rgba = (float *)(uintptr_t)lds_frame_out[f][0]; o_id = (uint32_t *)(uintptr_t)lds_frame_out[f][1];
                  o_t = (float *)(uintptr_t)lds_frame_out[f][2];    o_cnt = (uint32_t *)(uintptr_t)lds_frame_out[f][3];
              }
              if (phase & PH_PRIMARY) {
                // The primary hit record is final here, and so is the pixel's colour unless the shadow ray finds an
                // occluder: both are written now, and only the in-shadow colour waits (in LDS) for the shadow
                // traversal.  Nothing of the pixel but its output index stays in registers across that traversal
                // (keeping both candidate colours, t and the triangle id live cost the 64-VGPR build 41 spilled
                // registers: 164 MB of scratch writes per 1080p frame).
                if (o_t) o_t[out_i] = tr.t;
                if (tr.t < T_HIT_LIMIT) {
                    if (o_id) o_id[out_i] = tr.idx;
                    Ray sray;
                    V3 col_shadow, col_lit;
                    shade_primary<MAT, F8 ? 1 : 0>(s, p, tr, col_shadow, col_lit, sray);
                    if (o_rgba) store_frag_col(o_rgba, out_i, col_lit.x, col_lit.y, col_lit.z);
                    lds_shadow_col[tid] = col_shadow.x; lds_shadow_col[BLOCK + tid] = col_shadow.y; lds_shadow_col[2 * BLOCK + tid] = col_shadow.z;
                    tr.ray = sray;
                    trav_begin<COUNT>(s, tr, T_PRIMARY0, st.sp_init(tid), nbox);
                    phase = PH_SHADOW | ray_pos_flag(sray, s.octant_ok);
                } else {
                    V3 c = skybox(tr.ray.dx, tr.ray.dy, tr.ray.dz);
                    if (o_rgba) store_frag_col(o_rgba, out_i, c.x, c.y, c.z);
                    if (o_id) o_id[out_i] = 0xFFFFFFFFu;
                    if (COUNT && o_cnt) { o_cnt[2 * (size_t)out_i] = nbox; o_cnt[2 * (size_t)out_i + 1] = ntri; }
                    phase = PH_IDLE;
                }
              } else {  // PH_SHADOW finished: an occluded pixel takes its in-shadow colour (same lane, same address, program order)
                if (tr.t < T_HIT_LIMIT && o_rgba)
                    store_frag_col(o_rgba, out_i, lds_shadow_col[tid], lds_shadow_col[BLOCK + tid], lds_shadow_col[2 * BLOCK + tid]);
                if (COUNT && o_cnt) { o_cnt[2 * (size_t)out_i] = nbox; o_cnt[2 * (size_t)out_i + 1] = ntri; }
                phase = PH_IDLE;
              }
            }
        }
    }
    if (p.stats_cur && lane == 0 && (blockIdx.x & 7u) == 0u) {   // every 8th workgroup reports
        atomicAdd(&p.stats_cur[0], stat_uniform);
        atomicAdd(&p.stats_cur[1], stat_inner);
        atomicAdd(&p.stats_cur[2], stat_lanes);
        p.stats_cur[3] = (per_list ? 1u : dyn_thresh) | (dyn_age << 8);     // (a batch of lists reports 1: its waves chose list by list)
    }
#ifdef RTAMD_EXP_STAMPS
    if (lane == 0 && p.stamps) {
        unsigned long long *o = p.stamps + 16ull * (blockIdx.x * (BLOCK / 64) + (tid >> 6));
        o[4] = stamp_inner; o[5] = stamp_inner_lanes; o[6] = stamp_leaf | (stamp_uniform << 32); o[7] = stamp_leaf_lanes;
        unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (now - stamp_last > stamp_worst) { stamp_worst = now - stamp_last; stamp_worst_item = pool_item;
            w_inner = stamp_inner - snap_inner; w_leaf = stamp_leaf - snap_leaf; w_uniform = stamp_uniform - snap_uniform; w_lanes = stamp_inner_lanes - snap_lanes; }
        o[8] = w_inner; o[9] = w_leaf; o[10] = w_uniform; o[11] = w_lanes;
        o[12] = tm_inner; o[13] = tm_leaf; o[14] = __builtin_amdgcn_s_memtime() - tm_start; o[15] = now - stamp_last;
        o[0] = stamp_t0; o[1] = now; o[2] = stamp_items | (stamp_worst << 32); o[3] = xcc_id() | (stamp_worst_item << 8);
    }
#endif
}

#undef IS_BATCH

// ---------------------------------------------------------------------------
// k_static — A/B baseline: one lane per pixel / ray, one 64-lane item per wave,
// no persistence, no re-fill.  Same device functions, same results.
// ---------------------------------------------------------------------------
template <int MODE, bool COUNT, bool MAT = false>
__global__ void __launch_bounds__(BLOCK) k_static(const SceneDev s, const JobParams p) {
    __shared__ uint32_t lds_stack[LDS_STACK_ENTRIES * BLOCK];
#ifdef RTAMD_EXP_PAD_LDS   // occupancy experiment only: waste LDS to cap resident blocks
    __shared__ uint32_t lds_pad[RTAMD_EXP_PAD_LDS / 4];
    if (p.n_items == 0xFFFFFFFFu) lds_pad[threadIdx.x] = 1, p.rgba[0] = (float)lds_pad[threadIdx.x ^ 1];
#endif
    StackScratch st;
    st.lds = lds_stack;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t item = blockIdx.x * (BLOCK / 64) + (tid >> 6);
    if (item >= p.n_items) return;
    Trav tr;
    uint32_t nbox = 0, ntri = 0;
    if (MODE == MODE_TRACE) {
        const uint64_t ri = (uint64_t)item * 64u + lane;
        if (ri >= p.n_rays) return;
        const float *d = p.dir + 3 * ri, *o = p.org + 3 * ri;
        if (d[0] == 0.0f && d[1] == 0.0f && d[2] == 0.0f) {
            *reinterpret_cast<float4 *>(p.hits + 4 * ri) = make_float4(MISS, 0.0f, 0.0f, 0.0f);
            if (COUNT && p.counts) { p.counts[2 * ri] = 0; p.counts[2 * ri + 1] = 0; }
            return;
        }
        set_ray(tr.ray, o[0], o[1], o[2], d[0], d[1], d[2]);
        trav_begin<COUNT>(s, tr, MISS, st.sp_init(tid), nbox);
        trav_run<COUNT>(s, tr, st, tid, nbox, ntri);
        *reinterpret_cast<float4 *>(p.hits + 4 * ri) = make_float4(tr.t, tr.u, tr.v, __uint_as_float(tr.idx));
        if (COUNT && p.counts) { p.counts[2 * ri] = nbox; p.counts[2 * ri + 1] = ntri; }
        return;
    }
    PixelRef pr = item_pixel(p, item, lane);
    if (!pr.valid) return;
    float dx, dy, dz, ox, oy, oz;
    if (p.dir) {
        const float *d = p.dir + 3 * (size_t)pr.ray_index, *o = p.org + 3 * (size_t)pr.ray_index;
        dx = d[0]; dy = d[1]; dz = d[2]; ox = o[0]; oy = o[1]; oz = o[2];
    } else {
        gen_primary_dir(p.ipv, p.near_, p.far_, p.width, p.height, pr.px, pr.py, dx, dy, dz);
        ox = p.cam[0]; oy = p.cam[1]; oz = p.cam[2];
    }
    set_ray(tr.ray, ox, oy, oz, dx, dy, dz);
    trav_begin<COUNT>(s, tr, T_PRIMARY0, st.sp_init(tid), nbox);
    trav_run<COUNT>(s, tr, st, tid, nbox, ntri);
    const float prim_t = tr.t;
    const uint32_t prim_idx = tr.idx;
    V3 c;
    uint32_t id = 0xFFFFFFFFu;
    if (tr.t < T_HIT_LIMIT) {
        V3 col_shadow, col_lit;
        Ray sray;
        shade_primary<MAT>(s, p, tr, col_shadow, col_lit, sray);
        tr.ray = sray;
        trav_begin<COUNT>(s, tr, T_PRIMARY0, st.sp_init(tid), nbox);
        trav_run<COUNT>(s, tr, st, tid, nbox, ntri, p.shadow_anyhit != 0);
        c = (tr.t < T_HIT_LIMIT) ? col_shadow : col_lit;
        id = prim_idx;
    } else {
        c = skybox(dx, dy, dz);
    }
    if (p.rgba) *reinterpret_cast<float4 *>(p.rgba + 4 * (size_t)pr.out_index) = make_float4(c.x, c.y, c.z, 1.0f);
    if (p.hit_id) p.hit_id[pr.out_index] = id;
    if (p.hit_t) p.hit_t[pr.out_index] = prim_t;
    if (COUNT && p.counts) { p.counts[2 * (size_t)pr.out_index] = nbox; p.counts[2 * (size_t)pr.out_index + 1] = ntri; }
}

// ---------------------------------------------------------------------------
// small utility kernels
// ---------------------------------------------------------------------------
struct Mat4Arg { float m[16]; };   // inverse(proj * view), column-major, by value in the kernel arguments (scalar loads, no staging buffer)
__global__ void k_generate_rays(const Mat4Arg ipv, float near_, float far_, uint32_t width, uint32_t height, float off_x, float off_y,
                                uint32_t rank, uint32_t world, uint32_t tiles_x, uint32_t n, float *__restrict__ dir_out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t px, py;
    float dx = 0.0f, dy = 0.0f, dz = 0.0f;      // outside the frame: NO_RAY (ray_trace.frag:6)
    if (shard_pixel(i, width, height, rank, world, tiles_x, px, py))
        gen_primary_dir(ipv.m, near_, far_, width, height, px, py, dx, dy, dz, off_x, off_y);
    dir_out[3 * (size_t)i + 0] = dx; dir_out[3 * (size_t)i + 1] = dy; dir_out[3 * (size_t)i + 2] = dz;
}

// Hit records -> the G-buffer the dispatcher reads, for bounces the rasteriser cannot provide
// (the reference fills positionData / normalMatData with its rasteriser pass, src/main.rs:127-140,
// so it has one bounce; this is that pass restated for ray hits): position = org + dir * t,
// normal = triangleNormal(idx, uv) (ray_trace_and_shade.frag:165-178), w = the triangle's material
// index as bits, or 1e30 ("no surface", main.rs:130) for a miss or a NO_RAY record.
__global__ void __launch_bounds__(256) k_surface_from_hits(const SceneDev s, uint64_t n, const float *__restrict__ org, const float *__restrict__ dir,
                                                           const float *__restrict__ hits, uint32_t has_n, float *__restrict__ position, float *__restrict__ normal_mat) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 h = reinterpret_cast<const float4 *>(hits)[i];
    float px = 0.0f, py = 0.0f, pz = 0.0f;
    float4 nm = make_float4(0.0f, 0.0f, 0.0f, MISS);
    if (h.x < MISS) {
        const uint32_t idx = __float_as_uint(h.w);
        const float u = h.y, v = h.z;
        px = org[3 * i] + dir[3 * i] * h.x; py = org[3 * i + 1] + dir[3 * i + 1] * h.x; pz = org[3 * i + 2] + dir[3 * i + 2] * h.x;
        const RefTriangle tri = s.tri_idx[idx];
        float nx, ny, nz;
        if (has_n) {
            const float w = 1.0f - u - v;
            const float *n0 = s.normals + 3 * (size_t)tri.p0, *n1 = s.normals + 3 * (size_t)tri.p1, *n2 = s.normals + 3 * (size_t)tri.p2;
            nx = (n1[0] * u + n2[0] * v) + n0[0] * w;
            ny = (n1[1] * u + n2[1] * v) + n0[1] * w;
            nz = (n1[2] * u + n2[2] * v) + n0[2] * w;
        } else {
            const TriRecord &rec = s.tris[idx];
            nx = rec.e1[1] * rec.e2[2] - rec.e2[1] * rec.e1[2];
            ny = rec.e1[2] * rec.e2[0] - rec.e2[2] * rec.e1[0];
            nz = rec.e1[0] * rec.e2[1] - rec.e2[0] * rec.e1[1];
        }
        const float inv = 1.0f / sqrtf(dot_zyx(nx, ny, nz, nx, ny, nz));
        nm = make_float4(nx * inv, ny * inv, nz * inv, __uint_as_float(tri.mat_idx));
    }
    position[3 * i] = px; position[3 * i + 1] = py; position[3 * i + 2] = pz;
    reinterpret_cast<float4 *>(normal_mat)[i] = nm;
}

// ---------------------------------------------------------------------------
// k_dispatch_rays — ray_trace/ray_dispatcher.frag:26-41: secondary-ray origins and the
// shadow / reflection / ambient-occlusion directions from a G-buffer.
// HBM-bound streaming kernel: 28 B in + 48 B out per pixel, noise texels from cache.
// A block owns 256 consecutive pixels; the 12-B-per-pixel streams (position in, four
// RGB32F targets out) go through LDS so that every global access is a full 16-B lane
// access of a contiguous 3 KB run instead of three stride-12 dword accesses.
// ---------------------------------------------------------------------------
struct DispatchParams {
    const float *position;     // 3 floats per pixel
    const float *normal_mat;   // 4 floats per pixel; w == 1e30: no surface
    const uint32_t *noise;     // RGBA8 texels
    int32_t noise_w, noise_h;
    float light[3], cam[3], nos[4];
    uint32_t width, height, n;
    uint32_t rank, world, tiles_x;   // world > 1: the buffers hold this rank's tiles (compact tile-major layout), n = local tiles * 1024
    uint32_t tex_filter8;            // as JobParams::tex_filter8, for the noise fetch
    float *out[4];             // org, shadowDir, reflectDir, ambientDir (any may be null)
};

__global__ void __launch_bounds__(256) k_dispatch_rays(const DispatchParams p) {
    __shared__ float sh_in[768];
    __shared__ float sh_out[4][768];
    const uint32_t tid = threadIdx.x;
    const uint32_t base = blockIdx.x * 256u;
    const uint32_t count = min(256u, p.n - base);
    const bool full = count == 256u;
    // stage the block's positions: 768 floats = 192 float4 (3*base*4 B is 16-B aligned: base % 256 == 0)
    if (full) {
        if (tid < 192u) reinterpret_cast<float4 *>(sh_in)[tid] = reinterpret_cast<const float4 *>(p.position + 3 * (size_t)base)[tid];
    } else {
        for (uint32_t k = tid; k < 3u * count; k += 256u) sh_in[k] = p.position[3 * (size_t)base + k];
    }
    __syncthreads();
    const uint32_t i = base + tid;
    float o[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};   // NO_RAY: the targets' clear value (main.rs:157)
    if (tid < count) {
        const float4 nm = reinterpret_cast<const float4 *>(p.normal_mat)[i];
        uint32_t px, py;
        if (shard_pixel(i, p.width, p.height, p.rank, p.world, p.tiles_x, px, py) && !(nm.w == MISS)) {   // :28 discard
            const float fx = ((float)px + 0.5f) / (float)p.width, fy = ((float)py + 0.5f) / (float)p.height;   // quad01.vert:8 at the pixel centre
            const float x = sh_in[3 * tid], y = sh_in[3 * tid + 1], z = sh_in[3 * tid + 2];
            const V3 rnd = sample_albedo(p.noise, p.noise_w, p.noise_h, fx * p.nos[2] + p.nos[0], fy * p.nos[3] + p.nos[1], p.tex_filter8 != 0);   // :34
            o[0][0] = x + nm.x * 0.0001f; o[0][1] = y + nm.y * 0.0001f; o[0][2] = z + nm.z * 0.0001f;                             // :38
            {   // :39 normalize(lightPos - position)
                const float dx = p.light[0] - x, dy = p.light[1] - y, dz = p.light[2] - z;
                const float inv = 1.0f / sqrtf(dot_zyx(dx, dy, dz, dx, dy, dz));
                o[1][0] = dx * inv; o[1][1] = dy * inv; o[1][2] = dz * inv;
            }
            {   // :41 normalize(reflect(position - cameraPos, normal)), reflect(I,N) = I - 2*dot(N,I)*N
                const float ix = x - p.cam[0], iy = y - p.cam[1], iz = z - p.cam[2];
                const float two_d = 2.0f * dot_zyx(nm.x, nm.y, nm.z, ix, iy, iz);
                const float rx = ix - nm.x * two_d, ry = iy - nm.y * two_d, rz = iz - nm.z * two_d;
                const float inv = 1.0f / sqrtf(dot_zyx(rx, ry, rz, rx, ry, rz));
                o[2][0] = rx * inv; o[2][1] = ry * inv; o[2][2] = rz * inv;
            }
            {   // :42 normalize(normal + random * 2 - 1)
                const float ax = (nm.x + rnd.x * 2.0f) - 1.0f, ay = (nm.y + rnd.y * 2.0f) - 1.0f, az = (nm.z + rnd.z * 2.0f) - 1.0f;
                const float inv = 1.0f / sqrtf(dot_zyx(ax, ay, az, ax, ay, az));
                o[3][0] = ax * inv; o[3][1] = ay * inv; o[3][2] = az * inv;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) { sh_out[k][3 * tid] = o[k][0]; sh_out[k][3 * tid + 1] = o[k][1]; sh_out[k][3 * tid + 2] = o[k][2]; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float *dst = p.out[k];
        if (!dst) continue;
        dst += 3 * (size_t)base;
        if (full) {
            if (tid < 192u) reinterpret_cast<float4 *>(dst)[tid] = reinterpret_cast<const float4 *>(sh_out[k])[tid];
        } else {
            for (uint32_t e = tid; e < 3u * count; e += 256u) dst[e] = sh_out[k][e];
        }
    }
}

// Root side of the gather: rank-major tile blocks -> scan-line frame(s).  blockIdx.y = frame f of a batch:
// `gathered` is [rank][frame][tiles_per_rank * 1024 pixels], `out` is [frame][height * width].
__global__ void k_assemble_tiles(const uint32_t *__restrict__ gathered, uint32_t width, uint32_t height, uint32_t world,
                                 uint32_t tiles_x, uint32_t tiles_per_rank, uint32_t elems, uint32_t *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= width * height) return;
    const uint32_t f = blockIdx.y, n_frames = gridDim.y;
    uint32_t x = i % width, y = i / width;
    uint32_t g = (y >> 5) * tiles_x + (x >> 5);
    uint32_t r = g % world, j = g / world;
    size_t src = (((size_t)r * n_frames + f) * tiles_per_rank + j) * RTAMD_TILE_PIXELS + ((y & 31u) * 32u + (x & 31u));
    size_t dst = (size_t)f * width * height + i;
    for (uint32_t e = 0; e < elems; e++) out[dst * elems + e] = gathered[src * elems + e];
}

// The same for one-word pixels (RGBA8) of frames whose width is a multiple of 4: a thread moves four neighbouring pixels of a
// tile row (tiles are 32 wide, so the four share their tile) with one 16-byte load and one 16-byte store.  The one-pixel form
// moves 4 bytes per lane and sits at 3.5 TB/s; this one at the rate of the RGBA32F copy.
__global__ void k_assemble_tiles_px4(const uint32_t *__restrict__ gathered, uint32_t width, uint32_t height, uint32_t world,
                                     uint32_t tiles_x, uint32_t tiles_per_rank, uint32_t *__restrict__ out) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;       // group of four pixels of the frame, scan-line order
    if (q >= (width >> 2) * height) return;
    const uint32_t f = blockIdx.y, n_frames = gridDim.y;
    const uint32_t x = (q % (width >> 2)) << 2, y = q / (width >> 2);
    const uint32_t g = (y >> 5) * tiles_x + (x >> 5);
    const uint32_t r = g % world, j = g / world;
    const size_t src = (((size_t)r * n_frames + f) * tiles_per_rank + j) * RTAMD_TILE_PIXELS + ((y & 31u) * 32u + (x & 31u));
    const size_t dst = (size_t)f * width * height + (size_t)y * width + x;
    *reinterpret_cast<uint4 *>(out + dst) = *reinterpret_cast<const uint4 *>(gathered + src);
}

// util/display.frag:9 writes the float colour target to the window's 8-bit back buffer; the conversion is the
// GL one (4.6 spec 2.3.5: clamp to [0,1], x255, round to nearest; NaN -> 0 here).  HBM stream: 16 B in, 4 B out.
__device__ __forceinline__ uint32_t to_unorm8(float x) {
    x = (x == x) ? fminf(fmaxf(x, 0.0f), 1.0f) : 0.0f;
    return (uint32_t)__builtin_rintf(x * 255.0f);
}
__global__ void k_display_rgba8(const float4 *__restrict__ rgba, uint64_t n, uint32_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c = rgba[i];
    out[i] = to_unorm8(c.x) | (to_unorm8(c.y) << 8) | (to_unorm8(c.z) << 16) | (to_unorm8(c.w) << 24);
}

}  // namespace rtamd
