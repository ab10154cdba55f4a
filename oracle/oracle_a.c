/*
 * oracle_a.c — TEST INFRASTRUCTURE ONLY (oracle A).
 *
 * Scalar CPU restatement of the reference's ray-tracing hot path, written
 * against the REFERENCE data layout (36-byte nodes, 16-byte index triangles,
 * packed float attribute streams) and following the GLSL line by line.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker; the product (raytracer_amd/) never does.
 *
 * Restated functions (all citations relative to the reference tree):
 *   oa_aabb()          intersectAABB        res/shaders/ray_trace_and_shade.frag:53-61
 *   oa_tri()           intersectTriangle    ray_trace_and_shade.frag:63-87
 *   oa_traverse()      traverseBVH          ray_trace_and_shade.frag:112-151
 *                                           (== ray_trace/ray_trace.frag:104-129)
 *   oa_tri_normal()    triangleNormal       ray_trace_and_shade.frag:165-178
 *   oa_tri_texcoord()  triangleTexCoord     ray_trace_and_shade.frag:153-163
 *   oa_skybox()        skybox + constants   ray_trace_and_shade.frag:180-193
 *   oa_shade_pixel()   main                 ray_trace_and_shade.frag:195-235
 *   oa_trace()         main                 ray_trace/ray_trace.frag:131-143
 *   oa_raycreate()     main                 ray_trace/ray_create.frag:12 (+ util/quad-11.vert:8)
 *   oa_sample_albedo() texture() on the RGBA8 / LINEAR / REPEAT texture that
 *                      src/gl_wrapper/texture.rs:11-42,65-79 creates
 *
 * Arithmetic conventions: GLSL leaves the evaluation order of dot(), the NaN
 * behaviour of min()/max() and the precision of pow()/normalize() to the
 * implementation.  This file follows what Mesa 23.2 llvmpipe executes, because
 * llvmpipe running the unmodified shader is the pinned reference (oracle B,
 * oracle/glsl_ref.c): un-fused mul/add, dot(a,b) = (a.z*b.z + a.y*b.y) + a.x*b.x,
 * normalize(v) = v * (1/sqrt(dot(v,v))), correctly rounded 1/x, min/max that
 * return the non-NaN operand, f32 denormals flushed, `uv.x + uv.y > 1.0` evaluated as
 * f*(du + dv) > 1.0 (the compiler's factoring, oa_tri), unorm8 texels converted as
 * x * fl(1/255).  Texture filtering is float bilinear by default (what the library
 * computes); oa_set_texture_filter_8bit(1) restates llvmpipe's 8-bit filter, with which
 * textured colours and the dispatcher's noise fetch equal llvmpipe's to the last bit or
 * 0.003/255.  Parity pin: tests/test_oracle_golden.py compares this file with llvmpipe
 * outputs committed under tests/golden/; tests/golden/soak_oracle_vs_llvmpipe.py does the
 * same on randomised adversarial inputs in the build container.
 *
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <xmmintrin.h>

#define OA_MISS 1e30f
#define OA_EPSILON 0.000001f
#define OA_NODE_STACK_SIZE 100

typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;

typedef struct {
    float minx, miny, minz, maxx, maxy, maxz;
    uint32_t is_leaf, a, b;
} oa_node; /* 36 bytes, ray_trace_and_shade.frag:27-36 */

typedef struct { uint32_t p0, p1, p2, mat; } oa_triangle; /* :38-40 */

typedef struct {
    const oa_node *nodes; uint32_t n_nodes;
    const oa_triangle *triangles; uint32_t n_triangles;
    const float *positions; uint32_t n_vertices;
    const float *texcoords; /* may be NULL */
    const float *normals;   /* may be NULL */
} oa_scene;

typedef struct { const uint8_t *rgb; int w, h; } oa_texture;

/* options --------------------------------------------------------------- */
static int g_flush_denormals = 1;   /* set MXCSR FTZ|DAZ in the worker (llvmpipe rasteriser threads do) */
void oa_set_flush_denormals(int on) { g_flush_denormals = on; }
static int g_edge_sum_as_written = 0;   /* 1: evaluate :83 as the rounded sum of the two products (negative control of golden_edge) */
void oa_set_edge_sum_as_written(int on) { g_edge_sum_as_written = on; }
/* deepest shader-stack occupancy seen since the last reset (statistics only; racy max is fine) */
static volatile uint32_t g_max_stack = 0;
uint32_t oa_max_stack(int reset) { uint32_t v = g_max_stack; if (reset) g_max_stack = 0; return v; }

static void apply_fp_mode(void) {
    unsigned csr = _mm_getcsr();
    if (g_flush_denormals) csr |= 0x8040u; else csr &= ~0x8040u;
    _mm_setcsr(csr);
}

/* vector helpers with the llvmpipe evaluation order ---------------------- */
static inline float oa_dot(v3 a, v3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }
static inline v3 oa_cross(v3 a, v3 b) {
    v3 r = { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y };
    return r;
}
static inline v3 oa_sub(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static inline v3 oa_add(v3 a, v3 b) { v3 r = { a.x + b.x, a.y + b.y, a.z + b.z }; return r; }
static inline v3 oa_scale(v3 a, float s) { v3 r = { a.x * s, a.y * s, a.z * s }; return r; }
static inline v3 oa_mulv(v3 a, v3 b) { v3 r = { a.x * b.x, a.y * b.y, a.z * b.z }; return r; }
static inline v3 oa_normalize(v3 v) { float inv = 1.0f / sqrtf(oa_dot(v, v)); return oa_scale(v, inv); }
static inline float oa_min(float a, float b) { return fminf(a, b); }
static inline float oa_max(float a, float b) { return fmaxf(a, b); }
static inline float oa_clamp01(float x) { return oa_min(oa_max(x, 0.0f), 1.0f); }
/* pow(): undefined in GLSL for x < 0; llvmpipe yields NaN which the following
 * clamp turns into 0 (measured: SURVEY.md row a9). */
static inline float oa_pow_clamp01(float x, float y) {
    if (!(x >= 0.0f)) return 0.0f;
    return oa_clamp01(powf(x, y));
}

static inline v3 fetch_position(const oa_scene *s, uint32_t i) { v3 r = { s->positions[i * 3 + 0], s->positions[i * 3 + 1], s->positions[i * 3 + 2] }; return r; }
static inline v3 fetch_normal(const oa_scene *s, uint32_t i) { v3 r = { s->normals[i * 3 + 0], s->normals[i * 3 + 1], s->normals[i * 3 + 2] }; return r; }
static inline v2 fetch_texcoord(const oa_scene *s, uint32_t i) { v2 r = { s->texcoords[i * 2 + 0], s->texcoords[i * 2 + 1] }; return r; }

typedef struct { v3 org, dir, rdir; } oa_ray;

/* intersectAABB, :53-61 */
static inline float oa_aabb(const oa_ray *ray, const oa_node *n, float t) {
    float tx1 = (n->minx - ray->org.x) * ray->rdir.x, tx2 = (n->maxx - ray->org.x) * ray->rdir.x;
    float tmin = oa_min(tx1, tx2), tmax = oa_max(tx1, tx2);
    float ty1 = (n->miny - ray->org.y) * ray->rdir.y, ty2 = (n->maxy - ray->org.y) * ray->rdir.y;
    tmin = oa_max(tmin, oa_min(ty1, ty2)); tmax = oa_min(tmax, oa_max(ty1, ty2));
    float tz1 = (n->minz - ray->org.z) * ray->rdir.z, tz2 = (n->maxz - ray->org.z) * ray->rdir.z;
    tmin = oa_max(tmin, oa_min(tz1, tz2)); tmax = oa_min(tmax, oa_max(tz1, tz2));
    return (tmax >= tmin && tmin < t && tmax > 0.0f) ? tmin : OA_MISS;
}

/* intersectTriangle, :63-87 */
static inline float oa_tri(const oa_ray *ray, v3 p0, v3 p1, v3 p2, v2 *uv) {
    v3 edge1 = oa_sub(p1, p0);
    v3 edge2 = oa_sub(p2, p0);
    v3 h = oa_cross(ray->dir, edge2);
    float a = oa_dot(edge1, h);
    if (fabsf(a) < OA_EPSILON) return -1.0f;
    float f = 1.0f / a;
    v3 s = oa_sub(ray->org, p0);
    const float du = oa_dot(s, h);
    uv->x = f * du;
    if (uv->x < 0.0f || uv->x > 1.0f) return -1.0f;
    v3 q = oa_cross(s, edge1);
    const float dv = oa_dot(ray->dir, q);
    uv->y = f * dv;
    /* `uv.x + uv.y > 1.0` (:83) AS THE PINNED REFERENCE EXECUTES IT: Mesa's NIR algebraic pass factors f*du + f*dv into
     * f*(du + dv) (GLSL allows it without `precise`; ST_DEBUG=nir shows `fadd du, dv; fmul f, .; flt 1.0, .` in both
     * shaders), which decides differently from the rounded sum of the two products for rays within an ulp of the edge
     * u + v = 1.  Found by tests/golden/soak_oracle_vs_llvmpipe.py; pinned by tests/golden/golden_edge.npz. */
    const float edge = g_edge_sum_as_written ? uv->x + uv->y : f * (du + dv);
    if (uv->y < 0.0f || edge > 1.0f) return -1.0f;
    return f * oa_dot(edge2, q);
}

/* traverseBVH, :112-151.  counts[0] += AABB tests, counts[1] += triangle tests
 * (the shader's single `intersections` counter split by kind).  Returns 0, or
 * -1 when the 100-entry stack of the shader (:12,43) would overflow. */
static int oa_traverse(const oa_scene *sc, const oa_ray *ray, float *t, uint32_t *tri_idx, v2 *tri_uv, uint32_t counts[2]) {
    uint32_t stack[OA_NODE_STACK_SIZE];
    uint32_t idx;
    counts[0]++;
    idx = oa_aabb(ray, &sc->nodes[0], *t) == OA_MISS ? 0 : 1;
    stack[0] = 0;
    while (idx > 0) {
        const oa_node *node = &sc->nodes[stack[--idx]];
        if (node->is_leaf) {
            for (uint32_t i = node->a; i < node->a + node->b; i++) {
                counts[1]++;
                v2 uv = { 0, 0 };
                const oa_triangle *tr = &sc->triangles[i];
                float new_t = oa_tri(ray, fetch_position(sc, tr->p0), fetch_position(sc, tr->p1), fetch_position(sc, tr->p2), &uv);
                if (new_t > OA_EPSILON && new_t < *t) {
                    *tri_idx = i;
                    *tri_uv = uv;
                    *t = new_t;
                }
            }
        } else {
            counts[0] += 2;
            float dist0 = oa_aabb(ray, &sc->nodes[node->a], *t);
            float dist1 = oa_aabb(ray, &sc->nodes[node->b], *t);
#define OA_PUSH(x) do { if (idx >= OA_NODE_STACK_SIZE) return -1; stack[idx++] = (x); if (idx > g_max_stack) g_max_stack = idx; } while (0)
            if (dist0 < dist1) {
                if (dist1 != OA_MISS) OA_PUSH(node->b);
                if (dist0 != OA_MISS) OA_PUSH(node->a);
            } else {
                if (dist0 != OA_MISS) OA_PUSH(node->a);
                if (dist1 != OA_MISS) OA_PUSH(node->b);
            }
#undef OA_PUSH
        }
    }
    return 0;
}

/* triangleTexCoord, :153-163 */
static v2 oa_tri_texcoord(const oa_scene *sc, int has_texcoords, uint32_t idx, v2 uv) {
    v2 r = { 0, 0 };
    if (has_texcoords) {
        const oa_triangle *tr = &sc->triangles[idx];
        v2 t0 = fetch_texcoord(sc, tr->p0), t1 = fetch_texcoord(sc, tr->p1), t2 = fetch_texcoord(sc, tr->p2);
        float w = 1.0f - uv.x - uv.y;
        r.x = (t1.x * uv.x + t2.x * uv.y) + t0.x * w;
        r.y = (t1.y * uv.x + t2.y * uv.y) + t0.y * w;
    }
    return r;
}

/* triangleNormal, :165-178 */
static v3 oa_tri_normal(const oa_scene *sc, int has_normals, uint32_t idx, v2 uv) {
    const oa_triangle *tr = &sc->triangles[idx];
    if (has_normals) {
        v3 n0 = fetch_normal(sc, tr->p0), n1 = fetch_normal(sc, tr->p1), n2 = fetch_normal(sc, tr->p2);
        float w = 1.0f - uv.x - uv.y;
        v3 n = oa_add(oa_add(oa_scale(n1, uv.x), oa_scale(n2, uv.y)), oa_scale(n0, w));
        return oa_normalize(n);
    } else {
        v3 v0 = fetch_position(sc, tr->p0), v1 = fetch_position(sc, tr->p1), v2_ = fetch_position(sc, tr->p2);
        return oa_normalize(oa_cross(oa_sub(v1, v0), oa_sub(v2_, v0)));
    }
}

/* constants, :180-187.  SUN_DIR is folded by the GLSL compiler:
 * normalize(vec3(1,2,1)) = vec3(1,2,1) * (1/sqrt(6)). */
static const v3 TOP_SKY = { .5f, .7f, .9f };
static const v3 BOTTOM_SKY = { .2f, .5f, .8f };
static const v3 SUN_COL = { 1.0f, 0.97f, 0.86f };
#define AMBIENT 0.5f
#define DIFFUSE 0.8f
#define SPECULAR 0.4f
#define SPEC_POW 30.0f
static inline v3 sun_dir(void) {
    float inv = 1.0f / sqrtf(6.0f);
    v3 r = { 1.0f * inv, 2.0f * inv, 1.0f * inv };
    return r;
}

/* skybox, :189-193 */
static v3 oa_skybox(v3 dir) {
    float sun_fac = oa_pow_clamp01(oa_dot(dir, sun_dir()), 200.0f);
    float sky_fac = (dir.y + 1.0f) * .5f;
    v3 r = oa_add(oa_add(oa_scale(TOP_SKY, sky_fac), oa_scale(BOTTOM_SKY, 1.0f - sky_fac)), oa_scale(SUN_COL, sun_fac));
    return r;
}

/* texture(sampler2D, uv).xyz for an RGBA8-unorm, LINEAR, REPEAT, single-level
 * texture whose rows were uploaded first-row-first (texture.rs:11-26). */
static inline int wrap_repeat(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
/* unorm8 -> float as llvmpipe converts it: a multiplication by fl(1/255), not the division (the two differ in the last bit for
 * 126 of the 256 values).  With it the dispatcher's ambient directions equal llvmpipe's bit for bit (golden_dispatch*.npz). */
#define OA_UNORM8 (1.0f / 255.0f)
static int g_filter_8bit = 0;
/* 0 (default): float bilinear, what the library computes.  1: the filter as Mesa llvmpipe executes it on an RGBA8 texture --
 * weights rounded to 8 bits, three 8-bit lerps (x on both rows, then y) each rounded to 8 bits, the result an 8-bit value --
 * found by probing its sampler (tests/golden/soak_oracle_vs_llvmpipe.py describes the probe: 100 % of 1.8 M samples equal).
 * Test-only: it shows that what is left between the oracle and llvmpipe on textured pixels is this filter and nothing else. */
void oa_set_texture_filter_8bit(int on) { g_filter_8bit = on; }

static v3 oa_sample_albedo(const oa_texture *tex, float s, float t) {
    v3 out;
    const int W = tex->w, H = tex->h;
    float u = s * (float)W - 0.5f, v = t * (float)H - 0.5f;
    float fu = floorf(u), fv = floorf(v);
    float a = u - fu, b = v - fv;
    int i0 = wrap_repeat((int)fu, W), j0 = wrap_repeat((int)fv, H);
    int i1 = wrap_repeat(i0 + 1, W), j1 = wrap_repeat(j0 + 1, H);
    const uint8_t *t00 = tex->rgb + 3 * ((size_t)j0 * W + i0), *t10 = tex->rgb + 3 * ((size_t)j0 * W + i1);
    const uint8_t *t01 = tex->rgb + 3 * ((size_t)j1 * W + i0), *t11 = tex->rgb + 3 * ((size_t)j1 * W + i1);
    float c[3];
    if (g_filter_8bit) {
        /* unless BOTH dimensions are powers of two, coordinates are wrapped BEFORE they are scaled (fract(s), clamped below 1):
         * the same texel pair, but the products round at another magnitude, which moves a weight at a near-tie */
        const int npot = (W & (W - 1)) || (H & (H - 1));
        if (npot) {
            float fs = s - floorf(s);
            if (fs >= 1.0f) fs = 0.99999994f;
            u = fs * (float)W - 0.5f; fu = floorf(u); a = u - fu;
            i0 = wrap_repeat((int)fu, W); i1 = wrap_repeat(i0 + 1, W);
        }
        if (npot) {
            float ft = t - floorf(t);
            if (ft >= 1.0f) ft = 0.99999994f;
            v = ft * (float)H - 0.5f; fv = floorf(v); b = v - fv;
            j0 = wrap_repeat((int)fv, H); j1 = wrap_repeat(j0 + 1, H);
        }
        t00 = tex->rgb + 3 * ((size_t)j0 * W + i0); t10 = tex->rgb + 3 * ((size_t)j0 * W + i1);
        t01 = tex->rgb + 3 * ((size_t)j1 * W + i0); t11 = tex->rgb + 3 * ((size_t)j1 * W + i1);
        const int wx = (int)nearbyintf(a * 256.0f), wy = (int)nearbyintf(b * 256.0f);
        for (int k = 0; k < 3; k++) {
            const int top = t00[k] + (((t10[k] - t00[k]) * wx + 128) >> 8);
            const int bot = t01[k] + (((t11[k] - t01[k]) * wx + 128) >> 8);
            c[k] = (float)(top + (((bot - top) * wy + 128) >> 8)) * OA_UNORM8;
        }
        out.x = c[0]; out.y = c[1]; out.z = c[2];
        return out;
    }
    for (int k = 0; k < 3; k++) {
        float c00 = t00[k] * OA_UNORM8, c10 = t10[k] * OA_UNORM8, c01 = t01[k] * OA_UNORM8, c11 = t11[k] * OA_UNORM8;
        float top = c00 + a * (c10 - c00);
        float bot = c01 + a * (c11 - c01);
        c[k] = top + b * (bot - top);
    }
    out.x = c[0]; out.y = c[1]; out.z = c[2];
    return out;
}

/* texture() on n explicit coordinates (st: n x 2) -> rgb (n x 3); for the sampler fixture (tests/golden/golden_sampler.npz) */
void oa_sample_points(const uint8_t *rgb, int w, int h, uint64_t n, const float *st, float *out) {
    unsigned csr = _mm_getcsr();
    apply_fp_mode();
    const oa_texture tex = { rgb, w, h };
    for (uint64_t i = 0; i < n; i++) {
        const v3 c = oa_sample_albedo(&tex, st[2 * i], st[2 * i + 1]);
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
    _mm_setcsr(csr);
}

/* main, :195-235 */
typedef struct {
    float rgba[4];
    uint32_t hit_id;   /* 0xFFFFFFFF on miss */
    float hit_t;
    uint32_t counts[2];
    int shadow;        /* -1 miss, 0 lit, 1 in shadow */
    int overflow;
} oa_pixel;

/* Per-material shading (include/rtamd.h rtamd_upload_materials): the MTL
 * quantities of src/rendering/material.rs:5-16 in the places of the shader's constants.  Same field
 * order as rtamd_shading_material.  No reference output exists for a table other than the constants. */
typedef struct { float ka[3], kd[3], ks[3], ns; int32_t tex_slot; } oa_material;
#define OA_MAX_TEXTURE_SLOTS 16

static void oa_shade_pixel(const oa_scene *sc, v3 dir, v3 org, int has_texcoords, int has_normals,
                           const oa_texture tex[2], const oa_material *mats, const oa_texture *slots, oa_pixel *px) {
    float t = 1000000.0f;
    uint32_t tri = 0;
    v2 uv = { 0, 0 };
    px->counts[0] = px->counts[1] = 0;
    px->overflow = 0;
    oa_ray ray = { org, dir, { 1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z } };
    if (oa_traverse(sc, &ray, &t, &tri, &uv, px->counts)) px->overflow = 1;
    px->hit_t = t;
    v3 col;
    if (t < 1000.0f) {
        px->hit_id = tri;
        v3 normal = oa_tri_normal(sc, has_normals, tri, uv);
        v2 tc = oa_tri_texcoord(sc, has_texcoords, tri, uv);
        /* reflect(I,N) = I - 2*dot(N,I)*N */
        float two_d = 2.0f * oa_dot(normal, dir);
        v3 reflected = oa_sub(dir, oa_scale(normal, two_d));
        float shadow_t = 1000000.0f;
        v3 sd = sun_dir();
        uint32_t shadow_tri = 0;
        v2 suv = uv;
        oa_ray sray = { oa_add(oa_add(org, oa_scale(dir, t)), oa_scale(normal, 0.0001f)), sd, { 1.0f / sd.x, 1.0f / sd.y, 1.0f / sd.z } };
        if (oa_traverse(sc, &sray, &shadow_t, &shadow_tri, &suv, px->counts)) px->overflow = 1;
        int shadow = shadow_t < 1000.0f;
        px->shadow = shadow;
        uint32_t mat = sc->triangles[tri].mat;
        v3 ka = { AMBIENT, AMBIENT, AMBIENT }, kd = { DIFFUSE, DIFFUSE, DIFFUSE }, ks = { SPECULAR, SPECULAR, SPECULAR };
        float ns = SPEC_POW;
        v3 albedo;
        if (mats) {
            static const uint8_t white_rgb[3] = { 255, 255, 255 };
            const oa_texture white = { white_rgb, 1, 1 };
            const oa_material *m = &mats[mat];
            ka.x = m->ka[0]; ka.y = m->ka[1]; ka.z = m->ka[2];
            kd.x = m->kd[0]; kd.y = m->kd[1]; kd.z = m->kd[2];
            ks.x = m->ks[0]; ks.y = m->ks[1]; ks.z = m->ks[2];
            ns = m->ns;
            const oa_texture *t = (m->tex_slot >= 0 && slots[m->tex_slot].rgb) ? &slots[m->tex_slot] : &white;
            albedo = oa_sample_albedo(t, tc.x, -tc.y);
        } else {
            albedo = oa_sample_albedo(&tex[mat & 1u], tc.x, -tc.y);
        }
        v3 ambient = oa_mulv(oa_mulv(albedo, oa_skybox(normal)), ka);
        v3 diffuse = oa_mulv(oa_scale(oa_mulv(albedo, SUN_COL), oa_clamp01(oa_dot(normal, sd))), kd);
        v3 specular = oa_mulv(oa_scale(SUN_COL, oa_pow_clamp01(oa_dot(reflected, sd), ns)), ks);
        if (shadow) { diffuse = oa_scale(diffuse, 0.0f); specular = oa_scale(specular, 0.0f); }
        col = oa_add(oa_add(ambient, diffuse), specular);
    } else {
        px->hit_id = 0xFFFFFFFFu;
        px->shadow = -1;
        col = oa_skybox(dir);
    }
    px->rgba[0] = col.x; px->rgba[1] = col.y; px->rgba[2] = col.z; px->rgba[3] = 1.0f;
}

/* ---------------------------------------------------------------------- */
/* threaded drivers                                                        */
/* ---------------------------------------------------------------------- */
typedef struct {
    const oa_scene *sc;
    uint64_t n, begin, end;
    const float *dir, *org;
    int has_tc, has_n;
    oa_texture tex[2];
    const oa_material *mats; const oa_texture *slots;   /* NULL: the shader's constants and modelTextures[2] */
    float *rgba; uint32_t *hit_id; float *hit_t; uint32_t *counts; int8_t *shadow;
    float *hits; /* trace mode */
    int mode;    /* 0 render, 1 trace */
    int overflow;
} oa_job;

static void *oa_worker(void *arg) {
    oa_job *j = (oa_job *)arg;
    apply_fp_mode();
    for (uint64_t i = j->begin; i < j->end; i++) {
        v3 dir = { j->dir[3 * i], j->dir[3 * i + 1], j->dir[3 * i + 2] };
        v3 org = { j->org[3 * i], j->org[3 * i + 1], j->org[3 * i + 2] };
        if (j->mode == 0) {
            oa_pixel px;
            oa_shade_pixel(j->sc, dir, org, j->has_tc, j->has_n, j->tex, j->mats, j->slots, &px);
            if (px.overflow) j->overflow = 1;
            if (j->rgba) memcpy(j->rgba + 4 * i, px.rgba, 16);
            if (j->hit_id) j->hit_id[i] = px.hit_id;
            if (j->hit_t) j->hit_t[i] = px.hit_t;
            if (j->counts) { j->counts[2 * i] = px.counts[0]; j->counts[2 * i + 1] = px.counts[1]; }
            if (j->shadow) j->shadow[i] = (int8_t)px.shadow;
        } else {
            /* ray_trace.frag:131-143 */
            float *h = j->hits + 4 * i;
            uint32_t cnt[2] = { 0, 0 };
            if (dir.x == 0.0f && dir.y == 0.0f && dir.z == 0.0f) { /* discard: target keeps its clear value */
                h[0] = OA_MISS; h[1] = 0; h[2] = 0; h[3] = 0;
            } else {
                float t = OA_MISS;
                uint32_t tri = 0;
                v2 uv = { 0, 0 };
                oa_ray ray = { org, dir, { 1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z } };
                if (oa_traverse(j->sc, &ray, &t, &tri, &uv, cnt)) j->overflow = 1;
                h[0] = t; h[1] = uv.x; h[2] = uv.y;
                memcpy(&h[3], &tri, 4);
            }
            if (j->counts) { j->counts[2 * i] = cnt[0]; j->counts[2 * i + 1] = cnt[1]; }
        }
    }
    return NULL;
}

static int run_jobs(oa_job *proto, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    /* interleaved chunks so that sky rows and geometry rows spread over threads */
    const uint64_t chunk = 4096;
    uint64_t nchunks = (proto->n + chunk - 1) / chunk;
    if ((uint64_t)nthreads > nchunks) nthreads = (int)(nchunks ? nchunks : 1);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    oa_job *jobs = (oa_job *)malloc(sizeof(oa_job) * (size_t)nthreads);
    /* contiguous split of chunk indices; good enough for a baseline */
    int overflow = 0;
    for (int k = 0; k < nthreads; k++) {
        jobs[k] = *proto;
        uint64_t c0 = nchunks * (uint64_t)k / (uint64_t)nthreads, c1 = nchunks * (uint64_t)(k + 1) / (uint64_t)nthreads;
        jobs[k].begin = c0 * chunk;
        jobs[k].end = c1 * chunk < proto->n ? c1 * chunk : proto->n;
        jobs[k].overflow = 0;
    }
    if (nthreads == 1) {
        unsigned csr = _mm_getcsr();
        oa_worker(&jobs[0]);
        _mm_setcsr(csr);
    } else {
        for (int k = 0; k < nthreads; k++) pthread_create(&th[k], NULL, oa_worker, &jobs[k]);
        for (int k = 0; k < nthreads; k++) pthread_join(th[k], NULL);
    }
    for (int k = 0; k < nthreads; k++) overflow |= jobs[k].overflow;
    free(th); free(jobs);
    return overflow ? -1 : 0;
}

/* Fused shader over n pixels.  Any output pointer may be NULL.  Returns -1 if a
 * traversal would have overflowed the shader's 100-entry stack (results for
 * that pixel are then not meaningful — in GLSL it is undefined behaviour). */
int oa_render(const oa_scene *sc, uint64_t n, const float *dir, const float *org, int has_texcoords, int has_normals,
              const uint8_t *tex0, int tw0, int th0, const uint8_t *tex1, int tw1, int th1,
              float *rgba, uint32_t *hit_id, float *hit_t, uint32_t *counts, int8_t *shadow, int nthreads) {
    oa_job j;
    memset(&j, 0, sizeof j);
    j.sc = sc; j.n = n; j.dir = dir; j.org = org; j.has_tc = has_texcoords; j.has_n = has_normals;
    j.tex[0].rgb = tex0; j.tex[0].w = tw0; j.tex[0].h = th0;
    j.tex[1].rgb = tex1; j.tex[1].w = tw1; j.tex[1].h = th1;
    j.rgba = rgba; j.hit_id = hit_id; j.hit_t = hit_t; j.counts = counts; j.shadow = shadow; j.mode = 0;
    return run_jobs(&j, nthreads);
}

/* The fused shader with a material table: n_mats entries (every triangle's mat must be < n_mats,
 * else -2), textures[OA_MAX_TEXTURE_SLOTS] with rgb == NULL for an empty slot (samples white). */
int oa_render_materials(const oa_scene *sc, uint64_t n, const float *dir, const float *org, int has_texcoords, int has_normals,
                        const oa_material *mats, uint32_t n_mats, const uint8_t *const *tex_rgb, const int *tex_w, const int *tex_h,
                        float *rgba, uint32_t *hit_id, float *hit_t, uint32_t *counts, int8_t *shadow, int nthreads) {
    oa_texture slots[OA_MAX_TEXTURE_SLOTS];
    for (uint32_t i = 0; i < sc->n_triangles; i++) if (sc->triangles[i].mat >= n_mats) return -2;
    for (uint32_t i = 0; i < n_mats; i++) if (mats[i].tex_slot < -1 || mats[i].tex_slot >= OA_MAX_TEXTURE_SLOTS) return -2;
    for (int k = 0; k < OA_MAX_TEXTURE_SLOTS; k++) { slots[k].rgb = tex_rgb[k]; slots[k].w = tex_w[k]; slots[k].h = tex_h[k]; }
    oa_job j;
    memset(&j, 0, sizeof j);
    j.sc = sc; j.n = n; j.dir = dir; j.org = org; j.has_tc = has_texcoords; j.has_n = has_normals;
    j.mats = mats; j.slots = slots;
    j.rgba = rgba; j.hit_id = hit_id; j.hit_t = hit_t; j.counts = counts; j.shadow = shadow; j.mode = 0;
    return run_jobs(&j, nthreads);
}

/* Split-pipeline trace shader over n rays; hits = (t,u,v,bits(idx)) per ray. */
int oa_trace(const oa_scene *sc, uint64_t n, const float *org, const float *dir, float *hits, uint32_t *counts, int nthreads) {
    oa_job j;
    memset(&j, 0, sizeof j);
    j.sc = sc; j.n = n; j.dir = dir; j.org = org; j.hits = hits; j.counts = counts; j.mode = 1;
    return run_jobs(&j, nthreads);
}

/* ray_create.frag:12 evaluated at (off_x, off_y) inside each pixel, row 0 = bottom.  (0.5, 0.5) is the pixel centre
 * the reference rasterises (quad-11.vert:8); other offsets are the jittered samples of multi-sample frames, which the
 * reference does not have (one sample per pixel). */
/* ray_create.frag:12 for one value of the varying fragPos = (x, y) */
static inline v3 oa_raycreate_one(const float *m, float near_, float far_, float x, float y) {
    float v[4] = { x * far_ - x * near_, y * far_ - y * near_, 1.0f * far_ - (-1.0f) * near_, 1.0f * far_ - 1.0f * near_ };
    v3 d;
    float r[3];
    for (int k = 0; k < 3; k++) r[k] = ((m[0 + k] * v[0] + m[4 + k] * v[1]) + m[8 + k] * v[2]) + m[12 + k] * v[3];
    d.x = r[0]; d.y = r[1]; d.z = r[2];
    return oa_normalize(d);
}

void oa_raycreate_offset(uint32_t w, uint32_t h, const float *m /* column-major invProjView */, float near_, float far_,
                         float off_x, float off_y, float *dir_out) {
    for (uint32_t py = 0; py < h; py++) {
        float y = ((float)py + off_y) / (float)h * 2.0f - 1.0f;
        for (uint32_t px = 0; px < w; px++) {
            float x = ((float)px + off_x) / (float)w * 2.0f - 1.0f;
            const v3 d = oa_raycreate_one(m, near_, far_, x, y);
            float *o = dir_out + 3 * ((size_t)py * w + px);
            o[0] = d.x; o[1] = d.y; o[2] = d.z;
        }
    }
}

/* The same for n explicit values of the varying (frag_pos: n x 2).  The varying is the rasteriser's product, not the
 * shader's: tests/golden/soak_oracle_vs_llvmpipe.py reads llvmpipe's own values back and checks the shader arithmetic on them. */
void oa_raycreate_points(uint64_t n, const float *frag_pos, const float *m, float near_, float far_, float *dir_out) {
    for (uint64_t i = 0; i < n; i++) {
        const v3 d = oa_raycreate_one(m, near_, far_, frag_pos[2 * i], frag_pos[2 * i + 1]);
        dir_out[3 * i] = d.x; dir_out[3 * i + 1] = d.y; dir_out[3 * i + 2] = d.z;
    }
}

void oa_raycreate(uint32_t w, uint32_t h, const float *m, float near_, float far_, float *dir_out) {
    oa_raycreate_offset(w, h, m, near_, far_, 0.5f, 0.5f, dir_out);
}

/* ray_trace/ray_dispatcher.frag:26-41 over a w x h G-buffer (row 0 = bottom): secondary-ray
 * origins and the shadow / reflection / ambient-occlusion directions.  position: 3 floats
 * per pixel (RGB32F, NEAREST, framebuffer_manager.rs:33), normal_mat: 4 floats per pixel
 * (w == 1e30 marks "no surface": the fragment is discarded and the four targets keep their
 * clear value 0 = NO_RAY, main.rs:157-158).  fragPos is the quad01.vert varying at the
 * pixel centre; the blue-noise texture is an ordinary image texture (LINEAR, REPEAT).
 * materials[] is read at :36 but never used.  No threads: one memory-bound pass. */
/* Test-only: values of the varying fragPos to use instead of the pixel centres (w*h x 2, NULL = centres).  The varying is the
 * rasteriser's product; between texels an ulp of it can move an 8-bit filter weight, so the soak hands llvmpipe's own values in. */
static const float *g_dispatch_frag_pos = NULL;
void oa_set_dispatch_frag_pos(const float *xy) { g_dispatch_frag_pos = xy; }

void oa_dispatch(uint32_t w, uint32_t h, const float *position, const float *normal_mat,
                 const uint8_t *noise_rgb, int nw, int nh,
                 const float *light_pos, const float *camera_pos, const float *noise_offset_scale,
                 float *org, float *shadow_dir, float *reflect_dir, float *ambient_dir) {
    unsigned csr = _mm_getcsr();
    apply_fp_mode();
    oa_texture noise = { noise_rgb, nw, nh };
    const v3 light = { light_pos[0], light_pos[1], light_pos[2] }, cam = { camera_pos[0], camera_pos[1], camera_pos[2] };
    for (uint32_t py = 0; py < h; py++) {
        const float fy = ((float)py + 0.5f) / (float)h;
        for (uint32_t px = 0; px < w; px++) {
            const size_t i = (size_t)py * w + px;
            const float fx = ((float)px + 0.5f) / (float)w;
            float *o = org + 3 * i, *sd = shadow_dir + 3 * i, *rd = reflect_dir + 3 * i, *ad = ambient_dir + 3 * i;
            const float *nm = normal_mat + 4 * i;
            if (nm[3] == 1e30f) {                                  /* :28 discard */
                for (int k = 0; k < 3; k++) o[k] = sd[k] = rd[k] = ad[k] = 0.0f;
                continue;
            }
            const v3 n = { nm[0], nm[1], nm[2] };
            const v3 p = { position[3 * i], position[3 * i + 1], position[3 * i + 2] };
            const float px_x = g_dispatch_frag_pos ? g_dispatch_frag_pos[2 * i] : fx, px_y = g_dispatch_frag_pos ? g_dispatch_frag_pos[2 * i + 1] : fy;
            const v3 rnd = oa_sample_albedo(&noise, px_x * noise_offset_scale[2] + noise_offset_scale[0],
                                            px_y * noise_offset_scale[3] + noise_offset_scale[1]);   /* :34 */
            const v3 og = oa_add(p, oa_scale(n, 0.0001f));          /* :38 */
            const v3 s = oa_normalize(oa_sub(light, p));            /* :39 */
            const v3 inc = oa_sub(p, cam);
            const float two_d = 2.0f * oa_dot(n, inc);              /* reflect(I,N) = I - 2*dot(N,I)*N */
            const v3 r = oa_normalize(oa_sub(inc, oa_scale(n, two_d)));   /* :41 */
            const v3 av = { (n.x + rnd.x * 2.0f) - 1.0f, (n.y + rnd.y * 2.0f) - 1.0f, (n.z + rnd.z * 2.0f) - 1.0f };
            const v3 a = oa_normalize(av);                          /* :42 */
            o[0] = og.x; o[1] = og.y; o[2] = og.z;
            sd[0] = s.x; sd[1] = s.y; sd[2] = s.z;
            rd[0] = r.x; rd[1] = r.y; rd[2] = r.z;
            ad[0] = a.x; ad[1] = a.y; ad[2] = a.z;
        }
    }
    _mm_setcsr(csr);
}

/* Hit records (t,u,v,bits(idx)) of ray_trace.frag -> the G-buffer texels the dispatcher reads
 * (what the reference's rasteriser pass writes, src/main.rs:127-140, restated for ray hits):
 * position = org + dir * t, normal = triangleNormal (ray_trace_and_shade.frag:165-178),
 * w = bits(mat_idx), or position 0 / (0,0,0,1e30) for a miss or NO_RAY record. */
void oa_surface_from_hits(const oa_scene *sc, uint64_t n, const float *org, const float *dir, const float *hits, int has_normals,
                          float *position, float *normal_mat) {
    unsigned csr = _mm_getcsr();
    apply_fp_mode();
    for (uint64_t i = 0; i < n; i++) {
        const float *h = hits + 4 * i;
        float *p = position + 3 * i, *nm = normal_mat + 4 * i;
        if (!(h[0] < OA_MISS)) {
            p[0] = p[1] = p[2] = 0.0f;
            nm[0] = nm[1] = nm[2] = 0.0f; nm[3] = OA_MISS;
            continue;
        }
        uint32_t idx;
        memcpy(&idx, &h[3], 4);
        v2 uv = { h[1], h[2] };
        v3 nrm = oa_tri_normal(sc, has_normals, idx, uv);
        for (int k = 0; k < 3; k++) p[k] = org[3 * i + k] + dir[3 * i + k] * h[0];
        nm[0] = nrm.x; nm[1] = nrm.y; nm[2] = nrm.z;
        memcpy(&nm[3], &sc->triangles[idx].mat, 4);
    }
    _mm_setcsr(csr);
}
