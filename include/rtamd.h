/*
 * rtamd.h — C-ABI drop-in boundary for the ray-tracing hot path on MI355X (gfx950).
 *
 * This header is everything the reference's host would bind (bindgen / cc in
 * build.rs) to replace its OpenGL full-screen-pass dispatch of
 * res/shaders/ray_trace_and_shade.frag (and of the split-pipeline sibling
 * res/shaders/ray_trace/ray_trace.frag) by hand-written HIP kernels.  Plain
 * pointers and sizes only; no C++ or torch types.  All citations are relative
 * to the reference tree.
 *
 * Conventions
 *   - every function returns an rtamd_status (0 = ok, negative = error) and never
 *     unwinds across the boundary; rtamd_last_error() gives the text (the
 *     reference maps errors to Result<_, enum>, src/util/error.rs:4-59);
 *   - inputs are borrowed for the duration of the call and copied to HBM, as
 *     glBufferData does at src/gl_wrapper/buffer.rs:9-19; outputs go into
 *     caller-provided buffers;
 *   - a context is owned by one caller thread at a time (the reference's GL
 *     context is single-threaded too), one context per GPU / per process rank;
 *   - "host" entry points take host pointers and are synchronous; "_device"
 *     entry points take device pointers (HBM-resident inputs/outputs), enqueue on
 *     the context's stream and return without waiting;
 *   - pixel i of a frame uses ray i and produces output i: the library never
 *     flips rows.  With on-device ray generation row 0 is the BOTTOM row (GL
 *     window convention, util/quad-11.vert:8).
 *   - the library has no CPU fallback: without a usable GPU every compute entry
 *     point fails with RTAMD_ERR_DEVICE.
 */
#ifndef RTAMD_H
#define RTAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rtamd_status {
    RTAMD_OK = 0,
    RTAMD_ERR_INVALID_ARGUMENT = -1, /* NULL / zero-sized / inconsistent argument */
    RTAMD_ERR_EMPTY_SCENE = -2,      /* the reference panics on &data[0] of an empty Vec (buffer.rs:15) */
    RTAMD_ERR_NO_SCENE = -3,         /* render/trace before rtamd_upload_scene */
    RTAMD_ERR_MALFORMED_SCENE = -4,  /* node / triangle / vertex index out of range, cyclic or too deep tree */
    RTAMD_ERR_DEVICE = -5,           /* HIP runtime error or no GPU */
    RTAMD_ERR_OUT_OF_MEMORY = -6
} rtamd_status;

typedef struct rtamd_context rtamd_context;

/* Value written to hit_id outputs for pixels whose primary ray misses
 * (the fused shader leaves triangleIdx at its initial 0 and shades the sky,
 * ray_trace_and_shade.frag:201,232-234). */
/* Bumped whenever a struct layout or the meaning of an argument changes; a binder compares its compile-time
 * value with rtamd_abi_version() of the library it loaded.  (No counterpart in the reference.) */
#define RTAMD_ABI_VERSION 8
uint32_t rtamd_abi_version(void);

#define RTAMD_NO_HIT 0xFFFFFFFFu

/* Framebuffer tile used to shard a frame over ranks (interleaved tiles, scene
 * replicated): tile k (row-major over ceil(W/32) x ceil(H/32)) belongs to rank
 * k % tile_world. */
#define RTAMD_TILE_DIM 32u

/* ---- context ---------------------------------------------------------- */

/* Replaces GL context + program creation (src/main.rs:23,33-50). */
int rtamd_context_create(int device_ordinal, rtamd_context **out_ctx);
void rtamd_context_destroy(rtamd_context *ctx);

/* Text of the last error on this context (ctx == NULL: last error of a failed
 * rtamd_context_create on this thread).  Never NULL. */
const char *rtamd_last_error(const rtamd_context *ctx);

/* Use an existing hipStream_t (passed as void*) for all "_device" work of this
 * context from now on; NULL restores the context's own stream.
 *
 * Frames in flight: launches on one stream run in stream order; launches issued on
 * DIFFERENT streams may overlap on the GPU, and are meant to: the tail of one frame (its
 * last, slowest work items) then runs beside the start of the next, which on MI355X raises
 * the frame rate of the fused shader by about 1.4x (rtamd_render_batch_device
 * gets the same from ONE launch on one stream, and is the better tool).  The context keeps
 * separate launch scratch (queue counters, stack-spill area, timing events) per stream for
 * up to 8 streams, so a caller pipelines frames by alternating two streams of its own:
 *     set_stream(s[k % 2]); rtamd_render_device(..., out[k % 2]);
 * and orders consumers with ordinary stream/event dependencies.  Streams overlap only when
 * they sit on different hardware queues: HIP spreads a process's streams over
 * GPU_MAX_HW_QUEUES queues (default 4; read once when the runtime starts), so set it to 16 in
 * the environment when more than two frames are in flight.  A ninth stream takes over
 * the least recently used scratch after that one's last launch has finished.  Synchronize a
 * stream before destroying it.  (The reference's passes are serialised by the GL command
 * stream, src/main.rs:143-230; this has no counterpart there.) */
int rtamd_context_set_stream(rtamd_context *ctx, void *hip_stream);

/* Block until all work this context has enqueued on its CURRENT stream has finished. */
int rtamd_synchronize(rtamd_context *ctx);

/* ---- scene upload ------------------------------------------------------ */

/* Exactly the five slices the reference uploads as SSBOs at src/main.rs:93-102
 * (bindings 0..4 of ray_trace_and_shade.frag:47-51):
 *   nodes      BVH::data()            src/raytracing/bvh.rs:49-51     36-byte records
 *                                     (6 x f32 bounds, u32 is_leaf, u32 a, u32 b; types.rs:101-109)
 *   triangles  Model::triangles()     src/rendering/model.rs:20       16-byte records (p0,p1,p2,mat_idx), leaf order
 *   positions  Model::positions()     model.rs:22                     3 x f32 per vertex
 *   tex_coords Model::tex_coords()    model.rs:23   optional (NULL)   2 x f32 per vertex
 *   normals    Model::normals()       model.rs:24   optional (NULL)   3 x f32 per vertex
 */
typedef struct rtamd_scene_desc {
    const void *nodes;
    uint32_t node_count;
    const void *triangles;
    uint32_t triangle_count;
    const float *positions;
    uint32_t vertex_count;
    const float *tex_coords;
    const float *normals;
} rtamd_scene_desc;

/* Copies the scene to HBM and builds the device layout (64-byte two-child node
 * records, 48-byte precomputed-edge triangle records).  Validates every index.
 * Replaces ShaderStorageBuffer::new + buffer_data (src/gl_wrapper/buffer.rs:84-108). */
int rtamd_upload_scene(rtamd_context *ctx, const rtamd_scene_desc *scene);

/* Albedo texture for a slot: tightly packed RGB8 rows, first row first, as
 * Texture::from_data hands them to glTexImage2D (src/gl_wrapper/texture.rs:11-26,
 * 65-79).  Sampling is LINEAR min/mag, REPEAT wrap, no mip maps, RGBA8 unorm.
 * A slot never uploaded samples as opaque white.  Slots 0 and 1 are the shader's
 * modelTextures[2] (ray_trace_and_shade.frag:21), selected by the triangle's material
 * index; slots 2..RTAMD_MAX_TEXTURE_SLOTS-1 are reachable only through a material table
 * (rtamd_upload_materials). */
#define RTAMD_MAX_TEXTURE_SLOTS 16
int rtamd_upload_texture(rtamd_context *ctx, uint32_t slot, const uint8_t *rgb8, uint32_t width, uint32_t height);

/* Per-material shading for the fused render.  The reference parses
 * Ka / Kd / Ks / Ns and map_Kd into Material (src/rendering/material.rs:5-16,
 * resource_parser.rs:49-109) but its shader ignores them: AMBIENT 0.5, DIFFUSE 0.8,
 * SPECULAR 0.4, SPEC_POW 30 are constants (ray_trace_and_shade.frag:184-187) and only two
 * textures exist (:21).  With a table, pixel shading uses the hit triangle's material
 * (mat_idx, src/raytracing/types.rs:4-9) in the constants' places, component-wise:
 *     ambient  = albedo * skybox(n) * ambient_color
 *     diffuse  = albedo * SUN_COL * clamp(dot(n, SUN_DIR)) * diffuse_color
 *     specular = SUN_COL * clamp(pow(dot(reflected, SUN_DIR), specular_exp)) * specular_color
 * with albedo from texture_slot (-1: none = white).  The expressions are the shader's own
 * (:222-224), so a table of {0.5, 0.8, 0.4, 30, slot = material index} entries reproduces
 * the reference output bit for bit (tested); any other table has no reference output to
 * compare with and is checked against the CPU oracle only.
 * `count` must exceed every mat_idx of the uploaded scene; count 0 removes the table.
 * Uploading a scene removes the table.  No counterpart in the reference's shader inputs. */
typedef struct rtamd_shading_material {
    float ambient_color[3], diffuse_color[3], specular_color[3];   /* Ka, Kd, Ks */
    float specular_exp;                                            /* Ns */
    int32_t texture_slot;                                          /* map_Kd: 0..RTAMD_MAX_TEXTURE_SLOTS-1, or -1 */
} rtamd_shading_material;
int rtamd_upload_materials(rtamd_context *ctx, const rtamd_shading_material *materials, uint32_t count);

/* ---- fused render: ray_trace_and_shade.frag:main (:195-235) --------------- */

typedef struct rtamd_frame_desc {
    uint32_t width, height;
    /* Per-pixel rays: the `dir` / `org` RGB32F textures (uniform locations 0, 1;
     * :17-18,196-197), 3 floats per pixel.  If dir is NULL the rays are generated
     * on the device from the camera fields below (ray_create.frag:12). */
    const float *dir;
    const float *org;
    /* Camera for on-device ray generation: inverse(proj*view), column-major
     * (src/main.rs:149), near/far (camera.rs:6-7) and the eye position, which is
     * the origin of every primary ray. */
    float inv_proj_view[16];
    float near_plane, far_plane;
    float cam_pos[3];
    /* Uniform bools at locations 2 / 3 (:19-20). */
    uint32_t has_tex_coords;
    uint32_t has_normals;
    /* Interleaved-tile shard of the frame this call renders (tile_world >= 1,
     * tile_rank < tile_world).  tile_world == 1 renders the whole frame. */
    uint32_t tile_rank, tile_world;
} rtamd_frame_desc;

/* Output buffers; any pointer may be NULL (that output is skipped).
 * Layout when desc.tile_world == 1: scan-line order, index = y*width + x.
 * Layout when desc.tile_world  > 1: compact tile-major — local tile j (global
 * tile j*tile_world + tile_rank) occupies pixels [j*1024, (j+1)*1024), row-major
 * 32x32 inside the tile; see rtamd_tiles_per_rank / rtamd_assemble_tiles_device. */
typedef struct rtamd_frame_out {
    float *rgba32f;    /* fragCol (:231,233), 4 floats per pixel, unclamped */
    uint32_t *hit_id;  /* closest-hit triangle index (leaf order) or RTAMD_NO_HIT */
    float *hit_t;      /* primary t, or 1e6 (the shader's initial value, :199) on a miss */
    uint32_t *counts;  /* 2 per pixel: AABB tests, triangle tests (the shader's `intersections`, :115,123,138, split by kind; primary + shadow) */
    uint32_t *rgba8;   /* the pixel as the display pass writes it to the 8-bit back buffer (util/display.frag:9; GL float -> unorm8:
                        * clamp, x255, round to nearest), R in the low byte: rtamd_display_rgba8_device applied to rgba32f (which
                        * must be given too) by the same call, behind the render kernel on the same stream */
} rtamd_frame_out;

/* Host buffers in, host buffers out; synchronous.  Replaces bind + set_uniform_* +
 * quad_geometry.draw() + read-back (src/main.rs:204-230 pattern; shader.rs:128-174;
 * geometry_set.rs:34-44). */
int rtamd_render(rtamd_context *ctx, const rtamd_frame_desc *desc, const rtamd_frame_out *out);

/* Same, with desc->dir / desc->org and every pointer in `out` being DEVICE
 * pointers; enqueues on the context's stream and returns. */
int rtamd_render_device(rtamd_context *ctx, const rtamd_frame_desc *desc, const rtamd_frame_out *out);

/* Several frames in ONE launch.  The reference's frame loop draws one frame after another (src/main.rs:105-243); a
 * renderer that has the next frames' cameras at hand — or, on a multi-GPU split, the frames of one gather — submits them
 * together: the persistent kernel then pulls the work items of all n_frames frames from one queue, frame after frame, so
 * the slow tail of a frame (its last, most expensive items, during which most of the GPU idles) runs beside the next
 * frame's items.  On MI355X that is worth what several streams with a frame each are worth (about 1.7x the frame rate of
 * one frame at a time on the 1080p bench scene) without their needs: no extra hardware queues, one counter reset and
 * one launch per batch.  Frame f is rendered from descs[f] into outs[f], with exactly the bits rtamd_render_device
 * gives for it.  The frames share width, height, tile shard and the two has_* flags; rays / camera and every output
 * pointer are per frame (outputs may be NULL per frame; rgba8 per frame as in rtamd_render_device).  Limits: 1..
 * RTAMD_MAX_BATCH_FRAMES frames, fewer than 2^28 pixels per frame when n_frames > 1.  Device pointers; enqueues on the
 * context's stream and returns.  No counterpart in the reference. */
#define RTAMD_MAX_BATCH_FRAMES 16
int rtamd_render_batch_device(rtamd_context *ctx, uint32_t n_frames, const rtamd_frame_desc *descs, const rtamd_frame_out *outs);

/* rtamd_render_device for callers that cannot allocate: tight serving loops, harnesses that forbid hipMalloc.  No
 * counterpart in the reference.  Same arguments, same outputs bit for bit, but the call allocates nothing (no hipMalloc /
 * hipFree, no event or stream creation): the launch scratch the persistent kernel needs -- 8 work-queue counters and the
 * HBM columns that take the bottom of deep traversal stacks -- lives in a static device array of the library's code
 * object, 64 MiB + 2 KiB, which every GPU the library runs on holds until the process ends.  Limits: the uploaded
 * scene's tree may be at most 48 levels deep (rtamd_scene_info.max_depth; 16 stack entries per lane in LDS + 32 in the
 * static array), else RTAMD_ERR_INVALID_ARGUMENT; frame size as for rtamd_render_device.  It always runs the persistent
 * kernel with the context's fixed lane re-fill threshold (no coherence feedback, no per-launch timing events:
 * rtamd_last_kernel_ms / rtamd_last_launch_coherence do not see it).
 * All calls on one GPU share that GPU's array -- across streams and across contexts.  The library orders them itself: a
 * call's counter reset and kernel wait on the GPU (hipStreamWaitEvent on one per-device event created when the first
 * context on that GPU is) for the previous static call's kernel, so static calls never overlap one another, whatever
 * streams they are issued on; they may overlap every other kind of work.  The array keeps the last call's contents; the
 * call clears the counters itself (one 2 KB hipMemsetAsync on the context's stream).  Not meant for stream capture: the
 * call makes host-side look-ups (buffer range checks, current device) and waits on an event recorded outside a capture. */
int rtamd_render_device_static(rtamd_context *ctx, const rtamd_frame_desc *desc, const rtamd_frame_out *out);

/* ---- split-pipeline trace: ray_trace/ray_trace.frag:main (:131-143) -------- */

/* n rays in, n hit records out: hits[4*i..] = (t, u, v, uintBitsToFloat(triangle index)).
 * t starts at MISS = 1e30 (:136); a ray with dir == (0,0,0) is "no ray" (:133) and
 * yields the clear value (1e30, 0, 0, 0) that src/main.rs:179 puts in the target. */
int rtamd_trace(rtamd_context *ctx, uint64_t n, const float *org, const float *dir, float *hits);
int rtamd_trace_device(rtamd_context *ctx, uint64_t n, const float *org, const float *dir, float *hits);

/* Several ray lists in ONE launch: the persistent kernel pulls the work items of all n_jobs lists from one queue, so the
 * slow tail of a list (its last, most expensive rays, during which most of the GPU idles) runs beside the next list's rays —
 * what rtamd_render_batch_device does for frames.  The reference draws its three ray_trace passes one after the other
 * (src/main.rs:176-203: shadow, reflection, ambient rays of one dispatcher pass); they are independent, and so are the
 * passes of different samples of a multi-sample frame: hand them over together.  Job j's records are bit for bit those
 * of rtamd_trace_device(jobs[j]).  Limits: 1..RTAMD_MAX_BATCH_FRAMES lists, fewer than 2^28 rays per list when
 * n_jobs > 1; a list may be empty.  Device pointers; enqueues on the context's stream and returns. */
typedef struct rtamd_trace_job {
    uint64_t n;           /* rays in this list */
    const float *org;     /* 3 floats per ray */
    const float *dir;     /* 3 floats per ray; (0,0,0) = NO_RAY */
    float *hits;          /* 4 floats per ray: (t, u, v, bits(triangle index)) */
} rtamd_trace_job;
int rtamd_trace_batch_device(rtamd_context *ctx, uint32_t n_jobs, const rtamd_trace_job *jobs);

/* The same for ray lists that are IMAGES: list j holds one ray per pixel, in scan-line order with `width` rays per row
 * (tile_world <= 1; jobs[j].n a multiple of width) or in the compact tile-major shard layout (tile_world > 1: whole 32x32
 * tiles, row-major inside a tile; jobs[j].n a multiple of 1024; width is not used).  Which ray is traced with which is the
 * only difference: a work item is an 8x8 pixel block of the image -- the fused shader's item shape -- instead of 64
 * consecutive rays (a 64x1 strip of it), so that the rays of a wave stay together longer.  Records are bit for bit those of
 * rtamd_trace_batch_device, in the same places.  This is how the ray_trace passes of src/main.rs:176-203 run on their
 * full-screen targets; rtamd_multibounce_sample_batch_device uses it for every list.  1 <= n_jobs <= 16. */
int rtamd_trace_image_batch_device(rtamd_context *ctx, uint32_t n_jobs, const rtamd_trace_job *jobs, uint32_t width, uint32_t tile_world);

/* ---- ray generation: ray_trace/ray_create.frag:12 + util/quad-11.vert:8 ---- */

/* dir_out: 3 floats per pixel, scan-line order, row 0 = bottom row. */
int rtamd_generate_rays(rtamd_context *ctx, uint32_t width, uint32_t height, const float inv_proj_view[16],
                        float near_plane, float far_plane, float *dir_out);
int rtamd_generate_rays_device(rtamd_context *ctx, uint32_t width, uint32_t height, const float inv_proj_view[16],
                               float near_plane, float far_plane, float *dir_out);

/* As rtamd_generate_rays_device with the sample at (offset_x, offset_y) inside each pixel instead of the
 * centre (0.5, 0.5): jittered primaries for multi-sample frames (BASELINE config 5).  No counterpart in
 * the reference, which rasterises one sample per pixel. */
int rtamd_generate_rays_offset_device(rtamd_context *ctx, uint32_t width, uint32_t height, const float inv_proj_view[16],
                                      float near_plane, float far_plane, float offset_x, float offset_y, float *dir_out);

/* As rtamd_generate_rays_offset_device for one rank's tile shard of the frame (BASELINE config 5 split over GPUs: the
 * samples of a pixel stay on the GPU that owns its tile): dir_out holds rtamd_tiles_per_rank-style local tiles * 1024
 * rays in the compact tile-major layout, entries outside the frame are (0,0,0) = NO_RAY.  tile_world == 1 is
 * rtamd_generate_rays_offset_device.  No counterpart in the reference (one GPU, one sample). */
int rtamd_generate_rays_tiles_device(rtamd_context *ctx, uint32_t width, uint32_t height, const float inv_proj_view[16],
                                     float near_plane, float far_plane, float offset_x, float offset_y,
                                     uint32_t tile_rank, uint32_t tile_world, float *dir_out);

/* ---- hit records -> G-buffer ------------------------------------------------ */

/* For ray i with hit record hits[4i..] = (t, u, v, bits(idx)) (rtamd_trace): position[3i..] = org + dir * t,
 * normal_mat[4i..] = (triangleNormal(idx, uv) (ray_trace_and_shade.frag:165-178), bits(mat_idx)); a miss or
 * NO_RAY record (t = 1e30) gives position 0 and normal_mat (0,0,0,1e30), the G-buffer's "no surface" value
 * (src/main.rs:130).  This is the input rtamd_dispatch_rays takes: the reference fills it with its rasteriser
 * pass (src/main.rs:127-140) and therefore stops at one bounce; with this, dispatcher -> trace can repeat. */
int rtamd_surface_from_hits(rtamd_context *ctx, uint64_t n, const float *org, const float *dir, const float *hits, uint32_t has_normals,
                            float *position, float *normal_mat);
int rtamd_surface_from_hits_device(rtamd_context *ctx, uint64_t n, const float *org, const float *dir, const float *hits, uint32_t has_normals,
                                   float *position, float *normal_mat);

/* ---- secondary-ray generation: ray_trace/ray_dispatcher.frag:26-41 ---------- */

/* The blue-noise texture (uniform location 2, res/textures/blue_noise.png via
 * ResourceManager::get_texture, src/main.rs:53,171): an ordinary image texture, same
 * upload and sampling rules as rtamd_upload_texture. */
int rtamd_upload_noise_texture(rtamd_context *ctx, const uint8_t *rgb8, uint32_t width, uint32_t height);

typedef struct rtamd_dispatch_desc {
    uint32_t width, height;
    const float *position;        /* G-buffer positionData (:18): 3 floats per pixel, row 0 = bottom */
    const float *normal_mat;      /* G-buffer normalMatData (:19): 4 floats per pixel, xyz normal, w = material
                                   * index bits, or 1e30 = "no surface" (clear value, src/main.rs:130) */
    float light_pos[3];           /* uniform 3 */
    float camera_pos[3];          /* uniform 4 */
    float noise_offset_scale[4];  /* uniform 5: xy offset, zw scale (src/main.rs:161-166) */
    /* Tile shard (rtamd_dispatch_rays_device only; ABI 7): with tile_world > 1 the G-buffer and the four targets hold this
     * rank's tiles in the compact tile-major layout of rtamd_frame_out (local tiles * 1024 entries), and an entry's
     * noise texel is the one of its pixel in the whole frame, so a shard's rays are bit for bit the whole frame's.
     * Entries of edge tiles outside the frame are "no surface".  tile_world 0 or 1: the whole frame, scan-line order. */
    uint32_t tile_rank, tile_world;
} rtamd_dispatch_desc;

typedef struct rtamd_dispatch_out {
    /* the four RGB32F colour attachments of the ray buffer (src/main.rs:68-72), 3 floats per pixel;
     * any may be NULL.  Pixels without a surface get (0,0,0) = NO_RAY, the targets' clear value. */
    float *org, *shadow_dir, *reflect_dir, *ambient_dir;
} rtamd_dispatch_out;

/* One pass of the dispatcher over the G-buffer.  The outputs feed rtamd_trace[_device]
 * (org + one direction array per ray kind), exactly as src/main.rs:176-203 feeds ray_trace.frag. */
int rtamd_dispatch_rays(rtamd_context *ctx, const rtamd_dispatch_desc *desc, const rtamd_dispatch_out *out);
int rtamd_dispatch_rays_device(rtamd_context *ctx, const rtamd_dispatch_desc *desc, const rtamd_dispatch_out *out);

/* ---- multi-sample, multi-bounce chain: src/main.rs:143-203 repeated ---------------------------------- */

/* One sample of one frame, as the reference's frame loop wires it (src/main.rs:143-203) and repeated per bounce:
 *     primaries      ray_create.frag:12 at sample position s inside the pixel      rtamd_generate_rays_tiles_device
 *     closest hit    ray_trace.frag:131-143                                         rtamd_trace[_batch]_device
 *     for depth 1..max_depth:
 *         G-buffer   the rasteriser pass's outputs, from the hit records            rtamd_surface_from_hits_device
 *         dispatcher ray_dispatcher.frag:26-41                                       rtamd_dispatch_rays_device
 *         shadow ray ray_trace.frag on (org, shadow_dir)                             -> shadow_hits
 *         continuation (not after the last depth): ray_trace.frag on (org, reflect_dir) for even s, (org, ambient_dir)
 *                    for odd s                                                       -> hits (replacing the previous bounce's)
 * rtamd_multibounce_sample_batch_device runs this for up to 8 samples in lock step: every stage's traversals of all
 * samples go out as ONE rtamd_trace_batch_device launch (n primaries; then n shadow + n continuation lists per bounce), the
 * per-pixel stages as one launch per sample, all on the context's stream; it returns without waiting.  It is launches and
 * buffer rotation only: every buffer holds, bit for bit, what the single calls named above write.  Sample s sits at
 * (frac(0.5 + s*g), frac(0.5 + s*g*g)) inside the pixel (R2 sequence, g = 0.7548776662466927; s = 0 is the pixel centre the
 * reference rasterises); the dispatcher's noise offset for (s, depth) is ((7s + 3 depth) mod noise_size,
 * (11s + 5 depth) mod noise_size) / noise_size with scale (width, height) / noise_size (src/main.rs:161-166 uses a random
 * offset per frame).  No colour comes out: the reference's consumer of these records (shader.frag:120-124) is a stub. */
typedef struct rtamd_multibounce_desc {
    uint32_t width, height;
    float inv_proj_view[16];
    float near_plane, far_plane;
    float cam_pos[3];
    float light_pos[3];
    uint32_t has_normals;
    uint32_t max_depth;              /* bounces, 1..16 */
    uint32_t noise_size;             /* texel grid the noise offsets step over (the reference's blue_noise.png: 512) */
    uint32_t tile_rank, tile_world;  /* tile shard of the frame (tile_world 0 or 1: the whole frame); buffers then hold local
                                      * tiles * 1024 entries in the compact tile-major layout */
} rtamd_multibounce_desc;

/* The device buffers of one sample in flight, n = rtamd_multibounce_entries(desc) entries each (floats per entry in
 * brackets); 16-byte aligned.  cam_org is filled once by rtamd_multibounce_prepare_lane_device; everything else is
 * written by every sample.  The two sets of dispatcher targets alternate by depth (set depth & 1), because a bounce's
 * continuation rays must survive the next bounce's dispatcher pass. */
typedef struct rtamd_multibounce_lane {
    float *cam_org;        /* [3] the eye, once per entry */
    float *dir;            /* [3] primaries */
    float *hits;           /* [4] primary records, then each bounce's continuation records */
    float *position;       /* [3] G-buffer of the current bounce */
    float *normal_mat;     /* [4] */
    float *org[2], *shadow_dir[2], *reflect_dir[2], *ambient_dir[2];   /* [3] dispatcher targets */
    float *shadow_hits;    /* [4] */
} rtamd_multibounce_lane;

/* Entries per lane buffer; primaries per sample (the shard's pixels inside the frame). */
uint64_t rtamd_multibounce_entries(const rtamd_multibounce_desc *desc);
uint64_t rtamd_multibounce_primary_rays(const rtamd_multibounce_desc *desc);
int rtamd_multibounce_prepare_lane_device(rtamd_context *ctx, const rtamd_multibounce_desc *desc, const rtamd_multibounce_lane *lane);
/* samples: n_samples (1..8) sample indices, sample k in lanes[k] (distinct lanes).  rays_traced: optional device counter
 * that receives the number of traversals the call starts (NO_RAY entries excluded): primaries + per bounce one shadow ray
 * and, up to the last depth, one continuation ray per pixel with a surface. */
int rtamd_multibounce_sample_batch_device(rtamd_context *ctx, const rtamd_multibounce_desc *desc, const uint32_t *samples, uint32_t n_samples,
                                          const rtamd_multibounce_lane *lanes, uint64_t *rays_traced);

/* ---- multi-GPU frame assembly helpers -------------------------------------- */

/* Number of 32x32 tiles a rank owns, maximised over ranks (the per-rank block
 * size, in tiles, of the gather buffer). */
uint32_t rtamd_tiles_per_rank(uint32_t width, uint32_t height, uint32_t tile_world);

/* Root side of the RCCL gather: `gathered` holds tile_world rank-major blocks of
 * rtamd_tiles_per_rank()*1024 pixels, `elems_per_pixel` 4-byte words each (4 for
 * rgba32f, 1 for hit_id / hit_t); writes the scan-line frame.  Device pointers. */
int rtamd_assemble_tiles_device(rtamd_context *ctx, const void *gathered, uint32_t width, uint32_t height,
                                uint32_t tile_world, uint32_t elems_per_pixel, void *frame_out);

/* The same for a batch of n_frames frames moved by ONE gather (a rank sends its shares of n_frames
 * consecutive frames as one block, which divides the per-frame cost of the collective by n_frames):
 * `gathered` is [rank][frame][rtamd_tiles_per_rank()*1024 pixels], `frames_out` [frame][height*width]. */
int rtamd_assemble_tile_batch_device(rtamd_context *ctx, const void *gathered, uint32_t width, uint32_t height,
                                     uint32_t tile_world, uint32_t elems_per_pixel, uint32_t n_frames, void *frames_out);

/* ---- one process, several GPUs: tile split + RCCL gather inside the library ------------------------ */

/* The reference's caller is ONE single-threaded process (src/main.rs:105-243) that owns one GL context.  This is
 * its multi-GPU counterpart behind the same kind of handle: one rtamd_context per listed GPU (the "device list" of the
 * context), the scene replicated to each, every frame cut into interleaved 32x32 tiles (tile k -> GPU k % n, as
 * rtamd_frame_desc.tile_rank / tile_world), each GPU rendering its tiles of a batch of frames with ONE launch, and ONE
 * collective per batch -- ncclGather over xGMI (/opt/rocm/include/rccl/rccl.h:745; communicators from ncclCommInitAll,
 * :236, single process) -- bringing the shares to the first listed GPU, which assembles the scan-line frames
 * (rtamd_assemble_tile_batch_device).  No other data crosses between GPUs.  The host side is the calling thread alone.
 * No counterpart in the reference (one GPU); the per-rank form of the same path, one process per GPU with the collective
 * issued by the caller, is what rtamd_frame_desc.tile_* + rtamd_assemble_tile_batch_device serve. */
typedef struct rtamd_multi rtamd_multi;

/* rtamd_multi_create flags */
#define RTAMD_MULTI_TRANSPORT_RCCL 0u   /* default: ncclGather (librccl.so.1 is loaded when the handle is created) */
#define RTAMD_MULTI_TRANSPORT_COPY 1u   /* the root pulls the shares with hipMemcpyAsync (peer access over xGMI) instead of the
                                         * collective.  RCCL refuses two ranks on one GPU, so this is also how the N > 1 logic
                                         * (tile split, batch buffers, assembly) is exercised on a one-GPU box: with this flag
                                         * the device list may name a GPU several times. */

/* devices: n_devices HIP ordinals, devices[0] is the root (the assembled frames land there).  1 <= n_devices <= 64. */
int rtamd_multi_create(const int *devices, uint32_t n_devices, uint32_t flags, rtamd_multi **out);
void rtamd_multi_destroy(rtamd_multi *m);
/* m == NULL: the last error of a failed rtamd_multi_create on this thread.  Never NULL. */
const char *rtamd_multi_last_error(const rtamd_multi *m);
uint32_t rtamd_multi_device_count(const rtamd_multi *m);
/* The i-th GPU's context, borrowed (rtamd_get_scene_info, rtamd_last_launch_coherence ...).  Do not destroy it and do not
 * change its stream. */
rtamd_context *rtamd_multi_context(rtamd_multi *m, uint32_t i);

/* As the single-GPU calls, applied to every GPU of the list (the scene and textures are replicated). */
int rtamd_multi_upload_scene(rtamd_multi *m, const rtamd_scene_desc *scene);
int rtamd_multi_upload_texture(rtamd_multi *m, uint32_t slot, const uint8_t *rgb8, uint32_t width, uint32_t height);
int rtamd_multi_upload_materials(rtamd_multi *m, const rtamd_shading_material *materials, uint32_t count);
int rtamd_multi_set_option(rtamd_multi *m, const char *name, int64_t value);

/* What a frame is gathered as (= what crosses xGMI), and the element type of frames_out. */
#define RTAMD_GATHER_RGBA8 0     /* the display pass's 8-bit back buffer (util/display.frag:9): one uint32_t per pixel, R in the low byte */
#define RTAMD_GATHER_RGBA32F 1   /* fragCol itself: four floats per pixel (4x the bytes on the links) */

/* n_frames (1..RTAMD_MAX_BATCH_FRAMES) frames of the fused shader, split over the GPUs of the list and assembled on the
 * root.  descs[f] as for rtamd_render_batch_device (same width / height / has_* for all frames), except: tile_rank /
 * tile_world are ignored (the library sets them), and dir / org, if given, are HOST arrays of width*height*3 floats that
 * the call copies to every GPU before it returns (with dir == NULL the rays are generated on each GPU from the camera
 * fields: nothing to copy, the usual multi-GPU form).  frames_out: n_frames * width*height pixels of the gather format,
 * scan-line order, frame after frame -- a DEVICE pointer on the root GPU for rtamd_multi_render_batch_device, which
 * enqueues and returns (consecutive calls alternate between two sets of share buffers, so that the gather and assembly of
 * one batch run beside the rendering of the next; rtamd_multi_synchronize, or work ordered behind rtamd_multi_root_stream,
 * sees the frames complete), a HOST pointer for rtamd_multi_render_batch, which waits and copies.  Every pixel equals,
 * bit for bit, what one GPU renders alone (and converts with rtamd_display_rgba8_device for RGBA8). */
int rtamd_multi_render_batch_device(rtamd_multi *m, uint32_t n_frames, const rtamd_frame_desc *descs, int gather_format, void *frames_out);
int rtamd_multi_render_batch(rtamd_multi *m, uint32_t n_frames, const rtamd_frame_desc *descs, int gather_format, void *frames_out);
/* Wait for everything the handle has enqueued on every GPU. */
int rtamd_multi_synchronize(rtamd_multi *m);
/* The root GPU's hipStream_t (as void*) on which gathers and assemblies run: order consumers of frames_out behind it. */
void *rtamd_multi_root_stream(rtamd_multi *m);
/* Milliseconds the root spent in the last batch's gather (own share ready -> all shares landed) and assembly; blocks until
 * that batch is done.  Either may be NULL. */
int rtamd_multi_last_gather_ms(rtamd_multi *m, float *gather_ms, float *assemble_ms);

/* ---- display pass: util/display.frag:9 ------------------------------------------ */

/* The reference's last pass copies the float colour target to the window's 8-bit back buffer
 * (src/main.rs:232-238, util/display.frag:9); the conversion is OpenGL's float -> unorm8 (clamp to
 * [0,1], x255, round to nearest).  This is that conversion for n_pixels RGBA32F pixels (any layout:
 * scan-line frame or compact tiles) into packed RGBA8 (R in the low byte): a frame then leaves the
 * GPU, or crosses xGMI, at a quarter of the bytes.  NaN converts to 0.  Device pointers. */
int rtamd_display_rgba8_device(rtamd_context *ctx, const float *rgba32f, uint64_t n_pixels, uint32_t *rgba8_out);

/* ---- introspection ---------------------------------------------------------- */

typedef struct rtamd_scene_info {
    uint32_t node_count, triangle_count, vertex_count;
    uint32_t inner_node_count;   /* 64-byte device records */
    uint32_t big_leaf_count;     /* leaves that need the overflow table */
    uint32_t max_depth;          /* deepest root-to-leaf path */
    uint64_t device_bytes;       /* HBM held by the scene */
    uint32_t odd_box_count;      /* reachable nodes whose box has a non-finite bound or min > max on some axis.  The reference's
                                  * builder never makes one (types.rs:59-99), but intersectAABB (ray_trace_and_shade.frag:53-61) is
                                  * defined for any six floats and the C-ABI takes any caller's nodes: with one or more, every visit
                                  * of the scene uses the general slab test instead of the direction-octant form, which equals it
                                  * only for finite boxes with min <= max.  Results are the shader's either way. */
    uint32_t reserved;
} rtamd_scene_info;
int rtamd_get_scene_info(const rtamd_context *ctx, rtamd_scene_info *info);

/* Milliseconds the main kernel of the last *_device / host call took, measured
 * with HIP events on the stream it ran on (blocks until that kernel is done). */
int rtamd_last_kernel_ms(rtamd_context *ctx, float *ms);

/* What the last render / trace launch on the current stream measured about its rays (sampled over 1/8 of the
 * waves; blocks until that launch is done; any output may be NULL): the share of node visits in which every
 * active lane of a wave visited the same node, the share of lanes active per visit, and the lane re-fill
 * threshold the launch ran with.  These are the signals "dyn_thresh" = auto steers by from one launch to the next (fused
 * shader, plain ray lists).  Launches of image lists and of several lists report 1: there every wave chooses for itself, list by
 * list, from its own first visits in the list. */
int rtamd_last_launch_coherence(rtamd_context *ctx, float *uniform_fraction, float *lane_utilisation, uint32_t *dyn_thresh_used);

/* Kernel variant used by render/trace: 0 = default (persistent wave64 queue),
 * 1 = one lane per pixel, static 8x8 blocks (A/B baseline). */
int rtamd_set_kernel_variant(rtamd_context *ctx, int variant);

/* Tuning knobs of the persistent kernel (benchmarks / profiling; results never
 * change):
 *  "dyn_thresh" 0..64   re-fill idle lanes from the next work item when fewer than this many lanes of a
 *                       wave are still traversing (1 = only when the whole wave is done).  0 = auto (default):
 *                       each launch reads what the previous launch of the same kind on the same stream
 *                       measured and keeps 1 while at least 50 % of the lanes are active per node visit
 *                       (camera rays, sun shadow rays), switches to 32 below that (triangle soups, bounces)
 *                       for 64 launches, then probes one launch at 1 again; the first launch uses 1;
 *                       launches of image lists or of several ray lists (rtamd_trace_[image_]batch_device): every wave chooses
 *                       per list instead -- no re-fill until its own first 32 visits in a list had under half of the lanes active;
 *  "dyn_incoherent" 2..64  the threshold "dyn_thresh" = auto switches to for incoherent rays (default 32: measured flat over
 *                       24..32 on config 4 and on config 5's lists, 3-8 % slower at 16 and at 48, profiles/r03_ab_dyn_incoherent.txt);
 *  "leaf_thresh" 0..64  lanes waiting at a leaf are served once this many wait (64 = classic
 *                       while-while traversal, 1 = if-if), or once half of the wave's busy lanes wait, whichever is fewer (a wave
 *                       with few rays left does not make a lane at a leaf wait for a count the others can hardly reach).
 *                       Per-lane order is untouched, so no result bit depends on it.  0 (default) = by what the wave is doing:
 *                       20 while it runs without lane re-fill (coherent rays), 8 while a fused frame re-fills (soups), 12 while
 *                       a ray list re-fills (bounces); "leaf_thresh_coherent" / "leaf_thresh_refill" 1..64 set the two sides alone;
 *  "num_queues" 1..32   work-queue counters (8 = one per XCD);
 *  "queue_stride" 1..4096  u32 words between the counters (default 64 = 256 B, one memory
 *                       channel each);
 *  "max_blocks_per_cu" 0..64  cap on the persistent grid, in workgroups per CU.  0 (default) = automatic: all that
 *                       fit (8) for a launch on its own (7 for a fused launch of ONE frame: it is bounded by its slowest
 *                       pixels, and fewer waves per SIMD run each faster), 16 / n when n launches of this context are in
 *                       flight on different streams, so that the grids of overlapping frames are resident side by side;
 *  "queue_block" 1..65536  consecutive schedule positions (work items) a queue owns per turn, i.e. the size of
 *                       the screen region the waves of one XCD share (default 256 for frames, 1 for ray lists);
 *  "validate_buffers" 0/1  (default 1) every "_device" entry point checks, with hipMemGetAddressRange, that each buffer it is
 *                       given lies inside ONE device allocation for the size the call needs, and refuses the call with
 *                       RTAMD_ERR_INVALID_ARGUMENT otherwise: a short buffer would make the kernel read or write out of bounds,
 *                       and a GPU memory fault takes the process down.  (Pieces of a sub-allocator's larger allocation are
 *                       checked against that allocation's end.)  0 skips the look-ups (a few host microseconds per call);
 *  "launch_timing" 0/1  record a start event in front of every launch (default 1); 0 saves one
 *                        stream operation per launch, rtamd_last_kernel_ms then reports an error.
 *  "shadow_anyhit" 0/1  fused shader only (default 1): a shadow ray stops at the first leaf that
 *                       leaves its t below the 1000 limit.  The reference traverses on
 *                       (ray_trace_and_shade.frag:213-219) but consumes only `shadow_t < 1000`, which
 *                       cannot change once true, so every output is identical; only the optional
 *                       per-pixel test counters are smaller.  0 = traverse to the end (counter parity);
 *  "texture_filter_8bit" 0/1  (default 0) how texture() filters the RGBA8 albedo and noise images.  0: bilinear in float.
 *                       1: as the executor the parity tests are pinned to (Mesa llvmpipe) does it -- 8-bit weights, three rounded
 *                       8-bit lerps, an 8-bit result -- reproduced bit for bit on its probed samples; with it textured colours
 *                       are within 0.01/255 of that executor's instead of 1/255, and the dispatcher's ambient directions equal
 *                       its own between texels too.  GL permits both; float is the more exact filter;
 *  "kernel_variant"     as rtamd_set_kernel_variant. */
int rtamd_set_option(rtamd_context *ctx, const char *name, int64_t value);

const char *rtamd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_H */
