// C-ABI implementation (include/rtamd.h): context, scene upload + layout
// conversion, kernel launches.  gfx950 only; no CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../../include/rtamd.h"
#include "rt_kernels.hip"

using namespace rtamd;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

}  // namespace

struct rtamd_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // Per-stream launch scratch.  Launches on one stream are ordered by the stream; launches
    // on different streams may overlap on the GPU (frames in flight), so each stream the
    // context has been pointed at owns its queue counters, stack-spill area, schedule arrays
    // and timing events.  MAX_SLOTS streams are tracked; a further one takes over the least
    // recently used slot after that slot's last launch has finished.
    struct Slot {
        hipStream_t key = nullptr;
        bool bound = false;
        uint64_t last_use = 0;
        hipEvent_t ev_start = nullptr, ev_stop = nullptr;
        bool ev_valid = false;
        bool timed = false;   // ev_start was recorded for the last launch
        DevBuf queues, spill;
        uint64_t launches = 0;              // persistent-kernel launches on this stream since the statistics layout last changed
        int stats_mode = -1;                // mode of the launch that wrote the latest statistics
        uint32_t *stats_last = nullptr;     // where that launch put them
    };
    static constexpr int MAX_SLOTS = 8;
    Slot slots[MAX_SLOTS];
    uint64_t use_clock = 0;
    int last_slot = -1;
    size_t spill_bytes = 0;
    std::string err;
    int variant = 0;
    int num_cu = 0;
    uint32_t *static_scratch = nullptr;   // device address of the code object's static launch scratch on this GPU (rtamd_render_device_static)
    int persistent_blocks_per_cu[6] = {0, 0, 0, 0, 0, 0};  // [mode*2 + count]; [4 + count]: fused with a material table
    // scene
    bool has_scene = false;
    DevBuf nodes, tris, big, tri_idx, normals, texcoords;
    DevBuf tex[RTAMD_MAX_TEXTURE_SLOTS];
    DevBuf noise;
    int32_t noise_w = 0, noise_h = 0;
    int32_t tex_w[RTAMD_MAX_TEXTURE_SLOTS] = {}, tex_h[RTAMD_MAX_TEXTURE_SLOTS] = {};   // valid where tex[k].p is set
    DevBuf white;
    // optional per-material shading (rtamd_upload_materials)
    DevBuf materials, tex_slots;
    uint32_t n_materials = 0;
    uint32_t scene_max_mat = 0;       // largest mat_idx of the uploaded scene's triangles
#ifdef RTAMD_EXP_STAMPS
    DevBuf stamps;
#endif
    SceneDev sdev{};
    rtamd_scene_info info{};
    // measured defaults (tools/kbench.py): coherent camera/shadow rays run best without lane
    // re-fill (1), incoherent ray streams with re-fill below 32 busy lanes; leaves are
    // released when 8 lanes wait
    uint32_t dyn_thresh_fused = 1, dyn_thresh_trace = 1;   // starting values; with dyn_auto the kernel adapts them per launch
    bool dyn_auto = true;
    uint32_t dyn_incoherent = 32;
    // lanes waiting at a leaf that release the leaf phase (capped at half of the busy lanes): while a wave runs without lane re-fill, and
    // while it re-fills -- fused shader / ray lists.  Round-3 sweep with the half-of-busy rule in place (profiles/r03_ab_leaf_thresh_by_mode.txt):
    // coherent 8 -> 20: atrium batches +1.5 %, bunny +0.5 %; re-filling ray lists 8 -> 12: config 5 +1.6 %; re-filling soup frames want 8.
    uint32_t leaf_thresh = 20, leaf_thresh_refill_fused = 8, leaf_thresh_refill_trace = 12;
    bool shadow_anyhit = true;
    bool tex_filter8 = false;   // option "texture_filter_8bit"
    bool validate_buffers = true;     // *_device entry points check that every buffer lies inside one device allocation (check_range)
    bool launch_timing = true;        // record a start event per launch (rtamd_last_kernel_ms); the stop event is always recorded
    uint32_t queue_stride = 64;
    uint32_t max_blocks_per_cu = 0;   // 0 = as many as fit (8 workgroups = 8 waves per SIMD)
    uint32_t queue_block_fused = 256, queue_block_trace = 1;   // measured: 256-item regions per XCD turn help camera frames 1.5 %, not ray lists
    uint32_t n_queues = 8;
};

namespace {

int fail(rtamd_context *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ctx, e_ == hipErrorOutOfMemory ? RTAMD_ERR_OUT_OF_MEMORY : RTAMD_ERR_DEVICE, \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                            \
    } while (0)

int dev_alloc(rtamd_context *ctx, DevBuf &b, size_t bytes) {
    b.release();
    if (bytes == 0) bytes = 16;
    HIP_TRY(ctx, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return RTAMD_OK;
}

int dev_upload(rtamd_context *ctx, DevBuf &b, const void *src, size_t bytes) {
    int rc = dev_alloc(ctx, b, bytes);
    if (rc) return rc;
    if (bytes) HIP_TRY(ctx, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return RTAMD_OK;
}

// The pinned reference (llvmpipe) runs with denormals flushed (inputs and
// results); the device kernels are compiled the same way.  Precomputed edges are
// host arithmetic, so flush here explicitly.
inline float ftz(float x) { return std::fabs(x) < 1.17549435e-38f ? std::copysign(0.0f, x) : x; }

uint32_t n_tiles_of(uint32_t w, uint32_t h, uint32_t *tiles_x) {
    uint32_t tx = (w + RTAMD_TILE_DIM - 1) / RTAMD_TILE_DIM, ty = (h + RTAMD_TILE_DIM - 1) / RTAMD_TILE_DIM;
    if (tiles_x) *tiles_x = tx;
    return tx * ty;
}

uint32_t local_tiles(uint32_t n_tiles, uint32_t rank, uint32_t world) {
    return rank < n_tiles ? (n_tiles - rank + world - 1) / world : 0;
}

template <typename K>
int query_blocks(rtamd_context *ctx, K kernel, int *out) {
    int nb = 0;
    HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, BLOCK, 0));
    if (nb < 1) nb = 1;
    *out = nb;
    return RTAMD_OK;
}

// Every entry point that touches the GPU makes the context's device current for the duration of the
// call and puts the caller's device back on return: a process that drives several GPUs (one context each)
// keeps its own current device across calls into the library.
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    int enter(rtamd_context *ctx) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != ctx->device) {
            HIP_TRY(ctx, hipSetDevice(ctx->device));
            switched = true;
        }
        return RTAMD_OK;
    }
    ~DeviceScope() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
};

constexpr size_t QUEUE_HEADROOM_WORDS = 64;   // room for the odd launches' statistics pair in front of the counters
constexpr size_t QUEUE_BYTES = 256 + 32 * 16384 + 256;   // headroom + (num_queues <= 32) * (queue_stride <= 4096 words) + the even pair

// The scratch slot of the context's current stream (created on first use).
int acquire_slot(rtamd_context *ctx, rtamd_context::Slot **out) {
    rtamd_context::Slot *free_slot = nullptr, *lru = nullptr;
    for (auto &sl : ctx->slots) {
        if (sl.bound && sl.key == ctx->stream) { sl.last_use = ++ctx->use_clock; *out = &sl; return RTAMD_OK; }
        if (!sl.bound) { if (!free_slot) free_slot = &sl; }
        else if (!lru || sl.last_use < lru->last_use) lru = &sl;
    }
    rtamd_context::Slot *sl = free_slot ? free_slot : lru;
    if (sl->bound && sl->ev_valid) HIP_TRY(ctx, hipEventSynchronize(sl->ev_stop));   // take-over: its last launch must be done
    int rc;
    if (!sl->ev_start) HIP_TRY(ctx, hipEventCreate(&sl->ev_start));
    if (!sl->ev_stop) HIP_TRY(ctx, hipEventCreate(&sl->ev_stop));
    if (!sl->queues.p) {
        if ((rc = dev_alloc(ctx, sl->queues, QUEUE_BYTES))) return rc;
        HIP_TRY(ctx, hipMemset(sl->queues.p, 0, QUEUE_BYTES));
    }
    if (!sl->spill.p && (rc = dev_alloc(ctx, sl->spill, ctx->spill_bytes))) return rc;
    sl->launches = 0; sl->stats_last = nullptr;
    sl->key = ctx->stream; sl->bound = true; sl->ev_valid = false;
    sl->last_use = ++ctx->use_clock;
    *out = sl;
    return RTAMD_OK;
}

// The fused persistent kernel's instantiation for (per-pixel counters, material table, 8-bit texture filter).
void launch_persistent_fused(bool count, bool mat, bool f8, uint32_t grid, hipStream_t st, const SceneDev &s, const JobParams &p) {
// (one frame: the instantiation without the batch's frame look-ups; several: the one that serves both, see k_persistent)
#define RTAMD_FUSED(C, M, F) do { if (p.n_frames > 1u) hipLaunchKernelGGL((k_persistent<MODE_FUSED, C, M, F, false>), dim3(grid), dim3(BLOCK), 0, st, s, p); \
                                  else hipLaunchKernelGGL((k_persistent<MODE_FUSED, C, M, F, true>), dim3(grid), dim3(BLOCK), 0, st, s, p); } while (0)
    switch ((count ? 4 : 0) | (mat ? 2 : 0) | (f8 ? 1 : 0)) {
        case 0: RTAMD_FUSED(false, false, false); break;
        case 1: RTAMD_FUSED(false, false, true); break;
        case 2: RTAMD_FUSED(false, true, false); break;
        case 3: RTAMD_FUSED(false, true, true); break;
        case 4: RTAMD_FUSED(true, false, false); break;
        case 5: RTAMD_FUSED(true, false, true); break;
        case 6: RTAMD_FUSED(true, true, false); break;
        default: RTAMD_FUSED(true, true, true); break;
    }
#undef RTAMD_FUSED
}

// Launch the traversal kernel selected by (mode, count, variant) for `p`.
int launch_main(rtamd_context *ctx, int mode, bool count, JobParams &p) {
    if (p.n_items == 0) return RTAMD_OK;
    rtamd_context::Slot *sl = nullptr;
    int rc_slot = acquire_slot(ctx, &sl);
    if (rc_slot) return rc_slot;
    ctx->last_slot = (int)(sl - ctx->slots);
    p.queues = (uint32_t *)sl->queues.p + QUEUE_HEADROOM_WORDS;
    p.n_queues = ctx->n_queues;
    p.dyn_thresh = mode == MODE_FUSED ? ctx->dyn_thresh_fused : ctx->dyn_thresh_trace;
    p.dyn_auto = ctx->dyn_auto ? 1u : 0u;
    p.dyn_incoherent = ctx->dyn_incoherent;
    p.stats_prev = nullptr; p.stats_cur = nullptr;
    p.leaf_thresh = ctx->leaf_thresh; p.leaf_thresh_refill = mode == MODE_FUSED ? ctx->leaf_thresh_refill_fused : ctx->leaf_thresh_refill_trace;
    p.shadow_anyhit = ctx->shadow_anyhit ? 1u : 0u;
    p.tex_filter8 = ctx->tex_filter8 ? 1u : 0u;
    p.queue_stride = ctx->queue_stride; p.queue_block = mode == MODE_FUSED ? ctx->queue_block_fused : ctx->queue_block_trace;
    p.spill = (uint32_t *)sl->spill.p;
    p.spill_wave_words = (uint32_t)SPILL_STACK_ENTRIES * 64u;
#ifdef RTAMD_EXP_STAMPS
    p.stamps = (unsigned long long *)ctx->stamps.p;
#endif
    hipStream_t st = ctx->stream;
    const SceneDev &s = ctx->sdev;
    const bool mat = mode == MODE_FUSED && s.materials != nullptr;   // per-material shading: its own instantiation
    // (Letting the last wave of a launch zero the counters instead — one exit atomic per wave — was measured
    // equal for the fused shader and 5 % slower for ray_trace launches, whose waves all leave at once.)
    if (ctx->variant == 0) {
        // One memset per launch zeroes the queue counters and this launch's four statistics words; the four
        // the previous launch wrote sits on the other side of the counters and survives:
        //   [ stats of odd launches | queue counters (n_queues * stride words) | stats of even launches ]
        uint32_t *q = (uint32_t *)sl->queues.p + QUEUE_HEADROOM_WORDS;
        const size_t q_words = (size_t)ctx->n_queues * ctx->queue_stride;
        uint32_t *stats_odd = q - 4, *stats_even = q + q_words;
        const bool odd = (sl->launches & 1u) != 0;
        p.queues = q;
        p.stats_cur = odd ? stats_odd : stats_even;
        if (sl->launches > 0 && sl->stats_mode == mode) p.stats_prev = odd ? stats_even : stats_odd;   // the previous launch here was of this kind
        sl->launches++;
        sl->stats_mode = mode;
        sl->stats_last = p.stats_cur;
        HIP_TRY(ctx, hipMemsetAsync(odd ? (void *)stats_odd : (void *)q, 0, (q_words + 4) * sizeof(uint32_t), st));
    }
    sl->timed = ctx->launch_timing;
    if (sl->timed) HIP_TRY(ctx, hipEventRecord(sl->ev_start, st));
    if (ctx->variant == 0) {
        int per_cu = ctx->persistent_blocks_per_cu[(mat ? 4 : mode * 2) + (count ? 1 : 0)];
        if (ctx->max_blocks_per_cu) {
            if ((int)ctx->max_blocks_per_cu < per_cu) per_cu = (int)ctx->max_blocks_per_cu;
        } else {
            // Frames in flight: a persistent grid that fills the chip lets the next stream's launch in only as
            // its own workgroups retire.  With n launches in flight (this one + unfinished ones on the context's
            // other streams) each gets 16 / n workgroups per CU, so two grids are resident side by side:
            // measured on the 1080p atrium 0.251 -> 0.238 ms with 4 streams, a 1/4 tile share 91 -> 62 us with 8.
            int inflight = 1;
            for (const auto &o : ctx->slots)
                if (&o != sl && o.bound && o.ev_valid && hipEventQuery(o.ev_stop) == hipErrorNotReady) inflight++;
            (void)hipGetLastError();   // hipErrorNotReady is an answer, not a failure
            const int share = (16 * 256 / BLOCK) / inflight;   // 16 workgroups of 256 threads per CU, in units of this build's workgroup
            if (share < per_cu) per_cu = share < 1 ? 1 : share;
            // One frame in one launch with nothing beside it is bounded by its slowest items, not by throughput: 7 waves per SIMD instead of
            // 8 run each of them faster (a frame alone: bunny -5.5 %, 1 M soup -5.6 %, atrium +0.5 %, f16 equal; profiles/r03_ab_blocks_per_cu.txt).
            // Batches and frames pipelined over streams keep all that fit.
            const int lone = 7 * 256 / BLOCK;
            if (mode == MODE_FUSED && p.n_frames <= 1u && inflight == 1 && per_cu > lone) per_cu = lone;
        }
        uint32_t grid = (uint32_t)(ctx->num_cu * per_cu);
        uint32_t need = (p.n_items + (BLOCK / 64) - 1) / (BLOCK / 64);
        if (grid > need) grid = need;
        if (grid < 1) grid = 1;
        if (mode == MODE_FUSED) {
            launch_persistent_fused(count, mat, ctx->tex_filter8, grid, st, s, p);
        } else {
            if (count) hipLaunchKernelGGL((k_persistent<MODE_TRACE, true>), dim3(grid), dim3(BLOCK), 0, st, s, p);
            else hipLaunchKernelGGL((k_persistent<MODE_TRACE, false>), dim3(grid), dim3(BLOCK), 0, st, s, p);
        }
    } else {
        uint32_t grid = (p.n_items + (BLOCK / 64) - 1) / (BLOCK / 64);
        if (mode == MODE_FUSED && mat) {
            if (count) hipLaunchKernelGGL((k_static<MODE_FUSED, true, true>), dim3(grid), dim3(BLOCK), 0, st, s, p);
            else hipLaunchKernelGGL((k_static<MODE_FUSED, false, true>), dim3(grid), dim3(BLOCK), 0, st, s, p);
        } else if (mode == MODE_FUSED) {
            if (count) hipLaunchKernelGGL((k_static<MODE_FUSED, true>), dim3(grid), dim3(BLOCK), 0, st, s, p);
            else hipLaunchKernelGGL((k_static<MODE_FUSED, false>), dim3(grid), dim3(BLOCK), 0, st, s, p);
        } else {
            if (count) hipLaunchKernelGGL((k_static<MODE_TRACE, true>), dim3(grid), dim3(BLOCK), 0, st, s, p);
            else hipLaunchKernelGGL((k_static<MODE_TRACE, false>), dim3(grid), dim3(BLOCK), 0, st, s, p);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(sl->ev_stop, st));
    sl->ev_valid = true;
    return RTAMD_OK;
}

// ---------------------------------------------------------------------------
// Launch scratch of rtamd_render_device_static, the entry point that allocates nothing (include/rtamd.h).
// The persistent kernel needs two pieces of device memory beyond inputs and outputs: the work-queue
// counters and the per-wave HBM columns that take the bottom of deep traversal stacks.  Every other entry
// point gets them from hipMalloc (one set per stream, acquire_slot); this one keeps them in ONE array that
// is part of the code object, so every GPU the library runs on holds its own copy until the process ends:
//   [ 8 queue counters, 256 B apart (2 KB) | stack columns [8192 waves][32 entries][64 lanes] u32 (64 MiB) | guard ]
// 8192 waves = the full persistent grid (256 CUs x 8 workgroups x 4 waves); 32 entries beyond the 16 of the
// LDS ring = trees up to depth 48 (deeper scenes are refused by the entry point, and the kernel bounds every
// column access by the per-launch entry count it is given, spill_wave_words / 64, so a wave cannot reach its
// neighbour's column whatever the tree).
// All static calls on a GPU share the array, across streams and contexts: the entry point orders them with one
// event per device (g_static_done: each call's launches wait for the previous call's kernel), so overlapping
// callers are serialised on the GPU instead of popping each other's stack entries.
// `scratch` is a host-side stand-in here, not a device address: host code uses static_scratch_ptr().
constexpr uint32_t STATIC_WAVES = 8192;
constexpr uint32_t STATIC_SPILL_ENTRIES = 32;
constexpr uint32_t STATIC_MAX_DEPTH = (uint32_t)LDS_STACK_ENTRIES + STATIC_SPILL_ENTRIES;
constexpr uint32_t STATIC_QUEUES = 8, STATIC_QUEUE_STRIDE = 64;
constexpr size_t STATIC_QUEUE_WORDS = (size_t)STATIC_QUEUES * STATIC_QUEUE_STRIDE;
constexpr size_t STATIC_SPILL_WORDS = (size_t)STATIC_WAVES * STATIC_SPILL_ENTRIES * 64;
static __device__ __align__(256) uint32_t scratch[STATIC_QUEUE_WORDS + STATIC_SPILL_WORDS];

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
uint32_t *g_scratch_ptr[MAX_DEVICES] = {};
hipEvent_t g_static_done[MAX_DEVICES] = {};    // recorded behind the kernel of the last static call on the device

// The device address of `scratch` on the context's GPU (which must be current): looked up once per device.
int static_scratch_ptr(rtamd_context *ctx, uint32_t **out) {
    if (ctx->device < 0 || ctx->device >= MAX_DEVICES) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "device ordinal %d beyond %d", ctx->device, MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[ctx->device]) {
        uint32_t *ptr = nullptr;
        HIP_TRY(ctx, hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(ctx, RTAMD_ERR_DEVICE, "hipGetSymbolAddress returned NULL for the static launch scratch");
        hipEvent_t ev = nullptr;
        HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventRecord(ev, nullptr));      // a recorded event from the start: the first call's wait is a no-op
        g_static_done[ctx->device] = ev;
        g_scratch_ptr[ctx->device] = ptr;
    }
    *out = g_scratch_ptr[ctx->device];
    return RTAMD_OK;
}

// A *_device entry point gets raw device addresses; a buffer that is shorter than the call needs would be read or
// written out of bounds by the kernel, and on this hardware a GPU memory fault takes the whole process — possibly the
// machine's other users — down.  What the host can check it checks: that [ptr, ptr + bytes) lies inside ONE device
// allocation known to the HIP runtime.  (A sub-allocator such as torch's hands out pieces of larger allocations, so
// this catches runs past the end of the underlying allocation, the case that faults, not every short buffer.)
// Addresses the runtime does not know (hipErrorNotFound: e.g. memory mapped by another API) are let through.
int check_range(rtamd_context *ctx, const void *ptr, uint64_t bytes, const char *what) {
    if (!ptr || bytes == 0 || !ctx->validate_buffers) return RTAMD_OK;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr) != hipSuccess) { (void)hipGetLastError(); return RTAMD_OK; }
    const uintptr_t lo = (uintptr_t)base, hi = lo + size, a = (uintptr_t)ptr;
    if (a < lo || a > hi || bytes > hi - a)
        return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "%s: %llu bytes at %p run past the end of the device allocation they start in (%zu bytes at %p)",
                    what, (unsigned long long)bytes, ptr, size, base);
    return RTAMD_OK;
}

int check_frame(rtamd_context *ctx, const rtamd_frame_desc *d, const rtamd_frame_out *o) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    if (!d || !o) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "frame descriptor / output is NULL");
    if (!ctx->has_scene) return fail(ctx, RTAMD_ERR_NO_SCENE, "no scene uploaded");
    if (d->width == 0 || d->height == 0) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "empty frame %ux%u", d->width, d->height);
    if ((uint64_t)d->width * d->height > 0x7FFFFFFFull) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "frame too large");
    if (d->tile_world == 0 || d->tile_rank >= d->tile_world)
        return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "tile_rank %u / tile_world %u invalid", d->tile_rank, d->tile_world);
    if ((d->dir == nullptr) != (d->org == nullptr)) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "dir and org must both be given or both be NULL");
    if (d->has_tex_coords && !ctx->texcoords.p) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "has_tex_coords set but the scene has no tex_coords");
    if (d->has_normals && !ctx->normals.p) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "has_normals set but the scene has no normals");
    return RTAMD_OK;
}

void fill_job(const rtamd_context *ctx, const rtamd_frame_desc *d, const rtamd_frame_out *o, JobParams &p) {
    std::memset(&p, 0, sizeof p);
    p.width = d->width; p.height = d->height;
    p.dir = d->dir; p.org = d->org;
    std::memcpy(p.ipv, d->inv_proj_view, sizeof p.ipv);
    p.near_ = d->near_plane; p.far_ = d->far_plane;
    std::memcpy(p.cam, d->cam_pos, sizeof p.cam);
    p.has_tc = d->has_tex_coords ? 1u : 0u;
    p.has_n = d->has_normals ? 1u : 0u;
    p.rank = d->tile_rank; p.world = d->tile_world;
    uint32_t tiles_x = 0;
    uint32_t n_tiles = n_tiles_of(d->width, d->height, &tiles_x);
    p.tiles_x = tiles_x;
    p.n_items = local_tiles(n_tiles, d->tile_rank, d->tile_world) * 16u;
    p.rgba = o->rgba32f; p.hit_id = o->hit_id; p.hit_t = o->hit_t; p.counts = o->counts;
    p.n_frames = 1; p.items_per_frame = p.n_items;
    (void)ctx;
}

uint64_t frame_out_pixels(const rtamd_frame_desc *d) {
    const uint32_t n_tiles = n_tiles_of(d->width, d->height, nullptr);
    return d->tile_world == 1 ? (uint64_t)d->width * d->height : (uint64_t)local_tiles(n_tiles, d->tile_rank, d->tile_world) * RTAMD_TILE_PIXELS;
}

// Device buffers of one frame: rays for every pixel of the frame, outputs for the pixels this shard owns.
int check_frame_buffers(rtamd_context *ctx, const rtamd_frame_desc *d, const rtamd_frame_out *o) {
    const uint64_t n_px = (uint64_t)d->width * d->height, n_out = frame_out_pixels(d);
    int rc;
    if ((rc = check_range(ctx, d->dir, n_px * 12, "dir")) || (rc = check_range(ctx, d->org, n_px * 12, "org")) ||
        (rc = check_range(ctx, o->rgba32f, n_out * 16, "rgba32f")) || (rc = check_range(ctx, o->hit_id, n_out * 4, "hit_id")) ||
        (rc = check_range(ctx, o->hit_t, n_out * 4, "hit_t")) || (rc = check_range(ctx, o->counts, n_out * 8, "counts")) ||
        (rc = check_range(ctx, o->rgba8, n_out * 4, "rgba8")))
        return rc;
    return RTAMD_OK;
}

}  // namespace

extern "C" {

const char *rtamd_version(void) { return "rtamd 0.1 (gfx950)"; }

int rtamd_context_create(int device_ordinal, rtamd_context **out_ctx) {
    if (!out_ctx) return fail(nullptr, RTAMD_ERR_INVALID_ARGUMENT, "out_ctx is NULL");
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(nullptr, RTAMD_ERR_DEVICE, "no HIP device available (%s)", hipGetErrorString(e));
    if (device_ordinal < 0 || device_ordinal >= n) return fail(nullptr, RTAMD_ERR_INVALID_ARGUMENT, "device %d out of range (0..%d)", device_ordinal, n - 1);
    rtamd_context *ctx = new (std::nothrow) rtamd_context();
    if (!ctx) return fail(nullptr, RTAMD_ERR_OUT_OF_MEMORY, "out of host memory");
    ctx->device = device_ordinal;
    auto bail = [&](int code) { g_create_error = ctx->err; rtamd_context_destroy(ctx); return code; };
    int rc;
    DeviceScope dev_scope;
    if ((rc = dev_scope.enter(ctx))) return bail(rc);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) { ctx->err = "hipGetDeviceProperties failed"; return bail(RTAMD_ERR_DEVICE); }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        ctx->err = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return bail(RTAMD_ERR_DEVICE);
    }
    ctx->num_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { ctx->err = "hipStreamCreate failed"; return bail(RTAMD_ERR_DEVICE); }
    ctx->stream = ctx->own_stream;
    if ((rc = query_blocks(ctx, k_persistent<MODE_FUSED, false>, &ctx->persistent_blocks_per_cu[0]))) return bail(rc);
    if ((rc = query_blocks(ctx, k_persistent<MODE_FUSED, true>, &ctx->persistent_blocks_per_cu[1]))) return bail(rc);
    if ((rc = query_blocks(ctx, k_persistent<MODE_TRACE, false>, &ctx->persistent_blocks_per_cu[2]))) return bail(rc);
    if ((rc = query_blocks(ctx, k_persistent<MODE_TRACE, true>, &ctx->persistent_blocks_per_cu[3]))) return bail(rc);
    if ((rc = query_blocks(ctx, k_persistent<MODE_FUSED, false, true>, &ctx->persistent_blocks_per_cu[4]))) return bail(rc);
    if ((rc = query_blocks(ctx, k_persistent<MODE_FUSED, true, true>, &ctx->persistent_blocks_per_cu[5]))) return bail(rc);
    {
        int max_per_cu = 1;
        for (int k = 0; k < 6; k++) if (ctx->persistent_blocks_per_cu[k] > max_per_cu) max_per_cu = ctx->persistent_blocks_per_cu[k];
        size_t waves = (size_t)ctx->num_cu * max_per_cu * (BLOCK / 64);
        ctx->spill_bytes = waves * SPILL_STACK_ENTRIES * 64 * sizeof(uint32_t);
    }
#ifdef RTAMD_EXP_STAMPS
    if ((rc = dev_alloc(ctx, ctx->stamps, 65536 * 128 + 65536 * 32))) return bail(rc);
    (void)hipMemset(ctx->stamps.p, 0, ctx->stamps.bytes);
#endif
    if ((rc = static_scratch_ptr(ctx, &ctx->static_scratch))) return bail(rc);
    const uint32_t white = 0xFFFFFFFFu;
    if ((rc = dev_upload(ctx, ctx->white, &white, 4))) return bail(rc);
    {   // the slot of the context's own stream exists from the start: allocation failures surface here
        rtamd_context::Slot *sl = nullptr;
        if ((rc = acquire_slot(ctx, &sl))) return bail(rc);
    }
    *out_ctx = ctx;
    return RTAMD_OK;
}

void rtamd_context_destroy(rtamd_context *ctx) {
    if (!ctx) return;
    DeviceScope dev_scope;
    (void)dev_scope.enter(ctx);
    (void)hipDeviceSynchronize();   // launches may be in flight on several (caller-owned) streams
    ctx->nodes.release(); ctx->tris.release(); ctx->big.release(); ctx->tri_idx.release();
    ctx->normals.release(); ctx->texcoords.release(); for (auto &t : ctx->tex) t.release(); ctx->materials.release(); ctx->tex_slots.release(); ctx->noise.release();
    ctx->white.release();
    for (auto &sl : ctx->slots) {
        sl.queues.release(); sl.spill.release();
        if (sl.ev_start) (void)hipEventDestroy(sl.ev_start);
        if (sl.ev_stop) (void)hipEventDestroy(sl.ev_stop);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char *rtamd_last_error(const rtamd_context *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rtamd_context_set_stream(rtamd_context *ctx, void *hip_stream) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return RTAMD_OK;
}

int rtamd_synchronize(rtamd_context *ctx) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    DeviceScope dev_scope;
    int rc = dev_scope.enter(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RTAMD_OK;
}

int rtamd_set_kernel_variant(rtamd_context *ctx, int variant) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    if (variant != 0 && variant != 1) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "unknown kernel variant %d", variant);
    ctx->variant = variant;
    return RTAMD_OK;
}

int rtamd_set_option(rtamd_context *ctx, const char *name, int64_t value) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    if (!name) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "option name is NULL");
    if (!std::strcmp(name, "dyn_thresh")) {
        if (value < 0 || value > 64) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "dyn_thresh must be 0 (auto) or 1..64");
        if (value == 0) { ctx->dyn_auto = true; ctx->dyn_thresh_fused = 1; ctx->dyn_thresh_trace = 1; }
        else { ctx->dyn_auto = false; ctx->dyn_thresh_fused = ctx->dyn_thresh_trace = (uint32_t)value; }
        for (auto &sl : ctx->slots) sl.launches = 0;
    } else if (!std::strcmp(name, "dyn_incoherent")) {
        if (value < 2 || value > 64) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "dyn_incoherent must be 2..64");
        ctx->dyn_incoherent = (uint32_t)value;
    } else if (!std::strcmp(name, "leaf_thresh")) {
        if (value < 0 || value > 64) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "leaf_thresh must be 0..64");
        if (value == 0) { ctx->leaf_thresh = 20; ctx->leaf_thresh_refill_fused = 8; ctx->leaf_thresh_refill_trace = 12; }   // the defaults
        else ctx->leaf_thresh = ctx->leaf_thresh_refill_fused = ctx->leaf_thresh_refill_trace = (uint32_t)value;
    } else if (!std::strcmp(name, "leaf_thresh_coherent")) {
        if (value < 1 || value > 64) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "leaf_thresh_coherent must be 1..64");
        ctx->leaf_thresh = (uint32_t)value;
    } else if (!std::strcmp(name, "leaf_thresh_refill")) {
        if (value < 1 || value > 64) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "leaf_thresh_refill must be 1..64");
        ctx->leaf_thresh_refill_fused = ctx->leaf_thresh_refill_trace = (uint32_t)value;
    } else if (!std::strcmp(name, "queue_stride")) {
        if (value < 1 || value > 4096) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "queue_stride must be 1..4096 words");
        ctx->queue_stride = (uint32_t)value;
        for (auto &sl : ctx->slots) sl.launches = 0;
    } else if (!std::strcmp(name, "queue_block")) {
        if (value < 1 || value > 65536) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "queue_block must be 1..65536");
        ctx->queue_block_fused = ctx->queue_block_trace = (uint32_t)value;
    } else if (!std::strcmp(name, "max_blocks_per_cu")) {
        if (value < 0 || value > 64) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "max_blocks_per_cu must be 0..64");
        ctx->max_blocks_per_cu = (uint32_t)value;
    } else if (!std::strcmp(name, "validate_buffers")) {
        ctx->validate_buffers = value != 0;
    } else if (!std::strcmp(name, "launch_timing")) {
        ctx->launch_timing = value != 0;
    } else if (!std::strcmp(name, "shadow_anyhit")) {
        ctx->shadow_anyhit = value != 0;
    } else if (!std::strcmp(name, "texture_filter_8bit")) {
        ctx->tex_filter8 = value != 0;
    } else if (!std::strcmp(name, "num_queues")) {
        if (value < 1 || value > 32) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "num_queues must be 1..32");
        ctx->n_queues = (uint32_t)value;
        for (auto &sl : ctx->slots) sl.launches = 0;
    } else if (!std::strcmp(name, "kernel_variant")) {
        return rtamd_set_kernel_variant(ctx, (int)value);
    } else {
        return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "unknown option '%s'", name);
    }
    return RTAMD_OK;
}

int rtamd_upload_scene(rtamd_context *ctx, const rtamd_scene_desc *sc) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    if (!sc) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (!sc->nodes || !sc->triangles || !sc->positions || sc->node_count == 0 || sc->triangle_count == 0 || sc->vertex_count == 0)
        return fail(ctx, RTAMD_ERR_EMPTY_SCENE, "empty scene (nodes %u, triangles %u, vertices %u)", sc->node_count, sc->triangle_count, sc->vertex_count);
    if (sc->triangle_count > MAX_TRIANGLES) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "more than %u triangles are not supported", MAX_TRIANGLES);
    DeviceScope dev_scope;
    int rc = dev_scope.enter(ctx);
    if (rc) return rc;
    const RefNode *nodes = (const RefNode *)sc->nodes;
    const RefTriangle *tris = (const RefTriangle *)sc->triangles;
    const uint32_t n_nodes = sc->node_count, n_tris = sc->triangle_count, n_verts = sc->vertex_count;
    try {
        uint32_t max_mat = 0;
        for (uint32_t i = 0; i < n_tris; i++) {
            if (tris[i].p0 >= n_verts || tris[i].p1 >= n_verts || tris[i].p2 >= n_verts)
                return fail(ctx, RTAMD_ERR_MALFORMED_SCENE, "triangle %u references a vertex >= %u", i, n_verts);
            if (tris[i].mat_idx > max_mat) max_mat = tris[i].mat_idx;
        }

        // Walk the tree from the root: validates indices, rejects shared / cyclic
        // nodes, measures depth, and marks the reachable inner nodes.
        std::vector<uint8_t> state(n_nodes, 0);  // 1 = reachable inner, 2 = reachable leaf
        std::vector<std::pair<uint32_t, uint32_t>> todo;
        todo.emplace_back(0u, 0u);
        uint32_t max_depth = 0;
        // Boxes the octant slab test may not see (intersectAABB, ray_trace_and_shade.frag:53-61, is defined for any six floats;
        // slab_sel<OCT> equals it only for finite bounds with min <= max): counted here, over the reachable nodes; one such
        // box routes every visit of the scene through the general test (SceneDev::octant_ok).
        uint32_t odd_boxes = 0;
        while (!todo.empty()) {
            auto [ni, depth] = todo.back();
            todo.pop_back();
            if (ni >= n_nodes) return fail(ctx, RTAMD_ERR_MALFORMED_SCENE, "node index %u out of range (%u nodes)", ni, n_nodes);
            if (state[ni]) return fail(ctx, RTAMD_ERR_MALFORMED_SCENE, "node %u is reachable twice (not a tree)", ni);
            if (depth > max_depth) max_depth = depth;
            const RefNode &nd = nodes[ni];
            {
                bool odd = false;
                for (int k = 0; k < 3; k++) odd = odd || !std::isfinite(nd.bounds[k]) || !std::isfinite(nd.bounds[3 + k]) || !(nd.bounds[k] <= nd.bounds[3 + k]);
                if (odd) odd_boxes++;
            }
            if (nd.is_leaf) {
                state[ni] = 2;
                if ((uint64_t)nd.a + nd.b > n_tris) return fail(ctx, RTAMD_ERR_MALFORMED_SCENE, "leaf %u covers triangles [%u, %u+%u) beyond %u", ni, nd.a, nd.a, nd.b, n_tris);
            } else {
                state[ni] = 1;
                todo.emplace_back(nd.b, depth + 1);
                todo.emplace_back(nd.a, depth + 1);
            }
        }
        if (max_depth > (uint32_t)MAX_STACK_ENTRIES)
            return fail(ctx, RTAMD_ERR_MALFORMED_SCENE, "tree depth %u exceeds the supported traversal stack (%d; the reference shader's is 100)", max_depth, MAX_STACK_ENTRIES);

        std::vector<uint32_t> inner_index(n_nodes, 0);
        uint32_t n_inner = 0;
        for (uint32_t i = 0; i < n_nodes; i++) if (state[i] == 1) inner_index[i] = n_inner++;

        std::vector<BigLeaf> big;
        auto make_ref = [&](uint32_t ni) -> uint32_t {
            const RefNode &nd = nodes[ni];
            if (!nd.is_leaf) return inner_index[ni];
            if (nd.b >= 1 && nd.b <= REF_INLINE_MAX_COUNT) return REF_LEAF_BIT | (nd.b << REF_COUNT_SHIFT) | nd.a;
            big.push_back(BigLeaf{nd.a, nd.b});
            return REF_LEAF_BIT | (uint32_t)(big.size() - 1);
        };

        std::vector<WideNode> wide(n_inner ? n_inner : 1);
        std::memset(wide.data(), 0, wide.size() * sizeof(WideNode));
        for (uint32_t i = 0; i < n_nodes; i++) {
            if (state[i] != 1) continue;
            const RefNode &nd = nodes[i];
            WideNode &w = wide[inner_index[i]];
            const RefNode &ca = nodes[nd.a], &cb = nodes[nd.b];
            for (int k = 0; k < 3; k++) {
                w.a_min[k] = ca.bounds[k]; w.a_max[k] = ca.bounds[3 + k];
                w.b_min[k] = cb.bounds[k]; w.b_max[k] = cb.bounds[3 + k];
            }
            w.ref_a = make_ref(nd.a);
            w.ref_b = make_ref(nd.b);
        }
        if (big.size() > REF_FIRST_MASK) return fail(ctx, RTAMD_ERR_MALFORMED_SCENE, "too many oversized leaves");
        const uint32_t root_ref = make_ref(0);

        std::vector<TriRecord> recs(n_tris);
        const float *pos = sc->positions;
        for (uint32_t i = 0; i < n_tris; i++) {
            const float *p0 = pos + 3 * (size_t)tris[i].p0, *p1 = pos + 3 * (size_t)tris[i].p1, *p2 = pos + 3 * (size_t)tris[i].p2;
            TriRecord &r = recs[i];
            for (int k = 0; k < 3; k++) {
                float a = ftz(p0[k]);
                r.p0[k] = a;
                r.e1[k] = ftz(ftz(p1[k]) - a);  // edge1 = p1 - p0, ray_trace_and_shade.frag:66
                r.e2[k] = ftz(ftz(p2[k]) - a);  // edge2 = p2 - p0, :67
                r.pad[k] = 0.0f;
            }
        }

        // node records and triangle records share ONE allocation (nodes first, triangles
        // after them, 128-byte aligned) so that a 32-bit byte offset from one base reaches
        // either kind: the prefetch touch of a child needs no 64-bit address arithmetic.
        const size_t node_bytes = (wide.size() * sizeof(WideNode) + 127) & ~(size_t)127;
        const size_t tri_bytes = recs.size() * sizeof(TriRecord);
        if (node_bytes + tri_bytes > 0xFFFFFFF0ull) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "scene geometry exceeds the 4 GiB device layout limit");
        // from here on the previous scene's buffers are replaced: a failure leaves the context without a scene
        ctx->has_scene = false;
        ctx->sdev.materials = nullptr; ctx->n_materials = 0;   // a material table belongs to the scene it was checked against
        ctx->scene_max_mat = max_mat;
        if ((rc = dev_alloc(ctx, ctx->nodes, node_bytes + tri_bytes))) return rc;
        HIP_TRY(ctx, hipMemcpy(ctx->nodes.p, wide.data(), wide.size() * sizeof(WideNode), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy((char *)ctx->nodes.p + node_bytes, recs.data(), tri_bytes, hipMemcpyHostToDevice));
        ctx->tris.release();
        if ((rc = dev_upload(ctx, ctx->big, big.data(), big.size() * sizeof(BigLeaf)))) return rc;
        if ((rc = dev_upload(ctx, ctx->tri_idx, tris, (size_t)n_tris * sizeof(RefTriangle)))) return rc;
        ctx->normals.release(); ctx->texcoords.release();
        if (sc->normals && (rc = dev_upload(ctx, ctx->normals, sc->normals, (size_t)n_verts * 12))) return rc;
        if (sc->tex_coords && (rc = dev_upload(ctx, ctx->texcoords, sc->tex_coords, (size_t)n_verts * 8))) return rc;

        SceneDev &s = ctx->sdev;
        s.nodes = (const WideNode *)ctx->nodes.p;
        s.tris = (const TriRecord *)((const char *)ctx->nodes.p + node_bytes);
        s.tri_byte_offset = (uint32_t)node_bytes;
        s.big = (const BigLeaf *)ctx->big.p;
        s.tri_idx = (const RefTriangle *)ctx->tri_idx.p;
        s.normals = (const float *)ctx->normals.p;
        s.texcoords = (const float *)ctx->texcoords.p;
        for (int k = 0; k < 2; k++) {
            s.tex[k] = (const uint32_t *)(ctx->tex[k].p ? ctx->tex[k].p : ctx->white.p);
            s.tex_w[k] = ctx->tex[k].p ? ctx->tex_w[k] : 1;
            s.tex_h[k] = ctx->tex[k].p ? ctx->tex_h[k] : 1;
        }
        for (int k = 0; k < 6; k++) s.root_box[k] = nodes[0].bounds[k];
        s.root_ref = root_ref;
        s.octant_ok = odd_boxes == 0 ? 1u : 0u;

        ctx->info.node_count = n_nodes; ctx->info.triangle_count = n_tris; ctx->info.vertex_count = n_verts;
        ctx->info.inner_node_count = n_inner; ctx->info.big_leaf_count = (uint32_t)big.size(); ctx->info.max_depth = max_depth;
        ctx->info.odd_box_count = odd_boxes; ctx->info.reserved = 0;
        ctx->info.device_bytes = ctx->nodes.bytes + ctx->tris.bytes + ctx->big.bytes + ctx->tri_idx.bytes + ctx->normals.bytes + ctx->texcoords.bytes;
        ctx->has_scene = true;
    } catch (const std::bad_alloc &) {
        return fail(ctx, RTAMD_ERR_OUT_OF_MEMORY, "out of host memory while converting the scene");
    }
    return RTAMD_OK;
}

namespace {
// GL_RGB / UNSIGNED_BYTE rows (taken as tightly packed, first row first) -> RGBA8 texels in HBM.
int upload_rgb8(rtamd_context *ctx, DevBuf &dst, const uint8_t *rgb8, uint32_t width, uint32_t height) {
    if (!rgb8 || width == 0 || height == 0 || width > 32768 || height > 32768) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "bad texture %ux%u", width, height);
    DeviceScope dev_scope;
    int rc = dev_scope.enter(ctx);
    if (rc) return rc;
    try {
        // The reference uploads with the default GL_UNPACK_ALIGNMENT of 4 (texture.rs:14-24 never
        // changes it); a caller that hands over tightly packed rows whose length is not a multiple
        // of 4 would be mis-read by GL.  We take the rows as tightly packed (what the caller means).
        std::vector<uint32_t> rgba((size_t)width * height);
        for (uint32_t y = 0; y < height; y++) {
            const uint8_t *row = rgb8 + (size_t)y * (size_t)width * 3;
            for (uint32_t x = 0; x < width; x++)
                rgba[(size_t)y * width + x] = (uint32_t)row[3 * x] | ((uint32_t)row[3 * x + 1] << 8) | ((uint32_t)row[3 * x + 2] << 16) | 0xFF000000u;
        }
        // The new texels go into a buffer of their own first: if the allocation or the copy fails, the slot
        // (and every pointer the kernels hold to it: sdev.tex[], the tex_slots table) keeps the old, live texels.
        DevBuf fresh;
        if ((rc = dev_upload(ctx, fresh, rgba.data(), rgba.size() * 4))) { fresh.release(); return rc; }
        // a frame in flight may still sample the old texels
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { fresh.release(); return fail(ctx, RTAMD_ERR_DEVICE, "hipDeviceSynchronize failed: %s", hipGetErrorString(e)); }
        dst.release();
        dst = fresh;
    } catch (const std::bad_alloc &) {
        return fail(ctx, RTAMD_ERR_OUT_OF_MEMORY, "out of host memory");
    }
    return RTAMD_OK;
}
}  // namespace

namespace {
// The slot table the material path samples through: RTAMD_MAX_TEXTURE_SLOTS entries + the white texel.
int refresh_tex_slots(rtamd_context *ctx) {
    TexSlotDev table[RTAMD_MAX_TEXTURE_SLOTS + 1];
    for (int k = 0; k <= RTAMD_MAX_TEXTURE_SLOTS; k++) {
        const bool have = k < RTAMD_MAX_TEXTURE_SLOTS && ctx->tex[k].p;
        table[k].texels = (const uint32_t *)(have ? ctx->tex[k].p : ctx->white.p);
        table[k].w = have ? ctx->tex_w[k] : 1;
        table[k].h = have ? ctx->tex_h[k] : 1;
    }
    HIP_TRY(ctx, hipDeviceSynchronize());   // a frame in flight may still read the old table
    if (!ctx->tex_slots.p) { int rc = dev_alloc(ctx, ctx->tex_slots, sizeof table); if (rc) return rc; }
    HIP_TRY(ctx, hipMemcpy(ctx->tex_slots.p, table, sizeof table, hipMemcpyHostToDevice));
    ctx->sdev.tex_slots = (const TexSlotDev *)ctx->tex_slots.p;
    return RTAMD_OK;
}
}  // namespace

int rtamd_upload_texture(rtamd_context *ctx, uint32_t slot, const uint8_t *rgb8, uint32_t width, uint32_t height) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    if (slot >= RTAMD_MAX_TEXTURE_SLOTS) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "texture slot %u: slots are 0..%d (the shader's modelTextures[2] are slots 0 and 1)", slot, RTAMD_MAX_TEXTURE_SLOTS - 1);
    int rc = upload_rgb8(ctx, ctx->tex[slot], rgb8, width, height);
    if (rc) return rc;
    ctx->tex_w[slot] = (int32_t)width; ctx->tex_h[slot] = (int32_t)height;
    if (slot < 2) {
        ctx->sdev.tex[slot] = (const uint32_t *)ctx->tex[slot].p;
        ctx->sdev.tex_w[slot] = (int32_t)width; ctx->sdev.tex_h[slot] = (int32_t)height;
    }
    if (ctx->sdev.materials && (rc = refresh_tex_slots(ctx))) {
        // the slot table could not be rewritten and may still name the texels just freed: no launch may read it
        ctx->sdev.materials = nullptr; ctx->n_materials = 0;
        return rc;
    }
    return RTAMD_OK;
}

int rtamd_upload_materials(rtamd_context *ctx, const rtamd_shading_material *materials, uint32_t count) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    if (!ctx->has_scene) return fail(ctx, RTAMD_ERR_NO_SCENE, "no scene uploaded (the table is checked against the scene's material indices)");
    DeviceScope dev_scope;
    int rc = dev_scope.enter(ctx);
    if (rc) return rc;
    if (count == 0) {   // back to the shader's constants
        HIP_TRY(ctx, hipDeviceSynchronize());
        ctx->sdev.materials = nullptr; ctx->n_materials = 0;
        return RTAMD_OK;
    }
    if (!materials) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "materials is NULL");
    if (count <= ctx->scene_max_mat) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "%u materials, but the scene's triangles use material index %u", count, ctx->scene_max_mat);
    try {
        std::vector<MaterialDev> table(count);
        for (uint32_t i = 0; i < count; i++) {
            const rtamd_shading_material &m = materials[i];
            if (m.texture_slot < -1 || m.texture_slot >= RTAMD_MAX_TEXTURE_SLOTS)
                return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "material %u: texture slot %d is not -1..%d", i, m.texture_slot, RTAMD_MAX_TEXTURE_SLOTS - 1);
            MaterialDev &d = table[i];
            for (int k = 0; k < 3; k++) { d.ka[k] = m.ambient_color[k]; d.kd[k] = m.diffuse_color[k]; d.ks[k] = m.specular_color[k]; }
            d.ns = m.specular_exp;
            d.tex_slot = m.texture_slot < 0 ? (uint32_t)RTAMD_MAX_TEXTURE_SLOTS : (uint32_t)m.texture_slot;
            d.pad = 0;
        }
        HIP_TRY(ctx, hipDeviceSynchronize());   // a frame in flight may still read the old table
        if ((rc = dev_upload(ctx, ctx->materials, table.data(), table.size() * sizeof(MaterialDev)))) { ctx->sdev.materials = nullptr; ctx->n_materials = 0; return rc; }
    } catch (const std::bad_alloc &) {
        return fail(ctx, RTAMD_ERR_OUT_OF_MEMORY, "out of host memory");
    }
    if ((rc = refresh_tex_slots(ctx))) { ctx->sdev.materials = nullptr; ctx->n_materials = 0; return rc; }
    ctx->sdev.materials = (const MaterialDev *)ctx->materials.p;
    ctx->n_materials = count;
    return RTAMD_OK;
}

int rtamd_upload_noise_texture(rtamd_context *ctx, const uint8_t *rgb8, uint32_t width, uint32_t height) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    int rc = upload_rgb8(ctx, ctx->noise, rgb8, width, height);
    if (rc) return rc;
    ctx->noise_w = (int32_t)width; ctx->noise_h = (int32_t)height;
    return RTAMD_OK;
}

int rtamd_dispatch_rays_device(rtamd_context *ctx, const rtamd_dispatch_desc *d, const rtamd_dispatch_out *o) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    if (!d || !o) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "dispatch descriptor / output is NULL");
    if (!ctx->noise.p) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "no noise texture uploaded (rtamd_upload_noise_texture)");
    if (d->width == 0 || d->height == 0 || (uint64_t)d->width * d->height > 0x7FFFFFFFull) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "bad frame size %ux%u", d->width, d->height);
    if (!d->position || !d->normal_mat) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "position / normal_mat is NULL");
    if (((uintptr_t)d->position | (uintptr_t)d->normal_mat | (uintptr_t)o->org | (uintptr_t)o->shadow_dir | (uintptr_t)o->reflect_dir | (uintptr_t)o->ambient_dir) & 15u)
        return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "device buffers must be 16-byte aligned");
    const uint32_t tile_world = d->tile_world ? d->tile_world : 1u;     // 0 (a zero-initialised descriptor) means the whole frame
    if (d->tile_rank >= tile_world) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "tile_rank %u / tile_world %u invalid", d->tile_rank, d->tile_world);
    uint32_t tiles_x = 0;
    const uint32_t n_tiles = n_tiles_of(d->width, d->height, &tiles_x);
    const uint64_t n_entries = tile_world == 1 ? (uint64_t)d->width * d->height : (uint64_t)local_tiles(n_tiles, d->tile_rank, tile_world) * RTAMD_TILE_PIXELS;
    if (n_entries == 0) return RTAMD_OK;
    DeviceScope dev_scope;
    int rc = dev_scope.enter(ctx);
    if (rc) return rc;
    {
        const uint64_t n = n_entries;
        if ((rc = check_range(ctx, d->position, n * 12, "position")) || (rc = check_range(ctx, d->normal_mat, n * 16, "normal_mat")) ||
            (rc = check_range(ctx, o->org, n * 12, "org")) || (rc = check_range(ctx, o->shadow_dir, n * 12, "shadow_dir")) ||
            (rc = check_range(ctx, o->reflect_dir, n * 12, "reflect_dir")) || (rc = check_range(ctx, o->ambient_dir, n * 12, "ambient_dir")))
            return rc;
    }
    DispatchParams p{};
    p.position = d->position; p.normal_mat = d->normal_mat;
    p.noise = (const uint32_t *)ctx->noise.p; p.noise_w = ctx->noise_w; p.noise_h = ctx->noise_h;
    std::memcpy(p.light, d->light_pos, sizeof p.light);
    std::memcpy(p.cam, d->camera_pos, sizeof p.cam);
    std::memcpy(p.nos, d->noise_offset_scale, sizeof p.nos);
    p.width = d->width; p.height = d->height; p.n = (uint32_t)n_entries;
    p.rank = d->tile_rank; p.world = tile_world; p.tiles_x = tiles_x;
    p.tex_filter8 = ctx->tex_filter8 ? 1u : 0u;
    p.out[0] = o->org; p.out[1] = o->shadow_dir; p.out[2] = o->reflect_dir; p.out[3] = o->ambient_dir;
    hipLaunchKernelGGL(k_dispatch_rays, dim3((p.n + 255u) / 256u), dim3(256), 0, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return RTAMD_OK;
}

int rtamd_dispatch_rays(rtamd_context *ctx, const rtamd_dispatch_desc *d, const rtamd_dispatch_out *o) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    if (!d || !o) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "dispatch descriptor / output is NULL");
    if (d->width == 0 || d->height == 0 || !d->position || !d->normal_mat) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "bad dispatch arguments");
    DeviceScope dev_scope;
    int rc = dev_scope.enter(ctx);
    if (rc) return rc;
    const size_t n = (size_t)d->width * d->height;
    DevBuf d_pos, d_nm, d_out[4];
    float *host_out[4] = {o->org, o->shadow_dir, o->reflect_dir, o->ambient_dir};
    rtamd_dispatch_desc dd = *d;
    dd.tile_rank = 0; dd.tile_world = 1;      // the host-buffer form takes whole frames
    rtamd_dispatch_out od{};
    float **dev_out[4] = {&od.org, &od.shadow_dir, &od.reflect_dir, &od.ambient_dir};
    rc = RTAMD_OK;
    do {
        if ((rc = dev_upload(ctx, d_pos, d->position, n * 12))) break;
        if ((rc = dev_upload(ctx, d_nm, d->normal_mat, n * 16))) break;
        dd.position = (const float *)d_pos.p; dd.normal_mat = (const float *)d_nm.p;
        for (int k = 0; k < 4 && !rc; k++)
            if (host_out[k]) { if (!(rc = dev_alloc(ctx, d_out[k], n * 12))) *dev_out[k] = (float *)d_out[k].p; }
        if (rc) break;
        if ((rc = rtamd_dispatch_rays_device(ctx, &dd, &od))) break;
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { rc = fail(ctx, RTAMD_ERR_DEVICE, "dispatch kernel failed: %s", hipGetErrorString(e)); break; }
        for (int k = 0; k < 4; k++)
            if (host_out[k] && (e = hipMemcpy(host_out[k], d_out[k].p, n * 12, hipMemcpyDeviceToHost)) != hipSuccess) { rc = fail(ctx, RTAMD_ERR_DEVICE, "copy-back failed: %s", hipGetErrorString(e)); break; }
    } while (0);
    d_pos.release(); d_nm.release();
    for (auto &b : d_out) b.release();
    return rc;
}

int rtamd_render_device(rtamd_context *ctx, const rtamd_frame_desc *desc, const rtamd_frame_out *out) {
    int rc = check_frame(ctx, desc, out);
    if (rc) return rc;
    DeviceScope dev_scope;
    if ((rc = dev_scope.enter(ctx))) return rc;
    if (out->rgba8 && !out->rgba32f) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "rgba8 is made from rgba32f: pass both");
    if ((rc = check_frame_buffers(ctx, desc, out))) return rc;
    JobParams p;
    fill_job(ctx, desc, out, p);
    if ((rc = launch_main(ctx, MODE_FUSED, out->counts != nullptr, p))) return rc;
    if (out->rgba8) {
        // The display conversion runs as its own small launch behind the render kernel, on the same stream.  (Writing
        // the 8-bit pixel from inside k_persistent was measured: the three extra conversions at each of the kernel's
        // store sites cost the fused shader 18 % through register pressure; a separate instantiation was not worth it.)
        const uint32_t n_tiles = n_tiles_of(desc->width, desc->height, nullptr);
        const uint64_t n_out = desc->tile_world == 1 ? (uint64_t)desc->width * desc->height
                                                     : (uint64_t)local_tiles(n_tiles, desc->tile_rank, desc->tile_world) * RTAMD_TILE_PIXELS;
        return rtamd_display_rgba8_device(ctx, out->rgba32f, n_out, out->rgba8);
    }
    return RTAMD_OK;
}

int rtamd_render_batch_device(rtamd_context *ctx, uint32_t n_frames, const rtamd_frame_desc *descs, const rtamd_frame_out *outs) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    if (n_frames == 0 || n_frames > RTAMD_MAX_BATCH_FRAMES) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "a batch holds 1..%d frames, not %u", RTAMD_MAX_BATCH_FRAMES, n_frames);
    if (!descs || !outs) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "frame descriptors / outputs are NULL");
    int rc;
    bool any_counts = false;
    for (uint32_t f = 0; f < n_frames; f++) {
        if ((rc = check_frame(ctx, &descs[f], &outs[f]))) return rc;
        if (outs[f].rgba8 && !outs[f].rgba32f) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "frame %u: rgba8 is made from rgba32f: pass both", f);
        const rtamd_frame_desc &a = descs[0], &b = descs[f];
        if (a.width != b.width || a.height != b.height || a.tile_rank != b.tile_rank || a.tile_world != b.tile_world ||
            (a.has_tex_coords != 0) != (b.has_tex_coords != 0) || (a.has_normals != 0) != (b.has_normals != 0))
            return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "frame %u: the frames of a batch share size, tile shard and the has_tex_coords / has_normals flags", f);
        any_counts = any_counts || outs[f].counts != nullptr;
    }
    if (n_frames == 1) return rtamd_render_device(ctx, descs, outs);
    if (frame_out_pixels(&descs[0]) >= (1ull << BATCH_FRAME_SHIFT) || (uint64_t)descs[0].width * descs[0].height >= (1ull << BATCH_FRAME_SHIFT))
        return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "frames of a batch are limited to 2^%u - 1 pixels", BATCH_FRAME_SHIFT);
    if (ctx->variant != 0) {   // the A/B baseline kernel has no batch form: frame by frame
        for (uint32_t f = 0; f < n_frames; f++) if ((rc = rtamd_render_device(ctx, &descs[f], &outs[f]))) return rc;
        return RTAMD_OK;
    }
    DeviceScope dev_scope;
    if ((rc = dev_scope.enter(ctx))) return rc;
    for (uint32_t f = 0; f < n_frames; f++) if ((rc = check_frame_buffers(ctx, &descs[f], &outs[f]))) return rc;
    JobParams p;
    fill_job(ctx, &descs[0], &outs[0], p);
    if (p.n_items == 0) return RTAMD_OK;
    p.n_frames = n_frames;
    p.items_per_frame = p.n_items;
    p.n_items = p.items_per_frame * n_frames;     // <= 8 * 2^28 / 64 items
    for (uint32_t f = 0; f < n_frames; f++) {
        FrameIO &io = p.frames[f];
        io.dir = descs[f].dir; io.org = descs[f].org;
        io.rgba = outs[f].rgba32f; io.hit_id = outs[f].hit_id; io.hit_t = outs[f].hit_t; io.counts = outs[f].counts;
        std::memcpy(io.ipv, descs[f].inv_proj_view, sizeof io.ipv);
        io.near_ = descs[f].near_plane; io.far_ = descs[f].far_plane;
        std::memcpy(io.cam, descs[f].cam_pos, sizeof io.cam);
    }
    if ((rc = launch_main(ctx, MODE_FUSED, any_counts, p))) return rc;
    for (uint32_t f = 0; f < n_frames; f++)
        if (outs[f].rgba8 && (rc = rtamd_display_rgba8_device(ctx, outs[f].rgba32f, frame_out_pixels(&descs[f]), outs[f].rgba8))) return rc;
    return RTAMD_OK;
}

int rtamd_render_device_static(rtamd_context *ctx, const rtamd_frame_desc *desc, const rtamd_frame_out *out) {
    int rc = check_frame(ctx, desc, out);
    if (rc) return rc;
    if (out->rgba8 && !out->rgba32f) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "rgba8 is made from rgba32f: pass both");
    if (ctx->info.max_depth > STATIC_MAX_DEPTH)
        return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "the scene's tree is %u levels deep; the static launch scratch holds traversal stacks for at most %u "
                    "(use rtamd_render_device, which allocates its stack columns)", ctx->info.max_depth, STATIC_MAX_DEPTH);
    DeviceScope dev_scope;
    if ((rc = dev_scope.enter(ctx))) return rc;
    if ((rc = check_frame_buffers(ctx, desc, out))) return rc;
    uint32_t *const ptr = ctx->static_scratch;   // this GPU's copy of `scratch` (hipGetSymbolAddress at context creation)
    JobParams p;
    fill_job(ctx, desc, out, p);
    if (p.n_items == 0) return RTAMD_OK;
    p.queues = ptr;
    p.n_queues = STATIC_QUEUES;
    p.queue_stride = STATIC_QUEUE_STRIDE;
    p.queue_block = ctx->queue_block_fused;
    p.dyn_thresh = ctx->dyn_thresh_fused; p.dyn_auto = 0; p.dyn_incoherent = ctx->dyn_incoherent;
    p.stats_prev = nullptr; p.stats_cur = nullptr;   // no coherence feedback: the re-fill threshold is the context's fixed one
    p.leaf_thresh = ctx->leaf_thresh; p.leaf_thresh_refill = ctx->leaf_thresh_refill_fused;
    p.shadow_anyhit = ctx->shadow_anyhit ? 1u : 0u;
    p.tex_filter8 = ctx->tex_filter8 ? 1u : 0u;
    p.spill = ptr + STATIC_QUEUE_WORDS;
    p.spill_wave_words = STATIC_SPILL_ENTRIES * 64u;
#ifdef RTAMD_EXP_STAMPS
    p.stamps = nullptr;
#endif
    hipStream_t st = ctx->stream;
    // One static call at a time per GPU: this call's counter reset and kernel wait (on the GPU) for the kernel of the previous
    // static call on this device, whatever stream or context issued it, and the host-side sequence wait -> reset -> launch ->
    // record is atomic under the mutex, so two threads cannot interleave theirs.  Neither call allocates.
    std::lock_guard<std::mutex> static_lock(g_scratch_mutex);
    HIP_TRY(ctx, hipStreamWaitEvent(st, g_static_done[ctx->device], 0));
    // the array keeps the previous call's contents: the counters must start at zero (the stack columns need no clearing:
    // an entry is written before it is read)
    HIP_TRY(ctx, hipMemsetAsync(ptr, 0, STATIC_QUEUE_WORDS * sizeof(uint32_t), st));
    const SceneDev &s = ctx->sdev;
    const bool mat = s.materials != nullptr, count = out->counts != nullptr;
    int per_cu = ctx->persistent_blocks_per_cu[(mat ? 4 : 0) + (count ? 1 : 0)];
    if (ctx->max_blocks_per_cu && (int)ctx->max_blocks_per_cu < per_cu) per_cu = (int)ctx->max_blocks_per_cu;
    uint32_t grid = (uint32_t)(ctx->num_cu * per_cu);
    const uint32_t need = (p.n_items + (BLOCK / 64) - 1) / (BLOCK / 64), cap = STATIC_WAVES / (BLOCK / 64);
    if (grid > need) grid = need;
    if (grid > cap) grid = cap;     // the stack columns exist for STATIC_WAVES waves
    if (grid < 1) grid = 1;
    launch_persistent_fused(count, mat, ctx->tex_filter8, grid, st, s, p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(g_static_done[ctx->device], st));
    if (out->rgba8) {
        const uint32_t n_tiles = n_tiles_of(desc->width, desc->height, nullptr);
        const uint64_t n_out = desc->tile_world == 1 ? (uint64_t)desc->width * desc->height
                                                     : (uint64_t)local_tiles(n_tiles, desc->tile_rank, desc->tile_world) * RTAMD_TILE_PIXELS;
        return rtamd_display_rgba8_device(ctx, out->rgba32f, n_out, out->rgba8);
    }
    return RTAMD_OK;
}

int rtamd_render(rtamd_context *ctx, const rtamd_frame_desc *desc, const rtamd_frame_out *out) {
    int rc = check_frame(ctx, desc, out);
    if (rc) return rc;
    DeviceScope dev_scope;
    if ((rc = dev_scope.enter(ctx))) return rc;
    const size_t n_px = (size_t)desc->width * desc->height;
    uint32_t n_tiles = n_tiles_of(desc->width, desc->height, nullptr);
    const size_t n_out = desc->tile_world == 1 ? n_px : (size_t)local_tiles(n_tiles, desc->tile_rank, desc->tile_world) * RTAMD_TILE_PIXELS;
    DevBuf d_dir, d_org, d_rgba, d_id, d_t, d_cnt, d_rgba8;
    auto cleanup = [&]() { d_dir.release(); d_org.release(); d_rgba.release(); d_id.release(); d_t.release(); d_cnt.release(); d_rgba8.release(); };
    rtamd_frame_desc dd = *desc;
    rtamd_frame_out od{};
    rc = RTAMD_OK;
    do {
        if (desc->dir) {
            if ((rc = dev_upload(ctx, d_dir, desc->dir, n_px * 12))) break;
            if ((rc = dev_upload(ctx, d_org, desc->org, n_px * 12))) break;
            dd.dir = (const float *)d_dir.p; dd.org = (const float *)d_org.p;
        }
        if (out->rgba32f) { if ((rc = dev_alloc(ctx, d_rgba, n_out * 16))) break; od.rgba32f = (float *)d_rgba.p; }
        if (out->hit_id) { if ((rc = dev_alloc(ctx, d_id, n_out * 4))) break; od.hit_id = (uint32_t *)d_id.p; }
        if (out->hit_t) { if ((rc = dev_alloc(ctx, d_t, n_out * 4))) break; od.hit_t = (float *)d_t.p; }
        if (out->counts) { if ((rc = dev_alloc(ctx, d_cnt, n_out * 8))) break; od.counts = (uint32_t *)d_cnt.p; }
        if (out->rgba8) { if ((rc = dev_alloc(ctx, d_rgba8, n_out * 4))) break; od.rgba8 = (uint32_t *)d_rgba8.p; }
        // pixels of partial edge tiles that fall outside the frame are never written: define them as 0
        if (desc->tile_world != 1) {
            hipError_t e = hipSuccess;
            if (od.rgba32f) e = hipMemsetAsync(od.rgba32f, 0, n_out * 16, ctx->stream);
            if (e == hipSuccess && od.hit_id) e = hipMemsetAsync(od.hit_id, 0, n_out * 4, ctx->stream);
            if (e == hipSuccess && od.hit_t) e = hipMemsetAsync(od.hit_t, 0, n_out * 4, ctx->stream);
            if (e == hipSuccess && od.counts) e = hipMemsetAsync(od.counts, 0, n_out * 8, ctx->stream);
            if (e == hipSuccess && od.rgba8) e = hipMemsetAsync(od.rgba8, 0, n_out * 4, ctx->stream);
            if (e != hipSuccess) { rc = fail(ctx, RTAMD_ERR_DEVICE, "hipMemsetAsync failed: %s", hipGetErrorString(e)); break; }
        }
        if ((rc = rtamd_render_device(ctx, &dd, &od))) break;
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { rc = fail(ctx, RTAMD_ERR_DEVICE, "render kernel failed: %s", hipGetErrorString(e)); break; }
        if (out->rgba32f && (e = hipMemcpy(out->rgba32f, od.rgba32f, n_out * 16, hipMemcpyDeviceToHost)) != hipSuccess) { rc = fail(ctx, RTAMD_ERR_DEVICE, "copy-back failed: %s", hipGetErrorString(e)); break; }
        if (out->hit_id && (e = hipMemcpy(out->hit_id, od.hit_id, n_out * 4, hipMemcpyDeviceToHost)) != hipSuccess) { rc = fail(ctx, RTAMD_ERR_DEVICE, "copy-back failed: %s", hipGetErrorString(e)); break; }
        if (out->hit_t && (e = hipMemcpy(out->hit_t, od.hit_t, n_out * 4, hipMemcpyDeviceToHost)) != hipSuccess) { rc = fail(ctx, RTAMD_ERR_DEVICE, "copy-back failed: %s", hipGetErrorString(e)); break; }
        if (out->rgba8 && (e = hipMemcpy(out->rgba8, od.rgba8, n_out * 4, hipMemcpyDeviceToHost)) != hipSuccess) { rc = fail(ctx, RTAMD_ERR_DEVICE, "copy-back failed: %s", hipGetErrorString(e)); break; }
        if (out->counts && (e = hipMemcpy(out->counts, od.counts, n_out * 8, hipMemcpyDeviceToHost)) != hipSuccess) { rc = fail(ctx, RTAMD_ERR_DEVICE, "copy-back failed: %s", hipGetErrorString(e)); break; }
    } while (0);
    cleanup();
    return rc;
}

int rtamd_trace_device(rtamd_context *ctx, uint64_t n, const float *org, const float *dir, float *hits) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    if (!ctx->has_scene) return fail(ctx, RTAMD_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return RTAMD_OK;
    if (!org || !dir || !hits) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "org / dir / hits is NULL");
    if (n > 0xFFFFFFFFull) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "at most 2^32-1 rays per call");
    DeviceScope dev_scope;
    int rc = dev_scope.enter(ctx);
    if (rc) return rc;
    if ((rc = check_range(ctx, org, n * 12, "org")) || (rc = check_range(ctx, dir, n * 12, "dir")) || (rc = check_range(ctx, hits, n * 16, "hits"))) return rc;
    JobParams p;
    std::memset(&p, 0, sizeof p);
    p.dir = dir; p.org = org; p.hits = hits; p.n_rays = n;
    p.n_items = (uint32_t)((n + 63) / 64);
    p.world = 1;
    p.n_frames = 1;
    return launch_main(ctx, MODE_TRACE, false, p);
}

namespace {

// Work items of a ray list of n rays: 64 consecutive rays each, or -- for a list that is an image -- 8x8 pixel blocks
// (layout 1: scan-line order, `width` pixels per row; layout 2: the compact tile-major layout, 16 blocks per 32x32 tile).
uint32_t trace_items(uint64_t n, int layout, uint32_t width) {
    if (layout == 1) { const uint64_t rows = (n + width - 1) / width; return (uint32_t)(((uint64_t)(width + 7) / 8) * ((rows + 7) / 8)); }
    if (layout == 2) return (uint32_t)((n + RTAMD_TILE_PIXELS - 1) / RTAMD_TILE_PIXELS) * 16u;
    return (uint32_t)((n + 63) / 64);
}

int trace_batch(rtamd_context *ctx, uint32_t n_jobs, const rtamd_trace_job *jobs, int layout, uint32_t width) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    if (!ctx->has_scene) return fail(ctx, RTAMD_ERR_NO_SCENE, "no scene uploaded");
    if (n_jobs == 0 || n_jobs > RTAMD_MAX_BATCH_FRAMES) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "a batch holds 1..%d ray lists, not %u", RTAMD_MAX_BATCH_FRAMES, n_jobs);
    if (!jobs) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "jobs is NULL");
    if (layout == 1 && width == 0) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "an image in scan-line order needs its width");
    for (uint32_t j = 0; j < n_jobs; j++) {
        if (jobs[j].n && (!jobs[j].org || !jobs[j].dir || !jobs[j].hits)) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "job %u: org / dir / hits is NULL", j);
        if (jobs[j].n >= (1ull << BATCH_FRAME_SHIFT) && (n_jobs > 1 || layout != 0))
            return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "job %u: the ray lists of a batch are limited to 2^%u - 1 rays", j, BATCH_FRAME_SHIFT);
        if (layout == 1 && jobs[j].n % width) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "job %u: %llu rays are not rows of %u pixels", j, (unsigned long long)jobs[j].n, width);
        if (layout == 2 && jobs[j].n % RTAMD_TILE_PIXELS) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "job %u: %llu rays are not whole 32x32 tiles", j, (unsigned long long)jobs[j].n);
    }
    if (layout == 1) {
        // images of at most 8 columns, or so large that the kernel's division by multiplication would leave its exact range: the same
        // rays as plain lists (only the grouping into work items differs, never a record)
        uint64_t most = 0;
        for (uint32_t j = 0; j < n_jobs; j++) most = std::max<uint64_t>(most, trace_items(jobs[j].n, 1, width));
        if (width <= 8 || most * ((width + 7u) / 8u) >= (1ull << 32)) layout = 0;
    }
    int rc;
    if (ctx->variant != 0 || (n_jobs == 1 && layout == 0)) {      // the A/B baseline kernel has neither a batch form nor block items: list by list
        for (uint32_t j = 0; j < n_jobs; j++) if ((rc = rtamd_trace_device(ctx, jobs[j].n, jobs[j].org, jobs[j].dir, jobs[j].hits))) return rc;
        return RTAMD_OK;
    }
    DeviceScope dev_scope;
    if ((rc = dev_scope.enter(ctx))) return rc;
    JobParams p;
    std::memset(&p, 0, sizeof p);
    p.world = 1;
    p.n_frames = n_jobs;
    p.width = layout == 1 ? width : 0u;
    p.tiles_x = layout == 2 ? 1u : 0u;
    if (layout == 1) {      // the kernel divides item indices by the blocks per row with one multiplication: ceil(2^32 / d) is exact for items < 2^32 / d
        const uint32_t bpr = (width + 7u) / 8u;
        p.height = (uint32_t)(((1ull << 32) + bpr - 1) / bpr);
    }
    uint32_t items = 0;
    for (uint32_t j = 0; j < n_jobs; j++) {
        const uint64_t n = jobs[j].n;
        if ((rc = check_range(ctx, jobs[j].org, n * 12, "org")) || (rc = check_range(ctx, jobs[j].dir, n * 12, "dir")) || (rc = check_range(ctx, jobs[j].hits, n * 16, "hits"))) return rc;
        p.job_item_begin[j] = items;
        items += trace_items(n, layout, width);          // <= 16 * 2^22 items
        FrameIO &io = p.frames[j];
        io.org = jobs[j].org; io.dir = jobs[j].dir; io.rgba = jobs[j].hits; io.n_rays = (uint32_t)n;
    }
    for (uint32_t j = n_jobs; j <= (uint32_t)MAX_BATCH_FRAMES; j++) p.job_item_begin[j] = items;
    p.n_items = items;
    if (n_jobs == 1) { p.dir = jobs[0].dir; p.org = jobs[0].org; p.hits = jobs[0].hits; p.n_rays = jobs[0].n; }    // one list: the kernel's plain arguments
    return launch_main(ctx, MODE_TRACE, false, p);     // n_items == 0 (every list empty): nothing to launch
}

}  // namespace

int rtamd_trace_batch_device(rtamd_context *ctx, uint32_t n_jobs, const rtamd_trace_job *jobs) { return trace_batch(ctx, n_jobs, jobs, 0, 0); }

int rtamd_trace_image_batch_device(rtamd_context *ctx, uint32_t n_jobs, const rtamd_trace_job *jobs, uint32_t width, uint32_t tile_world) {
    return trace_batch(ctx, n_jobs, jobs, tile_world > 1 ? 2 : 1, width);
}

int rtamd_trace(rtamd_context *ctx, uint64_t n, const float *org, const float *dir, float *hits) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    if (!ctx->has_scene) return fail(ctx, RTAMD_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return RTAMD_OK;
    if (!org || !dir || !hits) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "org / dir / hits is NULL");
    if (n > 0xFFFFFFFFull) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "at most 2^32-1 rays per call");
    DeviceScope dev_scope;
    int rc = dev_scope.enter(ctx);
    if (rc) return rc;
    DevBuf d_org, d_dir, d_hits;
    do {
        if ((rc = dev_upload(ctx, d_org, org, n * 12))) break;
        if ((rc = dev_upload(ctx, d_dir, dir, n * 12))) break;
        if ((rc = dev_alloc(ctx, d_hits, n * 16))) break;
        if ((rc = rtamd_trace_device(ctx, n, (const float *)d_org.p, (const float *)d_dir.p, (float *)d_hits.p))) break;
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { rc = fail(ctx, RTAMD_ERR_DEVICE, "trace kernel failed: %s", hipGetErrorString(e)); break; }
        if ((e = hipMemcpy(hits, d_hits.p, n * 16, hipMemcpyDeviceToHost)) != hipSuccess) { rc = fail(ctx, RTAMD_ERR_DEVICE, "copy-back failed: %s", hipGetErrorString(e)); break; }
    } while (0);
    d_org.release(); d_dir.release(); d_hits.release();
    return rc;
}

int rtamd_generate_rays_offset_device(rtamd_context *ctx, uint32_t width, uint32_t height, const float inv_proj_view[16],
                                      float near_plane, float far_plane, float offset_x, float offset_y, float *dir_out) {
    return rtamd_generate_rays_tiles_device(ctx, width, height, inv_proj_view, near_plane, far_plane, offset_x, offset_y, 0, 1, dir_out);
}

int rtamd_generate_rays_tiles_device(rtamd_context *ctx, uint32_t width, uint32_t height, const float inv_proj_view[16],
                                     float near_plane, float far_plane, float offset_x, float offset_y,
                                     uint32_t tile_rank, uint32_t tile_world, float *dir_out) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    if (!inv_proj_view || !dir_out || width == 0 || height == 0) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "bad generate_rays arguments");
    if ((uint64_t)width * height > 0x7FFFFFFFull) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "frame too large");
    if (!(offset_x >= 0.0f && offset_x <= 1.0f && offset_y >= 0.0f && offset_y <= 1.0f)) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "sample offset must lie in [0,1]^2");
    if (tile_world == 0 || tile_rank >= tile_world) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "tile_rank %u / tile_world %u invalid", tile_rank, tile_world);
    uint32_t tiles_x = 0;
    const uint32_t n_tiles = n_tiles_of(width, height, &tiles_x);
    const uint64_t n_entries = tile_world == 1 ? (uint64_t)width * height : (uint64_t)local_tiles(n_tiles, tile_rank, tile_world) * RTAMD_TILE_PIXELS;
    if (n_entries == 0) return RTAMD_OK;
    DeviceScope dev_scope;
    int rc = dev_scope.enter(ctx);
    if (rc) return rc;
    if ((rc = check_range(ctx, dir_out, n_entries * 12, "dir_out"))) return rc;
    // the matrix travels in the kernel arguments: nothing of the caller's memory is read after return, and the call
    // does not wait for the stream (round 1 staged it through one device buffer and had to synchronise per call)
    Mat4Arg ipv;
    std::memcpy(ipv.m, inv_proj_view, sizeof ipv.m);
    const uint32_t n = (uint32_t)n_entries;
    hipLaunchKernelGGL(k_generate_rays, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ipv, near_plane, far_plane, width, height,
                       offset_x, offset_y, tile_rank, tile_world, tiles_x, n, dir_out);
    HIP_TRY(ctx, hipGetLastError());
    return RTAMD_OK;
}

int rtamd_generate_rays_device(rtamd_context *ctx, uint32_t width, uint32_t height, const float inv_proj_view[16],
                               float near_plane, float far_plane, float *dir_out) {
    return rtamd_generate_rays_offset_device(ctx, width, height, inv_proj_view, near_plane, far_plane, 0.5f, 0.5f, dir_out);
}

int rtamd_surface_from_hits_device(rtamd_context *ctx, uint64_t n, const float *org, const float *dir, const float *hits, uint32_t has_normals,
                                   float *position, float *normal_mat) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    if (!ctx->has_scene) return fail(ctx, RTAMD_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return RTAMD_OK;
    if (!org || !dir || !hits || !position || !normal_mat) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "surface_from_hits: NULL buffer");
    if (n > 0xFFFFFFFFull) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "at most 2^32-1 rays per call");
    if (has_normals && !ctx->normals.p) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "has_normals set but the scene has no normals");
    if (((uintptr_t)hits | (uintptr_t)normal_mat) & 15u) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "hits / normal_mat must be 16-byte aligned");
    DeviceScope dev_scope;
    int rc = dev_scope.enter(ctx);
    if (rc) return rc;
    if ((rc = check_range(ctx, org, n * 12, "org")) || (rc = check_range(ctx, dir, n * 12, "dir")) || (rc = check_range(ctx, hits, n * 16, "hits")) ||
        (rc = check_range(ctx, position, n * 12, "position")) || (rc = check_range(ctx, normal_mat, n * 16, "normal_mat"))) return rc;
    hipLaunchKernelGGL(k_surface_from_hits, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->sdev, n, org, dir, hits,
                       has_normals ? 1u : 0u, position, normal_mat);
    HIP_TRY(ctx, hipGetLastError());
    return RTAMD_OK;
}

int rtamd_surface_from_hits(rtamd_context *ctx, uint64_t n, const float *org, const float *dir, const float *hits, uint32_t has_normals,
                            float *position, float *normal_mat) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    if (n == 0) return RTAMD_OK;
    if (!org || !dir || !hits || !position || !normal_mat) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "surface_from_hits: NULL buffer");
    if (n > 0xFFFFFFFFull) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "at most 2^32-1 rays per call");
    DeviceScope dev_scope;
    int rc = dev_scope.enter(ctx);
    if (rc) return rc;
    DevBuf d_org, d_dir, d_hits, d_pos, d_nm;
    do {
        if ((rc = dev_upload(ctx, d_org, org, n * 12)) || (rc = dev_upload(ctx, d_dir, dir, n * 12)) || (rc = dev_upload(ctx, d_hits, hits, n * 16))) break;
        if ((rc = dev_alloc(ctx, d_pos, n * 12)) || (rc = dev_alloc(ctx, d_nm, n * 16))) break;
        if ((rc = rtamd_surface_from_hits_device(ctx, n, (const float *)d_org.p, (const float *)d_dir.p, (const float *)d_hits.p, has_normals,
                                                 (float *)d_pos.p, (float *)d_nm.p))) break;
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = hipMemcpy(position, d_pos.p, n * 12, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(normal_mat, d_nm.p, n * 16, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rc = fail(ctx, RTAMD_ERR_DEVICE, "surface_from_hits failed: %s", hipGetErrorString(e)); break; }
    } while (0);
    d_org.release(); d_dir.release(); d_hits.release(); d_pos.release(); d_nm.release();
    return rc;
}

int rtamd_generate_rays(rtamd_context *ctx, uint32_t width, uint32_t height, const float inv_proj_view[16],
                        float near_plane, float far_plane, float *dir_out) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    if (!inv_proj_view || !dir_out || width == 0 || height == 0) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "bad generate_rays arguments");
    DeviceScope dev_scope;
    int rc = dev_scope.enter(ctx);
    if (rc) return rc;
    DevBuf d;
    size_t bytes = (size_t)width * height * 12;
    if ((rc = dev_alloc(ctx, d, bytes))) return rc;
    rc = rtamd_generate_rays_device(ctx, width, height, inv_proj_view, near_plane, far_plane, (float *)d.p);
    if (!rc) {
        hipError_t e = hipStreamSynchronize(ctx->stream);   // the context's stream does not synchronise with the copy's null stream
        if (e == hipSuccess) e = hipMemcpy(dir_out, d.p, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(ctx, RTAMD_ERR_DEVICE, "generate_rays failed: %s", hipGetErrorString(e));
    }
    d.release();
    return rc;
}

uint32_t rtamd_tiles_per_rank(uint32_t width, uint32_t height, uint32_t tile_world) {
    if (tile_world == 0) return 0;
    return local_tiles(n_tiles_of(width, height, nullptr), 0, tile_world);
}

int rtamd_assemble_tile_batch_device(rtamd_context *ctx, const void *gathered, uint32_t width, uint32_t height,
                                     uint32_t tile_world, uint32_t elems_per_pixel, uint32_t n_frames, void *frames_out) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    if (!gathered || !frames_out || width == 0 || height == 0 || tile_world == 0 || elems_per_pixel == 0 || elems_per_pixel > 4 ||
        n_frames == 0 || n_frames > 65535u || (uint64_t)width * height > 0x7FFFFFFFull)
        return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "bad assemble_tiles arguments");
    DeviceScope dev_scope;
    int rc = dev_scope.enter(ctx);
    if (rc) return rc;
    uint32_t tiles_x = 0;
    uint32_t n_tiles = n_tiles_of(width, height, &tiles_x);
    uint32_t per_rank = local_tiles(n_tiles, 0, tile_world);
    uint32_t n = width * height;
    if ((rc = check_range(ctx, gathered, (uint64_t)tile_world * n_frames * per_rank * RTAMD_TILE_PIXELS * elems_per_pixel * 4, "gathered")) ||
        (rc = check_range(ctx, frames_out, (uint64_t)n_frames * n * elems_per_pixel * 4, "frames_out"))) return rc;
    if (elems_per_pixel == 1 && width % 4 == 0 && (((uintptr_t)gathered | (uintptr_t)frames_out) & 15u) == 0)      // RGBA8, the gather's usual format: four pixels per thread
        hipLaunchKernelGGL(k_assemble_tiles_px4, dim3((n / 4 + 255) / 256, n_frames), dim3(256), 0, ctx->stream, (const uint32_t *)gathered, width, height,
                           tile_world, tiles_x, per_rank, (uint32_t *)frames_out);
    else
        hipLaunchKernelGGL(k_assemble_tiles, dim3((n + 255) / 256, n_frames), dim3(256), 0, ctx->stream, (const uint32_t *)gathered, width, height,
                           tile_world, tiles_x, per_rank, elems_per_pixel, (uint32_t *)frames_out);
    HIP_TRY(ctx, hipGetLastError());
    return RTAMD_OK;
}

int rtamd_assemble_tiles_device(rtamd_context *ctx, const void *gathered, uint32_t width, uint32_t height,
                                uint32_t tile_world, uint32_t elems_per_pixel, void *frame_out) {
    return rtamd_assemble_tile_batch_device(ctx, gathered, width, height, tile_world, elems_per_pixel, 1, frame_out);
}

int rtamd_display_rgba8_device(rtamd_context *ctx, const float *rgba32f, uint64_t n_pixels, uint32_t *rgba8_out) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    if (n_pixels == 0) return RTAMD_OK;
    if (!rgba32f || !rgba8_out || n_pixels > 0x7FFFFFFFull * 256ull) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "bad display arguments");
    if ((uintptr_t)rgba32f & 15u) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "rgba32f must be 16-byte aligned");
    DeviceScope dev_scope;
    int rc = dev_scope.enter(ctx);
    if (rc) return rc;
    if ((rc = check_range(ctx, rgba32f, n_pixels * 16, "rgba32f")) || (rc = check_range(ctx, rgba8_out, n_pixels * 4, "rgba8_out"))) return rc;
    hipLaunchKernelGGL(k_display_rgba8, dim3((uint32_t)((n_pixels + 255) / 256)), dim3(256), 0, ctx->stream, (const float4 *)rgba32f, n_pixels, rgba8_out);
    HIP_TRY(ctx, hipGetLastError());
    return RTAMD_OK;
}

uint32_t rtamd_abi_version(void) { return RTAMD_ABI_VERSION; }

int rtamd_get_scene_info(const rtamd_context *ctx, rtamd_scene_info *info) {
    if (!ctx || !info) return RTAMD_ERR_INVALID_ARGUMENT;
    if (!ctx->has_scene) return RTAMD_ERR_NO_SCENE;
    *info = ctx->info;
    return RTAMD_OK;
}

#ifdef RTAMD_EXP_STAMPS
/* diagnostic build only (not declared in include/rtamd.h) */
int rtamd_exp_read_items(rtamd_context *ctx, unsigned long long *out, uint32_t items) {
    if (!ctx || !out || items > 65536) return RTAMD_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, (unsigned long long *)ctx->stamps.p + 1048576, (size_t)items * 32, hipMemcpyDeviceToHost));
    return RTAMD_OK;
}
int rtamd_exp_read_stamps(rtamd_context *ctx, unsigned long long *out, uint32_t waves) {
    if (!ctx || !out || waves > 65536) return RTAMD_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, ctx->stamps.p, (size_t)waves * 128, hipMemcpyDeviceToHost));
    return RTAMD_OK;
}
#endif

int rtamd_last_launch_coherence(rtamd_context *ctx, float *uniform_fraction, float *lane_utilisation, uint32_t *dyn_thresh_used) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    const rtamd_context::Slot *sl = nullptr;
    for (const auto &c : ctx->slots) if (c.bound && c.key == ctx->stream && c.ev_valid && c.stats_last) sl = &c;
    if (!sl) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "no persistent-kernel launch on the current stream yet");
    HIP_TRY(ctx, hipEventSynchronize(sl->ev_stop));
    uint32_t v[4] = {0, 0, 0, 0};
    HIP_TRY(ctx, hipMemcpy(v, sl->stats_last, sizeof v, hipMemcpyDeviceToHost));
    if (uniform_fraction) *uniform_fraction = v[1] ? (float)v[0] / (float)v[1] : 0.0f;
    if (lane_utilisation) *lane_utilisation = v[1] ? (float)v[2] / (64.0f * (float)v[1]) : 0.0f;
    if (dyn_thresh_used) *dyn_thresh_used = v[3] & 255u;
    return RTAMD_OK;
}

int rtamd_last_kernel_ms(rtamd_context *ctx, float *ms) {
    if (!ctx || !ms) return RTAMD_ERR_INVALID_ARGUMENT;
    // the current stream's last launch, else the most recent launch on any stream
    const rtamd_context::Slot *sl = nullptr;
    for (const auto &c : ctx->slots) if (c.bound && c.key == ctx->stream && c.ev_valid) sl = &c;
    if (!sl && ctx->last_slot >= 0 && ctx->slots[ctx->last_slot].ev_valid) sl = &ctx->slots[ctx->last_slot];
    if (!sl) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "no kernel has been launched yet");
    if (!sl->timed) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "the last launch was not timed (option launch_timing is 0)");
    HIP_TRY(ctx, hipEventSynchronize(sl->ev_stop));
    HIP_TRY(ctx, hipEventElapsedTime(ms, sl->ev_start, sl->ev_stop));
    return RTAMD_OK;
}

}  // extern "C"

#include "rtamd_chain.hip"
