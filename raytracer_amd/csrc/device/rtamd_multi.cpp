// rtamd_multi_* (include/rtamd.h): one process, several GPUs.  Host code only, written on top of the library's own
// single-GPU C entry points (one rtamd_context per GPU) plus HIP streams / events and RCCL:
//
//   per call (a batch of n frames), per GPU i of the list:
//     render stream i.b   wait "share buffer b is free"  ->  rtamd_render_batch_device(tile_rank = i, tile_world = N)
//                         ->  rtamd_display_rgba8_device over the whole batch (RGBA8 format)  ->  record "rendered"
//     comm stream i       wait "rendered"  ->  ncclGather(share -> root's `gathered`)  ->  record "share buffer b is free"
//   root:  comm stream 0  rtamd_assemble_tile_batch_device(gathered -> frames_out)
//
// Two sets of share buffers (b = call parity) and a render stream per set: the gather and assembly of one batch run
// beside the rendering of the next, exactly what bench.py does per rank over torch.distributed.  The collective is the
// only traffic between GPUs.  RCCL is loaded with dlopen when a handle asks for it, so that a process
// that never uses several GPUs does not load it.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../../include/rtamd.h"

namespace {

thread_local std::string g_multi_create_error;

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};
std::mutex g_rccl_mutex;
Rccl g_rccl;

// librccl.so.1 of the ROCm installation (or the copy a host such as PyTorch has already loaded under that name).
const char *load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return nullptr;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return "librccl.so.1 could not be loaded (dlopen)";
    Rccl r;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    r.Gather = (decltype(r.Gather))dlsym(h, "ncclGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    r.GetVersion = (decltype(r.GetVersion))dlsym(h, "ncclGetVersion");
    if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Gather || !r.GetErrorString) {
        dlclose(h);
        return "librccl.so.1 lacks ncclCommInitAll / ncclGather / ncclGroupStart / ncclGroupEnd";
    }
    r.handle = h;
    g_rccl = r;
    return nullptr;
}

struct Gpu {
    int device = -1;
    rtamd_context *ctx = nullptr;
    hipStream_t render[2] = {nullptr, nullptr};   // one per share-buffer set
    hipStream_t comm = nullptr;
    hipEvent_t rendered[2] = {nullptr, nullptr};  // set b's shares are rendered (and converted)
    hipEvent_t freed[2] = {nullptr, nullptr};     // set b's shares have been gathered: the set may be rendered into again
    bool freed_valid[2] = {false, false};
    void *share32[2] = {nullptr, nullptr};        // [frames][local pixels] RGBA32F
    void *share8[2] = {nullptr, nullptr};         // [frames][local pixels] RGBA8
    size_t share32_bytes[2] = {0, 0}, share8_bytes[2] = {0, 0};
    void *rays_dir = nullptr, *rays_org = nullptr;    // host rays of the current call, copied here ([frames][pixels][3])
    size_t rays_bytes = 0;
    ncclComm_t comm_handle = nullptr;
};

}  // namespace

struct rtamd_multi {
    std::vector<Gpu> gpus;
    uint32_t flags = 0;
    std::string err;
    uint64_t calls = 0;
    void *gathered = nullptr;        // root: [rank][frames][local pixels] of the gather format
    size_t gathered_bytes = 0;
    void *host_frames = nullptr;     // root: frames of rtamd_multi_render_batch before the copy to the host
    size_t host_frames_bytes = 0;
    hipEvent_t ev_share = nullptr, ev_gather = nullptr, ev_assembled = nullptr;   // root comm stream, last batch (timed)
    bool timed_valid = false;
};

namespace {

int mfail(rtamd_multi *m, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (m) m->err = buf; else g_multi_create_error = buf;
    return code;
}

#define MHIP(m, expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return mfail(m, e_ == hipErrorOutOfMemory ? RTAMD_ERR_OUT_OF_MEMORY : RTAMD_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define MNCCL(m, expr)                                                                                        \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess) return mfail(m, RTAMD_ERR_DEVICE, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)

// forwards a context's failure: the text comes from that context
int cfail(rtamd_multi *m, uint32_t i, int rc) {
    return mfail(m, rc, "GPU %u of the list (device %d): %s", i, m->gpus[i].device, rtamd_last_error(m->gpus[i].ctx));
}

struct CurrentDevice {
    int prev = -1;
    CurrentDevice() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~CurrentDevice() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int grow(rtamd_multi *m, void *&p, size_t &have, size_t need) {
    if (need <= have) return RTAMD_OK;
    if (p) MHIP(m, hipFree(p));
    p = nullptr; have = 0;
    MHIP(m, hipMalloc(&p, need));
    have = need;
    return RTAMD_OK;
}

}  // namespace

extern "C" {

int rtamd_multi_create(const int *devices, uint32_t n_devices, uint32_t flags, rtamd_multi **out) {
    if (!out) return mfail(nullptr, RTAMD_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!devices || n_devices == 0 || n_devices > 64) return mfail(nullptr, RTAMD_ERR_INVALID_ARGUMENT, "a device list holds 1..64 GPUs, not %u", n_devices);
    if (flags & ~RTAMD_MULTI_TRANSPORT_COPY) return mfail(nullptr, RTAMD_ERR_INVALID_ARGUMENT, "unknown flags 0x%x", flags);
    const bool copy = (flags & RTAMD_MULTI_TRANSPORT_COPY) != 0;
    if (!copy)
        for (uint32_t i = 0; i < n_devices; i++)
            for (uint32_t j = 0; j < i; j++)
                if (devices[i] == devices[j])
                    return mfail(nullptr, RTAMD_ERR_INVALID_ARGUMENT, "device %d is listed twice: RCCL takes one rank per GPU (RTAMD_MULTI_TRANSPORT_COPY allows it)", devices[i]);
    rtamd_multi *m = new (std::nothrow) rtamd_multi();
    if (!m) return mfail(nullptr, RTAMD_ERR_OUT_OF_MEMORY, "out of host memory");
    m->flags = flags;
    auto bail = [&](int code) { g_multi_create_error = m->err; rtamd_multi_destroy(m); return code; };
    try { m->gpus.resize(n_devices); } catch (const std::bad_alloc &) { m->err = "out of host memory"; return bail(RTAMD_ERR_OUT_OF_MEMORY); }
    CurrentDevice restore;
    for (uint32_t i = 0; i < n_devices; i++) {
        Gpu &g = m->gpus[i];
        int rc = rtamd_context_create(devices[i], &g.ctx);
        if (rc) { m->err = std::string("rtamd_context_create: ") + rtamd_last_error(nullptr); return bail(rc); }
        g.device = devices[i];      // only a GPU that has a context is touched again (destroy)
        hipError_t e = hipSetDevice(devices[i]);
        for (int b = 0; b < 2 && e == hipSuccess; b++) {
            e = hipStreamCreateWithFlags(&g.render[b], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&g.rendered[b], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&g.freed[b], hipEventDisableTiming);
        }
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&g.comm, hipStreamNonBlocking);
        if (e != hipSuccess) { m->err = std::string("stream / event creation failed: ") + hipGetErrorString(e); return bail(RTAMD_ERR_DEVICE); }
    }
    {
        hipError_t e = hipSetDevice(devices[0]);
        if (e == hipSuccess) e = hipEventCreate(&m->ev_share);
        if (e == hipSuccess) e = hipEventCreate(&m->ev_gather);
        if (e == hipSuccess) e = hipEventCreate(&m->ev_assembled);
        if (e != hipSuccess) { m->err = std::string("event creation failed: ") + hipGetErrorString(e); return bail(RTAMD_ERR_DEVICE); }
    }
    if (copy) {
        // the root reads the peers' share buffers: peer access where the GPUs differ (an error here only means the copy is staged)
        for (uint32_t i = 1; i < n_devices; i++)
            if (devices[i] != devices[0]) { (void)hipDeviceEnablePeerAccess(devices[i], 0); (void)hipGetLastError(); }
    } else {
        if (const char *why = load_rccl()) { m->err = why; return bail(RTAMD_ERR_DEVICE); }
        std::vector<ncclComm_t> comms(n_devices, nullptr);
        ncclResult_t r = g_rccl.CommInitAll(comms.data(), (int)n_devices, devices);
        if (r != ncclSuccess) { m->err = std::string("ncclCommInitAll failed: ") + g_rccl.GetErrorString(r); return bail(RTAMD_ERR_DEVICE); }
        for (uint32_t i = 0; i < n_devices; i++) m->gpus[i].comm_handle = comms[i];
    }
    *out = m;
    return RTAMD_OK;
}

void rtamd_multi_destroy(rtamd_multi *m) {
    if (!m) return;
    CurrentDevice restore;
    for (Gpu &g : m->gpus) {
        if (g.device < 0) continue;
        if (hipSetDevice(g.device) != hipSuccess) continue;
        (void)hipDeviceSynchronize();
    }
    for (Gpu &g : m->gpus) {
        if (g.device < 0 || hipSetDevice(g.device) != hipSuccess) continue;
        if (g.comm_handle && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g.comm_handle);
        for (int b = 0; b < 2; b++) {
            if (g.share32[b]) (void)hipFree(g.share32[b]);
            if (g.share8[b]) (void)hipFree(g.share8[b]);
            if (g.rendered[b]) (void)hipEventDestroy(g.rendered[b]);
            if (g.freed[b]) (void)hipEventDestroy(g.freed[b]);
        }
        if (g.rays_dir) (void)hipFree(g.rays_dir);
        if (g.rays_org) (void)hipFree(g.rays_org);
        if (g.ctx) rtamd_context_destroy(g.ctx);     // synchronises the device once more; the streams below are idle by then
        for (int b = 0; b < 2; b++) if (g.render[b]) (void)hipStreamDestroy(g.render[b]);
        if (g.comm) (void)hipStreamDestroy(g.comm);
    }
    if (!m->gpus.empty() && m->gpus[0].device >= 0 && hipSetDevice(m->gpus[0].device) == hipSuccess) {
        if (m->gathered) (void)hipFree(m->gathered);
        if (m->host_frames) (void)hipFree(m->host_frames);
        if (m->ev_share) (void)hipEventDestroy(m->ev_share);
        if (m->ev_gather) (void)hipEventDestroy(m->ev_gather);
        if (m->ev_assembled) (void)hipEventDestroy(m->ev_assembled);
    }
    (void)hipGetLastError();      // tear-down leaves no sticky error behind for the caller's next HIP call
    delete m;
}

const char *rtamd_multi_last_error(const rtamd_multi *m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }

uint32_t rtamd_multi_device_count(const rtamd_multi *m) { return m ? (uint32_t)m->gpus.size() : 0u; }

rtamd_context *rtamd_multi_context(rtamd_multi *m, uint32_t i) { return (m && i < m->gpus.size()) ? m->gpus[i].ctx : nullptr; }

void *rtamd_multi_root_stream(rtamd_multi *m) { return m ? (void *)m->gpus[0].comm : nullptr; }

int rtamd_multi_upload_scene(rtamd_multi *m, const rtamd_scene_desc *scene) {
    if (!m) return RTAMD_ERR_INVALID_ARGUMENT;
    for (uint32_t i = 0; i < m->gpus.size(); i++)
        if (int rc = rtamd_upload_scene(m->gpus[i].ctx, scene)) return cfail(m, i, rc);
    return RTAMD_OK;
}

int rtamd_multi_upload_texture(rtamd_multi *m, uint32_t slot, const uint8_t *rgb8, uint32_t width, uint32_t height) {
    if (!m) return RTAMD_ERR_INVALID_ARGUMENT;
    for (uint32_t i = 0; i < m->gpus.size(); i++)
        if (int rc = rtamd_upload_texture(m->gpus[i].ctx, slot, rgb8, width, height)) return cfail(m, i, rc);
    return RTAMD_OK;
}

int rtamd_multi_upload_materials(rtamd_multi *m, const rtamd_shading_material *materials, uint32_t count) {
    if (!m) return RTAMD_ERR_INVALID_ARGUMENT;
    for (uint32_t i = 0; i < m->gpus.size(); i++)
        if (int rc = rtamd_upload_materials(m->gpus[i].ctx, materials, count)) return cfail(m, i, rc);
    return RTAMD_OK;
}

int rtamd_multi_set_option(rtamd_multi *m, const char *name, int64_t value) {
    if (!m) return RTAMD_ERR_INVALID_ARGUMENT;
    for (uint32_t i = 0; i < m->gpus.size(); i++)
        if (int rc = rtamd_set_option(m->gpus[i].ctx, name, value)) return cfail(m, i, rc);
    return RTAMD_OK;
}

int rtamd_multi_synchronize(rtamd_multi *m) {
    if (!m) return RTAMD_ERR_INVALID_ARGUMENT;
    CurrentDevice restore;
    for (Gpu &g : m->gpus) {
        MHIP(m, hipSetDevice(g.device));
        for (int b = 0; b < 2; b++) MHIP(m, hipStreamSynchronize(g.render[b]));
        MHIP(m, hipStreamSynchronize(g.comm));
    }
    return RTAMD_OK;
}

int rtamd_multi_render_batch_device(rtamd_multi *m, uint32_t n_frames, const rtamd_frame_desc *descs, int gather_format, void *frames_out) {
    if (!m) return RTAMD_ERR_INVALID_ARGUMENT;
    if (n_frames == 0 || n_frames > RTAMD_MAX_BATCH_FRAMES) return mfail(m, RTAMD_ERR_INVALID_ARGUMENT, "a batch holds 1..%d frames, not %u", RTAMD_MAX_BATCH_FRAMES, n_frames);
    if (!descs || !frames_out) return mfail(m, RTAMD_ERR_INVALID_ARGUMENT, "descs / frames_out is NULL");
    if (gather_format != RTAMD_GATHER_RGBA8 && gather_format != RTAMD_GATHER_RGBA32F) return mfail(m, RTAMD_ERR_INVALID_ARGUMENT, "unknown gather format %d", gather_format);
    const uint32_t W = descs[0].width, H = descs[0].height;
    if (W == 0 || H == 0 || (uint64_t)W * H > 0x7FFFFFFFull) return mfail(m, RTAMD_ERR_INVALID_ARGUMENT, "bad frame size %ux%u", W, H);
    for (uint32_t f = 0; f < n_frames; f++) {
        if (descs[f].width != W || descs[f].height != H) return mfail(m, RTAMD_ERR_INVALID_ARGUMENT, "frame %u: the frames of a batch share their size", f);
        if ((descs[f].dir == nullptr) != (descs[f].org == nullptr)) return mfail(m, RTAMD_ERR_INVALID_ARGUMENT, "frame %u: dir and org must both be given or both be NULL", f);
        if ((descs[f].dir == nullptr) != (descs[0].dir == nullptr)) return mfail(m, RTAMD_ERR_INVALID_ARGUMENT, "frame %u: either every frame of a batch brings host rays or none does", f);
    }
    const uint32_t N = (uint32_t)m->gpus.size();
    const bool as8 = gather_format == RTAMD_GATHER_RGBA8, copy = (m->flags & RTAMD_MULTI_TRANSPORT_COPY) != 0;
    const uint64_t n_px = (uint64_t)W * H;
    const uint64_t local_px = N == 1 ? n_px : (uint64_t)rtamd_tiles_per_rank(W, H, N) * 1024u;    // per frame, the same for every rank (padded)
    const size_t px_bytes = as8 ? 4 : 16;
    const int b = (int)(m->calls & 1u);
    const bool host_rays = descs[0].dir != nullptr;
    CurrentDevice restore;
    int rc;

    // ---- every GPU renders (and converts) its tiles of the batch
    for (uint32_t i = 0; i < N; i++) {
        Gpu &g = m->gpus[i];
        MHIP(m, hipSetDevice(g.device));
        if ((rc = grow(m, g.share32[b], g.share32_bytes[b], (size_t)n_frames * local_px * 16))) return rc;
        if (as8 && (rc = grow(m, g.share8[b], g.share8_bytes[b], (size_t)n_frames * local_px * 4))) return rc;
        if (host_rays) {
            // both sets' renders read these buffers: the copy waits for what is in flight on this GPU
            MHIP(m, hipStreamSynchronize(g.render[0]));
            MHIP(m, hipStreamSynchronize(g.render[1]));
            const size_t need = (size_t)n_frames * n_px * 12;
            if (need > g.rays_bytes) {
                if (g.rays_dir) MHIP(m, hipFree(g.rays_dir));
                if (g.rays_org) MHIP(m, hipFree(g.rays_org));
                g.rays_dir = g.rays_org = nullptr; g.rays_bytes = 0;
                MHIP(m, hipMalloc(&g.rays_dir, need));
                MHIP(m, hipMalloc(&g.rays_org, need));
                g.rays_bytes = need;
            }
            for (uint32_t f = 0; f < n_frames; f++) {
                MHIP(m, hipMemcpy((char *)g.rays_dir + (size_t)f * n_px * 12, descs[f].dir, n_px * 12, hipMemcpyHostToDevice));
                MHIP(m, hipMemcpy((char *)g.rays_org + (size_t)f * n_px * 12, descs[f].org, n_px * 12, hipMemcpyHostToDevice));
            }
        }
        hipStream_t rs = g.render[b];
        if (g.freed_valid[b]) MHIP(m, hipStreamWaitEvent(rs, g.freed[b], 0));    // the set's previous shares have left
        rtamd_frame_desc d[RTAMD_MAX_BATCH_FRAMES];
        rtamd_frame_out o[RTAMD_MAX_BATCH_FRAMES];
        for (uint32_t f = 0; f < n_frames; f++) {
            d[f] = descs[f];
            d[f].tile_rank = i; d[f].tile_world = N;
            if (host_rays) {
                d[f].dir = (const float *)((const char *)g.rays_dir + (size_t)f * n_px * 12);
                d[f].org = (const float *)((const char *)g.rays_org + (size_t)f * n_px * 12);
            }
            std::memset(&o[f], 0, sizeof o[f]);
            o[f].rgba32f = (float *)((char *)g.share32[b] + (size_t)f * local_px * 16);
        }
        if ((rc = rtamd_context_set_stream(g.ctx, rs))) return cfail(m, i, rc);
        // (entries of edge tiles outside the frame, and of the padding tiles of ranks that own one tile fewer, are never written
        // by the kernel and never read by the assembly: they travel as they are)
        if ((rc = rtamd_render_batch_device(g.ctx, n_frames, d, o))) return cfail(m, i, rc);
        if (as8 && (rc = rtamd_display_rgba8_device(g.ctx, (const float *)g.share32[b], (uint64_t)n_frames * local_px, (uint32_t *)g.share8[b]))) return cfail(m, i, rc);
        MHIP(m, hipEventRecord(g.rendered[b], rs));
        MHIP(m, hipStreamWaitEvent(g.comm, g.rendered[b], 0));
    }

    // ---- the one collective: every GPU's share to the root
    Gpu &root = m->gpus[0];
    MHIP(m, hipSetDevice(root.device));
    const size_t block_bytes = (size_t)n_frames * local_px * px_bytes;      // one rank's share of the batch
    if ((rc = grow(m, m->gathered, m->gathered_bytes, block_bytes * N))) return rc;    // re-allocation waits for the device (hipFree)
    MHIP(m, hipEventRecord(m->ev_share, root.comm));
    if (copy) {
        for (uint32_t i = 0; i < N; i++) {
            Gpu &g = m->gpus[i];
            if (i) MHIP(m, hipStreamWaitEvent(root.comm, g.rendered[b], 0));
            MHIP(m, hipMemcpyAsync((char *)m->gathered + (size_t)i * block_bytes, as8 ? g.share8[b] : g.share32[b], block_bytes, hipMemcpyDeviceToDevice, root.comm));
        }
        MHIP(m, hipEventRecord(root.freed[b], root.comm));      // all shares of the set have been read, by the root's stream
        for (uint32_t i = 0; i < N; i++) {
            Gpu &g = m->gpus[i];
            if (i) {      // the peer's next render into the set waits for the root's copies
                MHIP(m, hipSetDevice(g.device));
                MHIP(m, hipStreamWaitEvent(g.comm, root.freed[b], 0));
                MHIP(m, hipEventRecord(g.freed[b], g.comm));
            }
            g.freed_valid[b] = true;
        }
        MHIP(m, hipSetDevice(root.device));
    } else {
        const size_t count = (size_t)n_frames * local_px * (as8 ? 1 : 4);
        const ncclDataType_t dt = as8 ? ncclInt32 : ncclFloat;
        MNCCL(m, g_rccl.GroupStart());
        for (uint32_t i = 0; i < N; i++) {
            Gpu &g = m->gpus[i];
            ncclResult_t r = g_rccl.Gather(as8 ? g.share8[b] : g.share32[b], i == 0 ? m->gathered : nullptr, count, dt, 0, g.comm_handle, g.comm);
            if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return mfail(m, RTAMD_ERR_DEVICE, "ncclGather failed on GPU %u of the list: %s", i, g_rccl.GetErrorString(r)); }
        }
        MNCCL(m, g_rccl.GroupEnd());
        for (uint32_t i = 0; i < N; i++) {
            Gpu &g = m->gpus[i];
            MHIP(m, hipSetDevice(g.device));
            MHIP(m, hipEventRecord(g.freed[b], g.comm));
            g.freed_valid[b] = true;
        }
        MHIP(m, hipSetDevice(root.device));
    }
    MHIP(m, hipEventRecord(m->ev_gather, root.comm));

    // ---- the root assembles the scan-line frames behind the gather
    if ((rc = rtamd_context_set_stream(root.ctx, root.comm))) return cfail(m, 0, rc);
    if (N == 1) {
        MHIP(m, hipMemcpyAsync(frames_out, m->gathered, block_bytes, hipMemcpyDeviceToDevice, root.comm));    // one GPU: the share is the frame
    } else if ((rc = rtamd_assemble_tile_batch_device(root.ctx, m->gathered, W, H, N, as8 ? 1 : 4, n_frames, frames_out))) {
        return cfail(m, 0, rc);
    }
    MHIP(m, hipEventRecord(m->ev_assembled, root.comm));
    m->timed_valid = true;
    m->calls++;
    return RTAMD_OK;
}

int rtamd_multi_render_batch(rtamd_multi *m, uint32_t n_frames, const rtamd_frame_desc *descs, int gather_format, void *frames_out) {
    if (!m) return RTAMD_ERR_INVALID_ARGUMENT;
    if (!descs || !frames_out || n_frames == 0 || n_frames > RTAMD_MAX_BATCH_FRAMES) return mfail(m, RTAMD_ERR_INVALID_ARGUMENT, "bad batch arguments");
    if (gather_format != RTAMD_GATHER_RGBA8 && gather_format != RTAMD_GATHER_RGBA32F) return mfail(m, RTAMD_ERR_INVALID_ARGUMENT, "unknown gather format %d", gather_format);
    const size_t bytes = (size_t)n_frames * descs[0].width * descs[0].height * (gather_format == RTAMD_GATHER_RGBA8 ? 4 : 16);
    if (bytes == 0) return mfail(m, RTAMD_ERR_INVALID_ARGUMENT, "empty frame");
    CurrentDevice restore;
    MHIP(m, hipSetDevice(m->gpus[0].device));
    int rc;
    if (bytes > m->host_frames_bytes) {
        if ((rc = rtamd_multi_synchronize(m))) return rc;     // an earlier batch may still be assembling into the old buffer
        MHIP(m, hipSetDevice(m->gpus[0].device));
        if ((rc = grow(m, m->host_frames, m->host_frames_bytes, bytes))) return rc;
    }
    if ((rc = rtamd_multi_render_batch_device(m, n_frames, descs, gather_format, m->host_frames))) return rc;
    MHIP(m, hipSetDevice(m->gpus[0].device));
    MHIP(m, hipStreamSynchronize(m->gpus[0].comm));
    MHIP(m, hipMemcpy(frames_out, m->host_frames, bytes, hipMemcpyDeviceToHost));
    return RTAMD_OK;
}

int rtamd_multi_last_gather_ms(rtamd_multi *m, float *gather_ms, float *assemble_ms) {
    if (!m) return RTAMD_ERR_INVALID_ARGUMENT;
    if (!m->timed_valid) return mfail(m, RTAMD_ERR_INVALID_ARGUMENT, "no batch has been rendered yet");
    CurrentDevice restore;
    MHIP(m, hipSetDevice(m->gpus[0].device));
    MHIP(m, hipEventSynchronize(m->ev_assembled));
    if (gather_ms) MHIP(m, hipEventElapsedTime(gather_ms, m->ev_share, m->ev_gather));
    if (assemble_ms) MHIP(m, hipEventElapsedTime(assemble_ms, m->ev_gather, m->ev_assembled));
    return RTAMD_OK;
}

}  // extern "C"
