// rtamd_multibounce_* (include/rtamd.h): the multi-sample, multi-bounce chain of BASELINE config 5 as a C entry point.
// The reference's frame loop (src/main.rs:143-203) is ray_create -> ray_dispatcher -> 3 x ray_trace for ONE bounce; here the
// same passes repeat per bounce, with rtamd_surface_from_hits_device standing in for the rasteriser's G-buffer.  This file
// is launches and buffer rotation only, written on top of the library's own public entry points; the two kernels below fill a
// constant origin array and add a host-known number to a device counter.
// Included at the end of rtamd_api.hip (one translation unit: the context's stream and error text are its internals).

namespace {

__global__ void k_fill_vec3(float *__restrict__ out, uint64_t n, float x, float y, float z) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[3 * i] = x; out[3 * i + 1] = y; out[3 * i + 2] = z;
}

__global__ void k_add_u64(unsigned long long *counter, unsigned long long v) { atomicAdd(counter, v); }

// counter += weight * (entries of a G-buffer that hold a surface: normal_mat.w != 1e30).  A grid-stride pass with one atomic per
// workgroup (1024 of them): an atomic per wave from inside the dispatcher serialised 130 000 atomics on one address per 4K frame.
__global__ void __launch_bounds__(256) k_count_surfaces(const float4 *__restrict__ normal_mat, uint64_t n, uint32_t weight, unsigned long long *counter) {
    __shared__ uint32_t part[4];
    uint32_t mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u)
        mine += normal_mat[i].w != 1e30f ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long total = (unsigned long long)part[0] + part[1] + part[2] + part[3];
        if (total) atomicAdd(counter, total * weight);
    }
}

constexpr double R2_G = 0.7548776662466927;   // plastic-number constant of the R2 low-discrepancy sequence

struct Shard { uint32_t rank, world; uint64_t entries, primaries; };

bool shard_of(const rtamd_multibounce_desc *d, Shard &s) {
    if (!d || d->width == 0 || d->height == 0 || (uint64_t)d->width * d->height > 0x7FFFFFFFull) return false;
    s.world = d->tile_world ? d->tile_world : 1u;
    s.rank = d->tile_rank;
    if (s.rank >= s.world) return false;
    const uint64_t n_px = (uint64_t)d->width * d->height;
    if (s.world == 1) { s.entries = s.primaries = n_px; return true; }
    const uint32_t tiles_x = (d->width + RTAMD_TILE_DIM - 1) / RTAMD_TILE_DIM, tiles_y = (d->height + RTAMD_TILE_DIM - 1) / RTAMD_TILE_DIM;
    uint64_t local = 0, inside = 0;
    for (uint32_t g = s.rank; g < tiles_x * tiles_y; g += s.world) {
        const uint32_t tx = g % tiles_x, ty = g / tiles_x;
        const uint32_t tw = d->width - tx * RTAMD_TILE_DIM < RTAMD_TILE_DIM ? d->width - tx * RTAMD_TILE_DIM : RTAMD_TILE_DIM;
        const uint32_t th = d->height - ty * RTAMD_TILE_DIM < RTAMD_TILE_DIM ? d->height - ty * RTAMD_TILE_DIM : RTAMD_TILE_DIM;
        local++; inside += (uint64_t)tw * th;
    }
    s.entries = local * RTAMD_TILE_DIM * RTAMD_TILE_DIM;
    s.primaries = inside;
    return true;
}

bool lane_complete(const rtamd_multibounce_lane &l) {
    return l.cam_org && l.dir && l.hits && l.position && l.normal_mat && l.org[0] && l.org[1] && l.shadow_dir[0] && l.shadow_dir[1] &&
           l.reflect_dir[0] && l.reflect_dir[1] && l.ambient_dir[0] && l.ambient_dir[1] && l.shadow_hits;
}

}  // namespace

extern "C" {

uint64_t rtamd_multibounce_entries(const rtamd_multibounce_desc *desc) {
    Shard s;
    return shard_of(desc, s) ? s.entries : 0;
}

uint64_t rtamd_multibounce_primary_rays(const rtamd_multibounce_desc *desc) {
    Shard s;
    return shard_of(desc, s) ? s.primaries : 0;
}

int rtamd_multibounce_prepare_lane_device(rtamd_context *ctx, const rtamd_multibounce_desc *desc, const rtamd_multibounce_lane *lane) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    Shard s;
    if (!shard_of(desc, s) || !lane || !lane->cam_org) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "multibounce: bad descriptor or lane");
    if (s.entries == 0) return RTAMD_OK;
    DeviceScope dev_scope;
    int rc = dev_scope.enter(ctx);
    if (rc) return rc;
    if ((rc = check_range(ctx, lane->cam_org, s.entries * 12, "cam_org"))) return rc;
    hipLaunchKernelGGL(k_fill_vec3, dim3((uint32_t)((s.entries + 255) / 256)), dim3(256), 0, ctx->stream, lane->cam_org, s.entries,
                       desc->cam_pos[0], desc->cam_pos[1], desc->cam_pos[2]);
    HIP_TRY(ctx, hipGetLastError());
    return RTAMD_OK;
}

int rtamd_multibounce_sample_batch_device(rtamd_context *ctx, const rtamd_multibounce_desc *desc, const uint32_t *samples, uint32_t n_samples,
                                          const rtamd_multibounce_lane *lanes, uint64_t *rays_traced) {
    if (!ctx) return RTAMD_ERR_INVALID_ARGUMENT;
    Shard sh;
    if (!shard_of(desc, sh)) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "multibounce: bad frame size or tile shard");
    if (!samples || !lanes || n_samples == 0 || n_samples > 8) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "multibounce: 1..8 samples, one lane each");
    if (desc->max_depth == 0 || desc->max_depth > 16) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "multibounce: max_depth must be 1..16");
    if (desc->noise_size == 0) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "multibounce: noise_size is 0");
    for (uint32_t k = 0; k < n_samples; k++) {
        if (!lane_complete(lanes[k])) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "multibounce: lane %u has a NULL buffer", k);
        for (uint32_t j = 0; j < k; j++)
            if (lanes[j].hits == lanes[k].hits) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "multibounce: lanes %u and %u share buffers", j, k);
    }
    const uint64_t n = sh.entries;
    if (n == 0) return RTAMD_OK;
    const uint32_t W = desc->width, H = desc->height, D = desc->max_depth, ns = desc->noise_size;
    int rc;

    // primaries of every sample, then ONE launch for all their traversals
    rtamd_trace_job jobs[16];
    for (uint32_t k = 0; k < n_samples; k++) {
        const double s = (double)samples[k];
        const float ox = (float)std::fmod(0.5 + s * R2_G, 1.0), oy = (float)std::fmod(0.5 + s * R2_G * R2_G, 1.0);
        if ((rc = rtamd_generate_rays_tiles_device(ctx, W, H, desc->inv_proj_view, desc->near_plane, desc->far_plane, ox, oy, sh.rank, sh.world, lanes[k].dir))) return rc;
        jobs[k].n = n; jobs[k].org = lanes[k].cam_org; jobs[k].dir = lanes[k].dir; jobs[k].hits = lanes[k].hits;
    }
    if ((rc = rtamd_trace_image_batch_device(ctx, n_samples, jobs, W, sh.world))) return rc;
    if (rays_traced) {
        if ((uintptr_t)rays_traced & 7u) return fail(ctx, RTAMD_ERR_INVALID_ARGUMENT, "multibounce: rays_traced must be 8-byte aligned");
        DeviceScope dev_scope;
        if ((rc = dev_scope.enter(ctx))) return rc;
        if ((rc = check_range(ctx, rays_traced, 8, "rays_traced"))) return rc;
        hipLaunchKernelGGL(k_add_u64, dim3(1), dim3(1), 0, ctx->stream, (unsigned long long *)rays_traced, (unsigned long long)(sh.primaries * n_samples));
        HIP_TRY(ctx, hipGetLastError());
    }

    const float *cur_org[8], *cur_dir[8];
    for (uint32_t k = 0; k < n_samples; k++) { cur_org[k] = lanes[k].cam_org; cur_dir[k] = lanes[k].dir; }
    for (uint32_t depth = 1; depth <= D; depth++) {
        const uint32_t t = depth & 1u;
        const bool last = depth == D;
        uint32_t n_jobs = 0;
        for (uint32_t k = 0; k < n_samples; k++) {
            const rtamd_multibounce_lane &ln = lanes[k];
            if ((rc = rtamd_surface_from_hits_device(ctx, n, cur_org[k], cur_dir[k], ln.hits, desc->has_normals, ln.position, ln.normal_mat))) return rc;
            rtamd_dispatch_desc dd;
            std::memset(&dd, 0, sizeof dd);
            dd.width = W; dd.height = H;
            dd.position = ln.position; dd.normal_mat = ln.normal_mat;
            std::memcpy(dd.light_pos, desc->light_pos, sizeof dd.light_pos);
            std::memcpy(dd.camera_pos, desc->cam_pos, sizeof dd.camera_pos);
            const uint64_t s = samples[k];
            dd.noise_offset_scale[0] = (float)((double)((7 * s + 3 * depth) % ns) / (double)ns);
            dd.noise_offset_scale[1] = (float)((double)((11 * s + 5 * depth) % ns) / (double)ns);
            dd.noise_offset_scale[2] = (float)((double)W / (double)ns);
            dd.noise_offset_scale[3] = (float)((double)H / (double)ns);
            dd.tile_rank = sh.rank; dd.tile_world = sh.world;
            rtamd_dispatch_out od;
            od.org = ln.org[t]; od.shadow_dir = ln.shadow_dir[t]; od.reflect_dir = ln.reflect_dir[t]; od.ambient_dir = ln.ambient_dir[t];
            if ((rc = rtamd_dispatch_rays_device(ctx, &dd, &od))) return rc;
            if (rays_traced) {      // a shadow ray per surface, and a continuation unless this is the last bounce
                DeviceScope dev_scope;
                if ((rc = dev_scope.enter(ctx))) return rc;
                const uint32_t blocks = (uint32_t)(n + 255) / 256 < 1024u ? (uint32_t)((n + 255) / 256) : 1024u;
                hipLaunchKernelGGL(k_count_surfaces, dim3(blocks), dim3(256), 0, ctx->stream, (const float4 *)ln.normal_mat, n, last ? 1u : 2u,
                                   (unsigned long long *)rays_traced);
                HIP_TRY(ctx, hipGetLastError());
            }
            jobs[n_jobs].n = n; jobs[n_jobs].org = ln.org[t]; jobs[n_jobs].dir = ln.shadow_dir[t]; jobs[n_jobs].hits = ln.shadow_hits;
            n_jobs++;
        }
        if (!last)
            for (uint32_t k = 0; k < n_samples; k++) {
                const rtamd_multibounce_lane &ln = lanes[k];
                const float *cont = (samples[k] % 2u == 0u) ? ln.reflect_dir[t] : ln.ambient_dir[t];
                // hits is read by nothing else from here on: the continuation's records replace the previous bounce's
                jobs[n_jobs].n = n; jobs[n_jobs].org = ln.org[t]; jobs[n_jobs].dir = cont; jobs[n_jobs].hits = ln.hits;
                n_jobs++;
                cur_org[k] = ln.org[t]; cur_dir[k] = cont;
            }
        if ((rc = rtamd_trace_image_batch_device(ctx, n_jobs, jobs, W, sh.world))) return rc;
    }
    return RTAMD_OK;
}

}  // extern "C"
