// Host side of the device library under AddressSanitizer + UBSan (GPU box; device code is NOT
// instrumented): rtamd_api.hip is compiled from source with `-Xarch_host -fsanitize=address,undefined`
// and driven through the C-ABI with good and malformed scenes, all entry points, several streams.
// Build + run: tools/sanitize/run_device.sh
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/rtamd.h"

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed line %d: %s (%s)\n", __LINE__, #c, rtamd_last_error(ctx)); failures++; } } while (0)

struct Node { float b[6]; uint32_t leaf, a, c; };
struct Tri { uint32_t p0, p1, p2, mat; };

int main() {
    rtamd_context *ctx = nullptr;
    if (rtamd_context_create(0, &ctx) != RTAMD_OK) { std::fprintf(stderr, "no context: %s\n", rtamd_last_error(nullptr)); return 2; }
    // chain scene of `depth` inner nodes (tests/test_gpu_api.py::_chain_scene)
    auto chain = [&](uint32_t depth, std::vector<Node> &nodes, std::vector<Tri> &tris, std::vector<float> &pos) {
        nodes.assign(2 * depth + 1, Node{}); tris.resize(depth + 1); pos.resize(9 * (depth + 1));
        for (uint32_t z = 0; z <= depth; z++) {
            const float p[9] = {-3, -3, (float)z, 3, -3, (float)z, 0, 3, (float)z};
            std::memcpy(&pos[9 * z], p, sizeof p);
            tris[z] = Tri{3 * z, 3 * z + 1, 3 * z + 2, z & 3u};
        }
        for (uint32_t k = 0; k < depth; k++) {
            nodes[2 * k] = Node{{-3, -3, (float)k, 3, 3, (float)depth}, 0, 2 * k + 1, 2 * k + 2};
            nodes[2 * k + 1] = Node{{-3, -3, (float)k, 3, 3, (float)k}, 1, k, 1};
        }
        nodes[2 * depth] = Node{{-3, -3, (float)depth, 3, 3, (float)depth}, 1, depth, 1};
    };
    std::vector<Node> nodes; std::vector<Tri> tris; std::vector<float> pos;
    chain(40, nodes, tris, pos);
    std::vector<float> nrm(pos.size(), 0.0f), uv(pos.size() / 3 * 2, 0.25f);
    for (size_t i = 0; i < nrm.size() / 3; i++) nrm[3 * i + 2] = 1.0f;
    rtamd_scene_desc sc{};
    sc.nodes = nodes.data(); sc.node_count = (uint32_t)nodes.size();
    sc.triangles = tris.data(); sc.triangle_count = (uint32_t)tris.size();
    sc.positions = pos.data(); sc.vertex_count = (uint32_t)(pos.size() / 3);
    sc.tex_coords = uv.data(); sc.normals = nrm.data();
    // malformed variants first: none may read or write out of bounds
    { auto bad = nodes; bad[0].a = 100000; rtamd_scene_desc b = sc; b.nodes = bad.data(); CHECK(rtamd_upload_scene(ctx, &b) == RTAMD_ERR_MALFORMED_SCENE); }
    { auto bad = nodes; bad[2].c = 0; rtamd_scene_desc b = sc; b.nodes = bad.data(); CHECK(rtamd_upload_scene(ctx, &b) == RTAMD_ERR_MALFORMED_SCENE); }
    { auto bad = nodes; bad[1].a = 39; bad[1].c = 5; rtamd_scene_desc b = sc; b.nodes = bad.data(); CHECK(rtamd_upload_scene(ctx, &b) == RTAMD_ERR_MALFORMED_SCENE); }
    { auto bad = tris; bad[3].p2 = 1u << 30; rtamd_scene_desc b = sc; b.triangles = bad.data(); CHECK(rtamd_upload_scene(ctx, &b) == RTAMD_ERR_MALFORMED_SCENE); }
    { rtamd_scene_desc b = sc; b.node_count = 0; CHECK(rtamd_upload_scene(ctx, &b) == RTAMD_ERR_EMPTY_SCENE); }
    { std::vector<Node> n2; std::vector<Tri> t2; std::vector<float> p2; chain(129, n2, t2, p2); rtamd_scene_desc b{}; b.nodes = n2.data(); b.node_count = (uint32_t)n2.size();
      b.triangles = t2.data(); b.triangle_count = (uint32_t)t2.size(); b.positions = p2.data(); b.vertex_count = (uint32_t)(p2.size() / 3); CHECK(rtamd_upload_scene(ctx, &b) == RTAMD_ERR_MALFORMED_SCENE); }
    CHECK(rtamd_upload_scene(ctx, &sc) == RTAMD_OK);
    rtamd_scene_info info{}; CHECK(rtamd_get_scene_info(ctx, &info) == RTAMD_OK && info.max_depth == 40);
    // textures, materials
    std::vector<uint8_t> tex(7 * 5 * 3, 200);
    for (uint32_t s = 0; s < 4; s++) CHECK(rtamd_upload_texture(ctx, s, tex.data(), 7, 5) == RTAMD_OK);
    CHECK(rtamd_upload_texture(ctx, 16, tex.data(), 7, 5) != RTAMD_OK);
    CHECK(rtamd_upload_noise_texture(ctx, tex.data(), 5, 7) == RTAMD_OK);
    rtamd_shading_material mats[4];
    for (int i = 0; i < 4; i++) mats[i] = rtamd_shading_material{{.5f, .5f, .5f}, {.8f, .8f, .8f}, {.4f, .4f, .4f}, 30.f, i - 1};
    CHECK(rtamd_upload_materials(ctx, mats, 3) != RTAMD_OK);
    CHECK(rtamd_upload_materials(ctx, mats, 4) == RTAMD_OK);
    // host-buffer entry points: ragged sizes
    for (uint32_t w : {1u, 37u, 130u}) {
        const uint32_t h = 19; const size_t n = (size_t)w * h;
        std::vector<float> dir(3 * n), org(3 * n), rgba(4 * n), hits(4 * n), t(n); std::vector<uint32_t> id(n), cnt(2 * n);
        for (size_t i = 0; i < n; i++) { org[3 * i] = -3.f + 6.f * (i % w) / w; org[3 * i + 1] = -3.f + 6.f * (i / w) / h; org[3 * i + 2] = 45.f; dir[3 * i + 2] = -1.f; }
        rtamd_frame_desc d{}; d.width = w; d.height = h; d.dir = dir.data(); d.org = org.data(); d.has_tex_coords = 1; d.has_normals = 1; d.tile_rank = 0; d.tile_world = 1;
        d.near_plane = 0.01f; d.far_plane = 1000.f;
        std::vector<uint32_t> px8(n);
        rtamd_frame_out o{rgba.data(), id.data(), t.data(), cnt.data(), px8.data()};
        CHECK(rtamd_render(ctx, &d, &o) == RTAMD_OK);
        CHECK(rtamd_trace(ctx, n, org.data(), dir.data(), hits.data()) == RTAMD_OK);
        size_t nh = 0; for (size_t i = 0; i < n; i++) nh += id[i] != RTAMD_NO_HIT;
        CHECK(nh > 0 || w == 1);
        std::vector<float> p3(3 * n), nm(4 * n);
        CHECK(rtamd_surface_from_hits(ctx, n, org.data(), dir.data(), hits.data(), 1, p3.data(), nm.data()) == RTAMD_OK);
        rtamd_dispatch_desc dd{}; dd.width = w; dd.height = h; dd.position = p3.data(); dd.normal_mat = nm.data(); dd.light_pos[1] = 10.f; dd.camera_pos[2] = 45.f; dd.noise_offset_scale[2] = 1.f; dd.noise_offset_scale[3] = 1.f;
        std::vector<float> o1(3 * n), o2(3 * n), o3(3 * n), o4(3 * n);
        rtamd_dispatch_out dout{o1.data(), o2.data(), o3.data(), o4.data()};
        CHECK(rtamd_dispatch_rays(ctx, &dd, &dout) == RTAMD_OK);
        // tile shards
        for (uint32_t r = 0; r < 3; r++) { d.tile_rank = r; d.tile_world = 3; std::vector<float> part((size_t)rtamd_tiles_per_rank(w, h, 3) * 1024 * 4);
            rtamd_frame_out po{part.data(), nullptr, nullptr, nullptr}; CHECK(rtamd_render(ctx, &d, &po) == RTAMD_OK); }
    }
    // device entry points on several streams (slot take-over beyond 8), options
    const uint32_t W = 64, H = 48; const size_t n = (size_t)W * H;
    float *d_rgba = nullptr; CHECK(hipMalloc(&d_rgba, 10 * n * 16) == hipSuccess);
    hipStream_t st[10]; for (auto &s : st) CHECK(hipStreamCreate(&s) == hipSuccess);
    rtamd_frame_desc d{}; d.width = W; d.height = H; d.has_tex_coords = 1; d.has_normals = 1; d.tile_world = 1; d.near_plane = 0.01f; d.far_plane = 1000.f;
    const float ipv[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, -1, 0, 0, 1, 1};
    std::memcpy(d.inv_proj_view, ipv, sizeof ipv); d.cam_pos[2] = 45.f;
    for (const char *opt : {"dyn_thresh", "leaf_thresh", "num_queues", "queue_stride", "queue_block", "max_blocks_per_cu"})
        for (int64_t v : {1, 3, 16}) {
            CHECK(rtamd_set_option(ctx, opt, v) == RTAMD_OK);
            for (int k = 0; k < 12; k++) { CHECK(rtamd_context_set_stream(ctx, st[k % 10]) == RTAMD_OK); rtamd_frame_out o{d_rgba + (k % 10) * n * 4, nullptr, nullptr, nullptr}; CHECK(rtamd_render_device(ctx, &d, &o) == RTAMD_OK); }
        }
    // round 2: several frames per launch, the entry point that allocates nothing, buffer range checks
    {
        CHECK(rtamd_set_option(ctx, "max_blocks_per_cu", 0) == RTAMD_OK); CHECK(rtamd_set_option(ctx, "dyn_thresh", 0) == RTAMD_OK);
        rtamd_frame_desc ds[17]; rtamd_frame_out os[17];
        for (int k = 0; k < 17; k++) { ds[k] = d; ds[k].cam_pos[0] = 0.1f * k; os[k] = rtamd_frame_out{d_rgba + (k % 10) * n * 4, nullptr, nullptr, nullptr, nullptr}; }
        for (uint32_t nf : {1u, 2u, 10u}) CHECK(rtamd_render_batch_device(ctx, nf, ds, os) == RTAMD_OK);
        CHECK(rtamd_render_batch_device(ctx, 17, ds, os) == RTAMD_ERR_INVALID_ARGUMENT);
        CHECK(rtamd_render_batch_device(ctx, 0, ds, os) == RTAMD_ERR_INVALID_ARGUMENT);
        ds[1].width = 2 * W; CHECK(rtamd_render_batch_device(ctx, 2, ds, os) == RTAMD_ERR_INVALID_ARGUMENT); ds[1].width = W;
        CHECK(rtamd_render_device_static(ctx, &d, &os[0]) == RTAMD_OK);
        // a buffer that would run past the allocation it starts in is refused on the host: d_rgba holds 10 frames
        rtamd_frame_out past{d_rgba + 9 * n * 4 + 4, nullptr, nullptr, nullptr, nullptr};
        CHECK(rtamd_render_device(ctx, &d, &past) == RTAMD_ERR_INVALID_ARGUMENT);
        CHECK(rtamd_render_device_static(ctx, &d, &past) == RTAMD_ERR_INVALID_ARGUMENT);
        uint32_t *d8 = nullptr; CHECK(hipMalloc(&d8, n * 4) == hipSuccess);
        CHECK(rtamd_display_rgba8_device(ctx, d_rgba, n, d8) == RTAMD_OK);
        CHECK(rtamd_display_rgba8_device(ctx, d_rgba, n + 1, d8) == RTAMD_ERR_INVALID_ARGUMENT);
        CHECK(hipDeviceSynchronize() == hipSuccess); (void)hipFree(d8);
    }
    // round 2, later: ray lists in one launch, tile shards of ray generation and of the dispatcher
    {
        const uint32_t tiles = rtamd_tiles_per_rank(W, H, 3);
        const size_t ne = (size_t)tiles * 1024;
        float *d_dir = nullptr, *d_org = nullptr, *d_hits = nullptr, *d_pos = nullptr, *d_nm = nullptr, *d_out = nullptr;
        CHECK(hipMalloc(&d_dir, ne * 12) == hipSuccess); CHECK(hipMalloc(&d_org, ne * 12) == hipSuccess); CHECK(hipMalloc(&d_hits, 3 * ne * 16) == hipSuccess);
        CHECK(hipMalloc(&d_pos, ne * 12) == hipSuccess); CHECK(hipMalloc(&d_nm, ne * 16) == hipSuccess); CHECK(hipMalloc(&d_out, 4 * ne * 12) == hipSuccess);
        CHECK(hipMemset(d_org, 0, ne * 12) == hipSuccess);
        for (uint32_t r = 0; r < 3; r++) CHECK(rtamd_generate_rays_tiles_device(ctx, W, H, ipv, 0.01f, 1000.f, 0.25f, 0.75f, r, 3, d_dir) == RTAMD_OK);
        CHECK(rtamd_generate_rays_tiles_device(ctx, W, H, ipv, 0.01f, 1000.f, 0.25f, 0.75f, 3, 3, d_dir) == RTAMD_ERR_INVALID_ARGUMENT);
        CHECK(rtamd_generate_rays_tiles_device(ctx, W, H, ipv, 0.01f, 1000.f, 0.25f, 0.75f, 0, 1, d_dir) == RTAMD_ERR_INVALID_ARGUMENT);   // whole frame > the shard buffer
        rtamd_trace_job jobs[17];
        for (int k = 0; k < 17; k++) jobs[k] = rtamd_trace_job{k % 3 == 2 ? 0u : (uint64_t)(ne - 7 * k), d_org, d_dir, d_hits + (size_t)(k % 3) * ne * 4};
        for (uint32_t nj : {1u, 3u, 16u}) CHECK(rtamd_trace_batch_device(ctx, nj, jobs) == RTAMD_OK);
        CHECK(rtamd_trace_batch_device(ctx, 17, jobs) == RTAMD_ERR_INVALID_ARGUMENT);
        CHECK(rtamd_trace_batch_device(ctx, 0, jobs) == RTAMD_ERR_INVALID_ARGUMENT);
        jobs[1].n = ne + 1; CHECK(rtamd_trace_batch_device(ctx, 2, jobs) == RTAMD_ERR_INVALID_ARGUMENT); jobs[1].n = ne - 7;   // past the end of org / dir
        // round 3: the same lists as images (8x8-pixel work items): tile-major shard layout, scan-line order, ragged and bad sizes
        for (int k = 0; k < 16; k++) jobs[k].n = k % 3 == 2 ? 0u : (uint64_t)(ne - 1024 * (k % 2));
        for (uint32_t nj : {1u, 3u, 16u}) CHECK(rtamd_trace_image_batch_device(ctx, nj, jobs, 0, 3) == RTAMD_OK);
        jobs[0].n = ne - 7; CHECK(rtamd_trace_image_batch_device(ctx, 1, jobs, 0, 3) == RTAMD_ERR_INVALID_ARGUMENT);     // not whole tiles
        for (uint32_t wd : {1u, 8u, 37u, 64u}) { jobs[0].n = (ne / wd) * wd; jobs[1].n = wd; CHECK(rtamd_trace_image_batch_device(ctx, 2, jobs, wd, 1) == RTAMD_OK); }
        jobs[0].n = 37 * 5 + 1; CHECK(rtamd_trace_image_batch_device(ctx, 1, jobs, 37, 1) == RTAMD_ERR_INVALID_ARGUMENT);    // not whole rows
        jobs[0].n = 37 * 5; CHECK(rtamd_trace_image_batch_device(ctx, 1, jobs, 0, 1) == RTAMD_ERR_INVALID_ARGUMENT);         // no width
        jobs[0].n = ne + 1024; CHECK(rtamd_trace_image_batch_device(ctx, 1, jobs, 0, 3) == RTAMD_ERR_INVALID_ARGUMENT);      // past the end of org / dir
        CHECK(rtamd_surface_from_hits_device(ctx, ne, d_org, d_dir, d_hits, 1, d_pos, d_nm) == RTAMD_OK);
        rtamd_dispatch_desc dd{}; dd.width = W; dd.height = H; dd.position = d_pos; dd.normal_mat = d_nm; dd.light_pos[1] = 10.f; dd.camera_pos[2] = 45.f;
        dd.noise_offset_scale[2] = 1.f; dd.noise_offset_scale[3] = 1.f; dd.tile_rank = 2; dd.tile_world = 3;
        rtamd_dispatch_out dout{d_out, d_out + ne * 3, d_out + 2 * ne * 3, d_out + 3 * ne * 3};
        CHECK(rtamd_dispatch_rays_device(ctx, &dd, &dout) == RTAMD_OK);
        dd.tile_rank = 3; CHECK(rtamd_dispatch_rays_device(ctx, &dd, &dout) == RTAMD_ERR_INVALID_ARGUMENT);
        dd.tile_rank = 0; dd.tile_world = 0; CHECK(rtamd_dispatch_rays_device(ctx, &dd, &dout) == RTAMD_ERR_INVALID_ARGUMENT);     // whole frame > the shard buffers
        CHECK(hipDeviceSynchronize() == hipSuccess);
        for (float *q : {d_dir, d_org, d_hits, d_pos, d_nm, d_out}) (void)hipFree(q);
    }
    // round 3: config 5's chain as one call (two samples, a tile shard and the whole frame), and the device list (one GPU named twice, copy transport)
    for (uint32_t world : {1u, 3u}) {
        rtamd_multibounce_desc md{}; md.width = W; md.height = H; std::memcpy(md.inv_proj_view, ipv, sizeof ipv); md.near_plane = 0.01f; md.far_plane = 1000.f;
        md.cam_pos[2] = 45.f; md.light_pos[1] = 10.f; md.has_normals = 1; md.max_depth = 3; md.noise_size = 5; md.tile_rank = world - 1; md.tile_world = world;
        const uint64_t ne = rtamd_multibounce_entries(&md);
        CHECK(ne > 0 && rtamd_multibounce_primary_rays(&md) <= ne);
        float *pool = nullptr; CHECK(hipMalloc(&pool, 2 * 14 * ne * 16 + 16) == hipSuccess);
        rtamd_multibounce_lane lanes[2];
        for (int k = 0; k < 2; k++) {
            float *q = pool + (size_t)k * 14 * ne * 4; auto take = [&]() { float *r = q; q += ne * 4; return r; };
            lanes[k].cam_org = take(); lanes[k].dir = take(); lanes[k].hits = take(); lanes[k].position = take(); lanes[k].normal_mat = take();
            for (int t = 0; t < 2; t++) { lanes[k].org[t] = take(); lanes[k].shadow_dir[t] = take(); lanes[k].reflect_dir[t] = take(); lanes[k].ambient_dir[t] = take(); }
            lanes[k].shadow_hits = take();
            CHECK(rtamd_multibounce_prepare_lane_device(ctx, &md, &lanes[k]) == RTAMD_OK);
        }
        uint64_t *counter = (uint64_t *)(pool + 2 * 14 * ne * 4); CHECK(hipMemset(counter, 0, 8) == hipSuccess);
        const uint32_t samples[2] = {4, 7};
        CHECK(rtamd_multibounce_sample_batch_device(ctx, &md, samples, 2, lanes, counter) == RTAMD_OK);
        CHECK(rtamd_multibounce_sample_batch_device(ctx, &md, samples, 9, lanes, counter) == RTAMD_ERR_INVALID_ARGUMENT);
        rtamd_multibounce_lane twice[2] = {lanes[0], lanes[0]};
        CHECK(rtamd_multibounce_sample_batch_device(ctx, &md, samples, 2, twice, counter) == RTAMD_ERR_INVALID_ARGUMENT);
        CHECK(hipDeviceSynchronize() == hipSuccess);
        uint64_t traced = 0; CHECK(hipMemcpy(&traced, counter, 8, hipMemcpyDeviceToHost) == hipSuccess); CHECK(traced >= 2 * rtamd_multibounce_primary_rays(&md));
        (void)hipFree(pool);
    }
    {
        const int devs[3] = {0, 0, 0};
        rtamd_multi *m = nullptr;
        CHECK(rtamd_multi_create(devs, 3, 0, &m) == RTAMD_ERR_INVALID_ARGUMENT && m == nullptr);       // RCCL: one rank per GPU
        CHECK(rtamd_multi_create(devs, 3, RTAMD_MULTI_TRANSPORT_COPY, &m) == RTAMD_OK && m);
        if (m) {
            CHECK(rtamd_multi_upload_scene(m, &sc) == RTAMD_OK);
            for (uint32_t s = 0; s < 2; s++) CHECK(rtamd_multi_upload_texture(m, s, tex.data(), 7, 5) == RTAMD_OK);
            rtamd_frame_desc fd[3]; for (auto &f : fd) f = d;
            std::vector<uint32_t> px(3 * n); std::vector<float> pf(3 * n * 4);
            for (int rep = 0; rep < 3; rep++) CHECK(rtamd_multi_render_batch(m, 3, fd, RTAMD_GATHER_RGBA8, px.data()) == RTAMD_OK);
            CHECK(rtamd_multi_render_batch(m, 2, fd, RTAMD_GATHER_RGBA32F, pf.data()) == RTAMD_OK);
            CHECK(rtamd_multi_render_batch(m, 17, fd, RTAMD_GATHER_RGBA8, px.data()) == RTAMD_ERR_INVALID_ARGUMENT);
            float g_ms = 0, a_ms = 0; CHECK(rtamd_multi_last_gather_ms(m, &g_ms, &a_ms) == RTAMD_OK);
            CHECK(rtamd_multi_synchronize(m) == RTAMD_OK);
            rtamd_multi_destroy(m);
        }
    }
    CHECK(rtamd_abi_version() == RTAMD_ABI_VERSION);
    CHECK(rtamd_set_option(ctx, "no_such", 1) != RTAMD_OK);
    CHECK(hipDeviceSynchronize() == hipSuccess);
    float u, l; uint32_t th; (void)rtamd_last_launch_coherence(ctx, &u, &l, &th);
    float ms; (void)rtamd_last_kernel_ms(ctx, &ms);
    CHECK(rtamd_context_set_stream(ctx, nullptr) == RTAMD_OK);
    CHECK(rtamd_upload_materials(ctx, nullptr, 0) == RTAMD_OK);
    CHECK(rtamd_upload_scene(ctx, &sc) == RTAMD_OK);   // re-upload over a live scene
    for (auto &s : st) (void)hipStreamDestroy(s);
    (void)hipFree(d_rgba);
    rtamd_context_destroy(ctx);
    std::printf(failures ? "FAILED: %d checks\n" : "ok\n", failures);
    return failures ? 1 : 0;
}
