/* A plain C99 caller of the C-ABI (include/rtamd.h) — what the reference's Rust side would do through `extern "C"`
 * (INTEGRATION.md), written in the one language every FFI agrees with: no C++, no Python, no torch.
 *
 * It follows the reference's frame set-up step by step (src/main.rs): create the context (:23,33-50), upload the five
 * scene slices (:93-102) and the two albedo textures (texture.rs:65-79), draw the fused pass into host buffers
 * (:204-230) and the split trace pass (ray_trace.frag), then tear down — and exercises the error contract on the way
 * (empty scene, render before upload, malformed tree).
 *
 *   drop_in <scene.bin> <out.bin> [--gpus n] [--copy]
 *       scene.bin / out.bin: see tests/test_c_caller.py, which writes one and reads the other.
 *       --gpus n: after the single-GPU pass, the same frame through the one-process multi-GPU entry points
 *       (rtamd_multi_*: tile split over n GPUs, ncclGather inside the library, assembly on the first) must come out bit for
 *       bit as the single context rendered it, in both gather formats.  --copy: n contexts on GPU 0 with the copy transport
 *       (RCCL takes one rank per GPU), which is how the n > 1 logic runs on a one-GPU box.
 *
 * Exit status: 0 ok; 3 no usable GPU (RTAMD_ERR_DEVICE from rtamd_context_create: the library has no CPU fallback);
 * anything else a failure, with the reason on stderr.  Built by __graft_entry__.build():
 *   gcc -std=c99 -Wall -Wextra -Werror -pedantic -Iinclude tests/c_caller/drop_in.c -Lraytracer_amd/lib -lrtamd ...
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtamd.h"

#define SCENE_MAGIC 0x314E4353u /* "SCN1" */

typedef struct scene_header {
    uint32_t magic, node_count, triangle_count, vertex_count, has_tex_coords, has_normals, width, height, tex_w[2], tex_h[2];
} scene_header;

static void *read_block(FILE *f, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (!p || (bytes && fread(p, 1, bytes, f) != bytes)) {
        fprintf(stderr, "drop_in: short read (%zu bytes)\n", bytes);
        exit(10);
    }
    return p;
}

static int expect(rtamd_context *ctx, int rc, int want, const char *what) {
    if (rc != want) {
        fprintf(stderr, "drop_in: %s returned %d, expected %d (%s)\n", what, rc, want, rtamd_last_error(ctx));
        return 1;
    }
    if (want != RTAMD_OK && rtamd_last_error(ctx)[0] == '\0') {
        fprintf(stderr, "drop_in: %s failed without an error text\n", what);
        return 1;
    }
    return 0;
}

/* The frame of `fd` (host rays) twice in one batch through rtamd_multi_*: every pixel of both frames, in both gather
 * formats, must equal what the single context produced (want32 / want8). */
static int multi_gpu_leg(uint32_t n, uint32_t copy, const rtamd_scene_desc *sd, uint8_t *tex[2], const uint32_t tex_w[2], const uint32_t tex_h[2],
                         const rtamd_frame_desc *fd, const float *want32, const uint32_t *want8) {
    int devices[64];
    for (uint32_t i = 0; i < n; i++) devices[i] = copy ? 0 : (int)i;
    rtamd_multi *m = NULL;
    int rc = rtamd_multi_create(devices, n, copy ? RTAMD_MULTI_TRANSPORT_COPY : RTAMD_MULTI_TRANSPORT_RCCL, &m);
    if (rc != RTAMD_OK) {
        fprintf(stderr, "drop_in: rtamd_multi_create(%u GPUs): %d (%s)\n", n, rc, rtamd_multi_last_error(NULL));
        return 30;
    }
    int bad = 0;
    const size_t n_px = (size_t)fd->width * fd->height;
    float *got32 = malloc(2 * n_px * 16);
    uint32_t *got8 = malloc(2 * n_px * 4);
    if (!got32 || !got8) return 17;
    if (rtamd_multi_device_count(m) != n || !rtamd_multi_context(m, n - 1) || rtamd_multi_context(m, n)) bad++;
    rc = rtamd_multi_upload_scene(m, sd);
    for (uint32_t k = 0; k < 2 && rc == RTAMD_OK; k++)
        if (tex[k]) rc = rtamd_multi_upload_texture(m, k, tex[k], tex_w[k], tex_h[k]);
    rtamd_frame_desc two[2];
    two[0] = *fd; two[1] = *fd;
    for (int pass = 0; pass < 3 && rc == RTAMD_OK; pass++) {      /* three calls: both sets of share buffers, and one of them again */
        if (rc == RTAMD_OK) rc = rtamd_multi_render_batch(m, 2, two, RTAMD_GATHER_RGBA8, got8);
        if (rc == RTAMD_OK) rc = rtamd_multi_render_batch(m, 2, two, RTAMD_GATHER_RGBA32F, got32);
        if (rc == RTAMD_OK)
            for (int f = 0; f < 2; f++) {
                if (memcmp(got8 + f * n_px, want8, n_px * 4)) { fprintf(stderr, "drop_in: multi-GPU RGBA8 frame %d differs (pass %d)\n", f, pass); bad++; }
                if (memcmp(got32 + 4 * f * n_px, want32, n_px * 16)) { fprintf(stderr, "drop_in: multi-GPU RGBA32F frame %d differs (pass %d)\n", f, pass); bad++; }
            }
    }
    float g_ms = -1.0f, a_ms = -1.0f;
    if (rc == RTAMD_OK) rc = rtamd_multi_last_gather_ms(m, &g_ms, &a_ms);
    if (rc == RTAMD_OK) rc = rtamd_multi_synchronize(m);
    if (rc != RTAMD_OK) { fprintf(stderr, "drop_in: multi-GPU leg: %d (%s)\n", rc, rtamd_multi_last_error(m)); bad++; }
    /* the error contract: 17 frames, an unknown format */
    if (rtamd_multi_render_batch(m, 17, two, RTAMD_GATHER_RGBA8, got8) != RTAMD_ERR_INVALID_ARGUMENT || rtamd_multi_last_error(m)[0] == '\0') bad++;
    if (rtamd_multi_render_batch(m, 2, two, 7, got8) != RTAMD_ERR_INVALID_ARGUMENT) bad++;
    rtamd_multi_destroy(m);
    free(got32); free(got8);
    if (bad) return 21;
    printf("multi ok: %u GPU(s), %s transport, gather %.3f ms, assembly %.3f ms\n", n, copy ? "copy" : "RCCL", g_ms, a_ms);
    return 0;
}

int main(int argc, char **argv) {
    if (rtamd_abi_version() != RTAMD_ABI_VERSION) {
        fprintf(stderr, "drop_in: header ABI %d, library ABI %u\n", RTAMD_ABI_VERSION, rtamd_abi_version());
        return 11;
    }
    rtamd_context *ctx = NULL;
    int rc = rtamd_context_create(0, &ctx);
    if (rc == RTAMD_ERR_DEVICE) {
        printf("no GPU: %s\n", rtamd_last_error(NULL));
        return ctx == NULL && rtamd_last_error(NULL)[0] != '\0' ? 3 : 12;
    }
    if (rc != RTAMD_OK) {
        fprintf(stderr, "drop_in: rtamd_context_create: %d (%s)\n", rc, rtamd_last_error(NULL));
        return 13;
    }
    uint32_t multi_gpus = 0, multi_copy = 0;
    for (int a = 3; a < argc; a++) {
        if (!strcmp(argv[a], "--gpus") && a + 1 < argc) multi_gpus = (uint32_t)strtoul(argv[++a], NULL, 10);
        else if (!strcmp(argv[a], "--copy")) multi_copy = 1;
        else argc = 0;
    }
    if (argc < 3 || multi_gpus > 64) {
        fprintf(stderr, "usage: drop_in scene.bin out.bin [--gpus n] [--copy]\n");
        rtamd_context_destroy(ctx);
        return 14;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 15; }
    scene_header h;
    if (fread(&h, sizeof h, 1, f) != 1 || h.magic != SCENE_MAGIC) { fprintf(stderr, "drop_in: bad scene file\n"); return 16; }
    const size_t n_px = (size_t)h.width * h.height;
    void *nodes = read_block(f, (size_t)h.node_count * 36);
    void *tris = read_block(f, (size_t)h.triangle_count * 16);
    float *pos = read_block(f, (size_t)h.vertex_count * 12);
    float *uv = h.has_tex_coords ? read_block(f, (size_t)h.vertex_count * 8) : NULL;
    float *nrm = h.has_normals ? read_block(f, (size_t)h.vertex_count * 12) : NULL;
    float *dir = read_block(f, n_px * 12), *org = read_block(f, n_px * 12);
    uint8_t *tex[2];
    for (int k = 0; k < 2; k++) tex[k] = h.tex_w[k] ? read_block(f, (size_t)h.tex_w[k] * h.tex_h[k] * 3) : NULL;
    fclose(f);

    int bad = 0;
    rtamd_frame_desc fd;
    rtamd_frame_out fo;
    memset(&fd, 0, sizeof fd);
    memset(&fo, 0, sizeof fo);
    fd.width = h.width; fd.height = h.height; fd.dir = dir; fd.org = org;
    fd.has_tex_coords = h.has_tex_coords; fd.has_normals = h.has_normals; fd.tile_rank = 0; fd.tile_world = 1;
    float *rgba = malloc(n_px * 16), *hit_t = malloc(n_px * 4), *hits = malloc(n_px * 16);
    uint32_t *hit_id = malloc(n_px * 4), *rgba8 = malloc(n_px * 4);
    if (!rgba || !hit_t || !hits || !hit_id || !rgba8) return 17;
    fo.rgba32f = rgba; fo.hit_id = hit_id; fo.hit_t = hit_t; fo.rgba8 = rgba8;

    /* the error contract first: nothing uploaded yet */
    bad += expect(ctx, rtamd_render(ctx, &fd, &fo), RTAMD_ERR_NO_SCENE, "render before upload");
    bad += expect(ctx, rtamd_trace(ctx, n_px, org, dir, hits), RTAMD_ERR_NO_SCENE, "trace before upload");
    rtamd_scene_desc sd;
    memset(&sd, 0, sizeof sd);
    sd.nodes = nodes; sd.node_count = 0; sd.triangles = tris; sd.triangle_count = h.triangle_count;
    sd.positions = pos; sd.vertex_count = h.vertex_count; sd.tex_coords = uv; sd.normals = nrm;
    bad += expect(ctx, rtamd_upload_scene(ctx, &sd), RTAMD_ERR_EMPTY_SCENE, "upload of an empty node slice");
    sd.node_count = h.node_count;
    if (h.node_count > 2) {   /* the root's first child points at the root itself: not a tree */
        uint32_t *root = (uint32_t *)nodes + 6;
        if (!root[0]) {
            const uint32_t keep = root[1];
            root[1] = 0;
            bad += expect(ctx, rtamd_upload_scene(ctx, &sd), RTAMD_ERR_MALFORMED_SCENE, "upload of a cyclic tree");
            root[1] = keep;
        }
    }
    bad += expect(ctx, rtamd_upload_texture(ctx, RTAMD_MAX_TEXTURE_SLOTS, (const uint8_t *)"\0\0\0", 1, 1), RTAMD_ERR_INVALID_ARGUMENT, "texture slot out of range");

    /* the reference's set-up, in its order */
    bad += expect(ctx, rtamd_upload_scene(ctx, &sd), RTAMD_OK, "rtamd_upload_scene");
    for (uint32_t k = 0; k < 2; k++)
        if (tex[k]) bad += expect(ctx, rtamd_upload_texture(ctx, k, tex[k], h.tex_w[k], h.tex_h[k]), RTAMD_OK, "rtamd_upload_texture");
    rtamd_scene_info info;
    bad += expect(ctx, rtamd_get_scene_info(ctx, &info), RTAMD_OK, "rtamd_get_scene_info");
    if (!bad && (info.node_count != h.node_count || info.triangle_count != h.triangle_count || info.vertex_count != h.vertex_count)) {
        fprintf(stderr, "drop_in: scene info does not match the upload\n");
        bad++;
    }
    bad += expect(ctx, rtamd_render(ctx, &fd, &fo), RTAMD_OK, "rtamd_render");
    bad += expect(ctx, rtamd_trace(ctx, n_px, org, dir, hits), RTAMD_OK, "rtamd_trace");
    bad += expect(ctx, rtamd_synchronize(ctx), RTAMD_OK, "rtamd_synchronize");
    rtamd_context_destroy(ctx);
    if (bad) return 20;
    if (multi_gpus) {
        const int mrc = multi_gpu_leg(multi_gpus, multi_copy, &sd, tex, h.tex_w, h.tex_h, &fd, rgba, rgba8);
        if (mrc) return mrc;
    }

    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 18; }
    if (fwrite(rgba, 16, n_px, f) != n_px || fwrite(hit_id, 4, n_px, f) != n_px || fwrite(hit_t, 4, n_px, f) != n_px ||
        fwrite(rgba8, 4, n_px, f) != n_px || fwrite(hits, 16, n_px, f) != n_px || fclose(f) != 0) {
        fprintf(stderr, "drop_in: short write\n");
        return 19;
    }
    printf("ok: %ux%u, %u triangles, tree depth %u\n", h.width, h.height, info.triangle_count, info.max_depth);
    free(nodes); free(tris); free(pos); free(uv); free(nrm); free(dir); free(org); free(tex[0]); free(tex[1]);
    free(rgba); free(hit_t); free(hits); free(hit_id); free(rgba8);
    return 0;
}
