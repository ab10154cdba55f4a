// Host-layer sanitizer driver (CPU only): the C API of include/rtamd_host.h linked from source with
// -fsanitize=address,undefined or -fsanitize=thread (tools/sanitize/run.sh).  Exercises the OBJ / MTL
// parsers on good and malformed input, and builds BVHs serially and on 2..8 threads from random
// soups, clustered duplicates and degenerate input, comparing the node arrays and triangle orders.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/rtamd_host.h"

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); failures++; } } while (0)

static void parsers() {
    char err[256];
    const char *objs[] = {
        "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n",
        "mtllib a.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nvn 0 0 1\nusemtl m\nf 1/1/1 2/2/1 3/3/1\nusemtl n\nf 3/3/1 2/2/1 1/1/1\n",
        "v 0 0\nf 1 2 3\n", "f 1 2 3\n", "v a b c\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n", "", "\n\n#c\n", "v 0 0 0\nf 1/ 2/ 3/\n",
        "v 0 0 0\nv 1 0 0\nv 0 1 0\nf -1 -2 -3\n", "v 1e999 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3", "usemtl\nf\nv\nvt\nvn\n",
    };
    for (const char *o : objs) {
        rtamd_model *m = nullptr;
        int rc = rtamd_model_parse_obj(o, std::strlen(o), &m, err, sizeof err);
        if (rc == 0) {
            rtamd_model_info inf; rtamd_model_get_info(m, &inf);
            if (inf.triangle_count) { CHECK(rtamd_model_build_bvh(m) == 0); rtamd_model_get_info(m, &inf); CHECK(inf.has_bvh); }
            for (uint32_t i = 0; i < inf.material_count + 1; i++) (void)rtamd_model_material_name(m, i);
            for (uint32_t i = 0; i < 3; i++) (void)rtamd_model_material_lib_name(m, i);
            rtamd_model_free(m);
        } else CHECK(err[0] != 0);
    }
    const char *mtls[] = {
        "newmtl a\nKa 0.1 0.2 0.3\nKd 1 1 1\nKs 0 0 0\nNs 5\nd 0.5\nNi 1.5\nmap_Kd x.png\nnewmtl b\nTr 0.2\nmap_Ka a\nmap_Ks b\nmap_Ns c\n",
        "Ka 1 1 1\n", "newmtl\n", "newmtl a\nKa 1 1\n", "newmtl a\nNs x\n", "", "newmtl a\nTf 1 2 3\nillum 2\n",
    };
    for (const char *t : mtls) {
        rtamd_material_lib *lib = nullptr;
        int rc = rtamd_material_lib_parse(t, std::strlen(t), &lib, err, sizeof err);
        if (rc == 0) {
            for (uint32_t i = 0; i < rtamd_material_lib_count(lib) + 1; i++) { rtamd_material mm; (void)rtamd_material_lib_get(lib, i, &mm); }
            rtamd_material_lib_free(lib);
        }
    }
}

struct Tri { uint32_t p0, p1, p2, mat; };
struct Node { float b[6]; uint32_t leaf, a, c; };

static void build_compare(int kind, uint32_t n, uint32_t seed) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(-10.f, 10.f), S(-0.05f, 0.05f);
    std::vector<float> pos(9 * (size_t)n);
    std::vector<Tri> tri(n);
    for (uint32_t i = 0; i < n; i++) {
        float c[3] = {U(rng), U(rng), U(rng)};
        if (kind == 1 && i % 10) { c[0] = c[1] = c[2] = 0.001f * (i % 7); }        // clustered, many equal centroids
        for (int v = 0; v < 3; v++) for (int k = 0; k < 3; k++) pos[9 * (size_t)i + 3 * v + k] = c[k] + (kind == 2 ? 0.0f : S(rng));   // kind 2: degenerate points
        if (kind == 3) for (int v = 0; v < 3; v++) pos[9 * (size_t)i + 3 * v + 1] = (i % 3 == 0) ? -0.0f : 0.0f;                    // flat, signed zeros
        tri[i] = Tri{3 * i, 3 * i + 1, 3 * i + 2, i & 1u};
    }
    rtamd_model *ref = nullptr;
    CHECK(rtamd_model_from_arrays(tri.data(), n, pos.data(), 3 * n, nullptr, nullptr, &ref) == 0);
    CHECK(rtamd_model_build_bvh(ref) == 0);
    rtamd_model_info ri; rtamd_model_get_info(ref, &ri);
    for (uint32_t threads : {2u, 3u, 8u}) {
        rtamd_model *m = nullptr;
        CHECK(rtamd_model_from_arrays(tri.data(), n, pos.data(), 3 * n, nullptr, nullptr, &m) == 0);
        CHECK(rtamd_model_build_bvh_parallel(m, threads) == 0);
        rtamd_model_info mi; rtamd_model_get_info(m, &mi);
        CHECK(mi.node_count == ri.node_count);
        if (mi.node_count == ri.node_count) {
            CHECK(std::memcmp(rtamd_model_bvh_nodes(m), rtamd_model_bvh_nodes(ref), sizeof(Node) * (size_t)ri.node_count) == 0);
            CHECK(std::memcmp(rtamd_model_triangles(m), rtamd_model_triangles(ref), sizeof(Tri) * (size_t)n) == 0);
        }
        rtamd_model_free(m);
    }
    rtamd_model_free(ref);
}

int main(int argc, char **argv) {
    const uint32_t big = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 60000u;
    parsers();
    for (int kind = 0; kind < 4; kind++) {
        build_compare(kind, 1, 1); build_compare(kind, 2, 2); build_compare(kind, 7, 3); build_compare(kind, 500, 4);
        build_compare(kind, big, 5);
    }
    float mats[3][16], pos[3] = {1, 2, 3}, up[3] = {0, 1, 0}, dir[3];
    rtamd_camera_direction(0.3f, -0.2f, dir);
    (void)mats; (void)pos; (void)up;
    std::printf(failures ? "FAILED: %d checks\n" : "ok\n", failures);
    return failures ? 1 : 0;
}
