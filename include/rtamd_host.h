/*
 * rtamd_host.h — C view of the host-side scene layer (librtamd_host.so).
 *
 * The reference's host is Rust; no Rust toolchain exists in the build image, so
 * the host classes that feed the hot path (ResourceParser::parse_model,
 * ModelBuilder, Model, BVHBuilder, BVH, Camera) are restated in C++ under
 * raytracer_amd/csrc/host/ with the reference's names and behaviour.  This header
 * exposes them through plain C so that tests, bench.py and non-C++ callers can
 * drive them; a Rust host would keep its own implementations and bind only
 * include/rtamd.h.  Pure CPU code: no GPU needed.
 */
#ifndef RTAMD_HOST_H
#define RTAMD_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtamd_model rtamd_model;

/* ResourceParser::parse_model (src/resource/resource_parser.rs:13-47) followed by
 * ModelBuilder::build (src/rendering/model.rs:101-149).  Returns 0 and a model,
 * or a negative code and an error text "<kind> at line <n>: <detail>" in err
 * (ResourceError::ResourceParseError, src/util/error.rs:24-33). */
int rtamd_model_parse_obj(const char *data, size_t len, rtamd_model **out, char *err, size_t err_len);

/* ---- material libraries: ResourceParser::parse_material_lib (resource_parser.rs:49-109),
 * Material / MaterialLibBuilder (src/rendering/material.rs:5-116); SURVEY section 8 row f4 ---- */
typedef struct rtamd_material_lib rtamd_material_lib;
typedef struct rtamd_material {
    const char *name;                       /* newmtl name; borrowed, valid until the library is freed */
    float ambient_color[3], diffuse_color[3], specular_color[3], transmission_color[3];   /* Ka Kd Ks Tf; defaults 0.3 / 1 / 1 / 1 */
    float specular_exp, transmission, optical_density;                                    /* Ns (10), Tr or 1 - d (0), Ni (1) */
    const char *ambient_tex, *diffuse_tex, *specular_tex, *specular_exp_tex;              /* map_Ka/Kd/Ks/Ns or NULL (Option::None) */
} rtamd_material;
/* Returns 0, or -(kind+1) with a "Kind at line N: detail" message in err (kinds as for
 * rtamd_model_parse_obj, plus NoMaterialNamed = a property before the first newmtl). */
int rtamd_material_lib_parse(const char *data, size_t len, rtamd_material_lib **out, char *err, size_t err_len);
void rtamd_material_lib_free(rtamd_material_lib *lib);
uint32_t rtamd_material_lib_count(const rtamd_material_lib *lib);
int rtamd_material_lib_get(const rtamd_material_lib *lib, uint32_t index, rtamd_material *out);

/* A model straight from arrays (synthetic scenes): triangles are 16-byte
 * (p0,p1,p2,mat_idx) records; tex_coords / normals may be NULL. */
int rtamd_model_from_arrays(const void *triangles, uint32_t triangle_count, const float *positions, uint32_t vertex_count,
                            const float *tex_coords, const float *normals, rtamd_model **out);

void rtamd_model_free(rtamd_model *m);

/* Model::build_bvh (model.rs:36) = BVHBuilder::new(model).build()
 * (src/raytracing/bvh.rs:61-79): binned SAH, reorders the model's triangles. */
int rtamd_model_build_bvh(rtamd_model *m);
/* The same BVH (nodes, numbering, triangle order: bit for bit) built on `threads` host threads;
 * SURVEY section 8 row f3, no counterpart in the reference (bvh.rs:150-242 is single-threaded). */
int rtamd_model_build_bvh_parallel(rtamd_model *m, uint32_t threads);

typedef struct rtamd_model_info {
    uint32_t triangle_count, vertex_count, node_count, material_count;
    uint32_t has_tex_coords, has_normals, has_bvh;
} rtamd_model_info;
void rtamd_model_get_info(const rtamd_model *m, rtamd_model_info *info);

/* Borrowed pointers into the model (valid until the model is freed or its BVH
 * rebuilt): Model::triangles / positions / tex_coords / normals (model.rs:20-24),
 * BVH::data (bvh.rs:49-51).  NULL when the Option is None. */
const void *rtamd_model_triangles(const rtamd_model *m);
const float *rtamd_model_positions(const rtamd_model *m);
const float *rtamd_model_tex_coords(const rtamd_model *m);
const float *rtamd_model_normals(const rtamd_model *m);
const void *rtamd_model_bvh_nodes(const rtamd_model *m);
/* Material name for mat_idx i in first-`usemtl` order (model.rs:89-97,136-137). */
const char *rtamd_model_material_name(const rtamd_model *m, uint32_t i);
/* i-th `mtllib` file name in file order (model.rs:85-87), NULL past the end. */
const char *rtamd_model_material_lib_name(const rtamd_model *m, uint32_t i);

/* Camera::view_proj_matrices (src/rendering/camera.rs:75-82) and the inverse of
 * proj*view that src/main.rs:149 uploads; all matrices column-major float[16].
 * Any output may be NULL.  Returns -1 when proj*view is singular. */
int rtamd_camera_matrices(float aspect, float fov_rad, const float position[3], const float up[3], const float direction[3],
                          float view[16], float proj[16], float inv_proj_view[16]);

/* Camera::set_rotation (camera.rs:48-56): yaw/pitch -> unit direction. */
void rtamd_camera_direction(float yaw, float pitch, float direction_out[3]);

/* Host mirror of ray_create.frag:12: one direction per pixel centre, row 0 = bottom. */
void rtamd_host_generate_primary_rays(const float inv_proj_view[16], float near_plane, float far_plane,
                                      uint32_t width, uint32_t height, float *dir_out);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_HOST_H */
