// C view of the host layer: see include/rtamd_host.h.
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "../../../include/rtamd_host.h"
#include "bvh.h"
#include "camera.h"
#include "model.h"
#include "resource_parser.h"

using namespace rtamd::host;

struct rtamd_model { Model model; };
struct rtamd_material_lib { std::vector<std::pair<std::string, Material>> materials; };

static const char *kind_name(ParseErrorKind k) {
    switch (k) {
        case ParseErrorKind::ParseIntError: return "ParseIntError";
        case ParseErrorKind::ParseFloatError: return "ParseFloatError";
        case ParseErrorKind::InvalidLineArgCount: return "InvalidLineArgCount";
        case ParseErrorKind::IndexOutOfRange: return "IndexOutOfRange";
        case ParseErrorKind::NoMaterialNamed: return "NoMaterialNamed";
        default: return "None";
    }
}

extern "C" {

int rtamd_model_parse_obj(const char *data, size_t len, rtamd_model **out, char *err, size_t err_len) {
    if (err && err_len) err[0] = 0;
    if (!data || !out) return -1;
    try {
        ParseError e;
        Model m = ResourceParser::parse_model(std::string(data, len), &e);
        if (e) {
            if (err && err_len) std::snprintf(err, err_len, "%s at line %u: %s", kind_name(e.kind), e.line, e.detail.c_str());
            return -(int)e.kind - 1;
        }
        *out = new rtamd_model{std::move(m)};
        return 0;
    } catch (const std::exception &ex) {
        if (err && err_len) std::snprintf(err, err_len, "exception: %s", ex.what());
        return -100;
    }
}

int rtamd_material_lib_parse(const char *data, size_t len, rtamd_material_lib **out, char *err, size_t err_len) {
    if (err && err_len) err[0] = 0;
    if (!data || !out) return -1;
    try {
        ParseError e;
        auto mats = ResourceParser::parse_material_lib(std::string(data, len), &e);
        if (e) {
            if (err && err_len) std::snprintf(err, err_len, "%s at line %u: %s", kind_name(e.kind), e.line, e.detail.c_str());
            return -(int)e.kind - 1;
        }
        *out = new rtamd_material_lib{std::move(mats)};
        return 0;
    } catch (const std::exception &ex) {
        if (err && err_len) std::snprintf(err, err_len, "exception: %s", ex.what());
        return -100;
    }
}

void rtamd_material_lib_free(rtamd_material_lib *lib) { delete lib; }
uint32_t rtamd_material_lib_count(const rtamd_material_lib *lib) { return lib ? (uint32_t)lib->materials.size() : 0; }

int rtamd_material_lib_get(const rtamd_material_lib *lib, uint32_t index, rtamd_material *out) {
    if (!lib || !out || index >= lib->materials.size()) return -1;
    const Material &m = lib->materials[index].second;
    out->name = lib->materials[index].first.c_str();
    const Vec3 *src[4] = {&m.ambient_color, &m.diffuse_color, &m.specular_color, &m.transmission_color};
    float *dst[4] = {out->ambient_color, out->diffuse_color, out->specular_color, out->transmission_color};
    for (int k = 0; k < 4; k++) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
    out->specular_exp = m.specular_exp; out->transmission = m.transmission; out->optical_density = m.optical_density;
    out->ambient_tex = m.has_ambient_tex ? m.ambient_tex.c_str() : nullptr;
    out->diffuse_tex = m.has_diffuse_tex ? m.diffuse_tex.c_str() : nullptr;
    out->specular_tex = m.has_specular_tex ? m.specular_tex.c_str() : nullptr;
    out->specular_exp_tex = m.has_specular_exp_tex ? m.specular_exp_tex.c_str() : nullptr;
    return 0;
}

int rtamd_model_from_arrays(const void *triangles, uint32_t triangle_count, const float *positions, uint32_t vertex_count,
                            const float *tex_coords, const float *normals, rtamd_model **out) {
    if (!triangles || !positions || !out) return -1;
    try {
        const Triangle *t = (const Triangle *)triangles;
        for (uint32_t i = 0; i < triangle_count; i++)
            if (t[i].p0 >= vertex_count || t[i].p1 >= vertex_count || t[i].p2 >= vertex_count) return -2;
        std::vector<Triangle> tris(t, t + triangle_count);
        const Vec3 *p = (const Vec3 *)positions;
        std::vector<Vec3> pos(p, p + vertex_count);
        std::vector<Vec2> tc;
        std::vector<Vec3> nr;
        if (tex_coords) tc.assign((const Vec2 *)tex_coords, (const Vec2 *)tex_coords + vertex_count);
        if (normals) nr.assign((const Vec3 *)normals, (const Vec3 *)normals + vertex_count);
        *out = new rtamd_model{Model::from_arrays(std::move(tris), std::move(pos), tex_coords ? &tc : nullptr, normals ? &nr : nullptr)};
        return 0;
    } catch (const std::exception &) {
        return -100;
    }
}

void rtamd_model_free(rtamd_model *m) { delete m; }

int rtamd_model_build_bvh(rtamd_model *m) {
    if (!m) return -1;
    try {
        m->model.build_bvh();
        return 0;
    } catch (const std::exception &) {
        return -100;
    }
}

int rtamd_model_build_bvh_parallel(rtamd_model *m, uint32_t threads) {
    if (!m) return -1;
    try {
        m->model.build_bvh_parallel(threads);
        return 0;
    } catch (const std::exception &) {
        return -100;
    }
}

void rtamd_model_get_info(const rtamd_model *m, rtamd_model_info *info) {
    if (!m || !info) return;
    const Model &md = m->model;
    info->triangle_count = (uint32_t)md.triangles().size();
    info->vertex_count = (uint32_t)md.positions().size();
    info->node_count = md.get_bvh() ? (uint32_t)md.get_bvh()->data().size() : 0;
    info->material_count = (uint32_t)md.get_materials().size();
    info->has_tex_coords = md.has_tex_coords();
    info->has_normals = md.has_normals();
    info->has_bvh = md.get_bvh() != nullptr;
}

const void *rtamd_model_triangles(const rtamd_model *m) { return m && !m->model.triangles().empty() ? m->model.triangles().data() : nullptr; }
const float *rtamd_model_positions(const rtamd_model *m) { return m && !m->model.positions().empty() ? &m->model.positions()[0].x : nullptr; }
const float *rtamd_model_tex_coords(const rtamd_model *m) {
    return m && m->model.tex_coords() && !m->model.tex_coords()->empty() ? &(*m->model.tex_coords())[0].x : nullptr;
}
const float *rtamd_model_normals(const rtamd_model *m) {
    return m && m->model.normals() && !m->model.normals()->empty() ? &(*m->model.normals())[0].x : nullptr;
}
const void *rtamd_model_bvh_nodes(const rtamd_model *m) { return m && m->model.get_bvh() ? m->model.get_bvh()->data().data() : nullptr; }
const char *rtamd_model_material_name(const rtamd_model *m, uint32_t i) {
    return m && i < m->model.get_materials().size() ? m->model.get_materials()[i].c_str() : nullptr;
}

const char *rtamd_model_material_lib_name(const rtamd_model *m, uint32_t i) {
    return m && i < m->model.get_material_libs().size() ? m->model.get_material_libs()[i].c_str() : nullptr;
}

int rtamd_camera_matrices(float aspect, float fov_rad, const float position[3], const float up[3], const float direction[3],
                          float view[16], float proj[16], float inv_proj_view[16]) {
    Camera cam(aspect, fov_rad, Vec3{position[0], position[1], position[2]}, Vec3{up[0], up[1], up[2]},
               Vec3{direction[0], direction[1], direction[2]});
    CameraViewProjMatrices vp = cam.view_proj_matrices();
    if (view) std::memcpy(view, vp.view.m, sizeof vp.view.m);
    if (proj) std::memcpy(proj, vp.proj.m, sizeof vp.proj.m);
    if (inv_proj_view) {
        Mat4 inv;
        if (!(vp.proj * vp.view).invert(&inv)) return -1;
        std::memcpy(inv_proj_view, inv.m, sizeof inv.m);
    }
    return 0;
}

void rtamd_camera_direction(float yaw, float pitch, float direction_out[3]) {
    Camera cam = Camera::new_default(1.0f);
    cam.set_rotation(yaw, pitch);
    Vec3 d = cam.direction();
    direction_out[0] = d.x; direction_out[1] = d.y; direction_out[2] = d.z;
}

void rtamd_host_generate_primary_rays(const float inv_proj_view[16], float near_plane, float far_plane,
                                      uint32_t width, uint32_t height, float *dir_out) {
    Mat4 m;
    std::memcpy(m.m, inv_proj_view, sizeof m.m);
    generate_primary_rays(m, near_plane, far_plane, width, height, dir_out);
}

}  // extern "C"
