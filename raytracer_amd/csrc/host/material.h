// Material + MaterialLibBuilder — Wavefront MTL records, API and defaults kept from the
// reference (src/rendering/material.rs:5-116).  The v0 shaders do not read them (the fused
// shader's Phong constants are hard-coded, ray_trace_and_shade.frag:223-229; the dispatcher's
// Material{bool reflect} is unused, ray_dispatcher.frag:36); they name the albedo textures
// (map_Kd) that main.rs binds by hand (src/main.rs:51-52).
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "types.h"

namespace rtamd {
namespace host {

struct Material {  // material.rs:5-21, defaults :24-38
    Vec3 ambient_color{0.3f, 0.3f, 0.3f};
    Vec3 diffuse_color{1.0f, 1.0f, 1.0f};
    Vec3 specular_color{1.0f, 1.0f, 1.0f};
    Vec3 transmission_color{1.0f, 1.0f, 1.0f};
    float specular_exp = 10.0f;
    float transmission = 0.0f;
    float optical_density = 1.0f;
    // Option<String>: has_* == false is None
    bool has_ambient_tex = false, has_diffuse_tex = false, has_specular_tex = false, has_specular_exp_tex = false;
    std::string ambient_tex, diffuse_tex, specular_tex, specular_exp_tex;

    std::vector<std::string> get_texture_names() const {  // material.rs:40-47 (ambient, diffuse, specular, specular_exp order)
        std::vector<std::string> names;
        if (has_ambient_tex) names.push_back(ambient_tex);
        if (has_diffuse_tex) names.push_back(diffuse_tex);
        if (has_specular_tex) names.push_back(specular_tex);
        if (has_specular_exp_tex) names.push_back(specular_exp_tex);
        return names;
    }
};

class MaterialLibBuilder {  // material.rs:50-116
public:
    std::vector<std::pair<std::string, Material>> build() { return std::move(materials_); }
    void add_material(const std::string &name) { materials_.emplace_back(name, Material()); }
    // every setter fails with NoMaterialNamed before the first `newmtl` (material.rs:62-64)
    Material *current() { return materials_.empty() ? nullptr : &materials_.back().second; }

private:
    std::vector<std::pair<std::string, Material>> materials_;
};

}  // namespace host
}  // namespace rtamd
