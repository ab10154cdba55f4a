// ResourceParser::parse_model — line-based Wavefront OBJ reader with the
// reference's exact behaviour (src/resource/resource_parser.rs:13-47,111-149,
// src/raytracing/types.rs:154-180):
//   * the file is split on '\n'; a line is recognised only by its literal prefix
//     "mtllib " / "usemtl " / "f " / "v " / "vt " / "vn " at column 0;
//   * tokens are split on ' ' and tokens that trim to "" are dropped;
//   * "v"/"vn" need exactly 3 values, "vt" 2 or 3 (only the first two are kept);
//   * "f" needs >= 3 corners and is fan-triangulated (0, i+1, i+2);
//   * a corner is split on '/' or '\\'; a missing or empty component stays 0;
//     positive indices are 1-based, zero stays 0, negatives are resolved against
//     the final attribute counts in ModelBuilder::build;
//   * material ids are handed out in first-`usemtl` order.
// Float parsing uses correctly-rounded decimal->binary32 conversion, as
// `str::parse::<f32>` does (resource_parser.rs:122).
#pragma once
#include <string>

#include "material.h"
#include "model.h"

namespace rtamd {
namespace host {

class ResourceParser {
public:
    // On failure returns an empty model and fills `err` (line is 1-based).
    static Model parse_model(const std::string &data, ParseError *err);
    // resource_parser.rs:49-109: "newmtl", "Ka", "Kd", "Ks", "Tf" (3 values), "Ns", "d" (stored as
    // transmission = 1 - d), "Tr", "Ni" (1 value), "map_Ka/Kd/Ks/Ns" (one name); lines are matched by
    // literal prefix at column 0 and everything else is ignored; a property before the first
    // `newmtl` is the NoMaterialNamed error.  Materials are returned in file order.
    static std::vector<std::pair<std::string, Material>> parse_material_lib(const std::string &data, ParseError *err);
};

}  // namespace host
}  // namespace rtamd
