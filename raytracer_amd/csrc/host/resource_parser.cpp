#include "resource_parser.h"

#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace rtamd {
namespace host {
namespace {

struct Tok { const char *p; size_t n; };

inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

inline Tok trim(Tok t) {
    while (t.n && is_ws(t.p[0])) { t.p++; t.n--; }
    while (t.n && is_ws(t.p[t.n - 1])) t.n--;
    return t;
}

inline bool starts_with(Tok line, const char *prefix) {
    size_t n = std::strlen(prefix);
    return line.n >= n && std::memcmp(line.p, prefix, n) == 0;
}

// str.split(' ').filter(|s| !s.trim().is_empty())   (resource_parser.rs:137-139)
void split_line(Tok line, std::vector<Tok> &out) {
    out.clear();
    size_t start = 0;
    for (size_t i = 0; i <= line.n; i++) {
        if (i == line.n || line.p[i] == ' ') {
            Tok t{line.p + start, i - start};
            if (trim(t).n != 0) out.push_back(t);
            start = i + 1;
        }
    }
}

ParseError arg_count_error(size_t count, Tok line) {
    ParseError e;
    e.kind = ParseErrorKind::InvalidLineArgCount;
    e.detail = "count " + std::to_string(count) + " in line: " + std::string(line.p, line.n);
    return e;
}

bool parse_f32(Tok t, float *out) {
    t = trim(t);
    if (t.n == 0 || t.n > 255) return false;
    char buf[256];
    std::memcpy(buf, t.p, t.n);
    buf[t.n] = 0;
    const char *s = buf;
    if (*s == '+' || *s == '-') s++;
    if (s[0] == '0' && (s[1] == 'x' || s[1] == 'X')) return false;  // Rust rejects hex floats
    if (*s == 0) return false;
    char *end = nullptr;
    errno = 0;
    float v = std::strtof(buf, &end);  // glibc: correctly rounded
    if (end == buf || *end != 0) return false;
    *out = v;
    return true;
}

// IndexBundle::parse (types.rs:177-180)
bool parse_index(Tok t, int32_t *out) {
    t = trim(t);
    if (t.n == 0 || t.n > 31) return false;
    char buf[32];
    std::memcpy(buf, t.p, t.n);
    buf[t.n] = 0;
    const char *s = buf;
    if (*s == '+' || *s == '-') s++;
    if (*s == 0) return false;
    for (const char *c = s; *c; c++) if (*c < '0' || *c > '9') return false;
    errno = 0;
    long long v = std::strtoll(buf, nullptr, 10);
    if (errno || v > std::numeric_limits<int32_t>::max() || v < std::numeric_limits<int32_t>::min()) return false;
    *out = (int32_t)(v > 0 ? v - 1 : v);
    return true;
}

ParseError parse_values(Tok line, size_t min_n, size_t max_n, std::vector<Tok> &toks, float *vals) {
    split_line(line, toks);
    size_t l = toks.size();
    if (l == 0 || l - 1 < min_n || l - 1 > max_n) return arg_count_error(l, line);
    for (size_t i = 1; i < l; i++) {
        if (!parse_f32(toks[i], &vals[i - 1])) {
            ParseError e;
            e.kind = ParseErrorKind::ParseFloatError;  // the reference panics here (unwrap, resource_parser.rs:122)
            e.detail = "invalid float '" + std::string(toks[i].p, toks[i].n) + "'";
            return e;
        }
    }
    return ParseError();
}

ParseError parse_string_line(Tok line, std::vector<Tok> &toks, std::string *out) {
    split_line(line, toks);
    if (toks.size() != 2) return arg_count_error(toks.size(), line);
    Tok v = trim(toks[1]);
    out->assign(v.p, v.n);
    return ParseError();
}

// IndexBundle::new (types.rs:155-161): split on '/' or '\\'
ParseError parse_bundle(Tok corner, IndexBundle *ib) {
    Tok parts[3] = {{nullptr, 0}, {nullptr, 0}, {nullptr, 0}};
    size_t np = 0, start = 0;
    for (size_t i = 0; i <= corner.n; i++) {
        if (i == corner.n || corner.p[i] == '/' || corner.p[i] == '\\') {
            if (np < 3) parts[np] = Tok{corner.p + start, i - start};
            np++;
            start = i + 1;
        }
    }
    int32_t *dst[3] = {&ib->pos_idx, &ib->tex_idx, &ib->nor_idx};
    for (size_t k = 0; k < 3 && k < np; k++) {
        if (parts[k].n == 0) continue;  // `!input[k].is_empty()` (no trim)
        if (!parse_index(parts[k], dst[k])) {
            ParseError e;
            e.kind = ParseErrorKind::ParseIntError;
            e.detail = std::string(parts[k].p, parts[k].n);
            return e;
        }
    }
    return ParseError();
}

}  // namespace

Model ResourceParser::parse_model(const std::string &data, ParseError *err) {
    ModelBuilder mb;
    std::vector<Tok> toks;
    std::vector<IndexBundle> bundles;
    ParseError e;
    uint32_t line_no = 0;
    size_t pos = 0;
    const size_t n = data.size();
    while (pos <= n) {
        size_t nl = data.find('\n', pos);
        if (nl == std::string::npos) nl = n;
        Tok line{data.data() + pos, nl - pos};
        line_no++;
        pos = nl + 1;

        if (starts_with(line, "mtllib ")) {
            std::string v;
            if ((e = parse_string_line(line, toks, &v))) break;
            mb.add_material_lib(v);
        }
        if (starts_with(line, "usemtl ")) {
            std::string v;
            if ((e = parse_string_line(line, toks, &v))) break;
            mb.add_material(v);
        }
        if (starts_with(line, "f ")) {
            split_line(line, toks);
            size_t l = toks.size();
            if (l < 4) { e = arg_count_error(l, line); break; }
            bundles.assign(l - 1, IndexBundle());
            bool bad = false;
            for (size_t i = 1; i < l; i++) {
                if ((e = parse_bundle(toks[i], &bundles[i - 1]))) { bad = true; break; }
            }
            if (bad) break;
            size_t num_tris = l - 3;  // polygon fan, resource_parser.rs:130-132
            for (size_t i = 0; i < num_tris; i++) mb.add_indices(bundles[0], bundles[i + 1], bundles[i + 2], mb.get_current_mat());
        }
        float vals[8];
        if (starts_with(line, "v ")) {
            if ((e = parse_values(line, 3, 3, toks, vals))) break;
            mb.add_position(Vec3{vals[0], vals[1], vals[2]});
        }
        if (starts_with(line, "vt ")) {
            if ((e = parse_values(line, 2, 3, toks, vals))) break;
            mb.add_tex_coord(Vec2{vals[0], vals[1]});
        }
        if (starts_with(line, "vn ")) {
            if ((e = parse_values(line, 3, 3, toks, vals))) break;
            mb.add_normal(Vec3{vals[0], vals[1], vals[2]});
        }
        if (nl == n) break;
    }
    if (e) {
        e.line = line_no;
        if (err) *err = e;
        return Model();
    }
    ParseError be;
    Model m = mb.build(&be);
    if (be) { if (err) *err = be; return Model(); }
    if (err) *err = ParseError();
    return m;
}

std::vector<std::pair<std::string, Material>> ResourceParser::parse_material_lib(const std::string &data, ParseError *err) {
    MaterialLibBuilder lb;
    std::vector<Tok> toks;
    ParseError e;
    uint32_t line_no = 0;
    size_t pos = 0;
    const size_t n = data.size();
    auto need_current = [&](Material **m) {
        *m = lb.current();
        if (*m) return true;
        e.kind = ParseErrorKind::NoMaterialNamed;
        e.detail = "property before the first newmtl";
        return false;
    };
    struct ColorProp { const char *prefix; Vec3 Material::*field; };
    static const ColorProp colors[] = {{"Ka ", &Material::ambient_color}, {"Kd ", &Material::diffuse_color},
                                       {"Ks ", &Material::specular_color}, {"Tf ", &Material::transmission_color}};
    struct TexProp { const char *prefix; std::string Material::*name; bool Material::*has; };
    static const TexProp texes[] = {{"map_Ka ", &Material::ambient_tex, &Material::has_ambient_tex},
                                    {"map_Kd ", &Material::diffuse_tex, &Material::has_diffuse_tex},
                                    {"map_Ks ", &Material::specular_tex, &Material::has_specular_tex},
                                    {"map_Ns ", &Material::specular_exp_tex, &Material::has_specular_exp_tex}};
    while (pos <= n && !e) {
        size_t nl = data.find('\n', pos);
        if (nl == std::string::npos) nl = n;
        Tok line{data.data() + pos, nl - pos};
        line_no++;
        pos = nl + 1;
        float vals[8];
        Material *m = nullptr;
        if (starts_with(line, "newmtl ")) {
            std::string v;
            if ((e = parse_string_line(line, toks, &v))) break;
            lb.add_material(v);
        }
        for (const ColorProp &c : colors) {
            if (!starts_with(line, c.prefix)) continue;
            if ((e = parse_values(line, 3, 3, toks, vals))) break;
            if (!need_current(&m)) break;
            m->*(c.field) = Vec3{vals[0], vals[1], vals[2]};
        }
        if (e) break;
        if (starts_with(line, "Ns ")) {
            if ((e = parse_values(line, 1, 1, toks, vals))) break;
            if (!need_current(&m)) break;
            m->specular_exp = vals[0];
        }
        if (starts_with(line, "d ")) {
            if ((e = parse_values(line, 1, 1, toks, vals))) break;
            if (!need_current(&m)) break;
            m->transmission = 1.0f - vals[0];   // resource_parser.rs:78
        }
        if (starts_with(line, "Tr ")) {
            if ((e = parse_values(line, 1, 1, toks, vals))) break;
            if (!need_current(&m)) break;
            m->transmission = vals[0];
        }
        if (starts_with(line, "Ni ")) {
            if ((e = parse_values(line, 1, 1, toks, vals))) break;
            if (!need_current(&m)) break;
            m->optical_density = vals[0];
        }
        for (const TexProp &t : texes) {
            if (!starts_with(line, t.prefix)) continue;
            std::string v;
            if ((e = parse_string_line(line, toks, &v))) break;
            if (!need_current(&m)) break;
            m->*(t.name) = v;
            m->*(t.has) = true;
        }
        if (e) break;
        if (nl == n) break;
    }
    if (e) {
        e.line = line_no;
        if (err) *err = e;
        return {};
    }
    if (err) *err = ParseError();
    return lb.build();
}

}  // namespace host
}  // namespace rtamd
