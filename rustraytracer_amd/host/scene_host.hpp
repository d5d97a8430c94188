// Scene and render configuration of the reference (TOML), flattened for the C ABI of the backend.
//
// Mirrors the serde structs of the reference (file:line in nero19960329/RustRayTracer):
//   SceneConfig / ObjectConfig      src/scene.rs:13-33, src/object.rs:26-39
//   CameraConfig (Perspective)      src/camera.rs:48-75
//   ShapeConfig (Sphere / Plane / Triangle / Quadrilateral / Mesh)   src/shapes/shape.rs:16-36
//   MaterialConfig (5 variants)     src/material.rs:286-315
//   RenderConfig and its tables     src/renderer.rs:12-37
//   SamplerConfig / TracerConfig    src/sampler.rs:168-187, src/tracers/tracer.rs:11-24
//   Matrix4DConfig (column-major)   src/math.rs:36-63
// Unknown keys are ignored, as serde does without deny_unknown_fields.  One rrt_prim per primitive in
// the order the reference's linear scan visits them, every `transform` applied the way each shape's
// `transform` method applies it.
#pragma once
#include <cmath>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rrt.h"
#include "ply_loader.hpp"
#include "toml_lite.hpp"

namespace host {

using toml_lite::Value;

struct ConfigError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

inline std::string read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw ConfigError("cannot open " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

inline const Value& need(const Value& table, const std::string& key, const std::string& where) {
    const Value* v = table.find(key);
    if (!v) throw ConfigError(where + ": missing field `" + key + "`");
    return *v;
}
inline double need_f64(const Value& table, const std::string& key, const std::string& where) {
    const Value& v = need(table, key, where);
    if (!v.is_number()) throw ConfigError(where + "." + key + ": expected a number");
    return v.number();
}
inline long long need_int(const Value& table, const std::string& key, const std::string& where) {
    const Value& v = need(table, key, where);
    if (v.kind != Value::INT) throw ConfigError(where + "." + key + ": expected an integer");
    return v.i;
}
inline uint32_t need_u32(const Value& table, const std::string& key, const std::string& where) {
    long long i = need_int(table, key, where);
    if (i < 0 || i > 0xffffffffll) throw ConfigError(where + "." + key + ": out of range for u32");
    return (uint32_t)i;
}
inline std::string need_str(const Value& table, const std::string& key, const std::string& where) {
    const Value& v = need(table, key, where);
    if (v.kind != Value::STRING) throw ConfigError(where + "." + key + ": expected a string");
    return v.s;
}
inline P3 vec3(const Value& v, const std::string& where) {
    if (v.kind != Value::TABLE) throw ConfigError(where + ": expected a table {x, y, z}");
    return P3{need_f64(v, "x", where), need_f64(v, "y", where), need_f64(v, "z", where)};
}

// cgmath Matrix4 as its four columns (Matrix4::new takes column-major arguments, src/math.rs:57-62)
struct Mat4 {
    double c[4][4];  // c[column][row]
};
inline Mat4 matrix_from_config(const Value* t) {
    Mat4 m;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) m.c[c][r] = c == r ? 1.0 : 0.0;  // EYE_MATRIX4D, src/math.rs:65-97
    if (!t) return m;
    if (t->kind != Value::TABLE) throw ConfigError("transform: expected a table");
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r)
            m.c[c][r] = need_f64(*t, "m" + std::to_string(c + 1) + std::to_string(r + 1), "transform");
    return m;
}
// Matrix4 * Vector4 in cgmath: self.x * v.x + self.y * v.y + self.z * v.z + self.w * v.w (columns)
inline void mul(const Mat4& m, const double v[4], double out[4]) {
    for (int r = 0; r < 4; ++r) out[r] = ((m.c[0][r] * v[0] + m.c[1][r] * v[1]) + m.c[2][r] * v[2]) + m.c[3][r] * v[3];
}
inline P3 transform_point3(const Mat4& m, P3 p) {  // src/math.rs:99-110
    double v[4] = {p.x, p.y, p.z, 1.0}, o[4];
    mul(m, v, o);
    if (o[3] == 0.0) throw ConfigError("transform maps a point to w = 0");
    return P3{o[0] / o[3], o[1] / o[3], o[2] / o[3]};
}
inline P3 transform_vec3(const Mat4& m, P3 p) {  // src/math.rs:112-115
    double v[4] = {p.x, p.y, p.z, 0.0}, o[4];
    mul(m, v, o);
    return P3{o[0], o[1], o[2]};
}
inline P3 normalize(P3 v) {
    double inv = 1.0 / std::sqrt(dot(v, v));
    return P3{v.x * inv, v.y * inv, v.z * inv};
}

struct SceneData {
    rrt_camera camera;
    std::vector<rrt_prim> prims;
    std::vector<rrt_material> materials;
    rrt_scene_desc desc() const {
        rrt_scene_desc d;
        d.camera = camera;
        d.n_prims = (uint32_t)prims.size();
        d.n_objects = (uint32_t)materials.size();
        d.prims = prims.data();
        d.materials = materials.data();
        return d;
    }
};

inline bool file_exists(const std::string& p) { return std::ifstream(p).good(); }

// Mesh paths are resolved like the reference does (relative to the working directory), then
// against the given directories (the scene file's own, and the directory two levels above it,
// which is the repository root for configs/<scene>/scene.toml).
inline std::string resolve(const std::string& path, const std::vector<std::string>& base_dirs) {
    if (!path.empty() && path[0] == '/') return path;
    if (file_exists(path)) return path;
    for (const std::string& b : base_dirs) {
        std::string p = b + "/" + path;
        if (file_exists(p)) return p;
    }
    return path;
}

inline void put(rrt_prim& p, int k, P3 v) {
    p.v[3 * k] = v.x;
    p.v[3 * k + 1] = v.y;
    p.v[3 * k + 2] = v.z;
}

inline void add_shape(const Value& shape, uint32_t object_id, const std::vector<std::string>& base_dirs,
                      std::vector<rrt_prim>& out) {
    const std::string type = need_str(shape, "type", "shape");
    const Mat4 m = matrix_from_config(shape.find("transform"));
    rrt_prim p;
    std::memset(&p, 0, sizeof(p));
    p.object_id = object_id;
    if (type == "Sphere") {  // src/shapes/sphere.rs:104-119: only the centre is transformed
        p.type = RRT_PRIM_SPHERE;
        put(p, 0, transform_point3(m, vec3(need(shape, "center", "Sphere"), "center")));
        p.v[3] = need_f64(shape, "radius", "Sphere");
        out.push_back(p);
    } else if (type == "Plane") {  // src/shapes/plane.rs:46-61
        p.type = RRT_PRIM_PLANE;
        P3 n = normalize(vec3(need(shape, "normal", "Plane"), "normal"));
        put(p, 0, transform_point3(m, vec3(need(shape, "point", "Plane"), "point")));
        put(p, 1, normalize(transform_vec3(m, n)));
        out.push_back(p);
    } else if (type == "Triangle" || type == "Quadrilateral") {  // triangle.rs:90-111, quadrilateral.rs:192-215
        const int k = type == "Triangle" ? 3 : 4;
        const Value& vs = need(shape, "vertices", type);
        if (vs.kind != Value::ARRAY || (int)vs.a.size() != k)
            throw ConfigError(type + ": expected " + std::to_string(k) + " vertices");
        p.type = k == 3 ? RRT_PRIM_TRIANGLE : RRT_PRIM_QUAD;
        for (int i = 0; i < k; ++i) put(p, i, transform_point3(m, vec3(vs.a[(size_t)i], "vertex")));
        out.push_back(p);
    } else if (type == "Mesh") {  // src/shapes/mesh.rs:91-114
        Mesh mesh = load_mesh(resolve(need_str(shape, "file", "Mesh"), base_dirs));
        std::vector<P3> tv(mesh.n_vertices());
        for (size_t i = 0; i < tv.size(); ++i)
            tv[i] = transform_point3(m, P3{mesh.vertices[3 * i], mesh.vertices[3 * i + 1], mesh.vertices[3 * i + 2]});
        out.reserve(out.size() + mesh.n_faces());
        for (size_t fi = 0; fi < mesh.n_faces(); ++fi) {
            const uint32_t b = mesh.face_first[fi], n = mesh.face_first[fi + 1] - b;
            std::memset(p.v, 0, sizeof(p.v));
            p.type = n == 3 ? RRT_PRIM_TRIANGLE : RRT_PRIM_QUAD;
            p.prim_id = (uint32_t)fi;
            for (uint32_t k = 0; k < n; ++k) put(p, (int)k, tv[mesh.face_index[b + k]]);
            out.push_back(p);
        }
    } else {
        throw ConfigError("unknown shape type \"" + type + "\"");
    }
}

inline rrt_material material(const Value& mat) {
    rrt_material m;
    std::memset(&m, 0, sizeof(m));
    const std::string type = need_str(mat, "type", "material");
    auto rgb = [&m](P3 c) { m.rgb[0] = c.x; m.rgb[1] = c.y; m.rgb[2] = c.z; };
    if (type == "Emissive") {
        m.type = RRT_MAT_EMISSIVE;
        rgb(vec3(need(mat, "color", type), "color"));
    } else if (type == "Lambertian") {
        m.type = RRT_MAT_LAMBERTIAN;
        rgb(vec3(need(mat, "albedo", type), "albedo"));
    } else if (type == "PhongSpecular") {
        m.type = RRT_MAT_PHONG;
        rgb(vec3(need(mat, "specular", type), "specular"));
        m.param = need_f64(mat, "shininess", type);
    } else if (type == "IdealReflector") {
        m.type = RRT_MAT_REFLECTOR;
    } else if (type == "IdealDielectric") {
        m.type = RRT_MAT_DIELECTRIC;
        m.param = need_f64(mat, "ior", type);
    } else {
        throw ConfigError("unknown material type \"" + type + "\"");
    }
    return m;
}

// Scene::from_config (src/scene.rs:20-33)
inline SceneData load_scene(const std::string& path) {
    Value cfg = toml_lite::parse(read_file(path));
    SceneData sc;
    const Value& cam = need(cfg, "camera", "scene");
    if (need_str(cam, "type", "camera") != "Perspective") throw ConfigError("camera: type must be \"Perspective\"");
    P3 from = vec3(need(cam, "look_from", "camera"), "look_from"), at = vec3(need(cam, "look_at", "camera"), "look_at"),
       up = vec3(need(cam, "vup", "camera"), "vup");
    const double f[3] = {from.x, from.y, from.z}, a[3] = {at.x, at.y, at.z}, u[3] = {up.x, up.y, up.z};
    if (rrt_camera_perspective(f, a, u, need_f64(cam, "vfov", "camera"), need_f64(cam, "aspect", "camera"), &sc.camera) != RRT_OK)
        throw ConfigError(std::string("camera: ") + rrt_last_error());
    std::vector<std::string> base_dirs;
    size_t slash = path.rfind('/');
    std::string dir = slash == std::string::npos ? "." : path.substr(0, slash);
    base_dirs.push_back(dir);
    base_dirs.push_back(dir + "/../..");
    if (const Value* objs = cfg.find("objects")) {
        if (objs->kind != Value::ARRAY) throw ConfigError("objects: expected an array of tables");
        for (size_t i = 0; i < objs->a.size(); ++i) {
            const Value& o = objs->a[i];
            add_shape(need(o, "shape", "object"), (uint32_t)i, base_dirs, sc.prims);
            sc.materials.push_back(material(need(o, "material", "object")));
        }
    }
    return sc;
}

struct RenderSettings {
    rrt_render_desc render;
    rrt_post_desc post;
    long long parallelism = 0;
};

// RenderConfig (src/renderer.rs:12-37).  `seed` is an extension key (serde ignores unknown keys).
inline RenderSettings load_render_config(const std::string& path) {
    Value cfg = toml_lite::parse(read_file(path));
    RenderSettings rs;
    std::memset(&rs.render, 0, sizeof(rs.render));
    std::memset(&rs.post, 0, sizeof(rs.post));
    const Value &tr = need(cfg, "tracer", "render config"), &im = need(cfg, "image", "render config"),
                &sa = need(cfg, "sampler", "render config"), &pp = need(cfg, "post_processing", "render config"),
                &perf = need(cfg, "performance", "render config");
    if (need_str(tr, "type", "tracer") != "mcpt") throw ConfigError("tracer.type must be \"mcpt\"");
    rs.render.min_depth = need_u32(tr, "min_depth", "tracer");
    rs.render.max_depth = need_u32(tr, "max_depth", "tracer");
    rs.render.width = need_u32(im, "width", "image");
    rs.render.height = need_u32(im, "height", "image");
    const std::string st = need_str(sa, "type", "sampler");
    rs.render.spp = need_u32(sa, "samples_per_pixel", "sampler");
    if (st == "Random") {
        rs.render.sampler_type = RRT_SAMPLER_RANDOM;
        rs.render.sample_end = (uint64_t)rs.render.spp + 1;  // RandomSampler traces spp + 1 samples, src/sampler.rs:48-55
    } else if (st == "Stratified") {
        rs.render.sampler_type = RRT_SAMPLER_STRATIFIED;
        rs.render.x_strata = need_u32(sa, "x_strata", "sampler");
        rs.render.y_strata = need_u32(sa, "y_strata", "sampler");
        if ((uint64_t)rs.render.x_strata * rs.render.y_strata != rs.render.spp)
            throw ConfigError("x_strata * y_strata != samples_per_pixel");  // src/sampler.rs:87-91
        rs.render.sample_end = rs.render.spp;
    } else {
        throw ConfigError("unknown sampler type \"" + st + "\"");
    }
    if (const Value* tm = pp.find("tone_mapping")) {  // Option<String>; unknown names are the identity, renderer.rs:52-58
        if (tm->kind != Value::STRING) throw ConfigError("post_processing.tone_mapping: expected a string");
        rs.post.tone_mapping = tm->s == "reinhard" ? 1 : 0;
    }
    const Value& g = need(pp, "gamma_correction", "post_processing");
    if (g.kind != Value::BOOL) throw ConfigError("post_processing.gamma_correction: expected a boolean");
    rs.post.gamma_correction = g.b ? 1 : 0;
    if (const Value* wb = pp.find("white_balance")) {
        P3 w = vec3(*wb, "white_balance");
        rs.post.has_white_balance = 1;
        rs.post.white_balance[0] = w.x;
        rs.post.white_balance[1] = w.y;
        rs.post.white_balance[2] = w.z;
    }
    if (const Value* par = perf.find("parallelism")) {
        if (par->kind != Value::INT) throw ConfigError("performance.parallelism: expected an integer");
        rs.parallelism = par->i;
    }
    if (const Value* seed = cfg.find("seed")) {
        if (seed->kind != Value::INT) throw ConfigError("seed: expected an integer");
        rs.render.seed = (uint64_t)seed->i;
    }
    return rs;
}

}  // namespace host
