// PLY mesh loading with the reference's strictness (src/shapes/utils.rs:16-120): ascii,
// binary_little_endian and binary_big_endian bodies (what ply-rs reads); vertex x, y, z, nx, ny, nz
// must be `float` and face.vertex_indices a list of `uint` (any count type), or loading fails with
// the reference's panic text; faces are then validated (3 or 4 indices, quads coplanar and convex).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace host {

struct Mesh {
    std::vector<double> vertices;              // 3 per vertex (f32 file values widened, utils.rs:29-37)
    std::vector<double> normals;               // 3 per vertex, normalised (utils.rs:54)
    std::vector<uint32_t> face_first;          // n_faces + 1 offsets into face_index
    std::vector<uint32_t> face_index;
    size_t n_vertices() const { return vertices.size() / 3; }
    size_t n_faces() const { return face_first.empty() ? 0 : face_first.size() - 1; }
};

namespace ply_detail {

struct Property {
    bool is_list = false;
    std::string name, type, count_type;
};
struct Element {
    std::string name;
    size_t count = 0;
    std::vector<Property> props;
};

inline int type_size(const std::string& t) {
    if (t == "char" || t == "int8" || t == "uchar" || t == "uint8") return 1;
    if (t == "short" || t == "int16" || t == "ushort" || t == "uint16") return 2;
    if (t == "int" || t == "int32" || t == "uint" || t == "uint32" || t == "float" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    return 0;
}
inline bool is_float32(const std::string& t) { return t == "float" || t == "float32"; }
inline bool is_uint32(const std::string& t) { return t == "uint" || t == "uint32"; }

// one scalar of type t from p (already in host byte order) as double
inline double scalar_value(const std::string& t, const unsigned char* p) {
    if (t == "char" || t == "int8") { int8_t v; std::memcpy(&v, p, 1); return v; }
    if (t == "uchar" || t == "uint8") { uint8_t v; std::memcpy(&v, p, 1); return v; }
    if (t == "short" || t == "int16") { int16_t v; std::memcpy(&v, p, 2); return v; }
    if (t == "ushort" || t == "uint16") { uint16_t v; std::memcpy(&v, p, 2); return v; }
    if (t == "int" || t == "int32") { int32_t v; std::memcpy(&v, p, 4); return v; }
    if (t == "uint" || t == "uint32") { uint32_t v; std::memcpy(&v, p, 4); return v; }
    if (t == "float" || t == "float32") { float v; std::memcpy(&v, p, 4); return v; }
    double v; std::memcpy(&v, p, 8); return v;
}

}  // namespace ply_detail

inline Mesh load_ply(const std::string& path) {
    using namespace ply_detail;
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open mesh " + path);
    std::string line;
    if (!std::getline(f, line) || line.substr(0, 3) != "ply") throw std::runtime_error(path + ": not a PLY file");
    std::string format;
    std::vector<Element> elements;
    for (;;) {
        if (!std::getline(f, line)) throw std::runtime_error(path + ": unexpected end of header");
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line);
        std::string tok;
        if (!(ls >> tok) || tok == "comment" || tok == "obj_info") continue;
        if (tok == "format") {
            ls >> format;
        } else if (tok == "element") {
            Element e;
            ls >> e.name >> e.count;
            elements.push_back(e);
        } else if (tok == "property") {
            if (elements.empty()) throw std::runtime_error(path + ": property before element");
            Property p;
            std::string t;
            ls >> t;
            if (t == "list") {
                p.is_list = true;
                ls >> p.count_type >> p.type >> p.name;
                if (!type_size(p.count_type)) throw std::runtime_error(path + ": unknown PLY type " + p.count_type);
            } else {
                p.type = t;
                ls >> p.name;
            }
            if (!type_size(p.type)) throw std::runtime_error(path + ": unknown PLY type " + p.type);
            elements.back().props.push_back(p);
        } else if (tok == "end_header") {
            break;
        }
    }
    const bool ascii = format == "ascii";
    const bool big = format == "binary_big_endian";
    if (!ascii && !big && format != "binary_little_endian")
        throw std::runtime_error(path + ": unsupported PLY format " + format);

    // the reference's type checks (utils.rs:28-64)
    const Element* ve = nullptr;
    const Element* fe = nullptr;
    for (const Element& e : elements) {
        if (e.name == "vertex") ve = &e;
        if (e.name == "face") fe = &e;
    }
    if (!ve) throw std::runtime_error(path + ": no vertex element");
    if (!fe) throw std::runtime_error(path + ": no face element");
    const char* need[6] = {"x", "y", "z", "nx", "ny", "nz"};
    int vcol[6];
    for (int k = 0; k < 6; ++k) {
        vcol[k] = -1;
        for (size_t j = 0; j < ve->props.size(); ++j)
            if (ve->props[j].name == need[k]) vcol[k] = (int)j;
        if (vcol[k] < 0) throw std::runtime_error(path + ": vertex property " + need[k] + " missing");
        const Property& p = ve->props[(size_t)vcol[k]];
        if (p.is_list || !is_float32(p.type)) throw std::runtime_error(std::string(need[k]) + "'s type unrecognized");
    }
    int fcol = -1;
    for (size_t j = 0; j < fe->props.size(); ++j)
        if (fe->props[j].name == "vertex_indices") fcol = (int)j;
    if (fcol < 0) throw std::runtime_error(path + ": face property vertex_indices missing");
    if (!fe->props[(size_t)fcol].is_list || !is_uint32(fe->props[(size_t)fcol].type))
        throw std::runtime_error("vertex_indices's type unrecognized");

    Mesh mesh;
    mesh.face_first.push_back(0);
    auto read_bytes = [&](unsigned char* dst, int n) {
        f.read(reinterpret_cast<char*>(dst), n);
        if (f.gcount() != n) throw std::runtime_error(path + ": unexpected end of data");
        if (big)
            for (int a = 0, b = n - 1; a < b; ++a, --b) std::swap(dst[a], dst[b]);
    };
    for (const Element& e : elements) {
        const bool is_v = &e == ve, is_f = &e == fe;
        if (is_v) {
            mesh.vertices.reserve(e.count * 3);
            mesh.normals.reserve(e.count * 3);
        }
        std::vector<double> row(e.props.size());
        for (size_t r = 0; r < e.count; ++r) {
            std::istringstream ls;
            if (ascii) {
                if (!std::getline(f, line)) throw std::runtime_error(path + ": unexpected end of data");
                ls.str(line);
            }
            for (size_t j = 0; j < e.props.size(); ++j) {
                const Property& p = e.props[j];
                if (!p.is_list) {
                    double v;
                    if (ascii) {
                        if (!(ls >> v)) throw std::runtime_error(path + ": malformed ascii row");
                        if (is_float32(p.type)) v = (double)(float)v;  // the file value is an f32
                    } else {
                        unsigned char buf[8];
                        read_bytes(buf, type_size(p.type));
                        v = scalar_value(p.type, buf);
                    }
                    row[j] = v;
                } else {
                    double n;
                    if (ascii) {
                        if (!(ls >> n)) throw std::runtime_error(path + ": malformed ascii row");
                    } else {
                        unsigned char buf[8];
                        read_bytes(buf, type_size(p.count_type));
                        n = scalar_value(p.count_type, buf);
                    }
                    if (n < 0 || n > 1e6) throw std::runtime_error(path + ": implausible list length");
                    for (int k = 0; k < (int)n; ++k) {
                        double v;
                        if (ascii) {
                            if (!(ls >> v)) throw std::runtime_error(path + ": malformed ascii row");
                        } else {
                            unsigned char buf[8];
                            read_bytes(buf, type_size(p.type));
                            v = scalar_value(p.type, buf);
                        }
                        if (is_f && (int)j == fcol) mesh.face_index.push_back((uint32_t)v);
                    }
                    if (is_f && (int)j == fcol) mesh.face_first.push_back((uint32_t)mesh.face_index.size());
                }
            }
            if (is_v) {
                for (int k = 0; k < 3; ++k) mesh.vertices.push_back(row[(size_t)vcol[k]]);
                double n[3] = {row[(size_t)vcol[3]], row[(size_t)vcol[4]], row[(size_t)vcol[5]]};
                double inv = 1.0 / std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);  // cgmath normalize
                for (int k = 0; k < 3; ++k) mesh.normals.push_back(n[k] * inv);
            }
        }
    }
    return mesh;
}

// src/shapes/quadrilateral.rs:23-54
struct P3 {
    double x, y, z;
};
inline P3 sub(P3 a, P3 b) { return P3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline P3 cross(P3 a, P3 b) { return P3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double dot(P3 a, P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline bool are_points_coplanar(P3 v0, P3 v1, P3 v2, P3 v3) {
    P3 n1 = cross(sub(v1, v0), sub(v2, v0)), n2 = cross(sub(v2, v0), sub(v3, v0));
    double c = dot(n1, n2) / (std::sqrt(dot(n1, n1)) * std::sqrt(dot(n2, n2)));
    return std::fabs(std::fabs(c) - 1.0) < 1e-3;
}
inline bool is_quadrilateral_convex(P3 v0, P3 v1, P3 v2, P3 v3) {
    if (!are_points_coplanar(v0, v1, v2, v3)) return false;
    P3 e01 = sub(v1, v0), e12 = sub(v2, v1), e23 = sub(v3, v2), e30 = sub(v0, v3);
    P3 c1 = cross(e01, e12), c2 = cross(e12, e23), c3 = cross(e23, e30), c4 = cross(e30, e01);
    return dot(c1, c2) >= 0.0 && dot(c2, c3) >= 0.0 && dot(c3, c4) >= 0.0 && dot(c4, c1) >= 0.0;
}

// load_mesh (src/shapes/utils.rs:84-120).  A face with more than four indices passes the
// reference's validation and panics at the first intersection (src/shapes/mesh.rs:75); here it is
// rejected at load time with that message.  An index past the vertex array is the reference's
// out-of-bounds panic.
inline Mesh load_mesh(const std::string& path) {
    size_t dot_pos = path.rfind('.');
    if (dot_pos == std::string::npos || path.substr(dot_pos + 1) != "ply")
        throw std::runtime_error("Unsupported mesh format: " + path);
    Mesh m = load_ply(path);
    for (size_t fi = 0; fi < m.n_faces(); ++fi) {
        const uint32_t b = m.face_first[fi], n = m.face_first[fi + 1] - b;
        if (n < 3) throw std::runtime_error("Invalid mesh: face " + std::to_string(fi) + " has fewer than 3 indices");
        if (n > 4) throw std::runtime_error("Mesh with non-triangle or non-quadrilateral face is not supported");
        P3 v[4];
        for (uint32_t k = 0; k < n; ++k) {
            uint32_t idx = m.face_index[b + k];
            if (idx >= m.n_vertices()) throw std::runtime_error("Invalid mesh: vertex index out of range");
            v[k] = P3{m.vertices[3 * idx], m.vertices[3 * idx + 1], m.vertices[3 * idx + 2]};
        }
        if (n == 4) {
            if (!are_points_coplanar(v[0], v[1], v[2], v[3]))
                throw std::runtime_error("Invalid mesh: face " + std::to_string(fi) + ", Reason: coplanar");
            if (!is_quadrilateral_convex(v[0], v[1], v[2], v[3]))
                throw std::runtime_error("Invalid mesh: face " + std::to_string(fi) + ", Reason: non-convex");
        }
    }
    return m;
}

}  // namespace host
