// pt_host.hpp -- the reference's host-side scene code in C++: OBJ parsing with parse-obj.js semantics, TRS
// matrices and transformed boxes with transform-utils.js arithmetic (doubles, rounded to f32 only where the reference
// rounds them).  Header-only; used by pt_cli.cc.  Pinned by the same golden vectors as the JS and Python hosts
// (tests/golden/host_fixtures.json, produced by running the reference's own files under Node).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

namespace pt_host {

struct Bounds { double mn[3], mx[3]; };
struct ParsedObj { std::vector<double> tris; Bounds bounds; int64_t num_triangles() const { return (int64_t)tris.size() / 12; } };

inline bool js_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

// JS parseFloat: longest numeric prefix after leading whitespace, else NaN (no hex, no "nan", no "inf")
inline double js_parse_float(const std::string& s) {
    size_t i = 0, n = s.size();
    while (i < n && js_space(s[i])) i++;
    size_t b = i;
    if (i < n && (s[i] == '+' || s[i] == '-')) i++;
    if (s.compare(i, 8, "Infinity") == 0) return s[b] == '-' ? -INFINITY : INFINITY;
    size_t d0 = i;
    while (i < n && s[i] >= '0' && s[i] <= '9') i++;
    bool int_digits = i > d0, frac_digits = false;
    if (i < n && s[i] == '.') {
        size_t f0 = ++i;
        while (i < n && s[i] >= '0' && s[i] <= '9') i++;
        frac_digits = i > f0;
        if (!int_digits && !frac_digits) return NAN;
    } else if (!int_digits) return NAN;
    if (i < n && (s[i] == 'e' || s[i] == 'E')) {
        size_t e = i + 1;
        if (e < n && (s[e] == '+' || s[e] == '-')) e++;
        size_t e0 = e;
        while (e < n && s[e] >= '0' && s[e] <= '9') e++;
        if (e > e0) i = e;
    }
    return std::strtod(s.substr(b, i - b).c_str(), nullptr);
}
// JS parseInt(s): optional sign + decimal digits prefix; ok = false for NaN
inline long long js_parse_int(const std::string& s, bool& ok) {
    size_t i = 0, n = s.size();
    while (i < n && js_space(s[i])) i++;
    size_t b = i;
    if (i < n && (s[i] == '+' || s[i] == '-')) i++;
    size_t d0 = i;
    while (i < n && s[i] >= '0' && s[i] <= '9') i++;
    ok = i > d0;
    return ok ? std::strtoll(s.substr(b, i - b).c_str(), nullptr, 10) : 0;
}
inline double js_min(double a, double b) { return (a != a || b != b) ? NAN : (a < b ? a : b); }   // Math.min propagates NaN
inline double js_max(double a, double b) { return (a != a || b != b) ? NAN : (a > b ? a : b); }

// line.split(/\s+/)
inline std::vector<std::string> split_ws(const std::string& line) {
    std::vector<std::string> out; size_t i = 0, n = line.size();
    for (;;) {
        size_t j = i;
        while (j < n && !js_space(line[j])) j++;
        out.push_back(line.substr(i, j - i));
        if (j >= n) break;
        while (j < n && js_space(line[j])) j++;
        i = j;
        if (i >= n) { out.push_back(""); break; }
    }
    return out;
}

// parse-obj.js:1-90.  Only lines starting with v / f / o matter; 1-based indices through a sentinel vertex; parseInt
// semantics ("3/1/2" -> 3); fan triangulation; "o" resets the vertex list; bounds over every v line; pad floats 3.1415.
inline ParsedObj parse_obj(const std::string& text) {
    ParsedObj r;
    for (int k = 0; k < 3; k++) { r.bounds.mn[k] = 1e30; r.bounds.mx[k] = -1e30; }
    std::vector<double> verts = {1e30, 1e30, 1e30};
    size_t pos = 0;
    for (;;) {
        size_t nl = text.find('\n', pos);
        std::string line = text.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
        if (line.size() > 1 && (line[0] == 'v' || line[0] == 'f' || line[0] == 'o')) {
            std::vector<std::string> tok = split_ws(line);
            if (tok[0] == "v") {
                for (int k = 0; k < 3; k++) {
                    double v = (size_t)(k + 1) < tok.size() ? js_parse_float(tok[k + 1]) : NAN;
                    r.bounds.mn[k] = js_min(r.bounds.mn[k], v); r.bounds.mx[k] = js_max(r.bounds.mx[k], v);
                    verts.push_back(v);
                }
            } else if (tok[0] == "f") {
                std::vector<long long> idx;
                for (size_t t = 1; t < tok.size(); t++) { bool ok; long long i = js_parse_int(tok[t], ok); if (ok) idx.push_back(i); }
                auto vert = [&](long long i, int k) -> double {
                    long long at = 3 * i + k;
                    return (at >= 0 && at < (long long)verts.size()) ? verts[(size_t)at] : NAN;
                };
                for (size_t i = 1; i + 1 < idx.size(); i++) {
                    const long long tri[3] = {idx[0], idx[i], idx[i + 1]};
                    for (int v = 0; v < 3; v++) { for (int k = 0; k < 3; k++) r.tris.push_back(vert(tri[v], k)); r.tris.push_back(3.1415); }
                }
            } else if (tok[0] == "o") {
                verts.assign({1e30, 1e30, 1e30});
            }
        }
        if (nl == std::string::npos) break;
        pos = nl + 1;
    }
    return r;
}

// ---- transform-utils.js (column-major 4x4 in doubles) --------------------------------------------------------------------
struct Mat4 { double m[16]; };
inline Mat4 mul4(const Mat4& a, const Mat4& b) {          // :61-80
    Mat4 o;
    for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++)
        o.m[4 * c + r] = a.m[r] * b.m[4 * c] + a.m[4 + r] * b.m[4 * c + 1] + a.m[8 + r] * b.m[4 * c + 2] + a.m[12 + r] * b.m[4 * c + 3];
    return o;
}
inline Mat4 translate(const double p[3]) { return Mat4{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, p[0], p[1], p[2], 1}}; }
inline Mat4 scale(const double s[3]) { return Mat4{{s[0], 0, 0, 0, 0, s[1], 0, 0, 0, 0, s[2], 0, 0, 0, 0, 1}}; }
inline void rot_xyz(const double rot[3], Mat4& rx, Mat4& ry, Mat4& rz) {     // :100-137 (and :139-176 with negated angles)
    double c = std::cos(rot[0]), s = std::sin(rot[0]);
    rx = Mat4{{1, 0, 0, 0, 0, c, s, 0, 0, -s, c, 0, 0, 0, 0, 1}};
    c = std::cos(rot[1]); s = std::sin(rot[1]);
    ry = Mat4{{c, 0, -s, 0, 0, 1, 0, 0, s, 0, c, 0, 0, 0, 0, 1}};
    c = std::cos(rot[2]); s = std::sin(rot[2]);
    rz = Mat4{{c, s, 0, 0, -s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
}
struct TransformMatrices { Mat4 l2w, w2l; };
// :45-59: L2W = T * (Rz*Ry*Rx) * S ; W2L = S^-1 * (Rx^-1*Ry^-1*Rz^-1) * T^-1
inline TransformMatrices compute_transform_matrices(const double position[3], const double rotation[3], const double scl[3]) {
    Mat4 rx, ry, rz, irx, iry, irz;
    rot_xyz(rotation, rx, ry, rz);
    const double neg[3] = {-rotation[0], -rotation[1], -rotation[2]};
    rot_xyz(neg, irx, iry, irz);
    const double inv_s[3] = {1.0 / scl[0], 1.0 / scl[1], 1.0 / scl[2]}, neg_p[3] = {-position[0], -position[1], -position[2]};
    TransformMatrices t;
    t.l2w = mul4(translate(position), mul4(mul4(rz, mul4(ry, rx)), scale(scl)));
    t.w2l = mul4(scale(inv_s), mul4(mul4(irx, mul4(iry, irz)), translate(neg_p)));
    return t;
}
// :11-43: box of the 8 transformed corners
inline Bounds get_transformed_bounding_box(const Mat4& l2w, const Bounds& b) {
    const double* m = l2w.m;
    const double pts[8][3] = {{b.mn[0], b.mn[1], b.mn[2]}, {b.mx[0], b.mn[1], b.mn[2]}, {b.mn[0], b.mx[1], b.mn[2]}, {b.mn[0], b.mn[1], b.mx[2]},
                              {b.mx[0], b.mn[1], b.mx[2]}, {b.mx[0], b.mx[1], b.mn[2]}, {b.mn[0], b.mx[1], b.mx[2]}, {b.mx[0], b.mx[1], b.mx[2]}};
    Bounds o;
    for (int k = 0; k < 3; k++) { o.mn[k] = 1e30; o.mx[k] = -1e30; }
    for (const auto& p : pts)
        for (int r = 0; r < 3; r++) {
            double q = m[r] * p[0] + m[4 + r] * p[1] + m[8 + r] * p[2] + m[12 + r];
            o.mn[r] = js_min(o.mn[r], q); o.mx[r] = js_max(o.mx[r], q);
        }
    return o;
}

}  // namespace pt_host
