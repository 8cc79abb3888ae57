#include "skeleton_host.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <vector>

namespace evm {
namespace {

constexpr float kMargin = 0.04f;        // Bullet CONVEX_DISTANCE_MARGIN
constexpr float kBreaking = 0.02f;      // Bullet gContactBreakingThreshold
constexpr float kDt = 1.f / 60.f;       // DELTA_T_MODEL, evo_motion_model/src/constants.h.in:8
constexpr float kEps = 1.1920928955078125e-7f;
constexpr float kPi = 3.1415926535897932384626433832795029f;

struct Mat {  // rows of a 3x3 + origin (an affine glm::mat4 / btTransform)
    float m[3][3];
    float o[3];
};

struct RawMember { std::string name, shape; float mass, friction, t[3], q[4], scale[3]; int ignore; };
struct RawCon { int type; std::string name, parent, child; float v[20]; };
struct RawMuscle { std::string name, a, b; float mass, scale[3], pa[3], pb[3], force, speed; };
struct RawShape { std::string name; std::vector<float> pts; };

bool next_float(std::istringstream &ss, float &f) {
    std::string tok;
    if (!(ss >> tok)) return false;
    f = strtof(tok.c_str(), nullptr);
    return true;
}
bool next_floats(std::istringstream &ss, float *dst, int n) {
    for (int i = 0; i < n; i++)
        if (!next_float(ss, dst[i])) return false;
    return true;
}

// glm::mat3_cast on a possibly non-unit quaternion (w,x,y,z), returned as rows.
void quat_to_rows(const float q[4], float m[3][3]) {
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float xx = x * x, yy = y * y, zz = z * z, xz = x * z, xy = x * y, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    m[0][0] = 1.f - 2.f * (yy + zz); m[1][0] = 2.f * (xy + wz); m[2][0] = 2.f * (xz - wy);
    m[0][1] = 2.f * (xy - wz); m[1][1] = 1.f - 2.f * (xx + zz); m[2][1] = 2.f * (yz + wx);
    m[0][2] = 2.f * (xz + wy); m[1][2] = 2.f * (yz - wx); m[2][2] = 1.f - 2.f * (xx + yy);
}
// column c of a row-stored matrix
inline void colv(const float m[3][3], int c, float out[3]) { out[0] = m[0][c]; out[1] = m[1][c]; out[2] = m[2][c]; }
inline float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline void cross3(const float a[3], const float b[3], float o[3]) {
    float r0 = a[1] * b[2] - a[2] * b[1], r1 = a[2] * b[0] - a[0] * b[2], r2 = a[0] * b[1] - a[1] * b[0];
    o[0] = r0; o[1] = r1; o[2] = r2;
}
float norm_angle(float a) {
    a = fmodf(a, 2.f * kPi);
    if (a < -kPi) return a + 2.f * kPi;
    if (a > kPi) return a - 2.f * kPi;
    return a;
}
void plane_space(const float n[3], float p[3], float q[3]) {
    if (fabsf(n[2]) > 0.7071067811865475244008443621048490f) {
        float a = n[1] * n[1] + n[2] * n[2], k = 1.f / sqrtf(a);
        p[0] = 0; p[1] = -n[2] * k; p[2] = n[1] * k;
        q[0] = a * k; q[1] = -n[0] * p[2]; q[2] = n[0] * p[1];
    } else {
        float a = n[0] * n[0] + n[1] * n[1], k = 1.f / sqrtf(a);
        p[0] = -n[1] * k; p[1] = n[0] * k; p[2] = 0;
        q[0] = -n[2] * p[1]; q[1] = n[2] * p[0]; q[2] = a * k;
    }
}
// Bullet quatRotate(q, v) with q = (x,y,z,w)
void quat_rotate(const float q[4], const float v[3], float out[3]) {
    const float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    float tx = qw * v[0] + qy * v[2] - qz * v[1];
    float ty = qw * v[1] + qz * v[0] - qx * v[2];
    float tz = qw * v[2] + qx * v[1] - qy * v[0];
    float tw = -qx * v[0] - qy * v[1] - qz * v[2];
    const float ix = -qx, iy = -qy, iz = -qz, iw = qw;
    out[0] = tw * ix + tx * iw + ty * iz - tz * iy;
    out[1] = tw * iy + ty * iw + tz * ix - tx * iz;
    out[2] = tw * iz + tz * iw + tx * iy - ty * ix;
}

// Box-like inertia of the margin-inflated local AABB + the relative contact breaking threshold
// (btPolyhedralConvexShape::calculateLocalInertia, btCollisionShape::getContactBreakingThreshold).
void shape_properties(const std::vector<float> &pts, const float scale[3], float mass, float inv_inertia[3],
                      float &break_thr, float *aabb_c = nullptr, float *aabb_h = nullptr) {
    float hi[3] = {-1e18f, -1e18f, -1e18f}, lo[3] = {1e18f, 1e18f, 1e18f};
    for (size_t i = 0; i + 2 < pts.size(); i += 3)
        for (int a = 0; a < 3; a++) {
            float s = pts[i + a] * scale[a];
            if (s > hi[a]) hi[a] = s;
            if (s < lo[a]) lo[a] = s;
        }
    float amin[3], amax[3];
    for (int a = 0; a < 3; a++) {
        float lmax = hi[a] + kMargin, lmin = lo[a] - kMargin;  // cached local AABB
        float half = 0.5f * (lmax - lmin);
        half += kMargin;                                       // getAabb adds the margin again
        float centre = 0.5f * (lmax + lmin);
        amin[a] = centre - half;
        amax[a] = centre + half;
        // btTransformAabb(localMin, localMax, margin, trans) works on exactly these; btCollisionWorld::updateSingleAabb
        // fattens the world box by gContactBreakingThreshold, which commutes with the transform when added to the extents
        if (aabb_c) aabb_c[a] = centre;
        if (aabb_h) aabb_h[a] = half;
    }
    float l[3];
    for (int a = 0; a < 3; a++) l[a] = 2.f * ((amax[a] - amin[a]) * 0.5f + kMargin);
    const float x2 = l[0] * l[0], y2 = l[1] * l[1], z2 = l[2] * l[2];
    const float sm = mass * 0.08333333f;
    const float in[3] = {sm * (y2 + z2), sm * (x2 + z2), sm * (x2 + y2)};
    for (int a = 0; a < 3; a++) inv_inertia[a] = in[a] != 0.f ? 1.f / in[a] : 0.f;
    float d[3] = {amax[0] - amin[0], amax[1] - amin[1], amax[2] - amin[2]};
    float c[3] = {(amin[0] + amax[0]) * 0.5f, (amin[1] + amax[1]) * 0.5f, (amin[2] + amax[2]) * 0.5f};
    float radius = sqrtf(dot3(d, d)) * 0.5f;
    break_thr = (radius + sqrtf(dot3(c, c))) * kBreaking;
}

void copy_rows(const float m[3][3], float out[9]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) out[3 * i + j] = m[i][j];
}

// ---- input formats --------------------------------------------------------------------------------------------
struct RawSkeleton {
    std::string root_name;
    std::vector<RawMember> members;
    std::vector<RawCon> cons;
    std::vector<RawMuscle> muscles;
    std::vector<RawShape> shapes;
};

// (1) the decoded text fixture written by tests/diag/decode_skeleton.py
int parse_fixture(const char *path, RawSkeleton &R, std::string &err) {
    std::ifstream f(path ? path : "");
    if (!f) { err = std::string("cannot open skeleton fixture: ") + (path ? path : "(null)"); return EVM_E_RUNTIME; }
    std::string &root_name = R.root_name;
    std::vector<RawMember> &members = R.members;
    std::vector<RawCon> &cons = R.cons;
    std::vector<RawMuscle> &muscles = R.muscles;
    std::vector<RawShape> &shapes = R.shapes;
    std::string line;
    int pts_left = 0;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        if (pts_left > 0) {
            float p[3];
            if (!next_floats(ss, p, 3)) { err = "malformed hull point"; return EVM_E_RUNTIME; }
            shapes.back().pts.insert(shapes.back().pts.end(), p, p + 3);
            pts_left--;
            continue;
        }
        std::string kw;
        ss >> kw;
        if (kw == "skeleton") {
            std::string robot, r;
            ss >> robot >> r >> root_name;
        } else if (kw == "member") {
            RawMember m;
            ss >> m.name >> m.shape;
            bool ok = next_float(ss, m.mass) && next_float(ss, m.friction) && next_floats(ss, m.t, 3) &&
                      next_floats(ss, m.q, 4) && next_floats(ss, m.scale, 3);
            if (!ok || !(ss >> m.ignore)) { err = "malformed member line"; return EVM_E_RUNTIME; }
            members.push_back(m);
        } else if (kw == "hinge" || kw == "fixed") {
            RawCon c;
            c.type = kw == "hinge" ? 0 : 1;
            ss >> c.name >> c.parent >> c.child;
            if (!next_floats(ss, c.v, 14)) { err = "malformed constraint line"; return EVM_E_RUNTIME; }
            cons.push_back(c);
        } else if (kw == "muscle") {
            RawMuscle m;
            ss >> m.name >> m.a >> m.b;
            bool ok = next_float(ss, m.mass) && next_floats(ss, m.scale, 3) && next_floats(ss, m.pa, 3) &&
                      next_floats(ss, m.pb, 3) && next_float(ss, m.force) && next_float(ss, m.speed);
            if (!ok) { err = "malformed muscle line"; return EVM_E_RUNTIME; }
            muscles.push_back(m);
        } else if (kw == "shape") {
            RawShape s;
            int n = 0, ndup = 0;
            ss >> s.name >> n >> ndup;
            shapes.push_back(s);
            pts_left = n;
        } else if (kw == "members" || kw == "constraints" || kw == "muscles" || kw == "shapes") {
        } else {
            err = "unknown keyword in skeleton fixture: " + kw;
            return EVM_E_RUNTIME;
        }
    }
    return EVM_OK;
}

// (2) the reference's own skeleton format: JSON written by JsonSerializer (evo_motion_model/src/json_serializer.cpp,
// read back at :113-168; skeleton layout evo_motion_model/src/robot/skeleton.cpp:27-53, member.cpp / constraint.cpp /
// muscle.cpp deserialising constructors) with every float as a 32-character IEEE-754 bit string
// (converter.cpp:138-147), plus the collision hulls from Wavefront OBJ files (shapes.cpp:24-56).
struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal *get(const std::string &k) const {
        for (auto &kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
};
struct JParser {
    const std::string &t;
    size_t i = 0;
    std::string err;
    explicit JParser(const std::string &text) : t(text) {}
    void ws() { while (i < t.size() && (t[i] == ' ' || t[i] == '\n' || t[i] == '\t' || t[i] == '\r')) i++; }
    bool fail(const std::string &m) { if (err.empty()) err = m + " at byte " + std::to_string(i); return false; }
    bool parse_string(std::string &out) {
        if (t[i] != '"') return fail("expected string");
        i++;
        out.clear();
        while (i < t.size() && t[i] != '"') {
            if (t[i] == '\\') {
                if (++i >= t.size()) return fail("bad escape");
                switch (t[i]) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': if (i + 4 >= t.size()) return fail("bad \\u escape"); out += '?'; i += 4; break;
                    default: out += t[i];
                }
                i++;
            } else out += t[i++];
        }
        if (i >= t.size()) return fail("unterminated string");
        i++;
        return true;
    }
    bool parse(JVal &v) {
        ws();
        if (i >= t.size()) return fail("unexpected end");
        const char c = t[i];
        if (c == '{') {
            v.kind = JVal::Obj;
            i++; ws();
            if (i < t.size() && t[i] == '}') { i++; return true; }
            for (;;) {
                ws();
                std::string k;
                if (!parse_string(k)) return false;
                ws();
                if (i >= t.size() || t[i] != ':') return fail("expected ':'");
                i++;
                JVal child;
                if (!parse(child)) return false;
                v.obj.emplace_back(std::move(k), std::move(child));
                ws();
                if (i < t.size() && t[i] == ',') { i++; continue; }
                if (i < t.size() && t[i] == '}') { i++; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.kind = JVal::Arr;
            i++; ws();
            if (i < t.size() && t[i] == ']') { i++; return true; }
            for (;;) {
                JVal child;
                if (!parse(child)) return false;
                v.arr.push_back(std::move(child));
                ws();
                if (i < t.size() && t[i] == ',') { i++; continue; }
                if (i < t.size() && t[i] == ']') { i++; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (c == '"') { v.kind = JVal::Str; return parse_string(v.str); }
        if (t.compare(i, 4, "true") == 0) { v.kind = JVal::Bool; v.b = true; i += 4; return true; }
        if (t.compare(i, 5, "false") == 0) { v.kind = JVal::Bool; v.b = false; i += 5; return true; }
        if (t.compare(i, 4, "null") == 0) { v.kind = JVal::Null; i += 4; return true; }
        char *end = nullptr;
        v.num = strtod(t.c_str() + i, &end);
        if (end == t.c_str() + i) return fail("unexpected character");
        v.kind = JVal::Num;
        i = (size_t) (end - t.c_str());
        return true;
    }
};

// binary_string_to_float (converter.cpp:138-147): 32 characters '0'/'1', most significant bit first
bool bits_to_float(const JVal *v, float &out) {
    if (!v || v->kind != JVal::Str || v->str.size() != 32) return false;
    uint32_t u = 0;
    for (char ch : v->str) {
        if (ch != '0' && ch != '1') return false;
        u = (u << 1) | (uint32_t) (ch == '1');
    }
    memcpy(&out, &u, 4);
    return true;
}
bool read_vec3(const JVal *o, float out[3]) {
    return o && o->kind == JVal::Obj && bits_to_float(o->get("x"), out[0]) && bits_to_float(o->get("y"), out[1]) &&
           bits_to_float(o->get("z"), out[2]);
}
bool read_quat(const JVal *o, float out[4]) {  // (w, x, y, z), the order of JsonDeserializer::read_quat
    return o && o->kind == JVal::Obj && bits_to_float(o->get("w"), out[0]) && bits_to_float(o->get("x"), out[1]) &&
           bits_to_float(o->get("y"), out[2]) && bits_to_float(o->get("z"), out[3]);
}
bool read_str(const JVal *v, std::string &out) {
    if (!v || v->kind != JVal::Str) return false;
    out = v->str;
    return true;
}

// ObjShape (shapes.cpp:24-56): 'v' lines parsed with stof, one hull point per face corner in face order; duplicates
// are then dropped keeping the FIRST occurrence, which preserves every first-strict-extremum support scan over the list.
int load_obj_hull(const std::string &file, RawShape &shape, std::string &err) {
    std::ifstream f(file);
    if (!f) { err = "cannot open hull file: " + file; return EVM_E_RUNTIME; }
    std::vector<float> verts;
    std::vector<int> order;
    std::string line;
    auto split = [](const std::string &s, char d) {
        std::vector<std::string> out;
        std::stringstream ss(s);
        std::string item;
        while (std::getline(ss, item, d)) out.push_back(item);
        return out;
    };
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const auto tok = split(line, ' ');
        if (tok.empty()) continue;
        if (tok[0] == "v") {
            if (tok.size() < 4) { err = "malformed vertex in " + file; return EVM_E_RUNTIME; }
            for (int a = 1; a <= 3; a++) verts.push_back(strtof(tok[a].c_str(), nullptr));
        } else if (tok[0] == "f") {
            if (tok.size() < 4) { err = "malformed face in " + file; return EVM_E_RUNTIME; }
            for (int a = 1; a <= 3; a++) order.push_back(atoi(split(tok[a], '/')[0].c_str()) - 1);
        }
    }
    std::map<std::array<uint32_t, 3>, int> seen;
    for (int idx : order) {
        if (idx < 0 || (size_t) (3 * idx + 2) >= verts.size()) { err = "face index out of range in " + file; return EVM_E_RUNTIME; }
        std::array<uint32_t, 3> key;
        memcpy(key.data(), &verts[3 * idx], 12);
        for (auto &k : key) if (k == 0x80000000u) k = 0;  // -0 and +0 are the same point
        if (seen.emplace(key, 1).second) shape.pts.insert(shape.pts.end(), verts.begin() + 3 * idx, verts.begin() + 3 * idx + 3);
    }
    if (shape.pts.empty()) { err = "no faces in " + file; return EVM_E_RUNTIME; }
    return EVM_OK;
}

int parse_json_skeleton(const char *path, RawSkeleton &R, std::string &err) {
    std::ifstream f(path);
    if (!f) { err = std::string("cannot open skeleton json: ") + path; return EVM_E_RUNTIME; }
    std::stringstream buf;
    buf << f.rdbuf();
    const std::string text = buf.str();
    JParser P(text);
    JVal root;
    if (!P.parse(root) || root.kind != JVal::Obj) { err = "skeleton json: " + (P.err.empty() ? std::string("not an object") : P.err); return EVM_E_RUNTIME; }
    auto bad = [&](const std::string &what) { err = "skeleton json: missing or malformed " + what; return EVM_E_RUNTIME; };
    if (!read_str(root.get("root_name"), R.root_name)) return bad("root_name");
    const JVal *jm = root.get("members"), *jc = root.get("constraints"), *jmu = root.get("muscles");
    if (!jm || jm->kind != JVal::Arr || !jc || jc->kind != JVal::Arr || !jmu || jmu->kind != JVal::Arr) return bad("members / constraints / muscles");
    for (auto &m : jm->arr) {
        RawMember r;
        const JVal *ign = m.get("ignore_collision");
        if (!read_str(m.get("name"), r.name) || !read_str(m.get("shape"), r.shape) || !bits_to_float(m.get("mass"), r.mass) ||
            !bits_to_float(m.get("friction"), r.friction) || !read_vec3(m.get("translation"), r.t) ||
            !read_quat(m.get("rotation"), r.q) || !read_vec3(m.get("scale"), r.scale) || !ign || ign->kind != JVal::Bool)
            return bad("member " + r.name);
        r.ignore = ign->b ? 1 : 0;
        R.members.push_back(r);
    }
    for (auto &c : jc->arr) {
        RawCon r;
        std::string type;
        if (!read_str(c.get("type"), type) || !read_str(c.get("name"), r.name) || !read_str(c.get("parent_name"), r.parent) ||
            !read_str(c.get("child_name"), r.child)) return bad("constraint header");
        if (type == "hinge") {
            r.type = 0;
            const JVal *lim = c.get("limit_radian");
            if (!read_vec3(c.get("pivot_in_parent"), r.v) || !read_vec3(c.get("pivot_in_child"), r.v + 3) ||
                !read_vec3(c.get("axis_in_parent"), r.v + 6) || !read_vec3(c.get("axis_in_child"), r.v + 9) || !lim ||
                !bits_to_float(lim->get("min"), r.v[12]) || !bits_to_float(lim->get("max"), r.v[13])) return bad("hinge " + r.name);
        } else if (type == "fixed") {
            r.type = 1;
            const JVal *fp = c.get("frame_in_parent"), *fc = c.get("frame_in_child");
            if (!fp || !fc || !read_vec3(fp->get("translation"), r.v) || !read_quat(fp->get("rotation"), r.v + 3) ||
                !read_vec3(fc->get("translation"), r.v + 7) || !read_quat(fc->get("rotation"), r.v + 10)) return bad("fixed " + r.name);
        } else {
            err = "skeleton json: unknown constraint type " + type;
            return EVM_E_RUNTIME;
        }
        R.cons.push_back(r);
    }
    for (auto &m : jmu->arr) {
        RawMuscle r;
        if (!read_str(m.get("name"), r.name) || !read_str(m.get("item_a"), r.a) || !read_str(m.get("item_b"), r.b) ||
            !bits_to_float(m.get("attach_mass"), r.mass) || !read_vec3(m.get("attach_scale"), r.scale) ||
            !read_vec3(m.get("pos_in_a"), r.pa) || !read_vec3(m.get("pos_in_b"), r.pb) || !bits_to_float(m.get("force"), r.force) ||
            !bits_to_float(m.get("speed"), r.speed)) return bad("muscle " + r.name);
        R.muscles.push_back(r);
    }
    // hulls: <dir of the json>/../obj/<shape>.obj (the reference's resources layout), else next to the json
    std::string dir = path;
    const size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
    std::vector<std::string> names;
    for (auto &m : R.members) if (std::find(names.begin(), names.end(), m.shape) == names.end()) names.push_back(m.shape);
    if (!R.muscles.empty() && std::find(names.begin(), names.end(), std::string("sphere")) == names.end()) names.push_back("sphere");
    for (auto &n : names) {
        RawShape sh;
        sh.name = n;
        std::string file = dir + "/../obj/" + n + ".obj";
        if (!std::ifstream(file)) file = dir + "/obj/" + n + ".obj";
        if (!std::ifstream(file)) file = dir + "/" + n + ".obj";
        const int rc = load_obj_hull(file, sh, err);
        if (rc) return rc;
        R.shapes.push_back(std::move(sh));
    }
    return EVM_OK;
}

}  // namespace

int load_skeleton_constants(const char *path, const EvmEnvParams &prm, EvmSkelC &S, std::string &err) {
    RawSkeleton RAW;
    {
        const std::string p = path ? path : "";
        const bool is_json = p.size() > 5 && p.compare(p.size() - 5, 5, ".json") == 0;
        const int rc = is_json ? parse_json_skeleton(p.c_str(), RAW, err) : parse_fixture(path, RAW, err);
        if (rc) return rc;
    }
    std::string &root_name = RAW.root_name;
    std::vector<RawMember> &members = RAW.members;
    std::vector<RawCon> &cons = RAW.cons;
    std::vector<RawMuscle> &muscles = RAW.muscles;
    std::vector<RawShape> &shapes = RAW.shapes;
    std::map<std::string, int> member_id, shape_id;
    for (size_t i = 0; i < members.size(); i++) member_id[members[i].name] = (int) i;
    for (size_t i = 0; i < shapes.size(); i++) shape_id[shapes[i].name] = (int) i;
    auto find_member = [&](const std::string &n, int &id) {
        auto it = member_id.find(n);
        if (it == member_id.end()) { err = "Member \"" + n + "\"not found"; return false; }  // skeleton.cpp:58
        id = it->second;
        return true;
    };

    const int nm = (int) members.size(), nmus = (int) muscles.size(), nb = nm + 2 * nmus;
    int nh = 0, nf = 0;
    for (auto &c : cons) (c.type == 0 ? nh : nf)++;
    if (nb > EVM_MAX_BODIES || nm > EVM_MAX_MEMBERS || nh > EVM_MAX_HINGES || nf > EVM_MAX_FIXED ||
        nmus > EVM_MAX_MUSCLES || nm < 1) { err = "skeleton exceeds the compiled capacity"; return EVM_E_UNSUPPORTED; }

    memset(&S, 0, sizeof(S));
    S.nb = nb; S.nm = nm; S.nh = nh; S.nf = nf; S.nmus = nmus;
    if (!find_member(root_name, S.root)) return EVM_E_RUNTIME;
    S.obs_dim = 19 * nm + 4 * nmus;
    S.act_dim = nmus;
    {
        int k = 0;
        S.state_member[k++] = S.root;
        for (int i = 0; i < nm; i++)
            if (i != S.root) S.state_member[k++] = i;
        for (int i = 0; i < nm; i++) S.state_index[S.state_member[i]] = i;
    }
    S.floor_o[0] = 0.f; S.floor_o[1] = -2.f; S.floor_o[2] = 2.f;  // robot_walk.cpp:24
    S.floor_top_y = S.floor_o[1] + 1.0f * 1.f;
    S.root_pos[0] = 1.f; S.root_pos[1] = 0.25f; S.root_pos[2] = 2.f;  // robot_walk.cpp:78
    S.min_vel = prm.minimal_velocity; S.target_vel = prm.target_velocity;
    S.max_steps = (int) (prm.max_episode_seconds / kDt);
    S.init_remaining = (int) (prm.initial_remaining_seconds / kDt);
    S.reset_frames = prm.reset_frames;
    S.env_kind = prm.env_kind;
    S.self_collision = prm.self_collision ? 1 : 0;
    S.settle_steps = prm.env_kind == 1 ? prm.reset_frames : 2 * prm.reset_frames;
    S.reset_angle_limit = prm.env_kind == 1 ? (float) 3.14159265358979323846 / 3.f : (float) 3.14159265358979323846 * 2.f / 3.f;

    // ---- bodies: members, then (attach_a, attach_b) per muscle ----
    std::vector<Mat> M0(nb);
    std::vector<int> body_shape(nb);
    std::vector<const float *> body_scale(nb);
    int hull_used = 0;
    std::map<std::pair<int, std::vector<float>>, int> hull_cache;
    const float floor_friction = 0.5f;  // robot_walk.cpp:33
    for (int i = 0; i < nm; i++) {
        const RawMember &m = members[i];
        auto sit = shape_id.find(m.shape);
        if (sit == shape_id.end()) { err = "unknown shape " + m.shape; return EVM_E_RUNTIME; }
        body_shape[i] = sit->second;
        body_scale[i] = m.scale;
        quat_to_rows(m.q, M0[i].m);
        for (int a = 0; a < 3; a++) M0[i].o[a] = m.t[a];
        S.body[i].mass = m.mass;
        S.body[i].friction = m.friction;
        // scaled hull points (btConvexHullShape::getScaledPoint), shared between members of equal shape+scale
        std::vector<float> key(m.scale, m.scale + 3);
        auto hk = std::make_pair(sit->second, key);
        auto hit = hull_cache.find(hk);
        const std::vector<float> &pts = shapes[sit->second].pts;
        if (hit == hull_cache.end()) {
            int n = (int) pts.size() / 3;
            const int npad = (n + 1) & ~1;  // pair layout (skel_const.h): an odd hull repeats its last vertex
            if (hull_used + npad > EVM_MAX_HULL_PTS) { err = "hull table overflow"; return EVM_E_UNSUPPORTED; }
            for (int p = 0; p < npad; p++) {
                const int src = p < n ? p : n - 1, g = hull_used + p;
                for (int a = 0; a < 3; a++) S.hull[6 * (g >> 1) + 2 * a + (g & 1)] = pts[3 * src + a] * m.scale[a];
            }
            hull_cache[hk] = hull_used;
            S.member[i].hull_off = hull_used;
            hull_used += npad;
        } else {
            S.member[i].hull_off = hit->second;
        }
        S.member[i].hull_n = (int) pts.size() / 3;
        S.member[i].contact_response = m.ignore ? 0 : 1;
        float mu = floor_friction * m.friction;
        S.member[i].mu = mu < -10.f ? -10.f : (mu > 10.f ? 10.f : mu);
    }
    auto sph = shape_id.find("sphere");
    for (int k = 0; k < nmus; k++) {
        const RawMuscle &mu = muscles[k];
        if (sph == shape_id.end()) { err = "sphere shape missing"; return EVM_E_RUNTIME; }
        int ma, mb;
        if (!find_member(mu.a, ma) || !find_member(mu.b, mb)) return EVM_E_RUNTIME;
        for (int side = 0; side < 2; side++) {
            const int bi = nm + 2 * k + side;
            const Mat &P = M0[side == 0 ? ma : mb];
            const float *pos = side == 0 ? mu.pa : mu.pb;
            // parent.model_matrix_without_scale() * glm::translate(I, pos)   (muscle.cpp:23,27)
            float c0[3], c1[3], c2[3];
            colv(P.m, 0, c0); colv(P.m, 1, c1); colv(P.m, 2, c2);
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) {
                    // column c of the product = P0*I[c][0] + P1*I[c][1] + P2*I[c][2]
                    float e0 = c == 0 ? 1.f : 0.f, e1 = c == 1 ? 1.f : 0.f, e2 = c == 2 ? 1.f : 0.f;
                    M0[bi].m[r][c] = c0[r] * e0 + c1[r] * e1 + c2[r] * e2;
                }
            for (int r = 0; r < 3; r++) M0[bi].o[r] = c0[r] * pos[0] + c1[r] * pos[1] + c2[r] * pos[2] + P.o[r];
            body_shape[bi] = sph->second;
            body_scale[bi] = mu.scale;
            S.body[bi].mass = mu.mass;
            S.body[bi].friction = 0.5f;
        }
    }
    for (int i = 0; i < nb; i++) {
        EvmBodyC &b = S.body[i];
        b.inv_mass = b.mass == 0.f ? 0.f : 1.0f / b.mass;
        float thr;
        if (i < nm) shape_properties(shapes[body_shape[i]].pts, body_scale[i], b.mass, b.inv_inertia, thr, S.member[i].aabb_c, S.member[i].aabb_h);
        else shape_properties(shapes[body_shape[i]].pts, body_scale[i], b.mass, b.inv_inertia, thr);
        if (i < nm) S.member[i].break_thr = thr;
        // applyGravity: m_gravity = g * (1 / invMass); solver: externalForceImpulse = (F * invMass) * dt
        float gy = b.inv_mass != 0.f ? -9.8f * (1.0f / b.inv_mass) : 0.f;
        b.ext_force_y = gy * b.inv_mass * kDt;
        copy_rows(M0[i].m, b.m0);
        for (int a = 0; a < 3; a++) b.t0[a] = M0[i].o[a];
    }

    // ---- skeleton constraints ----
    int hi = 0, fi = 0;
    S.ncon = (int) cons.size();
    for (size_t ci = 0; ci < cons.size(); ci++) {
        const RawCon &c = cons[ci];
        int pa, ch;
        if (!find_member(c.parent, pa) || !find_member(c.child, ch)) return EVM_E_RUNTIME;
        const float miA = S.body[pa].inv_mass, miB = S.body[ch].inv_mass, miS = miA + miB;
        if (c.type == 0) {
            EvmHingeC &h = S.hinge[hi];
            S.con_type[ci] = 0; S.con_idx[ci] = hi++;
            h.a = pa; h.b = ch;
            const float *pivA = c.v, *pivB = c.v + 3, *axA = c.v + 6, *axB = c.v + 9;
            const float lo = c.v[12], up = c.v[13];
            // btHingeConstraint pivot/axis constructor: frame A from body A's WORLD x axis at construction
            float a1[3], a2[3];
            colv(M0[pa].m, 0, a1);
            float proj = dot3(axA, a1);
            if (proj >= 1.0f - kEps) {
                float t[3]; colv(M0[pa].m, 2, t);
                a1[0] = -t[0]; a1[1] = -t[1]; a1[2] = -t[2];
                colv(M0[pa].m, 1, a2);
            } else if (proj <= -1.0f + kEps) {
                colv(M0[pa].m, 2, a1);
                colv(M0[pa].m, 1, a2);
            } else {
                cross3(axA, a1, a2);
                cross3(a2, axA, a1);
            }
            float FA[3][3] = {{a1[0], a2[0], axA[0]}, {a1[1], a2[1], axA[1]}, {a1[2], a2[2], axA[2]}};
            // shortestArcQuat(axisInA, axisInB)
            float cr[3]; cross3(axA, axB, cr);
            float d = dot3(axA, axB);
            float q[4];
            if (d < -1.0f + kEps) {
                float n[3], u[3]; plane_space(axA, n, u);
                q[0] = n[0]; q[1] = n[1]; q[2] = n[2]; q[3] = 0.f;
            } else {
                float s = sqrtf((1.0f + d) * 2.0f), rs = 1.0f / s;
                q[0] = cr[0] * rs; q[1] = cr[1] * rs; q[2] = cr[2] * rs; q[3] = s * 0.5f;
            }
            float b1[3], b2[3];
            quat_rotate(q, a1, b1);
            cross3(axB, b1, b2);
            float FB[3][3] = {{b1[0], b2[0], axB[0]}, {b1[1], b2[1], axB[1]}, {b1[2], b2[2], axB[2]}};
            copy_rows(FA, h.fa); copy_rows(FB, h.fb);
            for (int a = 0; a < 3; a++) { h.fao[a] = pivA[a]; h.fbo[a] = pivB[a]; }
            h.factA = miS > 0.f ? miB / miS : 0.5f;
            h.factB = 1.0f - h.factA;
            h.half_range = (up - lo) / 2.0f;           // btAngularLimit::set
            h.center = norm_angle(lo + h.half_range);
            h.bias = 0.3f; h.relaxation = 1.0f;        // setLimit defaults
        } else {
            EvmFixedC &x = S.fixed[fi];
            S.con_type[ci] = 1; S.con_idx[ci] = fi++;
            x.a = pa; x.b = ch;
            float FA[3][3], FB[3][3];
            quat_to_rows(c.v + 3, FA);
            quat_to_rows(c.v + 10, FB);
            copy_rows(FA, x.fa); copy_rows(FB, x.fb);
            for (int a = 0; a < 3; a++) { x.fao[a] = c.v[a]; x.fbo[a] = c.v[7 + a]; }
        }
    }
    for (int k = 0; k < nmus; k++) {
        const RawMuscle &mu = muscles[k];
        EvmMuscleC &m = S.muscle[k];
        find_member(mu.a, m.ma); find_member(mu.b, m.mb);
        m.sa = nm + 2 * k; m.sb = m.sa + 1;
        for (int a = 0; a < 3; a++) { m.piv_a[a] = mu.pa[a]; m.piv_b[a] = mu.pb[a]; }
        float d[3] = {M0[m.sa].o[0] - M0[m.sb].o[0], M0[m.sa].o[1] - M0[m.sb].o[1], M0[m.sa].o[2] - M0[m.sb].o[2]};
        m.upper_lin = 2.f * sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);  // muscle.cpp:43-49
        m.max_force = mu.force;
        m.max_impulse = mu.force / (1.f / kDt);
        m.speed = mu.speed;
        const float miA = S.body[m.sa].inv_mass, miB = S.body[m.sb].inv_mass, miS = miA + miB;
        m.factA = miS > 0.f ? miB / miS : 0.5f;
        m.factB = 1.0f - m.factA;
    }

    // ---- member pairs that may collide: all but constraint parent / child (setIgnoreCollisionCheck, constraint.cpp:65,147);
    // members without contact response (member.cpp:31-33) get no rows.  Lexicographic (a < b): body0 = a, body1 = b, and
    // this is the order of their rows in the solver ----
    S.npair = 0;
    S.hull_pts = hull_used;
    S.big_hull_off = -1; S.big_hull_n = 0;
    for (int i = 0; i < nm; i++)
        if (S.member[i].hull_n > EVM_BIG_HULL && S.member[i].hull_n > S.big_hull_n) { S.big_hull_off = S.member[i].hull_off; S.big_hull_n = S.member[i].hull_n; }
    if (S.self_collision && S.big_hull_n > EVM_MAX_HULL_PTS / 2) {
        err = "member-vs-member contacts: a hull has more vertices than the narrowphase kernel's LDS table holds";
        return EVM_E_UNSUPPORTED;
    }
    if (S.self_collision) {
        for (int i = 0; i < nm; i++)
            for (int j = i + 1; j < nm; j++) {
                if (!S.member[i].contact_response || !S.member[j].contact_response) continue;
                bool adjacent = false;
                for (int hq = 0; hq < nh; hq++) adjacent = adjacent || (S.hinge[hq].a == i && S.hinge[hq].b == j) || (S.hinge[hq].a == j && S.hinge[hq].b == i);
                for (int fq = 0; fq < nf; fq++) adjacent = adjacent || (S.fixed[fq].a == i && S.fixed[fq].b == j) || (S.fixed[fq].a == j && S.fixed[fq].b == i);
                if (adjacent) continue;
                EvmPairC &P = S.pair[S.npair++];
                P.a = (uint16_t) i; P.b = (uint16_t) j;
                P.thr = std::min(S.member[i].break_thr, S.member[j].break_thr);
                const float f = S.body[i].friction * S.body[j].friction;
                P.mu = f < -10.f ? -10.f : (f > 10.f ? 10.f : f);
            }
        std::vector<int> ord(S.npair);
        for (int k = 0; k < S.npair; k++) ord[k] = k;
        std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) {
            return S.member[S.pair[x].a].hull_n + S.member[S.pair[x].b].hull_n > S.member[S.pair[y].a].hull_n + S.member[S.pair[y].b].hull_n;
        });
        for (int k = 0; k < S.npair; k++) S.pair_order[k] = (uint16_t) ord[k];
    }

    // ---- scratch layout ----
    int o = 0;
    S.sc_r = o; o += 9 * nb;
    S.sc_ext = o; o += 3 * nb;
    S.sc_ms = o; o += 3 * nm;
    S.sc_pt = o; o += 6 * nm;
    S.sc_mobs = o; o += 4 * nmus;
    o = (o + 3) & ~3;  // records are quad-packed: bases on multiples of 4 slots (skel_const.h)
    S.sc_h = o; o += EVM_H_STRIDE * nh;
    S.sc_f = o; o += EVM_F_STRIDE * nf;
    S.sc_s = o; o += EVM_S_STRIDE * nmus;
    S.sc_p = o; o += EVM_P_STRIDE * 2 * nmus;
    S.sc_c = o; o += EVM_CM_STRIDE * nm;
    S.sc_rootms = o; o += 4;
    S.sc_snap = o; o += 4;
    S.sc_nexte = o; o += 12;
    S.sc_total = o;

    // ---- sweep visit list (Bullet order: skeleton constraints, then slider / p2p_a / p2p_b per muscle) ----
    int nv = 0;
    for (int ci = 0; ci < S.ncon; ci++) {
        EvmVisitC &v = S.visit[nv++];
        if (S.con_type[ci] == 0) {
            const EvmHingeC &h = S.hinge[S.con_idx[ci]];
            v.type = 0; v.slot = S.sc_h + EVM_H_STRIDE * S.con_idx[ci]; v.a = h.a; v.b = h.b; v.nslots = EVM_H_STRIDE;
        } else {
            const EvmFixedC &x = S.fixed[S.con_idx[ci]];
            v.type = 1; v.slot = S.sc_f + EVM_F_STRIDE * S.con_idx[ci]; v.a = x.a; v.b = x.b; v.nslots = EVM_F_STRIDE;
        }
    }
    for (int k = 0; k < nmus; k++) {
        const EvmMuscleC &m = S.muscle[k];
        EvmVisitC &v = S.visit[nv++];
        v.type = 2; v.slot = S.sc_s + EVM_S_STRIDE * k; v.a = m.sa; v.b = m.sb; v.nslots = EVM_S_STRIDE;
        EvmVisitC &pa = S.visit[nv++];
        pa.type = 3; pa.slot = S.sc_p + EVM_P_STRIDE * (2 * k); pa.a = m.ma; pa.b = m.sa; pa.nslots = EVM_P_STRIDE;
        EvmVisitC &pb = S.visit[nv++];
        pb.type = 3; pb.slot = S.sc_p + EVM_P_STRIDE * (2 * k + 1); pb.a = m.mb; pb.b = m.sb; pb.nslots = EVM_P_STRIDE;
    }
    for (int i = 0; i < nv; i++) {
        S.visit[i].imA = S.body[S.visit[i].a].inv_mass;
        S.visit[i].imB = S.body[S.visit[i].b].inv_mass;
    }
    {   // dataflow bookkeeping: per-body version each visit must observe before it may run
        std::vector<int> cnt(nb, 0);
        for (int i = 0; i < nv; i++) {
            S.visit[i].need = cnt[S.visit[i].a] | (cnt[S.visit[i].b] << 16);
            cnt[S.visit[i].a]++;
            cnt[S.visit[i].b]++;
        }
        for (int b = 0; b < nb; b++) S.body[b].per_sweep = cnt[b] + (b < nm ? 1 : 0);
        for (int b = 0; b < nb; b++)
            S.body[b].isotropic = S.body[b].inv_inertia[0] == S.body[b].inv_inertia[1] && S.body[b].inv_inertia[1] == S.body[b].inv_inertia[2];
    }
    S.nvisit = nv;

    // ---- sweep schedule (list scheduling of one Gauss-Seidel sweep onto EVM_NW waves) ----
    {
        // dependency depth, reported for information (the chain of visits on the root body)
        std::vector<int> lastl(nb, 0);
        int nlev = 0;
        for (int i = 0; i < nv; i++) {
            int l = 1 + std::max(lastl[S.visit[i].a], lastl[S.visit[i].b]);
            lastl[S.visit[i].a] = lastl[S.visit[i].b] = l;
            nlev = std::max(nlev, l);
        }
        S.nlevels = nlev;
        // items: joint visits in Bullet order, then one contact item per member (after all its joint visits)
        struct Item { int a, b; float cost; int entry; float pre, ca, cb; };  // muscle: pre before any wait, ca / cb after a / b
        std::vector<Item> items;
        // measured on MI355X (tools/stamps3.py, cycles with two waves per SIMD): hinge 1500, fixed 1300, a whole muscle
        // 3550 (slider rows ~1950, then ~800 per p2p), a member's contact rows 2050 (with random actions some env of a 64-env tile touches the ground
        // with nearly every member, so every member is costed as active); ~600 cycles of per-entry overhead
        // (descriptor + record prefetch issue), ~300 for a dependency that crosses waves
        const float ovh = 600.f;
        const float cost_of[4] = {1500.f + ovh, 1300.f + ovh, 2150.f + ovh, 900.f + ovh};
        // a muscle (slider, p2p_a, p2p_b: consecutive in Bullet's order, skeleton.cpp:83-89) is ONE item on the members its
        // two p2p constraints attach to; the attach spheres are private to it
        const int nskel = nv - 3 * nmus;  // skeleton constraints come first in the visit list
        for (int i = 0; i < nskel; i++) items.push_back({S.visit[i].a, S.visit[i].b, cost_of[S.visit[i].type], i, 0.f, 0.f, 0.f});
        for (int k = 0; k < nmus; k++)
            items.push_back({S.visit[nskel + 3 * k + 1].a, S.visit[nskel + 3 * k + 2].a, 3550.f + ovh, EVM_SCHED_MUSCLE | k, 1950.f + ovh, 800.f, 800.f});
        for (int m = 0; m < nm; m++) items.push_back({m, m, 2050.f + ovh, EVM_SCHED_CONTACT | m, 0.f, 0.f, 0.f});
        const int ni = (int) items.size();
        const float hop = 300.f;
        std::vector<std::vector<int>> preds(ni), succs(ni);
        {
            std::vector<int> last(nb, -1);
            for (int i = 0; i < ni; i++) {
                for (int body : {items[i].a, items[i].b}) {
                    if (last[body] >= 0 && (preds[i].empty() || preds[i].back() != last[body])) {
                        preds[i].push_back(last[body]);
                        succs[last[body]].push_back(i);
                    }
                }
                last[items[i].a] = last[items[i].b] = i;
            }
        }
        std::vector<float> bl(ni, 0.f);  // bottom level = longest path to the end of the sweep
        for (int i = ni - 1; i >= 0; i--) {
            float m = 0.f;
            for (int sidx : succs[i]) m = std::max(m, bl[sidx]);
            bl[i] = items[i].cost + m;
        }
        // 1. list scheduling of one sweep (bottom-level priority): gives the global order `topo` — a topological
        //    order of the sweep's dependency graph — and a first assignment
        std::vector<int> npred(ni), wave_of(ni, -1), topo;
        std::vector<float> finish(ni, 0.f), avail(EVM_NW, 0.f);
        for (int i = 0; i < ni; i++) npred[i] = (int) preds[i].size();
        std::vector<char> done(ni, 0);
        for (int step = 0; step < ni; step++) {
            int pick = -1;
            for (int i = 0; i < ni; i++)
                if (!done[i] && npred[i] == 0 && (pick < 0 || bl[i] > bl[pick])) pick = i;
            int bw = 0;
            float bstart = 1e30f;
            for (int w = 0; w < EVM_NW; w++) {
                float st = avail[w];
                for (int pidx : preds[pick]) st = std::max(st, finish[pidx] + (wave_of[pidx] != w ? hop : 0.f));
                if (st < bstart - 1e-3f) { bstart = st; bw = w; }
            }
            wave_of[pick] = bw;
            finish[pick] = bstart + items[pick].cost;
            avail[bw] = finish[pick];
            topo.push_back(pick);
            done[pick] = 1;
            for (int sidx : succs[pick]) npred[sidx]--;
        }
        // 2. the kernel runs the sweeps back to back without a barrier, so what matters is the steady state of the
        //    cyclic schedule, not one sweep's makespan: simulate NUM_ITER sweeps (each wave runs its entries in
        //    `topo` order; an entry starts when its wave is free and the last writers of its bodies are done) and
        //    improve the assignment by local search (move one entry to another wave while the total shrinks).
        //    Every wave's list stays sorted by the one global topological order, which is what rules out deadlock.
        auto simulate = [&](const std::vector<int> &asg) -> float {
            float wave_t[EVM_NW] = {0.f};
            std::vector<float> ready(nb, 0.f);
            std::vector<int> lastw(nb, -1);
            for (int sweep = 0; sweep < 10; sweep++)
                for (int i : topo) {
                    const int w = asg[i];
                    if (items[i].entry & EVM_SCHED_MUSCLE) {
                        // the slider rows touch only the private spheres; member a is waited for after them, member b
                        // after a's rows
                        float t = wave_t[w] + items[i].pre;
                        const int ab[2] = {items[i].a, items[i].b};
                        const float cc[2] = {items[i].ca, items[i].cb};
                        for (int q = 0; q < 2; q++) {
                            const float r = ready[ab[q]] + ((lastw[ab[q]] >= 0 && lastw[ab[q]] != w) ? hop : 0.f);
                            t = std::max(t, r) + cc[q];
                            ready[ab[q]] = t;
                            lastw[ab[q]] = w;
                        }
                        wave_t[w] = t;
                        continue;
                    }
                    float st = wave_t[w];
                    for (int body : {items[i].a, items[i].b}) {
                        const float r = ready[body] + ((lastw[body] >= 0 && lastw[body] != w) ? hop : 0.f);
                        if (r > st) st = r;
                    }
                    const float fn = st + items[i].cost;
                    wave_t[w] = fn;
                    ready[items[i].a] = ready[items[i].b] = fn;
                    lastw[items[i].a] = lastw[items[i].b] = w;
                }
            float m = 0.f;
            for (int w = 0; w < EVM_NW; w++) m = std::max(m, wave_t[w]);
            return m;
        };
        auto improve = [&](std::vector<int> &asg) -> float {
            float best = simulate(asg);
            for (int pass = 0; pass < 40; pass++) {
                bool moved = false;
                for (int i : topo) {
                    const int w0 = asg[i];
                    int bw = w0;
                    float bv = best;
                    for (int w = 0; w < EVM_NW; w++) {
                        if (w == w0) continue;
                        asg[i] = w;
                        const float v = simulate(asg);
                        if (v < bv - 0.5f) { bv = v; bw = w; }
                    }
                    asg[i] = bw;
                    if (bw != w0) { best = bv; moved = true; }
                }
                if (!moved) break;
            }
            return best;
        };
        float best_total = improve(wave_of);
        {   // a few deterministic restarts from pseudo-random assignments (LCG), keep the best
            uint32_t lcg = 12345u;
            for (int trial = 0; trial < 6 && EVM_NW > 1; trial++) {
                std::vector<int> asg(ni);
                for (int i = 0; i < ni; i++) { lcg = lcg * 1664525u + 1013904223u; asg[i] = (int) ((lcg >> 16) % EVM_NW); }
                const float v = improve(asg);
                if (v < best_total - 0.5f) { best_total = v; wave_of = asg; }
            }
        }
        S.sched_cycles = best_total;
        for (int w = 0; w < EVM_NW; w++) S.nsched[w] = S.nwsched[w] = 0;
        for (int i : topo) {
            const int w = wave_of[i];
            if (S.nsched[w] + 3 > EVM_MAX_SCHED || S.nwsched[w] >= EVM_MAX_WAVE_ENTRIES) { err = "sweep schedule overflow"; return EVM_E_UNSUPPORTED; }
            EvmEntryC &e = S.wsched[w][S.nwsched[w]++];
            memset(&e, 0, sizeof(e));
            const int code = items[i].entry;
            if (code & EVM_SCHED_CONTACT) {
                const int m = code & (EVM_SCHED_CONTACT - 1);
                e.type = 4; e.slot = S.sc_c + EVM_CM_STRIDE * m; e.a = e.b = m; e.nslots = EVM_CM_STRIDE;
                e.psA = e.psB = S.body[m].per_sweep;
                S.sched[w][S.nsched[w]++] = code;
            } else if (code & EVM_SCHED_MUSCLE) {
                const int k = code & (EVM_SCHED_MUSCLE - 1);
                const EvmVisitC &sl = S.visit[nskel + 3 * k], &pa = S.visit[nskel + 3 * k + 1], &pb = S.visit[nskel + 3 * k + 2];
                e.type = 5; e.slot = sl.slot; e.nslots = sl.nslots;
                e.a = pa.a; e.b = pb.a; e.imA = pa.imA; e.imB = pb.imA;
                e.need = (pa.need & 0xffff) | ((pb.need & 0xffff) << 16);
                e.psA = S.body[pa.a].per_sweep; e.psB = S.body[pb.a].per_sweep;
                e.spheres = sl.a | (sl.b << 16);
                e.imSa = sl.imA; e.imSb = sl.imB;
                e.iso = S.body[sl.a].isotropic && S.body[sl.b].isotropic;
                e.kA = S.body[sl.a].inv_inertia[0]; e.kB = S.body[sl.b].inv_inertia[0];
                for (int q = 0; q < 3; q++) S.sched[w][S.nsched[w]++] = nskel + 3 * k + q;  // the three visits, for the record
            } else {
                const EvmVisitC &v = S.visit[code];
                e.type = v.type; e.slot = v.slot; e.a = v.a; e.b = v.b; e.imA = v.imA; e.imB = v.imB; e.nslots = v.nslots;
                e.need = v.need; e.psA = S.body[v.a].per_sweep; e.psB = S.body[v.b].per_sweep;
                S.sched[w][S.nsched[w]++] = code;
            }
        }
        // hull scans -> waves (longest first); a hull of more than 64 vertices is cut into two slices
        S.nscan = 0;
        for (int m = 0; m < nm; m++) {
            S.member[m].scan_first = S.nscan;
            const int n = S.member[m].hull_n;
            if (S.member[m].contact_response) {
                const int parts = (n + 63) / 64;
                int begin = 0;
                for (int k = 0; k < parts; k++) {
                    int end = k == parts - 1 ? n : ((((k + 1) * n) / parts) + 1) & ~1;  // even boundaries (vertex pairs)
                    if (end > n) end = n;
                    if (S.nscan >= EVM_MAX_SCAN) { err = "too many hull slices"; return EVM_E_UNSUPPORTED; }
                    if (end > begin) S.scan[S.nscan++] = {m, begin, end, 0};
                    begin = end;
                }
            }
            S.member[m].scan_count = S.nscan - S.member[m].scan_first;
        }
        {
            std::vector<int> order(S.nscan);
            for (int i = 0; i < S.nscan; i++) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
                return S.scan[x].end - S.scan[x].begin > S.scan[y].end - S.scan[y].begin;
            });
            int load[EVM_NW] = {0};
            for (int i : order) {
                int best = 0;
                for (int w = 1; w < EVM_NW; w++) if (load[w] < load[best]) best = w;
                S.scan[i].wave = best;
                load[best] += S.scan[i].end - S.scan[i].begin + 16;
            }
        }
        // manifold maintenance + contact rows of a member: the same work for every member, dealt round robin
        {
            int k = 0;
            for (int m = 0; m < nm; m++) S.member_wave[m] = S.member[m].contact_response ? (k++ % EVM_NW) : 0;
        }
    }
    return EVM_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Lane-group sweep schedule (EvmGSchedC).  One sweep = the joint visits in Bullet's order (S.visit) followed by one
// contact visit per responding member.  Two visits that share no body commute exactly, so the only order that matters
// is, per body, the order of the visits that touch it (= Bullet's).  The schedule is built by list scheduling with
// batching: whenever a wave is free, the most urgent visit whose predecessors are done (urgency = longest dependent
// path through this sweep AND the next one: the sweeps run back to back, so the root body's chain of the next sweep
// hangs on this sweep's leg visits) opens a group entry, which is filled with up to three more ready visits of the same
// type that share no body with it.  Entries sorted by their simulated start time form one global order; every wave's list
// follows it, which rules out deadlock on the version counters.  The result is then scored by simulating the ten
// back-to-back sweeps and improved by moving entries between waves.
// ---------------------------------------------------------------------------------------------------------------------
int build_group_schedule(const EvmSkelC &S, int nwaves, EvmGSchedC &G, std::string &err) {
    // member-vs-member mode: the schedule holds the joint entries only; contact rows (floor and pairs) run after the joint
    // rows of each sweep as rounds of body-disjoint manifolds chosen per env at run time (sweep_groups.h)
    const bool with_contacts = !S.self_collision;
    memset(&G, 0, sizeof(G));
    if (nwaves < 1 || nwaves > EVM_G_MAX_WAVES) { err = "group schedule: bad wave count"; return EVM_E_INVALID; }
    struct V { int type, a, b, rec; float imA, imB, aux; int needA, needB; };
    std::vector<V> vs;
    for (int i = 0; i < S.nvisit; i++) {
        const EvmVisitC &v = S.visit[i];
        float aux = 0.f;
        if (v.type == 2) aux = (S.body[v.a].isotropic && S.body[v.b].isotropic) ? 1.f : 0.f;
        vs.push_back({v.type, v.a, v.b, (v.slot - S.sc_h) / 4, v.imA, v.imB, aux, 0, 0});
    }
    for (int m = 0; m < S.nm && with_contacts; m++)
        if (S.member[m].contact_response)
            vs.push_back({4, m, m, S.sc_c + EVM_CM_STRIDE * m, S.body[m].inv_mass, S.body[m].inv_mass, S.member[m].mu, 0, 0});
    const int nv = (int) vs.size();
    std::vector<int> cnt(S.nb, 0);
    for (auto &v : vs) {
        v.needA = cnt[v.a];
        v.needB = cnt[v.b];
        cnt[v.a]++;
        if (v.b != v.a) cnt[v.b]++;
    }
    // measured on MI355X (tools/gstamps.py, one wave per SIMD, cycles per group entry incl. its LDS round trips):
    // hinge 1770, fixed 1700, slider 1900, p2p 1000, contact rows 2100; a dependency that crosses waves ~300;
    // chain entries (below): ~700 + 1060 per hinge phase, ~600 + 600 per p2p phase (rows + the cross-lane hand-over)
    const float cost_of[5] = {1770.f, 1700.f, 1900.f, 1000.f, 2100.f}, hop = 300.f;
    // ---- items: single visits, and CHAINS = 3..4 consecutive visits of one type (hinge / p2p) on one shared body a with
    // distinct second bodies (the spider's root: its four hinges, the four p2p constraints of its muscles).  A chain runs as
    // one group entry whose lane groups take turns (sweep_groups.h, g_hinge_chain / g_p2p_chain).
    struct Item { int type; std::vector<int> vis; float cost; };
    std::vector<Item> items;
    std::vector<int> item_of(nv, -1);
    auto build_items = [&](bool with_chains) {
        items.clear();
        std::fill(item_of.begin(), item_of.end(), -1);
        if (with_chains && !getenv("EVM_G_NOCHAIN")) {
            for (int x = 0; x < S.nb; x++) {
                std::vector<int> seq;
                for (int i = 0; i < nv; i++) if (vs[i].a == x || vs[i].b == x) seq.push_back(i);
                size_t p0 = 0;
                while (p0 < seq.size()) {
                    const int t = vs[seq[p0]].type;
                    size_t p1 = p0;
                    if ((t == 0 || t == 3) && vs[seq[p0]].a == x && item_of[seq[p0]] < 0) {
                        p1 = p0 + 1;
                        while (p1 < seq.size() && p1 - p0 < EVM_G_SLOTS && vs[seq[p1]].type == t && vs[seq[p1]].a == x && item_of[seq[p1]] < 0) {
                            bool dup = false;
                            for (size_t q = p0; q < p1; q++) dup = dup || vs[seq[q]].b == vs[seq[p1]].b;
                            if (dup) break;
                            p1++;
                        }
                    }
                    if (p1 - p0 >= 3) {
                        Item it;
                        it.type = t == 0 ? 5 : 6;
                        for (size_t q = p0; q < p1; q++) { it.vis.push_back(seq[q]); item_of[seq[q]] = (int) items.size(); }
                        it.cost = t == 0 ? 700.f + 1060.f * (float) (p1 - p0) : 600.f + 600.f * (float) (p1 - p0);
                        items.push_back(it);
                        p0 = p1;
                    } else p0++;
                }
            }
        }
        for (int i = 0; i < nv; i++)
            if (item_of[i] < 0) { item_of[i] = (int) items.size(); items.push_back({vs[i].type, {i}, cost_of[vs[i].type]}); }
    };
    std::vector<std::vector<int>> preds, succs;
    std::vector<int> first_on(S.nb, -1), last_on(S.nb, -1);  // first / last ITEM on every body within a sweep
    auto build_graph = [&]() -> bool {
        const int ni = (int) items.size();
        preds.assign(ni, {});
        succs.assign(ni, {});
        std::fill(first_on.begin(), first_on.end(), -1);
        std::fill(last_on.begin(), last_on.end(), -1);
        for (int i = 0; i < nv; i++) {  // Bullet order
            const int me = item_of[i];
            for (int body : {vs[i].a, vs[i].b}) {
                const int pr = last_on[body];
                if (pr >= 0 && pr != me && std::find(preds[me].begin(), preds[me].end(), pr) == preds[me].end()) {
                    preds[me].push_back(pr);
                    succs[pr].push_back(me);
                }
                if (first_on[body] < 0) first_on[body] = me;
                last_on[body] = me;
            }
        }
        // acyclic?  (a chain may swallow a visit's predecessor AND successor)
        std::vector<int> indeg(ni);
        for (int k = 0; k < ni; k++) indeg[k] = (int) preds[k].size();
        std::vector<int> stack;
        for (int k = 0; k < ni; k++) if (!indeg[k]) stack.push_back(k);
        int seen = 0;
        while (!stack.empty()) {
            const int k = stack.back(); stack.pop_back(); seen++;
            for (int y : succs[k]) if (--indeg[y] == 0) stack.push_back(y);
        }
        return seen == ni;
    };
    build_items(true);
    if (!build_graph()) { build_items(false); build_graph(); }
    const int ni = (int) items.size();
    // bottom levels over two consecutive sweeps (items in topological order = by their first visit)
    std::vector<int> topo(ni);
    for (int k = 0; k < ni; k++) topo[k] = k;
    {
        // Kahn order
        std::vector<int> indeg(ni), order;
        for (int k = 0; k < ni; k++) indeg[k] = (int) preds[k].size();
        std::vector<int> q;
        for (int k = 0; k < ni; k++) if (!indeg[k]) q.push_back(k);
        while (!q.empty()) { const int k = q.back(); q.pop_back(); order.push_back(k); for (int y : succs[k]) if (--indeg[y] == 0) q.push_back(y); }
        topo = order;
    }
    std::vector<float> bl1(ni, 0.f), bl0(ni, 0.f);
    for (int t = ni - 1; t >= 0; t--) {
        const int k = topo[t];
        float m = 0.f;
        for (int y : succs[k]) m = std::max(m, bl1[y]);
        bl1[k] = items[k].cost + m;
    }
    for (int t = ni - 1; t >= 0; t--) {
        const int k = topo[t];
        float m = 0.f;
        for (int y : succs[k]) m = std::max(m, bl0[y]);
        for (int u : items[k].vis)
            for (int body : {vs[u].a, vs[u].b})
                if (last_on[body] == k) m = std::max(m, bl1[first_on[body]]);  // this body's next visit is in the next sweep
        bl0[k] = items[k].cost + m;
    }
    // ---- list scheduling with batching ----
    struct Ent { int type, wave; float start, cost; std::vector<int> vis; };
    std::vector<Ent> ents;
    {
        std::vector<float> finish(ni, -1.f), wave_free(nwaves, 0.f);
        std::vector<int> wave_of(ni, -1), npred(ni);
        std::vector<char> placed(ni, 0);
        for (int k = 0; k < ni; k++) npred[k] = (int) preds[k].size();
        int left = ni;
        auto ready_at = [&](int k, int w) {
            float r = 0.f;
            for (int pidx : preds[k]) r = std::max(r, finish[pidx] + (wave_of[pidx] != w ? hop : 0.f));
            return r;
        };
        auto clash = [&](int k, const std::vector<int> &have) {
            for (int h : have)
                for (int u : items[h].vis)
                    for (int v : items[k].vis)
                        if (vs[u].a == vs[v].a || vs[u].a == vs[v].b || vs[u].b == vs[v].a || vs[u].b == vs[v].b) return true;
            return false;
        };
        while (left > 0) {
            int bv = -1, bw = 0;
            float bstart = 1e30f;
            for (int k = 0; k < ni; k++) {
                if (placed[k] || npred[k] != 0) continue;
                for (int w = 0; w < nwaves; w++) {
                    const float st = std::max(wave_free[w], ready_at(k, w));
                    const bool better = st < bstart - 1.f || (st < bstart + 1.f && bv >= 0 && bl0[k] > bl0[bv] + 0.5f);
                    if (bv < 0 || better) { bv = k; bw = w; bstart = st; }
                }
            }
            std::vector<int> have = {bv};
            if (items[bv].type < 5) {
                // fill: ready single visits of the same type that share no body with the entry, most urgent first
                for (int q = 1; q < EVM_G_SLOTS; q++) {
                    int pick = -1;
                    for (int k = 0; k < ni; k++) {
                        if (placed[k] || npred[k] != 0 || items[k].type != items[bv].type) continue;
                        if (std::find(have.begin(), have.end(), k) != have.end() || clash(k, have) || ready_at(k, bw) > bstart + 1.f) continue;
                        if (pick < 0 || bl0[k] > bl0[pick]) pick = k;
                    }
                    if (pick < 0) break;
                    have.push_back(pick);
                }
            }
            Ent e;
            e.type = items[bv].type; e.wave = bw; e.start = bstart; e.cost = items[bv].cost;
            const float fin = bstart + e.cost;
            for (int h : have) {
                for (int u : items[h].vis) e.vis.push_back(u);
                placed[h] = 1; finish[h] = fin; wave_of[h] = bw; left--;
                for (int y : succs[h]) npred[y]--;
            }
            wave_free[bw] = fin;
            ents.push_back(e);
        }
    }
    // one global order: by simulated start time (a linear extension of the dependencies; per wave already increasing)
    std::stable_sort(ents.begin(), ents.end(), [](const Ent &x, const Ent &y) { return x.start < y.start; });
    const int ne = (int) ents.size();
    if (ne > EVM_G_MAX_ENTRIES) { err = "group schedule overflow"; return EVM_E_UNSUPPORTED; }
    // ---- score: the ten back-to-back sweeps with fixed per-wave lists; improve by moving entries between waves ----
    const float balance_w = getenv("EVM_G_BALANCE") ? (float) atof(getenv("EVM_G_BALANCE")) : 0.f;  // (A/B knob)
    const bool barrier_model = !(getenv("EVM_G_BARRIER_MODEL") && getenv("EVM_G_BARRIER_MODEL")[0] == '0');  // (A/B knob)
    auto simulate = [&](const std::vector<int> &asg) -> float {
        std::vector<float> wave_t(nwaves, 0.f), ready(S.nb, 0.f);
        std::vector<int> lastw(S.nb, -1);
        for (int sweep = 0; sweep < 10; sweep++) {
            for (int e = 0; e < ne; e++) {
                const int w = asg[e];
                float st = wave_t[w];
                for (int i : ents[e].vis)
                    for (int body : {vs[i].a, vs[i].b}) {
                        const float r = ready[body] + ((lastw[body] >= 0 && lastw[body] != w) ? hop : 0.f);
                        if (r > st) st = r;
                    }
                const float fn = st + ents[e].cost;
                wave_t[w] = fn;
                for (int i : ents[e].vis) { ready[vs[i].a] = ready[vs[i].b] = fn; lastw[vs[i].a] = lastw[vs[i].b] = w; }
            }
            // member-vs-member mode: the contact rounds sit between two workgroup barriers after every sweep's joint rows, so the
            // sweeps do not flow into each other — every sweep starts from a common time and what counts is ONE sweep's makespan
            if (!with_contacts && barrier_model) {
                float mx = 0.f;
                for (float t : wave_t) mx = std::max(mx, t);
                std::fill(wave_t.begin(), wave_t.end(), mx);
                std::fill(ready.begin(), ready.end(), mx);
                std::fill(lastw.begin(), lastw.end(), -1);
            }
        }
        float m = 0.f;
        for (float t : wave_t) m = std::max(m, t);
        // tie-break towards balanced waves: the cost model is approximate, and the busiest wave is the one that cannot absorb
        // an entry that runs longer than modelled
        std::vector<float> busy(nwaves, 0.f);
        for (int e = 0; e < ne; e++) busy[asg[e]] += ents[e].cost;
        float bmax = 0.f, bsum = 0.f;
        for (float b : busy) { bmax = std::max(bmax, b); bsum += b; }
        return m + balance_w * (bmax - bsum / (float) nwaves);
    };
    std::vector<int> asg(ne);
    for (int e = 0; e < ne; e++) asg[e] = ents[e].wave;
    float best = simulate(asg);
    if (nwaves > 1) {
        for (int pass = 0; pass < 40; pass++) {
            bool moved = false;
            for (int e = 0; e < ne; e++) {
                const int w0 = asg[e];
                int bw = w0;
                float bv = best;
                for (int w = 0; w < nwaves; w++) {
                    if (w == w0) continue;
                    asg[e] = w;
                    const float v = simulate(asg);
                    if (v < bv - 0.5f) { bv = v; bw = w; }
                }
                asg[e] = bw;
                if (bw != w0) { best = bv; moved = true; }
            }
            if (!moved) break;
        }
    }
    // ---- tables ----
    G.nwaves = nwaves;
    G.total = ne;
    G.est_cycles = best;
    G.nrq = (S.sc_c - S.sc_h) / 4;
    int k = 0;
    for (int w = 0; w < nwaves; w++) {
        G.first[w] = k;
        for (int e = 0; e < ne; e++) {
            if (asg[e] != w) continue;
            EvmGEntryC &E = G.entry[k];
            E.type = ents[e].type; E.order = e; E.iso = 1; E.members = 0;
            for (int sidx = 0; sidx < EVM_G_SLOTS; sidx++) {
                EvmGSlotC &sl = G.slot[k][sidx];
                memset(&sl, 0, sizeof(sl));
                sl.rec = -1;
                if (sidx >= (int) ents[e].vis.size()) continue;
                const V &v = vs[ents[e].vis[sidx]];
                sl.rec = v.rec; sl.a = v.a; sl.b = v.b; sl.imA = v.imA; sl.imB = v.imB; sl.aux = v.aux;
                sl.need = v.needA | (v.needB << 16);
                sl.ps = cnt[v.a] | (cnt[v.b] << 16);
                if (v.type == 2 && v.aux == 0.f) E.iso = 0;
                if (v.type == 4) E.members |= 1 << v.a;
            }
            if (E.type != 2) E.iso = 0;
            if (E.type >= 5) E.iso = (int) ents[e].vis.size();  // chain entries: filled slots, in Bullet's order
            k++;
        }
        G.count[w] = k - G.first[w];
        // run lengths: consecutive entries of one class (joint rows / contact rows) in this wave's list, in the high half of
        // `order` (the kernel walks a run of contact entries as one software-pipelined block)
        for (int q = G.first[w] + G.count[w] - 1, run = 0; q >= G.first[w]; q--) {
            const bool same = q + 1 < G.first[w] + G.count[w] && (G.entry[q + 1].type == 4) == (G.entry[q].type == 4);
            run = same ? run + 1 : 1;
            G.entry[q].order |= run << 16;
        }
    }
    // LDS image: per env (3 quads per body + the joint records), then the slot table (2 quads per slot) and the entry
    // headers (1 quad each), the version counters and the per-env residuals
    const size_t quads = (size_t) (3 * S.nb + G.nrq) * EVM_G_ENVS + (size_t) 3 * S.nb /* one pad quad per body row */ + (size_t) ne * (2 * EVM_G_SLOTS + 1);
    size_t bytes = quads * 16 + (size_t) ((S.nb + 3) / 4 * 4) * 4 + EVM_G_ENVS * 4;
    G.with_contacts = with_contacts ? 1 : 0;
    if (!with_contacts) {
        // behind the image: the contact program (2 banks x 16 slots x 16 envs words), four words per wave, the bodies'
        // inverse masses and the members' push / turn velocities of the split-impulse phase
        bytes = (bytes + 15) & ~(size_t) 15;
        bytes += (size_t) 2 * 16 * EVM_G_ENVS * 4 + 64 + (size_t) ((S.nb + 3) & ~3) * 4 + (size_t) 6 * S.nm * EVM_G_ENVS * 4;
        if (nwaves == 3) { err = "member-vs-member contacts: the lane-group sweeps kernel takes 1, 2 or 4 waves per workgroup"; return EVM_E_UNSUPPORTED; }
    }
    G.lds_bytes = (int) bytes;
    if (bytes > 160 * 1024) { err = "skeleton records exceed the LDS image of the lane-group sweeps"; return EVM_E_UNSUPPORTED; }
    return EVM_OK;
}

}  // namespace evm
