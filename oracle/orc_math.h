// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under evomotion_amd/ may include, link or call this.
//
// Scalar fp32 linear-algebra helpers written to follow Bullet3's LinearMath operation order
// (btVector3 / btMatrix3x3 / btQuaternion / btTransform, single precision, scalar code path) and the
// handful of GLM functions the reference calls at load / reset time.
//
// [UPSTREAM] Bullet3 and GLM are third-party dependencies that are NOT vendored in /root/reference and are
// not installed in the build image.  The formulas below restate their published
// algorithms; the reference call sites that make them relevant are cited next to each user.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

static constexpr float SIMD_EPSILON = 1.1920928955078125e-7f;  // FLT_EPSILON
static constexpr float SIMD_PI = 3.1415926535897932384626433832795029f;
static constexpr float SIMD_2_PI = 2.0f * SIMD_PI;
static constexpr float SIMD_HALF_PI = SIMD_PI * 0.5f;
static constexpr float SIMD_INFINITY = 3.402823466e+38f;  // FLT_MAX
static constexpr float SIMDSQRT12 = 0.7071067811865475244008443621048490f;
static constexpr float BT_LARGE_FLOAT = 1e18f;

struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float &at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(const V3 &a, const V3 &b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(const V3 &a, const V3 &b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(const V3 &a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(const V3 &a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, const V3 &a) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(const V3 &a, const V3 &b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 &operator+=(V3 &a, const V3 &b) { a = a + b; return a; }
inline V3 &operator-=(V3 &a, const V3 &b) { a = a - b; return a; }
inline V3 &operator*=(V3 &a, float s) { a = a * s; return a; }
inline float dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3 &a, const V3 &b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float length2(const V3 &a) { return dot(a, a); }
inline float length(const V3 &a) { return std::sqrt(length2(a)); }
// btVector3::normalize(): *this /= length()  ==  *this *= (1/length)
inline V3 normalized(const V3 &a) { return a * (1.0f / length(a)); }
inline V3 operator/(const V3 &a, float s) { return a * (1.0f / s); }

// btMatrix3x3: three ROW vectors.
struct M3 {
    V3 r[3];
    M3() {}
    M3(float xx, float xy, float xz, float yx, float yy, float yz, float zx, float zy, float zz) {
        r[0] = V3(xx, xy, xz); r[1] = V3(yx, yy, yz); r[2] = V3(zx, zy, zz);
    }
    static M3 identity() { return M3(1, 0, 0, 0, 1, 0, 0, 0, 1); }
    V3 col(int c) const { return V3(r[0][c], r[1][c], r[2][c]); }
    float tdotx(const V3 &v) const { return r[0].x * v.x + r[1].x * v.y + r[2].x * v.z; }
    float tdoty(const V3 &v) const { return r[0].y * v.x + r[1].y * v.y + r[2].y * v.z; }
    float tdotz(const V3 &v) const { return r[0].z * v.x + r[1].z * v.y + r[2].z * v.z; }
    M3 transpose() const {
        return M3(r[0].x, r[1].x, r[2].x, r[0].y, r[1].y, r[2].y, r[0].z, r[1].z, r[2].z);
    }
    // btMatrix3x3::scaled(s): scales the columns
    M3 scaled(const V3 &s) const {
        return M3(r[0].x * s.x, r[0].y * s.y, r[0].z * s.z, r[1].x * s.x, r[1].y * s.y, r[1].z * s.z,
                  r[2].x * s.x, r[2].y * s.y, r[2].z * s.z);
    }
    float cofac(int r1, int c1, int r2, int c2) const {
        return r[r1][c1] * r[r2][c2] - r[r1][c2] * r[r2][c1];
    }
    // btMatrix3x3::inverse() (cofactor form)
    M3 inverse() const {
        V3 co(cofac(1, 1, 2, 2), cofac(1, 2, 2, 0), cofac(1, 0, 2, 1));
        float det = dot(r[0], co);
        float s = 1.0f / det;
        return M3(co.x * s, cofac(0, 2, 2, 1) * s, cofac(0, 1, 1, 2) * s, co.y * s, cofac(0, 0, 2, 2) * s,
                  cofac(0, 2, 1, 0) * s, co.z * s, cofac(0, 1, 2, 0) * s, cofac(0, 0, 1, 1) * s);
    }
};
inline V3 operator*(const M3 &m, const V3 &v) { return V3(dot(m.r[0], v), dot(m.r[1], v), dot(m.r[2], v)); }
// btVector3 * btMatrix3x3  (row-vector times matrix)
inline V3 operator*(const V3 &v, const M3 &m) { return V3(m.tdotx(v), m.tdoty(v), m.tdotz(v)); }
inline M3 operator*(const M3 &a, const M3 &b) {
    return M3(b.tdotx(a.r[0]), b.tdoty(a.r[0]), b.tdotz(a.r[0]), b.tdotx(a.r[1]), b.tdoty(a.r[1]),
              b.tdotz(a.r[1]), b.tdotx(a.r[2]), b.tdoty(a.r[2]), b.tdotz(a.r[2]));
}
inline M3 operator+(const M3 &a, const M3 &b) {
    M3 m; for (int i = 0; i < 3; i++) m.r[i] = a.r[i] + b.r[i]; return m;
}
inline M3 operator-(const M3 &a, const M3 &b) {
    M3 m; for (int i = 0; i < 3; i++) m.r[i] = a.r[i] - b.r[i]; return m;
}
inline M3 operator*(const M3 &a, float s) {
    M3 m; for (int i = 0; i < 3; i++) m.r[i] = a.r[i] * s; return m;
}
// btVector3::getSkewSymmetricMatrix -> rows
inline M3 skew(const V3 &v) { return M3(0, -v.z, v.y, v.z, 0, -v.x, -v.y, v.x, 0); }
// btMatrix3x3::solve33 (Cramer on columns)
inline V3 solve33(const M3 &m, const V3 &b) {
    V3 c1 = m.col(0), c2 = m.col(1), c3 = m.col(2);
    float det = dot(c1, cross(c2, c3));
    if (std::fabs(det) > SIMD_EPSILON) {
        V3 x;
        x.x = dot(b, cross(c2, c3)) / det;
        x.y = dot(c1, cross(b, c3)) / det;
        x.z = dot(c1, cross(c2, b)) / det;
        return x;
    }
    return V3(0, 0, 0);
}

struct Q {
    float x, y, z, w;
    Q() : x(0), y(0), z(0), w(1) {}
    Q(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
};
inline Q operator*(const Q &q1, const Q &q2) {
    return Q(q1.w * q2.x + q1.x * q2.w + q1.y * q2.z - q1.z * q2.y,
             q1.w * q2.y + q1.y * q2.w + q1.z * q2.x - q1.x * q2.z,
             q1.w * q2.z + q1.z * q2.w + q1.x * q2.y - q1.y * q2.x,
             q1.w * q2.w - q1.x * q2.x - q1.y * q2.y - q1.z * q2.z);
}
inline Q operator*(const Q &q, const V3 &w) {
    return Q(q.w * w.x + q.y * w.z - q.z * w.y, q.w * w.y + q.z * w.x - q.x * w.z,
             q.w * w.z + q.x * w.y - q.y * w.x, -q.x * w.x - q.y * w.y - q.z * w.z);
}
inline Q inverse(const Q &q) { return Q(-q.x, -q.y, -q.z, q.w); }
inline float length2(const Q &q) { return q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w; }
inline V3 quatRotate(const Q &rot, const V3 &v) {
    Q q = rot * v;
    q = q * inverse(rot);
    return V3(q.x, q.y, q.z);
}
// btQuaternion::safeNormalize
inline Q safeNormalize(Q q) {
    float l2 = length2(q);
    if (l2 > SIMD_EPSILON) {
        float s = 1.0f / std::sqrt(l2);
        q = Q(q.x * s, q.y * s, q.z * s, q.w * s);
    }
    return q;
}
// btMatrix3x3::setRotation
inline M3 matFromQuat(const Q &q) {
    float d = length2(q);
    float s = 2.0f / d;
    float xs = q.x * s, ys = q.y * s, zs = q.z * s;
    float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
    float xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
    float yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    return M3(1.0f - (yy + zz), xy - wz, xz + wy, xy + wz, 1.0f - (xx + zz), yz - wx, xz - wy, yz + wx,
              1.0f - (xx + yy));
}
// btMatrix3x3::getRotation (scalar path)
inline Q quatFromMat(const M3 &m) {
    float trace = m.r[0].x + m.r[1].y + m.r[2].z;
    float t[4];
    if (trace > 0.0f) {
        float s = std::sqrt(trace + 1.0f);
        t[3] = s * 0.5f;
        s = 0.5f / s;
        t[0] = (m.r[2].y - m.r[1].z) * s;
        t[1] = (m.r[0].z - m.r[2].x) * s;
        t[2] = (m.r[1].x - m.r[0].y) * s;
    } else {
        int i = m.r[0].x < m.r[1].y ? (m.r[1].y < m.r[2].z ? 2 : 1) : (m.r[0].x < m.r[2].z ? 2 : 0);
        int j = (i + 1) % 3, k = (i + 2) % 3;
        float s = std::sqrt(m.r[i][i] - m.r[j][j] - m.r[k][k] + 1.0f);
        t[i] = s * 0.5f;
        s = 0.5f / s;
        t[3] = (m.r[k][j] - m.r[j][k]) * s;
        t[j] = (m.r[j][i] + m.r[i][j]) * s;
        t[k] = (m.r[k][i] + m.r[i][k]) * s;
    }
    return Q(t[0], t[1], t[2], t[3]);
}
// btQuaternion::getEulerZYX
inline void getEulerZYX(const Q &q, float &yawZ, float &pitchY, float &rollX) {
    float sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z, squ = q.w * q.w;
    float sarg = -2.0f * (q.x * q.z - q.w * q.y);
    if (sarg <= -0.99999f) {
        pitchY = -0.5f * SIMD_PI; rollX = 0; yawZ = 2.0f * std::atan2(q.x, -q.y);
    } else if (sarg >= 0.99999f) {
        pitchY = 0.5f * SIMD_PI; rollX = 0; yawZ = 2.0f * std::atan2(-q.x, q.y);
    } else {
        pitchY = std::asin(sarg);
        rollX = std::atan2(2.0f * (q.y * q.z + q.w * q.x), squ - sqx - sqy + sqz);
        yawZ = std::atan2(2.0f * (q.x * q.y + q.w * q.z), squ + sqx - sqy - sqz);
    }
}

struct Xf {
    M3 b;
    V3 o;
    static Xf identity() { Xf t; t.b = M3::identity(); t.o = V3(0, 0, 0); return t; }
    V3 operator()(const V3 &x) const { return V3(dot(b.r[0], x), dot(b.r[1], x), dot(b.r[2], x)) + o; }
    V3 invXform(const V3 &v) const { return b.transpose() * (v - o); }
};
inline Xf operator*(const Xf &a, const Xf &c) { Xf t; t.b = a.b * c.b; t.o = a(c.o); return t; }

inline float btNormalizeAngle(float a) {
    a = std::fmod(a, SIMD_2_PI);
    if (a < -SIMD_PI) return a + SIMD_2_PI;
    if (a > SIMD_PI) return a - SIMD_2_PI;
    return a;
}
inline float btAdjustAngleToLimits(float angle, float lo, float hi) {
    if (lo >= hi) return angle;
    if (angle < lo) {
        float diffLo = std::fabs(btNormalizeAngle(lo - angle));
        float diffHi = std::fabs(btNormalizeAngle(hi - angle));
        return (diffLo < diffHi) ? angle : (angle + SIMD_2_PI);
    }
    if (angle > hi) {
        float diffHi = std::fabs(btNormalizeAngle(angle - hi));
        float diffLo = std::fabs(btNormalizeAngle(angle - lo));
        return (diffLo < diffHi) ? (angle - SIMD_2_PI) : angle;
    }
    return angle;
}
inline void btPlaneSpace1(const V3 &n, V3 &p, V3 &q) {
    if (std::fabs(n.z) > SIMDSQRT12) {
        float a = n.y * n.y + n.z * n.z;
        float k = 1.0f / std::sqrt(a);
        p = V3(0, -n.z * k, n.y * k);
        q = V3(a * k, -n.x * p.z, n.x * p.y);
    } else {
        float a = n.x * n.x + n.y * n.y;
        float k = 1.0f / std::sqrt(a);
        p = V3(-n.y * k, n.x * k, 0);
        q = V3(-n.z * p.y, n.z * p.x, a * k);
    }
}
inline Q shortestArcQuat(const V3 &v0, const V3 &v1) {
    V3 c = cross(v0, v1);
    float d = dot(v0, v1);
    if (d < -1.0f + SIMD_EPSILON) {
        V3 n, unused;
        btPlaneSpace1(v0, n, unused);
        return Q(n.x, n.y, n.z, 0.0f);
    }
    float s = std::sqrt((1.0f + d) * 2.0f);
    float rs = 1.0f / s;
    return Q(c.x * rs, c.y * rs, c.z * rs, s * 0.5f);
}

// btTransformUtil::integrateTransform (exponential map, ANGULAR_MOTION_THRESHOLD = pi/4)
inline Xf integrateTransform(const Xf &cur, const V3 &linvel, const V3 &angvel, float dt, Q *out_q = nullptr) {
    Xf pred;
    pred.o = cur.o + linvel * dt;
    V3 axis;
    float fAngle2 = length2(angvel);
    float fAngle = 0;
    if (fAngle2 > SIMD_EPSILON) fAngle = std::sqrt(fAngle2);
    const float ANGULAR_MOTION_THRESHOLD = 0.5f * SIMD_HALF_PI;
    if (fAngle * dt > ANGULAR_MOTION_THRESHOLD) fAngle = ANGULAR_MOTION_THRESHOLD / dt;
    if (fAngle < 0.001f) {
        axis = angvel * (0.5f * dt - (dt * dt * dt) * 0.020833333333f * fAngle * fAngle);
    } else {
        axis = angvel * (std::sin(0.5f * fAngle * dt) / fAngle);
    }
    Q dorn(axis.x, axis.y, axis.z, std::cos(fAngle * dt * 0.5f));
    Q orn0 = quatFromMat(cur.b);
    Q predOrn = safeNormalize(dorn * orn0);
    if (length2(predOrn) > SIMD_EPSILON) {
        pred.b = matFromQuat(predOrn);
        if (out_q) *out_q = predOrn;
    } else {
        pred.b = cur.b;
        if (out_q) *out_q = orn0;
    }
    return pred;
}

// ---- GLM restatements (column-major mat4 reduced to basis+origin; the 4th row is always 0 0 0 1) ----
// glm::mat3_cast / mat4_cast of a (possibly non-unit) quaternion (w,x,y,z).  Returned as btMatrix3x3 rows,
// i.e. after btTransform::setFromOpenGLMatrix.  Reference: evo_motion_model/src/robot/member.cpp:26,
// evo_motion_model/src/item.cpp:32-33.
inline M3 glm_mat3_cast(float w, float x, float y, float z) {
    float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z, qwx = w * x, qwy = w * y,
          qwz = w * z;
    // Result[col][row]
    float c00 = 1.0f - 2.0f * (qyy + qzz), c01 = 2.0f * (qxy + qwz), c02 = 2.0f * (qxz - qwy);
    float c10 = 2.0f * (qxy - qwz), c11 = 1.0f - 2.0f * (qxx + qzz), c12 = 2.0f * (qyz + qwx);
    float c20 = 2.0f * (qxz + qwy), c21 = 2.0f * (qyz - qwx), c22 = 1.0f - 2.0f * (qxx + qyy);
    return M3(c00, c10, c20, c01, c11, c21, c02, c12, c22);
}
// glm::eulerAngleYXZ(yaw, pitch, roll) as rows.  Reference: evo_motion_model/src/env/robot_walk.cpp:85-86.
inline M3 glm_eulerAngleYXZ(float yaw, float pitch, float roll) {
    float ch = std::cos(yaw), sh = std::sin(yaw), cp = std::cos(pitch), sp = std::sin(pitch), cb = std::cos(roll),
          sb = std::sin(roll);
    float c00 = ch * cb + sh * sp * sb, c01 = sb * cp, c02 = -sh * cb + ch * sp * sb;
    float c10 = -ch * sb + sh * sp * cb, c11 = cb * cp, c12 = sb * sh + ch * sp * cb;
    float c20 = sh * cp, c21 = -sp, c22 = ch * cp;
    return M3(c00, c10, c20, c01, c11, c21, c02, c12, c22);
}
// glm mat4 * mat4 for affine matrices: column c of result = A0*B[c][0] + A1*B[c][1] + A2*B[c][2] (+ A3 for c==3).
inline Xf glm_mul(const Xf &A, const Xf &B) {
    Xf R;
    V3 a0 = A.b.col(0), a1 = A.b.col(1), a2 = A.b.col(2);
    V3 cols[3];
    for (int c = 0; c < 3; c++) {
        V3 bc = B.b.col(c);
        cols[c] = a0 * bc.x + a1 * bc.y + a2 * bc.z;
    }
    R.b = M3(cols[0].x, cols[1].x, cols[2].x, cols[0].y, cols[1].y, cols[2].y, cols[0].z, cols[1].z, cols[2].z);
    R.o = a0 * B.o.x + a1 * B.o.y + a2 * B.o.z + A.o;
    return R;
}

// ---- std::mt19937 + libstdc++ uniform_real_distribution<float>(0,1) ----
// Reference: evo_motion_model/src/env/robot_walk.h:34-35, robot_walk.cpp:21,82-84.
// libstdc++ generate_canonical<float,24>: one 32-bit draw, u = float(x) * 2^-32, and a result that
// rounds to 1.0f is replaced by nextafter(1,0).
struct MT19937 {
    uint32_t mt[624];
    int idx;
    void seed(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t) i;
        idx = 624;
    }
    void twist() {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
            mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        idx = 0;
    }
    uint32_t next() {
        if (idx >= 624) twist();
        uint32_t y = mt[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    float uniform01() {
        float r = (float) next() * 2.3283064365386963e-10f;  // float(x) / 2^32, both steps in fp32
        if (r >= 1.0f) r = 0.99999994f;
        return r;
    }
};

}  // namespace orc
