// Device-side fp32 helpers for the dynamics kernel (gfx950).  Plain structs in registers; no LDS, no memory.
#pragma once
#include <hip/hip_runtime.h>

#define DEV __device__ __forceinline__

namespace evm {

struct F3 {
    float x, y, z;
};
DEV F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
DEV F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
DEV F3 operator*(float s, F3 a) { return f3(a.x * s, a.y * s, a.z * s); }
DEV float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV F3 cross(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV float len2(F3 a) { return dot(a, a); }
DEV F3 normalize(F3 a) { return a * (1.0f / sqrtf(len2(a))); }
DEV float comp(F3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

struct M33 {  // rows (btMatrix3x3 convention)
    F3 r0, r1, r2;
};
DEV F3 mul(const M33 &m, F3 v) { return f3(dot(m.r0, v), dot(m.r1, v), dot(m.r2, v)); }
DEV F3 col(const M33 &m, int c) { return f3(comp(m.r0, c), comp(m.r1, c), comp(m.r2, c)); }
DEV F3 col0(const M33 &m) { return f3(m.r0.x, m.r1.x, m.r2.x); }
DEV F3 col1(const M33 &m) { return f3(m.r0.y, m.r1.y, m.r2.y); }
DEV F3 col2(const M33 &m) { return f3(m.r0.z, m.r1.z, m.r2.z); }
DEV M33 m33(F3 a, F3 b, F3 c) { M33 m; m.r0 = a; m.r1 = b; m.r2 = c; return m; }
DEV M33 transpose(const M33 &m) { return m33(col0(m), col1(m), col2(m)); }
DEV F3 tmul(const M33 &m, F3 v) {  // m^T * v
    return f3(m.r0.x * v.x + m.r1.x * v.y + m.r2.x * v.z, m.r0.y * v.x + m.r1.y * v.y + m.r2.y * v.z,
              m.r0.z * v.x + m.r1.z * v.y + m.r2.z * v.z);
}
DEV M33 mul(const M33 &a, const M33 &b) {  // a * b
    const F3 c0 = col0(b), c1 = col1(b), c2 = col2(b);
    return m33(f3(dot(a.r0, c0), dot(a.r0, c1), dot(a.r0, c2)), f3(dot(a.r1, c0), dot(a.r1, c1), dot(a.r1, c2)),
               f3(dot(a.r2, c0), dot(a.r2, c1), dot(a.r2, c2)));
}
DEV M33 load_m33(const float *p) {
    return m33(f3(p[0], p[1], p[2]), f3(p[3], p[4], p[5]), f3(p[6], p[7], p[8]));
}
DEV F3 load_f3(const float *p) { return f3(p[0], p[1], p[2]); }

struct S33 {  // symmetric 3x3
    float xx, xy, xz, yy, yz, zz;
};
DEV F3 mul(const S33 &s, F3 v) {
    return f3(s.xx * v.x + s.xy * v.y + s.xz * v.z, s.xy * v.x + s.yy * v.y + s.yz * v.z,
              s.xz * v.x + s.yz * v.y + s.zz * v.z);
}
// R * diag(d) * R^T  (btRigidBody::updateInertiaTensor), upper triangle
DEV S33 inertia_world(const M33 &R, F3 d) {
    const F3 a = f3(R.r0.x * d.x, R.r0.y * d.y, R.r0.z * d.z);
    const F3 b = f3(R.r1.x * d.x, R.r1.y * d.y, R.r1.z * d.z);
    const F3 c = f3(R.r2.x * d.x, R.r2.y * d.y, R.r2.z * d.z);
    S33 s;
    s.xx = dot(a, R.r0); s.xy = dot(a, R.r1); s.xz = dot(a, R.r2);
    s.yy = dot(b, R.r1); s.yz = dot(b, R.r2); s.zz = dot(c, R.r2);
    return s;
}

struct Q4 {
    float x, y, z, w;
};
DEV Q4 q4(float x, float y, float z, float w) { Q4 q; q.x = x; q.y = y; q.z = z; q.w = w; return q; }
DEV Q4 qmul(Q4 a, Q4 b) {
    return q4(a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
DEV Q4 qmulv(Q4 q, F3 w) {
    return q4(q.w * w.x + q.y * w.z - q.z * w.y, q.w * w.y + q.z * w.x - q.x * w.z, q.w * w.z + q.x * w.y - q.y * w.x,
              -q.x * w.x - q.y * w.y - q.z * w.z);
}
DEV Q4 qinv(Q4 q) { return q4(-q.x, -q.y, -q.z, q.w); }
DEV F3 quat_rotate(Q4 r, F3 v) {
    Q4 t = qmul(qmulv(r, v), qinv(r));
    return f3(t.x, t.y, t.z);
}
// ---- packed pairs -------------------------------------------------------------------------------------------
// gfx950 issues v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (two fp32 per lane per instruction) at the same rate as
// their scalar forms: the chip's fp32 vector peak is quoted with them.  The two sides of a constraint row (body A
// and body B) run the same arithmetic on different operands, so a row is written once on (A, B) pairs.
typedef float P2 __attribute__((ext_vector_type(2)));
DEV P2 p2(float a, float b) { P2 r; r.x = a; r.y = b; return r; }
struct F3P {
    P2 x, y, z;
};
DEV F3P f3p(P2 x, P2 y, P2 z) { F3P r; r.x = x; r.y = y; r.z = z; return r; }
DEV F3P pair(F3 a, F3 b) { return f3p(p2(a.x, b.x), p2(a.y, b.y), p2(a.z, b.z)); }
DEV F3 lo(const F3P &a) { return f3(a.x.x, a.y.x, a.z.x); }
DEV F3 hi(const F3P &a) { return f3(a.x.y, a.y.y, a.z.y); }
DEV F3P operator+(F3P a, F3P b) { return f3p(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV F3P operator-(F3P a, F3P b) { return f3p(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV F3P operator*(F3P a, P2 s) { return f3p(a.x * s, a.y * s, a.z * s); }
DEV F3P operator*(F3P a, float s) { return f3p(a.x * s, a.y * s, a.z * s); }
DEV P2 dot(F3P a, F3P b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV F3P cross(F3P a, F3P b) { return f3p(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
struct S33P {
    P2 xx, xy, xz, yy, yz, zz;
};
DEV F3P mul(const S33P &s, F3P v) {
    return f3p(s.xx * v.x + s.xy * v.y + s.xz * v.z, s.xy * v.x + s.yy * v.y + s.yz * v.z,
               s.xz * v.x + s.yz * v.y + s.zz * v.z);
}

// Individually rounded multiply / add (no fma contraction).  Used only where a DISCRETE decision of the
// reference algorithm hangs on rounding noise (zero-width slider limit, deepest-vertex choice), so that the
// decision is reproducible from identical inputs.
// (HIP's __fmul_rn/__fadd_rn are plain operators and DO get contracted under -ffp-contract=fast, hence the pragma.)
DEV float xm(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
DEV float xa(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
DEV float xs_(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
DEV P2 xm2(P2 a, P2 b) {
#pragma clang fp contract(off)
    return a * b;
}
DEV P2 xa2(P2 a, P2 b) {
#pragma clang fp contract(off)
    return a + b;
}
DEV float xdot(F3 a, F3 b) { return xa(xa(xm(a.x, b.x), xm(a.y, b.y)), xm(a.z, b.z)); }
// btMatrix3x3::setRotation, every operation individually rounded
DEV M33 mat_from_quat(Q4 q) {
    const float d = xa(xa(xa(xm(q.x, q.x), xm(q.y, q.y)), xm(q.z, q.z)), xm(q.w, q.w));
    const float s = 2.0f / d;
    const float xs = xm(q.x, s), ys = xm(q.y, s), zs = xm(q.z, s);
    const float wx = xm(q.w, xs), wy = xm(q.w, ys), wz = xm(q.w, zs);
    const float xx = xm(q.x, xs), xy = xm(q.x, ys), xz = xm(q.x, zs);
    const float yy = xm(q.y, ys), yz = xm(q.y, zs), zz = xm(q.z, zs);
    return m33(f3(xs_(1.0f, xa(yy, zz)), xs_(xy, wz), xa(xz, wy)), f3(xa(xy, wz), xs_(1.0f, xa(xx, zz)), xs_(yz, wx)),
               f3(xs_(xz, wy), xa(yz, wx), xs_(1.0f, xa(xx, yy))));
}
// E * M0 in glm's column order (glm::mat4 * glm::mat4), individually rounded
DEV M33 glm_mul_basis(const M33 &E, const M33 &M0) {
    const F3 e0 = col0(E), e1 = col1(E), e2 = col2(E);
#define EVM_COL(cx, cy, cz) f3(xa(xa(xm(e0.x, cx), xm(e1.x, cy)), xm(e2.x, cz)), xa(xa(xm(e0.y, cx), xm(e1.y, cy)), xm(e2.y, cz)), \
                               xa(xa(xm(e0.z, cx), xm(e1.z, cy)), xm(e2.z, cz)))
    const F3 k0 = EVM_COL(M0.r0.x, M0.r1.x, M0.r2.x);
    const F3 k1 = EVM_COL(M0.r0.y, M0.r1.y, M0.r2.y);
    const F3 k2 = EVM_COL(M0.r0.z, M0.r1.z, M0.r2.z);
#undef EVM_COL
    return m33(f3(k0.x, k1.x, k2.x), f3(k0.y, k1.y, k2.y), f3(k0.z, k1.z, k2.z));
}
// btMatrix3x3::getRotation, branch-free over the four pivots
DEV Q4 quat_from_mat(const M33 &m) {
    const float m00 = m.r0.x, m11 = m.r1.y, m22 = m.r2.z;
    const float trace = m00 + m11 + m22;
    Q4 q;
    if (trace > 0.0f) {
        float s = sqrtf(trace + 1.0f);
        q.w = s * 0.5f;
        s = 0.5f / s;
        q.x = (m.r2.y - m.r1.z) * s;
        q.y = (m.r0.z - m.r2.x) * s;
        q.z = (m.r1.x - m.r0.y) * s;
    } else {
        const int i = m00 < m11 ? (m11 < m22 ? 2 : 1) : (m00 < m22 ? 2 : 0);
        if (i == 0) {
            float s = sqrtf(m00 - m11 - m22 + 1.0f);
            q.x = s * 0.5f; s = 0.5f / s;
            q.w = (m.r2.y - m.r1.z) * s; q.y = (m.r1.x + m.r0.y) * s; q.z = (m.r2.x + m.r0.z) * s;
        } else if (i == 1) {
            float s = sqrtf(m11 - m22 - m00 + 1.0f);
            q.y = s * 0.5f; s = 0.5f / s;
            q.w = (m.r0.z - m.r2.x) * s; q.z = (m.r2.y + m.r1.z) * s; q.x = (m.r0.y + m.r1.x) * s;
        } else {
            float s = sqrtf(m22 - m00 - m11 + 1.0f);
            q.z = s * 0.5f; s = 0.5f / s;
            q.w = (m.r1.x - m.r0.y) * s; q.x = (m.r0.z + m.r2.x) * s; q.y = (m.r1.z + m.r2.y) * s;
        }
    }
    return q;
}

#define EVM_PI 3.1415926535897932384626433832795029f
#define EVM_2PI (2.0f * EVM_PI)
#define EVM_EPS 1.1920928955078125e-7f
#define EVM_INF 3.402823466e+38f

DEV float norm_angle(float a) {
    a = fmodf(a, EVM_2PI);
    if (a < -EVM_PI) return a + EVM_2PI;
    if (a > EVM_PI) return a - EVM_2PI;
    return a;
}
DEV void plane_space(F3 n, F3 &p, F3 &q) {
    if (fabsf(n.z) > 0.7071067811865475244008443621048490f) {
        const float a = n.y * n.y + n.z * n.z;
        const float k = 1.0f / sqrtf(a);
        p = f3(0.f, -n.z * k, n.y * k);
        q = f3(a * k, -n.x * p.z, n.x * p.y);
    } else {
        const float a = n.x * n.x + n.y * n.y;
        const float k = 1.0f / sqrtf(a);
        p = f3(-n.y * k, n.x * k, 0.f);
        q = f3(-n.z * p.y, n.z * p.x, a * k);
    }
}

// btTransformUtil::integrateTransform on (origin, basis) -> (origin', unit quaternion')
// position part of btTransformUtil::integrateTransform, with the contraction spelled out so that every kernel that
// evaluates it (the integrating wave, and the sweeps kernel predicting the root's motion state) gets the same bits
DEV F3 integ_pos(F3 o, F3 lin, float dt) { return f3(fmaf(lin.x, dt, o.x), fmaf(lin.y, dt, o.y), fmaf(lin.z, dt, o.z)); }
DEV void integrate_transform(F3 o, const M33 &R, F3 lin, F3 ang, float dt, F3 &o2, Q4 &q2) {
    o2 = integ_pos(o, lin, dt);
    const float a2 = len2(ang);
    float fAngle = 0.f;
    if (a2 > EVM_EPS) fAngle = sqrtf(a2);
    const float thr = 0.5f * (EVM_PI * 0.5f);
    if (fAngle * dt > thr) fAngle = thr / dt;
    F3 axis;
    if (fAngle < 0.001f) axis = ang * (0.5f * dt - (dt * dt * dt) * 0.020833333333f * fAngle * fAngle);
    else axis = ang * (sinf(0.5f * fAngle * dt) / fAngle);
    const Q4 dorn = q4(axis.x, axis.y, axis.z, cosf(fAngle * dt * 0.5f));
    const Q4 orn0 = quat_from_mat(R);
    Q4 p = qmul(dorn, orn0);
    const float l2 = p.x * p.x + p.y * p.y + p.z * p.z + p.w * p.w;
    if (l2 > EVM_EPS) {
        const float s = 1.0f / sqrtf(l2);
        p = q4(p.x * s, p.y * s, p.z * s, p.w * s);
    }
    q2 = p;  // (a zero-length product cannot occur for finite inputs: |dorn| ~ 1, |orn0| ~ 1)
}

// ---- no-contraction copies ------------------------------------------------------------------------------------
// btRigidBody::computeGyroscopicImpulseImplicit_Body is ill-conditioned for the (isotropic) attach spheres: the exact
// torque w x (I w) is zero, what the formula returns is its rounding noise divided by I — about 1e-7 |w|^2 rad/s, i.e.
// 1e-3 rad/s on a sphere spinning at 100 rad/s.  Which noise comes out depends on where the compiler fuses multiplies
// into adds, and that differs from kernel to kernel.  These copies are compiled without contraction (like the CPU
// oracle), so every kernel that prepares a body produces the same impulse.
#pragma clang fp contract(off)
namespace nc {
DEV F3 add(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV F3 sub(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV F3 scale(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
DEV float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV F3 cross(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV Q4 qmul(Q4 a, Q4 b) {
    return q4(a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
DEV Q4 qmulv(Q4 q, F3 w) {
    return q4(q.w * w.x + q.y * w.z - q.z * w.y, q.w * w.y + q.z * w.x - q.x * w.z, q.w * w.z + q.x * w.y - q.y * w.x,
              -q.x * w.x - q.y * w.y - q.z * w.z);
}
DEV F3 quat_rotate(Q4 r, F3 v) {
    const Q4 t = nc::qmul(nc::qmulv(r, v), q4(-r.x, -r.y, -r.z, r.w));
    return f3(t.x, t.y, t.z);
}
// omega2 - omega1 of computeGyroscopicImpulseImplicit_Body; q = the body's rotation, idl = local inertia diagonal
DEV F3 gyro_impulse(Q4 q, F3 idl, F3 omega1, float dt) {
    F3 ob = nc::quat_rotate(q4(-q.x, -q.y, -q.z, q.w), omega1);
    const F3 ibo = f3(idl.x * ob.x, idl.y * ob.y, idl.z * ob.z);
    const F3 f = nc::scale(nc::cross(ob, ibo), dt);
    // J = Ib + (skew(ob) * Ib - skew(Ib ob)) * dt, rows
    const F3 j0 = nc::add(f3(idl.x, 0.f, 0.f), nc::scale(nc::sub(f3(0.f * idl.x, -ob.z * idl.y, ob.y * idl.z), f3(0.f, -ibo.z, ibo.y)), dt));
    const F3 j1 = nc::add(f3(0.f, idl.y, 0.f), nc::scale(nc::sub(f3(ob.z * idl.x, 0.f * idl.y, -ob.x * idl.z), f3(ibo.z, 0.f, -ibo.x)), dt));
    const F3 j2 = nc::add(f3(0.f, 0.f, idl.z), nc::scale(nc::sub(f3(-ob.y * idl.x, ob.x * idl.y, 0.f * idl.z), f3(-ibo.y, ibo.x, 0.f)), dt));
    const F3 c1 = f3(j0.x, j1.x, j2.x), c2 = f3(j0.y, j1.y, j2.y), c3 = f3(j0.z, j1.z, j2.z);
    const float det = nc::dot(c1, nc::cross(c2, c3));
    F3 od = f3(0.f, 0.f, 0.f);
    if (fabsf(det) > 1.1920929e-07f)
        od = f3(nc::dot(f, nc::cross(c2, c3)) / det, nc::dot(c1, nc::cross(f, c3)) / det, nc::dot(c1, nc::cross(c2, f)) / det);
    ob = nc::sub(ob, od);
    return nc::sub(nc::quat_rotate(q, ob), omega1);
}
}  // namespace nc
#pragma clang fp contract(fast)

}  // namespace evm
