// Forward-mode derivative of the per-joint map of the physics post-optimisation (optimization.py:56 +
// smpl_layer.py:88): rotation matrix variable -> matrix_to_axis_angle (pytorch3d 0.7.2) -> the SMPL layer's own
// Rodrigues (rodrigues_layer.py:13-52) -> the rotation the kinematic chain uses.  Nine tangents ride along the value
// (one per input entry), which yields the 9x9 Jacobian in one pass; the reverse product with dL/dR' is then a dot.
// Sub-gradient conventions follow torch autograd at the non-smooth points the data really hits (identity hand joints):
// sqrt_positive_part and ||.|| have a ZERO derivative at 0, max(a, floor) passes the derivative of the larger argument.
#pragma once
#include <hip/hip_runtime.h>

namespace rotd {

constexpr int NT = 9;

struct Dual {
    float v;
    float d[NT];
};

#define ROTD_FN __host__ __device__ __forceinline__

ROTD_FN Dual cst(float c) {
    Dual r;
    r.v = c;
#pragma unroll
    for (int i = 0; i < NT; ++i) r.d[i] = 0.f;
    return r;
}
ROTD_FN Dual var(float c, int k) {
    Dual r = cst(c);
    r.d[k] = 1.f;
    return r;
}
ROTD_FN Dual operator+(const Dual &a, const Dual &b) {
    Dual r;
    r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < NT; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
ROTD_FN Dual operator-(const Dual &a, const Dual &b) {
    Dual r;
    r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < NT; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
ROTD_FN Dual operator*(const Dual &a, const Dual &b) {
    Dual r;
    r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < NT; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
ROTD_FN Dual operator/(const Dual &a, const Dual &b) {
    Dual r;
    const float inv = 1.0f / b.v;
    r.v = a.v * inv;
#pragma unroll
    for (int i = 0; i < NT; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
ROTD_FN Dual operator+(const Dual &a, float c) { Dual r = a; r.v += c; return r; }
ROTD_FN Dual operator*(const Dual &a, float c) {
    Dual r;
    r.v = a.v * c;
#pragma unroll
    for (int i = 0; i < NT; ++i) r.d[i] = a.d[i] * c;
    return r;
}
ROTD_FN Dual chain(const Dual &a, float val, float dval) {       // f(a) with f' known
    Dual r;
    r.v = val;
#pragma unroll
    for (int i = 0; i < NT; ++i) r.d[i] = a.d[i] * dval;
    return r;
}
ROTD_FN Dual sqrt_pos(const Dual &a) {                           // _sqrt_positive_part
    if (!(a.v > 0.f)) return cst(0.f);
    const float s = sqrtf(a.v);
    return chain(a, s, 0.5f / s);
}
ROTD_FN Dual norm3(const Dual &x, const Dual &y, const Dual &z) {   // torch.norm: zero sub-gradient at the origin
    const float n = sqrtf(x.v * x.v + y.v * y.v + z.v * z.v);
    Dual r;
    r.v = n;
    const float inv = n > 0.f ? 1.0f / n : 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) r.d[i] = (x.v * x.d[i] + y.v * y.d[i] + z.v * z.d[i]) * inv;
    return r;
}
ROTD_FN Dual sinc_half(const Dual &ang, const Dual &half) {      // sin(ang/2)/ang, series below 1e-6
    if (fabsf(ang.v) < 1e-6f) return chain(ang, 0.5f - ang.v * ang.v / 48.0f, -ang.v / 24.0f);
    return chain(half, sinf(half.v), cosf(half.v)) / ang;
}

// R (nine variables) -> quaternion (four-candidate method, floor 0.1, first maximum wins)
ROTD_FN void matrix_to_quaternion(const Dual *m, Dual *q) {
    const Dual one = cst(1.0f);
    const Dual a0 = sqrt_pos(one + m[0] + m[4] + m[8]), a1 = sqrt_pos(one + m[0] - m[4] - m[8]);
    const Dual a2 = sqrt_pos(one - m[0] + m[4] - m[8]), a3 = sqrt_pos(one - m[0] - m[4] + m[8]);
    int pick = 0;
    Dual best = a0;
    if (a1.v > best.v) { best = a1; pick = 1; }
    if (a2.v > best.v) { best = a2; pick = 2; }
    if (a3.v > best.v) { best = a3; pick = 3; }
    const Dual den = (best.v > 0.1f ? best : cst(0.1f)) * 2.0f;
    Dual c0, c1, c2, c3;
    if (pick == 0)      { c0 = a0 * a0;     c1 = m[7] - m[5]; c2 = m[2] - m[6]; c3 = m[3] - m[1]; }
    else if (pick == 1) { c0 = m[7] - m[5]; c1 = a1 * a1;     c2 = m[3] + m[1]; c3 = m[2] + m[6]; }
    else if (pick == 2) { c0 = m[2] - m[6]; c1 = m[3] + m[1]; c2 = a2 * a2;     c3 = m[5] + m[7]; }
    else                { c0 = m[3] - m[1]; c1 = m[6] + m[2]; c2 = m[7] + m[5]; c3 = a3 * a3; }
    q[0] = c0 / den; q[1] = c1 / den; q[2] = c2 / den; q[3] = c3 / den;
}

ROTD_FN void quaternion_to_axis_angle(const Dual *q, Dual *a) {
    const Dual n = norm3(q[1], q[2], q[3]);
    const float den = n.v * n.v + q[0].v * q[0].v;
    Dual half;                                                    // atan2(n, w)
    half.v = atan2f(n.v, q[0].v);
#pragma unroll
    for (int i = 0; i < NT; ++i) half.d[i] = (q[0].v * n.d[i] - n.v * q[0].d[i]) / den;
    const Dual ang = half * 2.0f;
    const Dual s = sinc_half(ang, half);
    a[0] = q[1] / s; a[1] = q[2] / s; a[2] = q[3] / s;
}

// SMPL layer's Rodrigues: theta = ||aa + 1e-8||, axis = aa / theta, renormalised quaternion -> matrix
ROTD_FN void rodrigues_smpl(const Dual *a, Dual *m) {
    const Dual ang = norm3(a[0] + 1e-8f, a[1] + 1e-8f, a[2] + 1e-8f);
    const Dual half = ang * 0.5f;
    const Dual sn = chain(half, sinf(half.v), cosf(half.v));
    Dual w = chain(half, cosf(half.v), -sinf(half.v));
    Dual x = sn * (a[0] / ang), y = sn * (a[1] / ang), z = sn * (a[2] / ang);
    const Dual n2 = w * w + x * x + y * y + z * z;
    const Dual nq = chain(n2, sqrtf(n2.v), 0.5f / sqrtf(n2.v));
    w = w / nq; x = x / nq; y = y / nq; z = z / nq;
    const Dual w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
    const Dual wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
    m[0] = w2 + x2 - y2 - z2;   m[1] = (xy - wz) * 2.0f;     m[2] = (wy + xz) * 2.0f;
    m[3] = (wz + xy) * 2.0f;    m[4] = w2 - x2 + y2 - z2;    m[5] = (yz - wx) * 2.0f;
    m[6] = (xz - wy) * 2.0f;    m[7] = (wx + yz) * 2.0f;     m[8] = w2 - x2 - y2 + z2;
}

// g_in[k] = sum_e g_out[e] * d R'_e / d R_k   for R' = rodrigues_smpl(matrix_to_axis_angle(R))
ROTD_FN void joint_map_vjp(const float *R, const float *g_out, float *g_in) {
    Dual m[9], q[4], a[3], o[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = var(R[k], k);
    matrix_to_quaternion(m, q);
    quaternion_to_axis_angle(q, a);
    rodrigues_smpl(a, o);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 9; ++e) s += g_out[e] * o[e].d[k];
        g_in[k] = s;
    }
}

}  // namespace rotd
