// Rotation conversions (device inline), semantics of pytorch3d.transforms 0.7.2 as used by the
// reference (eval_smpl_short.py:18,90-91,157-162; diffusion_smpl.py:212-213) plus the SMPL
// layer's own Rodrigues (libsmpl/smplpytorch/pytorch/rodrigues_layer.py:13-52).
// Quaternions are (w,x,y,z); matrices row-major m[9]; rot6d = first two rows.
#pragma once
#include <hip/hip_runtime.h>

namespace rot {

__device__ __forceinline__ void normalize3(float &x, float &y, float &z) {   // F.normalize, eps 1e-12
    const float n = fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);
    x /= n; y /= n; z /= n;
}

__device__ __forceinline__ void rot6d_to_matrix(const float *d, float *m) {
    float b1x = d[0], b1y = d[1], b1z = d[2];
    normalize3(b1x, b1y, b1z);
    const float dt = b1x * d[3] + b1y * d[4] + b1z * d[5];
    float b2x = d[3] - dt * b1x, b2y = d[4] - dt * b1y, b2z = d[5] - dt * b1z;
    normalize3(b2x, b2y, b2z);
    m[0] = b1x; m[1] = b1y; m[2] = b1z;
    m[3] = b2x; m[4] = b2y; m[5] = b2z;
    m[6] = b1y * b2z - b1z * b2y;
    m[7] = b1z * b2x - b1x * b2z;
    m[8] = b1x * b2y - b1y * b2x;
}

__device__ __forceinline__ float sinc_half(float ang, float half) {       // sin(ang/2)/ang
    return fabsf(ang) < 1e-6f ? 0.5f - ang * ang / 48.0f : sinf(half) / ang;
}

__device__ __forceinline__ void axis_angle_to_quaternion(const float *a, float *q) {
    const float ang = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    const float half = ang * 0.5f, s = sinc_half(ang, half);
    q[0] = cosf(half); q[1] = a[0] * s; q[2] = a[1] * s; q[3] = a[2] * s;
}

__device__ __forceinline__ void quaternion_to_matrix(const float *q, float *m) {
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float s2 = 2.0f / (r * r + i * i + j * j + k * k);
    m[0] = 1 - s2 * (j * j + k * k); m[1] = s2 * (i * j - k * r); m[2] = s2 * (i * k + j * r);
    m[3] = s2 * (i * j + k * r); m[4] = 1 - s2 * (i * i + k * k); m[5] = s2 * (j * k - i * r);
    m[6] = s2 * (i * k - j * r); m[7] = s2 * (j * k + i * r); m[8] = 1 - s2 * (i * i + j * j);
}

// four-candidate method, floor 0.1, first maximum wins, w sign NOT standardised (0.7.2)
__device__ __forceinline__ void matrix_to_quaternion(const float *m, float *q) {
    const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
    const float a0 = sqrtf(fmaxf(1.0f + m00 + m11 + m22, 0.f)), a1 = sqrtf(fmaxf(1.0f + m00 - m11 - m22, 0.f));
    const float a2 = sqrtf(fmaxf(1.0f - m00 + m11 - m22, 0.f)), a3 = sqrtf(fmaxf(1.0f - m00 - m11 + m22, 0.f));
    int pick = 0;
    float best = a0;
    if (a1 > best) { best = a1; pick = 1; }
    if (a2 > best) { best = a2; pick = 2; }
    if (a3 > best) { best = a3; pick = 3; }
    const float den = 2.0f * fmaxf(best, 0.1f);
    float c0, c1, c2, c3;
    if (pick == 0)      { c0 = a0 * a0;  c1 = m21 - m12; c2 = m02 - m20; c3 = m10 - m01; }
    else if (pick == 1) { c0 = m21 - m12; c1 = a1 * a1;  c2 = m10 + m01; c3 = m02 + m20; }
    else if (pick == 2) { c0 = m02 - m20; c1 = m10 + m01; c2 = a2 * a2;  c3 = m12 + m21; }
    else                { c0 = m10 - m01; c1 = m20 + m02; c2 = m21 + m12; c3 = a3 * a3; }
    q[0] = c0 / den; q[1] = c1 / den; q[2] = c2 / den; q[3] = c3 / den;
}

__device__ __forceinline__ void quaternion_to_axis_angle(const float *q, float *a) {
    const float n = sqrtf(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float half = atan2f(n, q[0]), ang = 2.0f * half, s = sinc_half(ang, half);
    a[0] = q[1] / s; a[1] = q[2] / s; a[2] = q[3] / s;
}

__device__ __forceinline__ void matrix_to_axis_angle(const float *m, float *a) {
    float q[4];
    matrix_to_quaternion(m, q);
    quaternion_to_axis_angle(q, a);
}

// SMPL layer's Rodrigues: theta = ||aa + 1e-8||, axis = aa / theta, renormalised quaternion
__device__ __forceinline__ void rodrigues_smpl(const float *a, float *m) {
    const float ex = a[0] + 1e-8f, ey = a[1] + 1e-8f, ez = a[2] + 1e-8f;
    const float ang = sqrtf(ex * ex + ey * ey + ez * ez);
    const float half = ang * 0.5f, sn = sinf(half);
    float w = cosf(half), x = sn * (a[0] / ang), y = sn * (a[1] / ang), z = sn * (a[2] / ang);
    const float nq = sqrtf(w * w + x * x + y * y + z * z);
    w /= nq; x /= nq; y /= nq; z /= nq;
    const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
    const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
    m[0] = w2 + x2 - y2 - z2; m[1] = 2 * xy - 2 * wz;    m[2] = 2 * wy + 2 * xz;
    m[3] = 2 * wz + 2 * xy;    m[4] = w2 - x2 + y2 - z2; m[5] = 2 * yz - 2 * wx;
    m[6] = 2 * xz - 2 * wy;    m[7] = 2 * wx + 2 * yz;    m[8] = w2 - x2 - y2 + z2;
}

}  // namespace rot
