// dev_math.hpp - small float32 vector / quaternion / spatial-algebra helpers shared by the device kernels.
// (Same conventions as MuJoCo's mju_* routines: quaternions are (w, x, y, z), spatial vectors are angular-then-linear.)
#pragma once
#include <hip/hip_runtime.h>

namespace dm {

struct V3 { float x, y, z; };
struct Q4 { float w, x, y, z; };
struct S6 { float a0, a1, a2, l0, l1, l2; };
typedef float f2 __attribute__((ext_vector_type(2)));  // register pair for v_pk_* arithmetic
struct M3 { float m0, m1, m2, m3, m4, m5, m6, m7, m8; };
struct I10 { float i0, i1, i2, i3, i4, i5, i6, i7, i8, i9; };

__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Q4 qnormalize(Q4 q) {
  float n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  if (n2 < 1e-30f) return {1.f, 0.f, 0.f, 0.f};
  float r = __builtin_amdgcn_rsqf(n2);
  r = r * (1.5f - 0.5f * n2 * r * r);
  return {q.w * r, q.x * r, q.y * r, q.z * r};
}
__device__ __forceinline__ M3 q2m(Q4 q) {
  float w = q.w, x = q.x, y = q.y, z = q.z;
  return {w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y),
          2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x),
          2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z};
}
__device__ __forceinline__ V3 mv(const M3 &m, V3 v) {
  return {m.m0 * v.x + m.m1 * v.y + m.m2 * v.z, m.m3 * v.x + m.m4 * v.y + m.m5 * v.z, m.m6 * v.x + m.m7 * v.y + m.m8 * v.z};
}
__device__ __forceinline__ V3 mtv(const M3 &m, V3 v) {
  return {m.m0 * v.x + m.m3 * v.y + m.m6 * v.z, m.m1 * v.x + m.m4 * v.y + m.m7 * v.z, m.m2 * v.x + m.m5 * v.y + m.m8 * v.z};
}
__device__ __forceinline__ V3 qrot(Q4 q, V3 v) { return mv(q2m(q), v); }
// sin and cos together for |x| up to a few hundred radians (joint half angles are a few radians): Cody-Waite reduction by
// pi/2 in three parts + the classic single-precision minimax polynomials on [-pi/4, pi/4]; max error 9e-8 (1.5 ulp at 1),
// about a third of the instructions of the library routine, whose large-argument path is never needed here.
__device__ __forceinline__ void fsincos(float x, float *sn, float *cs) {
  const float k = rintf(x * 0.6366197723675814f);
  float r = fmaf(-k, 1.5707855224609375f, x);
  r = fmaf(-k, 1.0804334124e-05f, r);
  r = fmaf(-k, 6.0770999344e-11f, r);
  const float r2 = r * r;
  const float s = fmaf(r * r2, fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f), r);
  const float c = fmaf(r2 * r2, fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f), fmaf(-0.5f, r2, 1.f));
  const int q = (int)k & 3;
  const float ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
  *sn = (q & 2) ? -ss : ss;
  *cs = ((q + 1) & 2) ? -cc : cc;
}
__device__ __forceinline__ Q4 axis_angle(V3 ax, float ang) {
  float s, c;
  fsincos(0.5f * ang, &s, &c);
  return {c, ax.x * s, ax.y * s, ax.z * s};
}
__device__ __forceinline__ S6 operator+(S6 a, S6 b) { return {a.a0 + b.a0, a.a1 + b.a1, a.a2 + b.a2, a.l0 + b.l0, a.l1 + b.l1, a.l2 + b.l2}; }
__device__ __forceinline__ S6 operator-(S6 a, S6 b) { return {a.a0 - b.a0, a.a1 - b.a1, a.a2 - b.a2, a.l0 - b.l0, a.l1 - b.l1, a.l2 - b.l2}; }
__device__ __forceinline__ S6 operator*(float s, S6 a) { return {s * a.a0, s * a.a1, s * a.a2, s * a.l0, s * a.l1, s * a.l2}; }
__device__ __forceinline__ float dot6(S6 a, S6 b) { return a.a0 * b.a0 + a.a1 * b.a1 + a.a2 * b.a2 + a.l0 * b.l0 + a.l1 * b.l1 + a.l2 * b.l2; }
__device__ __forceinline__ V3 ang(S6 s) { return {s.a0, s.a1, s.a2}; }
__device__ __forceinline__ V3 lin(S6 s) { return {s.l0, s.l1, s.l2}; }
__device__ __forceinline__ S6 mk6(V3 a, V3 l) { return {a.x, a.y, a.z, l.x, l.y, l.z}; }
__device__ __forceinline__ S6 zero6() { return {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; }
// mj: mju_crossMotion / mju_crossForce
__device__ __forceinline__ S6 cross_motion(S6 vel, S6 v) { return mk6(cross(ang(vel), ang(v)), cross(ang(vel), lin(v)) + cross(lin(vel), ang(v))); }
__device__ __forceinline__ S6 cross_force(S6 vel, S6 f) { return mk6(cross(ang(vel), ang(f)) + cross(lin(vel), lin(f)), cross(ang(vel), lin(f))); }
__device__ __forceinline__ S6 ld6(const float *p) { return {p[0], p[1], p[2], p[3], p[4], p[5]}; }
__device__ __forceinline__ void st6(float *p, S6 s) { p[0] = s.a0; p[1] = s.a1; p[2] = s.a2; p[3] = s.l0; p[4] = s.l1; p[5] = s.l2; }
__device__ __forceinline__ I10 ld10(const float *p) { return {p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9]}; }
__device__ __forceinline__ void st10(float *p, const I10 &i) { p[0] = i.i0; p[1] = i.i1; p[2] = i.i2; p[3] = i.i3; p[4] = i.i4; p[5] = i.i5; p[6] = i.i6; p[7] = i.i7; p[8] = i.i8; p[9] = i.i9; }
__device__ __forceinline__ I10 add10(const I10 &a, const I10 &b) { return {a.i0 + b.i0, a.i1 + b.i1, a.i2 + b.i2, a.i3 + b.i3, a.i4 + b.i4, a.i5 + b.i5, a.i6 + b.i6, a.i7 + b.i7, a.i8 + b.i8, a.i9 + b.i9}; }
// mj: mju_inertCom
__device__ __forceinline__ I10 inert_com(V3 in, const M3 &R, V3 d, float mass) {
  float t0 = R.m0 * in.x, t1 = R.m1 * in.y, t2 = R.m2 * in.z, t3 = R.m3 * in.x, t4 = R.m4 * in.y, t5 = R.m5 * in.z, t6 = R.m6 * in.x,
        t7 = R.m7 * in.y, t8 = R.m8 * in.z;
  float XX = t0 * R.m0 + t1 * R.m1 + t2 * R.m2, YY = t3 * R.m3 + t4 * R.m4 + t5 * R.m5, ZZ = t6 * R.m6 + t7 * R.m7 + t8 * R.m8;
  float XY = t0 * R.m3 + t1 * R.m4 + t2 * R.m5, XZ = t0 * R.m6 + t1 * R.m7 + t2 * R.m8, YZ = t3 * R.m6 + t4 * R.m7 + t5 * R.m8;
  return {XX + mass * (d.y * d.y + d.z * d.z), YY + mass * (d.x * d.x + d.z * d.z), ZZ + mass * (d.x * d.x + d.y * d.y),
          XY - mass * d.x * d.y, XZ - mass * d.x * d.z, YZ - mass * d.y * d.z, mass * d.x, mass * d.y, mass * d.z, mass};
}
// mj: mju_mulInertVec
__device__ __forceinline__ S6 mul_inert(const I10 &i, S6 v) {
  return {i.i0 * v.a0 + i.i3 * v.a1 + i.i4 * v.a2 - i.i8 * v.l1 + i.i7 * v.l2, i.i3 * v.a0 + i.i1 * v.a1 + i.i5 * v.a2 + i.i8 * v.l0 - i.i6 * v.l2,
          i.i4 * v.a0 + i.i5 * v.a1 + i.i2 * v.a2 - i.i7 * v.l0 + i.i6 * v.l1, i.i8 * v.a1 - i.i7 * v.a2 + i.i9 * v.l0,
          i.i6 * v.a2 - i.i8 * v.a0 + i.i9 * v.l1, i.i7 * v.a0 - i.i6 * v.a1 + i.i9 * v.l2};
}
__device__ __forceinline__ float rl_f(float v, int lane_idx) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_idx)); }
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// whole-wave shifts by one lane (DPP wave_shl:1 / wave_shr:1): lane i takes lane i + 1's (i - 1's) value; the end lane gets 0
__device__ __forceinline__ float wshl1(float v) { return dpp_move<0x130>(v); }
__device__ __forceinline__ float wshr1(float v) { return dpp_move<0x138>(v); }
// full-wave sum, result in every lane (DPP folds inside each row of 16, then four v_readlane)
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  v += dpp_move<0x140>(v);
  return (rl_f(v, 0) + rl_f(v, 16)) + (rl_f(v, 32) + rl_f(v, 48));
}

// sum within each row of 16 lanes, result in every lane of the row
__device__ __forceinline__ float row_sum(float v) {
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  v += dpp_move<0x140>(v);
  return v;
}
// 1/x: hardware reciprocal + one Newton step (full float32 accuracy without the IEEE division sequence)
__device__ __forceinline__ float frcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  return r * (2.f - x * r);
}
// sqrt(x) as x * rsq(x) with one Newton step on the reciprocal square root (full float32 accuracy, no IEEE fix-up code)
__device__ __forceinline__ float fsqrt(float x) {
  if (!(x > 0.f)) return 0.f;
  float r = __builtin_amdgcn_rsqf(x);
  r = r * (1.5f - 0.5f * x * r * r);
  return x * r;
}

// One workgroup = one wavefront: LDS operations of a wave execute in issue order, so publishing a lane's LDS write to
// the other lanes only needs the compiler not to reorder across this point.
#ifdef FFE_LOCAL_FENCE
#define DM_SYNC()                                                \
  do {                                                           \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local"); \
    __builtin_amdgcn_wave_barrier();                             \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local"); \
  } while (0)
#else
#define DM_SYNC()                                                \
  do {                                                           \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       \
    __builtin_amdgcn_wave_barrier();                             \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       \
  } while (0)
#endif

}  // namespace dm
