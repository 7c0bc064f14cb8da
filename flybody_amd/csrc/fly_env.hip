// fly_env.hip - MI355X (gfx950) batched fruit-fly environment: HIP kernels + the C ABI of
// include/flybody_env.h.
//
// Execution model: ONE 64-lane wavefront per environment instance, one workgroup = one wavefront, one
// launch = one control step of every env (task pre-step, nsub physics substeps, observation, reward,
// termination, auto-reset), so an env's state leaves the chip once per control step.
//   lanes <-> dofs (42)            for joint-space work (forces, limits, integration),
//   lanes <-> links (19)           for body-space work (kinematics, spatial inertia, velocities, forces),
//   lanes <-> fluid records (49)   for the inertia-box drag of the bodies welded to the thorax,
//   lanes <-> M entries (421, 7/lane) for the joint-space inertia,
// with per-env tiles (spatial quantities, M, its factor) staged in LDS and tree passes written as
// gather-sums over precomputed ancestor / subtree ranges so that a pass needs one LDS barrier, not one
// per tree level.  Reference semantics being reproduced are cited inline ("ref:" = /root/reference/vnl_ray,
// "mj:" = the MuJoCo stage the reference reaches through dm_control).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/flybody_env.h"
#include "ball_env.hpp"
#include "dev_model.hpp"
#include "launch_order.hpp"
#include "dev_math.hpp"
#include "convex.hpp"

#ifndef FFE_WAVES_PER_SIMD
#define FFE_WAVES_PER_SIMD 4  // register budget = 512 / this; picked by measurement: with the LDS tile under 10 KB, 16 waves fit a CU
#endif

namespace ffe {

constexpr int kMC = 6;   // contacts per env the solver carries (deepest kept; more is flagged)
constexpr int kNSD = 16;  // convex pairs whose separating direction is remembered from substep to substep (a wing near the abdomen brings six
                         // ellipsoid - cylinder pairs at once, whose search from scratch is the most expensive thing a wave can do: 25 - 100 us)

// ------------------------------------------------------------------------------------------------ state
struct alignas(64) EnvState {
  double rootpos[3];
  double wb_ctrl_freq;
  double ghost[8];  // pos[3], quat[4], pad
  double forced_phase;
  unsigned long long episode, lo_mask, hi_mask;
  float qpos[kMaxDof + 4];  // [0..2] unused (root position is held in float64 above), [3..6] root quat, hinges
  float qvel[kMaxDof + 4];
  int wb_step, wb_freq_idx, step_counter, traj_idx, needs_reset, nactive, solver_iters, forced_traj;
  // Position/velocity-stage results of the state above (dof axes, crb*axes, smooth joint forces, root frame, CoM): the
  // last stage-1 evaluation of a control step is exactly the first one of the next, so it is carried over instead of
  // being recomputed (bit-identical by construction; dropped whenever the state is written from outside).
  int s1_valid;
  int wb_off, wb_len, traj_row0;  // K.tab_off[wb_freq_idx], that table's length, K.traj_off[traj_idx]: kept with the state so that the
                                  // next launch can address its table / reference rows straight from the record
  unsigned long long in_lo_mask, in_hi_mask;  // limits instantiated but resolved inactive by the last substep's solve (first guess of the next)
  unsigned char cost_hist[32];  // work estimate of the last control step that ended in each of 32 wing-beat phase bins (launch order)
  float s1_cdof[kMaxDof * 6], s1_buf[kMaxDof * 6], s1_f[kLanePad], s1_misc[16];
  // contacts of the carried-over position stage (lane k = contact k) and the convex pairs' separating directions (flight_collide)
  float ct_f[kMC][9];
  int ct_i[kMC][3];
  float sd_n[kNSD][4];  // direction | the capsule's axis parameter of a capsule - convex pair
  int sd_pid[kNSD];
  int nct, sd_cnt, ct_pad[2];
};

struct TaskDev {
  int nfreq, ntraj, future_steps, time_limit_steps, episode_limit_steps, pad_first_obs, flags, obs_dim, canonical, clip;
  float act_lo[16], act_hi[16];
  double base_freq, rel_range, rate, dt_ctrl, terminal_com_dist, ghost_accel_z;
  double grid_inv_step;  // 1 / spacing of the beat-frequency grid when it is evenly spaced (np.linspace), else 0
  const double FFE_GLOBAL *beat_freqs, *phase, *phase_frac, *ref_qpos, *ref_qvel;
  const float FFE_GLOBAL *traj;
  const int FFE_GLOBAL *tab_off;
  const int FFE_GLOBAL *traj_off;  // [ntraj + 1] first reference row of each trajectory (ref: trajectory_loaders.py:98-100 ragged lengths)
  unsigned long long seed, env_id_base;
};

// ------------------------------------------------------------------------------------------------ LDS tile
struct alignas(16) Tile {
  float qpos[kMaxDof + 4];
  float qvel[kMaxDof + 4];
  float xpos[kMaxLink][3];    // world (root-relative) link origin
  float xmat[kMaxLink][9];    // world link orientation
  union {
    float cinert[kMaxLink][10]; // link spatial inertia about the com reference point (until the subtree sums)
    float crb[kMaxLink][10];    // composite (subtree) inertia (after them)
  };
  float cdof[kMaxDof][6];     // dof axes, com-centred world frame
  union {
    float cdofd[kMaxDof][6];  // cdof_dot; dead once the per-link sums A1 are taken
    float buf[kMaxDof][6];    // crb * cdof
  };
  union {
    float lT[kMaxLink][7];    // link frame in its parent link: pos[3], quat[4]; dead after K2
    float la[kMaxLink][6];    // per-link scratch
  };
  float lb[kMaxLink][6];
  float lc[kMaxLink][6];
  float2 LD[kMaxM];           // two factorisations side by side: .x = constraint Hessian, .y = M + h B (Euler)
  float frc[kMaxAct];    // actuator forces
  float ctrl[kMaxAct];   // control staging
  unsigned short colmadr[kMaxM + 8];  // for entry e = (row k, column a): start of row a in M/LD
  float sens[12];    // running sums of the buffered sensors: acc[3], gyro[3], vel[3]
  double rootpos[4];  // free-joint position in float64
  double ghost[16];   // pos[3], quat[4], vel[3], angvel[3]
  double park_d[2];   // wave-uniform task values parked over the physics loop (see flight_step_kernel)
  int park_i[8];
};

static_assert(offsetof(Tile, cdof) % 8 == 0 && offsetof(Tile, buf) % 8 == 0 && sizeof(float[6]) % 8 == 0, "ld6a needs 8-byte aligned 6-vector rows");

// ------------------------------------------------------------------------------------------------ maths
struct V3 { float x, y, z; };
struct Q4 { float w, x, y, z; };
struct S6 { float a0, a1, a2, l0, l1, l2; };  // spatial vector: angular then linear
struct M3 { float m0, m1, m2, m3, m4, m5, m6, m7, m8; };

__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Q4 qconj(Q4 q) { return {q.w, -q.x, -q.y, -q.z}; }
__device__ __forceinline__ Q4 qnormalize(Q4 q) {
  float n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  if (n2 < 1e-30f) return {1.f, 0.f, 0.f, 0.f};
  float r = __builtin_amdgcn_rsqf(n2);
  r = r * (1.5f - 0.5f * n2 * r * r);  // one Newton step: full float32 accuracy
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
__device__ __forceinline__ M3 mm(const M3 &a, const M3 &b) {
  return {a.m0 * b.m0 + a.m1 * b.m3 + a.m2 * b.m6, a.m0 * b.m1 + a.m1 * b.m4 + a.m2 * b.m7, a.m0 * b.m2 + a.m1 * b.m5 + a.m2 * b.m8,
          a.m3 * b.m0 + a.m4 * b.m3 + a.m5 * b.m6, a.m3 * b.m1 + a.m4 * b.m4 + a.m5 * b.m7, a.m3 * b.m2 + a.m4 * b.m5 + a.m5 * b.m8,
          a.m6 * b.m0 + a.m7 * b.m3 + a.m8 * b.m6, a.m6 * b.m1 + a.m7 * b.m4 + a.m8 * b.m7, a.m6 * b.m2 + a.m7 * b.m5 + a.m8 * b.m8};
}
__device__ __forceinline__ V3 qrot(Q4 q, V3 v) { return mv(q2m(q), v); }  // mj: mju_rotVecQuat
// sin/cos of a half joint angle.  Joint ranges keep |h| below ~1.6 rad (head twist +-3, wing pitch 2.92), where the
// Taylor polynomials below are accurate to 2e-7 (sin, degree 11) and 2e-8 (cos, degree 12); anything larger takes
// the library path.
__device__ __forceinline__ void sincos_half(float h, float *s, float *c) {
  if (fabsf(h) > 1.7f) { sincosf(h, s, c); return; }
  const float x2 = h * h;
  float ps = fmaf(x2, -2.50521084e-8f, 2.75573192e-6f);
  ps = fmaf(x2, ps, -1.98412698e-4f);
  ps = fmaf(x2, ps, 8.33333333e-3f);
  ps = fmaf(x2, ps, -1.66666667e-1f);
  *s = fmaf(h * x2, ps, h);
  float pc = fmaf(x2, 2.08767570e-9f, -2.75573192e-7f);
  pc = fmaf(x2, pc, 2.48015873e-5f);
  pc = fmaf(x2, pc, -1.38888889e-3f);
  pc = fmaf(x2, pc, 4.16666667e-2f);
  pc = fmaf(x2, pc, -0.5f);
  *c = fmaf(x2, pc, 1.0f);
}
__device__ __forceinline__ Q4 axis_angle(V3 ax, float ang) {
  float s, c;
  sincos_half(0.5f * ang, &s, &c);
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
__device__ __forceinline__ M3 ldm(const float *p) { return {p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]}; }
__device__ __forceinline__ M3 ldm_lane(const float FFE_GLOBAL *tab, int lane) {
  return {tab[0 * kLanePad + lane], tab[1 * kLanePad + lane], tab[2 * kLanePad + lane], tab[3 * kLanePad + lane], tab[4 * kLanePad + lane],
          tab[5 * kLanePad + lane], tab[6 * kLanePad + lane], tab[7 * kLanePad + lane], tab[8 * kLanePad + lane]};
}
__device__ __forceinline__ V3 ldv_lane(const float FFE_GLOBAL *tab, int lane) { return {tab[lane], tab[kLanePad + lane], tab[2 * kLanePad + lane]}; }

struct I10 { float i0, i1, i2, i3, i4, i5, i6, i7, i8, i9; };
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
// 6-vector rows of cdof / buf start on 8-byte boundaries (24-byte rows at 8-aligned offsets of the 16-aligned tile): three
// ds_read_b64 instead of six ds_read_b32
__device__ __forceinline__ S6 ld6a(const float *p) {
  const float2 *q = reinterpret_cast<const float2 *>(p);
  const float2 a = q[0], b = q[1], c = q[2];
  return {a.x, a.y, b.x, b.y, c.x, c.y};
}
__device__ __forceinline__ int rl_i(int v, int lane_idx) { return __builtin_amdgcn_readlane(v, lane_idx); }
__device__ __forceinline__ float rl_f(float v, int lane_idx) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_idx)); }
// Full-wave sum, result in every lane.  Four DPP steps fold each row of 16 lanes (quad swaps, half-row mirror, row
// mirror), then the four row totals are read with v_readlane: no LDS traffic, ~11 instructions.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror
  return (rl_f(v, 0) + rl_f(v, 16)) + (rl_f(v, 32) + rl_f(v, 48));
}
__device__ __forceinline__ S6 wave_sum6(S6 s) { return {wave_sum(s.a0), wave_sum(s.a1), wave_sum(s.a2), wave_sum(s.l0), wave_sum(s.l1), wave_sum(s.l2)}; }

// ref: quaternions.py:88-102 / :46-69 / :105-134 / :273-295 / :13-17 (float32 device versions)
__device__ __forceinline__ Q4 quat_recip(Q4 q) {
  float n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  return {q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
}
__device__ __forceinline__ V3 rotate_vec_with_quat(V3 v, Q4 q) {
  Q4 r = qmul(q, qmul(Q4{0.f, v.x, v.y, v.z}, quat_recip(q)));
  return {r.x, r.y, r.z};
}
__device__ __forceinline__ float quat_dist_short_arc(Q4 a, Q4 b) {
  // The reference evaluates acos(min(1, 2 (p.q)^2 - 1)) in float64.  In float32 that form loses half the digits
  // near zero angle (d acos/dx blows up at x = 1), so the same angle is taken from the relative quaternion
  // dq = conj(a) b as 2 atan2(|vec dq|, |w dq|), which is identical for unit quaternions and well conditioned.
  Q4 d = qmul(qconj(a), b);
  float v = sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
  return 2.0f * atan2f(v, fabsf(d.w));
}

// mj: inertia-box fluid model (mj_inertiaBoxFluidModel) for one body whose inertial frame sits at `rpos` with axes
// `rmat`, both in the coordinates of the link that carries it.  (w_b, v_b) = the link's angular velocity and the linear
// velocity of the link origin, in link axes.  Returns (torque about the link origin, force) in link axes, so the
// bodies welded to one link can be summed before a single rotation to the world.
__device__ __forceinline__ S6 box_fluid_local(const float FFE_GLOBAL *coef, int lane, V3 rpos, const M3 &rmat, V3 w_b, V3 v_b) {
  V3 lw = mtv(rmat, w_b), lv = mtv(rmat, v_b + cross(w_b, rpos));
  float c0 = coef[0 * kLanePad + lane], c1 = coef[1 * kLanePad + lane];
  V3 lt = {-c0 * lw.x - coef[5 * kLanePad + lane] * fabsf(lw.x) * lw.x, -c0 * lw.y - coef[6 * kLanePad + lane] * fabsf(lw.y) * lw.y,
           -c0 * lw.z - coef[7 * kLanePad + lane] * fabsf(lw.z) * lw.z};
  V3 lf = {-c1 * lv.x - coef[2 * kLanePad + lane] * fabsf(lv.x) * lv.x, -c1 * lv.y - coef[3 * kLanePad + lane] * fabsf(lv.y) * lv.y,
           -c1 * lv.z - coef[4 * kLanePad + lane] * fabsf(lv.z) * lv.z};
  V3 f = mv(rmat, lf);
  return mk6(mv(rmat, lt) + cross(rpos, f), f);
}
// mj: ellipsoid fluid model (mj_ellipsoidFluidModel + mj_addedMassForces + mj_viscousForces), same conventions
__device__ __forceinline__ S6 ell_fluid_local(const float FFE_GLOBAL *e, V3 rpos, const M3 &rmat, V3 w_b, V3 v_b) {
  V3 w = mtv(rmat, w_b), v = mtv(rmat, v_b + cross(w_b, rpos));
  V3 plin = {e[1] * v.x, e[2] * v.y, e[3] * v.z}, pang = {e[4] * w.x, e[5] * w.y, e[6] * w.z};
  V3 f = cross(plin, w);
  V3 t = cross(plin, v) + cross(pang, w);
  f = f + e[7] * cross(w, v);  // Magnus
  float p0 = e[8], p1 = e[9], p2 = e[10];
  float pd = p0 * p0 * p0 * p0 * v.x * v.x + p1 * p1 * p1 * p1 * v.y * v.y + p2 * p2 * p2 * p2 * v.z * v.z;
  float pn = (p0 * v.x) * (p0 * v.x) + (p1 * v.y) * (p1 * v.y) + (p2 * v.z) * (p2 * v.z);
  const float kPi = 3.14159265358979f;
  float A_proj = kPi * sqrtf(pd / fmaxf(1e-15f, pn));
  V3 nrm = {p0 * p0 * v.x, p1 * p1 * v.y, p2 * p2 * v.z};
  float speed = sqrtf(dot(v, v));
  float cos_alpha = pn / fmaxf(1e-15f, speed * pd);
  V3 circ = (e[11] * cos_alpha * A_proj) * cross(nrm, v);
  f = f + cross(circ, v);  // Kutta
  V3 mom = {w.x * (e[15] * e[19] + e[14] * (e[18] - e[19])), w.y * (e[15] * e[20] + e[14] * (e[18] - e[20])), w.z * (e[15] * e[21] + e[14] * (e[18] - e[21]))};
  float drag_lin = e[16] + e[22] * speed * (A_proj * e[13] + e[14] * (e[12] - A_proj));
  float drag_ang = e[17] + e[22] * sqrtf(dot(mom, mom));
  t = t - drag_ang * w;
  f = f - drag_lin * v;
  t = e[0] * t; f = e[0] * f;
  V3 fb = mv(rmat, f);
  return mk6(mv(rmat, t) + cross(rpos, fb), fb);
}

// mj: getimpedance.  x^p: the model's solimp power is 2 (MuJoCo default); other exponents go through exp2/log2.
__device__ __forceinline__ float pow_pos(float x, float p) {
  if (p == 2.f) return x * x;
  if (p == 1.f) return x;
  if (p == 3.f) return x * x * x;
  return __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(x));
}
__device__ __forceinline__ float impedance(float dmin, float dmax, float width, float mid, float power, float pos, float margin) {
  if (dmin == dmax || width <= 1e-15f) return 0.5f * (dmin + dmax);
  float x = fabsf((pos - margin) * __builtin_amdgcn_rcpf(width));
  if (x >= 1.f) return dmax;
  if (x <= 0.f) return dmin;
  float y;
  if (power == 1.f) y = x;
  else if (x <= mid) y = pow_pos(x, power) * __builtin_amdgcn_rcpf(pow_pos(mid, power - 1.f));
  else y = 1.f - pow_pos(1.f - x, power) * __builtin_amdgcn_rcpf(pow_pos(1.f - mid, power - 1.f));
  return dmin + y * (dmax - dmin);
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}
__device__ __forceinline__ unsigned long long env_rng(unsigned long long seed, unsigned long long env, unsigned long long episode, unsigned long long stream) {
  return splitmix64(splitmix64(splitmix64(seed ^ 0xF1B0D7ULL) + env) + (episode << 2) + stream);
}

// ------------------------------------------------------------------------------------------------ per-wave context
struct Ctx {
  const DevModel FFE_CONST *Mp;
  Tile &T;
  int lane;
  int flags;
  float f_smooth_nb;  // passive(spring+damper) - bias + fluid per dof lane (actuation is added in stage 2)
  float qacc;         // constrained acceleration (mj: d->qacc)
  float dinv[2];      // 1 / D of this lane's dof for the two resident factorisations
  unsigned la_pack;   // this lane's row start | depth << 10 | descendant count << 16 (read by every factor / solve)
  unsigned seq0, seq1, seq2, seq3;  // elimination order of the lane's branch, a byte per step (DevModel::br_seq)
  // contacts of the current position stage: lane k < nct holds contact k (normal from geom1's link to geom2's, position relative to
  // the root, distance, includemargin, sum of the two bodies' inverse weights, the two links, pair id)
  float ct_nx, ct_ny, ct_nz, ct_px, ct_py, ct_pz, ct_dist, ct_incl, ct_invw;
  int ct_l1, ct_l2, ct_pid, nct, ct_ovf;
  int nrare, nfac;  // second-pass collision calls and factorisations of this control step (wave-uniform): the launch-order key
  int pc;  // lane k: pair id of contact k of the previous substep's solve | 0x10000 if it was resolved inactive (-1: none): first guess of the next solve
  // separating-direction cache of the convex pairs: lane k < sd_cnt holds an entry
  float sd_nx, sd_ny, sd_nz, sd_t;
  int sd_pid, sd_cnt;
#ifdef FFE_STAMPS
  unsigned long long st_t0, st_acc[20];
#endif
#ifdef FFE_DBGCF
  int dbg_env;
#endif
#ifdef FFE_TRACE
  unsigned tr_coll;  // second-pass collision calls, factorisations, 100 MHz ticks spent in the collision stage
#endif
};

// Per-lane model constants are (re)read from the lane-major tables where they are used instead of being pinned in
// registers across the whole step: the tables are L1/L2 resident, the reads coalesce, and the register allocator is
// left with short live ranges in the dependent loops.  The pointer is laundered so the loads are not hoisted back.
__device__ __forceinline__ const DevModel FFE_CONST &model(const Ctx &c) {
  const DevModel FFE_CONST *m = c.Mp;
  asm volatile("" : "+s"(m));
  return *m;
}

// One workgroup = one wavefront, and a wavefront's LDS operations execute in issue order, so making one lane's LDS
// write visible to another lane needs no hardware wait at all - only the compiler must not reorder across the point.
#if defined(FFE_HW_SYNC)
#define SYNC() __syncthreads()
#elif defined(FFE_LOCAL_FENCE)
// experiment: order LDS traffic only, so that global (model table) loads may be scheduled across the hand-off points
#define SYNC()                                                      \
  do {                                                              \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local"); \
    __builtin_amdgcn_wave_barrier();                                \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local"); \
  } while (0)
#else
#define SYNC()                                                   \
  do {                                                           \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       \
    __builtin_amdgcn_wave_barrier();                             \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       \
  } while (0)
#endif
// Diagnostic build only (-DFFE_STAMPS): per-section shader-clock shares, summed over waves (cdna guide section 7,
// "In-kernel stamps").  The stamped build is never timed or shipped; read its SHARES, not its length.
#ifdef FFE_DBGCF
__device__ float g_dbgcf[64][4 + 6 * 12];  // diagnostic build: the contact rows and forces of the last constraint solve of envs 0 .. 63
#endif
#ifdef FFE_STAMPS
__device__ unsigned long long g_stamps[20];
#define STAMP_DECL unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_acc[20] = {0}
#define STAMP(k)                                                   \
  do {                                                             \
    __builtin_amdgcn_s_waitcnt(0);                                 \
    unsigned long long st_t1 = __builtin_amdgcn_s_memtime();       \
    c.st_acc[k] += st_t1 - c.st_t0;                                \
    c.st_t0 = st_t1;                                               \
  } while (0)
#define CSTAMP(k)                                                                                  \
  do {                                                                                             \
    __builtin_amdgcn_s_waitcnt(0);                                                                 \
    unsigned long long cs_t1 = __builtin_amdgcn_s_memtime();                                       \
    if (lane == 0) atomicAdd(&g_stamps[k], cs_t1 - cs_t0);                                         \
    cs_t0 = cs_t1;                                                                                 \
  } while (0)
#define CSTAMP_DECL unsigned long long cs_t0 = __builtin_amdgcn_s_memtime()
#else
#define STAMP(k) do {} while (0)
#define CSTAMP(k) do {} while (0)
#define CSTAMP_DECL do {} while (0)
#endif
// The whole-fly CoM (wave-uniform; stage 1 -> reward) lives in a spare corner of the LDS tile, not in three VGPRs.
__device__ __forceinline__ void set_com(Ctx &c, V3 v) { if (c.lane == 0) { c.T.sens[9] = v.x; c.T.sens[10] = v.y; c.T.sens[11] = v.z; } }
__device__ __forceinline__ V3 get_com(const Ctx &c) { return {c.T.sens[9], c.T.sens[10], c.T.sens[11]}; }
// Diagnostic build only (-DFFE_TRACE): start / end shader clock and hardware slot of every wave of the last launch, for the
// occupancy timeline of tools/wave_timeline.py (dispatch ramp, wave lifetimes, tail).
#ifdef FFE_TRACE
#define TRACE_HI(c) (((unsigned)(c).nrare & 0xffu) | (((unsigned)(c).nfac & 0xffu) << 8) | ((min((c).tr_coll, 0xffffu)) << 16))
__device__ unsigned long long g_trace[32768][4];
#define TRACE_BEGIN unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime(); const int tr_prev = cost[order[blockIdx.x]]
#define TRACE_END(slot, tr_extra, tr_hi)                                                                                              \
  do {                                                                                                                         \
    __builtin_amdgcn_s_waitcnt(0);                                                                                             \
    if (threadIdx.x == 0 && (slot) < 32768) {                                                                                  \
      g_trace[slot][0] = tr_t0; g_trace[slot][1] = __builtin_amdgcn_s_memrealtime();                                               \
      g_trace[slot][2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)tr_hi << 32);  /* HW_ID | caller's extra */                                               \
      g_trace[slot][3] = (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf) | ((unsigned long long)(tr_extra) << 8); /* XCC_ID | extra */ \
    }                                                                                                                          \
  } while (0)
#else
#define TRACE_BEGIN do {} while (0)
#define TRACE_HI(c) 0
#define TRACE_END(slot, tr_extra, tr_hi) do {} while (0)
#endif
// timing-only ablation switches (bench experiments; results are wrong when set)
// The in-kernel ones exist only in the -DFFE_ABLATION diagnostic build (tools/ablate.py); the shipped kernel carries neither the
// branches nor the flag bits through its loops.
#ifdef FFE_ABLATION
#define DBG(c, f) ((c).flags & (f))
#else
#define DBG(c, f) (0)
#endif
enum { DBG_SKIP_FACTOR = 1 << 16, DBG_SKIP_SOLVE = 1 << 17, DBG_SKIP_STAGE1 = 1 << 18, DBG_SKIP_MENTRIES = 1 << 19, DBG_SKIP_GHOST = 1 << 20, DBG_SKIP_WBPG = 1 << 21, DBG_SKIP_OBS = 1 << 22, DBG_NO_CARRY = 1 << 23, DBG_NO_ORDER = 1 << 24 };

// ------------------------------------------------------------------------------------------------ collision (flight)
// ref: tasks/base.py:299-302 disables the floor contacts only; the fly's own geoms (legs welded to the thorax in their retracted
// pose, head, mouth parts, wings, abdomen: fruitfly.xml:323-443, excludes :733-760) still collide in flight.  mj: mj_collision is part
// of the position stage: `flight_collide` runs after stage 1 on the link frames it leaves (T.xpos / T.xmat + the link quaternions),
// with the factor's workspace (T.LD) and the link scratch arrays (T.la .. T.lc) - both idle between stage 1 and stage 2 - as scratch.
struct CollA { float4 gc[kMaxGeom], gq[kMaxGeom]; float lq[kMaxLink][4]; float sdc[kNSD][5], sdn[kNSD][5]; };                                        // over T.LD
constexpr int kCL2 = 24;  // pairs one narrow phase takes (typical: the touching ones, 2 - 6)
struct CollB { unsigned short cl1[128], cl2[kCL2]; float rec[kMC][12]; float cl2n[kCL2][5]; };  // over T.la .. T.lc
static_assert(sizeof(CollA) <= sizeof(Tile::LD), "collision scratch A");
static_assert(sizeof(CollB) <= sizeof(Tile::lT) + sizeof(Tile::lb) + sizeof(Tile::lc) && offsetof(Tile, lc) == offsetof(Tile, lT) + sizeof(Tile::lT) + sizeof(Tile::lb), "collision scratch B");
__device__ __forceinline__ CollA &coll_a(Tile &T) { return *reinterpret_cast<CollA *>(&T.LD[0]); }
__device__ __forceinline__ CollB &coll_b(Tile &T) { return *reinterpret_cast<CollB *>(&T.lT[0][0]); }

__device__ __forceinline__ cvx::Geom load_geom(const CollA &A, const DevModel FFE_CONST &M, int g) {
  const float4 cc = A.gc[g], qq = A.gq[g];
  return cvx::Geom{dm::V3{cc.x, cc.y, cc.z}, dm::Q4{qq.x, qq.y, qq.z, qq.w}, M.cg_size[g], M.cg_size[kMaxGeom + g], M.cg_size[2 * kMaxGeom + g], M.cg_type[g]};
}

// Broad phase: MuJoCo's bounding-sphere test over the static candidate list, then a rigorous lower bound of the distance from a
// separating direction (two generic ones, cvx::separation_bound, and the one the narrow phase found for the pair on the last
// substep it ran: `sdc` in, `sdn` out).  Narrow phase: one lane per remaining pair (cvx::collide: mjc_CapsuleCapsule / the general
// convex collider restated in convex.hpp).  A contact inside its margin but outside margin - gap exerts no force and - with no
// adhesion actuator in the flight model - takes part in nothing: dropped.  The contacts (at most kMC, the deepest) are left in
// `rec`.  Returns count | overflow << 8 | new cache count << 16.
// Narrow phase + contact selection over the `n2` pairs the broad phase left in `cl2` (one lane per pair).  WITH_RARE = false leaves out
// the two classes that need the most registers (ellipsoid - cylinder: a wing near the abdomen; cylinder - cylinder) - see flight_collide.
template <bool WITH_RARE>
__device__ __forceinline__ int collide_narrow(Tile &T, const DevModel FFE_CONST &M, const int lane, int n2, int nk, int ovf) {
  CollA &A = coll_a(T);
  CollB &B = coll_b(T);
  const unsigned long long mm0 = M.cg_mmask[0], mm1 = M.cg_mmask[1];
  const float mclass = M.c_margin;
  auto pair_margin = [&](int a, int b) {
    const bool ma = a < 64 ? ((mm0 >> a) & 1ull) : ((mm1 >> (a - 64)) & 1ull), mb = b < 64 ? ((mm0 >> b) & 1ull) : ((mm1 >> (b - 64)) & 1ull);
    return (ma || mb) ? mclass : 0.f;
  };
  bool hit = false;
  float dist = 0.f, margin = 0.f, ctt = 0.f;
  dm::V3 nrm = {1.f, 0.f, 0.f}, cpos = {0.f, 0.f, 0.f};
  int a = 0, b = 0;
  unsigned w = 0u;
  if (lane < n2) {
    w = B.cl2[lane];
    a = w & 255; b = w >> 8;
    margin = pair_margin(a, b);
    const float *kn = B.cl2n[lane];
    const cvx::Contact ct = cvx::collide<WITH_RARE>(load_geom(A, M, a), load_geom(A, M, b), dm::V3{kn[0], kn[1], kn[2]}, kn[3] != 0.f, kn[4]);
    dist = ct.dist; nrm = ct.n; cpos = ct.pos; ctt = ct.t;
    hit = dist < margin - (margin != 0.f ? M.c_gap : 0.f);  // (mj: dist <= margin is detected, dist < margin - gap is active; an inactive one takes part in nothing here)
  }
  {
    // the pair's direction, for the next substep's bound and warm start; when the cache cannot take them all, the classes whose cold
    // search costs most go first (ellipsoid - cylinder, then ellipsoid - ellipsoid, then the capsule classes' axis parameter)
    const int ta = M.cg_type[a], tb = M.cg_type[b];
    const int prio = lane < n2 ? (cvx::rare_class(ta, tb) ? 2 : (ta >= cvx::ELLIPSOID ? 1 : 0)) : -1;
    const unsigned long long b2 = __ballot(prio == 2), b1 = __ballot(prio == 1), b0 = __ballot(prio == 0), below = (1ull << lane) - 1ull;
    const int pos = nk + (prio == 2 ? __popcll(b2 & below) : prio == 1 ? __popcll(b2) + __popcll(b1 & below) : __popcll(b2) + __popcll(b1) + __popcll(b0 & below));
    if (prio >= 0 && pos < kNSD) { float *o = A.sdn[pos]; o[0] = nrm.x; o[1] = nrm.y; o[2] = nrm.z; o[3] = __int_as_float((int)w); o[4] = ctt; }
  }
  nk = min(nk + n2, kNSD);
  unsigned long long bal = __ballot(hit);
  if (__popcll(bal) > kMC) {  // more contacts than the solver carries: the env is flagged and the deepest are kept
    ovf = 1;
    int rank = 0;
    for (unsigned long long m = bal; m; m &= m - 1) {
      const int j = __ffsll((long long)m) - 1;
      const float dj = __shfl(dist, j);
      if (dj < dist || (dj == dist && j < lane)) rank++;
    }
    hit = hit && rank < kMC;
    bal = __ballot(hit);
  }
  const int n = __popcll(bal), idx = __popcll(bal & ((1ull << lane) - 1ull));
  if (hit) {
    float *o = B.rec[idx];
    o[0] = nrm.x; o[1] = nrm.y; o[2] = nrm.z; o[3] = cpos.x; o[4] = cpos.y; o[5] = cpos.z; o[6] = dist;
    o[7] = margin - (margin != 0.f ? M.c_gap : 0.f); o[8] = M.cg_invw[a] + M.cg_invw[b];
    o[9] = __int_as_float(M.cg_link[a]); o[10] = __int_as_float(M.cg_link[b]); o[11] = __int_as_float((int)w);
  }
  SYNC();
  return n | (ovf << 8) | (nk << 16);
}


__device__ __noinline__ int flight_collide_a(Tile *Tp, const DevModel FFE_CONST *Mp, const int lane, const int ncache) {
  Tile &T = *Tp;
  const DevModel FFE_CONST &M = *Mp;
  CollA &A = coll_a(T);
  CollB &B = coll_b(T);
  const int ncg = M.ncg;
  CSTAMP_DECL;
  for (int g = lane; g < ncg; g += kWave) {
    const int l = M.cg_link[g];
    const float *lq = A.lq[l];
    const dm::Q4 xq = {lq[0], lq[1], lq[2], lq[3]};
    const dm::V3 gp = dm::V3{T.xpos[l][0], T.xpos[l][1], T.xpos[l][2]} + dm::qrot(xq, dm::V3{M.cg_pos[g], M.cg_pos[kMaxGeom + g], M.cg_pos[2 * kMaxGeom + g]});
    const dm::Q4 gq = dm::qnormalize(dm::qmul(xq, dm::Q4{M.cg_quat[g], M.cg_quat[kMaxGeom + g], M.cg_quat[2 * kMaxGeom + g], M.cg_quat[3 * kMaxGeom + g]}));
    A.gc[g] = make_float4(gp.x, gp.y, gp.z, M.cg_brad[g]);
    A.gq[g] = make_float4(gq.w, gq.x, gq.y, gq.z);
  }
  SYNC();
  const unsigned long long mm0 = M.cg_mmask[0], mm1 = M.cg_mmask[1];
  const float mclass = M.c_margin;
  auto pair_margin = [&](int a, int b) {
    const bool ma = a < 64 ? ((mm0 >> a) & 1ull) : ((mm1 >> (a - 64)) & 1ull), mb = b < 64 ? ((mm0 >> b) & 1ull) : ((mm1 >> (b - 64)) & 1ull);
    return (ma || mb) ? mclass : 0.f;
  };
  int n1 = 0, ovf = 0;
  const int ncp = M.ncp;
  unsigned pw[kMaxPair / kWave];
#pragma unroll
  for (int r = 0; r < kMaxPair / kWave; r++) pw[r] = M.cp_pair[r * kWave + lane];  // (issued together: one L2 round trip, not one per round)
#pragma unroll
  for (int r = 0; r < kMaxPair / kWave; r++) {
    const int base = r * kWave;
    if (base >= ncp) break;
    const unsigned w = pw[r];
    bool pass = false;
    if (w != 0xffffu) {
      const int a = w & 255, b = w >> 8;
      const float4 ca = A.gc[a], cb = A.gc[b];
      const float dx = cb.x - ca.x, dy = cb.y - ca.y, dz = cb.z - ca.z, reach = ca.w + cb.w + pair_margin(a, b);
      pass = dx * dx + dy * dy + dz * dz <= reach * reach;
    }
    const unsigned long long bal = __ballot(pass);
    const int idx = n1 + __popcll(bal & ((1ull << lane) - 1ull));
    if (pass && idx < 128) B.cl1[idx] = (unsigned short)w;
    n1 += __popcll(bal);
  }
  if (n1 > 128) { n1 = 128; ovf = 1; }
  SYNC();
  CSTAMP(17);  // geom frames + bounding spheres
  int n2 = 0, nk = 0;
  bool any_rare = false;
#pragma unroll 1
  for (int base = 0; base < n1; base += kWave) {
    bool pass = false, keep = false, have_kn = false;
    unsigned w = 0u;
    dm::V3 kn = {0.f, 0.f, 0.f};
    float kt = 0.f;
    if (base + lane < n1) {
      w = B.cl1[base + lane];
      const int a = w & 255, b = w >> 8;
      const cvx::Geom ga = load_geom(A, M, a), gb = load_geom(A, M, b);
      const float margin = pair_margin(a, b), incl = margin - (margin != 0.f ? M.c_gap : 0.f);  // (only a contact inside margin - gap matters here)
      pass = cvx::separation_bound(ga, gb) <= incl;
      if (pass) {
        int alt = -1;
        for (int k = 0; k < ncache; k++) {
          const int pk = __float_as_int(A.sdc[k][3]);
          if (pk == (int)w) {
            kn = dm::V3{A.sdc[k][0], A.sdc[k][1], A.sdc[k][2]};
            kt = A.sdc[k][4];
            have_kn = true;
          } else if ((pk & 255) == a && M.cg_type[pk >> 8] == gb.type) alt = k;
        }
        // A wing sweeps over the abdomen's stacked segments, a new pair every substep or two: a pair seen for the first time borrows the
        // direction its ellipsoid holds against a neighbouring geom of the same kind.  As a bound any direction is rigorous; as a start it
        // saves the narrow phase its search from scratch (the most expensive thing a wave does: it set the length of whole launches).
        if (!have_kn && alt >= 0 && ga.type == cvx::ELLIPSOID) { kn = dm::V3{A.sdc[alt][0], A.sdc[alt][1], A.sdc[alt][2]}; have_kn = true; }
        if (have_kn) keep = -cvx::overlap(ga, gb, kn) > incl;
        pass = !keep;
      }
    }
    any_rare = any_rare || __ballot(pass && cvx::rare_class(M.cg_type[w & 255], M.cg_type[w >> 8])) != 0ull;
    const unsigned long long bal = __ballot(pass), balk = __ballot(keep);
    const int idx = n2 + __popcll(bal & ((1ull << lane) - 1ull)), idk = nk + __popcll(balk & ((1ull << lane) - 1ull));
    if (pass && idx < kCL2) { B.cl2[idx] = (unsigned short)w; float *o = B.cl2n[idx]; o[0] = kn.x; o[1] = kn.y; o[2] = kn.z; o[3] = have_kn ? 1.f : 0.f; o[4] = kt; }
    if (keep && idk < kNSD) { A.sdn[idk][0] = kn.x; A.sdn[idk][1] = kn.y; A.sdn[idk][2] = kn.z; A.sdn[idk][3] = __int_as_float((int)w); A.sdn[idk][4] = kt; }
    n2 += __popcll(bal);
    nk = min(nk + __popcll(balk), kNSD);
  }
  if (n2 > kCL2) { n2 = kCL2; ovf = 1; }
  SYNC();
  CSTAMP(18);  // separating-direction bounds + cache
  // The narrow phase of the classes that run every substep is part of this (leaf) function: it stays within the caller-saved registers, so
  // the call costs no save / restore traffic.  A pair of a rare class sends the whole narrow phase to the second function.
  if (any_rare) return 0x40000000 | n2 | (nk << 8) | (ovf << 16);
  const int r = collide_narrow<false>(T, M, lane, n2, nk, ovf);
  CSTAMP(19);  // narrow phase (common classes)
  return r;
}

// the same with every pair class (entered only when the broad phase left an ellipsoid - cylinder or cylinder - cylinder pair)
__device__ __noinline__ int flight_collide_b(Tile *Tp, const DevModel FFE_CONST *Mp, const int lane, const int packed) {
  return collide_narrow<true>(*Tp, *Mp, lane, packed & 0xff, (packed >> 8) & 0xff, (packed >> 16) & 1);
}

// the position stage's collision: cache in, contacts + cache out (lane-resident, see Ctx)
__device__ __forceinline__ void flight_collide(Ctx &c) {
  const DevModel FFE_CONST &M = model(c);
  Tile &T = c.T;
  c.nct = 0;
  if (M.ncg == 0 || (c.flags & FFE_NO_CONTACT)) return;
  CollA &A = coll_a(T);
  CollB &B = coll_b(T);
  if (c.lane < kNSD) { float *o = A.sdc[c.lane]; o[0] = c.sd_nx; o[1] = c.sd_ny; o[2] = c.sd_nz; o[3] = __int_as_float(c.sd_pid); o[4] = c.sd_t; }
  SYNC();
#ifdef FFE_TRACE
  const unsigned long long tr_c0 = __builtin_amdgcn_s_memrealtime();
#endif
  int r = __builtin_amdgcn_readfirstlane(flight_collide_a(&T, c.Mp, c.lane, c.sd_cnt));
  if (r & 0x40000000) {
    r = __builtin_amdgcn_readfirstlane(flight_collide_b(&T, c.Mp, c.lane, r));
    c.nrare++;
  }
#ifdef FFE_TRACE
  c.tr_coll += (unsigned)(__builtin_amdgcn_s_memrealtime() - tr_c0);
#endif
  c.nct = r & 0xff;
  c.ct_ovf |= (r >> 8) & 0xff;
  c.sd_cnt = r >> 16;
  if (c.lane < kNSD) { const float *o = A.sdn[c.lane]; c.sd_nx = o[0]; c.sd_ny = o[1]; c.sd_nz = o[2]; c.sd_pid = __float_as_int(o[3]); c.sd_t = o[4]; }
  if (c.lane < kMC) {
    const float *o = B.rec[c.lane];
    c.ct_nx = o[0]; c.ct_ny = o[1]; c.ct_nz = o[2]; c.ct_px = o[3]; c.ct_py = o[4]; c.ct_pz = o[5]; c.ct_dist = o[6]; c.ct_incl = o[7]; c.ct_invw = o[8];
    c.ct_l1 = __float_as_int(o[9]); c.ct_l2 = __float_as_int(o[10]); c.ct_pid = __float_as_int(o[11]);
  }
  SYNC();
}

// Stage 1 = mj_fwdPosition + mj_fwdVelocity on the welded link model (mj_kinematics, mj_comPos, mj_crb,
// mj_comVel, mj_passive, mj_rne).  Needs T.qpos / T.qvel; leaves cdof, cdofd, xpos, xmat, M, f_smooth_nb.
__device__ __forceinline__ void stage1(Ctx &c) {
  const DevModel FFE_CONST &M = model(c);
  Tile &T = c.T;
  const int lane = c.lane;
  const bool is_link = lane < M.nlink, is_dof = lane < M.nv;
  const int l_dofadr = M.l_dofadr[lane], l_dofnum = M.l_dofnum[lane], l_sub = M.l_sub[lane];
  const int d_link = M.d_link[lane], d_kind = M.d_kind[lane];
  const unsigned anc_lo = M.l_anc[lane], anc_hi = M.l_anc[kLanePad + lane];  // ancestor links, nearest first, 0xff = none
  STAMP(8);  // everything between the last stamp and a stage 1 (integration, sensors, ghost, prologue)
  // inertial constants of this lane's link, needed after the frame composition below: requested now, so that their L2 round
  // trip runs under the kinematics instead of after its last hand-off
  const int d_plink_ = M.l_parent[d_link], d_dof0_ = M.l_dofadr[d_link];  // (V3 below: parent link and first dof of this dof's link)
  const float l_mass_ = M.l_mass[lane];
  const V3 l_ipos_ = ldv_lane(M.l_ipos, lane), l_inertia_ = ldv_lane(M.l_inertia, lane);
  const M3 l_imat_ = ldm_lane(M.l_imat, lane);

  // ---- K1: link frame in its parent (joint rotations folded in) + hinge axes in the final link frame
  if (is_link) {
    Q4 q;
    V3 p;
    if (lane == 0) {
      q = qnormalize(Q4{T.qpos[3], T.qpos[4], T.qpos[5], T.qpos[6]});
      p = {0.f, 0.f, 0.f};
    } else {
      Q4 qr = {1.f, 0.f, 0.f, 0.f};
      // a link carries at most 3 hinges: fetch axes / addresses / reference angles of all of them first
      V3 jax[3];
      int jqa[3];
      float jq0[3];
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int d = l_dofadr + (j < l_dofnum ? j : 0);
        jax[j] = ldv_lane(M.d_axis, d); jqa[j] = M.d_qadr[d]; jq0[j] = M.d_qpos0[d];
      }
#pragma unroll
      for (int j = 2; j >= 0; j--) {
        if (j < l_dofnum) {
          const int d = l_dofadr + j;
          const V3 bax = qrot(qconj(qr), jax[j]);
          T.cdof[d][0] = bax.x; T.cdof[d][1] = bax.y; T.cdof[d][2] = bax.z;
          qr = qmul(axis_angle(jax[j], T.qpos[jqa[j]] - jq0[j]), qr);
        }
      }
      q = qmul(Q4{M.l_quat[lane], M.l_quat[kLanePad + lane], M.l_quat[2 * kLanePad + lane], M.l_quat[3 * kLanePad + lane]}, qr);
      p = ldv_lane(M.l_pos, lane);
    }
    float *o = T.lT[lane];
    o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = q.w; o[4] = q.x; o[5] = q.y; o[6] = q.z;
  }
  SYNC();
  // ---- K2: compose up the ancestor chain (every lane walks its own chain; no per-level barrier)
  V3 xp = {0.f, 0.f, 0.f}, xip = {0.f, 0.f, 0.f};
  M3 xm = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}, xim = xm;
  float mass = 0.f;
  // Pointer doubling over the ancestor chain: after round r a link's frame is expressed in its 2^(r+1)-th ancestor,
  // so 3 rounds (LDS hand-offs) replace a walk of up to 7 dependent compositions.
  V3 kp = {0.f, 0.f, 0.f};
  Q4 kq = {1.f, 0.f, 0.f, 0.f};
  if (is_link) { const float *o = T.lT[lane]; kp = {o[0], o[1], o[2]}; kq = {o[3], o[4], o[5], o[6]}; }
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const int a = (int)((anc_lo >> (8 * ((1 << r) - 1))) & 0xffu);  // 1st, 2nd, 4th ancestor
    if (is_link && a != 0xff) {
      const float *oa = T.lT[a];
      const Q4 qa = {oa[3], oa[4], oa[5], oa[6]};
      kp = V3{oa[0], oa[1], oa[2]} + qrot(qa, kp);
      kq = qmul(qa, kq);
    }
    SYNC();
    if (r < 2) {
      if (is_link) { float *o = T.lT[lane]; o[0] = kp.x; o[1] = kp.y; o[2] = kp.z; o[3] = kq.w; o[4] = kq.x; o[5] = kq.y; o[6] = kq.z; }
      SYNC();
    }
  }
  if (is_link) {
    V3 p = kp;
    Q4 q = kq;
    q = qnormalize(q);
    { float *lq = coll_a(T).lq[lane]; lq[0] = q.w; lq[1] = q.x; lq[2] = q.y; lq[3] = q.z; }  // (for flight_collide: the factor's workspace is idle in stage 1)
    xp = p;
    xm = q2m(q);
    mass = l_mass_;
    xip = xp + mv(xm, l_ipos_);
    xim = mm(xm, l_imat_);
    T.xpos[lane][0] = xp.x; T.xpos[lane][1] = xp.y; T.xpos[lane][2] = xp.z;
    float *xo = T.xmat[lane];
    xo[0] = xm.m0; xo[1] = xm.m1; xo[2] = xm.m2; xo[3] = xm.m3; xo[4] = xm.m4; xo[5] = xm.m5; xo[6] = xm.m6; xo[7] = xm.m7; xo[8] = xm.m8;
  }
  // mj: mj_comPos - CoM of the whole tree (all lanes take part in the reduction)
  V3 com1;
  {
    float inv = 1.0f / M.total_mass;
    com1 = {wave_sum(mass * xip.x) * inv, wave_sum(mass * xip.y) * inv, wave_sum(mass * xip.z) * inv};
    set_com(c, com1);
  }
  I10 cin = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (is_link) {
    cin = inert_com(l_inertia_, xim, xip - com1, mass);
    st10(T.cinert[lane], cin);
  }
  SYNC();
  STAMP(0);  // kinematics + com + cinert
  // ---- dof axes in the com-centred world frame (mj: mju_dofCom)
  S6 cd = zero6();
  if (is_dof) {
    if (d_kind == 0) {
      int k = lane;  // translational root dofs are dofs 0..2
      cd = {0.f, 0.f, 0.f, k == 0 ? 1.f : 0.f, k == 1 ? 1.f : 0.f, k == 2 ? 1.f : 0.f};
    } else {
      const float *xo = T.xmat[d_link];
      V3 ax;
      if (d_kind == 1) {
        int k = lane - 3;
        ax = {xo[k], xo[3 + k], xo[6 + k]};
      } else {
        V3 b = {T.cdof[lane][0], T.cdof[lane][1], T.cdof[lane][2]};
        ax = mv(ldm(xo), b);
      }
      V3 off = get_com(c) - V3{T.xpos[d_link][0], T.xpos[d_link][1], T.xpos[d_link][2]};
      cd = mk6(ax, cross(ax, off));
    }
  }
  SYNC();  // every lane has consumed its link-frame axis before cdof is overwritten
  if (is_dof) st6(T.cdof[lane], cd);
  SYNC();
  // ---- V1: velocity increment contributed by each link's own dofs
  if (is_link) {
    S6 dv = zero6();
    for (int j = 0; j < l_dofnum; j++) dv = dv + T.qvel[l_dofadr + j] * ld6(T.cdof[l_dofadr + j]);
    st6(T.la[lane], dv);
  }
  SYNC();
  // ---- V2: mj_comVel - link velocity = sum over the ancestor chain
  S6 cvel = zero6();
  if (is_link) {
    cvel = ld6(T.la[lane]);
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int a = (int)(((it < 4 ? anc_lo : anc_hi) >> (8 * (it & 3))) & 0xffu);
      if (a != 0xff) cvel = cvel + ld6(T.la[a]);
    }
    st6(T.lb[lane], cvel);
  }
  SYNC();
  // ---- V3: cdof_dot = (velocity just before this dof) x cdof
  S6 cdd = zero6();
  if (is_dof) {
    if (d_kind == 2) {
      S6 cv = ld6(T.lb[d_plink_]);
      for (int e = d_dof0_; e < lane; e++) cv = cv + T.qvel[e] * ld6(T.cdof[e]);
      cdd = cross_motion(cv, ld6(T.cdof[lane]));
    } else if (d_kind == 1) {
      S6 cv = {0.f, 0.f, 0.f, T.qvel[0], T.qvel[1], T.qvel[2]};  // free joint: after the 3 translations only
      cdd = cross_motion(cv, ld6(T.cdof[lane]));
    }
    st6(T.cdofd[lane], cdd);
  }
  SYNC();
  STAMP(1);  // cdof + velocities
  // ---- A1: per-link sum of cdof_dot * qvel
  if (is_link) {
    S6 da = zero6();
    for (int j = 0; j < l_dofnum; j++) da = da + T.qvel[l_dofadr + j] * ld6(T.cdofd[l_dofadr + j]);
    st6(T.la[lane], da);
  }
  SYNC();
  // ---- A2: mj_rne forward pass + mj_passive fluid forces, per link
  S6 frc = zero6();
  if (is_link) {
    S6 cacc = ld6(T.la[lane]);
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int a = (int)(((it < 4 ? anc_lo : anc_hi) >> (8 * (it & 3))) & 0xffu);
      if (a != 0xff) cacc = cacc + ld6(T.la[a]);
    }
    if (!(c.flags & FFE_NO_GRAVITY)) { cacc.l0 -= M.gx; cacc.l1 -= M.gy; cacc.l2 -= M.gz; }
    const I10 cin2 = ld10(T.cinert[lane]);
    const S6 cvel2 = ld6(T.lb[lane]);
    frc = mul_inert(cin2, cacc) + cross_force(cvel2, mul_inert(cin2, cvel2));
  }
  if (!(c.flags & FFE_NO_FLUID)) {
    // mj_passive fluid forces, evaluated in link coordinates: (w_b, v_b) = link angular velocity / origin velocity
    const int ll = is_link ? lane : 0;
    const V3 xp2 = {T.xpos[ll][0], T.xpos[ll][1], T.xpos[ll][2]};
    const M3 xm2 = ldm(T.xmat[ll]);
    const S6 cvel3 = ld6(T.lb[ll]);
    const V3 off = xp2 - get_com(c);
    const V3 w_w = ang(cvel3);
    const V3 w_b = mtv(xm2, w_w), v_b = mtv(xm2, lin(cvel3) + cross(w_w, off));
    S6 wl = zero6();
    if (is_link) {
      const int kind = M.l_reckind[lane];
      if (kind == 1) wl = box_fluid_local(M.l_reccoef, lane, ldv_lane(M.l_recpos, lane), ldm_lane(M.l_recmat, lane), w_b, v_b);
      else if (kind == 2) wl = ell_fluid_local(M.ell + 32 * M.l_recell[lane], ldv_lane(M.l_recpos, lane), ldm_lane(M.l_recmat, lane), w_b, v_b);
    }
    // bodies welded to the root link: one inertia-box record per lane, summed across the wave in root-link axes
    const V3 w0 = {rl_f(w_b.x, 0), rl_f(w_b.y, 0), rl_f(w_b.z, 0)}, v0 = {rl_f(v_b.x, 0), rl_f(v_b.y, 0), rl_f(v_b.z, 0)};
    S6 wr = zero6();
    if (lane < M.nrootrec) wr = box_fluid_local(M.rr_coef, lane, ldv_lane(M.rr_pos, lane), ldm_lane(M.rr_mat, lane), w0, v0);
    wr = wave_sum6(wr);
    if (lane == 0) wl = wl + wr;
    if (is_link) {
      const V3 f_w = mv(xm2, lin(wl));
      frc = frc - mk6(mv(xm2, ang(wl)) + cross(off, f_w), f_w);
    }
  }
  if (is_link) st6(T.lc[lane], frc);
  SYNC();
  STAMP(2);  // rne forward + fluid
  // ---- A3: subtree sums (links are in depth-first order, so a subtree is a contiguous range) for forces and
  //          composite inertias (mj: mj_rne backward pass, mj_crb accumulation)
  {
    S6 fs = frc;
    const I10 z10 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const I10 ci = is_link ? ld10(T.cinert[lane]) : z10;
    I10 cr = ci;
    if (is_link && lane != 0) {
      for (int k = lane + 1; k < lane + l_sub; k++) {
        fs = fs + ld6(T.lc[k]);
        cr = add10(cr, ld10(T.cinert[k]));
      }
    }
    // the root link's subtree is the whole tree: reduce across the wave instead of an 18-step serial sum
    const S6 ftot = wave_sum6(is_link ? frc : zero6());
    const I10 ctot = {wave_sum(ci.i0), wave_sum(ci.i1), wave_sum(ci.i2), wave_sum(ci.i3), wave_sum(ci.i4), wave_sum(ci.i5), wave_sum(ci.i6), wave_sum(ci.i7), wave_sum(ci.i8), wave_sum(ci.i9)};
    if (lane == 0) { fs = ftot; cr = ctot; }
    SYNC();  // crb shares storage with cinert: every lane has finished reading the link inertias
    if (is_link) {
      st6(T.la[lane], fs);
      st10(T.crb[lane], cr);
    }
  }
  SYNC();
  // ---- joint space: bias projection, joint springs and dampers; crb * cdof
  if (is_dof) {
    const S6 cd2 = ld6(T.cdof[lane]);  // reloaded: keeping it in registers since the axes phase only raises pressure
    float bias = dot6(cd2, ld6(T.la[d_link]));
    float f = -bias;
    if (d_kind == 2) {
      if (!(c.flags & FFE_NO_SPRING)) f -= M.d_stiff[lane] * (T.qpos[M.d_qadr[lane]] - M.d_sref[lane]);
    }
    if (!(c.flags & FFE_NO_DAMPER)) f -= M.d_damp[lane] * T.qvel[lane];
    c.f_smooth_nb = f;
    st6(T.buf[lane], mul_inert(ld10(T.crb[d_link]), cd2));
  }
  SYNC();
  // M(i,j) = cdof_j . (crb_i cdof_i) is formed directly into the factor's working copy (see bfactor)
  STAMP(3);  // subtree sums + joint space
}


// mj: mj_factorI on (M + diag(add)), branch-parallel.  Tree-sparse L'DL in MuJoCo's row layout (row i = [M(i,i), M(i,parent), ...]);
// rows are left unscaled and 1/D is kept per lane (c.dinv).  Pivot k with ancestors a_1..a_n applies
// M(a_s,a_t) -= M(k,a_s) M(k,a_t) / M(k,k) for s <= t.  A substep needs two factorisations of the same sparsity that differ only
// in their diagonals (the constraint Hessian M + D_active and the implicit-damping matrix M + h B): DUAL factors both in one
// sweep (entries are float2, so the LDS instruction count and all index arithmetic are those of a single factorisation);
// non-DUAL refactors slot .x only (an active-set change) and leaves the Euler factor in .y untouched.  The dofs
// behind the free joint's root chain split into independent branches, so step t eliminates the t-th pivot of EVERY branch:
// 14 dependent steps + 5 for the root chain instead of 41.  The pair updates of a step's pivots are dealt over the 64 lanes in
// rounds by a host-built schedule (DevModel::fsched, one coalesced word per lane and round, fetched one round ahead);
// updates of the root block (rows of the free joint's dofs), which every branch touches, go to a private copy per branch
// (in the link scratch arrays, idle during stage 2) and are folded in before the root chain is eliminated.
static_assert(offsetof(Tile, la) % 8 == 0 && sizeof(Tile::la) + sizeof(Tile::lb) + sizeof(Tile::lc) >= 8 * 21 * 8, "root-block copies of up to 8 branches");
template <bool DUAL>
__device__ __forceinline__ void bfactor(Ctx &c, float add0, float add1) {
  const DevModel FFE_CONST &M = model(c);
  Tile &T = c.T;
  const int lane = c.lane;
  const int nv = M.nv;
  const int d_madr = c.la_pack & 0x3ff;
  STAMP(6);
  {  // mj_crb: M(i,j) = cdof_j . (crb_i cdof_i) over the 421 ancestor pairs, written straight into the working copy
    int ei[7], ej[7];
#pragma unroll
    for (int r = 0; r < 7; r++) {
      const int e = lane + r * kWave;
      const bool ok = e < M.nM;
      ei[r] = ok ? M.m_row[e] : 0;
      ej[r] = ok ? M.m_col[e] : 0;
    }
#pragma unroll
    for (int r = 0; r < 7; r++) {
      const int e = lane + r * kWave;
      if (e < M.nM) {
        float m = dot6(ld6a(T.cdof[ej[r]]), ld6a(T.buf[ei[r]]));
        if (ei[r] == ej[r]) m += M.d_arm[ei[r]] + (DBG(c, DBG_SKIP_MENTRIES) ? 1.f : 0.f);
        if (DUAL) T.LD[e] = make_float2(m, m); else T.LD[e].x = m;
      }
    }
  }
  float2 *dl = reinterpret_cast<float2 *>(&T.la[0][0]);
  const int ndl = 21 * M.nbranch;
  for (int e = lane; e < ndl; e += kWave) dl[e] = make_float2(0.f, 0.f);
  SYNC();
  if (lane < nv) {
    T.LD[d_madr].x += add0;
    if (DUAL) T.LD[d_madr].y += add1;
  }
  SYNC();
  STAMP(4);  // M entries
  if (!DBG(c, DBG_SKIP_FACTOR)) {
    const unsigned FFE_GLOBAL *tab = M.fsched + lane;
    const int nbr_rounds = M.fs_branch_rounds, nrounds = M.fs_rounds;
    auto round = [&](unsigned w) {
      if (w >> 31) {
        const int mk = (int)(w & 0x1ffu), sidx = (int)((w >> 9) & 0x1fu), tidx = (int)((w >> 14) & 0x1fu), tg = (int)((w >> 19) & 0x3ffu);
        float2 *tp = tg < 512 ? &T.LD[tg] : &dl[tg - 512];
        if (DUAL) {
          const float2 piv = T.LD[mk], a = T.LD[mk + sidx], b = T.LD[mk + tidx];
          float2 t = *tp;
          t.x -= a.x * __builtin_amdgcn_rcpf(piv.x) * b.x;
          t.y -= a.y * __builtin_amdgcn_rcpf(piv.y) * b.y;
          *tp = t;
        } else {
          tp->x -= T.LD[mk + sidx].x * __builtin_amdgcn_rcpf(T.LD[mk].x) * T.LD[mk + tidx].x;
        }
      }
    };
    // schedule words travel a group of four rounds ahead of their use (an L2 round trip is longer than one round)
    unsigned wq[4];
#pragma unroll
    for (int q = 0; q < 4; q++) wq[q] = tab[q * kWave];
#pragma unroll 1
    for (int base = 0; base < nbr_rounds; base += 4) {
      unsigned cw[4];
#pragma unroll
      for (int q = 0; q < 4; q++) cw[q] = wq[q];
#pragma unroll
      for (int q = 0; q < 4; q++) wq[q] = tab[(base + 4 + q) * kWave];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (base + q < nbr_rounds) {  // wave-uniform
          round(cw[q]);
          if (__builtin_amdgcn_readfirstlane(cw[q]) & 0x40000000u) SYNC();  // last round of a step: the next pivots read what this step wrote
        }
      }
    }
    SYNC();
    // fold the branches' root-block copies into the root rows, then eliminate the root chain
    if (lane < 21) {
      float2 acc = T.LD[lane];
      for (int b = 0; b < M.nbranch; b++) { const float2 d = dl[21 * b + lane]; acc.x += d.x; if (DUAL) acc.y += d.y; }
      if (DUAL) T.LD[lane] = acc; else T.LD[lane].x = acc.x;
    }
    SYNC();
    {  // root chain: nroot - 1 single-round steps
      unsigned rw[6];
#pragma unroll
      for (int q = 0; q < 6; q++) rw[q] = tab[(nbr_rounds + q) * kWave];  // (issued before the fold above completes: independent)
#pragma unroll
      for (int q = 0; q < 6; q++) {
        if (nbr_rounds + q < nrounds) {
          round(rw[q]);
          SYNC();
        }
      }
    }
  }
  if (DUAL) {
    const float2 d = T.LD[d_madr];
    c.dinv[0] = lane < nv ? __builtin_amdgcn_rcpf(d.x) : 0.f;
    c.dinv[1] = lane < nv ? __builtin_amdgcn_rcpf(d.y) : 0.f;
  } else {
    c.dinv[0] = lane < nv ? __builtin_amdgcn_rcpf(T.LD[d_madr].x) : 0.f;
  }
  STAMP(5);  // elimination
}

// mj: mj_solveLD, branch-parallel.  Behind the free joint's root chain (dofs 0..5) the dof tree splits into independent
// branches (abdomen chain 14, head subtree 14, wings 3 + 3, halteres 1 + 1), so the leaf-to-root substitution runs all branches
// at once: in step t every lane looks at the t-th pivot of ITS OWN branch (a byte of c.seq) and fetches that pivot's value
// from the owning lane with ds_bpermute - 14 steps instead of 41, each a handful of instructions.  The root chain is then
// coupled through six wave sums and finished with five uniform steps; the way back down mirrors it.
// MODE 0: factor .x only; 1: factor .y only; 2: both factors on the same right-hand side (returned as .x / .y).
__device__ __forceinline__ float bperm_f(float v, int src_lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v))); }
__device__ __forceinline__ int bperm_i(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
// MODE 0 / 1: one right-hand side through factor .x / .y; MODE 2: the same right-hand side through both factors; MODE 3: two
// right-hand sides (`rhs`, `rhs_b`) through factor .x at once (the contact columns of stage 2)
template <int MODE>
__device__ __forceinline__ float2 bsolve(Ctx &c, float rhs, float rhs_b = 0.f) {
  const DevModel FFE_CONST &M = *c.Mp;  // (three uniform scalars only: no need to launder the table pointer here)
  Tile &T = c.T;
  const int lane = c.lane;
  const int nv = M.nv, nroot = M.nroot, nbr = M.nbr_steps;
  const bool is_dof = lane < nv, branch = is_dof && lane >= nroot;
  const int d_madr = c.la_pack & 0x3ff, dep = (c.la_pack >> 10) & 0x3f, d_ndesc = c.la_pack >> 16;
  const int my_end = lane + d_ndesc, my_md = d_madr + dep;
  const float di0 = c.dinv[0], di1 = MODE == 3 ? c.dinv[0] : c.dinv[1];
  constexpr bool U0 = MODE != 1, U1 = MODE != 0;
  const float2 z2 = make_float2(0.f, 0.f);
  auto ldf = [&](int e) -> float2 {  // factor entry e, only the component(s) this instantiation uses
    if (MODE == 2) return T.LD[e];
    const float *p = reinterpret_cast<const float *>(T.LD) + 2 * e;
    if (MODE == 3) { const float v = p[0]; return make_float2(v, v); }
    return MODE == 0 ? make_float2(p[0], 0.f) : make_float2(0.f, p[1]);
  };
  if (DBG(c, DBG_SKIP_SOLVE)) return is_dof ? make_float2(rhs * di0, rhs * di1) : z2;
  STAMP(6);
  float x0 = is_dof ? rhs : 0.f, x1 = MODE == 3 ? (is_dof ? rhs_b : 0.f) : x0;
  // ---- x <- L^-T x, branches: step t eliminates the t-th pivot of every branch
#pragma unroll 1  // (rolled over the four sequence words: 26 KB less code over the kernel's five inlined copies, +0.3 %)
  for (int w = 0; w < 4; w++) {
    const unsigned word = w == 0 ? c.seq0 : (w == 1 ? c.seq1 : (w == 2 ? c.seq2 : c.seq3));
#pragma unroll
    for (int b = 0; b < 4; b++) {
      if (4 * w + b < nbr) {  // wave-uniform
        const int p = (int)((word >> (8 * b)) & 0xffu);
        const bool anc = branch && lane < p && p <= my_end;  // (0xff padding never passes)
        const int md_p = bperm_i(my_md, p);
        const float2 l = anc ? ldf(md_p - dep) : z2;
        if (U0) x0 -= l.x * bperm_f(x0 * di0, p);
        if (U1) x1 -= l.y * bperm_f(x1 * di1, p);
      }
    }
  }
  // ---- root chain: every branch dof i feeds root dof r with L(i, r) x_i / D_i
  {
    const float y0 = branch ? x0 * di0 : 0.f, y1 = branch ? x1 * di1 : 0.f;
#pragma unroll
    for (int r = 0; r < 6; r++) {
      if (r < nroot) {
        const float2 l = branch ? ldf(my_md - (r + 1)) : z2;
        if (U0) { const float s0 = wave_sum(l.x * y0); x0 = lane == r ? x0 - s0 : x0; }
        if (U1) { const float s1 = wave_sum(l.y * y1); x1 = lane == r ? x1 - s1 : x1; }
      }
    }
  }
#pragma unroll
  for (int i = 5; i > 0; i--) {
    if (i < nroot) {
      const float2 l = lane < i ? ldf(rl_i(my_md, i) - dep) : z2;
      if (U0) x0 -= l.x * (rl_f(x0, i) * rl_f(di0, i));
      if (U1) x1 -= l.y * (rl_f(x1, i) * rl_f(di1, i));
    }
  }
  // ---- x <- D^-1 x
  x0 *= di0; x1 *= di1;
  // ---- x <- L^-1 x: root chain, root -> branches, then down the branches (the sequences backwards)
#pragma unroll
  for (int j = 0; j < 5; j++) {
    if (j + 1 < nroot) {
      const float2 l = (lane > j && lane < nroot) ? ldf(my_md - (j + 1)) : z2;
      if (U0) x0 -= l.x * di0 * rl_f(x0, j);
      if (U1) x1 -= l.y * di1 * rl_f(x1, j);
    }
  }
#pragma unroll
  for (int r = 0; r < 6; r++) {
    if (r < nroot) {
      const float2 l = branch ? ldf(my_md - (r + 1)) : z2;
      if (U0) x0 -= l.x * di0 * rl_f(x0, r);
      if (U1) x1 -= l.y * di1 * rl_f(x1, r);
    }
  }
  const int pk = my_end | (dep << 8);
#pragma unroll 1
  for (int w = 3; w >= 0; w--) {
    const unsigned word = w == 0 ? c.seq0 : (w == 1 ? c.seq1 : (w == 2 ? c.seq2 : c.seq3));
#pragma unroll
    for (int b = 3; b >= 0; b--) {
      if (4 * w + b < nbr) {
        const int q = (int)((word >> (8 * b)) & 0xffu);
        const int pq = bperm_i(pk, q);
        const bool desc = branch && q < lane && lane <= (pq & 0xff);
        const float2 l = desc ? ldf(my_md - (pq >> 8)) : z2;
        if (U0) x0 -= l.x * di0 * bperm_f(x0, q);
        if (U1) x1 -= l.y * di1 * bperm_f(x1, q);
      }
    }
  }
  STAMP(7);
  return make_float2(x0, x1);
}

// Stage 2 = mj_fwdActuation, mj_fwdAcceleration, mj_fwdConstraint (joint limits), accelerometer, mj_Euler.
// `ctrl_force` is the per-dof generalized actuator force, already assembled.
__device__ __forceinline__ V3 stage2(Ctx &c, float qfrc_act, bool integrate, bool with_ghost, double ghost_accel_z, unsigned long long &lo_mask,
                     unsigned long long &hi_mask, unsigned long long &in_lo, unsigned long long &in_hi, int &iters_out) {
  const DevModel FFE_CONST &M = model(c);
  Tile &T = c.T;
  const int lane = c.lane;
  // the factor / solve index words are stage-2 business only: re-read here, they are not carried through stage 1
  c.la_pack = (unsigned)M.d_madr[lane] | ((unsigned)M.d_depth[lane] << 10) | ((unsigned)M.d_ndesc[lane] << 16);
  c.seq0 = M.br_seq[lane]; c.seq1 = M.br_seq[kWave + lane]; c.seq2 = M.br_seq[2 * kWave + lane]; c.seq3 = M.br_seq[3 * kWave + lane];
  const bool is_dof = lane < M.nv;
  const int d_kind = M.d_kind[lane];
  const float h = M.h;
  float f = is_dof ? c.f_smooth_nb + qfrc_act : 0.f;
  float qp = 0.f, qv = 0.f;
  if (is_dof) { qv = T.qvel[lane]; if (d_kind == 2) qp = T.qpos[M.d_qadr[lane]]; }

  // ---- mj_instantiateLimit / mj_makeImpedance / mj_referenceConstraint for this lane's hinge
  bool ex_lo = false, ex_hi = false;
  float D_lo = 0.f, D_hi = 0.f, ar_lo = 0.f, ar_hi = 0.f;
  if (is_dof && d_kind == 2 && M.d_limited[lane] && !(c.flags & FFE_NO_LIMIT)) {
    float margin = M.d_margin[lane];
    float dist_lo = qp - M.d_lo[lane], dist_hi = M.d_hi[lane] - qp;
    float dmin = M.d_solimp[lane], dmax = M.d_solimp[kLanePad + lane], width = M.d_solimp[2 * kLanePad + lane],
          mid = M.d_solimp[3 * kLanePad + lane], power = M.d_solimp[4 * kLanePad + lane];
    float K = M.d_K[lane], B = M.d_B[lane], iw = M.d_invw[lane];
    if (dist_lo < margin) {
      ex_lo = true;
      float imp = impedance(dmin, dmax, width, mid, power, dist_lo, margin);
      D_lo = imp * __builtin_amdgcn_rcpf(fmaxf(1e-30f, (1.f - imp) * iw));
      ar_lo = -B * qv - K * imp * (dist_lo - margin);
    }
    if (dist_hi < margin) {
      ex_hi = true;
      float imp = impedance(dmin, dmax, width, mid, power, dist_hi, margin);
      D_hi = imp * __builtin_amdgcn_rcpf(fmaxf(1e-30f, (1.f - imp) * iw));
      ar_hi = B * qv - K * imp * (dist_hi - margin);
    }
  }
  const unsigned long long ex_any = __ballot(ex_lo || ex_hi);
  STAMP(13);  // limit instantiation
  // Primal active-set Newton on  1/2 (a-a_s)'M(a-a_s) + sum 1/2 D min(0, J a - aref)^2 : with hinge limits the Hessian
  // is M + diag(D_active), i.e. the same sparse factorisation with a different diagonal; with no limit instantiated it
  // degenerates to qacc = M^-1 qfrc_smooth.  The implicit-damping Euler solve (M + h B) reuses the same code as a final
  // pass of the loop so that factor/solve are instantiated once.
  // first guess: an instantiated limit is active, unless the previous substep's solve found it instantiated but inactive (a
  // joint beyond its range that is already being driven back stays in that state for several substeps: without the memory
  // every one of them costs a second factorisation + solve)
  bool act_lo = ex_lo && !((in_lo >> lane) & 1ULL), act_hi = ex_hi && !((in_hi >> lane) & 1ULL);
  float a = 0.f, ae = 0.f, fc = 0.f;
  int iters = 0;
  const bool want_euler = integrate && !(c.flags & FFE_NO_DAMPER);
  const float hB = (is_dof && want_euler) ? h * M.d_damp[lane] : 0.f;
  // ---- contact rows of this position stage (mj: mj_makeConstraint for condim-1 contacts: one frictionless row each,
  //      J = n . (jacp of geom2's body - jacp of geom1's at the contact point), mj_makeImpedance, mj_referenceConstraint).  The row is
  //      kept by columns: this lane's dof entry of every contact's row in jk[]; lane k computes contact k's D and aref.
  const int nct = c.nct;
  float jk[kMC], cD = 0.f, car = 0.f;
#pragma unroll
  for (int k = 0; k < kMC; k++) jk[k] = 0.f;
  if (nct) {
    const S6 cd = is_dof ? ld6(T.cdof[lane]) : zero6();
    const V3 com = get_com(c);
    float dmin = fminf(fmaxf(M.c_solimp[0], 1e-4f), 0.9999f), dmax = fminf(fmaxf(M.c_solimp[1], 1e-4f), 0.9999f), width = fmaxf(0.f, M.c_solimp[2]),
          mid = fminf(fmaxf(M.c_solimp[3], 1e-4f), 0.9999f), power = fmaxf(1.f, M.c_solimp[4]);
#pragma unroll
    for (int k = 0; k < kMC; k++) {
      if (k < nct) {
        const V3 n = {rl_f(c.ct_nx, k), rl_f(c.ct_ny, k), rl_f(c.ct_nz, k)}, p = {rl_f(c.ct_px, k), rl_f(c.ct_py, k), rl_f(c.ct_pz, k)};
        const unsigned long long m1 = M.l_dofmask[rl_i(c.ct_l1, k)], m2 = M.l_dofmask[rl_i(c.ct_l2, k)];
        const int sg = (int)((m2 >> lane) & 1ULL) - (int)((m1 >> lane) & 1ULL);
        if (is_dof && sg != 0) { const V3 u = lin(cd) + cross(ang(cd), p - com); jk[k] = (float)sg * dot(n, u); }
        const float vel = wave_sum(jk[k] * qv);
        if (lane == k) {
          const float imp = impedance(dmin, dmax, width, mid, power, c.ct_dist, c.ct_incl);
          cD = imp * __builtin_amdgcn_rcpf(fmaxf(1e-30f, (1.f - imp) * c.ct_invw));
          car = -M.c_B * vel - M.c_K * imp * (c.ct_dist - c.ct_incl);
        }
      }
    }
  }
  if (ex_any == 0ULL && nct == 0) {
    // no limit instantiated, no contact: one dual factorisation, one dual solve
    bfactor<true>(c, 0.f, hB);
    c.nfac++;
    if (want_euler) { const float2 r = bsolve<2>(c, f); a = r.x; ae = r.y; }
    else { a = bsolve<0>(c, f).x; ae = a; }
  } else {
    // Contacts enter the same primal Newton through the matrix inversion lemma: with H0 = M + D_limits (the sparse factorisation) and
    // the active contact rows J (m <= kMC of them), (H0 + J' D J) a = rhs + J' D aref is solved as
    //   a = y0 + Y f,  y0 = H0^-1 rhs,  Y = H0^-1 J',  (D^-1 + J Y) f = -(J y0 - aref)      (f = the contacts' forces)
    // i.e. m more triangular solves with the resident factor and an m x m system (rows on lanes 0 .. m-1, Gauss-Jordan by readlane).
    // Y's columns live in LDS (T.crb / T.xmat[1..]: dead in stage 2) and are kept while the limit set - hence H0 - does not change.
    bool cact = lane < nct;   // first guess: a contact inside its includemargin pushes - unless the previous substep's solve found this pair separating
    if (nct) {                // (a pair drifting apart inside its margin stays that way for many substeps: each of them cost a second pass)
      const int me = c.ct_pid & 0xffff;
#pragma unroll
      for (int q = 0; q < kMC; q++) {
        const int pq = rl_i(c.pc, q);
        if ((pq & 0x1ffff) == (me | 0x10000)) cact = false;
      }
    }
    float cf = 0.f, y0 = 0.f;
    unsigned ymask = 0u;
    bool lim_changed = true;
    auto yrow = [&](int k) -> float * { return k < 4 ? &T.crb[0][0] + kMaxDof * k : &T.xmat[1][0] + kMaxDof * (k - 4); };
    static_assert(sizeof(T.crb) >= 4 * kMaxDof * 4 && sizeof(T.xmat) - 36 >= (kMC - 4) * kMaxDof * 4, "Y columns");
#pragma unroll 1
    for (int it = 0; it < 8; it++) {
      const float add = (act_lo ? D_lo : 0.f) + (act_hi ? D_hi : 0.f);
      const float rhs = f + (act_lo ? D_lo * ar_lo : 0.f) - (act_hi ? D_hi * ar_hi : 0.f);
      const unsigned am = (unsigned)__ballot(cact) & ((1u << nct) - 1u);
      bool need_y0 = false;
      if (lim_changed) {
        if (it == 0) bfactor<true>(c, add, hB);
        else bfactor<false>(c, add, 0.f);
        c.nfac++;
        need_y0 = true;
        ymask = 0u;
      }
      {
        // the solves still owed with this factor - y0 and the columns of the active contacts - two right-hand sides at a time
        unsigned todo = am & ~ymask;
        if (need_y0) {
          if (todo) {
            const int k = __ffs(todo) - 1;
            todo &= todo - 1u;
            float jv = 0.f;
#pragma unroll
            for (int q = 0; q < kMC; q++) jv = k == q ? jk[q] : jv;
            const float2 r2 = bsolve<3>(c, rhs, jv);
            y0 = r2.x;
            if (is_dof) yrow(k)[lane] = r2.y;
            ymask |= 1u << k;
          } else y0 = bsolve<0>(c, rhs).x;
        }
#pragma unroll 1
        while (todo) {
          const int k0 = __ffs(todo) - 1;
          todo &= todo - 1u;
          const int k1 = todo ? __ffs(todo) - 1 : -1;
          if (k1 >= 0) todo &= todo - 1u;
          float j0 = 0.f, j1 = 0.f;
#pragma unroll
          for (int q = 0; q < kMC; q++) { j0 = k0 == q ? jk[q] : j0; j1 = k1 == q ? jk[q] : j1; }
          if (k1 >= 0) {
            const float2 r2 = bsolve<3>(c, j0, j1);
            if (is_dof) { yrow(k0)[lane] = r2.x; yrow(k1)[lane] = r2.y; }
            ymask |= (1u << k0) | (1u << k1);
          } else {
            const float yk = bsolve<0>(c, j0).x;
            if (is_dof) yrow(k0)[lane] = yk;
            ymask |= 1u << k0;
          }
        }
      }
      a = y0;
      if (am) {
        SYNC();
        float srow[kMC], r = 0.f;
#pragma unroll
        for (int j = 0; j < kMC; j++) srow[j] = lane == j ? 1.f : 0.f;   // (inactive rows stay identity: zero force)
#pragma unroll
        for (int i = 0; i < kMC; i++) {
          if ((am >> i) & 1u) {
            const float ri = wave_sum(jk[i] * y0);
            if (lane == i) r = -(ri - car);
#pragma unroll
            for (int j = 0; j <= i; j++) {
              if ((am >> j) & 1u) {
                const float sv = wave_sum(jk[i] * (is_dof ? yrow(j)[lane] : 0.f));
                if (lane == i) srow[j] = sv + (i == j ? __builtin_amdgcn_rcpf(fmaxf(cD, 1e-30f)) : 0.f);
                if (lane == j && i != j) srow[i] = sv;
              }
            }
          }
        }
#pragma unroll
        for (int k = 0; k < kMC; k++) {
          if ((am >> k) & 1u) {
            const float ipiv = 1.f / rl_f(srow[k], k), rk = rl_f(r, k);
            float rowk[kMC];
#pragma unroll
            for (int j = 0; j < kMC; j++) rowk[j] = rl_f(srow[j], k);
            if (lane != k && lane < kMC) {
              const float fac = srow[k] * ipiv;
#pragma unroll
              for (int j = 0; j < kMC; j++) srow[j] -= fac * rowk[j];
              r -= fac * rk;
            }
          }
        }
        float diag = srow[0];
#pragma unroll
        for (int j = 1; j < kMC; j++) diag = lane == j ? srow[j] : diag;
        cf = (lane < kMC && ((am >> lane) & 1u)) ? r / diag : 0.f;
#pragma unroll
        for (int k = 0; k < kMC; k++)
          if ((am >> k) & 1u) a += (is_dof ? yrow(k)[lane] : 0.f) * rl_f(cf, k);
      } else cf = 0.f;
      iters++;
      const bool n_lo = ex_lo && (a - ar_lo < 0.f);
      const bool n_hi = ex_hi && (-a - ar_hi < 0.f);
      bool n_c = false;
#pragma unroll
      for (int k = 0; k < kMC; k++) {
        if (k < nct) {
          const float jar = wave_sum(jk[k] * a);
          if (lane == k) n_c = jar - car < 0.f;
        }
      }
      const bool lch = (n_lo != act_lo) || (n_hi != act_hi);
      const bool changed = lch || (lane < nct && n_c != cact);
      act_lo = n_lo; act_hi = n_hi; cact = n_c;
      lim_changed = __ballot(lch) != 0ULL;
      if (__ballot(changed) == 0ULL) break;
    }
#ifdef FFE_DBGCF
    if (c.dbg_env < 64) {
      float *o = g_dbgcf[c.dbg_env];
      const unsigned long long bc = __ballot(cact && lane < nct), bl = __ballot(act_lo), bh = __ballot(act_hi);
      if (lane == 0) { o[0] = (float)nct; o[1] = (float)(bc & 63ull); o[2] = (float)iters; o[3] = (float)(__popcll(bl) + __popcll(bh)); }
      if (lane < kMC) {
        float *q = o + 4 + 12 * lane;
        q[0] = c.ct_dist; q[1] = c.ct_nx; q[2] = c.ct_ny; q[3] = c.ct_nz; q[4] = c.ct_px; q[5] = c.ct_py; q[6] = c.ct_pz; q[7] = (float)c.ct_pid; q[8] = cD; q[9] = car; q[10] = cf; q[11] = c.ct_incl;
      }
    }
#endif
    if (act_lo) fc += D_lo * (ar_lo - a);
    if (act_hi) fc -= D_hi * (ar_hi + a);
#pragma unroll
    for (int k = 0; k < kMC; k++)
      if (k < nct) fc += jk[k] * rl_f(cact ? cf : 0.f, k);
    c.pc = lane < nct ? ((c.ct_pid & 0xffff) | (cact ? 0 : 0x10000)) : -1;
    ae = want_euler ? bsolve<1>(c, f + fc).y : a;
  }
  STAMP(14);  // (remaining glue inside the constraint/Euler block)
  lo_mask = __ballot(act_lo);
  hi_mask = __ballot(act_hi);
  in_lo = __ballot(ex_lo && !act_lo);
  in_hi = __ballot(ex_hi && !act_hi);
  iters_out = (ex_any || nct) ? iters : 0;
  c.qacc = a;
  // ---- accelerometer (mj: mj_rnePostConstraint + mj_objectAcceleration at the thorax site, which sits at the
  //      root-body origin): R^T (a_origin - g) with a_origin the free joint's linear acceleration
  V3 accel;
  {
    V3 ao = {__shfl(a, 0), __shfl(a, 1), __shfl(a, 2)};
    if (!(c.flags & FFE_NO_GRAVITY)) { ao.x -= M.gx; ao.y -= M.gy; ao.z -= M.gz; }
    accel = mtv(ldm(T.xmat[0]), ao);
  }
  if (!integrate) return accel;
  // ---- mj_Euler: semi-implicit update with the damping-implicit acceleration
  if (is_dof) {
    float nv_ = qv + h * ae;
    T.qvel[lane] = nv_;
    if (d_kind == 2) T.qpos[M.d_qadr[lane]] = qp + h * nv_;
  }
  SYNC();
  // Free-joint and ghost integration, one instruction stream for both: lanes 0-2 advance the root position, lanes 3-5
  // the ghost position (float64); lane 0 / lane 1 advance the root / ghost orientation (mju_quatIntegrate with the
  // body-frame angular velocity).  The ghost is an armature-1 free body coasting at the reference velocity.
  if (lane < 6) {
    const int k = lane < 3 ? lane : lane - 3;
    if (lane < 3) T.rootpos[k] += (double)h * (double)T.qvel[k];
    else if (with_ghost) {
      double gv = T.ghost[7 + k];
      if (k == 2) { gv += (double)h * ghost_accel_z; T.ghost[9] = gv; }
      T.ghost[k] += (double)h * gv;
    }
  }
  if (lane < 2 && (lane == 0 || with_ghost)) {
    Q4 q;
    V3 w;
    if (lane == 0) { q = {T.qpos[3], T.qpos[4], T.qpos[5], T.qpos[6]}; w = {T.qvel[3], T.qvel[4], T.qvel[5]}; }
    else { q = {(float)T.ghost[3], (float)T.ghost[4], (float)T.ghost[5], (float)T.ghost[6]}; w = {(float)T.ghost[10], (float)T.ghost[11], (float)T.ghost[12]}; }
    const float n2 = dot(w, w);
    q = qnormalize(q);
    if (n2 >= 1e-30f) {
      const float rn = __builtin_amdgcn_rsqf(n2);
      q = qnormalize(qmul(q, axis_angle(rn * w, n2 * rn * h)));
    }
    if (lane == 0) { T.qpos[3] = q.w; T.qpos[4] = q.x; T.qpos[5] = q.y; T.qpos[6] = q.z; }
    else { T.ghost[3] = q.w; T.ghost[4] = q.x; T.ghost[5] = q.y; T.ghost[6] = q.z; }
  }
  SYNC();
  return accel;
}

// mj: mj_fwdActuation.  Returns the generalized actuator force on this lane's dof.  `abase` = gain * clamp(ctrl) + bias0 of
// this lane's actuator: the control is fixed over a control step, so that part is evaluated once per launch (actuation_base);
// per substep only the length / velocity terms of the affine bias remain (velocity only if some actuator has a bias2).
__device__ __forceinline__ float actuation_base(Ctx &c, float ctrl) {
  const DevModel FFE_CONST &M = model(c);
  const int lane = c.lane;
  if ((c.flags & FFE_NO_ACTUATION) || lane >= M.nu) return 0.f;
  const float clo = M.a_clo[lane], chi = M.a_chi[lane];
  ctrl = M.a_cl[lane] ? fminf(fmaxf(ctrl, clo), chi) : ctrl;
  return M.a_gain[lane] * ctrl + M.a_b0[lane];
}
__device__ __forceinline__ float actuation(Ctx &c, float abase) {
  const DevModel FFE_CONST &M = model(c);
  Tile &T = c.T;
  const int lane = c.lane;
  if (c.flags & FFE_NO_ACTUATION) return 0.f;
  if (lane < M.nu) {
    // phase 1: every table read of this actuator is issued before anything waits (independent addresses)
    int tq[kMaxWrap], td[kMaxWrap];
    float tc[kMaxWrap];
    const bool use_vel = M.any_b2 != 0;  // uniform
#pragma unroll
    for (int w = 0; w < kMaxWrap; w++) {  // zero-padded transmission terms: joint actuators have one, fixed tendons several
      tq[w] = M.t_qadr[w * kMaxAct + lane]; tc[w] = M.t_coef[w * kMaxAct + lane];
      td[w] = use_vel ? M.t_dof[w * kMaxAct + lane] : 0;
    }
    const int fl = M.a_fl[lane];
    const float b1 = M.a_b1[lane], b2 = use_vel ? M.a_b2[lane] : 0.f, flo = M.a_flo[lane], fhi = M.a_fhi[lane];
    // phase 2
    float len = 0.f, vel = 0.f;
#pragma unroll
    for (int w = 0; w < kMaxWrap; w++) { len += tc[w] * T.qpos[tq[w]]; if (use_vel) vel += tc[w] * T.qvel[td[w]]; }
    float force = abase + b1 * len + b2 * vel;
    force = fl ? fminf(fmaxf(force, flo), fhi) : force;
    T.frc[lane] = force;
  }
  SYNC();
  float q = 0.f;
  if (lane < M.nv) {
    int a0 = M.d_act_id[lane], a1 = M.d_act_id[kLanePad + lane];
    if (a0 >= 0) q += M.d_act_coef[lane] * T.frc[a0];
    if (a1 >= 0) q += M.d_act_coef[kLanePad + lane] * T.frc[a1];
  }
  SYNC();
  return q;
}

// ------------------------------------------------------------------------------------------------ task helpers
// The wing-beat frequency filter feeds an argmin over a frequency grid, so one ulp decides which table is
// used: evaluate it exactly as numpy does (separate multiplies and adds, no FMA contraction).
__device__ __noinline__ double wbpg_filter(double cf, double rate, double base, double rel, double act) {
#pragma clang fp contract(off)
  double cmd = base * (1.0 + rel * act);
  double a = cf * rate;
  double b = cmd * (1.0 - rate);
  return a + b;
}
// argmin_i |table[i] - x| with numpy's first-minimum tie-break, across the wave (float64, bit-compatible
// with the reference's np.argmin(np.abs(...)) on the same tables; ref: pattern_generators.py:148,179,186)
__device__ int wave_argmin_absdiff(const double FFE_GLOBAL *tab, int n, double x, int lane) {
  double bv = 1e300;
  int bi = 0x7fffffff;
  for (int i = lane; i < n; i += kWave) {
    double v = fabs(x - tab[i]);
    if (v < bv) { bv = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    double ov = __shfl_xor(bv, o);
    int oi = __shfl_xor(bi, o);
    if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  return bi;
}

// argmin_i |table[i] - x| on a monotone, evenly spaced table (np.linspace: the 201 beat frequencies): the nearest grid point
// by arithmetic, then an exact comparison of it and its two neighbours with numpy's first-minimum tie-break.  Identical to
// the full scan whenever the table is within a quarter step of uniform, which the host checks (else grid_inv_step = 0).
__device__ __forceinline__ int grid_argmin_absdiff(const double FFE_GLOBAL *tab, int n, double x, double inv_step) {
  int k = (int)rint((x - tab[0]) * inv_step);
  k = k < 1 ? 1 : (k > n - 2 ? n - 2 : k);
  const double v0 = fabs(x - tab[k - 1]), v1 = fabs(x - tab[k]), v2 = fabs(x - tab[k + 1]);
  int bi = k - 1;
  double bv = v0;
  if (v1 < bv) { bv = v1; bi = k; }
  if (v2 < bv) { bv = v2; bi = k + 1; }
  return bi;
}

struct ObsLayout { int acc, gyro, jpos, jvel, vel, zaxis, rdisp, rquat; };
__device__ __forceinline__ ObsLayout obs_layout(int nj, int nref) {
  ObsLayout o;
  o.acc = 0; o.gyro = 3; o.jpos = 6; o.jvel = 6 + nj; o.vel = 6 + 2 * nj; o.zaxis = o.vel + 3; o.rdisp = o.zaxis + 3; o.rquat = o.rdisp + 3 * nref;
  return o;
}

// Observation assembly (ref: fruitfly.py:532-708 enabled set per tasks/base.py:167-168 + flight_imitation.py:84-85;
// ref_displacement / ref_root_quat: tasks/base.py:237-261).  Returns |ref_displacement[0]| and ref_root_quat[0].
__device__ __forceinline__ void write_obs(Ctx &c, const TaskDev FFE_CONST &K, float *obs, V3 s_acc, V3 s_gyro, V3 s_vel, int traj_row0, int step_counter,
                          float &com_dist, Q4 &rq0) {
  const DevModel FFE_CONST &M = model(c);
  Tile &T = c.T;
  const int lane = c.lane;
  const int nref = K.future_steps + 1;
  const ObsLayout L = obs_layout(M.nobsj, nref);
  if (lane == 0) {
    obs[L.acc] = s_acc.x; obs[L.acc + 1] = s_acc.y; obs[L.acc + 2] = s_acc.z;
    obs[L.gyro] = s_gyro.x; obs[L.gyro + 1] = s_gyro.y; obs[L.gyro + 2] = s_gyro.z;
    obs[L.vel] = s_vel.x; obs[L.vel + 1] = s_vel.y; obs[L.vel + 2] = s_vel.z;
    obs[L.zaxis] = T.xmat[0][6]; obs[L.zaxis + 1] = T.xmat[0][7]; obs[L.zaxis + 2] = T.xmat[0][8];
  }
  if (lane < M.nobsj) {
    obs[L.jpos + lane] = T.qpos[M.obsj_qadr[lane]];
    obs[L.jvel + lane] = T.qvel[M.obsj_dof[lane]];
  }
  float cd = 0.f;
  Q4 r0 = {1.f, 0.f, 0.f, 0.f};
  if (lane < nref) {
    const double *r = K.ref_qpos + ((size_t)traj_row0 + step_counter + lane) * 7;
    V3 dv = {(float)(r[0] - T.rootpos[0]), (float)(r[1] - T.rootpos[1]), (float)(r[2] - T.rootpos[2])};
    V3 e = mtv(ldm(T.xmat[0]), dv);
    obs[L.rdisp + 3 * lane] = e.x; obs[L.rdisp + 3 * lane + 1] = e.y; obs[L.rdisp + 3 * lane + 2] = e.z;
    Q4 fq = {T.qpos[3], T.qpos[4], T.qpos[5], T.qpos[6]};
    Q4 dq = qmul(quat_recip(fq), Q4{(float)r[3], (float)r[4], (float)r[5], (float)r[6]});
    obs[L.rquat + 4 * lane] = dq.w; obs[L.rquat + 4 * lane + 1] = dq.x; obs[L.rquat + 4 * lane + 2] = dq.y; obs[L.rquat + 4 * lane + 3] = dq.z;
    if (lane == 0) { cd = sqrtf(dot(e, e)); r0 = dq; }
  }
  com_dist = __shfl(cd, 0);
  rq0 = {__shfl(r0.w, 0), __shfl(r0.x, 0), __shfl(r0.y, 0), __shfl(r0.z, 0)};
}

__device__ __forceinline__ void load_lane_consts(Ctx &c) {
  const DevModel FFE_CONST &M = *c.Mp;
  c.dinv[0] = c.dinv[1] = 0.f;
  c.la_pack = (unsigned)M.d_madr[c.lane] | ((unsigned)M.d_depth[c.lane] << 10) | ((unsigned)M.d_ndesc[c.lane] << 16);
  c.seq0 = M.br_seq[c.lane]; c.seq1 = M.br_seq[kWave + c.lane]; c.seq2 = M.br_seq[2 * kWave + c.lane]; c.seq3 = M.br_seq[3 * kWave + c.lane];
  for (int e = c.lane; e < M.nM; e += kWave) c.T.colmadr[e] = M.colmadr[e];  // (host table: no dependent index chase at launch)
}

// ------------------------------------------------------------------------------------------------ the step kernel
// One launch = one dm_env step of every env.  mode: 0 = step (auto-reset envs that ended), 1 = reset all,
// 2 = bare physics: `nphys` mj_steps with ctrl taken verbatim from act[B][nu] (no task, no outputs) - BASELINE config 2.
__global__ __launch_bounds__(kWave, FFE_WAVES_PER_SIMD) void flight_step_kernel(const DevModel *__restrict__ Mp_, const TaskDev *__restrict__ Kp_, EnvState *__restrict__ states, const float *__restrict__ act,
                                                              float *__restrict__ obs_out, float *__restrict__ reward_out,
                                                              float *__restrict__ discount_out, int *__restrict__ step_type_out, int batch, int mode, int nphys,
                                                              const int *__restrict__ order, int *__restrict__ cost,
                                                              const unsigned char *__restrict__ reset_mask) {
  __shared__ Tile T;
  const DevModel FFE_CONST *Mp = (const DevModel FFE_CONST *)Mp_;
  const TaskDev FFE_CONST *Kp = (const TaskDev FFE_CONST *)Kp_;
  const DevModel FFE_CONST &M = *Mp;
  const TaskDev FFE_CONST &K = *Kp;
  if ((int)blockIdx.x >= batch) return;
  TRACE_BEGIN;
  const int env = order[blockIdx.x];  // most expensive envs first (launch_order.hpp)
  if (mode == 3) {  // ffe_reset_envs: only the masked envs start a new episode; the others keep state and output rows
    if (!reset_mask[env]) return;
    mode = 1;
  }
  const int lane = threadIdx.x;
  EnvState &S = states[env];
  Ctx c{Mp, T, lane, K.flags, 0.f, 0.f, {0.f, 0.f}, 0u, 0u, 0u, 0u, 0u};
#ifdef FFE_DBGCF
  c.dbg_env = env;
#endif
  c.pc = -1;
  c.nrare = 0; c.nfac = 0;
#ifdef FFE_TRACE
  c.tr_coll = 0u;
#endif
#ifdef FFE_STAMPS
  c.st_t0 = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < 20; k++) c.st_acc[k] = 0;
#endif
  load_lane_consts(c);
  if (lane < kNSD) { c.sd_nx = S.sd_n[lane][0]; c.sd_ny = S.sd_n[lane][1]; c.sd_nz = S.sd_n[lane][2]; c.sd_t = S.sd_n[lane][3]; c.sd_pid = S.sd_pid[lane]; }
  c.sd_cnt = S.s1_valid ? S.sd_cnt : 0;
  float *obs = obs_out + (size_t)env * K.obs_dim;
  const bool phys_only = (mode == 2);
  const int nsub = phys_only ? nphys : M.nsub;

  if (lane < 3) T.rootpos[lane] = S.rootpos[lane];
  if (lane < 9) T.sens[lane] = 0.f;
  unsigned long long lo_mask = S.lo_mask, hi_mask = S.hi_mask, in_lo = S.in_lo_mask, in_hi = S.in_hi_mask;
  int wb_step = S.wb_step, wb_idx = S.wb_freq_idx, step_counter = S.step_counter, traj_idx = S.traj_idx;
  double wb_cf = S.wb_ctrl_freq;
  int wb_off = S.wb_off, wb_len = S.wb_len, traj_row0 = S.traj_row0;
  const bool do_reset = !phys_only && ((mode == 1) || (S.needs_reset != 0));
  int iters = 0;
  // the stage-1 carry-over block of the previous launch (2.4 KB) depends on nothing the prologue computes: its loads are issued
  // here, ahead of the prologue's chain of dependent table reads, and land in LDS when the physics loop starts
  const bool have_saved = !do_reset && S.s1_valid != 0 && !DBG(c, DBG_NO_CARRY);
  constexpr int kCarry = (kMaxDof * 6 + kWave - 1) / kWave;
  float pre_c[kCarry], pre_b[kCarry], pre_f = 0.f, pre_m = 0.f;
  if (have_saved) {
#pragma unroll
    for (int k = 0; k < kCarry; k++) {
      const int e = lane + k * kWave;
      pre_c[k] = e < kMaxDof * 6 ? S.s1_cdof[e] : 0.f; pre_b[k] = e < kMaxDof * 6 ? S.s1_buf[e] : 0.f;
    }
    pre_f = S.s1_f[lane];
    pre_m = lane < 12 ? S.s1_misc[lane] : 0.f;
  }

  // ---- prepare: either start a new episode or load the env's state and run the task pre-step
  unsigned long long episode = S.episode;
  float ctrl_reg = 0.f;
  if (do_reset) {
    // ref: flight_imitation.py:93-147 + composer reset
    double phase;
    if (S.forced_traj >= 0) { traj_idx = S.forced_traj; phase = S.forced_phase; }
    else {
      traj_idx = (int)(env_rng(K.seed, K.env_id_base + env, episode, 0) % (unsigned long long)K.ntraj);
      phase = (double)(env_rng(K.seed, K.env_id_base + env, episode, 1) >> 11) * (1.0 / 9007199254740992.0);
    }
    episode++;
    step_counter = 0;
    // ref: pattern_generators.py:121-157 reset
    wb_cf = K.base_freq;
    wb_idx = wave_argmin_absdiff(K.beat_freqs, K.nfreq, wb_cf, lane);
    int off = K.tab_off[wb_idx], len = K.tab_off[wb_idx + 1] - off;
    wb_off = off; wb_len = len;
    wb_step = wave_argmin_absdiff(K.phase + off, len, phase, lane);
    const int nxt = wb_step + 1 < len ? wb_step + 1 : 0;  // the reference reads [step+1] unguarded
    const int row0 = K.traj_off[traj_idx];
    traj_row0 = row0;
    const double *rq = K.ref_qpos + (size_t)row0 * 7, *rv = K.ref_qvel + (size_t)row0 * 6;
    if (lane < kMaxDof + 4) { T.qpos[lane] = lane < M.nq ? M.qpos0[lane] : 0.f; T.qvel[lane] = 0.f; }
    SYNC();
    if (lane == 0) {
      T.qpos[3] = (float)rq[3]; T.qpos[4] = (float)rq[4]; T.qpos[5] = (float)rq[5]; T.qpos[6] = (float)rq[6];
      T.qvel[0] = (float)rv[0]; T.qvel[1] = (float)rv[1]; T.qvel[2] = (float)rv[2];  // initialize_qvel: linear only
    }
    if (lane < 7) T.ghost[lane] = rq[lane];
    if (lane < 3) T.rootpos[lane] = rq[lane];
    if (lane < M.nwing) {
      float q0 = K.traj[(size_t)(off + wb_step) * 6 + lane], q1 = K.traj[(size_t)(off + nxt) * 6 + lane];
      T.qpos[M.wing_qadr[lane]] = q0;
      T.qvel[M.wing_dof[lane]] = (float)(((double)q1 - (double)q0) / K.dt_ctrl);
    }
    lo_mask = hi_mask = in_lo = in_hi = 0ULL;
    SYNC();
  } else if (phys_only) {
    if (lane < kMaxDof + 4) { T.qpos[lane] = S.qpos[lane]; T.qvel[lane] = S.qvel[lane]; }
    if (lane < 13) T.ghost[lane] = lane == 3 ? 1.0 : 0.0;
    ctrl_reg = lane < M.nu ? act[(size_t)env * M.nu + lane] : 0.f;
    SYNC();
  } else {
    // before_step (ref: flight_imitation.py:149-167, base.py:190-193, fruitfly.py:480-492).  Every global read whose address the
    // state record already gives - the action entries, the wing-beat table row of the no-switch case, the reference rows - is
    // issued here in one batch, next to the state's own vectors: the prologue is a chain of dependent memory round trips at the
    // moment all resident waves start together, and each level removed from it is a microsecond or more per wave.
    const float *a_in = act + (size_t)env * M.naction;
    const int ai = lane < M.nu ? M.a_action[lane] : -1;
    float a_raw = ai >= 0 ? a_in[ai] : 0.f;
    float au_raw = a_in[M.user_action];
    const int step1 = wb_step + 1 < wb_len ? wb_step + 1 : 0;
    float tgt = lane < M.nwing ? K.traj[(size_t)(wb_off + step1) * 6 + lane] : 0.f;  // (re-read below if the table switches)
    const size_t row = (size_t)traj_row0 + step_counter;
    const double gh = lane < 7 ? K.ref_qpos[row * 7 + lane] : (lane < 13 ? K.ref_qvel[row * 6 + (lane - 7)] : 0.0);
    if (lane < kMaxDof + 4) { T.qpos[lane] = S.qpos[lane]; T.qvel[lane] = S.qvel[lane]; }
    // acme CanonicalSpecWrapper folded in (ref: train_dmpo_ray.py:128-129; tasks/task_utils.py:53-76 canonical2real)
    auto to_real = [&](float v, int k) -> float {
      if (!(v == v)) v = 0.f;  // NaN scrub
      if (K.canonical) {
        if (K.clip) v = fminf(fmaxf(v, -1.f), 1.f);
        v = 0.5f * (v + 1.f) * (K.act_hi[k] - K.act_lo[k]) + K.act_lo[k];
      }
      return v;
    };
    const float act_user = to_real(au_raw, M.user_action);
    {
      // ref: pattern_generators.py:159-191 step
      wb_step = step1;
      wb_cf = wbpg_filter(wb_cf, K.rate, K.base_freq, K.rel_range, (double)act_user);
      int idx_new = DBG(c, DBG_SKIP_WBPG) ? wb_idx
                    : (K.grid_inv_step > 0.0 ? grid_argmin_absdiff(K.beat_freqs, K.nfreq, wb_cf, K.grid_inv_step)
                                             : wave_argmin_absdiff(K.beat_freqs, K.nfreq, wb_cf, lane));
      if (idx_new != wb_idx) {
        double cur = K.phase_frac[wb_off + wb_step];
        int noff = K.tab_off[idx_new], nlen = K.tab_off[idx_new + 1] - noff;
        wb_step = wave_argmin_absdiff(K.phase_frac + noff, nlen, cur, lane);
        wb_idx = idx_new; wb_off = noff; wb_len = nlen;
        if (lane < M.nwing) tgt = K.traj[(size_t)(wb_off + wb_step) * 6 + lane];
      }
    }
    STAMP(15);  // prologue up to the WBPG step
    if (lane < M.nu) T.ctrl[lane] = ai >= 0 ? to_real(a_raw, ai) : 0.f;
    if (lane < 13) T.ghost[lane] = gh;
    SYNC();
    if (lane < M.nwing) {
      // action[wings] += target - qpos[wing]; the wing actuators are the ctrl slots fed by those action entries
      float add = tgt - T.qpos[M.wing_qadr[lane]];
      const int u = M.wing_ctrl[lane];  // the ctrl slot fed by this wing's action entry
      if (u >= 0) T.ctrl[u] += add;
    }
    SYNC();
    ctrl_reg = lane < M.nu ? T.ctrl[lane] : 0.f;
    SYNC();
    step_counter++;
  }
  // ---- physics.  dm_control's legacy step is mj_step2 then mj_step1, so the position/velocity stage is evaluated
  //      once up front and again after every integration; buffered sensors take one sample per substep.  A reset
  //      is the same pipeline run once without actuation and without integrating (mj_forward).
  {
    // per-lane, fixed over the control step: kept in the (now dead) control staging slots rather than in a register
    const float abase = do_reset ? 0.f : actuation_base(c, ctrl_reg);
    if (lane < kMaxAct) T.ctrl[lane] = abase;
  }
  // the task's wave-uniform values are not needed before the epilogue: parked in LDS, they do not sit in (or spill from)
  // vector registers through the physics loop
  if (lane == 0) {
    T.park_d[0] = wb_cf; T.park_d[1] = __longlong_as_double((long long)episode);
    T.park_i[0] = wb_step; T.park_i[1] = wb_idx; T.park_i[2] = step_counter; T.park_i[3] = traj_idx; T.park_i[4] = 0; T.park_i[5] = 0;
  }
  STAMP(11);  // prologue: state load, WBPG, action mixing (or episode reset)
  const int nst = do_reset ? 1 : nsub;
#pragma unroll 1
  for (int s = 0; s <= nst; s++) {
    if (s == 0 && have_saved) {
#pragma unroll
      for (int k = 0; k < kCarry; k++) {
        const int e = lane + k * kWave;
        if (e < kMaxDof * 6) { (&T.cdof[0][0])[e] = pre_c[k]; (&T.buf[0][0])[e] = pre_b[k]; }
      }
      c.f_smooth_nb = pre_f;
      if (lane < kMC) {  // the contacts of that position stage
        const float *o = S.ct_f[lane];
        c.ct_nx = o[0]; c.ct_ny = o[1]; c.ct_nz = o[2]; c.ct_px = o[3]; c.ct_py = o[4]; c.ct_pz = o[5]; c.ct_dist = o[6]; c.ct_incl = o[7]; c.ct_invw = o[8];
        c.ct_l1 = S.ct_i[lane][0]; c.ct_l2 = S.ct_i[lane][1]; c.ct_pid = S.ct_i[lane][2];
      }
      c.nct = S.nct;
      if (lane < 9) T.xmat[0][lane] = pre_m;
      else if (lane < 12) T.sens[lane] = pre_m;  // CoM (see set_com)
      SYNC();
    } else if (!DBG(c, DBG_SKIP_STAGE1) || s == 0) { stage1(c); flight_collide(c); STAMP(16); }
    if (lane < 6 && (do_reset || s > 0)) {
      // buffered velocity sensors at the thorax site: gyro = body-frame angular velocity, velocimeter = R^T v
      float add;
      if (lane < 3) add = T.xmat[0][lane] * T.qvel[0] + T.xmat[0][3 + lane] * T.qvel[1] + T.xmat[0][6 + lane] * T.qvel[2];
      else add = T.qvel[lane];
      T.sens[lane < 3 ? 6 + lane : lane] += add;
    }
    if (s == nst) break;
    float qa = 0.f;
    if (!do_reset) qa = actuation(c, lane < kMaxAct ? T.ctrl[lane] : 0.f);
    STAMP(12);  // sensor accumulation + actuation
    int it = 0;
    const V3 acc = stage2(c, qa, !do_reset, !do_reset && !phys_only && !DBG(c, DBG_SKIP_GHOST), K.ghost_accel_z, lo_mask, hi_mask, in_lo, in_hi, it);
    if (lane == 0) { T.park_i[4] += it; if (s < 4) T.park_i[5] |= min(c.nct, 15) << (4 * s); }
    if (lane < 3) T.sens[lane] += lane == 0 ? acc.x : (lane == 1 ? acc.y : acc.z);
    if (do_reset) break;
  }
  SYNC();
  wb_cf = T.park_d[0]; episode = (unsigned long long)__double_as_longlong(T.park_d[1]);
  wb_step = T.park_i[0]; wb_idx = T.park_i[1]; step_counter = T.park_i[2]; traj_idx = T.park_i[3]; iters = T.park_i[4];
  V3 s_acc = {T.sens[0], T.sens[1], T.sens[2]}, s_gyro = {T.sens[3], T.sens[4], T.sens[5]}, s_vel = {T.sens[6], T.sens[7], T.sens[8]};
  if (!phys_only) {
  const float inv = (do_reset && K.pad_first_obs) ? 1.f : 1.f / (float)nsub;
  float cdist;
  Q4 rq0;
  const int traj_row0 = K.traj_off[traj_idx], traj_rows = K.traj_off[traj_idx + 1] - traj_row0;
  if (DBG(c, DBG_SKIP_OBS)) { cdist = 0.f; rq0 = {1.f, 0.f, 0.f, 0.f}; }
  else write_obs(c, K, obs, inv * s_acc, inv * s_gyro, inv * s_vel, traj_row0, step_counter, cdist, rq0);
  if (do_reset) {
    if (lane == 0) { reward_out[env] = 0.f; discount_out[env] = 1.f; step_type_out[env] = FFE_STEP_FIRST; S.needs_reset = 0; S.forced_traj = -1; }
  } else {
    // ---- check_termination (ref: flight_imitation.py:198-209, base.py:214-217); qacc is the last substep's
    float qn2 = wave_sum(lane < M.nv ? c.qacc * c.qacc : 0.f);
    float height = (float)T.rootpos[2];
    // ref: flight_imitation.py:107-108, per episode: min(len(this trajectory), round(time_limit / control_timestep)) - (future_steps + 1)
    int lim = traj_rows < K.time_limit_steps ? traj_rows : K.time_limit_steps;
    int traj_timesteps = lim - (K.future_steps + 1);
    bool reached_end = (step_counter == traj_timesteps);
    bool bad = !(qn2 == qn2) || !(sqrtf(qn2) <= 1e14f);
    bool term = height < 0.2f || cdist > (float)K.terminal_com_dist || reached_end || bad;
    // ---- reward (ref: flight_imitation.py:169-196): ghost CoM vs walker CoM, and root orientation error
    if (lane == 0) {
      const double ox = -0.03697732, oy = 0.00029205, oz = -0.0142447;  // ref: task_utils.py:188
      const double *gpos = T.ghost, *rootpos = T.rootpos;
      const V3 com_e = get_com(c);
      double w = T.ghost[3], x = T.ghost[4], y = T.ghost[5], z = T.ghost[6];
      double n2 = w * w + x * x + y * y + z * z;
      double gx = gpos[0] + ((w * w + x * x - y * y - z * z) * ox + 2 * (x * y - w * z) * oy + 2 * (x * z + w * y) * oz) / n2;
      double gy = gpos[1] + (2 * (x * y + w * z) * ox + (w * w - x * x + y * y - z * z) * oy + 2 * (y * z - w * x) * oz) / n2;
      double gz = gpos[2] + (2 * (x * z - w * y) * ox + 2 * (y * z + w * x) * oy + (w * w - x * x - y * y + z * z) * oz) / n2;
      double dx = gx - (rootpos[0] + (double)com_e.x), dy = gy - (rootpos[1] + (double)com_e.y), dz = gz - (rootpos[2] + (double)com_e.z);
      float r1 = fmaxf(0.f, 1.f - (float)sqrt(dx * dx + dy * dy + dz * dz) / 0.4f);
      float r2 = fmaxf(0.f, 1.f - quat_dist_short_arc(Q4{1.f, 0.f, 0.f, 0.f}, rq0) / 3.14159265358979f);
      float reward = r1 * r2;
      float discount = (term && !reached_end) ? 0.f : 1.f;
      if (bad || !(reward == reward)) { reward = 0.f; discount = 0.f; }
      bool time_up = step_counter >= K.episode_limit_steps;  // composer.Environment: physics.time() >= time_limit
      reward_out[env] = reward; discount_out[env] = discount;
      step_type_out[env] = (term || time_up) ? FFE_STEP_LAST : FFE_STEP_MID;
      S.needs_reset = (term || time_up) ? 1 : 0;
    }
  }
  if (lane < 7) S.ghost[lane] = T.ghost[lane];
  if (lane == 0) S.episode = episode;
  }  // !phys_only
  STAMP(9);  // epilogue: observation, reward, termination
  // ---- store state
  if (lane < kMaxDof + 4) { S.qpos[lane] = T.qpos[lane]; S.qvel[lane] = T.qvel[lane]; }
  if (lane == 0) {
    S.rootpos[0] = T.rootpos[0]; S.rootpos[1] = T.rootpos[1]; S.rootpos[2] = T.rootpos[2];
    S.wb_ctrl_freq = wb_cf; S.wb_step = wb_step; S.wb_freq_idx = wb_idx; S.step_counter = step_counter; S.traj_idx = traj_idx;
    S.wb_off = wb_off; S.wb_len = wb_len; S.traj_row0 = traj_row0;
    S.lo_mask = lo_mask; S.hi_mask = hi_mask; S.in_lo_mask = in_lo; S.in_hi_mask = in_hi; S.solver_iters = iters;
    S.nactive = __popcll(lo_mask) + __popcll(hi_mask);
    // Key of the next launch's order = how long this env's wave is expected to run next step, in units of about 5 us.  A launch is two
    // rounds of resident waves and ends with its slowest one (wave lifetimes: mean 270 us, p99 390, longest 550 - 630), so the long ones
    // must start first.  The estimate is a least-squares fit of traced lifetimes (tools/wave_cost_fit.py): 14 us per factorisation, 2.6 us
    // per contact and substep, 18 us per second-pass collision call (a wing meeting the abdomen's cylinders).  Joint limits are hit, and
    // the wings pass the abdomen, at fixed phases of the wing beat, so the work is periodic in the WBPG phase: the key is the larger of this
    // step's work and the work recorded, one beat earlier, in the phase bins the next step will fall into.  Measured against the alternatives
    // (DESIGN.md section 6 item 9): round 2's key (solver passes only) 9.54 M env-steps/s, the wave's own measured lifetime 9.55 M (it mostly
    // records which round the wave ran in), lifetime + phase history 10.35 M, this one 10.48 M.
    const int hist_c = T.park_i[5], csum = (hist_c & 15) + ((hist_c >> 4) & 15) + ((hist_c >> 8) & 15) + ((hist_c >> 12) & 15);
    const int work = min(255, (11 * c.nfac + 2 * csum + 14 * c.nrare) / 2);
    int pred = work;
    if (!phys_only) {
      const int off = K.tab_off[wb_idx], len = K.tab_off[wb_idx + 1] - off;
      const int b = min(31, (int)(K.phase_frac[off + wb_step] * 32.0));
      const int nb = min(31, (int)(K.phase_frac[off + (wb_step + 1 < len ? wb_step + 1 : 0)] * 32.0));
      if (do_reset) { for (int k = 0; k < 32; k++) S.cost_hist[k] = 0; }
      else S.cost_hist[b] = (unsigned char)work;
      pred = max(pred, max((int)S.cost_hist[nb], (int)S.cost_hist[(nb + 1) & 31]));
    }
    cost[env] = pred;
  }
  // carry the final stage-1 results to the next launch (a reset's single evaluation is that of the FIRST state)
  for (int e = lane; e < kMaxDof * 6; e += kWave) { S.s1_cdof[e] = (&T.cdof[0][0])[e]; S.s1_buf[e] = (&T.buf[0][0])[e]; }
  S.s1_f[lane] = c.f_smooth_nb;
  if (lane < 9) S.s1_misc[lane] = T.xmat[0][lane];
  if (lane == 0) { const V3 com_e = get_com(c); S.s1_misc[9] = com_e.x; S.s1_misc[10] = com_e.y; S.s1_misc[11] = com_e.z; S.s1_valid = 1; }
  if (lane < kMC) {
    float *o = S.ct_f[lane];
    o[0] = c.ct_nx; o[1] = c.ct_ny; o[2] = c.ct_nz; o[3] = c.ct_px; o[4] = c.ct_py; o[5] = c.ct_pz; o[6] = c.ct_dist; o[7] = c.ct_incl; o[8] = c.ct_invw;
    S.ct_i[lane][0] = c.ct_l1; S.ct_i[lane][1] = c.ct_l2; S.ct_i[lane][2] = c.ct_pid;
  }
  if (lane < kNSD) { S.sd_n[lane][0] = c.sd_nx; S.sd_n[lane][1] = c.sd_ny; S.sd_n[lane][2] = c.sd_nz; S.sd_n[lane][3] = c.sd_t; S.sd_pid[lane] = c.sd_pid; }
  if (lane == 0) { S.nct = c.nct; S.sd_cnt = c.sd_cnt; S.ct_pad[0] = c.ct_ovf; S.ct_pad[1] = T.park_i[5]; }
#ifdef FFE_STAMPS
  STAMP(10);
  if (lane == 0) for (int k = 0; k < 20; k++) atomicAdd(&g_stamps[k], c.st_acc[k]);
#endif
  TRACE_END(blockIdx.x, (unsigned long long)((unsigned)(iters & 0xff) | ((unsigned)(__popcll(lo_mask) + __popcll(hi_mask)) << 8) | ((unsigned)(tr_prev & 0xffff) << 16)) | ((unsigned long long)(T.park_i[5] & 0xffff) << 32), TRACE_HI(c));  // this step's solver iterations, active limits at its end, the sort key it was launched with, the contacts each substep used (4 bits each)
}

__global__ void init_states_kernel(EnvState *states, int *order, int *cost, int batch) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch) return;
  EnvState z;
  memset(&z, 0, sizeof(z));
  z.needs_reset = 1; z.forced_traj = -1;
  states[i] = z;
  order[i] = i;
  cost[i] = 0;
}

__global__ void get_state_kernel(const EnvState *states, double *qpos, double *qvel, int batch, int nq, int nv) {
  int env = blockIdx.x, lane = threadIdx.x;
  if (env >= batch) return;
  const EnvState &S = states[env];
  if (lane < nq) qpos[(size_t)env * nq + lane] = lane < 3 ? S.rootpos[lane] : (double)S.qpos[lane];
  if (lane < nv) qvel[(size_t)env * nv + lane] = (double)S.qvel[lane];
}
__global__ void set_state_kernel(EnvState *states, const double *qpos, const double *qvel, int batch, int nq, int nv) {
  int env = blockIdx.x, lane = threadIdx.x;
  if (env >= batch) return;
  EnvState &S = states[env];
  if (lane < nq) { if (lane < 3) S.rootpos[lane] = qpos[(size_t)env * nq + lane]; else S.qpos[lane] = (float)qpos[(size_t)env * nq + lane]; }
  if (lane < nv) S.qvel[lane] = (float)qvel[(size_t)env * nv + lane];
  if (lane == 0) { S.lo_mask = 0; S.hi_mask = 0; S.in_lo_mask = 0; S.in_hi_mask = 0; S.s1_valid = 0; }
}
__global__ void get_task_state_kernel(const EnvState *states, int *ints, double *reals, int batch) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch) return;
  const EnvState &S = states[i];
  int *o = ints + (size_t)i * 8;
  o[0] = S.wb_step; o[1] = S.wb_freq_idx; o[2] = S.step_counter; o[3] = S.traj_idx; o[4] = S.needs_reset; o[5] = S.nactive; o[6] = S.solver_iters; o[7] = S.nct | ((S.ct_pad[0] & 255) << 8) | (S.ct_pad[1] << 16);  // contacts of the current position stage | some position stage of the last step met more contacts than the solver carries (the deepest kMC were kept) | contacts each of the last step's substeps used, 4 bits each
  double *r = reals + (size_t)i * 8;
  r[0] = S.wb_ctrl_freq;
  for (int k = 0; k < 7; k++) r[1 + k] = S.ghost[k];
}
__global__ void force_next_kernel(EnvState *states, const int *traj, const double *phase, int batch) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch) return;
  states[i].forced_traj = traj[i];
  states[i].forced_phase = phase[i];
}
__global__ void test_quat_kernel(int op, const float *a, const float *b, float *out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Q4 qa = {a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]}, qb = {b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]};
  Q4 r = {0, 0, 0, 0};
  if (op == 0) r = qmul(qa, qb);
  else if (op == 1) r = quat_recip(qa);
  else if (op == 2) { V3 v = rotate_vec_with_quat(V3{qa.w, qa.x, qa.y}, qb); r = {v.x, v.y, v.z, 0.f}; }
  else if (op == 3) r = {quat_dist_short_arc(qa, qb), 0.f, 0.f, 0.f};
  else if (op == 4) r = qmul(quat_recip(qa), qb);
  out[4 * i] = r.w; out[4 * i + 1] = r.x; out[4 * i + 2] = r.y; out[4 * i + 3] = r.z;
}

}  // namespace ffe

// ================================================================================================ C ABI
using namespace ffe;

struct ffe_env {
  ffb::BallEnv *ball = nullptr;  // walk_on_ball handles dispatch to ball_env.hip; everything below is the flight env
  int device = 0, batch = 0;
  DevModel dm{};
  TaskDev task{};
  DevModel *dm_dev = nullptr;
  TaskDev *task_dev = nullptr;
  HostModel host;
  EnvState *states = nullptr;
  int *order = nullptr, *cost = nullptr;  // launch order of the envs and its sort keys (launch_order.hpp)
  bool timing = false; double timing_ms = 0.0;  // ffe_time_kernel: events around the step kernel alone
  unsigned char *arena = nullptr;
  std::vector<void *> allocs;
  std::string err;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int *forced_traj_dev = nullptr;      // staging of ffe_force_next_episode (allocated on first use)
  double *forced_phase_dev = nullptr;
};

static thread_local std::string g_err;

// Every entry point runs on the handle's device and leaves the caller's current device untouched.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) { switched = (hipSetDevice(device) == hipSuccess); }
  }
  ~DeviceGuard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};

#define HIP_OK(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

template <typename T>
static T *upload(ffe_env *h, const T *src, size_t n) {
  T *p = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&p), (n ? n : 1) * sizeof(T)));
  h->allocs.push_back(p);
  if (n) HIP_OK(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
  return p;
}

// walk_on_ball handles: run `body` on the ball env, translating exceptions into the ABI's error codes
#define FFE_BALL_DISPATCH(h, body)                                    \
  if ((h) && (h)->ball) {                                             \
    try { body; } catch (const std::exception &e_) { (h)->err = e_.what(); return -2; } \
    return 0;                                                         \
  }

extern "C" {

int ffe_create_walk_on_ball(const void *model_blob, size_t blob_size, const ffe_ball_task *task, int batch, int device, ffe_handle *out) {
  if (!out) return -1;
  *out = nullptr;
  std::unique_ptr<ffe_env> h(new ffe_env());
  try {
    if (!task) throw std::runtime_error("ffe_create_walk_on_ball: bad arguments");
    ffb::BallTaskHost t{task->time_limit_steps, task->pad_first_obs, task->physics_flags, task->canonical_actions, task->clip_actions,
                        task->control_timestep};
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device: the MI355X path has no CPU fallback");
    if (device < 0 || device >= ndev) throw std::runtime_error("ffe_create_walk_on_ball: no such device");
    DeviceGuard guard(device);
    h->ball = ffb::ball_create(model_blob, blob_size, t, batch, device);
    h->device = device; h->batch = batch;
  } catch (const std::exception &e) { g_err = e.what(); return -1; }
  *out = h.release();
  return 0;
}
int ffe_get_act(ffe_handle h, double *act_dev, void *stream) {
  if (!h || !act_dev || !h->ball) return -1;
  DeviceGuard guard(h->device);
  FFE_BALL_DISPATCH(h, ffb::ball_get_act(h->ball, act_dev, stream));
  return -1;
}
int ffe_set_act(ffe_handle h, const double *act_dev, void *stream) {
  if (!h || !act_dev || !h->ball) return -1;
  DeviceGuard guard(h->device);
  FFE_BALL_DISPATCH(h, ffb::ball_set_act(h->ball, act_dev, stream));
  return -1;
}

const char *ffe_version(void) { return "flybody_amd 0.1 (gfx950, wave-per-env)"; }
const char *ffe_last_error(ffe_handle h) { return h ? h->err.c_str() : g_err.c_str(); }

int ffe_create_flight(const void *model_blob, size_t blob_size, const ffe_flight_task *task, int batch, int device, uint64_t seed,
                      uint64_t env_id_base, ffe_handle *out) {
  if (!out) return -1;
  *out = nullptr;
  std::unique_ptr<ffe_env> h(new ffe_env());
  std::unique_ptr<DeviceGuard> guard;
  try {
    if (!model_blob || !task || batch <= 0) throw std::runtime_error("ffe_create_flight: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      throw std::runtime_error("no HIP device: the MI355X path has no CPU fallback");
    if (device < 0 || device >= ndev) throw std::runtime_error("ffe_create_flight: no such device");
    guard.reset(new DeviceGuard(device));
    Blob blob(model_blob, blob_size);
    h->host = build_host_model(blob);
    h->device = device; h->batch = batch;
    h->arena = upload(h.get(), h->host.arena.data(), h->host.arena.size());
    h->host.fixup(h->dm, h->arena);
    const ffe_flight_task &t = *task;
    if (t.wb_nfreq <= 0 || !t.wb_beat_freqs || !t.wb_tab_off || !t.wb_traj || !t.wb_phase || t.ntraj <= 0 || t.traj_len <= 0 || !t.ref_qpos || !t.ref_qvel)
      throw std::runtime_error("ffe_create_flight: incomplete task tables");
    if (t.future_steps + 1 > kMaxFuture) throw std::runtime_error("future_steps too large");
    // per-trajectory row offsets (ref: trajectory_loaders.py:98-100 - trajectories of different lengths)
    std::vector<int> toff((size_t)t.ntraj + 1);
    for (int i = 0; i <= t.ntraj; i++) toff[i] = t.traj_off ? t.traj_off[i] : i * t.traj_len;
    if (toff[0] != 0) throw std::runtime_error("traj_off[0] must be 0");
    for (int i = 0; i < t.ntraj; i++)
      if (toff[i + 1] - toff[i] < t.future_steps + 2) throw std::runtime_error("trajectories too short");
    const size_t ref_rows = (size_t)toff[t.ntraj];
    if (h->dm.user_action < 0 || h->dm.nwing != 6) throw std::runtime_error("model is not the flight model");
    h->dm.nsub = (int)llround(t.wb_dt_ctrl / (double)blob.get("opt").f(0));
    const int rows = t.wb_tab_off[t.wb_nfreq];
    std::vector<double> frac(rows);
    std::vector<float> trajf((size_t)rows * 6);
    for (int i = 0; i < rows; i++) frac[i] = std::fmod(t.wb_phase[i], 1.0);
    for (size_t i = 0; i < trajf.size(); i++) trajf[i] = (float)t.wb_traj[i];
    TaskDev &K = h->task;
    K.nfreq = t.wb_nfreq; K.ntraj = t.ntraj; K.future_steps = t.future_steps; K.time_limit_steps = t.time_limit_steps;
    K.episode_limit_steps = t.episode_limit_steps > 0 ? t.episode_limit_steps : t.time_limit_steps;
    K.pad_first_obs = t.pad_first_obs; K.flags = t.physics_flags; K.canonical = t.canonical_actions; K.clip = t.clip_actions;
    for (int k = 0; k < 16; k++) { K.act_lo[k] = k < h->dm.naction ? h->host.action_min[k] : 0.f; K.act_hi[k] = k < h->dm.naction ? h->host.action_max[k] : 0.f; }
    K.base_freq = t.wb_base_freq; K.rel_range = t.wb_rel_range; K.rate = t.wb_rate; K.dt_ctrl = t.wb_dt_ctrl;
    K.terminal_com_dist = t.terminal_com_dist; K.ghost_accel_z = t.ghost_accel_z;
    K.grid_inv_step = 0.0;
    if (t.wb_nfreq >= 3) {  // evenly spaced and increasing (ref: pattern_generators.py:65-69 np.linspace)? then the lookup is arithmetic
      const double step = (t.wb_beat_freqs[t.wb_nfreq - 1] - t.wb_beat_freqs[0]) / (t.wb_nfreq - 1);
      bool even = step > 0;
      for (int i = 0; i < t.wb_nfreq && even; i++) even = std::fabs(t.wb_beat_freqs[i] - (t.wb_beat_freqs[0] + i * step)) < 0.25 * step;
      if (even) K.grid_inv_step = 1.0 / step;
    }
    set_off(K.beat_freqs, (size_t)upload(h.get(), t.wb_beat_freqs, (size_t)t.wb_nfreq));
    set_off(K.tab_off, (size_t)upload(h.get(), t.wb_tab_off, (size_t)t.wb_nfreq + 1));
    set_off(K.phase, (size_t)upload(h.get(), t.wb_phase, (size_t)rows));
    set_off(K.phase_frac, (size_t)upload(h.get(), frac.data(), frac.size()));
    set_off(K.traj, (size_t)upload(h.get(), trajf.data(), trajf.size()));
    set_off(K.ref_qpos, (size_t)upload(h.get(), t.ref_qpos, ref_rows * 7));
    set_off(K.ref_qvel, (size_t)upload(h.get(), t.ref_qvel, ref_rows * 6));
    set_off(K.traj_off, (size_t)upload(h.get(), toff.data(), toff.size()));
    K.seed = seed; K.env_id_base = env_id_base;
    K.obs_dim = 12 + 2 * h->dm.nobsj + 7 * (t.future_steps + 1);
    h->host.nobs = K.obs_dim;
    h->dm_dev = upload(h.get(), &h->dm, 1);
    h->task_dev = upload(h.get(), &h->task, 1);
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&h->states), sizeof(EnvState) * (size_t)batch));
    h->allocs.push_back(h->states);
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&h->order), sizeof(int) * (size_t)batch));
    h->allocs.push_back(h->order);
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&h->cost), sizeof(int) * (size_t)batch));
    h->allocs.push_back(h->cost);
    hipLaunchKernelGGL(init_states_kernel, dim3((batch + 255) / 256), dim3(256), 0, 0, h->states, h->order, h->cost, batch);
    HIP_OK(hipGetLastError());
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipEventCreate(&h->ev0));
    HIP_OK(hipEventCreate(&h->ev1));
  } catch (const std::exception &e) {
    g_err = e.what();
    for (void *p : h->allocs) (void)hipFree(p);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    return -1;
  }
  *out = h.release();
  return 0;
}

int ffe_destroy(ffe_handle h) {
  if (!h) return -1;
  DeviceGuard guard(h->device);
  if (h->ball) { ffb::ball_destroy(h->ball); delete h; return 0; }
  for (void *p : h->allocs) (void)hipFree(p);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  delete h;
  return 0;
}

int ffe_spec(ffe_handle h, ffe_spec_t *s) {
  if (!h || !s) return -1;
  if (h->ball) {
    std::memset(s, 0, sizeof(*s));
    ffb::ball_spec(h->ball, &s->nq, &s->nv, &s->nu, &s->action_dim, &s->obs_dim, &s->nsub, &s->physics_timestep, &s->control_timestep);
    s->batch = h->batch;
    // walk_on_ball row: accelerometer 3 | actuator_activation 59 | appendages_pos 21 | ball_qvel 3 | force 18 | gyro 3 |
    //                   joints_pos 85 | joints_vel 85 | touch 6 | velocimeter 3 | world_zaxis 3
    s->off_accelerometer = 0; s->off_gyro = 104; s->off_joints_pos = 107; s->off_joints_vel = 192; s->off_velocimeter = 283; s->off_world_zaxis = 286;
    s->off_ref_displacement = -1; s->off_ref_root_quat = -1; s->n_obs_joints = 85; s->n_ref = 0;
    return 0;
  }
  const DevModel &M = h->dm;
  const int nref = h->task.future_steps + 1, nj = M.nobsj;
  s->batch = h->batch; s->nq = M.nq; s->nv = M.nv; s->nu = M.nu; s->action_dim = M.naction; s->obs_dim = h->task.obs_dim; s->nsub = M.nsub;
  s->physics_timestep = (double)M.h; s->control_timestep = h->task.dt_ctrl;
  s->off_accelerometer = 0; s->off_gyro = 3; s->off_joints_pos = 6; s->off_joints_vel = 6 + nj; s->off_velocimeter = 6 + 2 * nj;
  s->off_world_zaxis = 9 + 2 * nj; s->off_ref_displacement = 12 + 2 * nj; s->off_ref_root_quat = 12 + 2 * nj + 3 * nref;
  s->n_obs_joints = nj; s->n_ref = nref;
  return 0;
}

int ffe_action_bounds(ffe_handle h, float *mn, float *mx) {
  if (!h || !mn || !mx) return -1;
  FFE_BALL_DISPATCH(h, ffb::ball_action_bounds(h->ball, mn, mx));
  std::memcpy(mn, h->host.action_min.data(), h->host.action_min.size() * sizeof(float));
  std::memcpy(mx, h->host.action_max.data(), h->host.action_max.size() * sizeof(float));
  return 0;
}

// the ordering kernel pays for itself only while the launch is a couple of rounds of resident waves: +1.7 % at 8 192 envs,
// -0.8 % at 16 384, -1.9 % at 32 768 (measured with the phase-aware key)
#ifndef FFE_ORDER_MAX_BATCH
#define FFE_ORDER_MAX_BATCH 8192
#endif
static int launch_step(ffe_handle h, const float *act, float *obs, float *rew, float *disc, int32_t *st, void *stream, int mode, int nphys = 0,
                       const uint8_t *mask = nullptr) {
  if (!h) return -1;
  DeviceGuard guard(h->device);
  FFE_BALL_DISPATCH(h, ffb::ball_launch(h->ball, act, obs, rew, disc, st, stream, mode, nphys, mask));
  if (mode != 2 && (!obs || !rew || !disc || !st || (mode == 0 && !act))) { h->err = "null device buffer"; return -1; }
  if (h->timing && hipEventRecord(h->ev0, static_cast<hipStream_t>(stream)) != hipSuccess) return -2;
  hipLaunchKernelGGL(flight_step_kernel, dim3(h->batch), dim3(kWave), 0, static_cast<hipStream_t>(stream), h->dm_dev, h->task_dev, h->states, act, obs, rew,
                     disc, st, h->batch, mode, nphys, h->order, h->cost, mask);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { h->err = hipGetErrorString(e); return -2; }
  if (h->timing && hipEventRecord(h->ev1, static_cast<hipStream_t>(stream)) != hipSuccess) return -2;
  // measured: +2 % env-steps/s at B = 8 192 (two rounds of the 4 096 resident waves); beyond that the tail the order shortens
  // is a smaller share of the launch than the serialised sort kernel itself (-1.5 % at 16 384, -2 % at 32 768): not sorted
  if (mode == 0 && h->batch > 1 && h->batch <= FFE_ORDER_MAX_BATCH && !(h->task.flags & DBG_NO_ORDER)) {
    hipLaunchKernelGGL(ffe_order::order_by_cost, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), h->cost, h->order, h->batch);
    e = hipGetLastError();
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return -2; }
  }
  if (h->timing) {
    float t = 0.f;
    if (hipEventSynchronize(h->ev1) != hipSuccess || hipEventElapsedTime(&t, h->ev0, h->ev1) != hipSuccess) return -2;
    h->timing_ms += t;
  }
  return 0;
}

int ffe_reset(ffe_handle h, float *obs, float *rew, float *disc, int32_t *st, void *stream) { return launch_step(h, nullptr, obs, rew, disc, st, stream, 1); }
int ffe_reset_envs(ffe_handle h, const uint8_t *mask, float *obs, float *rew, float *disc, int32_t *st, void *stream) {
  if (!h) return -1;
  if (!mask) { h->err = "null reset mask"; return -1; }
  return launch_step(h, nullptr, obs, rew, disc, st, stream, 3, 0, mask);
}
int ffe_step(ffe_handle h, const float *act, float *obs, float *rew, float *disc, int32_t *st, void *stream) {
  return launch_step(h, act, obs, rew, disc, st, stream, 0);
}

int ffe_physics_step(ffe_handle h, const float *ctrl, int nsteps, void *stream) {
  if (!h || !ctrl || nsteps <= 0) return -1;
  return launch_step(h, ctrl, nullptr, nullptr, nullptr, nullptr, stream, 2, nsteps);
}

int ffe_force_next_episode(ffe_handle h, const int32_t *traj, const double *phase, void *stream) {
  if (!h || !traj || !phase) return -1;
  if (h->ball) { h->err = "walk_on_ball episodes have no per-episode randomness"; return -1; }
  DeviceGuard guard(h->device);
  try {
    for (int i = 0; i < h->batch; i++)
      if (traj[i] >= h->task.ntraj) throw std::runtime_error("ffe_force_next_episode: trajectory index out of range");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!h->forced_traj_dev) {
      HIP_OK(hipMalloc(reinterpret_cast<void **>(&h->forced_traj_dev), sizeof(int) * h->batch));
      h->allocs.push_back(h->forced_traj_dev);
      HIP_OK(hipMalloc(reinterpret_cast<void **>(&h->forced_phase_dev), sizeof(double) * h->batch));
      h->allocs.push_back(h->forced_phase_dev);
    }
    // pageable host memory: the copies are staged before the calls return, ordered on `s` with the kernel below; the stream
    // is synchronised once so that a second call cannot overwrite the staging buffers under a pending kernel
    HIP_OK(hipMemcpyAsync(h->forced_traj_dev, traj, sizeof(int) * h->batch, hipMemcpyHostToDevice, s));
    HIP_OK(hipMemcpyAsync(h->forced_phase_dev, phase, sizeof(double) * h->batch, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(force_next_kernel, dim3((h->batch + 255) / 256), dim3(256), 0, s, h->states, h->forced_traj_dev, h->forced_phase_dev, h->batch);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(s));
  } catch (const std::exception &e) { h->err = e.what(); return -1; }
  return 0;
}

int ffe_get_state(ffe_handle h, double *qpos, double *qvel, void *stream) {
  if (!h || !qpos || !qvel) return -1;
  DeviceGuard guard(h->device);
  FFE_BALL_DISPATCH(h, ffb::ball_get_state(h->ball, qpos, qvel, stream));
  hipLaunchKernelGGL(get_state_kernel, dim3(h->batch), dim3(kWave), 0, static_cast<hipStream_t>(stream), h->states, qpos, qvel, h->batch, h->dm.nq, h->dm.nv);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
int ffe_set_state(ffe_handle h, const double *qpos, const double *qvel, void *stream) {
  if (!h || !qpos || !qvel) return -1;
  DeviceGuard guard(h->device);
  FFE_BALL_DISPATCH(h, ffb::ball_set_state(h->ball, qpos, qvel, stream));
  hipLaunchKernelGGL(set_state_kernel, dim3(h->batch), dim3(kWave), 0, static_cast<hipStream_t>(stream), h->states, qpos, qvel, h->batch, h->dm.nq, h->dm.nv);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
int ffe_get_task_state(ffe_handle h, int32_t *ints, double *reals, void *stream) {
  if (!h || !ints || !reals) return -1;
  DeviceGuard guard(h->device);
  FFE_BALL_DISPATCH(h, ffb::ball_get_task_state(h->ball, ints, reals, stream));
  hipLaunchKernelGGL(get_task_state_kernel, dim3((h->batch + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), h->states, ints, reals, h->batch);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int ffe_time_steps(ffe_handle h, const float *act, float *obs, float *rew, float *disc, int32_t *st, int iters, void *stream, float *ms) {
  if (!h || !ms || iters <= 0) return -1;
  DeviceGuard guard(h->device);
  FFE_BALL_DISPATCH(h, *ms = ffb::ball_time_steps(h->ball, act, obs, rew, disc, st, iters, stream));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipEventRecord(h->ev0, s) != hipSuccess) return -2;
  for (int i = 0; i < iters; i++) {
    int rc = launch_step(h, act, obs, rew, disc, st, stream, 0);
    if (rc) return rc;
  }
  if (hipEventRecord(h->ev1, s) != hipSuccess) return -2;
  if (hipEventSynchronize(h->ev1) != hipSuccess) return -2;
  float total = 0.f;
  if (hipEventElapsedTime(&total, h->ev0, h->ev1) != hipSuccess) return -2;
  *ms = total / (float)iters;
  return 0;
}

int ffe_time_kernel(ffe_handle h, const float *act, float *obs, float *rew, float *disc, int32_t *st, int iters, void *stream, float *ms) {
  if (!h || !ms || iters <= 0) return -1;
  DeviceGuard guard(h->device);
  FFE_BALL_DISPATCH(h, *ms = ffb::ball_time_kernel(h->ball, act, obs, rew, disc, st, iters, stream));
  h->timing = true; h->timing_ms = 0.0;
  int rc = 0;
  for (int i = 0; i < iters && !rc; i++) rc = launch_step(h, act, obs, rew, disc, st, stream, 0);
  h->timing = false;
  *ms = (float)(h->timing_ms / iters);
  return rc;
}

#ifdef FFE_STAMPS
int ffe_debug_read_stamps(unsigned long long *out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps), 20 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[20] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

#ifdef FFE_DBGCF
int ffe_debug_read_cf(float *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbgcf), sizeof(float) * 64 * (4 + 6 * 12)) == hipSuccess ? 0 : -1; }
#endif

#ifdef FFE_TRACE
// rows of {start clock, end clock, HW_ID, XCC_ID} per workgroup of the last flight launch (walk_on_ball: ffb_debug_read_trace)
int ffe_debug_read_trace(unsigned long long *out, int nrows) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), (size_t)nrows * 4 * sizeof(unsigned long long)) != hipSuccess) return -1;
  return 0;
}
#endif

int ffe_test_quat(int op, const float *a, const float *b, float *out, int n, void *stream) {
  if (!a || !b || !out || n <= 0) return -1;
  hipLaunchKernelGGL(test_quat_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), op, a, b, out, n);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // extern "C"
