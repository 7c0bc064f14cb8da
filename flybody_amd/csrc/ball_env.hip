// ball_env.hip - MI355X (gfx950) batched walk_on_ball environment (ref: fly_envs.py:125-157, tasks/walk_on_ball.py).
//
// ONE 64-lane wavefront per environment, one launch = one control step (10 physics substeps of 2e-4 s, dm_control's
// legacy mj_step2;mj_step1 order, buffered sensors, observation, reward, termination, auto-reset).  Lane mapping and
// the model tables are described in ball_model.hpp.  Per substep (mj: mj_step restated):
//   stage 1  kinematics, velocities / bias accelerations (root-to-leaf passes by pointer jumping), spatial inertias about the
//            fixed thorax origin, inertia-box drag, composite inertias and subtree forces (gathers over depth-first lane
//            ranges), joint-space inertia (582 entries across the lanes), block factorisation of M and M + hB together,
//            ball-capsule collision;
//   stage 2  filtered actuators (+ adhesion through the contact normals), smooth acceleration, joint-limit and
//            elliptic-cone contact rows, the constraint solve in the space of those rows (G = J M^-1 J' from block solves,
//            dense Newton with the exact cone Hessian in registers, lane = row), noslip sweeps on the same G,
//            touch / force sensors, implicit-in-damping Euler integration.
// Parity tests: tests/test_gpu_ball.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <vector>

#include "ball_env.hpp"
#include "launch_order.hpp"
#include "ball_model.hpp"
#include "dev_math.hpp"
#include "convex.hpp"

namespace ffb {
using namespace dm;

enum { BF_NO_FLUID = 1, BF_NO_LIMIT = 2, BF_NO_DAMPER = 4, BF_NO_SPRING = 8, BF_NO_GRAVITY = 16, BF_NO_ACTUATION = 32,
       BF_NO_CONTACT = 64, BF_NO_NOSLIP = 128, BF_NO_ADHESION = 256 };
// Keeps the fully unrolled per-entry loops from being interleaved into one huge basic block of loads: without it the
// scheduler hoists every entry's LDS reads to the top and the kernel needs > 500 VGPRs.
#define ENTRY_FENCE() __builtin_amdgcn_sched_barrier(0)
// Loop-invariant code motion otherwise precomputes every LDS address derived from the per-lane entry / slot words once per
// launch and keeps ~150 of them alive across the substep loop; an opaque copy forces the (cheap) address math to stay local.
__device__ __forceinline__ unsigned opq(unsigned x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ int opq(int x) { asm volatile("" : "+v"(x)); return x; }
#ifdef FFB_STAMPS
__device__ unsigned long long g_bstamps[24];
#define BSTAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); c.st_acc[k] += t_ - c.st_t0; c.st_t0 = t_; } while (0)
#else
#define BSTAMP(k) do { } while (0)
#endif
// Diagnostic build only (-DFFE_TRACE): start / end shader clock and hardware slot of every wave of the last launch (tools/wave_timeline.py)
#ifdef FFE_TRACE
__device__ unsigned long long g_btrace[32768][4];
#endif
constexpr int kMaxNewton = 12;
constexpr float kNewtonTol2 = 1e-8f;  // stop when |grad| <= 1e-4 |force scale| (M^-1 metric; 1e-7 costs 4x in parity for 1% speed)
constexpr int kLsIter = 10;
constexpr float kLsTol = 1e-2f;  // |phi'(alpha)| <= tol |phi'(0)|: an inexact line search, the Newton loop converges the rest

struct alignas(16) BState {
  float q[NDP], v[NDP], act[64];
  float ballq[4], ballw[4];
  float qacc_ws[NDP], wsc[NL][3];  // warm start of the constraint solver: last limit force per dof, last contact force per link
  int step_counter, needs_reset, overflow, iters, ncon, have_ws, nself, pad1;
  unsigned con_hist[2];  // active (inside includemargin) contacts of each of the control step's first 16 substeps, 4 bits each: parity tooling
  unsigned det_hist[2];  // detected (inside margin) contacts of its first 12 substeps, 5 bits each
  float sd_n[NSD][4];    // separating-direction cache of the convex pairs (convex_collide)
  unsigned short sd_pid[NSD];
  int sd_cnt, ws_n, pad2[2];
  float ws_f[NC];        // forces of the fly-fly contacts of the last substep, by pair id (warm start)
  unsigned short ws_pid[NC];
};

struct BTaskDev {
  int time_limit_steps, pad_first_obs, flags, canonical, clip;
};

struct alignas(16) BTile {
  // One scratch region, used by the phases of a substep in turn (7 488 B):
  //   stage 1:    X4 / dadd / Mq (inertia, factorisation), F then lk (assembly, tree passes)
  //   collision:  gc / gq / lp (geom frames, link poses) over the dead X4 .. Mq; primitive geom slots + the pair list in lk
  //   stage 2:    Q, V / X4 (block solves) | G over the dead Mq and lk | Newton: S over everything (G is in registers by then)
  union {
    struct {
      union {
        struct { float Q[NDP], V[NDP]; };  // joint state staged for the actuators / contact rows (start of stage 2) and the observation
        float4 X4[NDP];                     // right-hand sides / solutions of the block solves
      };
      float dadd[NDP];                      // h * damping: only read by the stage-1 factorisation
      union {
        struct {
          float Mq[NMMAX];                  // joint-space inertia: only read by the stage-1 factorisation
          union {
            float F[NDP][6];                // crb * cdof during the inertia assembly
            float lk[NL][12];               // link exchange of the tree passes: pose (7) | motion vector (6) | crb (10) | force (6)
            struct { float pg[NPG * 9]; unsigned short cl1[CL1], cl2[CL2], tc_pid[NSD]; float tc_n[NSD][4], cl2n[CL2][5]; };  // collision: primitive geom slots, pair lists, next cache
          };
        };
        float G[RMAX * (RMAX + 1) / 2];     // stage 2: G = J M^-1 J' over the constraint rows, lower triangle packed by rows (lane r owns row r)
      };
    };
    float S[RMAX * (RMAX + 1) / 2];         // Newton: the factor of I + L' G L, transposed through here (packed lower triangle)
    struct { float4 gc[NG], gq[NG]; float lp[NL][8]; };  // collision: geom centre | bounding radius, geom quaternion; link pose
  };
  float Lm[NMMAX], Lh[NMMAX];               // factors of M and of M + h B (Euler), made together in stage 1
  float dinv_m[NDP], dinv_h[NDP];
  float C[NDP][6];
  float frc[64];
  int c_link[NC], c_blk[NC], c_excl[NC], c_nch[NC], c_adh[NC];
  unsigned c_amask[NC], c_bmask[NBLK];  // c_bmask[b]: contacts whose chain lies in block b
  unsigned char c_chain[NC][16];
  float c_par[NC][5];  // K, B, invweight, friction, includemargin
  float c_pos[NC][3], c_frame[NC][9], c_dist[NC];
  float c_D[NC], c_mu[NC], c_aref[NC][3], c_f[NC][3], c_w[NC][3];
  // constraint rows: 3 per contact (normal, two tangents), then the instantiated joint limits
  float r_y0[RMAX], r_lam[RMAX], r_f[RMAX];
  float r_sgn[RMAX], r_D[RMAX];
  unsigned char r_blk[RMAX], r_col[RMAX], r_dof[RMAX], rowof[NBLK][KCOL];
  float sens[24];  // running sums of the buffered sensors: force 18, touch 6
  // separating directions of the convex pairs that reached the narrow phase, kept from substep to substep (convex_collide)
  float sd_n[NSD][4];  // direction | the capsule's axis parameter of a capsule - convex pair
  unsigned short sd_pid[NSD];
  int sd_cnt;
  // fly-fly contacts: pair id of each slot, and the pairs' forces of the last substep (warm start of their rows)
  unsigned short c_pid[NC], ws_pid[NC];
  float ws_f[NC];
  int ws_n;
};

// ------------------------------------------------------------------------------------------------ per-lane context
// Only what must survive between the stages lives here; model constants are re-read from the (L2-resident) tables
// where they are used, through a laundered pointer so the compiler does not hoist them back into registers.
struct Ctx {
  const BallModel *M;
  BTile *T;
  int lane, flags;
  unsigned lpack;         // parent + 1 | depth << 8 | ndof << 12
  int sdof[3];
  unsigned sbl[3];        // block | local index << 8 of each slot's dof
  float q[3], v[3], fnb[3];
  int xh;                 // this lane carries a haltere in slot 2 (closed-form single hinge, see ball_model.hpp)
  V3 xp, xip;
  Q4 xq;
  S6 cvel, caccb;         // link velocity and bias acceleration (gravity + velocity products) about the thorax origin
  float mass;
  Q4 bq;
  V3 bw, btau;
  int nc, nact;           // ball contacts detected / of those, the ones inside includemargin (they get constraint rows)
  int nsc;                // fly-fly contacts of this substep: slots nc .. nc + nsc - 1 of the contact arrays (see self_collide)
  int ndrop;              // detected fly-fly contacts without a slot (inactive, no adhesion actuator involved)
  float wsl[3], wsc[3];   // warm start of the constraint solver: last substep's limit force per slot, contact force of this link
  int have_ws, overflow;
#ifdef FFB_STAMPS
  unsigned long long st_t0, st_acc[24];
#endif
};

__device__ __forceinline__ bool slot_on(const Ctx &c, int s) { return c.sdof[s] >= 0; }
__device__ __forceinline__ int l_parent(const Ctx &c) { return (int)(c.lpack & 0xffu) - 1; }
__device__ __forceinline__ int l_ndof(const Ctx &c) { return (int)((c.lpack >> 12) & 0x3u); }
// The tables are read through an address-space-1 pointer: a generic pointer makes every table read a FLAT load, which
// counts against the LDS wait counter as well, so each LDS wait would also wait for the schedule prefetch.
typedef const BallModel FFE_GLOBAL *ModelPtr;
__device__ __forceinline__ const BallModel FFE_GLOBAL &model(const Ctx &c) {
  ModelPtr m = (ModelPtr)c.M;
  asm volatile("" : "+s"(m));
  return *m;
}

// ------------------------------------------------------------------------------------------------ block factorisation
// mj: mj_factorI on M's 12 independent blocks, all blocks in lock step (step s eliminates every block's s-th pivot from the
// leaf end).  The matrix lives in LDS, so any lane can apply any update: the host lays the ~2300 updates
//   L[e] -= L[ki] * L[kj] / L[kk]        (e = (i, j), i a proper ancestor of the pivot k; values stay unscaled until the end)
// out in step order as `nfs` slots of 64 independent updates (ball_model.hpp), read coalesced and one slot ahead.
// LDS operations of a wave complete in issue order, which is all the ordering the steps need.
__device__ __forceinline__ void factor2(Ctx &c) {
  // factorises M into T.Lm and M + diag(T.dadd) into T.Lh in one pass over the schedule (same elimination order, so the
  // schedule words, address arithmetic and control flow are shared)
  BTile &T = *c.T;
  const BallModel FFE_GLOBAL &M = model(c);
  const int lane = c.lane;
#pragma unroll
  for (int t = 0; t < ECAP; t++) {
    const unsigned ea = M.ent_a[t][lane];
    if (ea >> 31) {
      const unsigned i = ea & 0xffu, j = (ea >> 8) & 0xffu, adr = (ea >> 16) & 0x3ffu;
      const float vv = T.Mq[adr];
      T.Lm[adr] = vv;
      T.Lh[adr] = i == j ? vv + T.dadd[i] : vv;
    }
  }
  DM_SYNC();
  // Schedule words are fetched one group of four slots ahead (tables are zero padded past nfs).  No fence inside the loop:
  // the LDS accesses of consecutive slots may alias, so the compiler keeps their order, and the hardware executes a
  // wave's LDS operations in issue order.
  const int nfs = M.nfs;
  unsigned wa[4], wb[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { wa[q] = M.fac_a[q][lane]; wb[q] = M.fac_b[q][lane]; }
#pragma unroll 1
  for (int base = 0; base < nfs; base += 4) {
    unsigned ca[4], cb[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { ca[q] = wa[q]; cb[q] = wb[q]; }
#pragma unroll
    for (int q = 0; q < 4; q++) { wa[q] = M.fac_a[base + 4 + q][lane]; wb[q] = M.fac_b[base + 4 + q][lane]; }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const unsigned a = ca[q], b = cb[q];
      if (a >> 31) {
        const unsigned akk = (a >> 10) & 0x3ffu, aki = (a >> 20) & 0x3ffu, akj = b & 0x3ffu, ae = a & 0x3ffu;
        const float mkk = T.Lm[akk], mki = T.Lm[aki], mkj = T.Lm[akj], hkk = T.Lh[akk], hki = T.Lh[aki], hkj = T.Lh[akj];
        T.Lm[ae] -= mki * mkj * frcp(mkk);
        T.Lh[ae] -= hki * hkj * frcp(hkk);
      }
    }
  }
  DM_SYNC();
#pragma unroll
  for (int t = 0; t < ECAP; t++) {
    const unsigned ea = M.ent_a[t][lane];
    if ((ea >> 31) && (ea & 0xffu) == ((ea >> 8) & 0xffu)) {
      T.dinv_m[ea & 0xffu] = frcp(T.Lm[(ea >> 16) & 0x3ffu]);
      T.dinv_h[ea & 0xffu] = frcp(T.Lh[(ea >> 16) & 0x3ffu]);
    }
  }
  DM_SYNC();
#pragma unroll
  for (int t = 0; t < ECAP; t++) {
    const unsigned ea = M.ent_a[t][lane];
    if ((ea >> 31) && (ea & 0xffu) != ((ea >> 8) & 0xffu)) {
      T.Lm[(ea >> 16) & 0x3ffu] *= T.dinv_m[ea & 0xffu];
      T.Lh[(ea >> 16) & 0x3ffu] *= T.dinv_h[ea & 0xffu];
    }
  }
  DM_SYNC();
}

// mj: mj_solveLD on T.X4 (four right-hand sides at once): rows leaf -> root, D^-1, columns root -> leaf, from the two
// schedules p1 / p2 (slots of 64 independent updates in step order)
__device__ __forceinline__ void solve4(Ctx &c, const float *L, const float *dinv) {
  BTile &T = *c.T;
  const BallModel FFE_GLOBAL &M = model(c);
  const int lane = c.lane;
  {
    const int n = M.np1;
    unsigned wq[4];
#pragma unroll
    for (int q = 0; q < 4; q++) wq[q] = M.p1[q][lane];
#pragma unroll 1
    for (int base = 0; base < n; base += 4) {
      unsigned cw[4];
#pragma unroll
      for (int q = 0; q < 4; q++) cw[q] = wq[q];
#pragma unroll
      for (int q = 0; q < 4; q++) wq[q] = M.p1[base + 4 + q][lane];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const unsigned w = cw[q];
        if (w >> 31) {
          const unsigned i = (w >> 10) & 0x7fu, j = (w >> 17) & 0x7fu;
          const float l = L[w & 0x3ffu];
          const float4 xi = T.X4[i];
          float4 xj = T.X4[j];
          xj.x -= l * xi.x; xj.y -= l * xi.y; xj.z -= l * xi.z; xj.w -= l * xi.w;
          T.X4[j] = xj;
        }
      }
    }
    DM_SYNC();
  }
  for (int f = lane; f < ND; f += 64) {
    const float dv = dinv[f];
    float4 x = T.X4[f];
    x.x *= dv; x.y *= dv; x.z *= dv; x.w *= dv;
    T.X4[f] = x;
  }
  DM_SYNC();
  {
    const int n = M.np2;
    unsigned wq[4];
#pragma unroll
    for (int q = 0; q < 4; q++) wq[q] = M.p2[q][lane];
#pragma unroll 1
    for (int base = 0; base < n; base += 4) {
      unsigned cw[4];
#pragma unroll
      for (int q = 0; q < 4; q++) cw[q] = wq[q];
#pragma unroll
      for (int q = 0; q < 4; q++) wq[q] = M.p2[base + 4 + q][lane];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const unsigned w = cw[q];
        if (w >> 31) {
          const unsigned i = (w >> 10) & 0x7fu, j = (w >> 17) & 0x7fu;
          const float l = L[w & 0x3ffu];
          const float4 xj = T.X4[j];
          float4 xi = T.X4[i];
          xi.x -= l * xj.x; xi.y -= l * xj.y; xi.z -= l * xj.z; xi.w -= l * xj.w;
          T.X4[i] = xi;
        }
      }
    }
    DM_SYNC();
  }
}

// Two single-right-hand-side solves with two factors of the same structure in one pass over the schedules:
// T.X4[.].x <- (L_A D_A L_A')^-1 x, T.X4[.].y <- (L_B D_B L_B')^-1 y   (final acceleration with M, Euler with M + h B)
__device__ __forceinline__ void solve_dual(Ctx &c, const float *LA, const float *dinvA, const float *LB, const float *dinvB) {
  BTile &T = *c.T;
  const BallModel FFE_GLOBAL &M = model(c);
  const int lane = c.lane;
  {
    const int n = M.np1;
    unsigned wq[4];
#pragma unroll
    for (int q = 0; q < 4; q++) wq[q] = M.p1[q][lane];
#pragma unroll 1
    for (int base = 0; base < n; base += 4) {
      unsigned cw[4];
#pragma unroll
      for (int q = 0; q < 4; q++) cw[q] = wq[q];
#pragma unroll
      for (int q = 0; q < 4; q++) wq[q] = M.p1[base + 4 + q][lane];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const unsigned w = cw[q];
        if (w >> 31) {
          const unsigned i = (w >> 10) & 0x7fu, j = (w >> 17) & 0x7fu;
          const float la = LA[w & 0x3ffu], lb = LB[w & 0x3ffu];
          const float4 xi = T.X4[i];
          float4 xj = T.X4[j];
          xj.x -= la * xi.x; xj.y -= lb * xi.y;
          T.X4[j] = xj;
        }
      }
    }
    DM_SYNC();
  }
  for (int f = lane; f < ND; f += 64) {
    float4 x = T.X4[f];
    x.x *= dinvA[f]; x.y *= dinvB[f];
    T.X4[f] = x;
  }
  DM_SYNC();
  {
    const int n = M.np2;
    unsigned wq[4];
#pragma unroll
    for (int q = 0; q < 4; q++) wq[q] = M.p2[q][lane];
#pragma unroll 1
    for (int base = 0; base < n; base += 4) {
      unsigned cw[4];
#pragma unroll
      for (int q = 0; q < 4; q++) cw[q] = wq[q];
#pragma unroll
      for (int q = 0; q < 4; q++) wq[q] = M.p2[base + 4 + q][lane];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const unsigned w = cw[q];
        if (w >> 31) {
          const unsigned i = (w >> 10) & 0x7fu, j = (w >> 17) & 0x7fu;
          const float la = LA[w & 0x3ffu], lb = LB[w & 0x3ffu];
          const float4 xj = T.X4[j];
          float4 xi = T.X4[i];
          xi.x -= la * xj.x; xi.y -= lb * xj.y;
          T.X4[i] = xi;
        }
      }
    }
    DM_SYNC();
  }
}

// ------------------------------------------------------------------------------------------------ impedance
// mj: getimpedance (margin folded into `x` by the caller: x = |pos - margin|)
__device__ __forceinline__ float impedance(const float *si, float x) {
  float d0 = fminf(fmaxf(si[0], 1e-4f), 0.9999f), d1 = fminf(fmaxf(si[1], 1e-4f), 0.9999f);
  const float width = fmaxf(0.f, si[2]), mid = fminf(fmaxf(si[3], 1e-4f), 0.9999f), power = fmaxf(1.f, si[4]);
  if (d0 == d1 || width <= 1e-15f) return 0.5f * (d0 + d1);
  x = x * frcp(width);
  if (x >= 1.f) return d1;
  if (x <= 0.f) return d0;
  float y;
  if (power == 1.f) y = x;
  else if (power == 2.f) y = x <= mid ? x * x * frcp(mid) : 1.f - (1.f - x) * (1.f - x) * frcp(1.f - mid);
  else if (x <= mid) y = powf(x, power) / powf(mid, power - 1.f);
  else y = 1.f - powf(1.f - x, power) / powf(1.f - mid, power - 1.f);
  return d0 + y * (d1 - d0);
}

// ------------------------------------------------------------------------------------------------ fly-fly collision
// mj: mj_collision narrow phase for the fly's own sphere / capsule pairs: mjc_CapsuleCapsule (closest points of the two
// segments, one contact; a sphere is a capsule of zero length, for which the same formulas reduce to mjc_SphereCapsule /
// mjraw_SphereSphere).  Broad phase = MuJoCo's bounding-sphere test.  A detected contact (dist <= margin) takes the next
// free slot of the contact arrays after the ball contacts: c_link = link of geom1 | link of geom2 << 8, c_frame[0..2] =
// normal from geom1 to geom2, c_par = {K, B, invweight, 0, margin - gap}.  Returns the number of fly-fly contacts stored.
__device__ __noinline__ int self_collide(BTile *Tp, ModelPtr Mp, const int lane, const int nc) {
  BTile &T = *Tp;
  const BallModel FFE_GLOBAL &M = *Mp;
  // per slot: (centre, bounding radius), (axis, half length), radius
  const float4 *pgc = reinterpret_cast<const float4 *>(&T.lk[0][0]), *pga = pgc + NPG;
  const float *pgr = reinterpret_cast<const float *>(pgc + 2 * NPG);
  const float mclaw = M.sc_margin;
  const unsigned long long claws = M.sp_claw;
  const int sphere = M.sp_sphere;
  // lane s owns slot s and meets slots s + 1 .. s + NPG / 2 (mod NPG): every unordered pair once
  const bool own = lane < NPG;
  const unsigned long long partners = own ? M.sp_mask[lane] : 0ull;
  const float4 o0 = own ? pgc[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
  const float oreach = o0.w + (((claws >> lane) & 1ull) ? mclaw : 0.f);
  unsigned near = 0u;  // bit t - 1: the pair (lane, lane + t) passed the bounding-sphere test
#pragma unroll 4
  for (int t = 1; t <= NPG / 2; t++) {
    int j = lane + t;
    j = j >= NPG ? j - NPG : j;
    if (own && ((partners >> j) & 1ull) && (t < NPG / 2 || lane < NPG / 2)) {
      const float4 p0 = pgc[j];
      const float dx = p0.x - o0.x, dy = p0.y - o0.y, dz = p0.z - o0.z;
      const float reach = oreach + p0.w + mclaw;  // (the claw margin on both sides: a bound is all that is needed here)
      if (dx * dx + dy * dy + dz * dz <= reach * reach) near |= 1u << (t - 1);
    }
  }
  int nsc = 0, ovf = 0, ndrop = 0;
#pragma unroll 1
  while (__ballot(near != 0u) != 0ull) {
    bool hit = false;
    float dist = 0.f, margin = 0.f;
    V3 nrm = {1.f, 0.f, 0.f}, cpos = {0.f, 0.f, 0.f};
    int s1 = 0, s2 = 0;
    if (near) {
      const int t = __ffs(near);
      near &= near - 1u;
      int j = lane + t;
      j = j >= NPG ? j - NPG : j;
      // mj: geom1 is the one with the lower type code (the sphere), else the one that comes first in the model
      const bool swap = j == sphere || (lane != sphere && j < lane);
      s1 = swap ? j : lane; s2 = swap ? lane : j;
      const float4 ca = pgc[s1], aa = pga[s1], cb_ = pgc[s2], ab = pga[s2];
      const V3 p1 = {ca.x, ca.y, ca.z}, a1 = {aa.x, aa.y, aa.z}, p2 = {cb_.x, cb_.y, cb_.z}, a2 = {ab.x, ab.y, ab.z};
      const float l1 = aa.w, r1 = pgr[s1], l2 = ab.w, r2 = pgr[s2];
      margin = (((claws >> s1) | (claws >> s2)) & 1ull) ? mclaw : 0.f;
      const V3 dif = p1 - p2;
      const float mb = -dot(a1, a2), u = -dot(a1, dif), v = dot(a2, dif), det = 1.f - mb * mb;
      float x1, x2;
      if (fabsf(det) >= 1e-6f) {
        const float idet = 1.f / det;
        x1 = (u - mb * v) * idet; x2 = (v - mb * u) * idet;
        if (x1 > l1) { x1 = l1; x2 = v - mb * l1; } else if (x1 < -l1) { x1 = -l1; x2 = v + mb * l1; }
        if (x2 > l2) { x2 = l2; x1 = fminf(fmaxf(u - mb * l2, -l1), l1); }
        else if (x2 < -l2) { x2 = -l2; x1 = fminf(fmaxf(u + mb * l2, -l1), l1); }
      } else {  // parallel axes: centre of the overlapping stretch
        const float c2 = u, lo = fmaxf(-l1, c2 - l2), hi = fminf(l1, c2 + l2);
        x1 = lo <= hi ? 0.5f * (lo + hi) : (c2 > 0.f ? l1 : -l1);
        x2 = fminf(fmaxf((x1 - c2) * (mb < 0.f ? 1.f : -1.f), -l2), l2);
      }
      const V3 q1 = p1 + x1 * a1, q2 = p2 + x2 * a2, d12 = q2 - q1;
      const float cd = fsqrt(dot(d12, d12));
      hit = cd <= margin + r1 + r2;
      if (cd >= 1e-15f) nrm = frcp(cd) * d12;
      dist = cd - r1 - r2;
      cpos = q1 + (r1 + 0.5f * dist) * nrm;
    }
    // A contact inside its margin but outside margin - gap exerts no force; it only matters as one of the contacts an adhesion
    // actuator spreads its pull over (mjTRN_BODY).  Without such an actuator on either link it gets no slot (and no solver row);
    // it is still counted (ndrop) so that the number of detected contacts stays MuJoCo's.
    const bool dropped = hit && dist >= margin - (margin != 0.f ? M.sc_gap : 0.f) && M.l_adh[M.pgs_link[s1]] < 0 && M.l_adh[M.pgs_link[s2]] < 0;
    hit = hit && !dropped;
    ndrop += __popcll(__ballot(dropped));
    const unsigned long long bal = __ballot(hit);
    if (bal) {
      const int idx = nc + nsc + __popcll(bal & ((1ull << lane) - 1ull));
      auto store = [&](int at) {
        const float incl = margin - (margin != 0.f ? M.sc_gap : 0.f);
        T.c_link[at] = M.pgs_link[s1] | (M.pgs_link[s2] << 8);
        T.c_pid[at] = (unsigned short)(0x8000 | s1 | (s2 << 6));
        T.c_excl[at] = dist >= incl ? 1 : 0;
        T.c_dist[at] = dist;
        T.c_pos[at][0] = cpos.x; T.c_pos[at][1] = cpos.y; T.c_pos[at][2] = cpos.z;
        T.c_frame[at][0] = nrm.x; T.c_frame[at][1] = nrm.y; T.c_frame[at][2] = nrm.z;
        T.c_par[at][0] = M.sc_K; T.c_par[at][1] = M.sc_B; T.c_par[at][2] = M.pgs_invw[s1] + M.pgs_invw[s2]; T.c_par[at][3] = 0.f;
        T.c_par[at][4] = incl;
      };
      if (hit && idx < NC) store(idx);
      const int n = __popcll(bal);
      const bool over = nc + nsc + n > NC;
      nsc = min(nsc + n, NC - nc);
      if (over && nc < NC) {
        // More contacts than the tile holds: the env is flagged, and a hit without a slot takes the place of the shallowest fly-fly
        // contact held so far if it is deeper (as for the ball contacts and the convex pairs: a deep contact is never the one dropped).
        ovf = 1;
        DM_SYNC();
        for (unsigned long long left = __ballot(hit && idx >= NC); left; left &= left - 1) {
          const int j = __ffsll((long long)left) - 1;
          float shallow = -1e30f;
          int at = -1;
          for (int k = nc; k < NC; k++) { const float dk = T.c_dist[k]; if (dk > shallow) { shallow = dk; at = k; } }
          if (__shfl(dist, j) < shallow) { if (lane == j) store(at); DM_SYNC(); }
        }
      } else if (over) ovf = 1;
    }
  }
  return nsc | (ovf << 8) | (ndrop << 16);
}

// ------------------------------------------------------------------------------------------------ convex pairs
// mj: mjc_Convex for the pairs with an ellipsoid or a cylinder on one side (thorax, head, rostrum, labrum, wings, coxae, abdomen
// segments; fruitfly.xml:323-443), restated in csrc/convex.hpp.  The geoms' world frames are made once per substep from the link
// poses the lanes publish (`T.lp`) into `T.gc` (centre | bounding radius) / `T.gq` (orientation).
__device__ __forceinline__ cvx::Geom load_geom(const BTile &T, const BallModel FFE_GLOBAL &M, int g) {
  const float4 cc = T.gc[g], qq = T.gq[g];
  return cvx::Geom{V3{cc.x, cc.y, cc.z}, Q4{qq.x, qq.y, qq.z, qq.w}, M.cg_size[0][g], M.cg_size[1][g], M.cg_size[2][g], M.cg_type[g]};
}

// Ball (geom1, a sphere) against the ellipsoids / cylinders that can reach it (abdomen segments: mjc_SphereCylinder; labrum, wings:
// the sphere's centre against the ellipsoid's signed distance): condim-3 ball contacts like the leg capsules', appended to the
// `nc0` ball contacts stage 1 has just stored.  Also makes the geom frames for `convex_collide`.  Returns the number appended
// | overflow << 8.
#ifndef FFB_BALLCVX_ATTR
#define FFB_BALLCVX_ATTR __noinline__
#endif
__device__ FFB_BALLCVX_ATTR int ball_convex(BTile *Tp, ModelPtr Mp, const int lane, const int nc0) {
  BTile &T = *Tp;
  const BallModel FFE_GLOBAL &M = *Mp;
  const int ncg = M.ncg;
  for (int g = lane; g < ncg; g += 64) {
    const int l = M.cg_link[g];
    V3 gp = {M.cg_pos[0][g], M.cg_pos[1][g], M.cg_pos[2][g]};
    Q4 gq = {M.cg_quat[0][g], M.cg_quat[1][g], M.cg_quat[2][g], M.cg_quat[3][g]};
    if (l >= 0) {
      const float *lp = T.lp[l];
      const Q4 xq = {lp[3], lp[4], lp[5], lp[6]};
      gp = V3{lp[0], lp[1], lp[2]} + qrot(xq, gp);
      gq = qnormalize(qmul(xq, gq));
    }
    T.gc[g] = make_float4(gp.x, gp.y, gp.z, M.cg_brad[g]);
    T.gq[g] = make_float4(gq.w, gq.x, gq.y, gq.z);
  }
  DM_SYNC();
  bool hit = false;
  float dist = 0.f, incl = 0.f;
  V3 nrm = {1.f, 0.f, 0.f}, cpos = {0.f, 0.f, 0.f};
  const int nxb = M.nxb;
  int g = 0;
  if (lane < nxb) {
    g = M.xb_geom[lane];
    const cvx::Geom ball = {V3{M.b_center[0], M.b_center[1], M.b_center[2]}, Q4{1.f, 0.f, 0.f, 0.f}, M.b_radius, 0.f, 0.f, cvx::SPHERE};
    const float4 cc = T.gc[g];
    const float dx = cc.x - ball.c.x, dy = cc.y - ball.c.y, dz = cc.z - ball.c.z, reach = M.b_radius + cc.w + M.xb_margin[lane];
    if (dx * dx + dy * dy + dz * dz <= reach * reach) {  // mj: the bounding-sphere test of mj_collision
      const cvx::PResult r = cvx::prim_convex<1>(ball, load_geom(T, M, g));
      dist = r.dist; nrm = r.n; cpos = r.pos;
      hit = dist <= M.xb_margin[lane];
      incl = M.xb_margin[lane] - M.xb_gap[lane];
    }
  }
  const unsigned long long bal = __ballot(hit);
  const int n = __popcll(bal), idx = nc0 + __popcll(bal & ((1ull << lane) - 1ull));
  if (hit && idx < NC) {
    const int link = M.cg_link[g], nch = M.l_nchain[link], last = M.l_chain[nch - 1][link];
    T.c_link[idx] = link;
    T.c_blk[idx] = M.d_blk[last]; T.c_amask[idx] = M.d_amask[last];
    T.c_excl[idx] = dist >= incl ? 1 : 0;
    T.c_dist[idx] = dist;
    T.c_pos[idx][0] = cpos.x; T.c_pos[idx][1] = cpos.y; T.c_pos[idx][2] = cpos.z;
    T.c_nch[idx] = nch;
    for (int p = 0; p < NCH; p++) T.c_chain[idx][p] = (unsigned char)(p < nch ? M.l_chain[p][link] : 0);
    T.c_par[idx][0] = M.xb_K[lane]; T.c_par[idx][1] = M.xb_B[lane]; T.c_par[idx][2] = M.xb_invw[lane]; T.c_par[idx][3] = M.xb_fric[lane];
    T.c_par[idx][4] = incl;
    T.c_adh[idx] = M.l_adh[link];
    V3 t1 = (nrm.y < 0.5f && nrm.y > -0.5f) ? V3{0.f, 1.f, 0.f} : V3{0.f, 0.f, 1.f};  // mj: mju_makeFrame
    t1 = t1 - dot(nrm, t1) * nrm;
    t1 = frcp(fsqrt(dot(t1, t1))) * t1;
    const V3 t2 = cross(nrm, t1);
    float *fr = T.c_frame[idx];
    fr[0] = nrm.x; fr[1] = nrm.y; fr[2] = nrm.z; fr[3] = t1.x; fr[4] = t1.y; fr[5] = t1.z; fr[6] = t2.x; fr[7] = t2.y; fr[8] = t2.z;
  }
  DM_SYNC();
  return min(n, NC - nc0) | ((nc0 + n > NC ? 1 : 0) << 8);
}

// The fly's own convex pairs.  Broad phase: MuJoCo's bounding-sphere test over the static candidate list (flybody_amd/model/reach.py),
// then - on the survivors - a rigorous lower bound of the distance from one separating direction (cvx::separation_bound); a pair
// whose bound exceeds its margin cannot touch.  Narrow phase: one lane per remaining pair (cvx::collide).  A contact (dist <= margin)
// takes the next free slot after the `nprev` contacts already stored, in the same layout as self_collide's; a geom of the thorax
// (fixed to the world) has no link: the moving geom's link is stored first and the normal turned, so that it still points from the
// first link's geom to the second's.  Returns the number of contacts stored | overflow << 8.
#ifndef FFB_CONVEX_ATTR
#define FFB_CONVEX_ATTR __noinline__
#endif
__device__ FFB_CONVEX_ATTR int convex_collide(BTile *Tp, ModelPtr Mp, const int lane, const int nprev) {
  BTile &T = *Tp;
  const BallModel FFE_GLOBAL &M = *Mp;
  const unsigned long long mm0 = M.cg_mmask[0], mm1 = M.cg_mmask[1];
  const float mclaw = M.sc_margin;
  auto pair_margin = [&](int a, int b) {
    const bool ma = a < 64 ? ((mm0 >> a) & 1ull) : ((mm1 >> (a - 64)) & 1ull), mb = b < 64 ? ((mm0 >> b) & 1ull) : ((mm1 >> (b - 64)) & 1ull);
    return (ma || mb) ? mclaw : 0.f;
  };
  int n1 = 0, ovf = 0;
#ifdef CVXDBG_NO_SPHERE
  const int ncp = 0;
#else
  const int ncp = M.ncp;
#endif
#pragma unroll 2
  for (int base = 0; base < ncp; base += 64) {
    const unsigned w = M.cp_pair[base + lane];
    bool pass = false;
    if (w != 0xffffu) {
      const int a = w & 255, b = w >> 8;
      const float4 ca = T.gc[a], cb = T.gc[b];
      const float dx = cb.x - ca.x, dy = cb.y - ca.y, dz = cb.z - ca.z, reach = ca.w + cb.w + pair_margin(a, b);
      pass = dx * dx + dy * dy + dz * dz <= reach * reach;
    }
    const unsigned long long bal = __ballot(pass);
    const int idx = n1 + __popcll(bal & ((1ull << lane) - 1ull));
    if (pass && idx < CL1) T.cl1[idx] = (unsigned short)w;
    n1 += __popcll(bal);
  }
  if (n1 > CL1) { n1 = CL1; ovf = 1; }
#ifdef CVXDBG_NO_BOUND
  n1 = 0;
#endif
  DM_SYNC();
  // Second test: a separating direction proves the pair apart.  Besides the two directions of cvx::separation_bound, the one the
  // narrow phase found for this pair on the last substep it ran (`sd_*`): pairs that stay near each other without touching (a hind
  // coxa beside the abdomen, stacked abdomen discs) then cost two support evaluations per substep instead of a narrow phase.
  int n2 = 0, nk = 0;
  const int ncache = T.sd_cnt;
#pragma unroll 1
  for (int base = 0; base < n1; base += 64) {
    bool pass = false, keep = false, have_kn = false;
    unsigned w = 0u;
    V3 kn = {0.f, 0.f, 0.f};
    float kt = 0.f;
    if (base + lane < n1) {
      w = T.cl1[base + lane];
      const int a = w & 255, b = w >> 8;
      const cvx::Geom ga = load_geom(T, M, a), gb = load_geom(T, M, b);
      // the distance that matters: the margin where an adhesion actuator shares its pull over every detected contact of its body,
      // margin - gap (a contact beyond it exerts no force and gets no slot) otherwise
      float thr = pair_margin(a, b);
      {
        const int la_ = M.cg_link[a], lb_ = M.cg_link[b];
        if ((la_ < 0 || M.l_adh[la_] < 0) && (lb_ < 0 || M.l_adh[lb_] < 0) && thr != 0.f) thr -= M.sc_gap;
      }
      pass = cvx::separation_bound(ga, gb) <= thr;
      if (pass) {
        for (int k = 0; k < ncache; k++) {
          if (T.sd_pid[k] == w) {
            kn = V3{T.sd_n[k][0], T.sd_n[k][1], T.sd_n[k][2]};
            kt = T.sd_n[k][3];
            keep = -cvx::overlap(ga, gb, kn) > thr;
            have_kn = true;
          }
        }
        pass = !keep;
      }
    }
    const unsigned long long bal = __ballot(pass), balk = __ballot(keep);
    const int idx = n2 + __popcll(bal & ((1ull << lane) - 1ull)), idk = nk + __popcll(balk & ((1ull << lane) - 1ull));
    if (pass && idx < CL2) { T.cl2[idx] = (unsigned short)w; float *o = T.cl2n[idx]; o[0] = kn.x; o[1] = kn.y; o[2] = kn.z; o[3] = have_kn ? 1.f : 0.f; o[4] = kt; }
    if (keep && idk < NSD) { T.tc_pid[idk] = (unsigned short)w; T.tc_n[idk][0] = kn.x; T.tc_n[idk][1] = kn.y; T.tc_n[idk][2] = kn.z; T.tc_n[idk][3] = kt; }
    n2 += __popcll(bal);
    nk = min(nk + __popcll(balk), NSD);
  }
  if (n2 > CL2) { n2 = CL2; ovf = 1; }
#ifdef CVXDBG_NO_NARROW
  n2 = 0;
#endif
  DM_SYNC();
  bool hit = false, dropped = false;
  float dist = 0.f, margin = 0.f, ctt = 0.f;
  V3 nrm = {1.f, 0.f, 0.f}, cpos = {0.f, 0.f, 0.f};
  int a = 0, b = 0;
  if (lane < n2) {
    const unsigned w = T.cl2[lane];
    a = w & 255; b = w >> 8;
    margin = pair_margin(a, b);
    const float *kn = T.cl2n[lane];
    const cvx::Contact ct = cvx::collide(load_geom(T, M, a), load_geom(T, M, b), V3{kn[0], kn[1], kn[2]}, kn[3] != 0.f, kn[4]);
    dist = ct.dist; nrm = ct.n; cpos = ct.pos; ctt = ct.t;
    hit = dist <= margin;
    // (an inactive contact no adhesion actuator takes part in gets no slot: see self_collide)
    if (hit && dist >= margin - (margin != 0.f ? M.sc_gap : 0.f)) {
      const int la_ = M.cg_link[a], lb_ = M.cg_link[b];
      dropped = (la_ < 0 || M.l_adh[la_] < 0) && (lb_ < 0 || M.l_adh[lb_] < 0);
      hit = !dropped;
    }
    if (nk + lane < NSD) {  // its direction, for the next substep's second test
      T.tc_pid[nk + lane] = (unsigned short)w; T.tc_n[nk + lane][0] = nrm.x; T.tc_n[nk + lane][1] = nrm.y; T.tc_n[nk + lane][2] = nrm.z; T.tc_n[nk + lane][3] = ctt;
    }
  }
  nk = min(nk + n2, NSD);
  DM_SYNC();
  if (lane < nk) { T.sd_pid[lane] = T.tc_pid[lane]; T.sd_n[lane][0] = T.tc_n[lane][0]; T.sd_n[lane][1] = T.tc_n[lane][1]; T.sd_n[lane][2] = T.tc_n[lane][2]; T.sd_n[lane][3] = T.tc_n[lane][3]; }
  if (lane == 0) T.sd_cnt = nk;
  unsigned long long bal = __ballot(hit);
  if (nprev + __popcll(bal) > NC) {  // more than the free slots: the env is flagged and the deepest are kept (as for the ball contacts)
    ovf = 1;
    int rank = 0;
    for (unsigned long long m = bal; m; m &= m - 1) {
      const int j = __ffsll((long long)m) - 1;
      const float dj = __shfl(dist, j);
      if (dj < dist || (dj == dist && j < lane)) rank++;
    }
    hit = hit && rank < NC - nprev;
    bal = __ballot(hit);
  }
  const int n = __popcll(bal), idx = nprev + __popcll(bal & ((1ull << lane) - 1ull));
  if (hit && idx < NC) {
    int la = M.cg_link[a], lb = M.cg_link[b];
    if (la < 0) { la = lb; lb = 255; nrm = V3{-nrm.x, -nrm.y, -nrm.z}; }
    else if (lb < 0) lb = 255;
    const float incl = margin - (margin != 0.f ? M.sc_gap : 0.f);
    T.c_link[idx] = la | (lb << 8);
    T.c_pid[idx] = (unsigned short)(a | (b << 8));
    T.c_excl[idx] = dist >= incl ? 1 : 0;
    T.c_dist[idx] = dist;
    T.c_pos[idx][0] = cpos.x; T.c_pos[idx][1] = cpos.y; T.c_pos[idx][2] = cpos.z;
    T.c_frame[idx][0] = nrm.x; T.c_frame[idx][1] = nrm.y; T.c_frame[idx][2] = nrm.z;
    T.c_par[idx][0] = M.sc_K; T.c_par[idx][1] = M.sc_B; T.c_par[idx][2] = M.cg_invw[a] + M.cg_invw[b]; T.c_par[idx][3] = 0.f;
    T.c_par[idx][4] = incl;
  }
  DM_SYNC();
  return n | (ovf << 8) | (__popcll(__ballot(dropped)) << 16);
}

// Fly-fly contact slots keep the dofs of geom2's chain (14 bytes), and the block-local solve column of that chain (byte 14),
// in the slot's unused ball Jacobian.
__device__ __forceinline__ unsigned char *sc_chain_b(BTile &T, int k) { return reinterpret_cast<unsigned char *>(&T.c_frame[k][3]); }  // (a fly-fly slot uses row 0 of its frame only: 24 spare bytes)
__device__ __forceinline__ const unsigned char *sc_chain_b(const BTile &T, int k) { return reinterpret_cast<const unsigned char *>(&T.c_frame[k][3]); }

// The contact rows' Jacobian entries are evaluated where they are used, from the dof's motion axes (T.C) and the contact's point and
// frame.  As a table (16 contacts x 3 rows x 14 chain dofs, plus the ball's 3 x 3 per contact) they took 3.2 KB of the tile - the
// difference between 6 and 8 resident waves per CU, i.e. between three rounds of waves per launch and two (walk_on_ball +39 %).
// cj_u: velocity of contact k's point per unit rate of fly dof f (about the fixed thorax origin c0).
__device__ __forceinline__ V3 cj_u(const BTile &T, int k, int f, V3 c0) {
  const S6 cd = ld6(T.C[f]);
  const V3 r = V3{T.c_pos[k][0], T.c_pos[k][1], T.c_pos[k][2]} - c0;
  return lin(cd) + cross(ang(cd), r);
}
__device__ __forceinline__ float cj_row(const BTile &T, int k, int r, V3 u) { const float *fr = &T.c_frame[k][3 * r]; return fr[0] * u.x + fr[1] * u.y + fr[2] * u.z; }
// the ball's three entries of row r of ball contact k (geom1 = the ball, rates = its angular velocity in body axes; Rb = its rotation)
__device__ __forceinline__ V3 cj_ball(const BTile &T, int k, int r, const M3 &Rb, V3 bc) {
  const V3 rr = V3{T.c_pos[k][0], T.c_pos[k][1], T.c_pos[k][2]} - bc;
  const float *fr = &T.c_frame[k][3 * r];
  const V3 u0 = cross(V3{Rb.m0, Rb.m3, Rb.m6}, rr), u1 = cross(V3{Rb.m1, Rb.m4, Rb.m7}, rr), u2 = cross(V3{Rb.m2, Rb.m5, Rb.m8}, rr);
  return V3{-(fr[0] * u0.x + fr[1] * u0.y + fr[2] * u0.z), -(fr[0] * u1.x + fr[1] * u1.y + fr[2] * u1.z), -(fr[0] * u2.x + fr[1] * u2.y + fr[2] * u2.z)};
}

// mj: mj_makeConstraint rows of the fly-fly contacts (condim 1: one frictionless row each, J = n . (jacp2 - jacp1) at the
// contact point), mj_makeImpedance, mj_referenceConstraint; and mj_transmission mjTRN_BODY for the adhesion actuators: an
// adhesion force is spread evenly over ALL contacts of its body (ball and fly-fly, also the ones inside the gap), which
// rewrites the weights the ball-only path set.  Slot k = nc + j: c_J[k][0] / c_J[k][1] = the row over the chain dofs of
// geom1's / geom2's link, c_chain[k] / sc_chain_b = those dofs, c_nch / c_blk = counts / blocks (second << 8),
// c_amask = chain masks (second << 16), c_adh = adhesion actuators of the two links (255 = none).
__device__ __noinline__ void self_rows(BTile *Tp, ModelPtr Mp, const int lane, const int nc, const int nsc) {
  BTile &T = *Tp;
  const BallModel FFE_GLOBAL &M = *Mp;
  const V3 c0 = {M.thorax_pos[0], M.thorax_pos[1], M.thorax_pos[2]};
  for (int j = 0; j < nsc; j++) {
    const int k = nc + j, la = T.c_link[k] & 255, lb = T.c_link[k] >> 8;
    const int half = lane >> 4, p = lane & 15, link = half == 0 ? la : lb;
    const int nch = (half < 2 && link != 255) ? M.l_nchain[link] : 0;  // (link 255: a geom of the thorax, fixed to the world - no chain)
    float jv = 0.f, vv = 0.f;
    int f = 0;
    if (half < 2 && p < nch) {
      f = M.l_chain[p][link];
      const S6 cd = ld6(T.C[f]);
      const V3 r = V3{T.c_pos[k][0], T.c_pos[k][1], T.c_pos[k][2]} - c0;
      const V3 u = lin(cd) + cross(ang(cd), r);
      jv = (T.c_frame[k][0] * u.x + T.c_frame[k][1] * u.y + T.c_frame[k][2] * u.z) * (half == 0 ? -1.f : 1.f);
      vv = T.V[f];
    }
    if (half < 2 && p < NCH) {
      if (half == 0) T.c_chain[k][p] = (unsigned char)f; else sc_chain_b(T, k)[p] = (unsigned char)f;
    }
    const float vel = wave_sum(jv * vv);
    if (lane == 0) {
      const int na = M.l_nchain[la], nb = lb != 255 ? M.l_nchain[lb] : 0, fa = M.l_chain[na - 1][la], fb = lb != 255 ? M.l_chain[nb - 1][lb] : fa;
      T.c_nch[k] = na | (nb << 8);
      T.c_blk[k] = (int)M.d_blk[fa] | ((int)M.d_blk[fb] << 8);
      T.c_amask[k] = (unsigned)M.d_amask[fa] | ((lb != 255 ? (unsigned)M.d_amask[fb] : 0u) << 16);
      T.c_adh[k] = (M.l_adh[la] & 255) | (((lb != 255 ? M.l_adh[lb] : -1) & 255) << 8);
      const float K = T.c_par[k][0], B = T.c_par[k][1], invw = T.c_par[k][2], incl = T.c_par[k][4], dist = T.c_dist[k];
      float si_[5];
#pragma unroll
      for (int q2 = 0; q2 < 5; q2++) si_[q2] = M.sc_solimp[q2];
      const float imp = impedance(si_, fabsf(dist - incl));
      const float R0 = fmaxf(1e-15f, (1.f - imp) * invw * frcp(imp));
      T.c_D[k] = T.c_excl[k] ? 0.f : frcp(R0);
      T.c_mu[k] = 0.f;
      T.c_aref[k][0] = -B * vel - K * imp * (dist - incl);
      T.c_aref[k][1] = T.c_aref[k][2] = 0.f;
    }
  }
  DM_SYNC();
  // adhesion weights over all contacts of the claw's body
  if (lane < nc + nsc) {
    const int k = lane;
    auto ids = [&](int kk, int &a1, int &a2) {
      if (kk < nc) { a1 = T.c_adh[kk] >= 0 ? T.c_adh[kk] : 255; a2 = 255; }
      else { a1 = T.c_adh[kk] & 255; a2 = (T.c_adh[kk] >> 8) & 255; }
    };
    int a1, a2;
    ids(k, a1, a2);
    int n1 = 0, n2 = 0;
    for (int kk = 0; kk < nc + nsc; kk++) {
      int b1, b2;
      ids(kk, b1, b2);
      n1 += (a1 != 255 && (b1 == a1 || b2 == a1)) ? 1 : 0;
      n2 += (a2 != 255 && (b1 == a2 || b2 == a2)) ? 1 : 0;
    }
    float w = 0.f;
    if (a1 != 255) w -= T.frc[a1] / (float)n1;
    if (a2 != 255 && a2 != a1) w -= T.frc[a2] / (float)n2;
    T.c_w[k][0] = w; T.c_w[k][1] = 0.f; T.c_w[k][2] = 0.f;
  }
  DM_SYNC();
}

// J' of the fly-fly rows into the right-hand sides of a block solve: over geom1's chain, then (the chains may share dofs)
// accumulated over geom2's chain.  X4 has been zeroed and filled by the other rows; columns as assigned in stage 2.
__device__ __noinline__ void self_rhs(BTile *Tp, const int lane, const int nc, const int nsc, const int nrc, const int cb, const float c0x, const float c0y, const float c0z) {
  BTile &T = *Tp;
  const V3 c0 = {c0x, c0y, c0z};
#pragma unroll 1
  for (int half = 0; half < 2; half++) {
    DM_SYNC();
    for (int item = lane; item < nsc * 16; item += 64) {
      const int j = item >> 4, p = item & 15, k = nc + j;
      const int nch = (T.c_nch[k] >> (8 * half)) & 0xff;
      const int col = (half == 0 ? (int)T.r_col[nrc + j] : (int)sc_chain_b(T, k)[14]) + 1 - cb;
      if (p < nch && col >= 0 && col < 4) {
        const int f = half == 0 ? (int)T.c_chain[k][p] : (int)sc_chain_b(T, k)[p];
        (&T.X4[f].x)[col] += (half == 0 ? -1.f : 1.f) * cj_row(T, k, 0, cj_u(T, k, f, c0));
      }
    }
  }
}
// rows of G owned by fly-fly contacts: J over each of the two chains against the solved columns of that chain's block
__device__ __noinline__ void self_gacc(BTile *Tp, const int lane, const int nc, const int nsc, const int nrc, const int cb, const float c0x, const float c0y, const float c0z) {
  BTile &T = *Tp;
  const V3 c0 = {c0x, c0y, c0z};
  if (lane < nrc || lane >= nrc + nsc) return;
  const int k = nc + lane - nrc, gtri = lane * (lane + 1) / 2;
#pragma unroll 1
  for (int half = 0; half < 2; half++) {
    const int nch = (T.c_nch[k] >> (8 * half)) & 0xff, b = (T.c_blk[k] >> (8 * half)) & 0xff;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < nch; p++) {
      const int f = half == 0 ? (int)T.c_chain[k][p] : (int)sc_chain_b(T, k)[p];
      const float jv = (half == 0 ? -1.f : 1.f) * cj_row(T, k, 0, cj_u(T, k, f, c0));
      const float4 y = T.X4[f];
      acc.x += jv * y.x; acc.y += jv * y.y; acc.z += jv * y.z; acc.w += jv * y.w;
    }
#pragma unroll
    for (int q2 = 0; q2 < 4; q2++) {
      const int gc = cb + q2;
      const float av = q2 == 0 ? acc.x : (q2 == 1 ? acc.y : (q2 == 2 ? acc.z : acc.w));
      if (gc == 0) T.r_y0[lane] += av;  // J a_s
      else if (gc - 1 < KCOL) {
        const int r2 = T.rowof[b][gc - 1];
        if (r2 <= lane) T.G[gtri + r2] += av;  // (255 = no such row)
      }
    }
  }
}

// sum over the fly-fly contacts whose row touches fly dof (blk, li) of J[c][p] * w[c][0]
__device__ __forceinline__ float self_gather(const Ctx &c, unsigned sbl, int f, V3 c0, const float (*w)[3]) {
  const BTile &T = *c.T;
  const unsigned li = sbl >> 8, blk = sbl & 0xffu;
  float acc = 0.f;
  for (int k = c.nc; k < c.nc + c.nsc; k++) {
    const unsigned bl = (unsigned)T.c_blk[k], am = T.c_amask[k];
    const bool in1 = (bl & 0xffu) == blk && ((am >> li) & 1u), in2 = (bl >> 8) == blk && ((am >> (16 + li)) & 1u);
    if (in1 || in2) {  // (dof f lies on geom1's chain: -n . u, on geom2's: +n . u, on both: they cancel)
      const float jn = cj_row(T, k, 0, cj_u(T, k, f, c0));
      acc += ((in2 ? jn : 0.f) - (in1 ? jn : 0.f)) * w[k][0];
    }
  }
  return acc;
}

// ------------------------------------------------------------------------------------------------ stage 1
__device__ __forceinline__ void stage1(Ctx &c) {
  BTile &T = *c.T;
  const BallModel FFE_GLOBAL &M = model(c);
  const int lane = c.lane, parent = l_parent(c), ndof = l_ndof(c);
  const V3 c0 = {M.thorax_pos[0], M.thorax_pos[1], M.thorax_pos[2]};
  const V3 pos = {M.l_pos[0][lane], M.l_pos[1][lane], M.l_pos[2][lane]};
  const Q4 quat = {M.l_quat[0][lane], M.l_quat[1][lane], M.l_quat[2][lane], M.l_quat[3][lane]};
  V3 axis[3];
#pragma unroll
  for (int s = 0; s < 3; s++) axis[s] = {M.s_axis[0][s][lane], M.s_axis[1][s][lane], M.s_axis[2][s][lane]};
  // Every joint of this model sits at its body's origin (checked on the host), so a link's origin does not depend on
  // its own joint angles and everything that does not involve the parent is done once, before the tree pass:
  // the link's orientation relative to its parent after 0, 1, 2, 3 of its joints and the joint axes in the parent frame.
  Q4 qrel = quat;
  V3 axp[3];
#pragma unroll
  for (int s = 0; s < 3; s++) {
    axp[s] = qrot(qrel, axis[s]);
    if (s < ndof) {
      float sn, cs;
      fsincos(0.5f * c.q[s], &sn, &cs);
      qrel = qmul(qrel, Q4{cs, axis[s].x * sn, axis[s].y * sn, axis[s].z * sn});
    }
  }
  // ---- mj: mj_kinematics by pointer jumping instead of one sweep per tree level: after round r the pose (xp, xq) is
  //      relative to the frame above the link's 2^(r+1)-th ancestor; three rounds cover the deepest chain (8 links) with every
  //      lane at work in every round (a level sweep runs its body once per level with one level's lanes active).
  const unsigned tree = M.l_tree[lane];
  const int sub = (int)(tree & 0xffu), anc2 = (int)((tree >> 8) & 0xffu) - 1, anc4 = (int)((tree >> 16) & 0xffu) - 1;
  V3 axw[3], anc[3];
  {
    V3 xp = pos;
    Q4 xq = qrel;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int a = r == 0 ? parent : (r == 1 ? anc2 : anc4);
      float *o = T.lk[lane];
      o[0] = xp.x; o[1] = xp.y; o[2] = xp.z; o[3] = xq.w; o[4] = xq.x; o[5] = xq.y; o[6] = xq.z;
      DM_SYNC();
      if (a >= 0) {
        const float *p = T.lk[a];
        const V3 pp = {p[0], p[1], p[2]};
        const Q4 pq = {p[3], p[4], p[5], p[6]};
        xp = pp + mv(q2m(pq), xp);
        xq = qmul(pq, xq);
      }
      DM_SYNC();
    }
    xq = qnormalize(xq);
    c.xp = xp; c.xq = xq;
    float *o = T.lk[lane];
    o[3] = xq.w; o[4] = xq.x; o[5] = xq.y; o[6] = xq.z;
    DM_SYNC();
    Q4 pq = {1.f, 0.f, 0.f, 0.f};
    if (parent >= 0) { const float *p = T.lk[parent]; pq = {p[3], p[4], p[5], p[6]}; }
    const M3 Rp = q2m(pq);
#pragma unroll
    for (int s = 0; s < 3; s++) { axw[s] = mv(Rp, axp[s]); anc[s] = xp; }
    DM_SYNC();
  }
  BSTAMP(0);  // kinematics
  const M3 xmat = q2m(c.xq);
  c.xip = c.xp + mv(xmat, V3{M.l_ipos[0][lane], M.l_ipos[1][lane], M.l_ipos[2][lane]});
  const M3 ximat = q2m(qmul(c.xq, Q4{M.l_iquat[0][lane], M.l_iquat[1][lane], M.l_iquat[2][lane], M.l_iquat[3][lane]}));
  c.mass = M.l_mass[lane];
  // ---- mj: mj_comPos with the fixed thorax origin as the reference point
  const I10 cinert = inert_com(V3{M.l_inertia[0][lane], M.l_inertia[1][lane], M.l_inertia[2][lane]}, ximat, c.xip - c0, c.mass);
  S6 cdof[3];
#pragma unroll
  for (int s = 0; s < 3; s++) cdof[s] = s < ndof ? mk6(axw[s], cross(axw[s], c0 - anc[s])) : zero6();
  // ---- mj: mj_comVel + the acceleration half of mj_rne.  All motion vectors refer to the fixed thorax origin, so a link's
  //      velocity is the plain sum of v * cdof over its ancestor path: path sums by pointer jumping (published inclusive,
  //      the exclusive one - the parent's velocity - kept privately), the velocity products locally, then the same path
  //      sum for the bias accelerations.
  {
    S6 dv = zero6();
#pragma unroll
    for (int s = 0; s < 3; s++) if (s < ndof) dv = dv + c.v[s] * cdof[s];
    S6 sv = dv, pv = zero6();
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int a = r == 0 ? parent : (r == 1 ? anc2 : anc4);
      st6(T.lk[lane], sv);
      DM_SYNC();
      if (a >= 0) { const S6 t = ld6(T.lk[a]); sv = sv + t; pv = pv + t; }
      DM_SYNC();
    }
    S6 da = zero6();
#pragma unroll
    for (int s = 0; s < 3; s++) {
      if (s < ndof) {
        const S6 cdd = cross_motion(pv, cdof[s]);
        pv = pv + c.v[s] * cdof[s];
        da = da + c.v[s] * cdd;
      }
    }
    c.cvel = pv;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int a = r == 0 ? parent : (r == 1 ? anc2 : anc4);
      st6(T.lk[lane], da);
      DM_SYNC();
      if (a >= 0) da = da + ld6(T.lk[a]);
      DM_SYNC();
    }
    da.l2 += (c.flags & BF_NO_GRAVITY) ? 0.f : -M.gz;
    c.caccb = da;
  }
  BSTAMP(1);  // velocities + bias accelerations
  // ---- body forces: rigid-body bias (mj_rne) minus inertia-box drag (mj_inertiaBoxFluidModel), about c0
  S6 ftot;
  {
    const S6 t1 = mul_inert(cinert, c.caccb), t2 = mul_inert(cinert, c.cvel);
    ftot = t1 + cross_force(c.cvel, t2);
    if (!(c.flags & BF_NO_FLUID)) {
      float fl[8];
#pragma unroll
      for (int k = 0; k < 8; k++) fl[k] = M.l_fl[k][lane];
      const V3 r = c.xip - c0;
      const V3 wl = mtv(ximat, ang(c.cvel)), vl = mtv(ximat, lin(c.cvel) + cross(ang(c.cvel), r));
      const V3 Tl = {-fl[0] * wl.x - fl[5] * fabsf(wl.x) * wl.x, -fl[0] * wl.y - fl[6] * fabsf(wl.y) * wl.y, -fl[0] * wl.z - fl[7] * fabsf(wl.z) * wl.z};
      const V3 Fl = {-fl[1] * vl.x - fl[2] * fabsf(vl.x) * vl.x, -fl[1] * vl.y - fl[3] * fabsf(vl.y) * vl.y, -fl[1] * vl.z - fl[4] * fabsf(vl.z) * vl.z};
      const V3 Tw = mv(ximat, Tl), Fw = mv(ximat, Fl);
      ftot = ftot - mk6(Tw + cross(r, Fw), Fw);
    }
  }
  // ---- subtree sums (mj_crb's composite inertia, then mj_rne's backward pass), leaves first
  //      Links are numbered depth first, so the subtree of link l is lanes l .. l + sub - 1: every lane gathers its own
  //      range from one publication of the per-link values (no level order, no barrier inside the loop).
  I10 crb = cinert;
  const int maxsub = M.maxsub;
  st10(T.lk[lane], cinert);
  DM_SYNC();
#pragma unroll 1
  for (int t = 1; t < maxsub; t++) if (t < sub) crb = add10(crb, ld10(T.lk[lane + t]));
  DM_SYNC();
  st6(T.lk[lane], ftot);
  DM_SYNC();
#pragma unroll 1
  for (int t = 1; t < maxsub; t++) if (t < sub) ftot = ftot + ld6(T.lk[lane + t]);
  DM_SYNC();
  BSTAMP(2);  // body forces + subtree sums
  // ---- smooth joint forces without actuation: springs, dampers, -(bias - drag)
#pragma unroll
  for (int s = 0; s < 3; s++) {
    float f = 0.f;
    if (s < ndof) {
      if (!(c.flags & BF_NO_SPRING)) f -= M.s_stiff[s][lane] * (c.q[s] - M.s_sref[s][lane]);
      if (!(c.flags & BF_NO_DAMPER)) f -= M.s_damp[s][lane] * c.v[s];
      f -= dot6(cdof[s], ftot);
    }
    c.fnb[s] = f;
  }
  if (c.xh) {  // halteres: closed form (see ball_model.hpp)
    float sn, cs;
    fsincos(c.q[2], &sn, &cs);
    float f = 0.f;
    if (!(c.flags & BF_NO_SPRING)) f -= M.s_stiff[2][lane] * (c.q[2] - M.s_sref[2][lane]);
    if (!(c.flags & BF_NO_DAMPER)) f -= M.s_damp[2][lane] * c.v[2];
    if (!(c.flags & BF_NO_GRAVITY)) f += M.x_Gc[lane] * cs + M.x_Gs[lane] * sn;
    if (!(c.flags & BF_NO_FLUID)) f -= M.x_cv[lane] * c.v[2] + M.x_cq[lane] * fabsf(c.v[2]) * c.v[2];
    c.fnb[2] = f;
  }
  // ball: isotropic sphere about its centre, only the box drag acts (mj_inertiaBoxFluidModel in the inertial frame)
  {
    V3 tau = {0.f, 0.f, 0.f};
    if (!(c.flags & BF_NO_FLUID)) {
      const M3 Ri = q2m(Q4{M.b_iquat[0], M.b_iquat[1], M.b_iquat[2], M.b_iquat[3]});
      const V3 wl = mtv(Ri, c.bw);
      const V3 Tl = {-M.b_fl[0] * wl.x - M.b_fl[5] * fabsf(wl.x) * wl.x, -M.b_fl[0] * wl.y - M.b_fl[6] * fabsf(wl.y) * wl.y, -M.b_fl[0] * wl.z - M.b_fl[7] * fabsf(wl.z) * wl.z};
      tau = mv(Ri, Tl);
    }
    c.btau = tau;
  }
  // ---- mj: mj_crb joint-space inertia, one entry per (lane, slot t)
#pragma unroll
  for (int s = 0; s < 3; s++) {
    if (s < ndof) { st6(T.F[opq(c.sdof[s])], mul_inert(crb, cdof[s])); st6(T.C[opq(c.sdof[s])], cdof[s]); }
  }
  if (c.xh) { st6(T.F[c.sdof[2]], S6{M.x_M[lane], 0.f, 0.f, 0.f, 0.f, 0.f}); st6(T.C[c.sdof[2]], S6{1.f, 0.f, 0.f, 0.f, 0.f, 0.f}); }
#pragma unroll
  for (int s = 0; s < 3; s++) if (slot_on(c, s)) T.dadd[opq(c.sdof[s])] = (c.flags & BF_NO_DAMPER) ? 0.f : M.h * M.s_damp[s][lane];
  DM_SYNC();
#pragma unroll
  for (int t = 0; t < ECAP; t++) {
    const unsigned ea = M.ent_a[t][lane];
    if (ea >> 31) {
      const unsigned i = ea & 0xffu, j = (ea >> 8) & 0xffu;
      float mij = dot6(ld6(T.C[j]), ld6(T.F[i]));
      if (i == j) mij += M.d_arm[i];
      T.Mq[(ea >> 16) & 0x3ffu] = mij;
    }
  }
  DM_SYNC();
  BSTAMP(3);  // joint forces + inertia assembly
  factor2(c);
  BSTAMP(4);  // factor M and M + h B
  // ---- mj: mj_collision, ball (geom1, sphere) against this link's capsule: mjc_SphereCapsule
  bool hit = false;
  float dist = 0.f, margin = 0.f, gap = 0.f;
  V3 nrm = {1.f, 0.f, 0.f}, cpos = {0.f, 0.f, 0.f};
  if (M.g_has[lane] && !(c.flags & BF_NO_CONTACT)) {
    const V3 bc = {M.b_center[0], M.b_center[1], M.b_center[2]};
    const V3 gp = c.xp + mv(xmat, V3{M.g_pos[0][lane], M.g_pos[1][lane], M.g_pos[2][lane]});
    const V3 ax = mv(xmat, V3{M.g_axis[0][lane], M.g_axis[1][lane], M.g_axis[2][lane]});
    const float half = M.g_half[lane], rad = M.g_rad[lane], x = fminf(fmaxf(dot(ax, bc - gp), -half), half);
    {  // publish the geom for the fly-fly pair tests below (the link-exchange area is free between the factorisation and stage 2)
      // two float4 arrays (centre | radius, axis | half length), slot-major: consecutive lanes read consecutive 16-byte words
      float4 *o = reinterpret_cast<float4 *>(&T.lk[0][0]);
      const int sl = M.g_slot[lane];
      o[sl] = make_float4(gp.x, gp.y, gp.z, half + rad);  // bounding radius: all the broad phase reads of a partner
      o[NPG + sl] = make_float4(ax.x, ax.y, ax.z, half);
      reinterpret_cast<float *>(o + 2 * NPG)[sl] = rad;
    }
    margin = M.g_margin[lane]; gap = M.g_gap[lane];
    const V3 dif = gp + x * ax - bc;
    const float cd = fsqrt(dot(dif, dif));
    dist = cd - M.b_radius - rad;
    hit = cd <= margin + M.b_radius + rad;
    if (cd >= 1e-15f) nrm = frcp(cd) * dif;
    cpos = bc + (M.b_radius + 0.5f * dist) * nrm;
  }
  BSTAMP(5);  // collision: the ball against the leg capsules
  unsigned long long bal = __ballot(hit);
  if (__popcll(bal) > NC) {
    // more contacts than the tile holds: the env is flagged and the NC DEEPEST are kept (dropping a deep one lets the leg sink in
    // until a later substep answers with an enormous force: tools/soak.py saw such envs blow up at saturated actions)
    c.overflow |= 1;
    int rank = 0;
    for (unsigned long long m = bal; m; m &= m - 1) {
      const int j = __ffsll((long long)m) - 1;
      const float dj = __shfl(dist, j);
      if (dj < dist || (dj == dist && j < lane)) rank++;
    }
    hit = hit && rank < NC;
    bal = __ballot(hit);
  }
  const int idx = __popcll(bal & ((1ull << lane) - 1ull));
  c.nc = min(__popcll(bal), NC);
  c.nact = __popcll(__ballot(hit && idx < NC && !(dist >= margin - gap)));
  if (hit && idx < NC) {
    T.c_link[idx] = lane;
    const int last = ndof == 1 ? c.sdof[0] : (ndof == 2 ? c.sdof[1] : c.sdof[2]);
    T.c_blk[idx] = M.d_blk[last]; T.c_amask[idx] = M.d_amask[last];
    T.c_excl[idx] = dist >= margin - gap ? 1 : 0;
    T.c_dist[idx] = dist;
    T.c_pos[idx][0] = cpos.x; T.c_pos[idx][1] = cpos.y; T.c_pos[idx][2] = cpos.z;
    const int nch = M.l_nchain[lane];
    T.c_nch[idx] = nch;
    for (int p = 0; p < NCH; p++) T.c_chain[idx][p] = (unsigned char)(p < nch ? M.l_chain[p][lane] : 0);
    T.c_par[idx][0] = M.g_K[lane]; T.c_par[idx][1] = M.g_B[lane]; T.c_par[idx][2] = M.g_invw[lane]; T.c_par[idx][3] = M.g_fric[lane];
    T.c_par[idx][4] = margin - gap;
    T.c_adh[idx] = M.l_adh[lane];
    // mj: mju_makeFrame
    V3 t1 = (nrm.y < 0.5f && nrm.y > -0.5f) ? V3{0.f, 1.f, 0.f} : V3{0.f, 0.f, 1.f};
    t1 = t1 - dot(nrm, t1) * nrm;
    t1 = frcp(fsqrt(dot(t1, t1))) * t1;
    const V3 t2 = cross(nrm, t1);
    float *fr = T.c_frame[idx];
    fr[0] = nrm.x; fr[1] = nrm.y; fr[2] = nrm.z; fr[3] = t1.x; fr[4] = t1.y; fr[5] = t1.z; fr[6] = t2.x; fr[7] = t2.y; fr[8] = t2.z;
  }
  DM_SYNC();
  // ---- the ball against the ellipsoids / cylinders that can reach it (more condim-3 ball contacts), and the geom frames of the
  //      convex pairs below, from the link poses published here
  if (!(c.flags & BF_NO_CONTACT)) {
    float *o = T.lp[lane];
    o[0] = c.xp.x; o[1] = c.xp.y; o[2] = c.xp.z; o[3] = c.xq.w; o[4] = c.xq.x; o[5] = c.xq.y; o[6] = c.xq.z;
    DM_SYNC();
    const int xb = __builtin_amdgcn_readfirstlane(ball_convex(c.T, (ModelPtr)c.M, lane, c.nc));
    if (xb >> 8) c.overflow |= 1;
    for (int k = c.nc; k < c.nc + (xb & 0xff); k++) c.nact += T.c_excl[k] ? 0 : 1;
    c.nc += xb & 0xff;
  }
  BSTAMP(10);  // geom frames + the ball against its convex partners
  if (lane < NBLK) {
    unsigned bm = 0u;
    for (int k = 0; k < c.nc; k++) if (T.c_blk[k] == lane) bm |= 1u << k;
    T.c_bmask[lane] = bm;
  }
  // ---- mj: mj_collision over the fly's own sphere / capsule pairs (condim 1).  Every link publishes its primitive geoms
  //      (world centre, axis, half length, radius) into the link-exchange area, which is free between the factorisation and
  //      stage 2; the pair tests read them back from there.
  c.nsc = 0; c.ndrop = 0;
  if (!(c.flags & BF_NO_CONTACT)) {
    if (lane == M.pg2_lane) {  // the rostrum's second capsule
      const V3 gc = c.xp + mv(xmat, V3{M.pg2_pos[0], M.pg2_pos[1], M.pg2_pos[2]});
      const V3 ga = mv(xmat, V3{M.pg2_axis[0], M.pg2_axis[1], M.pg2_axis[2]});
      float4 *o = reinterpret_cast<float4 *>(&T.lk[0][0]);
      o[M.pg2_slot] = make_float4(gc.x, gc.y, gc.z, M.pg2_half + M.pg2_rad);
      o[NPG + M.pg2_slot] = make_float4(ga.x, ga.y, ga.z, M.pg2_half);
      reinterpret_cast<float *>(o + 2 * NPG)[M.pg2_slot] = M.pg2_rad;
    }
    DM_SYNC();
#ifdef FFB_SC_NOCALL
    const int sc = 0;
#else
    const int sc = __builtin_amdgcn_readfirstlane(self_collide(c.T, (ModelPtr)c.M, lane, c.nc));  // wave-uniform by construction
#endif
    c.nsc = sc & 0xff;
    if ((sc >> 8) & 0xff) c.overflow |= 1;
    c.ndrop = sc >> 16;
    BSTAMP(11);  // sphere / capsule pairs
    // ---- and over its pairs with an ellipsoid or cylinder on one side
    const int cc = __builtin_amdgcn_readfirstlane(convex_collide(c.T, (ModelPtr)c.M, lane, c.nc + c.nsc));
    c.nsc += cc & 0xff;
    if ((cc >> 8) & 0xff) c.overflow |= 1;
    c.ndrop += cc >> 16;
    for (int k = c.nc; k < c.nc + c.nsc; k++) c.nact += T.c_excl[k] ? 0 : 1;
  }
  DM_SYNC();
  BSTAMP(12);  // convex pairs
}

// sum over the contacts whose chain contains fly dof (blk, li) of  sum_r J[c][r][p] * w[c][r]
__device__ __forceinline__ float contact_gather(const Ctx &c, unsigned sbl, int f, V3 c0, const float (*w)[3]) {
  const BTile &T = *c.T;
  const unsigned li = sbl >> 8;
  unsigned bm = T.c_bmask[sbl & 0xffu];
  float acc = 0.f;
  while (bm) {
    const int k = __ffs(bm) - 1;
    bm &= bm - 1u;
    const unsigned am = T.c_amask[k];
    if ((am >> li) & 1u) {
      const V3 u = cj_u(T, k, f, c0);
      acc += cj_row(T, k, 0, u) * w[k][0] + cj_row(T, k, 1, u) * w[k][1] + cj_row(T, k, 2, u) * w[k][2];
    }
  }
  return acc;
}

// mj: PrimalUpdateConstraint for one elliptic condim-3 contact.  With U = mu (jar_n, jar_t1, jar_t2), N = U0, T = |U_t|:
// zero force in the top zone N >= mu T, fully quadratic in the bottom zone mu N + T <= 0, and the cone-surface cost
// Dm/2 (N - mu T)^2, Dm = D / (mu^2 (1 + mu^2)), in between.  Hc (optional) is the 3x3 Hessian of that cost in jar.
__device__ __forceinline__ void cone_force(float D, float mu, float j0, float j1, float j2, float &f0, float &f1, float &f2, float *Hc) {
  const float N = j0 * mu, U1 = j1 * mu, U2 = j2 * mu, Tn = fsqrt(U1 * U1 + U2 * U2);
  if (Hc) for (int k = 0; k < 9; k++) Hc[k] = 0.f;
  if (N >= mu * Tn || (Tn <= 0.f && N >= 0.f)) { f0 = f1 = f2 = 0.f; return; }
  if (mu * N + Tn <= 0.f || (Tn <= 0.f && N < 0.f)) {
    f0 = -D * j0; f1 = -D * j1; f2 = -D * j2;
    if (Hc) { Hc[0] = D; Hc[4] = D; Hc[8] = D; }
    return;
  }
  const float Dm = D * frcp(fmaxf(1e-15f, mu * mu * (1.f + mu * mu))), NT = N - mu * Tn, iT = frcp(Tn);
  f0 = -Dm * NT * mu;
  f1 = -f0 * iT * U1 * mu;
  f2 = -f0 * iT * U2 * mu;
  if (Hc) {
    const float a = mu * N * iT * iT * iT, bd = mu * mu - mu * N * iT, s2 = Dm * mu * mu;
    Hc[0] = s2;
    Hc[1] = Hc[3] = s2 * (-mu * U1 * iT);
    Hc[2] = Hc[6] = s2 * (-mu * U2 * iT);
    Hc[4] = s2 * (a * U1 * U1 + bd);
    Hc[8] = s2 * (a * U2 * U2 + bd);
    Hc[5] = Hc[7] = s2 * (a * U1 * U2);
  }
}

// mj: mju_QCQP2 with equal friction coefficients d: min 1/2 x'Ax + x'b, |x| <= d r (root finding differs, see below)
__device__ __forceinline__ bool qcqp2(float &x0, float &x1, float A00, float A01, float A11, float b0, float b1, float d, float r) {
  const float B0 = b0 * d, B1 = b1 * d, P00 = A00 * d * d, P01 = A01 * d * d, P11 = A11 * d * d, r2 = r * r;
  float la = 0.f, v0 = 0.f, v1 = 0.f;
  bool active = false;
  for (int it = 0; it < 20; it++) {
    const float det = (P00 + la) * (P11 + la) - P01 * P01;
    if (det < 1e-10f) { x0 = x1 = 0.f; return false; }
    const float idet = frcp(det), i00 = (P11 + la) * idet, i11 = (P00 + la) * idet, i01 = -P01 * idet;
    v0 = -i00 * B0 - i01 * B1; v1 = -i01 * B0 - i11 * B1;
    const float q = v0 * v0 + v1 * v1, val = q - r2;
    if (val < 1e-10f) break;
    const float deriv = -2.f * (i00 * v0 * v0 + i11 * v1 * v1 + 2.f * i01 * v0 * v1);
    // Newton on 1/r - 1/|v(la)| (nearly linear in la) instead of mju_QCQP2's Newton on |v|^2 - r^2: same root, same stopping
    // tests, monotone from the left as well, 3.4 instead of 5.5 iterations on the sliding contacts of this task
    const float delta = -2.f * q * (fsqrt(q) - r) * frcp(r * deriv);
    if (delta < 1e-10f) break;
    la += delta;
    active = true;
  }
  x0 = v0 * d; x1 = v1 * d;
  return active;
}

// Dense Newton of the constraint solve on R <= RB rows (RB = 16 / 24 / 32 picked per substep): lane = row, everything in
// registers - the lane's row of G and of S = I + L' G L, other lanes' scalars by v_readlane (uniform index), neighbours'
// by bpermute shifts.  Loops are fully unrolled over RB so that the register arrays keep static indices.
// A real call (not inlined): the iteration loop then has the whole register file to itself - inlined, the allocator
// kept the caller's ~90 live values resident and spilled 120-200 of the loop's own values per iteration instead.
template <int RB>
__device__ __noinline__ int dense_newton(BTile *Tp, ModelPtr Mp, const int lane, const int R, const int nrc, const float scale2, const int nslip,
                                         const int have_ws) {
  BTile &T = *Tp;
  int iters = 0;
    // The row arrays are kept as float pairs so that the broadcast-multiply-accumulate loops (products with G, the Cholesky
    // updates) issue as v_pk_fma_f32 with the two v_readlane results as an SGPR pair: 3 instructions per 2 columns, not 4.
    f2 Gp[RB / 2], Sp[RB / 2];
    auto G_ = [&](int j) -> float { return (j & 1) ? Gp[j >> 1].y : Gp[j >> 1].x; };
    auto S_ = [&](int j) -> float { return (j & 1) ? Sp[j >> 1].y : Sp[j >> 1].x; };
    auto setS = [&](int j, float v) { if (j & 1) Sp[j >> 1].y = v; else Sp[j >> 1].x = v; };
    auto gload = [&](int j) -> float { return (lane < R && j < R) ? T.G[j <= lane ? lane * (lane + 1) / 2 + j : j * (j + 1) / 2 + lane] : 0.f; };
#pragma unroll
    for (int m = 0; m < RB / 2; m++) Gp[m] = f2{gload(2 * m), gload(2 * m + 1)};
    float lam = lane < R ? T.r_lam[lane] : 0.f;
    const float y0v = lane < R ? T.r_y0[lane] : 0.f;
    const bool crow = lane < nrc;           // contact row (else limit row or idle lane)
    const int sub = crow ? lane % 3 : 0;    // position inside the contact's 3-row block
    auto gdot = [&](float vreg) {           // (G v)[lane], v given as one value per lane
      f2 acc = f2{0.f, 0.f};
#pragma unroll
      for (int m = 0; m < RB / 2; m++) acc = __builtin_elementwise_fma(Gp[m], f2{rl_f(vreg, 2 * m), rl_f(vreg, 2 * m + 1)}, acc);  // G = 0 beyond the R live rows
      return acc.x + acc.y;
    };
    float fv = 0.f, yv = 0.f;
    float L0 = 0.f, L1 = 0.f, L2 = 0.f;  // this row of the block-lower Cholesky factor of W: L[row][first .. first+2]
    auto eval = [&](float y, bool want_L) {
      // a contact's three rows are evaluated on all three lanes (the residuals come by shifts inside the block)
      // (by DPP wave shifts: no LDS round trip in the line search's inner evaluation)
      const float u1 = wshl1(y), u2 = wshl1(u1), d1 = wshr1(y), d2 = wshr1(d1);
      const float ya = sub == 0 ? y : (sub == 1 ? d1 : d2), yb = sub == 0 ? u1 : (sub == 1 ? y : d1), yc = sub == 0 ? u2 : (sub == 1 ? u1 : y);
      float f = 0.f;
      if (want_L) L0 = L1 = L2 = 0.f;
      if (crow) {
        const int k = lane / 3;
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, Hc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (!T.c_excl[k]) cone_force(T.c_D[k], T.c_mu[k], ya, yb, yc, f0, f1, f2, want_L ? Hc : nullptr);
        f = sub == 0 ? f0 : (sub == 1 ? f1 : f2);
        if (want_L) {  // Hc = L L' (positive semi-definite: a vanishing pivot zeroes its column)
          const float l00 = Hc[0] > 1e-30f ? fsqrt(Hc[0]) : 0.f, i00 = l00 > 0.f ? frcp(l00) : 0.f;
          const float l10 = Hc[3] * i00, l20 = Hc[6] * i00;
          const float d1 = Hc[4] - l10 * l10, l11 = d1 > 1e-7f * Hc[4] ? fsqrt(d1) : 0.f, i11 = l11 > 0.f ? frcp(l11) : 0.f;
          const float l21 = (Hc[7] - l20 * l10) * i11;
          const float d2 = Hc[8] - l20 * l20 - l21 * l21, l22 = d2 > 1e-7f * Hc[8] ? fsqrt(d2) : 0.f;
          if (sub == 0) { L0 = l00; } else if (sub == 1) { L0 = l10; L1 = l11; } else { L0 = l20; L1 = l21; L2 = l22; }
        }
      } else if (lane < R) {
        const float D = T.r_D[lane];
        f = y < 0.f ? -D * y : 0.f;
        if (want_L) L0 = y < 0.f ? fsqrt(D) : 0.f;
      }
      return f;
    };
    bool fresh = false;  // yv, fv belong to the current lam
    yv = y0v + gdot(lam);
#pragma unroll 1
    for (int it = 0; it < kMaxNewton; it++) {
      fresh = true;
      fv = eval(yv, true);
      const float ev = lam - fv;
      const float pv = gdot(ev);
      const float gn2 = wave_sum(lane < R ? pv * ev : 0.f);
      if ((it > 0 || have_ws) && gn2 <= kNewtonTol2 * scale2 + 1e-30f) break;
      iters++;
      // S = I + L' G L.  Column j: t_j = (G L)[lane][j] from this lane's G row and L's column j (read from the owning lanes),
      // then S[i][j] = delta_ij + sum over the rows a >= i of i's block of L[a][i] t_j(a) (neighbour lanes, shifted in).
      // wshl1(x): lane i takes lane i + 1's x (DPP wave shift: no LDS round trip, unlike a bpermute)
      const bool has1 = crow && sub < 2, has2 = crow && sub == 0;
      // (shifts are whole-wave operations: taken unconditionally, selected afterwards)
      const float Ls1 = wshl1(sub == 1 ? L0 : (sub == 2 ? L1 : 0.f)), Ls2 = wshl1(wshl1(L0));
      const float La1 = has1 ? Ls1 : 0.f;  // L[lane+1][lane] (the source lane picks its entry one column left of its diagonal)
      const float La2 = has2 ? Ls2 : 0.f;  // L[lane+2][lane] (only for sub == 0)
      const float Ld = sub == 0 ? L0 : (sub == 1 ? L1 : L2);                         // L[lane][lane]
#pragma unroll
      for (int j = 0; j < RB; j++) {
        if (j < R) {
        float tj;
        if (j < nrc) {
          const int fj = j - j % 3, sj = j % 3;
          // L[fj + b][j] for b = sj .. 2: lane fj+b holds it at position sj
          tj = G_(j) * rl_f(sj == 0 ? L0 : (sj == 1 ? L1 : L2), j);
          if (sj < 2) tj += G_(fj + sj + 1 < RB ? fj + sj + 1 : 0) * rl_f(sj == 0 ? L0 : L1, fj + sj + 1 < RB ? fj + sj + 1 : 0);
          if (sj < 1) tj += G_(fj + 2 < RB ? fj + 2 : 0) * rl_f(L0, fj + 2 < RB ? fj + 2 : 0);
        } else tj = G_(j) * rl_f(L0, j);
        const float t1 = wshl1(tj), t2 = wshl1(t1);
        setS(j, (j == lane ? 1.f : 0.f) + Ld * tj + La1 * t1 + La2 * t2);
        } else setS(j, j == lane ? 1.f : 0.f);
      }
      // rhs = L' p
      float w;
      {
        const float p1 = wshl1(pv), p2 = wshl1(p1);
        w = Ld * pv + La1 * p1 + La2 * p2;
      }
      // Cholesky of S (row per lane) as S = Lt D Lt', Lt unit lower triangular: Sr[k] ends as Lt[lane][k] below the diagonal and 0
      // on and above it, so that the substitutions are one unconditional FMA per step; ipp = 1 / D[lane]
      float ipp = 1.f;
#pragma unroll
      for (int k = 0; k < RB; k++) {
        if (k < R) {
          const float ip = __builtin_amdgcn_rsqf(rl_f(S_(k), k));
          const float lik = S_(k) * ip;
          if ((k & 1) == 0) Sp[k >> 1].y -= lik * rl_f(lik, k + 1);  // the pair partner of an even pivot column
          const f2 nl = f2{-lik, -lik};
#pragma unroll
          for (int m = (k >> 1) + 1; m < RB / 2; m++)  // rows / columns beyond R are identity: no-ops
            Sp[m] = __builtin_elementwise_fma(nl, f2{rl_f(lik, 2 * m), rl_f(lik, 2 * m + 1)}, Sp[m]);
          setS(k, lane > k ? lik * ip : 0.f);
          ipp = lane == k ? ip * ip : ipp;
        }
      }
      // forward substitution Lt z = w
#pragma unroll
      for (int k = 0; k < RB; k++)
        if (k < R) w -= S_(k) * rl_f(w, k);
      w *= ipp;
      // transpose through LDS (this lane's column of Lt), then backward substitution Lt' u = D^-1 z
      // (only the strict lower triangle is non-zero: packed by rows, row i at i (i - 1) / 2)
      if (lane < R) {
        const int tri = lane * (lane - 1) / 2;
#pragma unroll
        for (int j = 0; j < RB; j++) if (j < lane) T.S[tri + j] = S_(j);
      }
      DM_SYNC();
#pragma unroll
      for (int m = 0; m < RB / 2; m++) Sp[m] = f2{0.f, 0.f};
      if (lane < R) {
#pragma unroll
        for (int k = 0; k < RB; k++) if (k < R && k > lane) setS(k, T.S[k * (k - 1) / 2 + lane]);
      }
      DM_SYNC();
#pragma unroll
      for (int k = RB - 1; k >= 0; k--)
        if (k < R) w -= S_(k) * rl_f(w, k);
      // d = -e + L u, jd = G d
      float dl;
      {
        const float u1 = wshr1(w), u2 = wshr1(u1);
        dl = -ev + Ld * w + (crow && sub >= 1 ? (sub == 1 ? L0 : L1) : 0.f) * u1 + (crow && sub == 2 ? L0 : 0.f) * u2;
      }
      if (lane >= R) dl = 0.f;
      const float jdv = gdot(dl);
      const float c1s = wave_sum(lane < R ? dl * jdv : 0.f);
      const float d0 = wave_sum(lane < R ? ev * jdv : 0.f);  // phi'(0) = (lambda - f(y)) . G d: the forces at y are already known
      // exact line search on the convex phi(alpha): root of phi'(alpha) = phi'(0) + alpha c1 - sum_rows (f(y + alpha jd) - f(y)) jd
      auto dphi = [&](float al) {
        const float fa = eval(yv + al * jdv, false);
        float acc = lane < R ? (fv - fa) * jdv : 0.f;
        return d0 + al * c1s + wave_sum(acc);
      };
      float alpha = 0.f;
      {
        if (!(d0 < 0.f)) break;  // not a descent direction any more: converged to rounding
        float lo = 0.f, hi = 1.f, dlo = d0, dhi = dphi(1.f);
        int guard = 0;
        while (dhi < 0.f && fabsf(dhi) > kLsTol * fabsf(d0) && guard++ < 8) { lo = hi; dlo = dhi; hi *= 2.f; dhi = dphi(hi); }
        if (dhi < 0.f || fabsf(dhi) <= kLsTol * fabsf(d0)) alpha = hi;  // full (or doubled) Newton step: |phi'| already small
        else {
#pragma unroll 1
          for (int ls = 0; ls < kLsIter; ls++) {
            float mid = lo - dlo * (hi - lo) / (dhi - dlo);
            if (!(mid > lo + 0.05f * (hi - lo)) || !(mid < hi - 0.05f * (hi - lo))) mid = 0.5f * (lo + hi);
            const float dm_ = dphi(mid);
            if (dm_ < 0.f) { lo = mid; dlo = dm_; } else { hi = mid; dhi = dm_; }
            if (fabsf(dm_) <= kLsTol * fabsf(d0) || hi - lo <= 1e-6f * hi) break;
          }
          alpha = (dhi - dlo) != 0.f ? lo - dlo * (hi - lo) / (dhi - dlo) : hi;
          if (!(alpha >= lo) || !(alpha <= hi)) alpha = 0.5f * (lo + hi);
        }
      }
      lam += alpha * dl;
      yv += alpha * jdv;  // y = y0 + G lambda stays current without another product
      fresh = false;
    }
    // forces at the solution
    if (!fresh) fv = eval(yv, false);
    // ---- mj: mj_solNoSlip (ref: fruitfly.xml:4 noslip_iterations="3"): Gauss-Seidel on the tangential rows with the
    //      unregularised A = J M^-1 J' - which is G - normal and limit forces held fixed; res = G f + (J a_s - aref).
    //      G is symmetric, so row r of G against f is a wave sum over the lanes' column-r entries.
    if (nslip > 0) {
      const BallModel FFE_GLOBAL &M = *Mp;
      const float scale = 1.f / (M.meaninertia * 105.f);
      const int nc = nrc / 3;
      float res = y0v + gdot(fv);  // residual row per lane, kept current as the tangential forces move (G is symmetric)
      for (int iter = 0; iter < nslip; iter++) {
        float improvement = 0.f;
#pragma unroll
        for (int k = 0; k < NC; k++) {
          if (3 * k + 2 < RB && k < nc && !T.c_excl[k]) {
            const int r0 = 3 * k + 1, r1 = 3 * k + 2;
            const float res0 = rl_f(res, r0), res1 = rl_f(res, r1);
            const float o0 = rl_f(fv, r0), o1 = rl_f(fv, r1), fn = rl_f(fv, 3 * k);
            const float A00 = rl_f(G_(r0), r0), A01 = rl_f(G_(r1), r0), A11 = rl_f(G_(r1), r1);
            float v0 = 0.f, v1 = 0.f;
            if (fn >= 1e-15f) {
              const float b0 = res0 - A00 * o0 - A01 * o1, b1 = res1 - A01 * o0 - A11 * o1, mu = T.c_mu[k];
              const bool active = qcqp2(v0, v1, A00, A01, A11, b0, b1, mu, fn);
              if (active) {
                const float ssum = (v0 * v0 + v1 * v1) * frcp(mu * mu);
                const float sc2 = fn * __builtin_amdgcn_rsqf(fmaxf(1e-15f, ssum));
                v0 *= sc2; v1 *= sc2;
              }
            }
            float d0 = v0 - o0, d1 = v1 - o1;
            // mj: costChange reverts an update whose cost change is > 1e-10 (a failed QCQP).  In float32 the two terms below
            // cancel to ~1e-7 of their size, so the threshold is taken relative to them; a genuine failure is far above it.
            const float lin_ = d0 * res0 + d1 * res1, quad_ = 0.5f * (d0 * (A00 * d0 + A01 * d1) + d1 * (A01 * d0 + A11 * d1));
            float change = lin_ + quad_;
            if (change > 1e-10f + 1e-4f * (fabsf(lin_) + fabsf(quad_))) { v0 = o0; v1 = o1; d0 = d1 = 0.f; change = 0.f; }
            improvement -= fminf(change, 0.f);
            fv = lane == r0 ? v0 : (lane == r1 ? v1 : fv);
            res += G_(r0) * d0 + G_(r1) * d1;
          }
        }
        if (improvement * scale < 1e-6f) break;
      }
    }
    if (lane < R) { T.r_lam[lane] = lam; T.r_f[lane] = fv; }
    DM_SYNC();
  return iters;
}

// ------------------------------------------------------------------------------------------------ stage 2
__device__ __forceinline__ void stage2(Ctx &c, float act_reg, float ctrl_reg, float &act_out, bool integrate, int &iters_out, float *qacc_norm2) {
  BTile &T = *c.T;
  const BallModel FFE_GLOBAL &M = model(c);
  const int lane = c.lane;
  const int nc = c.nc;
  const float h = M.h;
  const V3 c0 = {M.thorax_pos[0], M.thorax_pos[1], M.thorax_pos[2]};
  const V3 bc = {M.b_center[0], M.b_center[1], M.b_center[2]};
#pragma unroll
  for (int s = 0; s < 3; s++) if (slot_on(c, s)) { T.Q[opq(c.sdof[s])] = c.q[s]; T.V[opq(c.sdof[s])] = c.v[s]; }
  DM_SYNC();
  // ---- mj: mj_fwdActuation: first-order activation filter, affine position servo on the activation
  float act_dot = 0.f;
  if (lane < NU) {
    float force = 0.f;
    if (!(c.flags & BF_NO_ACTUATION)) {
      float ctrl = ctrl_reg;
      if (M.a_climited[lane]) ctrl = fminf(fmaxf(ctrl, M.a_clo[lane]), M.a_chi[lane]);
      act_dot = (ctrl - act_reg) * frcp(M.a_tau[lane]);
      float length = 0.f, vel = 0.f;
      const int nw = M.a_nwrap[lane];
      for (int w = 0; w < nw; w++) { const int f = M.a_wdof[w][lane]; const float cf = M.a_wcoef[w][lane]; length += cf * T.Q[f]; vel += cf * T.V[f]; }
      force = M.a_gain[lane] * act_reg + M.a_b0[lane] + M.a_b1[lane] * length + M.a_b2[lane] * vel;
      if (M.a_flimited[lane]) force = fminf(fmaxf(force, M.a_flo[lane]), M.a_fhi[lane]);
      if (M.a_trn[lane] == 2 && (c.flags & BF_NO_ADHESION)) force = 0.f;
    }
    T.frc[lane] = force;
  }
  act_out = act_reg + h * act_dot;
  // ---- contact rows: Jacobians over the chain dofs (jac2 - jac1, geom1 = ball), impedance, reference acceleration
  const M3 Rb = q2m(c.bq);
  DM_SYNC();
  DM_SYNC();
  // per-contact parameters and the adhesion pull (mj: mj_transmission mjTRN_BODY: -force along the normal row).
  // Row-parallel: a row of 16 lanes handles one contact, lane p of the row the p-th dof of its chain.
  for (int base = 0; base < nc; base += 4) {
    const int k = base + (lane >> 4), p = lane & 15;
    const bool on = k < nc;
    const bool pv = on && p < T.c_nch[k];
    const float vv = pv ? T.V[T.c_chain[k][p]] : 0.f;
    const V3 uj = pv ? cj_u(T, k, T.c_chain[k][p], c0) : V3{0.f, 0.f, 0.f};
    float vel[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      vel[r] = row_sum(pv ? cj_row(T, k, r, uj) * vv : 0.f);
      if (on) { const V3 jb = cj_ball(T, k, r, Rb, bc); vel[r] += jb.x * c.bw.x + jb.y * c.bw.y + jb.z * c.bw.z; }
    }
    if (on && p == 0) {
      const float K = T.c_par[k][0], B = T.c_par[k][1], invw = T.c_par[k][2], incl = T.c_par[k][4], dist = T.c_dist[k];
      float si_[5];
#pragma unroll
      for (int q2 = 0; q2 < 5; q2++) si_[q2] = M.c_solimp[q2];
      const float imp = impedance(si_, fabsf(dist - incl));
      const float R0 = fmaxf(1e-15f, (1.f - imp) * invw * frcp(imp));
      T.c_D[k] = T.c_excl[k] ? 0.f : frcp(R0);
      T.c_mu[k] = T.c_par[k][3];
      T.c_aref[k][0] = -B * vel[0] - K * imp * (dist - incl);
      T.c_aref[k][1] = -B * vel[1];
      T.c_aref[k][2] = -B * vel[2];
      const int adh = T.c_adh[k];
      T.c_w[k][0] = adh >= 0 ? -T.frc[adh] : 0.f; T.c_w[k][1] = 0.f; T.c_w[k][2] = 0.f;
    }
  }
  DM_SYNC();
#ifdef FFB_SC_NOSTAGE2
  const int nsc = 0;
#else
  const int nsc = c.nsc;
#endif
  if (nsc) self_rows(c.T, (ModelPtr)c.M, lane, nc, nsc);  // fly-fly contacts (rare): rows, impedance, adhesion shares
  BSTAMP(6);  // actuation + contact rows
  // ---- smooth forces (mj: mj_fwdAcceleration); the smooth acceleration a_s = M^-1 qfrc_smooth rides as column 0 of the
  //      first block solve of the G build below
  float qs[3], am[3];
#pragma unroll
  for (int s = 0; s < 3; s++) {
    qs[s] = 0.f; am[s] = 0.f;
    if (slot_on(c, s)) {
      float f = c.fnb[s];
      const int a0 = M.s_act[0][s][lane], a1 = M.s_act[1][s][lane];
      if (a0 >= 0) f += M.s_actcoef[0][s][lane] * T.frc[a0];
      if (a1 >= 0) f += M.s_actcoef[1][s][lane] * T.frc[a1];
      if (nc) f += contact_gather(c, opq(c.sbl[s]), opq(c.sdof[s]), c0, T.c_w);
      if (nsc) f += self_gather(c, opq(c.sbl[s]), opq(c.sdof[s]), c0, T.c_w);
      qs[s] = f;
    }
  }
  V3 qsb = c.btau;
  for (int k = 0; k < nc; k++) {
    const V3 jb = cj_ball(T, k, 0, Rb, bc);
    qsb.x += jb.x * T.c_w[k][0]; qsb.y += jb.y * T.c_w[k][0]; qsb.z += jb.z * T.c_w[k][0];
  }
  const float Ib = M.b_I;
  const V3 amb = frcp(Ib) * qsb;
  BSTAMP(7);  // smooth forces
  // ---- joint-limit rows (mj: mj_instantiateLimit, margin 0): sign, D, aref per slot
  float lsgn[3], lD[3], laref[3];
#pragma unroll
  for (int s = 0; s < 3; s++) {
    lsgn[s] = 0.f; lD[s] = 0.f; laref[s] = 0.f;
    if (slot_on(c, s) && M.s_limited[s][lane] && !(c.flags & BF_NO_LIMIT)) {
      const float dlo = c.q[s] - M.s_lo[s][lane], dhi = M.s_hi[s][lane] - c.q[s];
      float dist = 0.f;
      if (dlo < 0.f) { lsgn[s] = 1.f; dist = dlo; }
      else if (dhi < 0.f) { lsgn[s] = -1.f; dist = dhi; }
      if (lsgn[s] != 0.f) {
        float si_[5];
#pragma unroll
        for (int q2 = 0; q2 < 5; q2++) si_[q2] = M.j_solimp[q2];
        const float imp = impedance(si_, fabsf(dist));
        lD[s] = frcp(fmaxf(1e-15f, (1.f - imp) * M.s_invw[s][lane] * frcp(imp)));
        laref[s] = -M.s_B[s][lane] * (lsgn[s] * c.v[s]) - M.s_K[s][lane] * imp * dist;
      }
    }
  }
  // ---- mj: mj_fwdConstraint in the space of the constraint rows.  With Y = M^-1 J', G = J Y and a = a_s + Y lambda the
  //      primal cost 1/2 (a - a_s)' M (a - a_s) + s(J a - aref) becomes 1/2 lambda' G lambda + s(y0 + G lambda), y0 = J a_s - aref;
  //      its Newton step is  d = -(I + W G)^-1 (lambda - f),  W = d2s/dy2 (3x3 per contact, scalar per limit).  M is factorised
  //      once per substep (stage 1); the rows' columns of G come four per block solve, rows of different blocks sharing a
  //      column because M^-1 is block diagonal; every Newton iteration is then dense work on <= 32 rows.
  int lrow[3] = {-1, -1, -1};
  const int nrc = 3 * nc;
  int R = nrc;
  if (nsc) {  // fly-fly contact rows: one unilateral row each (condim 1), handled by the solver like a joint-limit row
    if (lane < nsc) {
      const int row = nrc + lane, k = nc + lane;
      T.r_sgn[row] = 0.f; T.r_D[row] = T.c_D[k]; T.r_dof[row] = (unsigned char)k; T.r_blk[row] = (unsigned char)(T.c_blk[k] & 0xff);
      T.r_y0[row] = -T.c_aref[k][0];
      float l0 = 0.f;  // the same pair's force of the last substep
      if (c.have_ws) { const int nw = T.ws_n; const unsigned pid = T.c_pid[k]; for (int j = 0; j < nw; j++) if (T.ws_pid[j] == pid) l0 = T.ws_f[j]; }
      T.r_lam[row] = l0;
    }
    R += nsc;
  }
  {  // limit rows: one per instantiated limit, in (slot, lane) order; y0 starts as -aref, J a_s is added by the first solve
#pragma unroll
    for (int s = 0; s < 3; s++) {
      const bool on = lsgn[s] != 0.f;
      const unsigned long long bal = __ballot(on);
      const int idx = R + __popcll(bal & ((1ull << lane) - 1ull));
      if (on && idx < RMAX) {
        lrow[s] = idx;
        T.r_sgn[idx] = lsgn[s]; T.r_D[idx] = lD[s]; T.r_dof[idx] = (unsigned char)opq(c.sdof[s]); T.r_blk[idx] = (unsigned char)(opq(c.sbl[s]) & 0xffu);
        T.r_y0[idx] = -laref[s];
        T.r_lam[idx] = c.have_ws ? c.wsl[s] : 0.f;
      }
      R += __popcll(bal);
    }
    if (R > RMAX) { R = RMAX; c.overflow |= 2; }
  }
  const bool constrained = R > 0;
  int iters = 0;
  if (lane < nrc) {  // contact rows: the ball's share of J a_s minus aref; the fly's share comes with the first solve
    const int k = lane / 3, r = lane - 3 * k;
    T.r_blk[lane] = (unsigned char)T.c_blk[k];
    const V3 jb = cj_ball(T, k, r, Rb, bc);
    T.r_y0[lane] = jb.x * amb.x + jb.y * amb.y + jb.z * amb.z - T.c_aref[k][r];
  }
  for (int k = lane; k < NBLK * KCOL; k += 64) (&T.rowof[0][0])[k] = 255;
  // warm start from the force this link's contact carried last substep
  if (c.have_ws) { for (int k = 0; k < nc; k++) if (T.c_link[k] == lane) { T.r_lam[3 * k] = c.wsc[0]; T.r_lam[3 * k + 1] = c.wsc[1]; T.r_lam[3 * k + 2] = c.wsc[2]; } }
  else if (lane < nrc) T.r_lam[lane] = 0.f;
  DM_SYNC();
  // columns: a row's column is its rank among the rows of its block
  int ncol;
  {
    int mycol = 0, mycol2 = 0;
    const int myb = lane < R ? (int)T.r_blk[lane] : -1;
    int myb2 = -1;  // a fly-fly row whose two chains sit in different blocks needs a column in each of them
    if (nsc && lane >= nrc && lane < nrc + nsc) { myb2 = T.c_blk[nc + lane - nrc] >> 8; if (myb2 == myb) myb2 = -1; }
    for (int b = 0; b < NBLK; b++) {  // rank among the lower rows of the same block, by ballots
      const unsigned long long mb = __ballot(myb == b || myb2 == b);
      if (myb == b) mycol = __popcll(mb & ((1ull << lane) - 1ull));
      if (myb2 == b) mycol2 = __popcll(mb & ((1ull << lane) - 1ull));
    }
    if (lane < R) {
      const int b = myb;
      T.r_col[lane] = (unsigned char)mycol;
      if (mycol < KCOL) T.rowof[b][mycol] = (unsigned char)lane;
      if (myb2 >= 0) {
        sc_chain_b(T, nc + lane - nrc)[14] = (unsigned char)mycol2;
        if (mycol2 < KCOL) T.rowof[myb2][mycol2] = (unsigned char)lane;
        mycol = max(mycol, mycol2);
      } else if (nsc && lane >= nrc && lane < nrc + nsc) sc_chain_b(T, nc + lane - nrc)[14] = (unsigned char)mycol;
      mycol += 1;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mycol = max(mycol, __shfl_xor(mycol, off));
    if (mycol > KCOL) c.overflow |= 4;  // more than KCOL rows in one block of M: the env is flagged
                                     // (wave-uniform here: lane 0 reports it) and the rows without a column exert no force (below)
    ncol = min(mycol, KCOL);
    if (mycol > KCOL) {
      // A row without a column would keep its force law but lose its own response (its row of G stays empty): an explicit spring
      // of stiffness D - the blow-ups tools/soak.py saw at saturated actions.  Such rows are switched off for this substep
      // instead: a contact as a whole (all three rows), a limit or fly-fly row through D = 0.
      DM_SYNC();
      const bool nocol = lane < R && ((int)T.r_col[lane] >= KCOL || (myb2 >= 0 && mycol2 >= KCOL));
      if (nocol) { if (lane < nrc) T.c_excl[lane / 3] = 1; else T.r_D[lane] = 0.f; }
    }
  }
  DM_SYNC();
  // G: ball coupling between contact rows, then the fly part M_blk^-1 from the block solves
  const float iIb = frcp(Ib);
  const int gtri = lane * (lane + 1) / 2;  // G is symmetric: row `lane` keeps its columns r2 <= lane
  {
    V3 ja = {0.f, 0.f, 0.f};  // this row's ball Jacobian (zero for limit rows); the other rows' come by v_readlane
    if (lane < nrc) { const V3 jp = cj_ball(T, lane / 3, lane % 3, Rb, bc); ja = {jp.x * iIb, jp.y * iIb, jp.z * iIb}; }
    for (int r2 = 0; r2 < R; r2++) {
      const float gv = (ja.x * rl_f(ja.x, r2) + ja.y * rl_f(ja.y, r2) + ja.z * rl_f(ja.z, r2)) * Ib;
      if (lane < R && r2 <= lane) T.G[gtri + r2] = gv;
    }
  }
  // solve columns: 0 = qfrc_smooth, 1 + c = the rows whose block-local column is c
#pragma unroll 1
  for (int cb = 0; cb < ncol + 1; cb += 4) {
    for (int f = lane; f < ND; f += 64) T.X4[f] = make_float4(0.f, 0.f, 0.f, 0.f);
    DM_SYNC();
    if (cb == 0) {
#pragma unroll
      for (int s = 0; s < 3; s++) if (slot_on(c, s)) T.X4[opq(c.sdof[s])].x = qs[s];
    }
    for (int item = lane; item < nrc * NCH; item += 64) {  // contact rows: J' over the chain
      const int r = item / NCH, p = item - r * NCH, k = r / 3, col = (int)T.r_col[r] + 1 - cb;
      if (col >= 0 && col < 4 && p < T.c_nch[k]) { const int f = T.c_chain[k][p]; (&T.X4[f].x)[col] = cj_row(T, k, r - 3 * k, cj_u(T, k, f, c0)); }
    }
    if (lane >= nrc + nsc && lane < R) {
      const int col = (int)T.r_col[lane] + 1 - cb;
      if (col >= 0 && col < 4) (&T.X4[T.r_dof[lane]].x)[col] = T.r_sgn[lane];
    }
    if (nsc) self_rhs(c.T, lane, nc, nsc, nrc, cb, c0.x, c0.y, c0.z);
    DM_SYNC();
    solve4(c, T.Lm, T.dinv_m);
    if (cb == 0) {
#pragma unroll
      for (int s = 0; s < 3; s++) if (slot_on(c, s)) am[s] = T.X4[opq(c.sdof[s])].x;
    }
    if (lane < R) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane < nrc) {
        const int k = lane / 3, r = lane - 3 * k, nch = T.c_nch[k];
        for (int p = 0; p < nch; p++) {
          const int f = T.c_chain[k][p];
          const float jv = cj_row(T, k, r, cj_u(T, k, f, c0));
          const float4 y = T.X4[f];
          acc.x += jv * y.x; acc.y += jv * y.y; acc.z += jv * y.z; acc.w += jv * y.w;
        }
      } else if (lane < nrc + nsc) {  // fly-fly row: self_gacc below
      } else {
        const float sg = T.r_sgn[lane];
        const float4 y = T.X4[T.r_dof[lane]];
        acc = make_float4(sg * y.x, sg * y.y, sg * y.z, sg * y.w);
      }
      const int b = T.r_blk[lane];
#pragma unroll
      for (int q2 = 0; q2 < 4; q2++) {
        const int gc = cb + q2;
        const float av = q2 == 0 ? acc.x : (q2 == 1 ? acc.y : (q2 == 2 ? acc.z : acc.w));
        if (gc == 0) T.r_y0[lane] += av;  // J a_s
        else if (gc - 1 < KCOL) {
          const int r2 = T.rowof[b][gc - 1];
          if (r2 <= lane) T.G[gtri + r2] += av;  // (255 = no such row)
        }
      }
    }
    if (nsc) self_gacc(c.T, lane, nc, nsc, nrc, cb, c0.x, c0.y, c0.z);
    DM_SYNC();
  }
  BSTAMP(8);  // constraint rows + G
  const float scale2 = [&]() { float sq = 0.f;
#pragma unroll
    for (int s = 0; s < 3; s++) if (slot_on(c, s)) sq += qs[s] * am[s];
    return wave_sum(sq) + dot(qsb, amb); }();
  if (constrained) {
    int nact_ = 0;
    for (int k = 0; k < nc; k++) nact_ += T.c_excl[k] ? 0 : 1;
    const int nslip = (nact_ > 0 && !(c.flags & BF_NO_NOSLIP)) ? M.noslip_iterations : 0;
    ModelPtr mp_ = (ModelPtr)c.M;
    // the dense solver's cost grows with the square of its register-resident size: sizes around the common row counts (3 rows per
    // contact + ~5 joint limits: 17, 20, 23, 26) are instantiated more finely (+1.1 % env-steps/s; every even size: no further gain)
    if (R <= 12) iters = dense_newton<12>(c.T, mp_, lane, R, nrc, scale2, nslip, c.have_ws);
    else if (R <= 16) iters = dense_newton<16>(c.T, mp_, lane, R, nrc, scale2, nslip, c.have_ws);
    else if (R <= 18) iters = dense_newton<18>(c.T, mp_, lane, R, nrc, scale2, nslip, c.have_ws);
    else if (R <= 20) iters = dense_newton<20>(c.T, mp_, lane, R, nrc, scale2, nslip, c.have_ws);
    else if (R <= 22) iters = dense_newton<22>(c.T, mp_, lane, R, nrc, scale2, nslip, c.have_ws);
    else if (R <= 24) iters = dense_newton<24>(c.T, mp_, lane, R, nrc, scale2, nslip, c.have_ws);
    else if (R <= 28) iters = dense_newton<28>(c.T, mp_, lane, R, nrc, scale2, nslip, c.have_ws);
    else if (R <= 32) iters = dense_newton<32>(c.T, mp_, lane, R, nrc, scale2, nslip, c.have_ws);
    else if (R <= 40) iters = dense_newton<40>(c.T, mp_, lane, R, nrc, scale2, nslip, c.have_ws);
    else iters = dense_newton<RMAX>(c.T, mp_, lane, R, nrc, scale2, nslip, c.have_ws);
    BSTAMP(9);  // newton + noslip (dense, in registers)
  }
  iters_out += iters;
  BSTAMP(16);  // noslip
  // ---- constraint forces in joint space, final acceleration a = a_s + M^-1 J' f
  if (lane < nrc) T.c_f[lane / 3][lane % 3] = constrained ? T.r_f[lane] : 0.f;
  if (nsc && lane < nsc) {
    T.c_f[nc + lane][0] = T.r_f[nrc + lane]; T.c_f[nc + lane][1] = 0.f; T.c_f[nc + lane][2] = 0.f;
    T.ws_pid[lane] = T.c_pid[nc + lane]; T.ws_f[lane] = constrained ? T.r_f[nrc + lane] : 0.f;
  }
  if (lane == 0) T.ws_n = nsc;
  DM_SYNC();
  float qc[3], a[3];
#pragma unroll
  for (int s = 0; s < 3; s++) {
    qc[s] = 0.f;
    if (slot_on(c, s)) {
      if (lrow[s] >= 0) { const float lf_ = T.r_f[lrow[s]]; qc[s] = lsgn[s] * lf_; c.wsl[s] = lf_; } else c.wsl[s] = 0.f;
      if (nc) qc[s] += contact_gather(c, opq(c.sbl[s]), opq(c.sdof[s]), c0, T.c_f);
      if (nsc) qc[s] += self_gather(c, opq(c.sbl[s]), opq(c.sdof[s]), c0, T.c_f);
    }
  }
  c.wsc[0] = c.wsc[1] = c.wsc[2] = 0.f;
  for (int k = 0; k < nc; k++) if (T.c_link[k] == lane) { c.wsc[0] = T.c_f[k][0]; c.wsc[1] = T.c_f[k][1]; c.wsc[2] = T.c_f[k][2]; }
  V3 qcb = {0.f, 0.f, 0.f};
  for (int k = 0; k < nc; k++) {
    const float f0 = T.c_f[k][0], f1 = T.c_f[k][1], f2 = T.c_f[k][2];
    const V3 b0 = cj_ball(T, k, 0, Rb, bc), b1 = cj_ball(T, k, 1, Rb, bc), b2 = cj_ball(T, k, 2, Rb, bc);
    qcb.x += b0.x * f0 + b1.x * f1 + b2.x * f2;
    qcb.y += b0.y * f0 + b1.y * f1 + b2.y * f2;
    qcb.z += b0.z * f0 + b1.z * f1 + b2.z * f2;
  }
  // final acceleration a = a_s + M^-1 J' f and the Euler acceleration (M + h B)^-1 (qfrc_smooth + J' f) (mj: mj_Euler, implicit in
  // the joint damping) in one pass: component x through M's factor, component y through the factor of M + h B
  V3 ab = amb + frcp(Ib) * qcb;
  float qe[3];
#pragma unroll
  for (int s = 0; s < 3; s++) if (slot_on(c, s)) T.X4[opq(c.sdof[s])] = make_float4(qc[s], qs[s] + qc[s], 0.f, 0.f);
  DM_SYNC();
  solve_dual(c, T.Lm, T.dinv_m, T.Lh, T.dinv_h);
#pragma unroll
  for (int s = 0; s < 3; s++) {
    const float4 x = slot_on(c, s) ? T.X4[opq(c.sdof[s])] : make_float4(0.f, 0.f, 0.f, 0.f);
    a[s] = am[s] + x.x; qe[s] = x.y;
  }
  DM_SYNC();
  {
    float n2 = 0.f;
#pragma unroll
    for (int s = 0; s < 3; s++) if (slot_on(c, s)) n2 += a[s] * a[s];
    *qacc_norm2 = wave_sum(n2) + dot(ab, ab);
    c.have_ws = 1;
  }
  BSTAMP(17);  // constraint forces, final acceleration
  // ---- sensors (mj: mj_rnePostConstraint + mj_sensorAcc): touch = normal force on the claw, force = interaction
  //      force on the tarsus from its parent, in the tarsus site frame.  Only the linear part of the spatial force is
  //      needed: m (a_lin + alpha x r + w x (v_lin + w x r)) with r = CoM - origin.
  {
    const int parent = l_parent(c), ndof = l_ndof(c);
    const unsigned tree = M.l_tree[lane];
    const int sub = (int)(tree & 0xffu), anc2 = (int)((tree >> 8) & 0xffu) - 1, anc4 = (int)((tree >> 16) & 0xffu) - 1;
    V3 fext = {0.f, 0.f, 0.f};
    float touch = 0.f;
    for (int k = 0; k < nc; k++) {
      if (T.c_link[k] == lane && !T.c_excl[k]) {
        const float *fr = T.c_frame[k];
        const float f0 = T.c_f[k][0], f1 = T.c_f[k][1], f2 = T.c_f[k][2];
        fext = fext + V3{fr[0] * f0 + fr[3] * f1 + fr[6] * f2, fr[1] * f0 + fr[4] * f1 + fr[7] * f2, fr[2] * f0 + fr[5] * f1 + fr[8] * f2};
        if (f0 > 0.f) touch += f0;
      }
    }
    for (int k = nc; k < nc + nsc; k++) {  // fly-fly contacts: -n f on geom1's link, +n f on geom2's (mj_rnePostConstraint's cfrc_ext)
      const int la = T.c_link[k] & 255, lb = T.c_link[k] >> 8;
      if ((la == lane || lb == lane) && !T.c_excl[k]) {
        const float f0 = T.c_f[k][0] * (lb == lane ? 1.f : -1.f);
        fext = fext + V3{T.c_frame[k][0] * f0, T.c_frame[k][1] * f0, T.c_frame[k][2] * f0};
        if (T.c_f[k][0] > 0.f) touch += T.c_f[k][0];  // the contact point lies inside the claw's (larger) touch site
      }
    }
    S6 dacc = zero6();  // sum over the dofs of the ancestor path of cdof * qacc (path sum by pointer jumping, as in stage 1)
#pragma unroll
    for (int s = 0; s < 3; s++) if (s < ndof) dacc = dacc + a[s] * ld6(T.C[opq(c.sdof[s])]);
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int an = r == 0 ? parent : (r == 1 ? anc2 : anc4);
      st6(T.lk[lane], dacc);
      DM_SYNC();
      if (an >= 0) dacc = dacc + ld6(T.lk[an]);
      DM_SYNC();
    }
    const S6 cacc = c.caccb + dacc;
    const V3 r = c.xip - V3{M.thorax_pos[0], M.thorax_pos[1], M.thorax_pos[2]};
    const V3 w = ang(c.cvel);
    V3 fint = c.mass * (lin(cacc) + cross(ang(cacc), r) + cross(w, lin(c.cvel) + cross(w, r))) - fext;
    {  // subtree sum over the link's depth-first lane range
      float *o = T.lk[lane];
      o[0] = fint.x; o[1] = fint.y; o[2] = fint.z;
      DM_SYNC();
      const int maxsub = M.maxsub;
#pragma unroll 1
      for (int t = 1; t < maxsub; t++) if (t < sub) { const float *p = T.lk[lane + t]; fint = fint + V3{p[0], p[1], p[2]}; }
      DM_SYNC();
    }
    const int fi = M.l_force[lane], ti = M.l_touch[lane];
    if (fi >= 0) {
      const Q4 sq = qmul(c.xq, Q4{M.l_fsite[0][lane], M.l_fsite[1][lane], M.l_fsite[2][lane], M.l_fsite[3][lane]});
      const V3 fl_ = mtv(q2m(sq), fint);
      T.sens[3 * fi] += fl_.x; T.sens[3 * fi + 1] += fl_.y; T.sens[3 * fi + 2] += fl_.z;
    }
    if (ti >= 0) T.sens[18 + ti] += touch;
    DM_SYNC();
  }
  BSTAMP(18);  // sensors
  if (!integrate) return;  // mj_forward: state untouched
  // ---- integrate with the Euler acceleration computed above
#pragma unroll
  for (int s = 0; s < 3; s++) {
    if (slot_on(c, s)) { c.v[s] += h * qe[s]; c.q[s] += h * c.v[s]; }
  }
  // ball: no damping, so its Euler acceleration is ab = (tau_smooth + J_b' f) / I
  c.bw = c.bw + h * ab;
  {
    const float wn = fsqrt(dot(c.bw, c.bw));
    if (wn >= 1e-15f) {
      const Q4 dq = axis_angle(frcp(wn) * c.bw, wn * h);
      c.bq = qnormalize(qmul(qnormalize(c.bq), dq));
    }
  }
  BSTAMP(19);  // Euler
}

// ------------------------------------------------------------------------------------------------ kernel
__global__ __launch_bounds__(64, 2) void ball_step_kernel(const BallModel *__restrict__ Mp, BTaskDev K, BState *__restrict__ states,
                                                         const float *__restrict__ act, float *__restrict__ obs, float *__restrict__ rew,
                                                         float *__restrict__ disc, int *__restrict__ st, int batch, int mode, int nphys,
                                                         const int *__restrict__ order, int *__restrict__ cost,
                                                         const unsigned char *__restrict__ reset_mask) {
  // Workgroups are dispatched in index order; `order` lists the envs by decreasing cost of their previous control step, so the
  // expensive ones (more contacts, more Newton iterations) start first and the launch does not end on a few long waves
  // running alone at low occupancy (launch_order.hpp; walk_on_ball, B = 4 096: -3 % launch time).
  if ((int)blockIdx.x >= batch) return;
#ifdef FFE_TRACE
  const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime();
  const int tr_prev = cost[order[blockIdx.x]];
#endif
  const int env = order[blockIdx.x], lane = threadIdx.x;
  if (mode == 3) {  // ffe_reset_envs: only the masked envs start a new episode; the others keep state and output rows
    if (!reset_mask[env]) return;
    mode = 1;
  }
  __shared__ BTile T;
  const BallModel &M = *Mp;
  BState &S = states[env];
  Ctx c;
  c.M = Mp; c.T = &T; c.lane = lane; c.flags = K.flags; c.overflow = 0;
#ifdef FFB_STAMPS
  c.st_t0 = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < 24; k++) c.st_acc[k] = 0;
#endif
  c.lpack = M.l_pack[lane]; c.xh = M.x_on[lane];
#pragma unroll
  for (int s = 0; s < 3; s++) {
    const int f = M.s_dof[s][lane];
    c.sdof[s] = f;
    c.sbl[s] = f >= 0 ? ((unsigned)M.d_blk[f] | ((unsigned)M.d_li[f] << 8)) : 0u;
  }
  const bool do_reset = (mode == 1) || (mode == 0 && S.needs_reset != 0);
  const bool phys_only = (mode == 2);
  float act_reg = 0.f, ctrl_reg = 0.f;
  int step_counter = S.step_counter, iters = 0;
  if (do_reset) {
    // ref: walk_on_ball.py:52-54 + fruitfly.py:330-340: qpos0, zero velocity / activation, wings folded to their spring reference
#pragma unroll
    for (int s = 0; s < 3; s++) { c.q[s] = 0.f; c.v[s] = 0.f; }
#pragma unroll
    for (int s = 0; s < 3; s++)
      if (slot_on(c, s)) for (int w = 0; w < M.nwing; w++) if (M.wing_dof[w] == c.sdof[s]) c.q[s] = M.qspring[c.sdof[s]];
    c.bq = {1.f, 0.f, 0.f, 0.f}; c.bw = {0.f, 0.f, 0.f};
    c.have_ws = 0;
#pragma unroll
    for (int s = 0; s < 3; s++) { c.wsl[s] = 0.f; c.wsc[s] = 0.f; }
    step_counter = 0;
  } else {
#pragma unroll
    for (int s = 0; s < 3; s++) { c.q[s] = slot_on(c, s) ? S.q[c.sdof[s]] : 0.f; c.v[s] = slot_on(c, s) ? S.v[c.sdof[s]] : 0.f; }
    c.bq = {S.ballq[0], S.ballq[1], S.ballq[2], S.ballq[3]}; c.bw = {S.ballw[0], S.ballw[1], S.ballw[2]};
    c.have_ws = S.have_ws;
#pragma unroll
    for (int s = 0; s < 3; s++) { c.wsl[s] = slot_on(c, s) ? S.qacc_ws[c.sdof[s]] : 0.f; c.wsc[s] = S.wsc[lane][s]; }
    act_reg = lane < NU ? S.act[lane] : 0.f;
    if (lane < NU) {
      if (phys_only) ctrl_reg = act[(size_t)env * NU + lane];
      else {
        // ref: fruitfly.py:480-492 apply_action (+ the CanonicalSpecWrapper the reference wraps every env in)
        const int ai = M.a_action[lane];
        float av = ai >= 0 ? act[(size_t)env * NACT + ai] : 0.f;
        if (!(av == av)) av = 0.f;
        if (K.canonical && ai >= 0) {
          if (K.clip) av = fminf(fmaxf(av, -1.f), 1.f);
          av = M.act_lo[ai] + 0.5f * (av + 1.f) * (M.act_hi[ai] - M.act_lo[ai]);
        }
        ctrl_reg = av;
      }
    }
    if (!phys_only) step_counter++;
  }
  if (lane < 24) T.sens[lane] = 0.f;
  if (lane < NSD) { T.sd_pid[lane] = S.sd_pid[lane]; T.sd_n[lane][0] = S.sd_n[lane][0]; T.sd_n[lane][1] = S.sd_n[lane][1]; T.sd_n[lane][2] = S.sd_n[lane][2]; T.sd_n[lane][3] = S.sd_n[lane][3]; }
  if (lane == 0) { T.sd_cnt = do_reset ? 0 : S.sd_cnt; T.ws_n = do_reset ? 0 : S.ws_n; }
  if (lane < NC) { T.ws_pid[lane] = S.ws_pid[lane]; T.ws_f[lane] = S.ws_f[lane]; }
  DM_SYNC();
  const int nsub = phys_only ? nphys : M.nsub;
  float qn2 = 0.f;
  // one copy of each stage: stage1 ; [stage2 ; stage1] x nsub.  A reset is stage1 ; stage2 without actuation and without
  // integrating (mj_forward, dm_control's after_reset); its sensors are the first sample of the buffers.
  if (do_reset) c.flags |= BF_NO_ACTUATION;
  unsigned long long con_hist = 0ull, det_hist = 0ull;
#pragma unroll 1
  for (int s = 0;; s++) {
    stage1(c);
    if (!do_reset && s == nsub) break;
    if (s < 16) con_hist |= (unsigned long long)min(c.nact, 15) << (4 * s);
    if (s < 12) det_hist |= (unsigned long long)min(c.nc + c.nsc, 31) << (5 * s);
    float act_new;
    stage2(c, act_reg, ctrl_reg, act_new, !do_reset, iters, &qn2);
    if (do_reset) break;
    act_reg = act_new;
  }
  c.flags = K.flags;
  // ---- store state
#pragma unroll
  for (int s = 0; s < 3; s++) if (slot_on(c, s)) { S.q[c.sdof[s]] = c.q[s]; S.v[c.sdof[s]] = c.v[s]; S.qacc_ws[c.sdof[s]] = c.wsl[s]; }
#pragma unroll
  for (int s = 0; s < 3; s++) S.wsc[lane][s] = c.wsc[s];
  if (lane < NU) S.act[lane] = do_reset ? 0.f : act_reg;
  if (lane < NSD) { S.sd_pid[lane] = T.sd_pid[lane]; S.sd_n[lane][0] = T.sd_n[lane][0]; S.sd_n[lane][1] = T.sd_n[lane][1]; S.sd_n[lane][2] = T.sd_n[lane][2]; S.sd_n[lane][3] = T.sd_n[lane][3]; }
  if (lane == 0) { S.sd_cnt = T.sd_cnt; S.ws_n = T.ws_n; }
  if (lane < NC) { S.ws_pid[lane] = T.ws_pid[lane]; S.ws_f[lane] = T.ws_f[lane]; }
  if (lane == 0) {
    S.ballq[0] = c.bq.w; S.ballq[1] = c.bq.x; S.ballq[2] = c.bq.y; S.ballq[3] = c.bq.z;
    S.ballw[0] = c.bw.x; S.ballw[1] = c.bw.y; S.ballw[2] = c.bw.z;
    S.step_counter = step_counter; S.iters = iters; S.ncon = c.nc + c.nsc; S.nself = c.nsc;  // (contacts that take part in something: an inactive one without an adhesion actuator on either body is not kept)
    S.con_hist[0] = (unsigned)con_hist; S.con_hist[1] = (unsigned)(con_hist >> 32);
    S.det_hist[0] = (unsigned)det_hist; S.det_hist[1] = (unsigned)(det_hist >> 32);
    // key of the next launch's order: what a wave's lifetime varies with - Newton iterations over the step's substeps (4 us each)
    // and the number of contacts (40 us each; least-squares fit of the lifetimes in tools/wave_timeline.py's trace)
    cost[env] = min(255, iters + 10 * (c.nc + c.nsc));
    if (do_reset) S.overflow = 0; else S.overflow |= c.overflow;  // sticky over the episode: 1 contacts > 10, 2 constraint rows > 32, 4 rows of one block > 12
    S.have_ws = do_reset ? 0 : c.have_ws;
  }
#ifdef FFB_STAMPS
  BSTAMP(20);  // prologue + state store
  if (lane == 0) for (int k = 0; k < 24; k++) atomicAdd(&g_bstamps[k], c.st_acc[k]);
#endif
  if (phys_only) return;
  // ---- observation (ref: SURVEY App. A order): accelerometer 3 | actuator_activation 59 | appendages_pos 21 | ball_qvel 3 |
  //      force 18 | gyro 3 | joints_pos 85 | joints_vel 85 | touch 6 | velocimeter 3 | world_zaxis 3
#pragma unroll
  for (int s = 0; s < 3; s++) if (slot_on(c, s)) { T.Q[c.sdof[s]] = c.q[s]; T.V[c.sdof[s]] = c.v[s]; }
  {
    float *o = T.lk[lane];
    o[0] = c.xp.x; o[1] = c.xp.y; o[2] = c.xp.z; o[3] = c.xq.w; o[4] = c.xq.x; o[5] = c.xq.y; o[6] = c.xq.z;
  }
  DM_SYNC();
  float *ob = obs + (size_t)env * NOBS;
  const float inv = (do_reset && K.pad_first_obs) ? 1.f : 1.f / (float)M.nsub;
  const int nsamp = do_reset ? 1 : M.nsub;
  const M3 Rt = q2m(Q4{M.thorax_quat[0], M.thorax_quat[1], M.thorax_quat[2], M.thorax_quat[3]});
  const M3 Rs = q2m(Q4{M.site_quat[0], M.site_quat[1], M.site_quat[2], M.site_quat[3]});
  if (lane < 3) {
    // the thorax is welded to the world: accelerometer = R_site' (0, 0, -g), gyro = velocimeter = 0
    const V3 g = mtv(Rs, V3{0.f, 0.f, (c.flags & BF_NO_GRAVITY) ? 0.f : -M.gz});
    ob[lane] = (lane == 0 ? g.x : (lane == 1 ? g.y : g.z)) * (float)nsamp * inv;
    ob[3 + NU + 21 + 3 + 18 + lane] = 0.f;                       // gyro
    ob[3 + NU + 21 + 3 + 18 + 3 + 2 * NOBSJ + 6 + lane] = 0.f;   // velocimeter
    ob[3 + NU + 21 + 3 + 18 + 3 + 2 * NOBSJ + 6 + 3 + lane] = lane == 0 ? Rt.m6 : (lane == 1 ? Rt.m7 : Rt.m8);  // world_zaxis = xmat[6:9]
    ob[3 + NU + 21 + lane] = lane == 0 ? c.bw.x : (lane == 1 ? c.bw.y : c.bw.z);  // ball_qvel
  }
  if (lane < NU) ob[3 + lane] = do_reset ? 0.f : act_reg;
  if (lane < 7) {  // appendages_pos: (x_site - x_thorax) . R_thorax (ref: fruitfly.py:629-638)
    const int l = M.app_link[lane];
    const float *p = T.lk[l];
    const V3 sp = V3{p[0], p[1], p[2]} + qrot(Q4{p[3], p[4], p[5], p[6]}, V3{M.app_pos[lane][0], M.app_pos[lane][1], M.app_pos[lane][2]});
    const V3 rel = mtv(Rt, sp - V3{M.thorax_pos[0], M.thorax_pos[1], M.thorax_pos[2]});
    ob[3 + NU + 3 * lane] = rel.x; ob[3 + NU + 3 * lane + 1] = rel.y; ob[3 + NU + 3 * lane + 2] = rel.z;
  }
  if (lane < 18) ob[3 + NU + 21 + 3 + lane] = T.sens[lane] * inv;
  if (lane < 6) ob[3 + NU + 21 + 3 + 18 + 3 + 2 * NOBSJ + lane] = T.sens[18 + lane] * inv;
  for (int k = lane; k < NOBSJ; k += 64) {
    ob[3 + NU + 21 + 3 + 18 + 3 + k] = T.Q[M.obs_dof[k]];
    ob[3 + NU + 21 + 3 + 18 + 3 + NOBSJ + k] = T.V[M.obs_dof[k]];
  }
  // ---- reward / termination (ref: walk_on_ball.py:61-79, base.py:213-217)
  if (lane == 0) {
    if (do_reset) { rew[env] = 0.f; disc[env] = 1.f; st[env] = 0; S.needs_reset = 0; }
    else {
      float r = fmaxf(0.f, 1.f - fabsf(c.bw.x) / 6.f) * fmaxf(0.f, 1.f - fabsf(c.bw.y + 5.f) / 6.f) * fmaxf(0.f, 1.f - fabsf(c.bw.z) / 6.f);
      const bool bad = !(qn2 == qn2) || !(sqrtf(qn2) <= 1e14f);
      const bool timeup = step_counter >= K.time_limit_steps;
      if (bad && !(r == r)) r = 0.f;
      rew[env] = r; disc[env] = bad ? 0.f : 1.f; st[env] = (bad || timeup) ? 2 : 1;
      S.needs_reset = (bad || timeup) ? 1 : 0;
    }
  }
#ifdef FFE_TRACE
  __builtin_amdgcn_s_waitcnt(0);
  if (lane == 0 && blockIdx.x < 32768) {
    g_btrace[blockIdx.x][0] = tr_t0; g_btrace[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
    g_btrace[blockIdx.x][2] = __builtin_amdgcn_s_getreg((31 << 11) | 4); g_btrace[blockIdx.x][3] = (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf) | ((unsigned long long)((unsigned)(iters & 0xff) | ((unsigned)((c.nc + c.nsc) & 0xff) << 8) | ((unsigned)(tr_prev & 0xffff) << 16)) << 8);
  }
#endif
}

__global__ void ball_init_states(BState *states, int *order, int *cost, int batch) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch) return;
  BState z;
  memset(&z, 0, sizeof(z));
  z.needs_reset = 1; z.ballq[0] = 1.f;
  states[i] = z;
  order[i] = i;
  cost[i] = 0;
}
__global__ void ball_get_state_kernel(const BState *states, double *qpos, double *qvel, int batch) {
  const int env = blockIdx.x, t = threadIdx.x;
  if (env >= batch) return;
  const BState &S = states[env];
  for (int k = t; k < 106; k += blockDim.x) qpos[(size_t)env * 106 + k] = k < 4 ? (double)S.ballq[k] : (double)S.q[k - 4];
  for (int k = t; k < 105; k += blockDim.x) qvel[(size_t)env * 105 + k] = k < 3 ? (double)S.ballw[k] : (double)S.v[k - 3];
}
__global__ void ball_set_state_kernel(BState *states, const double *qpos, const double *qvel, int batch) {
  const int env = blockIdx.x, t = threadIdx.x;
  if (env >= batch) return;
  BState &S = states[env];
  for (int k = t; k < 106; k += blockDim.x) { if (k < 4) S.ballq[k] = (float)qpos[(size_t)env * 106 + k]; else S.q[k - 4] = (float)qpos[(size_t)env * 106 + k]; }
  for (int k = t; k < 105; k += blockDim.x) { if (k < 3) S.ballw[k] = (float)qvel[(size_t)env * 105 + k]; else S.v[k - 3] = (float)qvel[(size_t)env * 105 + k]; }
  if (t == 0) { S.have_ws = 0; S.sd_cnt = 0; S.ws_n = 0; }
}
__global__ void ball_act_kernel(BState *states, double *act, int batch, int set) {
  const int env = blockIdx.x, t = threadIdx.x;
  if (env >= batch || t >= NU) return;
  if (set) states[env].act[t] = (float)act[(size_t)env * NU + t];
  else act[(size_t)env * NU + t] = (double)states[env].act[t];
}
__global__ void ball_task_state_kernel(const BState *states, int *ints, double *reals, int batch) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch) return;
  const BState &S = states[i];
  int *o = ints + (size_t)i * 8;
  o[0] = (int)S.con_hist[0]; o[1] = (int)S.con_hist[1]; o[2] = S.step_counter; o[3] = S.nself; o[4] = S.needs_reset; o[5] = S.ncon; o[6] = S.iters; o[7] = S.overflow;
  for (int k = 0; k < 8; k++) reals[(size_t)i * 8 + k] = 0.0;
  reals[(size_t)i * 8] = (double)((unsigned long long)S.det_hist[0] | ((unsigned long long)S.det_hist[1] << 32));  // (60 bits used: exact up to 10 substeps)
}

// ================================================================================================ host side
#define HIPB_OK(expr)                                                                               \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

struct BallEnv {
  int device = 0, batch = 0;
  BallHost host;
  BTaskDev task{};
  BallModel *model_dev = nullptr;
  BState *states = nullptr;
  int *order = nullptr, *cost = nullptr;  // launch order of the envs and its sort keys (launch_order.hpp)
  bool timing = false; double timing_ms = 0.0;  // ball_time_kernel: events around the step kernel alone
  double control_timestep = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

struct BallEnvDeleter { void operator()(BallEnv *e) const { ball_destroy(e); } };

// the caller (fly_env.hip) has made `device` current
BallEnv *ball_create(const void *blob, size_t blob_size, const BallTaskHost &task, int batch, int device) {
  if (!blob || batch <= 0) throw std::runtime_error("ffe_create_walk_on_ball: bad arguments");
  std::unique_ptr<BallEnv, BallEnvDeleter> e(new BallEnv());  // frees the device allocations made so far if a later step throws
  Blob b(blob, blob_size);
  e->host = build_ball_model(b);
  e->device = device; e->batch = batch;
  e->control_timestep = task.control_timestep;
  e->host.m.nsub = (int)llround(task.control_timestep / (double)e->host.m.h);
  if (e->host.m.nsub < 1 || e->host.m.nsub > 64) throw std::runtime_error("walk_on_ball: bad control timestep");
  e->task.time_limit_steps = task.time_limit_steps; e->task.pad_first_obs = task.pad_first_obs; e->task.flags = task.physics_flags;
  e->task.canonical = task.canonical_actions; e->task.clip = task.clip_actions;
  HIPB_OK(hipMalloc((void **)&e->model_dev, sizeof(BallModel)));
  HIPB_OK(hipMemcpy(e->model_dev, &e->host.m, sizeof(BallModel), hipMemcpyHostToDevice));
  HIPB_OK(hipMalloc((void **)&e->states, sizeof(BState) * (size_t)batch));
  HIPB_OK(hipMalloc((void **)&e->order, sizeof(int) * (size_t)batch));
  HIPB_OK(hipMalloc((void **)&e->cost, sizeof(int) * (size_t)batch));
  hipLaunchKernelGGL(ball_init_states, dim3((batch + 63) / 64), dim3(64), 0, 0, e->states, e->order, e->cost, batch);
  HIPB_OK(hipGetLastError());
  HIPB_OK(hipDeviceSynchronize());
  HIPB_OK(hipEventCreate(&e->ev0));
  HIPB_OK(hipEventCreate(&e->ev1));
  return e.release();
}
void ball_destroy(BallEnv *e) {
  if (!e) return;
  if (e->model_dev) (void)hipFree(e->model_dev);
  if (e->states) (void)hipFree(e->states);
  if (e->order) (void)hipFree(e->order);
  if (e->cost) (void)hipFree(e->cost);
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  delete e;
}
void ball_spec(const BallEnv *e, int *nq, int *nv, int *nu, int *action_dim, int *obs_dim, int *nsub, double *h, double *ctrl_dt) {
  *nq = 106; *nv = 105; *nu = NU; *action_dim = NACT; *obs_dim = NOBS; *nsub = e->host.m.nsub; *h = e->host.m.h; *ctrl_dt = e->control_timestep;
}
void ball_action_bounds(const BallEnv *e, float *mn, float *mx) {
  for (int k = 0; k < NACT; k++) { mn[k] = e->host.action_min[k]; mx[k] = e->host.action_max[k]; }
}
void ball_launch(BallEnv *e, const float *act, float *obs, float *rew, float *disc, int32_t *st, void *stream, int mode, int nphys,
                 const uint8_t *mask) {
  if (mode != 1 && mode != 3 && !act) throw std::runtime_error("walk_on_ball: null action buffer");
  if (mode != 2 && (!obs || !rew || !disc || !st)) throw std::runtime_error("walk_on_ball: null output buffer");
  if (mode == 3 && !mask) throw std::runtime_error("walk_on_ball: null reset mask");
  if (e->timing) HIPB_OK(hipEventRecord(e->ev0, (hipStream_t)stream));
  hipLaunchKernelGGL(ball_step_kernel, dim3(e->batch), dim3(64), 0, (hipStream_t)stream, e->model_dev, e->task, e->states, act, obs, rew, disc, st,
                     e->batch, mode, nphys, e->order, e->cost, mask);
  HIPB_OK(hipGetLastError());
  if (e->timing) HIPB_OK(hipEventRecord(e->ev1, (hipStream_t)stream));
  if (mode == 0 && e->batch > 1) {
    hipLaunchKernelGGL(ffe_order::order_by_cost, dim3(1), dim3(1024), 0, (hipStream_t)stream, e->cost, e->order, e->batch);
    HIPB_OK(hipGetLastError());
  }
  if (e->timing) {
    float t = 0.f;
    HIPB_OK(hipEventSynchronize(e->ev1));
    HIPB_OK(hipEventElapsedTime(&t, e->ev0, e->ev1));
    e->timing_ms += t;
  }
}
void ball_get_state(BallEnv *e, double *qpos, double *qvel, void *stream) {
  hipLaunchKernelGGL(ball_get_state_kernel, dim3(e->batch), dim3(128), 0, (hipStream_t)stream, e->states, qpos, qvel, e->batch);
  HIPB_OK(hipGetLastError());
}
void ball_set_state(BallEnv *e, const double *qpos, const double *qvel, void *stream) {
  hipLaunchKernelGGL(ball_set_state_kernel, dim3(e->batch), dim3(128), 0, (hipStream_t)stream, e->states, qpos, qvel, e->batch);
  HIPB_OK(hipGetLastError());
}
void ball_get_act(BallEnv *e, double *act, void *stream) {
  hipLaunchKernelGGL(ball_act_kernel, dim3(e->batch), dim3(64), 0, (hipStream_t)stream, e->states, act, e->batch, 0);
  HIPB_OK(hipGetLastError());
}
void ball_set_act(BallEnv *e, const double *act, void *stream) {
  hipLaunchKernelGGL(ball_act_kernel, dim3(e->batch), dim3(64), 0, (hipStream_t)stream, e->states, const_cast<double *>(act), e->batch, 1);
  HIPB_OK(hipGetLastError());
}
void ball_get_task_state(BallEnv *e, int32_t *ints, double *reals, void *stream) {
  hipLaunchKernelGGL(ball_task_state_kernel, dim3((e->batch + 63) / 64), dim3(64), 0, (hipStream_t)stream, e->states, ints, reals, e->batch);
  HIPB_OK(hipGetLastError());
}
#ifdef FFE_TRACE
extern "C" int ffb_debug_read_trace(unsigned long long *out, int nrows) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_btrace), (size_t)nrows * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef FFB_STAMPS
extern "C" int ffb_debug_read_stamps(unsigned long long *out24, int reset) {
  if (hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_bstamps), 24 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[24] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_bstamps), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
float ball_time_steps(BallEnv *e, const float *act, float *obs, float *rew, float *disc, int32_t *st, int iters, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  HIPB_OK(hipEventRecord(e->ev0, s));
  for (int k = 0; k < iters; k++) ball_launch(e, act, obs, rew, disc, st, stream, 0, 0, nullptr);
  HIPB_OK(hipEventRecord(e->ev1, s));
  HIPB_OK(hipEventSynchronize(e->ev1));
  float ms = 0.f;
  HIPB_OK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
  return ms / (float)iters;
}

float ball_time_kernel(BallEnv *e, const float *act, float *obs, float *rew, float *disc, int32_t *st, int iters, void *stream) {
  e->timing = true; e->timing_ms = 0.0;
  for (int k = 0; k < iters; k++) ball_launch(e, act, obs, rew, disc, st, stream, 0, 0, nullptr);
  e->timing = false;
  return (float)(e->timing_ms / iters);
}

}  // namespace ffb
