// ball_model.hpp - host-side construction of the device tables for the walk_on_ball environment.
//
// The walk_on_ball model (ref: fly_envs.py:125-157, tasks/walk_on_ball.py, tasks/arenas/ball.py) is a fly whose
// thorax is welded to the world standing on a free-spinning ball: 66 moving fly bodies (102 hinge dofs), one ball
// (3 dofs), 59 filtered actuators.  Mapping onto one 64-lane wavefront per environment:
//   lanes <-> the 64 fly links that carry legs, head, mouth, antennae, wings and abdomen (body space work);
//            each lane also owns the <= 3 hinge dofs of its link (joint space work);
//   the two halteres (single hinges on the fixed thorax, no collision geoms, no actuators) are closed-form
//            one-dof systems parked in a free dof slot of two lanes;
//   the ball is a sphere spinning about its fixed centre: isotropic inertia, no bias force, closed-form drag;
//            it only couples to the legs through the contact rows;
//   the joint-space inertia (580 non-zeros in 12 independent blocks) lives in LDS, so any lane can touch any entry:
//            assembly, the block factorisation and the triangular solves follow host-built schedules of
//            64-wide slots (fac_a/fac_b, p1, p2, ent_a/ent_b below).
// All tables are lane-major ([field][lane]) so a wavefront's read is one coalesced transaction.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "dev_model.hpp"

namespace ffb {

using ffe::Blob;
using ffe::Tensor;

constexpr int NL = 64;      // fly links handled by lanes
constexpr int ND = 102;     // fly hinge dofs (oracle dof index - 3)
constexpr int NDP = 104;    // padded
constexpr int NMMAX = 584;  // M entries (580) padded
constexpr int ECAP = 10;    // M entries per lane
constexpr int NSTEP = 14;   // pivots per block (largest block: abdomen / head tree, 14 dofs)
constexpr int NBLK = 12;    // independent blocks of M (6 legs, head tree, abdomen, 2 wings, 2 halteres)
#ifndef FFB_NC
#define FFB_NC 16
#define FFB_RMAX 48
#define FFB_KCOL 24
#endif
constexpr int NC = FFB_NC;      // contact capacity per env, ball and fly-fly contacts together (overflow is flagged)
constexpr int RMAX = FFB_RMAX;    // constraint rows per env: 3 per ball contact + 1 per fly-fly contact + instantiated joint limits (overflow is flagged)
constexpr int KCOL = FFB_KCOL;    // solve columns per block of M = constraint rows one block can carry (a leg at saturated actions: 5 ball contacts + limits)
constexpr int NCH = 14;     // fly dofs a contact row can touch (deepest chain)
constexpr int NU = 59;      // actuators
constexpr int NWRAP = 7;    // transmission terms per actuator
constexpr int NOBSJ = 85;
constexpr int NOBS = 289;
constexpr int NACT = 59;
constexpr int MAXDEPTH = 8;
constexpr int NFS = 56;     // slots of the factorisation schedule (64 entry updates per slot)
constexpr int NPS = 24;     // slots of each triangular-solve schedule
constexpr int NPG = 48;     // sphere / capsule collision geoms on fly links ("primitive geoms": legs 42, mouth 3, antennae 2, abdomen_7)
constexpr int NG = 72;      // all collision geoms of the fly (70), padded
constexpr int NCP = 1152;   // candidate pairs with an ellipsoid or cylinder on one side (static list, flybody_amd/model/reach.py), padded to 64
constexpr int NXB = 12;     // ellipsoid / cylinder geoms that can reach the ball
constexpr int CL1 = 192;    // pairs that may pass the bounding-sphere test of one substep (seen: 120)
constexpr int CL2 = 24;     // pairs one narrow phase takes (seen: 9)
constexpr int NSD = 12;     // convex pairs whose separating direction is remembered from substep to substep

struct BallModel {
  // ---- options
  float h, gz, rho, beta;
  int nsub, nM, maxdepth, nblk;
  // ---- links (lane = link)
  int l_parent[NL], l_depth[NL], l_nchild[NL], l_child[3][NL], l_ndof[NL], l_body[NL];
  float l_pos[3][NL], l_quat[4][NL];  // frame in parent link; world frame for depth-1 links (thorax is fixed)
  float l_mass[NL], l_ipos[3][NL], l_iquat[4][NL], l_inertia[3][NL], l_fl[8][NL];  // l_fl: inertia-box drag coefficients
  // capsule geom for ball contacts (one per leg link; abdomen_7's sphere is a zero-length capsule)
  int g_has[NL];
  float g_pos[3][NL], g_axis[3][NL], g_rad[NL], g_half[NL], g_margin[NL], g_gap[NL], g_fric[NL], g_K[NL], g_B[NL], g_solimp[5][NL],
      g_invw[NL];
  int l_touch[NL], l_force[NL], l_adh[NL];  // sensor / adhesion-actuator index carried by this link, or -1
  float l_fsite[4][NL];                     // force-sensor site orientation in the link frame
  int l_chain[NCH][NL], l_nchain[NL];       // fly dofs from the chain root down to this link's last dof
  // ---- dof slots (slot s of lane = s-th hinge of the link); x_on lanes carry a haltere in slot 2 (arrays keep 4 slots)
  int s_dof[4][NL];
  float s_axis[3][3][NL], s_jpos[3][3][NL];
  float s_stiff[4][NL], s_sref[4][NL], s_damp[4][NL], s_arm[4][NL], s_lo[4][NL], s_hi[4][NL], s_invw[4][NL], s_K[4][NL], s_B[4][NL],
      s_solimp[5][4][NL];
  int s_limited[4][NL], s_act[2][4][NL];  // s_act[0]: joint actuator id, s_act[1]: tendon actuator id (or -1)
  float s_actcoef[2][4][NL];
  // halteres (x_on lanes): constant inertia, gravity torque Gc cos q + Gs sin q, drag -cv qd - cq |qd| qd
  float x_M[NL], x_Gc[NL], x_Gs[NL], x_cv[NL], x_cq[NL];
  int x_on[NL];  // lane carries a haltere in slot 2
  // ---- ball
  float b_I, b_center[3], b_radius, b_iquat[4], b_fl[8], b_fric, b_K_unused;
  // ---- joint-space inertia structure (fly dofs)
  short d_parent[NDP], d_madr[NDP], d_blk[NDP], d_li[NDP];
  unsigned short d_amask[NDP];
  float d_arm[NDP];
  // Schedules (any lane may process any entry: the matrices live in LDS).  One slot = up to 64 independent updates.
  //   fac_a: adr_e | adr_kk << 10 | adr_ki << 20 | valid << 31, fac_b: adr_kj      L[e] -= L[ki] L[kj] / L[kk]
  //   p1 / p2: adr_e | i << 10 | j << 17 | valid << 31                              x[j] -= L[e] x[i]  /  x[i] -= L[e] x[j]
  //   ent_a: i | j << 8 | adr << 16 | valid << 31, ent_b: blk | li_i << 4 | li_j << 8 (assembly, final scaling)
  unsigned fac_a[NFS][NL], fac_b[NFS][NL], p1[NPS][NL], p2[NPS][NL], ent_a[ECAP][NL], ent_b[ECAP][NL];
  int nfs, np1, np2;
  // ---- actuators (lane = actuator)
  int a_trn[NL], a_nwrap[NL], a_wdof[NWRAP][NL], a_action[NL], a_link[NL];
  float a_wcoef[NWRAP][NL], a_gain[NL], a_b0[NL], a_b1[NL], a_b2[NL], a_clo[NL], a_chi[NL], a_flo[NL], a_fhi[NL], a_tau[NL];
  int a_climited[NL], a_flimited[NL];
  // ---- observation bookkeeping
  int obs_dof[NOBSJ + 3], app_link[8];
  float app_pos[8][3], thorax_pos[3], thorax_quat[4], site_quat[4];
  int dof_of_oracle[ND + 8];  // oracle dof (>= 3) -> fly dof; identity minus 3, kept for clarity
  float qpos0[ND + 8], qspring[ND + 8];
  int wing_dof[8], nwing;
  float act_lo[NL], act_hi[NL];
  float meaninertia;
  int noslip_iterations;
  float j_solimp[5], c_solimp[5];  // joint-limit / ball-contact impedance parameters (uniform over the model; checked on the host)
  unsigned l_pack[NL], l_kids[NL];  // parent+1 | depth << 8 | ndof << 12 ;  child0 | child1 << 8 | child2 << 16 | nchild << 24
  unsigned l_tree[NL];              // subtree size (links are in depth-first order: the subtree of l is lanes l .. l + size - 1)
                                    // | (ancestor 2 levels up) + 1 << 8 | (ancestor 4 levels up) + 1 << 16
  int maxsub;                       // largest subtree size
  // ---- fly-fly collision (mj: mj_collision over the sphere / capsule pairs that survive MuJoCo's filters: same weld body,
  //      parent-child, <exclude> (fruitfly.xml:733-760 + walk_on_ball.py:33-40), contype / conaffinity); condim 1, frictionless
  //      (fruitfly.xml:17).  A link publishes up to two primitive geoms into LDS slots; lane s then owns slot s and tests it
  //      against slots s + 1 .. s + NPG / 2 (mod NPG), so every unordered pair is visited once; sp_mask[s] bit j = (s, j) is a
  //      candidate pair.  geom1 / geom2 in MuJoCo's order: the sphere first (lower type code), else the lower slot (checked).
  int g_slot[NL];                              // slot of the link's (first) primitive geom = the g_* capsule above, or -1
  int pg2_lane, pg2_slot;                      // the one link with a second primitive geom (rostrum: left + right capsule)
  float pg2_pos[3], pg2_axis[3], pg2_half, pg2_rad;
  int pgs_link[NPG];
  float pgs_invw[NPG];
  unsigned long long sp_mask[NL], sp_claw;  // candidate partners of each slot; slots whose geom carries the claw margin / gap
  int npg, nsp, sp_sphere;                  // sp_sphere: slot of the one sphere (abdomen_7), or -1
  float sc_margin, sc_gap, sc_K, sc_B, sc_solimp[5];  // margin / gap of a pair with a claw geom (others 0); K, B, solimp uniform (checked)
  // ---- pairs with an ellipsoid or a cylinder on one side (mj: mjc_Convex; thorax, head, rostrum, labrum, wings, coxae, abdomen
  //      segments: fruitfly.xml:323-443).  All 70 fly geoms in model order; cg_link = lane of the geom's link or -1 (thorax: fixed to
  //      the world, cg_pos / cg_quat then hold the world frame); cp_pair = g1 | g2 << 8 with g1 the lower type code (mj_collision's
  //      order), 0xffff = padding; xb_* = the ellipsoids / cylinders among the ball's candidate partners with their contact parameters
  //      (mj_contactParam against the ball's sphere, as g_* above).
  int ncg, ncp, nxb;
  int cg_link[NG], cg_type[NG];
  float cg_pos[3][NG], cg_quat[4][NG], cg_size[3][NG], cg_brad[NG], cg_invw[NG];
  unsigned long long cg_mmask[2];  // geoms that carry the claw margin / gap
  unsigned short cp_pair[NCP];
  int xb_geom[NXB];
  float xb_margin[NXB], xb_gap[NXB], xb_fric[NXB], xb_K[NXB], xb_B[NXB], xb_invw[NXB];
};

namespace detail {
inline void q2m(const double *q, double *m) { ffe::quat2mat_d(q, m); }
inline void qmul(const double *a, const double *b, double *r) {
  r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  r[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  r[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  r[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
inline void rot(const double *q, const double *v, double *r) {
  double m[9];
  q2m(q, m);
  for (int k = 0; k < 3; k++) r[k] = m[3 * k] * v[0] + m[3 * k + 1] * v[1] + m[3 * k + 2] * v[2];
}
inline void cross(const double *a, const double *b, double *r) {
  r[0] = a[1] * b[2] - a[2] * b[1]; r[1] = a[2] * b[0] - a[0] * b[2]; r[2] = a[0] * b[1] - a[1] * b[0];
}
inline void kb(double tc, double dr, double dmax, double h, double *K, double *B) {
  dmax = std::min(std::max(dmax, 1e-4), 0.9999);
  if (tc > 0) {
    tc = std::max(tc, 2 * h);
    *K = 1.0 / std::max(1e-15, dmax * dmax * tc * tc * dr * dr);
    *B = 2.0 / std::max(1e-15, dmax * tc);
  } else { *K = -tc / std::max(1e-15, dmax * dmax); *B = -dr / std::max(1e-15, dmax); }
}
}  // namespace detail

struct BallHost {
  BallModel m;
  std::vector<float> action_min, action_max;
  int nq = 106, nv = 105;
};

inline BallHost build_ball_model(const Blob &b) {
  using namespace detail;
  BallHost H;
  BallModel &M = H.m;
  std::memset(&M, 0, sizeof(M));
  const Tensor &opt = b.get("opt");
  const double h = opt.f(0), rho = opt.f(1), beta = opt.f(2), gz = opt.f(5);
  M.h = (float)h; M.gz = (float)gz; M.rho = (float)rho; M.beta = (float)beta;
  const Tensor &bpar = b.get("body_parentid"), &bpos = b.get("body_pos"), &bquat = b.get("body_quat"), &bmass = b.get("body_mass"),
               &bipos = b.get("body_ipos"), &biquat = b.get("body_iquat"), &binert = b.get("body_inertia"), &bbox = b.get("body_box"),
               &bjadr = b.get("body_jntadr"), &bjnum = b.get("body_jntnum"), &bdadr = b.get("body_dofadr"), &bdnum = b.get("body_dofnum");
  const Tensor &jtype = b.get("jnt_type"), &jpos = b.get("jnt_pos"), &jaxis = b.get("jnt_axis"), &jlim = b.get("jnt_limited"),
               &jrange = b.get("jnt_range"), &jstiff = b.get("jnt_stiffness"), &jsolref = b.get("jnt_solref"), &jsolimp = b.get("jnt_solimp"),
               &jqadr = b.get("jnt_qposadr"), &jdadr = b.get("jnt_dofadr"), &jmargin = b.get("jnt_margin");
  const Tensor &qpos0 = b.get("qpos0"), &qspring = b.get("qpos_spring");
  const Tensor &dpar = b.get("dof_parentid"), &ddamp = b.get("dof_damping"), &darm = b.get("dof_armature"), &dinvw = b.get("dof_invweight0"),
               &dM0 = b.get("dof_M0"), &djnt = b.get("dof_jntid"), &dbody = b.get("dof_bodyid");
  const int nb = (int)bpar.count, nv = (int)dpar.count, njnt = (int)jtype.count;
  if (nv != ND + 3) throw std::runtime_error("ball model: expected 105 dofs");
  // ---- identify ball, thorax, halteres
  int ball = -1, thorax = (int)b.get("site").f(0);
  for (int j = 0; j < njnt; j++) if (jtype.i(j) == 1) ball = b.get("jnt_bodyid").i(j);
  if (ball != 1 || jdadr.i(bjadr.i(ball)) != 0) throw std::runtime_error("ball model: the ball joint must come first");
  if (bjnum.i(thorax) != 0 || bpar.i(thorax) != 0) throw std::runtime_error("ball model: thorax must be fixed to the world");
  std::vector<int> nchild(nb, 0);
  for (int i = 1; i < nb; i++) nchild[bpar.i(i)]++;
  std::vector<int> lane_of(nb, -1), halt;
  int nl = 0;
  for (int i = 1; i < nb; i++) {
    if (i == ball || i == thorax) continue;
    if (bjnum.i(i) < 1 || bjnum.i(i) > 3) throw std::runtime_error("ball model: every fly body must carry 1-3 hinges");
    for (int j = bjadr.i(i); j < bjadr.i(i) + bjnum.i(i); j++)
      if (jtype.i(j) != 3) throw std::runtime_error("ball model: hinge joints only");
    if (bpar.i(i) == thorax && nchild[i] == 0 && bdnum.i(i) == 1) { halt.push_back(i); continue; }
    if (nl >= NL) throw std::runtime_error("ball model: more than 64 fly links");
    lane_of[i] = nl++;
  }
  if (nl != NL || halt.size() != 2) throw std::runtime_error("ball model: expected 64 links + 2 halteres");
  double tpos[3], tquat[4];
  for (int k = 0; k < 3; k++) tpos[k] = bpos.f(3 * thorax + k);
  for (int k = 0; k < 4; k++) tquat[k] = bquat.f(4 * thorax + k);
  for (int k = 0; k < 3; k++) M.thorax_pos[k] = (float)tpos[k];
  for (int k = 0; k < 4; k++) M.thorax_quat[k] = (float)tquat[k];
  {
    const Tensor &site = b.get("site");
    double sq[4] = {site.f(4), site.f(5), site.f(6), site.f(7)}, wq[4];
    qmul(tquat, sq, wq);
    for (int k = 0; k < 4; k++) M.site_quat[k] = (float)wq[k];
  }
  // ---- geoms / sites / sensors by body
  const Tensor &gbody = b.get("geom_bodyid"), &gtype = b.get("geom_type"), &gsize = b.get("geom_size"), &gpos = b.get("geom_pos"),
               &gquat = b.get("geom_quat"), &gfric = b.get("geom_friction"), &gmargin = b.get("geom_margin"), &ggap = b.get("geom_gap"),
               &gsolref = b.get("geom_solref"), &gsolimp = b.get("geom_solimp"), &gsolmix = b.get("geom_solmix"), &gcondim = b.get("geom_condim");
  const Tensor &binvw = b.get("body_invweight0");
  int ball_geom = -1;
  for (int g = 0; g < (int)gbody.count; g++) if (gbody.i(g) == ball) ball_geom = g;
  if (ball_geom < 0 || gtype.i(ball_geom) != 2) throw std::runtime_error("ball model: ball sphere geom missing");
  M.b_radius = (float)gsize.f(3 * ball_geom);
  for (int k = 0; k < 3; k++) M.b_center[k] = (float)bpos.f(3 * ball + k);
  M.b_I = (float)binert.f(3 * ball);
  for (int k = 0; k < 4; k++) M.b_iquat[k] = (float)biquat.f(4 * ball + k);
  {
    double box[3] = {bbox.f(3 * ball), bbox.f(3 * ball + 1), bbox.f(3 * ball + 2)};
    ffe::BoxCoef c = ffe::box_coefs(box, rho, beta);
    for (int k = 0; k < 8; k++) M.b_fl[k] = c.c[k];
  }
  // ---- links
  int maxdepth = 0;
  for (int i = 1; i < nb; i++) {
    int l = lane_of[i];
    if (l < 0) continue;
    M.l_body[l] = i;
    int p = bpar.i(i);
    M.l_parent[l] = p == thorax ? -1 : lane_of[p];
    if (p != thorax && lane_of[p] < 0) throw std::runtime_error("ball model: unexpected parent");
    M.l_depth[l] = p == thorax ? 1 : M.l_depth[lane_of[p]] + 1;
    maxdepth = std::max(maxdepth, M.l_depth[l]);
    double pos[3] = {bpos.f(3 * i), bpos.f(3 * i + 1), bpos.f(3 * i + 2)}, quat[4] = {bquat.f(4 * i), bquat.f(4 * i + 1), bquat.f(4 * i + 2), bquat.f(4 * i + 3)};
    if (p == thorax) {  // compose with the fixed thorax pose
      double wp[3], wq[4];
      rot(tquat, pos, wp);
      for (int k = 0; k < 3; k++) pos[k] = tpos[k] + wp[k];
      qmul(tquat, quat, wq);
      std::memcpy(quat, wq, sizeof(wq));
    } else {
      int pl = lane_of[p];
      if (M.l_nchild[pl] >= 3) throw std::runtime_error("ball model: more than 3 children");
      M.l_child[M.l_nchild[pl]++][pl] = l;
    }
    for (int k = 0; k < 3; k++) M.l_pos[k][l] = (float)pos[k];
    for (int k = 0; k < 4; k++) M.l_quat[k][l] = (float)quat[k];
    M.l_mass[l] = (float)bmass.f(i);
    for (int k = 0; k < 3; k++) { M.l_ipos[k][l] = (float)bipos.f(3 * i + k); M.l_inertia[k][l] = (float)binert.f(3 * i + k); }
    for (int k = 0; k < 4; k++) M.l_iquat[k][l] = (float)biquat.f(4 * i + k);
    double box[3] = {bbox.f(3 * i), bbox.f(3 * i + 1), bbox.f(3 * i + 2)};
    ffe::BoxCoef c = ffe::box_coefs(box, rho, beta);
    if (bmass.f(i) < 1e-15) std::memset(&c, 0, sizeof(c));
    for (int k = 0; k < 8; k++) M.l_fl[k][l] = c.c[k];
    M.l_ndof[l] = bdnum.i(i);
    M.l_touch[l] = M.l_force[l] = M.l_adh[l] = -1;
  }
  if (maxdepth > MAXDEPTH) throw std::runtime_error("ball model: tree deeper than expected");
  M.maxdepth = maxdepth;
  // ---- dof slots
  auto fill_slot = [&](int s, int l, int j) {
    int od = jdadr.i(j), f = od - 3, qa = jqadr.i(j);
    M.s_dof[s][l] = f;
    M.s_stiff[s][l] = (float)jstiff.f(j); M.s_sref[s][l] = (float)qspring.f(qa); M.s_damp[s][l] = (float)ddamp.f(od);
    M.s_arm[s][l] = (float)darm.f(od); M.s_lo[s][l] = (float)jrange.f(2 * j); M.s_hi[s][l] = (float)jrange.f(2 * j + 1);
    M.s_limited[s][l] = jlim.i(j); M.s_invw[s][l] = (float)dinvw.f(od);
    if (jmargin.f(j) != 0) throw std::runtime_error("ball model: joint margins are not supported");
    double K, B;
    kb(jsolref.f(2 * j), jsolref.f(2 * j + 1), jsolimp.f(5 * j + 1), h, &K, &B);
    M.s_K[s][l] = (float)K; M.s_B[s][l] = (float)B;
    for (int k = 0; k < 5; k++) M.s_solimp[k][s][l] = (float)jsolimp.f(5 * j + k);
    M.s_act[0][s][l] = M.s_act[1][s][l] = -1;
    M.qpos0[f] = (float)qpos0.f(qa); M.qspring[f] = (float)qspring.f(qa);
    if (qpos0.f(qa) != 0) throw std::runtime_error("ball model: non-zero joint reference");
  };
  for (int l = 0; l < NL; l++) for (int s = 0; s < 4; s++) M.s_dof[s][l] = -1;
  for (int l = 0; l < NL; l++) {
    int i = M.l_body[l];
    for (int s = 0; s < M.l_ndof[l]; s++) {
      int j = bjadr.i(i) + s;
      fill_slot(s, l, j);
      for (int k = 0; k < 3; k++) {
        M.s_axis[k][s][l] = (float)jaxis.f(3 * j + k); M.s_jpos[k][s][l] = (float)jpos.f(3 * j + k);
        if (jpos.f(3 * j + k) != 0.0) throw std::runtime_error("ball model: joints are expected at their body's origin");
      }
    }
  }
  // ---- halteres: closed-form single hinge on the fixed thorax
  int hl[2] = {-1, -1};
  for (int l = 0, k = 0; l < NL && k < 2; l++) if (M.l_ndof[l] <= 2) hl[k++] = l;
  if (hl[1] < 0) throw std::runtime_error("ball model: no free dof slot for the halteres");
  for (int xx = 0; xx < 2; xx++) {
    const int x = hl[xx];
    int i = halt[xx], j = bjadr.i(i), od = jdadr.i(j);
    fill_slot(2, x, j);
    M.x_on[x] = 1;
    double pos[3] = {bpos.f(3 * i), bpos.f(3 * i + 1), bpos.f(3 * i + 2)}, quat[4] = {bquat.f(4 * i), bquat.f(4 * i + 1), bquat.f(4 * i + 2), bquat.f(4 * i + 3)};
    double wq[4], ax_b[3] = {jaxis.f(3 * j), jaxis.f(3 * j + 1), jaxis.f(3 * j + 2)}, jp[3] = {jpos.f(3 * j), jpos.f(3 * j + 1), jpos.f(3 * j + 2)};
    (void)pos;
    qmul(tquat, quat, wq);
    double a[3], r0b[3] = {bipos.f(3 * i) - jp[0], bipos.f(3 * i + 1) - jp[1], bipos.f(3 * i + 2) - jp[2]}, r0[3];
    rot(wq, ax_b, a);
    rot(wq, r0b, r0);
    const double mass = bmass.f(i), F[3] = {0, 0, mass * gz};
    double ar = a[0] * r0[0] + a[1] * r0[1] + a[2] * r0[2], rperp[3], axr[3], t1[3], t2[3];
    for (int k = 0; k < 3; k++) rperp[k] = r0[k] - a[k] * ar;
    cross(a, r0, axr);
    cross(rperp, F, t1);
    cross(axr, F, t2);
    M.x_M[x] = (float)dM0.f(od);  // includes the armature
    M.x_Gc[x] = (float)(a[0] * t1[0] + a[1] * t1[1] + a[2] * t1[2]);
    M.x_Gs[x] = (float)(a[0] * t2[0] + a[1] * t2[1] + a[2] * t2[2]);
    // drag: local (inertial-frame) angular velocity alpha qd, linear velocity at the CoM beta qd
    double iq[4] = {biquat.f(4 * i), biquat.f(4 * i + 1), biquat.f(4 * i + 2), biquat.f(4 * i + 3)}, im[9];
    q2m(iq, im);
    double vb[3], al[3], be[3];
    cross(ax_b, r0b, vb);
    for (int k = 0; k < 3; k++) {
      al[k] = im[k] * ax_b[0] + im[3 + k] * ax_b[1] + im[6 + k] * ax_b[2];
      be[k] = im[k] * vb[0] + im[3 + k] * vb[1] + im[6 + k] * vb[2];
    }
    double box[3] = {bbox.f(3 * i), bbox.f(3 * i + 1), bbox.f(3 * i + 2)};
    ffe::BoxCoef c = ffe::box_coefs(box, rho, beta);
    double cv = 0, cq = 0;
    for (int k = 0; k < 3; k++) {
      cv += c.c[0] * al[k] * al[k] + c.c[1] * be[k] * be[k];
      cq += c.c[5 + k] * std::fabs(al[k]) * al[k] * al[k] + c.c[2 + k] * std::fabs(be[k]) * be[k] * be[k];
    }
    M.x_cv[x] = (float)cv; M.x_cq[x] = (float)cq;
  }
  // ---- joint-space inertia structure over fly dofs
  std::vector<int> par(ND), madr(ND), blk(ND), li(ND), depth(ND);
  std::vector<unsigned> amask(ND);
  int adr = 0, nblk = 0;
  std::vector<int> blk_start;
  for (int f = 0; f < ND; f++) {
    int p = dpar.i(f + 3);
    par[f] = p < 0 ? -1 : p - 3;
    if (p >= 0 && p < 3) throw std::runtime_error("ball model: fly dof parented to the ball");
    madr[f] = adr;
    depth[f] = par[f] < 0 ? 1 : depth[par[f]] + 1;
    adr += depth[f];
    if (par[f] < 0) { blk_start.push_back(f); nblk++; }
    blk[f] = nblk - 1;
    li[f] = f - blk_start.back();
    if (par[f] >= 0 && blk[par[f]] != blk[f]) throw std::runtime_error("ball model: block structure");
    amask[f] = (1u << li[f]) | (par[f] < 0 ? 0u : amask[par[f]]);
    if (li[f] >= NSTEP) throw std::runtime_error("ball model: block larger than 14 dofs");
    M.d_parent[f] = (short)par[f]; M.d_madr[f] = (short)madr[f]; M.d_blk[f] = (short)blk[f]; M.d_li[f] = (short)li[f];
    M.d_amask[f] = (unsigned short)amask[f]; M.d_arm[f] = (float)darm.f(f + 3);
  }
  if (nblk > NBLK || adr > NMMAX) throw std::runtime_error("ball model: inertia structure exceeds capacities");
  for (int l = 0; l < NL; l++) if (M.x_on[l]) M.d_arm[M.s_dof[2][l]] = 0.f;  // x_M already holds the haltere armature
  M.nM = adr; M.nblk = nblk;
  blk_start.push_back(ND);
  // entries of M with the elimination steps that touch them
  struct Ent { int i, j, adr, work; unsigned short fmask; int rowstep, colstep; };
  std::vector<Ent> ents;
  for (int i = 0; i < ND; i++) {
    int j = i, a = madr[i];
    while (j >= 0) {
      Ent e{i, j, a, 0, 0, 0, 0};
      int bb = blk[i], n = blk_start[bb + 1] - blk_start[bb];
      for (int s = 0; s < n; s++) {
        int k = blk_start[bb + 1] - 1 - s;
        if (k != i && (amask[k] >> li[i]) & 1u) { e.fmask |= (unsigned short)(1u << s); e.work++; }
        if (k == i) e.rowstep = s;
      }
      e.colstep = li[j];
      ents.push_back(e);
      j = par[j]; a++;
    }
  }
  // ---- schedules
  {
    auto anc_off = [&](int k, int i) { return (int)__builtin_popcount(amask[k] & ~((2u << li[i]) - 1u)); };
    int slot = 0;
    for (int s = 0; s < NSTEP; s++) {  // factor: pivots of step s, every (i, j) with i a proper ancestor of the pivot
      int fill = 0;
      for (const Ent &e : ents) {
        if (!((e.fmask >> s) & 1u)) continue;
        int bb = blk[e.i], k = blk_start[bb + 1] - 1 - s;
        if (fill == 0 && slot >= NFS) throw std::runtime_error("ball model: factor schedule too long");
        M.fac_a[slot][fill] = (unsigned)e.adr | ((unsigned)madr[k] << 10) | ((unsigned)(madr[k] + anc_off(k, e.i)) << 20) | 0x80000000u;
        M.fac_b[slot][fill] = (unsigned)(madr[k] + anc_off(k, e.j));
        if (++fill == NL) { fill = 0; slot++; }
      }
      if (fill) slot++;
    }
    M.nfs = slot;
    if (slot + 8 > NFS) throw std::runtime_error("ball model: factor schedule needs padding room");
    slot = 0;
    for (int s = 0; s < NSTEP; s++) {  // pass 1: rows leaf -> root
      int fill = 0;
      for (const Ent &e : ents) {
        if (e.i == e.j || e.rowstep != s) continue;
        if (fill == 0 && slot >= NPS) throw std::runtime_error("ball model: solve schedule too long");
        M.p1[slot][fill] = (unsigned)e.adr | ((unsigned)e.i << 10) | ((unsigned)e.j << 17) | 0x80000000u;
        if (++fill == NL) { fill = 0; slot++; }
      }
      if (fill) slot++;
    }
    M.np1 = slot;
    if (slot + 8 > NPS) throw std::runtime_error("ball model: solve schedule needs padding room");
    slot = 0;
    for (int r = 0; r < NSTEP; r++) {  // pass 2: columns root -> leaf
      int fill = 0;
      for (const Ent &e : ents) {
        if (e.i == e.j || e.colstep != r) continue;
        if (fill == 0 && slot >= NPS) throw std::runtime_error("ball model: solve schedule too long");
        M.p2[slot][fill] = (unsigned)e.adr | ((unsigned)e.i << 10) | ((unsigned)e.j << 17) | 0x80000000u;
        if (++fill == NL) { fill = 0; slot++; }
      }
      if (fill) slot++;
    }
    M.np2 = slot;
    if (slot + 8 > NPS) throw std::runtime_error("ball model: solve schedule needs padding room");
    int k = 0;
    for (const Ent &e : ents) {
      int t = k / NL, l = k % NL;
      M.ent_a[t][l] = (unsigned)e.i | ((unsigned)e.j << 8) | ((unsigned)e.adr << 16) | 0x80000000u;
      M.ent_b[t][l] = (unsigned)blk[e.i] | ((unsigned)li[e.i] << 4) | ((unsigned)li[e.j] << 8);
      k++;
    }
    if ((int)ents.size() > ECAP * NL) throw std::runtime_error("ball model: too many M entries");
  }
  // ---- chains (fly dofs from the chain root to each link's last dof), capsules, sensors
  for (int l = 0; l < NL; l++) {
    int last = M.s_dof[M.l_ndof[l] - 1][l];
    std::vector<int> ch;
    for (int f = last; f >= 0; f = par[f]) ch.push_back(f);
    std::reverse(ch.begin(), ch.end());
    if ((int)ch.size() > NCH) throw std::runtime_error("ball model: chain too long");
    M.l_nchain[l] = (int)ch.size();
    for (int k = 0; k < NCH; k++) M.l_chain[k][l] = k < (int)ch.size() ? ch[k] : -1;
  }
  const double bfric = gfric.f(3 * ball_geom), bmix = gsolmix.f(ball_geom);
  for (int g = 0; g < (int)gbody.count; g++) {
    int l = lane_of[gbody.i(g)];
    if (l < 0) continue;
    int ty = gtype.i(g);
    if (ty != 2 && ty != 3) continue;  // ellipsoids / cylinders cannot reach the ball
    if (M.g_has[l]) continue;  // a second capsule on the link (rostrum): fly-fly collision only, it cannot reach the ball
    M.g_has[l] = 1;
    double q[4] = {gquat.f(4 * g), gquat.f(4 * g + 1), gquat.f(4 * g + 2), gquat.f(4 * g + 3)}, mm[9];
    q2m(q, mm);
    for (int k = 0; k < 3; k++) { M.g_pos[k][l] = (float)gpos.f(3 * g + k); M.g_axis[k][l] = (float)mm[3 * k + 2]; }
    M.g_rad[l] = (float)gsize.f(3 * g); M.g_half[l] = ty == 3 ? (float)gsize.f(3 * g + 1) : 0.f;
    M.g_margin[l] = (float)std::max(gmargin.f(g), gmargin.f(ball_geom)); M.g_gap[l] = (float)std::max(ggap.f(g), ggap.f(ball_geom));
    M.g_fric[l] = (float)std::max(gfric.f(3 * g), bfric);
    if (std::max(gcondim.i(g), gcondim.i(ball_geom)) != 3) throw std::runtime_error("ball model: ball contacts must be condim 3");
    double mix = gsolmix.f(g) / (gsolmix.f(g) + bmix), sr[2], si[5];  // mj: mj_contactParam, geom1 = ball
    mix = 1.0 - mix;  // weight of the ball (geom1)
    for (int k = 0; k < 2; k++) sr[k] = mix * gsolref.f(2 * ball_geom + k) + (1 - mix) * gsolref.f(2 * g + k);
    for (int k = 0; k < 5; k++) si[k] = mix * gsolimp.f(5 * ball_geom + k) + (1 - mix) * gsolimp.f(5 * g + k);
    double K, B;
    kb(sr[0], sr[1], si[1], h, &K, &B);
    M.g_K[l] = (float)K; M.g_B[l] = (float)B;
    for (int k = 0; k < 5; k++) M.g_solimp[k][l] = (float)si[k];
    M.g_invw[l] = (float)(binvw.f(2 * gbody.i(g)) + binvw.f(2 * ball));
  }
  const Tensor &sbody = b.get("sites_bodyid"), &squat = b.get("sites_quat"), &spos = b.get("sites_pos"), &tsite = b.get("touch_site"),
               &fsite = b.get("force_site"), &asite = b.get("appendage_site");
  for (int t = 0; t < (int)tsite.count; t++) M.l_touch[lane_of[sbody.i(tsite.i(t))]] = t;
  for (int t = 0; t < (int)fsite.count; t++) {
    int s = fsite.i(t), l = lane_of[sbody.i(s)];
    M.l_force[l] = t;
    for (int k = 0; k < 4; k++) M.l_fsite[k][l] = (float)squat.f(4 * s + k);
  }
  if (asite.count > 8) throw std::runtime_error("ball model: too many appendage sites");
  for (int t = 0; t < (int)asite.count; t++) {
    int s = asite.i(t);
    M.app_link[t] = lane_of[sbody.i(s)];
    for (int k = 0; k < 3; k++) M.app_pos[t][k] = (float)spos.f(3 * s + k);
  }
  // ---- actuators
  const Tensor &atrn = b.get("act_trntype"), &atid = b.get("act_trnid"), &agear = b.get("act_gear"), &again = b.get("act_gainprm"),
               &abias = b.get("act_biasprm"), &acl = b.get("act_ctrllimited"), &acr = b.get("act_ctrlrange"), &afl = b.get("act_forcelimited"),
               &afr = b.get("act_forcerange"), &adyn = b.get("act_dyntype"), &adp = b.get("act_dynprm"), &aact = b.get("act_action");
  const Tensor &tadr = b.get("ten_adr"), &tnum = b.get("ten_num"), &wdof = b.get("wrap_dof"), &wcoef = b.get("wrap_coef");
  const int nu = (int)atrn.count;
  if (nu != NU) throw std::runtime_error("ball model: expected 59 actuators");
  std::vector<int> slot_lane(ND, -1), slot_idx(ND, -1);
  for (int l = 0; l < NL; l++) for (int s = 0; s < 4; s++) if (M.s_dof[s][l] >= 0) { slot_lane[M.s_dof[s][l]] = l; slot_idx[M.s_dof[s][l]] = s; }
  for (int a = 0; a < nu; a++) {
    if (adyn.i(a) != 1) throw std::runtime_error("ball model: every actuator is expected to be filtered");
    if (agear.f(a) != 1.0) throw std::runtime_error("ball model: gear != 1");
    M.a_trn[a] = atrn.i(a); M.a_gain[a] = (float)again.f(a);
    M.a_b0[a] = (float)abias.f(3 * a); M.a_b1[a] = (float)abias.f(3 * a + 1); M.a_b2[a] = (float)abias.f(3 * a + 2);
    M.a_climited[a] = acl.i(a); M.a_clo[a] = (float)acr.f(2 * a); M.a_chi[a] = (float)acr.f(2 * a + 1);
    M.a_flimited[a] = afl.i(a); M.a_flo[a] = (float)afr.f(2 * a); M.a_fhi[a] = (float)afr.f(2 * a + 1);
    M.a_tau[a] = (float)std::max(1e-15, adp.f(a)); M.a_action[a] = aact.i(a); M.a_link[a] = -1;
    auto bind = [&](int od, double coef, int which) {
      int f = od - 3, l = slot_lane[f], s = slot_idx[f];
      if (f < 0 || l < 0) throw std::runtime_error("ball model: actuator on an unknown dof");
      if (M.s_act[which][s][l] >= 0) throw std::runtime_error("ball model: two actuators of a kind on one dof");
      M.s_act[which][s][l] = a; M.s_actcoef[which][s][l] = (float)coef;
      if (M.a_nwrap[a] >= NWRAP) throw std::runtime_error("ball model: tendon too long");
      M.a_wdof[M.a_nwrap[a]][a] = f; M.a_wcoef[M.a_nwrap[a]][a] = (float)coef; M.a_nwrap[a]++;
    };
    if (atrn.i(a) == 0) bind(jdadr.i(atid.i(a)), 1.0, 0);
    else if (atrn.i(a) == 1) { int t = atid.i(a); for (int w = tadr.i(t); w < tadr.i(t) + tnum.i(t); w++) bind(wdof.i(w), wcoef.f(w), 1); }
    else { int l = lane_of[atid.i(a)]; if (l < 0) throw std::runtime_error("ball model: adhesion body"); M.a_link[a] = l; M.l_adh[l] = a; }
  }
  const Tensor &amin = b.get("action_min"), &amax = b.get("action_max");
  H.action_min.resize(amin.count); H.action_max.resize(amax.count);
  for (size_t k = 0; k < amin.count; k++) { H.action_min[k] = (float)amin.f(k); H.action_max[k] = (float)amax.f(k); M.act_lo[k] = H.action_min[k]; M.act_hi[k] = H.action_max[k]; }
  const Tensor &oj = b.get("obs_jnt");
  if ((int)oj.count != NOBSJ) throw std::runtime_error("ball model: expected 85 observable joints");
  for (int k = 0; k < NOBSJ; k++) M.obs_dof[k] = jdadr.i(oj.i(k)) - 3;
  const Tensor &wj = b.get("wing_jnt");
  M.nwing = (int)wj.count;
  for (int k = 0; k < M.nwing && k < 8; k++) M.wing_dof[k] = jdadr.i(wj.i(k)) - 3;
  double mi = 0;
  for (int k = 0; k < nv; k++) mi += dM0.f(k) / nv;
  M.meaninertia = (float)mi;
  const Tensor &so = b.get("solver_opt");
  if ((int)so.f(0) != 1) throw std::runtime_error("ball model: elliptic cones expected");
  if (so.f(2) != 1.0) throw std::runtime_error("ball model: impratio != 1 unsupported");
  M.noslip_iterations = (int)so.f(1);
  for (int k = 0; k < 5; k++) { M.j_solimp[k] = (float)jsolimp.f(k); M.c_solimp[k] = -1.f; }
  for (int j = 0; j < njnt; j++) for (int k = 0; k < 5; k++)
    if (jtype.i(j) == 3 && jsolimp.f(5 * j + k) != jsolimp.f(k)) throw std::runtime_error("ball model: joint solimp is expected to be uniform");
  for (int l = 0; l < NL; l++) {
    if (!M.g_has[l]) continue;
    for (int k = 0; k < 5; k++) {
      if (M.c_solimp[k] < 0) M.c_solimp[k] = M.g_solimp[k][l];
      else if (M.c_solimp[k] != M.g_solimp[k][l]) throw std::runtime_error("ball model: contact solimp is expected to be uniform");
    }
  }
  M.maxsub = 1;
  for (int l = 0; l < NL; l++) {
    auto up = [&](int x, int n) { for (int k = 0; k < n && x >= 0; k++) x = M.l_parent[x]; return x; };
    int size = 1;  // depth-first numbering: the subtree is the run of following links that are deeper than l
    while (l + size < NL && M.l_body[l + size] > 0 && M.l_depth[l + size] > M.l_depth[l]) size++;
    for (int k = l + 1; k < NL; k++) {  // check: exactly the links of that run have l among their ancestors
      bool desc = false;
      for (int x = M.l_parent[k]; x >= 0; x = M.l_parent[x]) desc |= (x == l);
      if (M.l_body[l] > 0 && M.l_body[k] > 0 && desc != (k < l + size)) throw std::runtime_error("ball model: links are not in depth-first order");
    }
    if (M.l_body[l] <= 0) size = 1;
    M.maxsub = std::max(M.maxsub, size);
    M.l_tree[l] = (unsigned)size | ((unsigned)(up(l, 2) + 1) << 8) | ((unsigned)(up(l, 4) + 1) << 16);
    M.l_pack[l] = (unsigned)(M.l_parent[l] + 1) | ((unsigned)M.l_depth[l] << 8) | ((unsigned)M.l_ndof[l] << 12);
    M.l_kids[l] = (unsigned)M.l_child[0][l] | ((unsigned)M.l_child[1][l] << 8) | ((unsigned)M.l_child[2][l] << 16) | ((unsigned)M.l_nchild[l] << 24);
  }
  // ---- fly-fly candidate pairs over the primitive geoms, filtered as mj_collision filters them
  {
    const Tensor &gct = b.get("geom_contype"), &gca = b.get("geom_conaffinity"), &excl = b.get("exclude_pairs"), &weld = b.get("body_weldid");
    std::vector<int> slot_of((size_t)gbody.count, -1);
    for (int l = 0; l < NL; l++) M.g_slot[l] = -1;
    M.pg2_lane = M.pg2_slot = -1;
    int npg = 0;
    double sref[2] = {0, 0}, simp[5] = {0, 0, 0, 0, 0};
    for (int g = 0; g < (int)gbody.count; g++) {
      const int l = lane_of[gbody.i(g)], ty = gtype.i(g);
      if (l < 0 || (ty != 2 && ty != 3)) continue;
      if (npg >= NPG) throw std::runtime_error("ball model: more primitive geoms than slots");
      slot_of[g] = npg; M.pgs_link[npg] = l;
      double q[4] = {gquat.f(4 * g), gquat.f(4 * g + 1), gquat.f(4 * g + 2), gquat.f(4 * g + 3)}, mm[9];
      q2m(q, mm);
      if (M.g_slot[l] < 0) {  // the link's first primitive geom is the g_* capsule filled above (same geom order)
        M.g_slot[l] = npg;
        if (!M.g_has[l] || M.g_rad[l] != (float)gsize.f(3 * g) || M.g_pos[0][l] != (float)gpos.f(3 * g)) throw std::runtime_error("ball model: geom tables out of step");
      } else {
        if (M.pg2_lane >= 0) throw std::runtime_error("ball model: one link with two primitive geoms expected");
        M.pg2_lane = l; M.pg2_slot = npg;
        for (int k = 0; k < 3; k++) { M.pg2_pos[k] = (float)gpos.f(3 * g + k); M.pg2_axis[k] = ty == 3 ? (float)mm[3 * k + 2] : (k == 2 ? 1.f : 0.f); }
        M.pg2_rad = (float)gsize.f(3 * g); M.pg2_half = ty == 3 ? (float)gsize.f(3 * g + 1) : 0.f;
      }
      M.pgs_invw[npg] = (float)binvw.f(2 * gbody.i(g));
      if (gcondim.i(g) != 1) throw std::runtime_error("ball model: fly geoms are expected to be condim 1");
      if (npg == 0) { for (int k = 0; k < 2; k++) sref[k] = gsolref.f(2 * g + k); for (int k = 0; k < 5; k++) simp[k] = gsolimp.f(5 * g + k); }
      for (int k = 0; k < 2; k++) if (gsolref.f(2 * g + k) != sref[k]) throw std::runtime_error("ball model: fly geom solref is expected to be uniform");
      for (int k = 0; k < 5; k++) if (gsolimp.f(5 * g + k) != simp[k]) throw std::runtime_error("ball model: fly geom solimp is expected to be uniform");
      if (gmargin.f(g) != 0) {
        if (M.sc_margin != 0 && (M.sc_margin != (float)gmargin.f(g) || M.sc_gap != (float)ggap.f(g))) throw std::runtime_error("ball model: one margin class expected");
        M.sc_margin = (float)gmargin.f(g); M.sc_gap = (float)ggap.f(g);
      }
      npg++;
    }
    M.npg = npg;
    double K, B;
    kb(sref[0], sref[1], simp[1], h, &K, &B);  // mj_contactParam: equal solmix weights of equal parameters
    M.sc_K = (float)K; M.sc_B = (float)B;
    for (int k = 0; k < 5; k++) M.sc_solimp[k] = (float)simp[k];
    M.sp_sphere = -1;
    for (int g = 0; g < (int)gbody.count; g++) {
      if (slot_of[g] < 0) continue;
      if (gmargin.f(g) != 0) M.sp_claw |= 1ull << slot_of[g];
      if (gtype.i(g) == 2) { if (M.sp_sphere >= 0) throw std::runtime_error("ball model: one sphere expected among the fly geoms"); M.sp_sphere = slot_of[g]; }
    }
    if (npg != NPG) throw std::runtime_error("ball model: the pair enumeration expects exactly NPG primitive geoms");
    const int nex = (int)excl.count / 2;
    for (int b1 = 0; b1 < nb; b1++) for (int b2 = b1 + 1; b2 < nb; b2++) {  // body-pair major order, as the oracle (and mj_collision) walk them
      if (b1 == ball || b2 == ball) continue;
      const int w1 = weld.i(b1), w2 = weld.i(b2);
      if (w1 == w2) continue;
      const int wp1 = weld.i(bpar.i(w1)), wp2 = weld.i(bpar.i(w2));
      if (w1 != 0 && w2 != 0 && (w1 == wp2 || w2 == wp1)) continue;
      bool skip = false;
      for (int e = 0; e < nex; e++) skip |= (excl.i(2 * e) == b1 && excl.i(2 * e + 1) == b2) || (excl.i(2 * e) == b2 && excl.i(2 * e + 1) == b1);
      if (skip) continue;
      for (int ga = 0; ga < (int)gbody.count; ga++) {
        if (gbody.i(ga) != b1 || slot_of[ga] < 0) continue;
        for (int gb = 0; gb < (int)gbody.count; gb++) {
          if (gbody.i(gb) != b2 || slot_of[gb] < 0) continue;
          if (!((gct.i(ga) & gca.i(gb)) || (gct.i(gb) & gca.i(ga)))) continue;
          int g1 = ga, g2 = gb;
          if (gtype.i(g1) > gtype.i(g2)) std::swap(g1, g2);  // mj: lower type code first
          const int s1 = slot_of[g1], s2 = slot_of[g2];
          // the kernel derives the order instead of storing it
          const int e1 = (s1 == M.sp_sphere || s2 == M.sp_sphere) ? M.sp_sphere : std::min(s1, s2);
          if (e1 != s1) throw std::runtime_error("ball model: unexpected geom order of a fly-fly pair");
          M.sp_mask[s1] |= 1ull << s2; M.sp_mask[s2] |= 1ull << s1;
          M.nsp++;
        }
      }
    }
  }
  // ---- convex pairs: geom tables, the static candidate list, the ball's convex partners
  {
    const Tensor &cg1 = b.get("cand_g1"), &cg2 = b.get("cand_g2");
    const int ng = (int)gbody.count;
    if (ng - 1 > NG) throw std::runtime_error("ball model: more collision geoms than the geom table holds");
    std::vector<int> fg((size_t)ng, -1);  // model geom -> fly geom index
    int n = 0;
    for (int g = 0; g < ng; g++) {
      if (g == ball_geom) continue;
      const int body = gbody.i(g), l = lane_of[body], ty = gtype.i(g);
      if (l < 0 && body != thorax) throw std::runtime_error("ball model: collision geom on an unexpected body");
      if (ty < 2 || ty > 5) throw std::runtime_error("ball model: unexpected geom type");
      fg[g] = n;
      M.cg_link[n] = l; M.cg_type[n] = ty;
      double pos[3] = {gpos.f(3 * g), gpos.f(3 * g + 1), gpos.f(3 * g + 2)}, quat[4] = {gquat.f(4 * g), gquat.f(4 * g + 1), gquat.f(4 * g + 2), gquat.f(4 * g + 3)};
      if (l < 0) {  // thorax: compose with its fixed pose
        double wp[3], wq[4];
        rot(tquat, pos, wp);
        for (int k = 0; k < 3; k++) pos[k] = tpos[k] + wp[k];
        qmul(tquat, quat, wq);
        std::memcpy(quat, wq, sizeof(wq));
      }
      for (int k = 0; k < 3; k++) { M.cg_pos[k][n] = (float)pos[k]; M.cg_size[k][n] = (float)gsize.f(3 * g + k); }
      for (int k = 0; k < 4; k++) M.cg_quat[k][n] = (float)quat[k];
      const double s0 = gsize.f(3 * g), s1 = gsize.f(3 * g + 1), s2 = gsize.f(3 * g + 2);
      M.cg_brad[n] = (float)(ty == 2 ? s0 : ty == 3 ? s0 + s1 : ty == 5 ? std::sqrt(s0 * s0 + s1 * s1) : std::max(s0, std::max(s1, s2)));
      M.cg_invw[n] = (float)binvw.f(2 * body);
      if (gmargin.f(g) != 0) {
        if ((float)gmargin.f(g) != M.sc_margin || (float)ggap.f(g) != M.sc_gap) throw std::runtime_error("ball model: one margin class expected");
        M.cg_mmask[n >> 6] |= 1ull << (n & 63);
      }
      if (gcondim.i(g) != 1) throw std::runtime_error("ball model: fly geoms are expected to be condim 1");
      {  // solref / solimp uniform over the fly's geoms (mj_contactParam then mixes equal values): the primitive ones set sc_K, sc_B
        double K, B;
        kb(gsolref.f(2 * g), gsolref.f(2 * g + 1), gsolimp.f(5 * g + 1), h, &K, &B);
        if ((float)K != M.sc_K || (float)B != M.sc_B) throw std::runtime_error("ball model: fly geom solref is expected to be uniform");
        for (int q = 0; q < 5; q++) if ((float)gsolimp.f(5 * g + q) != M.sc_solimp[q]) throw std::runtime_error("ball model: fly geom solimp is expected to be uniform");
      }
      n++;
    }
    M.ncg = n;
    for (int k = 0; k < NCP; k++) M.cp_pair[k] = 0xffffu;
    const double bfric2 = gfric.f(3 * ball_geom);
    for (size_t k = 0; k < cg1.count; k++) {
      int g1 = cg1.i(k), g2 = cg2.i(k);
      if (gtype.i(g1) > gtype.i(g2)) std::swap(g1, g2);
      const bool convex = gtype.i(g1) >= 4 || gtype.i(g2) >= 4;
      if (!convex) continue;
      if (g1 == ball_geom || g2 == ball_geom) {
        const int g = g1 == ball_geom ? g2 : g1;
        if (M.nxb >= NXB) throw std::runtime_error("ball model: more convex partners of the ball than expected");
        if (lane_of[gbody.i(g)] < 0) throw std::runtime_error("ball model: a thorax geom among the ball's partners");
        const int x = M.nxb++;
        M.xb_geom[x] = fg[g];
        M.xb_margin[x] = (float)std::max(gmargin.f(g), gmargin.f(ball_geom)); M.xb_gap[x] = (float)std::max(ggap.f(g), ggap.f(ball_geom));
        M.xb_fric[x] = (float)std::max(gfric.f(3 * g), bfric2);
        if (std::max(gcondim.i(g), gcondim.i(ball_geom)) != 3) throw std::runtime_error("ball model: ball contacts must be condim 3");
        double mix = 1.0 - gsolmix.f(g) / (gsolmix.f(g) + gsolmix.f(ball_geom)), sr[2], si[5];
        for (int q = 0; q < 2; q++) sr[q] = mix * gsolref.f(2 * ball_geom + q) + (1 - mix) * gsolref.f(2 * g + q);
        for (int q = 0; q < 5; q++) si[q] = mix * gsolimp.f(5 * ball_geom + q) + (1 - mix) * gsolimp.f(5 * g + q);
        double K, B;
        kb(sr[0], sr[1], si[1], h, &K, &B);
        M.xb_K[x] = (float)K; M.xb_B[x] = (float)B;
        for (int q = 0; q < 5; q++) if ((float)si[q] != M.c_solimp[q]) throw std::runtime_error("ball model: contact solimp is expected to be uniform");
        M.xb_invw[x] = (float)(binvw.f(2 * gbody.i(g)) + binvw.f(2 * ball));
        continue;
      }
      if (M.ncp >= NCP) throw std::runtime_error("ball model: more convex candidate pairs than the list holds");
      M.cp_pair[M.ncp++] = (unsigned short)(fg[g1] | (fg[g2] << 8));
    }
  }
  (void)dbody; (void)djnt; (void)gcondim;
  return H;
}

}  // namespace ffb
