// Host build of the kernels' float32 narrow phase (flybody_amd/csrc/convex.hpp compiled with -DCVX_HOST) for
// tests/test_convex_f32_cpu.py: test infrastructure only, never linked into the product library.
#define CVX_HOST 1
#ifdef CVX_DEBUG
#define CVX_TRACE 1
#endif
#include "../flybody_amd/csrc/convex.hpp"

using namespace cvx;

static Geom mk(const float *p) { return Geom{{p[0], p[1], p[2]}, {p[3], p[4], p[5], p[6]}, p[7], p[8], p[9], (int)p[10]}; }

extern "C" {
#ifdef CVX_DEBUG
int cvxh_trace(float *out) { for (int k = 0; k < cvx::g_ntrace; k++) out[k] = cvx::g_trace[k]; int n = cvx::g_ntrace; cvx::g_ntrace = 0; return n; }
#endif
// geom = centre[3], quat[4], size[3], type
float cvxh_sdf(const float *g, const float *x, float *grad, float *H6) {
  V3 gr; Sym3 H;
  const float f = sdf<true>(mk(g), V3{x[0], x[1], x[2]}, gr, H);
  grad[0] = gr.x; grad[1] = gr.y; grad[2] = gr.z;
  H6[0] = H.xx; H6[1] = H.yy; H6[2] = H.zz; H6[3] = H.xy; H6[4] = H.xz; H6[5] = H.yz;
  return f;
}
void cvxh_support(const float *g, const float *n, float *out) {
  const V3 s = support(mk(g), V3{n[0], n[1], n[2]});
  out[0] = s.x; out[1] = s.y; out[2] = s.z;
}
float cvxh_distance(const float *g1, const float *g2, const float *n0, int have_n_, const float *x0, int have_x_, float sgap, float cull, int outer, int inner, float *nrm, float *pos) {
  Result r;
  const V3 n = {n0[0], n0[1], n0[2]};
  const bool have_n = have_n_ != 0, have_x = have_x_ != 0;
  const V3 xw = {x0[0], x0[1], x0[2]};
  if (outer == 1 && inner == 3) r = distance<1, 3>(mk(g1), mk(g2), n, have_n, sgap, cull, xw, have_x);
  else if (outer == 1 && inner == 2) r = distance<1, 2>(mk(g1), mk(g2), n, have_n, sgap, cull, xw, have_x);
  else if (outer == 2 && inner == 2) r = distance<2, 2>(mk(g1), mk(g2), n, have_n, sgap, cull, xw, have_x);
  else if (outer == 2 && inner == 3) r = distance<2, 3>(mk(g1), mk(g2), n, have_n, sgap, cull, xw, have_x);
  else if (outer == 3 && inner == 3) r = distance<3, 3>(mk(g1), mk(g2), n, have_n, sgap, cull, xw, have_x);
  else if (outer == 4 && inner == 3) r = distance<4, 3>(mk(g1), mk(g2), n, have_n, sgap, cull, xw, have_x);
  else if (outer == 6 && inner == 3) r = distance<6, 3>(mk(g1), mk(g2), n, have_n, sgap, cull, xw, have_x);
  else if (outer == 6 && inner == 4) r = distance<6, 4>(mk(g1), mk(g2), n, have_n, sgap, cull, xw, have_x);
  else r = distance<10, 5>(mk(g1), mk(g2), n, have_n, sgap, cull, xw, have_x);
  nrm[0] = r.n.x; nrm[1] = r.n.y; nrm[2] = r.n.z; pos[0] = r.pos.x; pos[1] = r.pos.y; pos[2] = r.pos.z;
  return r.dist;
}
// class-specialised routines: capsule / sphere against ellipsoid / cylinder, ellipsoid against ellipsoid
float cvxh_prim_convex(const float *g1, const float *g2, int iters, float *nrm, float *pos, int *shallow) {
  PResult r;
  if (iters == 3) r = prim_convex<3>(mk(g1), mk(g2));
  else if (iters == 4) r = prim_convex<4>(mk(g1), mk(g2));
  else if (iters == 5) r = prim_convex<5>(mk(g1), mk(g2));
  else if (iters == 6) r = prim_convex<6>(mk(g1), mk(g2));
  else r = prim_convex<12>(mk(g1), mk(g2));
  nrm[0] = r.n.x; nrm[1] = r.n.y; nrm[2] = r.n.z; pos[0] = r.pos.x; pos[1] = r.pos.y; pos[2] = r.pos.z;
  *shallow = r.shallow ? 1 : 0;
  return r.dist;
}
float cvxh_ell_ell(const float *g1, const float *g2, const float *n0, int have_n, int iters, float *nrm, float *pos) {
  Result r;
  const V3 n = {n0[0], n0[1], n0[2]};
  if (iters == 3) r = ell_ell<3>(mk(g1), mk(g2), n, have_n != 0);
  else if (iters == 4) r = ell_ell<4>(mk(g1), mk(g2), n, have_n != 0);
  else if (iters == 6) r = ell_ell<6>(mk(g1), mk(g2), n, have_n != 0);
  else if (iters == 8) r = ell_ell<8>(mk(g1), mk(g2), n, have_n != 0);
  else r = ell_ell<16>(mk(g1), mk(g2), n, have_n != 0);
  nrm[0] = r.n.x; nrm[1] = r.n.y; nrm[2] = r.n.z; pos[0] = r.pos.x; pos[1] = r.pos.y; pos[2] = r.pos.z;
  return r.dist;
}
float cvxh_ell_cyl(const float *g1, const float *g2, int iters, float *nrm, float *pos) {
  Result r;
  if (iters == 3) r = ell_cyl<3>(mk(g1), mk(g2));
  else if (iters == 4) r = ell_cyl<4>(mk(g1), mk(g2));
  else if (iters == 6) r = ell_cyl<6>(mk(g1), mk(g2));
  else if (iters == 8) r = ell_cyl<8>(mk(g1), mk(g2));
  else r = ell_cyl<16>(mk(g1), mk(g2));
  nrm[0] = r.n.x; nrm[1] = r.n.y; nrm[2] = r.n.z; pos[0] = r.pos.x; pos[1] = r.pos.y; pos[2] = r.pos.z;
  return r.dist;
}
// the kernels' entry points: the pair dispatcher and the broad phase's separating-direction bound
float cvxh_collide(const float *g1, const float *g2, float *nrm, float *pos) {
  const Contact r = collide(mk(g1), mk(g2));
  nrm[0] = r.n.x; nrm[1] = r.n.y; nrm[2] = r.n.z; pos[0] = r.pos.x; pos[1] = r.pos.y; pos[2] = r.pos.z;
  return r.dist;
}
float cvxh_separation_bound(const float *g1, const float *g2) { return separation_bound(mk(g1), mk(g2)); }
float cvxh_overlap(const float *g1, const float *g2, const float *u) { return overlap(mk(g1), mk(g2), V3{u[0], u[1], u[2]}); }
}
