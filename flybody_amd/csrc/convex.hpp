// convex.hpp - float32 narrow phase for the fly's convex geom pairs (mj: mjc_Convex), shared by the step kernels.
//
// MuJoCo gives every pair with an ellipsoid or a cylinder on one side (thorax, head, coxae, wings, abdomen segments against
// anything; ref: fruitfly/assets/fruitfly.xml:323-443) to its general convex routine, which returns ONE contact: depth, the
// direction of least penetration, a point between the two witness points.  What that iterative routine converges to is the
// minimum-translation problem
//     dist = - min over unit n of o(n),   o(n) = h_1(n) + h_2(-n) - n . (c_2 - c_1)   (overlap of the two geoms along n)
// and this file solves it as the majorisation the float64 CPU checker of the tests also uses: for the current n
// geom2 is moved out along n until the pair is separated by `sgap`, the closest points of the disjoint pair are found by
// Newton's method on G(x) = 1/2 d_1(x)^2 + 1/2 d_2(x)^2 (d = distance to the geom; minimiser = midpoint of the closest pair;
// Hessian d H + g g' is bounded and regular), and n becomes the direction between them.  o(n) never increases; the fixed
// point has the witness points facing each other along n.  One lane works on one pair, everything in registers.
//
// The pair classes that dominate the fly's contacts have cheaper exact formulations (further down: `prim_convex`, `ell_ell`,
// `ell_cyl`); `collide` picks per pair.
// The header also compiles for the host (-DCVX_HOST: tests/test_convex_f32_cpu.py runs this float32 code without a GPU);
// the product includes it from the HIP kernels only.
#pragma once

#ifdef CVX_HOST
#include <cmath>
#define CVX_FN static inline
namespace dm {
struct V3 { float x, y, z; };
struct Q4 { float w, x, y, z; };
struct M3 { float m0, m1, m2, m3, m4, m5, m6, m7, m8; };
CVX_FN V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
CVX_FN V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
CVX_FN V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
CVX_FN V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
CVX_FN float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
CVX_FN M3 q2m(Q4 q) {
  float w = q.w, x = q.x, y = q.y, z = q.z;
  return {w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), w * w - x * x + y * y - z * z,
          2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z};
}
CVX_FN V3 mv(const M3 &m, V3 v) { return {m.m0 * v.x + m.m1 * v.y + m.m2 * v.z, m.m3 * v.x + m.m4 * v.y + m.m5 * v.z, m.m6 * v.x + m.m7 * v.y + m.m8 * v.z}; }
CVX_FN V3 mtv(const M3 &m, V3 v) { return {m.m0 * v.x + m.m3 * v.y + m.m6 * v.z, m.m1 * v.x + m.m4 * v.y + m.m7 * v.z, m.m2 * v.x + m.m5 * v.y + m.m8 * v.z}; }
CVX_FN float frcp(float x) { return 1.f / x; }
CVX_FN float fsqrt(float x) { return x > 0.f ? std::sqrt(x) : 0.f; }
}  // namespace dm
#else
#include "dev_math.hpp"
#define CVX_FN __device__ __forceinline__
#endif

namespace cvx {
using namespace dm;

enum { SPHERE = 2, CAPSULE = 3, ELLIPSOID = 4, CYLINDER = 5 };  // MuJoCo's geom type codes

// centre, orientation, sizes (sphere: radius; capsule / cylinder: radius, half length along the local z axis; ellipsoid: semi-axes)
struct Geom { V3 c; Q4 q; float s0, s1, s2; int type; };
struct Sym3 { float xx, yy, zz, xy, xz, yz; };

CVX_FN V3 zaxis(Q4 q) { return {2.f * (q.x * q.z + q.w * q.y), 2.f * (q.y * q.z - q.w * q.x), q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z}; }
CVX_FN Sym3 outer_s(V3 a, float s) { return {s * a.x * a.x, s * a.y * a.y, s * a.z * a.z, s * a.x * a.y, s * a.x * a.z, s * a.y * a.z}; }
CVX_FN Sym3 add_s(Sym3 a, Sym3 b) { return {a.xx + b.xx, a.yy + b.yy, a.zz + b.zz, a.xy + b.xy, a.xz + b.xz, a.yz + b.yz}; }
CVX_FN Sym3 scale_s(Sym3 a, float s) { return {s * a.xx, s * a.yy, s * a.zz, s * a.xy, s * a.xz, s * a.yz}; }

// Signed distance of x to the geom (negative inside), its gradient (unit) and - when WANT_H - its Hessian, world axes.
template <bool WANT_H>
CVX_FN float sdf(const Geom &g, V3 x, V3 &grad, Sym3 &H) {
  const V3 v = x - g.c;
  H = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (g.type == SPHERE || g.type == CAPSULE) {
    const V3 a = zaxis(g.q);
    const float half = g.type == CAPSULE ? g.s1 : 0.f, z = dot(a, v), t = fminf(fmaxf(z, -half), half);
    const V3 w = v - t * a;
    const float d2 = dot(w, w), d = fsqrt(d2), id = d > 1e-20f ? frcp(d) : 0.f;
    grad = d > 1e-20f ? id * w : V3{1.f, 0.f, 0.f};
    if (WANT_H) {
      H = outer_s(grad, -id);
      H.xx += id; H.yy += id; H.zz += id;
      if (fabsf(z) < half) H = add_s(H, outer_s(a, -id));
    }
    return d - g.s0;
  }
  if (g.type == CYLINDER) {
    const V3 a = zaxis(g.q);
    const float z = dot(a, v), sz = z >= 0.f ? 1.f : -1.f;
    V3 w = v - z * a;
    w = w - dot(a, w) * a;  // (a point on the axis leaves only rounding in w)
    const float rho = fsqrt(dot(w, w)), irho = rho > 1e-20f ? frcp(rho) : 0.f;
    V3 e = irho * w;
    if (!(rho > 1e-20f)) { e = fabsf(a.x) < 0.9f ? cross(a, V3{1.f, 0.f, 0.f}) : cross(a, V3{0.f, 1.f, 0.f}); e = frcp(fsqrt(dot(e, e))) * e; }
    const float dr = rho - g.s0, dz = fabsf(z) - g.s1;
    const V3 t = cross(a, e);
    if (dr > 0.f && dz > 0.f) {  // nearest point on the rim
      const float phi = fsqrt(dr * dr + dz * dz), ip = frcp(phi);
      grad = (dr * ip) * e + (dz * sz * ip) * a;
      if (WANT_H) {
        const V3 u = (-dz * ip) * e + (dr * sz * ip) * a;
        H = add_s(outer_s(u, ip), outer_s(t, dr * ip * irho));
      }
      return phi;
    }
    if (dr > dz) {  // side wall (outside it, or inside and nearer to it than to a cap)
      grad = e;
      if (WANT_H) H = outer_s(t, irho);
      return dr;
    }
    grad = sz * a;  // cap: flat
    return dz;
  }
  // ellipsoid: nearest surface point q_i = s_i^2 y_i / (s_i^2 + tau), tau the root of F(tau) = sum (s_i y_i / (s_i^2 + tau))^2 = 1 above
  // -min s_i^2, bracketed by lo = max_i (s_i |y_i| - s_i^2) (that term alone is 1 there) and hi = sqrt(sum s_i^2 y_i^2) - min s_i^2
  // (every denominator is at least min s_i^2 + tau).  Newton on 1 / sqrt(F) - 1 - exactly linear in tau where one term dominates; plain
  // Newton on F - 1 only gains a factor 1.5 per step next to a pole (a thin blade seen from its flat side) - kept inside the bracket,
  // bisection when a step leaves it (the wing blades' three length scales make 1 / sqrt(F) rise, level off and rise again).
  // 3 - 5 steps for the fly's rounder ellipsoids, up to 10 for the wing blades.
  const M3 R = q2m(g.q);
  const V3 y = mtv(R, v);
  const float sx = g.s0 * g.s0, sy = g.s1 * g.s1, sz2 = g.s2 * g.s2, smin = fminf(fminf(sx, sy), sz2);
  const float px = sx * y.x * y.x, py = sy * y.y * y.y, pz = sz2 * y.z * y.z;
  float lo = fmaxf(fmaxf(g.s0 * fabsf(y.x) - sx, g.s1 * fabsf(y.y) - sy), g.s2 * fabsf(y.z) - sz2);
  float hi = fmaxf(fsqrt(px + py + pz) - smin, lo);
  float tau = lo;
  bool done = false;
#pragma unroll 1
  for (int it = 0; it < 10 && !done; it++) {
    const float ax = frcp(sx + tau), ay = frcp(sy + tau), az = frcp(sz2 + tau);
    const float bx = px * ax * ax, by = py * ay * ay, bz = pz * az * az;
    const float F = bx + by + bz, dF = -2.f * (bx * ax + by * ay + bz * az);
    if (F > 1.f) lo = tau; else hi = tau;
    float tn = dF < 0.f ? tau + 2.f * (F - F * fsqrt(F)) * frcp(dF) : hi;
    if (!(tn >= lo && tn <= hi)) tn = 0.5f * (lo + hi);
    done = fabsf(tn - tau) <= 2e-7f * (fabsf(tau) + smin) || hi - lo <= 2e-7f * (fabsf(tau) + smin);
    tau = tn;
  }
  const float ax = frcp(sx + tau), ay = frcp(sy + tau), az = frcp(sz2 + tau);
  const V3 m = {y.x * ax, y.y * ay, y.z * az};
  const float mn = fsqrt(dot(m, m)), imn = mn > 1e-30f ? frcp(mn) : 0.f;
  const V3 gl = mn > 1e-30f ? imn * m : V3{1.f, 0.f, 0.f};
  grad = mv(R, gl);
  if (WANT_H) {
    // d m_i = a_i dy_i - m_i a_i dtau, dtau = sum_j k_j dy_j / K with k_j = s_j^2 m_j a_j, K = sum_i s_i^2 m_i^2 a_i;
    // H_local = (I - n n') (diag(a) - (m a)(k / K)') / |m|.  Symmetric; assembled as P A P / |m| with A = diag(a) - c c' / K' where it
    // is cheaper to take the symmetric part of the product directly.
    const V3 k = {sx * m.x * ax, sy * m.y * ay, sz2 * m.z * az};
    const float K = k.x * m.x + k.y * m.y + k.z * m.z, iK = frcp(K);
    const V3 ma = {m.x * ax, m.y * ay, m.z * az};
    // J = diag(a) - ma (k' iK)
    const float J00 = ax - ma.x * k.x * iK, J01 = -ma.x * k.y * iK, J02 = -ma.x * k.z * iK;
    const float J10 = -ma.y * k.x * iK, J11 = ay - ma.y * k.y * iK, J12 = -ma.y * k.z * iK;
    const float J20 = -ma.z * k.x * iK, J21 = -ma.z * k.y * iK, J22 = az - ma.z * k.z * iK;
    // rows of (I - n n') J: r_i = J_i - n_i (n' J)
    const V3 nJ = {gl.x * J00 + gl.y * J10 + gl.z * J20, gl.x * J01 + gl.y * J11 + gl.z * J21, gl.x * J02 + gl.y * J12 + gl.z * J22};
    const float h00 = (J00 - gl.x * nJ.x) * imn, h01 = (J01 - gl.x * nJ.y) * imn, h02 = (J02 - gl.x * nJ.z) * imn;
    const float h11 = (J11 - gl.y * nJ.y) * imn, h12 = (J12 - gl.y * nJ.z) * imn, h22 = (J22 - gl.z * nJ.z) * imn;
    // world: R Hl R'
    const V3 c0 = mv(R, V3{h00, h01, h02}), c1 = mv(R, V3{h01, h11, h12}), c2 = mv(R, V3{h02, h12, h22});  // columns of R Hl
    // (R Hl R')_ij = sum_k (R Hl)_ik R_jk with (R Hl)_ik = c_k[i]
    H.xx = c0.x * R.m0 + c1.x * R.m1 + c2.x * R.m2; H.xy = c0.x * R.m3 + c1.x * R.m4 + c2.x * R.m5; H.xz = c0.x * R.m6 + c1.x * R.m7 + c2.x * R.m8;
    H.yy = c0.y * R.m3 + c1.y * R.m4 + c2.y * R.m5; H.yz = c0.y * R.m6 + c1.y * R.m7 + c2.y * R.m8;
    H.zz = c0.z * R.m6 + c1.z * R.m7 + c2.z * R.m8;
  }
  return tau * mn;
}

// support point of the geom in the world direction n (unit)
CVX_FN V3 support(const Geom &g, V3 n) {
  if (g.type == ELLIPSOID) {
    const M3 R = q2m(g.q);
    const V3 l = mtv(R, n);
    const V3 d = {g.s0 * g.s0 * l.x, g.s1 * g.s1 * l.y, g.s2 * g.s2 * l.z};
    const float den = fsqrt(g.s0 * g.s0 * l.x * l.x + g.s1 * g.s1 * l.y * l.y + g.s2 * g.s2 * l.z * l.z);
    return g.c + mv(R, frcp(fmaxf(den, 1e-30f)) * d);
  }
  const V3 a = zaxis(g.q);
  const float z = dot(a, n);
  if (g.type == CYLINDER) {
    V3 w = n - z * a;
    w = w - dot(a, w) * a;  // n nearly along the axis leaves only rounding in w: keep at least its direction across the axis
    const float rn = fsqrt(dot(w, w));
    const V3 rad = rn > 1e-20f ? (g.s0 * frcp(rn)) * w : V3{0.f, 0.f, 0.f};
    return g.c + rad + (z >= 0.f ? g.s1 : -g.s1) * a;
  }
  const float half = g.type == CAPSULE ? g.s1 : 0.f;
  return g.c + g.s0 * n + (z >= 0.f ? half : -half) * a;
}

CVX_FN V3 solve_sym3(const Sym3 &A, V3 b) {
  const float c0 = A.yy * A.zz - A.yz * A.yz, c1 = A.yz * A.xz - A.xy * A.zz, c2 = A.xy * A.yz - A.yy * A.xz;
  const float idet = frcp(A.xx * c0 + A.xy * c1 + A.xz * c2);
  const float c3 = A.xx * A.zz - A.xz * A.xz, c4 = A.xy * A.xz - A.xx * A.yz, c5 = A.xx * A.yy - A.xy * A.xy;
  return {(c0 * b.x + c1 * b.y + c2 * b.z) * idet, (c1 * b.x + c3 * b.y + c4 * b.z) * idet, (c2 * b.x + c4 * b.y + c5 * b.z) * idet};
}

struct Result { float dist; V3 n, pos; };
#ifdef CVX_TRACE
static float g_trace[4096]; static int g_ntrace;
#define CVX_TR(...) do { float v_[] = {__VA_ARGS__}; for (float f_ : v_) if (g_ntrace < 4096) g_trace[g_ntrace++] = f_; } while (0)
#else
#define CVX_TR(...) do {} while (0)
#endif

// A segment inside the geom that its surface follows: the axis segment of a capsule / cylinder, the centre of a sphere, the
// stretch of an ellipsoid's longest axis that its two shorter axes do not reach.  The closest points of two such segments
// start the iteration next to the region where the two surfaces face each other.
CVX_FN void core_segment(const Geom &g, V3 &axis, float &half) {
  if (g.type == ELLIPSOID) {
    const M3 R = q2m(g.q);
    const float mx = fmaxf(fmaxf(g.s0, g.s1), g.s2), mid = g.s0 + g.s1 + g.s2 - mx - fminf(fminf(g.s0, g.s1), g.s2);
    axis = g.s0 == mx ? V3{R.m0, R.m3, R.m6} : (g.s1 == mx ? V3{R.m1, R.m4, R.m7} : V3{R.m2, R.m5, R.m8});
    half = mx - mid;
    return;
  }
  axis = zaxis(g.q);
  half = g.type == SPHERE ? 0.f : g.s1;
}
// closest points p1 + x1 a1, p2 + x2 a2 of two segments (mj: mjc_CapsuleCapsule's parametrisation)
CVX_FN void segment_closest(V3 p1, V3 a1, float l1, V3 p2, V3 a2, float l2, float &x1, float &x2) {
  const V3 dif = p1 - p2;
  const float mb = -dot(a1, a2), u = -dot(a1, dif), v = dot(a2, dif), det = 1.f - mb * mb;
  if (fabsf(det) >= 1e-6f) {
    const float idet = frcp(det);
    x1 = (u - mb * v) * idet; x2 = (v - mb * u) * idet;
    if (x1 > l1) { x1 = l1; x2 = v - mb * l1; } else if (x1 < -l1) { x1 = -l1; x2 = v + mb * l1; }
    if (x2 > l2) { x2 = l2; x1 = fminf(fmaxf(u - mb * l2, -l1), l1); }
    else if (x2 < -l2) { x2 = -l2; x1 = fminf(fmaxf(u + mb * l2, -l1), l1); }
    else x2 = fminf(fmaxf(x2, -l2), l2);
  } else {
    const float c2 = u, lo = fmaxf(-l1, c2 - l2), hi = fminf(l1, c2 + l2);
    x1 = lo <= hi ? 0.5f * (lo + hi) : (c2 > 0.f ? l1 : -l1);
    x2 = fminf(fmaxf((x1 - c2) * (mb < 0.f ? 1.f : -1.f), -l2), l2);
  }
}

// Narrow phase of one pair.  `n` = start direction from geom1 to geom2 when `have_n` (e.g. the pair's normal of the previous
// substep; `xw` = its contact position then, when `have_x`), else it is taken from the two core segments; `sgap` = separation kept while the closest points are found;
// `cull` = stop once -o(n) > cull proves there is no contact (dist then holds that lower bound).  Returns dist, the normal
// from geom1 to geom2 and the contact position.
template <int MAX_OUTER, int N_INNER>
CVX_FN Result distance(Geom a, Geom b, V3 n, bool have_n, float sgap, float cull, V3 xw = V3{0.f, 0.f, 0.f}, bool have_x = false) {
  // work about geom1's centre: the cancellations below then happen between numbers of the geoms' own size
  const V3 origin = a.c;
  b.c = b.c - a.c;
  a.c = {0.f, 0.f, 0.f};
  Result r;
  V3 q1, q2;
  {
    V3 a1, a2;
    float l1, l2, x1, x2;
    core_segment(a, a1, l1); core_segment(b, a2, l2);
    segment_closest(a.c, a1, l1, b.c, a2, l2, x1, x2);
    q1 = x1 * a1; q2 = b.c + x2 * a2;
    V3 u = q2 - q1;
    const float ul = fsqrt(dot(u, u));
    if (ul > 1e-6f * (sgap + 1e-12f)) u = frcp(ul) * u;
    else { u = b.c; const float bl = fsqrt(dot(u, u)); u = bl > 1e-20f ? frcp(bl) * u : V3{1.f, 0.f, 0.f}; }
    if (!have_n) n = u;
  }
  r.dist = 0.f; r.n = n; r.pos = {0.f, 0.f, 0.f};
  V3 p1 = {0.f, 0.f, 0.f}, p2 = {0.f, 0.f, 0.f};
  bool done = false;
#pragma unroll 1
  for (int outer = 0; outer < MAX_OUTER && !done; outer++) {
    const V3 s1 = support(a, r.n), s2 = support(b, V3{-r.n.x, -r.n.y, -r.n.z});
    const float o = dot(r.n, s1 - s2);
    if (-o > cull) { r.dist = -o; break; }
    const float T = fmaxf(0.f, o + sgap);
    const V3 shift = T * r.n;
    Geom bs = b;
    bs.c = b.c + shift;
    // start: on the line through the two core points, half-way between the two supporting planes with normal n
    V3 x;
    if (outer == 0 && have_x) x = xw - origin + 0.5f * shift;  // the pair's last contact position (warm start)
    else if (outer == 0) {
      const float t1 = dot(r.n, s1 - q1), t2 = dot(r.n, q2 - s2), len = dot(r.n, q2 + shift - q1);
      x = q1 + (t1 + 0.5f * (len - t1 - t2)) * r.n;
      x = x + 0.5f * ((q2 + shift - q1) - len * r.n);  // and half-way across, where the two core points are not in line with n
    } else x = 0.5f * (p1 + p2 + shift);
    x = x + (dot(r.n, s1 - x) + 0.5f * (o + T)) * r.n;  // onto the mid-plane between the two supporting planes: outside both geoms
    V3 ga = r.n, gb = r.n;
    float fa = 0.f, fb = 0.f;
    Sym3 Ha, Hb;
#pragma unroll 1
    for (int it = 0; it < N_INNER; it++) {
      fa = sdf<true>(a, x, ga, Ha); fb = sdf<true>(bs, x, gb, Hb);
      const float da = fmaxf(fa, 0.f), db = fmaxf(fb, 0.f);
      Sym3 A = add_s(add_s(scale_s(Ha, da), outer_s(ga, fa > 0.f ? 1.f : 0.f)), add_s(scale_s(Hb, db), outer_s(gb, fb > 0.f ? 1.f : 0.f)));
      const float mu = 1e-4f * (A.xx + A.yy + A.zz) + 1e-30f;
      A.xx += mu; A.yy += mu; A.zz += mu;
      const V3 rhs = {-(da * ga.x + db * gb.x), -(da * ga.y + db * gb.y), -(da * ga.z + db * gb.z)};
      V3 dx = da + db > 0.f ? solve_sym3(A, rhs) : V3{0.f, 0.f, 0.f};
      // trust region: no step longer than twice the distance the point still has to both geoms together
      const float dl2 = dot(dx, dx), lim = 2.f * (da + db) + sgap;
      if (dl2 > lim * lim) dx = (lim * frcp(fsqrt(dl2))) * dx;
      CVX_TR((float)outer, (float)it, fa, fb, x.x, x.y, x.z, r.n.x, r.n.y, r.n.z, T, o);
      x = x + dx;
    }
    Sym3 dummy;
    fa = sdf<false>(a, x, ga, dummy); fb = sdf<false>(bs, x, gb, dummy);
    V3 nn = ga - gb;
    const float nl = fsqrt(dot(nn, nn));
    nn = nl > 1e-20f ? frcp(nl) * nn : r.n;
    const V3 dn = nn - r.n;
    done = T == 0.f || dot(dn, dn) < 1e-12f;
    r.dist = fa + fb - T;
    p1 = x - fa * ga; p2 = x - fb * gb - shift;
    r.n = nn;
  }
  r.pos = origin + 0.5f * (p1 + p2);
  return r;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Pair classes with a cheaper exact formulation than the general iteration above.  Both solve the same minimum-translation
// problem (same dist / normal / position as `distance`, to rounding) for contacts no deeper than the thinner geom's core.

// (1) A capsule or sphere (geom1: lower type code) against an ellipsoid or cylinder: the capsule is its axis segment p + t a,
// |t| <= half, swept by a ball of radius r, so dist = min over t of phi_G(p + t a) - r with phi_G the convex, C1 signed distance to
// geom2: a one-dimensional convex minimisation, f'(t) = grad phi . a monotone.  Newton on f' inside a bracket.  A cylinder's
// f' is only piecewise smooth - its curvature jumps by orders of magnitude where the axis point passes over the rim (|z| = H) or
// over the side wall (rho = R) - so for a cylinder a Newton step never leaves the smooth piece it starts in: it stops just past
// the piece's end (those breakpoints are roots of a linear and a quadratic equation in t) and continues there with the next
// piece's curvature; f' being monotone, the walk crosses each breakpoint at most once.
// Exact while the axis segment stays outside geom2 (phi > 0 at the minimiser: penetration shallower than the capsule's radius);
// deeper, it returns the largest depth of an axis point, a lower bound of the true penetration (`shallow` = false).
struct PResult { float dist; V3 n, pos; bool shallow; float t; };
// (`t0`, when `have_t`: the minimiser of the last substep - Newton then starts next to the answer)
template <int N_ITER>
CVX_FN PResult prim_convex(Geom p, Geom g, float t0 = 0.f, bool have_t = false) {
  const V3 origin = g.c;
  p.c = p.c - g.c;
  g.c = {0.f, 0.f, 0.f};
  const V3 a = zaxis(p.q);
  const float half = p.type == CAPSULE ? p.s1 : 0.f, r = p.s0;
  // breakpoints of a cylinder's f' (ellipsoid: none)
  float bk0 = 2.f * half + 1.f, bk1 = bk0, bk2 = bk0, bk3 = bk0;  // (beyond the segment = absent)
  if (g.type == CYLINDER) {
    const V3 ga = zaxis(g.q);
    const float az = dot(a, ga), z0 = dot(p.c, ga);
    if (fabsf(az) > 1e-6f) { const float iz = frcp(az); bk0 = (g.s1 - z0) * iz; bk1 = (-g.s1 - z0) * iz; }
    const V3 ap = a - az * ga, pp = p.c - z0 * ga;
    const float qa = dot(ap, ap), qb = dot(ap, pp), qc = dot(pp, pp) - g.s0 * g.s0, disc = qb * qb - qa * qc;
    if (qa > 1e-12f && disc > 0.f) { const float sq = fsqrt(disc), iq = frcp(qa); bk2 = (-qb - sq) * iq; bk3 = (-qb + sq) * iq; }
  }
  const float nudge = 2e-6f * (half + r);
  float lo = -half, hi = half;
  bool lo_ev = false, hi_ev = false, conv = half == 0.f;
  float t = fminf(fmaxf(have_t ? t0 : -dot(a, p.c), lo), hi);
  V3 grad = {1.f, 0.f, 0.f};
  Sym3 H;
  float f = 0.f;
#pragma unroll 1
  for (int it = 0; it < N_ITER && !conv; it++) {
    const V3 x = p.c + t * a;
    f = sdf<true>(g, x, grad, H);
    const float d1 = dot(grad, a);
    const float d2 = a.x * (H.xx * a.x + H.xy * a.y + H.xz * a.z) + a.y * (H.xy * a.x + H.yy * a.y + H.yz * a.z) + a.z * (H.xz * a.x + H.yz * a.y + H.zz * a.z);
    const bool neg = d1 <= 0.f;  // the root lies at or beyond t
    if (neg) { lo = t; lo_ev = true; } else { hi = t; hi_ev = true; }
    // end of the smooth piece in the direction of the root
    float pe;
    if (neg) {
      pe = hi;
      if (bk0 > t + nudge && bk0 < pe) pe = bk0;
      if (bk1 > t + nudge && bk1 < pe) pe = bk1;
      if (bk2 > t + nudge && bk2 < pe) pe = bk2;
      if (bk3 > t + nudge && bk3 < pe) pe = bk3;
    } else {
      pe = lo;
      if (bk0 < t - nudge && bk0 > pe) pe = bk0;
      if (bk1 < t - nudge && bk1 > pe) pe = bk1;
      if (bk2 < t - nudge && bk2 > pe) pe = bk2;
      if (bk3 < t - nudge && bk3 > pe) pe = bk3;
    }
    float tn = d2 > 1e-9f ? t - d1 * frcp(d2) : (neg ? pe + 1.f : pe - 1.f);
    bool newton = true;
    if (neg ? tn >= pe : tn <= pe) {  // the step leaves the piece: stop just past its end (a bracket end: on it, or bisect if evaluated)
      newton = false;
      const bool at_end = neg ? pe >= hi : pe <= lo;
      tn = at_end ? ((neg ? hi_ev : lo_ev) ? 0.5f * (lo + hi) : pe) : (neg ? fminf(pe + nudge, hi) : fmaxf(pe - nudge, lo));
    }
    const float tol = 1e-7f * (half + r);
    conv = fabsf(d1) <= 1e-7f || hi - lo <= tol || (newton && fabsf(tn - t) <= tol) || (neg && t >= half) || (!neg && t <= -half);
    t = tn;
  }
  const V3 x = p.c + t * a;
  f = sdf<false>(g, x, grad, H);
  PResult out;
  out.dist = f - r;
  out.n = V3{-grad.x, -grad.y, -grad.z};
  out.pos = origin + x - (0.5f * (r + f)) * grad;
  out.shallow = f > 0.f;
  out.t = t;
  return out;
}

// (2) Two ellipsoids: their support functions h_i(n) = sqrt(n' A_i n), A_i = R_i S_i^2 R_i', are smooth, so the overlap
// o(n) = h_1(n) + h_2(n) - n . (c_2 - c_1) is minimised over the unit sphere directly: Newton in the tangent plane of n on the
// Lagrangian (multiplier = o(n) by homogeneity; grad o = p_1 - p_2, the difference of the two witness points), with a step limit
// and backtracking on o.  dist = -min o.
CVX_FN Sym3 ell_matrix(const Geom &g) {
  const M3 R = q2m(g.q);
  const float a = g.s0 * g.s0, b = g.s1 * g.s1, c = g.s2 * g.s2;
  return {a * R.m0 * R.m0 + b * R.m1 * R.m1 + c * R.m2 * R.m2, a * R.m3 * R.m3 + b * R.m4 * R.m4 + c * R.m5 * R.m5,
          a * R.m6 * R.m6 + b * R.m7 * R.m7 + c * R.m8 * R.m8, a * R.m0 * R.m3 + b * R.m1 * R.m4 + c * R.m2 * R.m5,
          a * R.m0 * R.m6 + b * R.m1 * R.m7 + c * R.m2 * R.m8, a * R.m3 * R.m6 + b * R.m4 * R.m7 + c * R.m5 * R.m8};
}
CVX_FN V3 sym_mv(const Sym3 &A, V3 v) { return {A.xx * v.x + A.xy * v.y + A.xz * v.z, A.xy * v.x + A.yy * v.y + A.yz * v.z, A.xz * v.x + A.yz * v.y + A.zz * v.z}; }
template <int N_ITER>
CVX_FN Result ell_ell(const Geom &a, const Geom &b, V3 n, bool have_n) {
  const Sym3 A1 = ell_matrix(a), A2 = ell_matrix(b);
  const V3 c = b.c - a.c;
  if (!have_n) {
    const float cl = fsqrt(dot(c, c));
    n = cl > 1e-20f ? frcp(cl) * c : V3{1.f, 0.f, 0.f};
  }
  V3 u1 = sym_mv(A1, n), u2 = sym_mv(A2, n);
  float h1 = fsqrt(dot(n, u1)), h2 = fsqrt(dot(n, u2)), o = h1 + h2 - dot(n, c);
#pragma unroll 1
  for (int it = 0; it < N_ITER; it++) {
    const float i1 = frcp(h1), i2 = frcp(h2);
    const V3 gr = i1 * u1 + i2 * u2 - c;
    // Hessian of h_1 + h_2 minus o I, restricted to the tangent plane of n
    const float k1 = i1 * i1 * i1, k2 = i2 * i2 * i2;
    Sym3 Hs = add_s(add_s(scale_s(A1, i1), outer_s(u1, -k1)), add_s(scale_s(A2, i2), outer_s(u2, -k2)));
    V3 t1 = fabsf(n.x) < 0.6f ? V3{0.f, -n.z, n.y} : V3{-n.z, 0.f, n.x};
    t1 = frcp(fsqrt(dot(t1, t1))) * t1;
    const V3 t2 = cross(n, t1);
    const V3 Ht1 = sym_mv(Hs, t1), Ht2 = sym_mv(Hs, t2);
    float m00 = dot(t1, Ht1) - o, m01 = dot(t1, Ht2), m11 = dot(t2, Ht2) - o;
    const float r0 = -dot(t1, gr), r1 = -dot(t2, gr);
    float det = m00 * m11 - m01 * m01;
    if (!(m00 > 0.f && det > 1e-6f * m00 * m11)) {  // not positive definite (a deep overlap): fall back to the curvature part alone
      m00 += fmaxf(o, 0.f) * 1.5f; m11 += fmaxf(o, 0.f) * 1.5f;
      m00 = fmaxf(m00, 1e-12f); m11 = fmaxf(m11, 1e-12f);
      det = fmaxf(m00 * m11 - m01 * m01, 1e-3f * m00 * m11);
    }
    const float idet = frcp(det);
    float d0 = (m11 * r0 - m01 * r1) * idet, d1 = (m00 * r1 - m01 * r0) * idet;
    const float dl2 = d0 * d0 + d1 * d1;
    if (dl2 > 0.25f) { const float s = 0.5f * frcp(fsqrt(dl2)); d0 *= s; d1 *= s; }
    bool moved = false;
#pragma unroll 1
    for (int bt = 0; bt < 4 && !moved; bt++) {
      V3 nn = n + d0 * t1 + d1 * t2;
      nn = frcp(fsqrt(dot(nn, nn))) * nn;
      const V3 v1 = sym_mv(A1, nn), v2 = sym_mv(A2, nn);
      const float g1 = fsqrt(dot(nn, v1)), g2 = fsqrt(dot(nn, v2)), on = g1 + g2 - dot(nn, c);
      if (on <= o + 1e-7f * (h1 + h2)) { n = nn; u1 = v1; u2 = v2; h1 = g1; h2 = g2; o = on; moved = true; }
      else { d0 *= 0.5f; d1 *= 0.5f; }
    }
    if (!moved) break;
  }
  Result r;
  r.dist = -o;
  r.n = n;
  const V3 p1 = frcp(h1) * u1, p2 = c - frcp(h2) * u2;
  r.pos = a.c + 0.5f * (p1 + p2);
  return r;
}


// (3) An ellipsoid (geom1) against a cylinder (geom2), in the same dual form as (2).  The cylinder's support function
// h_C(n) = R |n_perp| + H |n_z| (n_z = n . axis) has kinks on the great circle n_z = 0 (witness anywhere along a side-wall line) and at
// the poles n = +-axis (witness anywhere on a cap), and those kinks are exactly where contacts with a side wall or a cap sit.  The
// minimiser of o(n) = h_E(n) + h_C(n) + n . c (c = ellipsoid centre - cylinder centre, n from the ellipsoid to the cylinder) is
// looked for on the three kinds of cylinder feature, each with its exact optimality test:
//   cap:  n = -+axis; optimal iff the ellipsoid's witness point lies over the cap disc;
//   side: the minimiser on the circle n_z = 0 (coarse scan of eight directions, then one-dimensional Newton); optimal iff the
//         ellipsoid's witness lies between the caps;
//   rim:  the cylinder's witness is a point z(psi) of a rim circle; the contact is where the ellipsoid's signed distance along the
//         circle is least (positive: gap, negative: depth along the ellipsoid's own normal) - one-dimensional Newton on psi with
//         backtracking from the best of a few starts (the azimuth of the side witness; the rim points deepest under either face of
//         the ellipsoid's thinnest axis: a wing blade lying across the abdomen's edge); the candidate's value is its dual bound -o(n).
// Separated geoms have one stationary direction; overlapping ones may have one per feature: the least overlap among them is kept
// (every direction's -o(n) is a lower bound of dist).  With the pair's direction of the last substep (`have_n`) only the feature
// that direction lies on is refined; the full search runs when that does not reproduce a valid contact near the old direction.
template <int N_ITER>
CVX_FN Result ell_cyl(const Geom &e, const Geom &cy, V3 n0 = V3{0.f, 0.f, 0.f}, bool have_n = false) {
  const Sym3 A = ell_matrix(e);
  const V3 c = e.c - cy.c, ax = zaxis(cy.q);
  const float R = cy.s0, Hh = cy.s1;
  Result r;
  r.dist = -1e30f; r.n = V3{1.f, 0.f, 0.f}; r.pos = cy.c;
  Geom eg = e;
  eg.c = c;  // (coordinates relative to the cylinder's centre)

  auto cap = [&](float sgn) {
    const V3 n = (-sgn) * ax, u = sym_mv(A, n);
    const float h = fsqrt(dot(n, u));
    const V3 pe = c + frcp(h) * u, pr = pe - dot(pe, ax) * ax;
    const float d = -(h + Hh + dot(n, c));
    if (dot(pr, pr) <= R * R && d > r.dist) { r.dist = d; r.n = n; r.pos = cy.c + 0.5f * (pe + pr + (sgn * Hh) * ax); }
  };
  // side: Newton on the circle from `n` (unit, across the axis); returns the axial coordinate of the ellipsoid's witness
  auto side = [&](V3 n, V3 &pe_out) -> float {
    V3 u = sym_mv(A, n);
    float h = fsqrt(dot(n, u)), o = h + R + dot(n, c);
#pragma unroll 1
    for (int it = 0; it < N_ITER; it++) {
      const float ih = frcp(h);
      const V3 t = cross(ax, n), gr = ih * u + c, At = sym_mv(A, t);
      const float ut = dot(u, t), d1 = dot(gr, t), d2 = dot(t, At) * ih - ut * ut * ih * ih * ih - dot(n, gr);
      if (fabsf(d1) < 1e-9f) break;
      float dth = d2 > 1e-9f ? -d1 * frcp(d2) : (d1 > 0.f ? -0.3f : 0.3f);
      dth = fminf(fmaxf(dth, -0.5f), 0.5f);
      bool moved = false;
#pragma unroll 1
      for (int bt = 0; bt < 4 && !moved; bt++) {
        V3 nn = n + dth * t;
        nn = frcp(fsqrt(dot(nn, nn))) * nn;
        const V3 un = sym_mv(A, nn);
        const float hn = fsqrt(dot(nn, un)), on = hn + R + dot(nn, c);
        if (on <= o + 1e-7f * (h + R)) { n = nn; u = un; h = hn; o = on; moved = true; }
        else dth *= 0.5f;
      }
      if (!moved) break;
    }
    const V3 pe = c + frcp(h) * u;
    const float za = dot(pe, ax);
    if (fabsf(za) <= Hh && -o > r.dist) { r.dist = -o; r.n = n; r.pos = cy.c + 0.5f * (pe + za * ax - R * n); }
    pe_out = pe;
    return za;
  };
  auto rim_f = [&](float sr, V3 e1) { V3 g; Sym3 Hq; return sdf<false>(eg, (sr * Hh) * ax + R * e1, g, Hq); };
  auto rim = [&](float sr, V3 e1) {
    V3 e2 = cross(ax, e1), grad;
    const V3 zc = (sr * Hh) * ax;
    V3 zr = R * e1;
    Sym3 Hq;
    float f = sdf<true>(eg, zc + zr, grad, Hq);
#pragma unroll 1
    for (int it = 0; it < N_ITER; it++) {
      const V3 zp = R * e2;  // z = zc + zr, z' = R e2 (e2 = axis x e1), z'' = -zr
      const float d1 = dot(grad, zp), d2 = dot(zp, sym_mv(Hq, zp)) - dot(grad, zr);
      if (fabsf(d1) < 1e-9f * R) break;
      float dps = d2 > 1e-12f ? -d1 * frcp(d2) : (d1 > 0.f ? -0.3f : 0.3f);
      dps = fminf(fmaxf(dps, -0.6f), 0.6f);
      bool moved = false;
#pragma unroll 1
      for (int bt = 0; bt < 4 && !moved; bt++) {
        V3 f1 = e1 + dps * e2;  // (a rotation to first order, renormalised: the step size itself needs no accuracy)
        f1 = frcp(fsqrt(dot(f1, f1))) * f1;
        V3 g2; Sym3 H2;
        const float fn = sdf<true>(eg, zc + R * f1, g2, H2);
        if (fn <= f + 1e-7f * (fabsf(f) + R)) { e1 = f1; e2 = cross(ax, e1); zr = R * e1; f = fn; grad = g2; Hq = H2; moved = true; }
        else dps *= 0.5f;
      }
      if (!moved) break;
    }
    // the candidate's direction n = the ellipsoid's outward normal at its witness (from the ellipsoid to the cylinder); its rigorous
    // value is the dual bound -o(n), which equals f when n lies in the rim's normal cone and falls far below it when it does not
    const V3 un = sym_mv(A, grad);
    const float nzr = dot(grad, ax);
    const V3 mp = grad - nzr * ax;
    const float orim = fsqrt(dot(grad, un)) + R * fsqrt(dot(mp, mp)) + Hh * fabsf(nzr) + dot(grad, c);
    if (-orim > r.dist) { r.dist = -orim; r.n = grad; r.pos = cy.c + zc + zr - (0.5f * f) * grad; }
  };
  // rim, dual form: o is smooth where n_z and n_perp both differ from zero - two-dimensional Newton as in (2), kept on the side
  // sz of the kink n_z = 0.  Needed where the ellipsoid's own edge is the contact (a wing blade's edge on the abdomen's rim: the
  // overlap then exceeds the blade edge's radius of curvature and the deepest rim point is no longer the witness).
  auto rim_dual = [&](V3 n, float sz) {
    {
      const float nz = dot(n, ax);
      if (nz * sz < 0.1f) { n = n + (0.1f * sz - nz) * ax; n = frcp(fsqrt(dot(n, n))) * n; }
    }
    auto hc = [&](V3 m, V3 &gc) {  // h_C and its gradient (the cylinder's support point relative to its centre)
      const float mz = dot(m, ax);
      const V3 mp = m - mz * ax;
      const float ml = fsqrt(dot(mp, mp));
      gc = (ml > 1e-20f ? R * frcp(ml) : 0.f) * mp + (sz * Hh) * ax;
      return R * ml + Hh * fabsf(mz);
    };
    V3 gc, u = sym_mv(A, n);
    float h = fsqrt(dot(n, u)), h2 = hc(n, gc), o = h + h2 + dot(n, c);
#pragma unroll 1
    for (int it = 0; it < N_ITER; it++) {
      const float ih = frcp(h);
      const V3 gr = ih * u + gc + c;
      Sym3 Hs = add_s(scale_s(A, ih), outer_s(u, -ih * ih * ih));
      {  // cylinder: R (I - ax ax' - e e') / |n_perp|
        const float mz = dot(n, ax);
        const V3 mp = n - mz * ax;
        const float ml = fsqrt(dot(mp, mp)), k = ml > 1e-6f ? R * frcp(ml) : 0.f;
        const V3 ee = ml > 1e-6f ? frcp(ml) * mp : V3{0.f, 0.f, 0.f};
        Hs.xx += k; Hs.yy += k; Hs.zz += k;
        Hs = add_s(Hs, add_s(outer_s(ax, -k), outer_s(ee, -k)));
      }
      V3 t1 = fabsf(n.x) < 0.6f ? V3{0.f, -n.z, n.y} : V3{-n.z, 0.f, n.x};
      t1 = frcp(fsqrt(dot(t1, t1))) * t1;
      const V3 t2 = cross(n, t1), Ht1 = sym_mv(Hs, t1), Ht2 = sym_mv(Hs, t2);
      float m00 = dot(t1, Ht1) - o, m01 = dot(t1, Ht2), m11 = dot(t2, Ht2) - o;
      const float r0 = -dot(t1, gr), r1 = -dot(t2, gr);
      if (r0 * r0 + r1 * r1 < 1e-17f) break;
      float det = m00 * m11 - m01 * m01;
      if (!(m00 > 0.f && det > 1e-6f * m00 * m11)) {
        m00 += fmaxf(o, 0.f) * 1.5f; m11 += fmaxf(o, 0.f) * 1.5f;
        m00 = fmaxf(m00, 1e-12f); m11 = fmaxf(m11, 1e-12f);
        det = fmaxf(m00 * m11 - m01 * m01, 1e-3f * m00 * m11);
      }
      const float idet = frcp(det);
      float d0 = (m11 * r0 - m01 * r1) * idet, d1 = (m00 * r1 - m01 * r0) * idet;
      const float dl2 = d0 * d0 + d1 * d1;
      if (dl2 > 0.09f) { const float sc = 0.3f * frcp(fsqrt(dl2)); d0 *= sc; d1 *= sc; }
      bool moved = false;
#pragma unroll 1
      for (int bt = 0; bt < 5 && !moved; bt++) {
        V3 nn = n + d0 * t1 + d1 * t2;
        nn = frcp(fsqrt(dot(nn, nn))) * nn;
        V3 gcn;
        const V3 un = sym_mv(A, nn);
        const float hn = fsqrt(dot(nn, un)), h2n = hc(nn, gcn), on = hn + h2n + dot(nn, c);
        if (dot(nn, ax) * sz > 1e-4f && on <= o + 1e-7f * (h + h2)) { n = nn; u = un; h = hn; h2 = h2n; gc = gcn; o = on; moved = true; }
        else { d0 *= 0.5f; d1 *= 0.5f; }
      }
      if (!moved) break;
    }
    if (-o > r.dist + 2e-7f * R) {  // (a candidate equal to the one already held to rounding does not replace it)
      r.dist = -o; r.n = n;
      const V3 p1 = c + frcp(h) * u, p2 = V3{-gc.x, -gc.y, -gc.z};
      r.pos = cy.c + 0.5f * (p1 + p2);
    }
  };
  auto across = [&](V3 v, V3 fallback) {  // unit vector of v's part across the axis
    const V3 w = v - dot(v, ax) * ax;
    const float wl = fsqrt(dot(w, w));
    return wl > 1e-12f ? frcp(wl) * w : fallback;
  };
  V3 b1 = fabsf(ax.x) < 0.9f ? cross(ax, V3{1.f, 0.f, 0.f}) : cross(ax, V3{0.f, 1.f, 0.f});
  b1 = frcp(fsqrt(dot(b1, b1))) * b1;

  if (have_n) {  // refine the feature the pair's last direction lies on
    const float nz = dot(n0, ax);
    V3 pe;
    if (fabsf(nz) > 0.99995f) cap(nz > 0.f ? -1.f : 1.f);
    else if (fabsf(nz) < 2e-3f) side(across(n0, b1), pe);
    else rim_dual(n0, nz > 0.f ? 1.f : -1.f);
    if (r.dist > -1e29f && dot(r.n, n0) > 0.95f) return r;
    r.dist = -1e30f;
  }
  const float cz = dot(c, ax);
  cap(cz >= 0.f ? 1.f : -1.f);
  if (r.dist >= 0.f) return r;
  // side: best of eight directions around the axis as the start
  V3 pe;
  float za;
  {
    const V3 b2 = cross(ax, b1);
    V3 nbest = b1;
    float obest = 1e30f;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const float cs = k == 0 ? 1.f : k == 1 ? 0.70710678f : k == 2 ? 0.f : k == 3 ? -0.70710678f : k == 4 ? -1.f : k == 5 ? -0.70710678f : k == 6 ? 0.f : 0.70710678f;
      const float sn = k == 0 ? 0.f : k == 1 ? 0.70710678f : k == 2 ? 1.f : k == 3 ? 0.70710678f : k == 4 ? 0.f : k == 5 ? -0.70710678f : k == 6 ? -1.f : -0.70710678f;
      const V3 n = cs * b1 + sn * b2;
      const float o = fsqrt(dot(n, sym_mv(A, n))) + dot(n, c);
      if (o < obest) { obest = o; nbest = n; }
    }
    const float dprev = r.dist;
    za = side(nbest, pe);
    if (r.dist >= 0.f && r.dist > dprev) return r;
  }
  // rim: the best of a few starts
  {
    const M3 Re = q2m(e.q);
    const V3 m = e.s0 <= e.s1 && e.s0 <= e.s2 ? V3{Re.m0, Re.m3, Re.m6} : (e.s1 <= e.s2 ? V3{Re.m1, Re.m4, Re.m7} : V3{Re.m2, Re.m5, Re.m8});  // thinnest axis
    const V3 em = across(m, b1), es = across(pe, V3{-b1.x, -b1.y, -b1.z});
    const float s0 = za > 0.f ? 1.f : -1.f;
    float sr = s0, fb = rim_f(s0, es);
    V3 eb = es;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float s = k < 2 ? 1.f : -1.f;
      const V3 e1 = (k & 1) ? em : V3{-em.x, -em.y, -em.z};
      const float f = rim_f(s, e1);
      if (f < fb) { fb = f; sr = s; eb = e1; }
    }
    rim(sr, eb);
    const V3 nb = r.dist > -1e29f ? r.n : V3{-es.x, -es.y, -es.z};
    rim_dual(nb, 1.f);
    rim_dual(nb, -1.f);
    // and from the ellipsoid's thinnest axis (a blade pressed flat over the rim: the direction of least overlap is close to the
    // blade's normal), towards the cylinder
    const V3 mt = dot(m, c) > 0.f ? V3{-m.x, -m.y, -m.z} : m;
    rim_dual(mt, 1.f);
    rim_dual(mt, -1.f);
  }
  return r;
}

// ---------------------------------------------------------------------------------------------------------------------------
// What the kernels call.  geom1 has the lower type code (mj_collision's order), so the normal points from geom1 to geom2.
struct Contact { float dist; V3 n, pos; float t; };  // (t: the capsule's axis parameter of a capsule / sphere - convex pair, kept as the next substep's start)

// Overlap of the two geoms along the unit direction u (from geom1 to geom2): -o(u) is a rigorous lower bound of the pair's
// distance for every u.
CVX_FN float overlap(const Geom &a, const Geom &b, V3 u) {
  const V3 s1 = support(a, u), s2 = support(b, V3{-u.x, -u.y, -u.z});
  return dot(u, s1 - s2);
}
// The broad phase's second test: the better of two such bounds - u between the closest points of the two core segments (long
// geoms side by side) and u between the two centres (flat geoms stacked on each other: the abdomen's discs).  A pair whose bound
// exceeds its margin cannot touch.
CVX_FN float separation_bound(const Geom &a, const Geom &b) {
  V3 a1, a2;
  float l1, l2, x1, x2, best = -1e30f;
  core_segment(a, a1, l1); core_segment(b, a2, l2);
  segment_closest(a.c, a1, l1, b.c, a2, l2, x1, x2);
  V3 u = (b.c + x2 * a2) - (a.c + x1 * a1);
  float ul = fsqrt(dot(u, u));
  if (ul > 1e-12f) best = -overlap(a, b, frcp(ul) * u);  // (crossing cores: no bound from this direction)
  u = b.c - a.c;
  ul = fsqrt(dot(u, u));
  if (ul > 1e-12f) best = fmaxf(best, -overlap(a, b, frcp(ul) * u));
  return best;
}

// mj: mjc_CapsuleCapsule (closest points of the two axis segments, one contact); a sphere is a capsule of zero length, for which
// the same formulas are mjc_SphereCapsule / mjraw_SphereSphere
CVX_FN Contact capsule_capsule(const Geom &a, const Geom &b) {
  const V3 a1 = zaxis(a.q), a2 = zaxis(b.q);
  const float l1 = a.type == CAPSULE ? a.s1 : 0.f, l2 = b.type == CAPSULE ? b.s1 : 0.f;
  float x1, x2;
  {  // (mjc_CapsuleCapsule's own case split: kept apart from segment_closest, whose clamping order differs in the last digit)
    const V3 dif = a.c - b.c;
    const float mb = -dot(a1, a2), u = -dot(a1, dif), v = dot(a2, dif), det = 1.f - mb * mb;
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
  }
  const V3 q1 = a.c + x1 * a1, q2 = b.c + x2 * a2, d12 = q2 - q1;
  const float cd = fsqrt(dot(d12, d12));
  Contact out;
  out.t = 0.f;
  out.n = cd >= 1e-15f ? frcp(cd) * d12 : V3{1.f, 0.f, 0.f};
  out.dist = cd - a.s0 - b.s0;
  out.pos = q1 + (a.s0 + 0.5f * out.dist) * out.n;
  return out;
}

// cylinder - cylinder (never taken on the tasks' workloads: the separating-direction test culls the abdomen's stacked discs): the
// general iteration.  Inline like everything else here: the kernels' collision passes are out-of-line LEAF functions, which the
// register allocator keeps inside the caller-saved registers; a call inside one of them would turn it into a non-leaf function that
// saves and restores every callee-saved register it keeps live across that call on every entry (measured: 1.1 GB of scratch traffic
// per launch of the flight kernel).
CVX_FN Result cyl_cyl(const Geom &g1, const Geom &g2) {
  return distance<6, 4>(g1, g2, V3{0.f, 0.f, 0.f}, false, 0.05f * fminf(g1.s0, g2.s0), 1e30f);
}

#ifndef CVX_PRIM_ITERS
#define CVX_PRIM_ITERS 8
#endif
#ifndef CVX_ELLELL_ITERS
#define CVX_ELLELL_ITERS 8
#endif
#ifndef CVX_ELLCYL_ITERS
#define CVX_ELLCYL_ITERS 8   // (12 measured the same parity and cost 4 % of the flight throughput: the second collision pass sets the length of the launch; 6 passes the host bounds but loses a cold wing-blade contact: tests/test_gpu_parity.py::test_forced_contacts_one_substep)
#endif
// (`n0`: the pair's direction of the last substep, when `have_n`: the ellipsoid classes then refine it instead of searching)
template <bool WITH_RARE = true>
CVX_FN Contact collide(const Geom &g1, const Geom &g2, V3 n0 = V3{0.f, 0.f, 0.f}, bool have_n = false, float t0 = 0.f) {
  Contact out;
  out.t = 0.f;
  if (g2.type <= CAPSULE) return capsule_capsule(g1, g2);
  if (g1.type <= CAPSULE) {  // sphere / capsule against ellipsoid / cylinder
    const PResult r = prim_convex<CVX_PRIM_ITERS>(g1, g2, t0, have_n);
    out.dist = r.dist; out.n = r.n; out.pos = r.pos; out.t = r.t;
  } else if (g2.type == ELLIPSOID) {
    const Result r = ell_ell<CVX_ELLELL_ITERS>(g1, g2, n0, have_n);
    out.dist = r.dist; out.n = r.n; out.pos = r.pos;
  } else if (!WITH_RARE) {  // (two-pass callers: the rare classes are the second pass's)
    out.dist = 1e30f; out.n = V3{1.f, 0.f, 0.f}; out.pos = g1.c;
  } else if (g1.type == ELLIPSOID) {
    const Result r = ell_cyl<CVX_ELLCYL_ITERS>(g1, g2, n0, have_n);
    out.dist = r.dist; out.n = r.n; out.pos = r.pos;
  } else {  // cylinder - cylinder (the abdomen's segments among themselves; they never come near each other): the general iteration
    const Result r = cyl_cyl(g1, g2);
    out.dist = r.dist; out.n = r.n; out.pos = r.pos;
  }
  return out;
}
// the pair classes `collide<false>` leaves out (ellipsoid - cylinder: a wing near the abdomen; cylinder - cylinder)
CVX_FN bool rare_class(int type1, int type2) { return type1 >= ELLIPSOID && type2 == CYLINDER; }

}  // namespace cvx
