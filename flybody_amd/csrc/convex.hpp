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
  // ellipsoid: nearest surface point q_i = s_i^2 y_i / (s_i^2 + tau), tau the root of f = sum (s_i y_i / (s_i^2 + tau))^2 - 1 above
  // -min s_i^2.  f is convex and decreasing there, so Newton from a point left of the root (f >= 0) climbs to it monotonically:
  // tau_0 = max_i (s_i |y_i| - s_i^2) is such a point (its own term is 1, every term is <= 1, so 0 <= f <= 2), inside and outside.
  const M3 R = q2m(g.q);
  const V3 y = mtv(R, v);
  const float sx = g.s0 * g.s0, sy = g.s1 * g.s1, sz2 = g.s2 * g.s2;
  const float px = sx * y.x * y.x, py = sy * y.y * y.y, pz = sz2 * y.z * y.z;
  float tau = fmaxf(fmaxf(g.s0 * fabsf(y.x) - sx, g.s1 * fabsf(y.y) - sy), g.s2 * fabsf(y.z) - sz2);
#pragma unroll
  for (int it = 0; it < 6; it++) {
    const float ax = frcp(sx + tau), ay = frcp(sy + tau), az = frcp(sz2 + tau);
    const float bx = px * ax * ax, by = py * ay * ay, bz = pz * az * az;
    const float f = bx + by + bz - 1.f, df = -2.f * (bx * ax + by * ay + bz * az);
    tau = (f > 0.f && df < 0.f) ? tau - f * frcp(df) : tau;
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

}  // namespace cvx
