// dev_model.hpp - host-side construction of the device-resident model tables.
//
// Input: the compiled model blob (flybody_amd/model/blob.py layout).  Output: one HBM arena holding
// lane-major ("[field][lane]") tables so that a wavefront's per-dof / per-link reads coalesce, plus the
// index tables that drive the sparse factorisation and solves.  Everything here runs once per handle.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

// Pointers stored inside structs that live in device memory lose their address space when the compiler cannot trace
// them back to a kernel argument, and are then dereferenced with FLAT loads (slower, and they tie the LDS and vector
// memory wait counters together).  Declaring them global on the device side makes every table read a global_load.
#if defined(__HIP_DEVICE_COMPILE__)
#define FFE_GLOBAL __attribute__((address_space(1)))
#define FFE_CONST __attribute__((address_space(4)))  // read-only struct of table pointers: uniform fields come by s_load
#else
#define FFE_GLOBAL
#define FFE_CONST
#endif

namespace ffe {

constexpr int kWave = 64;
constexpr int kMaxLink = 19;  // LDS capacity, sized to the flight model (19 links)
constexpr int kMaxDof = 42;   // LDS capacity (flight model: 42 dofs)
constexpr int kMaxM = 422;    // LDS capacity (flight model: 421 entries)
constexpr int kMaxAct = 16;
constexpr int kMaxWrap = 8;   // transmission terms per actuator (joint: 1, fixed tendon: its joints)
constexpr int kMaxObsJ = 32;
constexpr int kMaxFuture = 8;
constexpr int kLanePad = 64;  // every lane-major table is padded to a full wave
constexpr int kMaxGeom = 72;   // collision geoms (70), padded
constexpr int kMaxPair = 384;  // candidate collision pairs (362), padded to whole waves

// ------------------------------------------------------------------------------------------------ blob
struct Tensor {
  int dtype = 0;  // 0 f64, 1 i32
  std::vector<uint32_t> dims;
  const unsigned char *data = nullptr;
  size_t count = 0;
  double f(size_t i) const { double v; std::memcpy(&v, data + 8 * i, 8); return v; }
  int32_t i(size_t k) const { int32_t v; std::memcpy(&v, data + 4 * k, 4); return v; }
};

class Blob {
 public:
  Blob(const void *p, size_t n) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    if (n < 12 || std::memcmp(b, "FFMB", 4) != 0) throw std::runtime_error("model blob: bad magic");
    uint32_t ver, cnt;
    std::memcpy(&ver, b + 4, 4);
    std::memcpy(&cnt, b + 8, 4);
    if (ver != 1) throw std::runtime_error("model blob: unsupported version");
    size_t off = 12;
    for (uint32_t k = 0; k < cnt; k++) {
      uint16_t nl;
      if (off + 2 > n) throw std::runtime_error("model blob: truncated");
      std::memcpy(&nl, b + off, 2); off += 2;
      std::string name(reinterpret_cast<const char *>(b + off), nl); off += nl;
      Tensor t;
      t.dtype = b[off]; int ndim = b[off + 1]; off += 2;
      t.count = 1;
      for (int d = 0; d < ndim; d++) { uint32_t v; std::memcpy(&v, b + off, 4); off += 4; t.dims.push_back(v); t.count *= v; }
      off += (8 - off % 8) % 8;
      t.data = b + off;
      off += t.count * (t.dtype == 0 ? 8 : 4);
      if (off > n) throw std::runtime_error("model blob: truncated tensor " + name);
      tensors_[name] = t;
    }
  }
  bool has(const std::string &name) const { return tensors_.count(name) != 0; }
  const Tensor &get(const std::string &name) const {
    auto it = tensors_.find(name);
    if (it == tensors_.end()) throw std::runtime_error("model blob: missing tensor " + name);
    return it->second;
  }

 private:
  std::map<std::string, Tensor> tensors_;
};

// ------------------------------------------------------------------------------------------------ device view
// Pointers into the arena (device addresses once uploaded).  Lane-major tables: element (c, lane) at [c*64+lane].
struct DevModel {
  int nlink, nv, nq, nu, nM, nrootrec, nobsj, nwing, naction, nsub;
  float h, gx, gy, gz, total_mass;
  // per dof
  const int FFE_GLOBAL *d_link, *d_madr, *d_depth, *d_kind, *d_qadr, *d_limited, *d_act_id, *d_ndesc;  // d_act_id: [2][64]
  const unsigned int FFE_GLOBAL *pairtab;  // [256] elimination pairs (s | t << 8), sorted by t then s
  // Branch-parallel triangular solves: the dofs behind the root chain (the free joint's 6 dofs) split into independent
  // branches (abdomen chain, head subtree, wings, halteres); br_seq[w][lane] packs, 4 bytes per word, the elimination order
  // (leaf end first) of the branch the lane's dof belongs to, 0xff-padded; root-chain lanes and idle lanes hold 0xff only.
  const unsigned int FFE_GLOBAL *br_seq;   // [4][64]
  int nbr_steps, nroot;                    // longest branch sequence; dofs of the root chain (6)
  int any_b2;                              // some actuator has a velocity term in its affine bias (none in the flight model)
  // Branch-parallel factorisation schedule: step t eliminates the t-th pivot of every branch at once; the pair updates
  // M(a_s, a_t) -= M(k, a_s) M(k, a_t) / M(k, k) of all those pivots are dealt over the 64 lanes in rounds.  One word per
  // (round, lane): row start of the pivot | s << 9 | t << 14 | target << 19 | last round of its step << 30 | valid << 31, the
  // target being an entry of M, or 512 + 21 * branch + e for entry e of the root block (rows of the free joint's dofs), which
  // concurrent branches accumulate in private copies (folded in before the root chain is eliminated).
  const unsigned int FFE_GLOBAL *fsched;   // [fs_rounds + 1][64]
  int fs_branch_rounds, fs_rounds, nbranch;
  const float FFE_GLOBAL *d_axis, *d_arm, *d_damp, *d_stiff, *d_sref, *d_lo, *d_hi, *d_margin, *d_invw, *d_K, *d_B, *d_solimp,
      *d_act_coef, *d_qpos0;  // d_axis [3][64]; d_solimp [5][64]; d_act_coef [2][64]
  // per link
  const int FFE_GLOBAL *l_parent, *l_dofadr, *l_dofnum, *l_sub, *l_reckind, *l_recell;
  const unsigned int FFE_GLOBAL *l_anc;  // [2][64] ancestor links packed as bytes, nearest first, 0xff = none
  const float FFE_GLOBAL *l_pos, *l_quat, *l_ipos, *l_imat, *l_inertia, *l_mass, *l_recpos, *l_recmat, *l_reccoef;
  // fluid records on the root link, one per lane
  const float FFE_GLOBAL *rr_pos, *rr_mat, *rr_coef;
  // ellipsoid parameter blocks [nell][32]
  const float FFE_GLOBAL *ell;
  // sparse-M index tables
  const unsigned char FFE_GLOBAL *m_row, *m_col;
  const unsigned short FFE_GLOBAL *colmadr;  // [nM] for entry e = (row k, column a): start of row a
  // actuators
  const int FFE_GLOBAL *a_cl, *a_fl, *a_action, *t_qadr, *t_dof;
  const float FFE_GLOBAL *t_coef;
  const float FFE_GLOBAL *a_gain, *a_b0, *a_b1, *a_b2, *a_clo, *a_chi, *a_flo, *a_fhi;
  // task bookkeeping
  const int FFE_GLOBAL *wing_dof, *wing_qadr, *wing_action, *wing_ctrl, *obsj_qadr, *obsj_dof;  // wing_ctrl: ctrl slot fed by the wing's action entry
  int user_action;
  const float FFE_GLOBAL *qpos0;  // [nq]
  // ---- collision geoms (ref: tasks/base.py:299-302 disables the floor only: the fly's own geoms still collide in flight).  All
  //      70 geoms in model order, frames in their link (cg_pos [3][kMaxGeom], cg_quat [4][.], cg_size [3][.]); cp_pair = the static
  //      candidate list (flybody_amd/model/reach.py), g1 | g2 << 8 with g1 the lower type code (mj_collision's order), 0xffff padding;
  //      l_dofmask[l] = dofs that move link l.  ncg == 0: a model blob without geoms (contact-free configurations).
  int ncg, ncp;
  float c_margin, c_gap, c_K, c_B, c_solimp[5];
  unsigned long long cg_mmask[2];  // geoms that carry the margin / gap (labrum, claws)
  const int FFE_GLOBAL *cg_link, *cg_type;
  const float FFE_GLOBAL *cg_pos, *cg_quat, *cg_size, *cg_brad, *cg_invw;
  const unsigned short FFE_GLOBAL *cp_pair;
  const unsigned long long FFE_GLOBAL *l_dofmask;
};

struct BoxCoef { float c[8]; };  // visc_ang, visc_lin, quad_lin[3], quad_ang[3]

inline BoxCoef box_coefs(const double *box, double rho, double beta) {
  const double kPi = 3.14159265358979323846;
  BoxCoef o;
  double diam = (box[0] + box[1] + box[2]) / 3.0;
  o.c[0] = static_cast<float>(beta > 0 ? kPi * diam * diam * diam * beta : 0.0);
  o.c[1] = static_cast<float>(beta > 0 ? 3.0 * kPi * diam * beta : 0.0);
  o.c[2] = static_cast<float>(0.5 * rho * box[1] * box[2]);
  o.c[3] = static_cast<float>(0.5 * rho * box[0] * box[2]);
  o.c[4] = static_cast<float>(0.5 * rho * box[0] * box[1]);
  o.c[5] = static_cast<float>(rho * box[0] * (std::pow(box[1], 4) + std::pow(box[2], 4)) / 64.0);
  o.c[6] = static_cast<float>(rho * box[1] * (std::pow(box[0], 4) + std::pow(box[2], 4)) / 64.0);
  o.c[7] = static_cast<float>(rho * box[2] * (std::pow(box[0], 4) + std::pow(box[1], 4)) / 64.0);
  return o;
}

inline void quat2mat_d(const double *q, double *m) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}

// Bump allocator for the arena; returns byte offsets, 16-byte aligned.
class Arena {
 public:
  template <typename T>
  size_t put(const std::vector<T> &v) {
    size_t off = (buf_.size() + 15) & ~size_t(15);
    buf_.resize(off + v.size() * sizeof(T) + 16);
    std::memcpy(buf_.data() + off, v.data(), v.size() * sizeof(T));
    return off;
  }
  const std::vector<unsigned char> &bytes() const { return buf_; }

 private:
  std::vector<unsigned char> buf_;
};

struct HostModel {
  DevModel view{};           // pointers hold arena *offsets* until fixup()
  std::vector<unsigned char> arena;
  std::vector<float> action_min, action_max;
  std::vector<double> qpos0;
  int nobs = 0;

  void fixup(DevModel &dst, const unsigned char *base) const {
    dst = view;
    auto fix = [&](auto &p) {
      using P = std::remove_reference_t<decltype(p)>;
      p = (P)((size_t)base + (size_t)p);  // C-style: the member may carry a device address-space qualifier
    };
    fix(dst.d_link); fix(dst.d_madr); fix(dst.d_depth); fix(dst.d_kind); fix(dst.d_qadr);
    fix(dst.d_limited); fix(dst.d_act_id); fix(dst.d_ndesc); fix(dst.pairtab); fix(dst.br_seq); fix(dst.fsched); fix(dst.d_axis); fix(dst.d_arm); fix(dst.d_damp); fix(dst.d_stiff);
    fix(dst.d_sref); fix(dst.d_lo); fix(dst.d_hi); fix(dst.d_margin); fix(dst.d_invw); fix(dst.d_K); fix(dst.d_B);
    fix(dst.d_solimp); fix(dst.d_act_coef); fix(dst.d_qpos0);
    fix(dst.l_anc); fix(dst.l_parent); fix(dst.l_dofadr); fix(dst.l_dofnum); fix(dst.l_sub); fix(dst.l_reckind); fix(dst.l_recell);
    fix(dst.l_pos); fix(dst.l_quat); fix(dst.l_ipos); fix(dst.l_imat); fix(dst.l_inertia); fix(dst.l_mass);
    fix(dst.l_recpos); fix(dst.l_recmat); fix(dst.l_reccoef);
    fix(dst.rr_pos); fix(dst.rr_mat); fix(dst.rr_coef); fix(dst.ell);
    fix(dst.m_row); fix(dst.m_col); fix(dst.colmadr);
    fix(dst.a_cl); fix(dst.a_fl); fix(dst.a_action); fix(dst.t_qadr); fix(dst.t_dof); fix(dst.t_coef);
    fix(dst.a_gain); fix(dst.a_b0); fix(dst.a_b1); fix(dst.a_b2); fix(dst.a_clo); fix(dst.a_chi); fix(dst.a_flo);
    fix(dst.a_fhi);
    fix(dst.wing_dof); fix(dst.wing_qadr); fix(dst.wing_action); fix(dst.wing_ctrl); fix(dst.obsj_qadr); fix(dst.obsj_dof);
    fix(dst.qpos0);
    fix(dst.cg_link); fix(dst.cg_type); fix(dst.cg_pos); fix(dst.cg_quat); fix(dst.cg_size); fix(dst.cg_brad); fix(dst.cg_invw);
    fix(dst.cp_pair); fix(dst.l_dofmask);
  }
};

template <typename P>
inline void set_off(P &p, size_t off) { p = (P)off; }  // arena offset now, device address after fixup()

inline HostModel build_host_model(const Blob &b) {
  HostModel H;
  DevModel &V = H.view;
  Arena A;
  const Tensor &opt = b.get("opt");
  const double h = opt.f(0), rho = opt.f(1), beta = opt.f(2);
  const Tensor &lparent = b.get("link_parent");
  const int nl = static_cast<int>(lparent.count);
  const Tensor &dofpar = b.get("dof_parentid");
  const int nv = static_cast<int>(dofpar.count);
  const int nq = static_cast<int>(b.get("qpos0").count);
  const int nu = static_cast<int>(b.get("act_trntype").count);
  if (nl > kMaxLink || nv > kMaxDof || nu > kMaxAct) throw std::runtime_error("model exceeds kernel capacities");
  V.nlink = nl; V.nv = nv; V.nq = nq; V.nu = nu;
  V.h = static_cast<float>(h);
  V.gx = static_cast<float>(opt.f(3)); V.gy = static_cast<float>(opt.f(4)); V.gz = static_cast<float>(opt.f(5));

  auto lane_f = [&](int comps) { return std::vector<float>(static_cast<size_t>(comps) * kLanePad, 0.f); };
  auto lane_i = [&](int comps, int fill = 0) { return std::vector<int>(static_cast<size_t>(comps) * kLanePad, fill); };

  // ---- body -> link / dof maps -----------------------------------------------------------------
  const Tensor &body_link = b.get("body_link");
  const Tensor &dof_body = b.get("dof_bodyid");
  const Tensor &dof_jnt = b.get("dof_jntid");
  const Tensor &jnt_type = b.get("jnt_type");
  const Tensor &jnt_qadr = b.get("jnt_qposadr");
  const Tensor &jnt_dadr = b.get("jnt_dofadr");
  const Tensor &jnt_axis = b.get("jnt_axis");
  const Tensor &jnt_pos = b.get("jnt_pos");
  for (size_t k = 0; k < jnt_pos.count; k++)
    if (jnt_pos.f(k) != 0.0) throw std::runtime_error("joint anchors away from the body origin are not supported");
  if (jnt_type.i(0) != 0) throw std::runtime_error("root joint must be free");

  auto d_parent = lane_i(1, -1), d_link = lane_i(1), d_madr = lane_i(1), d_depth = lane_i(1), d_kind = lane_i(1),
       d_qadr = lane_i(1), d_limited = lane_i(1), d_act_id = lane_i(2, -1);
  auto d_axis = lane_f(3), d_arm = lane_f(1), d_damp = lane_f(1), d_stiff = lane_f(1), d_sref = lane_f(1),
       d_lo = lane_f(1), d_hi = lane_f(1), d_margin = lane_f(1), d_invw = lane_f(1), d_K = lane_f(1), d_B = lane_f(1),
       d_solimp = lane_f(5), d_act_coef = lane_f(2), d_qpos0 = lane_f(1);
  std::vector<unsigned char> m_row, m_col;
  int maxdepth = 0;
  for (int d = 0; d < nv; d++) {
    d_parent[d] = dofpar.i(d);
    d_link[d] = body_link.i(dof_body.i(d));
    int j = dof_jnt.i(d), jt = jnt_type.i(j);
    int within = d - jnt_dadr.i(j);
    if (jt == 0) { d_kind[d] = within < 3 ? 0 : 1; d_qadr[d] = within % 3; }
    else if (jt == 3) { d_kind[d] = 2; d_qadr[d] = jnt_qadr.i(j); }
    else throw std::runtime_error("only free and hinge joints are supported");
    for (int c = 0; c < 3; c++) d_axis[c * kLanePad + d] = static_cast<float>(jnt_axis.f(3 * j + c));
    d_arm[d] = static_cast<float>(b.get("dof_armature").f(d));
    d_damp[d] = static_cast<float>(b.get("dof_damping").f(d));
    d_invw[d] = static_cast<float>(b.get("dof_invweight0").f(d));
    if (jt == 3) {
      d_qpos0[d] = static_cast<float>(b.get("qpos0").f(jnt_qadr.i(j)));
      d_stiff[d] = static_cast<float>(b.get("jnt_stiffness").f(j));
      d_sref[d] = static_cast<float>(b.get("qpos_spring").f(jnt_qadr.i(j)));
      d_limited[d] = b.get("jnt_limited").i(j);
      d_lo[d] = static_cast<float>(b.get("jnt_range").f(2 * j));
      d_hi[d] = static_cast<float>(b.get("jnt_range").f(2 * j + 1));
      d_margin[d] = static_cast<float>(b.get("jnt_margin").f(j));
      const Tensor &si = b.get("jnt_solimp"), &sr = b.get("jnt_solref");
      double s[5];
      for (int c = 0; c < 5; c++) s[c] = si.f(5 * j + c);
      s[0] = std::fmin(std::fmax(s[0], 1e-4), 0.9999); s[1] = std::fmin(std::fmax(s[1], 1e-4), 0.9999);
      s[2] = std::fmax(0.0, s[2]); s[3] = std::fmin(std::fmax(s[3], 1e-4), 0.9999); s[4] = std::fmax(1.0, s[4]);
      for (int c = 0; c < 5; c++) d_solimp[c * kLanePad + d] = static_cast<float>(s[c]);
      double tc = sr.f(2 * j), dr = sr.f(2 * j + 1), dmax = s[1], K, B;
      if (tc > 0) {
        tc = std::fmax(tc, 2 * h);  // refsafe
        K = 1.0 / std::fmax(1e-15, dmax * dmax * tc * tc * dr * dr);
        B = 2.0 / std::fmax(1e-15, dmax * tc);
      } else { K = -tc / std::fmax(1e-15, dmax * dmax); B = -dr / std::fmax(1e-15, dmax); }
      d_K[d] = static_cast<float>(K); d_B[d] = static_cast<float>(B);
    }
    d_madr[d] = static_cast<int>(m_row.size());
    int depth = 0;
    for (int a = d; a >= 0; a = dofpar.i(a)) { m_row.push_back(static_cast<unsigned char>(d)); m_col.push_back(static_cast<unsigned char>(a)); depth++; }
    d_depth[d] = depth;
    maxdepth = depth > maxdepth ? depth : maxdepth;
  }
  // descendants of a dof are the contiguous index range that follows it (bodies and dofs are in depth-first order)
  auto d_ndesc = lane_i(1);
  for (int d = 0; d < nv; d++)
    for (int a = dofpar.i(d); a >= 0; a = dofpar.i(a)) d_ndesc[a]++;
  for (int d = 0; d < nv; d++)
    for (int e = d + 1; e <= d + d_ndesc[d]; e++) {
      bool ok = false;
      for (int a = dofpar.i(e); a >= 0; a = dofpar.i(a)) ok |= (a == d);
      if (!ok) throw std::runtime_error("dof tree is not in depth-first order");
    }
  std::vector<unsigned int> pairtab(4 * kWave, 0xffffu);
  {
    int p = 0;
    for (int t = 1; t <= 22 && p < 4 * kWave; t++)
      for (int sidx = 1; sidx <= t && p < 4 * kWave; sidx++) pairtab[p++] = static_cast<unsigned>(sidx) | (static_cast<unsigned>(t) << 8);
  }
  if ((maxdepth - 1) * maxdepth / 2 > 4 * kWave) throw std::runtime_error("dof chains too deep for the pair table");
  // root chain = the free joint's dofs 0..5 (each the parent of the next); every other dof hangs off dof 5 through a branch
  std::vector<unsigned int> br_seq(4 * kWave, 0xffffffffu);
  int nroot = 0, nbr_steps = 0;
  {
    while (nroot < nv && dofpar.i(nroot) == nroot - 1 && d_ndesc[nroot] == nv - 1 - nroot) nroot++;
    if (nroot != 6) throw std::runtime_error("expected a 6-dof root chain (free joint) ahead of the branches");
    for (int d = nroot; d < nv; d++) {
      if (dofpar.i(d) != nroot - 1) continue;          // d starts a branch: dofs d .. d + ndesc
      const int len = d_ndesc[d] + 1;
      if (len > 16) throw std::runtime_error("branch longer than the 16-step solve sequence");
      nbr_steps = len > nbr_steps ? len : nbr_steps;
      for (int e = d; e < d + len; e++)
        for (int t = 0; t < len; t++) {               // elimination order: highest dof index first (descendants before ancestors)
          unsigned int &w = br_seq[(t >> 2) * kWave + e];
          w = (w & ~(0xffu << (8 * (t & 3)))) | (static_cast<unsigned>(d + len - 1 - t) << (8 * (t & 3)));
        }
    }
  }
  V.nbr_steps = nbr_steps; V.nroot = nroot;
  std::vector<unsigned int> fsched;
  {
    std::vector<int> bstart, blen;
    for (int d = nroot; d < nv; d++) if (dofpar.i(d) == nroot - 1) { bstart.push_back(d); blen.push_back(d_ndesc[d] + 1); }
    V.nbranch = static_cast<int>(bstart.size());
    if (V.nbranch > 8 || d_madr[nroot] != nroot * (nroot + 1) / 2) throw std::runtime_error("unexpected branch structure behind the root chain");
    auto anc = [&](int k, int sidx) { int a = k; for (int i = 0; i < sidx; i++) a = dofpar.i(a); return a; };
    auto emit_step = [&](const std::vector<std::pair<int, int>> &pivots) {  // (pivot dof, branch or -1 for a root pivot)
      std::vector<unsigned int> words;
      for (auto [k, br] : pivots) {
        const int n = d_depth[k] - 1;
        for (int t = 1; t <= n; t++)
          for (int sidx = 1; sidx <= t; sidx++) {
            const int as = anc(k, sidx);
            int tgt = d_madr[as] + (t - sidx);
            if (br >= 0 && as < nroot) tgt = 512 + 21 * br + tgt;  // root block: private copy of this branch
            if (d_madr[k] >= 512 || tgt >= 1024) throw std::runtime_error("factor schedule field overflow");
            words.push_back(static_cast<unsigned>(d_madr[k]) | (static_cast<unsigned>(sidx) << 9) | (static_cast<unsigned>(t) << 14) |
                            (static_cast<unsigned>(tgt) << 19) | 0x80000000u);
          }
      }
      const size_t rounds = (words.size() + kWave - 1) / kWave;
      for (size_t r = 0; r < rounds; r++)
        for (int l = 0; l < kWave; l++) {
          const size_t i = r * kWave + l;
          unsigned w = i < words.size() ? words[i] : 0u;
          if (r + 1 == rounds) w |= 0x40000000u;  // (every lane of the round carries the flag)
          fsched.push_back(w);
        }
    };
    for (int t = 0; t < nbr_steps; t++) {
      std::vector<std::pair<int, int>> piv;
      for (int bi = 0; bi < V.nbranch; bi++) if (t < blen[bi]) piv.push_back({bstart[bi] + blen[bi] - 1 - t, bi});
      emit_step(piv);
    }
    V.fs_branch_rounds = static_cast<int>(fsched.size() / kWave);
    for (int k = nroot - 1; k > 0; k--) emit_step({{k, -1}});
    V.fs_rounds = static_cast<int>(fsched.size() / kWave);
    if (V.fs_rounds - V.fs_branch_rounds > 6) throw std::runtime_error("root chain longer than the kernel unrolls");
    for (int l = 0; l < 10 * kWave; l++) fsched.push_back(0u);  // read-ahead padding
  }
  const int nM = static_cast<int>(m_row.size());
  if (nM > kMaxM) throw std::runtime_error("mass matrix exceeds kernel capacity");
  V.nM = nM;
  // ---- links ---------------------------------------------------------------------------------------
  auto l_parent = lane_i(1, -1), l_dofadr = lane_i(1), l_dofnum = lane_i(1), l_sub = lane_i(1), l_reckind = lane_i(1),
       l_recell = lane_i(1);
  auto l_pos = lane_f(3), l_quat = lane_f(4), l_ipos = lane_f(3), l_imat = lane_f(9), l_inertia = lane_f(3),
       l_mass = lane_f(1), l_recpos = lane_f(3), l_recmat = lane_f(9), l_reccoef = lane_f(8);
  double total_mass = 0;
  for (int k = 0; k < nl; k++) {
    l_parent[k] = lparent.i(k);
    l_dofadr[k] = b.get("link_dofadr").i(k);
    l_dofnum[k] = b.get("link_dofnum").i(k);
    if (k > 0 && l_dofnum[k] > 3) throw std::runtime_error("a non-root link carries more than 3 hinges");
    l_sub[k] = b.get("link_subtree").i(k);
    for (int c = 0; c < 3; c++) l_pos[c * kLanePad + k] = static_cast<float>(b.get("link_pos").f(3 * k + c));
    for (int c = 0; c < 4; c++) l_quat[c * kLanePad + k] = static_cast<float>(b.get("link_quat").f(4 * k + c));
    for (int c = 0; c < 3; c++) l_ipos[c * kLanePad + k] = static_cast<float>(b.get("link_ipos").f(3 * k + c));
    double q[4], mat[9];
    for (int c = 0; c < 4; c++) q[c] = b.get("link_iquat").f(4 * k + c);
    quat2mat_d(q, mat);
    for (int c = 0; c < 9; c++) l_imat[c * kLanePad + k] = static_cast<float>(mat[c]);
    for (int c = 0; c < 3; c++) l_inertia[c * kLanePad + k] = static_cast<float>(b.get("link_inertia").f(3 * k + c));
    l_mass[k] = static_cast<float>(b.get("link_mass").f(k));
    total_mass += b.get("link_mass").f(k);
  }
  V.total_mass = static_cast<float>(total_mass);
  std::vector<unsigned int> l_anc(2 * kLanePad, 0xffffffffu);
  for (int k = 0; k < nl; k++) {
    int a = lparent.i(k), it = 0;
    unsigned long long packed = ~0ULL;
    while (a >= 0) {
      if (it >= 8) throw std::runtime_error("link tree deeper than 8 ancestors");
      packed = (packed & ~(0xffULL << (8 * it))) | (static_cast<unsigned long long>(a) << (8 * it));
      a = lparent.i(a); it++;
    }
    l_anc[k] = static_cast<unsigned int>(packed & 0xffffffffu);
    l_anc[kLanePad + k] = static_cast<unsigned int>(packed >> 32);
  }
  // fluid records: root link's go one per lane, every other link carries at most one record of its own
  auto rr_pos = lane_f(3), rr_mat = lane_f(9), rr_coef = lane_f(8);
  int nrr = 0;
  const Tensor &fb_link = b.get("fbox_link");
  for (size_t r = 0; r < fb_link.count; r++) {
    int k = fb_link.i(r);
    BoxCoef bc = box_coefs(reinterpret_cast<const double *>(b.get("fbox_box").data) + 3 * r, rho, beta);
    if (k == 0) {
      if (nrr >= kWave) throw std::runtime_error("too many fluid records on the root link");
      for (int c = 0; c < 3; c++) rr_pos[c * kLanePad + nrr] = static_cast<float>(b.get("fbox_pos").f(3 * r + c));
      for (int c = 0; c < 9; c++) rr_mat[c * kLanePad + nrr] = static_cast<float>(b.get("fbox_mat").f(9 * r + c));
      for (int c = 0; c < 8; c++) rr_coef[c * kLanePad + nrr] = bc.c[c];
      nrr++;
    } else {
      if (l_reckind[k] != 0) throw std::runtime_error("more than one fluid record on a non-root link");
      l_reckind[k] = 1;
      for (int c = 0; c < 3; c++) l_recpos[c * kLanePad + k] = static_cast<float>(b.get("fbox_pos").f(3 * r + c));
      for (int c = 0; c < 9; c++) l_recmat[c * kLanePad + k] = static_cast<float>(b.get("fbox_mat").f(9 * r + c));
      for (int c = 0; c < 8; c++) l_reccoef[c * kLanePad + k] = bc.c[c];
    }
  }
  V.nrootrec = nrr;
  const Tensor &fe_link = b.get("fell_link");
  std::vector<float> ell(32 * (fe_link.count ? fe_link.count : 1), 0.f);
  const double kPi = 3.14159265358979323846;
  for (size_t g = 0; g < fe_link.count; g++) {
    int k = fe_link.i(g);
    if (k == 0 || l_reckind[k] != 0) throw std::runtime_error("unsupported ellipsoid fluid geom placement");
    l_reckind[k] = 2; l_recell[k] = static_cast<int>(g);
    for (int c = 0; c < 3; c++) l_recpos[c * kLanePad + k] = static_cast<float>(b.get("fell_pos").f(3 * g + c));
    for (int c = 0; c < 9; c++) l_recmat[c * kLanePad + k] = static_cast<float>(b.get("fell_mat").f(9 * g + c));
    const double *s = reinterpret_cast<const double *>(b.get("fell_size").data) + 3 * g;
    const double *cf = reinterpret_cast<const double *>(b.get("fell_coef").data) + 12 * g;
    double dmax = std::fmax(std::fmax(s[0], s[1]), s[2]), dmin = std::fmin(std::fmin(s[0], s[1]), s[2]);
    double dmid = s[0] + s[1] + s[2] - dmax - dmin;
    double vol = 4.0 / 3.0 * kPi * s[0] * s[1] * s[2], eqD = 2.0 / 3.0 * (s[0] + s[1] + s[2]);
    float *e = ell.data() + 32 * g;
    e[0] = static_cast<float>(cf[0]);                                  // interaction coefficient
    for (int c = 0; c < 3; c++) e[1 + c] = static_cast<float>(rho * cf[6 + c]);   // rho * virtual mass
    for (int c = 0; c < 3; c++) e[4 + c] = static_cast<float>(rho * cf[9 + c]);   // rho * virtual inertia
    e[7] = static_cast<float>(cf[5] * rho * vol);                      // Magnus
    e[8] = static_cast<float>(s[1] * s[2]); e[9] = static_cast<float>(s[2] * s[0]); e[10] = static_cast<float>(s[0] * s[1]);
    e[11] = static_cast<float>(cf[4] * rho);                           // Kutta
    e[12] = static_cast<float>(kPi * dmax * dmid);                     // A_max
    e[13] = static_cast<float>(cf[1]); e[14] = static_cast<float>(cf[2]); e[15] = static_cast<float>(cf[3]);  // blunt, slender, angular
    e[16] = static_cast<float>(beta * 3.0 * kPi * eqD);
    e[17] = static_cast<float>(beta * kPi * eqD * eqD * eqD);
    e[18] = static_cast<float>(8.0 / 15.0 * kPi * dmid * dmax * dmax * dmax * dmax);  // I_max
    e[19] = static_cast<float>(8.0 / 15.0 * kPi * s[0] * std::pow(std::fmax(s[1], s[2]), 4));
    e[20] = static_cast<float>(8.0 / 15.0 * kPi * s[1] * std::pow(std::fmax(s[0], s[2]), 4));
    e[21] = static_cast<float>(8.0 / 15.0 * kPi * s[2] * std::pow(std::fmax(s[0], s[1]), 4));
    e[22] = static_cast<float>(rho);
  }

  // ---- actuators -----------------------------------------------------------------------------------
  std::vector<int> a_trn(kMaxAct, 0), a_dof(kMaxAct, 0), a_qadr(kMaxAct, 0), a_cl(kMaxAct, 0), a_fl(kMaxAct, 0),
      a_action(kMaxAct, -1), a_wrap_off(kMaxAct + 1, 0), w_qadr, w_dof;
  std::vector<float> a_gain(kMaxAct, 0), a_b0(kMaxAct, 0), a_b1(kMaxAct, 0), a_b2(kMaxAct, 0), a_clo(kMaxAct, 0),
      a_chi(kMaxAct, 0), a_flo(kMaxAct, 0), a_fhi(kMaxAct, 0), w_coef;
  const Tensor &ten_adr = b.get("ten_adr"), &ten_num = b.get("ten_num"), &wrap_dof = b.get("wrap_dof"), &wrap_coef = b.get("wrap_coef");
  auto add_coupling = [&](int dof, int act, float coef) {
    for (int s = 0; s < 2; s++)
      if (d_act_id[s * kLanePad + dof] < 0) { d_act_id[s * kLanePad + dof] = act; d_act_coef[s * kLanePad + dof] = coef; return; }
    throw std::runtime_error("more than two actuators drive one dof");
  };
  for (int u = 0; u < nu; u++) {
    int trn = b.get("act_trntype").i(u), id = b.get("act_trnid").i(u);
    double gear = b.get("act_gear").f(u);
    if (b.get("act_dyntype").i(u) != 0) throw std::runtime_error("actuator activation dynamics are not supported yet");
    a_trn[u] = trn;
    a_gain[u] = static_cast<float>(b.get("act_gainprm").f(u));
    a_b0[u] = static_cast<float>(b.get("act_biasprm").f(3 * u));
    a_b1[u] = static_cast<float>(b.get("act_biasprm").f(3 * u + 1) * gear);
    a_b2[u] = static_cast<float>(b.get("act_biasprm").f(3 * u + 2) * gear);
    a_cl[u] = b.get("act_ctrllimited").i(u);
    a_clo[u] = static_cast<float>(b.get("act_ctrlrange").f(2 * u)); a_chi[u] = static_cast<float>(b.get("act_ctrlrange").f(2 * u + 1));
    a_fl[u] = b.get("act_forcelimited").i(u);
    a_flo[u] = static_cast<float>(b.get("act_forcerange").f(2 * u)); a_fhi[u] = static_cast<float>(b.get("act_forcerange").f(2 * u + 1));
    a_action[u] = b.get("act_action").i(u);
    a_wrap_off[u] = static_cast<int>(w_qadr.size());
    if (trn == 0) {
      a_dof[u] = jnt_dadr.i(id); a_qadr[u] = jnt_qadr.i(id);
      add_coupling(a_dof[u], u, static_cast<float>(gear));
    } else if (trn == 1) {
      for (int w = ten_adr.i(id); w < ten_adr.i(id) + ten_num.i(id); w++) {
        int dof = wrap_dof.i(w);
        w_dof.push_back(dof); w_qadr.push_back(jnt_qadr.i(dof_jnt.i(dof))); w_coef.push_back(static_cast<float>(wrap_coef.f(w)));
        add_coupling(dof, u, static_cast<float>(gear * wrap_coef.f(w)));
      }
    } else throw std::runtime_error("body (adhesion) transmissions are not supported yet");
  }
  for (int u = nu; u <= kMaxAct; u++) a_wrap_off[u] = static_cast<int>(w_qadr.size());
  V.any_b2 = 0;
  for (int u = 0; u < nu; u++) if (a_b2[u] != 0.f) V.any_b2 = 1;
  // transposed, zero-padded transmission table [term][actuator]: every actuator - joint or tendon driven - is a sum of
  // up to kMaxWrap (coef, dof) terms, so the kernel evaluates length and velocity without a data-dependent loop
  std::vector<int> t_qadr(kMaxWrap * kMaxAct, 0), t_dof(kMaxWrap * kMaxAct, 0);
  std::vector<float> t_coef(kMaxWrap * kMaxAct, 0.f);
  for (int u = 0; u < nu; u++) {
    if (a_trn[u] == 0) { t_qadr[u] = a_qadr[u]; t_dof[u] = a_dof[u]; t_coef[u] = 1.f; }
    else {
      int n = a_wrap_off[u + 1] - a_wrap_off[u];
      if (n > kMaxWrap) throw std::runtime_error("tendon with too many joints");
      for (int w = 0; w < n; w++) {
        t_qadr[w * kMaxAct + u] = w_qadr[a_wrap_off[u] + w]; t_dof[w * kMaxAct + u] = w_dof[a_wrap_off[u] + w];
        t_coef[w * kMaxAct + u] = w_coef[a_wrap_off[u] + w];
      }
    }
  }
  if (w_qadr.empty()) { w_qadr.push_back(0); w_dof.push_back(0); w_coef.push_back(0.f); }

  // ---- task bookkeeping ----------------------------------------------------------------------------
  const Tensor &wing_jnt = b.get("wing_jnt"), &wing_act = b.get("wing_action"), &obs_jnt = b.get("obs_jnt");
  std::vector<int> wing_dof(8, 0), wing_qadr(8, 0), wing_action(8, 0), obsj_qadr(kMaxObsJ, 0), obsj_dof(kMaxObsJ, 0);
  V.nwing = static_cast<int>(wing_jnt.count);
  std::vector<int> wing_ctrl(8, -1);
  for (int k = 0; k < V.nwing; k++) {
    wing_dof[k] = jnt_dadr.i(wing_jnt.i(k)); wing_qadr[k] = jnt_qadr.i(wing_jnt.i(k)); wing_action[k] = wing_act.i(k);
    for (int u = 0; u < nu; u++)
      if (a_action[u] == wing_action[k]) {
        if (wing_ctrl[k] >= 0) throw std::runtime_error("a wing action entry feeds more than one actuator");
        wing_ctrl[k] = u;
      }
  }
  std::vector<unsigned short> colmadr(m_row.size() + 8, 0);
  for (size_t e = 0; e < m_row.size(); e++) colmadr[e] = static_cast<unsigned short>(d_madr[m_col[e]]);
  V.nobsj = static_cast<int>(obs_jnt.count);
  if (V.nobsj > kMaxObsJ) throw std::runtime_error("too many observable joints");
  for (int k = 0; k < V.nobsj; k++) { obsj_qadr[k] = jnt_qadr.i(obs_jnt.i(k)); obsj_dof[k] = jnt_dadr.i(obs_jnt.i(k)); }
  V.user_action = b.get("user_action").count ? b.get("user_action").i(0) : -1;
  V.naction = static_cast<int>(b.get("action_min").count);
  for (int k = 0; k < V.naction; k++) { H.action_min.push_back(static_cast<float>(b.get("action_min").f(k))); H.action_max.push_back(static_cast<float>(b.get("action_max").f(k))); }
  std::vector<float> qpos0(nq);
  for (int k = 0; k < nq; k++) { qpos0[k] = static_cast<float>(b.get("qpos0").f(k)); H.qpos0.push_back(b.get("qpos0").f(k)); }
  const Tensor &site = b.get("site");
  for (int c = 1; c <= 3; c++) if (site.f(c) != 0.0) throw std::runtime_error("sensor site must sit at the root body origin");
  if (site.f(4) != 1.0) throw std::runtime_error("sensor site must share the root body orientation");

  // ---- collision ------------------------------------------------------------------------------------
  std::vector<int> cg_link(kMaxGeom, 0), cg_type(kMaxGeom, 0);
  std::vector<float> cg_pos(3 * kMaxGeom, 0.f), cg_quat(4 * kMaxGeom, 0.f), cg_size(3 * kMaxGeom, 0.f), cg_brad(kMaxGeom, 0.f), cg_invw(kMaxGeom, 0.f);
  std::vector<unsigned short> cp_pair(kMaxPair, 0xffffu);
  std::vector<unsigned long long> l_dofmask(kMaxLink + 1, 0ull);
  V.ncg = V.ncp = 0; V.cg_mmask[0] = V.cg_mmask[1] = 0ull;
  V.c_margin = V.c_gap = V.c_K = V.c_B = 0.f;
  for (int d = 0; d < nv; d++) {  // a dof moves its own link and every link below it
    const int lk = d_link[d];
    for (int k = 0; k < nl; k++) {
      bool below = false;
      for (int a = k; a >= 0; a = lparent.i(a)) below |= (a == lk);
      if (below) l_dofmask[k] |= 1ull << d;
    }
  }
  if (b.has("cgeom_link")) {
    const Tensor &gl = b.get("cgeom_link"), &gp = b.get("cgeom_pos"), &gq = b.get("cgeom_quat"), &gt = b.get("geom_type"), &gs = b.get("geom_size"),
                 &gm = b.get("geom_margin"), &gg = b.get("geom_gap"), &gsr = b.get("geom_solref"), &gsi = b.get("geom_solimp"), &gcd = b.get("geom_condim"),
                 &gb = b.get("geom_bodyid"), &biw = b.get("body_invweight0"), &c1 = b.get("cand_g1"), &c2 = b.get("cand_g2");
    const int ng = static_cast<int>(gl.count);
    if (ng > kMaxGeom || static_cast<int>(c1.count) > kMaxPair) throw std::runtime_error("collision tables exceed kernel capacities");
    V.ncg = ng;
    for (int g = 0; g < ng; g++) {
      const int ty = gt.i(g);
      if (ty < 2 || ty > 5) throw std::runtime_error("flight model: unexpected geom type");
      if (gcd.i(g) != 1) throw std::runtime_error("flight model: fly geoms are expected to be condim 1");
      cg_link[g] = gl.i(g); cg_type[g] = ty;
      for (int c = 0; c < 3; c++) { cg_pos[c * kMaxGeom + g] = static_cast<float>(gp.f(3 * g + c)); cg_size[c * kMaxGeom + g] = static_cast<float>(gs.f(3 * g + c)); }
      for (int c = 0; c < 4; c++) cg_quat[c * kMaxGeom + g] = static_cast<float>(gq.f(4 * g + c));
      const double s0 = gs.f(3 * g), s1 = gs.f(3 * g + 1), s2 = gs.f(3 * g + 2);
      cg_brad[g] = static_cast<float>(ty == 2 ? s0 : ty == 3 ? s0 + s1 : ty == 5 ? std::sqrt(s0 * s0 + s1 * s1) : std::fmax(s0, std::fmax(s1, s2)));
      cg_invw[g] = static_cast<float>(biw.f(2 * gb.i(g)));
      // mj_contactParam mixes equal parameters of equal weights: uniform K, B, solimp over the fly's geoms
      double tc = gsr.f(2 * g), dr = gsr.f(2 * g + 1), dmax = std::fmin(std::fmax(gsi.f(5 * g + 1), 1e-4), 0.9999), K, B;
      if (tc > 0) { tc = std::fmax(tc, 2 * h); K = 1.0 / std::fmax(1e-15, dmax * dmax * tc * tc * dr * dr); B = 2.0 / std::fmax(1e-15, dmax * tc); }
      else { K = -tc / std::fmax(1e-15, dmax * dmax); B = -dr / std::fmax(1e-15, dmax); }
      if (g == 0) { V.c_K = static_cast<float>(K); V.c_B = static_cast<float>(B); for (int c = 0; c < 5; c++) V.c_solimp[c] = static_cast<float>(gsi.f(5 * g + c)); }
      if (static_cast<float>(K) != V.c_K || static_cast<float>(B) != V.c_B) throw std::runtime_error("flight model: geom solref is expected to be uniform");
      for (int c = 0; c < 5; c++) if (static_cast<float>(gsi.f(5 * g + c)) != V.c_solimp[c]) throw std::runtime_error("flight model: geom solimp is expected to be uniform");
      if (gm.f(g) != 0) {
        if (V.c_margin != 0.f && (V.c_margin != static_cast<float>(gm.f(g)) || V.c_gap != static_cast<float>(gg.f(g)))) throw std::runtime_error("flight model: one margin class expected");
        V.c_margin = static_cast<float>(gm.f(g)); V.c_gap = static_cast<float>(gg.f(g));
        V.cg_mmask[g >> 6] |= 1ull << (g & 63);
      }
    }
    for (size_t k = 0; k < c1.count; k++) {
      int g1 = c1.i(k), g2 = c2.i(k);
      if (gt.i(g1) > gt.i(g2)) std::swap(g1, g2);  // mj: lower type code first
      if (gl.i(g1) == gl.i(g2)) continue;            // (same link: welded together, filtered by mj_collision already)
      cp_pair[V.ncp++] = static_cast<unsigned short>(g1 | (g2 << 8));
    }
  }
  // ---- pack ------------------------------------------------------------------------------------------
  set_off(V.d_link, A.put(d_link));
  set_off(V.d_madr, A.put(d_madr)); set_off(V.d_depth, A.put(d_depth));
  set_off(V.d_kind, A.put(d_kind)); set_off(V.d_qadr, A.put(d_qadr));
  set_off(V.d_limited, A.put(d_limited)); set_off(V.d_act_id, A.put(d_act_id));
  set_off(V.d_ndesc, A.put(d_ndesc)); set_off(V.pairtab, A.put(pairtab)); set_off(V.br_seq, A.put(br_seq)); set_off(V.fsched, A.put(fsched));
  set_off(V.d_axis, A.put(d_axis)); set_off(V.d_arm, A.put(d_arm));
  set_off(V.d_damp, A.put(d_damp)); set_off(V.d_stiff, A.put(d_stiff));
  set_off(V.d_sref, A.put(d_sref)); set_off(V.d_lo, A.put(d_lo));
  set_off(V.d_hi, A.put(d_hi)); set_off(V.d_margin, A.put(d_margin));
  set_off(V.d_invw, A.put(d_invw)); set_off(V.d_K, A.put(d_K));
  set_off(V.d_B, A.put(d_B)); set_off(V.d_solimp, A.put(d_solimp));
  set_off(V.d_act_coef, A.put(d_act_coef)); set_off(V.d_qpos0, A.put(d_qpos0));
  set_off(V.l_anc, A.put(l_anc));
  set_off(V.l_parent, A.put(l_parent)); set_off(V.l_dofadr, A.put(l_dofadr));
  set_off(V.l_dofnum, A.put(l_dofnum)); set_off(V.l_sub, A.put(l_sub));
  set_off(V.l_reckind, A.put(l_reckind)); set_off(V.l_recell, A.put(l_recell));
  set_off(V.l_pos, A.put(l_pos)); set_off(V.l_quat, A.put(l_quat));
  set_off(V.l_ipos, A.put(l_ipos)); set_off(V.l_imat, A.put(l_imat));
  set_off(V.l_inertia, A.put(l_inertia)); set_off(V.l_mass, A.put(l_mass));
  set_off(V.l_recpos, A.put(l_recpos)); set_off(V.l_recmat, A.put(l_recmat));
  set_off(V.l_reccoef, A.put(l_reccoef));
  set_off(V.rr_pos, A.put(rr_pos)); set_off(V.rr_mat, A.put(rr_mat));
  set_off(V.rr_coef, A.put(rr_coef)); set_off(V.ell, A.put(ell));
  set_off(V.m_row, A.put(m_row)); set_off(V.m_col, A.put(m_col)); set_off(V.colmadr, A.put(colmadr));
  set_off(V.a_cl, A.put(a_cl));
  set_off(V.a_fl, A.put(a_fl)); set_off(V.a_action, A.put(a_action));
  set_off(V.t_qadr, A.put(t_qadr)); set_off(V.t_dof, A.put(t_dof)); set_off(V.t_coef, A.put(t_coef));
  set_off(V.a_gain, A.put(a_gain)); set_off(V.a_b0, A.put(a_b0));
  set_off(V.a_b1, A.put(a_b1)); set_off(V.a_b2, A.put(a_b2));
  set_off(V.a_clo, A.put(a_clo)); set_off(V.a_chi, A.put(a_chi));
  set_off(V.a_flo, A.put(a_flo)); set_off(V.a_fhi, A.put(a_fhi));
  set_off(V.wing_dof, A.put(wing_dof)); set_off(V.wing_qadr, A.put(wing_qadr));
  set_off(V.wing_action, A.put(wing_action)); set_off(V.wing_ctrl, A.put(wing_ctrl));
  set_off(V.obsj_qadr, A.put(obsj_qadr)); set_off(V.obsj_dof, A.put(obsj_dof));
  set_off(V.qpos0, A.put(qpos0));
  set_off(V.cg_link, A.put(cg_link)); set_off(V.cg_type, A.put(cg_type)); set_off(V.cg_pos, A.put(cg_pos)); set_off(V.cg_quat, A.put(cg_quat));
  set_off(V.cg_size, A.put(cg_size)); set_off(V.cg_brad, A.put(cg_brad)); set_off(V.cg_invw, A.put(cg_invw));
  set_off(V.cp_pair, A.put(cp_pair)); set_off(V.l_dofmask, A.put(l_dofmask));
  H.arena = A.bytes();
  return H;
}

}  // namespace ffe
