// ch_analysis.hpp — host-side structural analysis of a flat circuit description.
//
// Plays the role DAECompiler's structural simplification plays for the reference
// (CircuitIRODESystem → IRODESystem: alias elimination + state selection,
// src/circuitodesystem.jl:147-164, doc/circuit_simulation.jmd:211 "feed 40 equations … end up
// solving 5"), restricted to what maps well onto the GPU:
//   1. nodes reached from ground through chains of voltage sources become KNOWN (time-dependent
//      constants): no unknown, no branch current, no KCL row;
//   2. floating 0 V DC sources (the SPICE ammeter idiom, e.g. `VQ Q Q_tmp 0`,
//      test/DFF/DFF_cap_all.cir:9) merge their two nodes (ALIAS);
//   3. the remaining unknowns are partitioned into connected components of the Jacobian graph —
//      independent diagonal blocks (for the tiled DFF array: one 12-unknown block per flip-flop);
//   4. structurally identical blocks share one "class" (gather lists are stored once per class).
// Everything else (branch currents of L, VCVS and non-eliminable V sources) stays a regular MNA unknown.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/cedarhip.h"

namespace chip {

// engine-internal device kinds (VCVS is split into its branch part and its control part)
enum { K_R = 1, K_C, K_L, K_V, K_I, K_VCVS_A, K_VCVS_B, K_VCCS, K_MOS, K_VA };
constexpr int NTERM = 8;  // terminals of an engine device record (compiled Verilog-A modules use up to 8 nodes)

struct HSource {
  int kind;
  double dc;
  double ac = 0.0;  // small-signal magnitude
  double par[CH_SRC_NPAR];
  std::vector<double> ts, ys;
};

struct HDev {
  int kind;
  int node[CH_DEV_NNODE];
  int ipar[CH_DEV_NIPAR];
  double par[CH_DEV_NPAR];
  double mult;
  int va_nt = 0;          // CH_DEV_VA: nodes of the compiled module
  unsigned va_qmask = 0;  // CH_DEV_VA: nodes touched by ddt() contributions
  int branch;      // MNA branch index or -1
  bool eliminated; // V source removed by known/alias analysis
};

struct KnownDef {  // value(t) = sum_j sign_j * src_j(t)
  std::vector<std::pair<int, double>> terms;
};

// one engine device record (what a lane evaluates)
struct EDev {
  int kind;
  int term[NTERM];   // >= 0: global unknown index; < 0: -(known index + 1)
  int nt = 4;        // terminals in use (K_VA: nodes of the module)
  unsigned qmask = 0; // K_VA: nodes that receive ddt() contributions
  int hdev;      // index into the description's device list (parameters, multiplier)
  int src;       // source index or -1
  int mos;       // MOS instance index or -1
};

struct CompClass {
  int nc = 0, ndev = 0;
  bool nonlinear = false;                  // contains a MOSFET: DC Newton steps are voltage-limited
  std::vector<int> mat_ptr, vec_ptr;       // CSR over nc*nc matrix targets / nc vector targets
  std::vector<uint16_t> mat_src, vec_src;  // staging offsets dev_local*stride + slot (G block / F block)
};

struct Analysis {
  int n_nodes = 0, n_branch = 0, n_mna = 0;
  std::vector<int> node_unknown;   // [n_nodes+1] global unknown index or -1
  std::vector<int> node_known;     // [n_nodes+1] known index or -1 (ground = known 0)
  std::vector<int> branch_unknown; // [n_branch]  global unknown index or -1 (eliminated)
  std::vector<KnownDef> known;     // known[0] = ground
  int n_alias = 0;
  int n_unk = 0;
  std::vector<int> unk_mna;        // [n_unk] a representative MNA index (node-1 or n_nodes+branch)
  std::vector<uint8_t> diff_mask;  // [n_unk] 1 = differential unknown (appears under d/dt)
  // components
  int n_comp = 0;
  std::vector<int> comp_uofs, comp_nc, comp_dofs, comp_ndev, comp_class;
  std::vector<EDev> edev;          // component-ordered
  std::vector<CompClass> classes;
  int max_nc = 0, max_ndev = 0;
  // tearing (bordered block-diagonal form): a few high-degree unknowns (supply rails behind a resistance) are taken out as the BORDER;
  // every block carries replicas of them as its last nb local unknowns (local index comp_no[c] + b), so that all per-block tables
  // keep their shape; devices that touch border unknowns only (the rail resistors) are listed apart
  int nb = 0;                      // border unknowns (0: not torn)
  int n_glob = 0;                  // unknowns of the untorn system (own unknowns of all blocks + nb)
  std::vector<int> comp_no;        // [n_comp] own unknowns of the block (comp_nc = comp_no + nb)
  std::vector<uint8_t> replica;    // [n_unk] 1 = border replica that is NOT the counted one (block 0 holds the counted replicas)
  struct BorderDev { int kind; int ta, tb; int hdev; };   // terminals: >= 0 border index, < 0 -(known index + 1)
  std::vector<BorderDev> border_dev;
  bool force_sparse = false;       // a block has more devices than 16-bit staging offsets address: it takes the sparse path (no dense gather lists)
  bool wide = false;               // a compiled Verilog-A device is present: stamp records are [I(8)|Q(8)|G(64)|C(64)]
  int stride() const { return wide ? 145 : 41; }   // odd number of doubles per record: the lanes of a wave write different LDS banks
  int g_ofs() const { return wide ? 16 : 8; }
  int g_ld() const { return wide ? 8 : 4; }
  std::vector<int> mos_hdev;       // MOS instance -> description device index
  std::string err;
};

// slot mask per engine kind: which of the NTERM vector slots / NTERM² matrix slots a device writes
inline void kind_mask(int kind, bool vec[NTERM], bool mat[NTERM * NTERM], int nt = 4) {
  for (int i = 0; i < NTERM; ++i) vec[i] = false;
  for (int i = 0; i < NTERM * NTERM; ++i) mat[i] = false;
  auto M = [&](int r, int c) { mat[r * NTERM + c] = true; };
  switch (kind) {
    case K_R: case K_C: vec[0] = vec[1] = true; M(0, 0); M(0, 1); M(1, 0); M(1, 1); break;
    case K_I: vec[0] = vec[1] = true; break;
    case K_V: case K_L: case K_VCVS_A: vec[0] = vec[1] = vec[2] = true; M(0, 2); M(1, 2); M(2, 0); M(2, 1); M(2, 2); break;
    case K_VCVS_B: vec[0] = true; M(0, 1); M(0, 2); break;
    case K_VCCS: vec[0] = vec[1] = true; M(0, 2); M(0, 3); M(1, 2); M(1, 3); break;
    case K_MOS: for (int i = 0; i < 4; ++i) { vec[i] = true; for (int j = 0; j < 4; ++j) M(i, j); } break;
    case K_VA: for (int i = 0; i < nt; ++i) { vec[i] = true; for (int j = 0; j < nt; ++j) M(i, j); } break;
  }
}

struct UnionFind {
  std::vector<int> p;
  explicit UnionFind(int n) : p(n) { std::iota(p.begin(), p.end(), 0); }
  int find(int x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }
  void unite(int a, int b) { a = find(a); b = find(b); if (a != b) p[std::max(a, b)] = std::min(a, b); }
};

// protect[d] = true: V source d must keep its branch unknown (its current is observed).
// swept_src[s] = true: source s has a runtime (per-sample) parameter → never treated as a constant 0 V alias.
inline int analyse(int n_nodes, std::vector<HDev>& dev, const std::vector<HSource>& src, const std::vector<char>& protect,
                   const std::vector<char>& swept_src, Analysis& A, bool tear = false) {
  A = Analysis();
  A.n_nodes = n_nodes;
  int nb = 0;
  for (auto& d : dev) { d.branch = (d.kind == CH_DEV_L || d.kind == CH_DEV_V || d.kind == CH_DEV_VCVS) ? nb++ : -1; d.eliminated = false; }
  A.n_branch = nb;
  A.n_mna = n_nodes + nb;

  // ---- 1+2: known nodes and aliases ----
  UnionFind uf(n_nodes + 1);
  std::vector<int> known_of(n_nodes + 1, -1);  // indexed by alias representative
  A.known.clear();
  A.known.push_back(KnownDef());  // ground
  known_of[0] = 0;
  auto is_zero_const = [&](const HSource& s, int si) {
    if (swept_src[si]) return false;
    if (s.dc != 0.0) return false;
    if (s.kind == CH_SRC_DC) return s.par[0] == 0.0;
    if (s.kind == CH_SRC_PWL) { for (double y : s.ys) if (y != 0.0) return false; return true; }
    return false;
  };
  bool changed = true;
  while (changed) {
    changed = false;
    for (size_t i = 0; i < dev.size(); ++i) {
      HDev& d = dev[i];
      if (d.kind != CH_DEV_V || d.eliminated || protect[i]) continue;
      int a = uf.find(d.node[0]), b = uf.find(d.node[1]);
      if (a == b) { A.err = "voltage source with both terminals on one node"; return CH_ERR_SINGULAR; }
      int ka = known_of[a], kb = known_of[b];
      const int si = d.ipar[0];
      if (ka >= 0 && kb >= 0) { A.err = "loop of voltage sources"; return CH_ERR_SINGULAR; }
      if (ka >= 0 || kb >= 0) {
        // V(a) - V(b) = src  →  unknown side becomes known
        KnownDef kd = A.known[ka >= 0 ? ka : kb];
        kd.terms.push_back({si, ka >= 0 ? -1.0 : +1.0});  // V(b) = V(a) - src ; V(a) = V(b) + src
        known_of[ka >= 0 ? b : a] = (int)A.known.size();
        A.known.push_back(kd);
        d.eliminated = true; changed = true;
      } else if (is_zero_const(src[si], si)) {
        // merge; the merged class keeps a known tag if either side had one (neither has, here)
        uf.unite(a, b);
        d.eliminated = true; changed = true; A.n_alias++;
      }
    }
  }
  // after merging, a class representative may have changed: re-resolve known tags
  {
    std::vector<int> k2(n_nodes + 1, -1);
    for (int n = 0; n <= n_nodes; ++n) if (known_of[n] >= 0) k2[uf.find(n)] = known_of[n];
    known_of.swap(k2);
  }

  // ---- unknown numbering (provisional): node representatives, then surviving branches ----
  std::vector<int> prov_node(n_nodes + 1, -1);
  int np = 0;
  for (int n = 1; n <= n_nodes; ++n) { int r = uf.find(n); if (known_of[r] < 0 && prov_node[r] < 0) prov_node[r] = np++; }
  std::vector<int> prov_branch(nb, -1);
  for (auto& d : dev) if (d.branch >= 0 && !d.eliminated) prov_branch[d.branch] = np++;
  const int nprov = np;

  // ---- engine device records with provisional terminals ----
  auto term_of_node = [&](int node) {
    int r = uf.find(node);
    if (known_of[r] >= 0) return -(known_of[r] + 1);
    return prov_node[r];
  };
  std::vector<EDev> recs;
  for (size_t i = 0; i < dev.size(); ++i) {
    const HDev& d = dev[i];
    if (d.eliminated) continue;
    EDev e; e.hdev = (int)i; e.src = -1; e.mos = -1;
    for (int k = 0; k < NTERM; ++k) e.term[k] = -1;  // known 0 = ground
    e.nt = 4; e.qmask = 0;
    switch (d.kind) {
      case CH_DEV_R: e.kind = K_R; e.term[0] = term_of_node(d.node[0]); e.term[1] = term_of_node(d.node[1]); recs.push_back(e); break;
      case CH_DEV_C: e.kind = K_C; e.term[0] = term_of_node(d.node[0]); e.term[1] = term_of_node(d.node[1]); recs.push_back(e); break;
      case CH_DEV_I: e.kind = K_I; e.src = d.ipar[0]; e.term[0] = term_of_node(d.node[0]); e.term[1] = term_of_node(d.node[1]); recs.push_back(e); break;
      case CH_DEV_V: e.kind = K_V; e.src = d.ipar[0]; e.term[0] = term_of_node(d.node[0]); e.term[1] = term_of_node(d.node[1]); e.term[2] = prov_branch[d.branch]; recs.push_back(e); break;
      case CH_DEV_L: e.kind = K_L; e.term[0] = term_of_node(d.node[0]); e.term[1] = term_of_node(d.node[1]); e.term[2] = prov_branch[d.branch]; recs.push_back(e); break;
      case CH_DEV_VCVS: {
        e.kind = K_VCVS_A; e.term[0] = term_of_node(d.node[0]); e.term[1] = term_of_node(d.node[1]); e.term[2] = prov_branch[d.branch]; recs.push_back(e);
        EDev f = e; f.kind = K_VCVS_B; f.term[0] = prov_branch[d.branch]; f.term[1] = term_of_node(d.node[2]); f.term[2] = term_of_node(d.node[3]); f.term[3] = -1; recs.push_back(f);
      } break;
      case CH_DEV_VCCS: e.kind = K_VCCS; for (int k = 0; k < 4; ++k) e.term[k] = term_of_node(d.node[k]); recs.push_back(e); break;
      case CH_DEV_MOS: e.kind = K_MOS; e.mos = (int)A.mos_hdev.size(); A.mos_hdev.push_back((int)i); for (int k = 0; k < 4; ++k) e.term[k] = term_of_node(d.node[k]); recs.push_back(e); break;
      case CH_DEV_VA: {
        e.kind = K_VA; e.nt = d.va_nt; e.qmask = d.va_qmask; A.wide = true;
        for (int k = 0; k < e.nt; ++k) e.term[k] = term_of_node(d.node[k]);
        recs.push_back(e);
      } break;
      default: A.err = "unknown device kind"; return CH_ERR_INVALID;
    }
  }

  // ---- 2b: tearing — up to two unknowns that (almost) every device group hangs on ----
  std::vector<int> border_of(nprov, -1);
  int nbord = 0;
  if (tear) {
    std::vector<int> deg(nprov, 0);
    for (const EDev& e : recs) for (int k = 0; k < NTERM; ++k) if (e.term[k] >= 0) deg[e.term[k]]++;
    std::vector<int> cand;
    for (int u = 0; u < nprov; ++u) if (deg[u] >= 64) cand.push_back(u);
    std::sort(cand.begin(), cand.end(), [&](int x, int y) { return deg[x] != deg[y] ? deg[x] > deg[y] : x < y; });
    if (cand.empty() || cand.size() > 2) { A.err = "tearing: the circuit has no border of one or two high-degree nodes"; return CH_ERR_UNSUPPORTED; }
    std::sort(cand.begin(), cand.end());
    for (int u : cand) border_of[u] = nbord++;
    for (int b = 0; b < nb; ++b) if (prov_branch[b] >= 0 && border_of[prov_branch[b]] >= 0) { A.err = "tearing: a branch current cannot be a border unknown"; return CH_ERR_UNSUPPORTED; }
  }
  A.nb = nbord;

  // ---- 3: connected components over the provisional unknowns (the border taken out) ----
  UnionFind cu(nprov);
  std::vector<char> touched(nprov, 0);
  for (const EDev& e : recs) {
    int first = -1;
    for (int k = 0; k < NTERM; ++k) if (e.term[k] >= 0) {
      touched[e.term[k]] = 1;
      if (border_of[e.term[k]] >= 0) continue;
      if (first < 0) first = e.term[k]; else cu.unite(first, e.term[k]);
    }
  }
  for (int u = 0; u < nprov; ++u) if (!touched[u]) { A.err = "floating node without any device"; return CH_ERR_SINGULAR; }
  std::map<int, int> comp_of_root;
  std::vector<int> comp_id(nprov, -1);
  for (int u = 0; u < nprov; ++u) {
    if (border_of[u] >= 0) continue;
    int r = cu.find(u);
    auto it = comp_of_root.find(r);
    if (it == comp_of_root.end()) it = comp_of_root.insert({r, (int)comp_of_root.size()}).first;
    comp_id[u] = it->second;
  }
  A.n_comp = (int)comp_of_root.size();
  if (tear && A.n_comp < 2) { A.err = "tearing: removing the border leaves one block"; return CH_ERR_UNSUPPORTED; }
  A.comp_no.assign(A.n_comp, 0);
  for (int u = 0; u < nprov; ++u) if (comp_id[u] >= 0) A.comp_no[comp_id[u]]++;
  A.comp_nc.assign(A.n_comp, 0);
  for (int c = 0; c < A.n_comp; ++c) A.comp_nc[c] = A.comp_no[c] + nbord;
  A.comp_uofs.assign(A.n_comp, 0);
  for (int c = 1; c < A.n_comp; ++c) A.comp_uofs[c] = A.comp_uofs[c - 1] + A.comp_nc[c - 1];
  // final numbering: component-major, provisional order inside a component, then the block's replicas of the border
  std::vector<int> fin(nprov, -1), fill(A.n_comp, 0);
  for (int u = 0; u < nprov; ++u) if (comp_id[u] >= 0) { int c = comp_id[u]; fin[u] = A.comp_uofs[c] + fill[c]++; }
  A.n_unk = A.n_comp > 0 ? A.comp_uofs[A.n_comp - 1] + A.comp_nc[A.n_comp - 1] : 0;
  A.n_glob = nprov;
  for (int u = 0; u < nprov; ++u) if (border_of[u] >= 0) fin[u] = A.comp_uofs[0] + A.comp_no[0] + border_of[u];   // the counted replica (block 0)
  A.replica.assign(A.n_unk, 0);
  for (int c = 1; c < A.n_comp; ++c) for (int b = 0; b < nbord; ++b) A.replica[A.comp_uofs[c] + A.comp_no[c] + b] = 1;

  A.node_unknown.assign(n_nodes + 1, -1);
  A.node_known.assign(n_nodes + 1, -1);
  A.unk_mna.assign(A.n_unk, -1);
  for (int n = 0; n <= n_nodes; ++n) {
    int r = uf.find(n);
    if (known_of[r] >= 0) A.node_known[n] = known_of[r];
    else { A.node_unknown[n] = fin[prov_node[r]]; if (A.unk_mna[fin[prov_node[r]]] < 0) A.unk_mna[fin[prov_node[r]]] = n - 1; }
  }
  A.branch_unknown.assign(nb, -1);
  for (auto& d : dev) if (d.branch >= 0 && !d.eliminated) { A.branch_unknown[d.branch] = fin[prov_branch[d.branch]]; A.unk_mna[fin[prov_branch[d.branch]]] = n_nodes + d.branch; }
  for (int c = 1; c < A.n_comp; ++c) for (int b = 0; b < nbord; ++b) A.unk_mna[A.comp_uofs[c] + A.comp_no[c] + b] = A.unk_mna[A.comp_uofs[0] + A.comp_no[0] + b];

  // ---- devices per component (component-major order) ----
  std::vector<std::vector<int>> cdev(A.n_comp);
  for (size_t i = 0; i < recs.size(); ++i) {
    EDev& e = recs[i];
    int c = -1;
    bool any_unknown = false;
    for (int k = 0; k < NTERM; ++k) if (e.term[k] >= 0) { any_unknown = true; if (border_of[e.term[k]] < 0) c = comp_id[e.term[k]]; }
    if (c < 0 && any_unknown) {
      // every unknown terminal is a border unknown: a two-terminal linear element of the border itself (the rail resistance,
      // a decoupling capacitor to ground or between the rails); its stamps are added to the reduced system by every wavefront
      const bool cap_ok = e.kind == K_C && (e.term[0] >= 0 || e.term[0] == -1) && (e.term[1] >= 0 || e.term[1] == -1);
      if (!(e.kind == K_R || cap_ok)) { A.err = "tearing: a device other than a resistor or a grounded / rail-to-rail capacitor sits on the border alone"; return CH_ERR_UNSUPPORTED; }
      Analysis::BorderDev bd; bd.kind = e.kind; bd.hdev = e.hdev;
      bd.ta = e.term[0] >= 0 ? border_of[e.term[0]] : e.term[0];
      bd.tb = e.term[1] >= 0 ? border_of[e.term[1]] : e.term[1];
      A.border_dev.push_back(bd);
      continue;
    }
    for (int k = 0; k < NTERM; ++k) if (e.term[k] >= 0) e.term[k] = border_of[e.term[k]] >= 0 ? A.comp_uofs[c] + A.comp_no[c] + border_of[e.term[k]] : fin[e.term[k]];
    if (c >= 0) cdev[c].push_back((int)i);  // devices between known nodes only do not enter the system
  }
  A.comp_dofs.assign(A.n_comp, 0);
  A.comp_ndev.assign(A.n_comp, 0);
  for (int c = 0; c < A.n_comp; ++c) {
    A.comp_dofs[c] = (int)A.edev.size();
    A.comp_ndev[c] = (int)cdev[c].size();
    for (int i : cdev[c]) A.edev.push_back(recs[i]);
    A.max_nc = std::max(A.max_nc, A.comp_nc[c]);
    A.max_ndev = std::max(A.max_ndev, A.comp_ndev[c]);
  }
  if (tear && A.max_nc > 16) { A.err = "tearing: a block with its border replicas has more than 16 unknowns"; return CH_ERR_UNSUPPORTED; }
  // differential mask
  A.diff_mask.assign(A.n_unk, 0);
  for (const EDev& e : A.edev) {
    auto mark = [&](int t) { if (t >= 0) A.diff_mask[t] = 1; };
    if (e.kind == K_C) { mark(e.term[0]); mark(e.term[1]); }
    else if (e.kind == K_L) mark(e.term[2]);
    else if (e.kind == K_MOS) for (int k = 0; k < 4; ++k) mark(e.term[k]);
    else if (e.kind == K_VA) for (int k = 0; k < e.nt; ++k) if (e.qmask & (1u << k)) mark(e.term[k]);
  }

  if (nbord > 0) {   // a border unknown is differential when any block (or a border capacitor) says so; every replica carries the flag
    std::vector<uint8_t> bd(nbord, 0);
    for (int c = 0; c < A.n_comp; ++c) for (int b = 0; b < nbord; ++b) bd[b] |= A.diff_mask[A.comp_uofs[c] + A.comp_no[c] + b];
    for (const auto& d : A.border_dev) if (d.kind == K_C) { if (d.ta >= 0) bd[d.ta] = 1; if (d.tb >= 0) bd[d.tb] = 1; }
    for (int c = 0; c < A.n_comp; ++c) for (int b = 0; b < nbord; ++b) A.diff_mask[A.comp_uofs[c] + A.comp_no[c] + b] = bd[b];
  }

  // ---- 4: classes + gather lists ----
  std::map<std::vector<int>, int> class_of_sig;
  A.comp_class.assign(A.n_comp, 0);
  for (int c = 0; c < A.n_comp; ++c) {
    std::vector<int> sig;
    sig.push_back(A.comp_nc[c]);
    sig.push_back(A.comp_ndev[c]);
    for (int i = 0; i < A.comp_ndev[c]; ++i) {
      const EDev& e = A.edev[A.comp_dofs[c] + i];
      sig.push_back(e.kind);
      for (int k = 0; k < NTERM; ++k) sig.push_back(e.term[k] >= 0 ? e.term[k] - A.comp_uofs[c] : -1);
      sig.push_back(e.nt);
    }
    auto it = class_of_sig.find(sig);
    if (it != class_of_sig.end()) { A.comp_class[c] = it->second; continue; }
    const int id = (int)A.classes.size();
    class_of_sig[sig] = id;
    A.comp_class[c] = id;
    CompClass cl;
    cl.nc = A.comp_nc[c]; cl.ndev = A.comp_ndev[c];
    const int stride = A.stride(), gofs = A.g_ofs(), gld = A.g_ld();
    if (cl.nc > 64 || (long)cl.ndev * stride > 65535) {   // such a circuit takes the sparse path (its own CSR assembly): no dense gather lists, no size limit here
      if (cl.nc <= 64) A.force_sparse = true;   // device-heavy small block (e.g. thousands of parallel instances on a few nodes): staging would not fit LDS either
      for (int i = 0; i < cl.ndev; ++i) { const int kd = A.edev[A.comp_dofs[c] + i].kind; if (kd == K_MOS || kd == K_VA) cl.nonlinear = true; }
      A.classes.push_back(std::move(cl));
      continue;
    }
    std::vector<std::vector<uint16_t>> ml((size_t)cl.nc * cl.nc), vl(cl.nc);
    for (int i = 0; i < cl.ndev; ++i) {
      const EDev& e = A.edev[A.comp_dofs[c] + i];
      bool vm[NTERM], mm[NTERM * NTERM];
      kind_mask(e.kind, vm, mm, e.nt);
      if (e.kind == K_MOS || e.kind == K_VA) cl.nonlinear = true;
      int row[NTERM];
      for (int k = 0; k < NTERM; ++k) row[k] = e.term[k] >= 0 ? e.term[k] - A.comp_uofs[c] : -1;
      for (int k = 0; k < NTERM; ++k) if (vm[k] && row[k] >= 0) vl[row[k]].push_back((uint16_t)(i * stride + k));
      for (int k = 0; k < NTERM; ++k) for (int j = 0; j < NTERM; ++j)
        if (mm[k * NTERM + j] && row[k] >= 0 && row[j] >= 0) ml[(size_t)row[k] * cl.nc + row[j]].push_back((uint16_t)(i * stride + gofs + k * gld + j));
    }
    cl.mat_ptr.push_back(0);
    for (auto& l : ml) { cl.mat_src.insert(cl.mat_src.end(), l.begin(), l.end()); cl.mat_ptr.push_back((int)cl.mat_src.size()); }
    cl.vec_ptr.push_back(0);
    for (auto& l : vl) { cl.vec_src.insert(cl.vec_src.end(), l.begin(), l.end()); cl.vec_ptr.push_back((int)cl.vec_src.size()); }
    A.classes.push_back(std::move(cl));
  }
  return CH_OK;
}

}  // namespace chip
