// Host-only fuzz of the structural analysis (cedarsim.jl_amd/csrc/ch_analysis.hpp): random device tables, compiled with
// -fsanitize=address,undefined and libstdc++ assertions by tests/test_host_analysis_fuzz.py.  No GPU, no oracle.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <numeric>
#include <random>
#include <set>
#include <string>
#include <vector>
#include "cedarhip.h"
#include "ch_analysis.hpp"
using namespace chip;
int main() {
  std::mt19937 rng(42);
  long ok = 0, err = 0;
  for (int trial = 0; trial < 3000; ++trial) {
    const int n_nodes = 1 + rng() % 12;
    const int n_src = 1 + rng() % 4;
    std::vector<HSource> src(n_src);
    for (auto& s : src) { s.kind = rng() % 2 ? CH_SRC_DC : CH_SRC_PWL; s.dc = (rng() % 3) ? 1.0 : 0.0; for (double& p : s.par) p = 0.0; s.par[0] = s.dc; if (s.kind == CH_SRC_PWL) { s.ts = {0.0, 1.0}; s.ys = {rng() % 2 ? 0.0 : 1.0, 0.0}; } }
    const int n_dev = 1 + rng() % 20;
    std::vector<HDev> dev;
    for (int i = 0; i < n_dev; ++i) {
      HDev d; std::memset(&d, 0, sizeof(d));
      const int kinds[] = {CH_DEV_R, CH_DEV_C, CH_DEV_L, CH_DEV_V, CH_DEV_I, CH_DEV_VCVS, CH_DEV_VCCS, CH_DEV_MOS, CH_DEV_VA};
      d.kind = kinds[rng() % 9];
      for (int k = 0; k < CH_DEV_NNODE; ++k) d.node[k] = 0;
      const int nt = d.kind == CH_DEV_VA ? 2 + rng() % 7 : 4;
      for (int k = 0; k < nt; ++k) d.node[k] = rng() % (n_nodes + 1);
      if (d.kind == CH_DEV_V || d.kind == CH_DEV_I) d.ipar[0] = rng() % n_src;
      d.par[0] = 1.0; d.mult = 1.0; d.va_nt = nt; d.va_qmask = rng() % 256;
      dev.push_back(d);
    }
    std::vector<char> protect(dev.size(), 0), swept(src.size(), 0);
    for (auto& p : protect) p = rng() % 5 == 0;
    for (auto& p : swept) p = rng() % 5 == 0;
    Analysis A;
    const int rc = analyse(n_nodes, dev, src, protect, swept, A);
    if (rc != CH_OK) { ++err; continue; }
    ++ok;
    // consistency of the result
    if ((int)A.node_unknown.size() != n_nodes + 1 || (int)A.unk_mna.size() != A.n_unk) { printf("size mismatch\n"); return 1; }
    int tot = 0; for (int c = 0; c < A.n_comp; ++c) tot += A.comp_nc[c];
    if (tot != A.n_unk) { printf("component sizes do not add up\n"); return 1; }
    for (const EDev& e : A.edev) for (int k = 0; k < NTERM; ++k) if (e.term[k] >= A.n_unk || e.term[k] < -(int)A.known.size()) { printf("terminal out of range\n"); return 1; }
    for (const CompClass& c : A.classes) {
      if (c.nc > 64) { if (!c.mat_ptr.empty() || !c.vec_ptr.empty()) { printf("sparse-path class with gather lists\n"); return 1; } continue; }
      if ((int)c.mat_ptr.size() != c.nc * c.nc + 1 || (int)c.vec_ptr.size() != c.nc + 1) { printf("list sizes\n"); return 1; }
      for (uint16_t o : c.mat_src) if (o >= c.ndev * A.stride()) { printf("mat offset out of range\n"); return 1; }
      for (uint16_t o : c.vec_src) if (o >= c.ndev * A.stride()) { printf("vec offset out of range\n"); return 1; }
    }
  }
  printf("analysed %ld circuits, %ld rejected with an error code\n", ok, err);
  // ---- tearing: tiles of MOSFETs / resistors / capacitors behind one or two shared rails with a series resistance ----
  long torn = 0, refused = 0;
  for (int trial = 0; trial < 300; ++trial) {
    const int tiles = 8 + rng() % 40, per = 2 + rng() % 5, rails = 1 + rng() % 2;
    // nodes: 1 = ideal supply, 2.. = rails, then `per` private nodes per tile
    const int rail0 = 2, priv0 = rail0 + rails, n_nodes = priv0 + tiles * per - 1;
    std::vector<HSource> src(1); src[0].kind = CH_SRC_DC; src[0].dc = 1.0; for (double& q : src[0].par) q = 0.0; src[0].par[0] = 1.0;
    std::vector<HDev> dev;
    auto add = [&](int kind, int a, int b, int c2 = 0, int d2 = 0) { HDev d; std::memset(&d, 0, sizeof(d)); d.kind = kind; d.node[0] = a; d.node[1] = b; d.node[2] = c2; d.node[3] = d2; d.par[0] = 1.0; d.mult = 1.0; d.va_nt = 4; dev.push_back(d); };
    add(CH_DEV_V, 1, 0);
    for (int r = 0; r < rails; ++r) { add(CH_DEV_R, r == 0 ? 1 : 0, rail0 + r); if (rng() % 2) add(CH_DEV_C, rail0 + r, 0); }
    if (rails == 2 && rng() % 2) add(CH_DEV_C, rail0, rail0 + 1);
    for (int t = 0; t < tiles; ++t) {
      const int base = priv0 + t * per;
      for (int k = 0; k < per; ++k) {   // every private node hangs on a rail through a MOSFET and on its neighbour through a resistor
        add(CH_DEV_MOS, base + k, base + (k + 1) % per, rail0 + (int)(rng() % rails), rail0 + (int)(rng() % rails));
        add(CH_DEV_MOS, rail0, base + k, base + (k + 1) % per, rail0 + rails - 1);
        add(CH_DEV_MOS, base + k, rail0 + rails - 1, rail0, rail0);
        add(CH_DEV_R, base + k, base + (k + 1) % per);
      }
      add(CH_DEV_C, base, 0);
    }
    std::vector<char> protect(dev.size(), 0), swept(src.size(), 0);
    Analysis A;
    const int rc = analyse(n_nodes, dev, src, protect, swept, A, true);
    if (rc != CH_OK) { ++refused; if (tiles * per * 3 >= 64 * 2) { printf("tearing refused a tiled array: %s\n", A.err.c_str()); return 1; } continue; }
    ++torn;
    if (A.nb != rails || A.n_comp != tiles || A.n_glob != tiles * per + rails) { printf("tearing: nb %d comps %d n_glob %d (want %d %d %d)\n", A.nb, A.n_comp, A.n_glob, rails, tiles, tiles * per + rails); return 1; }
    int tot = 0;
    for (int c = 0; c < A.n_comp; ++c) { tot += A.comp_nc[c]; if (A.comp_nc[c] != A.comp_no[c] + A.nb || A.comp_no[c] != per) { printf("tearing: block sizes\n"); return 1; } }
    if (tot != A.n_unk || (int)A.replica.size() != A.n_unk || (int)A.unk_mna.size() != A.n_unk) { printf("tearing: totals\n"); return 1; }
    if (A.classes.size() != 1) { /* random rail choices make tiles differ: allowed */ }
    for (int c = 0; c < A.n_comp; ++c) for (int b = 0; b < A.nb; ++b) {
      const int u = A.comp_uofs[c] + A.comp_no[c] + b;
      if (A.replica[u] != (c > 0) || A.unk_mna[u] != rail0 + b - 1) { printf("tearing: replica flags / MNA map\n"); return 1; }
    }
    for (int c = 0; c < A.n_comp; ++c) for (int i = 0; i < A.comp_ndev[c]; ++i) {
      const EDev& e = A.edev[A.comp_dofs[c] + i];
      for (int k = 0; k < NTERM; ++k) if (e.term[k] >= 0 && (e.term[k] < A.comp_uofs[c] || e.term[k] >= A.comp_uofs[c] + A.comp_nc[c])) { printf("tearing: a terminal leaves its block\n"); return 1; }
    }
    for (const auto& d : A.border_dev) if (d.ta >= A.nb || d.tb >= A.nb || (d.ta < 0 && d.tb < 0)) { printf("tearing: border device terminals\n"); return 1; }
    for (int r = 0; r < rails; ++r) if (A.node_unknown[rail0 + r] != A.comp_uofs[0] + A.comp_no[0] + r) { printf("tearing: rail node map\n"); return 1; }
  }
  printf("tearing: %ld arrays torn, %ld refused (too few devices on the rails)\n", torn, refused);
  return 0;
}
