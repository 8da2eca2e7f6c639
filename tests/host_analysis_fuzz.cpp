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
  return 0;
}
