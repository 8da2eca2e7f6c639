// ch_bsim4.hpp — BSIM4 (level 54, v4.5) MOSFET for the HIP engine.
//
// Replaces the VA-generated device functor the reference evaluates per instance per residual call
// (src/vasim.jl:853-867; `bsim4.va` from package BSIM4 0.5.0, used at test/gf180_dff.jl:11,
// benchmarks/benchmark_common.jl:57).  Two halves:
//   * b4_pack()   host: card ⊕ instance ⊕ temperature → packed parameter column (what the
//                 reference constant-folds per instance), one column per distinct
//                 (model, W, L, NF, AS, AD, PS, PD[, sample]) class;
//   * b4_device() device: one lane evaluates one instance.  The core is differentiated with a
//                 3-partial dual number in the MODE frame (Vgs', Vds', Vbs' after source/drain
//                 interchange) and chained to the 4x4 terminal stamp once at the end; junction
//                 diodes, junction charges and overlap charges are 1-D functions with analytic
//                 derivatives.  (The reference uses ForwardDiff duals on all of it,
//                 src/vasim.jl:347-357.)
// The model equations are the public BSIM4.5.0 definition for mobMod 0/1/2, capMod 2 (and 0 = no
// intrinsic charge), dioMod 1, rdsMod 0, no gate current / NQS / rgate / rbody.
#pragma once
#include <cmath>
#include <cstdint>

#include "../../include/cedarhip.h"

#if defined(__HIPCC__)
#define CH_HD __host__ __device__
#define CH_D __device__ __forceinline__
#else
#define CH_HD
#define CH_D inline
#endif

namespace chip {

// Packed per-class parameter column.  X(name)
#define CH_B4_PACKED(X)                                                                              \
  X(type) X(mobmod) X(capmod) X(nf) X(leff) X(weff) X(weffCV) X(weffCJ) X(vtm) X(tratio1) X(toxe)    \
  X(coxe) X(coxp) X(factor1) X(phi) X(sqrtPhi) X(Xdep0) X(vbi_phi) X(cdep0) X(litl) X(ldeb) X(k1)    \
  X(k1ox) X(k2ox) X(vbsc) X(vfb_phi) X(vth0t) X(k3) X(k3b) X(vthNarrowW) X(lpe0term) X(LpeVb)        \
  X(dvt0) X(dvt1L) X(dvt2) X(dvt0w) X(dvt1wWL) X(dvt2w) X(eta0) X(etab) X(kt1eff) X(kt2t)            \
  X(theta0vb0) X(thetaRout) X(vfbzb) X(vtfbphi1) X(vtfbphi2) X(nfactor) X(cdsc) X(cdscb) X(cdscd)    \
  X(cit) X(voffcbn) X(mstar) X(ngate) X(polyT1) X(dvtp0) X(dvtp1) X(ua) X(ub) X(uc) X(u0temp) X(eu)  \
  X(vsattemp) X(a0) X(agsa0) X(a1) X(a2) X(b0term) X(keta) X(xj) X(dwg) X(dwb) X(rds0h) X(rdswmin)   \
  X(prwg) X(prwb) X(delta) X(pclmlitl) X(pdiblb) X(fproutL) X(pdits) X(pditsd) X(pditslL) X(pscbe1l) \
  X(pscbe2) X(pvag) X(alphaL) X(beta0) X(agidlW) X(bgidl) X(cgidl) X(egidl) X(toxe3) X(xpart)        \
  X(cgso) X(cgdo) X(cgbo) X(cgslW) X(cgdlW) X(ckappas) X(ckappad) X(abulkCVfactor) X(acde) X(moinVtm) \
  X(noff) X(voffcv) X(CoxWL) X(toxp8) X(Isbs) X(Isbd) X(Nvtms) X(Nvtmd) X(vjsmFwd) X(vjdmFwd)        \
  X(IVjsmFwd) X(IVjdmFwd) X(czbs) X(czbssw) X(czbsswg) X(czbd) X(czbdsw) X(czbdswg) X(PhiBS)         \
  X(PhiBSWS) X(PhiBSWGS) X(PhiBD) X(PhiBSWD) X(PhiBSWGD) X(mjs) X(mjsws) X(mjswgs) X(mjd) X(mjswd)   \
  X(mjswgd)

enum B4Idx {
#define X(n) B4I_##n,
  CH_B4_PACKED(X)
#undef X
      B4I_COUNT
};

namespace k {
constexpr double EPS0 = 8.85418e-12, EPSSI = 1.03594e-10, KboQ = 8.617087e-5, Q = 1.60219e-19;
constexpr double MAX_EXP = 5.834617425e14, MIN_EXP = 1.713908431e-15, EXP_TH = 34.0;
constexpr double PI = 3.14159265358979323846;
}  // namespace k

// ------------------------------------------------------------------------------------------------
// Host: build one packed column.  mp = model card (CH_B4_NPAR, NaN = not given),
// ip = instance {w,l,nf,as,ad,ps,pd} (already multiplied by .option scale), temp in Celsius.
inline int b4_pack(const double* mp, const double* ip, double temp_c, double* out) {
  using namespace k;
  auto has = [&](int i) { return !std::isnan(mp[i]); };
  auto par = [&](int i, double d) { return std::isnan(mp[i]) ? d : mp[i]; };
  for (int i = 0; i < B4I_COUNT; ++i) out[i] = 0.0;
#define O(n) out[B4I_##n]
  const double type = par(CH_B4_type, 1.0);
  const int mobmod = (int)par(CH_B4_mobmod, 0.0), capmod = (int)par(CH_B4_capmod, 2.0);
  const int permod = (int)par(CH_B4_permod, 1.0), binunit = (int)par(CH_B4_binunit, 1.0);
  if ((int)par(CH_B4_rdsmod, 0.0) != 0 || (int)par(CH_B4_rgatemod, 0.0) != 0 || (int)par(CH_B4_rbodymod, 0.0) != 0 ||
      (int)par(CH_B4_igcmod, 0.0) != 0 || (int)par(CH_B4_igbmod, 0.0) != 0 || (int)par(CH_B4_trnqsmod, 0.0) != 0 ||
      (int)par(CH_B4_geomod, 0.0) != 0 || (int)par(CH_B4_diomod, 1.0) != 1 || mobmod < 0 || mobmod > 2 ||
      !(capmod == 0 || capmod == 2))
    return CH_ERR_UNSUPPORTED;
  const double W = ip[CH_MOS_W], L = ip[CH_MOS_L];
  const double nf = std::isnan(ip[CH_MOS_NF]) ? 1.0 : ip[CH_MOS_NF];
  if (!(W > 0) || !(L > 0) || !(nf >= 1)) return CH_ERR_INVALID;

  // temperatures
  const double Tn = par(CH_B4_tnom, 27.0) + 273.15, T = temp_c + 273.15;
  const double tr = T / Tn, dT = T - Tn, vtm0 = KboQ * Tn, vtm = KboQ * T;
  const double Eg0 = 1.16 - 7.02e-4 * Tn * Tn / (Tn + 1108.0), Eg = 1.16 - 7.02e-4 * T * T / (T + 1108.0);
  const double ni = 1.45e10 * (Tn / 300.15) * std::sqrt(Tn / 300.15) * std::exp(21.5565981 - Eg0 / (2.0 * vtm0));
  // oxide
  const double toxe = par(CH_B4_toxe, 3e-9), toxp = par(CH_B4_toxp, toxe), toxm = par(CH_B4_toxm, toxe);
  const double epsrox = par(CH_B4_epsrox, 3.9), coxe = epsrox * EPS0 / toxe, coxp = epsrox * EPS0 / toxp;
  const double factor1 = std::sqrt(EPSSI / (epsrox * EPS0) * toxe);
  // effective dimensions
  const double lint = par(CH_B4_lint, 0), wint = par(CH_B4_wint, 0);
  const double ll = par(CH_B4_ll, 0), lw = par(CH_B4_lw, 0), lwl = par(CH_B4_lwl, 0);
  const double wl = par(CH_B4_wl, 0), ww = par(CH_B4_ww, 0), wwl = par(CH_B4_wwl, 0);
  const double lln = par(CH_B4_lln, 1), lwn = par(CH_B4_lwn, 1), wln = par(CH_B4_wln, 1), wwn = par(CH_B4_wwn, 1);
  const double dlc0 = par(CH_B4_dlc, lint), dwc0 = par(CH_B4_dwc, wint), dwj0 = par(CH_B4_dwj, dwc0);
  const double Ln = L + par(CH_B4_xl, 0), Wn = W / nf + par(CH_B4_xw, 0);
  const double pl1 = std::pow(Ln, lln), pw1 = std::pow(Wn, lwn), pl2 = std::pow(Ln, wln), pw2 = std::pow(Wn, wwn);
  const double dl = lint + ll / pl1 + lw / pw1 + lwl / (pl1 * pw1);
  const double dlc = dlc0 + par(CH_B4_llc, ll) / pl1 + par(CH_B4_lwc, lw) / pw1 + par(CH_B4_lwlc, lwl) / (pl1 * pw1);
  const double dw = wint + wl / pl2 + ww / pw2 + wwl / (pl2 * pw2);
  const double dwx = par(CH_B4_wlc, wl) / pl2 + par(CH_B4_wwc, ww) / pw2 + par(CH_B4_wwlc, wwl) / (pl2 * pw2);
  const double leff = Ln - 2 * dl, weff = Wn - 2 * dw, leffCV = Ln - 2 * dlc, weffCV = Wn - 2 * (dwc0 + dwx),
               weffCJ = Wn - 2 * (dwj0 + dwx);
  if (leff <= 0 || weff <= 0 || leffCV <= 0 || weffCV <= 0 || weffCJ <= 0) return CH_ERR_INVALID;
  const double iL = (binunit == 1 ? 1e-6 : 1.0) / leff, iW = (binunit == 1 ? 1e-6 : 1.0) / weff, iLW = iL * iW;
  auto bin = [&](int i, double d) {
    auto z = [&](int j) { return std::isnan(mp[j]) ? 0.0 : mp[j]; };
    return par(i, d) + z(i + 1) * iL + z(i + 2) * iW + z(i + 3) * iLW;
  };
  // binned card values
  const double xj = bin(CH_B4_xj, 1.5e-7), ndep = bin(CH_B4_ndep, 1.7e17), nsd = bin(CH_B4_nsd, 1e20);
  const double nsub = bin(CH_B4_nsub, 6e16), ngate = bin(CH_B4_ngate, 0), phin = bin(CH_B4_phin, 0);
  double vbm = bin(CH_B4_vbm, -3.0);
  const double lpe0 = bin(CH_B4_lpe0, 1.74e-7), lpeb = bin(CH_B4_lpeb, 0), w0 = bin(CH_B4_w0, 2.5e-6);
  const double k3 = bin(CH_B4_k3, 80.0), k3b = bin(CH_B4_k3b, 0);
  const double dvt0 = bin(CH_B4_dvt0, 2.2), dvt1 = bin(CH_B4_dvt1, 0.53), dvt2 = bin(CH_B4_dvt2, -0.032);
  const double dvt0w = bin(CH_B4_dvt0w, 0), dvt1w = bin(CH_B4_dvt1w, 5.3e6), dvt2w = bin(CH_B4_dvt2w, -0.032);
  const double drout = bin(CH_B4_drout, 0.56), dsub = has(CH_B4_dsub) ? bin(CH_B4_dsub, 0) : drout;
  const double kt1 = bin(CH_B4_kt1, -0.11), kt1l = bin(CH_B4_kt1l, 0), kt2 = bin(CH_B4_kt2, 0.022);
  double u0 = bin(CH_B4_u0, type > 0 ? 0.067 : 0.025);
  if (u0 > 1.0) u0 *= 1e-4;
  const double tr1 = tr - 1.0;
  O(type) = type; O(mobmod) = mobmod; O(capmod) = capmod; O(nf) = nf;
  O(leff) = leff; O(weff) = weff; O(weffCV) = weffCV; O(weffCJ) = weffCJ;
  O(vtm) = vtm; O(tratio1) = tr1; O(toxe) = toxe; O(coxe) = coxe; O(coxp) = coxp; O(factor1) = factor1;
  const double phi = vtm0 * std::log(ndep / ni) + phin + 0.4, sqrtPhi = std::sqrt(phi);
  O(phi) = phi; O(sqrtPhi) = sqrtPhi;
  const double Xdep0 = std::sqrt(2.0 * EPSSI / (Q * ndep * 1e6)) * sqrtPhi;
  O(Xdep0) = Xdep0;
  const double vbi = vtm0 * std::log(nsd * ndep / (ni * ni));
  O(vbi_phi) = vbi - phi;
  O(cdep0) = std::sqrt(Q * EPSSI * ndep * 1e6 / 2.0 / phi);
  const double litl = std::sqrt(3.0 * xj * toxe);
  O(litl) = litl;
  O(ldeb) = std::sqrt(EPSSI * vtm0 / (Q * ndep * 1e6)) / 3.0;
  double k1, k2;
  if (has(CH_B4_k1) || has(CH_B4_k2)) { k1 = bin(CH_B4_k1, 0.53); k2 = bin(CH_B4_k2, -0.0186); }
  else {
    const double g1 = has(CH_B4_gamma1) ? bin(CH_B4_gamma1, 0) : 5.753e-12 * std::sqrt(ndep) / coxe;
    const double g2 = has(CH_B4_gamma2) ? bin(CH_B4_gamma2, 0) : 5.753e-12 * std::sqrt(nsub) / coxe;
    const double xt = bin(CH_B4_xt, 1.55e-7);
    double vbx = has(CH_B4_vbx) ? bin(CH_B4_vbx, 0) : phi - 7.7348e-4 * ndep * xt * xt;
    if (vbx > 0) vbx = -vbx;
    if (vbm > 0) vbm = -vbm;
    const double a = std::sqrt(phi - vbx) - sqrtPhi, b = std::sqrt(phi * (phi - vbm)) - phi;
    k2 = (g1 - g2) * a / (2.0 * b + vbm);
    k1 = g2 - 2.0 * k2 * std::sqrt(phi - vbm);
  }
  double vbsc = -30.0;
  if (k2 < 0) { const double h = 0.5 * k1 / k2; vbsc = 0.9 * (phi - h * h); vbsc = vbsc > -3.0 ? -3.0 : (vbsc < -30.0 ? -30.0 : vbsc); }
  if (vbsc > vbm) vbsc = vbm;
  const double k1ox = k1 * toxe / toxm, k2ox = k2 * toxe / toxm;
  O(k1) = k1; O(k1ox) = k1ox; O(k2ox) = k2ox; O(vbsc) = vbsc;
  const bool vth0Given = has(CH_B4_vth0);
  double vth0 = bin(CH_B4_vth0, type > 0 ? 0.7 : -0.7), vfb;
  if (has(CH_B4_vfb)) vfb = bin(CH_B4_vfb, -1.0);
  else if (vth0Given) vfb = type * vth0 - phi - k1 * sqrtPhi;
  else vfb = -1.0;
  if (!vth0Given) vth0 = type * (vfb + phi + k1ox * sqrtPhi);
  O(vfb_phi) = vfb + phi;
  O(vth0t) = type * vth0;
  O(k3) = k3; O(k3b) = k3b;
  const double vthNW = toxe * phi / (weff + w0);
  O(vthNarrowW) = vthNW;
  const double lpe0s = std::sqrt(1.0 + lpe0 / leff);
  O(lpe0term) = k1ox * (lpe0s - 1.0) * sqrtPhi;
  O(LpeVb) = std::sqrt(1.0 + lpeb / leff);
  O(dvt0) = dvt0; O(dvt1L) = dvt1 * leff; O(dvt2) = dvt2;
  O(dvt0w) = dvt0w; O(dvt1wWL) = dvt1w * weff * leff; O(dvt2w) = dvt2w;
  O(eta0) = bin(CH_B4_eta0, 0.08); O(etab) = bin(CH_B4_etab, -0.07);
  O(kt1eff) = (kt1 + kt1l / leff) * tr1; O(kt2t) = kt2 * tr1;
  auto sce = [&](double x) {
    if (x < EXP_TH) { const double e = std::exp(x), m1 = e - 1.0; return e / (m1 * m1 + 2.0 * e * MIN_EXP); }
    return 1.0 / (MAX_EXP - 2.0);
  };
  const double lt0 = std::sqrt(EPSSI / (epsrox * EPS0) * toxe * Xdep0);
  O(theta0vb0) = sce(dsub * leff / lt0);
  O(thetaRout) = bin(CH_B4_pdiblc1, 0.39) * sce(drout * leff / lt0) + bin(CH_B4_pdiblc2, 0.0086);
  {
    const double v0 = vbi - phi, f1x = factor1 * std::sqrt(Xdep0);
    const double t8 = dvt0w * sce(dvt1w * weff * leff / f1x) * v0, t9 = dvt0 * sce(dvt1 * leff / f1x) * v0;
    const double t5 = k1ox * (lpe0s - 1.0) * sqrtPhi + (kt1 + kt1l / leff) * tr1;
    O(vfbzb) = type * vth0 - t8 - t9 + k3 * vthNW + t5 - phi - k1 * sqrtPhi;
  }
  {
    const double t3 = type * vth0 - vfb - phi;
    double a = type > 0 ? 2.0 * t3 : 2.5 * t3, b = 4.0 * t3;
    O(vtfbphi1) = a < 0 ? 0 : a; O(vtfbphi2) = b < 0 ? 0 : b;
  }
  O(nfactor) = bin(CH_B4_nfactor, 1.0); O(cdsc) = bin(CH_B4_cdsc, 2.4e-4); O(cdscb) = bin(CH_B4_cdscb, 0);
  O(cdscd) = bin(CH_B4_cdscd, 0); O(cit) = bin(CH_B4_cit, 0);
  O(voffcbn) = (bin(CH_B4_voff, -0.08) + par(CH_B4_voffl, 0) / leff) * (1.0 + bin(CH_B4_tvoff, 0) * dT);
  O(mstar) = 0.5 + std::atan(bin(CH_B4_minv, 0)) / PI;
  O(ngate) = ngate;
  O(polyT1) = 1e6 * Q * EPSSI * ngate / (coxe * coxe);
  O(dvtp0) = bin(CH_B4_dvtp0, 0); O(dvtp1) = bin(CH_B4_dvtp1, 0);
  O(ua) = bin(CH_B4_ua, mobmod == 2 ? 1e-15 : 1e-9) + bin(CH_B4_ua1, 1e-9) * tr1;
  O(ub) = bin(CH_B4_ub, 1e-19) + bin(CH_B4_ub1, -1e-18) * tr1;
  O(uc) = bin(CH_B4_uc, mobmod == 1 ? -0.0465 : -0.0465e-9) + bin(CH_B4_uc1, mobmod == 1 ? -0.056 : -0.056e-9) * tr1;
  O(u0temp) = u0 * std::pow(tr, bin(CH_B4_ute, -1.5));
  O(eu) = bin(CH_B4_eu, type > 0 ? 1.67 : 1.0);
  O(vsattemp) = bin(CH_B4_vsat, 8e4) - bin(CH_B4_at, 3.3e4) * tr1;
  const double a0 = bin(CH_B4_a0, 1.0);
  O(a0) = a0; O(agsa0) = bin(CH_B4_ags, 0) * a0; O(a1) = bin(CH_B4_a1, 0); O(a2) = bin(CH_B4_a2, 1.0);
  O(b0term) = bin(CH_B4_b0, 0) / (weff + bin(CH_B4_b1, 0));
  O(keta) = bin(CH_B4_keta, -0.047); O(xj) = xj; O(dwg) = bin(CH_B4_dwg, 0); O(dwb) = bin(CH_B4_dwb, 0);
  {
    const double prt = bin(CH_B4_prt, 0), pww = std::pow(weffCJ * 1e6, bin(CH_B4_wr, 1.0)) * nf;
    double r0 = (bin(CH_B4_rdsw, 200.0) + prt * tr1) * nf / pww, rmin = (par(CH_B4_rdswmin, 0) + prt * tr1) * nf / pww;
    O(rds0h) = 0.5 * (r0 < 0 ? 0 : r0); O(rdswmin) = rmin < 0 ? 0 : rmin;
  }
  O(prwg) = bin(CH_B4_prwg, 1.0); O(prwb) = bin(CH_B4_prwb, 0); O(delta) = bin(CH_B4_delta, 0.01);
  O(pclmlitl) = bin(CH_B4_pclm, 1.3) * litl;  // pclm*litl; pclm <= MIN_EXP encoded as 0
  if (bin(CH_B4_pclm, 1.3) <= MIN_EXP) O(pclmlitl) = 0.0;
  O(pdiblb) = bin(CH_B4_pdiblcb, 0);
  { const double fp = bin(CH_B4_fprout, 0); O(fproutL) = fp <= 0 ? 0.0 : fp * std::sqrt(leff); }
  O(pdits) = bin(CH_B4_pdits, 0); O(pditsd) = bin(CH_B4_pditsd, 0); O(pditslL) = 1.0 + par(CH_B4_pditsl, 0) * leff;
  O(pscbe1l) = bin(CH_B4_pscbe1, 4.24e8) * litl; O(pscbe2) = bin(CH_B4_pscbe2, 1e-5); O(pvag) = bin(CH_B4_pvag, 0);
  { const double al = bin(CH_B4_alpha0, 0) + bin(CH_B4_alpha1, 0) * leff; O(alphaL) = al <= 0 ? 0.0 : al / leff; }
  O(beta0) = bin(CH_B4_beta0, 0);
  {
    const double ag = bin(CH_B4_agidl, 0), bg = bin(CH_B4_bgidl, 2.3e9), cg = bin(CH_B4_cgidl, 0.5);
    O(agidlW) = (ag > 0 && bg > 0 && cg > 0) ? ag * weffCJ * nf : 0.0;
    O(bgidl) = bg; O(cgidl) = cg; O(egidl) = bin(CH_B4_egidl, 0.8); O(toxe3) = 3.0 * toxe;
  }
  // charge model
  O(xpart) = par(CH_B4_xpart, 0);
  const double cgsl = bin(CH_B4_cgsl, 0), cgdl = bin(CH_B4_cgdl, 0);
  const double cf = has(CH_B4_cf) ? bin(CH_B4_cf, 0) : 2.0 * epsrox * EPS0 / PI * std::log(1.0 + 0.4e-6 / toxe);
  const bool dlcOK = has(CH_B4_dlc) && dlc0 > 0;
  const double cgdo = has(CH_B4_cgdo) ? mp[CH_B4_cgdo] : (dlcOK ? dlc0 * coxe - cgdl : 0.6 * xj * coxe);
  const double cgso = has(CH_B4_cgso) ? mp[CH_B4_cgso] : (dlcOK ? dlc0 * coxe - cgsl : 0.6 * xj * coxe);
  const double cgbo = has(CH_B4_cgbo) ? mp[CH_B4_cgbo] : 2.0 * dwc0 * coxe;
  O(cgdo) = (cgdo + cf) * weffCV * nf; O(cgso) = (cgso + cf) * weffCV * nf; O(cgbo) = cgbo * leffCV * nf;
  O(cgslW) = cgsl * weffCV * nf; O(cgdlW) = cgdl * weffCV * nf;
  O(ckappas) = bin(CH_B4_ckappas, 0.6);
  O(ckappad) = has(CH_B4_ckappad) ? bin(CH_B4_ckappad, 0.6) : out[B4I_ckappas];
  O(abulkCVfactor) = 1.0 + std::pow(bin(CH_B4_clc, 1e-7) / leffCV, bin(CH_B4_cle, 0.6));
  O(acde) = bin(CH_B4_acde, 1.0) * std::pow(ndep / 2e16, -0.25);
  O(moinVtm) = bin(CH_B4_moin, 15.0) * vtm;
  O(noff) = bin(CH_B4_noff, 1.0); O(voffcv) = bin(CH_B4_voffcv, 0);
  O(CoxWL) = coxe * weffCV * leffCV * nf; O(toxp8) = 1e8 * toxp;
  // junctions
  {
    double jss = par(CH_B4_jss, 1e-4), jsws = par(CH_B4_jsws, 0), jswgs = par(CH_B4_jswgs, 0);
    double jsd = par(CH_B4_jsd, jss), jswd = par(CH_B4_jswd, jsws), jswgd = par(CH_B4_jswgd, jswgs);
    const double njs = par(CH_B4_njs, 1.0), njd = par(CH_B4_njd, njs);
    const double xtis = par(CH_B4_xtis, 3.0), xtid = par(CH_B4_xtid, xtis);
    if (dT != 0.0) {
      const double e0 = Eg0 / vtm0 - Eg / vtm, lt = std::log(tr);
      const double fs = std::exp((e0 + xtis * lt) / njs), fd = std::exp((e0 + xtid * lt) / njd);
      jss *= fs; jsws *= fs; jswgs *= fs; jsd *= fd; jswd *= fd; jswgd *= fd;
    }
    auto ct = [&](double c, double tc) { const double f = 1.0 + tc * dT; return f > 0 ? c * f : 0.0; };
    auto pt = [&](double p, double tp) { const double r = p - tp * dT; return r < 0.01 ? 0.01 : r; };
    const double tcj = par(CH_B4_tcj, 0), tcjsw = par(CH_B4_tcjsw, 0), tcjswg = par(CH_B4_tcjswg, 0);
    const double tpb = par(CH_B4_tpb, 0), tpbsw = par(CH_B4_tpbsw, 0), tpbswg = par(CH_B4_tpbswg, 0);
    const double cjs = par(CH_B4_cjs, 5e-4), cjd = par(CH_B4_cjd, cjs);
    const double cjsws = par(CH_B4_cjsws, 5e-10), cjswd = par(CH_B4_cjswd, cjsws);
    const double cjswgs = par(CH_B4_cjswgs, cjsws), cjswgd = par(CH_B4_cjswgd, cjswgs);
    const double pbs = par(CH_B4_pbs, 1.0), pbd = par(CH_B4_pbd, pbs), pbsws = par(CH_B4_pbsws, 1.0);
    const double pbswd = par(CH_B4_pbswd, pbsws), pbswgs = par(CH_B4_pbswgs, pbsws), pbswgd = par(CH_B4_pbswgd, pbswgs);
    O(mjs) = par(CH_B4_mjs, 0.5); O(mjd) = par(CH_B4_mjd, out[B4I_mjs]);
    O(mjsws) = par(CH_B4_mjsws, 0.33); O(mjswd) = par(CH_B4_mjswd, out[B4I_mjsws]);
    O(mjswgs) = par(CH_B4_mjswgs, out[B4I_mjsws]); O(mjswgd) = par(CH_B4_mjswgd, out[B4I_mjswgs]);
    O(PhiBS) = pt(pbs, tpb); O(PhiBD) = pt(pbd, tpb); O(PhiBSWS) = pt(pbsws, tpbsw); O(PhiBSWD) = pt(pbswd, tpbsw);
    O(PhiBSWGS) = pt(pbswgs, tpbswg); O(PhiBSWGD) = pt(pbswgd, tpbswg);
    // geoMod 0 diffusion geometry (isolated end diffusions, shared internal ones)
    const double dmcg = par(CH_B4_dmcg, 0), dmci = par(CH_B4_dmci, dmcg);
    const double Piso = 2.0 * (dmcg + dmci) + weffCJ, Aiso = (dmcg + dmci) * weffCJ, Psha = 2.0 * dmcg, Asha = dmcg * weffCJ;
    double eS, iS, eD, iD;
    if (((long)nf) % 2) { eS = eD = 1.0; iS = iD = (nf - 1.0) / 2.0; } else { eD = 0.0; iD = nf / 2.0; eS = 2.0; iS = nf / 2.0 - 1.0; }
    auto g = [&](int kpar) { return !std::isnan(ip[kpar]); };
    const double As = g(CH_MOS_AS) ? ip[CH_MOS_AS] : eS * Aiso + iS * Asha, Ad = g(CH_MOS_AD) ? ip[CH_MOS_AD] : eD * Aiso + iD * Asha;
    double Ps = g(CH_MOS_PS) ? (permod == 0 ? ip[CH_MOS_PS] : ip[CH_MOS_PS] - weffCJ * nf) : eS * Piso + iS * Psha;
    double Pd = g(CH_MOS_PD) ? (permod == 0 ? ip[CH_MOS_PD] : ip[CH_MOS_PD] - weffCJ * nf) : eD * Piso + iD * Psha;
    if (Ps < 0) Ps = 0;
    if (Pd < 0) Pd = 0;
    const double Isbs = As * jss + Ps * jsws + weffCJ * nf * jswgs, Isbd = Ad * jsd + Pd * jswd + weffCJ * nf * jswgd;
    O(Isbs) = Isbs; O(Isbd) = Isbd; O(Nvtms) = vtm * njs; O(Nvtmd) = vtm * njd;
    const double ijs = par(CH_B4_ijthsfwd, 0.1), ijd = par(CH_B4_ijthdfwd, ijs);
    if (Isbs > 0) { O(vjsmFwd) = vtm * njs * std::log(ijs / Isbs + 1.0); O(IVjsmFwd) = Isbs * std::exp(out[B4I_vjsmFwd] / (vtm * njs)); }
    if (Isbd > 0) { O(vjdmFwd) = vtm * njd * std::log(ijd / Isbd + 1.0); O(IVjdmFwd) = Isbd * std::exp(out[B4I_vjdmFwd] / (vtm * njd)); }
    O(czbs) = ct(cjs, tcj) * As; O(czbssw) = ct(cjsws, tcjsw) * Ps; O(czbsswg) = ct(cjswgs, tcjswg) * weffCJ * nf;
    O(czbd) = ct(cjd, tcj) * Ad; O(czbdsw) = ct(cjswd, tcjsw) * Pd; O(czbdswg) = ct(cjswgd, tcjswg) * weffCJ * nf;
  }
#undef O
  return CH_OK;
}

#if defined(__HIPCC__)
// ------------------------------------------------------------------------------------------------
// Device side.
//
// fp64 division and square root dominate the instruction count of a compact model; the IEEE
// sequences hipcc emits cost ~11 / ~15 VALU instructions.  frcp/fsqrt use the hardware seeds
// (v_rcp_f64, v_rsq_f64) plus Newton steps: <= 1-2 ulp, which is far inside the 1e-10 parity bar.
// v_rcp_f64 is good to 4.6e-8 relative on gfx950 (scripts/rcp_accuracy.hip); one cubically convergent correction
// r·(1 + e + e²), e = 1 − x·r, brings it to the rounding limit with three FMAs
CH_D double frcp(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  const double e = fma(-x, r, 1.0);
  return fma(r, fma(e, e, e), r);
}
CH_D double fsqrt(double x) {  // x > 0
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  g = fma(fma(-g, g, x), h, g);
  return g;
}

// value + N partial derivatives.  N = 3: (d/dVgs', d/dVds', d/dVbs') in one lane.
// N = 1: one partial per lane, four lanes per instance (the fourth carries a zero seed).
template <int N>
struct DN {
  double v, p[N];
};
template <int N> CH_D DN<N> mkd(double v) { DN<N> r; r.v = v; for (int i = 0; i < N; ++i) r.p[i] = 0.0; return r; }
template <int N> CH_D DN<N> operator+(DN<N> a, DN<N> b) { DN<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.p[i] = a.p[i] + b.p[i]; return r; }
template <int N> CH_D DN<N> operator-(DN<N> a, DN<N> b) { DN<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.p[i] = a.p[i] - b.p[i]; return r; }
template <int N> CH_D DN<N> operator-(DN<N> a) { DN<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.p[i] = -a.p[i]; return r; }
template <int N> CH_D DN<N> operator+(DN<N> a, double b) { a.v += b; return a; }
template <int N> CH_D DN<N> operator+(double b, DN<N> a) { a.v += b; return a; }
template <int N> CH_D DN<N> operator-(DN<N> a, double b) { a.v -= b; return a; }
template <int N> CH_D DN<N> operator-(double b, DN<N> a) { DN<N> r; r.v = b - a.v; for (int i = 0; i < N; ++i) r.p[i] = -a.p[i]; return r; }
template <int N> CH_D DN<N> operator*(DN<N> a, double b) { DN<N> r; r.v = a.v * b; for (int i = 0; i < N; ++i) r.p[i] = a.p[i] * b; return r; }
template <int N> CH_D DN<N> operator*(double b, DN<N> a) { return a * b; }
template <int N> CH_D DN<N> operator*(DN<N> a, DN<N> b) { DN<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.p[i] = fma(a.p[i], b.v, a.v * b.p[i]); return r; }
template <int N> CH_D DN<N> recip(DN<N> a) { const double r = frcp(a.v), m = -r * r; DN<N> o; o.v = r; for (int i = 0; i < N; ++i) o.p[i] = a.p[i] * m; return o; }
template <int N> CH_D DN<N> operator/(DN<N> a, DN<N> b) {
  const double r = frcp(b.v), q = a.v * r;
  DN<N> o; o.v = q; for (int i = 0; i < N; ++i) o.p[i] = fma(-q, b.p[i], a.p[i]) * r; return o;
}
template <int N> CH_D DN<N> operator/(DN<N> a, double b) { return a * frcp(b); }
template <int N> CH_D DN<N> operator/(double a, DN<N> b) { return recip(b) * a; }
template <int N> CH_D DN<N> chain(DN<N> a, double f, double fp) { DN<N> r; r.v = f; for (int i = 0; i < N; ++i) r.p[i] = a.p[i] * fp; return r; }
template <int N> CH_D DN<N> dsqrt(DN<N> a) { const double s = fsqrt(a.v); return chain(a, s, 0.5 * frcp(s)); }
template <int N> CH_D DN<N> dexp(DN<N> a) { const double e = exp(a.v); return chain(a, e, e); }
// Natural logarithm for positive finite normal arguments (every use below: junction terms, smoothing functions and
// mobility powers).  x = m·2^e with m in [sqrt(1/2), sqrt(2)); s = (m−1)/(m+1); ln m = 2s(1 + s²/3 + … + s¹⁸/19),
// |s| ≤ 0.1716 so the truncation error is 2e-17.  ≈ 28 instructions against 118 for the library log on gfx950
// (76 of them fp64); 0, negative, NaN and +inf arguments are handled by a final selection.
CH_D double flog(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double m = __builtin_amdgcn_frexp_mant(x);          // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool lo = m < 0.70710678118654752440;
  m = lo ? 2.0 * m : m;
  e = lo ? e - 1 : e;
  const double s = (m - 1.0) * frcp(m + 1.0);
  const double z = s * s;
  double p = 1.0 / 19.0;
  p = fma(p, z, 1.0 / 17.0); p = fma(p, z, 1.0 / 15.0); p = fma(p, z, 1.0 / 13.0); p = fma(p, z, 1.0 / 11.0);
  p = fma(p, z, 1.0 / 9.0); p = fma(p, z, 1.0 / 7.0); p = fma(p, z, 1.0 / 5.0); p = fma(p, z, 1.0 / 3.0);
  const double lm = fma(2.0 * s, p * z, 2.0 * s);
  const double de = (double)e;
  const double r = fma(de, 6.93147180369123816490e-01, fma(de, 1.90821492927058770002e-10, lm));
  // special cases by selection (no branch): 0 -> -inf, negative / NaN -> NaN, +inf -> +inf
  return (x > 0.0 && x < __builtin_inf()) ? r : (x == 0.0 ? -__builtin_inf() : (x > 0.0 ? x : __builtin_nan("")));
#else
  return log(x);
#endif
}
template <int N> CH_D DN<N> dlog(DN<N> a) { return chain(a, flog(a.v), frcp(a.v)); }

// smooth SCE factor e^x/((e^x-1)^2 + 2 e^x MIN_EXP)
template <int N> CH_D DN<N> d_sce(DN<N> x) {
  if (x.v < k::EXP_TH) {
    const DN<N> e = dexp(x), m1 = e - 1.0;
    return e / (m1 * m1 + e * (2.0 * k::MIN_EXP));
  }
  return mkd<N>(1.0 / (k::MAX_EXP - 2.0));
}

struct B4Col {
  // One packed column per class, parameters contiguous ([column][B4I_COUNT]): the ~140 values one
  // lane needs sit in 9 consecutive cache lines, and the 8 classes of a DFF block fit in L1.
  const double* p;  // base of this lane's column
  CH_D double operator[](int i) const { return p[i]; }
};
CH_D B4Col b4_col(const double* table, long col) { return B4Col{table + col * (long)B4I_COUNT}; }

// junction diode current and conductance (dioMod 1) incl. gmin
CH_D void diode(double vb, double Is, double Nvtm, double vjm, double IVjm, double gmin, double& i, double& g) {
  if (Is <= 0.0) { i = gmin * vb; g = gmin; return; }
  const double iN = frcp(Nvtm);
  if (vb <= vjm) {
    const double t = vb * iN;
    if (t < -k::EXP_TH) { i = Is * (k::MIN_EXP - 1.0) + gmin * vb; g = gmin; }
    else { const double e = exp(t); i = Is * (e - 1.0) + gmin * vb; g = Is * e * iN + gmin; }
  } else {
    const double s = IVjm * iN;
    i = IVjm - Is + s * (vb - vjm) + gmin * vb; g = s + gmin;
  }
}
// junction depletion charge and capacitance
CH_D void junction(double vb, double cz, double czsw, double czswg, double pb, double pbsw, double pbswg, double mj,
                   double mjsw, double mjswg, double& q, double& c) {
  if (vb < 0.0) {
    q = 0.0; c = 0.0;
    if (cz > 0.0) { const double a = 1.0 - vb * frcp(pb), s = exp(-mj * flog(a)); q += pb * cz * (1.0 - a * s) * frcp(1.0 - mj); c += cz * s; }
    if (czsw > 0.0) { const double a = 1.0 - vb * frcp(pbsw), s = exp(-mjsw * flog(a)); q += pbsw * czsw * (1.0 - a * s) * frcp(1.0 - mjsw); c += czsw * s; }
    if (czswg > 0.0) { const double a = 1.0 - vb * frcp(pbswg), s = exp(-mjswg * flog(a)); q += pbswg * czswg * (1.0 - a * s) * frcp(1.0 - mjswg); c += czswg * s; }
  } else {
    const double t0 = cz + czsw + czswg, t1 = cz * mj * frcp(pb) + czsw * mjsw * frcp(pbsw) + czswg * mjswg * frcp(pbswg);
    q = vb * (t0 + 0.5 * t1 * vb); c = t0 + t1 * vb;
  }
}
// bias-dependent overlap charge (capMod 2) and capacitance for one side
CH_D void overlap(double vg, double cov, double clW, double ckappa, double& q, double& c) {
  const double t0 = vg + 0.02, t1 = fsqrt(t0 * t0 + 0.08), t2 = 0.5 * (t0 - t1), dt2 = 0.5 * (1.0 - t0 * frcp(t1));
  const double t4 = fsqrt(1.0 - 4.0 * t2 * frcp(ckappa));
  q = (cov + clW) * vg - clW * (t2 + 0.5 * ckappa * (t4 - 1.0));
  c = (cov + clW) - clW * (dt2 - dt2 * frcp(t4));
}

// Mode-frame core: currents into the mode drain / source / bulk and the intrinsic charges, as
// dual numbers w.r.t. whatever seeds Vgs, Vds, Vbs carry.
template <int N>
struct B4Core {
  DN<N> iD, iS, iB, qd, qg, qs, qb;
};

template <int N>
CH_D void b4_core(const B4Col P, const DN<N> Vgs, const DN<N> Vds, const DN<N> Vbs, B4Core<N>& o) {
  using namespace k;
  typedef DN<N> D;
  const double phi = P[B4I_phi], sqrtPhi = P[B4I_sqrtPhi], Vtm = P[B4I_vtm], Leff = P[B4I_leff];
  const double coxe = P[B4I_coxe], itoxe = frcp(P[B4I_toxe]), icoxe = frcp(coxe), iLeff = frcp(Leff);

  // ---- effective body bias ----
  D Vbseff;
  {
    const double vbsc = P[B4I_vbsc];
    const D t0 = Vbs - (vbsc + 0.001);
    const D t1 = dsqrt(t0 * t0 - 0.004 * vbsc);
    if (t0.v >= 0.0) Vbseff = 0.5 * (t0 + t1) + vbsc;
    else Vbseff = vbsc * (1.0 + (-0.002) / (t1 - t0));
    const double t9 = 0.95 * phi;
    const D u0 = t9 - Vbseff - 0.001;
    Vbseff = t9 - 0.5 * (u0 + dsqrt(u0 * u0 + 0.004 * t9));
  }
  const D Phis = phi - Vbseff;
  const D sqrtPhis = dsqrt(Phis);
  const D Xdep = sqrtPhis * (P[B4I_Xdep0] * frcp(sqrtPhi));

  // ---- threshold voltage ----
  const D sqXdep = dsqrt(Xdep);
  const double f1 = P[B4I_factor1], V0 = P[B4I_vbi_phi];
  auto ltf = [&](double dv2) -> D {
    const D a = Vbseff * dv2;
    const D b = (a.v >= -0.5) ? a + 1.0 : (1.0 + 3.0 * a) / (3.0 + 8.0 * a);
    return sqXdep * b * f1;
  };
  const D Theta0 = d_sce(P[B4I_dvt1L] / ltf(P[B4I_dvt2]));
  const D Delt_vth = Theta0 * (P[B4I_dvt0] * V0);
  const double dvt0w = P[B4I_dvt0w];
  D T2w = mkd<N>(0.0);
  if (dvt0w != 0.0) T2w = d_sce(P[B4I_dvt1wWL] / ltf(P[B4I_dvt2w])) * (dvt0w * V0);
  const D T1t = P[B4I_lpe0term] + P[B4I_kt1eff] + Vbseff * P[B4I_kt2t];
  D T3d = P[B4I_eta0] + P[B4I_etab] * Vbseff;
  if (T3d.v < 1.0e-4) T3d = (2.0e-4 - T3d) / (3.0 - 2.0e4 * T3d);
  const D DIBL_Sft = T3d * Vds * P[B4I_theta0vb0];
  const double vthNW = P[B4I_vthNarrowW], LpeVb = P[B4I_LpeVb], k1ox = P[B4I_k1ox], k2ox = P[B4I_k2ox];
  D Vth = P[B4I_vth0t] + (k1ox * sqrtPhis - P[B4I_k1] * sqrtPhi) * LpeVb - k2ox * Vbseff - Delt_vth - T2w +
          (P[B4I_k3] + P[B4I_k3b] * Vbseff) * vthNW + T1t - DIBL_Sft;

  // ---- subthreshold swing ----
  D n;
  {
    const D c1 = EPSSI / Xdep;
    const D t4 = (P[B4I_nfactor] * c1 + (P[B4I_cdsc] + P[B4I_cdscb] * Vbseff + P[B4I_cdscd] * Vds) * Theta0 + P[B4I_cit]) * icoxe;
    n = (t4.v >= -0.5) ? t4 + 1.0 : (1.0 + 3.0 * t4) / (3.0 + 8.0 * t4);
  }
  {
    const double dvtp0 = P[B4I_dvtp0];
    if (dvtp0 > 0.0) {
      const D t0 = -P[B4I_dvtp1] * Vds;
      const D t2 = (t0.v < -EXP_TH) ? mkd<N>(MIN_EXP) : dexp(t0);
      const D t3 = Leff + dvtp0 * (1.0 + t2);
      Vth = Vth - n * (Vtm * dlog(Leff / t3));
    }
  }
  // ---- poly depletion ----
  D Vgs_eff = Vgs;
  {
    const double ngate = P[B4I_ngate], vfbphi = P[B4I_vfb_phi];
    if (ngate > 1.0e18 && ngate < 1.0e25 && Vgs.v > vfbphi) {
      const double iT1p = frcp(P[B4I_polyT1]);
      const D t8 = Vgs - vfbphi;
      const D t4 = dsqrt(1.0 + t8 * (2.0 * iT1p));
      const D t2 = 2.0 * t8 / (t4 + 1.0);
      const D t3 = t2 * t2 * (0.5 * iT1p);
      const D t7 = 1.07 - t3;
      const D t6 = dsqrt(t7 * t7 + 0.224);
      Vgs_eff = Vgs - (1.12 - 0.5 * (t7 + t6));
    }
  }
  const D Vgst = Vgs_eff - Vth;

  // ---- Vgsteff ----
  D Vgsteff;
  const D inVt = recip(n * Vtm);
  {
    const double mstar = P[B4I_mstar], cc = coxe * frcp(P[B4I_cdep0]);
    const D t1 = mstar * Vgst;
    const D t2 = t1 * inVt;
    D t10;
    if (t2.v > EXP_TH) t10 = t1;
    else if (t2.v < -EXP_TH) t10 = n * (Vtm * log(1.0 + MIN_EXP));
    else t10 = (n * Vtm) * dlog(1.0 + dexp(t2));
    const D h2 = (P[B4I_voffcbn] - (1.0 - mstar) * Vgst) * inVt;
    D t9;
    if (h2.v < -EXP_TH) t9 = mstar + n * (cc * MIN_EXP);
    else if (h2.v > EXP_TH) t9 = mstar + n * (cc * MAX_EXP);
    else t9 = mstar + n * (cc * dexp(h2));
    Vgsteff = t10 / t9;
  }

  // ---- Weff, Rds ----
  const D dsp = sqrtPhis - sqrtPhi;
  D Weff = P[B4I_weff] - 2.0 * (P[B4I_dwg] * Vgsteff + P[B4I_dwb] * dsp);
  if (Weff.v < 2.0e-8) Weff = 2.0e-8 * (4.0e-8 - Weff) / (6.0e-8 - 2.0 * Weff);
  D Rds;
  {
    const D t2 = recip(1.0 + P[B4I_prwg] * Vgsteff) + P[B4I_prwb] * dsp;
    Rds = P[B4I_rdswmin] + (t2 + dsqrt(t2 * t2 + 0.01)) * P[B4I_rds0h];
  }
  // ---- Abulk ----
  D Abulk0, Abulk;
  {
    const D t1 = (0.5 * k1ox * LpeVb) / sqrtPhis + (k2ox - P[B4I_k3b] * vthNW);
    const D t5 = Leff / (Leff + 2.0 * dsqrt(P[B4I_xj] * Xdep));
    const D t2 = P[B4I_a0] * t5 + P[B4I_b0term];
    Abulk0 = 1.0 + t1 * t2;
    const D dAdVg = -t1 * (P[B4I_agsa0] * (t5 * t5 * t5));
    Abulk = Abulk0 + dAdVg * Vgsteff;
    if (Abulk0.v < 0.1) Abulk0 = (0.2 - Abulk0) / (3.0 - 20.0 * Abulk0);
    if (Abulk.v < 0.1) Abulk = (0.2 - Abulk) / (3.0 - 20.0 * Abulk);
    const D t2k = P[B4I_keta] * Vbseff;
    const D t0k = (t2k.v >= -0.9) ? recip(1.0 + t2k) : (17.0 + 20.0 * t2k) / (0.8 + t2k);
    Abulk = Abulk * t0k;
    Abulk0 = Abulk0 * t0k;
  }
  // ---- mobility ----
  D iueff;  // 1/ueff
  {
    const int mobmod = (int)P[B4I_mobmod];
    D t5;
    if (mobmod == 0) {
      const D t3 = (Vgsteff + Vth + Vth) * itoxe;
      t5 = t3 * (P[B4I_ua] + P[B4I_uc] * Vbseff + P[B4I_ub] * t3);
    } else if (mobmod == 1) {
      const D t3 = (Vgsteff + Vth + Vth) * itoxe;
      t5 = t3 * (P[B4I_ua] + P[B4I_ub] * t3) * (1.0 + P[B4I_uc] * Vbseff);
    } else {
      const D t0 = (Vgsteff + P[B4I_vtfbphi1]) * itoxe;
      t5 = dexp(P[B4I_eu] * dlog(t0)) * (P[B4I_ua] + P[B4I_uc] * Vbseff);
    }
    const D den = (t5.v >= -0.8) ? t5 + 1.0 : (0.6 + t5) / (7.0 + 10.0 * t5);
    iueff = den * frcp(P[B4I_u0temp]);
  }
  const D ueff = recip(iueff);
  // ---- Vdsat ----
  const double vsat = P[B4I_vsattemp];
  const D WVCoxRds = Weff * Rds * (vsat * coxe);
  const D Esat = (2.0 * vsat) * iueff;
  const D EsatL = Esat * Leff;
  const D iEsatL = recip(EsatL);
  D Lambda;
  {
    const double a1 = P[B4I_a1], a2 = P[B4I_a2];
    if (a1 == 0.0) Lambda = mkd<N>(a2);
    else if (a1 > 0.0) {
      const double t0 = 1.0 - a2;
      const D t1 = t0 - a1 * Vgsteff - 0.0001;
      Lambda = a2 + t0 - 0.5 * (t1 + dsqrt(t1 * t1 + 0.0004 * t0));
    } else {
      const D t1 = a2 + a1 * Vgsteff - 0.0001;
      Lambda = 0.5 * (t1 + dsqrt(t1 * t1 + 0.0004 * a2));
    }
  }
  const D iLam = recip(Lambda);
  const D Vgst2Vtm = Vgsteff + 2.0 * Vtm;
  const D iVgst2Vtm = recip(Vgst2Vtm);
  D Vdsat;
  if (Rds.v == 0.0 && Lambda.v == 1.0) Vdsat = EsatL * Vgst2Vtm / (Abulk * EsatL + Vgst2Vtm);
  else {
    const D t9 = Abulk * WVCoxRds;
    const D t0 = 2.0 * Abulk * (t9 - 1.0 + iLam);
    const D t1 = Vgst2Vtm * (2.0 * iLam - 1.0) + Abulk * EsatL + 3.0 * (Vgst2Vtm * t9);
    const D t2 = Vgst2Vtm * (EsatL + 2.0 * (Vgst2Vtm * WVCoxRds));
    Vdsat = (t1 - dsqrt(t1 * t1 - 2.0 * t0 * t2)) / t0;
  }
  // ---- Vdseff ----
  D Vdseff;
  {
    const double delta = P[B4I_delta];
    const D t1 = Vdsat - Vds - delta;
    const D t2 = dsqrt(t1 * t1 + 4.0 * delta * Vdsat);
    if (t1.v >= 0.0) Vdseff = Vdsat - 0.5 * (t1 + t2);
    else Vdseff = Vdsat * (1.0 - (2.0 * delta) / (t2 - t1));
    if (Vds.v == 0.0) Vdseff = Vds;
    if (Vdseff.v > Vds.v) Vdseff = Vds;
  }
  const D diffVds = Vds - Vdseff;
  // ---- Vasat ----
  const D Vasat = (EsatL + Vdsat + 2.0 * (WVCoxRds * Vgsteff) * (1.0 - 0.5 * Abulk * Vdsat * iVgst2Vtm)) / ((2.0 * iLam - 1.0) + WVCoxRds * Abulk);
  // ---- channel conductance ----
  D Idl;
  {
    const D t0 = (Vgsteff + P[B4I_vtfbphi2]) * (0.5 * frcp(P[B4I_toxp8]));
    const D Tcen = 1.9e-9 / (1.0 + dexp(0.7 * dlog(t0)));
    const double coxp = P[B4I_coxp];
    const D Coxeff = (EPSSI * coxp) / (EPSSI + coxp * Tcen);
    const D beta = ueff * Coxeff * Weff * iLeff;
    const D fg1 = Vgsteff * (1.0 - 0.5 * Vdseff * Abulk * iVgst2Vtm);
    const D gche = beta * fg1 / (1.0 + Vdseff * iEsatL);
    Idl = gche / (1.0 + gche * Rds);
  }
  // ---- output resistance ----
  const double fpL = P[B4I_fproutL];
  const D FP = (fpL <= 0.0) ? mkd<N>(1.0) : recip(1.0 + fpL * iVgst2Vtm);
  D Pvag;
  {
    const D t9 = (P[B4I_pvag] * iEsatL) * Vgsteff;
    Pvag = (t9.v > -0.9) ? t9 + 1.0 : (0.8 + t9) / (17.0 + 20.0 * t9);
  }
  D iCclm, VACLM;  // 1/Cclm
  {
    const double pl = P[B4I_pclmlitl];
    if (pl > 0.0 && diffVds.v > 1.0e-10) {
      const D Cclm = FP * Pvag * (1.0 + Rds * Idl) * (Leff + Vdsat * ueff * (0.5 * frcp(vsat))) * frcp(pl);
      iCclm = recip(Cclm);
      VACLM = Cclm * diffVds;
    } else { iCclm = mkd<N>(1.0 / MAX_EXP); VACLM = mkd<N>(MAX_EXP); }
  }
  D iVADIBL;
  {
    const double th = P[B4I_thetaRout];
    if (th > MIN_EXP) {
      const D t8 = Abulk * Vdsat;
      D va = (Vgst2Vtm - Vgst2Vtm * t8 / (Vgst2Vtm + t8)) * frcp(th);
      const D t7 = P[B4I_pdiblb] * Vbseff;
      va = va * ((t7.v >= -0.9) ? recip(1.0 + t7) : (17.0 + 20.0 * t7) / (0.8 + t7));
      iVADIBL = recip(va * Pvag);
    } else iVADIBL = mkd<N>(1.0 / MAX_EXP);
  }
  D iVADITS;
  {
    const double pdits = P[B4I_pdits];
    if (pdits > MIN_EXP) {
      const D t0 = P[B4I_pditsd] * Vds;
      const D t1 = (t0.v > EXP_TH) ? mkd<N>(MAX_EXP) : dexp(t0);
      iVADITS = recip((1.0 + P[B4I_pditslL] * t1) * frcp(pdits) * FP);
    } else iVADITS = mkd<N>(1.0 / MAX_EXP);
  }
  D iVASCBE;
  {
    const double ps2 = P[B4I_pscbe2], ps1l = P[B4I_pscbe1l];
    if (ps2 > 0.0) {
      if (diffVds.v > ps1l / EXP_TH) iVASCBE = (ps2 * iLeff) * dexp(-ps1l / diffVds);
      else iVASCBE = mkd<N>(ps2 * iLeff / MAX_EXP);
    } else iVASCBE = mkd<N>(1.0 / MAX_EXP);
  }
  D Idsa = Idl * (1.0 + diffVds * iVADIBL);
  Idsa = Idsa * (1.0 + diffVds * iVADITS);
  Idsa = Idsa * (1.0 + dlog((Vasat + VACLM) / Vasat) * iCclm);
  D Isub = mkd<N>(0.0);
  {
    const double aL = P[B4I_alphaL], beta0 = P[B4I_beta0];
    if (aL > 0.0 && beta0 > 0.0) {
      D t1;
      if (diffVds.v > beta0 / EXP_TH) t1 = aL * diffVds * dexp(-beta0 / diffVds);
      else t1 = (aL * MIN_EXP) * diffVds;
      Isub = t1 * (Idsa * Vdseff);
    }
  }
  const double nf = P[B4I_nf];
  const D Ids = Idsa * (1.0 + diffVds * iVASCBE) * Vdseff * nf;
  Isub = Isub * nf;
  // ---- GIDL / GISL ----
  D Igidl = mkd<N>(0.0), Igisl = mkd<N>(0.0);
  {
    const double agW = P[B4I_agidlW];
    if (agW > 0.0) {
      const double bg = P[B4I_bgidl], cg = P[B4I_cgidl], eg = P[B4I_egidl], it3x = frcp(P[B4I_toxe3]);
      const D Vbd = Vbs - Vds;
      const D t1 = (Vds - Vgs_eff - eg) * it3x;
      if (t1.v > 0.0 && Vbd.v <= 0.0) {
        const D t2 = bg / t1;
        const D ig = (t2.v < 100.0) ? agW * t1 * dexp(-t2) : (agW * 3.720075976e-44) * t1;
        const D t5 = -Vbd * Vbd * Vbd;
        Igidl = ig * (t5 / (cg + t5));
      }
      const D s1 = (-Vgs_eff - eg) * it3x;
      if (s1.v > 0.0 && Vbs.v <= 0.0) {
        const D t2 = bg / s1;
        const D ig = (t2.v < 100.0) ? agW * s1 * dexp(-t2) : (agW * 3.720075976e-44) * s1;
        const D t5 = -Vbs * Vbs * Vbs;
        Igisl = ig * (t5 / (cg + t5));
      }
    }
  }
  // ---- intrinsic charges (capMod 2) ----
  D qg = mkd<N>(0.0), qb = mkd<N>(0.0), qsm = mkd<N>(0.0), qdm = mkd<N>(0.0);
  const double xpart = P[B4I_xpart];
  if (xpart >= 0.0 && (int)P[B4I_capmod] != 0) {
    const D VbseffCV = (Vbseff.v < 0.0) ? Vbseff : phi - Phis;
    const double CoxWL = P[B4I_CoxWL], vfbzb = P[B4I_vfbzb], Cox = P[B4I_coxp], Tox = P[B4I_toxp8], ldeb = P[B4I_ldeb];
    const double iTox = frcp(Tox), cwl = CoxWL * icoxe;
    const D T0q = n * (Vtm * P[B4I_noff]);
    const D T1q = (Vgst - P[B4I_voffcv]) / T0q;
    D Vgc;
    if (T1q.v > EXP_TH) Vgc = Vgst - P[B4I_voffcv];
    else if (T1q.v < -EXP_TH) Vgc = T0q * log(1.0 + MIN_EXP);
    else Vgc = T0q * dlog(1.0 + dexp(T1q));
    const D V3 = vfbzb - Vgs_eff + VbseffCV - 0.02;
    const D Vfbeff = vfbzb - 0.5 * (V3 + dsqrt(V3 * V3 + ((vfbzb <= 0.0) ? -0.08 * vfbzb : 0.08 * vfbzb)));
    const D tm = (Vgs_eff - VbseffCV - vfbzb) * (P[B4I_acde] * iTox);
    D Tcen;
    if (tm.v > -EXP_TH && tm.v < EXP_TH) Tcen = ldeb * dexp(tm);
    else Tcen = mkd<N>(tm.v <= -EXP_TH ? ldeb * MIN_EXP : ldeb * MAX_EXP);
    const double LINK = 1.0e-11 * Tox;  // 1e-3 * toxp
    const D V3c = ldeb - Tcen - LINK;
    Tcen = ldeb - 0.5 * (V3c + dsqrt(V3c * V3c + 4.0 * LINK * ldeb));
    // Coxeff = Cox*Ccen/(Cox+Ccen) with Ccen = EPSSI/Tcen  =  Cox*EPSSI/(Cox*Tcen + EPSSI)
    D CoxWLcen = (Cox * EPSSI * cwl) / (Cox * Tcen + EPSSI);
    const D Qac0 = CoxWLcen * (Vfbeff - vfbzb);
    const double h0 = 0.5 * k1ox;
    const D T3s = Vgs_eff - Vfbeff - VbseffCV - Vgc;
    D T1s;
    if (k1ox == 0.0) T1s = mkd<N>(0.0);
    else if (T3s.v < 0.0) T1s = h0 + T3s * frcp(k1ox);
    else T1s = dsqrt(h0 * h0 + T3s);
    const D Qsub0 = CoxWLcen * (k1ox * (T1s - h0));
    double Den, T0d;
    if (k1ox <= 0.0) { Den = 0.25 * P[B4I_moinVtm]; T0d = 0.5 * sqrtPhi; }
    else { Den = P[B4I_moinVtm] * k1ox * k1ox; T0d = k1ox * sqrtPhi; }
    const D DeltaPhi = Vtm * dlog(1.0 + (2.0 * T0d + Vgc) * Vgc * frcp(Den));
    const D T3t = 4.0 * (Vth - vfbzb - phi);
    const D T0t = ((T3t.v >= 0.0) ? Vgc + T3t : Vgc + 1.0e-20) * (0.5 * iTox);
    Tcen = 1.9e-9 / (1.0 + dexp(0.7 * dlog(T0t)));
    CoxWLcen = (Cox * EPSSI * cwl) / (Cox * Tcen + EPSSI);
    const D AbulkCV = Abulk0 * P[B4I_abulkCVfactor];
    const D T1 = Vgc - DeltaPhi;
    const D VdsatCV = T1 / AbulkCV;
    const D T0v = VdsatCV - Vds - 0.02;
    const D T1v = dsqrt(T0v * T0v + 0.08 * VdsatCV);
    D VdseffCV;
    if (T0v.v >= 0.0) VdseffCV = VdsatCV - 0.5 * (T0v + T1v);
    else VdseffCV = VdsatCV * (1.0 - 0.04 / (T1v - T0v));
    if (Vds.v == 0.0) VdseffCV = Vds;
    const D T0 = AbulkCV * VdseffCV;
    const D T2 = 12.0 * (T1 - 0.5 * T0 + 1.0e-20);
    const D iT2 = recip(T2);
    const D T3 = T0 * iT2;
    qg = CoxWLcen * (T1 - T0 * (0.5 - T3));
    qb = CoxWLcen * (1.0 - AbulkCV) * (0.5 * VdseffCV - T3 * VdseffCV);
    if (xpart > 0.5) qsm = -CoxWLcen * (0.5 * T1 + 0.25 * T0 - 0.5 * T0 * T3);
    else if (xpart < 0.5) {
      const D T4b = T1 * ((2.0 / 3.0) * T0 * T0 + T1 * (T1 - (4.0 / 3.0) * T0)) - (2.0 / 15.0) * T0 * T0 * T0;
      qsm = -(72.0 * CoxWLcen * (iT2 * iT2)) * T4b;  // 0.5/(T2/12)^2 = 72/T2^2
    } else qsm = -0.5 * qg;
    qg = qg + Qac0 + Qsub0 - qb;
    qb = qb - (Qac0 + Qsub0);
    qdm = -(qg + qb + qsm);
  }
  // currents into the device: D': Ids + Isub + Igidl ; S': -Ids + Igisl ; B: -Isub - Igidl - Igisl
  o.iD = Ids + Isub + Igidl; o.iS = Igisl - Ids; o.iB = -(Isub + Igidl + Igisl);
  o.qd = qdm; o.qg = qg; o.qs = qsm; o.qb = qb;
}

// Junction diodes, junction charges and overlap charges on the TRUE terminals (type-normalised
// voltages): 5 two-terminal elements (a, b, value x, derivative dx w.r.t. v_a - v_b).
struct B4Two {
  double ibs, gbs, ibd, gbd, qbs, cbs, qbd, cbd, qgd, cgd, qgs, cgs, qgb, cgb;
};
CH_D void b4_two_terminal(const B4Col P, double vgs, double vds, double vbs, double gmin, B4Two& t) {
  const double vbd = vbs - vds;
  diode(vbs, P[B4I_Isbs], P[B4I_Nvtms], P[B4I_vjsmFwd], P[B4I_IVjsmFwd], gmin, t.ibs, t.gbs);
  diode(vbd, P[B4I_Isbd], P[B4I_Nvtmd], P[B4I_vjdmFwd], P[B4I_IVjdmFwd], gmin, t.ibd, t.gbd);
  junction(vbs, P[B4I_czbs], P[B4I_czbssw], P[B4I_czbsswg], P[B4I_PhiBS], P[B4I_PhiBSWS], P[B4I_PhiBSWGS], P[B4I_mjs], P[B4I_mjsws], P[B4I_mjswgs], t.qbs, t.cbs);
  junction(vbd, P[B4I_czbd], P[B4I_czbdsw], P[B4I_czbdswg], P[B4I_PhiBD], P[B4I_PhiBSWD], P[B4I_PhiBSWGD], P[B4I_mjd], P[B4I_mjswd], P[B4I_mjswgd], t.qbd, t.cbd);
  if ((int)P[B4I_capmod] != 0) {
    overlap(vgs - vds, P[B4I_cgdo], P[B4I_cgdlW], P[B4I_ckappad], t.qgd, t.cgd);
    overlap(vgs, P[B4I_cgso], P[B4I_cgslW], P[B4I_ckappas], t.qgs, t.cgs);
    t.cgb = P[B4I_cgbo]; t.qgb = t.cgb * (vgs - vbs);
  } else { t.qgd = t.cgd = t.qgs = t.cgs = t.qgb = t.cgb = 0.0; }
}

// One lane per instance.  out[40] = {I[4], Q[4], G[16], C[16]} for terminals (d,g,s,b);
// the multiplier is applied by the caller.
CH_D void b4_device(const B4Col P, double vd, double vg, double vs, double vb, double gmin, double* out) {
  const double tp = P[B4I_type];
  const double vds = tp * (vd - vs), vgs = tp * (vg - vs), vbs = tp * (vb - vs);
  const bool fwd = vds >= 0.0;
  typedef DN<3> D3;
  D3 Vgs = mkd<3>(fwd ? vgs : vgs - vds), Vds = mkd<3>(fwd ? vds : -vds), Vbs = mkd<3>(fwd ? vbs : vbs - vds);
  Vgs.p[0] = 1.0; Vds.p[1] = 1.0; Vbs.p[2] = 1.0;
  B4Core<3> c;
  b4_core<3>(P, Vgs, Vds, Vbs, c);
  double* I = out; double* Qo = out + 4; double* G = out + 8; double* C = out + 24;
  // mode-frame stamp, rows/cols (D', G, S', B); d/dS' = -(sum of the three partials)
  auto put = [&](int row, const D3& x, double* val, double* M) {
    val[row] = x.v; M[row * 4 + 0] = x.p[1]; M[row * 4 + 1] = x.p[0]; M[row * 4 + 3] = x.p[2]; M[row * 4 + 2] = -(x.p[0] + x.p[1] + x.p[2]);
  };
  put(0, c.iD, I, G); put(1, mkd<3>(0.0), I, G); put(2, c.iS, I, G); put(3, c.iB, I, G);
  put(0, c.qd, Qo, C); put(1, c.qg, Qo, C); put(2, c.qs, Qo, C); put(3, c.qb, Qo, C);
  // map mode frame -> true (d,g,s,b): exchange index 0 and 2 (rows and columns) when reversed.
  // Static indices under one runtime predicate keep everything in registers (no scratch).
  if (!fwd) {
    auto sw = [](double& x, double& y) { const double t = x; x = y; y = t; };
    sw(I[0], I[2]); sw(Qo[0], Qo[2]);
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) { sw(G[0 * 4 + cc], G[2 * 4 + cc]); sw(C[0 * 4 + cc], C[2 * 4 + cc]); }
#pragma unroll
    for (int r = 0; r < 4; ++r) { sw(G[r * 4 + 0], G[r * 4 + 2]); sw(C[r * 4 + 0], C[r * 4 + 2]); }
  }
  B4Two t;
  b4_two_terminal(P, vgs, vds, vbs, gmin, t);
  auto two = [&](double* val, double* M, int a, int b, double x, double dx) {
    val[a] += x; val[b] -= x;
    M[a * 4 + a] += dx; M[a * 4 + b] -= dx; M[b * 4 + a] -= dx; M[b * 4 + b] += dx;
  };
  two(I, G, 3, 2, t.ibs, t.gbs); two(I, G, 3, 0, t.ibd, t.gbd);
  two(Qo, C, 3, 2, t.qbs, t.cbs); two(Qo, C, 3, 0, t.qbd, t.cbd);
  two(Qo, C, 1, 0, t.qgd, t.cgd); two(Qo, C, 1, 2, t.qgs, t.cgs); two(Qo, C, 1, 3, t.qgb, t.cgb);
#pragma unroll
  for (int r = 0; r < 4; ++r) { I[r] *= tp; Qo[r] *= tp; }
}

// Four lanes per instance (a quad, lanes 4q..4q+3 of one wavefront).  Lane `sub` seeds ONE partial
// (0: Vgs', 1: Vds', 2: Vbs', 3: none), so the dual-number arithmetic per lane shrinks from
// 1+3 to 1+1 components; the S' column is the negated sum of the other three, formed with two
// quad shuffles.  Each lane then owns ONE column of the 4x4 G and C stamps and writes it straight
// into the 40-slot staging record `st` (scaled by the multiplier m); lane 3 also writes I and Q.
CH_D void b4_device_quad(const B4Col P, double vd, double vg, double vs, double vb, double gmin, int sub, double m, double* st) {
  const double tp = P[B4I_type];
  const double vds = tp * (vd - vs), vgs = tp * (vg - vs), vbs = tp * (vb - vs);
  const bool fwd = vds >= 0.0;
  typedef DN<1> D1;
  D1 Vgs = mkd<1>(fwd ? vgs : vgs - vds), Vds = mkd<1>(fwd ? vds : -vds), Vbs = mkd<1>(fwd ? vbs : vbs - vds);
  Vgs.p[0] = sub == 0 ? 1.0 : 0.0; Vds.p[0] = sub == 1 ? 1.0 : 0.0; Vbs.p[0] = sub == 2 ? 1.0 : 0.0;
  B4Core<1> c;
  b4_core<1>(P, Vgs, Vds, Vbs, c);
  // this lane's column in the mode frame (D'=0, G=1, S'=2, B=3): sub 0 -> G, 1 -> D', 2 -> B, 3 -> S'
  auto quad_sum = [](double x) { x += __shfl_xor(x, 1); x += __shfl_xor(x, 2); return x; };
  auto colval = [&](const D1& x) { const double sum = quad_sum(x.p[0]); return sub == 3 ? -sum : x.p[0]; };
  // mode-frame rows D', G, S', B
  double gI[4] = {colval(c.iD), 0.0, colval(c.iS), colval(c.iB)};
  double gQ[4] = {colval(c.qd), colval(c.qg), colval(c.qs), colval(c.qb)};
  double I4[4] = {c.iD.v, 0.0, c.iS.v, c.iB.v}, Q4[4] = {c.qd.v, c.qg.v, c.qs.v, c.qb.v};
  int col = sub == 0 ? 1 : (sub == 1 ? 0 : (sub == 2 ? 3 : 2));
  if (!fwd) {  // mode frame -> true frame: exchange D' and S' (rows, and this lane's column label)
    auto sw = [](double& x, double& y) { const double t = x; x = y; y = t; };
    sw(gI[0], gI[2]); sw(gQ[0], gQ[2]); sw(I4[0], I4[2]); sw(Q4[0], Q4[2]);
    col = col == 0 ? 2 : (col == 2 ? 0 : col);
  }
  B4Two t;
  b4_two_terminal(P, vgs, vds, vbs, gmin, t);
  // two-terminal element between rows (a,b): this lane's column gets +dx on row a / -dx on row b if
  // col == a, the opposite if col == b
  auto two = [&](double* val, double* gcol, int a, int b, double x, double dx) {
    val[a] += x; val[b] -= x;
    const double s = col == a ? dx : (col == b ? -dx : 0.0);
    gcol[a] += s; gcol[b] -= s;
  };
  two(I4, gI, 3, 2, t.ibs, t.gbs); two(I4, gI, 3, 0, t.ibd, t.gbd);
  two(Q4, gQ, 3, 2, t.qbs, t.cbs); two(Q4, gQ, 3, 0, t.qbd, t.cbd);
  two(Q4, gQ, 1, 0, t.qgd, t.cgd); two(Q4, gQ, 1, 2, t.qgs, t.cgs); two(Q4, gQ, 1, 3, t.qgb, t.cgb);
#pragma unroll
  for (int r = 0; r < 4; ++r) { st[8 + r * 4 + col] = m * gI[r]; st[24 + r * 4 + col] = m * gQ[r]; }
  if (sub == 3) {
    const double mt = m * tp;
#pragma unroll
    for (int r = 0; r < 4; ++r) { st[r] = mt * I4[r]; st[4 + r] = mt * Q4[r]; }
  }
}
#endif  // __HIPCC__

}  // namespace chip
