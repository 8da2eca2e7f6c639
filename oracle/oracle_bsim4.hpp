// oracle_bsim4.hpp — CPU restatement of the BSIM4 (level 54, v4.5) MOSFET used by the GF180 DFF
// benchmark (test/DFF/*.ngspice, test/gf180_dff.jl:11 `load_VA_model(BSIM4.bsim4_va)`).
//
// TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED for device arithmetic: the reference takes the model
// from package BSIM4 0.5.0 (Manifest.toml:223-227), whose bsim4.va is NOT under /root/reference,
// and no reference test pins its currents beyond logic levels (test/gf180_dff.jl:29-33,
// test/inverter.jl:40-50).  The equations below restate the public BSIM4.5.0 model definition
// for the configuration the in-tree GF180 card uses (test/binning/bins.cir:7-34: mobmod=0,
// capmod=2, diomod=1, rdsmod=0, rbodymod=0, rgatemod=0, igcmod=igbmod=0, geomod=0, permod=1),
// plus mobmod 1/2.  Derivatives come from dual numbers exactly as in the reference
// (ForwardDiff.Dual over the VA-generated code, src/vasim.jl:347-357).
//
// What is modelled: Vth (SCE, narrow width, DIBL, pocket lpe0/lpeb, temperature), subthreshold
// swing n, poly depletion, Vgsteff, bias-dependent Weff/Rds (rdsMod 0), Abulk, mobility,
// Vdsat/Vdseff, CLM/DIBL/DITS/SCBE output resistance, impact-ionisation Isub, GIDL/GISL,
// source/drain junction diodes (dioMod 1) with gmin, capMod-2 intrinsic charges with 40/60,
// 50/50 and 0/100 partition, bias-dependent overlap charges, junction charges.
// Not modelled (engine rejects or ignores): gate tunnelling, NQS, rgate/rbody networks, stress.
#pragma once
#include <cmath>
#include <cstdint>
#include "../include/cedarhip.h"
#include "oracle_dual.hpp"

namespace oracle {

struct B4Size {
  double type;
  int mobmod, capmod;
  double nf;
  // geometry
  double leff, weff, leffCV, weffCV, weffCJ;
  // temperature
  double vtm, vtm0, tratio, deltemp;
  // oxide
  double toxe, toxp, coxe, coxp, factor1;
  // threshold
  double phi, sqrtPhi, phis3, Xdep0, vbi, cdep0, litl, ldeb, k1, k2, k1ox, k2ox, vbsc, vfb, vth0;
  double k3, k3b, w0, lpe0, lpeb, dvt0, dvt1, dvt2, dvt0w, dvt1w, dvt2w, dsub, eta0, etab;
  double kt1, kt1l, kt2, theta0vb0, thetaRout, vfbzb, vtfbphi1, vtfbphi2;
  double nfactor, cdsc, cdscb, cdscd, cit, voffcbn, mstar, ngate, dvtp0, dvtp1;
  // mobility / saturation
  double ua, ub, uc, u0temp, eu, vsattemp, a0, ags, a1, a2, b0, b1, keta, xj, dwg, dwb;
  double rds0, rdswmin, prwg, prwb, delta, pclm, pdiblb, fprout, pdits, pditsd, pditsl;
  double pscbe1, pscbe2, pvag, alpha0, alpha1, beta0;
  double agidl, bgidl, cgidl, egidl;
  // charge
  double xpart, cgso, cgdo, cgbo, cgsl, cgdl, ckappas, ckappad, abulkCVfactor, acde, moin, noff, voffcv;
  // junctions
  double Isbs, Isbd, Nvtms, Nvtmd, vjsmFwd, vjdmFwd, IVjsmFwd, IVjdmFwd;
  double czbs, czbssw, czbsswg, czbd, czbdsw, czbdswg;
  double PhiBS, PhiBSWS, PhiBSWGS, PhiBD, PhiBSWD, PhiBSWGD;
  double mjs, mjsws, mjswgs, mjd, mjswd, mjswgd;
};

namespace b4c {
constexpr double EPS0 = 8.85418e-12;
constexpr double EPSSI = 1.03594e-10;
constexpr double KboQ = 8.617087e-5;
constexpr double Charge_q = 1.60219e-19;
constexpr double MAX_EXP = 5.834617425e14;
constexpr double MIN_EXP = 1.713908431e-15;
constexpr double EXP_THRESHOLD = 34.0;
constexpr double DELTA_1 = 0.02;
constexpr double DELTA_3 = 0.02;
constexpr double DELTA_4 = 0.02;
constexpr double PI = 3.14159265358979323846;
}  // namespace b4c

// Size- and temperature-dependent precompute (the work the reference constant-folds per instance,
// card ⊕ instance overrides → instance parameter set).
inline int b4_setup(const double* mp, const double* ip, double temp_c, B4Size& s) {
  using namespace b4c;
  auto given = [&](int idx) { return !std::isnan(mp[idx]); };
  auto get = [&](int idx, double dflt) { return std::isnan(mp[idx]) ? dflt : mp[idx]; };
  auto igiven = [&](int k) { return !std::isnan(ip[k]); };

  s.type = get(CH_B4_type, 1.0);
  s.mobmod = (int)get(CH_B4_mobmod, 0.0);
  s.capmod = (int)get(CH_B4_capmod, 2.0);
  int permod = (int)get(CH_B4_permod, 1.0);
  int binunit = (int)get(CH_B4_binunit, 1.0);
  double W = ip[CH_MOS_W], L = ip[CH_MOS_L];
  double nf = igiven(CH_MOS_NF) ? ip[CH_MOS_NF] : 1.0;
  s.nf = nf;
  if (!(W > 0.0) || !(L > 0.0)) return CH_ERR_INVALID;

  double tnom = get(CH_B4_tnom, 27.0) + 273.15;
  double Temp = temp_c + 273.15;
  s.tratio = Temp / tnom;
  s.deltemp = Temp - tnom;
  s.vtm0 = KboQ * tnom;
  s.vtm = KboQ * Temp;
  double Eg0 = 1.16 - 7.02e-4 * tnom * tnom / (tnom + 1108.0);
  double ni = 1.45e10 * (tnom / 300.15) * std::sqrt(tnom / 300.15) * std::exp(21.5565981 - Eg0 / (2.0 * s.vtm0));
  double Eg = 1.16 - 7.02e-4 * Temp * Temp / (Temp + 1108.0);

  s.toxe = get(CH_B4_toxe, 3.0e-9);
  s.toxp = get(CH_B4_toxp, s.toxe);
  double toxm = get(CH_B4_toxm, s.toxe);
  double epsrox = get(CH_B4_epsrox, 3.9);
  s.coxe = epsrox * EPS0 / s.toxe;
  s.coxp = epsrox * EPS0 / s.toxp;
  s.factor1 = std::sqrt(EPSSI / (epsrox * EPS0) * s.toxe);

  // ---- effective geometry ----
  double lint = get(CH_B4_lint, 0.0), wint = get(CH_B4_wint, 0.0);
  double ll = get(CH_B4_ll, 0.0), lw = get(CH_B4_lw, 0.0), lwl = get(CH_B4_lwl, 0.0);
  double wl = get(CH_B4_wl, 0.0), ww = get(CH_B4_ww, 0.0), wwl = get(CH_B4_wwl, 0.0);
  double lln = get(CH_B4_lln, 1.0), lwn = get(CH_B4_lwn, 1.0), wln = get(CH_B4_wln, 1.0), wwn = get(CH_B4_wwn, 1.0);
  double llc = get(CH_B4_llc, ll), lwc = get(CH_B4_lwc, lw), lwlc = get(CH_B4_lwlc, lwl);
  double wlc = get(CH_B4_wlc, wl), wwc = get(CH_B4_wwc, ww), wwlc = get(CH_B4_wwlc, wwl);
  double dlc0 = get(CH_B4_dlc, lint), dwc0 = get(CH_B4_dwc, wint), dwj0 = get(CH_B4_dwj, dwc0);
  double Lnew = L + get(CH_B4_xl, 0.0);
  double Wnew = W / nf + get(CH_B4_xw, 0.0);
  double T0 = std::pow(Lnew, lln), T1 = std::pow(Wnew, lwn);
  double dl = lint + ll / T0 + lw / T1 + lwl / (T0 * T1);
  double dlc = dlc0 + llc / T0 + lwc / T1 + lwlc / (T0 * T1);
  double T2 = std::pow(Lnew, wln), T3 = std::pow(Wnew, wwn);
  double dw = wint + wl / T2 + ww / T3 + wwl / (T2 * T3);
  double tmpc = wlc / T2 + wwc / T3 + wwlc / (T2 * T3);
  double dwc = dwc0 + tmpc, dwj = dwj0 + tmpc;
  s.leff = Lnew - 2.0 * dl;
  s.weff = Wnew - 2.0 * dw;
  s.leffCV = Lnew - 2.0 * dlc;
  s.weffCV = Wnew - 2.0 * dwc;
  s.weffCJ = Wnew - 2.0 * dwj;
  if (s.leff <= 0 || s.weff <= 0 || s.leffCV <= 0 || s.weffCV <= 0 || s.weffCJ <= 0) return CH_ERR_INVALID;
  double Inv_L, Inv_W, Inv_LW;
  if (binunit == 1) { Inv_L = 1.0e-6 / s.leff; Inv_W = 1.0e-6 / s.weff; Inv_LW = 1.0e-12 / (s.leff * s.weff); }
  else { Inv_L = 1.0 / s.leff; Inv_W = 1.0 / s.weff; Inv_LW = 1.0 / (s.leff * s.weff); }
  // binned value:  P = P0 + lP/Leff + wP/Weff + pP/(Leff*Weff)
  auto bin = [&](int idx, double dflt) {
    double base = std::isnan(mp[idx]) ? dflt : mp[idx];
    double lp = std::isnan(mp[idx + 1]) ? 0.0 : mp[idx + 1];
    double wp = std::isnan(mp[idx + 2]) ? 0.0 : mp[idx + 2];
    double pp = std::isnan(mp[idx + 3]) ? 0.0 : mp[idx + 3];
    return base + lp * Inv_L + wp * Inv_W + pp * Inv_LW;
  };

  // ---- binned parameters ----
  s.cdsc = bin(CH_B4_cdsc, 2.4e-4); s.cdscb = bin(CH_B4_cdscb, 0.0); s.cdscd = bin(CH_B4_cdscd, 0.0);
  s.cit = bin(CH_B4_cit, 0.0); s.nfactor = bin(CH_B4_nfactor, 1.0); s.xj = bin(CH_B4_xj, 1.5e-7);
  double vsat = bin(CH_B4_vsat, 8.0e4), at = bin(CH_B4_at, 3.3e4);
  s.a0 = bin(CH_B4_a0, 1.0); s.ags = bin(CH_B4_ags, 0.0); s.a1 = bin(CH_B4_a1, 0.0); s.a2 = bin(CH_B4_a2, 1.0);
  s.keta = bin(CH_B4_keta, -0.047);
  double nsub = bin(CH_B4_nsub, 6.0e16), ndep = bin(CH_B4_ndep, 1.7e17), nsd = bin(CH_B4_nsd, 1.0e20);
  double phin = bin(CH_B4_phin, 0.0);
  s.ngate = bin(CH_B4_ngate, 0.0);
  if (s.ngate > 0 && s.ngate <= 1.0e23) { /* cm^-3 as given */ }
  double vbm = bin(CH_B4_vbm, -3.0), xt = bin(CH_B4_xt, 1.55e-7);
  s.kt1 = bin(CH_B4_kt1, -0.11); s.kt1l = bin(CH_B4_kt1l, 0.0); s.kt2 = bin(CH_B4_kt2, 0.022);
  s.k3 = bin(CH_B4_k3, 80.0); s.k3b = bin(CH_B4_k3b, 0.0); s.w0 = bin(CH_B4_w0, 2.5e-6);
  s.dvtp0 = bin(CH_B4_dvtp0, 0.0); s.dvtp1 = bin(CH_B4_dvtp1, 0.0);
  s.lpe0 = bin(CH_B4_lpe0, 1.74e-7); s.lpeb = bin(CH_B4_lpeb, 0.0);
  s.dvt0 = bin(CH_B4_dvt0, 2.2); s.dvt1 = bin(CH_B4_dvt1, 0.53); s.dvt2 = bin(CH_B4_dvt2, -0.032);
  s.dvt0w = bin(CH_B4_dvt0w, 0.0); s.dvt1w = bin(CH_B4_dvt1w, 5.3e6); s.dvt2w = bin(CH_B4_dvt2w, -0.032);
  double drout = bin(CH_B4_drout, 0.56);
  s.dsub = bin(CH_B4_dsub, given(CH_B4_dsub) ? 0.0 : drout);
  if (!given(CH_B4_dsub)) s.dsub = drout;  // unbinned default: dsub = drout
  double ua = bin(CH_B4_ua, s.mobmod == 2 ? 1.0e-15 : 1.0e-9), ua1 = bin(CH_B4_ua1, 1.0e-9);
  double ub = bin(CH_B4_ub, 1.0e-19), ub1 = bin(CH_B4_ub1, -1.0e-18);
  double uc = bin(CH_B4_uc, s.mobmod == 1 ? -0.0465 : -0.0465e-9);
  double uc1 = bin(CH_B4_uc1, s.mobmod == 1 ? -0.056 : -0.056e-9);
  double u0 = bin(CH_B4_u0, s.type > 0 ? 0.067 : 0.025);
  if (u0 > 1.0) u0 /= 1.0e4;  // given in cm^2/Vs
  s.eu = bin(CH_B4_eu, s.type > 0 ? 1.67 : 1.0);
  double ute = bin(CH_B4_ute, -1.5);
  double voff = bin(CH_B4_voff, -0.08), tvoff = bin(CH_B4_tvoff, 0.0), minv = bin(CH_B4_minv, 0.0);
  s.delta = bin(CH_B4_delta, 0.01);
  double rdsw = bin(CH_B4_rdsw, 200.0), prt = bin(CH_B4_prt, 0.0);
  s.prwg = bin(CH_B4_prwg, 1.0); s.prwb = bin(CH_B4_prwb, 0.0);
  s.eta0 = bin(CH_B4_eta0, 0.08); s.etab = bin(CH_B4_etab, -0.07);
  s.pclm = bin(CH_B4_pclm, 1.3);
  double pdibl1 = bin(CH_B4_pdiblc1, 0.39), pdibl2 = bin(CH_B4_pdiblc2, 0.0086);
  s.pdiblb = bin(CH_B4_pdiblcb, 0.0);
  s.fprout = bin(CH_B4_fprout, 0.0); s.pdits = bin(CH_B4_pdits, 0.0); s.pditsd = bin(CH_B4_pditsd, 0.0);
  s.pditsl = get(CH_B4_pditsl, 0.0);
  s.pscbe1 = bin(CH_B4_pscbe1, 4.24e8); s.pscbe2 = bin(CH_B4_pscbe2, 1.0e-5); s.pvag = bin(CH_B4_pvag, 0.0);
  double wr = bin(CH_B4_wr, 1.0);
  s.dwg = bin(CH_B4_dwg, 0.0); s.dwb = bin(CH_B4_dwb, 0.0); s.b0 = bin(CH_B4_b0, 0.0); s.b1 = bin(CH_B4_b1, 0.0);
  s.alpha0 = bin(CH_B4_alpha0, 0.0); s.alpha1 = bin(CH_B4_alpha1, 0.0); s.beta0 = bin(CH_B4_beta0, 0.0);
  s.agidl = bin(CH_B4_agidl, 0.0); s.bgidl = bin(CH_B4_bgidl, 2.3e9); s.cgidl = bin(CH_B4_cgidl, 0.5);
  s.egidl = bin(CH_B4_egidl, 0.8);
  s.cgsl = bin(CH_B4_cgsl, 0.0); s.cgdl = bin(CH_B4_cgdl, 0.0);
  s.ckappas = bin(CH_B4_ckappas, 0.6);
  s.ckappad = given(CH_B4_ckappad) ? bin(CH_B4_ckappad, 0.6) : s.ckappas;
  double cf = given(CH_B4_cf) ? bin(CH_B4_cf, 0.0) : 2.0 * epsrox * EPS0 / PI * std::log(1.0 + 0.4e-6 / s.toxe);
  double clc = bin(CH_B4_clc, 1.0e-7), cle = bin(CH_B4_cle, 0.6);
  s.acde = bin(CH_B4_acde, 1.0); s.moin = bin(CH_B4_moin, 15.0); s.noff = bin(CH_B4_noff, 1.0);
  s.voffcv = bin(CH_B4_voffcv, 0.0);
  s.xpart = get(CH_B4_xpart, 0.0);

  // ---- temperature scaling ----
  double TR1 = s.tratio - 1.0;
  s.abulkCVfactor = 1.0 + std::pow(clc / s.leffCV, cle);
  double PowWeffWr = std::pow(s.weffCJ * 1.0e6, wr) * nf;
  s.ua = ua + ua1 * TR1; s.ub = ub + ub1 * TR1; s.uc = uc + uc1 * TR1;
  s.vsattemp = vsat - at * TR1;
  s.u0temp = u0 * std::pow(s.tratio, ute);
  s.rds0 = (rdsw + prt * TR1) * nf / PowWeffWr;
  s.rdswmin = (get(CH_B4_rdswmin, 0.0) + prt * TR1) * nf / PowWeffWr;
  if (s.rds0 < 0) s.rds0 = 0;
  if (s.rdswmin < 0) s.rdswmin = 0;

  // ---- overlap capacitance ----
  double cgdo_m, cgso_m, cgbo_m;
  if (given(CH_B4_cgdo)) cgdo_m = mp[CH_B4_cgdo];
  else if (given(CH_B4_dlc) && dlc0 > 0.0) cgdo_m = dlc0 * s.coxe - s.cgdl;
  else cgdo_m = 0.6 * s.xj * s.coxe;
  if (given(CH_B4_cgso)) cgso_m = mp[CH_B4_cgso];
  else if (given(CH_B4_dlc) && dlc0 > 0.0) cgso_m = dlc0 * s.coxe - s.cgsl;
  else cgso_m = 0.6 * s.xj * s.coxe;
  cgbo_m = given(CH_B4_cgbo) ? mp[CH_B4_cgbo] : 2.0 * dwc0 * s.coxe;
  s.cgdo = (cgdo_m + cf) * s.weffCV;
  s.cgso = (cgso_m + cf) * s.weffCV;
  s.cgbo = cgbo_m * s.leffCV * nf;

  // ---- threshold-voltage related ----
  s.phi = s.vtm0 * std::log(ndep / ni) + phin + 0.4;
  s.sqrtPhi = std::sqrt(s.phi);
  s.phis3 = s.sqrtPhi * s.phi;
  s.Xdep0 = std::sqrt(2.0 * EPSSI / (Charge_q * ndep * 1.0e6)) * s.sqrtPhi;
  s.litl = std::sqrt(3.0 * s.xj * s.toxe);
  s.vbi = s.vtm0 * std::log(nsd * ndep / (ni * ni));
  s.cdep0 = std::sqrt(Charge_q * EPSSI * ndep * 1.0e6 / 2.0 / s.phi);
  s.ldeb = std::sqrt(EPSSI * s.vtm0 / (Charge_q * ndep * 1.0e6)) / 3.0;
  s.acde *= std::pow(ndep / 2.0e16, -0.25);

  if (given(CH_B4_k1) || given(CH_B4_k2)) {
    s.k1 = bin(CH_B4_k1, 0.53);
    s.k2 = bin(CH_B4_k2, -0.0186);
  } else {
    double gamma1 = given(CH_B4_gamma1) ? bin(CH_B4_gamma1, 0.0) : 5.753e-12 * std::sqrt(ndep) / s.coxe;
    double gamma2 = given(CH_B4_gamma2) ? bin(CH_B4_gamma2, 0.0) : 5.753e-12 * std::sqrt(nsub) / s.coxe;
    double vbx = given(CH_B4_vbx) ? bin(CH_B4_vbx, 0.0) : s.phi - 7.7348e-4 * ndep * xt * xt;
    if (vbx > 0) vbx = -vbx;
    if (vbm > 0) vbm = -vbm;
    double T0g = gamma1 - gamma2;
    double T1g = std::sqrt(s.phi - vbx) - s.sqrtPhi;
    double T2g = std::sqrt(s.phi * (s.phi - vbm)) - s.phi;
    s.k2 = T0g * T1g / (2.0 * T2g + vbm);
    s.k1 = gamma2 - 2.0 * s.k2 * std::sqrt(s.phi - vbm);
  }
  if (s.k2 < 0.0) {
    double T0k = 0.5 * s.k1 / s.k2;
    s.vbsc = 0.9 * (s.phi - T0k * T0k);
    if (s.vbsc > -3.0) s.vbsc = -3.0;
    else if (s.vbsc < -30.0) s.vbsc = -30.0;
  } else s.vbsc = -30.0;
  if (s.vbsc > vbm) s.vbsc = vbm;
  s.k1ox = s.k1 * s.toxe / toxm;
  s.k2ox = s.k2 * s.toxe / toxm;

  bool vth0Given = given(CH_B4_vth0);
  double vth0 = bin(CH_B4_vth0, s.type > 0 ? 0.7 : -0.7);
  if (given(CH_B4_vfb)) s.vfb = bin(CH_B4_vfb, -1.0);
  else if (vth0Given) s.vfb = s.type * vth0 - s.phi - s.k1 * s.sqrtPhi;
  else s.vfb = -1.0;
  if (!vth0Given) vth0 = s.type * (s.vfb + s.phi + s.k1ox * s.sqrtPhi);
  s.vth0 = vth0;

  {
    double T3v = s.type * s.vth0 - s.vfb - s.phi;
    double T4v = T3v + T3v, T5v = 2.5 * T3v;
    s.vtfbphi1 = (s.type > 0) ? T4v : T5v;
    if (s.vtfbphi1 < 0.0) s.vtfbphi1 = 0.0;
    s.vtfbphi2 = 4.0 * T3v;
    if (s.vtfbphi2 < 0.0) s.vtfbphi2 = 0.0;
  }
  {
    double tmp = std::sqrt(EPSSI / (epsrox * EPS0) * s.toxe * s.Xdep0);
    double T0t = s.dsub * s.leff / tmp;
    if (T0t < EXP_THRESHOLD) {
      double T1t = std::exp(T0t), T2t = T1t - 1.0, T3t = T2t * T2t, T4t = T3t + 2.0 * T1t * MIN_EXP;
      s.theta0vb0 = T1t / T4t;
    } else s.theta0vb0 = 1.0 / (MAX_EXP - 2.0);
    T0t = drout * s.leff / tmp;
    double T5t;
    if (T0t < EXP_THRESHOLD) {
      double T1t = std::exp(T0t), T2t = T1t - 1.0, T3t = T2t * T2t, T4t = T3t + 2.0 * T1t * MIN_EXP;
      T5t = T1t / T4t;
    } else T5t = 1.0 / (MAX_EXP - 2.0);
    s.thetaRout = pdibl1 * T5t + pdibl2;
  }
  {
    double tmp = std::sqrt(s.Xdep0), tmp1 = s.vbi - s.phi, tmp2 = s.factor1 * tmp;
    double T0t = s.dvt1w * s.weff * s.leff / tmp2, T8t, T9t;
    if (T0t < EXP_THRESHOLD) {
      double T1t = std::exp(T0t), T2t = T1t - 1.0, T3t = T2t * T2t, T4t = T3t + 2.0 * T1t * MIN_EXP;
      T8t = T1t / T4t;
    } else T8t = 1.0 / (MAX_EXP - 2.0);
    T8t = s.dvt0w * T8t * tmp1;
    T0t = s.dvt1 * s.leff / tmp2;
    if (T0t < EXP_THRESHOLD) {
      double T1t = std::exp(T0t), T2t = T1t - 1.0, T3t = T2t * T2t, T4t = T3t + 2.0 * T1t * MIN_EXP;
      T9t = T1t / T4t;
    } else T9t = 1.0 / (MAX_EXP - 2.0);
    T9t = s.dvt0 * T9t * tmp1;
    double T4t = s.toxe * s.phi / (s.weff + s.w0);
    double T0l = std::sqrt(1.0 + s.lpe0 / s.leff);
    double T5t = s.k1ox * (T0l - 1.0) * s.sqrtPhi + (s.kt1 + s.kt1l / s.leff) * TR1;
    double tmp3 = s.type * s.vth0 - T8t - T9t + s.k3 * T4t + T5t;
    s.vfbzb = tmp3 - s.phi - s.k1 * s.sqrtPhi;
  }
  s.voffcbn = voff + get(CH_B4_voffl, 0.0) / s.leff;
  s.voffcbn *= (1.0 + tvoff * s.deltemp);
  s.mstar = 0.5 + std::atan(minv) / PI;

  // ---- junction diodes: saturation currents, capacitances (temperature adjusted) ----
  double jss = get(CH_B4_jss, 1.0e-4), jsws = get(CH_B4_jsws, 0.0), jswgs = get(CH_B4_jswgs, 0.0);
  double jsd = get(CH_B4_jsd, jss), jswd = get(CH_B4_jswd, jsws), jswgd = get(CH_B4_jswgd, jswgs);
  double njs = get(CH_B4_njs, 1.0), njd = get(CH_B4_njd, njs);
  double xtis = get(CH_B4_xtis, 3.0), xtid = get(CH_B4_xtid, xtis);
  double T0j = Eg0 / s.vtm0 - Eg / s.vtm, T1j = std::log(s.tratio);
  double T3s = std::exp((T0j + xtis * T1j) / njs), T3d = std::exp((T0j + xtid * T1j) / njd);
  if (s.deltemp == 0.0) { T3s = 1.0; T3d = 1.0; }
  jss *= T3s; jsws *= T3s; jswgs *= T3s; jsd *= T3d; jswd *= T3d; jswgd *= T3d;

  double cjs = get(CH_B4_cjs, 5.0e-4), cjd = get(CH_B4_cjd, cjs);
  double cjsws = get(CH_B4_cjsws, 5.0e-10), cjswd = get(CH_B4_cjswd, cjsws);
  double cjswgs = get(CH_B4_cjswgs, cjsws), cjswgd = get(CH_B4_cjswgd, cjswgs);
  s.mjs = get(CH_B4_mjs, 0.5); s.mjd = get(CH_B4_mjd, s.mjs);
  s.mjsws = get(CH_B4_mjsws, 0.33); s.mjswd = get(CH_B4_mjswd, s.mjsws);
  s.mjswgs = get(CH_B4_mjswgs, s.mjsws); s.mjswgd = get(CH_B4_mjswgd, s.mjswgs);
  double pbs = get(CH_B4_pbs, 1.0), pbd = get(CH_B4_pbd, pbs);
  double pbsws = get(CH_B4_pbsws, 1.0), pbswd = get(CH_B4_pbswd, pbsws);
  double pbswgs = get(CH_B4_pbswgs, pbsws), pbswgd = get(CH_B4_pbswgd, pbswgs);
  double tcj = get(CH_B4_tcj, 0.0), tcjsw = get(CH_B4_tcjsw, 0.0), tcjswg = get(CH_B4_tcjswg, 0.0);
  double tpb = get(CH_B4_tpb, 0.0), tpbsw = get(CH_B4_tpbsw, 0.0), tpbswg = get(CH_B4_tpbswg, 0.0);
  auto tcap = [&](double c, double tc) { double f = 1.0 + tc * s.deltemp; return f > 0 ? c * f : 0.0; };
  cjs = tcap(cjs, tcj); cjd = tcap(cjd, tcj);
  cjsws = tcap(cjsws, tcjsw); cjswd = tcap(cjswd, tcjsw);
  cjswgs = tcap(cjswgs, tcjswg); cjswgd = tcap(cjswgd, tcjswg);
  auto tphi = [&](double p, double tp) { double r = p - tp * s.deltemp; return r < 0.01 ? 0.01 : r; };
  s.PhiBS = tphi(pbs, tpb); s.PhiBD = tphi(pbd, tpb);
  s.PhiBSWS = tphi(pbsws, tpbsw); s.PhiBSWD = tphi(pbswd, tpbsw);
  s.PhiBSWGS = tphi(pbswgs, tpbswg); s.PhiBSWGD = tphi(pbswgd, tpbswg);

  // effective source/drain area & perimeter (geoMod 0, isolated S and D; BSIM4PAeffGeo)
  double dmcg = get(CH_B4_dmcg, 0.0), dmci = get(CH_B4_dmci, dmcg);
  double PSiso = 2.0 * (dmcg + dmci) + s.weffCJ, ASiso = (dmcg + dmci) * s.weffCJ;
  // nf fingers: number of end / internal diffusions (BSIM4NumFingerDiff, minSD default)
  double nuEndS, nuIntS, nuEndD, nuIntD;
  {
    int inf = (int)nf;
    if (inf % 2 != 0) { nuEndD = nuEndS = 1.0; nuIntD = nuIntS = (nf - 1.0) / 2.0; }
    else { nuEndD = 0.0; nuIntD = nf / 2.0; nuEndS = 2.0; nuIntS = nf / 2.0 - 1.0; }  // minSD = drain
  }
  double PSsha = 2.0 * dmcg, ASsha = dmcg * s.weffCJ;
  double Aseff, Adeff, Pseff, Pdeff;
  Aseff = igiven(CH_MOS_AS) ? ip[CH_MOS_AS] : nuEndS * ASiso + nuIntS * ASsha;
  Adeff = igiven(CH_MOS_AD) ? ip[CH_MOS_AD] : nuEndD * ASiso + nuIntD * ASsha;
  if (igiven(CH_MOS_PS)) Pseff = (permod == 0) ? ip[CH_MOS_PS] : ip[CH_MOS_PS] - s.weffCJ * nf;
  else Pseff = nuEndS * PSiso + nuIntS * PSsha;
  if (igiven(CH_MOS_PD)) Pdeff = (permod == 0) ? ip[CH_MOS_PD] : ip[CH_MOS_PD] - s.weffCJ * nf;
  else Pdeff = nuEndD * PSiso + nuIntD * PSsha;
  if (Pseff < 0) Pseff = 0;
  if (Pdeff < 0) Pdeff = 0;

  s.Isbs = Aseff * jss + Pseff * jsws + s.weffCJ * nf * jswgs;
  s.Isbd = Adeff * jsd + Pdeff * jswd + s.weffCJ * nf * jswgd;
  s.Nvtms = s.vtm * njs; s.Nvtmd = s.vtm * njd;
  double ijthsfwd = get(CH_B4_ijthsfwd, 0.1), ijthdfwd = get(CH_B4_ijthdfwd, ijthsfwd);
  s.vjsmFwd = s.Isbs > 0 ? s.Nvtms * std::log(ijthsfwd / s.Isbs + 1.0) : 0.0;
  s.IVjsmFwd = s.Isbs > 0 ? s.Isbs * std::exp(s.vjsmFwd / s.Nvtms) : 0.0;
  s.vjdmFwd = s.Isbd > 0 ? s.Nvtmd * std::log(ijthdfwd / s.Isbd + 1.0) : 0.0;
  s.IVjdmFwd = s.Isbd > 0 ? s.Isbd * std::exp(s.vjdmFwd / s.Nvtmd) : 0.0;
  s.czbs = cjs * Aseff; s.czbssw = cjsws * Pseff; s.czbsswg = cjswgs * s.weffCJ * nf;
  s.czbd = cjd * Adeff; s.czbdsw = cjswd * Pdeff; s.czbdswg = cjswgd * s.weffCJ * nf;
  return CH_OK;
}

// smooth exponential helper used by the SCE terms:  T1/(T2^2 + 2 T1 MIN_EXP), T1 = exp(x)
template <class S>
inline S b4_sce(const S& x) {
  using namespace b4c;
  if (val(x) < EXP_THRESHOLD) {
    S T1 = exp(x), T2 = T1 - 1.0, T3 = T2 * T2, T4 = T3 + 2.0 * T1 * MIN_EXP;
    return T1 / T4;
  }
  return S(1.0 / (MAX_EXP - 2.0));
}

// junction diode current (dioMod 1: forward-bias linearisation above ijthfwd, no breakdown) + gmin
template <class S>
inline S b4_diode_i(const S& vb, double Is, double Nvtm, double vjmFwd, double IVjmFwd, double gmin) {
  using namespace b4c;
  if (Is <= 0.0) return gmin * vb;
  if (val(vb) <= vjmFwd) {
    S T0 = vb / Nvtm;
    S ev = (val(T0) < -EXP_THRESHOLD) ? S(MIN_EXP) : exp(T0);
    return Is * (ev - 1.0) + gmin * vb;
  }
  double T0 = IVjmFwd / Nvtm;
  return (IVjmFwd - Is) + T0 * (vb - vjmFwd) + gmin * vb;
}

// junction depletion charge (bottom + sidewall + gate-edge sidewall)
template <class S>
inline S b4_junction_q(const S& vb, double cz, double czsw, double czswg, double pb, double pbsw, double pbswg,
                       double mj, double mjsw, double mjswg) {
  if (val(vb) == 0.0) {
    return (cz + czsw + czswg) * vb;  // value 0, derivative = zero-bias capacitance
  } else if (val(vb) < 0.0) {
    S q(0.0);
    if (cz > 0.0) { S arg = 1.0 - vb / pb; S sarg = exp(-mj * log(arg)); q += pb * cz * (1.0 - arg * sarg) / (1.0 - mj); }
    if (czsw > 0.0) { S arg = 1.0 - vb / pbsw; S sarg = exp(-mjsw * log(arg)); q += pbsw * czsw * (1.0 - arg * sarg) / (1.0 - mjsw); }
    if (czswg > 0.0) { S arg = 1.0 - vb / pbswg; S sarg = exp(-mjswg * log(arg)); q += pbswg * czswg * (1.0 - arg * sarg) / (1.0 - mjswg); }
    return q;
  }
  double T0 = cz + czsw + czswg;
  double T1c = cz * mj / pb + czsw * mjsw / pbsw + czswg * mjswg / pbswg;
  return vb * (T0 + 0.5 * T1c * vb);
}

// Full device evaluation.  vd,vg,vs,vb are the true terminal voltages; outputs are the currents
// flowing INTO the device at each terminal (d,g,s,b) and the terminal charges, i.e. the kcl!
// contributions of the VA functor (src/vasim.jl:836-839) before the multiplier.
template <class S>
inline void b4_eval(const B4Size& p, const S& vd_, const S& vg_, const S& vs_, const S& vb_, double gmin,
                    S I[4], S Q[4]) {
  using namespace b4c;
  const double tp = p.type;
  // polarity-normalised branch voltages
  S vds = tp * (vd_ - vs_), vgs = tp * (vg_ - vs_), vbs = tp * (vb_ - vs_);
  S vbd = vbs - vds, vgd = vgs - vds, vgb = vgs - vbs;
  // mode selection: forward (vds>=0) or reverse (source/drain interchanged)
  bool fwd = val(vds) >= 0.0;
  S Vds = fwd ? vds : -vds;
  S Vgs = fwd ? vgs : vgd;
  S Vbs = fwd ? vbs : vbd;

  // ---- effective body bias ----
  S T0 = Vbs - p.vbsc - 0.001;
  S T1 = sqrt(T0 * T0 - 0.004 * p.vbsc);
  S Vbseff;
  if (val(T0) >= 0.0) Vbseff = p.vbsc + 0.5 * (T0 + T1);
  else { S T2 = -0.002 / (T1 - T0); Vbseff = p.vbsc * (1.0 + T2); }
  {  // forward body bias clamp at 0.95 phi
    double T9 = 0.95 * p.phi;
    S T0b = T9 - Vbseff - 0.001;
    S T1b = sqrt(T0b * T0b + 0.004 * T9);
    Vbseff = T9 - 0.5 * (T0b + T1b);
  }
  S Phis = p.phi - Vbseff;
  S sqrtPhis = sqrt(Phis);
  S Xdep = p.Xdep0 * sqrtPhis / p.sqrtPhi;
  const double Leff = p.leff, Vtm = p.vtm;

  // ---- threshold voltage ----
  S T3 = sqrt(Xdep);
  double V0 = p.vbi - p.phi;
  S lt1, ltw;
  {
    S T0a = p.dvt2 * Vbseff, T1a;
    if (val(T0a) >= -0.5) T1a = 1.0 + T0a; else { S T4 = 1.0 / (3.0 + 8.0 * T0a); T1a = (1.0 + 3.0 * T0a) * T4; }
    lt1 = p.factor1 * T3 * T1a;
    S T0w = p.dvt2w * Vbseff, T1w;
    if (val(T0w) >= -0.5) T1w = 1.0 + T0w; else { S T4 = 1.0 / (3.0 + 8.0 * T0w); T1w = (1.0 + 3.0 * T0w) * T4; }
    ltw = p.factor1 * T3 * T1w;
  }
  S Theta0 = b4_sce(S(p.dvt1 * Leff / lt1));
  S Delt_vth = p.dvt0 * Theta0 * V0;
  S T5w = b4_sce(S(p.dvt1w * p.weff * Leff / ltw));
  S T2w = p.dvt0w * T5w * V0;
  double TempRatio = p.tratio - 1.0;
  double T0l = std::sqrt(1.0 + p.lpe0 / Leff);
  S T1t = p.k1ox * (T0l - 1.0) * p.sqrtPhi + (p.kt1 + p.kt1l / Leff + p.kt2 * Vbseff) * TempRatio;
  double Vth_NarrowW = p.toxe * p.phi / (p.weff + p.w0);
  S T3d = p.eta0 + p.etab * Vbseff;
  if (val(T3d) < 1.0e-4) { S T9 = 1.0 / (3.0 - 2.0e4 * T3d); T3d = (2.0e-4 - T3d) * T9; }
  S DIBL_Sft = T3d * p.theta0vb0 * Vds;
  double Lpe_Vb = std::sqrt(1.0 + p.lpeb / Leff);
  S Vth = tp * p.vth0 + (p.k1ox * sqrtPhis - p.k1 * p.sqrtPhi) * Lpe_Vb - p.k2ox * Vbseff - Delt_vth - T2w +
          (p.k3 + p.k3b * Vbseff) * Vth_NarrowW + T1t - DIBL_Sft;

  // ---- subthreshold swing factor n ----
  S tmp1 = EPSSI / Xdep;
  S tmp2 = p.nfactor * tmp1;
  S tmp3 = p.cdsc + p.cdscb * Vbseff + p.cdscd * Vds;
  S tmp4 = (tmp2 + tmp3 * Theta0 + p.cit) / p.coxe;
  S n;
  if (val(tmp4) >= -0.5) n = 1.0 + tmp4; else { S T0n = 1.0 / (3.0 + 8.0 * tmp4); n = (1.0 + 3.0 * tmp4) * T0n; }

  // ---- pocket-implant Vth correction (dvtp0 > 0) ----
  if (p.dvtp0 > 0.0) {
    S T0p = -p.dvtp1 * Vds;
    S T2p = (val(T0p) < -EXP_THRESHOLD) ? S(MIN_EXP) : exp(T0p);
    S T3p = Leff + p.dvtp0 * (1.0 + T2p);
    S T4p = Vtm * log(Leff / T3p);
    Vth -= n * T4p;
  }

  // ---- poly-gate depletion ----
  auto polydep = [&](const S& Vg) -> S {
    double T0p = p.vfb + p.phi;
    if (p.ngate > 1.0e18 && p.ngate < 1.0e25 && val(Vg) > T0p) {
      double T1p = 1.0e6 * Charge_q * EPSSI * p.ngate / (p.coxe * p.coxe);
      S T8 = Vg - T0p;
      S T4 = sqrt(1.0 + 2.0 * T8 / T1p);
      S T2 = 2.0 * T8 / (T4 + 1.0);
      S T3p = 0.5 * T2 * T2 / T1p;
      S T7 = 1.12 - T3p - 0.05;
      S T6 = sqrt(T7 * T7 + 0.224);
      S T5 = 1.12 - 0.5 * (T7 + T6);
      return Vg - T5;
    }
    return Vg;
  };
  S Vgs_eff = polydep(Vgs);
  S Vgst = Vgs_eff - Vth;

  // ---- effective gate overdrive Vgsteff ----
  S Vgsteff;
  {
    S T0g = n * Vtm;
    S T1g = p.mstar * Vgst;
    S T2g = T1g / T0g;
    S T10;
    if (val(T2g) > EXP_THRESHOLD) T10 = T1g;
    else if (val(T2g) < -EXP_THRESHOLD) T10 = Vtm * std::log(1.0 + MIN_EXP) * n;
    else { S ExpVgst = exp(T2g); T10 = n * (Vtm * log(1.0 + ExpVgst)); }
    S T1h = p.voffcbn - (1.0 - p.mstar) * Vgst;
    S T2h = T1h / T0g;
    S T9;
    if (val(T2h) < -EXP_THRESHOLD) T9 = p.mstar + (p.coxe * MIN_EXP / p.cdep0) * n;
    else if (val(T2h) > EXP_THRESHOLD) T9 = p.mstar + (p.coxe * MAX_EXP / p.cdep0) * n;
    else { S ExpVgst = exp(T2h); T9 = p.mstar + n * ((p.coxe / p.cdep0) * ExpVgst); }
    Vgsteff = T10 / T9;
  }

  // ---- effective width, Rds ----
  S T9w = sqrtPhis - p.sqrtPhi;
  S Weff = p.weff - 2.0 * (p.dwg * Vgsteff + p.dwb * T9w);
  if (val(Weff) < 2.0e-8) { S T0w2 = 1.0 / (6.0e-8 - 2.0 * Weff); Weff = 2.0e-8 * (4.0e-8 - Weff) * T0w2; }
  S Rds;
  {
    S T0r = 1.0 + p.prwg * Vgsteff;
    S T1r = p.prwb * T9w;
    S T2r = 1.0 / T0r + T1r;
    S T3r = T2r + sqrt(T2r * T2r + 0.01);
    Rds = p.rdswmin + T3r * (p.rds0 * 0.5);
  }

  // ---- bulk charge effect Abulk ----
  S Abulk0, Abulk;
  {
    S T9a = 0.5 * p.k1ox * Lpe_Vb / sqrtPhis;
    S T1a = T9a + p.k2ox - p.k3b * Vth_NarrowW;
    S T9b = sqrt(p.xj * Xdep);
    S tmp1a = Leff + 2.0 * T9b;
    S T5a = Leff / tmp1a;
    S tmp2a = p.a0 * T5a;
    double tmp4a = p.b0 / (p.weff + p.b1);
    S T2a = tmp2a + tmp4a;
    S T6a = T5a * T5a, T7a = T5a * T6a;
    Abulk0 = 1.0 + T1a * T2a;
    S T8a = p.ags * p.a0 * T7a;
    S dAbulk_dVg = -T1a * T8a;
    Abulk = Abulk0 + dAbulk_dVg * Vgsteff;
    if (val(Abulk0) < 0.1) { S T9c = 1.0 / (3.0 - 20.0 * Abulk0); Abulk0 = (0.2 - Abulk0) * T9c; }
    if (val(Abulk) < 0.1) { S T9c = 1.0 / (3.0 - 20.0 * Abulk); Abulk = (0.2 - Abulk) * T9c; }
    S T2k = p.keta * Vbseff, T0k;
    if (val(T2k) >= -0.9) T0k = 1.0 / (1.0 + T2k);
    else { S T1k = 1.0 / (0.8 + T2k); T0k = (17.0 + 20.0 * T2k) * T1k; }
    Abulk *= T0k;
    Abulk0 *= T0k;
  }

  // ---- mobility ----
  S T5m;
  if (p.mobmod == 0) {
    S T0m = Vgsteff + Vth + Vth;
    S T2m = p.ua + p.uc * Vbseff;
    S T3m = T0m / p.toxe;
    T5m = T3m * (T2m + p.ub * T3m);
  } else if (p.mobmod == 1) {
    S T0m = Vgsteff + Vth + Vth;
    S T2m = 1.0 + p.uc * Vbseff;
    S T3m = T0m / p.toxe;
    S T4m = T3m * (p.ua + p.ub * T3m);
    T5m = T4m * T2m;
  } else {
    S T0m = (Vgsteff + p.vtfbphi1) / p.toxe;
    S T1m = exp(p.eu * log(T0m));
    S T2m = p.ua + p.uc * Vbseff;
    T5m = T1m * T2m;
  }
  S Denomi;
  if (val(T5m) >= -0.8) Denomi = 1.0 + T5m; else { S T9m = 1.0 / (7.0 + 10.0 * T5m); Denomi = (0.6 + T5m) * T9m; }
  S ueff = p.u0temp / Denomi;

  // ---- saturation voltage Vdsat ----
  S WVCox = Weff * p.vsattemp * p.coxe;
  S WVCoxRds = WVCox * Rds;
  S Esat = 2.0 * p.vsattemp / ueff;
  S EsatL = Esat * Leff;
  S Lambda;
  if (p.a1 == 0.0) Lambda = S(p.a2);
  else if (p.a1 > 0.0) {
    double T0s = 1.0 - p.a2;
    S T1s = T0s - p.a1 * Vgsteff - 0.0001;
    S T2s = sqrt(T1s * T1s + 0.0004 * T0s);
    Lambda = p.a2 + T0s - 0.5 * (T1s + T2s);
  } else {
    S T1s = p.a2 + p.a1 * Vgsteff - 0.0001;
    S T2s = sqrt(T1s * T1s + 0.0004 * p.a2);
    Lambda = 0.5 * (T1s + T2s);
  }
  S Vgst2Vtm = Vgsteff + 2.0 * Vtm;
  S Vdsat;
  if (val(Rds) == 0.0 && val(Lambda) == 1.0) {
    S T0s = 1.0 / (Abulk * EsatL + Vgst2Vtm);
    Vdsat = EsatL * Vgst2Vtm * T0s;
  } else {
    S T9s = Abulk * WVCoxRds;
    S T7s = Vgst2Vtm * T9s;
    S T6s = Vgst2Vtm * WVCoxRds;
    S T0s = 2.0 * Abulk * (T9s - 1.0 + 1.0 / Lambda);
    S T1s = Vgst2Vtm * (2.0 / Lambda - 1.0) + Abulk * EsatL + 3.0 * T7s;
    S T2s = Vgst2Vtm * (EsatL + 2.0 * T6s);
    S T3s = sqrt(T1s * T1s - 2.0 * T0s * T2s);
    Vdsat = (T1s - T3s) / T0s;
  }

  // ---- effective Vds ----
  S Vdseff;
  {
    S T1e = Vdsat - Vds - p.delta;
    S T2e = sqrt(T1e * T1e + 4.0 * p.delta * Vdsat);
    if (val(T1e) >= 0.0) Vdseff = Vdsat - 0.5 * (T1e + T2e);
    else { S T4e = 2.0 * p.delta / (T2e - T1e); Vdseff = Vdsat * (1.0 - T4e); }
    if (val(Vds) == 0.0) Vdseff = Vds * 1.0;  // value 0; keep dVdseff/dVds = 1 limit
    if (val(Vdseff) > val(Vds)) Vdseff = Vds;
  }
  S diffVds = Vds - Vdseff;

  // ---- Vasat ----
  S Vasat;
  {
    S tmp4v = 1.0 - 0.5 * Abulk * Vdsat / Vgst2Vtm;
    S T9v = WVCoxRds * Vgsteff;
    S T0v = EsatL + Vdsat + 2.0 * T9v * tmp4v;
    S T9x = WVCoxRds * Abulk;
    S T1v = 2.0 / Lambda - 1.0 + T9x;
    Vasat = T0v / T1v;
  }

  // ---- channel conductance Idl = gche/(1+gche*Rds) without output-resistance effects ----
  S Idl, Coxeff_dc;
  {
    double tmp2i = 2.0e8 * p.toxp;
    S T0i = (Vgsteff + p.vtfbphi2) / tmp2i;
    S tmp3i = exp(0.7 * log(T0i));
    S Tcen = 1.9e-9 / (1.0 + tmp3i);
    Coxeff_dc = EPSSI * p.coxp / (EPSSI + p.coxp * Tcen);
    S CoxeffWovL = Coxeff_dc * Weff / Leff;
    S beta = ueff * CoxeffWovL;
    S AbovVgst2Vtm = Abulk / Vgst2Vtm;
    S T0j = 1.0 - 0.5 * Vdseff * AbovVgst2Vtm;
    S fgche1 = Vgsteff * T0j;
    S fgche2 = 1.0 + Vdseff / EsatL;
    S gche = beta * fgche1 / fgche2;
    Idl = gche / (1.0 + gche * Rds);
  }

  // ---- output resistance: CLM, DIBL, DITS, SCBE ----
  S FP;
  if (p.fprout <= 0.0) FP = S(1.0);
  else { S T9f = p.fprout * std::sqrt(Leff) / Vgst2Vtm; FP = 1.0 / (1.0 + T9f); }
  S PvagTerm;
  {
    S T9p = (p.pvag / EsatL) * Vgsteff;
    if (val(T9p) > -0.9) PvagTerm = 1.0 + T9p; else { S T4p = 1.0 / (17.0 + 20.0 * T9p); PvagTerm = (0.8 + T9p) * T4p; }
  }
  S Cclm, VACLM;
  if (p.pclm > MIN_EXP && val(diffVds) > 1.0e-10) {
    S T0c = 1.0 + Rds * Idl;
    S T2c = Vdsat / Esat;
    S T1c = Leff + T2c;
    Cclm = FP * PvagTerm * T0c * T1c / (p.pclm * p.litl);
    VACLM = Cclm * diffVds;
  } else { VACLM = S(MAX_EXP); Cclm = S(MAX_EXP); }
  S VADIBL;
  if (p.thetaRout > MIN_EXP) {
    S T8d = Abulk * Vdsat;
    S T0d = Vgst2Vtm * T8d;
    S T1d = Vgst2Vtm + T8d;
    VADIBL = (Vgst2Vtm - T0d / T1d) / p.thetaRout;
    S T7d = p.pdiblb * Vbseff;
    if (val(T7d) >= -0.9) VADIBL *= 1.0 / (1.0 + T7d);
    else { S T4d = 1.0 / (0.8 + T7d); VADIBL *= (17.0 + 20.0 * T7d) * T4d; }
    VADIBL *= PvagTerm;
  } else VADIBL = S(MAX_EXP);
  S VADITS;
  {
    S T0t = p.pditsd * Vds;
    S T1x = (val(T0t) > EXP_THRESHOLD) ? S(MAX_EXP) : exp(T0t);
    if (p.pdits > MIN_EXP) { double T2x = 1.0 + p.pditsl * Leff; VADITS = (1.0 + T2x * T1x) / p.pdits; VADITS *= FP; }
    else VADITS = S(MAX_EXP);
  }
  S VASCBE;
  if (p.pscbe2 > 0.0) {
    if (val(diffVds) > p.pscbe1 * p.litl / EXP_THRESHOLD) { S T0b = p.pscbe1 * p.litl / diffVds; VASCBE = Leff * exp(T0b) / p.pscbe2; }
    else VASCBE = S(MAX_EXP * Leff / p.pscbe2);
  } else VASCBE = S(MAX_EXP);

  S Idsa = Idl * (1.0 + diffVds / VADIBL);
  Idsa *= (1.0 + diffVds / VADITS);
  {
    S Va = Vasat + VACLM;
    S T0a = log(Va / Vasat);
    Idsa *= (1.0 + T0a / Cclm);
  }
  // ---- substrate (impact ionisation) current ----
  S Isub;
  {
    double tmpa = p.alpha0 + p.alpha1 * Leff;
    if (tmpa <= 0.0 || p.beta0 <= 0.0) Isub = S(0.0);
    else {
      double T2b = tmpa / Leff;
      S T1b;
      if (val(diffVds) > p.beta0 / EXP_THRESHOLD) { S T0b = -p.beta0 / diffVds; T1b = T2b * diffVds * exp(T0b); }
      else T1b = (T2b * MIN_EXP) * diffVds;
      Isub = T1b * (Idsa * Vdseff);
    }
  }
  // Idl/Idsa are channel conductances (gche/(1+gche*Rds)); the drain current is Ids*Vdseff
  S Ids = Idsa * (1.0 + diffVds / VASCBE) * Vdseff;
  Ids *= p.nf;
  Isub *= p.nf;

  // ---- GIDL (mode drain) / GISL (mode source) ----
  S Igidl(0.0), Igisl(0.0);
  if (p.agidl > 0.0 && p.bgidl > 0.0 && p.cgidl > 0.0) {
    double T0g = 3.0 * p.toxe;
    S Vbd_m = Vbs - Vds;
    S T1g = (Vds - Vgs_eff - p.egidl) / T0g;
    if (val(T1g) > 0.0 && val(Vbd_m) <= 0.0) {
      S T2g = p.bgidl / T1g;
      S Ig = (val(T2g) < 100.0) ? p.agidl * p.weffCJ * T1g * exp(-T2g) : (p.agidl * p.weffCJ * 3.720075976e-44) * T1g;
      S T4g = Vbd_m * Vbd_m, T5g = -Vbd_m * T4g;
      Igidl = Ig * (T5g / (p.cgidl + T5g)) * p.nf;
    }
    S T1s = (-Vgs_eff - p.egidl) / T0g;
    if (val(T1s) > 0.0 && val(Vbs) <= 0.0) {
      S T2g = p.bgidl / T1s;
      S Ig = (val(T2g) < 100.0) ? p.agidl * p.weffCJ * T1s * exp(-T2g) : (p.agidl * p.weffCJ * 3.720075976e-44) * T1s;
      S T4g = Vbs * Vbs, T5g = -Vbs * T4g;
      Igisl = Ig * (T5g / (p.cgidl + T5g)) * p.nf;
    }
  }

  // ---- junction diodes on the TRUE source and drain ----
  S Ibs = b4_diode_i(vbs, p.Isbs, p.Nvtms, p.vjsmFwd, p.IVjsmFwd, gmin);
  S Ibd = b4_diode_i(vbd, p.Isbd, p.Nvtmd, p.vjdmFwd, p.IVjdmFwd, gmin);

  // ---- intrinsic charges (capMod 2, charge-thickness model) in mode orientation ----
  S qgate(0.0), qbulk(0.0), qsrcm(0.0), qdrnm(0.0);
  if (p.xpart >= 0.0 && p.capmod != 0) {
    S VbseffCV = (val(Vbseff) < 0.0) ? Vbseff : (p.phi - Phis);
    double CoxWL = p.coxe * p.weffCV * p.leffCV * p.nf;
    S noffn = n * p.noff;
    S T0q = Vtm * noffn;
    S T1q = (Vgst - p.voffcv) / T0q;
    S VgsteffCV;
    if (val(T1q) > EXP_THRESHOLD) VgsteffCV = Vgst - p.voffcv;
    else if (val(T1q) < -EXP_THRESHOLD) VgsteffCV = T0q * std::log(1.0 + MIN_EXP);
    else VgsteffCV = T0q * log(1.0 + exp(T1q));

    S V3 = p.vfbzb - Vgs_eff + VbseffCV - DELTA_3;
    S T0f = (p.vfbzb <= 0.0) ? sqrt(V3 * V3 - 4.0 * DELTA_3 * p.vfbzb) : sqrt(V3 * V3 + 4.0 * DELTA_3 * p.vfbzb);
    S Vfbeff = p.vfbzb - 0.5 * (V3 + T0f);
    double Cox = p.coxp;
    double Tox = 1.0e8 * p.toxp;
    S T0c = (Vgs_eff - VbseffCV - p.vfbzb) / Tox;
    S tmpc = T0c * p.acde;
    S Tcen;
    if (val(tmpc) > -EXP_THRESHOLD && val(tmpc) < EXP_THRESHOLD) Tcen = p.ldeb * exp(tmpc);
    else if (val(tmpc) <= -EXP_THRESHOLD) Tcen = S(p.ldeb * MIN_EXP);
    else Tcen = S(p.ldeb * MAX_EXP);
    double LINK = 1.0e-3 * p.toxp;
    S V3c = p.ldeb - Tcen - LINK;
    S V4c = sqrt(V3c * V3c + 4.0 * LINK * p.ldeb);
    Tcen = p.ldeb - 0.5 * (V3c + V4c);
    S Ccen = EPSSI / Tcen;
    S Coxeff = (Cox / (Cox + Ccen)) * Ccen;
    S CoxWLcen = CoxWL * Coxeff / p.coxe;
    S Qac0 = CoxWLcen * (Vfbeff - p.vfbzb);

    double T0s = 0.5 * p.k1ox;
    S T3s = Vgs_eff - Vfbeff - VbseffCV - VgsteffCV;
    S T1s;
    if (p.k1ox == 0.0) T1s = S(0.0);
    else if (val(T3s) < 0.0) T1s = T0s + T3s / p.k1ox;
    else T1s = sqrt(T0s * T0s + T3s);
    S Qsub0 = CoxWLcen * p.k1ox * (T1s - T0s);

    double Denomi_q, T0d;
    if (p.k1ox <= 0.0) { Denomi_q = 0.25 * p.moin * Vtm; T0d = 0.5 * p.sqrtPhi; }
    else { Denomi_q = p.moin * Vtm * p.k1ox * p.k1ox; T0d = p.k1ox * p.sqrtPhi; }
    S T1d = 2.0 * T0d + VgsteffCV;
    S DeltaPhi = Vtm * log(1.0 + T1d * VgsteffCV / Denomi_q);

    S T3t = 4.0 * (Vth - p.vfbzb - p.phi);
    double Tox2 = Tox + Tox;
    S T0t = (val(T3t) >= 0.0) ? (VgsteffCV + T3t) / Tox2 : (VgsteffCV + 1.0e-20) / Tox2;
    S tmpt = exp(0.7 * log(T0t));
    Tcen = 1.9e-9 / (1.0 + tmpt);
    Ccen = EPSSI / Tcen;
    Coxeff = (Cox / (Cox + Ccen)) * Ccen;
    CoxWLcen = CoxWL * Coxeff / p.coxe;

    S AbulkCV = Abulk0 * p.abulkCVfactor;
    S VdsatCV = (VgsteffCV - DeltaPhi) / AbulkCV;
    S T0v = VdsatCV - Vds - DELTA_4;
    S T1v = sqrt(T0v * T0v + 4.0 * DELTA_4 * VdsatCV);
    S VdseffCV;
    if (val(T0v) >= 0.0) VdseffCV = VdsatCV - 0.5 * (T0v + T1v);
    else { S T3v = 2.0 * DELTA_4 / (T1v - T0v); VdseffCV = VdsatCV * (1.0 - T3v); }
    if (val(Vds) == 0.0) VdseffCV = Vds * 1.0;

    S T0 = AbulkCV * VdseffCV;
    S T1 = VgsteffCV - DeltaPhi;
    S T2 = 12.0 * (T1 - 0.5 * T0 + 1.0e-20);
    S T3 = T0 / T2;
    qgate = CoxWLcen * (T1 - T0 * (0.5 - T3));
    S T7 = 1.0 - AbulkCV;
    qbulk = CoxWLcen * T7 * (0.5 * VdseffCV - T0 * VdseffCV / T2);
    if (p.xpart > 0.5) {
      qsrcm = -CoxWLcen * (T1 / 2.0 + T0 / 4.0 - 0.5 * T0 * T0 / T2);
    } else if (p.xpart < 0.5) {
      S T2b = T2 / 12.0;
      S T3b = 0.5 * CoxWLcen / (T2b * T2b);
      S T4b = T1 * (2.0 * T0 * T0 / 3.0 + T1 * (T1 - 4.0 * T0 / 3.0)) - 2.0 * T0 * T0 * T0 / 15.0;
      qsrcm = -T3b * T4b;
    } else {
      qsrcm = -0.5 * qgate;
    }
    qgate += Qac0 + Qsub0 - qbulk;
    qbulk -= (Qac0 + Qsub0);
    qdrnm = -(qgate + qbulk + qsrcm);
  }
  // map mode-oriented intrinsic charges to true drain/source
  S qd_i = fwd ? qdrnm : qsrcm;
  S qs_i = fwd ? qsrcm : qdrnm;

  // ---- bias-dependent overlap charges on true terminals ----
  S qgdo, qgso;
  {
    S T0o = vgd + DELTA_1;
    S T1o = sqrt(T0o * T0o + 4.0 * DELTA_1);
    S T2o = 0.5 * (T0o - T1o);
    double T3o = p.weffCV * p.cgdl;
    S T4o = sqrt(1.0 - 4.0 * T2o / p.ckappad);
    qgdo = (p.cgdo + T3o) * vgd - T3o * (T2o + 0.5 * p.ckappad * (T4o - 1.0));
    S T0p = vgs + DELTA_1;
    S T1p = sqrt(T0p * T0p + 4.0 * DELTA_1);
    S T2p = 0.5 * (T0p - T1p);
    double T3p = p.weffCV * p.cgsl;
    S T4p = sqrt(1.0 - 4.0 * T2p / p.ckappas);
    qgso = (p.cgso + T3p) * vgs - T3p * (T2p + 0.5 * p.ckappas * (T4p - 1.0));
    qgdo *= p.nf;
    qgso *= p.nf;
  }
  S qgb = p.cgbo * vgb;

  // ---- junction charges ----
  S qbs = b4_junction_q(vbs, p.czbs, p.czbssw, p.czbsswg, p.PhiBS, p.PhiBSWS, p.PhiBSWGS, p.mjs, p.mjsws, p.mjswgs);
  S qbd = b4_junction_q(vbd, p.czbd, p.czbdsw, p.czbdswg, p.PhiBD, p.PhiBSWD, p.PhiBSWGD, p.mjd, p.mjswd, p.mjswgd);

  // ---- terminal currents (into the device) ----
  // mode-oriented channel, substrate and GIDL currents flow from mode-drain
  S Id_ch = fwd ? Ids : -Ids;              // true drain -> true source through the channel
  S Isub_d = fwd ? Isub : S(0.0), Isub_s = fwd ? S(0.0) : Isub;
  S Igd = fwd ? Igidl : Igisl, Igs = fwd ? Igisl : Igidl;  // true drain->bulk, true source->bulk
  S Id = Id_ch + Isub_d + Igd - Ibd;
  S Is = -Id_ch + Isub_s + Igs - Ibs;
  S Ib = -(Isub_d + Isub_s) - Igd - Igs + Ibs + Ibd;
  I[0] = tp * Id; I[1] = S(0.0); I[2] = tp * Is; I[3] = tp * Ib;

  // ---- terminal charges ----
  S Qg = qgate + qgdo + qgso + qgb;
  S Qb = qbulk - qgb + qbs + qbd;
  S Qd = qd_i - qgdo - qbd;
  S Qs = qs_i - qgso - qbs;
  Q[0] = tp * Qd; Q[1] = tp * Qg; Q[2] = tp * Qs; Q[3] = tp * Qb;
}

}  // namespace oracle
