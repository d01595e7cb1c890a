// Host side of the L-BFGS driver: the Moré–Thuente line search (MINPACK-2 dcsrch / dcstep with L-BFGS-B's constants) and
// the unbounded L-BFGS-B iteration that scipy.optimize.fmin_l_bfgs_b runs for lbfgs.py:64 (spec: SURVEY.md 8c; the
// Python restatement fastoptsolver_amd/_linesearch.py + lbfgs.py is checked against SciPy's DCSRCH and goldens, this is
// the same logic without an interpreter between the kernels).  Only scalars live here; every vector operation is a
// device kernel (fos_gemv_pair_dd, lbfgs_two_loop_kernel<double>, vec_stats / vec_axpby in fp64).
// Written from the published algorithm (Moré & Thuente, ACM TOMS 20, 1994; Byrd, Lu, Nocedal, Zhu 1995).
#pragma once
#include <algorithm>
#include <cmath>

#include "../../include/fos.h"

namespace fos {

constexpr double LS_FTOL = 1e-3, LS_GTOL = 0.9, LS_XTOL = 0.1, LS_STPMIN = 0.0, LS_STPMAX = 1e10;

struct Interval {
  double lo, flo, dlo, hi, fhi, dhi, smin, smax;
  bool brackt;
};

inline void cubic_ingredients(double a, double fa, double da, double c, double fc, double dc, bool clamp_sqrt, double* theta,
                              double* gamma) {
  const double th = 3.0 * (fa - fc) / (c - a) + da + dc;
  const double scale = std::max(std::fabs(th), std::max(std::fabs(da), std::fabs(dc)));
  double rad = (th / scale) * (th / scale) - (da / scale) * (dc / scale);
  if (clamp_sqrt) rad = std::max(0.0, rad);
  *theta = th;
  *gamma = scale * std::sqrt(rad);
}

// One dcstep: update the interval with the trial (stp, f, d) and return the next trial step.
inline double next_trial(Interval& iv, double stp, double f, double d) {
  const double lo = iv.lo, flo = iv.flo, dlo = iv.dlo, hi = iv.hi, fhi = iv.fhi, dhi = iv.dhi;
  const bool opposite = d * (dlo / std::fabs(dlo)) < 0.0;
  double nw, theta, gam;
  if (f > flo) {                                             // value went up: minimiser is bracketed
    cubic_ingredients(lo, flo, dlo, stp, f, d, false, &theta, &gam);
    if (stp < lo) gam = -gam;
    const double r = ((gam - dlo) + theta) / (((gam - dlo) + gam) + d);
    const double cub = lo + r * (stp - lo);
    const double quad = lo + ((dlo / ((flo - f) / (stp - lo) + dlo)) / 2.0) * (stp - lo);
    nw = std::fabs(cub - lo) < std::fabs(quad - lo) ? cub : cub + (quad - cub) / 2.0;
    iv.brackt = true;
  } else if (opposite) {                                     // value down, slope changed sign
    cubic_ingredients(lo, flo, dlo, stp, f, d, false, &theta, &gam);
    if (stp > lo) gam = -gam;
    const double r = ((gam - d) + theta) / (((gam - d) + gam) + dlo);
    const double cub = stp + r * (lo - stp);
    const double sec = stp + (d / (d - dlo)) * (lo - stp);
    nw = std::fabs(cub - stp) > std::fabs(sec - stp) ? cub : sec;
    iv.brackt = true;
  } else if (std::fabs(d) < std::fabs(dlo)) {                // value down, same-sign slope, flatter
    cubic_ingredients(lo, flo, dlo, stp, f, d, true, &theta, &gam);
    if (stp > lo) gam = -gam;
    const double r = ((gam - d) + theta) / ((gam + (dlo - d)) + gam);
    double cub;
    if (r < 0.0 && gam != 0.0) cub = stp + r * (lo - stp);
    else cub = stp > lo ? iv.smax : iv.smin;
    const double sec = stp + (d / (d - dlo)) * (lo - stp);
    if (iv.brackt) {
      nw = std::fabs(cub - stp) < std::fabs(sec - stp) ? cub : sec;
      const double lim = stp + 0.66 * (hi - stp);
      nw = stp > lo ? std::min(lim, nw) : std::max(lim, nw);
    } else {
      nw = std::fabs(cub - stp) > std::fabs(sec - stp) ? cub : sec;
      nw = std::max(iv.smin, std::min(iv.smax, nw));
    }
  } else {                                                   // value down, same-sign slope, not flatter
    if (iv.brackt) {
      cubic_ingredients(stp, f, d, hi, fhi, dhi, false, &theta, &gam);
      if (stp > hi) gam = -gam;
      const double r = ((gam - d) + theta) / (((gam - d) + gam) + dhi);
      nw = stp + r * (hi - stp);
    } else {
      nw = stp > lo ? iv.smax : iv.smin;
    }
  }
  if (f > flo) {
    iv.hi = stp; iv.fhi = f; iv.dhi = d;
  } else {
    if (opposite) { iv.hi = lo; iv.fhi = flo; iv.dhi = dlo; }
    iv.lo = stp; iv.flo = f; iv.dlo = d;
  }
  return nw;
}

}  // namespace fos

// C layout (include/fos.h): reverse-communication state of one search.
inline double fos_ls_begin_impl(fos_linesearch* ls, double stp, double f0, double d0) {
  if (stp < fos::LS_STPMIN || stp > fos::LS_STPMAX || d0 >= 0.0) {
    ls->status = FOS_LS_ERROR;
    return stp;
  }
  ls->f0 = f0; ls->d0 = d0; ls->dtest = fos::LS_FTOL * d0;
  ls->stage1 = 1;
  ls->width = fos::LS_STPMAX - fos::LS_STPMIN;
  ls->width1 = 2.0 * (fos::LS_STPMAX - fos::LS_STPMIN);
  ls->lo = 0.0; ls->flo = f0; ls->dlo = d0; ls->hi = 0.0; ls->fhi = f0; ls->dhi = d0;
  ls->brackt = 0; ls->smin = 0.0; ls->smax = 5.0 * stp;
  ls->status = FOS_LS_FG;
  return stp;
}

inline double fos_ls_step_impl(fos_linesearch* ls, double stp, double f, double d) {
  using namespace fos;
  const double ftest = ls->f0 + stp * ls->dtest;
  if (ls->stage1 && f <= ftest && d >= 0.0) ls->stage1 = 0;
  int status = FOS_LS_FG;
  if (ls->brackt && (stp <= ls->smin || stp >= ls->smax)) status = FOS_LS_WARNING;                 // rounding errors
  if (ls->brackt && ls->smax - ls->smin <= LS_XTOL * ls->smax) status = FOS_LS_WARNING;            // xtol test
  if (stp == LS_STPMAX && f <= ftest && d <= ls->dtest) status = FOS_LS_WARNING;                   // stp = stpmax
  if (stp == LS_STPMIN && (f > ftest || d >= ls->dtest)) status = FOS_LS_WARNING;                  // stp = stpmin
  if (f <= ftest && std::fabs(d) <= LS_GTOL * (-ls->d0)) status = FOS_LS_CONVERGENCE;
  ls->status = status;
  if (status != FOS_LS_FG) return stp;
  Interval iv{ls->lo, ls->flo, ls->dlo, ls->hi, ls->fhi, ls->dhi, ls->smin, ls->smax, ls->brackt != 0};
  if (ls->stage1 && f <= iv.flo && f > ftest) {
    // first stage: search on psi(t) = f(t) - f0 - ftol*d0*t
    const double gt = ls->dtest;
    Interval sh{iv.lo, iv.flo - iv.lo * gt, iv.dlo - gt, iv.hi, iv.fhi - iv.hi * gt, iv.dhi - gt, iv.smin, iv.smax, iv.brackt};
    stp = next_trial(sh, stp, f - stp * gt, d - gt);
    iv.lo = sh.lo; iv.flo = sh.flo + sh.lo * gt; iv.dlo = sh.dlo + gt;
    iv.hi = sh.hi; iv.fhi = sh.fhi + sh.hi * gt; iv.dhi = sh.dhi + gt;
    iv.brackt = sh.brackt;
  } else {
    stp = next_trial(iv, stp, f, d);
  }
  if (iv.brackt) {
    if (std::fabs(iv.hi - iv.lo) >= 0.66 * ls->width1) stp = iv.lo + 0.5 * (iv.hi - iv.lo);
    ls->width1 = ls->width;
    ls->width = std::fabs(iv.hi - iv.lo);
    iv.smin = std::min(iv.lo, iv.hi);
    iv.smax = std::max(iv.lo, iv.hi);
  } else {
    iv.smin = stp + 1.1 * (stp - iv.lo);
    iv.smax = stp + 4.0 * (stp - iv.lo);
  }
  stp = std::min(std::max(stp, LS_STPMIN), LS_STPMAX);
  if (iv.brackt && (stp <= iv.smin || stp >= iv.smax || iv.smax - iv.smin <= LS_XTOL * iv.smax)) stp = iv.lo;
  ls->lo = iv.lo; ls->flo = iv.flo; ls->dlo = iv.dlo; ls->hi = iv.hi; ls->fhi = iv.fhi; ls->dhi = iv.dhi;
  ls->smin = iv.smin; ls->smax = iv.smax; ls->brackt = iv.brackt ? 1 : 0;
  return stp;
}
