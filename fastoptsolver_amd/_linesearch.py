"""Host-side scalar logic of the L-BFGS line search: Moré–Thuente (MINPACK-2 dcsrch/dcstep) with the constants
L-BFGS-B uses (ftol 1e-3, gtol 0.9, xtol 0.1, stpmin 0, stpmax 1e10).  Only scalars live here; every vector
operation of the search is a device kernel.  Written from the published algorithm (Moré & Thuente, ACM TOMS 20,
1994); checked in tests against SciPy's DCSRCH and against the oracle's independent restatement."""
import math

FTOL, GTOL, XTOL, STPMIN, STPMAX = 1e-3, 0.9, 0.1, 0.0, 1e10


def _cubic_min(a, fa, da, c, fc, dc, clamp_sqrt=False):
    """Ingredients of the cubic through (a, fa, da), (c, fc, dc): returns (theta, gamma)."""
    theta = 3.0 * (fa - fc) / (c - a) + da + dc
    scale = max(abs(theta), abs(da), abs(dc))
    rad = (theta / scale) ** 2 - (da / scale) * (dc / scale)
    if clamp_sqrt:
        rad = max(0.0, rad)
    return theta, scale * math.sqrt(rad)


def next_trial(iv, stp, f, d):
    """One dcstep: update the interval dict ``iv`` (keys lo/flo/dlo, hi/fhi/dhi, brackt, smin, smax) with the
    trial (stp, f, d) and return the next trial step."""
    lo, flo, dlo = iv["lo"], iv["flo"], iv["dlo"]
    hi, fhi, dhi = iv["hi"], iv["fhi"], iv["dhi"]
    opposite = d * (dlo / abs(dlo)) < 0.0
    if f > flo:                                             # value went up: minimiser is bracketed
        theta, gam = _cubic_min(lo, flo, dlo, stp, f, d)
        if stp < lo:
            gam = -gam
        r = ((gam - dlo) + theta) / (((gam - dlo) + gam) + d)
        cub = lo + r * (stp - lo)
        quad = lo + ((dlo / ((flo - f) / (stp - lo) + dlo)) / 2.0) * (stp - lo)
        new = cub if abs(cub - lo) < abs(quad - lo) else cub + (quad - cub) / 2.0
        iv["brackt"] = True
    elif opposite:                                          # value down, slope changed sign
        theta, gam = _cubic_min(lo, flo, dlo, stp, f, d)
        if stp > lo:
            gam = -gam
        r = ((gam - d) + theta) / (((gam - d) + gam) + dlo)
        cub = stp + r * (lo - stp)
        sec = stp + (d / (d - dlo)) * (lo - stp)
        new = cub if abs(cub - stp) > abs(sec - stp) else sec
        iv["brackt"] = True
    elif abs(d) < abs(dlo):                                 # value down, same-sign slope, flatter
        theta, gam = _cubic_min(lo, flo, dlo, stp, f, d, clamp_sqrt=True)
        if stp > lo:
            gam = -gam
        r = ((gam - d) + theta) / ((gam + (dlo - d)) + gam)
        if r < 0.0 and gam != 0.0:
            cub = stp + r * (lo - stp)
        else:
            cub = iv["smax"] if stp > lo else iv["smin"]
        sec = stp + (d / (d - dlo)) * (lo - stp)
        if iv["brackt"]:
            new = cub if abs(cub - stp) < abs(sec - stp) else sec
            lim = stp + 0.66 * (hi - stp)
            new = min(lim, new) if stp > lo else max(lim, new)
        else:
            new = cub if abs(cub - stp) > abs(sec - stp) else sec
            new = max(iv["smin"], min(iv["smax"], new))
    else:                                                   # value down, same-sign slope, not flatter
        if iv["brackt"]:
            theta, gam = _cubic_min(stp, f, d, hi, fhi, dhi)
            if stp > hi:
                gam = -gam
            r = ((gam - d) + theta) / (((gam - d) + gam) + dhi)
            new = stp + r * (hi - stp)
        else:
            new = iv["smax"] if stp > lo else iv["smin"]
    if f > flo:
        iv["hi"], iv["fhi"], iv["dhi"] = stp, f, d
    else:
        if opposite:
            iv["hi"], iv["fhi"], iv["dhi"] = lo, flo, dlo
        iv["lo"], iv["flo"], iv["dlo"] = stp, f, d
    return new


class LineSearch:
    """Reverse communication: ``begin(stp, f0, d0)`` then ``step(stp, f, d) -> next stp`` until
    ``status`` is no longer 'FG' ('CONVERGENCE', 'WARNING: ...', 'ERROR: ...')."""

    def begin(self, stp, f0, d0):
        if stp < STPMIN or stp > STPMAX or d0 >= 0.0:
            self.status = "ERROR: bad initial step or non-descent direction"
            return stp
        self.f0, self.d0, self.dtest = f0, d0, FTOL * d0
        self.stage1 = True
        self.width, self.width1 = STPMAX - STPMIN, 2.0 * (STPMAX - STPMIN)
        self.iv = dict(lo=0.0, flo=f0, dlo=d0, hi=0.0, fhi=f0, dhi=d0, brackt=False, smin=0.0, smax=5.0 * stp)
        self.status = "FG"
        return stp

    def step(self, stp, f, d):
        iv = self.iv
        ftest = self.f0 + stp * self.dtest
        if self.stage1 and f <= ftest and d >= 0.0:
            self.stage1 = False
        status = "FG"
        if iv["brackt"] and (stp <= iv["smin"] or stp >= iv["smax"]):
            status = "WARNING: ROUNDING ERRORS PREVENT PROGRESS"
        if iv["brackt"] and iv["smax"] - iv["smin"] <= XTOL * iv["smax"]:
            status = "WARNING: XTOL TEST SATISFIED"
        if stp == STPMAX and f <= ftest and d <= self.dtest:
            status = "WARNING: STP = STPMAX"
        if stp == STPMIN and (f > ftest or d >= self.dtest):
            status = "WARNING: STP = STPMIN"
        if f <= ftest and abs(d) <= GTOL * (-self.d0):
            status = "CONVERGENCE"
        self.status = status
        if status != "FG":
            return stp
        if self.stage1 and f <= iv["flo"] and f > ftest:
            # first stage: search on psi(t) = f(t) - f0 - ftol*d0*t
            gt = self.dtest
            shifted = dict(lo=iv["lo"], flo=iv["flo"] - iv["lo"] * gt, dlo=iv["dlo"] - gt,
                           hi=iv["hi"], fhi=iv["fhi"] - iv["hi"] * gt, dhi=iv["dhi"] - gt,
                           brackt=iv["brackt"], smin=iv["smin"], smax=iv["smax"])
            stp = next_trial(shifted, stp, f - stp * gt, d - gt)
            iv.update(lo=shifted["lo"], flo=shifted["flo"] + shifted["lo"] * gt, dlo=shifted["dlo"] + gt,
                      hi=shifted["hi"], fhi=shifted["fhi"] + shifted["hi"] * gt, dhi=shifted["dhi"] + gt,
                      brackt=shifted["brackt"])
        else:
            stp = next_trial(iv, stp, f, d)
        if iv["brackt"]:
            if abs(iv["hi"] - iv["lo"]) >= 0.66 * self.width1:
                stp = iv["lo"] + 0.5 * (iv["hi"] - iv["lo"])
            self.width1, self.width = self.width, abs(iv["hi"] - iv["lo"])
            iv["smin"], iv["smax"] = min(iv["lo"], iv["hi"]), max(iv["lo"], iv["hi"])
        else:
            iv["smin"] = stp + 1.1 * (stp - iv["lo"])
            iv["smax"] = stp + 4.0 * (stp - iv["lo"])
        stp = min(max(stp, STPMIN), STPMAX)
        if iv["brackt"] and (stp <= iv["smin"] or stp >= iv["smax"] or iv["smax"] - iv["smin"] <= XTOL * iv["smax"]):
            stp = iv["lo"]
        return stp
