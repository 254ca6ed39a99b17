#!/usr/bin/env python
"""Emulation (numpy, fp64) of VERDICT r05 Next 2: start the Newton-Schulz iteration on a PROVISIONAL shift while the lambda_min
filter is still squaring, re-establish the coupled invariant affinely when delta is final (sigma_ns.hip; covo.py:116-132).

What the filter knows from its NORM HISTORY alone at squaring j (no Rayleigh-Ritz evaluation): with X_j = T_(2^j)(Y0) / t_j
normalised, nrm_j = |X_j|_F^2, e_j = (1 - nrm_j) / 2:
  * T_(2^j)(y_1) <= t_j sqrt(nrm_j)  =>  acosh y_1 <= acosh(t_j sqrt(nrm_j)) / 2^j  =>  a rigorous LOWER bound lam_lo of lambda_min;
  * once e_j < 0.05: the gap bound of ritz_eval with y_1's upper bound in place of the Rayleigh quotient.
So at j* = first j with e_j < 0.05 the iteration can start on B' = A + delta' I, delta' = 1e-2 - lam_lo >= delta, with the table of
the DEFLATED interval [1e-2 + gap', s'] (the bottom eigenvalue sits below the interval and merely grows; its direction is put
right at the end with the Ritz vector, exactly as the deflation's fix-up today).  When the evaluation of X_kwin delivers delta:
Y <- Y - ((delta' - delta) / s') Z  (no product), lower bound rescaled by (lo' - tau) / lo', carry on.

Prints per Hessian: kwin, j*, tau = delta' - delta, iteration counts (today's deflated chain | provisional for m iterations +
rest), and the error of Sigma against eigh.  usage: prov_shift_emul.py [hessians.npz]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from filter_emul import gap_bound, ritz, run_filter, stats  # noqa: E402


def ns_coef(l):
    rho = np.sqrt(3.0 / (1 + l + l * l)) if l < 1 - 1e-9 else 1.0
    return 1.5 * rho, -0.5 * rho ** 3


def sigma_ref(A, sigma=0.5):
    w, U = np.linalg.eigh(A)
    o = w - w[0] + 1e-2
    n = len(w)
    log_const = (2 * n * 2 * np.log(sigma) + np.log(o).sum()) / n
    return (U * np.exp(0.5 * log_const - 0.5 * np.log(o))) @ U.T


def chain(A, m_switch=None, sigma=0.5, verbose=False):
    """m_switch None: today's chain (deflated, final delta from the start).  Else: provisional start, switch after m iterations."""
    A = 0.5 * (A + A.T)
    n = A.shape[0]
    ev = np.linalg.eigvalsh(A)
    hi, md = stats(A)
    cut = md + (hi - md) / 1024
    Xs, norms, ts, stop = run_filter(A, cut, hi)
    inv = 1 / (hi - cut)
    alpha, beta = (hi + cut) * inv, 2 * inv
    kwin = None
    for k in range(2, stop + 1):
        if ts[k] <= 1e10:
            continue
        w, u, res = ritz(A, Xs[k - 1], 4)
        g = 0.7 * gap_bound(norms, k, hi, cut, w[0])
        if g > 2e-2 and res <= 1e-8 * g and 0.5 * (1 - norms[k]) < 0.05:
            kwin, gapf, lmin, uvec = k, g, w[0], u
            break
    if kwin is None:
        return None
    delta = 1e-2 - lmin
    I = np.eye(n)
    out = dict(kwin=kwin)
    if m_switch is None:
        B = A + delta * I
        s = np.sqrt((B * B).sum()) * (1 + 1e-12)
        lo = 1e-2 + gapf
        tau_d = np.sqrt(lo * s)
        Y = B / s + ((tau_d - 1e-2) / s) * np.outer(uvec, uvec)
        Z = I.copy()
        l = np.sqrt(lo / s)
        its = 0
        while True:
            P = Z @ Y
            if ((P - I) ** 2).sum() < 1e-8 or its >= 14:
                break
            a, b = ns_coef(l)
            T = a * I + b * P
            Y, Z = Y @ T, T @ Z
            l = min(1.0, l * (a + b * l * l))
            its += 1
        Zf = Z + np.sqrt(s) * (10.0 - 1 / np.sqrt(tau_d)) * np.outer(uvec, uvec)
        scale = s
        out.update(its=its)
    else:
        logt = [0.0]
        for j in range(len(norms) - 1):
            logt.append(np.log(2) + 2 * logt[-1] + np.log(norms[j]))
        jstar = None
        for j in range(2, kwin + 1):
            if 0.5 * (1 - norms[j]) < 0.05 and logt[j] > np.log(1e4):
                jstar = j
                break
        if jstar is None:
            return None
        j = jstar
        e = 0.5 * (1 - norms[j])
        logT = logt[j] + 0.5 * np.log(norms[j])
        ach_up = (logT + np.log(2.0)) / 2 ** j if logT > 30 else np.arccosh(np.exp(logT)) / 2 ** j
        lam_lo = (alpha - np.cosh(ach_up)) / beta - 1e-9 * hi
        D = -np.log(e) * 2.0 ** (-j)
        y2 = np.cosh(max(ach_up - D, 0.0))
        gap_p = 0.7 * max((alpha - y2) / beta - lam_lo, 0.0)
        dp = 1e-2 - lam_lo
        tau = dp - delta
        assert tau >= 0, tau
        Bp = A + dp * I
        s = np.sqrt((Bp * Bp).sum()) * (1 + 1e-12)
        lo = 1e-2 + gap_p
        Y = Bp / s
        Z = I.copy()
        l = np.sqrt(lo / s)
        p1 = (1e-2 + tau) / s          # the bottom direction's eigenvalue of Z Y (tracked as a scalar; below the table's interval)
        z1 = 1.0                       # ... and of Z
        its = 0
        switched = False
        while True:
            if its == m_switch and not switched:
                Y = Y - (tau / s) * Z           # the switch: B' -> B, no product
                p1 *= 1e-2 / (1e-2 + tau)
                l = l * np.sqrt((lo - tau) / lo)
                switched = True
            P = Z @ Y
            if switched and ((P - I) ** 2).sum() - (1 - p1) ** 2 < 1e-8 or its >= 14:
                break
            a, b = ns_coef(l)
            T = a * I + b * P
            Y, Z = Y @ T, T @ Z
            t1 = a + b * p1
            p1, z1 = p1 * t1 * t1, z1 * t1
            l = min(1.0, l * (a + b * l * l))
            its += 1
        Zf = Z + (np.sqrt(s) * 10.0 - z1) * np.outer(uvec, uvec)
        scale = s
        out.update(its=its, jstar=jstar, tau=tau, gap_p=gap_p, gapf=gapf)
    # Sigma = c (B)^(-1/2) = c Zf / sqrt(scale)
    B = A + delta * I
    logdetB = np.linalg.slogdet(B)[1]
    c = np.exp(2 * np.log(sigma) + logdetB / (2 * n))
    S = c * 0.5 * (Zf + Zf.T) / np.sqrt(scale)
    ref = sigma_ref(A, sigma)
    out.update(err=np.linalg.norm(S - ref) / np.linalg.norm(ref))
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    z = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "..", "tests", "golden", "hessians_r03.npz"))
    print("task          kwin j*  tau      gap'/gap   today: its err      provisional (m = 1 / 2 / 3): its err")
    for task in z.files:
        for A in z[task]:
            base = chain(A)
            if base is None:
                print(task[:12], "no converged pair")
                continue
            row = f"{task[:12]:12s}  {base['kwin']:3d}"
            provs = [chain(A, m) for m in (1, 2, 3)]
            if provs[0] is None:
                print(row, "  no provisional trigger")
                continue
            p = provs[0]
            row += f" {p['jstar']:3d}  {p['tau']:.1e}  {p['gap_p']:.3f}/{p['gapf']:.3f}   {base['its']:2d} {base['err']:.1e}     "
            row += "  ".join(f"{q['its']:2d} {q['err']:.1e}" for q in provs)
            print(row)
