#!/usr/bin/env python
"""Offline (numpy) emulation of the lambda_min filter of sigma_ns.hip on dumped closed-loop Hessians
(scripts/dump_hessians.py): how many squarings the stopping rule takes, and what the Rayleigh-Ritz step
would already deliver after k squarings for different block sizes and cuts.  Analysis only."""
import sys
import numpy as np

def stats(A):
    gm = np.abs(A).sum(1).max(); f2 = (A * A).sum(); md = A.diagonal().min()
    hi = min(gm, np.sqrt(f2)) * (1 + 1e-12) + 1e-3
    return hi, md

def run_filter(A, cut, hi, kmax=16):
    n = A.shape[0]
    alpha = (hi + cut) / (hi - cut); beta = 2 / (hi - cut)
    X = alpha * np.eye(n) - beta * A
    nrm = (X * X).sum(); t = 1.0
    Xs = []; norms = [nrm]; ts = [t]
    stop = None
    prev = None
    for step in range(kmax):
        if stop is None and step >= 2 and t > 1e10 and abs(nrm - prev) <= 1e-7 * nrm:
            stop = step
        with np.errstate(over='ignore'):
            t_out = 2 * t * t * nrm
        X = X @ X / nrm - np.eye(n) / t_out
        X = 0.5 * (X + X.T)
        prev = nrm; nrm = (X * X).sum(); t = t_out
        Xs.append(X); norms.append(nrm); ts.append(t)
    if stop is None: stop = kmax
    return Xs, norms, ts, stop

def ritz(A, X, r):
    d = X.diagonal().copy()
    # null rows out of the picks
    null = np.abs(A).sum(1) == 0
    d[null] = -1e300
    idx = np.argsort(-d)[:r]
    V, _ = np.linalg.qr(X[:, idx])
    Hm = V.T @ A @ V
    w, c = np.linalg.eigh(0.5 * (Hm + Hm.T))
    u = V @ c[:, 0]
    res = np.linalg.norm(A @ u - w[0] * u)
    return w, u, res

if __name__ == "__main__":
    z = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/hessians.npz")
    for task in z.files:
        Rs = z[task]
        print("==", task, Rs.shape)
        rows = []
        for A in Rs:
            A = 0.5 * (A + A.T)
            ev = np.linalg.eigvalsh(A)
            hi, md = stats(A)
            cut0 = md + (hi - md) / 1024
            Xs, norms, ts, stop = run_filter(A, cut0, hi)
            # accuracy after k squarings for block sizes 4, 8, 16
            acc = {}
            for r in (4, 8, 16):
                ks = None
                for k in range(2, 16):
                    w, u, res = ritz(A, Xs[k - 1], r)
                    if abs(w[0] - ev[0]) < 1e-11 * max(1, abs(ev[0])) and res < 1e-8 * max(ev[1] - ev[0], 1e-3):
                        ks = k; break
                acc[r] = ks
            rows.append((ev[0], ev[1] - ev[0], ev[4] - ev[0], ev[-1], hi, md, cut0 - ev[0], stop, acc[4], acc[8], acc[16]))
        rows = np.array([[np.nan if v is None else v for v in r] for r in rows], dtype=float)
        np.set_printoptions(linewidth=200, precision=3, suppress=True)
        print("lmin      gap12   gap15   lmax    hi      mindiag  cut-l1  stop  k(r=4) k(r=8) k(r=16)")
        for r in rows[::4]: print(" ".join(f"{v:8.3f}" for v in r))
        print("mean stop %.2f  k4 %.2f k8 %.2f k16 %.2f" % tuple(np.nanmean(rows[:, 7:11], axis=0)))

def gap_bound(norms, k, hi, cut, l1):
    """ns_ritz_kernel's lower bound of lambda_2 - lambda_1 from the norm history up to squaring k"""
    inv = 1.0 / (hi - cut); alpha = (hi + cut) * inv; beta = 2 * inv
    y1 = alpha - beta * l1
    for j in range(1, k + 1):
        e = 0.5 * (1.0 - norms[j])
        if 1e-13 < e < 0.05 and y1 > 1.0:
            D = -np.log(e) * 2.0 ** (-j)
            ac = np.log(y1 + np.sqrt(y1 * y1 - 1)) - D
            ex = np.exp(max(ac, 0.0)); y2 = 0.5 * (ex + 1 / ex)
            return max((alpha - y2) / beta - l1, 0.0)
    return 0.0

def early_rule(task_R, r=4, k0=4):
    out = []
    for A in task_R:
        A = 0.5 * (A + A.T)
        ev = np.linalg.eigvalsh(A)
        hi, md = stats(A); cut = md + (hi - md) / 1024
        Xs, norms, ts, stop = run_filter(A, cut, hi)
        kwin = None
        for k in range(k0, stop + 1):
            if ts[k] <= 1e10: continue
            w, u, res = ritz(A, Xs[k - 1], r)
            g = 0.7 * gap_bound(norms, k, hi, cut, w[0])
            if g > 2e-2 and res <= 1e-8 * g:
                kwin = k; err = w[0] - ev[0]; break
        if kwin is None:
            kwin = stop; w, u, res = ritz(A, Xs[stop - 1], r); err = w[0] - ev[0]
        out.append((stop, kwin, err, res, ev[1] - ev[0]))
    return np.array(out)

if __name__ == "__main__":
    for task in z.files:
        for r in (4, 8):
            o = early_rule(z[task], r)
            print(task, "r=%d" % r, "mean stop %.2f  mean kwin %.2f  max |lambda err| %.2e  max res/gap %.2e" %
                  (o[:, 0].mean(), o[:, 1].mean(), np.abs(o[:, 2]).max(), (o[:, 3] / o[:, 4]).max()),
                  " hist(stop-kwin):", np.bincount((o[:, 0] - o[:, 1]).astype(int)))

def more_pairs(task_R):
    """at kwin (first iterate whose bottom pair passes): how good are Ritz pairs 2 .. 4?"""
    rows = []
    for A in task_R:
        A = 0.5 * (A + A.T)
        ev = np.linalg.eigvalsh(A)
        hi, md = stats(A); cut = md + (hi - md) / 1024
        Xs, norms, ts, stop = run_filter(A, cut, hi)
        for k in range(2, stop + 1):
            if ts[k] <= 1e10: continue
            d = Xs[k - 2].diagonal().copy(); d[np.abs(A).sum(1) == 0] = -1e300
            idx = np.argsort(-d)[:4]
            V, _ = np.linalg.qr(Xs[k - 1][:, idx])
            Hm = V.T @ A @ V; w, c = np.linalg.eigh(0.5 * (Hm + Hm.T)); U = V @ c
            res = np.linalg.norm(A @ U - U * w, axis=0)
            g = 0.7 * gap_bound(norms, k, hi, cut, w[0])
            if g > 2e-2 and res[0] <= 1e-8 * g:
                rows.append((k, *(res / np.maximum(ev[1:5] - ev[0:4], 1e-9)), *(w - ev[:4]), ev[1] - ev[0], ev[3] - ev[0], ev[4] - ev[0], ev[-1] - ev[0]))
                break
    return np.array(rows)

if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "pairs":
    for task in z.files:
        r = more_pairs(z[task])
        print(task, "residual/gap of Ritz pairs 1..4 at kwin (median, max):", np.median(r[:, 1:5], 0), r[:, 1:5].max(0))
        print("   gap12 / gap14 / gap15 medians:", np.median(r[:, 9]), np.median(r[:, 10]), np.median(r[:, 11]))
