"""Second, independent restatement (numpy / scipy, float64) of the per-stage arithmetic of SURVEY.md Appendix A,
used to cross-check the C++ oracle (SURVEY §4 test plan item 3).  Small inputs only."""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def se3_exp(d):
    """Rodrigues + V matrix; d = [omega, v] (fast_gicp so3.hpp)."""
    w, v = np.asarray(d[:3], float), np.asarray(d[3:], float)
    th = np.linalg.norm(w)
    W = skew(w)
    if th < 1e-10:
        R = np.eye(3) + W
        V = R
    else:
        R = np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th**2 * W @ W
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * W + (th - np.sin(th)) / th**3 * W @ W
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ v
    return T


def gicp_covariances(xyz, k=20):
    xyz = np.asarray(xyz, np.float64)
    tree = cKDTree(xyz)
    _, idx = tree.query(xyz, k=k)
    nb = xyz[idx]                                  # [n,k,3]
    c = nb - nb.mean(axis=1, keepdims=True)
    C = np.einsum("nki,nkj->nij", c, c) / k
    C = C + 1e-3 * np.eye(3)
    Ci = np.linalg.inv(C)
    Ci = Ci / np.linalg.norm(Ci, axis=(1, 2), keepdims=True)
    return np.linalg.inv(Ci)


def gicp_linearize(src, tgt, cov_s, cov_t, T, max_corr):
    src, tgt = np.asarray(src, np.float64), np.asarray(tgt, np.float64)
    R, t = T[:3, :3], T[:3, 3]
    q = src @ R.T + t
    d, j = cKDTree(tgt).query(q)
    valid = d * d < max_corr**2
    H, b, err = np.zeros((6, 6)), np.zeros(6), 0.0
    for i in np.flatnonzero(valid):
        M = np.linalg.inv(cov_t[j[i]] + R @ cov_s[i] @ R.T)
        e = tgt[j[i]] - q[i]
        J = np.hstack([skew(q[i]), -np.eye(3)])
        H += J.T @ M @ J
        b += J.T @ M @ e
        err += e @ M @ e
    return H, b, err, np.where(valid, j, -1)


def fitness(src, tgt, T, max_range):
    q = np.asarray(src, np.float64) @ T[:3, :3].T + T[:3, 3]
    d, _ = cKDTree(np.asarray(tgt, np.float64)).query(q)
    d2 = d * d
    m = d2 <= max_range
    return (d2[m].mean() if m.any() else np.finfo(np.float64).max), int(m.sum())


def ndt_cells(xyz, res, min_points=6):
    """dict (i,j,k) -> (n, mean, icov) following VoxelGridCovariance::applyFilter."""
    xyz32 = np.asarray(xyz, np.float32)
    inv = np.float32(1.0) / np.float32(res)
    ijk = np.floor(xyz32 * inv).astype(np.int64)
    cells = {}
    x64 = xyz32.astype(np.float64)
    order = np.lexsort((ijk[:, 0], ijk[:, 1], ijk[:, 2]))
    ijk_s = ijk[order]
    brk = np.flatnonzero(np.concatenate([[True], np.any(ijk_s[1:] != ijk_s[:-1], axis=1)]))
    ends = np.concatenate([brk[1:], [len(order)]])
    for s, e in zip(brk, ends):
        n = e - s
        if n < min_points:
            continue
        p = x64[order[s:e]]
        mean = p.sum(0) / n
        cov = (p.T @ p - 2 * np.outer(p.sum(0), mean)) / n + np.outer(mean, mean)
        cov *= (n - 1.0) / n
        w, V = np.linalg.eigh((cov + cov.T) / 2)
        if w[0] < 0 or w[1] < 0 or w[2] <= 0:
            continue
        mn = 0.01 * w[2]
        if w[0] < mn:
            w = w.copy()
            w[0] = mn
            w[1] = max(w[1], mn)
            cov = V @ np.diag(w) @ np.linalg.inv(V)
        cells[tuple(ijk_s[s])] = (n, mean, np.linalg.inv(cov))
    return cells


def ndt_pose(p):
    cx, sx, cy, sy, cz, sz = np.cos(p[3]), np.sin(p[3]), np.cos(p[4]), np.sin(p[4]), np.cos(p[5]), np.sin(p[5])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rx @ Ry @ Rz
    T[:3, 3] = p[:3]
    return T


def ndt_gauss(res, outlier_ratio=0.55):
    c1 = 10 * (1 - outlier_ratio)
    c2 = outlier_ratio / res**3
    d3 = -np.log(c2)
    d1 = -np.log(c1 + c2) - d3
    d2 = -2 * np.log((-np.log(c1 * np.exp(-0.5) + c2) - d3) / d1)
    return d1, d2


def ndt_score(src, cells, p, res, direct7=True, freeze_p=None):
    """Score only (float64), for finite-difference checks of the oracle's analytic gradient/Hessian."""
    d1, d2 = ndt_gauss(res)
    # cell assignment optionally frozen at pose freeze_p (the score is discontinuous where points change cells)
    T = ndt_pose(p if freeze_p is None else freeze_p).astype(np.float32)
    src32 = np.asarray(src, np.float32)
    q = (src32 @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    inv = np.float32(1.0) / np.float32(res)
    ijk = np.floor(q * inv).astype(np.int64)
    offs = [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)] if direct7 else [(0, 0, 0)]
    # exact (float64) transform for the smooth part so FD is not dominated by float32 rounding
    q64 = np.asarray(src, np.float64) @ ndt_pose(p)[:3, :3].T + np.asarray(p[:3], np.float64)
    score = 0.0
    for i in range(len(q)):
        for o in offs:
            c = cells.get((ijk[i, 0] + o[0], ijk[i, 1] + o[1], ijk[i, 2] + o[2]))
            if c is None:
                continue
            x = q64[i] - c[1]
            e = np.exp(-d2 / 2 * x @ c[2] @ x)
            if not (0 <= d2 * e <= 1):
                continue
            score += -d1 * e
    return score


def _rot_parts(r, snap=True):
    """Rx, Ry, Rz and their first / second derivatives with respect to their own angle.  snap: ndt_omp's
    computeAngleDerivatives treats |angle| < 10e-5 as exactly 0 (cos 1, sin 0)."""
    out = []
    for axis, a in enumerate(r):
        if snap and abs(a) < 10e-5:
            c, s = 1.0, 0.0
        else:
            c, s = np.cos(a), np.sin(a)
        i, j = [(1, 2), (2, 0), (0, 1)][axis]
        R, dR, ddR = np.eye(3), np.zeros((3, 3)), np.zeros((3, 3))
        R[i, i], R[i, j], R[j, i], R[j, j] = c, -s, s, c
        dR[i, i], dR[i, j], dR[j, i], dR[j, j] = -s, -c, c, -s
        ddR[i, i], ddR[i, j], ddR[j, i], ddR[j, j] = -c, s, -s, -c
        out.append((R, dR, ddR))
    return out


def ndt_derivatives(src, cells, p, res, direct7=True, upstream_hd1_sign=True, outlier_ratio=0.55):
    """Score, gradient (6) and Hessian (6x6) of ndt_omp's objective at p = (t, rx, ry, rz), T(p) = Trans(t) Rx Ry Rz — an
    INDEPENDENT float64 restatement of computeDerivatives / updateDerivatives (Magnusson 2009 eq. 6.12 / 6.13): the Jacobian and
    the second derivatives of T(p) x come from products of elementary rotation derivatives, not from ndt_omp's j_ang / h_ang
    tables (which is what oracle/ndt.hpp and the device restate), so a transcription error in those tables shows up here.
    upstream_hd1_sign reproduces the one known quirk of those tables: the z entry of h_ang_d1 is +sin(ry) instead of -sin(ry).
    Cells are assigned from the float32 transformed point like ndt_omp does."""
    d1, d2 = ndt_gauss(res, outlier_ratio)
    p = np.asarray(p, np.float64)
    (Rx, dRx, ddRx), (Ry, dRy, ddRy), (Rz, dRz, ddRz) = _rot_parts(p[3:6], snap=True)   # derivative tables: snapped angles
    R = ndt_pose(p)[:3, :3]                                                               # the transform itself: exact angles
    dR = [dRx @ Ry @ Rz, Rx @ dRy @ Rz, Rx @ Ry @ dRz]
    ddR = [[ddRx @ Ry @ Rz, dRx @ dRy @ Rz, dRx @ Ry @ dRz],
           [None, Rx @ ddRy @ Rz, Rx @ dRy @ dRz],
           [None, None, Rx @ Ry @ ddRz]]
    if upstream_hd1_sign:
        # d2(T x)_x / dry2 = (-cy cz, cy sz, -sy) . x ; ndt_omp's table carries +sy in the last slot
        sy = 0.0 if abs(p[4]) < 10e-5 else np.sin(p[4])
        ddR[1][1] = ddR[1][1].copy()
        ddR[1][1][0, 2] += 2.0 * sy
    for a in range(3):
        for b in range(a):
            ddR[a][b] = ddR[b][a]
    T32 = np.eye(4, dtype=np.float32)
    T32[:3, :3], T32[:3, 3] = R.astype(np.float32), p[:3].astype(np.float32)
    src32 = np.asarray(src, np.float32)
    # the float transform as a chain of fused multiply-adds, fmaf(T2, z, fmaf(T1, y, fmaf(T0, x, T3))) (a float product is exact
    # in double, so each fma is one double multiply-add rounded to float): points near a cell face must land in the same cell
    t64, x64 = T32.astype(np.float64), src32.astype(np.float64)
    q32 = np.empty((len(src32), 3), np.float32)
    for r_ in range(3):
        acc = (t64[r_, 0] * x64[:, 0] + t64[r_, 3]).astype(np.float32).astype(np.float64)
        acc = (t64[r_, 1] * x64[:, 1] + acc).astype(np.float32).astype(np.float64)
        q32[:, r_] = (t64[r_, 2] * x64[:, 2] + acc).astype(np.float32)
    ijk = np.floor(q32 * (np.float32(1.0) / np.float32(res))).astype(np.int64)
    offs = [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)] if direct7 else [(0, 0, 0)]
    xs = np.asarray(src, np.float64)
    xt = q32.astype(np.float64)   # ndt_omp evaluates the residual on the FLOAT transformed cloud (4e-6 m at 50 m: 1e-3 of a score term)
    score, g, H = 0.0, np.zeros(6), np.zeros((6, 6))
    for i in range(len(xs)):
        x = xs[i]
        J = np.zeros((3, 6))
        J[:, :3] = np.eye(3)
        for a in range(3):
            J[:, 3 + a] = dR[a] @ x
        for o in offs:
            c = cells.get((ijk[i, 0] + o[0], ijk[i, 1] + o[1], ijk[i, 2] + o[2]))
            if c is None:
                continue
            q = xt[i] - c[1]
            S = c[2]
            e = np.exp(-d2 / 2 * q @ S @ q)
            if not (0 <= d2 * e <= 1):
                continue
            score += -d1 * e
            w = d1 * d2 * e
            qSJ = q @ S @ J
            g += w * qSJ
            Hp = -d2 * np.outer(qSJ, qSJ) + J.T @ S @ J
            for a in range(3):
                for b in range(3):
                    Hp[3 + a, 3 + b] += q @ S @ (ddR[a][b] @ x)
            H += w * Hp
    return score, g, H


def ndt_newton_step(g, H, step_size=0.1, trans_eps=0.01):
    """One iteration of ndt_omp's computeTransformation as it actually runs (the More-Thuente loop never executes): the
    Newton direction by SVD (pseudo-inverse), normalised, flipped if it is not a descent direction of -score, with the step
    length min(max(|dp|, eps / 2), step_size).  Returns (unit direction, step length)."""
    dp = np.linalg.pinv(H, rcond=1e-15) @ (-g)
    n = np.linalg.norm(dp)
    if n == 0 or not np.isfinite(n):
        return dp, 0.0
    dp = dp / n
    d_phi_0 = -(g @ dp)
    if d_phi_0 >= 0:
        if d_phi_0 == 0:
            return dp, 0.0
        dp = -dp
    return dp, max(min(n, step_size), trans_eps / 2)


def vgicp_linearize(src, tgt, cov_s, cov_t, T, resolution, offsets=((0, 0, 0),)):
    """fast_gicp::FastVGICP (ADDITIVE voxels): voxel = {n, mean of the points, mean of their
    covariances}, key floor(p / res - 0.5); every source point is matched against the voxel(s) of T a_i with weight
    sqrt(n).  Returns H, b, error and the number of voxel correspondences per source point."""
    src, tgt = np.asarray(src, np.float64), np.asarray(tgt, np.float64)
    key_t = np.floor(tgt / resolution - 0.5).astype(np.int64)
    vox = {}
    for k, p, c in zip(map(tuple, key_t), tgt, cov_t):
        v = vox.setdefault(k, [0, np.zeros(3), np.zeros((3, 3))])
        v[0] += 1
        v[1] += p
        v[2] += c
    R, t = T[:3, :3], T[:3, 3]
    q = src @ R.T + t
    key_s = np.floor(q / resolution - 0.5).astype(np.int64)
    H, b, err, hits = np.zeros((6, 6)), np.zeros(6), 0.0, np.zeros(len(src), np.int32)
    for i in range(len(src)):
        for o in offsets:
            v = vox.get((key_s[i, 0] + o[0], key_s[i, 1] + o[1], key_s[i, 2] + o[2]))
            if v is None:
                continue
            hits[i] += 1
            n, mean, cov = v[0], v[1] / v[0], v[2] / v[0]
            M = np.sqrt(n) * np.linalg.inv(cov + R @ cov_s[i] @ R.T)
            e = mean - q[i]
            J = np.hstack([skew(q[i]), -np.eye(3)])
            H += J.T @ M @ J
            b += J.T @ M @ e
            err += e @ M @ e
    return H, b, err, hits


# ---- LsqRegistration::computeTransformation / step_lm / is_converged, the GICP control path -----------
def gicp_error_frozen(src, tgt, T, corr, maha):
    """compute_error: the correspondences and Mahalanobis matrices of the last linearisation, residuals at T."""
    q = np.asarray(src, np.float64) @ T[:3, :3].T + T[:3, 3]
    v = corr >= 0
    e = np.asarray(tgt, np.float64)[corr[v]] - q[v]
    return float(np.einsum("ni,nij,nj->", e, maha[v], e))


def gicp_align(src, tgt, cov_s, cov_t, guess, max_corr=2.5, max_iterations=64, rot_eps=2e-3, trans_eps=0.01, lm_max_iterations=10,
               lm_init_lambda_factor=1e-9):
    """The whole LM run written from the text of Appendix A.2 (not from oracle/gicp.hpp): returns (final T, outer iterations,
    converged, total LM tries).  The correspondence search follows update_correspondences: the query is T a_i in FLOAT."""
    src, tgt = np.asarray(src, np.float64), np.asarray(tgt, np.float64)
    tree = cKDTree(tgt)

    def is_converged(delta):
        return max(np.abs(delta[:3, :3] - np.eye(3)).max() / rot_eps, np.abs(delta[:3, 3]).max() / trans_eps) < 1.0

    def linearize(T):
        R, t = T[:3, :3], T[:3, 3]
        Tf = T.astype(np.float32)
        q32 = (src.astype(np.float32) @ Tf[:3, :3].T + Tf[:3, 3]).astype(np.float64)   # the search point is a float cloud
        d, j = tree.query(q32)
        corr = np.where(d * d < max_corr**2, j, -1)
        v = corr >= 0
        maha = np.zeros((len(src), 3, 3))
        maha[v] = np.linalg.inv(cov_t[corr[v]] + R @ cov_s[v] @ R.T)
        q = src @ R.T + t
        H, b, err = np.zeros((6, 6)), np.zeros(6), 0.0
        for i in np.flatnonzero(v):
            e = tgt[corr[i]] - q[i]
            J = np.hstack([skew(q[i]), -np.eye(3)])
            H += J.T @ maha[i] @ J
            b += J.T @ maha[i] @ e
            err += e @ maha[i] @ e
        return err, H, b, corr, maha

    x0 = np.array(guess, np.float64)
    lam, converged, iterations, tries_total = -1.0, False, 0, 0
    while iterations < max_iterations and not converged:
        iterations += 1
        y0, H, b, corr, maha = linearize(x0)
        if lam < 0:
            lam = lm_init_lambda_factor * np.abs(np.diag(H)).max()
        nu, stepped, delta = 2.0, False, np.eye(4)
        for _ in range(lm_max_iterations):
            tries_total += 1
            d = np.linalg.solve(H + lam * np.eye(6), -b)
            delta = se3_exp(d)
            xi = delta @ x0
            yi = gicp_error_frozen(src, tgt, xi, corr, maha)
            rho = (y0 - yi) / (d @ (lam * d - b))
            if rho < 0:
                if is_converged(delta):
                    stepped = True
                    break
                lam, nu = nu * lam, 2 * nu
                continue
            x0 = xi
            lam = lam * max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3)
            stepped = True
            break
        if not stepped:          # "lm not converged!!": the outer loop breaks
            break
        converged = is_converged(delta)
    return x0, iterations, converged, tries_total
