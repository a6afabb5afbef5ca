"""Deterministic synthetic CRF inputs with the SLAM path's statistics.

TUM / Bonn sequences and ORBvoc.txt are not available offline, so BASELINE.json's
configs are restated as synthetic frames: uniform keypoints in a 640x480 image, one
to three "walking person" boxes holding ~20 % of the points, per-point observation
counts and reprojection errors drawn from the static / dynamic distributions, and an
initial label that is the truth flipped with p=0.15 (stands in for
Tracking::RroughClassify, /root/reference/src/Tracking.cc:1961-2013).

Pure numpy; used by tests/, bench.py and __graft_entry__.smoke().  The feature
assembly mirrors PottsPotential3D::appearanceKernel / smoothKernel
(/root/reference/Thirdparty/DenseCRF/include/pairwise3d.h:37-71): a float32 division
by the kernel's standard deviation.
"""
import numpy as np

# Examples/RGB-D/TUM3.yaml:78-101 (CRF block; BONN.yaml is identical)
TUM3 = dict(w1=10.0, w2=30.0, u_alpha=1.7, stdev_alpha=0.6, u_beta=5.4, stdev_beta=1.5,
            u_gamma=0.3, stdev_gamma=0.2, point3d_stdev=0.5, point2d_stdev=18.0,
            u_depth=2.75, pth=0.8, confidence=0.7)

IMG_W, IMG_H = 640, 480


def slam_frame(N, seed, obs_cap=None, dyn_frac=0.2, flip=0.15):
    """One frame's CRF inputs.  Label 0 = dynamic (moving), 1 = static."""
    rng = np.random.default_rng([int(seed), int(N)])
    uv = (rng.random((N, 2)) * np.array([IMG_W, IMG_H])).astype(np.float32)
    n_box = int(rng.integers(1, 4))
    dyn = np.zeros(N, bool)
    for _ in range(n_box):
        area = dyn_frac / n_box * IMG_W * IMG_H
        aspect = rng.uniform(0.4, 1.2)               # w/h of a standing person-ish blob
        bw = min(np.sqrt(area * aspect), IMG_W)
        bh = min(area / bw, IMG_H)
        x0 = rng.uniform(0, IMG_W - bw)
        y0 = rng.uniform(0, IMG_H - bh)
        dyn |= (uv[:, 0] >= x0) & (uv[:, 0] < x0 + bw) & (uv[:, 1] >= y0) & (uv[:, 1] < y0 + bh)
    err = np.where(dyn, np.abs(rng.normal(4.5, 1.5, N)), np.abs(rng.normal(1.7, 0.6, N)))
    obs = 1 + np.where(dyn, rng.poisson(1.0, N), rng.poisson(5.0, N))
    if obs_cap is not None:                          # config C3: "10-KF window"
        obs = np.minimum(obs, obs_cap)
    depth = rng.uniform(0.5, 6.0, N)
    truth = np.where(dyn, 0, 1).astype(np.int16)
    flips = rng.random(N) < flip
    init = np.where(flips, 1 - truth, truth).astype(np.int16)
    return dict(N=N, uv=uv, obs=obs.astype(np.float32), err=err.astype(np.float32),
                depth=depth.astype(np.float32), truth=truth, init_label=init)


def appearance_features(frame, p=TUM3):
    """pairwise3d.h:37-48: (n_obs / stdev_beta, reproj_err / stdev_alpha)."""
    f = np.empty((frame["N"], 2), np.float32)
    f[:, 0] = frame["obs"] / np.float32(p["stdev_beta"])
    f[:, 1] = frame["err"] / np.float32(p["stdev_alpha"])
    return f


def smooth_features(frame, p=TUM3):
    """pairwise3d.h:51-71 (2-D branch): (u, v) / point2d_stdev."""
    return (frame["uv"] / np.float32(p["point2d_stdev"])).astype(np.float32)


def slam_problem(N, seed, obs_cap=None, p=TUM3):
    """Everything the call site at Tracking.cc:1919-1930 hands to the CRF."""
    fr = slam_frame(N, seed, obs_cap=obs_cap)
    return dict(N=N, L=2, label=fr["init_label"], conf=np.float32(p["confidence"]),
                kernels=[(appearance_features(fr, p), np.float32(p["w1"])),
                         (smooth_features(fr, p), np.float32(p["w2"]))],
                truth=fr["truth"], frame=fr)


def bilateral_problem(N, seed, conf=0.7, w=10.0):
    """Config C5: one 6-D kernel (x/60, y/60, depth/0.5, r/20, g/20, b/20), L=2."""
    rng = np.random.default_rng([int(seed), int(N), 6])
    x = rng.uniform(0, IMG_W, N)
    y = rng.uniform(0, IMG_H, N)
    depth = rng.uniform(0.5, 6.0, N)
    rgb = rng.uniform(0, 255, (N, 3))
    f = np.empty((N, 6), np.float32)
    f[:, 0] = x.astype(np.float32) / np.float32(60)
    f[:, 1] = y.astype(np.float32) / np.float32(60)
    f[:, 2] = depth.astype(np.float32) / np.float32(0.5)
    f[:, 3:] = rgb.astype(np.float32) / np.float32(20)
    # two-blob rule for the truth, flipped with p=0.15 for the initial label
    c = np.array([[200.0, 240.0], [460.0, 200.0]])
    r2 = ((x[:, None] - c[None, :, 0]) ** 2 + (y[:, None] - c[None, :, 1]) ** 2).min(1)
    truth = np.where(r2 < 110.0 ** 2, 0, 1).astype(np.int16)
    init = np.where(rng.random(N) < 0.15, 1 - truth, truth).astype(np.int16)
    return dict(N=N, L=2, label=init, conf=np.float32(conf),
                kernels=[(f, np.float32(w))], truth=truth)


def generic_problem(N, d_list, L, seed, spread=4.0, lattice_ties=False):
    """Generic-template cases: arbitrary d / L, negative features,
    optionally points sitting exactly on lattice-cell boundaries (ties for the
    rounding and rank compares)."""
    rng = np.random.default_rng([int(seed), int(N), int(L)] + [int(d) for d in d_list])
    kernels = []
    for d in d_list:
        f = rng.normal(0.0, spread, (N, d)).astype(np.float32)
        if lattice_ties:
            q = rng.random(N) < 0.5
            f[q] = np.round(f[q] * 2) / 2          # many exact .0 / .5 coordinates
            f[rng.random(N) < 0.1] = 0.0
        kernels.append((f, np.float32(rng.uniform(1.0, 12.0))))
    unary = rng.uniform(0.05, 3.0, (N, L)).astype(np.float32)
    return dict(N=N, L=L, unary=unary, kernels=kernels)
