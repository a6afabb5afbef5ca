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


def image_features(W, H, posdev, rgb=None, featuredev=0.0):
    """PairwisePotential's image features (pairwise_cpu.h:33-51, FromImage / examples/example_cpu.cpp:86-91): (x, y) / posdev and,
    with an image, its channels / featuredev -- float divisions of exact small integers, so numpy gives the reference's bits."""
    y, x = np.mgrid[0:H, 0:W].astype(np.float32)
    cols = [x.ravel() / np.float32(posdev), y.ravel() / np.float32(posdev)]
    if rgb is not None:
        c = np.ascontiguousarray(rgb, np.uint8).reshape(W * H, -1).astype(np.float32)
        cols += [c[:, i] / np.float32(featuredev) for i in range(c.shape[1])]
    return np.ascontiguousarray(np.stack(cols, 1), np.float32)


def map_point_scene(n_points, n_kf, seed, max_obs=12):
    """A synthetic local map for the unary builder (Tracking.cc:1803-1839): keyframes on a short arc
    looking roughly down +z, map points in front of them, each point observed by a random subset of
    keyframes at its projection plus pixel noise (static points: ~1 px, dynamic ones: several px).
    Some observations fall behind the camera or outside the image, some points have none."""
    rng = np.random.default_rng([int(seed), int(n_points), int(n_kf)])
    fx = fy = np.float32(535.4)
    cx, cy = np.float32(320.1), np.float32(247.6)
    poses = np.zeros((n_kf, 3, 4), np.float32)
    for k in range(n_kf):
        a = rng.normal(0, 0.08)
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        b = rng.normal(0, 0.05)
        Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
        poses[k, :, :3] = (R @ Rx).astype(np.float32)
        poses[k, :, 3] = rng.normal(0, 0.3, 3).astype(np.float32)
    if n_kf:
        poses[n_kf // 2, 2, 2] *= -1            # one keyframe that sees most points behind it
    intr = np.tile(np.array([fx, fy, cx, cy], np.float32), (n_kf, 1))
    bounds = np.tile(np.array([0, 640, 0, 480], np.float32), (n_kf, 1))
    Xw = np.stack([rng.uniform(-2.5, 2.5, n_points), rng.uniform(-1.8, 1.8, n_points),
                   rng.uniform(0.6, 6.0, n_points)], 1).astype(np.float32)
    dyn = rng.random(n_points) < 0.2
    counts = np.where(dyn, rng.poisson(1.0, n_points), 1 + rng.poisson(5.0, n_points))
    counts = np.minimum(counts, min(max_obs, n_kf)).astype(np.int32)
    obs_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    obs_kf = np.empty(obs_ptr[-1], np.int32)
    obs_kp = np.empty((obs_ptr[-1], 2), np.float64)
    for i in range(n_points):
        ks = np.sort(rng.choice(n_kf, counts[i], replace=False)) if counts[i] else np.zeros(0, np.int64)
        obs_kf[obs_ptr[i]:obs_ptr[i + 1]] = ks
        for j, k in enumerate(ks):
            xc = poses[k, :, :3].astype(np.float64) @ Xw[i].astype(np.float64) + poses[k, :, 3]
            z = xc[2] if abs(xc[2]) > 1e-6 else 1e-6
            u, v = fx * xc[0] / z + cx, fy * xc[1] / z + cy
            noise = rng.normal(0, 4.5 if dyn[i] else 1.2, 2)
            obs_kp[obs_ptr[i] + j] = (u + noise[0], v + noise[1])
    return dict(Xw=Xw, obs_ptr=obs_ptr, obs_kf=obs_kf, obs_kp=obs_kp, kf_pose=poses.reshape(n_kf, 12),
                kf_intr=intr, kf_bounds=bounds, dynamic=dyn)


def pose_scene(n, seed, noise=0.7, outlier_frac=0.15, mono_frac=0.2, n_invalid=0):
    """A synthetic frame for Optimizer::PoseOptimization (src/Optimizer.cc:239-450): map points in front of an RGB-D
    camera (TUM fr3 intrinsics, bf = 40), keypoints at their projections under a true pose plus pixel noise, a fraction
    of gross outliers (moving points the CRF missed), a fraction of monocular observations (no depth: u_right < 0),
    octave-dependent information (1.2^-2k), and an initial pose a few centimetres / a degree off."""
    rng = np.random.default_rng([int(seed), int(n), 77])
    fx = fy = 535.4
    cx, cy, bf = 320.1, 247.6, 40.0
    a, b = 0.05, 0.03
    Ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    R, t = Ry @ Rx, np.array([0.12, -0.05, 0.08])
    Xw = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(1.0, 6.0, n)], 1).astype(np.float32)
    pc = Xw.astype(np.float64) @ R.T + t
    u, v = fx * pc[:, 0] / pc[:, 2] + cx, fy * pc[:, 1] / pc[:, 2] + cy
    ur = u - bf / pc[:, 2]
    kp = np.stack([u, v], 1) + rng.normal(0, noise, (n, 2)) if noise else np.stack([u, v], 1)
    ur = ur + (rng.normal(0, noise, n) if noise else 0.0)
    bad = rng.random(n) < outlier_frac
    kp[bad] += rng.normal(0, 25, (int(bad.sum()), 2))
    mono = rng.random(n) < mono_frac
    ur = np.where(mono, -1.0, ur)
    octave = rng.integers(0, 4, n)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    a0 = a + 0.02
    T0 = np.eye(4)
    T0[:3, :3] = np.array([[np.cos(a0), 0, np.sin(a0)], [0, 1, 0], [-np.sin(a0), 0, np.cos(a0)]])
    T0[:3, 3] = t + np.array([0.03, -0.02, 0.04])
    valid = np.ones(n, np.uint8)
    if n_invalid:
        valid[rng.choice(n, n_invalid, replace=False)] = 0
    return dict(Xw=Xw, kp=kp.astype(np.float32), u_right=ur.astype(np.float32),
                inv_sigma2=(1.0 / 1.2 ** (2 * octave)).astype(np.float32), valid=valid,
                K4=np.array([fx, fy, cx, cy], np.float32), bf=np.float32(bf), T_true=T, T_init=T0.astype(np.float32), gross=bad)
