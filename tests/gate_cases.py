"""Correspondences planted around every decision of the linearisation (camera.h:28-35 depth and image gates,
picp_solver.cpp:78 the chi^2 test) and the float32, reference-order values those decisions are taken on.

ref_values() restates the reference's operation order in numpy float32 (every numpy float32 operation rounds once, like
the reference's SSE2 build): pc = t + (R0 x + (R1 y + R2 z)) (Eigen's 3-term inner product order, vo_math.h: dot3),
ph = K pc likewise, iz = 1/ph.z, (u, v) = ph.xy * iz, chi = e0 e0 + e1 e1.  plant() builds, for each of the seven gates,
thousands of correspondences whose deciding value lies within a fraction of an ulp up to ~1000 ulp of the gate, on both
sides; the exact distance of each (in ulps of the gate, or of the magnitudes that cancel where the gate is 0) is MEASURED
from ref_values, not assumed."""
import numpy as np

F = np.float32
GATES = ("z_far", "z_near", "u_lo", "u_hi", "v_lo", "v_hi", "chi")


def ref_values(T, K, world, meas):
    """T: 4x4 float32 (world -> camera), K: 3x3 float32, world (N,3), meas (N,2) float32 -> dict of float32 arrays"""
    T = np.asarray(T, F); K = np.asarray(K, F); w = np.asarray(world, F); z = np.asarray(meas, F)
    R, t = T[:3, :3], T[:3, 3]
    pc = [t[i] + (R[i, 0] * w[:, 0] + (R[i, 1] * w[:, 1] + R[i, 2] * w[:, 2])) for i in range(3)]          # camera.h:27
    ph = [K[i, 0] * pc[0] + (K[i, 1] * pc[1] + K[i, 2] * pc[2]) for i in range(3)]                        # camera.h:30
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        iz = F(1.0) / ph[2]                                                                               # camera.h:31
        u, v = ph[0] * iz, ph[1] * iz
        e0, e1 = u - z[:, 0], v - z[:, 1]                                                                 # picp_solver.cpp:35
        chi = e0 * e0 + e1 * e1                                                                           # :75
    return dict(pc2=pc[2], u=u, v=v, chi=chi)


def classify(vals, rows, cols, z_near, z_far, thr):
    """0 = skipped before any statistic (gated out), 1 = inlier, 2 = outlier -- camera.h:28,32-35, picp_solver.cpp:72-88"""
    with np.errstate(invalid="ignore"):
        z_out = (vals["pc2"] > F(z_far)) | (vals["pc2"] < F(z_near))
        img_out = (vals["u"] < F(0)) | (vals["u"] > F(cols - 1)) | (vals["v"] < F(0)) | (vals["v"] > F(rows - 1))
        cls = np.where(vals["chi"] > F(thr), 2, 1)
    cls[z_out | img_out] = 0
    return cls.astype(np.int32)


def _ulp(x):
    x = F(abs(x))
    return float(np.spacing(x)) if x > 0 else float(np.spacing(F(1.0)))


def plant(n_per_gate=7000, seed=0, rows=480, cols=640, z_near=1, z_far=10, thr=100.0, K=None, max_angle=0.25, max_t=0.4):
    """-> dict(world, meas, T, K, gate (index into GATES per correspondence), dist (float64: signed distance of the deciding
    value from its gate in UNITS), unit (the unit per gate), cls (reference decisions), cam = (rows, cols, z_near, z_far), thr)"""
    rng = np.random.default_rng(seed)
    if K is None:
        K = np.array([[180.0, 0.0, 320.0], [0.0, 180.0, 240.0], [0.0, 0.0, 1.0]], F)
    K64 = np.asarray(K, np.float64)
    # a general small motion (double, orthonormal to rounding), stored as float32
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    ang = rng.uniform(0.5, 1.0) * max_angle
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = rng.uniform(-max_t, max_t, 3)
    T = T.astype(F)
    T64 = T.astype(np.float64)
    Ri, ti = np.linalg.inv(T64[:3, :3]), T64[:3, 3]
    Ki = np.linalg.inv(K64)
    cm1, rm1 = cols - 1, rows - 1
    # offsets from a fraction of a unit to ~1000 units, both signs, log-uniform; a tenth of the points exactly "on" (offset 0)
    def offsets(n):
        mag = 2.0 ** rng.uniform(-2, 10, n)
        mag[rng.random(n) < 0.1] = 0.0
        return mag * rng.choice([-1.0, 1.0], n)
    units = dict(z_far=_ulp(z_far), z_near=_ulp(z_near if z_near != 0 else 1.0), u_lo=_ulp(K64[0, 2]), u_hi=_ulp(cm1),
                 v_lo=_ulp(K64[1, 2]), v_hi=_ulp(rm1), chi=_ulp(thr))
    world, meas, gate = [], [], []
    for g, name in enumerate(GATES):
        n = n_per_gate
        off = offsets(n)
        z = rng.uniform(2.0, 8.0, n)
        u = rng.uniform(60.0, cm1 - 60.0, n); v = rng.uniform(60.0, rm1 - 60.0, n)
        mu, mv = u + rng.normal(0, 0.5, n), v + rng.normal(0, 0.5, n)          # measurements: half a pixel off, far below thr
        if name == "z_far": z = z_far + off * units[name]
        elif name == "z_near": z = z_near + off * units[name]
        elif name == "u_lo": u = 0.0 + off * units[name]; mu = u + rng.normal(0, 0.5, n)
        elif name == "u_hi": u = cm1 + off * units[name]; mu = u + rng.normal(0, 0.5, n)
        elif name == "v_lo": v = 0.0 + off * units[name]; mv = v + rng.normal(0, 0.5, n)
        elif name == "v_hi": v = rm1 + off * units[name]; mv = v + rng.normal(0, 0.5, n)
        else:
            r = np.sqrt(thr + off * units[name]); phi = rng.uniform(0, 2 * np.pi, n)
            mu, mv = u - r * np.cos(phi), v - r * np.sin(phi)
        pc = (Ki @ np.stack([u * z, v * z, z])).T                               # camera-frame points that project to (u, v) at depth z
        w = (pc - ti) @ Ri.T
        world.append(w.astype(F)); meas.append(np.stack([mu, mv], 1).astype(F)); gate.append(np.full(n, g, np.int32))
    world = np.concatenate(world); meas = np.concatenate(meas); gate = np.concatenate(gate)
    vals = ref_values(T, K, world, meas)
    gv = dict(z_far=(vals["pc2"], z_far), z_near=(vals["pc2"], z_near), u_lo=(vals["u"], 0.0), u_hi=(vals["u"], cm1),
              v_lo=(vals["v"], 0.0), v_hi=(vals["v"], rm1), chi=(vals["chi"], thr))
    dist = np.zeros(len(world))
    for g, name in enumerate(GATES):
        m = gate == g
        dist[m] = (gv[name][0][m].astype(np.float64) - float(F(gv[name][1]))) / units[name]
    return dict(world=world, meas=meas, T=T, K=np.asarray(K, F), gate=gate, dist=dist, unit=units,
                cls=classify(vals, rows, cols, z_near, z_far, thr), cam=(rows, cols, z_near, z_far), thr=float(thr), vals=vals)
