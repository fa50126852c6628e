"""Mixed models (tests, bench legs): systems whose nonlinear sub-DAGs come in SEVERAL shapes, written once against an
expression module passed as `m` (heyoka_amd itself in the product's tests and bench legs; the tests pass the checker's
expression module as well: both expose var / make_vars, the arithmetic operators and the elementary functions of
include/heyoka/math.hpp). Workload definitions like heyoka_amd/configs.py - nothing here is imported by the library."""
import numpy as np


def _pow(m, b, e):
    return (m.pow if hasattr(m, "pow") else m.pow_)(b, e)


def _vars(m, names):
    if hasattr(m, "make_vars"):
        return list(m.make_vars(*names))
    return [m.var(n) for n in names]


def nbody_j2(m, n_bodies, masses, G, J2R2):
    """Point masses (all pairs, like model::nbody) plus the oblateness (J2) of body 0 acting on every other body:
    a_i += -(3/2) J2 R^2 G m_0 r^-5 (x (1 - 5 z^2 / r^2), y (1 - 5 z^2 / r^2), z (3 - 5 z^2 / r^2)), r = r_i - r_0, with its
    reaction on body 0. The pair clusters of the point masses are one class of clusters, the oblateness terms another."""
    names = []
    for i in range(n_bodies):
        names += ["x_%d" % i, "y_%d" % i, "z_%d" % i, "vx_%d" % i, "vy_%d" % i, "vz_%d" % i]
    v = _vars(m, names)
    X = [v[6 * i:6 * i + 3] for i in range(n_bodies)]
    V = [v[6 * i + 3:6 * i + 6] for i in range(n_bodies)]
    acc = [[0.0, 0.0, 0.0] for _ in range(n_bodies)]
    for i in range(n_bodies):
        for j in range(i + 1, n_bodies):
            d = [X[j][c] - X[i][c] for c in range(3)]
            r3 = _pow(m, d[0] * d[0] + d[1] * d[1] + d[2] * d[2], -1.5)
            for c in range(3):
                acc[i][c] = acc[i][c] + (G * masses[j]) * (d[c] * r3)
                acc[j][c] = acc[j][c] - (G * masses[i]) * (d[c] * r3)
    for i in range(1, n_bodies):
        d = [X[i][c] - X[0][c] for c in range(3)]
        r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
        r5 = _pow(m, r2, -2.5)
        q = (d[2] * d[2]) * _pow(m, r2, -1.0)
        f = [d[0] * (1.0 - 5.0 * q), d[1] * (1.0 - 5.0 * q), d[2] * (3.0 - 5.0 * q)]
        for c in range(3):
            t = (1.5 * J2R2 * G) * (f[c] * r5)
            acc[i][c] = acc[i][c] - masses[0] * t
            acc[0][c] = acc[0][c] + masses[i] * t
    sys_ = []
    for i in range(n_bodies):
        for c in range(3):
            sys_.append((X[i][c], V[i][c]))
        for c in range(3):
            sys_.append((V[i][c], acc[i][c]))
    return sys_


def sine_lattice(m, n_sites, k=0.7, beta=0.4, gamma=0.01):
    """A chain of pendula coupled by springs with a cubic term (Frenkel-Kontorova / FPU-beta) and a linear drag:
    th_i'' = -sin th_i + sum over the bonds of i of +-(k d + beta d^3) - gamma om_i, d = th_{i+1} - th_i. The sin / cos pairs
    of the sites (their argument is a state variable) are one class of clusters, the cubes of the bonds another."""
    names = ["th_%d" % i for i in range(n_sites)] + ["om_%d" % i for i in range(n_sites)]
    v = _vars(m, names)
    th, om = v[:n_sites], v[n_sites:]
    force = [-1.0 * m.sin(th[i]) - gamma * om[i] for i in range(n_sites)]
    for i in range(n_sites - 1):
        d = th[i + 1] - th[i]
        # (The cube as products: the recurrence of pow(d, 3) divides by d^[0], and the bonds swing through d = 0.)
        f = k * d + beta * ((d * d) * d)
        force[i] = force[i] + f
        force[i + 1] = force[i + 1] - f
    return [(th[i], om[i]) for i in range(n_sites)] + [(om[i], force[i]) for i in range(n_sites)]


def sine_lattice_state(n_sites, n, seed=5):
    """Angles alternating around +-0.6 rad (+- 0.2 of scatter), small angular velocities: every bond is stretched by 0.8 ... 1.6
    rad."""
    rng = np.random.RandomState(seed)
    th = 0.6 * np.where(np.arange(n_sites) % 2 == 0, 1.0, -1.0)[:, None] + rng.uniform(-0.2, 0.2, (n_sites, n))
    return np.ascontiguousarray(np.concatenate([th, rng.uniform(-0.3, 0.3, (n_sites, n))]))


def lattice_centres(m, centres, charges, k_lat=2.0, v_lat=0.3):
    """A particle among fixed attracting centres (model::fixed_centres) in a periodic ("optical lattice") potential
    V = v_lat (cos k x + cos k y + cos k z): a = sum_j q_j (c_j - r) |c_j - r|^-3 + v_lat k (sin k x, sin k y, sin k z). The
    distance clusters of the centres are one class, the sin / cos pairs of the three coordinates (functions of state
    variables) another."""
    x, y, z, vx, vy, vz = _vars(m, ["x", "y", "z", "vx", "vy", "vz"])
    r = [x, y, z]
    acc = [(v_lat * k_lat) * m.sin(k_lat * r[c]) for c in range(3)]
    for cj, qj in zip(centres, charges):
        d = [cj[c] - r[c] for c in range(3)]
        r3 = _pow(m, d[0] * d[0] + d[1] * d[1] + d[2] * d[2], -1.5)
        for c in range(3):
            acc[c] = acc[c] + qj * (d[c] * r3)
    return [(x, vx), (y, vy), (z, vz), (vx, acc[0]), (vy, acc[1]), (vz, acc[2])]


def lattice_centres_setup(n_centres=12, seed=3):
    rng = np.random.RandomState(seed)
    centres = rng.uniform(-4.0, 4.0, (n_centres, 3))
    centres[np.linalg.norm(centres, axis=1) < 1.5] *= 2.5  # (nothing close to the particle's orbit around the origin)
    charges = rng.uniform(0.2, 1.0, n_centres)
    return centres.tolist(), charges.tolist()


def lattice_centres_state(n, seed=9):
    rng = np.random.RandomState(seed)
    base = np.array([0.5, 0.1, -0.2, 0.0, 0.4, 0.1])[:, None]
    return np.ascontiguousarray(base + 0.05 * rng.uniform(-1, 1, (6, n)))
