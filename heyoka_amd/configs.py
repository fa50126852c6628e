"""Synthetic inputs of the BASELINE.json configurations (deterministic, no data files needed).

Sources (reference tree, bluescarni/heyoka v7.12.0):
  outer Solar System ICs / masses / G ... benchmark/outer_ss_long_term_batch.cpp:60-92
  perturbation scheme .................... benchmark/outer_ss_long_term_batch.cpp:103-108 (std::mt19937 seed 42,
                                           x += |x| * (U(-1,1) * perturb), one draw per array element in array order)
  centre-of-mass shift ................... benchmark/outer_ss_long_term_batch.cpp:200-221
  two-body circular orbit ................ benchmark/two_body_step_batch.cpp:37-54
  forced damped pendulum (batch 4) ....... tutorial/batch_mode.cpp:41-48
All arrays use the batch layout array[row, lane].
"""

import numpy as np

OUTER_SS_MASSES = [1.00000597682, 1 / 1047.355, 1 / 3501.6, 1 / 22869.0, 1 / 19314.0, 7.4074074e-09]
OUTER_SS_G = 0.01720209895 * 0.01720209895 * 365 * 365

_OUTER_SS_IC = [
    # Sun
    -4.06428567034226e-3, -6.08813756435987e-3, -1.66162304225834e-6,
    +6.69048890636161e-6 * 365, -6.33922479583593e-6 * 365, -3.13202145590767e-9 * 365,
    # Jupiter
    +3.40546614227466e0, +3.62978190075864e0, +3.42386261766577e-2,
    -5.59797969310664e-3 * 365, +5.51815399480116e-3 * 365, -2.66711392865591e-6 * 365,
    # Saturn
    +6.60801554403466e0, +6.38084674585064e0, -1.36145963724542e-1,
    -4.17354020307064e-3 * 365, +3.99723751748116e-3 * 365, +1.67206320571441e-5 * 365,
    # Uranus
    +1.11636331405597e1, +1.60373479057256e1, +3.61783279369958e-1,
    -3.25884806151064e-3 * 365, +2.06438412905916e-3 * 365, -2.17699042180559e-5 * 365,
    # Neptune
    -3.01777243405203e1, +1.91155314998064e0, -1.53887595621042e-1,
    -2.17471785045538e-4 * 365, -3.11361111025884e-3 * 365, +3.58344705491441e-5 * 365,
    # Pluto
    -2.13858977531573e1, +3.20719104739886e1, +2.49245689556096e0,
    -1.76936577252484e-3 * 365, -2.06720938381724e-3 * 365, +6.58091931493844e-4 * 365,
]


def mt19937_uniform_m1_1(n, seed):
    """n draws of std::uniform_real_distribution<double>(-1, 1) driven by std::mt19937(seed), as
    libstdc++ computes them: generate_canonical<double, 53> consumes two 32-bit outputs a, b and
    returns (a + b * 2^32) / 2^64 (clamped below 1), mapped to [-1, 1)."""
    rs = np.random.RandomState(int(seed))  # legacy seeding == init_genrand(seed) == std::mt19937(seed)
    raw = rs._bit_generator.random_raw(2 * int(n)).astype(np.float64)
    x = (raw[0::2] + raw[1::2] * 4294967296.0) / 18446744073709551616.0
    x = np.minimum(x, np.nextafter(1.0, 0.0))
    return x * 2.0 - 1.0


def outer_ss_state(n_systems, perturb=1e-12, seed=42, com_shift=True):
    """(36, n_systems) initial conditions of the perturbed outer Solar System ensemble."""
    n = int(n_systems)
    st = np.repeat(np.array(_OUTER_SS_IC, dtype=np.float64)[:, None], n, axis=1)
    if perturb != 0:
        r = mt19937_uniform_m1_1(36 * n, seed).reshape(36, n)
        st += np.abs(st) * (r * perturb)
    if com_shift:
        m = np.array(OUTER_SS_MASSES)[:, None, None]
        s3 = st.reshape(6, 6, n)
        tot = np.sum(OUTER_SS_MASSES)
        com = (s3[:, 0:3, :] * m).sum(axis=0) / tot
        com_v = (s3[:, 3:6, :] * m).sum(axis=0) / tot
        s3[:, 0:3, :] -= com[None]
        s3[:, 3:6, :] -= com_v[None]
        st = s3.reshape(36, n)
    return np.ascontiguousarray(st)


def nbody_energy(state, masses, G):
    """Total energy per lane of an N-body state (n*6, lanes) (cf. model::nbody_energy,
    src/model/nbody.cpp:209-235)."""
    nb = len(masses)
    s = np.asarray(state, dtype=np.float64).reshape(nb, 6, -1)
    m = np.asarray(masses, dtype=np.float64)
    kin = 0.5 * (m[:, None] * (s[:, 3:6, :] ** 2).sum(axis=1)).sum(axis=0)
    pot = np.zeros(s.shape[2])
    for i in range(nb):
        for j in range(i + 1, nb):
            d = s[j, 0:3, :] - s[i, 0:3, :]
            pot -= G * m[i] * m[j] / np.sqrt((d * d).sum(axis=0))
    return kin + pot


def two_body_state(n_systems, perturb=0.0, seed=42):
    """(12, n_systems): masses {1, 0}, circular orbit r = 1, v = 1 (identical lanes by default, as
    in the reference benchmark; optional relative perturbation of the second body)."""
    n = int(n_systems)
    st = np.zeros((12, n))
    st[6, :] = 1.0
    st[10, :] = 1.0
    if perturb != 0:
        r = mt19937_uniform_m1_1(2 * n, seed).reshape(2, n)
        st[6, :] += r[0] * perturb
        st[10, :] += r[1] * perturb
    return st


def plummer_nbody_state(n_bodies, n_systems, seed=1234, soft=0.05, jitter=1e-9):
    """(6 * n_bodies, n_systems) initial conditions for model::nbody(n_bodies) with unit masses and
    G = 1: a seeded Plummer-like cloud (scale radius 1, velocities from the local escape speed),
    re-drawn until the minimum pair separation exceeds `soft`, copied to all lanes with a small
    per-lane relative jitter. The reference has no physical IC set for this configuration
    (benchmark/n_body_creation.cpp:48-49 uses iota)."""
    rng = np.random.RandomState(seed)
    nb = int(n_bodies)
    while True:
        u = rng.uniform(0.05, 0.95, nb)
        r = 1.0 / np.sqrt(u ** (-2.0 / 3.0) - 1.0)
        ct = rng.uniform(-1, 1, nb)
        ph = rng.uniform(0, 2 * np.pi, nb)
        stt = np.sqrt(1 - ct * ct)
        pos = np.stack([r * stt * np.cos(ph), r * stt * np.sin(ph), r * ct], axis=1)
        d = pos[:, None, :] - pos[None, :, :]
        dist = np.sqrt((d * d).sum(-1)) + np.eye(nb) * 1e9
        if dist.min() > soft:
            break
    vesc = np.sqrt(2.0 * nb) * (1.0 + r * r) ** (-0.25)
    q = rng.uniform(0.1, 0.6, nb)
    vmag = q * vesc
    ct = rng.uniform(-1, 1, nb)
    ph = rng.uniform(0, 2 * np.pi, nb)
    stt = np.sqrt(1 - ct * ct)
    vel = np.stack([vmag * stt * np.cos(ph), vmag * stt * np.sin(ph), vmag * ct], axis=1)
    pos -= pos.mean(axis=0)
    vel -= vel.mean(axis=0)
    one = np.concatenate([pos, vel], axis=1).reshape(-1)  # body-major: x y z vx vy vz
    st = np.repeat(one[:, None], int(n_systems), axis=1)
    if jitter != 0 and n_systems > 1:
        st *= 1.0 + jitter * rng.uniform(-1, 1, st.shape)
    return np.ascontiguousarray(st)


FORCED_PENDULUM = dict(
    x0=[0.01, 0.02, 0.03, 0.04],
    v0=[1.85, 1.86, 1.87, 1.88],
    alpha=[0.10, 0.11, 0.12, 0.13],
)
