"""The FDTD example's host-side set-up, restated in numpy with the C++ expression types followed
operation by operation (float32 unless the C++ expression promotes to double), so that a test can hand
the oracle / the HIP kernels exactly the inputs the unchanged example hands its transition function.

Follows (reference, read-only): examples/fdtd/src/Parameters.hpp:139-262 (derived quantities),
defines.hpp:36-42 (constants), material/Material.hpp:24-71 (coefficients),
material/CoefResolver.hpp:33-52 (cell of a ring), Kernel.hpp:61-78 (kernel constants),
fdtd.cpp:188-212 (grid initialisation), fdtd.cpp:224-247 (snapshot loop)."""
import math

import numpy as np

f32, f64 = np.float32, np.float64

C0 = f32(299792458.0)                      # defines.hpp:37
SQRT_2 = f32(1.4142135623730951)           # defines.hpp:40
PI = f32(3.1415926535897932384626433)      # defines.hpp:42
MU_0 = f32(f64(4.0) * f64(PI) * f64(1.0e-7))            # Material.hpp:33 (double expression -> float)
EPS_0 = f32(f64(1.0) / f64(f32(f32(C0 * C0) * MU_0)))   # Material.hpp:36


class Experiment:
    """Parameters.hpp: the JSON fields (stored as float) and the quantities derived from them."""

    def __init__(self, config):
        self.tau = f32(config["tau"])
        self.dx = f32(config["dx"])
        t = config["time"]
        self.t_cutoff_factor, self.t_detect_factor = f32(t["t_cutoff"]), f32(t["t_detect"])
        self.t_max_factor = f32(t["t_max"])
        self.t_snap_factor = f32(t["t_snap"]) if "t_snap" in t else None
        s = config["source"]
        self.frequency, self.t_0_factor = f32(s["frequency"]), f32(s["phase"])
        self.source_x, self.source_y, self.source_radius = f32(s["x"]), f32(s["y"]), f32(s["radius"])
        self.rings = [(f32(r["radius"]), f32(r["mu_r"]), f32(r["eps_r"]), f32(r["sigma"]))
                      for r in config["cavity_rings"]]

    # Parameters.hpp:218-262
    def dt(self):
        return f32(f64(f32(self.dx / f32(C0 * SQRT_2))) * f64(0.99))

    def n_timesteps(self):
        return int(math.ceil(f32(f32(self.t_max_factor * self.tau) / self.dt())))

    def n_snap_timesteps(self):
        if self.t_snap_factor is None:
            return None
        return int(math.ceil(f32(f32(self.t_snap_factor * self.tau) / self.dt())))

    def omega(self):
        return f32(f64(2.0) * f64(PI) * f64(self.frequency))

    def grid_width(self):
        outer = f32(0.0)
        for radius, *_ in self.rings:
            outer = f32(outer + radius)
        return int(math.ceil(f32(f32(f32(f32(2) * outer) / self.dx) + f32(2))))

    def source_r(self):
        return int(f32(f32(self.grid_width() // 2) + f32(self.source_y / self.dx)))

    def source_c(self):
        return int(f32(f32(self.grid_width() // 2) + f32(self.source_x / self.dx)))

    # Material.hpp:38-58 (all float: the integer literals convert to float)
    def ring_coefficients(self, ring):
        _radius, mu_r, eps_r, sigma = self.rings[ring]
        dx, dt, one, two = self.dx, self.dt(), f32(1), f32(2)
        sdt = f32(sigma * dt)
        ca = f32(f32(one - sdt) / f32(one + sdt))
        da = ca
        if np.isinf(eps_r):
            cb = f32(0.0)
        else:
            cb = f32(f32(dt / f32(f32(EPS_0 * eps_r) * dx)) / f32(one + f32(sdt / f32(f32(two * EPS_0) * eps_r))))
        if np.isinf(mu_r):
            db = f32(0.0)
        else:
            db = f32(f32(dt / f32(f32(MU_0 * mu_r) * dx)) / f32(one + f32(sdt / f32(f32(two * MU_0) * mu_r))))
        return ca, cb, da, db

    # Kernel.hpp:61-78
    def kernel_constants(self):
        dt = self.dt()
        ratio = f32(self.source_radius / self.dx)
        source_r, source_c = f32(self.source_r()), f32(self.source_c())
        bound = f32(ratio * ratio)
        bound = f32(bound - f32(f32(source_c * source_c) + f32(source_r * source_r)))
        return dict(
            dt=dt, t_0=f32(self.t_0_factor * self.tau), tau=self.tau, omega=self.omega(),
            cutoff_iteration=int(math.floor(f32(f32(self.t_cutoff_factor * self.tau) / dt))),
            detect_iteration=int(math.floor(f32(f32(self.t_detect_factor * self.tau) / dt))),
            source_radius_squared=f32(ratio * ratio), source_r=source_r, source_c=source_c,
            source_distance_bound=bound, double_center_rc=f32(self.grid_width()))

    # fdtd.cpp:188-212
    def initial_grid(self, cell_dtype):
        n = self.grid_width()
        cells = np.zeros((n, n), dtype=cell_dtype)  # beyond the last ring: MaterialCell::halo(), all zero
        half = f64(f32(n)) / f64(2.0)
        a = (np.arange(n, dtype=f32).astype(f64) - half).astype(f32)  # float(r) - float(n) / 2.0 -> float
        distance = (self.dx * np.sqrt((a * a)[:, None] + (a * a)[None, :], dtype=f32)).astype(f32)
        assigned = np.zeros((n, n), dtype=bool)
        radius = f32(0.0)
        for i in range(len(self.rings)):
            radius = f32(radius + self.rings[i][0])
            here = (distance < radius) & ~assigned
            ca, cb, da, db = self.ring_coefficients(i)
            for name, v in (("ca", ca), ("cb", cb), ("da", da), ("db", db)):
                cells[name][here] = v
            assigned |= here
        return cells

    # fdtd.cpp:224-247: (iteration_offset, n_iterations, label of the hz snapshot or None)
    def update_calls(self):
        n, snap = self.n_timesteps(), self.n_snap_timesteps()
        if snap is None:
            return [(0, n, None)]
        return [(i, snap, i + snap) for i in range(0, n, snap)]


def load_csv(path):
    """A frame written by fdtd.cpp:118-161 (default ostream formatting: 6 significant digits)."""
    return np.loadtxt(path, delimiter=",", dtype=np.float64, ndmin=2)
