"""Problem description: every literal of one reference script in one object.

AS = AllScripts/ of the reference; C6 = AS/centralized_six_robots_implementation.py.
The reference hard-codes these as module globals (C6:197-205, 252-266, 349-352) and copies
the file per scenario; here they are runtime parameters of one generic solver."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

from ._lib import CConfig, NMPC_MAX_OBSTACLES


@dataclass
class ProblemConfig:
    m: int = 2
    N: int = 20
    T: float = 0.05
    dmin: float = 0.15                                   # pair rows >= dmin**2 (C6:349)
    q: Tuple[float, float, float] = (1.0, 5.0, 0.1)      # C6:252-257
    r: Tuple[float, float] = (0.5, 0.05)                 # C6:259-264
    v_max: float = 0.22
    w_max: float = 2.84
    xy_max: float = 10.0                                 # C6:351-352
    th_max: float = math.inf                             # finite only in the obstacle scripts
    obstacles: List[Tuple[float, float, float]] = field(default_factory=list)   # (ox, oy, obs_r)
    rob_dim: float = 0.2
    margin: float = 0.1
    pad_value: float = 3.5                               # C6:278
    pad_rows: bool = True
    pair_rows: bool = True                               # False: AS/mpc_online_casadi_tb3_multi_centralized.py:115-148 (no collision rows)
    # 'ipopt' options of C6:345 that the solve honours
    tol: float = 1e-8
    mu_init: float = 0.5
    max_iter: int = 2000

    @property
    def nx(self): return 3 * self.m
    @property
    def nu(self): return 2 * self.m
    @property
    def M(self): return self.m * (self.m - 1) // 2 if self.pair_rows else 0
    @property
    def n_var(self): return self.nx * (self.N + 1) + self.nu * self.N
    @property
    def rows0(self): return self.nx + (self.M if self.pad_rows else 0)
    @property
    def rows_k(self): return self.nx + self.M + self.m * len(self.obstacles)
    @property
    def n_g(self): return self.rows0 + self.rows_k * self.N
    @property
    def n_p(self): return 2 * self.nx

    def to_c(self) -> CConfig:
        if len(self.obstacles) > NMPC_MAX_OBSTACLES:
            raise ValueError("too many obstacles")
        c = CConfig()
        c.m, c.N, c.n_obs, c.pad_rows = self.m, self.N, len(self.obstacles), int(self.pad_rows)
        c.T, c.dmin = self.T, self.dmin
        c.q[:] = self.q; c.r[:] = self.r
        c.v_max, c.w_max, c.xy_max, c.th_max = self.v_max, self.w_max, self.xy_max, self.th_max
        c.rob_dim, c.margin, c.pad_value = self.rob_dim, self.margin, self.pad_value
        for i, (ox, oy, orad) in enumerate(self.obstacles):
            c.obs[3 * i], c.obs[3 * i + 1], c.obs[3 * i + 2] = ox, oy, orad
        c.tol, c.mu_init, c.max_iter = self.tol, self.mu_init, self.max_iter
        c.pair_rows = int(self.pair_rows)
        return c

    # bounds exactly as the scripts build them (C6:349-352; third_scenario_mpc_obstacle_avoidance.py:175-177)
    def bounds(self):
        inf = math.inf
        lbs = np.tile(np.array([-self.xy_max, -self.xy_max, -self.th_max]), self.m)
        lbu = np.tile(np.array([-self.v_max, -self.w_max]), self.m)
        lbx = np.concatenate([np.tile(lbs, self.N + 1), np.tile(lbu, self.N)])
        ubx = -lbx
        K = len(self.obstacles)
        lb0 = np.concatenate([np.zeros(self.nx), np.full(self.rows0 - self.nx, self.dmin ** 2)])
        ub0 = np.concatenate([np.zeros(self.nx), np.full(self.rows0 - self.nx, inf)])
        lbk = np.concatenate([np.zeros(self.nx), np.full(self.M, self.dmin ** 2), np.full(self.m * K, self.margin)])
        ubk = np.concatenate([np.zeros(self.nx), np.full(self.M + self.m * K, inf)])
        return lbx, ubx, np.concatenate([lb0, np.tile(lbk, self.N)]), np.concatenate([ub0, np.tile(ubk, self.N)])


# ---- the reference scripts as presets (file's own N unless overridden) --------------------------
def centralized_one_robot(N: int = 100) -> ProblemConfig:        # AS/centralized_one_robots_implementation.py:58-63
    return ProblemConfig(m=1, N=N, T=0.05, dmin=0.0, v_max=0.22, w_max=2.84, pad_rows=False)

def centralized_two_robots(N: int = 70) -> ProblemConfig:        # AS/centralized_two_robots_implementation.py:101-109
    return ProblemConfig(m=2, N=N, T=0.05, dmin=0.15, v_max=0.22, w_max=2.84)

def centralized_six_robots(N: int = 35) -> ProblemConfig:        # C6:197-205
    return ProblemConfig(m=6, N=N, T=0.3, dmin=0.4, v_max=0.15, w_max=1.5)

def two_robots_no_collision_rows(N: int = 50) -> ProblemConfig:  # AS/mpc_online_casadi_tb3_multi_centralized.py:60-66,115-148
    return ProblemConfig(m=2, N=N, T=0.01, dmin=0.0, v_max=0.22, w_max=2.84, pad_rows=False, pair_rows=False)

def ten_robots_collision_avoidance(N: int = 20) -> ProblemConfig:  # AS/mpc_online_casadi_tb3_ten_multi_centralized_collision_avoidance.py:169-177
    return ProblemConfig(m=10, N=N, T=0.1, dmin=0.3, v_max=0.22, w_max=2.84)

def third_scenario_obstacles(N: int = 100) -> ProblemConfig:     # AS/third_scenario_mpc_obstacle_avoidance.py:56-63,97-119,175-177
    return ProblemConfig(m=1, N=N, T=0.2, dmin=0.0, v_max=0.2, w_max=1.0, th_max=2 * math.pi, pad_rows=False,
                         rob_dim=0.2, margin=0.1,
                         obstacles=[(-0.6, 3.3, 0.2), (0.6, 3.3, 0.125), (0.0, 2.3, 0.15),
                                    (1.0, 2.3, 0.15), (-0.6, 1.3, 0.2), (0.6, 1.3, 0.175)])

def six_robots_eight_obstacles(N: int = 25, obstacles=None) -> ProblemConfig:
    """BASELINE.json config 5: synthetic composite (C6 pair rows + obstacle rows); no reference script."""
    c = centralized_six_robots(N)
    c.obstacles = list(obstacles or [])
    c.rob_dim, c.margin = 0.2, 0.1
    return c


# ---- every remaining script of AS/ that builds this NLP, as a parameter set ---------------------
# (the reference copies one file per scenario; the NLP differs only in these literals.)
# name -> (m, N, T, dmin, extra kwargs); cited lines hold T/N/m/dmin, v/omega limits and `args`.
_PAIR = dict(v_max=0.22, w_max=2.84)
_SOLO = dict(v_max=0.22, w_max=2.84, dmin=0.0, pad_rows=False)
_OBS = dict(dmin=0.0, pad_rows=False, v_max=0.2, w_max=math.pi / 4, th_max=2 * math.pi, rob_dim=0.15, margin=0.05)
_SCRIPT_TABLE = {
    "first_scenario":                   (1, 100, 0.05, dict(_SOLO)),                  # AS/first_scenario.py:58-63,131-133
    "second_scenario":                  (2, 50, 0.1, dict(_PAIR, dmin=0.25)),         # AS/second_scenario.py:80-88,181-183
    "third_scenario":                   (3, 50, 0.05, dict(_PAIR, dmin=0.3)),         # AS/third_scenario.py:92-100,206-209
    "fourth_scenario":                  (4, 50, 0.1, dict(_PAIR, dmin=0.3)),          # AS/fourth_scenario.py:104-112,229-232
    "fifth_scenario":                   (5, 35, 0.1, dict(_PAIR, dmin=0.3)),          # AS/fifth_scenario.py:115-123,241-244
    "sixth_scenario":                   (6, 35, 0.3, dict(_PAIR, dmin=0.3)),          # AS/sixth_scenario.py:127-135,279-282
    "decentralized_first_scenario":     (1, 200, 0.05, dict(_SOLO, xy_max=2.0)),      # AS/decentralized_first_scenario.py:94-100,183-185
    "decentralized_two_robots":         (2, 50, 0.1, dict(_PAIR, dmin=0.25)),         # AS/decentralized_two_robots.py:80-88,180-182
    "centralized_three_robots":         (3, 60, 0.05, dict(_PAIR, dmin=0.15)),        # AS/centralized_three_robots_implementation.py:127-135,241-244
    "centralized_four_robots":          (4, 45, 0.1, dict(_PAIR, dmin=0.4)),          # AS/centralized_four_robots_implementation.py:150-158,275-278
    "centralized_five_robots":          (5, 40, 0.1, dict(_PAIR, dmin=0.4)),          # AS/centralized_five_robots_implementation.py:174-182,300-303
    "tb3_two_collision_free":           (2, 100, 0.02, dict(_PAIR, dmin=0.25)),       # AS/mpc_online_casadi_tb3_two_centralized_collision_free.py:80-88,180-182
    "tb3_five_collision_free":          (5, 70, 0.02, dict(_PAIR, dmin=0.3)),         # AS/mpc_online_casadi_tb3_multi_centralized_collision_free.py:115-123,241-244
    "tb3_six_collision_free":           (6, 35, 0.2, dict(_PAIR, dmin=0.3)),          # AS/mpc_online_casadi_tb3_six_multi_centralized_collision_free.py:127-135,279-282
    "tb3_eight_collision_free":         (8, 5, 0.02, dict(_PAIR, dmin=0.25)),         # AS/mpc_online_casadi_tb3_eight_multi_centralized_collision_free.py:148-156,329-332
    "mpc_online_casadi":                (1, 50, 0.01, dict(_SOLO)),                   # AS/mpc_online_casadi.py:56-61,129-131
    "mpc_online_casadi_tb3_1":          (1, 200, 0.01, dict(_SOLO)),                  # AS/mpc_online_casadi_tb3_1.py:56-61,129-131 (tb3_2, tb3_3: same literals)
    "casadi_test":                      (1, 25, 0.25, dict(_SOLO)),                   # AS/casadi_test.py:34-39,107-109
    "casadi_test_mpc":                  (1, 50, 0.02, dict(_SOLO)),                   # AS/casadi_test_mpc.py:56-61,129-131
    "first_scenario_obstacle":          (1, 100, 0.1, dict(_OBS, obstacles=[(0.4, 1.1, 0.15)])),   # AS/first_scenario_mpc_obstacle_avoidance.py:56-63,97-99,150
    "second_scenario_obstacles":        (1, 100, 0.1, dict(_OBS, obstacles=[(1.0, 0.5, 0.15), (-0.75, 0.0, 0.125),
                                                                          (0.0, -1.25, 0.15), (0.0, 1.0, 0.125)])),  # AS/second_scenario_mpc_obstacle_avoidance.py:56-63,97-111,164
}
_NAMED = {
    "centralized_one_robot": centralized_one_robot, "centralized_two_robots": centralized_two_robots,
    "centralized_six_robots": centralized_six_robots, "two_robots_no_collision_rows": two_robots_no_collision_rows,
    "ten_robots_collision_avoidance": ten_robots_collision_avoidance, "third_scenario_obstacles": third_scenario_obstacles,
}


def script_names() -> List[str]:
    """Every reference script (of those that build the CasADi NLP of SURVEY 8) that has a preset."""
    return sorted(list(_SCRIPT_TABLE) + list(_NAMED))


def script_preset(name: str, N: int = None) -> ProblemConfig:
    """ProblemConfig holding the literals of reference script `name` (see `script_names()`); `N` overrides
    the file's own horizon.  Unknown names raise KeyError listing the known ones."""
    if name in _NAMED:
        return _NAMED[name]() if N is None else _NAMED[name](N)
    if name not in _SCRIPT_TABLE:
        raise KeyError(f"no preset {name!r}; known: {', '.join(script_names())}")
    m, n_file, T, kw = _SCRIPT_TABLE[name]
    kw = dict(kw)
    if "obstacles" in kw:
        kw["obstacles"] = list(kw["obstacles"])
    return ProblemConfig(m=m, N=n_file if N is None else N, T=T, **kw)
