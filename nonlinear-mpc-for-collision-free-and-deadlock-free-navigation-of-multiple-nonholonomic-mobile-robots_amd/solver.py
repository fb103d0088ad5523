"""Host-side mirror of the reference's call surface for the per-step solve.

    solver = nlpsol('solver', 'ipopt', cfg, opts)                 # C6:345-346
    sol    = solver(x0=, p=, lbx=, ubx=, lbg=, ubg=)              # C6:432 ; sol['x'] (n_var x 1)
    t0, u0 = shift(T, t0, u)                                      # C6:160-169,450

(C6 = AllScripts/centralized_six_robots_implementation.py.)  The batched overloads take a
leading batch dimension and keep everything on the GPU as torch tensors.  torch is used for
device memory and streams only; all arithmetic happens in libnmpc_hip.so through its C ABI.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np

from . import _lib
from .problem import ProblemConfig


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: the NMPC solve has no CPU fallback")
    return torch


def shift(T, t0, u):
    """C6:160-169: t0 += T ; u0 = [u[1:]; u[-1]] (drop first row, duplicate last)."""
    u = np.asarray(u)
    u0 = np.concatenate((u[1:u.shape[0], 0:], u[u.shape[0] - 1:u.shape[0], 0:]), axis=0)
    return t0 + T, u0


def shift_states(X, N=None):
    """C6:465: X0 = [X[1:]; X[N-1]] — the appended row is row N-1 of the (N+1)-row array."""
    X = np.asarray(X)
    N = X.shape[0] - 1 if N is None else N
    return np.concatenate((X[1:], X[N - 1:N]), axis=0)


def cold_start(cfg: ProblemConfig, x0) -> np.ndarray:
    """C6:398-400,423: X0 = repmat(x0), u0 = 0, packed [vec(X); vec(U)]."""
    x0 = np.asarray(x0, dtype=np.float64).reshape(-1)
    return np.concatenate([np.tile(x0, cfg.N + 1), np.zeros(cfg.nu * cfg.N)])


def odometry_to_global(odom, init, device=None, wrap_2pi: bool = False):
    """Batched odometry callback of the scripts (C2:18-37): odom [n,4] = (x_r, y_r, q_z, q_w), init [n,3] = (x, y, th) of the
    robot's start frame -> pose [n,3] in the global frame (device tensor).  Runs nmpc_odometry_batch.
    wrap_2pi: the modify() of the scripts without collision rows (AS/mpc_online_casadi.py:24-33), yaw in [-pi,0) -> +2 pi."""
    torch = _torch()
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    o = torch.as_tensor(odom, dtype=torch.float64, device=dev).reshape(-1, 4).contiguous()
    i0 = torch.as_tensor(init, dtype=torch.float64, device=dev).reshape(-1, 3).contiguous()
    if o.shape[0] != i0.shape[0]:
        raise ValueError(f"odom has {o.shape[0]} rows, init {i0.shape[0]}")
    pose = torch.empty((o.shape[0], 3), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.nmpc_odometry_batch(o.shape[0], o.data_ptr(), i0.data_ptr(), pose.data_ptr(), int(bool(wrap_2pi)), torch.cuda.current_stream().cuda_stream), "nmpc_odometry_batch")
    return pose


def split_swarm(p, m: int):
    """Independent-robot mode (AS/mpc_online_casadi_tb3_1.py et al. run one 3-state NLP per robot):
    p [B, 6m] = [x0 (3m); xs (3m)] of B swarms -> p1 [B*m, 6] of B*m single-robot problems (robot-major inside a swarm).
    Solve those with a ProblemConfig(m=1) solver; merge_swarm() puts the single-robot plans back side by side."""
    p = np.asarray(p, dtype=np.float64).reshape(-1, 6 * m)
    B = p.shape[0]
    return np.concatenate([p[:, : 3 * m].reshape(B, m, 3), p[:, 3 * m:].reshape(B, m, 3)], axis=2).reshape(B * m, 6)


def merge_swarm(w1, m: int, N: int):
    """w1 [B*m, 3(N+1)+2N] single-robot solutions -> w [B, (3m)(N+1)+(2m)N] in the centralized packing (C6:339)."""
    w1 = np.asarray(w1, dtype=np.float64)
    B = w1.shape[0] // m
    X = w1[:, : 3 * (N + 1)].reshape(B, m, N + 1, 3).transpose(0, 2, 1, 3).reshape(B, -1)
    U = w1[:, 3 * (N + 1):].reshape(B, m, N, 2).transpose(0, 2, 1, 3).reshape(B, -1)
    return np.concatenate([X, U], axis=1)


class NmpcSolver:
    """The object nlpsol() returns.  Owns a device workspace sized for `max_batch` instances."""

    def __init__(self, cfg: ProblemConfig, max_batch: int = 1, device: Optional[int] = None, kernel: Optional[int] = None,
                 trace_instance: Optional[int] = None):
        """kernel: None / 0 = the library picks the solve kernel per batch size; 1 / 2 / 3 pins the HBM-resident / element-per-lane /
        column-per-lane (throughput shape) kernel, 4 / 5 the column kernel's latency shape with two / four wavefronts per instance
        (nmpc_options_t).  The library itself reads no environment variable; this host class does, for development: NMPC_KERNEL (its
        first character, '1'..'5'; anything else = 0) and NMPC_TRACE_INST supply the two options when the arguments are not given."""
        import os
        self.cfg = cfg
        self.torch = _torch()
        self.lib = _lib.load()
        self.device = self.torch.device("cuda", self.torch.cuda.current_device() if device is None else device)
        self._ccfg = cfg.to_c()
        self._h = C.c_void_p()
        self.max_batch = int(max_batch)
        with self.torch.cuda.device(self.device):
            kv = os.environ.get("NMPC_KERNEL", "")
            opts = _lib.COptions(kernel=int(kernel) if kernel is not None else (int(kv[:1]) if kv[:1] in ("1", "2", "3", "4", "5") else 0),
                                 trace_instance=int(trace_instance) if trace_instance is not None else int(os.environ.get("NMPC_TRACE_INST", "-1")))
            _lib.check(self.lib.nmpc_create_opts(C.byref(self._ccfg), self.max_batch, C.byref(opts), C.byref(self._h)), "nmpc_create_opts")
        self.n_var, self.n_g, self.n_p = cfg.n_var, cfg.n_g, cfg.n_p
        assert self.n_var == self.lib.nmpc_n_var(C.byref(self._ccfg)) and self.n_g == self.lib.nmpc_n_g(C.byref(self._ccfg))
        self._stats: Dict = {}
        self._bounds = cfg.bounds()

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self.lib.nmpc_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    @property
    def workspace_bytes(self) -> int:
        return int(self.lib.nmpc_workspace_bytes(self._h))

    def kernel_for_batch(self, B: int, ordered: bool = False) -> int:
        """the solve kernel nmpc_solve_batch launches for a batch of B (ordered: with a dispatch-order hint): 3 / 4 column-per-lane in its
        throughput / latency shape (4 covers two and four wavefronts per instance), 2 element-per-lane, 1 HBM-resident (nmpc_query)"""
        return int(self.lib.nmpc_query(self._h, _lib.QUERY_KERNEL_FOR_ORDERED_BATCH if ordered else _lib.QUERY_KERNEL_FOR_BATCH, int(B)))

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, a, shape):
        t = self.torch.as_tensor(a, dtype=self.torch.float64, device=self.device).reshape(shape).contiguous()
        return t

    # ---- batched device API -----------------------------------------------------------------
    def solve_batch(self, p, w0, want_fg: bool = False, order=None):
        """p [B, 2 n_x], w0 [B, n_var] (torch cuda / numpy) -> dict of torch cuda tensors.

        order: optional permutation of range(B) (dispatch-order hint, nmpc_solve_batch_ordered): workgroup g solves instance
        order[g]; put the instances expected to need the most iterations first.  Results are unaffected."""
        torch = self.torch
        p = self._dev(p, (-1, self.n_p)); B = p.shape[0]
        w0 = self._dev(w0, (B, self.n_var))
        if B > self.max_batch:
            raise ValueError(f"batch {B} exceeds max_batch {self.max_batch}")
        w = torch.empty((B, self.n_var), dtype=torch.float64, device=self.device)
        obj = torch.empty(B, dtype=torch.float64, device=self.device)
        kkt = torch.empty(B, dtype=torch.float64, device=self.device)
        status = torch.empty(B, dtype=torch.int32, device=self.device)
        iters = torch.empty(B, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            if order is None:
                _lib.check(self.lib.nmpc_solve_batch(self._h, B, p.data_ptr(), w0.data_ptr(), w.data_ptr(), obj.data_ptr(),
                                                     status.data_ptr(), iters.data_ptr(), kkt.data_ptr(), self._stream()), "nmpc_solve_batch")
            else:
                od = torch.as_tensor(order, device=self.device).to(torch.int32).contiguous()
                if od.shape != (B,):
                    raise ValueError(f"order must have shape ({B},)")
                _lib.check(self.lib.nmpc_solve_batch_ordered(self._h, B, p.data_ptr(), w0.data_ptr(), w.data_ptr(), obj.data_ptr(), status.data_ptr(),
                                                             iters.data_ptr(), kkt.data_ptr(), od.data_ptr(), self._stream()), "nmpc_solve_batch_ordered")
        out = dict(x=w, f=obj, status=status, iters=iters, kkt=kkt)
        if want_fg:
            f, g = self.eval_batch(p, w)
            out["f"], out["g"] = f, g
        return out

    def step_batch(self, p, w, order=None):
        """One control period on the device (nmpc_step_batch): solve from the guess w, then IN PLACE w <- shifted solution and
        p[:, :n_x] <- x0 + T f(x0, u_0); `order` [B] int32 device tensor (in: dispatch order of this period, out: the order for the
        next one, sorted by this period's iteration counts, longest first) or None.  p and w must be contiguous float64 device
        tensors owned by the caller (they are modified).  Returns sol['x'] of this period and the per-instance outputs."""
        torch = self.torch
        B = p.shape[0]
        if not (torch.is_tensor(p) and torch.is_tensor(w) and p.is_cuda and w.is_cuda and p.dtype == torch.float64 and w.dtype == torch.float64
                and p.is_contiguous() and w.is_contiguous() and p.shape == (B, self.n_p) and w.shape == (B, self.n_var)):
            raise ValueError("step_batch works in place: p [B, n_p] and w [B, n_var] must be contiguous float64 device tensors")
        if B > self.max_batch:
            raise ValueError(f"batch {B} exceeds max_batch {self.max_batch}")
        if order is not None and not (torch.is_tensor(order) and order.is_cuda and order.dtype == torch.int32 and order.is_contiguous() and order.shape == (B,)):
            raise ValueError("order must be a contiguous int32 device tensor of shape (B,)")
        x = torch.empty((B, self.n_var), dtype=torch.float64, device=self.device)
        obj = torch.empty(B, dtype=torch.float64, device=self.device); kkt = torch.empty(B, dtype=torch.float64, device=self.device)
        status = torch.empty(B, dtype=torch.int32, device=self.device); iters = torch.empty(B, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nmpc_step_batch(self._h, B, p.data_ptr(), w.data_ptr(), x.data_ptr(), obj.data_ptr(), status.data_ptr(), iters.data_ptr(),
                                                kkt.data_ptr(), order.data_ptr() if order is not None else None, self._stream()), "nmpc_step_batch")
        return dict(x=x, f=obj, status=status, iters=iters, kkt=kkt, order=order)

    def eval_batch(self, p, w):
        torch = self.torch
        p = self._dev(p, (-1, self.n_p)); B = p.shape[0]
        w = self._dev(w, (B, self.n_var))
        f = torch.empty(B, dtype=torch.float64, device=self.device)
        g = torch.empty((B, self.n_g), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nmpc_eval_batch(self._h, B, p.data_ptr(), w.data_ptr(), f.data_ptr(), g.data_ptr(), self._stream()), "nmpc_eval_batch")
        return f, g

    def shift_batch(self, p, w, plant: bool = True):
        """device warm-start shift (C6:160-169,460-465) and, if plant, x0 + T f(x0,u0) (casadi_test.py:17-26)."""
        torch = self.torch
        p = self._dev(p, (-1, self.n_p)); B = p.shape[0]
        w = self._dev(w, (B, self.n_var))
        wn = torch.empty_like(w)
        x0n = torch.empty((B, self.cfg.nx), dtype=torch.float64, device=self.device) if plant else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nmpc_shift_batch(self._h, B, p.data_ptr(), w.data_ptr(), wn.data_ptr(),
                                                 x0n.data_ptr() if plant else None, self._stream()), "nmpc_shift_batch")
        return wn, x0n

    # ---- the reference's single-instance keyword call (C6:432) ----------------------------------
    def __call__(self, x0=None, p=None, lbx=None, ubx=None, lbg=None, ubg=None, **kw):
        if kw:
            raise TypeError(f"unexpected arguments {sorted(kw)}")
        if p is None or x0 is None:
            raise ValueError("x0 and p are required")
        p = np.asarray(p, dtype=np.float64).reshape(-1)
        w0 = np.asarray(x0, dtype=np.float64).reshape(-1)
        if p.size != self.n_p:
            raise ValueError(f"p has {p.size} entries, expected {self.n_p}")
        if w0.size != self.n_var:
            raise ValueError(f"x0 has {w0.size} entries, expected {self.n_var}")
        # bounds are structural here: accept (n,1)/(1,n)/flat, and insist they are the configured ones
        for name, given, want in (("lbx", lbx, self._bounds[0]), ("ubx", ubx, self._bounds[1]),
                                  ("lbg", lbg, self._bounds[2]), ("ubg", ubg, self._bounds[3])):
            if given is None:
                continue
            g = np.asarray(given, dtype=np.float64).reshape(-1)
            if g.size != want.size:
                raise ValueError(f"{name} has {g.size} entries, expected {want.size}")
            if not np.array_equal(g, want):
                raise ValueError(f"{name} differs from the bounds of the configured problem; build a new config instead")
        r = self.solve_batch(p[None, :], w0[None, :], want_fg=True)
        self.torch.cuda.synchronize(self.device)
        st = int(r["status"][0])
        self._stats = dict(return_status=_lib.STATUS_NAMES.get(st, str(st)), success=(st == 0), iter_count=int(r["iters"][0]),
                           kkt_error=float(r["kkt"][0]), status_code=st)
        return {"x": r["x"][0].cpu().numpy().reshape(-1, 1), "f": float(r["f"][0]), "g": r["g"][0].cpu().numpy().reshape(-1, 1)}

    def stats(self) -> Dict:
        """CasADi's solver.stats(); the reference never reads it, non-convergence is reported here only."""
        return dict(self._stats)


def nlpsol(name: str, plugin: str, problem, opts: Optional[dict] = None, max_batch: int = 1) -> NmpcSolver:
    """Drop-in for casadi.nlpsol(name,'ipopt',nlp_prob,opts) (C6:345-346) with `problem` a ProblemConfig.

    Honoured keys of opts['ipopt']: max_iter, tol / acceptable_tol, mu_init; print_* are accepted and ignored."""
    if plugin != "ipopt":
        raise ValueError("only the 'ipopt' call surface of the reference is mirrored")
    if not isinstance(problem, ProblemConfig):
        raise TypeError("problem must be a ProblemConfig (the symbolic CasADi graph is replaced by its parameters)")
    ip = dict((opts or {}).get("ipopt", {}))
    cfg = ProblemConfig(**{**problem.__dict__})
    if "max_iter" in ip: cfg.max_iter = int(ip["max_iter"])
    if "tol" in ip: cfg.tol = float(ip["tol"])
    elif "acceptable_tol" in ip: cfg.tol = float(ip["acceptable_tol"])
    if "mu_init" in ip: cfg.mu_init = float(ip["mu_init"])
    return NmpcSolver(cfg, max_batch=max_batch)


class PipelinedSolver:
    """Several launches in flight: `depth` handles (one workspace each) on `depth` HIP streams, used round-robin.  A launch lasts as long as its longest
    solve, so towards its end most of the device idles; the next batch's launch on another stream fills it (bench.py `two_streams`: six
    robots, B = 4096, 277 k -> 445 k solves/s with identical results).  Not part of the reference's call surface: a host-side convenience over the
    stream-ordered C ABI.

        pipe = PipelinedSolver(cfg, max_batch=4096)
        pending = [pipe.solve_batch(p_k, w0_k) for p_k, w0_k in batches]      # returns at once; results are ordered on the slot's stream
        pipe.synchronize()                                                    # or: r["event"].synchronize() for one result
    """

    def __init__(self, cfg: ProblemConfig, max_batch: int = 1, depth: int = 2, **kw):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.solvers = [NmpcSolver(cfg, max_batch=max_batch, **kw) for _ in range(depth)]
        self.torch = self.solvers[0].torch
        self.device = self.solvers[0].device
        with self.torch.cuda.device(self.device):
            self.streams = [self.torch.cuda.Stream() for _ in range(depth)]
        self._next = 0

    def solve_batch(self, p, w0, **kw):
        """NmpcSolver.solve_batch on the next slot's stream.  The slot's stream first waits for the caller's current stream (the inputs may still be in
        production there); the result carries `event`, recorded on the slot's stream after the solve — wait on it (or call synchronize()) before reading."""
        torch = self.torch
        k = self._next
        self._next = (k + 1) % len(self.solvers)
        st = self.streams[k]
        sol = self.solvers[k]
        p = sol._dev(p, (-1, sol.n_p))                       # on the caller's stream (a host array is copied there)
        w0 = sol._dev(w0, (p.shape[0], sol.n_var))
        st.wait_stream(torch.cuda.current_stream(self.device))
        p.record_stream(st); w0.record_stream(st)            # the caller may drop its references at once: the allocator must not hand the inputs out again before the solve has read them
        with torch.cuda.stream(st):
            r = sol.solve_batch(p, w0, **kw)
            ev = torch.cuda.Event()
            ev.record(st)
        r["event"] = ev
        r["slot"] = k
        return r

    def synchronize(self):
        for st in self.streams:
            st.synchronize()
