"""Development aid: per-phase cycle shares of the LDS-resident solve kernel (NMPC_PROFILE build).

    NMPC_PROFILE=1 NMPC_FORCE_BUILD=1 python tools/phase_profile.py [batch]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NMPC_PROFILE"] = "1"
import importlib
import torch
import nmpc_amd
from oracle import nlp_ref as R
from tests import helpers as Hh

if not os.environ.get("NMPC_SO"):      # NMPC_SO: a profile build made beforehand (variants/)
    importlib.import_module("nmpc_amd.build").build(force=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
REPS = int(os.environ.get("PHASE_REPS", "1"))
name = sys.argv[2] if len(sys.argv) > 2 else "six"
ocfg = {"six": R.cfg_six(20), "two": R.cfg_two(20), "ten": R.cfg_ten(30), "ten20": R.cfg_ten(20)}[name]
cfg = Hh.to_product_cfg(ocfg, max_iter=2000)
P, W0 = Hh.batch(ocfg, B, 2)
s = nmpc_amd.NmpcSolver(cfg, max_batch=B)
L = s.lib
out = (C.c_int64 * 12)()
r = s.solve_batch(P, W0); torch.cuda.synchronize()
L.nmpc_debug_profile(s._h, out, 1)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); r = s.solve_batch(P, W0); t1.record(); torch.cuda.synchronize()
L.nmpc_debug_profile(s._h, out, 1)
it = r["iters"].cpu().numpy()
tot_it = it.sum()
names = ["setup", "A kkt-error", "B0 stage packs", "B riccati: schur+gains+rest", "C forward", "D frac-to-bnd", "E line search", "F multipliers", "G update", "B riccati: pack+G pass", "B riccati: assembly", "B riccati: elimination"]
cyc = np.array([out[i] for i in range(12)], dtype=np.float64)
print(f"batch {B} ({name}): kernel {t0.elapsed_time(t1):.2f} ms, mean iters {it.mean():.1f}, max {it.max()}")
for n, c in zip(names, cyc):
    print(f"  {n:30s} {c / tot_it:12.0f} clock64-ticks/iter  {100 * c / cyc.sum():5.1f} %")
print(f"  total            {cyc.sum() / tot_it:12.0f} ticks/iter  (clock64 = 100 MHz realtime on gfx9: x10 ns)")
