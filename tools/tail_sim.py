"""Development aid (CPU only): what a launch costs as a function of the iteration counts of its instances, and what two ways of treating the
stragglers could buy.  Iteration counts come from the C oracle on the bench batch (seconds on 8 cores); the per-iteration times of a wave are
the measured ones (130 us alone on its SIMD, 184 us with the chip full, 104 us in the two-wavefront latency shape).

    python tools/tail_sim.py [workload] [B]

  one launch                     greedy dispatch onto 2048 wave slots, per-iteration time interpolated with the number of active waves
  capped launch + resume         every solve stops after K iterations, the survivors are resumed in the latency shape by a second launch
  dynamic helper (upper bound)   as soon as half the slots are idle every running solve proceeds at the latency shape's rate
Result for six robots, B=4096 (round 3): one launch 19.5 ms (measured 19.0); capped + resume 19.7 .. 22.3 ms for K = 30 .. 80 (never a gain:
the first launch cannot end before its own last solve); dynamic helper 17.6 ms (+10 %), which would need the kernel to change its shape
in flight.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import oracle_lib as O

name = sys.argv[1] if len(sys.argv) > 1 else "six"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
from tests import helpers as Hh
ocfg, B, P, W0 = Hh.bench_batch(name, B)
it = O.solve_batch(O.make_config(ocfg, max_iter=2000), P, W0)["iters"].astype(int)
print("%s B=%d: mean %.1f, max %d iterations; longest: %s" % (name, B, it.mean(), it.max(), np.argsort(-it)[:6]))
LONE, LOADED, LAT_LONE, LAT_LOADED = 130e-6, 184e-6, 104e-6, 150e-6


def t_iter(nact, slots, lone, loaded):
    x = np.clip((nact - slots / 4) / (slots - slots / 4), 0, 1)
    return lone + (loaded - lone) * x


def run(iters, slots, lone, loaded, helper=None):
    n = len(iters)
    nxt = min(slots, n)
    rem = iters.astype(float).copy()
    active = np.arange(nxt)
    t = 0.0
    while len(active):
        ti = t_iter(len(active), slots, lone, loaded)
        if helper is not None and len(active) <= slots / 2:
            ti *= helper / lone
        r = rem[active]
        t += r.min() * ti
        rem[active] -= r.min()
        keep = rem[active] > 1e-9
        k = min(int((~keep).sum()), n - nxt)
        active = active[keep]
        if k > 0:
            active = np.concatenate([active, np.arange(nxt, nxt + k)])
            nxt += k
    return t


print("one launch                    %.2f ms" % (1e3 * run(it, 2048, LONE, LOADED)))
for K in (30, 40, 48, 64, 80):
    t1 = run(np.minimum(it, K), 2048, LONE, LOADED)
    surv = it[it > K] - K
    t2 = run(surv, 1024, LAT_LONE, LAT_LOADED) if len(surv) else 0.0
    print("cap %3d + resume (%4d left)   %.2f + %.2f = %.2f ms" % (K, len(surv), 1e3 * t1, 1e3 * t2, 1e3 * (t1 + t2)))
print("dynamic helper (upper bound)  %.2f ms" % (1e3 * run(it, 2048, LONE, LOADED, helper=LAT_LONE)))
