"""Development aid (GPU): lone-iteration time of each kernel shape = launch time of a batch small enough that every instance has a SIMD
(or CU) to itself, divided by the launch's longest solve.   python tools/lone_iter.py [workload] [B ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import nmpc_amd

wname = sys.argv[1] if len(sys.argv) > 1 else "six"
Bs = [int(a) for a in sys.argv[2:]] or [64, 256, 1024]
for kern in (3, 4, 5):
    for B in Bs:
        cfg, _, p, w0 = bench.make_batch(wname, 0, B, None, 2000)
        try:
            s = nmpc_amd.NmpcSolver(cfg, max_batch=B, kernel=kern)
        except Exception as e:
            print("kernel", kern, "B", B, "unavailable:", e); continue
        pd = torch.as_tensor(p, device="cuda"); wd = torch.as_tensor(w0, device="cuda")
        r = s.solve_batch(pd, wd); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t = time.perf_counter(); r = s.solve_batch(pd, wd); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        it = r["iters"].cpu().numpy() if hasattr(r["iters"], "cpu") else np.asarray(r["iters"])
        print("workload %s kernel %d B %5d: %.3f ms, iters mean %.1f max %d -> %.1f us per iteration of the longest solve; %.0f solves/s"
              % (wname, kern, B, 1e3 * min(ts), it.mean(), it.max(), 1e6 * min(ts) / it.max(), B / min(ts)), flush=True)
