"""Development aid (GPU box): the column kernel's latency shape (two wavefronts per instance, nmpc_options_t.kernel = 4) against its
throughput shape (kernel = 3): results and launch times on the bench batch.   python tools/dbg_lat.py [six|two|ten|ten20|composite] [B ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nmpc_amd, bench
from tests import helpers as Hh
name = sys.argv[1] if len(sys.argv) > 1 else "six"
sizes = [int(a) for a in sys.argv[2:]] or [1, 64, 512, 1024, 2048, 4096]
from tests import helpers as Hh
ocfg, _, P, W0 = Hh.bench_batch(name, max(sizes))
cfg = Hh.to_product_cfg(ocfg)
s3 = nmpc_amd.NmpcSolver(cfg, max_batch=max(sizes), kernel=3)
s4 = nmpc_amd.NmpcSolver(cfg, max_batch=max(sizes), kernel=int(os.environ.get("LAT_KERNEL", "4")))      # LAT_KERNEL=5: four wavefronts per instance (five / six robots)
print("kernel ids", s3.kernel_for_batch(sizes[-1]), s4.kernel_for_batch(sizes[-1]), flush=True)
for B in sizes:
    dP = torch.as_tensor(P[:B], device="cuda"); dW = torch.as_tensor(W0[:B], device="cuda")
    out = []
    for s in (s3, s4):
        s.solve_batch(dP, dW); torch.cuda.synchronize()
        t = time.perf_counter(); r = s.solve_batch(dP, dW); torch.cuda.synchronize(); dt = time.perf_counter() - t
        out.append((dt, r))
    r3, r4 = out[0][1], out[1][1]
    dw = (r3["x"] - r4["x"]).abs().amax(dim=1).cpu().numpy()
    print(f"{name} B={B:5d}: throughput shape {out[0][0]*1e3:8.2f} ms ({B/out[0][0]:9.0f}/s)  latency shape {out[1][0]*1e3:8.2f} ms ({B/out[1][0]:9.0f}/s)  "
          f"status equal {bool((r3['status'] == r4['status']).all())} iters equal {(r3['iters'] == r4['iters']).float().mean().item():.4f} same point {(dw <= 1e-6).mean():.4f} max iters {int(r4['iters'].max())}", flush=True)
