"""Development aid (GPU box): warm closed loop of the bench batch as ONE fleet (one handle, one nmpc_step_batch per period) and as TWO fleets of
half the size on two HIP streams.   python tools/cl_two_streams.py [periods]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import nmpc_amd
periods = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg, B, P, W0 = bench.make_batch("six", 0, 0, None, 2000)
dP = torch.as_tensor(P, device="cuda"); dW = torch.as_tensor(W0, device="cuda")


def run(nfleet):
    Bf = B // nfleet
    sol = [nmpc_amd.NmpcSolver(cfg, max_batch=Bf) for _ in range(nfleet)]
    sts = [torch.cuda.Stream() for _ in range(nfleet)]
    Pc = [dP[f * Bf:(f + 1) * Bf].clone() for f in range(nfleet)]; Wc = [dW[f * Bf:(f + 1) * Bf].clone() for f in range(nfleet)]
    order = [torch.arange(Bf, dtype=torch.int32, device="cuda") for _ in range(nfleet)]
    torch.cuda.synchronize(); t = time.perf_counter()
    its = []
    for _ in range(periods):
        for f in range(nfleet):
            with torch.cuda.stream(sts[f]):
                rr = sol[f].step_batch(Pc[f], Wc[f], order[f]); its.append(rr["iters"])
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    it = torch.stack([i.double().mean() for i in its]).mean().item()
    print("fleets %d x %d: %.0f solves/s, %.2f ms per period, mean iterations %.2f" % (nfleet, Bf, B * periods / dt, 1e3 * dt / periods, it), flush=True)


for n in [int(a) for a in os.environ.get("FLEETS", "1,2,4,1,2").split(",")]:
    run(n)
