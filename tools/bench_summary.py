"""Development aid: one line per workload of a bench.py JSON line (stdin or file)."""
import json, sys
d = json.load(open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin)
print("six", round(d["value"]), "closed-loop", round(d["closed_loop"]["solves_per_s"]), "host-buffers", round(d["host_buffers"]["solves_per_s"]),
      "same-basin vs cpu", d["cpu_baseline"].get("same_basin_frac_vs_gpu"), "conv", d["solve_stats"]["converged_frac"], "iters", round(d["solve_stats"]["mean_iters"], 2))
for s in d.get("sweep", []):
    print("  ", s["workload"][:40], round(s["value"]), "iters", round(s["mean_iters"], 2), "conv", s["converged_frac"])
print(d["library"])
