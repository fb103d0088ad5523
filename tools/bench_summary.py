"""Development aid: one line per workload of a bench.py JSON line.   python tools/bench_summary.py [bench.json]"""
import json, sys
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench.json"))
rl = d["roofline"]; ss = d["solve_stats"]
print("%-46s %9.0f solves/s %7.2f ms  iters %.2f / %d  frac %.4f (sustained %.4f)  traffic %s" % (
    d["config"]["workload"][:46], d["value"], d["ms_per_step"], ss["mean_iters"], ss["max_iters"], rl["frac"], rl.get("frac_sustained", 0.0),
    ("%.0f GB/s" % rl["traffic_GBps"]) if rl.get("traffic") else "null"))
for k in ("closed_loop", "host_buffers", "two_streams"):
    if k in d:
        print("%-46s %9.0f solves/s %7.2f ms" % (k, d[k]["solves_per_s"], d[k].get("ms_per_step", d[k].get("ms_per_launch", 0.0))))
for s in d.get("sweep", []):
    r = s["roofline"]
    print("%-46s %9.0f solves/s %7.2f ms  iters %.2f / %d  frac %.4f  traffic %s  %s" % (
        s["workload"][:46], s["value"], s["ms_per_step"], s["mean_iters"], s["max_iters"], r["frac"], ("%.0f GB/s" % r["traffic_GBps"]) if r.get("traffic") else "null", s["status_counts"]))
for k in ("cpu_baseline", "casadi"):
    if k in d:
        print(k, d[k] if isinstance(d[k], str) else {a: d[k][a] for a in ("value", "cores", "kind") if a in d[k]})
