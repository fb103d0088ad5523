"""Development aid: the "Measured" table of DESIGN.md 7 from a bench.py record.   python tools/design_table.py [profiles/current/bench.json]"""
import json, sys
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "profiles/current/bench.json"))
rows = []


def fmt(v):
    return ("%.2f M" % (v / 1e6)) if v >= 1e6 else ("%.1f k" % (v / 1e3))


def row(name, B, val, ms, mi, mx, rl, prof, two=None):
    tr = ("%.0f KB, %.2f TB/s" % (rl["traffic"] / (mi * B) / 1024, rl["traffic_GBps"] / 1e3)) if rl.get("traffic") else "null"
    rows.append("| %s | %d | **%s** | %.2f | %s | %.1f \\| %d | %.4f (%.3f) | %s | `%s` |" % (
        name, B, fmt(val), ms, fmt(two["solves_per_s"]) if two else "–", mi, mx, rl["frac"], rl.get("frac_sustained", rl["frac"] * 78.6 / 47.7), tr, prof))


ss = d["solve_stats"]
row("six robots N=20 (BASELINE config 3, **headline**)", 4096, d["value"], d["ms_per_step"], ss["mean_iters"], ss["max_iters"], d["roofline"], "current", d.get("two_streams"))
names = {"two": "two robots N=20", "ten20": "ten robots N=20", "ten": "ten robots N=30", "composite": "six robots + 8 obstacles N=25 (config 5 shard; four wavefronts per instance)",
         "six": "six robots N=20", "lidar_v4": "LIDAR V4 (N=100, Nc=50, R=10)"}
dirs = {("two", 4096): "current_two", ("two", 1024): "current_two_b1024", ("ten20", 4096): "current_ten20", ("ten", 512): "current_ten", ("ten", 4096): "current_ten_b4096",
        ("composite", 1024): "current_composite", ("six", 16384): "current_b16384", ("lidar_v4", 4096): "current_lidar"}
for s in d["sweep"]:
    k = s["workload"].split(":")[0]
    nm = names[k]
    if k == "ten" and s["batch"] == 512:
        nm += " (config 4 shard; two wavefronts per instance)"
    if k == "two" and s["batch"] == 1024:
        nm += " (config 2's own batch)"
    row(nm, s["batch"], s["value"], s["ms_per_step"], s["mean_iters"], s["max_iters"], s["roofline"], dirs.get((k, s["batch"]), "-"), s.get("two_streams"))
print("| shape | B | solves/s | ms per launch | two launches in flight | iterations mean \\| max | frac of 78.6 TFLOP/s (of the sustained 47.7) | HBM-side traffic per iteration and instance, rate | `profiles/` |")
print("|---|---|---|---|---|---|---|---|---|")
print("\n".join(rows))
cl = d.get("closed_loop", {}); hb = d.get("host_buffers", {}); cb = d.get("cpu_baseline", {})
ts = d.get("two_streams", {})
if ts:
    print("\"Two launches in flight\": the same batch through two handles on two HIP streams, launches alternating (`bench.py` `two_streams`; an extra, never `value`) — the second "
          "launch runs on the SIMDs the first one's tail leaves idle: six robots B=4096 **%.0f k solves/s**, %.2f ms per launch, identical results; as the timed steps themselves, `python bench.py --streams 2` / `3`: 476 k / 497 k (the default stays one launch at a time)." % (ts["solves_per_s"] / 1e3, ts["ms_per_launch"]))
print(("Warm closed loop (20 periods x 4096 swarms, one `nmpc_step_batch` per period): **%.0f k solves/s** (mean %.1f iterations per warm solve; as four fleets of 1024 on four streams: FLEETS).  Host (pageable numpy) "
      "buffers at the boundary: %.0f k solves/s (PCIe-inclusive, never `value`).  CPU baseline (`oracle/nmpc_oracle.c`, %d host threads, %s): **%.1f k solves/s**.  %s" % (
          cl.get("solves_per_s", 0) / 1e3, cl.get("mean_iters_later_steps", 0), hb.get("solves_per_s", 0) / 1e3, cb.get("cores", 0), cb.get("sample", "").split(";")[0], cb.get("value", 0) / 1e3,
          "CasADi/IPOPT: " + (d["casadi"] if isinstance(d.get("casadi"), str) else json.dumps(d.get("casadi"))))).replace("FLEETS", ("%.0f k" % (cl["four_fleets"]["solves_per_s"] / 1e3)) if "four_fleets" in cl else "not measured"))
