#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/collect_profiles.sh into profiles/<tag>/ (small, tracked files).

    python tools/summarize_profiles.py gpurun_out/prof_r1d profiles/r1_final

  kernel_stats.csv  per-kernel calls / total / average duration (the --kernel-trace --stats pass)
  hbm_traffic.json  FETCH_SIZE / WRITE_SIZE of the solve kernel per launch, corrected as MI355X_MICROARCH.md prescribes:
                    both counters are in KB; on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B -> doubled
  sq_counters.json  SQ instruction counters of the solve kernel per launch
"""
import csv, json, os, shutil, sqlite3, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
KERNEL = os.environ.get("NMPC_PROFILE_KERNEL", "solve_lds_kernel")


def db(name):
    return sqlite3.connect(os.path.join(src, name, "run_results.db"))


# --- kernel stats
c = db("stats")
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(os.path.join(dst, "kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.4f" % (100.0 * r[2] / tot), r[4], r[5]])
solve = [r for r in rows if KERNEL in r[0]][0]
# the symbol of the kernel that was actually dispatched (the library holds one instantiation per team size)
sym = c.execute("select kernel_name, group_segment_size, private_segment_size, sgpr_count, arch_vgpr_count, accum_vgpr_count from kernel_symbols where display_name = ?", (solve[0],)).fetchall()
disp = c.execute("select grid_x, workgroup_x, lds_size, scratch_size from kernels where name like ? limit 1", ("%" + KERNEL + "%",)).fetchone()


def counters(name):
    c = db(name)
    out = {}
    for cn, n, s in c.execute("select counter_name, count(*), sum(value) from counters_collection where kernel_name like ? group by counter_name", ("%" + KERNEL + "%",)):
        out[cn] = {"launches": n, "per_launch": s / n}
    return out


fe, wr = counters("fetch"), counters("write")
fetch_kb, write_kb = fe["FETCH_SIZE"]["per_launch"], wr["WRITE_SIZE"]["per_launch"]
traffic = {
    "kernel": solve[0], "avg_duration_ms_trace_pass": solve[3] * 1e-6, "launches_trace_pass": solve[1],
    "FETCH_SIZE_KB_per_launch_raw": fetch_kb, "WRITE_SIZE_KB_per_launch_raw": write_kb,
    "read_bytes_per_launch": 2.0 * fetch_kb * 1024.0, "write_bytes_per_launch": write_kb * 1024.0,
    "hbm_bytes_per_launch": 2.0 * fetch_kb * 1024.0 + write_kb * 1024.0,
    "correction": "KB units; FETCH_SIZE doubled (gfx950 counts 128-B read requests as 64 B); WRITE_SIZE as read; separate --pmc passes",
    "workload": json.load(open(os.path.join(src, "bench_fetch.json")))["config"],
    "dispatch": {"grid_x": disp[0], "workgroup_x": disp[1], "lds_bytes": disp[2], "scratch_bytes_per_lane": disp[3]},
    "registers": [{"kernel": s[0][:60], "lds_static": s[1], "scratch_bytes_per_lane": s[2], "sgpr": s[3], "arch_vgpr_as_rocprofv3_reports": s[4], "accum_vgpr": s[5],
                   "note": "rocprofv3's arch_vgpr_count is half of the unified 512-entry file's allocation on gfx950 (a kernel compiled to 244-256 VGPRs shows 128); "
                           "the code object's own .vgpr_count (hipcc -S) is the figure DESIGN.md quotes"} for s in sym],
}
bj = json.load(open(os.path.join(src, "bench_fetch.json")))
import re as _re
_m = _re.search(r"src=([0-9a-f]{16})", bj.get("library", ""))
traffic["library_src_hash"] = _m.group(1) if _m else None      # bench.py only uses this profile for the build it was taken from
traffic["library"] = bj.get("library")
traffic["total_iterations_per_launch"] = bj["solve_stats"]["mean_iters"] * bj["config"]["batch_per_gpu"]
traffic["hbm_bytes_per_iteration"] = traffic["hbm_bytes_per_launch"] / traffic["total_iterations_per_launch"]
traffic["hbm_GBps"] = traffic["hbm_bytes_per_launch"] / (solve[3] * 1e-9) / 1e9
json.dump(traffic, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
sq = counters("sq")
json.dump({k: v["per_launch"] for k, v in sq.items()}, open(os.path.join(dst, "sq_counters.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "bench_stats.json"), os.path.join(dst, "bench_under_rocprof.json"))
print(json.dumps(traffic, indent=1)); print(json.dumps({k: v["per_launch"] for k, v in sq.items()}, indent=1))
