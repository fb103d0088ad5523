#!/bin/bash
# Run ON THE GPU BOX (through gpurun): every profile directory bench.py's roofline blocks look for, collected and summarised in one call.
#   bash tools/collect_all_profiles.sh <tag>     ->  gpurun_out/profiles_<tag>/<dir>/{kernel_stats.csv,hbm_traffic.json,sq_counters.json,bench_under_rocprof.json}
# Afterwards, in the build container:  for d in gpurun_out/profiles_<tag>/*; do rm -rf profiles/$(basename $d); cp -r $d profiles/; done
TAG=${1:-r4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
run() {   # <dir> <workload> [batch]
  echo "== $1 ($2 $3)"
  bash tools/collect_profiles.sh ${TAG}_$1 "$2" $3 > gpurun_out/prof_${TAG}_$1.log 2>&1 || { echo "collect failed: $1"; tail -5 gpurun_out/prof_${TAG}_$1.log; return; }
  NMPC_PROFILE_KERNEL=${KERN:-solve_col_kernel} python3 tools/summarize_profiles.py gpurun_out/prof_${TAG}_$1 gpurun_out/profiles_$TAG/$1 > gpurun_out/prof_${TAG}_$1.sum 2>&1 || { echo "summarise failed: $1"; tail -5 gpurun_out/prof_${TAG}_$1.sum; return; }
  rm -rf gpurun_out/prof_${TAG}_$1       # the rocpd databases are large; the summaries are what is kept
  python3 -c "import json; t=json.load(open('gpurun_out/profiles_$TAG/$1/hbm_traffic.json')); print('   %.2f ms, %.0f KB per iteration and instance, %.0f GB/s' % (t['avg_duration_ms_trace_pass'], t['hbm_bytes_per_iteration']/1024, t['hbm_GBps']))"
}
run current six
run current_b16384 six 16384
run current_two two
run current_two_b1024 two 1024
run current_ten20 ten20
run current_ten ten
run current_ten_b4096 ten 4096
run current_composite composite
KERN=lidar_solve_kernel run current_lidar "tools/bench_lidar.py 4096"
