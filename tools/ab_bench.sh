#!/bin/bash
# Development aid (GPU box): A/B the library variants under variants/ in ONE session (run-to-run noise between boxes is several %).
#   [WORKLOAD=two] [BATCH=16384] [PIN=3|0] [REPS=3] bash tools/ab_bench.sh name1 name2 ...
# PIN=3 (default) pins the column kernel's throughput shape; PIN=0 leaves the choice to the library (latency shapes for small batches).
W=${WORKLOAD:-six}
BA=${BATCH:-0}
PIN=${PIN:-3}
for rep in $(seq 1 ${REPS:-3}); do
  for v in "$@"; do
    NMPC_SO=$PWD/variants/libnmpc_$v.so NMPC_KERNEL=$PIN timeout -k 10 200 python bench.py --workload $W --batch $BA --cpu-sample 0 --closed-loop 0 --sweep 0 --steps 5 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('AB $W $v rep $rep', round(d['value']), round(d['ms_per_step'],2), d['solve_stats']['mean_iters'], d['solve_stats']['converged_frac'])"
  done
done
