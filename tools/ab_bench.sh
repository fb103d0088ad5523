#!/bin/bash
# Development aid (GPU box): A/B the library variants under variants/ in ONE session (run-to-run noise between boxes is several %).
#   [WORKLOAD=two] [BATCH=16384] bash tools/ab_bench.sh name1 name2 ...
W=${WORKLOAD:-six}
BA=${BATCH:-0}
for rep in 1 2 3; do
  for v in "$@"; do
    NMPC_SO=$PWD/variants/libnmpc_$v.so NMPC_KERNEL=3 timeout -k 10 200 python bench.py --workload $W --batch $BA --cpu-sample 0 --closed-loop 0 --sweep 0 --steps 5 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('AB $v rep $rep', round(d['value']), round(d['ms_per_step'],2))"
  done
done
