#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 passes for one bench workload.
#   bash tools/collect_profiles.sh <tag> [workload] [batch]    ->  gpurun_out/prof_<tag>/{stats,fetch,write,sq}/...   (workload: six | two | ten20 | ten | composite; batch: instances per GPU, default the workload's)
#   bash tools/collect_profiles.sh <tag> "tools/bench_lidar.py 4096"     (another command that prints one bench-style JSON line)
# Counter passes are separate from the trace pass and from each other (FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -e
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 5 --warmup 1 --cpu-sample 0 --closed-loop 0 --sweep 0"
case "$2" in
  ""|six) ;;
  two|ten20|ten|composite) ARGS="$ARGS --workload $2" ;;
  *) ARGS="$ROOT/$2" ;;
esac
if [ -n "$3" ]; then ARGS="$ARGS --batch $3"; fi
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python3 $ARGS > $OUT/bench_stats.json 2> $OUT/stats.err
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o run -- python3 $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o run -- python3 $ARGS > $OUT/bench_write.json 2> $OUT/write.err
timeout -k 10 280 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES -d $OUT/sq -o run -- python3 $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err
find $OUT -name "*.csv" | head -20
