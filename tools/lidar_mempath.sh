#!/bin/bash
# Run ON THE GPU BOX (through gpurun): memory-path counters of the LIDAR solve kernel (texture addresser / L1 vector cache / L2 requests),
# one rocprofv3 --pmc pass per counter group, plus the launch time at batches that put 0.5 / 1 / 2 waves on a SIMD.
#   bash tools/lidar_mempath.sh <tag>      ->  gpurun_out/mem_<tag>/{g1,g2,g3,g4}/...  and  gpurun_out/mem_<tag>/sizes.log
set -e
TAG=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/mem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for b in 512 1024 2048 4096; do python3 $ROOT/tools/bench_lidar.py $b 2>> $OUT/sizes.log > /dev/null; done
G1="TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
G2="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum"
G3="TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
G4="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
i=0
for G in "$G1" "$G2" "$G3" "$G4"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $G -d $OUT/g$i -o run -- python3 $ROOT/tools/bench_lidar.py 4096 > $OUT/g$i.json 2> $OUT/g$i.err || echo "group $i failed" >> $OUT/sizes.log
done
find $OUT -name "*counter_collection.csv" | head
