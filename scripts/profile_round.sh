#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats + PMC passes of the bench command, outputs under gpurun_out/prof_<tag>/.
# usage: scripts/profile_round.sh TAG
tag=${1:-r01}
repo=${GRAFT_REPO_ROOT:-$PWD}
out=$repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $repo/bench.py --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o s -- $B --steps 20 --warmup 3 > $out/stats.log 2>&1 && \
  find /tmp/p_stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \; && tail -1 $out/stats.log | cut -c1-300 && \
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -o f -- $B --steps 5 --warmup 2 > $out/fetch.log 2>&1 && \
  find /tmp/p_fetch -name "*counter_collection.csv" -exec cp {} $out/fetch_counter_collection.csv \; && \
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -o w -- $B --steps 5 --warmup 2 > $out/write.log 2>&1 && \
  find /tmp/p_write -name "*counter_collection.csv" -exec cp {} $out/write_counter_collection.csv \; && \
timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/p_sq1 -o a -- $B --steps 5 --warmup 2 > $out/sq1.log 2>&1 && \
  find /tmp/p_sq1 -name "*counter_collection.csv" -exec cp {} $out/sq1_counter_collection.csv \; && \
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/p_sq2 -o b -- $B --steps 5 --warmup 2 > $out/sq2.log 2>&1 && \
  find /tmp/p_sq2 -name "*counter_collection.csv" -exec cp {} $out/sq2_counter_collection.csv \;
ls -la $out
cd /tmp
timeout -k 10 100 rocprofv3 --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_ta1 -o t -- $B --steps 5 --warmup 2 > $out/ta1.log 2>&1 && \
  find /tmp/p_ta1 -name "*counter_collection.csv" -exec cp {} $out/ta1_counter_collection.csv \; && \
timeout -k 10 100 rocprofv3 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum --output-format csv -d /tmp/p_ta2 -o t -- $B --steps 5 --warmup 2 > $out/ta2.log 2>&1 && \
  find /tmp/p_ta2 -name "*counter_collection.csv" -exec cp {} $out/ta2_counter_collection.csv \;
ls $out | wc -l
