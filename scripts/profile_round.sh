#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats + PMC passes of the bench command, outputs under gpurun_out/prof_<tag>/.
# usage: scripts/profile_round.sh TAG     (then: python scripts/summarise_profiles.py TAG  ->  profiles/<TAG>_*)
tag=${1:-r02}
repo=${GRAFT_REPO_ROOT:-$PWD}
out=$repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $repo/bench.py --no-cpu-baseline --no-pmc"      # the program itself after `--`; bench.py's own live PMC children are off
pass() {   # name, steps, warmup, rocprofv3 arguments...
  name=$1; steps=$2; warm=$3; shift 3
  timeout -k 10 200 rocprofv3 "$@" --output-format csv -d /tmp/p_$name -o p -- $B --steps $steps --warmup $warm > $out/$name.log 2>&1 || { echo "$name FAILED"; tail -3 $out/$name.log; return 1; }
}
pass stats 20 5 --kernel-trace --stats && find /tmp/p_stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \; && tail -1 $out/stats.log | cut -c1-200
SUNRAY_BENCH_PIPELINE=0 pass stats_seq 20 5 --kernel-trace --stats && find /tmp/p_stats_seq -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_sequential.csv \;
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  pass pmc$i 5 3 --pmc $grp && find /tmp/p_pmc$i -name "*counter_collection.csv" -exec cp {} $out/pmc${i}_counter_collection.csv \;
done
ls $out
