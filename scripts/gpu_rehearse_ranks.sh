#!/bin/bash
# Rehearsal of the N-rank bench on a ONE-GPU box (all ranks on cuda:0, gather over gloo): the strip / halo / gather code path
# end to end through bench.py's self-launch; every N must print the 1-GPU frame CRC.   usage: scripts/gpu_rehearse_ranks.sh [N ...]
cd "$(dirname "$0")/.."
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: crc %s  %.1f Mray/s  %.3f ms  rays/frame %.0f  %s' % (d['config']['last_frame_crc32'], d['value'], d['ms_per_step'], d['config']['rays_per_frame'], d['config']['parallelism'][:110]))"; }
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | show "N=1"
for n in ${@:-2 4}; do
  for axis in cols rows; do
    SUNRAY_BENCH_ONE_DEVICE=1 SUNRAY_BENCH_AXIS=$axis timeout -k 10 400 python bench.py --gpus $n --steps 6 --warmup 2 2>gpurun_out/reh_$n_$axis.err | tail -1 | show "N=$n $axis" || tail -5 gpurun_out/reh_$n_$axis.err
  done
done
