#!/bin/bash
# Experiment: cooperative (LDS-staged) node fetch vs per-lane gathers, stand-alone tracer + bench frame, one box.
cd "$(dirname "$0")/.."
V=sunray_amd/_variants
for lib in "" $V/libsunray_hip_pad16.so $V/libsunray_hip_coop.so; do
  echo "=== tracer ${lib:-product}"
  SUNRAY_HIP_LIB=$lib timeout -k 10 200 python scripts/gpu_tracer_bench.py 2>&1 | grep -v amdgpu.ids
done
for lib in "" $V/libsunray_hip_coop.so; do
  echo "=== bench ${lib:-product}"
  SUNRAY_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Mray/s frame %.3f ms dominant %.3f other %.3f crc %s' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['other_pass_avg_ms'], d['config']['last_frame_crc32']))"
done
