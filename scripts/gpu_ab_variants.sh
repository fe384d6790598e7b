#!/bin/bash
# A/B of library variants on one box: scripts/gpu_ab_variants.sh NAME... ("product" = the in-tree library); two rounds of the
# stand-alone tracer bench and the bench frame (sequential passes, no PMC) per variant.
cd "$(dirname "$0")/.."
V=sunray_amd/_variants
for round in 1 2; do
for name in "$@"; do
  lib=""; [ "$name" != "product" ] && lib=$V/libsunray_hip_$name.so
  echo "=== $name (round $round)"
  [ $round = 1 ] && [ -z "$SKIP_TRACER" ] && SUNRAY_HIP_LIB=$lib timeout -k 10 200 python scripts/gpu_tracer_bench.py 2>&1 | grep -E "Mray/s"
  for pipe in 0 1; do
  SUNRAY_HIP_LIB=$lib SUNRAY_BENCH_PIPELINE=$pipe timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  frames in flight %d: %.1f Mray/s frame %.3f ms  %s %.3f  %s %.3f crc %s' % (d['config']['frames_in_flight'], d['value'], d['ms_per_step'], r['kernel'][:5], r['avg_launch_ms'], r['other_pass']['kernel'][:5], r['other_pass']['avg_launch_ms'], d['config']['last_frame_crc32']))"
  done
done
done
