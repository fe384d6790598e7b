#!/bin/bash
# A/B of run-time switches of the product library on one box: scripts/ab_env.sh "VAR=a" "VAR=b" ... (two rounds each)
cd "$(dirname "$0")/.."
for round in 1 2; do
  for v in "$@"; do
    env $v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-24s round $round: %.1f Mray/s  frame %.3f ms  final %.3f  ris %.3f  crc %s' % ('$v', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['other_pass']['avg_launch_ms'], d['config']['last_frame_crc32']))"
  done
done
