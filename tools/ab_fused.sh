#!/bin/bash
# A/B the fused program kernel: rays per thread.
for v in 0 1; do
  echo "== PXF_PROGRAM_V=$v"
  PXF_PROGRAM_V=$v timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('eager %.4g rays/s | fused %.4g rays/s  launch %.3f ms  %.0f GB/s' % (j['value'], j['fused']['value'], j['fused']['avg_launch_ms'], j['fused']['GBps']))
" || exit 1
done
