#!/bin/bash
# A/B an environment knob on the real bench, alternating runs on the same box (e.g. PXF_NT_MIN_RAYS,
# PXF_PROGRAM_NO_CHAIN; the round-1 grid knobs PXF_BPC / PXF_XCD_REMAP no longer reach the ray kernels, which have
# no grid-stride loop since round 3).
# usage: tools/ab_env.sh VAR val1 val2 [rounds]
VAR=$1; A=$2; B=$3; R=${4:-2}
for r in $(seq 1 $R); do
  for v in $A $B; do
    echo -n "$VAR=$v : "
    env $VAR=$v timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-fused 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.4g rays/s  chain %.0f GB/s | ' % (j['value'], j['chain_GBps']) + '  '.join('%s %.3f(%.0f)' % (k['kernel'][:6], k['avg_ms'], k['GBps']) for k in j['kernels']))"
  done
done
