#!/bin/bash
# A/B bench variants in one gpurun call: grid cap per CU (PXF_BPC) x {default, non-temporal} library.
ROOT=${GRAFT_REPO_ROOT:-.}
for lib in "" "$ROOT/build/libpxf_hip_nt.so"; do
  for bpc in 8 0 16; do
    echo "== PXF_LIB=${lib##*/} PXF_BPC=$bpc"
    PXF_LIB=$lib PXF_BPC=$bpc timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4g rays/s  ms/step %.3f  chain %.0f GB/s' % (j['value'], j['ms_per_step'], j['chain_GBps']))
print('  '+'  '.join('%s %.3f(%.0f)' % (k['kernel'][:6], k['avg_ms'], k['GBps']) for k in j['kernels']))

" || exit 1
  done
done
