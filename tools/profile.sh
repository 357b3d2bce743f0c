#!/bin/bash
# Profile bench.py on the GPU box: kernel-trace stats, then HBM read / write counters in two
# separate PMC passes (TCC has 4 slots; FETCH_SIZE takes 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md).
# Usage (inside gpurun):  bash tools/profile.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="${@:---steps 5 --warmup 1 --no-cpu-baseline}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err" || { echo "trace failed"; tail -5 "$OUT/trace.err"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_pmc_fetch.json" 2> "$OUT/pmc_fetch.err" || { echo "pmc fetch failed"; tail -5 "$OUT/pmc_fetch.err"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_pmc_write.json" 2> "$OUT/pmc_write.err" || { echo "pmc write failed"; tail -5 "$OUT/pmc_write.err"; exit 1; }
find "$OUT" -name "*.csv" | head -30
du -sh "$OUT"
