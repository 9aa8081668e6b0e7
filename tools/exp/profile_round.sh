#!/bin/bash
# Collect the per-round profiles on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01 [bf16x3]
# 1. rocprofv3 --kernel-trace --stats of the default bench command (default --steps / --warmup; --no-cpu-baseline only drops
#    the CPU baseline and the secondary legs, whose kernels would pollute the table)  -> per-kernel average durations
# 2. FETCH_SIZE and WRITE_SIZE in two separate --pmc passes (MI355X_MICROARCH.md, HBM section: the two
#    counters do not fit one pass; never combined with any trace domain other than the kernel trace)
# Raw output lands in gpurun_out/; tools/summarize_profile.py turns it into the files under profiles/.
set -e -o pipefail
TAG=${1:-r01}; PREC=${2:-bf16x3}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "[profile] kernel stats" >&2
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" \
    --precision $PREC --no-cpu-baseline > "$OUT/bench_under_profiler.json" 2> "$OUT/stats.err"
for C in FETCH_SIZE WRITE_SIZE; do
  echo "[profile] pmc $C" >&2
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 \
      --precision $PREC --no-cpu-baseline > /dev/null 2> "$OUT/pmc_$C.err"
done
cd "$ROOT"
python3 tools/summarize_profile.py "$OUT" "$TAG" "$PREC"
