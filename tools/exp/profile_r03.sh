#!/bin/bash
# Round-3 profile collection on the GPU box (run through gpurun from the repo root):   bash tools/profile_r03.sh
# Raw output under gpurun_out/prof_r03/; the summaries (the files to copy into profiles/) under gpurun_out/prof_r03/summary/.
#   1. rocprofv3 --kernel-trace --stats of the default bench command (secondary legs off)      -> r03_fp16_kernel_stats.csv
#   2. HBM bytes per step, FETCH_SIZE / WRITE_SIZE in separate --pmc passes (tools/hbm_traffic.py) -> r03_fp16_hbm_traffic.json
#   3. SQ counters of the step's kernels, two --pmc passes (tools/pmc_sq.sh)                     -> r03_fp16_sq_counters.txt
#   4. the full default bench line (CPU baseline, modes, variants, eval path)                    -> r03_bench_fp16.json
#   5. bench.py --gpus 2 through gloo on this ONE GPU (plumbing of the N > 1 code path only)      -> r03_bench_2rank_gloo.json
# A step that times out or is killed ends the script (no GPU step is started behind a hung one).
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r03; SUM=$OUT/summary
rm -rf "$OUT"; mkdir -p "$SUM"
step() { echo "[profile] $1" >&2; shift; "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[profile] timed out / killed (rc=$rc): stopping" >&2; exit $rc; fi; return 0; }
export TMPDIR=/tmp
cd /tmp
step "kernel stats" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" \
    --no-cpu-baseline --no-live-traffic > "$OUT/bench_under_profiler.json" 2> "$OUT/stats.err"
cd "$ROOT"
cp $(find "$OUT/stats" -name "*kernel_stats.csv" | head -1) "$SUM/r03_fp16_kernel_stats.csv" 2>/dev/null
step "hbm traffic" timeout -k 10 400 python3 tools/hbm_traffic.py "$OUT/traffic" > "$OUT/traffic.txt" 2>&1
cp "$OUT/traffic/hbm_traffic.json" "$SUM/r03_fp16_hbm_traffic.json" 2>/dev/null
# the bench line below quotes this measurement (bench.py accepts the file only if its csrc hash matches the code it times)
cp "$OUT/traffic/hbm_traffic.json" "$ROOT/profiles/r03_fp16_hbm_traffic.json" 2>/dev/null
step "sq counters" timeout -k 10 700 bash tools/pmc_sq.sh r03 tools/train_steps.py --steps 3 > "$OUT/sq.log" 2>&1
cp "$ROOT/gpurun_out/sq_r03/sq_summary.txt" "$SUM/r03_fp16_sq_counters.txt" 2>/dev/null
cd /tmp
step "v1 kernel stats" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_v1" -- python3 "$ROOT/tools/train_steps.py" \
    --model v1 --steps 20 > "$OUT/v1_under_profiler.log" 2> "$OUT/stats_v1.err"
cd "$ROOT"
cp $(find "$OUT/stats_v1" -name "*kernel_stats.csv" | head -1) "$SUM/r03_v1_fp16_kernel_stats.csv" 2>/dev/null
step "v1 step" timeout -k 10 300 python3 tools/bench_v1.py fp16 > "$SUM/r03_v1_step_kernels.txt" 2> "$OUT/v1_step.err"
step "bench" timeout -k 10 500 python3 bench.py --no-live-traffic > "$SUM/r03_bench_fp16.json" 2> "$OUT/bench.err"
step "bench 2 ranks (gloo, one GPU)" env NRMS_DIST_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > "$SUM/r03_bench_2rank_gloo.json" 2> "$OUT/bench2.err"
step "bench 2 ranks sharded (gloo, one GPU)" env NRMS_DIST_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 5 --warmup 2 --grad-sync sharded > "$SUM/r03_bench_2rank_gloo_sharded.json" 2> "$OUT/bench2s.err"
ls -la "$SUM" >&2
