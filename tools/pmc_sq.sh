#!/bin/bash
# SQ-level counters (MFMA busy, LDS/VMEM activity, waits) per kernel, two --pmc passes.
#   bash tools/pmc_sq.sh <tag> [script.py args...]     -> gpurun_out/sq_<tag>/sq_summary.txt
# (default program: bench.py --steps 2 --warmup 1 --no-cpu-baseline)
set -e -o pipefail
TAG=${1:-x}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/sq_$TAG
shift || true
if [ $# -gt 0 ]; then PROG=("$ROOT/$1" "${@:2}"); else PROG=("$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline); fi
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS"
i=0
for SET in "$A" "$B"; do
  i=$((i+1)); echo "[pmc] pass $i" >&2
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/p$i" -- python3 "${PROG[@]}" > /dev/null 2> "$OUT/p$i.err"
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float)); dur = {}
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(lambda: defaultdict(float)); names = {}; t = {}
    for r in csv.DictReader(open(f)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
        t[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    big = {}
    for d, n in names.items():                      # the longest launch of each kernel
        if n not in big or t[d] > t[big[n]]: big[n] = d
    for n, d in big.items():
        for c, v in per[d].items(): agg[n][c] = v
        dur[n] = t[d]
with open(os.path.join(out, "sq_summary.txt"), "w") as f:
    for n in sorted(dur, key=lambda k: -dur[k])[:12]:
        f.write("%s  (%.3f ms under pmc)\n" % (n, dur[n] / 1e6))
        for c, v in sorted(agg[n].items()): f.write("    %-32s %.4g\n" % (c, v))
print(open(os.path.join(out, "sq_summary.txt")).read())
PY
