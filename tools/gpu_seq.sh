#!/bin/bash
# Run GPU steps one after another on the box; a step that TIMES OUT or is KILLED (exit 124 / 137) ends the call -- no further
# GPU step is started behind a hung one -- while an ordinary failure (a failing test) does not.
#   bash tools/gpu_seq.sh "cmd 1" "cmd 2" ...
for c in "$@"; do
  echo "[gpu_seq] $c" >&2
  bash -o pipefail -c "$c"; rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[gpu_seq] step timed out / was killed (rc=$rc): stopping" >&2; exit $rc; fi
done
exit 0
