#!/bin/bash
# Back-to-back launches of the headline case for SECONDS of wall time (default 8), with the card's clocks and power
# sampled once a second beside it.  Run on the GPU box: bash tools/sustained.sh [seconds]; output in gpurun_out/sustained/.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/sustained"
S=${1:-8}
mkdir -p "$OUT"
( for i in $(seq 1 $((S + 6))); do date +%s.%N; rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|mclk|power"; sleep 1; done ) > "$OUT/clocks.log" 2>&1 &
SMI=$!
timeout -k 10 $((S + 60)) "$ROOT/tools/vbench" sustained_u8_1024 $S > "$OUT/sustained.json"
RC=$?
wait $SMI
cat "$OUT/sustained.json" | cut -c1-700
grep -ic sclk "$OUT/clocks.log"
exit $RC
