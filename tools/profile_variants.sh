#!/bin/bash
# Profiles the cases of tools/vbench one by one with rocprofv3: a kernel-trace pass, a counter pass
# and a GRBM pass (counters are never combined with tracing) and, for the STFT cases, the two HBM traffic passes.
# Run on the GPU box:   bash tools/profile_variants.sh [case ...]      (default: every case of `vbench list`)
# One summary per case lands in gpurun_out/variants/<case>_summary.txt; copy what should be kept into profiles/.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/variants"
VB="$ROOT/tools/vbench"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CASES="$*"
[ -z "$CASES" ] && CASES=$("$VB" list | grep -v sustained)
for c in $CASES; do
    D="$OUT/$c"
    mkdir -p "$D"
    echo "== $c"
    timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -o t -- "$VB" $c 10 > "$D/trace.log" 2>&1 || { echo "trace pass of $c failed"; tail -5 "$D/trace.log"; exit 1; }
    grep -h '^{' "$D/trace.log" | tail -1 > "$D/vbench.json"
    timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS \
        --output-format csv -d "$D/sq" -o sq -- "$VB" $c 4 > "$D/sq.log" 2>&1 || { echo "counter pass of $c failed"; tail -5 "$D/sq.log"; exit 1; }
    timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$D/grbm" -o grbm -- "$VB" $c 4 > "$D/grbm.log" 2>&1 || { echo "GRBM pass of $c failed"; tail -5 "$D/grbm.log"; exit 1; }
    case "$c" in
        u8_*|features_*|band_1024_50|mag_*|filterbank_*|survey_*)
            timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/fetch" -o fetch -- "$VB" $c 4 > "$D/fetch.log" 2>&1 || { echo "fetch pass of $c failed"; exit 1; }
            timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$D/write" -o write -- "$VB" $c 4 > "$D/write.log" 2>&1 || { echo "write pass of $c failed"; exit 1; }
            ;;
    esac
    python3 "$ROOT/tools/variant_summary.py" "$D" $c > "$OUT/${c}_summary.txt" 2>&1
    tail -4 "$OUT/${c}_summary.txt"
    rm -rf "$D/trace" "$D/sq" "$D/grbm" "$D/fetch" "$D/write"          # the summaries carry what is kept
done
