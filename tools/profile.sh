#!/bin/bash
# Profiles bench.py's timed kernel with rocprofv3: one kernel-trace pass and
# separate counter passes (never combined with tracing).  Run on the GPU box:
#   bash tools/profile.sh [extra bench args]
# Summaries land in gpurun_out/prof/; copy what should be kept into profiles/.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-end-to-end --no-latency $*"
TRACE_BENCH="python3 $ROOT/bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-end-to-end --no-latency $*"    # the default step count: its average is the one bench.py reports

timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $TRACE_BENCH > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }

pass() {
    name=$1; shift
    timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o $name -- $BENCH > "$OUT/$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE
python3 "$ROOT/tools/profile_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep the output small: the per-dispatch CSVs of the torch synthesis kernels are not needed
for f in $(find "$OUT" -name "*.csv"); do
    case "$f" in
        *kernel_stats.csv|*agent_info.csv|*domain_stats.csv) ;;
        *) { head -1 "$f"; grep stft_ "$f"; } > "$f.tmp"; mv "$f.tmp" "$f" ;;
    esac
done
