#!/bin/bash
# Counter passes over tools/kbench (run on the GPU box): bash tools/pmc_kbench.sh <name> <counters...>
# One rocprofv3 --pmc run per call; results in gpurun_out/pmc_<name>/.  KB_LIB / KB_ARGS (environment) name another
# library build and another case ("fft noverlap seconds reps"; default: the headline case).
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
name=$1; shift
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d "$ROOT/gpurun_out/pmc_$name" -o $name -- "$ROOT/tools/kbench" "${KB_LIB:-$ROOT/spectrogram_amd/lib/libsonostft.so}" ${KB_ARGS:-1024 768 3600 3} > "$ROOT/gpurun_out/pmc_$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$ROOT/gpurun_out/pmc_$name.log"; exit 1; }
python3 - "$ROOT/gpurun_out/pmc_$name/${name}_counter_collection.csv" <<'PY'
import csv, collections, sys
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'stft' in r['Kernel_Name']:
        d[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(d.items()):
    print(f"{k:40s} {sum(v) / len(v):18.1f}  ({len(v)} dispatches)")
PY
