#!/bin/bash
# One profiling call on the GPU box for the build in the tree: the cfg2 trace and counter passes (tools/profile.sh), the
# default bench line, every workload's bench line, and the whole-file detection's kernel trace and counters (one stage
# at a time, so the kernels' durations add up).  Everything lands under gpurun_out/; afterwards, in the build
# container, `python tools/keep_profiles.py r03` files it under profiles/.
#   gpurun --timeout 1200 -- 'bash tools/profile_all.sh'
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out/keep gpurun_out/w gpurun_out/prof18 gpurun_out/prof19
timeout -k 10 900 bash tools/profile.sh > gpurun_out/profile_sh.log 2>&1; echo "profile rc=$?"; tail -4 gpurun_out/profile_sh.log
cp gpurun_out/prof/hbm_traffic.json gpurun_out/prof/valu_counters.json gpurun_out/keep/
cp gpurun_out/prof/hbm_traffic.json profiles/hbm_traffic.json; cp gpurun_out/prof/valu_counters.json profiles/valu_counters.json   # the lines below quote them
O=gpurun_out/w
timeout -k 10 300 python bench.py > $O/cfg2.json 2> $O/cfg2.err &&
timeout -k 10 200 python bench.py --workload cfg2_pcm16 > $O/cfg2_pcm16.json 2> $O/cfg2_pcm16.err &&
timeout -k 10 200 python bench.py --workload cfg4 > $O/cfg4.json 2> $O/cfg4.err &&
timeout -k 10 300 python bench.py --workload cfg5_pcm16 > $O/cfg5_pcm16.json 2> $O/cfg5_pcm16.err &&
timeout -k 10 300 python bench.py --workload cfg5 --seconds 3600 --steps 3 --warmup 1 > $O/cfg5_1h.json 2> $O/cfg5_1h.err &&
timeout -k 10 600 python bench.py --workload cfg5 > $O/cfg5_12h.json 2> $O/cfg5_12h.err &&
timeout -k 10 600 python bench.py --workload cfg3 > $O/cfg3.json 2> $O/cfg3.err || { echo "a workload failed"; exit 1; }
cd /tmp && export TMPDIR=/tmp
export SONO_NO_MEASURE_OVERLAP=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof18" -o p -- python3 "$ROOT/bench.py" --workload cfg5 --seconds 3600 --steps 2 --warmup 1 > "$ROOT/gpurun_out/prof18/log.txt" 2>&1 || { echo "detection trace failed"; exit 1; }
find "$ROOT/gpurun_out/prof18" -name "*kernel_trace.csv" -delete
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$ROOT/gpurun_out/prof19" -o p -- python3 "$ROOT/bench.py" --workload cfg5 --seconds 1200 --steps 1 --warmup 0 > "$ROOT/gpurun_out/prof19/log.txt" 2>&1 || { echo "detection counters failed"; exit 1; }
cd "$ROOT" && f=$(find gpurun_out/prof19 -name "*counter_collection.csv" | head -1) && python3 - "$f" > gpurun_out/prof19/summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "sono" not in k: continue
    k = k.replace("sono::(anonymous namespace)::", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
print("counters of the library's kernels in one whole-file detection of 20 min of audio (bench.py --workload cfg5 --seconds 1200 --steps 1 --warmup 0), summed over the kernel's dispatches")
for k, v in acc.items():
    print(k, "dispatches", len(calls[k])); [print("    %-20s %.4g" % (c, x)) for c, x in sorted(v.items())]
PY
find gpurun_out/prof19 -name "*.csv" -size +2M -delete
echo done
