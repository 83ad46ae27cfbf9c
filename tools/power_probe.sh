#!/bin/bash
# sustained headline launches with the card's power / clock read beside them
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $ROOT/gpurun_out/power
( sleep 3; for i in 1 2 3; do rocm-smi --showpower --showclocks --showtemp --showuse 2>&1; sleep 1.5; done ) > $ROOT/gpurun_out/power/smi_load.txt 2>&1 &
SMI=$!
timeout -k 10 60 $ROOT/tools/vbench sustained_u8_1024 10 > $ROOT/gpurun_out/power/sustained.json
wait $SMI
rocm-smi --showpower --showclocks --showmaxpower 2>&1 > $ROOT/gpurun_out/power/smi_idle.txt
grep -iE "power|sclk|temp|use|fclk|socclk" $ROOT/gpurun_out/power/smi_load.txt | head -40
echo ---- idle
grep -iE "power|sclk" $ROOT/gpurun_out/power/smi_idle.txt | head
cut -c1-300 $ROOT/gpurun_out/power/sustained.json
