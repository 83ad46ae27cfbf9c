# usage: bash tools/kb_sweep.sh lib.so [lib.so ...] -- times the standard set of shapes with each library
for L in "$@"; do
  timeout -k 10 60 tools/kbench $L 1024 768 3600 10 || exit 1
  timeout -k 10 60 tools/kbench $L 512 256 3600 10 || exit 1
  timeout -k 10 60 tools/kbench $L 256 128 3600 10 || exit 1
  timeout -k 10 60 tools/kbench $L 2048 1536 3600 10 || exit 1
  timeout -k 10 60 tools/kbench $L 4096 3891 600 10 || exit 1
done
