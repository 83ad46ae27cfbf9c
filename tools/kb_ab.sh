# usage: bash tools/kb_ab.sh <fft> <noverlap> <seconds> libA.so libB.so [...] -- times the libraries in turn, three rounds, on one box
N=$1; V=$2; S=$3; shift 3
for round in 1 2 3; do
  for L in "$@"; do
    timeout -k 10 60 tools/kbench $L $N $V $S 10 || exit 1
  done
done
