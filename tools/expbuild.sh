#!/bin/bash
# Builds an experimental variant of libsonostft.so with extra compiler flags:
#   tools/expbuild.sh <name> [-DSONO_...]   ->  gpurun_exp/lib_<name>.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; shift
mkdir -p "$ROOT/gpurun_exp/obj_$name"
cd "$ROOT/spectrogram_amd/csrc"
for f in sono_tile_u8.hip sono_tile_mag.hip sono_tile_band.hip sono_tile_survey.hip sono_tile.hip sono_kernels.hip sono_detect.hip sono_capi.hip sono_capi_detect.hip sono_capi_rois.hip sono_capi_noalloc.hip sono_stream.hip sono_measure_dev.hip; do
  extra=""; [ $f = sono_measure_dev.hip ] && extra="-ffp-contract=off"     # as spectrogram_amd/build.py
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize $extra "$@" -c $f -o "$ROOT/gpurun_exp/obj_$name/$f.o" &
done
for c in sono_host sono_framing sono_measure; do g++ -O2 -std=c++17 -fPIC -ffp-contract=off -c $c.cpp -o "$ROOT/gpurun_exp/obj_$name/$c.o" & done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o "$ROOT/gpurun_exp/lib_$name.so" "$ROOT/gpurun_exp/obj_$name"/*.o
rm -rf "$ROOT/gpurun_exp/obj_$name"
echo "built gpurun_exp/lib_$name.so"
