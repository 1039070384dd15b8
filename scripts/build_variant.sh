#!/bin/bash
# dev tool: build a variant of the engine with extra -D flags on ONE kernel file
#   scripts/build_variant.sh geo_kernels.hip "-DSAGE_EXP=1" sage_slam_amd/_variants/libsage_geo1.so
set -e
cd "$(dirname "$0")/.."
SRC=$1; FLAGS=$2; OUT=$3
mkdir -p "$(dirname "$OUT")" /tmp/sage_var
OBJ=/tmp/sage_var/$(basename "$OUT" .so).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Isage_slam_amd/csrc $FLAGS -x hip -c sage_slam_amd/csrc/$SRC -o "$OBJ"
OTHERS=$(ls sage_slam_amd/csrc/_obj/*.o | grep -v "$(basename "$SRC" .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ" $OTHERS -lpthread
echo "built $OUT"
