#!/bin/bash
# A what-if build of ONE source with extra flags, linked with the product's other objects:
#   tools/build_alt.sh <name> <source.hip> <extra hipcc flags...>   ->  commonscenes_amd/alt/libcommonscenes_hip_<name>.so
# Run timing A/Bs with CS_LIB_PATH=<that file> (lib.load).  The product library must be built first (objects in build/).
set -e
NAME=$1; SRC=$2; shift 2
ROOT=$(cd $(dirname $0)/.. && pwd)
PKG=$ROOT/commonscenes_amd
mkdir -p $PKG/alt
OBJ=$PKG/alt/${NAME}_$(basename $SRC .hip).o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $PKG/csrc/$SRC -o $OBJ
OBJS=""
for o in $PKG/build/*.o; do
  if [ "$(basename $o)" != "$(basename $SRC .hip).o" ]; then OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $OBJ -o $PKG/alt/libcommonscenes_hip_${NAME}.so
echo $PKG/alt/libcommonscenes_hip_${NAME}.so
