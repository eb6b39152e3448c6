#!/bin/bash
# Builds a variant of librtamd.so with extra hipcc flags into build_ab/librtamd_<name>.so
# (select it at run time with RTAMD_LIB=<path>).  Usage: build_variant.sh <name> <flags...>
set -e
cd "$(dirname "$0")/../raytracer_amd/csrc"
NAME=$1; shift
OUT=../../build_ab
mkdir -p $OUT/obj_$NAME
make -s front/blob.o >/dev/null
for f in front/*.c; do o=front/$(basename ${f%.c}).o; [ -f $o ] || make -s $o; done
for f in device/*.hip; do
  /opt/rocm/bin/hipcc -O3 -ffp-contract=off -fPIC -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -Wall \
    -Wno-unused-parameter -I../../include -Ifront -Idevice "$@" -c $f -o $OUT/obj_$NAME/$(basename ${f%.hip}).o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/librtamd_$NAME.so front/*.o $OUT/obj_$NAME/*.o -lm
echo built $OUT/librtamd_$NAME.so
