#!/bin/bash
# GPU box: times the key configurations (64 samples per pixel) with one or more builds
# of the library.  Usage: ab_timings.sh [name ...]   (names of build_ab/librtamd_<name>.so;
# "product" = raytracer_amd/csrc/librtamd.so).  Extra environment is passed through.
cd "$(dirname "$0")/.."
CONFIGS=${CONFIGS:-"showcase-4k:3840:2160 7-reflections:640:640 8-glossy:640:640 6-soft:1024:512 3-aa:512:512 4-environmentmap:1024:1024"}
for name in "${@:-product}"; do
  if [ "$name" = product ]; then unset RTAMD_LIB; else export RTAMD_LIB=$PWD/build_ab/librtamd_$name.so; fi
  for c in $CONFIGS; do
    IFS=: read scene w h <<< "$c"
    echo -n "[$name] "
    timeout -k 10 120 python3 scripts/profile_run.py --repeat 3 --scene $scene --width $w --height $h || exit 1
  done
done
