#!/bin/bash
# A/B of two builds of the library (IQLHIP_LIB=path), interleaved: steady bench (5 000 steps) and the driver's command.
# usage: tools/ab_libs.sh OUT libA.so libB.so [runs]
out=$1; a=$2; b=$3; n=${4:-3}
BENCH_ARGS="--steps 5000 --warmup 200 --repeats 1" tools/ab_bench20.sh "$out.steady" "IQLHIP_LIB=$a" "IQLHIP_LIB=$b" "$n" > /dev/null || exit 1
tools/ab_bench20.sh "$out.n20" "IQLHIP_LIB=$a" "IQLHIP_LIB=$b" "$n" > /dev/null || exit 1
cat "$out.steady" "$out.n20" > "$out"; rm -f "$out.steady" "$out.n20"; cat "$out"
