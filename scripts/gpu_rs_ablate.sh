#!/bin/bash
# Timing ablations of conv_rs.hip (alt builds from scripts/build_alt.sh rsabl<N> conv_rs.hip -DSNAP_RS_ABLATE=<N>)
export PYTHONPATH=.
ARGS="${RS_ARGS:---no-res --stats none}"
for a in ${RS_ABL:-0 1 2 4 8 16 6 7}; do
  if [ "$a" = 0 ]; then unset SNAP_HIP_LIB; else export SNAP_HIP_LIB=snap_amd/lib/alt_rsabl$a/libsnap_hip.so; fi
  for sh in ${RS_SHAPES:-0 1 2}; do
    echo "abl=$a $(timeout 100 python tools/rs_bench.py --only $sh $ARGS 2>/dev/null | tail -1)"
  done
done
