#!/bin/bash
# A/B of the order of the stored rows of the structured P2 system (PHX_P2S_TILE: 0 = by length alone, T = tiles of T^3
# fine points first; PHX_P2S_LENQ: length step of the key): the stored rows alone (PHX_SELL_EXP=0) and the padded entry
# count.  Same box.
set -e
n=${1:-256}
for t in ${TILES_AB:-0 4 8 16}; do
  for q in ${LENQ_AB:-4}; do
    echo -n "tile $t lenq $q: "
    PHX_P2S_TILE=$t PHX_P2S_LENQ=$q PHX_SELL_EXP=0 timeout -k 10 300 python tools/r04/spmv_parts.py p2:$n 20 2>&1 | tail -1 | cut -c1-140
  done
done
