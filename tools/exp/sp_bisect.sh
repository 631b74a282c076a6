#!/bin/bash
# which class of split-kernel instantiations breaks a test?  usage: sp_bisect.sh <pytest node id>
T=$1
for m in 127 0 1 2 4 8 16 32 64; do
  r=$(LIME_SP_MASK=$m timeout -k 5 300 python -m pytest "$T" -x -q 2>&1 | tail -1)
  echo "mask $m: $r"
done
