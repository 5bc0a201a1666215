#!/bin/bash
# usage (GPU box): bash profiles/cli_lib_ab.sh <other libdipper_hip.so>  -> whole-CLI wall time at 30 000 x 10 000 with the built library and with another build
python3 - <<PY
import numpy as np, sys
sys.path.insert(0, ".")
from tests import _util
import bench
seqs = bench.make_input(30000, 10000, 1)
_util.write_fasta("/tmp/in.fa", ["T%d" % (i + 1) for i in range(30000)], seqs, width=0)
PY
cp dipper_amd/libdipper_hip.so /tmp/new.so
for round in 1 2; do
  for lib in /tmp/new.so $1; do
    cp $lib dipper_amd/libdipper_hip.so
    for i in 1 2 3 4 5 6; do
      s=$(date +%s%N); ./dipper_amd/bin/dipper -i m -I /tmp/in.fa -O /tmp/o.nwk -m 2 -d 2 >/dev/null 2>&1; e=$(date +%s%N); echo $(( (e - s) / 1000000 ))
    done | tr '\n' ' '
    echo " <- $lib"
  done
done
cp /tmp/new.so dipper_amd/libdipper_hip.so
