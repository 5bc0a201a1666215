#!/bin/bash
# experiment: waves-per-SIMD hint on the large-shape post kernel of the pruned NJ, NJ at 100 000 x 10 000
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd $REPO/dipper_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wall -Wno-unused-result -Wno-pass-failed"
for w in 1 3 4; do
  /opt/rocm/bin/hipcc $FLAGS -DDPR_NJP_BIG_WAVES=$w -c njp.hip -o njp.o 2> $OUT/build_w$w.err
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdipper_hip.so ctx.o nj.o njs.o njp.o msa.o mash.o mash_index.o place.o dc.o exact.o -ldl
  echo "== waves $w"
  (cd $REPO && timeout -k 10 200 python profiles/nj_big.py 100000 10000 2) 2>&1 | tail -2
done
