#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for f in 0 16 32 64 112; do echo "flags=$f"; DPR_NJP_FLAGS=$f timeout -k 10 200 python profiles/nj_kt.py 100000 10000 4000 4 2>&1 | tail -1 | cut -c60-330; done
