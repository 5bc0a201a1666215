#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for n in 5000 8000; do echo "n=$n"; timeout -k 10 300 python profiles/cli_overhead.py $n 2000 3 2>&1 | grep plain | cut -c1-100; done
echo "n=12000 stream"; DPR_NJ_MODE=stream timeout -k 10 300 python profiles/cli_overhead.py 12000 2000 3 2>&1 | grep plain | cut -c1-100
echo "n=12000 nograph"; DPR_NJ_NOGRAPH=1 timeout -k 10 300 python profiles/cli_overhead.py 12000 2000 3 2>&1 | grep plain | cut -c1-100
