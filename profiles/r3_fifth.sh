#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd $REPO
step() {
    local lim=$1 name=$2; shift 2
    echo "=== $name" | tee -a $OUT/steps5.log
    local t0=$(date +%s)
    timeout -k 10 $lim "$@" > $OUT/$name.out 2> $OUT/$name.err
    local rc=$?
    echo "rc=$rc wall=$(( $(date +%s) - t0 ))s" | tee -a $OUT/steps5.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name killed at its limit: stopping" | tee -a $OUT/steps5.log; exit 1; fi
}
step 600 tests_a python -m pytest tests/test_gpu_sharded.py tests/test_gpu_multiproc.py tests/test_gpu_nj.py tests/test_gpu_mash_place.py tests/test_gpu_dc.py -m gpu -x -q
tail -4 $OUT/tests_a.out
step 600 tests_b python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_accuracy.py -m gpu -x -q
tail -4 $OUT/tests_b.out
step 300 nj_worstcase3 python profiles/nj_worstcase.py 30000 10000 2e-05,const,ints,random
cat $OUT/nj_worstcase3.out
cd /tmp && export TMPDIR=/tmp
step 150 vworld8_mailbox rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vw8c -o v -- python3 $REPO/profiles/njs_vworld_stats.py 30000 10000 256 8 2
cat $OUT/vworld8_mailbox.out
step 150 vworld8_peer rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vw8d -o v -- python3 $REPO/profiles/njs_vworld_stats.py 100000 10000 24 8 1
cat $OUT/vworld8_peer.out
step 400 add_mash rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/addmash -o v -- python3 $REPO/profiles/add_bench.py 500000 50000 3000 r
cat $OUT/add_mash.out
step 300 add_msa rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/addmsa -o v -- python3 $REPO/profiles/add_bench.py 500000 50000 1000 m
cat $OUT/add_msa.out
find $OUT -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv, glob
for tag in ("vw8c", "vw8d", "addmash", "addmsa"):
    for f in glob.glob("$OUT/%s/**/v_kernel_stats.csv" % tag, recursive=True) + glob.glob("$OUT/%s/v_kernel_stats.csv" % tag):
        print("==", tag)
        for r in list(csv.DictReader(open(f)))[:7]:
            print(r['Name'][:64].ljust(64), r['Calls'].rjust(7), "%10.1f ms" % (float(r['TotalDurationNs'])/1e6), "%9.2f us avg" % (float(r['AverageNs'])/1e3), r['Percentage'])
        break
PY
