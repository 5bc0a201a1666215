#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd $REPO
step() {
    local lim=$1 name=$2; shift 2
    echo "=== $name" | tee -a $OUT/steps3.log
    timeout -k 10 $lim "$@" > $OUT/$name.out 2> $OUT/$name.err
    local rc=$?
    echo "rc=$rc" | tee -a $OUT/steps3.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name killed at its limit: stopping" | tee -a $OUT/steps3.log; exit 1; fi
}
step 300 tests_nj python -m pytest tests/test_gpu_nj.py -m gpu -x -q
tail -5 $OUT/tests_nj.out
step 600 nj_worstcase2 python profiles/nj_worstcase.py 30000 10000
cat $OUT/nj_worstcase2.out
step 150 pmc_sq bash profiles/pmc_njp.sh sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE"
cat $OUT/pmc_sq.out
step 150 pmc_tcc bash profiles/pmc_njp.sh tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
cat $OUT/pmc_tcc.out
step 150 pmc_fetch bash profiles/pmc_njp.sh fetch "FETCH_SIZE"
cat $OUT/pmc_fetch.out
step 150 pmc_write bash profiles/pmc_njp.sh write "WRITE_SIZE"
cat $OUT/pmc_write.out
cd /tmp && export TMPDIR=/tmp
for plan in 2 0; do
  step 150 vworld8_plan$plan rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vw8_$plan -o v -- python3 $REPO/profiles/njs_vworld_stats.py 30000 10000 256 8 $plan
  cat $OUT/vworld8_plan$plan.out
  python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/vw8_$plan/**/v_kernel_stats.csv", recursive=True) + glob.glob("$OUT/vw8_$plan/v_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r['Name'][:60].ljust(60), r['Calls'].rjust(7), "%10.1f ms" % (float(r['TotalDurationNs'])/1e6), "%9.2f us avg" % (float(r['AverageNs'])/1e3), r['Percentage'])
    break
PY
  find $OUT/vw8_$plan -name "*kernel_trace.csv" -delete
done
