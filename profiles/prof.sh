#!/bin/bash
# ONE profiling driver for the GPU box (round 4; replaces the r3_*.sh command lists):
#   bash profiles/prof.sh stats  <tag> <program> [args...]    rocprofv3 --kernel-trace --stats; keeps <tag>_kernel_stats.csv
#   bash profiles/prof.sh trace  <tag> <program> [args...]    ... and the per-dispatch distribution (profiles/nj_trace_summary.py) as <tag>_trace_summary.json
#   bash profiles/prof.sh pmc    <tag> "<counters>" <program> [args...]    one --pmc pass (kernel trace only); per-kernel mean of every counter
# <program> is started directly behind `--` (python3 script.py ... or a binary), never through a shell.
# Output under gpurun_out/${DPR_ROUND:-r5}/<tag>/ ; the per-dispatch CSVs are deleted (tens of MB).
MODE=$1; TAG=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/${DPR_ROUND:-r5}/$TAG
mkdir -p $OUT
# rocprofv3 wants /tmp as the working directory: paths relative to the repository become absolute
ARGS=()
for a in "$@"; do if [ -e "$GRAFT_REPO_ROOT/$a" ]; then ARGS+=("$GRAFT_REPO_ROOT/$a"); else ARGS+=("$a"); fi; done
set -- "${ARGS[@]}"
cd /tmp && export TMPDIR=/tmp
case $MODE in
stats|trace)
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- "$@" > $OUT/out.txt 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
    ST=$(find $OUT -name "p_kernel_stats.csv" | head -1)
    cp $ST $OUT/${TAG}_kernel_stats.csv
    if [ $MODE = trace ]; then python3 $GRAFT_REPO_ROOT/profiles/nj_trace_summary.py $OUT $OUT/${TAG}_trace_summary.json > $OUT/${TAG}_trace_summary.txt; cat $OUT/${TAG}_trace_summary.txt; fi
    python3 - <<PY
import csv
for r in list(csv.DictReader(open('$OUT/${TAG}_kernel_stats.csv')))[:14]:
    print(r['Name'][:52].ljust(52), r['Calls'].rjust(7), "%10.1f ms"%(float(r['TotalDurationNs'])/1e6), "%9.2f us avg"%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
    ;;
pmc)
    CNT=$1; shift
    rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT -o p -- "$@" > $OUT/out.txt 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
    python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(list)
for f in glob.glob('$OUT/**/p_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r['Kernel_Name'].split('(')[0].replace('void dpr::', ''), r['Counter_Name'])].append(float(r['Counter_Value']))
with open('$OUT/${TAG}_pmc.csv', 'w') as o:
    o.write('tag,kernel,counter,dispatches,mean\n')
    for (k, c), v in sorted(acc.items()):
        line = '$TAG,%s,%s,%d,%.6e' % (k[:60], c, len(v), sum(v) / len(v))
        o.write(line + '\n'); print(line)
PY
    ;;
esac
tail -3 $OUT/out.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
