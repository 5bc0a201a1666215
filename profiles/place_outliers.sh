# which launches are the outliers of the placement update kernels?  kernel trace of a 100 000-tip placement and of --add 50 000 onto
# 500 000; per kernel: the five longest dispatches with their position in the run
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6/place_outliers; mkdir -p $OUT
for job in "scratch python3 $GRAFT_REPO_ROOT/profiles/place_walks_scratch.py 100000 1000" "add python3 $GRAFT_REPO_ROOT/profiles/place_walks.py 500000 50000 1000"; do
  set -- $job; tag=$1; shift
  rm -rf $OUT/$tag; mkdir -p $OUT/$tag
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$tag -o p -- "$@" > $OUT/$tag/out.txt 2> $OUT/$tag/err.txt
  python3 - <<PY
import csv, glob
rows = []
for f in glob.glob('$OUT/$tag/**/p_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ','').replace('dpr::','')))
rows.sort()
byk = {}
for s, e, k in rows:
    byk.setdefault(k, []).append(e - s)
for k in ('place_update_kernel', 'place_update_multi_kernel', 'place_tip_edges_kernel', 'place_tip_multi_kernel'):
    v = byk.get(k)
    if not v: continue
    import statistics
    top = sorted(range(len(v)), key=lambda i: -v[i])[:5]
    print('$tag', k, 'calls', len(v), 'mean %.1f us' % (sum(v)/len(v)/1e3), 'median %.1f' % (statistics.median(v)/1e3), 'longest (dispatch index: us):', [(i, round(v[i]/1e3,1)) for i in top],
          'over 10x mean: %d launches, %.2f %% of the kernel total' % (sum(1 for x in v if x > 10*sum(v)/len(v)), 100.0*sum(x for x in v if x > 10*sum(v)/len(v))/sum(v)))
PY
  tail -1 $OUT/$tag/out.txt | cut -c1-400
  find $OUT/$tag -name "*.csv" -delete
done
