set -e
cd $GRAFT_REPO_ROOT
G=tools/bin/gen_synth
$G --tips 20000 --sites 1000 --seed 3 --fasta /tmp/a20k.fa >/dev/null
B=dipper_amd/bin/dipper
for dev in "" "--devices 0,0" "--devices 0,0,0,0"; do
  echo "== placement m=1 $dev"; ( time $B -i m -I /tmp/a20k.fa -m 1 -d 2 -O /tmp/p_"${dev// /_}".nwk $dev ) 2>&1 | grep -E "Ranks|real|Tree Operation|Distance Operation|ERROR|rror"
done
md5sum /tmp/p_*.nwk
for dev in "" "--devices 0,0,0"; do
  echo "== dc m=3 $dev"; ( time DPR_COMM_WINDOW_MB=1 $B -i m -I /tmp/a20k.fa -m 3 -d 2 -O /tmp/d_"${dev// /_}".nwk $dev ) 2>&1 | grep -E "Ranks|real|Finished|ERROR|rror"
done
md5sum /tmp/d_*.nwk
for dev in "" "--devices 0,0"; do
  echo "== mash placement $dev"; ( time DPR_COMM_WINDOW_MB=2 $B -i r -I /tmp/a20k.fa -m 1 -O /tmp/r_"${dev// /_}".nwk $dev ) 2>&1 | grep -E "Ranks|real|Operation|ERROR|rror"
done
md5sum /tmp/r_*.nwk
for e in "" "DPR_NJ_MULTI=rows" "DPR_NJ_MULTI=shard" "DPR_NJ_MODE=stream"; do
  echo "== nj $e"; ( time env $e $B -i m -I /tmp/a20k.fa -m 2 -d 2 -O /tmp/n_"$e".nwk --devices 0,0 ) 2>&1 | grep -E "Ranks|real|NJ over|Tree Created|ERROR|rror"
done
$B -i m -I /tmp/a20k.fa -m 2 -d 2 -O /tmp/n_one.nwk 2>/dev/null
md5sum /tmp/n_*.nwk
