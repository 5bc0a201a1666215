# `dipper --devices ...` with FIVE process ranks on one GPU and with inputs smaller than / not divisible by the rank count:
# every multi-rank Newick must equal the one-rank file.  bash profiles/multirank_cli_edge_cases.sh  (prints md5 pairs + OK/DIFF)
cd $GRAFT_REPO_ROOT
G=tools/bin/gen_synth
B=dipper_amd/bin/dipper
fail=0
cmp_run() {   # tag, ranks spec, then the command's arguments
  tag=$1; devs=$2; shift 2
  timeout -k 5 120 $B "$@" -O /tmp/e_${tag}_1.nwk > /tmp/e_${tag}_1.log 2>&1; r1=$?
  timeout -k 5 120 $B "$@" -O /tmp/e_${tag}_m.nwk --devices $devs > /tmp/e_${tag}_m.log 2>&1; rm=$?
  if [ $r1 -ne 0 ] || [ $rm -ne 0 ]; then
     if [ $r1 -eq $rm ]; then echo "$tag: both exit $r1 (same refusal): $(grep -m1 -i error /tmp/e_${tag}_m.log)"; else echo "$tag: EXIT CODES DIFFER one-rank $r1, ranks $rm"; tail -3 /tmp/e_${tag}_m.log; fail=1; fi
     return
  fi
  if cmp -s /tmp/e_${tag}_1.nwk /tmp/e_${tag}_m.nwk; then echo "$tag: OK ($(wc -c < /tmp/e_${tag}_m.nwk) bytes, $(grep -m1 Ranks /tmp/e_${tag}_m.log))"; else echo "$tag: DIFF"; fail=1; fi
}
$G --tips 3000 --sites 500 --seed 5 --fasta /tmp/e3k.fa >/dev/null
$G --tips 7 --sites 200 --seed 6 --fasta /tmp/e7.fa >/dev/null
$G --tips 4 --sites 200 --seed 7 --fasta /tmp/e4.fa >/dev/null
$G --tips 1501 --sites 300 --seed 8 --fasta /tmp/e1501.fa >/dev/null
cmp_run nj5 0,0,0,0,0 -i m -I /tmp/e3k.fa -m 2 -d 2
cmp_run place5 0,0,0,0,0 -i m -I /tmp/e3k.fa -m 1 -d 2
cmp_run dc5 0,0,0,0,0 -i m -I /tmp/e3k.fa -m 3 -d 2
cmp_run mash5 0,0,0,0,0 -i r -I /tmp/e3k.fa -m 1
cmp_run nj_7tips_3ranks 0,0,0 -i m -I /tmp/e7.fa -m 2 -d 2
cmp_run place_7tips_3ranks 0,0,0 -i m -I /tmp/e7.fa -m 1 -d 2
cmp_run nj_4tips_5ranks 0,0,0,0,0 -i m -I /tmp/e4.fa -m 2 -d 2
cmp_run place_4tips_5ranks 0,0,0,0,0 -i m -I /tmp/e4.fa -m 1 -d 2
cmp_run place_1501_4ranks 0,0,0,0 -i m -I /tmp/e1501.fa -m 1 -d 2
cmp_run nj_rows_1501_3ranks 0,0,0 -i m -I /tmp/e1501.fa -m 2 -d 2
DPR_NJ_MULTI=rows cmp_run nj_rows_forced_1501 0,0,0 -i m -I /tmp/e1501.fa -m 2 -d 2
DPR_NJ_MULTI=shard cmp_run nj_shard_forced_1501 0,0,0 -i m -I /tmp/e1501.fa -m 2 -d 2
# --add: backbone of 1 000 tips (one rank), 501 queries over four ranks
head -n 2000 /tmp/e1501.fa > /tmp/e_bb.fa
$B -i m -I /tmp/e_bb.fa -m 2 -d 2 -O /tmp/e_bb.nwk > /dev/null 2>&1
cmp_run add_501_4ranks 0,0,0,0 -i m -I /tmp/e1501.fa -a -t /tmp/e_bb.nwk -d 2
[ $fail -eq 0 ] && echo "ALL EDGE CASES OK" || echo "SOME EDGE CASES FAILED"
exit $fail
