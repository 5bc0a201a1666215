# round 4, call h: do the latency-bound loops run at reduced clocks?  (clocks sampled during the loop; a background FMA load)
O=gpurun_out/r4/h; mkdir -p $O
rocm-smi --showclocks > $O/clocks_idle.txt 2>&1
export DPR_NJP_SMALL=fused
for sp in 0 8 64 256 1024; do
  echo "== fused kernel, spin $sp"; python3 profiles/nj_target.py --no-torch --reps 2 --spin $sp --clocks 0.25 2>&1 | grep -o '"nj_ms": [0-9.]*\|"clocks": {[^}]*}' | tee -a $O/spin_30k.txt
done
unset DPR_NJP_SMALL
for sp in 0 64; do
  echo "== post3, spin $sp"; python3 profiles/nj_target.py --no-torch --reps 2 --spin $sp 2>&1 | grep -o '"nj_ms": [0-9.]*' | tee -a $O/spin_30k_post3.txt
done
