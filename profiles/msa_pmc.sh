# PMC passes on msa_dist_kernel<JC> at the authors' sequence length (profiles/msa_block_bench.py); separate passes, kernel trace only
# usage: bash profiles/msa_pmc.sh [bench args, e.g. --gap]
cd $GRAFT_REPO_ROOT
export DPR_ROUND=r6
python3 profiles/msa_block_bench.py "$@" --reps 5
bash profiles/prof.sh pmc msa_p1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" python3 profiles/msa_block_bench.py "$@" --reps 2 | grep -i msa_dist
bash profiles/prof.sh pmc msa_p2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" python3 profiles/msa_block_bench.py "$@" --reps 2 | grep -i msa_dist
