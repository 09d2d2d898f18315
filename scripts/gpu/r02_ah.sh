cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02ah
mkdir -p $O
for B in 3 4 3 4 6; do
PRV_BLOCKS_PER_CU=$B python3 scripts/kbench.py --reps 5 --tag bpc 2>&1 | grep "^bpc" | sed 's/eval_exact=[0-9]* //' | tee -a $O/kbench.txt
done
for B in 3 4; do
PRV_BLOCKS_PER_CU=$B python3 scripts/kbench.py --reps 5 --field 512 --tag bpc512 2>&1 | grep "^bpc" | sed 's/eval_exact=[0-9]* //' | tee -a $O/kbench.txt
PRV_BLOCKS_PER_CU=$B python3 scripts/kbench.py --reps 5 --views 540 --width 80 --height 45 --spp 16 --min-t 0.01 --tag bpc_ref 2>&1 | grep "^bpc" | sed 's/eval_exact=[0-9]* //' | tee -a $O/kbench.txt
done
