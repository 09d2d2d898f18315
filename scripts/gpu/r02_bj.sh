cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
F='s/eval_exact=[0-9]* //; s/RM=8 rounds.*render=/render=/; s/kernel_rate.*ns/ns/'
for B in 2 3 4 2 3; do
PRV_BLOCKS_PER_CU=$B python3 scripts/kbench.py --reps 8 --tag bpc 2>&1 | grep "^bpc" | sed "$F"
done
