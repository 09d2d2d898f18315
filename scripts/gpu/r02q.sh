cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02q
mkdir -p $O
for M in 0 16; do
PRV_MERGE_MAX=$M python3 scripts/kbench.py --reps 5 --views 540 --width 80 --height 45 --spp 16 --min-t 0.01 --tag ref_merge=$M 2>&1 | grep "ref_" | tee -a $O/kbench.txt
PRV_MERGE_MAX=$M python3 scripts/kbench.py --reps 5 --views 540 --width 80 --height 45 --spp 16 --min-t 0.0001 --tag ref_minT1e-4_merge=$M 2>&1 | grep "ref_" | tee -a $O/kbench.txt
done
PRV_RENDER64=0 python3 scripts/kbench.py --reps 5 --views 540 --width 80 --height 45 --spp 16 --min-t 0.01 --tag ref_r32 2>&1 | grep "ref_" | tee -a $O/kbench.txt
python3 scripts/kbench.py --reps 5 --views 64 --width 800 --height 800 --spp 1 --min-t 0.01 --tag big_minT0.01 2>&1 | grep "big_" | tee -a $O/kbench.txt
python3 scripts/kbench.py --reps 5 --views 34 --width 320 --height 180 --spp 16 --min-t 0.01 --tag mid_spp16 2>&1 | grep "mid_" | tee -a $O/kbench.txt
python3 scripts/kbench.py --reps 5 --views 540 --width 320 --height 180 --spp 1 --min-t 0.01 --tag mid_spp1 2>&1 | grep "mid_" | tee -a $O/kbench.txt
