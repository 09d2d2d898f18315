cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02j
mkdir -p $O
for B in 3; do
PRV_BLOCKS_PER_CU=$B python3 scripts/kbench.py --reps 5 --tag r64_3w 2>&1 | grep "r64" | tee -a $O/kbench.txt
PRV_BLOCKS_PER_CU=$B python3 scripts/kbench.py --reps 5 --tag r64_3w 2>&1 | grep "r64" | tee -a $O/kbench.txt
done
PRV_RENDER64=0 python3 scripts/kbench.py --reps 5 --tag r64_off 2>&1 | grep "r64" | tee -a $O/kbench.txt
PRV_BLOCKS_PER_CU=3 python3 scripts/kbench.py --reps 5 --field 512 --tag f512_r64_3w 2>&1 | grep "r64" | tee -a $O/kbench.txt
export PRV_BLOCKS_PER_CU=3
python3 scripts/kbench.py --reps 1 --tag f256 > $O/kbench_256.txt 2>&1
bash scripts/pmc.sh $O/pmc256 1,2,3 > $O/pmc256.log 2>&1
grep -A22 "render_queue64" $O/pmc256/summary.txt | grep "INSTS_VALU\|ACTIVE_INST_VALU\|MFMA\|BUSY_CYCLES"
cat $O/kbench_256.txt | grep f256
PRV_R64_WAVES=4 python3 nerf_prv_amd/build.py --force > $O/build4.log 2>&1
PRV_BLOCKS_PER_CU=4 python3 scripts/kbench.py --reps 5 --tag r64_4w 2>&1 | grep "r64" | tee -a $O/kbench.txt
