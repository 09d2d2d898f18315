cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02ak
mkdir -p $O
F='s/eval_exact=[0-9]* //; s/BPC.*render=/render=/; s/kernel_rate.*ns/ns/'
KBENCH_CLOCKS=1 python3 scripts/kbench.py --reps 60 --tag long 2>&1 | grep "^long" | sed "$F" | tee -a $O/kbench.txt
python3 scripts/kbench.py --reps 5 --field 512 --tag f512 2>&1 | grep "^f512" | sed "$F" | tee -a $O/kbench.txt
PRV_RENDER64=0 python3 scripts/kbench.py --reps 5 --tag slot32 2>&1 | grep "^slot32" | sed "$F" | tee -a $O/kbench.txt
python3 scripts/kbench.py --reps 5 --views 540 --width 80 --height 45 --spp 16 --min-t 0.01 --tag ref 2>&1 | grep "^ref" | sed "$F" | tee -a $O/kbench.txt
PRV_ABLATE=16 python3 nerf_prv_amd/build.py --force > $O/build.log 2>&1
python3 scripts/kbench.py --reps 5 --tag window 2>&1 | grep "^window" | sed "$F" | tee -a $O/kbench.txt
PRV_ABLATE= python3 nerf_prv_amd/build.py --force > /dev/null 2>&1
rocm-smi --showpower --showclocks 2>/dev/null | head -30 > $O/smi.txt
