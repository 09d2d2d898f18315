cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02ai
mkdir -p $O
i=0
while read -r FL; do
i=$((i+1))
PRV_EXTRA_HIPFLAGS="$FL" timeout 400 python3 nerf_prv_amd/build.py --force > $O/build_$i.log 2>&1 || { echo "v$i build failed: $FL" | tee -a $O/kbench.txt; continue; }
for r in 1 2; do python3 scripts/kbench.py --reps 5 --tag "v$i" 2>&1 | grep "^v$i" | sed 's/eval_exact=[0-9]* //; s/BPC.*render=/render=/; s/kernel_rate.*ns/ns/' | sed "s|^|[$FL] |" | tee -a $O/kbench.txt; done
done <<'LIST'
-mllvm -amdgpu-sched-strategy=max-ilp
-mllvm -amdgpu-sched-strategy=max-memory-clause
-mllvm -amdgpu-sched-strategy=iterative-ilp
-mllvm -amdgpu-sched-strategy=iterative-minreg
-mllvm -amdgpu-schedule-metric-bias=0
-mllvm -amdgpu-schedule-metric-bias=40
LIST
python3 nerf_prv_amd/build.py --force > /dev/null 2>&1
