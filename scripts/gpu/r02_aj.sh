cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02aj
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sweep.py -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -2 $O/pytest.log
CF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -Wall -Iinclude"
D=nerf_prv_amd
for ST in iterative-ilp max-ilp iterative-minreg iterative-maxocc default; do
  if [ $ST = default ]; then X=""; else X="-mllvm -amdgpu-sched-strategy=$ST"; fi
  hipcc $CF $X -c -o $D/prv_march.o $D/csrc/prv_march.hip > $O/build_$ST.log 2>&1 || { echo "$ST: build failed"; continue; }
  hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libprv_hip.so $D/prv_kernels.o $D/prv_train.o $D/prv_api.o $D/prv_march.o -ldl -lz >> $O/build_$ST.log 2>&1
  for r in 1 2; do python3 scripts/kbench.py --reps 5 --tag "m" 2>&1 | grep "^m " | sed 's/eval_exact=[0-9]* //; s/BPC.*render=/render=/; s/kernel_rate.*ns/ns/' | sed "s|^|[$ST] |" | tee -a $O/kbench.txt; done
  python3 scripts/kbench.py --reps 5 --views 540 --width 80 --height 45 --spp 16 --min-t 0.01 --tag m 2>&1 | grep "^m " | sed 's/eval_exact=[0-9]* //; s/BPC.*render=/render=/; s/kernel_rate.*ns/ns/' | sed "s|^|[$ST ref] |" | tee -a $O/kbench.txt
done
