# timing ablation (wrong pixels): the two 64 -> 16 MLP layers on 16x16x32 MFMAs, without / with the operand re-layout swaps
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
for AB in 0 16 48; do
  PRV_ABLATE=$AB python3 -c "from nerf_prv_amd import build as b; b.build_hip(force=True)" > $O/build_$AB.log 2>&1 || tail -5 $O/build_$AB.log
  for i in 1 2; do
    python3 scripts/kbench.py --scene baseline --tag "ablate=$AB 256"
    python3 scripts/kbench.py --scene dense --tag "ablate=$AB 256"
  done 2>&1 | grep -v amdgpu.ids | tee -a $O/kbench.txt
done
