# hashed-level gathers with the nt cache policy vs default, on the HBM-bound 512^3 field and the 256^3 one (dev experiment)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
for NT in 0 1; do
  PRV_HASH_NT=$NT python3 -c "from nerf_prv_amd import build as b; b.build_hip(force=True)" > $O/build_$NT.log 2>&1
  for i in 1 2; do
    python3 scripts/kbench.py --scene dense --field 512 --tag "nt=$NT 512"
    python3 scripts/kbench.py --scene baseline --tag "nt=$NT 256"
  done 2>&1 | grep -v amdgpu.ids | tee -a $O/kbench.txt
done
