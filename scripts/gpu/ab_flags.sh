# A/B of extra compile flags for libprv_hip.so on the kernel microbench (dev): variants separated by '|'
#   scripts/gpu/ab_flags.sh "|-DFOO|-DFOO -DBAR=2" [kbench args]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
VARIANTS=$1; shift
O=gpurun_out/ab_flags; mkdir -p $O
IFS='|' read -ra VS <<< "$VARIANTS"
for V in "${VS[@]}"; do
  PRV_EXTRA_HIPFLAGS="$V" python3 -c "from nerf_prv_amd import build as b; b.build_hip(force=True)" > $O/build.log 2>&1 || tail -5 $O/build.log
  for i in 1 2; do
    python3 scripts/kbench.py --scene baseline --tag "[$V]" "$@"
    python3 scripts/kbench.py --scene dense --tag "[$V]" "$@"
  done 2>&1 | grep -v amdgpu.ids | tee -a $O/kbench.txt
  python3 scripts/kbench.py --scene dense --field 512 --tag "[$V] 512" "$@" 2>&1 | grep -v amdgpu.ids | tee -a $O/kbench.txt
done
