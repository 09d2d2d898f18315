# A/B of compile-time variants of libprv_hip.so on the kernel microbench (dev): scripts/gpu/ab.sh VAR "v1 v2 ..." [kbench args]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
VAR=$1; VALS=$2; shift; shift
O=gpurun_out/ab_$VAR; mkdir -p $O
for V in $VALS; do
  env $VAR=$V python3 -c "from nerf_prv_amd import build as b; b.build_hip(force=True)" > $O/build_$V.log 2>&1 || tail -5 $O/build_$V.log
  for i in 1 2; do
    python3 scripts/kbench.py --scene baseline --tag "$VAR=$V" "$@"
    python3 scripts/kbench.py --scene dense --tag "$VAR=$V" "$@"
  done 2>&1 | grep -v amdgpu.ids | tee -a $O/kbench.txt
done
