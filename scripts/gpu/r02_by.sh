cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02by
mkdir -p $O
for i in $(seq 1 10); do
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest_$i.log 2>&1; echo "run $i rc=$? $(grep -o '[0-9]* passed\|[0-9]* failed' $O/pytest_$i.log | tr '\n' ' ')"
done
