cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02l
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python3 scripts/kbench.py --reps 1 --tag f256 > $O/kbench_256.txt 2>&1
python3 scripts/kbench.py --reps 1 --field 512 --tag f512 > $O/kbench_512.txt 2>&1
bash scripts/pmc.sh $O/pmc256 1,2,3,4,5,6 > $O/pmc256.log 2>&1
bash scripts/pmc.sh $O/pmc512 1,2,4,5,6 --field 512 > $O/pmc512.log 2>&1
cat $O/kbench_256.txt $O/kbench_512.txt | grep "f256\|f512"
