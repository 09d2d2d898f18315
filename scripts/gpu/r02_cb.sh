cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for R in 4096 16384 65536; do
python3 scripts/trainbench.py --rays $R --steps 600 --members 5 2>&1 | tail -1 | sed "s/^/[rays=$R] /"
done
