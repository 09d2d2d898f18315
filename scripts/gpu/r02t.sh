cd $GRAFT_REPO_ROOT
python3 scripts/stepdiag.py 2>&1 | grep " it"
