cd $GRAFT_REPO_ROOT
timeout 300 python3 scripts/gpu/r02_ca.py 2>&1 | tail -5
