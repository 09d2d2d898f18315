cd $GRAFT_REPO_ROOT
python3 scripts/gpu/r02_bl.py 2>&1 | tail -3
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
