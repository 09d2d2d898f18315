cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 300 python3 bench.py --gpus 2 --steps 2 --warmup 1 > /tmp/b2.out 2> /tmp/b2.err; echo "rc=$? after $(( $(date +%s) - t0 )) s"
tail -3 /tmp/b2.err | cut -c1-300
