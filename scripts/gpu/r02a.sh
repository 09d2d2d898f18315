set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02a
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/pytest.log
timeout 600 python scripts/relerr_diag.py > gpurun_out/r02a/relerr.txt 2>&1
timeout 600 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; echo "bench rc=$?" >> gpurun_out/r02a/bench.err
tail -5 gpurun_out/r02a/pytest.log
