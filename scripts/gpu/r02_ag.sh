cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02ag
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 600 python3 bench.py --no-cpu-baseline --no-training > $O/bench.json 2> $O/bench.err
python3 -c "
import json;d=json.load(open('$O/bench.json'));r=d['roofline'];print(round(d['value']/1e9,2),round(d['ms_per_step'],3),r['avg_launch_ms'],r.get('shader_clock_ghz_measured'),r.get('valu_issue_frac'),r.get('valu_issue_frac_at_measured_clock'),r.get('mfma_pipe_frac_at_measured_clock'))
x=d['field512']['roofline'];print(x.get('shader_clock_ghz_measured'),x['frac'])"
