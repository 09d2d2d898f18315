cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python3 bench.py --mode strong --views-total 1024 --field 512 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-training > /tmp/strong.json 2> /tmp/strong.err; echo "rc=$?"
python3 -c "
import json;d=json.load(open('/tmp/strong.json'));print(d['config'], round(d['value']/1e9,2), round(d['ms_per_step'],2), d['roofline']['frac'], d['scaling'])"
