"""dev: build the inputs of tests/test_gpu_ingp.py::test_planner_scores_with_members_given_as_ingp_snapshots once, print the command"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pathlib import Path
from oracle import oracle
from tests.test_gpu_ingp import snapshot_of, SMALL_NGP, GOLD, ROOT
from tests.test_gpu_planner import YAML
pre = Path(sys.argv[1])
(pre / "models" / "objA").mkdir(parents=True, exist_ok=True)
for e in range(2):
    snapshot_of(oracle, SMALL_NGP, 4000 + e, pre / "models" / "objA" / f"member_{e}.ingp")
cfg = pre / "cfg.yaml"
cfg.write_text(YAML.format(pre=pre, vs=os.path.join(GOLD, "hemisphere"), method=2, model_source="pretrained_members: 1").replace("field_density_bias: 3.0", "field_density_bias: 0.0"))
print(cfg)
