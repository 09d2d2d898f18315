"""PMC summaries (scripts/pmc.sh) + the kbench line of the same workload -> the two JSON files bench.py reads:
profiles/r06_round_cost.json (VALU / MFMA instructions per 32-sample wave-round of render_queue64) and
profiles/r06_pmc_traffic.json (fabric-side bytes per launch), keyed "<variant> <scene>" and stamped with the sha256 of
the libprv_hip.so they were measured on (bench.py compares it with the library it loads).  usage:
  python scripts/pmc_to_json.py <key e.g. "64<4, 5> baseline"> <pmc summary.txt> <kbench log> <tag>"""
import hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variant, summary, kbench_log, tag = sys.argv[1:5]
cur, counters = None, {}
for line in open(summary):
    if line.startswith("=="):
        cur = line[2:].strip()
        counters[cur] = {}
    elif cur and "per_dispatch=" in line:
        counters[cur][line.split()[0]] = float(line.split("per_dispatch=")[1])
rq = next(v for k, v in counters.items() if "render_queue" in k)
kb = open(kbench_log).read()
ev, rounds = int(re.search(r"eval_exact=(\d+)", kb).group(1)), int(re.search(r"rounds=(\d+)", kb).group(1))

sys.path.insert(0, ROOT)
from nerf_prv_amd import _lib  # noqa: E402

LIB_SHA = _lib.device_code_digest()  # the gfx950 code objects (.hip_fatbin), not the whole file


def update(path, entry):
    entry = dict(entry, device_code_sha256=LIB_SHA)
    full = os.path.join(ROOT, path)
    data = json.load(open(full)) if os.path.exists(full) else {}
    data[variant] = entry
    json.dump(data, open(full, "w"), indent=1)

if "SQ_INSTS_VALU" in rq:
    update("profiles/r06_round_cost.json", {
        "valu_insts_per_round": rq["SQ_INSTS_VALU"] / rounds,
        "mfma_insts_per_round": rq.get("SQ_INSTS_MFMA", 0) / rounds if "SQ_INSTS_MFMA" in rq else None,
        "valu_active_quadcycles_per_inst": rq["SQ_ACTIVE_INST_VALU"] / rq["SQ_INSTS_VALU"] if "SQ_ACTIVE_INST_VALU" in rq else None,
        "sq_insts_valu_per_launch": rq["SQ_INSTS_VALU"], "wave_rounds_per_launch": rounds, "samples_evaluated_per_launch": ev,
        "source": f"rocprofv3 --kernel-trace --pmc (scripts/pmc.sh), one render_queue launch of scripts/kbench.py; {tag}"})
if "FETCH_SIZE" in rq and "WRITE_SIZE" in rq:
    update("profiles/r06_pmc_traffic.json", {
        "fetch_kib_per_launch": rq["FETCH_SIZE"], "write_kib_per_launch": rq["WRITE_SIZE"],
        "tcc_hit_per_launch": rq.get("TCC_HIT_sum"), "tcc_miss_per_launch": rq.get("TCC_MISS_sum"),
        "samples_evaluated_per_launch": ev,
        "source": f"rocprofv3 --kernel-trace --pmc, separate passes (scripts/pmc.sh sets 4,5,6), scripts/kbench.py one launch; {tag}. "
                  "FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE not doubled: per-lane gathers, one 64-B request per missing load "
                  "(profiles/archive/r01_gather_calib.txt)"})
print(json.dumps({k: v for k, v in rq.items()}, indent=1))
