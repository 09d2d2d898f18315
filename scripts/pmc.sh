#!/bin/bash
# PMC counters for the render path, one rocprofv3 pass per counter set, each under its own timeout
# usage: scripts/pmc.sh <outdir> <sets: comma list of 1..7> [kbench args...]
export TMPDIR=/tmp
OUT=$1; SETS=$2; shift; shift
mkdir -p $OUT
declare -A S
S[1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU"
S[2]="SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES"
S[3]="SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA"
S[4]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
S[5]="FETCH_SIZE"
S[6]="WRITE_SIZE"
S[8]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVES SQ_BUSY_CYCLES"
S[9]="TA_TA_BUSY_sum TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum"
S[10]="TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
S[11]="TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_TCR_TCP_STALL_CYCLES_sum"
S[7]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum"
for i in ${SETS//,/ }; do
  timeout 150 rocprofv3 --kernel-trace --pmc ${S[$i]} --output-format csv -d $OUT/p$i -- python3 ${PMC_SCRIPT:-scripts/kbench.py} ${PMC_SCRIPT:+} $( [ -z "$PMC_SCRIPT" ] && echo --reps 1 ) "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed/timeout"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("prv::", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
with open(out + "/summary.txt", "w") as w:
    for k in agg:
        if "render_queue" not in k and "march" not in k and "train_" not in k: continue
        w.write(f"== {k}\n")
        for c in sorted(agg[k]):
            w.write(f"{c:32s} total={agg[k][c]:.6g} dispatches={cnt[k][c]} per_dispatch={agg[k][c]/cnt[k][c]:.6g}\n")
print(open(out + "/summary.txt").read())
PY
