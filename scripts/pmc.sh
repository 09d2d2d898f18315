#!/bin/bash
# collect PMC counters for the render kernel in separate passes (--pmc only with --kernel-trace)
# usage: scripts/pmc.sh <outdir> [kbench args...]
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
i=0
for set in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" \
 "SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_LDS SQ_INSTS_SALU" \
 "SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM" \
 "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY" \
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
 "TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
 "FETCH_SIZE" \
 "WRITE_SIZE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 scripts/kbench.py --reps 1 "$@" > $OUT/p$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
with open(out + "/summary.txt", "w") as w:
    for k in agg:
        if "render_queue" not in k and "march" not in k: continue
        w.write(f"== {k}\n")
        for c in sorted(agg[k]):
            w.write(f"{c:45s} total={agg[k][c]:.6g} dispatches={cnt[k][c]} per_dispatch={agg[k][c]/cnt[k][c]:.6g}\n")
print(open(out + "/summary.txt").read())
PY
