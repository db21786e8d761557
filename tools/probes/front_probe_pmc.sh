#!/bin/bash
# counters of the L2's memory side for the three bare-store patterns of front_probe.hip (short-lived, long-lived, persistent waves)
# usage (GPU box, repo root): bash tools/probes/front_probe_pmc.sh  -> gpurun_out/r03/front_pmc/
set -e
OUT=$PWD/gpurun_out/r03/front_pmc
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/probes/front_probe.hip -o /tmp/front_probe
/tmp/front_probe 1048576 4 pmc > $OUT/unprofiled.log
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_BUSY_sum" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_CYCLE_sum" \
           "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCC_WRITE_sum TCC_NORMAL_WRITEBACK_sum" \
           "TCC_REQ_sum TCC_STREAMING_REQ_sum TCC_WRITE_SECTORS_sum TCC_WRITEBACK_sum"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p$i --output-format csv -- /tmp/front_probe 1048576 1 pmc > $OUT/p$i.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(acc):
        fh.write(k + "\n")
        for c in sorted(acc[k]):
            v = acc[k][c]
            fh.write("   %-44s mean %.4g over %d launches\n" % (c, sum(v) / len(v), len(v)))
print(open(out + "/summary.txt").read())
PY
