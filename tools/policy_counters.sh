#!/bin/bash
# LDS counters of the fused policy encoder for the default library and the variants named on the command line (GPU box, repo root).
set -o pipefail
R=$PWD; O=$R/gpurun_out/pol_ctr; mkdir -p $O
exec < /dev/null
cd /tmp && export TMPDIR=/tmp
for v in "" "$@"; do
  n=${v:-default}
  LPBOX_LIB_VARIANT=$v timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/$n -o c -- python3 $R/tools/policy_body.py 128000 2 > $O/$n.log 2>&1 || exit 1
  python3 - "$O/$n" "$n" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(float); n = 0
for r in csv.DictReader(open(f)):
    if "policy_body" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n += 1
k = max(1, n // max(1, len(acc)))
print(sys.argv[2], {a: round(b / k) for a, b in acc.items()},
      "conflict share %.3f" % (acc["SQ_LDS_BANK_CONFLICT"] / max(acc["SQ_LDS_IDX_ACTIVE"], 1)),
      "lds active / busy %.3f" % (acc["SQ_LDS_IDX_ACTIVE"] / max(acc["SQ_BUSY_CU_CYCLES"], 1)),
      "mfma busy / busy %.3f" % (acc["SQ_VALU_MFMA_BUSY_CYCLES"] / max(acc["SQ_BUSY_CU_CYCLES"], 1)))
PY
done
