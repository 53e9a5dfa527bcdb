#!/bin/bash
# SQ / LDS hardware counters behind the statements in DESIGN.md (run on the GPU box from the repo root; separate --pmc passes,
# --kernel-trace only).  Raw CSVs land in gpurun_out/ctr_*; tools/summarise_counters.py writes profiles/<round>_counters.json.
set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
exec < /dev/null
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/ctr_lp_lds -o c -- python3 $R/tools/window.py 2000 1 > $O/ctr_lp_lds.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/ctr_lp_sq -o c -- python3 $R/tools/window.py 2000 1 > $O/ctr_lp_sq.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 --kernel-trace --output-format csv -d $O/ctr_lp_mix -o c -- python3 $R/tools/window.py 2000 1 > $O/ctr_lp_mix.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/ctr_policy -o c -- python3 $R/tools/policy_prof.py 128000 fused 3 > $O/ctr_policy.log 2>&1
echo "collect_counters rc=$?"
