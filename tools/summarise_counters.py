"""gpurun_out/ctr_* (tools/collect_counters.sh) -> profiles/<round>_counters.json: counter sums per kernel and the ratios DESIGN.md quotes."""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"


def sums(d, kernel):
    acc = {}
    for r in csv.DictReader(open(os.path.join(O, d, "c_counter_collection.csv"))):
        if kernel in r["Kernel_Name"]:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return acc


lp = {}
for d in ("ctr_lp_lds", "ctr_lp_sq", "ctr_lp_mix"):
    for k, v in sums(d, "lp_window_kernel").items():
        lp.setdefault(k, v)
pol = sums("ctr_policy", "policy_body_kernel")
out = {
    "lp_window_kernel": {"workload": "tools/window.py 2000 1 (256 instances, 2000 iterations each)", "counters": lp, "ratios": {
        "lds_bank_conflict_over_lds_active": lp["SQ_LDS_BANK_CONFLICT"] / lp["SQ_LDS_IDX_ACTIVE"],
        "lds_active_over_cu_busy": lp["SQ_LDS_IDX_ACTIVE"] / lp["SQ_BUSY_CU_CYCLES"],
        "valu_active_per_wave_cycle": lp["SQ_ACTIVE_INST_VALU"] / lp["SQ_WAVE_CYCLES"],
        "wait_any_per_wave_cycle": lp["SQ_WAIT_ANY"] / lp["SQ_WAVE_CYCLES"],
        "fp64_share_of_valu_instructions": (lp["SQ_INSTS_VALU_ADD_F64"] + lp["SQ_INSTS_VALU_MUL_F64"] + lp.get("SQ_INSTS_VALU_FMA_F64", 0.0)) / lp["SQ_INSTS_VALU"]}},
    "policy_body_kernel": {"workload": "tools/policy_prof.py 128000 fused 3", "counters": pol, "ratios": {
        "lds_bank_conflict_over_lds_active": pol["SQ_LDS_BANK_CONFLICT"] / pol["SQ_LDS_IDX_ACTIVE"],
        "lds_active_over_cu_busy": pol["SQ_LDS_IDX_ACTIVE"] / pol["SQ_BUSY_CU_CYCLES"],
        "mfma_busy_over_4x_cu_busy": pol["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * pol["SQ_BUSY_CU_CYCLES"]),
        "wait_any_per_wave_cycle": pol["SQ_WAIT_ANY"] / pol["SQ_WAVE_CYCLES"]}},
    "note": "rocprofv3 --pmc passes with --kernel-trace only; sums over all dispatches of the kernel in the run; ratios, not absolute cycles, are what DESIGN.md uses",
}
json.dump(out, open(os.path.join(P, rnd + "_counters.json"), "w"), indent=1)
print(json.dumps({k: v["ratios"] for k, v in out.items() if isinstance(v, dict)}, indent=1))
