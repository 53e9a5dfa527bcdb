"""Turn gpurun_out/prof_* (tools/collect_profiles.sh) into the committed round summaries under profiles/."""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"


def stats(src, dst, keep=12):
    rows = list(csv.reader(open(src)))
    with open(dst, "w", newline="") as f:
        csv.writer(f).writerows(rows[:keep + 1])


def counter(path, name, kernel_substr):
    vals = {}
    for r in csv.DictReader(open(path)):
        # (the default PCG instantiation only: the bench also runs the opt-in direct mode once, template flag "true>")
        if r["Counter_Name"] == name and kernel_substr in r["Kernel_Name"] and "true>" not in r["Kernel_Name"]:
            vals.setdefault(r["Dispatch_Id"], 0.0)
            vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
    v = list(vals.values())
    return sum(v) / len(v), len(v)


stats(os.path.join(O, "prof_bench", "bench_kernel_stats.csv"), os.path.join(P, rnd + "_kernel_stats.csv"))
if os.path.exists(os.path.join(O, "prof_segb", "segb_kernel_stats.csv")):
    stats(os.path.join(O, "prof_segb", "segb_kernel_stats.csv"), os.path.join(P, rnd + "_seg_batch_kernel_stats.csv"))
if os.path.exists(os.path.join(O, "prof_c4", "c4_kernel_stats.csv")):
    stats(os.path.join(O, "prof_c4", "c4_kernel_stats.csv"), os.path.join(P, rnd + "_config4_kernel_stats.csv"))
stats(os.path.join(O, "prof_policy", "policy_kernel_stats.csv"), os.path.join(P, rnd + "_policy_kernel_stats.csv"))
if os.path.exists(os.path.join(O, "prof_policy32", "policy32_kernel_stats.csv")):
    stats(os.path.join(O, "prof_policy32", "policy32_kernel_stats.csv"), os.path.join(P, rnd + "_policy_f32_kernel_stats.csv"))
stats(os.path.join(O, "prof_big", "big_kernel_stats.csv"), os.path.join(P, rnd + "_big_kernel_stats.csv"))
stats(os.path.join(O, "prof_seg", "seg_kernel_stats.csv"), os.path.join(P, rnd + "_seg_kernel_stats.csv"))
fetch_kb, nf = counter(os.path.join(O, "prof_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE", "lp_window_kernel")
write_kb, nw = counter(os.path.join(O, "prof_write", "write_counter_collection.csv"), "WRITE_SIZE", "lp_window_kernel")
out = {
    "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 ; same with --pmc WRITE_SIZE (two separate passes, tools/collect_profiles.sh)",
    "kernel": "lp_window_kernel<512,1> (config 2: 256 instances j=100/k=500)",
    "instances": 256,
    "launches_sampled": [nf, nw],
    "FETCH_SIZE_KB_per_launch": fetch_kb,
    "WRITE_SIZE_KB_per_launch": write_kb,
    "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024,
    "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B; calibrated there for wide coalesced reads, our 8-byte/2-byte loads are not, so this is an upper estimate). One launch = one full solve of the 256-instance batch: the solve is register/LDS resident, HBM sees the state once in and once out.",
}
json.dump(out, open(os.path.join(P, rnd + "_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
for name in ("prof_bench", "prof_policy", "prof_big", "prof_seg"):
    log = os.path.join(O, name + ".log")
    for ln in open(log):
        if ln.startswith("{") or ln.startswith("rows ") or ln.startswith("n="):
            print(name, ln.strip()[:300])
        if ln.startswith("{") and name in ("prof_bench", "prof_c4", "prof_big", "prof_seg"):     # the bench line of the profiled run itself
            d = json.loads(ln)
            d["profiler_mode"] = True
            d["profiler_note"] = ("printed by the run rocprofv3 was tracing" + (" with hipGraph replay OFF (LPBOX_*_NOGRAPH=1: eager launches, so that "
                                  "every kernel appears in the trace)" if name in ("prof_big", "prof_seg") else "") +
                                  ": it documents what the kernel statistics next to it belong to, it is NOT the benchmark figure of this "
                                  "configuration (that is " + rnd + "_bench_config*.json / the driver's BENCH record)")
            open(os.path.join(P, rnd + "_" + name[5:] + "_profiled_run_line.json"), "w").write(json.dumps(d) + "\n")
