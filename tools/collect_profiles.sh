#!/bin/bash
# Round profiles (run on the GPU box from the repo root): rocprofv3 kernel stats of the headline bench and of the other configs, two
# separate PMC passes for the HBM traffic of the dominant kernel.  Results land in gpurun_out/prof_*; the summaries that are judged
# are copied into profiles/ by tools/summarise_profiles.py <round>.
set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
exec < /dev/null
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 $R/bench.py --config 2 --steps 5 --warmup 1 --cpu-sample 0 > $O/prof_bench.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_fetch -o fetch -- python3 $R/bench.py --config 2 --steps 2 --warmup 1 --cpu-sample 0 > $O/prof_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_write -o write -- python3 $R/bench.py --config 2 --steps 2 --warmup 1 --cpu-sample 0 > $O/prof_write.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o c4 -- python3 $R/bench.py --config 4 --steps 2 --warmup 1 --cpu-sample 0 > $O/prof_c4.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_policy -o policy -- python3 $R/tools/policy_prof.py 128000 fused 10 > $O/prof_policy.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_policy32 -o policy32 -- python3 $R/tools/policy_prof.py 128000 mfma32 5 > $O/prof_policy32.log 2>&1 &&
LPBOX_BIG_NOGRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_big -o big -- python3 $R/bench.py --config 5 --steps 2 --warmup 1 --cpu-sample 0 > $O/prof_big.log 2>&1 &&
LPBOX_SEG_NOGRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_seg -o seg -- python3 $R/bench.py --config 3 --steps 3 --warmup 1 --cpu-sample 0 > $O/prof_seg.log 2>&1
echo "collect_profiles rc=$?"
cd /tmp; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_segb -o segb -- python3 $R/tools/seg_batch_bench.py 100 10000 > $O/prof_segb.log 2>&1; echo "seg batch profile rc=$?"
# gpurun copies back at most 64 MiB: the per-dispatch traces are not needed by tools/summarise_profiles.py
find $O/prof_* -name "*_kernel_trace.csv" -size +2M -delete 2>/dev/null; true
