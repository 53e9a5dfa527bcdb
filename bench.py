#!/usr/bin/env python3
"""Headline benchmark: ADMM iterations/sec on a batch of 256 combinatorial-auction LP instances (j=100 items / k=500 bids),
fp64, per GPU (BASELINE.json configs[1]); wall-clock to converge = ms_per_step.

A "step" = one complete solve of the batch: ADMM_lp_iters_init + ADMM_lp_iters(0, 2e4) semantics for every instance
(each stops on its own reference stop test, LPcpp:934/:977), i.e. one launch of the persistent window kernel.
`value` = instance-iterations executed by all ranks / time.  With --gpus N every rank holds its own 256-instance shard
(instance-sharded, no data-path collective; weak scaling).

Launch:  python bench.py [--gpus 1] [--steps K] [--warmup W]
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "accelerated-lpbox-admm_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FIXTURE = os.path.join(ROOT, "tests", "golden", "lp_100_500_seed0.npz")
MAX_ITERS = 20000       # LPcpp:496 max_iters = 2e4 (test.cpp:14)


def load_instances(path):
    d = np.load(path)
    out = []
    cp = ri = pr = 0
    for n, l, nnz in zip(d["n"], d["l"], d["nnz"]):
        n, l, nnz = int(n), int(l), int(nnz)
        out.append(dict(n=n, l=l, nnz=nnz, colptr=d["colptr"][cp:cp + n + 1].astype(np.int32),
                        rowidx=d["rowidx"][ri:ri + nnz].astype(np.int32), b=-1.0 * d["price"][pr:pr + n]))
        cp += n + 1; ri += nnz; pr += n
    return out


def byte_model(I):
    """Algorithmic bytes per outer iteration = B_fixed + K * B_pcg (SURVEY.md section 8d / BASELINE.md section 3)."""
    n, l, nnz = I["n"], I["l"], I["nnz"]
    m_E = 12 * nnz + 4 * (l + 1)
    m_Et = 12 * nnz + 4 * (n + 1)
    b_pcg = m_E + m_Et + 8 * (13 * n + 2 * l)
    b_fixed = 3 * m_E + 2 * m_Et + 8 * (30 * n + 12 * l)
    return b_fixed, b_pcg


def cpu_baseline(insts, sample):
    """The CPU oracle (a port: the reference's Eigen path cannot be built here), 1 thread, on a bounded sample."""
    from oracle import oracle as O
    O.build()
    iters = 0
    t0 = time.perf_counter()
    for I in insts[:sample]:
        s = O.LpOracle(0, order=O.ORDER_EIGEN)
        s.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"])
        s.solve_init()
        s.solve_iter(0, MAX_ITERS)
        iters += s.total_outer_iters
    dt = time.perf_counter() - t0
    return dict(value=iters / dt, unit="instance-iterations/s", cores=1, kind="port",
                sample=f"first {sample} instances of the batch solved to convergence by oracle/lpbox_oracle.c "
                       f"(gcc -O3, Eigen reduction order), {iters} iterations in {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="instances per GPU (default: the 256 of BASELINE configs[1])")
    ap.add_argument("--cpu-sample", type=int, default=32, help="instances solved by the CPU oracle for cpu_baseline (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    from lpbox_hip import _lib
    from lpbox_hip.lp import LpBatch

    L = _lib.load()
    if L.lpbox_device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible; the HIP path has no CPU fallback")
    # rehearsal on a one-GPU box only: LPBOX_BENCH_BACKEND=gloo LPBOX_BENCH_DEVICE=0 puts every rank on one card
    backend = os.environ.get("LPBOX_BENCH_BACKEND", "nccl")
    if "LPBOX_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["LPBOX_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    insts = load_instances(FIXTURE)
    shard = [insts[(i + 0) % len(insts)] for i in range(args.batch)]   # every rank: the same synthetic 256-instance shard
    batch = LpBatch(shard, device=local_rank)
    cfg = batch.config()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def step():
        batch.solve_init()
        batch.solve_iter(0, MAX_ITERS)      # synchronous: returns when every instance has stopped

    for _ in range(args.warmup):
        step()
    batch.kernel_time(reset=True)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0

    ctr = [batch.counters(i) for i in range(args.batch)]       # of the last step (identical every step: deterministic)
    outer = np.array([c[0] for c in ctr], np.float64)
    pcg = np.array([c[1] for c in ctr], np.float64)
    iters_per_step = float(outer.sum())
    bm = np.array([byte_model(I) for I in shard], np.float64)
    alg_bytes_per_launch = float((outer * bm[:, 0] + pcg * bm[:, 1]).sum())
    k_ms, k_launches = batch.kernel_time()
    objs = np.array([-batch.cal_obj(i) for i in range(args.batch)])
    infeasible = int(sum(batch.check_infeasible_l2f(i) > 0 for i in range(min(args.batch, 32))))

    t_max, it_total = dt, iters_per_step * args.steps
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ii = torch.tensor([it_total], dtype=torch.float64, device=red_dev)
        dist.all_reduce(ii, op=dist.ReduceOp.SUM)
        t_max, it_total = float(tt.item()), float(ii.item())

    if rank == 0:
        kernel_s = (k_ms / 1e3) / max(k_launches, 1)
        achieved = alg_bytes_per_launch / kernel_s / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "admm_iters_per_sec_batched_lp_j100_k500",
            "value": it_total / t_max,
            "unit": "instance-iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * t_max / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (reference generator, RandomState(0), first 256 draws j=100/k=500; same shard on every rank)",
            "config": {"workload": "batch of 256 j=100/k=500 combinatorial-auction LP instances per GPU, fp64, "
                                   "full solve (init + ADMM_lp_iters(0,2e4)) = 1 step",
                       "instances_per_gpu": args.batch, "threads_per_instance": cfg["threads"],
                       "lds_bytes_per_instance": cfg["lds_bytes"], "parallelism": f"instance-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "lp_window_kernel", "kernel_ms_per_launch": 1e3 * kernel_s,
                         "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                         "note": "state is register/LDS resident, so algorithmic (streaming-model) GB/s may exceed HBM peak"},
            "detail": {"instance_iters_per_step": iters_per_step, "mean_outer_iters": float(outer.mean()),
                       "max_outer_iters": float(outer.max()), "mean_pcg_per_outer": float(pcg.sum() / outer.sum()),
                       "wall_clock_to_converge_ms": 1e3 * t_max / args.steps,
                       "mean_objective": float(objs.mean()), "infeasible_in_first_32": infeasible},
        }
        if world == 1 and args.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline(shard, args.cpu_sample)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
