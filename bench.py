#!/usr/bin/env python3
"""Benchmarks of the MI355X-native Lp-Box ADMM inner solver on BASELINE.json's configurations.

  --config 2 (default)  batch of 256 j=100/k=500 combinatorial-auction LPs per GPU, fp64      (BASELINE configs[1], the headline)
  --config 4            batch of 256 j=500/k=2000 LPs per GPU                                  (configs[3]; 2048 over 8 GPUs)
  --config 3            one full-resolution segmentation MRF (n = 187 500)                     (configs[2])
  --config 5            one LP with 10^6 variables, variable-sharded over the ranks            (configs[4])

Configs 2 / 4: a "step" = one complete solve of the batch (ADMM_lp_iters_init + ADMM_lp_iters(0, 2e4) for every instance, each
stopping on its own reference stop test, LPcpp:934/:977) = one launch of the persistent window kernel; `value` = instance-iterations
executed by all ranks / time; ms_per_step = wall-clock to converge.  With --gpus N every rank holds its own 256-instance shard
(instance-sharded, no data-path collective, weak scaling): rank 0 the fixture = instances 0..255 of the reference generator's seed-0
stream, rank r > 0 instances 256 r .. 256 r + 255 of the SAME stream (BASELINE configs[3]: 2048 draws over 8 GPUs), produced on the spot by
the draw-for-draw restatement of the generator in lpbox_hip/auction.py from the generator states stored in tests/golden/lp_stream_*.npz
and checked against the digests stored there.  (LPBOX_BENCH_SHARD=relabel, or a missing stream fixture: the round-2 behaviour -- rank 0's
LPs with bids and items relabelled by a rank-seeded permutation.)
Config 3: step = one ADMM_bqp_unconstrained_legacy solve; config 5: step = init + 100 ADMM iterations of the sharded instance.

Launch:  python bench.py [--config C] [--gpus N] [--steps K] [--warmup W]
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
With --gpus N > 1 and no WORLD_SIZE in the environment bench.py starts its N ranks itself: N fresh child processes (one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), started BEFORE the parent has made any GPU call; the parent only relays rank 0's line.

Without --config (what the driver runs) the headline line is config 2 and, on one GPU, short passes of configs 4, 3 and 5 follow
OUTSIDE the headline's timed region; each lands under detail.configs.<c> as a complete line of its own (value, roofline, cpu_baseline).
"""
import argparse
import glob
import json
import os
import platform
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "accelerated-lpbox-admm_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

# /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy); LDS "aggregate with every CU streaming:
# ~150 TB/s for ds_read_b64/b128" (256 CUs x 256 B/clk x ~2.4 GHz)
HBM_PEAK_GBS = 8000.0
LDS_PEAK_GBS = 150000.0
GOLDEN = os.path.join(ROOT, "tests", "golden")
FIXTURE = os.path.join(GOLDEN, "lp_100_500_seed0.npz")
FIXTURE_C4 = os.path.join(GOLDEN, "lp_500_2000_seed0.npz")
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


def relabel(I, seed):
    """The same LP with its variables (bids) and rows (items) renumbered by a seeded permutation."""
    rs = np.random.RandomState(seed)
    n, l = I["n"], I["l"]
    pc, pr_ = rs.permutation(n), rs.permutation(l)            # new index of old column j / old row i
    lens = np.diff(I["colptr"])
    cols_old = np.repeat(np.arange(n), lens)
    new_c, new_r = pc[cols_old], pr_[I["rowidx"]]
    order = np.lexsort((new_r, new_c))
    colptr = np.zeros(n + 1, np.int32)
    np.add.at(colptr, new_c + 1, 1)
    b = np.zeros(n)
    b[pc] = I["b"]
    return dict(n=n, l=l, nnz=I["nnz"], colptr=np.cumsum(colptr).astype(np.int32), rowidx=new_r[order].astype(np.int32), b=b)


def byte_model(I):
    """Algorithmic bytes per outer iteration = B_fixed + K * B_pcg (SURVEY.md section 8d / BASELINE.md section 3)."""
    n, l, nnz = I["n"], I["l"], I["nnz"]
    m_E = 12 * nnz + 4 * (l + 1)
    m_Et = 12 * nnz + 4 * (n + 1)
    b_pcg = m_E + m_Et + 8 * (13 * n + 2 * l)
    b_fixed = 3 * m_E + 2 * m_Et + 8 * (30 * n + 12 * l)
    return b_fixed, b_pcg


def host_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        cc = subprocess.check_output(["gcc", "--version"], text=True).splitlines()[0]
    except Exception:
        cc = "gcc (version unknown)"
    return model, cc + " -O3 -ffp-contract=off (oracle/Makefile; the reference builds with g++ -O3, LP/cython_solver/Makefile:9)"


def _eigen_solve(I, log=None):
    from oracle import oracle as O
    s = O.LpOracle(0, order=O.ORDER_EIGEN)
    s.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"])
    s.solve_init()
    if log:
        s.set_log(log)
    s.solve_iter(0, MAX_ITERS)
    return -s.cal_Obj(), s.total_outer_iters


def usable_cpus():
    """CPUs this process may really use: the affinity mask, capped by a cgroup-v2 CPU quota if one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def _pool_pass(insts, count, workers):
    """`count` solves (the batch cyclically) by `workers` single-thread processes, one instance per process at a time."""
    from concurrent.futures import ProcessPoolExecutor
    tasks = [insts[i % len(insts)] for i in range(count)]
    t0 = time.perf_counter()
    with ProcessPoolExecutor(workers) as ex:
        res = list(ex.map(_eigen_solve, tasks, chunksize=1))
    return res, time.perf_counter() - t0


def cpu_baseline_lp(insts, sample, pool_count, with_log=True):
    """The CPU oracle (a port: the reference's Eigen path cannot be built here) in Eigen's reduction order, logging off:
    (i) 1 thread on a bounded sample, (ii) one single-thread process per usable host core (`nproc` processes) over
    `pool_count` solves, (iii) the same with 16 processes (the figure rounds 1-2 reported as "all cores").
    Runs BEFORE the GPU is initialised (the pools fork)."""
    from oracle import oracle as O
    O.build()
    model, cc = host_info()
    t0 = time.perf_counter()
    one = [_eigen_solve(I) for I in insts[:sample]]
    dt1 = time.perf_counter() - t0
    it1 = sum(r[1] for r in one)
    ncpu = usable_cpus()
    host = os.cpu_count() or ncpu
    quota = (f"this job may use {ncpu} of the host's {host} hardware threads (affinity mask / cgroup cpu.max quota)" if ncpu < host
             else f"all {host} hardware threads of the host are usable")
    out = dict(value=it1 / dt1, unit="instance-iterations/s", cores=1, kind="port",
               sample=f"first {sample} instances of the batch solved to convergence by oracle/lpbox_oracle.c (Eigen reduction order, "
                      f"logging off), {it1} iterations in {dt1:.1f} s", cpu_model=model, compiler=cc, host_cores=host,
               usable_cores=ncpu, cpu_quota=quota)
    res = one
    if pool_count > sample and ncpu > 1:
        # enough solves that the tail (the slowest instance of the batch runs 1.5 x the mean) does not dominate a wide pool
        count = max(pool_count, 4 * ncpu) if ncpu > 64 else pool_count
        res, dtp = _pool_pass(insts, count, ncpu)
        out["all_cores"] = dict(value=sum(r[1] for r in res) / dtp, unit="instance-iterations/s", cores=ncpu, processes=ncpu,
                                sample=f"{count} solves (the batch's first {min(count, len(insts))} instances"
                                       + (", cyclically" if count > len(insts) else "") + f") by {ncpu} single-thread processes = one "
                                       f"per usable host core ({quota}), one instance per process at a time, {dtp:.1f} s")
        if ncpu < host:
            # SURVEY 8d(ii) asks for nproc processes; under a CPU quota more processes only time-share the same cores.  The
            # whole-host figure can therefore only be extrapolated -- an upper bound: SMT siblings and memory do not scale linearly
            out["all_cores"]["extrapolated_to_host_cores"] = dict(
                value=out["all_cores"]["value"] * host / ncpu, cores=host,
                note=f"NOT measured: the {ncpu}-core figure x {host}/{ncpu}, linear-scaling upper bound for the whole host")
        if ncpu > 16:
            r16, dt16 = _pool_pass(insts, min(pool_count, 64), 16)
            out["processes_16"] = dict(value=sum(r[1] for r in r16) / dt16, unit="instance-iterations/s", cores=16, processes=16,
                                       sample=f"first {len(r16)} instances by 16 single-thread processes, {dt16:.1f} s "
                                              "(what rounds 1-2 printed under the name all_cores)")
    if with_log:
        # footnote: what ./test really does -- the reference's per-iteration text log is on by default (LPh:148, LPcpp:1013-1067)
        import tempfile
        k = max(1, min(sample, 8))
        with tempfile.TemporaryDirectory() as td:
            t0 = time.perf_counter()
            itl = sum(_eigen_solve(I, os.path.join(td, "log_%d.txt" % j))[1] for j, I in enumerate(insts[:k]))
            dtl = time.perf_counter() - t0
        out["with_reference_default_log"] = dict(value=itl / dtl, unit="instance-iterations/s", cores=1,
                                                 sample=f"first {k} instances again with the per-iteration log (7 norms, 12 lines per iteration) "
                                                        f"appended to a temporary file, {dtl:.1f} s")
    out["note"] = "value / all_cores: logging off, as in every solver object this repository creates"
    k = min(len(res), len(insts))
    return out, np.array([r[0] for r in res[:k]]), np.array([r[1] for r in res[:k]])


def pmc_traffic(kernel_tag, instances):
    """HBM bytes per launch measured by a rocprofv3 --pmc pass (tools/collect_profiles.sh -> profiles/rNN_pmc_traffic.json); only a
    record of the same kernel and batch is reported, with its source -- never a stale constant."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if kernel_tag in str(d.get("kernel", "")) and int(d.get("instances", -1)) == int(instances):
            return d.get("hbm_bytes_per_launch"), os.path.relpath(path, ROOT)
    return None, None


def dist_setup(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    return rank, local_rank, world


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes.  This parent has not touched the GPU
    (no HIP call, no torch import) and never will; it waits for the ranks, relays rank 0's JSON line and returns the worst exit code."""
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # a rank that dies early would leave the others waiting in the rendezvous or a barrier: watch all of them, and if one fails
    # stop the rest instead of hanging until a collective times out
    import threading
    out_chunks = []
    reader = threading.Thread(target=lambda: out_chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
        time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        sys.stderr.write(f"bench.py: rank {failed} exited with code {procs[failed].returncode}; the other ranks were stopped\n")
    reader.join(timeout=10)
    sys.stdout.write("".join(c for c in out_chunks if c))
    sys.stdout.flush()
    return max(abs(p.returncode or 0) for p in procs) if failed is None else (abs(procs[failed].returncode) or 1)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=None, choices=(2, 3, 4, 5),
                    help="default: config 2 as the headline + short passes of 4, 3, 5 under detail.configs (one GPU)")
    ap.add_argument("--batch", type=int, default=256, help="instances per GPU (configs 2 and 4)")
    ap.add_argument("--cpu-sample", type=int, default=None, help="instances solved by ONE CPU thread for cpu_baseline (0 = skip)")
    ap.add_argument("--cpu-pool", type=int, default=None, help="instances solved by the all-cores CPU pass")
    ap.add_argument("--variables", type=float, default=1e6, help="variables of the config 5 instance")
    ap.add_argument("--no-extra-configs", action="store_true", help="default run: headline only, no detail.configs")
    args = ap.parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, argv)
    rank, local_rank, world = dist_setup(args)
    if os.environ.get("LPBOX_BENCH_FAIL_RANK") == str(rank) and STUB:           # test hook (tests/test_bench_launch.py): a rank that dies at start
        raise SystemExit(3)
    headline = args.config or 2
    extras = [] if (args.config is not None or world > 1 or args.no_extra_configs or STUB) else [4, 3, 5]

    def args_for(c, nested):
        a = argparse.Namespace(**vars(args))
        a.config = c
        if nested or a.steps is None:
            a.steps = {2: 5, 4: 2, 3: 5, 5: 3}[c] if not nested else {4: 2, 3: 3, 5: 2}[c]
        if nested:
            a.warmup = 1
        return a

    # CPU legs of the LP configs first: their process pools fork, which must happen before this process initialises the GPU
    cpu = {}
    if world == 1 and rank == 0:
        for c in [headline] + extras:
            if c in (2, 4):
                cpu[c] = cpu_leg_lp(args_for(c, c != headline), nested=c != headline)
    ctx = gpu_dist_init(rank, local_rank, world)
    line = RUN[headline](args_for(headline, False), ctx, cpu.get(headline))
    if rank == 0 and extras:
        line["detail"]["configs"] = {}
        for c in extras:
            try:
                sub = RUN[c](args_for(c, True), ctx, cpu.get(c))
            except Exception as e:                      # the headline stands; say what failed
                sub = {"error": f"{type(e).__name__}: {e}"}
            line["detail"]["configs"][str(c)] = sub
        line["detail"]["configs"]["note"] = ("BASELINE configs[3], [2], [4] measured in this same run AFTER the headline's timed region: "
                                             "each entry is a complete bench line of its own (python bench.py --config C prints it alone)")
    if rank == 0:
        print(json.dumps(line), flush=True)
    finish(world)
    return 0


STUB = os.environ.get("LPBOX_BENCH_STUB", "") not in ("", "0")      # tests only: no GPU, tests/bench_stub.py in place of LpBatch


class Ctx:
    pass


def gpu_dist_init(rank, local_rank, world):
    ctx = Ctx()
    ctx.rank, ctx.world = rank, world
    # rehearsal on a one-GPU box only: LPBOX_BENCH_BACKEND=gloo LPBOX_BENCH_DEVICE=0 puts every rank on one card
    backend = os.environ.get("LPBOX_BENCH_BACKEND", "nccl")
    if STUB:
        if backend == "nccl":
            raise SystemExit("LPBOX_BENCH_STUB needs LPBOX_BENCH_BACKEND=gloo")
        import torch.distributed as dist
        import torch
        if world > 1:
            dist.init_process_group(backend)
        gpu_sync = lambda: None
    else:
        import torch
        import torch.distributed as dist
        from lpbox_hip import _lib
        L = _lib.load()
        if L.lpbox_device_count() < 1:
            raise SystemExit("bench.py: no HIP device visible; the HIP path has no CPU fallback")
        if "LPBOX_BENCH_DEVICE" in os.environ:
            local_rank = int(os.environ["LPBOX_BENCH_DEVICE"])
        torch.cuda.set_device(local_rank)
        if world > 1:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
        gpu_sync = torch.cuda.synchronize
    red_dev = "cuda" if backend == "nccl" else "cpu"

    def sync():
        gpu_sync()
        if world > 1:
            dist.barrier()
            gpu_sync()

    def allred(v, op):
        if world == 1:
            return float(v)
        t = torch.tensor([v], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=getattr(dist.ReduceOp, op))
        return float(t.item())

    ctx.local_rank, ctx.sync, ctx.allred, ctx.backend = local_rank, sync, allred, backend
    return ctx


def finish(world):
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------------
# configs 2 and 4: batches of independent LP instances on the persistent one-workgroup-per-instance kernel
# ------------------------------------------------------------------------------------------------------------------------
SHARD_SOURCE = {}       # rank -> how its shard was made (for the `data` field)


def lp_shard(args, rank):
    c4 = args.config == 4
    insts = load_instances(FIXTURE_C4 if c4 else FIXTURE)
    shard = [insts[i % len(insts)] for i in range(args.batch)]
    SHARD_SOURCE[rank] = "fixture"
    if rank > 0:
        items, bids = (500, 2000) if c4 else (100, 500)
        from lpbox_hip import auction
        fx = auction.default_stream_fixture(items, bids)
        first = len(insts) * rank
        if (not STUB and os.environ.get("LPBOX_BENCH_SHARD", "stream") != "relabel" and args.batch == len(insts) and os.path.exists(fx)
                and first + args.batch <= len(np.load(fx)["n"])):
            shard = auction.stream_instances(items, bids, first, args.batch, fixture=fx, workers=max(1, min(16, usable_cpus() // max(1, args.gpus))))
            SHARD_SOURCE[rank] = "stream"
        else:
            shard = [relabel(I, 1000 * rank + i) for i, I in enumerate(shard)]
            SHARD_SOURCE[rank] = "relabel"
    return shard, len(insts)


def cpu_leg_lp(args, nested=False):
    """cpu_baseline of configs 2 / 4 (rank 0 at N = 1 only), before the GPU is initialised."""
    c4 = args.config == 4
    sample = args.cpu_sample if args.cpu_sample is not None else ((3 if nested else 6) if c4 else 32)
    if sample <= 0:
        return None
    shard, _ = lp_shard(args, 0)
    ncpu = usable_cpus()
    pool = args.cpu_pool if args.cpu_pool is not None else (min(args.batch, max(32, ncpu)) if c4 else args.batch)
    return cpu_baseline_lp(shard, sample, pool, with_log=not c4)


def run_lp_batch(args, ctx, cpu_leg=None):
    rank, world, local_rank, sync, allred = ctx.rank, ctx.world, ctx.local_rank, ctx.sync, ctx.allred
    c4 = args.config == 4
    shard, n_distinct = lp_shard(args, rank)
    shard_source = "stream" if world > 1 and allred(1.0 if SHARD_SOURCE.get(rank) in ("stream", "fixture") else 0.0, "MIN") > 0 else "relabel"
    cpu, cpu_obj, cpu_it = cpu_leg if cpu_leg is not None else (None, None, None)
    if STUB:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from bench_stub import StubBatch as LpBatch
    else:
        from lpbox_hip.lp import LpBatch
    batch = LpBatch(shard, device=local_rank)
    cfg = batch.config()

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

    B = args.batch
    ctr = [batch.counters(i) for i in range(B)]                 # of the last step (identical every step: deterministic)
    outer = np.array([c[0] for c in ctr], np.float64)
    pcg = np.array([c[1] for c in ctr], np.float64)
    iters_per_step = float(outer.sum())
    bm = np.array([byte_model(I) for I in shard], np.float64)
    nnz = np.array([I["nnz"] for I in shard], np.float64)
    alg_bytes = float((outer * bm[:, 0] + pcg * bm[:, 1]).sum())
    lds_gather_bytes = float(((5 * outer + 2 * pcg) * nnz * 8).sum())      # (5 + 2K) sparse gathers of nnz f64 per outer iteration
    k_ms, k_launches = batch.kernel_time()
    objs = np.array([-batch.cal_obj(i) for i in range(B)])
    infeasible = int(sum(batch.check_infeasible_l2f(i) > 0 for i in range(B)))

    # Outside the timed region, config 2 only: the opt-in direct x-update (lpbox_set_x_update; NOT the reference's PCG, so it is reported
    # beside the headline, never as `value`) on the same batch
    direct = None
    if not c4 and world == 1 and not STUB:
        try:
            batch.set_x_update("direct")
            batch.solve_init(); batch.solve_iter(0, MAX_ITERS)          # warm
            batch.kernel_time(reset=True)
            batch.solve_init(); batch.solve_iter(0, MAX_ITERS)
            d_ms, _ = batch.kernel_time()
            d_outer = np.array([batch.counters(i)[0] for i in range(B)], np.float64)
            d_objs = np.array([-batch.cal_obj(i) for i in range(B)])
            d_gap = (d_objs - objs) / objs
            direct = {"ms_per_step": d_ms, "instance_iterations_per_s": float(d_outer.sum()) / (d_ms / 1e3),
                      "mean_outer_iters": float(d_outer.mean()), "max_outer_iters": float(d_outer.max()),
                      "us_per_outer_iteration_slowest_instance": 1e3 * d_ms / float(d_outer.max()),
                      "instances_at_max_iters": int(sum(batch.stop(i)[0] == 0 for i in range(B))),
                      "infeasible_instances": int(sum(batch.check_infeasible_l2f(i) > 0 for i in range(B))),
                      "mean_objective": float(d_objs.mean()), "mean_paired_gap_vs_pcg_mode": float(d_gap.mean()),
                      "stderr": float(d_gap.std(ddof=1) / np.sqrt(B)),
                      "note": "exact x-update through an on-chip inverse (DESIGN.md section 17): a different algorithm step than the "
                              "reference's PCG-to-1e-3, bit-exact only against its own oracle mirror; informational"}
        except Exception as e:                                      # a batch the mode does not fit: say so, the headline stands
            direct = {"error": str(e)}
        batch.set_x_update("pcg")

    t_max = allred(dt, "MAX")
    it_total = allred(iters_per_step * args.steps, "SUM")
    line = None
    if rank == 0:
        kernel_s = (k_ms / 1e3) / max(k_launches, 1)
        kname = "lp_window_kernel<%d,%d>" % (cfg["threads"], cfg["elems_per_thread"])
        traffic, traffic_src = pmc_traffic(kname, B)
        lds_gbs = lds_gather_bytes / kernel_s / 1e9
        size = "j=500/k=2000" if c4 else "j=100/k=500"
        line = {
            "metric": "admm_iters_per_sec_batched_lp_" + ("j500_k2000" if c4 else "j100_k500"),
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
            "data": f"synthetic (reference generator, RandomState(0), draws 0..{min(B, n_distinct) - 1} {size} on rank 0"
                    + ("" if world == 1 else (f"; rank r: draws {B} r .. {B} r + {B - 1} of the same stream, regenerated draw for draw by lpbox_hip/auction.py "
                                              "and checked against the reference generator's digests" if shard_source == "stream" else
                                              "; rank r > 0: the same LPs relabelled by a rank-seeded permutation"))
                    + ")" + (" -- STUB SOLVER, launch-path test only, not a measurement" if STUB else ""),
            "config": {"workload": f"batch of {B} {size} combinatorial-auction LP instances per GPU, fp64, full solve "
                                   "(init + ADMM_lp_iters(0,2e4)) = 1 step (BASELINE configs[%d])" % (3 if c4 else 1),
                       "instances_per_gpu": B, "threads_per_instance": cfg["threads"], "slots_per_thread": cfg["elems_per_thread"],
                       "lds_bytes_per_instance": cfg["lds_bytes"], "parallelism": f"instance-sharded x{world}"},
            # The kernel keeps all state in registers/LDS: HBM is touched once per launch, so HBM cannot bound it.  The roof that
            # can is the LDS gather path of the two sparse products; the remaining gap is dependency latency.
            "roofline": {"bound": "lds", "achieved": lds_gbs, "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": lds_gbs / LDS_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kname, "kernel_ms_per_launch": 1e3 * kernel_s,
                         "lds_gather_bytes_per_launch": lds_gather_bytes,
                         "hbm": {"algorithmic_bytes_per_launch": alg_bytes, "effective_GBps": alg_bytes / kernel_s / 1e9,
                                 "measured_GBps": (traffic / kernel_s / 1e9) if traffic else None,
                                 "frac_of_peak_measured": (traffic / kernel_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                                 "peak": HBM_PEAK_GBS,
                                 "note": "effective = SURVEY 8d streaming model / time, informational: state is on-chip, so it "
                                         "may exceed the HBM peak and is NOT a roofline fraction"},
                         "note": "bytes = (5+2K)*nnz*8 per instance-iteration (f64 LDS gathers of E*v and E^T*w) against the guide's "
                                 "aggregate ds_read_b64 rate; issue/latency analysis in DESIGN.md section 5, counters in profiles/"},
            "detail": {"instance_iters_per_step": iters_per_step, "mean_outer_iters": float(outer.mean()),
                       "max_outer_iters": float(outer.max()), "mean_pcg_per_outer": float(pcg.sum() / outer.sum()),
                       "wall_clock_to_converge_ms": 1e3 * t_max / args.steps,
                       # the lottery-free figure of the kernel: which instance is slowest, and for how many
                       # iterations, is a draw of the summation order; time per iteration of that instance is not
                       "us_per_outer_iteration_slowest_instance": 1e6 * kernel_s / float(outer.max()),
                       "mean_objective": float(objs.mean()), "infeasible_instances": infeasible},
        }
        if direct is not None:
            line["detail"]["direct_x_update_mode"] = direct
        if cpu is not None:
            k = len(cpu_obj)
            gap = (objs[:k] - cpu_obj) / cpu_obj
            line["cpu_baseline"] = cpu
            line["detail"]["objective_vs_eigen_order_oracle"] = {
                "instances": k, "gpu_mean_objective": float(objs[:k].mean()), "eigen_order_mean_objective": float(cpu_obj.mean()),
                "mean_paired_gap": float(gap.mean()), "stderr": float(gap.std(ddof=1) / np.sqrt(k)) if k > 1 else None,
                "gpu_better_equal_worse": [int((gap > 0).sum()), int((gap == 0).sum()), int((gap < 0).sum())],
                "eigen_order_mean_outer_iters": float(cpu_it.mean()), "eigen_order_max_outer_iters": float(cpu_it.max()),
                "note": "objective = sum of accepted bid prices (maximisation); the two summation orders end on different, "
                        "statistically equivalent binary solutions (DESIGN.md section 3)"}
        if STUB:
            line["stub"] = True
    batch.close()
    return line if rank == 0 else None


# ------------------------------------------------------------------------------------------------------------------------
# config 3: one full-resolution segmentation MRF
# ------------------------------------------------------------------------------------------------------------------------
def run_seg(args, ctx, cpu_leg=None):
    rank, world, local_rank, sync, allred = ctx.rank, ctx.world, ctx.local_rank, ctx.sync, ctx.allred
    from lpbox_hip.seg import PyLPboxADMMsolver, load_gray
    gray = load_gray(os.path.join(GOLDEN, "seg", "0.jpg"))       # the reference's own sample image 0 (500 x 375)
    s = PyLPboxADMMsolver(0, gray.size, rank)
    s.write_files = False
    s.set_image(gray)
    P = s.get_problem()

    def step():
        s.solve_init()
        return s.solve_iter()

    for _ in range(args.warmup):
        step()
    s.kernel_time(reset=True)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        energy = step()
    sync()
    dt = time.perf_counter() - t0
    o, p = s.counters()
    ms, nl = s.kernel_time()
    n, nnz = P["n"], len(P["colidx"])
    mA = 12 * nnz + 4 * (n + 1)
    bytes_solve = o * (3 * mA + 256 * n) + p * (mA + 104 * n)     # B_iter = 3 m_A + 8*32 n + K (m_A + 8*13 n)
    t_max = allred(dt, "MAX")
    it_total = allred(o * args.steps, "SUM")
    line = None
    if rank == 0:
        chain_s = ms / 1e3 / args.steps
        line = {"metric": "admm_iters_per_sec_segmentation_mrf_full_resolution", "value": it_total / t_max, "unit": "iterations/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * t_max / args.steps,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                "data": "the reference's sample image Segmentation/cython/data/0.jpg (committed copy tests/golden/seg/0.jpg), decoded by PIL",
                "config": {"workload": "one VOC2012-size segmentation MRF, n = %d variables, nnz = %d (7-diagonal), full resolution, "
                                       "ADMM_bqp_unconstrained_legacy to convergence = 1 step (BASELINE configs[2])" % (n, nnz),
                           "parallelism": f"replicas x{world}"},
                "roofline": {"bound": "hbm", "achieved": bytes_solve / chain_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": bytes_solve / chain_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                             "kernel": "segmentation launch chain (seg_k_prep/yrhs/resid/matvec/update/post, hipGraph replay)",
                             "kernel_ms_per_launch": 1e3 * chain_s / max(nl / args.steps, 1), "launches_per_solve": nl / args.steps,
                             "chain_ms_per_solve": 1e3 * chain_s, "algorithmic_bytes_per_solve": bytes_solve,
                             "note": "chain-level: algorithmic bytes of a whole solve / stream time of its launches (HIP events on the "
                                     "solver's stream); the 34 MB working set is L2/MALL resident, per-kernel stats in profiles/"},
                "detail": {"outer_iters": o, "pcg_per_outer": p / o, "energy": energy, "us_per_iteration": 1e6 * chain_s / o}}
        if world == 1 and (args.cpu_sample is None or args.cpu_sample > 0):
            from oracle import oracle as O
            so = O.SegOracle(0, gray.size, 0)
            so.set_problem(P)
            so.solve_init()
            t0 = time.perf_counter()
            so.solve_iter()
            dtc = time.perf_counter() - t0
            model, cc = host_info()
            line["cpu_baseline"] = dict(value=so.total_outer_iters / dtc, unit="iterations/s", cores=1, kind="port",
                                        sample=f"the same full solve by oracle/seg_oracle.c: {so.total_outer_iters} iterations in {dtc:.1f} s",
                                        cpu_model=model, compiler=cc)
    s.close()
    return line


# ------------------------------------------------------------------------------------------------------------------------
# config 5: one large LP, variable-sharded over the ranks
# ------------------------------------------------------------------------------------------------------------------------
XGMI_LINK_GBS = 153.0          # /opt/skills/guides/MI355X_MICROARCH.md: 7 xGMI links x ~153 GB/s per GPU, fully connected
RCCL_OP_LATENCY_US = 15.0      # ASSUMED small-message latency of one RCCL operation on an 8-GPU node (10-20 us); no hardware to measure it


def model_8_ranks(args, n, l, K, s_iter_1gpu, device):
    """What the variable-sharded run should do on 8 ranks -- a MODEL put on record before hardware exists (no multi-GPU box was available):
    local work of one rank + the exchanges of one outer iteration x an assumed latency + the bandwidth term of the l-vector exchanges.
    Local work is MEASURED here on one GPU with an instance of n/8 variables (same generator; its row count is l/8, a real shard keeps
    all l rows with 1/8 of their entries -- the l-sized kernels are under-counted, the launch-bound floor is not)."""
    from lpbox_hip.big import BigLp
    from lpbox_hip.synth import make_auction_like
    import torch
    P8 = make_auction_like(max(n // 8, 1000), 0)

    def local(mode):
        g8 = BigLp(P8, device=device, use_torch_stream=True, pcg_mode=mode)
        g8.solve_init(); g8.solve_iter(0, 100)                              # warm (graphs, launch counts)
        g8.solve_init()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g8.solve_iter(0, 100)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        k = g8.scalar("pcg_total") / max(g8.scalar("outer_total"), 1)
        g8.close()
        return dt, k

    local_s, _ = local("reference")
    lean_s, lean_K = local("lean")
    lean_ops = 7 + 3 * lean_K    # comm-lean PCG: K x (E*p with the p.p / q.q partials riding along 2, (r.r, r.z) 1)
    lean_t = lean_s + lean_ops * RCCL_OP_LATENCY_US * 1e-6 + (lean_K + 2) * 2 * (l * 8 / 8) / (XGMI_LINK_GBS * 1e9)
    ops = 7 + 4 * K              # per outer iteration: prep 1, E*y1 2, resid 1, K x (E*p 2, p.Mp 1, (r.r, r.z) 1), post 1, E*x 2
    vec_exchanges = K + 2        # E*v exchanges: row blocks sent to their owner (all 7 links at once), reduced blocks gathered back
    bw_s = vec_exchanges * 2 * (l * 8 / 8) / (XGMI_LINK_GBS * 1e9)
    lat_s = ops * RCCL_OP_LATENCY_US * 1e-6
    t = local_s + lat_s + bw_s
    return {"ms_per_iteration": 1e3 * t, "iterations_per_s": 1.0 / t, "speedup_vs_1_gpu_measured_here": s_iter_1gpu / t,
            "local_work_ms": 1e3 * local_s, "rccl_operations_per_outer_iteration": ops, "latency_ms": 1e3 * lat_s, "bandwidth_ms": 1e3 * bw_s,
            "assumed_us_per_rccl_operation": RCCL_OP_LATENCY_US, "xgmi_link_GBps": XGMI_LINK_GBS,
            "comm_lean_pcg": {"ms_per_iteration": 1e3 * lean_t, "speedup_vs_1_gpu_measured_here": s_iter_1gpu / lean_t,
                              "local_work_ms": 1e3 * lean_s, "rccl_operations_per_outer_iteration": lean_ops,
                              "note": "opt-in pcg_mode='lean' (NOT the reference's arithmetic): 3 instead of 4 dependent exchanges per PCG iteration"},
            "note": "MODEL, not a measurement: the three exchanges of a PCG iteration are sequentially dependent in the reference's "
                    "arithmetic, so their latency adds up; at the assumed latency 8 ranks buy little or nothing over one GPU "
                    "(DESIGN.md section 10)"}


def run_big(args, ctx, cpu_leg=None):
    rank, world, local_rank, sync, allred, backend = ctx.rank, ctx.world, ctx.local_rank, ctx.sync, ctx.allred, ctx.backend
    from lpbox_hip.big import BigLp
    from lpbox_hip.synth import make_auction_like
    n = int(args.variables)
    P = make_auction_like(n, 0)
    g = BigLp(P, rank=rank, world=world, device=local_rank, use_torch_stream=True)
    window = 100

    def step():
        g.solve_init()
        g.solve_iter(0, window)

    for _ in range(args.warmup):
        step()
    sync()
    c0, l0 = g.scalar("collectives"), g.scalar("launches")          # host-side counters, cumulative since the handle was created
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    t_max = allred(dt, "MAX")
    # the opt-in comm-lean PCG beside it (every rank: it has its own exchanges); reported under detail, never as `value`
    gl = BigLp(P, rank=rank, world=world, device=local_rank, use_torch_stream=True, pcg_mode="lean")
    gl.solve_init(); gl.solve_iter(0, window)
    sync()
    cl0, ll0 = gl.scalar("collectives"), gl.scalar("launches")
    t0 = time.perf_counter()
    lean_steps = max(1, min(args.steps, 5))
    for _ in range(lean_steps):
        gl.solve_init(); gl.solve_iter(0, window)
    sync()
    lean_t = allred(time.perf_counter() - t0, "MAX") / (lean_steps * window)
    lean = {"ms_per_iteration": 1e3 * lean_t, "iterations_per_s": 1.0 / lean_t, "pcg_per_outer": gl.scalar("pcg_total") / max(gl.scalar("outer_total"), 1),
            "launches_per_iteration": (gl.scalar("launches") - ll0) / (lean_steps * max(gl.scalar("outer_total"), 1)),
            "exchanges_per_outer_iteration": (gl.scalar("collectives") - cl0) / (lean_steps * max(gl.scalar("outer_total"), 1)),
            "note": "opt-in pcg_mode='lean': p.Mp = dI (p.p) + r4Et (q.q) -- not the reference's arithmetic, outside the parity claim "
                    "(bit-exact against its own oracle mirror, tests/test_big_gpu_parity.py)"}
    gl.close()
    line = None
    if rank == 0:
        o, p = g.scalar("outer_total"), g.scalar("pcg_total")        # of the last step
        per_iter = 1.0 / (args.steps * max(o, 1))
        coll_per_iter, launches_per_iter = (g.scalar("collectives") - c0) * per_iter, (g.scalar("launches") - l0) * per_iter
        nnz, l = len(P["rowidx"]), P["l"]
        mE, mEt = 12 * nnz + 4 * (l + 1), 12 * nnz + 4 * (n + 1)
        K = p / o
        b_iter = 3 * mE + 2 * mEt + 8 * (30 * n + 12 * l) + K * (mE + mEt + 8 * (13 * n + 2 * l))
        s_iter = t_max / (args.steps * window)
        line = {"metric": "admm_iters_per_sec_single_lp_variable_sharded", "value": 1.0 / s_iter, "unit": "iterations/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * t_max / args.steps,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
                "data": "synthetic auction-like LP (lpbox_hip/synth.py, seed 0; the reference generator cannot produce this size)",
                "config": {"workload": "one LP with n = %d variables, l = %d rows, nnz = %d, variable-sharded over %d rank(s); "
                                       "init + %d ADMM iterations = 1 step (BASELINE configs[4])" % (n, l, nnz, world, window),
                           "parallelism": f"variable-sharded x{world}",
                           "exchanges_per_outer_iteration": coll_per_iter},
                "roofline": {"bound": "hbm", "achieved": b_iter / s_iter / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": b_iter / s_iter / 1e9 / HBM_PEAK_GBS / world, "traffic": None,
                             "kernel": "large-instance launch chain (big_k_*)", "algorithmic_bytes_per_iteration": b_iter,
                             "launches_per_iteration": launches_per_iter,
                             "note": "chain-level: algorithmic bytes of an outer iteration (SURVEY 8d) / time per iteration, divided by "
                                     "the number of GPUs; per-kernel stats in profiles/"},
                "detail": {"pcg_per_outer": K, "ms_per_iteration": 1e3 * s_iter, "backend": backend if world > 1 else None}}
        line["detail"]["launches_per_iteration"] = launches_per_iter
        line["detail"]["folded_reductions"] = bool(g.scalar("folded_reductions"))
        line["detail"]["comm_lean_pcg"] = lean
        if world == 1:
            line["detail"]["model_8_ranks"] = model_8_ranks(args, n, l, K, s_iter, local_rank)
        if world == 1 and (args.cpu_sample is None or args.cpu_sample > 0):
            from oracle import oracle as O
            s = O.LpOracle(0)
            s.set_problem(P["n"], P["l"], P["colptr"], P["rowidx"], P["b"])
            s.solve_init()
            t0 = time.perf_counter()
            s.solve_iter(0, 5)
            dtc = time.perf_counter() - t0
            model, cc = host_info()
            line["cpu_baseline"] = dict(value=5 / dtc, unit="iterations/s", cores=1, kind="port",
                                        sample=f"first 5 iterations of the same instance by oracle/lpbox_oracle.c in {dtc:.1f} s",
                                        cpu_model=model, compiler=cc)
    g.close()
    return line


RUN = {2: run_lp_batch, 4: run_lp_batch, 3: run_seg, 5: run_big}

if __name__ == "__main__":
    sys.exit(main())
