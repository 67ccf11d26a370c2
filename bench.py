#!/usr/bin/env python3
"""Headline benchmark: LM iterations/sec (+ final reprojection RMS) on the 64-camera / 100k-point
synthetic point-model problem (BASELINE.json configs[2]), one process per GPU.

    python bench.py --gpus N --steps K --warmup W

A *step* is one Levenberg-Marquardt iteration of the hot path: linearise (residuals + 2x6/2x3 Jacobian
blocks), eliminate the points into the 6C x 6C reduced camera system, Cholesky-solve it, back-substitute the
points, evaluate the candidate, decide.  Inputs are resident in HBM before the timed region starts.  The timed
region runs exactly K iterations from the uploaded start (tolerances off so that the count is exact),
bracketed by barrier + torch.cuda.synchronize; the maximum over ranks is reported.

What runs, in this order, on ONE solver (same buffers, streams, communicator):
  1. a solve with the reference's own options (iterations to converge, final RMS) on a solver of its own,
  2. N > 1 only: a 2-iteration PREFLIGHT whose iteration logs are compared bit for bit across the ranks (a mismatch ends the job
     with a non-zero exit; a stalled in-kernel wait selects the sequential multi-GPU schedule, and the line says so),
  3. the per-kernel pass: K iterations with HIP events around every launch (untimed; fills "kernels"),
  4. untimed filler up to --preload iterations in all, then a DIAGNOSTIC timed region of K iterations: "ms_per_step_steady",
  5. W warm-up iterations, then THE timed region: exactly K iterations — "ms_per_step", and `value` is quoted on it; "warmup" = W.
Round 4 quoted `value` on (4) and reported "warmup": 80 for a command that said 5; now the region the command line describes is
the last thing that runs and the one the headline is quoted on, and everything in front of it is listed ("untimed_iterations_before").
"run_overhead_us": what the timed run cost beyond its K steps (its wall time - K x the median of its own rows' iteration times).

N > 1 (launched by torch.distributed.run): the default config (cfg3) scales WEAKLY — every rank holds all 64 cameras and its own
block of 100k points; `--config cfg4|cfg5` run BASELINE's totals (1M / 500k points) divided over the ranks ("strong").  Per iteration
the packed reduced camera system is all-reduced over RCCL.  `value` is the whole-job rate in units of the N=1 workload:
    value = LM iterations/s x (total observations / 2,000,000)
so at N = 1 it is exactly LM iterations/s on BASELINE's 64-cam x 100k-point problem.
`--comm shm`: the ranks' collectives go through shared memory instead of RCCL (csrc/ba_comm.hpp ShmComm) and the ranks share the
visible GPUs round-robin — the launcher, the bootstrap and one process per rank end to end on a ONE-GPU box; it measures nothing.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz; vector and matrix fp64 share this peak
                           # (the guide's Matrix-cores table has no fp64 row: AMD datasheet figure)
PEAK_SOURCE = "AMD MI355X datasheet: 78.6 TFLOP/s fp64 vector = fp64 matrix (256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz); MI355X_MICROARCH.md has no fp64 row"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="timed LM iterations (50 = Ceres' max_num_iterations default, SURVEY 8d)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--preload", type=int, default=80,
                    help="untimed iterations the GPU has run in all (per-kernel pass + this filler + warm-up) when the timed region starts: "
                         "the clocks take ~30 ms of load to come up; 0: no filler")
    ap.add_argument("--config", default="cfg3", help="cfg2 | cfg3 (default; the metric's config) | cfg5-like via --points")
    ap.add_argument("--points", type=int, default=None, help="points per rank (default: the config's)")
    ap.add_argument("--schur-impl", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--no-events", action="store_true", help="do not bracket kernels with HIP events in the timed region")
    ap.add_argument("--comm", choices=("rccl", "shm"), default="rccl",
                    help="N > 1: rccl (one GPU per rank, the product) | shm (processes sharing the visible GPUs, collectives through shared memory: "
                         "the whole multi-process path on a one-GPU box)")
    ap.add_argument("--mg-pipeline", action="store_true", help="N > 1 over RCCL: opt into the pipelined multi-GPU schedule (RSBA_PIPELINE_MG=1); "
                                                                 "the preflight falls back to the sequential one if it stalls")
    return ap.parse_args()


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU, torch.distributed.run
    with a rendezvous on 127.0.0.1) BEFORE this process touches HIP, relay their output and leave with their exit code.
    Children, never a re-exec: this process only counts devices (which does not initialise the GPU on this image)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n and "shm" not in sys.argv[1:] and "--comm=shm" not in sys.argv[1:]:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (--comm shm shares them between the ranks)" % (n, have))
    if have < 1:
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("RSBA_BENCH_JOB", "%d_%d" % (os.getpid(), int(time.time())))   # names the shared-memory group of --comm shm
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: one rank per GPU" % (args.gpus, world))

    # Kernel arguments in device memory instead of host memory: every workgroup of a launch reads its arguments first thing, and
    # the step's kernels are short — 0.422 -> 0.418 ms per iteration here (a runtime setting, read when HIP initialises, below;
    # DESIGN.md section 7).
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    if world > 1 or os.environ.get("RSBA_FORCE_COMM"):
        # With a communicator in the process (RCCL's queues beside the solver's three streams) the runtime's default of four
        # hardware queues for its stream pool put ~40 us between a launch of the next step's kernels and their start; with
        # eight (or one, or two) it is gone: 0.471 -> 0.426 ms per iteration for the pipelined multi-GPU schedule at one rank,
        # nothing for a solver without a communicator.  Read by the HIP runtime when it initialises — below.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    shm = world > 1 and args.comm == "shm"
    device = local_rank % torch.cuda.device_count() if shm else local_rank
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shm:   # (RCCL refuses two ranks of one communicator on one device: the plumbing goes over gloo)
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        if args.mg_pipeline and not shm:
            os.environ["RSBA_PIPELINE_MG"] = "1"

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    from realsensecalibration_amd import capi
    from realsensecalibration_amd import distributed as rd
    from realsensecalibration_amd import synthetic as syn

    # ---- workload: this rank's shard
    C, P_cfg, k, seed, outl, huber = syn.CONFIGS[args.config]
    # cfg3 (the metric's config) scales weakly: P_cfg points PER RANK.  cfg4 / cfg5 are BASELINE's multi-GPU configs: their named
    # totals (1M / 500k points) are divided over the ranks unless --points says otherwise (round 4 ran 8 x the total there).
    strong = args.points is None and args.config in ("cfg4", "cfg5") and world > 1
    if strong:
        P_total = P_cfg
        lo, hi = rd.shard_range(P_total, rank, world)
        P_rank = hi - lo
    else:
        P_rank = args.points or P_cfg
        P_total = P_rank * world
        lo, hi = rd.shard_range(P_total, rank, world)
    t0 = time.time()
    prob = syn.make_problem(C, P_total, k, seed, point_range=(lo, hi), outlier_frac=outl)
    prob["huber_delta"] = huber
    gen_s = time.time() - t0
    N_rank, N_total = prob["N"], P_total * min(k, C)

    uid = None
    if world > 1:
        if shm:
            box = [os.environ.get("RSBA_BENCH_JOB", "job%d" % os.getppid()) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            uid = ctypes.create_string_buffer(capi.comm_shm_id("bench_%s" % box[0]), 128)
        else:
            uid = rd.broadcast_unique_id(dist, capi, rank)

    def options(**kw):
        o = capi.default_options(device=device, huber_delta=huber, rank=rank, world_size=world, **kw)
        if args.schur_impl is not None:
            o.schur_impl = args.schur_impl
        if uid is not None:
            o.comm_unique_id = ctypes.cast(uid, ctypes.c_void_p)
        return o

    problem = capi.Problem.points(prob)

    # ---- reference run with the reference's own options (tolerances on): iterations to converge, final RMS
    sv = capi.Solver(problem, options())
    s_conv = sv.run()
    _, sumsq = sv.final_costs()
    rms = float(np.sqrt(sumsq / (2.0 * N_total))) if sumsq > 0 else float("nan")
    sv.close()

    # ---- throughput runs: exactly W then exactly K iterations from the uploaded start
    # negative tolerances: no convergence test can fire (0 would still stop on a bitwise-equal candidate cost)
    fixed = dict(function_tolerance=-1.0, parameter_tolerance=-1.0, gradient_tolerance=-1.0,
                 max_num_consecutive_invalid_steps=1 << 30, min_trust_region_radius=0.0)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if shm else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    sv_k = capi.Solver(problem, options(max_num_iterations=args.steps, profile_kernels=0 if args.no_events else 1, **fixed))
    # ---- N > 1: preflight.  Two forced iterations; every rank's iteration log must equal rank 0's bit for bit (the ranks factor the
    # identical all-reduced system and take identical decisions — anything else is a broken collective), and a stalled in-kernel wait
    # of the pipelined multi-GPU schedule selects the sequential one for the whole job: every rank takes the same decision from the
    # same gathered facts.  A rank that fails inside (RCCL error, a peer gone: bounded waits, csrc/ba_comm.hpp) raises and the
    # launcher ends the job with its exit code — never a re-exec of a process that has touched the GPU.
    preflight = None
    if world > 1:
        sv_k.configure_run(2, 0)
        s_pf = sv_k.run()
        mine = dict(log=sv_k.iterations()[:, :8].tobytes(), info=sv_k.schedule_info(), iters=int(s_pf.num_iterations))
        box = [None] * world
        dist.all_gather_object(box, mine)
        same = all(b["log"] == box[0]["log"] and b["iters"] == 2 for b in box)
        stalls = sum(b["info"]["stalls"] for b in box)
        preflight = {"iterations": 2, "logs_bitwise_equal_across_ranks": bool(same), "stalls": int(stalls),
                     "schedule": box[0]["info"]["schedule"], "fell_back_to_sequential": False}
        if not same:
            if rank == 0:
                print("bench.py: preflight FAILED: the ranks' iteration logs differ (schedules: %s)" % [b["info"]["schedule"] for b in box], file=sys.stderr, flush=True)
            sys.exit(3)
        if stalls > 0 and box[0]["info"]["schedule"] == "pipelined_mg":
            os.environ["RSBA_PIPELINE_MG"] = "0"
            sv_k.close()
            sv_k = capi.Solver(problem, options(max_num_iterations=args.steps, profile_kernels=0 if args.no_events else 1, **fixed))
            preflight["fell_back_to_sequential"] = True
    # One solver for all of it (same device buffers, same streams, one communicator).  The per-kernel pass comes FIRST so that the
    # GPU has been under load when the timed regions start: after the host-side set-up the clocks take ~10-30 ms of work to come
    # up (an iteration measured 0.455 ms in the first ten steps behind an idle gap and 0.415 ms forty steps later).
    stats = {}
    ran_before = 2 if world > 1 and not (preflight or {}).get("fell_back_to_sequential") else 0
    if not args.no_events:
        sv_k.configure_run(args.steps, 1)
        sv_k.run()
        stats = sv_k.kernel_stats()
        ran_before += args.steps

    def timed_run():
        """Exactly K iterations, bracketed by barrier + synchronize; the maximum over the ranks."""
        sv_k.configure_run(args.steps, 0 if args.no_events else 2)
        sync()
        t0 = time.perf_counter()
        s = sv_k.run()
        sync()
        dt = max_over_ranks(time.perf_counter() - t0)
        assert s.num_iterations == args.steps, (s.num_iterations, args.steps)
        return s, dt

    # the communicator the timed solver all-reduces over must span exactly --gpus ranks (1 = no communicator)
    rccl_nranks = sv_k.comm_nranks()
    if rccl_nranks != args.gpus:
        raise SystemExit("communicator spans %d rank(s), --gpus %d" % (rccl_nranks, args.gpus))
    # diagnostic region ("ms_per_step_steady"): filler up to --preload untimed iterations in all, then K timed
    elapsed_steady = None
    preload = 0
    if args.preload > 0:
        preload = max(0, args.preload - ran_before)
        if preload > 0:
            sv_k.configure_run(preload, 0)
            s_p = sv_k.run()
            assert s_p.num_iterations == preload, (s_p.num_iterations, preload)
            ran_before += preload
        _, elapsed_steady = timed_run()
        ran_before += args.steps
    # THE region the command line describes: W warm-up iterations, then exactly K timed ones — the last thing that runs.
    # A line that names a schedule must have run it: a stalled in-kernel wait inside the region (a step repeated sequentially after its
    # 0.5 s budget) or a fallback taken there makes the region worthless — one rank: it is run again (at most twice more) on the same
    # solver; a region that still is not clean is printed as such (`timed_region.clean` false) and the process leaves with exit code 4.
    attempts, clean, info_before = 0, False, sv_k.schedule_info()
    while True:
        attempts += 1
        info0 = sv_k.schedule_info()
        if args.warmup > 0:
            sv_k.configure_run(args.warmup, 0)
            s_w = sv_k.run()
            assert s_w.num_iterations == args.warmup, (s_w.num_iterations, args.warmup)
        s_k, elapsed = timed_run()
        assert s_k.num_iterations == args.steps, (s_k.num_iterations, args.steps)
        sched = sv_k.schedule_info()
        clean = sched["stalls"] == info0["stalls"] and sched["fallbacks"] == info0["fallbacks"] and sched["schedule"] == info0["schedule"]
        if clean or world > 1 or attempts >= 3:
            break
        ran_before += args.warmup + args.steps
        print("bench.py: a stall / fallback inside the timed region (%s -> %s): the region is run again" % (info0, sched), file=sys.stderr, flush=True)
    timed_region = {"attempts": attempts, "clean": bool(clean), "stalls_inside": sched["stalls"] - info0["stalls"],
                    "fallbacks_inside": sched["fallbacks"] - info0["fallbacks"], "stalls_before": info_before["stalls"], "fallbacks_before": info_before["fallbacks"]}
    stats_timed = sv_k.kernel_stats()
    # What THE RUN just timed cost beyond its steps (the first step's full point pass, the gradient at the last accepted point, reset and
    # synchronisation, the call itself): its wall time minus K times the median of its own rows' iteration times (the host's stamps
    # between two results; rows 2 .. K - 1).  ms_per_step carries run_overhead_us / K of it.  (Two runs of different lengths do not
    # measure this: far behind convergence the steps themselves get cheaper — (t_2K - t_K) / K was 6 % below the first K steps' period.)
    run_overhead_us = None
    it_times = sv_k.iteration_times(cap=max(256, args.steps + 2))
    if len(it_times) >= 6:
        run_overhead_us = 1e6 * (elapsed - args.steps * float(np.median(it_times[2:-1])))
    sv_k.close()
    if not args.no_events:
        stats.update(stats_timed)  # the roofline kernels: durations measured inside the timed region
    if world > 1:
        box = [None] * world
        dist.all_gather_object(box, sched)
        sched = dict(box[0], stalls=sum(b["stalls"] for b in box), fallbacks=sum(b["fallbacks"] for b in box))
        boxt = [None] * world
        dist.all_gather_object(boxt, timed_region)
        timed_region = dict(timed_region, clean=all(b["clean"] for b in boxt), stalls_inside=sum(b["stalls_inside"] for b in boxt),
                            fallbacks_inside=sum(b["fallbacks_inside"] for b in boxt))

    if rank != 0:
        if world > 1:
            capi.load().rsba_comm_finalize()
            dist.destroy_process_group()
        return

    iters_per_s = args.steps / elapsed
    value = iters_per_s * (N_total / 2_000_000.0) if args.config == "cfg3" else iters_per_s
    out = {
        "metric": "LM iterations/sec (64 cams x 100k pts point model; + final reprojection RMS px)",
        "value": value, "unit": "LM iterations/s (of the 2M-observation workload)" if args.config == "cfg3" else "LM iterations/s", "n_gpus": world, "rccl_nranks": rccl_nranks,
        "steps": args.steps, "warmup": args.warmup, "untimed_iterations_before": ran_before,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "ms_per_step_steady": (1e3 * elapsed_steady / args.steps) if elapsed_steady is not None else None,
        "run_overhead_us": run_overhead_us,   # per RUN, beyond its steps: wall time of the timed run - K x the median of its own rows' iteration times
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %d cams x %d points, %d observations (%d views/point), point model <2,6,3>, "
                               "DENSE_SCHUR-equivalent; %d points per GPU" % (args.config, C, P_total, N_total, k, P_rank),
                   "sharding": ("points by contiguous block, cameras replicated, %s all-reduce of the reduced system" % ("shared-memory (host-staged)" if shm else "RCCL")) if world > 1 else "single GPU",
                   "schur_impl": int(sched["schur_impl"]), "seed": seed},
        "schedule": sched["schedule"], "stalls": sched["stalls"], "fallbacks": sched["fallbacks"], "comm": sched["comm_kind"], "timed_region": timed_region,
        "factorisation_workgroups": sched["chol_workgroups"],   # (33 .. 64 cameras on one rank: six of the diagonal-chain kernel + the border's; resident tiles above 64 cameras)
        "lm_iterations_per_s": iters_per_s, "observations_per_s": iters_per_s * N_total,
        "final_reprojection_rms_px": rms, "iterations_to_converge": int(s_conv.num_iterations),
        "converged_final_cost": s_conv.final_cost, "termination": int(s_conv.termination_type),
        "problem_generation_s": gen_s, "upload_s": s_k.setup_seconds,
    }
    if preflight is not None:
        out["preflight"] = preflight

    # ---- roofline of the dominant kernel, from HIP-event durations recorded in the timed region
    if stats:
        # pipelined solve: the Cholesky kernel is launched first and sleeps on ready flags until its columns exist; its
        # event span (and rocprofv3's duration) includes that sleep, which the kernel measures itself (wall-clock
        # ticks) and reports under ":waiting".  Its own work = span - waiting; that is what the roofline is quoted on.
        waiting = {n.split(":")[0]: v for n, v in stats.items() if n.endswith(":waiting")}
        spans = {n: v for n, v in stats.items() if not n.endswith(":waiting")}
        # averages first (a kernel's span and its waiting may come from different passes: the timed pass records only the
        # two roofline kernels, the per-kernel pass all of them), then own work = average span - average waiting
        def avg(v):
            return v[1] / max(v[0], 1)
        stats = {n: (c, c * (avg((c, ms)) - avg(waiting[n]))) if n in waiting else (c, ms) for n, (c, ms) in spans.items()}
        per = {n: (c, ms / max(c, 1)) for n, (c, ms) in stats.items()}
        views = np.full(P_rank, min(k, C), np.float64)
        schur_flops = syn.schur_flops_per_iteration(views)
        schur_flops_r04 = syn.schur_flops_per_iteration(views, full_diagonal_blocks=True)   # rounds 1-4 counted 216 flop for the diagonal blocks too
        nc = 6 * C
        chol_flops = nc ** 3 / 3.0 + 2.0 * nc ** 2
        b_iter = syn.algorithmic_bytes_per_iteration(C, P_rank, N_rank)
        # HBM traffic per launch: PMC counters of separate rocprofv3 --pmc passes over THIS configuration (tools/profile_r03.sh
        # writes profiles/r03_pmc_<workload>.json); a line for a workload without such a file carries traffic = null
        pmc, pmc_source = {}, None
        wl = args.config if args.points is None else "%s_%d" % (args.config, args.points)
        for cand in ("r06_pmc_%s.json" % wl, "r05_pmc_%s.json" % wl, "r04_pmc_%s.json" % wl, "r03_pmc_%s.json" % wl) + (("r02_pmc.json",) if wl == "cfg3" else ()):
            q = os.path.join(ROOT, "profiles", cand)
            if os.path.exists(q):
                try:
                    pmc = json.load(open(q))
                    sched = pmc.pop("__schedule__", "SEQUENTIAL schedule (counter collection serialises kernels; the pipelined factorisation "
                                                    "would only time out)")
                    pmc_source = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `bench.py` on this workload, %s; "
                                  "FETCH_SIZE x 2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes; a committed constant, not collected by "
                                  "this run" % (cand, sched))
                    break
                except Exception:
                    pmc = {}

        def roof(name):
            ms = per[name][1]
            # the pipelined solve launches the pair kernel and the Cholesky once per camera group: per-launch
            # algorithmic work = per-iteration work / launches per iteration (the launches of an iteration add up to it)
            lpi = max(1.0, round(per[name][0] / float(args.steps)))
            # (the PMC tables carry the kernels' own names: the timer's "k_backsub_candidate" is k_backsub_candidate_proj<...> there)
            prow = pmc.get(name) or next((v for k, v in sorted(pmc.items()) if k.startswith(name)), {})
            traffic = prow.get("hbm_bytes_per_launch")
            if "schur_tiles" in name or "schur_pairs" in name or "linearize_schur" in name:
                ach = schur_flops / lpi / (ms * 1e-3) / 1e12
                # the three factors of `frac` from the same committed counter passes as `traffic` (tools/pmc_to_json.py): the algorithmic
                # multiply-adds as wave instructions / the vector instructions issued; active lanes per instruction; the SIMDs' busy share
                issue = None
                if prow.get("SQ_ACTIVE_INST_VALU"):
                    issue = {"useful_instruction_ratio": (schur_flops / lpi / 2.0 / 64.0) / prow["SQ_ACTIVE_INST_VALU"],
                             "lane_utilisation": prow.get("lane_utilisation"), "valu_busy": prow.get("valu_busy"),
                             "mfma_ops": prow.get("SQ_INSTS_VALU_MFMA_MOPS_F64"), "source": "the counter passes named in traffic_source"}
                return {"kernel": name, "bound": "mfma", "bound_pipe": "fp64_valu", "pipe": "v_fma_f64 (fp64 VALU issue; the kernel executes no MFMA)",
                        "achieved": ach, "peak": FP64_PEAK_TFLOPS, "peak_source": PEAK_SOURCE, "unit": "TFLOP/s",
                        "frac": ach / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": pmc_source if traffic is not None else None,
                        "issue": issue,
                        "avg_launch_us": 1e3 * ms,
                        "algorithmic_flops_per_launch": schur_flops / lpi, "launches_per_iteration": lpi,
                        "frac_with_full_diagonal_blocks": schur_flops_r04 / lpi / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                        "note": "compute-bound point elimination on the fp64 vector pipe: per point with k views k(k-1)/2 off-diagonal blocks "
                                "x 216 flop + k diagonal blocks x 126 flop (their 21 unique entries) + k x 36 (rhs); rounds 1-4 counted the diagonal "
                                "blocks at 216 as well (frac_with_full_diagonal_blocks, +4 %); "
                                "\"bound\" keeps the schema's compute label (hbm | mfma), \"bound_pipe\" / \"pipe\" name the real one: the fp64 vector pipe "
                                "(PMC: zero MFMA ops in this kernel; same 78.6 TFLOP/s peak)"}
            if "reduced_system" in name or "chol_tiles" in name or "chol_step" in name:
                ach = chol_flops / lpi / (ms * 1e-3) / 1e12
                return {"kernel": name, "bound": "mfma", "bound_pipe": "fp64_mfma", "pipe": "v_mfma_f64_16x16x4_f64", "achieved": ach, "peak": FP64_PEAK_TFLOPS,
                        "peak_source": PEAK_SOURCE, "unit": "TFLOP/s",
                        "frac": ach / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": pmc_source if traffic is not None else None,
                        "avg_launch_us": 1e3 * ms,
                        "algorithmic_flops_per_launch": chol_flops / lpi, "launches_per_iteration": lpi,
                        "note": "dense (6C)^3/3 + 2(6C)^2 Cholesky solve: a chain of 32 x 32 factorisations with matrix-core updates around it (six workgroups up to 64 cameras, one resident workgroup per 64 x 64 tile above), latency-bound by construction"}
            n_red = 6 * C
            # the block back-substitution of the reduced system reads the factor's lower triangle once (and x, y)
            share = {"k_point_pass": 24 * N_rank + 48 * P_rank, "k_backsub_candidate": 24 * N_rank + 48 * P_rank,
                     "k_backsub_chain": 8 * (n_red * (n_red + 1) // 2 + 3 * n_red), "k_backsub_multi": 8 * (n_red * (n_red + 1) // 2 + 3 * n_red),
                     "k_chol_finish": 8 * (n_red * (n_red + 1) // 2 + 3 * n_red)}.get(name, b_iter)
            ach = share / lpi / (ms * 1e-3) / 1e9
            return {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "peak_source": "MI355X_MICROARCH.md: HBM3E 8.0 TB/s",
                    "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": pmc_source if traffic is not None else None,
                    "avg_launch_us": 1e3 * ms, "algorithmic_bytes_per_launch": share / lpi,
                    "launches_per_iteration": lpi}

        # dominant kernel = most CU-time: duration x share of the chip it occupies (the one-workgroup kernels hold 1 of
        # 256 CUs; the pipelined Cholesky runs beside k_schur_tiles on its reserved CU)
        one_wg = ("k_reduced_system_solve", "k_finish_candidate", "k_chol_finish", "k_publish_result", "k_finish_linearize")
        # per ITERATION (average launch x launches per iteration): the timed region samples the roofline kernel on every
        # fourth step only, so totals over the recorded launches are not comparable between kernels
        def cu_time(kv):
            n, (c, _) = kv
            return per[n][1] * max(1.0, round(c / float(args.steps))) * (1.0 / 256.0 if n in one_wg else 1.0)
        order = sorted(stats.items(), key=lambda kv: -cu_time(kv))
        out["roofline"] = roof(order[0][0])
        it_s = elapsed / args.steps
        out["roofline"]["iteration_hbm_view"] = {"algorithmic_bytes_per_iteration": b_iter, "achieved_GBps": b_iter / it_s / 1e9,
                                                 "frac_of_8TBps": b_iter / it_s / 1e9 / HBM_PEAK_GBS}
        rest = sorted(order[1:], key=lambda kv: -per[kv[0]][1] * max(1.0, round(kv[1][0] / float(args.steps))))  # the others by plain duration per iteration
        out["roofline_other_kernels"] = [roof(n) for n, _ in rest[:3]]
        out["kernels"] = {n: {"launches": int(c), "avg_us": 1e3 * a, "measured": "timed region" if n in stats_timed else "per-kernel pass"}
                          for n, (c, a) in sorted(per.items())}
        out["kernel_timing"] = ("HIP events on the launching stream; in the timed region only the dominant kernel (k_schur_tiles) is "
                                "recorded, on every fourth LM step (an event record costs ~10 us of host time between two launches, "
                                "which is on the critical path of a 0.43 ms step); the other kernels come from an identical K-step pass "
                                "with every launch recorded, run on the same solver before the warm-up")
        for n, (c, ms) in waiting.items():
            if n not in out["kernels"]:
                continue
            out["kernels"][n]["avg_span_us"] = 1e3 * spans[n][1] / max(spans[n][0], 1)
            out["kernels"][n]["avg_waiting_us"] = 1e3 * ms / max(c, 1)
            out["kernels"][n]["note"] = "avg_us = span - waiting (launched ahead of its inputs, sleeps on a flag inside the kernel)"

    # ---- CPU baseline: the oracle (a port of the Ceres-1.14 path; real Ceres cannot be built here) on this box
    if not args.no_cpu_baseline and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        o = oracle_lib.load()
        ncpu = len(os.sched_getaffinity(0))
        # what this process may really use: the cgroup's CPU quota (this pool's boxes show 256 hardware threads and grant 16 CPUs
        # — 256 OpenMP threads then time-slice and spin at every barrier: 0.6 it/s where 64 threads gave 7.4)
        quota = ncpu
        try:
            q, per_ = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                quota = max(1, min(ncpu, int(round(float(q) / float(per_)))))
        except Exception:
            pass
        res = {}
        # 1 thread is what the reference runs (Solver::Options::num_threads default, bundle_adjustment_manager.cpp:90-92);
        # all host cores is SURVEY 8(d)'s second figure: the granted CPUs, and 2x / 4x that many threads (the elimination's
        # dynamic schedule gains from oversubscription while the quota is not the bound), the best of them reported
        for nt in sorted({1, quota, min(ncpu, 2 * quota), min(ncpu, 4 * quota)}):
            oo = o.options(max_num_iterations=args.cpu_iters, num_threads=nt, function_tolerance=-1.0, parameter_tolerance=-1.0,
                           gradient_tolerance=-1.0, huber_delta=huber)
            x_cpu, s_cpu, log_cpu = o.solve_points(prob, oo)
            res[nt] = s_cpu.num_iterations / s_cpu.minimizer_seconds
        best = max(res, key=lambda n: res[n])
        # the same forced iterations on the GPU from the same start: parity of the hot path at the benchmark's full size
        # (the oracle is the checker here, never the thing measured)
        sv_c = capi.Solver(problem, options(max_num_iterations=args.cpu_iters, **fixed))
        s_gpu = sv_c.run()
        log_gpu = sv_c.iterations()
        sv_c.download()
        x_gpu = np.array(problem.params, copy=True)
        sv_c.close()
        blocks = [slice(6 * c, 6 * c + 6) for c in range(C)] + [slice(6 * C + 3 * j, 6 * C + 3 * j + 3) for j in range(0, P_rank, max(1, P_rank // 2000))]
        rel = max(float(np.abs(x_gpu[b] - x_cpu[b]).max() / max(np.abs(x_cpu[b]).max(), 1e-12)) for b in blocks)
        out["full_size_parity"] = {"forced_iterations": int(s_gpu.num_iterations), "final_cost_gpu": s_gpu.final_cost, "final_cost_oracle": s_cpu.final_cost,
                                   "final_cost_rel_diff": abs(s_gpu.final_cost - s_cpu.final_cost) / s_cpu.final_cost,
                                   "max_rel_diff_per_parameter_block": rel,
                                   "same_accept_reject_sequence": bool(np.array_equal(np.asarray(log_gpu)[:, 7], np.asarray(log_cpu)[:, 7])),
                                   "note": "all camera blocks and every %d-th point block; tolerance of the parity tests: 1e-6" % max(1, P_rank // 2000)}
        out["cpu_baseline"] = {"value": res[best], "unit": "LM iterations/s", "cores": int(best), "kind": "port",
                               "sample": "%d LM iterations of the same %s problem (oracle/: Jet AutoDiff + Schur + dense LLT, "
                                         "-O3 -march=native, OpenMP over points), the best of %s threads; single thread = %.4f it/s"
                                         % (args.cpu_iters, args.config, " / ".join(str(k_) for k_ in sorted(res)), res[1]),
                               "single_thread_value": res[1], "all_cores_value": res.get(quota), "all_cores_threads": quota,
                               "by_threads": {str(k_): v_ for k_, v_ in sorted(res.items())}, "host_hardware_threads": ncpu,
                               "cpu_quota_cores": quota}
        out["speedup_vs_cpu_baseline"] = iters_per_s / res[best]
    # RCCL prints a version banner through C stdio, which sits in libc's buffer until exit when stdout is a pipe and
    # would land AFTER the JSON line: push it out first so that the JSON is the last line of stdout
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)
    if world > 1:
        capi.load().rsba_comm_finalize()
        dist.destroy_process_group()
    if not timed_region["clean"]:
        print("bench.py: the timed region was not clean (%s): the line above does not measure the schedule it names" % timed_region, file=sys.stderr, flush=True)
        sys.exit(4)


if __name__ == "__main__":
    main()
