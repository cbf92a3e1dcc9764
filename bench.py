#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X simplex pivot path (contract: see the task / DESIGN.md).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload tableau|tableau1k|batch]

The workload is BASELINE.json's metric configuration: gen_primal_columns(m=20000, n=40000, seed 0), fp64 tableau
resident in HBM (condensed: the 20001 x 20001 nonbasic part, 3.2 GB), slack starting basis, the reference's pivot rules.
A "step" is ONE PASS OF THE HOT PATH OVER THE TABLEAU.  The large-tableau loop defers its updates
(lpx_tableau_blocked.hpp): the decisions of 32 pivots (pricing -> column + ratio test -> eta / pivot row) are taken
from the stale tableau plus the pending (eta column, pivot row) pairs, then one kernel streams the tableau through HBM
once and applies the 32 rank-1 updates to every element in order -- so a step is 32 pivots (`config.pivots_per_step`),
`value` is pivots/s = steps x pivots_per_step / time, and every pivot is the reference's pivot (the run is parity-gated
against the oracle's eager loop).  Tableaus that run one update per pivot (the 1000 x 2000 leg, N > 1) have
pivots_per_step = 1 or do 32 eager pivots per step.  N=1: unpartitioned.  N>1: the tableau is column-partitioned over
the N ranks (one process per GPU, torch.distributed / RCCL) and every pivot does one all-gather of the ranks'
proposals ("scaling": "strong", total work fixed).  The timed steps are enqueued back to back with no host
synchronisation.

Rank 0 prints ONE JSON line with `roofline` (the kernel that streams the tableau, HIP events on the launch stream) and,
at N=1, `cpu_baseline` (the oracle's tableau loop on the host cores over a bounded number of pivots of the
same workload; its pivot log is also the parity check of the GPU run's first pivots).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="tableau", choices=["tableau", "tableau1k", "batch", "revised1k", "revised10k", "revised20k", "dual10k", "dual1k", "config5"])
    ap.add_argument("--cpu-pivots", type=int, default=None, help="pivots of the CPU baseline sample (0 = skip)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--m", type=int, default=None, help="tableau workloads: rows (default: the config's)")
    ap.add_argument("--n", type=int, default=None, help="tableau workloads: columns")
    ap.add_argument("--headline-only", action="store_true", help="default workload: skip the nested batch / tableau1k legs")
    ap.add_argument("--batch-scaling", default="strong", choices=["strong", "weak"],
                    help="batch workload at N > 1: 4096 LPs in total (BASELINE config 3) or 4096 per GPU")
    return ap.parse_args()


_JSON_FD = None


def emit(out):
    """The one JSON line, on the process's original stdout."""
    line = (json.dumps(out) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.buffer.write(line)
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def main():
    global _JSON_FD
    args = parse()
    # RCCL writes its version banner to stdout from C when the first communicator is created: everything but the
    # JSON line goes to stderr
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    import linprog_amd  # noqa: F401
    from linprog_amd import _lib as _lpx

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    # rehearsal knobs (one-GPU box): LPX_BENCH_BACKEND=gloo LPX_BENCH_ONE_DEVICE=1 put every rank on cuda:0
    backend = os.environ.get("LPX_BENCH_BACKEND", "nccl")
    if os.environ.get("LPX_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    _lpx.selfcheck()  # every kernel instance of liblpx resolves on this device (an error, not an abort, if one does not)
    group = None
    # LPX_BENCH_FORCE_DIST=1: also build the process group at world size 1 (a one-GPU box can then run the RCCL
    # collectives of the partitioned path with a single rank)
    if world > 1 or os.environ.get("LPX_BENCH_FORCE_DIST"):
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            # RCCL's kernels on a high-priority stream: the per-pivot all-gather has to find CUs while the update
            # kernel's 10^5..10^6 short workgroups keep the chip full
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), pg_options=opts)
            except (AttributeError, TypeError):
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    def finish(out):
        if rank == 0 and out is not None:
            emit(out)
        if group is not None:
            dist.barrier()  # rank 0's oracle checks are done: nobody tears the communicator down under a busy peer
            dist.destroy_process_group()

    if args.workload == "batch":
        return finish(bench_batch(args, rank, world, group, args.steps, args.warmup))
    if args.workload.startswith("revised"):
        return bench_revised(args)
    if args.workload == "config5":
        return bench_config5(args)
    if args.workload in ("dual10k", "dual1k"):
        return bench_dual(args)
    if args.workload == "tableau1k":
        return finish(bench_tableau(args, rank, world, group, args.m or 1000, args.n or 2000, args.steps, args.warmup))

    # default: the headline leg (20000 x 40000 tableau) and, in the same JSON line, the metric's other legs as nested
    # records with their own roofline / cpu_baseline: "batch" = config 3 (4096 LPs of 64 x 128, sharded over the ranks),
    # "tableau1k" = config 2 (1000 x 2000, one GPU).  --headline-only skips the nested legs.
    out = bench_tableau(args, rank, world, group, args.m or 20000, args.n or 40000, args.steps, args.warmup, blocks=True)
    if not args.headline_only and not (args.m or args.n):
        import torch

        torch.cuda.empty_cache()
        sub = bench_batch(args, rank, world, group, 40, 5)
        if rank == 0:
            out["batch"] = sub
        if world == 1:
            sub = bench_tableau(args, rank, world, None, 1000, 2000, 4000, 100)
            if rank == 0:
                out["tableau1k"] = sub
    finish(out)


# AMD's published MI355X figure, 78.6 TFLOP/s vector fp64 = 39.3 T fma/s (MI355X_MICROARCH.md has no fp64 row; on random
# operands the chip sustains 26-28 T fma/s: tools/fma_rate.hip)
FP64_FMA_PEAK = 78.6e12 / 2


def bench_tableau(args, rank, world, group, m, n, K, W, blocks=False):
    """One leg: K timed steps of the tableau loop on gen_primal_columns(m, n).  blocks: a step is one pass over the
    tableau = 32 pivots (the deferred-update block; 32 eager pivots where the loop runs one update per pivot); else a
    step is one pivot.  Returns the record (every rank)."""
    import ctypes as C

    import torch
    import torch.distributed as dist

    from linprog_amd import _lib, problems
    from linprog_amd.tableau import PrimalTableauSimplexSolver

    backend = os.environ.get("LPX_BENCH_BACKEND", "nccl")
    # the tableau carries the nonbasic columns only (condensed form): with the slack starting basis these are the n - m
    # structural columns, partitioned evenly over the ranks; LPX_TABLEAU_CONDENSED=0: every column, n / world per rank
    condensed = PrimalTableauSimplexSolver.CONDENSED
    ncarry = n - m if condensed else n
    j0, j1 = ncarry * rank // world, ncarry * (rank + 1) // world
    t_setup = time.time()
    A_cols, b, c, basis = problems.gen_primal_columns(m, n, args.seed, j0, j1)
    PPS = 32 if blocks else 1  # pivots per step
    Solver = type("BenchTableau", (PrimalTableauSimplexSolver,), {"_LOG_CAP": PPS * (W + 2 * K) + 8})
    SUSTAIN_S = float(os.environ.get("LPX_BENCH_SUSTAINED_S", "2.0")) if blocks else 0.0
    solver = Solver(c, A_cols, b, basis, group=group, columns=(j0, j1), n_global=n)
    eng = solver._engine
    lib = _lib.load()
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup
    kb = int(lib.lpx_tableau_block_size(C.byref(eng.s)))  # pivots per pass over T (this rank's slice), 0: one update per pivot
    if blocks and kb not in (0, PPS):
        PPS = kb
    pipelined = bool(kb) and group is None and bool(lib.lpx_tableau_pipelined(C.byref(eng.s)))
    chain_kind = int(lib.lpx_tableau_chain_kind(C.byref(eng.s))) if group is None else -1
    small_chain = chain_kind == _lib.CHAIN_SMALL
    one_launch = chain_kind == _lib.CHAIN_FUSED
    K_piv, W_piv = K * PPS, W * PPS

    def barrier():
        if group is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The update kernel is timed with HIP events on the launch stream.  At HBM-bound sizes they are recorded inside the
    # timed region.  Launch-bound sizes are replayed as hipGraphs (event records cannot ride along), so there the kernel
    # is timed in a separate eager pass of further pivots afterwards.
    events_in_timed = (m >= 10000 or group is not None) and os.environ.get("LPX_BENCH_K1_EVENTS", "1") != "0"
    _lib.check(lib.lpx_ctrl_begin_solve(eng.s.ctrl, torch.cuda.current_stream().cuda_stream))
    eng.enqueue(W_piv)
    barrier()
    if events_in_timed:
        _lib.check(lib.lpx_profile_k1(1))
    barrier()
    t0 = time.perf_counter()
    eng.enqueue(K_piv)
    barrier()
    dt = time.perf_counter() - t0
    extra = 0
    if not events_in_timed:
        extra = min(K_piv, 500)
        _lib.check(lib.lpx_profile_k1(1))
        eng.enqueue(extra)
        barrier()
    if group is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    k1_ms, k1_n = C.c_double(), C.c_int64()
    _lib.check(lib.lpx_profile_k1_read(C.byref(k1_ms), C.byref(k1_n)))
    _lib.check(lib.lpx_profile_k1(0))
    # a second, LONG window (>= SUSTAIN_S seconds of the same loop, same state carried on): the K-step window above is a
    # few tens of milliseconds
    sustained = None
    if SUSTAIN_S > 0 and group is None:
        S_piv = max(PPS, int(SUSTAIN_S * K_piv / dt) // PPS * PPS)
        barrier()
        t1 = time.perf_counter()
        eng.enqueue(S_piv)
        barrier()
        ds = time.perf_counter() - t1
        extra += S_piv
        sustained = {"value": S_piv / ds, "unit": "pivots/s", "seconds": ds, "pivots": S_piv, "steps": S_piv // PPS,
                     "ms_per_step": ds / (S_piv // PPS) * 1e3,
                     "gate": "same run, state carried on from the timed window: pivot count and RUNNING status checked, the "
                             "log's prefix against the oracle (cpu_baseline.parity), x_B >= 0 and the cost below the timed "
                             "window's afterwards"}
    # The per-pivot form of the same update (SURVEY 8(d)'s figure, north_star's ">= 70 % of peak HBM bandwidth on the pivot
    # update"): a few more pivots of the SAME run with the deferred chain switched off, i.e. one streaming rank-1 update of
    # the whole tableau per pivot (k_rank1), timed with HIP events like the pass above.  Same arithmetic, same pivots.
    eager = None
    if blocks and kb and group is None and os.environ.get("LPX_BENCH_EAGER", "1") != "0":
        E_piv = 12
        knobs = {"LPX_TABLEAU_BLOCK_MIN_MB": str(1 << 30),  # liblpx reads these per call: no tableau is that large,
                 "LPX_TABLEAU_FUSED": "0"}                  # and no two-buffer chain: the plain column -> prepare -> K1 loop
        saved = {k: os.environ.get(k) for k in knobs}
        os.environ.update(knobs)
        try:
            if int(lib.lpx_tableau_block_size(C.byref(eng.s))) == 0 and int(lib.lpx_tableau_chain_kind(C.byref(eng.s))) == _lib.CHAIN_EAGER:
                eng.enqueue(2)  # the chain's own first launches (pricing scan of a fresh call)
                barrier()
                _lib.check(lib.lpx_profile_k1(1))
                t2 = time.perf_counter()
                eng.enqueue(E_piv)
                barrier()
                de = time.perf_counter() - t2
                e_ms, e_n = C.c_double(), C.c_int64()
                _lib.check(lib.lpx_profile_k1_read(C.byref(e_ms), C.byref(e_n)))
                _lib.check(lib.lpx_profile_k1(0))
                extra += 2 + E_piv
                if e_n.value > 0:
                    e_avg = e_ms.value / 1e3 / e_n.value
                    eb = 16.0 * (m + 1) * (ncarry + 1)
                    eager = {"kernel": "k_rank1<tableau>: ONE rank-1 update of the whole tableau per pivot (the deferred chain switched "
                                       "off for these pivots; same run, same arithmetic)",
                             "pivots": E_piv, "pivots_per_s": E_piv / de, "launches": e_n.value, "avg_launch_ms": e_avg * 1e3,
                             "bytes_per_launch": eb, "achieved": eb / e_avg / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": eb / e_avg / 1e9 / HBM_PEAK_GBS,
                             "note": "16 B per element and pivot: SURVEY 8(d)'s algorithmic bytes for the layout that ran"}
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    ctrl = eng.read_ctrl()
    pivots_done = int(ctrl.pivots)
    if pivots_done != W_piv + K_piv + extra or ctrl.status != _lib.RUNNING:
        sys.exit(f"bench invalid: {pivots_done} pivots done of {W_piv + K_piv + extra}, status {ctrl.status}")
    if sustained is not None:
        xb_min = float(eng.T[:m, eng.n].min().item())
        if xb_min < 0.0:
            sys.exit(f"bench invalid: x_B went negative in the sustained window ({xb_min})")
        sustained["min_xB"] = xb_min
        sustained["objective"] = -float(eng.T[m, eng.n].item())

    # roofline of the update kernel on THIS rank's slice: algorithmic bytes = 16 B x (m+1) x (n_local+1) per launch
    n_loc = j1 - j0
    bytes_per_launch = 16.0 * (m + 1) * (n_loc + 1)
    if k1_n.value == 0:
        sys.exit("bench invalid: no launch of the update kernel was timed")
    k1_avg_s = (k1_ms.value / 1e3) / max(1, k1_n.value)
    achieved = bytes_per_launch / k1_avg_s / 1e9
    traffic = traffic_src = None
    tpath = os.path.join(ROOT, "profiles", "tb_flush_traffic.json" if kb else "k1_traffic.json")
    if os.path.exists(tpath) and m == 20000 and n == 40000 and world == 1:
        try:
            rec = json.load(open(tpath))
            # the PMC figure belongs to one layout: use it only for the layout it was measured on
            if 0.9 * bytes_per_launch <= rec.get("hbm_bytes_per_launch", 0) <= 1.2 * bytes_per_launch:
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_src = (f"NOT measured in this run: read from profiles/{os.path.basename(tpath)} (rocprofv3 --pmc FETCH_SIZE / "
                               f"WRITE_SIZE passes of this command, tag {rec.get('tag')}: 2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes)")
        except Exception:
            traffic = None
    out = {
        "metric": f"simplex pivots/sec ({m}x{n} fp64 tableau)",
        "value": K_piv / dt,
        "unit": "pivots/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": dt / K * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"gen_primal_columns(m={m}, n={n}, seed={args.seed}) dense primal tableau simplex, "
                               f"slack start, first-negative-index pricing",
                   "m": m, "n": n,
                   "form": "condensed tableau: (m+1) x (n-m+1), the nonbasic columns and x_B (basic columns are unit "
                           "vectors no pivot changes)" if condensed else "full tableau: (m+1) x (n+1)",
                   "step": (f"one pass of the hot path over the tableau: the decisions of {PPS} pivots from the stale tableau "
                            f"plus the pending (eta column, pivot row) pairs, then ONE kernel applies the {PPS} rank-1 updates "
                            f"to every element in order (deferred updates, bit-identical to one update per pivot)") if kb
                   else (f"{PPS} pivots, one rank-1 update of the whole tableau each" if PPS > 1 else "one pivot"),
                   "chain": ("pipelined: the pass of block k streams T -> T_alt (and back) on the CUs of one stream while block "
                             "k+1 is decided on the CUs of another from the buffer the pass reads plus the pending pairs of "
                             "both blocks (chains of up to 64)") if pipelined
                   else ("in place: decisions, then the pass" if kb else
                         "one launch per block of 8 pivots: one workgroup decides the block (rows' multipliers and columns' "
                         "pivot-row entries in registers, exchanges through LDS) while the other workgroups apply the previous "
                         "block, two buffers" if small_chain else "one update per pivot"),
                   "pivots_per_step": PPS, "pivots_timed": K_piv, "us_per_pivot": dt / K_piv * 1e6,
                   "launches_per_pivot": ((2 if group is None else 3) + 1.0 / kb) if kb else (0.125 if small_chain else
                                                                                                  1 if one_launch else 3),
                   "partition": f"columns/{world}",
                   "setup_s": round(t_setup, 2)},
        "roofline": {"bound": "hbm", "kernel": "k_tb_flush2 (the pass that applies a block of pending rank-1 updates)" if kb
                     else "k_tb_small (one workgroup decides 8 pivots, the others apply the previous 8: T -> T_alt)" if small_chain
                     else "k_t_fused (update + next pivot's pricing / column / ratio test, two buffers)"
                     if one_launch else "k_rank1<tableau>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "bytes_per_launch": bytes_per_launch, "layout_bytes_full": 16.0 * (m + 1) * (n + 1),
                     "avg_launch_ms": k1_avg_s * 1e3, "launches": k1_n.value,
                     "k1_timing": "HIP events in the timed region" if events_in_timed
                     else "HIP events in a separate eager pass (timed region replays hipGraphs)",
                     "whole_pivot_GBs": 16.0 * (m + 1) * (ncarry + 1) * K_piv / dt / 1e9},
    }
    if sustained is not None:
        out["sustained"] = sustained
    if eager is not None:
        out["roofline"]["eager_update"] = eager
    if kb:
        r = out["roofline"]
        del r["whole_pivot_GBs"]
        r["pivots_per_launch"] = kb
        r["note"] = (f"`achieved` = the 16 B per element this kernel really moves (each element read once, written once per "
                     f"launch) / its launch time.  One launch carries {kb} pivots, so SURVEY 8(d)'s per-pivot figure "
                     f"(16 (m+1)(n+1) B per pivot) is NOT what reaches HBM any more: see `per_pivot_formula_GBs`, an "
                     f"equivalent rate, not a bandwidth")
        r["per_pivot_formula_GBs"] = bytes_per_launch * K_piv / dt / 1e9
        fma = float(kb) * (m + 1) * (n_loc + 1) / k1_avg_s
        r["valu"] = {"fma_per_s": fma, "peak_fma_per_s": FP64_FMA_PEAK, "frac": fma / FP64_FMA_PEAK,
                     "note": f"{kb} fp64 fma per element and launch against the vector fp64 peak"}
        r["share_of_step"] = k1_ms.value / 1e3 / max(1, k1_n.value) * k1_n.value / dt
    if not events_in_timed:
        out["roofline"]["note"] = ("launch/latency-bound size (the tableau is " + f"{8e-6 * (m + 1) * (n_loc + 1):.0f}"
                                   " MB: Infinity-Cache resident): the fraction of HBM peak is reported, not claimed as a "
                                   "bandwidth result")
    if rank == 0 and world == 1:
        A_full = A_cols if not condensed else problems.gen_primal_columns(m, n, args.seed, 0, n)[0]
        out["cpu_baseline"] = cpu_baseline(args, m, n, A_full, b, c, basis, solver, condensed)
    elif rank == 0:
        # N>1: rank 0 regenerates the whole matrix and checks the partitioned run's first pivots against the oracle
        full = problems.gen_primal_columns(m, n, args.seed, 0, n)
        kc = args.cpu_pivots if args.cpu_pivots is not None else (100 if m >= 10000 else 400)
        chk = cpu_baseline(args, m, n, full[0], full[1], full[2], full[3], solver, condensed, kc=kc)
        out["parity"] = None if chk is None else chk["parity"]
    del solver, eng
    return out


def cpu_baseline(args, m, n, A, b, c, basis, solver, condensed, kc=None):
    """Oracle tableau loop (C, OpenMP) on the host cores IN THE LAYOUT THE GPU RAN (condensed: nonbasic columns only, so
    both sides move the same bytes per pivot), bounded sample; also the parity gate of the timed run."""
    from oracle import oracle as O

    cores = min(16, os.cpu_count() or 1)
    if kc is None:
        kc = args.cpu_pivots
    if kc is None:
        kc = 640 if m >= 10000 else 4000  # ~10 s of host work at the headline size
    if kc <= 0:
        return None
    os.environ["OMP_NUM_THREADS"] = str(cores)
    ncarry = n - m if condensed else n
    o = O.OracleTableau(c, A, b, basis, log_cap=kc + 1, threads=cores, ld=ncarry + 1 + ((ncarry + 1) % 2), condensed=condensed)
    t0 = time.perf_counter()
    r = o.solve(kc)
    dt = time.perf_counter() - t0
    # the timed GPU run did W + K pivots; the oracle's sample may be longer: parity on the common prefix
    gpu_log = solver.pivot_log()
    both = min(len(gpu_log), len(r.log))
    if both == 0 or r.log[:both].tolist() != gpu_log[:both].tolist():
        sys.exit("bench invalid: GPU pivot sequence differs from the oracle's on the same workload")
    done = len(r.log)
    return {"value": done / dt, "unit": "pivots/s", "cores": cores, "kind": "port",
            "layout": f"{'condensed' if condensed else 'full'} tableau, {m + 1} x {ncarry + 1} (the layout the GPU ran)",
            "sample": f"first {done} pivots of the same workload, oracle tableau loop (C + OpenMP rank-1 update, {cores} threads)",
            "parity": f"GPU and oracle (row, entering, leaving) sequences identical on the first {both} pivots"}


def bench_revised(args):
    """Revised form (explicit B^-1, the drop-in classes' engine): pivots/s of PrimalRevisedSimplexSolver's loop."""
    import ctypes as C

    import torch

    from linprog_amd import _lib, problems
    from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver
    from oracle import oracle as O

    m, n = {"revised1k": (1000, 2000), "revised10k": (10000, 20000), "revised20k": (20000, 40000)}[args.workload]
    K, W = args.steps, args.warmup
    A, b, c, basis = problems.gen_primal_columns(m, n, args.seed, 0, n)
    Solver = type("BenchRevised", (PrimalRevisedSimplexSolver,), {"_LOG_CAP": W + 2 * K + 8})
    solver = Solver(c, A, b, basis)
    eng, lib = solver._engine, _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.lpx_ctrl_begin_solve(eng.s.ctrl, st))
    eng.enqueue(_lib.KIND_PRIMAL, W)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.enqueue(_lib.KIND_PRIMAL, K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    extra = min(K, 200)
    _lib.check(lib.lpx_profile_k1(1))
    eng.enqueue(_lib.KIND_PRIMAL, extra)
    torch.cuda.synchronize()
    k1_ms, k1_n = C.c_double(), C.c_int64()
    _lib.check(lib.lpx_profile_k1_read(C.byref(k1_ms), C.byref(k1_n)))
    _lib.check(lib.lpx_profile_k1(0))
    ctrl = eng.read_ctrl()
    if ctrl.pivots != W + K + extra or ctrl.status != _lib.RUNNING:
        sys.exit(f"bench invalid: {ctrl.pivots} pivots, status {ctrl.status}")
    kc = args.cpu_pivots if args.cpu_pivots is not None else (300 if m <= 1000 else 20 if m <= 10000 else 6)
    cpu = None
    if kc > 0:
        o = O.OracleRevised(c, A, b, basis, log_cap=kc + 1)
        t1 = time.perf_counter()
        r = o.solve(kc)
        dc = time.perf_counter() - t1
        if r.log.tolist() != solver.pivot_log()[:kc].tolist():
            sys.exit("bench invalid: GPU pivot sequence differs from the oracle's")
        cpu = {"value": kc / dc, "unit": "pivots/s", "cores": 1, "kind": "port",
               "sample": f"first {kc} pivots, oracle revised loop (C, single thread)",
               "parity": f"sequences identical on these {kc} pivots"}
    k1_s = k1_ms.value / 1e3 / max(1, k1_n.value)
    iter_bytes = 8.0 * (4 * m * m + m * n / 2)  # 16m^2 update + 8m^2 y + 8m^2 direction + 4mn pricing (float32 A)
    out = {"metric": f"simplex pivots/sec ({m}x{n} revised form, fp64 B^-1)", "value": K / dt, "unit": "pivots/s",
           "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"gen_primal_columns(m={m}, n={n}, seed={args.seed}) PrimalRevisedSimplexSolver loop",
                      "m": m, "n": n, "form": "revised"},
           "roofline": {"bound": "hbm", "kernel": "k_rank1_y<revised> (B^-1 update + the next iteration's y partials)", "achieved": 16.0 * m * m / k1_s / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 16.0 * m * m / k1_s / 1e9 / HBM_PEAK_GBS,
                        "traffic": None, "avg_launch_ms": k1_s * 1e3,
                        "whole_iteration_GBs": iter_bytes * K / dt / 1e9},
           "cpu_baseline": cpu}
    emit(out)


def bench_dual(args):
    """The dual simplex (dual_solvers.py:82-108) at 10000 x 20000 (dual10k) or 1000 x 2000 (dual1k: the launch-bound size, one
    launch per pivot) from gen_dual's dual-feasible slack basis, in both forms: the carried (condensed) tableau and the
    revised form.  Value = the tableau form's pivots/s."""
    import ctypes as C

    import torch

    from linprog_amd import _lib, problems
    from linprog_amd.dual_solvers import DualRevisedSimplexSolver
    from linprog_amd.tableau import DualTableauSimplexSolver
    from oracle import oracle as O

    m, n = (args.m or 1000, args.n or 2000) if args.workload == "dual1k" else (args.m or 10000, args.n or 20000)
    K, W = args.steps, args.warmup
    c, A, b, basis = problems.gen_dual(m, n, args.seed)
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream

    def run(solver, enqueue):
        eng = solver._engine
        _lib.check(lib.lpx_ctrl_begin_solve(eng.s.ctrl, st))
        enqueue(eng, W)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        enqueue(eng, K)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        _lib.check(lib.lpx_profile_k1(1))
        enqueue(eng, min(K, 100))
        torch.cuda.synchronize()
        k1_ms, k1_n = C.c_double(), C.c_int64()
        _lib.check(lib.lpx_profile_k1_read(C.byref(k1_ms), C.byref(k1_n)))
        _lib.check(lib.lpx_profile_k1(0))
        ctrl = eng.read_ctrl()
        if ctrl.pivots != W + K + min(K, 100) or ctrl.status != _lib.RUNNING:
            sys.exit(f"bench invalid: {ctrl.pivots} pivots, status {ctrl.status}")
        return dt, k1_ms.value / 1e3 / max(1, k1_n.value)

    Tab = type("BenchDualTableau", (DualTableauSimplexSolver,), {"_LOG_CAP": W + 2 * K + 8})
    tab = Tab(c, A, b, basis)
    dt_t, k1_t = run(tab, lambda eng, k: eng.enqueue(k))
    log_t = tab.pivot_log()
    ncarry = tab._engine.n
    kb = int(lib.lpx_tableau_block_size(C.byref(tab._engine.s)))  # deferred updates: pivots per pass over the tableau
    one_launch = kb == 0 and tab._engine.T_alt is not None and os.environ.get("LPX_TABLEAU_FUSED", "1") != "0"
    del tab
    torch.cuda.empty_cache()
    Rev = type("BenchDualRevised", (DualRevisedSimplexSolver,), {"_LOG_CAP": W + 2 * K + 8})
    rev = Rev(c, A, b, basis)
    dt_r, k1_r = run(rev, lambda eng, k: eng.enqueue(_lib.KIND_DUAL, k))
    log_r = rev.pivot_log()
    same = min(len(log_t), len(log_r))
    if log_t[:same].tolist() != log_r[:same].tolist():
        sys.exit("bench invalid: the tableau and the revised form took different dual pivots")
    kc = args.cpu_pivots if args.cpu_pivots is not None else (6 if m >= 5000 else min(K, 300))
    cpu = None
    if kc > 0:
        o = O.OracleRevised(c, A, b, basis, kind="dual", log_cap=kc + 1)
        t1 = time.perf_counter()
        r = o.solve(kc)
        dc = time.perf_counter() - t1
        if r.log.tolist() != log_t[:kc].tolist():
            sys.exit("bench invalid: GPU dual pivot sequence differs from the oracle's")
        cpu = {"value": kc / dc, "unit": "pivots/s", "cores": 1, "kind": "port",
               "sample": f"first {kc} pivots, oracle dual revised loop (C, single thread)",
               "parity": f"tableau form, revised form and oracle: sequences identical on these {kc} pivots; the two GPU "
                         f"forms on all {same}"}
    bytes_t = 16.0 * (m + 1) * (ncarry + 1)
    out = {"metric": f"dual simplex pivots/sec ({m}x{n})", "value": K / dt_t, "unit": "pivots/s", "n_gpus": 1, "steps": K,
           "warmup": W, "ms_per_step": dt_t / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"gen_dual(m={m}, n={n}, seed={args.seed}) DualTableauSimplexSolver loop (condensed tableau, "
                                  f"{m + 1} x {ncarry + 1})", "m": m, "n": n,
                      "launches_per_pivot": (2 + 1.0 / kb) if kb else (1 if one_launch else 4),
                      "pivots_per_pass_over_the_tableau": kb if kb else 1,
                      "revised_form_pivots_per_s": K / dt_r, "revised_form_k1_ms": k1_r * 1e3},
           "roofline": {"bound": "hbm", "kernel": "k_tb_flush2 (the pass that applies a block of pending rank-1 updates)" if kb
                        else "k_td_fused (update + next pivot's leaving row / ratio test / column, two "
                             "buffers)" if one_launch else "k_rank1<tableau>", "achieved": bytes_t / k1_t / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": bytes_t / k1_t / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "bytes_per_launch": bytes_t, "avg_launch_ms": k1_t * 1e3},
           "cpu_baseline": cpu}
    emit(out)


def bench_config5(args):
    """Config 5: small box-bounded LPs through BoundedVariablePrimalSimplexSolver (gen_box, m <= 170, the
    reference's own limit) and gen_dual LPs through DualRevisedSimplexSolver and PrimalDualAlgorithm.  A step = one full
    solve of every problem of the set; each GPU result is checked against the oracle's (pivot events / iters / cost)."""
    import torch

    from linprog_amd import problems
    from linprog_amd.dual_solvers import DualRevisedSimplexSolver
    from linprog_amd.special_solvers import BoundedVariablePrimalSimplexSolver, PrimalDualAlgorithm
    from oracle import oracle as O

    sizes = [16, 64, 128, 170]
    cases = []
    for m in sizes:
        box = problems.gen_box(m, 2 * m, 0)
        du = problems.gen_dual(m, 2 * m, 0)
        cases.append(("bounded", m, lambda box=box: BoundedVariablePrimalSimplexSolver(*box).solve(100000),
                      lambda box=box: O.OracleBounded(*box).solve(100000)))
        cases.append(("dual", m, lambda du=du: DualRevisedSimplexSolver(*du).solve(100000),
                      lambda du=du: O.OracleRevised(*du, kind="dual", log_cap=4).solve(100000)))
        if m <= 64:
            cases.append(("primal_dual", m, lambda du=du: PrimalDualAlgorithm(*du[:3]).solve(100000, 100000),
                          lambda du=du: O.OraclePrimalDual(*du[:3]).solve(100000, 100000)))
    detail = {}
    for _ in range(args.warmup):
        for _, _, run, _ in cases:
            run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for _, _, run, _ in cases:
            run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cpu_t = 0.0
    for name, m, run, ora in cases:
        t1 = time.perf_counter()
        res = run()
        torch.cuda.synchronize()
        tg = time.perf_counter() - t1
        t1 = time.perf_counter()
        want = ora()
        tc = time.perf_counter() - t1
        cpu_t += tc
        if res.iters != want.iters or abs(res.cost - want.cost) > 1e-9 * max(1.0, abs(want.cost)):
            sys.exit(f"bench invalid: {name} m={m}: GPU iters/cost {res.iters}/{res.cost} vs oracle {want.iters}/{want.cost}")
        detail[f"{name}_m{m}"] = {"iters": int(res.iters), "gpu_ms": round(tg * 1e3, 3), "oracle_ms": round(tc * 1e3, 3)}
    # the MI355X-shaped form of config 5: MANY box-bounded LPs per launch (BatchedBoundedSolver, one workgroup per LP)
    from linprog_amd import _lib as _l
    from linprog_amd.batched import BatchedBoundedSolver

    nb, mb = 1024, 64
    boxes = [problems.gen_box(mb, 2 * mb, s) for s in range(nb)]
    cB, AB, bB, lbB, ubB, basB = [np.stack([p[k] for p in boxes]) for k in range(6)]
    vsB = np.full((nb, 2 * mb), _l.VS_LOWER, dtype=np.int32)
    bs = BatchedBoundedSolver(cB, AB, bB, lbB, ubB, basB, vsB)
    ctrl0, basis0, vs0, binv0 = bs.ctrl.clone(), bs.basis.clone(), bs.var_state.clone(), bs.Binv.clone()

    def run_batch():
        bs.ctrl.copy_(ctrl0); bs.basis.copy_(basis0); bs.var_state.copy_(vs0); bs.Binv.copy_(binv0)
        bs.enqueue(100000)

    run_batch()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        run_batch()
    torch.cuda.synchronize()
    tb = (time.perf_counter() - t1) / 5
    rb = bs.results()
    t1 = time.perf_counter()
    nchk = 64
    for i in range(nchk):  # oracle, single thread: CPU rate and bit-exact check of a sample
        o = O.OracleBounded(*boxes[i]).solve(100000)
        if not (np.array_equal(rb["x"][i], o.x) and rb["iters"][i] == o.iters):
            sys.exit(f"bench invalid: batched bounded LP {i} differs from the oracle")
    tcb = (time.perf_counter() - t1) / nchk
    batched = {"workload": f"{nb} x gen_box({mb},{2 * mb},seed), full solves, one workgroup per LP (lpx_revised_solve_small_batch)",
               "value": nb / tb, "unit": "LPs/s", "ms_per_launch": tb * 1e3, "all_optimal": bool((rb["status"] == 1).all()),
               "iterations_per_s": float(rb["iters"].sum()) / tb,
               "cpu_baseline": {"value": 1.0 / tcb, "unit": "LPs/s", "cores": 1, "kind": "port",
                                "sample": f"the first {nchk} LPs through the oracle's bounded loop (C, single thread, Python "
                                          f"constructor included)",
                                "parity": f"x and iteration counts of those {nchk} LPs bit-identical"}}
    out = {"metric": "config-5 solve sets/sec (bounded, dual, primal-dual; m <= 170)", "value": args.steps / dt, "unit": "sets/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "gen_box(m,2m,0) bounded + gen_dual(m,2m,0) dual [+ primal-dual for m <= 64], "
                                  "m in 16,64,128,170, full solves incl. construction (replicas only: no sharding); a lone LP "
                                  "that fits one CU's LDS runs as ONE launch (lpx_revised_solve_small), larger ones on the "
                                  "multi-kernel chain",
                      "per_problem": detail},
           "batched_bounded": batched,
           "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                        "note": "launch/latency-bound at these sizes (m <= 170: B^-1 is 231 KB); no bandwidth claim"},
           "cpu_baseline": {"value": 1.0 / cpu_t, "unit": "sets/s", "cores": 1, "kind": "port",
                            "sample": "the same set once through the oracle (C, single thread, Python constructors included)",
                            "parity": "iters equal and cost within 1e-9 for every problem of the set"}}
    emit(out)


# AMD's published MI355X figure, 78.6 TFLOP/s vector fp64 = 39.3 T fma/s (MI355X_MICROARCH.md has no fp64 row; on random
# operands the chip sustains 26-28 T fma/s: tools/fma_rate.hip)
FP64_FMA_PEAK = 78.6e12 / 2


def bench_batch(args, rank, world, group, K, W):
    """Config 3: 4096 x gen_primal(64, 128, seed) full solves.  N > 1 ("strong", BASELINE config 3): the FIXED batch of
    4096 LPs is split over the ranks by LP index (ShardedBatch), no collective during the solve, one all-gather of the
    results per step inside the timed region.  --batch-scaling weak: 4096 LPs per GPU, no gather.  Returns the record."""
    import torch
    import torch.distributed as dist

    from linprog_amd import problems
    from linprog_amd.batched import BatchedPrimalSolver, BatchedTableauSolver, ResultGather, shard

    # carried state per LP: the condensed tableau in registers (default) or B^-1 in LDS (LPX_BATCH_FORM=revised)
    tableau_form = os.environ.get("LPX_BATCH_FORM", "tableau") != "revised"
    Batched = BatchedTableauSolver if tableau_form else BatchedPrimalSolver
    total = 4096
    weak = args.batch_scaling == "weak" and world > 1
    if weak:
        lo, hi = rank * total, (rank + 1) * total
    else:
        lo, hi = shard(total, rank, world)
    count = hi - lo
    m, n = 64, 128
    probs = [problems.gen_primal(m, n, s) for s in range(lo, hi)]
    args4 = [np.stack([p[k] for p in probs]) for k in range(4)]
    # Two solver objects on two streams, alternating: consecutive batches overlap, so the tail of one launch (the last
    # long LPs running alone on a mostly idle chip) is covered by the head of the next.  Every step is still one full
    # pass over one batch with its own inputs and outputs.  The single-stream figure (no overlap) is measured as well.
    solvers = [Batched(*args4) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    gathers = None
    if group is not None and not weak:
        gathers = [ResultGather(group, total, m, n, "cuda") for _ in range(2)]
    solver = solvers[0]
    basis0 = solver.basis.clone()
    torch.cuda.synchronize()

    def barrier():
        if group is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(cnt, two_streams):
        for i in range(cnt):
            q = (i & 1) if two_streams else 0
            with torch.cuda.stream(streams[q]):
                s = solvers[q]
                s.basis.copy_(basis0)
                s.enqueue(100000)
                if gathers is not None:  # the one collective: every rank ends the step holding all 4096 results
                    gathers[q].pack(s.x, s.cost, s.basis, s.status, s.iters, s.pivots)
                    gathers[q].all_gather()

    def timed(two_streams):
        run(W, two_streams)
        barrier()
        t0 = time.perf_counter()
        run(K, two_streams)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    dt1 = timed(False)
    dt = timed(True)
    piv_t = torch.tensor([float(solver.pivots.sum().item())], dtype=torch.float64, device="cuda")
    if world > 1:
        if dist.get_backend() != "nccl":
            piv_t = piv_t.cpu()
        dist.all_reduce(piv_t)
    piv = int(piv_t.item())  # pivots of one step over all ranks
    ok = bool((solver.status == 1).all().item())
    lps = total * world if weak else total
    # Bounds.  The solver state lives in registers, so HBM sees one load of (A, b, c, basis) and one store of
    # (x, basis, cost, status, iters, pivots) per LP and launch: `achieved` is that true traffic over the launch time -- a
    # tiny fraction of the HBM peak, which is the point.  The kernel is bound by fp64 VALU issue: `valu` prices the fma
    # count of the rank-1 updates, m x (k + 1) per pivot, against the vector fp64 peak.  SURVEY.md 8(d)'s tableau-formula
    # rate 16 (m+1)(n_carried+1) B per pivot is served from registers and is given as information only.
    ncarry = n - m if tableau_form else n
    hbm_per_launch = count * (4.0 * (m * n + m + n) + 8 * m + 8.0 * n + 8 * m + 8 + 4 + 8 + 8)
    per_launch_s = dt / K
    achieved = hbm_per_launch / per_launch_s / 1e9
    fma_rate = piv * float(m) * (ncarry + 1) * K / dt
    out = {"metric": "batched LPs solved/sec (m=64 n=128)", "value": lps * K / dt, "unit": "LPs/s",
           "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
           "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f64",
           "data": "synthetic",
           "config": {"workload": (f"{total} x gen_primal(64,128,seed) per GPU" if weak else
                                   f"{total} x gen_primal(64,128,seed) in total, split over the ranks by LP index, one "
                                   f"all-gather of the results per step") +
                                  ", full solves (maxiters 100000), one wave per LP; `value`: consecutive batches alternate "
                                  "between two streams; `single_stream`: back to back on one stream, no overlap",
                      "form": "condensed tableau in registers (lpx_batched_tableau_solve)" if tableau_form
                      else "revised form, B^-1 in LDS (lpx_batched_primal_solve)",
                      "lps_per_gpu": count, "pivots_per_step": piv, "all_optimal": ok,
                      "pivots_per_s": piv * K / dt},
           "single_stream": {"value": lps * K / dt1, "unit": "LPs/s", "ms_per_step": dt1 / K * 1e3,
                             "pivots_per_s": piv * K / dt1},
           "roofline": {"bound": "hbm", "kernel": "k_batched_tab" if tableau_form else "k_batched_wave",
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": hbm_per_launch,
                        "traffic_source": "computed: bytes of the kernel's inputs and outputs per launch (this rank)",
                        "valu": {"fma_per_s": fma_rate, "peak_fma_per_s": FP64_FMA_PEAK * world,
                                 "frac": fma_rate / (FP64_FMA_PEAK * world),
                                 "note": "fma of the rank-1 updates, m x (carried columns + 1) per pivot, against the vector "
                                         "fp64 peak: the bound that matters for a register-resident solver"},
                        "tableau_formula_GBs": 16.0 * (m + 1) * (ncarry + 1) * piv * K / dt / 1e9,
                        "note": "register-resident solver state: HBM traffic is inputs + outputs only, so `frac` is small "
                                "by design; see `valu`"}}
    if rank == 0 and (args.cpu_pivots is None or args.cpu_pivots > 0):
        # rank 0's LPs through the oracle, one LP per core: CPU baseline and bit-exact parity gate in one
        from oracle import oracle as O

        cores = min(16, os.cpu_count() or 1)
        c, A, b, basis = [np.stack([p[k] for p in probs]) for k in range(4)]
        O.primal_solve_batch(c[:64], A[:64], b[:64], basis[:64], 100000, threads=cores)  # thread pool, page faults
        reps, dc = 0, 0.0
        while dc < 4.0 and reps < 50:
            t1 = time.perf_counter()
            r = O.primal_solve_batch(c, A, b, basis, 100000, threads=cores)
            dc += time.perf_counter() - t1
            reps += 1
        res = solver.results()
        if tableau_form:
            # same pivots as the revised-form oracle on every LP (basis, iteration count), x and cost to 1e-9; and bit for
            # bit the oracle's TABLEAU loop, LP by LP (single thread, Python loop: a sample)
            same = (np.array_equal(res["basis"], r["basis"]) and np.array_equal(res["iters"], r["iters"])
                    and np.allclose(res["x"], r["x"], rtol=1e-9, atol=1e-9) and np.allclose(res["cost"], r["cost"], rtol=1e-9, atol=1e-9))
            step = max(1, count // 256)
            for i in range(0, count, step):
                o = O.OracleTableau(*probs[i], log_cap=4).solve(100000)
                same = same and np.array_equal(res["x"][i], o.x) and res["cost"][i] == o.cost and res["iters"][i] == o.iters
            how = (f"basis and iters of all {count} LPs of rank 0 identical to the revised-form oracle's, x and cost within 1e-9; "
                   f"x, cost and iters of {len(range(0, count, step))} sampled LPs bit-identical to the oracle's tableau loop")
        else:
            same = (np.array_equal(res["x"], r["x"]) and np.array_equal(res["cost"], r["cost"])
                    and np.array_equal(res["basis"], r["basis"]) and np.array_equal(res["iters"], r["iters"]))
            how = f"x, cost, basis and iters of all {count} LPs of rank 0 bit-identical to the GPU's"
        if same and gathers is not None:
            # the gathered batch: rank 0's rows are its own results, every row is an optimal solve
            g = gathers[0].unpack()
            same = bool(np.array_equal(g["x"][lo:hi], res["x"]) and (g["status"] == 1).all() and g["x"].shape[0] == total)
            how += f"; gathered batch: {total} rows, all optimal, rank 0's rows equal its own results"
        if not same:
            sys.exit("bench invalid: batched GPU results differ from the oracle's on the same batch")
        if world == 1:
            out["cpu_baseline"] = {"value": count * reps / dc, "unit": "LPs/s", "cores": cores, "kind": "port",
                                   "sample": f"the same {count} LPs x {reps} passes, oracle revised loop (C), one LP per "
                                             f"OpenMP thread",
                                   "parity": how}
        else:
            out["parity"] = "rank 0: " + how
    del solvers, gathers
    return out


if __name__ == "__main__":
    main()
