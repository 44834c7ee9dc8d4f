#!/usr/bin/env python3
"""bench.py -- headline benchmark of the seed-extension hot path on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md section 8d "config 2 headline"): searchp, BLOSUM62 11/1,
100 000 synthetic 150-aa queries x 32 candidate windows of Lq + 2b = 176 residues = 3.2 M extensions =
84.48 Gcells of full-rectangle (parity mode) DP per GPU.  A *step* is one pass of the hot path over that batch:
pass 1 (score every window) [+ e-value filter + pass 2 traceback of the survivors once --with-trace is on].
Inputs are resident in HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Multi-GPU: the path shards by query with no data-path collective -> weak scaling, every
rank owns its own 100 k queries; the only collective is the final gather of per-rank result counts/top hits,
which is outside the hot loop but inside the timed region's last step.

Rank 0 prints ONE JSON line.  `value` = GCUPS = (sum over ranks of Lq*Ls cells of pass 1) * K / max-over-ranks
seconds / 1e9.  The oracle is used ONLY for the cpu_baseline leg (rank 0, N = 1), never for `value`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# integer-VALU roofline: 10 algorithmic integer ops per cell (4 add + 6 max, score-only
# Gotoh) against 256 CU x 128 lanes x 2.4 GHz = 78.6 T int32 lane-ops/s.  See DESIGN.md "Roofline" for the
# measured per-instruction issue rates (tools/ubench.hip) this peak is compared with.
ALGO_OPS_PER_CELL = 10
PEAK_INT32_TOPS = 256 * 128 * 2.4e9 / 1e12
# kernels that compute two cells per lane-op (packed 16-bit): SURVEY.md section 8d quotes 157 Tops/s for them
PEAK_PACKED16_TOPS = 2 * PEAK_INT32_TOPS
ALGO_BYTES_PER_EXT_EXTRA = 16 + 4  # extension record read + score written


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--queries", type=int, default=100_000, help="queries per GPU")
    ap.add_argument("--lq", type=int, default=150)
    ap.add_argument("--windows", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-queries", type=int, default=100_000)
    ap.add_argument("--pass1-only", action="store_true", help="time the score kernel alone (no filter, no traceback)")
    ap.add_argument("--db-length", type=int, default=205_000_000, help="dbTotalLength for the e-value (Swiss-Prot sized)")
    ap.add_argument("--max-evalue", type=float, default=1e-2)
    ap.add_argument("--max-matches", type=int, default=25, help="HSPs kept per query for the final gather (maxMatches)")
    return ap.parse_args()


def usable_cpus() -> tuple[int, str]:
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes expose all
    256 hardware threads but `cpu.max` grants 16 CPUs' worth of time -- oversubscribing them only gets throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = f"{n} hardware threads visible"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = max(1, int(int(quota) / int(period) + 0.5))
            if q < n:
                note += f", cgroup cpu.max = {quota}/{period} -> {q} CPUs"
                n = q
    except Exception:
        pass
    return n, note


def cpu_baseline(args, cores: int, note: str = ""):
    """Times the oracle's inter-sequence int16 SIMD batch scorer (the shape of the reference's CPU path,
    oracle/lx_oracle_simd.cpp) on a bounded sample of the same workload, on this box's host cores."""
    from lambda_amd import capi, synth
    from tests import oracle_lib

    orc = oracle_lib.load()
    nq = min(args.cpu_sample_queries, args.queries)
    q, s, ext = synth.make_batch_np(nq, args.lq, args.windows, seed=0x1A3BDA02)
    sc = oracle_lib.scoring_from(capi.builtin_scoring(62, gap_open=-11, gap_extend=-1))
    cells = float((ext["q_len"].astype(np.float64) * ext["s_len"]).sum())
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        orc.score_batch(q, s, ext, sc, threads=cores, simd=True)
        best = min(best, time.perf_counter() - t0)
    return {
        "value": round(cells / best / 1e9, 3),
        "unit": "GCUPS",
        "cores": cores,
        "kind": "port",
        "sample": f"{nq} queries x {args.windows} windows ({len(ext)} extensions, {cells / 1e9:.2f} Gcells), "
                  f"oracle inter-sequence int16 SIMD restatement (NOT SeqAn), OpenMP with {cores} threads ({note}), "
                  f"best of 3, {best:.3f} s",
    }


def pmc_traffic(kernel_name: str):
    """HBM bytes per step of a kernel (kernel_name = substring of its rocprofv3 name) from the committed rocprofv3 PMC passes (profiles/*_pmc.json):
    (2 x FETCH_SIZE + WRITE_SIZE) KiB -- FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md.
    PMC counters cannot be read from inside a timed run, so this is the figure of the profiled run of the same
    command; None if no profile of this kernel instantiation is committed."""
    try:
        files = sorted((ROOT / "profiles").glob("*_pmc.json"))
        for f in reversed(files):
            for name, d in json.loads(f.read_text())["kernels"].items():
                c = d.get("counters_per_step_mean", d.get("counters_per_launch_mean", {}))
                if kernel_name in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                    return ((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
                            f"{f.name}: (2*FETCH_SIZE + WRITE_SIZE) KiB per step of {name}")
    except Exception:
        pass
    return None, "no committed PMC profile for this kernel instantiation"


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from lambda_amd import capi, shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus) and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("LX_BENCH_FORCE_DIST") == "1"  # the env switch lets one GPU exercise the RCCL path
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- workload, generated directly in HBM; every rank owns different queries (shard by query) ----
    h = capi.Handle(local_rank)
    h.set_scoring(capi.builtin_scoring(62, gap_open=-11, gap_extend=-1), 0)
    h.set_option(capi.LX_OPT_MAX_QLEN, args.lq)
    h.set_option(capi.LX_OPT_QUERY_RUN, args.windows if args.windows % 8 == 0 else 0)
    d_q, d_s, d_ext, ext = synth.make_batch_torch(args.queries, args.lq, args.windows, 0x1A3BDA02 + rank, dev)
    pad = torch.zeros(256, dtype=torch.uint8, device=dev)
    d_q = torch.cat([d_q, pad])
    d_s = torch.cat([d_s, pad])
    n = len(ext)
    cells_rank = float((ext["q_len"].astype(np.float64) * ext["s_len"]).sum())
    d_score = torch.zeros(n, dtype=torch.int32, device=dev)
    # pass-2 outputs (worst case sizes: every extension may survive)
    h.set_option(capi.LX_OPT_MAX_SLEN, int(ext["s_len"].max()))
    sizes = ext["q_len"].astype(np.uint64) + ext["s_len"].astype(np.uint64)
    off = np.zeros(n, dtype=np.uint64)
    off[1:] = np.cumsum(sizes)[:-1]
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_ops = torch.zeros(int(sizes.sum()) + 16, dtype=torch.uint8, device=dev)
    d_hsp = torch.zeros(n * 48, dtype=torch.uint8, device=dev)
    d_count = torch.zeros(2, dtype=torch.int64, device=dev)
    # e-value filter of iterateMatchesFullSimd (maxEValue 1e-2, src/search_options.hpp:96) as an integer score cut-off
    ka = capi.karlin_params(62, gap_open=-11, gap_extend=-1)
    lib = capi.load()
    import ctypes as C
    adj = lib.lx_length_adjustment(args.db_length, args.lq, C.byref(ka))
    min_score = 1
    while lib.lx_evalue(min_score, args.lq - adj, args.db_length - adj, C.byref(ka)) > args.max_evalue:
        min_score += 1
    # a non-default torch stream: its handle is non-NULL, so the kernels really run on the stream the timing events
    # are recorded on (NULL would select the lx handle's private stream)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()

    def step():
        if args.pass1_only:
            h.score_batch_dev(d_q, d_s, d_ext, n, d_score, stream=stream.cuda_stream)
        else:
            h.extend_batch_dev(d_q, d_s, d_ext, n, min_score, d_score, d_hsp, d_ops, d_off, d_count,
                               stream=stream.cuda_stream)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def gather():
        # the path's only exchange: every rank keeps, per query, its best `maxMatches` = 25
        # HSPs (src/search_options.hpp:99) among the survivors of the e-value filter, then one gather of fixed-size
        # records [global extension id, lx_hsp (48 B)] = 56 B each -- RCCL over xGMI.  Query ranges are disjoint, so
        # rank order is the final order.
        if args.pass1_only or args.windows <= 0:
            hit = torch.nonzero(d_score >= min_score).flatten()
            return shard.gather_hits(torch.stack([hit + rank * n, d_score[hit].to(torch.int64)], dim=1), dst=None)
        sc2 = d_score.view(args.queries, args.windows)
        k = min(args.max_matches, args.windows)
        top, idx = torch.topk(sc2, k, dim=1)
        keep = top >= min_score
        ext_id = (idx + torch.arange(args.queries, device=dev).unsqueeze(1) * args.windows)[keep]
        hsp64 = d_hsp.view(torch.int64).view(n, 6)[ext_id]
        rec = torch.cat([(ext_id + rank * n).unsqueeze(1), hsp64], dim=1)
        # all_gather (dst=None): ~2 ms for 8 x 90 MB over xGMI, once per run; a gather to the writing rank costs the same
        return shard.gather_hits(rec, dst=None)

    for _ in range(args.warmup):
        step()
    if use_dist and args.warmup > 0:
        torch.cuda.synchronize()
        gather()  # untimed: the first collective of each kind sets up RCCL's channels
    fence()
    h.synchronize()  # surfaces any device-side error flag before timing

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)
        step()
        ev[k][1].record(stream)
    n_hits_total = None
    if use_dist:
        stream.synchronize()
        all_hits = gather()
        n_hits_total = int(all_hits.shape[0])
    fence()
    dt = time.perf_counter() - t0
    h.synchronize()
    kernel_name = h.last_kernel_name()
    survivors = int(d_count.cpu()[1]) if not args.pass1_only else 0

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    # device time per phase of the LAST timed step (HIP events recorded by the library on the launch stream, around
    # each kernel launch): 0 = pass-1 score kernel, 1 = selection, 2 = pass-2 forward kernel, 3 = pass-2 backtrace
    phase_ms = {ph: h.last_phase_ms(ph) for ph in (0, 1, 2, 3)}
    trace_kernel_name = h.last_trace_kernel_name()

    if rank == 0:
        total_cells = cells_rank * world
        gcups = total_cells * args.steps / dt / 1e9
        lq, ls = args.lq, synth.window_len(args.lq)

        def roofline(kernel, ms, launches, cells, pmc_key, algo_bytes):
            """The DP kernels are bound by integer VALU issue (not HBM, not MFMA), so `bound` is
            "valu" and achieved/peak count lane-ops; the HBM side of the same kernel -- algorithmic bytes per step over
            its duration against 8 TB/s -- is reported next to it as evidence that memory is not the limit."""
            gc = cells / (ms * 1e-3) / 1e9
            tops = gc * ALGO_OPS_PER_CELL / 1e3
            traffic, note = pmc_traffic(pmc_key)
            packed = "pair_kernel" in kernel
            peak = PEAK_PACKED16_TOPS if packed else PEAK_INT32_TOPS
            hbm = algo_bytes / (ms * 1e-3) / 1e9
            return {
                "bound": "valu", "kernel": kernel, "achieved": round(tops, 3), "peak": round(peak, 2),
                "unit": "Tops/s (%s lane-ops; 10 algorithmic ops per cell)" % ("packed 16-bit" if packed else "int32"),
                "frac": round(tops / peak, 4),
                "kernel_ms_per_step": round(ms, 4), "launches_per_step": launches, "kernel_gcups": round(gc, 1),
                "cells_per_step": cells,
                "hbm": {"bound": "hbm", "achieved": round(hbm, 2), "peak": 8000, "unit": "GB/s", "frac": round(hbm / 8000, 4),
                        "algorithmic_bytes_per_step": algo_bytes},
                "traffic": traffic, "traffic_note": note,
            }

        # algorithmic bytes: pass 1 reads every window once, every query once per run, one 24-byte record per extension and
        # writes one int32 score; pass 2 forward reads the same per survivor (+ its score) and writes 4 direction bits per cell
        algo_bytes = float(ext["q_len"].sum()) / args.windows + float(ext["s_len"].sum()) + n * ALGO_BYTES_PER_EXT_EXTRA
        if "single sweep" in kernel_name:
            # the sweep also writes the checkpoints of every extension: one boundary pair per strip and row, one pair per
            # column every 16 rows -- 2-byte codes from the packed-half kernel, int16 pairs from the int32 kernel -- and a
            # 16-byte end record
            pair_bytes = 2 if "pair_kernel" in kernel_name else 4
            algo_bytes += n * (ls * (-(-lq // 19)) * pair_bytes + (ls / 16.0) * lq * pair_bytes + 16)
        r_score = roofline(kernel_name, phase_ms[0][0], phase_ms[0][1], cells_rank,
                           "score_pair_kernel" if "pair_kernel" in kernel_name else "score_kernel", algo_bytes)
        rooflines = [r_score]
        if not args.pass1_only and phase_ms[2][0] > 0:
            cells2 = float(survivors) * lq * ls
            if "ckpt" in trace_kernel_name:
                # checkpoint mode: one 4-byte boundary pair per strip and row + one 4-byte pair per column every 16 rows
                strips = -(-lq // 19)
                stored = ls * strips * 4 + (ls / 16.0) * lq * 4
                pmc_key = "ckpt_forward_kernel"
            else:
                stored = lq * ls / 2.0  # 4 direction bits per cell
                pmc_key = "trace_forward_kernel"
            algo2 = survivors * (stored + ls + lq / 4.0 + ALGO_BYTES_PER_EXT_EXTRA + 4)
            rooflines.append(roofline(trace_kernel_name, phase_ms[2][0], phase_ms[2][1], cells2, pmc_key, algo2))
        rooflines.sort(key=lambda r: -r["kernel_ms_per_step"])
        out = {
            "metric": "GCUPS (gapped extension, full-rectangle parity mode; pass-1 cells per second of whole step) searchp BLOSUM62",
            "value": round(gcups, 2),
            "unit": "GCUPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16x2 (exact small integers) + int32" if "pair_kernel" in kernel_name else "int32",
            "data": "synthetic",
            "config": {
                "workload": f"searchp BLOSUM62 gap 11/1, {args.queries} x {args.lq} aa queries x {args.windows} "
                            f"windows of {synth.window_len(args.lq)} aa per GPU (BASELINE.json configs[1]), "
                            f"cells = sum Lq*Ls (full rectangle, band off as in the reference)",
                "extensions_per_gpu": n,
                "gcells_per_gpu": round(cells_rank / 1e9, 3),
                "parallelism": f"query-sharded x{world}, no data-path collective; one gather of the per-query top-"
                               f"{args.max_matches} HSP records (56 B) after the last step"
                               + (f" ({n_hits_total} records)" if n_hits_total is not None else ""),
                "step": ("pass 1 score kernel over the whole batch" if args.pass1_only else
                         ("single sweep: " if "single sweep" in kernel_name else "") +
                         f"pass 1 (score all) -> e-value filter (E<={args.max_evalue:g} at db {args.db_length}, i.e. score>={min_score}) "
                         f"-> pass 2 (traceback of the {survivors} survivors), all on the GPU; GCUPS counts pass-1 cells only"),
                "survivors_per_gpu": survivors,
                "pass2_gcells_per_gpu": round(survivors * args.lq * synth.window_len(args.lq) / 1e9, 3),
            },
            "alignments_per_s": round(n * world * args.steps / dt, 1),
            "traced_per_s": round(survivors * world * args.steps / dt, 1),
            "total_gcups_both_passes": round((cells_rank + float(survivors) * lq * ls) * world * args.steps / dt / 1e9, 2),
            "phase_ms_last_step": {"score": round(phase_ms[0][0], 3), "select": round(phase_ms[1][0], 3),
                                   "trace_forward": round(phase_ms[2][0], 3), "backtrace": round(phase_ms[3][0], 3),
                                   "events_step_ms": round(kern_ms, 3)},
            "roofline": rooflines[0],           # the kernel that takes most of the step
            "roofline_other": rooflines[1:],
        }
        if world == 1 and not args.no_cpu_baseline:
            cores, note = usable_cpus()
            out["cpu_baseline"] = cpu_baseline(args, cores, note)
        print(json.dumps(out), flush=True)
    h.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
