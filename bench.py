#!/usr/bin/env python3
"""bench.py -- benchmark of the seed-extension hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config {1,2,3,4}]

Default workload = BASELINE.json configs[1] (SURVEY.md section 8d "config 2 headline"): searchp, BLOSUM62 11/1, 100 000
synthetic 150-aa queries x 32 candidate windows of Lq + 2b = 176 residues = 3.2 M extensions = 84.48 Gcells of
full-rectangle (parity mode) DP per GPU.  `--config k` selects BASELINE.json configs[k] (lambda_amd/workloads.py):
2 = searchn (1 M x 150 bp, 8 windows, +2/-3, 5/2, 1 % N), 3 = searchp scale-out (1 M x 200 aa x 32 windows split by query
over the ranks), 4 = bisulfite (500 k x 150 bp, both matrices / two scoring slots).

A *step* is one pass of the hot path over the rank's share of the job: pass 1 (score every window) -> e-value filter ->
pass 2 (traceback of the survivors), all on the GPU, inputs resident in HBM before the timed region starts.  A share
that does not fit one device call (checkpoint slots of configs[3] at N = 1) is processed in several calls per step.

Multi-GPU: one process per GPU.  Launched by the driver as `python -m torch.distributed.run --nproc-per-node N ...
bench.py --gpus N ...`; started plainly with --gpus N > 1 it re-executes itself under torch.distributed.run (and fails
loudly when the node has fewer than N devices).  The path shards by query with no data-path collective (SURVEY.md section
8e); the only collective is the final gather of the per-query top hits, inside the timed region after the last step.
Weak scaling (every rank owns a full per-GPU share; default of configs 1, 2) or strong scaling (the job's queries split over
the ranks: default of configs 3, 4; `--total-queries T` forces it, `--queries Q` forces weak with Q per rank).

Rank 0 prints ONE JSON line.  `value` = GCUPS = (pass-1 cells of all ranks, sum Lq*Ls) * K / max-over-ranks seconds / 1e9.
The oracle is used ONLY for the cpu_baseline leg (rank 0, N = 1), never for `value`.
`--dry-run` walks through launch, rendezvous (gloo) and sharding without touching a GPU (CPU tests of the launch path).
"""
from __future__ import annotations

import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# integer-VALU roofline: 10 algorithmic integer ops per cell (4 add + 6 max, score-only
# Gotoh) against 256 CU x 128 lanes x 2.4 GHz = 78.6 T int32 lane-ops/s.  See DESIGN.md section 4.2 / 6 for the
# measured per-instruction issue rates (tools/ubench.hip) this peak is compared with.
ALGO_OPS_PER_CELL = 10
PEAK_INT32_TOPS = 256 * 128 * 2.4e9 / 1e12
# kernels that compute two cells per lane-op (packed 16-bit): SURVEY.md section 8d quotes 157 Tops/s for them
PEAK_PACKED16_TOPS = 2 * PEAK_INT32_TOPS
ALGO_BYTES_PER_EXT_EXTRA = 24 + 4  # extension record read + score written


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4], help="BASELINE.json configs[k]")
    ap.add_argument("--queries", type=int, default=None, help="weak scaling: queries per GPU (default: the config's per-GPU share)")
    ap.add_argument("--total-queries", type=int, default=None, help="strong scaling: queries of the whole job, split over the ranks")
    ap.add_argument("--batch-queries", type=int, default=None, help="queries per device call (default: the config's)")
    ap.add_argument("--lq", type=int, default=None, help="override the config's query length")
    ap.add_argument("--windows", type=int, default=None, help="override the config's windows per query")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-queries", type=int, default=None)
    ap.add_argument("--pass1-only", action="store_true", help="time the score kernel alone (no filter, no traceback)")
    ap.add_argument("--db-length", type=int, default=None, help="dbTotalLength for the e-value (default: the config's)")
    ap.add_argument("--max-evalue", type=float, default=None)
    ap.add_argument("--max-matches", type=int, default=25, help="HSPs kept per query for the final gather (maxMatches)")
    ap.add_argument("--trace-bytes", type=int, default=160 << 30, help="LX_OPT_TRACE_BYTES: HBM the checkpoint slots may take")
    ap.add_argument("--extend-chunk", type=int, default=None, help="--host-path / --ragged: LX_OPT_EXTEND_CHUNK (extensions per chunk of the pipeline; default: the library's)")
    ap.add_argument("--band", type=int, default=0, help="band mode (LX_OPT_BAND, not the reference's configuration): half width in "
                    "diagonals around the window's seed diagonal; the metric then counts cells = sum Lq * min(Ls, 2 band + 1)")
    ap.add_argument("--host-path", action="store_true", help="time lx_extend_batch on HOST buffers (what a lambda3 binding calls, INTEGRATION.md "
                    "level 1/2): PCIe and the host's share included, subjects resident (lx_set_subjects); a secondary line, never `value` of the headline")
    ap.add_argument("--entry", choices=("bytes", "rle", "list", "host", "dev"), default="bytes", help="--host-path / --ragged: lx_extend_batch (ops as column bytes, n "
                    "records), lx_extend_batch_rle (run-length codes, n records) or lx_extend_batch_list (the survivors as a list, as the "
                    "reference's filter loop leaves them)")
    ap.add_argument("--survivor-rate", type=float, default=None, help="share of homologous windows in the synthetic batch (default 0.5: "
                    "half of the windows pass the e-value cut-off, far more than a real seed set); 0.02 / 0.1 show the adaptive pass-2 mode "
                    "(LX_OPT_ADAPT_PERMILLE)")
    ap.add_argument("--adapt-permille", type=int, default=None, help="LX_OPT_ADAPT_PERMILLE (0 = always the single sweep)")
    ap.add_argument("--query-run", type=int, default=0, help="development aid: LX_OPT_QUERY_RUN promise for the device-resident step (default: the "
                    "workload's windows per query)")
    ap.add_argument("--mq-sweep", type=int, default=None, help="development aid: LX_OPT_MQ_SWEEP (2 = the multi-query sweep for every run that is a "
                    "multiple of 4)")
    ap.add_argument("--ragged", action="store_true", help="--host-path on a ragged seed list as lambda really produces them (query lengths "
                    "50-400, windows per query geometric with mean 12, 10 %% merged windows of up to 3 Lq): GCUPS and the padded share")
    ap.add_argument("--ragged-queries", type=int, default=50_000)
    ap.add_argument("--lq-range", type=int, nargs=2, default=None, metavar=("LO", "HI"), help="--ragged: query lengths (default 50 400)")
    ap.add_argument("--ragged-mean-windows", type=float, default=None, help="--ragged: mean windows per query (default 12)")
    ap.add_argument("--strong", action="store_true", help="--ragged --lq-range 500 800 --strong: the long strong-hit list of VERDICT r3 / "
                    "tools/dev/long_queries.py -- 8 000 queries, 8 windows each on average, half of them homologous at the workload's substitution "
                    "rate, i.e. scoring 2 000-3 500: beyond the compact checkpoint codes")
    ap.add_argument("--iterate", action="store_true", help="time lx_iterate_matches_dev -- the whole of iterateMatchesFullSimd on a DEVICE match "
                    "list (widen, sort, merge, unique, both passes, records) -- on a synthetic seed list of configs[2]'s size, or with --config 1 "
                    "of configs[1]'s (searchp BLOSUM62: 100 000 queries x ~32 windows); --entry host: "
                    "lx_iterate_matches on the same list in host memory")
    ap.add_argument("--iterate-reads", type=int, default=1_000_000)
    ap.add_argument("--iterate-mbp", type=float, default=100.0)
    ap.add_argument("--iterate-queries", type=int, default=100_000, help="--iterate --config 1: queries of the protein seed list (configs[1]: 100 000 x 150 aa, "
                    "~32 windows each after merging)")
    ap.add_argument("--cold", action="store_true", help="--iterate: every timed call is the FIRST call of a fresh handle (lx_create, the sequence "
                    "sets, lx_reserve with hints, then the one call a search makes): what `lambda3 searchn` pays, not the tenth call")
    ap.add_argument("--dry-run", action="store_true", help="launch + rendezvous (gloo) + sharding only; no GPU, value = null")
    args = ap.parse_args(argv)
    # (--iterate without --config keeps timing the read set of configs[2]; `--iterate --config 1` is the metric's own program: the protein list)
    args.config_given = any(a == "--config" or a.startswith("--config=") for a in (sys.argv[1:] if argv is None else argv))
    return args


def workload_of(args):
    import dataclasses

    from lambda_amd import workloads

    w = workloads.WORKLOADS[args.config]
    over = {}
    if args.lq is not None:
        over["lq"] = args.lq
    if args.windows is not None:
        over["windows"] = args.windows
    if args.db_length is not None:
        over["db_length"] = args.db_length
    if args.max_evalue is not None:
        over["max_evalue"] = args.max_evalue
    return dataclasses.replace(w, **over) if over else w


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def maybe_self_spawn(args) -> None:
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    if not args.dry_run:
        import torch

        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this node exposes {have} GPU(s); refusing to run fewer ranks than asked for")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


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


def cpu_baseline(w, sample_queries: int, cores: int, note: str = ""):
    """Times the oracle's inter-sequence int16 SIMD batch scorer (the shape of the reference's CPU path,
    oracle/lx_oracle_simd.cpp) on a bounded sample of the same workload, on this box's host cores."""
    from lambda_amd import capi, synth, workloads
    from tests import oracle_lib

    orc = oracle_lib.load()
    d = w.directions[0]
    q, s, ext = synth.make_batch_np(sample_queries, w.lq, w.windows, seed=w.seed, alphabet=workloads.alphabet_array(w),
                                    sub_rate=w.sub_rate, indel_rate=w.indel_rate, n_rate=w.n_rate, n_rank=w.n_rank,
                                    convert=d.convert, convert_rate=w.convert_rate)
    m, ma, mi, go, ge = d.scoring
    sc = oracle_lib.scoring_from(capi.builtin_scoring(m, match=ma, mismatch=mi, gap_open=go, gap_extend=ge))
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
        "sample": f"{sample_queries} queries x {w.windows} windows ({len(ext)} extensions, {cells / 1e9:.2f} Gcells) of this workload"
                  f"{' (forward direction)' if len(w.directions) > 1 else ''}, "
                  f"oracle inter-sequence int16 SIMD restatement (NOT SeqAn), OpenMP with {cores} threads ({note}), "
                  f"best of 3, {best:.3f} s",
    }


def kernel_instantiation(name: str) -> str:
    """`lx::score_pair_kernel<8,19,true>` from either spelling of a kernel's name: the library's ("lx::score_pair_kernel<8,19,true>
    (single sweep)") or rocprofv3's ("void lx::score_pair_kernel<8, 19, true>(lx::ScoreParams)").  The template arguments are part
    of the identity: `<8,19,true>` writes 23 GB of checkpoints per launch, `<8,19>` writes the scores."""
    s = name.strip()
    if s.startswith("void "):
        s = s[5:]
    m = re.match(r"([A-Za-z_][\w:]*)\s*(<[^()]*>)?", s)
    if not m:
        return s.replace(" ", "")
    base = m.group(1)
    if base.startswith("lx::"):
        base = base[4:]  # the namespace is not part of what distinguishes two profiles
    return (base + (m.group(2) or "")).replace(" ", "")


def pmc_traffic(kernel_name: str, command_has: str | None = None, command_also: str | None = None, command_lacks: str | None = None):
    """HBM bytes per launch of ONE kernel instantiation (kernel_name in the library's or rocprofv3's spelling, compared by
    `kernel_instantiation`: full template argument list, not a substring) from the committed rocprofv3 PMC passes
    (profiles/*_pmc.json, newest name first): (2 x FETCH_SIZE + WRITE_SIZE) KiB -- FETCH_SIZE doubled per the gfx950
    correction in MI355X_MICROARCH.md.  PMC counters cannot be read from inside a timed run, so this is the figure of the
    profiled run of the same command.  Returns (bytes or None, note); files of another shape (no "kernels" table) are skipped
    and a file that cannot be read is NAMED in the note instead of being swallowed.  `command_has`: only profiles of a command line
    that contains it (the secondary lines: "--ragged", "--host-path"); None: only profiles of the default command."""
    want = kernel_instantiation(kernel_name)
    skipped = []
    for f in sorted((ROOT / "profiles").glob("*_pmc.json"), reverse=True):
        try:
            doc = json.loads(f.read_text())
            kernels = doc.get("kernels")
        except Exception as e:  # unreadable / not JSON: say so, go on to the older profiles
            skipped.append(f"{f.name}: {type(e).__name__}: {e}")
            continue
        if not isinstance(kernels, dict):
            continue  # a profile of something else (e.g. the front end's seeding kernel): no per-kernel table
        cmd = str(doc.get("command", ""))
        if (command_has is None and any(k in cmd for k in ("--ragged", "--host-path", "--iterate"))) or (command_has is not None and command_has not in cmd):
            continue  # the profile of another workload
        if (command_also is not None and command_also not in cmd) or (command_lacks is not None and command_lacks in cmd):
            continue  # ... of another list of the same entry point (--ragged --config 2 against the protein list)
        for name, d in kernels.items():
            if not isinstance(d, dict) or kernel_instantiation(name) != want:
                continue
            c = d.get("counters_per_launch_mean") or d.get("counters_per_step_mean") or {}
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                return ((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
                        f"{f.name}: (2*FETCH_SIZE + WRITE_SIZE) KiB per launch of {name}; {profile_sources_note(doc)}"
                        + (f"; skipped {'; '.join(skipped)}" if skipped else ""))
    return None, (f"no committed PMC profile holds FETCH_SIZE and WRITE_SIZE of {want}"
                  + (f"; skipped {'; '.join(skipped)}" if skipped else ""))


def git_blob(path) -> str:
    """the git blob id of a file's content (`git hash-object`)"""
    import hashlib

    data = Path(path).read_bytes()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def stale_profile_sources(doc) -> dict:
    """{source: (blob id at profile time, blob id now)} for the kernel sources a PMC profile was made from that read differently today;
    None when the profile does not say what it was made from (rounds 1-4)."""
    srcs = doc.get("kernel_sources")
    if not isinstance(srcs, dict):
        return None
    out = {}
    for rel, then in srcs.items():
        try:
            now = git_blob(ROOT / rel)
        except OSError:
            now = "missing"
        if now != then:
            out[rel] = (then, now)
    return out


def profile_sources_note(doc) -> str:
    stale = stale_profile_sources(doc)
    if stale is None:
        return "the profile does not record the kernel sources it was made from (made before round 5)"
    srcs = doc["kernel_sources"]
    txt = "kernel sources at profile time " + ", ".join(f"{Path(k).name} {v[:10]}" for k, v in sorted(srcs.items()))
    if stale:
        txt += " -- STALE: " + ", ".join(f"{Path(k).name} is {now[:10]} now" for k, (_, now) in sorted(stale.items())) + " (re-run tools/profile_round.sh)"
    else:
        txt += " (= the tree's)"
    return txt


def issue_ceiling():
    """What gfx950 really issues for the packed sweep's instruction mix (3 packed adds, 3 packed max3, 1 permute per pair of
    cells -- every one a half-rate VOP3P / VOP3 instruction) at the sweep's occupancy, from the committed run of
    tools/ubench.hip (profiles/*_ubench_valu_issue_rates.txt, line "sweep_mix(x7) waves/SIMD=3"): lane-instructions per
    second, and the ceiling it puts on the 10-ops-per-cell accounting of SURVEY.md section 8d when a cell costs the bare
    recurrence's 3.75 lane-instructions.  The roofline's `frac` is against the guide's 157 Tops/s; this is the part of it
    the instruction set lets a kernel of this formulation reach."""
    rate, src = 35.1e12, "default (no committed ubench run found)"
    try:
        for f in sorted((ROOT / "profiles").glob("*_ubench_valu_issue_rates.txt"), reverse=True):
            for line in f.read_text().splitlines():
                if line.startswith("sweep_mix(x7)") and "waves/SIMD=3" in line:
                    rate = float(line.split("=")[-1].split()[0]) * 1e12
                    src = f"{f.name}: sweep_mix(x7) at 3 wavefronts per SIMD"
                    raise StopIteration
    except StopIteration:
        pass
    except Exception:
        pass
    bare = 3.75
    tops = rate / bare * ALGO_OPS_PER_CELL / 1e12
    return {"lane_instr_per_s": rate, "source": src, "bare_lane_instr_per_cell": bare, "ceiling_tops": round(tops, 2),
            "issue_ceiling_frac": round(tops / PEAK_PACKED16_TOPS, 4)}


def dry_run(args, w, world, rank):
    """No GPU: rendezvous over gloo, plan the sharding, gather the plans, print the line the real run would print (value
    null).  What the CPU tests and `python bench.py --gpus 2 --config 3 --dry-run` exercise."""
    import torch.distributed as dist

    from lambda_amd import workloads

    pl = workloads.plan(w, world, rank, args.total_queries, args.queries, args.batch_queries)
    # the library's host threads as THIS rank's process sizes them (no device needed): parts per host loop = min(affinity, cgroup quota)
    # / LOCAL_WORLD_SIZE -- the ranks of a node share its CPUs
    import ctypes as C

    from lambda_amd import capi

    wdt, granted, lws = C.c_uint32(), C.c_uint32(), C.c_uint32()
    capi.load().lx_host_threads_info(C.byref(wdt), C.byref(granted), C.byref(lws))
    mine = {"rank": rank, "q_lo": pl.q_lo, "q_hi": pl.q_hi, "calls": [[b.direction.slot, b.n_queries] for b in pl.batches],
            "host_threads": wdt.value, "granted_cpus": granted.value, "local_world_size": lws.value}
    plans = [mine]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        plans = [None] * world
        dist.all_gather_object(plans, mine)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({
            "metric": "GCUPS (gapped extension, full-rectangle parity mode)", "value": None, "unit": "GCUPS", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "scaling": pl.scaling, "dry_run": True,
            "config": {"workload": workloads.describe(w), "job_queries": pl.job_queries,
                       "job_gcells": round(workloads.cells_of(w, pl.job_queries) / 1e9, 3)},
            "ranks": plans}), flush=True)


def host_path(args, w, pl, world, rank, local_rank, dev, use_dist):
    """The step through the host-buffer entry point: every call takes the extension list from host memory, returns scores,
    records and ops to host memory; the database (here: the synthetic windows) stays on the device between calls, as
    lx_set_subjects is meant to be used.  Wall-clock around K calls."""
    import ctypes as C

    import torch
    import torch.distributed as dist

    from lambda_amd import capi, synth, workloads

    h = capi.Handle(local_rank)
    for d in w.directions:
        m, ma, mi, go, ge = d.scoring
        h.set_scoring(capi.builtin_scoring(m, match=ma, mismatch=mi, gap_open=go, gap_extend=ge), d.slot)
    h.set_option(capi.LX_OPT_BS_MATCH_RULE, 1 if len(w.directions) > 1 else 0)
    h.set_option(capi.LX_OPT_TRACE_BYTES, args.trace_bytes)
    if args.extend_chunk:
        h.set_option(capi.LX_OPT_EXTEND_CHUNK, args.extend_chunk)
    ka = capi.karlin_params(*w.karlin)
    lib = capi.load()
    adj = lib.lx_length_adjustment(w.db_length, w.lq, C.byref(ka))
    min_score = 1
    while lib.lx_evalue(min_score, w.lq - adj, w.db_length - adj, C.byref(ka)) > w.max_evalue:
        min_score += 1
    # host copies of the synthetic batches; all windows of the rank form its resident "database"
    parts, s_all, s_at = [], [], 0
    if args.ragged:
        if args.strong:  # tools/dev/long_queries.py's list
            nq, mw, seed = (8000 if args.ragged_queries == 50_000 else args.ragged_queries), args.ragged_mean_windows or 8.0, 5
        else:
            nq, mw, seed = args.ragged_queries, args.ragged_mean_windows or 12.0, 0x1A3BDA07 + rank
        q_np, s_np, ext = synth.make_ragged_lists_np(nq, seed=seed, alphabet=workloads.alphabet_array(w), lq_range=tuple(args.lq_range or (50, 400)),
                                                      mean_windows=mw, sub_rate=0.25 if args.strong else w.sub_rate,
                                                      indel_rate=0.02 if args.strong else w.indel_rate)
        args.ragged_queries = nq
        s_all.append(s_np)
        parts.append((0, q_np, ext))
    for b in ([] if args.ragged else pl.batches):
        d_q, d_s, _, ext = synth.make_batch_torch(b.n_queries, w.lq, w.windows, b.seed, dev, alphabet=workloads.alphabet_array(w),
                                                  sub_rate=w.sub_rate, indel_rate=w.indel_rate, n_rate=w.n_rate, n_rank=w.n_rank,
                                                  convert=b.direction.convert, convert_rate=w.convert_rate)
        ext = ext.copy()
        ext["s_off"] += s_at
        s_np = d_s.cpu().numpy()
        s_at += len(s_np)
        s_all.append(s_np)
        parts.append((b.direction.slot, d_q.cpu().numpy(), ext))
        del d_q, d_s
    h.set_subjects(np.concatenate(s_all))
    cells_rank = sum(float((e["q_len"].astype(np.float64) * e["s_len"]).sum()) for _, _, e in parts)
    keep, counts = [None] * len(parts), [0] * len(parts)

    def step():
        surv = 0
        for i, (slot, q, ext) in enumerate(parts):
            if args.entry == "list":
                r = h.extend_batch_list(q, None, ext, min_score, slot=slot, copy=False, out_score=None if keep[i] is None else keep[i][0])
                keep[i], counts[i] = (r[0], None, None), len(r[1])  # (the list's views die with the next call)
                continue
            r = h.extend_batch(q, None, ext, min_score, slot=slot, copy_ops=False, out=keep[i], rle=args.entry == "rle")
            keep[i] = r[:3]  # the caller keeps its result arrays between calls, like lambda's per-thread holders
            surv += int((r[1]["n_ops"] > 0).sum()) if args.steps <= 1 else 0
        return surv

    for _ in range(max(args.warmup, 1)):
        step()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dt = time.perf_counter() - t0
    survivors = sum(counts) if args.entry == "list" else sum(int((k[1]["n_ops"] > 0).sum()) for k in keep)
    # the sweep of the LAST call of the last step (the library's HIP events around every launch of the call: lx_last_phase_ms sums a
    # call's chunks), its executed cells (lx_last_extend_stats) and its algorithmic bytes
    st_last = h.last_extend_stats()
    sweep_ms, sweep_launches = h.last_phase_ms(0)
    bt_ms, bt_launches = h.last_phase_ms(3)
    sweep_name = h.last_kernel_name()
    last_slot, last_q, last_ext = parts[-1]
    tot = torch.tensor([dt, cells_rank, float(sum(len(e) for _, _, e in parts)), float(survivors)], dtype=torch.float64, device=dev)
    if use_dist:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt = float(mx[0].item())
    if rank == 0:
        total_cells, total_ext, total_surv = float(tot[1].item()), float(tot[2].item()), float(tot[3].item())
        up = total_ext / world * 28 + sum(len(q) for _, q, _ in parts)
        print(json.dumps({
            "metric": "GCUPS (gapped extension through the HOST-buffer entry point lx_extend_batch: PCIe and host work included; "
                      "full-rectangle parity mode, pass-1 cells per second of whole step) " + ("searchp BLOSUM62" if w.program == "blastp" else w.name),
            "value": round(total_cells * args.steps / dt / 1e9, 2), "unit": "GCUPS", "n_gpus": world, "steps": args.steps,
            "warmup": max(args.warmup, 1), "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": pl.scaling,
            "vs_baseline": None, "dtype": "f16x2 (exact small integers) + int32", "data": "synthetic",
            "config": {"workload": workloads.describe(w) if not args.ragged else
                       f"RAGGED seed list, {w.program} scheme of configs[{w.key}]: {args.ragged_queries} queries of {(args.lq_range or (50, 400))[0]}-"
                       f"{(args.lq_range or (50, 400))[1]} residues, windows per query geometric (mean {args.ragged_mean_windows or (8 if args.strong else 12):g}), "
                       f"10 % merged windows of up to 3 Lq, half homologous" + (" (strong hits: scores beyond the compact codes' 2046)" if args.strong else "") +
                       "; cells = sum Lq*Ls",
                       "baseline_config": args.config, "host_path": True, "ragged": bool(args.ragged),
                       "padding": (lambda st: {"extensions": st[0], "slots": st[1], "cells": st[2], "executed_cells": st[3],
                                               "padded_share_of_executed_cells": round(1 - st[2] / max(st[3], 1), 4)})(h.last_extend_stats()),
                       "entry_point": {"bytes": "lx_extend_batch", "rle": "lx_extend_batch_rle", "list": "lx_extend_batch_list"}[args.entry] +
                                      "(host buffers; subjects resident via lx_set_subjects; result arrays kept by the caller)",
                       "extensions_job": int(total_ext), "gcells_job": round(total_cells / 1e9, 3), "survivors_job": int(total_surv),
                       "bytes_up_per_step_rank0": int(up),
                       "bytes_down_per_step_rank0_approx": int(total_ext / world * 4 + total_surv / world * 60),
                       "step": f"lx_extend_batch per device call: host grouping + padding -> chunk pipeline (upload, single sweep, selection, "
                               f"backtrace, run-length ops) -> scores, records and ops back in the caller's arrays; cut-off score>={min_score}"},
            "alignments_per_s": round(total_ext * args.steps / dt, 1), "traced_per_s": round(total_surv * args.steps / dt, 1),
            "roofline": host_roofline(args, sweep_name, sweep_ms, sweep_launches, bt_ms, bt_launches, st_last, last_q, last_ext),
            "cpu_baseline": host_cpu_baseline(args, w, last_q, np.concatenate(s_all), last_ext) if (world == 1 and not args.no_cpu_baseline) else None,
            "note": "secondary line: this one prices the boundary a binding crosses; its roofline is the sweep of the last device call of the step "
                    "(all its launches), on the cells of the list and on the cells the wavefronts execute (padding included)",
        }), flush=True)
    h.close()
    if use_dist:
        dist.destroy_process_group()


def iterate_path(args, world, rank, local_rank, dev, use_dist):
    """The Level-2 driver as a caller sees it: ONE call = iterateMatchesFullSimd (/root/reference/src/search_algo.hpp:1177-1332) over
    the seed list of a whole read set -- widen, sort, merge, unique (:1136-1175), pass 1, filter, pass 2, records -- wall clock
    around K calls, the matches in device memory (`lx_iterate_matches_dev`, the sequence sets resident) or, `--entry host`, in host
    memory (`lx_iterate_matches`).  Synthetic list of BASELINE configs[2]'s size (lambda_amd/synth.py make_seed_list_np)."""
    import torch
    import torch.distributed as dist

    from lambda_amd import capi, synth, workloads

    # --config 1: the protein list of the metric's own program (searchp BLOSUM62, configs[1]: 100 000 queries x ~32 windows);
    # default (and --config 2): the read set of configs[2]
    protein = getattr(args, "config", None) == 1 and getattr(args, "config_given", False)
    w = workloads.WORKLOADS[1 if protein else 2]
    d = w.directions[0]
    h = capi.Handle(local_rank)
    m_, ma, mi, go, ge = d.scoring
    h.set_scoring(capi.builtin_scoring(m_, match=ma, mismatch=mi, gap_open=go, gap_extend=ge), d.slot)
    h.set_option(capi.LX_OPT_TRACE_BYTES, args.trace_bytes)
    ka = capi.karlin_params(*w.karlin)
    if protein:
        n_q = args.iterate_queries
        q, qoff, qlen, qorig, s, soff, slen, m = synth.make_protein_seed_list_np(n_q, seed=w.seed + rank, lq=w.lq, homologs=w.windows // 2, spurious=w.windows // 2)
        frames = 1
        # (dbTotalLength is configs[1]'s -- a Swiss-Prot-sized database, SURVEY.md section 8d -- whatever the resident subject set holds: the
        # database only enters through it, src/search_algo.hpp:317-319)
        params = capi.SearchParams(w.max_evalue, -1, 0, w.db_length, 0, 1, 1, 0, capi.LX_FRAMES_NONE, capi.LX_FRAMES_NONE, ka)
    else:
        q, qoff, qlen, qorig, s, soff, slen, m = synth.make_seed_list_np(args.iterate_reads, args.iterate_mbp, seed=0x1A3BDA03 + rank)
        frames = 2
        params = capi.SearchParams(w.max_evalue, -1, 0, int(slen.sum()), 0, 2, 1, 0, capi.LX_FRAMES_REVCOMP, capi.LX_FRAMES_NONE, ka)
    h.set_subjects(s)
    on_dev = args.entry != "host"
    if on_dev:
        h.set_subject_seqs(soff, slen)
        h.set_queries(q, qoff, qlen, qorig, frames)
        d_m = torch.from_numpy(m.view(np.uint8).copy()).to(dev)
        torch.cuda.synchronize()
    import ctypes as C

    lib = capi.load()

    def timed_call():
        r = C.c_void_p()
        if on_dev:
            t0 = time.perf_counter()
            h._check(lib.lx_iterate_matches_dev(h.h, 0, d_m.data_ptr(), len(m), C.byref(params), C.byref(r)))
            return r, time.perf_counter() - t0
        mm = m.copy()  # (the host entry point works in place, like the reference's span)
        t0 = time.perf_counter()
        h._check(lib.lx_iterate_matches(h.h, 0, capi._ptr(q), q.size, capi._ptr(qoff), capi._ptr(qlen), len(qoff), capi._ptr(qorig), None, 0,
                                        capi._ptr(soff), capi._ptr(slen), len(soff), capi._ptr(mm), len(mm), C.byref(params), C.byref(r)))
        return r, time.perf_counter() - t0

    def digest(r):
        """count, statistics and a checksum of the records + columns of a result (xxhash of the library's own arrays)"""
        import xxhash

        n = int(lib.lx_iterate_result_count(r))
        st_ = lib.lx_iterate_result_stats(r)
        rows = C.string_at(lib.lx_iterate_result_matches(r), n * capi.BLAST_MATCH_DTYPE.itemsize) if n else b""
        cols = b""
        ops = lib.lx_iterate_result_ops(r)
        if n and ops:
            bm = np.frombuffer(rows, dtype=capi.BLAST_MATCH_DTYPE)
            cols = C.string_at(ops, int(bm["ops_off"][-1]) + int(bm["n_ops"][-1]))
        return n, st_, xxhash.xxh64(rows).hexdigest() + xxhash.xxh64(cols).hexdigest()

    # hints a search has before the call: the seeding stage's match count; windows and records as a share of it (this list: 7.6 matches
    # per window, 12.7 per record) with a margin -- estimates that fall short only move an allocation back into the call
    hints = (len(m), len(m) // 2, len(m) // 4, (len(m) // 4) * 160) if protein else (len(m), len(m) // 7, len(m) // 12, (len(m) // 12) * 160)
    stats = None
    for _ in range(max(args.warmup, 1)):
        r, _ = timed_call()
        stats = digest(r)
        lib.lx_iterate_result_free(r)
    if use_dist:
        dist.barrier()
    times, setup = [], []
    for _ in range(args.steps):
        if args.cold and on_dev:
            h.close()
            t0 = time.perf_counter()
            h = capi.Handle(local_rank)
            h.set_scoring(capi.builtin_scoring(m_, match=ma, mismatch=mi, gap_open=go, gap_extend=ge), d.slot)
            h.set_option(capi.LX_OPT_TRACE_BYTES, args.trace_bytes)
            h.set_subjects(s)
            h.set_subject_seqs(soff, slen)
            h.set_queries(q, qoff, qlen, qorig, frames)
            h.reserve(*hints)
            setup.append(time.perf_counter() - t0)
        r, dt = timed_call()
        times.append(dt)
        got = digest(r)  # (outside the timed region)
        lib.lx_iterate_result_free(r)  # (the caller's to free: not part of the call)
        if got[0] != stats[0] or got[2] != stats[2] or got[1].num_ext_ali != stats[1].num_ext_ali:
            raise SystemExit(f"bench.py --iterate: a timed call returned other records than the warm-up call ({got[0]} / {stats[0]} records, checksums {got[2]} / {stats[2]})")
    dt = float(sum(times))
    n_hsp, st, checksum = stats
    if n_hsp != st.num_ext_ali - st.failed_identity:
        raise SystemExit(f"bench.py --iterate: {n_hsp} records of {st.num_ext_ali} traced windows and {st.failed_identity} below the identity cut-off")
    n_win = len(m) - st.hits_duplicate
    xs = h.last_extend_stats()
    # (pass 1 of the last call: all launches of its sweep; the window list again, through the widen entry point, for the byte count and the
    # CPU sample -- outside the timed region)
    sweep_ms, sweep_launches = h.last_phase_ms(0)
    bt_ms, bt_launches = h.last_phase_ms(3)
    sweep_name = h.last_kernel_name()
    roof = base = None
    if rank == 0 and on_dev:
        win = h.widen_and_preprocess_dev(d_m, len(m))
        ext = np.zeros(len(win), dtype=capi.EXT_DTYPE)
        ext["q_off"], ext["q_len"] = qoff[win["qryId"]], qlen[win["qryId"]]
        ext["s_off"] = soff[win["subjId"]] + win["subjStart"]
        ext["s_len"] = win["subjEnd"] - win["subjStart"]
        roof = host_roofline(args, sweep_name, sweep_ms, sweep_launches, bt_ms, bt_launches, xs, q, ext, flag="--iterate")
        if world == 1 and not args.no_cpu_baseline:
            base = host_cpu_baseline(args, w, q, s, ext)
            base["sample"] += "; pass 1 of the driver only (the sort, pass 2 and the records are not in it)"
    if rank == 0:
        print(json.dumps({
            "metric": ("ms per FIRST call of a fresh handle" if args.cold and on_dev else "ms per call")
                      + " of the Level-2 driver (iterateMatchesFullSimd: widen + sort + merge + unique, pass 1, filter, pass 2, records) on a "
                      + ("DEVICE match list, lx_iterate_matches_dev" if on_dev else "HOST match list, lx_iterate_matches")
                      + ("; searchp BLOSUM62 scheme of configs[1]" if protein else "; searchn scheme of configs[2]"),
            "value": round(dt / args.steps * 1e3, 3), "unit": "ms", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
            "ms_per_step": round(dt / args.steps * 1e3, 3), "ms_min": round(min(times) * 1e3, 3), "ms_max": round(max(times) * 1e3, 3),
            "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f16x2 (exact small integers) + int32 + u64 sort words",
            "data": "synthetic",
            "config": {"workload": (f"seed list of {len(qoff)} queries x {int(qorig[0])} aa against {len(soff)} proteins ({s.size / 1e6:.1f} M residues resident; "
                                    f"dbTotalLength {w.db_length} as configs[1] states): " if protein else
                                    f"seed list of {args.iterate_reads} reads x {int(qorig[0])} bp (two query frames each) against {args.iterate_mbp:g} Mbp in "
                                    f"{len(soff)} contigs: ")
                                   + f"{len(m)} matches in the order a seeding kernel's lanes emit them -> {n_win} windows "
                                   f"({xs[2] / 1e9:.1f} Gcells) -> {st.num_ext_ali} traced -> {n_hsp} HSPs; E <= {w.max_evalue:g}",
                       "baseline_config": w.key,
                       "entry_point": "lx_iterate_matches_dev (matches in device memory, sequence sets resident: lx_set_queries / lx_set_subjects / "
                                      "lx_set_subject_seqs)" if on_dev else "lx_iterate_matches (host buffers, subjects resident)",
                       "matches": len(m), "windows": int(n_win), "traced": int(st.num_ext_ali), "hsps": n_hsp, "records_checksum": checksum,
                       **({"cold": "every timed call is the first call of a fresh handle: lx_create + lx_set_scoring / lx_set_subjects / lx_set_subject_seqs / "
                                   "lx_set_queries + lx_reserve%r outside the timed region (%.1f ms on average), then ONE lx_iterate_matches_dev"
                                   % (hints, 1e3 * sum(setup) / max(len(setup), 1))} if args.cold and on_dev else {}),
                       "padding": {"extensions": xs[0], "slots": xs[1], "cells": xs[2], "executed_cells": xs[3]}},
            "gcups_of_window_cells": round(xs[2] * args.steps / dt / 1e9, 1),
            "matches_per_s": round(len(m) * args.steps / dt, 1),
            "roofline": roof, "cpu_baseline": base,
            "note": "secondary line: prices the whole driver call around the kernels of the headline line; its roofline is pass 1's sweep (all "
                    "launches of the last call)",
        }), flush=True)
    h.close()
    if use_dist:
        dist.destroy_process_group()


def host_roofline(args, kernel, ms, launches, bt_ms, bt_launches, st, q, ext, flag=None):
    """Roofline of the sweep of ONE host-buffer call (all its launches: a call is a pipeline of chunks): integer-VALU bound as the headline
    line's; `achieved` counts the LIST's cells (10 algorithmic ops each), `frac_executed` what the wavefronts execute
    (padded columns and rows included: lx_last_extend_stats) -- the distance between the two is the padding of the plan."""
    if ms <= 0:
        return None
    cells, executed = float(st[2]), float(st[3])
    packed = "pair" in kernel or "sweep_mq" in kernel
    peak = PEAK_PACKED16_TOPS if packed else PEAK_INT32_TOPS
    tops = cells / (ms * 1e-3) * ALGO_OPS_PER_CELL / 1e12
    tops_x = executed / (ms * 1e-3) * ALGO_OPS_PER_CELL / 1e12
    # algorithmic bytes: every window once, every query once, one 24-byte record + one score per extension
    qkeys = np.unique(ext["q_off"])
    algo = float(ext["s_len"].sum()) + float(len(q) if len(qkeys) else 0) + len(ext) * ALGO_BYTES_PER_EXT_EXTRA
    flag = flag or ("--ragged" if args.ragged else "--host-path")
    # (the PMC passes profile the kernel instantiation the sweep ran as; its name in the library's spelling ends at the first blank)
    other_config = getattr(args, "config", 1) not in (None, 1) and flag in ("--ragged", "--host-path")
    # (--iterate: the read set of configs[2] unless --config 1 asks for the protein list)
    iterate_protein = flag == "--iterate" and getattr(args, "config", None) == 1 and getattr(args, "config_given", False)
    traffic, note = pmc_traffic(kernel.split(" (")[0], command_has=flag,
                                command_also=f"--config {args.config}" if other_config else "--config 1" if iterate_protein else None,
                                command_lacks=None if other_config or iterate_protein else "--config")
    ceil = issue_ceiling() if packed else None
    return {
        **({"issue_ceiling_frac": ceil["issue_ceiling_frac"]} if ceil else {}),
        "bound": "valu", "kernel": kernel, "achieved": round(tops, 3), "peak": round(peak, 2),
        "unit": "Tops/s (%s lane-ops; 10 algorithmic ops per cell of the list)" % ("packed 16-bit" if packed else "int32"),
        "frac": round(tops / peak, 4), "frac_executed": round(tops_x / peak, 4),
        "kernel_ms_per_call": round(ms, 4), "launches_per_call": launches, "kernel_gcups": round(cells / (ms * 1e-3) / 1e9, 1),
        "kernel_gcups_executed": round(executed / (ms * 1e-3) / 1e9, 1), "cells_per_call": cells, "executed_cells_per_call": executed,
        "backtrace_ms_per_call": round(bt_ms, 4), "backtrace_launches_per_call": bt_launches,
        "hbm": {"bound": "hbm", "algorithmic_bytes_per_call": algo, "achieved": round(algo / (ms * 1e-3) / 1e9, 2), "peak": 8000, "unit": "GB/s",
                "frac": round(algo / (ms * 1e-3) / 1e9 / 8000, 5)},
        "traffic": traffic, "traffic_over_algorithmic": round(traffic / algo, 2) if traffic else None, "traffic_note": note,
    }


def host_cpu_baseline(args, w, q, s, ext):
    """The oracle's inter-sequence int16 SIMD scorer on a bounded sample of the SAME list (its first queries' windows, ~20 Gcells), on this
    box's host cores."""
    from lambda_amd import capi
    from tests import oracle_lib

    cores, note = usable_cpus()
    orc = oracle_lib.load()
    d = w.directions[0]
    m, ma, mi, go, ge = d.scoring
    sc = oracle_lib.scoring_from(capi.builtin_scoring(m, match=ma, mismatch=mi, gap_open=go, gap_extend=ge))
    cum = np.cumsum(ext["q_len"].astype(np.float64) * ext["s_len"])
    k = int(np.searchsorted(cum, 20e9)) + 1
    sample = np.ascontiguousarray(ext[:k])
    cells = float(cum[min(k, len(cum)) - 1])
    best = float("inf")
    for _ in range(2):
        t0 = time.perf_counter()
        orc.score_batch(q, s, sample, sc, threads=cores, simd=True)
        best = min(best, time.perf_counter() - t0)
    return {"value": round(cells / best / 1e9, 3), "unit": "GCUPS", "cores": cores, "kind": "port",
            "sample": f"the first {len(sample)} extensions of this list ({cells / 1e9:.2f} Gcells), oracle inter-sequence int16 SIMD restatement (NOT SeqAn), "
                      f"OpenMP with {cores} threads ({note}), best of 2, {best:.3f} s"}


class DevBatch:
    """One device call's inputs and outputs, resident in HBM."""

    def __init__(self, w, b, dev, min_score, band=0, homolog_frac=0.5):
        import torch

        from lambda_amd import synth, workloads

        self.slot = b.direction.slot
        self.n_queries = b.n_queries
        self.q_first = b.q_lo
        d_q, d_s, self.d_ext, ext = synth.make_batch_torch(
            b.n_queries, w.lq, w.windows, b.seed, dev, alphabet=workloads.alphabet_array(w), sub_rate=w.sub_rate,
            indel_rate=w.indel_rate, n_rate=w.n_rate, n_rank=w.n_rank, convert=b.direction.convert, convert_rate=w.convert_rate,
            homolog_frac=homolog_frac)
        pad = torch.zeros(256, dtype=torch.uint8, device=dev)
        self.d_q = torch.cat([d_q, pad])
        self.d_s = torch.cat([d_s, pad])
        self.n = n = len(ext)
        self.max_slen = int(ext["s_len"].max())
        self.cells = float((ext["q_len"].astype(np.float64) * (np.minimum(ext["s_len"], 2 * band + 1) if band > 0 else ext["s_len"])).sum())
        self.q_bytes, self.s_bytes = float(ext["q_len"].sum()) / w.windows, float(ext["s_len"].sum())
        self.d_score = torch.zeros(n, dtype=torch.int32, device=dev)
        # pass-2 outputs (worst case sizes: every extension may survive)
        sizes = ext["q_len"].astype(np.uint64) + ext["s_len"].astype(np.uint64)
        off = np.zeros(n, dtype=np.uint64)
        off[1:] = np.cumsum(sizes)[:-1]
        self.d_off = torch.from_numpy(off.view(np.int64)).to(dev)
        self.d_ops = torch.zeros(int(sizes.sum()) + 16, dtype=torch.uint8, device=dev)
        self.d_hsp = torch.zeros(n * 48, dtype=torch.uint8, device=dev)
        self.d_count = torch.zeros(2, dtype=torch.int64, device=dev)
        self.min_score = min_score


def main():
    args = parse()
    maybe_self_spawn(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    w = workload_of(args)
    if args.dry_run:
        return dry_run(args, w, world, rank)

    import ctypes as C

    import torch
    import torch.distributed as dist

    from lambda_amd import capi, shard, synth, workloads

    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: no device {local_rank} on this node ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("LX_BENCH_FORCE_DIST") == "1"  # the env switch lets one GPU exercise the RCCL path
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    pl = workloads.plan(w, world, rank, args.total_queries, args.queries, args.batch_queries)
    if args.iterate:
        return iterate_path(args, world, rank, local_rank, dev, use_dist)
    if args.host_path or args.ragged:
        return host_path(args, w, pl, world, rank, local_rank, dev, use_dist)

    # ---- scoring schemes and the e-value filter of iterateMatchesFullSimd as an integer score cut-off
    h = capi.Handle(local_rank)
    for d in w.directions:
        m, ma, mi, go, ge = d.scoring
        h.set_scoring(capi.builtin_scoring(m, match=ma, mismatch=mi, gap_open=go, gap_extend=ge), d.slot)
    h.set_option(capi.LX_OPT_BS_MATCH_RULE, 1 if len(w.directions) > 1 else 0)
    h.set_option(capi.LX_OPT_MAX_QLEN, w.lq)
    h.set_option(capi.LX_OPT_QUERY_RUN, args.query_run if args.query_run else (w.windows if w.windows % 8 == 0 else 0))
    if args.mq_sweep is not None:
        h.set_option(capi.LX_OPT_MQ_SWEEP, args.mq_sweep)
    if args.adapt_permille is not None:
        h.set_option(capi.LX_OPT_ADAPT_PERMILLE, args.adapt_permille)
    h.set_option(capi.LX_OPT_TRACE_BYTES, args.trace_bytes)
    if args.band > 0:
        h.set_band(args.band)  # default centres: min(_bandSize(Lq), Ls - Lq), the seed diagonal of the synthetic windows
    ka = capi.karlin_params(*w.karlin)
    lib = capi.load()
    adj = lib.lx_length_adjustment(w.db_length, w.lq, C.byref(ka))
    min_score = 1
    while lib.lx_evalue(min_score, w.lq - adj, w.db_length - adj, C.byref(ka)) > w.max_evalue:
        min_score += 1

    # ---- workload, generated directly in HBM; every rank owns different queries (shard by query)
    batches = [DevBatch(w, b, dev, min_score, args.band, 0.5 if args.survivor_rate is None else args.survivor_rate) for b in pl.batches]
    if not batches:
        raise SystemExit(f"rank {rank}: no queries to process (job of {pl.job_queries} queries over {world} ranks)")
    h.set_option(capi.LX_OPT_MAX_SLEN, max(b.max_slen for b in batches))
    cells_rank = sum(b.cells for b in batches)
    n_rank = sum(b.n for b in batches)
    # a non-default torch stream: its handle is non-NULL, so the kernels really run on the stream the timing events
    # are recorded on (NULL would select the lx handle's private stream)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()

    def step():
        for b in batches:
            if args.pass1_only:
                h.score_batch_dev(b.d_q, b.d_s, b.d_ext, b.n, b.d_score, slot=b.slot, stream=stream.cuda_stream)
            else:
                h.extend_batch_dev(b.d_q, b.d_s, b.d_ext, b.n, b.min_score, b.d_score, b.d_hsp, b.d_ops, b.d_off, b.d_count,
                                   slot=b.slot, stream=stream.cuda_stream)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def gather():
        # the path's only exchange: every rank keeps, per query, its best `maxMatches` = 25
        # HSPs (src/search_options.hpp:99) among the survivors of the e-value filter, then one gather of fixed-size
        # records [global extension id, lx_hsp (48 B)] = 56 B each -- RCCL over xGMI.  Query ranges are disjoint, so
        # rank order is the final order.
        recs = []
        for b in batches:
            base = b.q_first * w.windows  # global extension numbering of this direction's queries
            if args.pass1_only or w.windows <= 0:
                hit = torch.nonzero(b.d_score >= b.min_score).flatten()
                recs.append(torch.stack([hit + base, b.d_score[hit].to(torch.int64)], dim=1))
                continue
            sc2 = b.d_score.view(b.n_queries, w.windows)
            k = min(args.max_matches, w.windows)
            top, idx = torch.topk(sc2, k, dim=1)
            keep = top >= b.min_score
            ext_id = (idx + torch.arange(b.n_queries, device=dev).unsqueeze(1) * w.windows)[keep]
            hsp64 = b.d_hsp.view(torch.int64).view(b.n, 6)[ext_id]
            recs.append(torch.cat([(ext_id + base).unsqueeze(1), hsp64], dim=1))
        # all_gather (dst=None): ~2 ms for 8 x 90 MB over xGMI, once per run; a gather to the writing rank costs the same
        return shard.gather_hits(torch.cat(recs, dim=0), dst=None)

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
    survivors = sum(int(b.d_count.cpu()[1]) for b in batches) if not args.pass1_only else 0
    last = batches[-1]  # the library's phase events are those of the most recent call: the last batch of the last step
    survivors_last = int(last.d_count.cpu()[1]) if not args.pass1_only else 0

    # whole-job figures need every rank's share (strong scaling: ranks own different numbers of queries)
    tot = torch.tensor([dt, cells_rank, float(n_rank), float(survivors)], dtype=torch.float64, device=dev)
    if use_dist:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt = float(mx[0].item())
    total_cells, total_ext, total_surv = float(tot[1].item()), float(tot[2].item()), float(tot[3].item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    # device time per phase of the LAST call of the last timed step (HIP events recorded by the library on the launch
    # stream, around each kernel launch): 0 = pass-1 score kernel, 1 = selection, 2 = pass-2 forward kernel, 3 = backtrace
    phase_ms = {ph: h.last_phase_ms(ph) for ph in (0, 1, 2, 3)}
    trace_kernel_name = h.last_trace_kernel_name()

    if rank == 0:
        gcups = total_cells * args.steps / dt / 1e9
        lq, ls = w.lq, synth.window_len(w.lq)

        def roofline(kernel, ms, launches, cells, pmc_key, algo_bytes, stored_bytes):
            """The DP kernels are bound by integer VALU issue (not HBM, not MFMA), so `bound` is
            "valu" and achieved/peak count lane-ops of ONE launch (cells of the last device call / its duration); the HBM
            side of the same launch -- algorithmic bytes (windows, queries, records, scores), the
            checkpoint bytes the kernel additionally writes for pass 2, and the PMC traffic -- is reported next to it."""
            gc = cells / (ms * 1e-3) / 1e9
            tops = gc * ALGO_OPS_PER_CELL / 1e3
            headline = args.config == 1 and args.lq is None and args.windows is None and args.band <= 0
            traffic, note = pmc_traffic(pmc_key) if headline else (None, "PMC passes are committed for the headline batch only")
            packed = "pair" in kernel or "sweep_mq" in kernel
            peak = PEAK_PACKED16_TOPS if packed else PEAK_INT32_TOPS
            hbm = (algo_bytes + stored_bytes) / (ms * 1e-3) / 1e9
            ceil = issue_ceiling() if packed else None
            return {
                **({"issue_ceiling_frac": ceil["issue_ceiling_frac"], "frac_of_issue_ceiling": round(tops / ceil["ceiling_tops"], 4),
                    "issue_ceiling": ceil} if ceil else {}),
                "bound": "valu", "kernel": kernel, "achieved": round(tops, 3), "peak": round(peak, 2),
                "unit": "Tops/s (%s lane-ops; 10 algorithmic ops per cell)" % ("packed 16-bit" if packed else "int32"),
                "frac": round(tops / peak, 4),
                "kernel_ms_per_launch": round(ms, 4), "launches": launches, "kernel_gcups": round(gc, 1),
                "cells_per_launch": cells,
                "hbm": {"bound": "hbm", "achieved": round(hbm, 2), "peak": 8000, "unit": "GB/s", "frac": round(hbm / 8000, 4),
                        "algorithmic_bytes_per_launch": algo_bytes, "checkpoint_bytes_written_per_launch": stored_bytes,
                        "write_amplification": round((algo_bytes + stored_bytes) / algo_bytes, 2)},
                "traffic": traffic,
                "traffic_over_algorithmic": round(traffic / algo_bytes, 2) if traffic else None,
                "traffic_note": note,
            }

        # algorithmic bytes of pass 1: every window once, every query once per run, one 24-byte
        # record per extension, one int32 score written
        algo_bytes = last.q_bytes + last.s_bytes + last.n * ALGO_BYTES_PER_EXT_EXTRA
        stored = 0.0
        if "single sweep" in kernel_name:
            # the sweep also writes the checkpoints of every extension: one boundary pair per strip and row, one pair per
            # column every 16 rows -- 2-byte codes from the packed-half kernel, int16 pairs otherwise -- and a 16-byte end
            # record.  NOT algorithmic: the price of fusing pass 2's forward half into pass 1.
            pair_bytes = 2 if "score_pair_kernel" in kernel_name else 4
            strip = 13 if "<16,13" in kernel_name else 19
            stored = last.n * (ls * (-(-lq // strip)) * pair_bytes + (ls / 16.0) * lq * pair_bytes + 16)
        r_score = roofline(kernel_name, phase_ms[0][0], phase_ms[0][1], last.cells, kernel_name, algo_bytes, stored)
        rooflines = [r_score]
        if not args.pass1_only and phase_ms[2][0] > 0:
            cells2 = float(survivors_last) * lq * ls
            if "ckpt" in trace_kernel_name:
                strips = -(-lq // 19)
                stored2 = survivors_last * (ls * strips * 4 + (ls / 16.0) * lq * 4)
                pmc_key = trace_kernel_name
            else:
                stored2 = survivors_last * lq * ls / 2.0  # 4 direction bits per cell
                pmc_key = trace_kernel_name
            algo2 = survivors_last * (ls + lq / 4.0 + ALGO_BYTES_PER_EXT_EXTRA + 4)
            rooflines.append(roofline(trace_kernel_name, phase_ms[2][0], phase_ms[2][1], cells2, pmc_key, algo2, stored2))
        rooflines.sort(key=lambda r: -r["kernel_ms_per_launch"])
        packed_name = "score_pair_kernel" in kernel_name
        out = {
            "metric": ("GCUPS (gapped extension, full-rectangle parity mode; pass-1 cells per second of whole step) " if args.band <= 0 else
                       f"GCUPS (gapped extension, BAND MODE band={args.band}: cells = sum Lq*min(Ls, 2*band+1); not the reference's "
                       f"configuration, never to be mixed with the parity line) ")
                      + ("searchp BLOSUM62" if w.program == "blastp" else w.name),
            "value": round(gcups, 2),
            "unit": "GCUPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": pl.scaling,
            "vs_baseline": None,
            "dtype": ("f16x2 (exact small integers) + int32" if packed_name else
                      "i16x2 (packed 16-bit integers) + int32" if "pair16" in kernel_name else "int32"),
            "data": "synthetic",
            "config": {
                "workload": workloads.describe(w) if args.band <= 0 else
                            workloads.describe(w).replace("cells = sum Lq*Ls (full rectangle, band off as in the reference)",
                                                          f"band={args.band} around the seed diagonal: cells = sum Lq*min(Ls, {2 * args.band + 1})"),
                "baseline_config": args.config,
                "band": args.band,
                "job_queries": pl.job_queries,
                "queries_rank0": pl.n_queries,
                "device_calls_per_step_rank0": len(batches),
                "extensions_job": int(total_ext),
                "gcells_job": round(total_cells / 1e9, 3),
                "parallelism": f"query-sharded x{world} ({pl.scaling} scaling), no data-path collective; one gather of the per-query "
                               f"top-{args.max_matches} HSP records (56 B) after the last step"
                               + (f" ({n_hits_total} records)" if n_hits_total is not None else ""),
                "step": ("pass 1 score kernel over the whole share" if args.pass1_only else
                         ("single sweep: " if "single sweep" in kernel_name else "") +
                         f"pass 1 (score all) -> e-value filter (E<={w.max_evalue:g} at db {w.db_length}, i.e. score>={min_score}) "
                         f"-> pass 2 (traceback of the {int(total_surv)} survivors of the job), all on the GPU; GCUPS counts pass-1 cells only"),
                "survivors_job": int(total_surv),
                "pass2_gcells_job": round(total_surv * lq * ls / 1e9, 3),
            },
            "alignments_per_s": round(total_ext * args.steps / dt, 1),
            "traced_per_s": round(total_surv * args.steps / dt, 1),
            "total_gcups_both_passes": round((total_cells + total_surv * lq * ls) * args.steps / dt / 1e9, 2),
            "phase_ms_last_call": {"score": round(phase_ms[0][0], 3), "select": round(phase_ms[1][0], 3),
                                   "trace_forward": round(phase_ms[2][0], 3), "backtrace": round(phase_ms[3][0], 3),
                                   "events_step_ms": round(kern_ms, 3)},
            "roofline": rooflines[0],           # the kernel that takes most of the step
            "roofline_other": rooflines[1:],
        }
        if world == 1 and not args.no_cpu_baseline:
            cores, note = usable_cpus()
            # ~84 Gcells of CPU work per repetition at most (a second or so on the box's 16 granted CPUs), three repetitions
            default_sample = max(1, min(pl.n_queries, int(84.5e9 / (w.windows * lq * ls))))
            out["cpu_baseline"] = cpu_baseline(w, min(args.cpu_sample_queries or default_sample, pl.n_queries), cores, note)
        print(json.dumps(out), flush=True)
    h.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
