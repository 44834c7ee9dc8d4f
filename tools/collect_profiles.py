"""Condenses a gpurun_out/<dir> produced by the round's rocprofv3 runs into the tracked summaries under profiles/.

    python tools/collect_profiles.py gpurun_out/r1b r01

Inputs (made on the GPU box):
    <dir>/stats/bench_kernel_stats.csv          rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py ...
    <dir>/pmc_sq|pmc_sq_wait|pmc_fetch|pmc_write/pmc_counter_collection.csv   separate --pmc passes
    <dir>/bench.log, bench_pass1.log, stats.log  the bench JSON lines of the plain / pass-1-only / profiled runs
"""
import collections
import csv
import json
import sys
from pathlib import Path

src, tag = Path(sys.argv[1]), sys.argv[2]
out = Path(__file__).resolve().parent.parent / "profiles"
out.mkdir(exist_ok=True)
ROOT = out.parent


def git_blob(path: Path) -> str:
    """the git blob id of a file's content (what `git hash-object` prints)"""
    import hashlib

    data = path.read_bytes()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


# the sources of the kernels the PMC figures belong to, as they stood when the passes ran: bench.py names them beside the traffic it
# reads from these files and tests/test_bench_cli.py fails when a kernel changed after its newest profile
KERNEL_SOURCES = {f"lambda_amd/csrc/{n}": git_blob(ROOT / "lambda_amd" / "csrc" / n)
                  for n in ("lx_score_f16.hip", "lx_sweep_mq.hip", "lx_ckpt.hip", "lx_dp_common.h", "lx_device.h")}


def condense(names, steps):
    kern = {}
    for name in names:
        f = src / name / "pmc_counter_collection.csv"
        if not f.exists():
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        meta, launches = {}, collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "lx::" not in k:
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r["Dispatch_Id"])
            meta[k] = {x: r[x] for x in ("Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size")}
        for k, v in acc.items():
            d = kern.setdefault(k, {"dispatch": meta[k], "counters_per_step_mean": {}, "counters_per_launch_mean": {}})
            for c, x in v.items():
                d["counters_per_step_mean"][c] = x / steps
                d["counters_per_launch_mean"][c] = x / len(launches[k])
    return kern

rows = list(csv.reader(open(src / "stats" / "bench_kernel_stats.csv")))
with open(out / f"{tag}_bench_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:]:
        if "lx::" in r[0] or float(r[4]) > 1.0:
            w.writerow(r)

steps_profiled = 3  # --steps 2 --warmup 1
kern = {}
for name in ("pmc_sq", "pmc_sq_wait", "pmc_fetch", "pmc_write"):
    if not (src / name / "pmc_counter_collection.csv").exists():
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    meta, launches = {}, collections.defaultdict(set)
    for r in csv.DictReader(open(src / name / "pmc_counter_collection.csv")):
        k = r["Kernel_Name"]
        if "lx::" not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
        meta[k] = {x: r[x] for x in ("Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size")}
    for k, v in acc.items():
        d = kern.setdefault(k, {"dispatch": meta[k], "counters_per_step_mean": {}, "counters_per_launch_mean": {}})
        for c, x in v.items():
            d["counters_per_step_mean"][c] = x / steps_profiled
            d["counters_per_launch_mean"][c] = x / len(launches[k])
json.dump({
    "command": "rocprofv3 --pmc <counters> --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "
               "(separate passes: SQ_* instruction counts, SQ_* wave-cycle breakdown, FETCH_SIZE, WRITE_SIZE)",
    "note": "FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads 1/2 of the bytes of wide coalesced reads "
            "(MI355X_MICROARCH.md, HBM) -> doubled before use.  The score kernels are launched once per step, the trace "
            "kernels once per chunk, so per-step means are the comparable figures.  SQ_WAVE_CYCLES ~ SQ_WAIT_ANY (parked on "
            "s_waitcnt) + SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY, in quad-cycles summed over wavefronts.",
    "kernel_sources": KERNEL_SOURCES, "kernels": kern}, open(out / f"{tag}_bench_pmc.json", "w"), indent=1)

for log, dst in (("bench.log", f"{tag}_bench_line.json"), ("bench_pass1.log", f"{tag}_bench_line_pass1_only.json"),
                 ("stats.log", f"{tag}_bench_line_under_rocprof.json")):
    if not (src / log).exists():  # (tools/profile_round.sh <dir> profiles: the passes first, the bench lines once they are condensed)
        continue
    lines = [l for l in open(src / log) if l.startswith('{"metric"')]
    if lines:
        (out / dst).write_text(lines[-1])
# the secondary lines of the round (configs 2-4, band, host path, ragged, survivor rates) and the ragged list's kernel stats
for log in sorted(src.glob("bench_*.log")):
    if log.name in ("bench_pass1.log",):
        continue
    lines = [l for l in open(log) if l.startswith('{"metric"')]
    if lines:
        (out / f"{tag}_{log.stem}.json").write_text(lines[-1])
if (src / "host_curve.jsonl").exists():
    (out / f"{tag}_host_curve.jsonl").write_text("".join(l for l in open(src / "host_curve.jsonl") if l.startswith("{")))
rs = src / "stats_ragged" / "ragged_kernel_stats.csv"
if rs.exists():
    rows = list(csv.reader(open(rs)))
    with open(out / f"{tag}_ragged_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            if "lx::" in r[0]:
                w.writerow(r)
hs = src / "stats_host" / "host_kernel_stats.csv"  # the five chunks of lx_extend_batch_list on the headline batch (DESIGN_LOG.md section 8.8)
if hs.exists():
    rows = list(csv.reader(open(hs)))
    with open(out / f"{tag}_host_path_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            if "lx::" in r[0]:
                w.writerow(r)
lw = src / "pmc_write_lowsurv" / "pmc_counter_collection.csv"
if lw.exists():
    acc, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(lw)):
        if "lx::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]] += float(r["Counter_Value"])
            n[r["Kernel_Name"]].add(r["Dispatch_Id"])
    json.dump({"command": "rocprofv3 --pmc WRITE_SIZE -- python bench.py --steps 2 --warmup 3 --survivor-rate 0.02 --no-cpu-baseline",
               "note": "WRITE_SIZE in KiB per launch: the adaptive pass-2 mode at 2 % survivors (pass 1 writes scores only; checkpoints "
                       "are written for the survivors by ckpt_forward_kernel)",
               "write_kib_per_launch": {k: acc[k] / len(n[k]) for k in acc}}, open(out / f"{tag}_pmc_write_survivors_0.02.json", "w"), indent=1)
for k, d in kern.items():
    print(k[:60], {c: "%.4g" % x for c, x in d["counters_per_step_mean"].items()})

# round 4: PMC passes of the ragged list (the multi-query sweep and its backtrace), the Level-2 driver's kernels, the front end at scale
kern = {}
for name in ("pmc_ragged_sq", "pmc_ragged_sq_wait", "pmc_ragged_fetch", "pmc_ragged_write"):
    f = src / name / "pmc_counter_collection.csv"
    if not f.exists():
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    meta, launches = {}, collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "lx::" not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
        meta[k] = {x: r[x] for x in ("Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size")}
    for k, v in acc.items():
        d = kern.setdefault(k, {"dispatch": meta[k], "counters_per_step_mean": {}, "counters_per_launch_mean": {}})
        for c, x in v.items():
            d["counters_per_step_mean"][c] = x / steps_profiled
            d["counters_per_launch_mean"][c] = x / len(launches[k])
if kern:
    json.dump({
        "command": "rocprofv3 --pmc <counters> --output-format csv -- python bench.py --ragged --entry list --steps 2 --warmup 1 --no-cpu-baseline "
                   "(separate passes: SQ_* instruction counts, SQ_* wave-cycle breakdown, FETCH_SIZE, WRITE_SIZE)",
        "note": "the ragged list of bench.py (50 000 queries of 50-400 aa, 596 k windows, 44.3 G real / 57.8 G executed cells per call): per-STEP means "
                "are per lx_extend_batch_list call (two sweep launches: pool, stream).  FETCH_SIZE/WRITE_SIZE in KiB, FETCH_SIZE doubled before use.",
        "kernel_sources": KERNEL_SOURCES, "kernels": kern}, open(out / f"{tag}_ragged_pmc.json", "w"), indent=1)
rs = src / "stats_iterate" / "iterate_kernel_stats.csv"
if rs.exists():
    rows = list(csv.reader(open(rs)))
    with open(out / f"{tag}_iterate_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            if "lx::" in r[0]:
                w.writerow(r)
rs = src / "stats_iterate_protein" / "iterate_kernel_stats.csv"
if rs.exists():
    rows = list(csv.reader(open(rs)))
    with open(out / f"{tag}_iterate_protein_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            if "lx::" in r[0]:
                w.writerow(r)
lines = []
for log in ("cli_nucl.log", "cli_nucl_host_list.log"):
    if (src / log).exists():
        lines += [f"== {log} (tools/cli_scale_nucl.py 1000000 100" + ("; LAMBDA3_HOST_LIST=1 LX_ITERATE_ON_HOST=1: the list on the host, as round 3)" if "host" in log else ")")]
        lines += [l.rstrip() for l in open(src / log) if l.startswith(("lambda3 ", "output sha256", "rc ")) or "lx_iterate_matches_dev:" in l or "iterateMatchesFullSimd (" in l]
if lines:
    (out / f"{tag}_cli_end_to_end.txt").write_text("\n".join(lines) + "\n")

# round 5: FETCH_SIZE / WRITE_SIZE of the solo sweep (sweep_mq_kernel<19,false,false>) on the Level-2 driver's list and on the configs[2]-sized
# ragged list, so that those bench lines carry `traffic` too
for passes, cmd, dst in ((("pmc_iterate_sq", "pmc_iterate_sq_wait", "pmc_iterate_fetch", "pmc_iterate_write"), "--iterate", "iterate_pmc"),
                         (("pmc_ragged_nucl_fetch", "pmc_ragged_nucl_write"), "--ragged --entry list --config 2", "ragged_nucl_pmc"),
                         # round 6: the Level-2 driver on the protein list of configs[1]; the long strong-hit list
                         (("pmc_iterate_protein_sq", "pmc_iterate_protein_sq_wait", "pmc_iterate_protein_fetch", "pmc_iterate_protein_write"), "--iterate --config 1",
                          "iterate_protein_pmc"),
                         (("pmc_strong_sq", "pmc_strong_sq_wait", "pmc_strong_fetch", "pmc_strong_write"), "--ragged --entry list --lq-range 500 800 --strong",
                          "ragged_long_strong_pmc")):
    k = condense(passes, steps_profiled)
    if k:
        json.dump({"command": f"rocprofv3 --pmc <counters> --output-format csv -- python bench.py {cmd} --steps 2 --warmup 1 --no-cpu-baseline (separate passes)",
                   "note": "per-STEP means are per call of the entry point (all launches of the kernel in it), per-LAUNCH means per kernel launch.  "
                           "FETCH_SIZE/WRITE_SIZE in KiB, FETCH_SIZE doubled before use.",
                   "kernel_sources": KERNEL_SOURCES, "kernels": k}, open(out / f"{tag}_{dst}.json", "w"), indent=1)


# round 5: the spread of the secondary lines over runs and boxes (tools/dev/rerun_lines.sh writes it)
if (src / "line_spread.txt").exists():
    (out / f"{tag}_line_spread.txt").write_text((src / "line_spread.txt").read_text())
