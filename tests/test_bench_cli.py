"""CPU tests of bench.py's launch path and of the workload table (lambda_amd/workloads.py): `--gpus N` started without a
launcher re-executes itself under torch.distributed.run, ranks rendezvous (gloo in --dry-run), strong scaling splits the
job's queries like the reference splits them over its threads (/root/reference/src/search.cpp:384-385)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

from lambda_amd import synth, workloads

ROOT = Path(__file__).resolve().parent.parent


def run_bench(*argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_plain_launch_with_two_gpus_spawns_two_ranks():
    d = run_bench("--gpus", "2", "--config", "3", "--dry-run", "--total-queries", "1001")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["dry_run"]
    assert [r["rank"] for r in d["ranks"]] == [0, 1]
    assert d["ranks"][0]["q_lo"] == 0 and d["ranks"][0]["q_hi"] == d["ranks"][1]["q_lo"] and d["ranks"][1]["q_hi"] == 1001
    assert sum(c[1] for r in d["ranks"] for c in r["calls"]) == 1001
    assert "configs[3]" in d["config"]["workload"] and "200 aa" in d["config"]["workload"]


def test_ranks_of_a_node_share_its_cpus():
    """One process per GPU: every rank's library sizes its host threads to the granted CPUs divided by LOCAL_WORLD_SIZE (eight
    ranks must not start eight full pools on the CPUs of one box)."""
    d = run_bench("--gpus", "8", "--config", "3", "--dry-run", "--total-queries", "8000")
    assert d["n_gpus"] == 8 and len(d["ranks"]) == 8
    for r in d["ranks"]:
        assert r["local_world_size"] == 8 and 1 <= r["host_threads"] <= max(1, r["granted_cpus"] // 8), r
    assert sum(r["host_threads"] for r in d["ranks"]) <= max(8, d["ranks"][0]["granted_cpus"])


def test_single_rank_default_is_the_headline_config():
    d = run_bench("--dry-run")
    assert d["n_gpus"] == 1 and d["scaling"] == "weak"
    assert d["config"]["job_queries"] == 100_000 and abs(d["config"]["job_gcells"] - 84.48) < 1e-6
    assert d["ranks"][0]["calls"] == [[0, 100_000]]


def test_refuses_more_ranks_than_devices():
    env = {k: v for k, v in os.environ.items() if k != "WORLD_SIZE"}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "64"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "--gpus 64" in (r.stderr + r.stdout)


@pytest.mark.parametrize("key", [1, 2, 3, 4])
def test_workload_table_matches_baseline_configs(key):
    w = workloads.WORKLOADS[key]
    base = json.loads((ROOT / "BASELINE.json").read_text())["configs"][key]
    assert f"{w.lq}" in base  # query length as BASELINE.json states it
    per_ext = w.lq * synth.window_len(w.lq)
    # 84.5 Gcells (headline), 211 Gcells (searchn), 1.47 Tcells (scale-out)
    want = {1: 84.48e9, 2: 211.2e9, 3: 1.472e12, 4: 500_000 * 8 * 26_400}[key]
    assert abs(workloads.cells_of(w, w.queries_total) - want) / want < 1e-3
    assert per_ext == {150: 26_400, 200: 46_000}[w.lq]


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_strong_scaling_plan_partitions_the_job(world):
    w = workloads.WORKLOADS[3]
    seen = 0
    for r in range(world):
        p = workloads.plan(w, world, r)
        assert p.scaling == "strong" and p.job_queries == 1_000_000
        assert sum(b.n_queries for b in p.batches) == p.n_queries
        assert all(b.n_queries <= 125_000 for b in p.batches)
        seen += p.n_queries
    assert seen == 1_000_000
    assert len(workloads.plan(w, 8, 0).batches) == 1 and len(workloads.plan(w, 1, 0).batches) == 8


def test_bisulfite_plan_deals_reads_to_both_directions():
    w = workloads.WORKLOADS[4]
    for world in (1, 4):
        tot = {0: 0, 1: 0}
        for r in range(world):
            p = workloads.plan(w, world, r)
            for b in p.batches:
                tot[b.direction.slot] += b.n_queries
            assert sum(b.n_queries for b in p.batches) == p.n_queries
        assert tot[0] + tot[1] == 500_000 and abs(tot[0] - tot[1]) <= world
    # odd range start: the first query of a rank decides which direction gets the extra read
    p = workloads.plan(w, 3, 1, total_queries=10)
    assert (p.q_lo, p.q_hi) == (4, 7) and sorted((b.direction.slot, b.n_queries) for b in p.batches) == [(0, 2), (1, 1)]


def test_weak_scaling_gives_every_rank_its_own_queries():
    w = workloads.WORKLOADS[1]
    a, b = workloads.plan(w, 2, 0), workloads.plan(w, 2, 1)
    assert a.scaling == "weak" and a.n_queries == b.n_queries == 100_000 and a.job_queries == 200_000
    assert a.batches[0].seed != b.batches[0].seed


def test_bisulfite_conversion_of_synthetic_reads():
    q, s, ext = synth.make_batch_np(50, 100, 4, seed=3, alphabet=workloads.alphabet_array(workloads.WORKLOADS[4]),
                                    homolog_frac=1.0, sub_rate=0.0, indel_rate=0.0, convert="CT", convert_rate=1.0)
    assert (q != 1).all()  # no C left in the reads
    b = synth.band_size(100)
    w0 = s[: synth.window_len(100)][b: b + 100]
    # the window still holds the unconverted read: where it has C the read has T, everything else is equal
    assert ((w0 == q[:100]) | ((w0 == 1) & (q[:100] == 3))).all() and (w0 == 1).any()


def test_pmc_traffic_matches_the_full_instantiation_and_reports_what_it_skipped(tmp_path, monkeypatch):
    """a profile of another shape (no "kernels" table) made the driver's line lose its traffic figure, and a
    substring match attached the 24 GB of `score_pair_kernel<8, 19, true>` (checkpoint writer) to `<8, 19>` (writes the scores)."""
    import bench

    for spelling in ("score_pair_kernel<8, 19, true>", "lx::score_pair_kernel<8,19,true> (single sweep)",
                     "void lx::score_pair_kernel<8, 19, true>(lx::ScoreParams)"):
        t, note = bench.pmc_traffic(spelling)
        assert t is not None and 1e10 < t < 5e10, (spelling, note)   # 24.4 GB per headline launch
        assert "score_pair_kernel<8, 19, true>" in note
    for other in ("score_pair_kernel<8, 19>", "lx::score_pair_kernel<8,19>", "score_pair_kernel", "score_pair_kernel<8, 19, false>"):
        t, note = bench.pmc_traffic(other)
        assert t is None and "no committed PMC profile" in note, (other, t)
    # files that are not a per-kernel table, or not JSON at all, neither hide the good profile nor vanish silently
    good = json.loads((ROOT / "profiles" / "r03_bench_pmc.json").read_text())
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "r03_bench_pmc.json").write_text(json.dumps(good))
    (tmp_path / "profiles" / "r09_other_kernel_pmc.json").write_text(json.dumps({"kernel": "x", "SQ_WAVES": 1}))
    (tmp_path / "profiles" / "r10_broken_pmc.json").write_text("{not json")
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    t, note = bench.pmc_traffic("score_pair_kernel<8, 19, true>")
    assert t is not None and "r10_broken_pmc.json" in note and "r03_bench_pmc.json" in note
    assert bench.kernel_instantiation("lx::select_scan_kernel(lx::SelectParams, unsigned long)") == "select_scan_kernel"


def test_newest_pmc_profiles_were_made_from_the_kernels_in_the_tree():
    """bench.py reads `roofline.traffic` from the committed PMC passes, so a kernel that changed after its newest
    profile would leave a stale ratio in a driver-stamped line.  Every profile of the newest round that records its kernel sources
    (tools/collect_profiles.py, from round 5 on) must have been made from the sources in the tree: re-run tools/profile_round.sh +
    tools/collect_profiles.py after touching lx_score_f16.hip / lx_sweep_mq.hip / lx_ckpt.hip."""
    sys.path.insert(0, str(ROOT))
    import bench

    profs = sorted((ROOT / "profiles").glob("*_pmc.json"), reverse=True)
    recorded = []
    for f in profs:
        try:
            doc = json.loads(f.read_text())
        except Exception:
            continue
        if isinstance(doc, dict) and isinstance(doc.get("kernel_sources"), dict):
            recorded.append((f, doc))
    if not recorded:
        pytest.skip("no committed PMC profile records its kernel sources yet")
    newest_round = recorded[0][0].name.split("_")[0]
    for f, doc in recorded:
        if f.name.split("_")[0] != newest_round:
            continue
        stale = bench.stale_profile_sources(doc)
        assert not stale, f"{f.name} was profiled on other kernel sources than the tree's: {stale}"
        assert "STALE" not in bench.profile_sources_note(doc)
    # ... and the note names a change
    doc = json.loads(recorded[0][0].read_text())
    k = next(iter(doc["kernel_sources"]))
    doc["kernel_sources"][k] = "0" * 40
    assert "STALE" in bench.profile_sources_note(doc) and k in bench.stale_profile_sources(doc)

