"""The ONE selection function of the fused step (plan_step, lambda_amd/csrc/lx_api.cpp) walked on the CPU through lx_plan_step:
query widths 1 ... 1 300 x query runs {0, 2, 4, 7, 8, 16, 24, 32} x the four scoring schemes of the reference's programs.  Every
choice must be the documented one for its (width, run, alphabet) and must satisfy its own preconditions: the panels hold the
query, the wavefront's LDS fits, compact codes only with gap costs they can hold, an int32 fix-up launch exactly where the
a-priori bound of the widest admitted query exceeds the codes' 2046 (or the sweep tests the codes after the fact).  A wrong
condition in the table sends a shape to a kernel half as fast without any parity test noticing -- this one does."""
import math

import pytest

from lambda_amd import capi
from tests.test_oracle import SCHEMES

RUNS = (0, 2, 4, 7, 8, 16, 24, 32)


def _expected(name, q, run, nrows):
    small = nrows <= 8  # nucleotide / bisulfite alphabets: two packed-half profiles cost no occupancy
    if run == 2:  # the free packing: pairs of one query, up to four queries per wavefront -- only the multi-query sweep serves it
        return capi.PLAN_MQ, 4
    if run == 0 or run % 4 != 0:
        return capi.PLAN_NO_SWEEP, None
    if run == 4:
        return capi.PLAN_MQ, 4
    if run % 16 != 0:  # 8, 24
        if not small or q > 208:
            return capi.PLAN_MQ, 2
        return capi.PLAN_HALF, (2 if q <= 152 else 1)
    if q <= 208:
        return capi.PLAN_HALF, 1
    return capi.PLAN_I16_COMPACT_WIDE, 1


@pytest.mark.parametrize("name", ["blosum62", "nucl", "bs_fwd", "bs_rev"])
def test_plan_table(name):
    sc = SCHEMES[name]
    nrows = (sc.alphabet_size + 1 + 3) // 4 * 4
    seen = set()
    for run in RUNS:
        for q in range(1, 1301):
            ls = q + 2 * (int(math.sqrt(q)) + 1)
            p = capi.plan_step(sc, q, ls, run, 1000)
            fam, qpw = _expected(name, q, run, nrows)
            assert p.family == fam, (name, q, run, p.family, p.name)
            if fam == capi.PLAN_NO_SWEEP:
                continue
            seen.add((p.family, p.group_lanes, p.strip_cols))
            assert p.queries_per_wavefront == qpw, (name, q, run, p.queries_per_wavefront, p.name)
            # preconditions of the choice
            assert p.group_lanes * p.strip_cols * p.panels >= q and p.group_lanes * p.strip_cols * (p.panels - 1) < q, (q, run, p.name)
            assert p.lds_bytes <= 64 * 1024, (q, run, p.lds_bytes)
            if fam == capi.PLAN_MQ:
                assert p.group_lanes == 8 and p.strip_cols in (11, 13, 19) and p.compact_codes == 1
                assert p.lds_bytes <= 20480  # four protein profiles + staging at two wavefronts per SIMD (eight per CU)
            if fam == capi.PLAN_HALF:
                assert p.panels == 1 and p.compact_codes == 1
                want_geo = (8, 13) if q <= 104 else (8, 19) if q <= 152 else (8, 25) if (q <= 200 and run % 16 == 0) else (16, 13)
                assert (p.group_lanes, p.strip_cols) == want_geo, (q, run, p.name)
            if fam == capi.PLAN_I16_COMPACT_WIDE:
                assert (p.group_lanes, p.strip_cols) == (8, 19) and p.panels == -(-q // 152) and p.compact_codes == 1
            if p.compact_codes:
                assert -sc.gap_open <= 31
            # a fix-up launch exactly where the widest admitted query can exceed what the codes hold, or the panels are several
            assert bool(p.may_decline) == (p.score_bound > 2046 or p.panels > 1), (q, run, p.score_bound, p.panels, p.name)
            assert p.slot_bytes > 0 and p.name.decode().startswith("lx::")
    assert len(seen) >= 7


def test_plan_follows_budget_survivors_and_switches():
    sc = SCHEMES["blosum62"]
    # a batch whose checkpoints do not fit the budget: no sweep
    assert capi.plan_step(sc, 150, 176, 32, 3_200_000, trace_bytes=1 << 30).family == capi.PLAN_NO_SWEEP
    assert capi.plan_step(sc, 150, 176, 32, 3_200_000).family == capi.PLAN_HALF
    # few survivors last time: the one-query-per-wavefront sweep gives way to mode 1, the multi-query plan keeps its sweep
    p = capi.plan_step(sc, 150, 176, 32, 3_200_000, survivor_share=0.01)
    assert p.family == capi.PLAN_NO_SWEEP and p.adapted == 1
    assert capi.plan_step(sc, 150, 176, 32, 3_200_000, survivor_share=0.01, adapt_permille=0).family == capi.PLAN_HALF
    assert capi.plan_step(sc, 150, 176, 32, 3_200_000, survivor_share=0.2).family == capi.PLAN_HALF
    assert capi.plan_step(sc, 150, 176, 4, 100_000, survivor_share=0.01).family == capi.PLAN_MQ
    # the options: LX_OPT_MQ_SWEEP = 0 / 2, LX_OPT_PACKED_HALF = 0, LX_OPT_PASS2_MODE
    assert capi.plan_step(sc, 150, 176, 8, 1000, mq_sweep=0).family == capi.PLAN_HALF  # two profiles per wavefront at reduced occupancy
    assert capi.plan_step(sc, 150, 176, 2, 1000, mq_sweep=0).family == capi.PLAN_NO_SWEEP  # pairs have no other sweep
    assert b"free packing" in capi.plan_step(sc, 150, 176, 2, 1000).name
    assert capi.plan_step(sc, 150, 176, 32, 1000, mq_sweep=2).family == capi.PLAN_MQ
    assert capi.plan_step(sc, 150, 176, 32, 1000, packed_half=0).family == capi.PLAN_INT32
    assert capi.plan_step(sc, 150, 176, 32, 1000, pass2_mode=1).family == capi.PLAN_NO_SWEEP
    # gap costs beyond the compact codes (first character dearer than 31): int16-pair slots
    wide_gap = capi.builtin_scoring(62, gap_open=-35, gap_extend=-1)
    p = capi.plan_step(wide_gap, 150, 176, 32, 1000)
    assert p.family in (capi.PLAN_I16_PAIRS, capi.PLAN_INT32) and p.compact_codes == 0
    assert capi.plan_step(wide_gap, 150, 176, 4, 1000).family == capi.PLAN_NO_SWEEP


def test_solo_packing_only_where_sixteen_profiles_fit():
    """LX_OPT_QUERY_RUN = 1 -- no promise, a byte profile per window (lx_sweep_mq.hip's solo packing) -- is served where 16 profiles
    fit a wavefront's share of the LDS: nucleotide and bisulfite schemes (5 letters + pad), one to several panels; 16 protein
    profiles (27 letters) do not fit, and there 1 is what 0 is: no promise, no sweep."""
    prot = capi.builtin_scoring(62, gap_open=-11, gap_extend=-1)
    for lq in (100, 150, 300, 900):
        pl = capi.plan_step(prot, lq, lq + 30, 1, 10_000)
        assert "sweep_mq_kernel" not in pl.name.decode()
    for method in (0, -1, -2):
        sc = capi.builtin_scoring(method, match=2, mismatch=-3, gap_open=-5, gap_extend=-2)
        for lq, slen in ((100, 130), (150, 176), (150, 2000), (400, 450), (1000, 1100)):
            pl = capi.plan_step(sc, lq, slen, 1, 10_000)
            name = pl.name.decode()
            assert "solo packing: up to 16 queries per wavefront" in name, (method, lq, name)
            assert pl.queries_per_wavefront == 16 and pl.lds_bytes <= 20 * 1024 and pl.compact_codes == 1
            assert pl.panels * pl.group_lanes * pl.strip_cols >= lq
            # a fix-up launch exactly where the a-priori bound exceeds the codes or the panels are several
            assert bool(pl.may_decline) == (pl.score_bound > 2046 or pl.panels > 1), (lq, slen, pl.score_bound)
            assert ("int32 fix-up" in name) == bool(pl.may_decline)
