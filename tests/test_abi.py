"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol include/lambda_ext.h
declares, refuses to create a handle without a device (no CPU fallback), and its host-side mirrors of the
reference's helpers agree with the oracle."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from lambda_amd import capi
from tests import oracle_lib

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    txt = (ROOT / "include" / "lambda_ext.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lx_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(lx_lib):
    decl = declared_symbols()
    assert len(decl) >= 20
    missing = [s for s in decl if not hasattr(lx_lib, s)]
    assert not missing, missing
    assert sorted(capi.EXPORTED_SYMBOLS) == decl
    assert lx_lib.lx_abi_version() == 3


def test_library_matches_the_source_tree(lx_lib):
    """The binary reports the hash of the sources it was compiled from; a stale .so (edited tree, old build) cannot pass."""
    from lambda_amd import build

    assert lx_lib.lx_build_id().decode() == build.source_id() == build.library_id()


def test_struct_layouts_match_header():
    assert C.sizeof(capi.Scoring) == 16 + 1024
    assert capi.EXT_DTYPE.itemsize == 24 and capi.HSP_DTYPE.itemsize == 48
    assert capi.MATCH_DTYPE.itemsize == 48 and capi.SEED_DTYPE.itemsize == 40
    assert C.sizeof(capi.SearchParams) == 96 and capi.BLAST_MATCH_DTYPE.itemsize == 128


def test_no_device_means_loud_failure(lx_lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert lx_lib.lx_device_count() == 0
    with pytest.raises(capi.LambdaExtError) as ei:
        capi.Handle(0)
    assert "no CPU fallback" in str(ei.value) or "HIP" in str(ei.value)


def test_widen_and_preprocess_matches_oracle(oracle):
    rng = np.random.default_rng(42)
    nq, ns = 30, 40
    qlens = rng.integers(20, 400, nq).astype(np.uint64)
    slens = rng.integers(50, 3000, ns).astype(np.uint64)
    n = 3000
    m = np.zeros(n, dtype=capi.MATCH_DTYPE)
    m["qryId"] = rng.integers(0, nq, n)
    m["subjId"] = rng.integers(0, 6, n)  # few subjects -> many overlapping windows
    seedlen = rng.integers(8, 15, n)
    m["qryStart"] = (rng.random(n) * np.maximum(qlens[m["qryId"]].astype(np.int64) - seedlen, 1)).astype(np.uint64)
    m["qryEnd"] = np.minimum(m["qryStart"] + seedlen.astype(np.uint64), qlens[m["qryId"]])
    m["subjStart"] = (rng.random(n) * np.maximum(slens[m["subjId"]].astype(np.int64) - seedlen, 1)).astype(np.uint64)
    m["subjEnd"] = np.minimum(m["subjStart"] + seedlen.astype(np.uint64), slens[m["subjId"]])
    want = oracle.widen_and_preprocess(m.astype(oracle_lib.MATCH_DTYPE), qlens, slens)
    got = capi.widen_and_preprocess(m, qlens, slens)
    assert len(got) == len(want) and len(got) < n
    assert (got.view(np.uint64) == want.view(np.uint64)).all()


def test_blast_statistics_match_oracle(oracle):
    for args, (lam, K) in (((62, 2, -3, -11, -1), (0.267, 0.041)), ((0, 2, -3, -5, -2), (0.625, 0.41))):
        ka = capi.karlin_params(*args)
        assert (ka.lambda_, ka.K) == (lam, K)
        oka = oracle_lib.Karlin(ka.lambda_, ka.K, ka.H, ka.alpha, ka.beta)
        lib = capi.load()
        for db in (10_000, 3_000_000, 205_000_000, 200_000_000_000):
            for ql in (10, 50, 100, 150, 200, 1000):
                a = lib.lx_length_adjustment(db, ql, C.byref(ka))
                assert a == oracle.length_adjustment(db, ql, oka)
                for sc in (1, 30, 77, 200, 1500):
                    assert lib.lx_evalue(sc, ql - a if ql > a else 0, db - a, C.byref(ka)) == oracle.evalue(sc, ql - a if ql > a else 0, db - a, oka)
                    assert lib.lx_bitscore(sc, C.byref(ka)) == oracle.bitscore(sc, oka)
    with pytest.raises(capi.LambdaExtError):
        capi.karlin_params(62, gap_open=-3, gap_extend=-3)  # no published values -> prepareScoring would throw


def test_widen_and_preprocess_large_lists_in_any_order(oracle):
    """Lists beyond the size where the driver's loops go to the host threads: grouped by query (one piece per thread, cut at
    query boundaries) and in RANDOM order (what the GPU seeding stage hands over: dealt to buckets of consecutive queries first) give
    what one thread gives -- the oracle's list."""
    rng = np.random.default_rng(11)
    n = 120_000
    m = np.zeros(n, dtype=capi.MATCH_DTYPE)
    m["qryId"] = np.sort(rng.integers(0, 9000, n))
    m["subjId"] = rng.integers(0, 40, n)
    qlens = rng.integers(30, 200, 9000).astype(np.uint64)
    slens = rng.integers(300, 2000, 40).astype(np.uint64)
    m["qryStart"] = rng.integers(0, 20, n)
    m["qryEnd"] = m["qryStart"] + 10
    m["subjStart"] = (rng.integers(0, 1500, n) // 7) * 7  # duplicates and overlaps abound
    m["subjEnd"] = m["subjStart"] + 10
    want = oracle.widen_and_preprocess(m.astype(oracle_lib.MATCH_DTYPE), qlens, slens)
    for order in (np.arange(n), rng.permutation(n), np.arange(n)[::-1]):
        got = capi.widen_and_preprocess(m[order], qlens, slens)
        assert len(got) == len(want) and len(got) < n
        for f in ("qryId", "subjId", "qryStart", "qryEnd", "subjStart", "subjEnd"):
            assert (got[f] == want[f]).all(), f


def test_rank_bridge():
    """lx_convert_ranks = seqan2_to_rank_inner (src/seqan2_to_biocpp.hpp:352-395): aa27 moves X behind Z, bisulfite
    dna5 moves N behind T, match/mismatch alphabets pass through; out-of-alphabet ranks are an error."""
    import numpy as np
    import pytest

    aa_bio = "ABCDEFGHIJKLMNOPQRSTUVWXYZ*"   # BioC++ aa27 rank order
    aa_seqan = "ABCDEFGHIJKLMNOPQRSTUVWYZX*"  # SeqAn AminoAcid rank order (the order of the scoring tables)
    got = capi.convert_ranks(capi.LX_RANKS_AA27, np.arange(27, dtype=np.uint8))
    assert [aa_seqan[r] for r in got] == list(aa_bio)
    dna_bio, dna_seqan = "ACGNT", "ACGTN"
    got = capi.convert_ranks(capi.LX_RANKS_DNA5_BS, np.arange(5, dtype=np.uint8))
    assert [dna_seqan[r] for r in got] == list(dna_bio)
    x = np.array([4, 3, 2, 1, 0, 3], dtype=np.uint8)
    assert (capi.convert_ranks(capi.LX_RANKS_SIMPLE, x) == x).all()
    # the BLOSUM62 table is indexed by SeqAn ranks: W/W = 11, Y/Y = 7, X/X = -1 after the bridge
    m = capi.builtin_scoring(62).matrix_np()
    w, y, xx = capi.convert_ranks(capi.LX_RANKS_AA27, np.array([aa_bio.index(c) for c in "WYX"], dtype=np.uint8))
    assert (m[w, w], m[y, y], m[xx, xx]) == (11, 7, -1)
    with pytest.raises(capi.LambdaExtError):
        capi.convert_ranks(capi.LX_RANKS_AA27, np.array([27], dtype=np.uint8))
    with pytest.raises(capi.LambdaExtError):
        capi.convert_ranks(7, np.array([0], dtype=np.uint8))
