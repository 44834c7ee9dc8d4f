"""Synthetic seed-extension workloads (SURVEY.md section 8d / BASELINE.md section 3).

No index is built: candidate windows are generated directly, with the geometry the reference's window builder
produces (whole query x subject window of Lq + 2*b residues, b = floor(sqrt(Lq)) + 1;
/root/reference/src/search_misc.hpp:46-50, src/search_algo.hpp:919-938).  Half of the windows hold a mutated
copy of the query (substitutions + indels) between random flanks, half are random.

Two generators with the same structure: numpy (deterministic fixtures, CPU tests) and torch (full-size batches
generated directly in HBM for bench.py).
"""
from __future__ import annotations

import math

import numpy as np

from .capi import EXT_DTYPE

# SeqAn AminoAcid ranks ("ABCDEFGHIJKLMNOPQRSTUVWYZX*") of the 20 standard residues ACDEFGHIKLMNPQRSTVWY
STD20 = np.array([0, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 15, 16, 17, 18, 19, 21, 22, 23], dtype=np.uint8)


# bisulfite conversions in SeqAn Dna5 ranks (A, C, G, T, N): read letter that replaces the genome's
CONVERSIONS = {"CT": (1, 3), "GA": (2, 0)}


def band_size(lq: int) -> int:
    return int(math.sqrt(lq)) + 1


def window_len(lq: int) -> int:
    return lq + 2 * band_size(lq)


def make_batch_np(n_queries: int, lq: int, windows_per_query: int, seed: int, alphabet: np.ndarray = STD20,
                  homolog_frac: float = 0.5, sub_rate: float = 0.25, indel_rate: float = 0.02, n_rate: float = 0.0,
                  n_rank: int = 4, convert: str | None = None, convert_rate: float = 0.99):
    """Returns (q_res, s_res, ext): uint8 residue buffers and the EXT_DTYPE extension list.

    Extensions are ordered query-major (all windows of query 0, then query 1, ...), i.e. runs of
    `windows_per_query` consecutive entries share one query slice.

    convert = "CT" / "GA": bisulfite reads (SeqAn Dna5 ranks A, C, G, T = 0..3) -- after the windows have been copied
    from the unconverted read, `convert_rate` of the read's C become T (forward strand) / of its G become A (reverse).
    """
    rng = np.random.default_rng(seed)
    b = band_size(lq)
    ls = lq + 2 * b
    na = len(alphabet)
    q_idx = rng.integers(0, na, size=(n_queries, lq), dtype=np.int64)
    n_ext = n_queries * windows_per_query
    w_idx = rng.integers(0, na, size=(n_ext, ls), dtype=np.int64)

    homolog = rng.random(n_ext) < homolog_frac
    nh = int(homolog.sum())
    if nh:
        hq = np.repeat(np.arange(n_queries), windows_per_query)[homolog]
        ev = rng.random((nh, ls))
        dele = ev < indel_rate / 2           # skip one query residue
        ins = (ev >= indel_rate / 2) & (ev < indel_rate)  # insert a random residue
        shift = np.cumsum(dele, axis=1) - np.cumsum(ins, axis=1)
        src = np.arange(ls)[None, :] - b + shift
        inside = (src >= 0) & (src < lq) & ~ins
        copied = q_idx[hq[:, None], np.clip(src, 0, lq - 1)]
        sub = rng.random((nh, ls)) < sub_rate
        keep = inside & ~sub
        rows = w_idx[homolog]
        rows[keep] = copied[keep]
        w_idx[homolog] = rows

    q_res = alphabet[q_idx].astype(np.uint8)
    s_res = alphabet[w_idx].astype(np.uint8)
    if convert:
        frm, to = CONVERSIONS[convert]
        q_res[(q_res == frm) & (rng.random(q_res.shape) < convert_rate)] = to
    if n_rate > 0:
        q_res[rng.random(q_res.shape) < n_rate] = n_rank
        s_res[rng.random(s_res.shape) < n_rate] = n_rank

    ext = np.zeros(n_ext, dtype=EXT_DTYPE)
    ext["q_off"] = np.repeat(np.arange(n_queries, dtype=np.uint64) * lq, windows_per_query)
    ext["q_len"] = lq
    ext["s_off"] = np.arange(n_ext, dtype=np.uint64) * ls
    ext["s_len"] = ls
    return q_res.reshape(-1), s_res.reshape(-1), ext


def make_ragged_np(n_ext: int, seed: int, alphabet: np.ndarray = STD20, lq_range=(1, 400), ls_extra=(0, 80),
                   homolog_frac: float = 0.6, sub_rate: float = 0.2):
    """Ragged batch for edge-case parity tests: every extension has its own query and window length."""
    rng = np.random.default_rng(seed)
    na = len(alphabet)
    q_parts, s_parts = [], []
    ext = np.zeros(n_ext, dtype=EXT_DTYPE)
    qo = so = 0
    for i in range(n_ext):
        lq = int(rng.integers(lq_range[0], lq_range[1] + 1))
        ls = max(1, lq + int(rng.integers(ls_extra[0] - lq // 2, ls_extra[1] + 1)))
        q = rng.integers(0, na, size=lq)
        s = rng.integers(0, na, size=ls)
        if rng.random() < homolog_frac and lq > 4:
            a = int(rng.integers(0, max(1, ls - lq // 2)))
            seg = q[: min(lq, ls - a)].copy()
            mut = rng.random(len(seg)) < sub_rate
            seg[mut] = rng.integers(0, na, size=int(mut.sum()))
            # one deletion and one insertion now and then
            if len(seg) > 10 and rng.random() < 0.5:
                cut = int(rng.integers(2, len(seg) - 2))
                seg = np.concatenate([seg[:cut], seg[cut + int(rng.integers(1, 4)):]])
            s[a:a + len(seg)] = seg[: ls - a]
        q_parts.append(alphabet[q])
        s_parts.append(alphabet[s])
        ext[i] = (qo, so, lq, ls)
        qo += lq
        so += ls
    return np.concatenate(q_parts).astype(np.uint8), np.concatenate(s_parts).astype(np.uint8), ext


def make_batch_torch(n_queries: int, lq: int, windows_per_query: int, seed: int, device, alphabet: np.ndarray = STD20,
                     homolog_frac: float = 0.5, sub_rate: float = 0.25, indel_rate: float = 0.02,
                     n_rate: float = 0.0, n_rank: int = 4, chunk_queries: int = 8192, convert: str | None = None,
                     convert_rate: float = 0.99):
    """Same workload as make_batch_np, generated directly on `device` (torch).  Returns torch tensors
    (q_res u8 [n_queries*lq], s_res u8 [n_ext*ls], ext u8 view of EXT_DTYPE records [n_ext*24])."""
    import torch

    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    b = band_size(lq)
    ls = lq + 2 * b
    alpha = torch.as_tensor(np.asarray(alphabet, dtype=np.int64), device=device)
    na = len(alphabet)
    n_ext = n_queries * windows_per_query
    q_res = torch.empty((n_queries, lq), dtype=torch.uint8, device=device)
    s_res = torch.empty((n_ext, ls), dtype=torch.uint8, device=device)
    pos = torch.arange(ls, device=device)[None, :]
    for q0 in range(0, n_queries, chunk_queries):
        q1 = min(n_queries, q0 + chunk_queries)
        nq = q1 - q0
        q_idx = torch.randint(0, na, (nq, lq), generator=gen, device=device)
        ne = nq * windows_per_query
        w_idx = torch.randint(0, na, (ne, ls), generator=gen, device=device)
        homolog = torch.rand(ne, generator=gen, device=device) < homolog_frac
        hq = torch.arange(nq, device=device).repeat_interleave(windows_per_query)
        ev = torch.rand((ne, ls), generator=gen, device=device)
        dele = ev < indel_rate / 2
        ins = (ev >= indel_rate / 2) & (ev < indel_rate)
        shift = torch.cumsum(dele.to(torch.int32), 1) - torch.cumsum(ins.to(torch.int32), 1)
        src = pos - b + shift
        inside = (src >= 0) & (src < lq) & ~ins
        copied = torch.gather(q_idx[hq], 1, src.clamp(0, lq - 1))
        sub = torch.rand((ne, ls), generator=gen, device=device) < sub_rate
        keep = inside & ~sub & homolog[:, None]
        w_idx = torch.where(keep, copied, w_idx)
        qr = alpha[q_idx].to(torch.uint8)
        sr = alpha[w_idx].to(torch.uint8)
        if convert:
            frm, to = CONVERSIONS[convert]
            qr[(qr == frm) & (torch.rand(qr.shape, generator=gen, device=device) < convert_rate)] = to
        if n_rate > 0:
            qr[torch.rand(qr.shape, generator=gen, device=device) < n_rate] = n_rank
            sr[torch.rand(sr.shape, generator=gen, device=device) < n_rate] = n_rank
        q_res[q0:q1] = qr
        s_res[q0 * windows_per_query:q1 * windows_per_query] = sr
    ext = np.zeros(n_ext, dtype=EXT_DTYPE)
    ext["q_off"] = np.repeat(np.arange(n_queries, dtype=np.uint64) * lq, windows_per_query)
    ext["q_len"] = lq
    ext["s_off"] = np.arange(n_ext, dtype=np.uint64) * ls
    ext["s_len"] = ls
    d_ext = torch.from_numpy(ext.view(np.uint8).copy()).to(device)
    return q_res.reshape(-1), s_res.reshape(-1), d_ext, ext


def make_ragged_lists_np(n_queries: int, seed: int, alphabet: np.ndarray = STD20, lq_range=(50, 400), mean_windows: float = 12.0,
                         merged_frac: float = 0.10, homolog_frac: float = 0.5, sub_rate: float = 0.25, indel_rate: float = 0.02):
    """A seed list as lambda really hands it over (after _widenAndPreprocessMatches, src/search_algo.hpp:1136-1175): queries
    of mixed lengths, a geometric number of windows per query (mean `mean_windows`), most windows Lq + 2b long, a tenth of
    them merged ones of up to 3 Lq; grouped by query.  Returns (q_res, s_res, ext)."""
    rng = np.random.default_rng(seed)
    na = len(alphabet)
    lqs = rng.integers(lq_range[0], lq_range[1] + 1, n_queries)
    nwin = rng.geometric(1.0 / mean_windows, n_queries)
    q_off = np.concatenate([[0], np.cumsum(lqs)[:-1]])
    q_idx = rng.integers(0, na, int(lqs.sum()), dtype=np.uint8)
    n_ext = int(nwin.sum())
    ext = np.zeros(n_ext, dtype=EXT_DTYPE)
    qid = np.repeat(np.arange(n_queries), nwin)
    lq_e = lqs[qid]
    b_e = (np.sqrt(lq_e).astype(np.int64) + 1)
    ls_e = lq_e + 2 * b_e
    merged = rng.random(n_ext) < merged_frac
    ls_e = np.where(merged, rng.integers(ls_e, np.maximum(3 * lq_e, ls_e) + 1), ls_e)  # (tiny queries: Lq + 2b exceeds 3 Lq)
    s_off = np.concatenate([[0], np.cumsum(ls_e)[:-1]])
    s_idx = rng.integers(0, na, int(ls_e.sum()), dtype=np.uint8)
    homolog = np.nonzero(rng.random(n_ext) < homolog_frac)[0]
    if len(homolog):
        # vectorised: the (substituted) query copied into the window at offset a -- b for a plain window, anywhere for a merged
        # one -- with up to two residues of the query skipped (a gap in the alignment each)
        lq_h, ls_h, b_h = lq_e[homolog], ls_e[homolog], b_e[homolog]
        a_h = np.where(merged[homolog], (rng.random(len(homolog)) * np.maximum(1, ls_h - lq_h - b_h + 1)).astype(np.int64), b_h)
        ndel = rng.binomial(2, min(1.0, indel_rate * float(lqs.mean()) / 2.0), len(homolog))
        p1 = (rng.random(len(homolog)) * lq_h).astype(np.int64)
        p2 = (rng.random(len(homolog)) * lq_h).astype(np.int64)
        len_h = np.minimum(lq_h - ndel, ls_h - a_h)
        tot = int(len_h.sum())
        first = np.concatenate([[0], np.cumsum(len_h)[:-1]])
        k = np.arange(tot) - np.repeat(first, len_h)                    # position inside the copied piece
        rep = lambda x: np.repeat(x, len_h)
        skip = (rep(ndel) >= 1) * (k >= rep(p1)) + (rep(ndel) >= 2) * (k >= rep(p2))
        src = rep(q_off[qid[homolog]]) + np.minimum(k + skip, rep(lq_h) - 1)
        val = q_idx[src]
        mut = rng.random(tot) < sub_rate
        val[mut] = rng.integers(0, na, int(mut.sum()), dtype=np.uint8)
        s_idx[rep(s_off[homolog] + a_h) + k] = val
    ext["q_off"], ext["q_len"], ext["s_off"], ext["s_len"] = q_off[qid], lq_e, s_off, ls_e
    return alphabet[q_idx].astype(np.uint8), alphabet[s_idx].astype(np.uint8), ext


def make_seed_list_np(n_reads: int, genome_mbp: float, seed: int, read_len: int = 150, seeds_per_read: int = 12, contigs: int = 20,
                      homolog_frac: float = 0.75, sub_rate: float = 0.02, spurious_per_read: float = 0.5):
    """A seed list of the kind `lambda3 searchn` hands to iterateMatches (/root/reference/src/search_algo.hpp:1364-1385) at BASELINE
    configs[2]'s size: reads of `read_len` bp cut from a random genome (both strands, `sub_rate` substitutions; the rest random),
    every read as two query frames (itself, its reverse complement: add_reverse_complement, src/shared_definitions.hpp:260), for
    every homologous read `seeds_per_read` seed hits along its diagonal on the frame that matches (a tenth of them one base off the
    diagonal, as seeds next to an indel are) + spurious hits elsewhere.  Ranks: A, C, G, T = 0, 1, 2, 4 (BioC++ dna5 without N).
    Returns q_res, q_off, q_len (frame-expanded, 2 n_reads), q_orig_len (n_reads), s_res, s_off, s_len, matches (MATCH_DTYPE, in the
    arbitrary order a seeding kernel's lanes emit them)."""
    from .capi import MATCH_DTYPE

    rng = np.random.default_rng(seed)
    nt = np.array([0, 1, 2, 4], dtype=np.uint8)
    comp = np.zeros(5, dtype=np.uint8)
    comp[[0, 1, 2, 4]] = [4, 2, 1, 0]
    clen = int(genome_mbp * 1e6 / contigs)
    s_len = np.full(contigs, clen, dtype=np.uint64)
    s_off = (np.arange(contigs, dtype=np.uint64) * np.uint64(clen))
    s_res = nt[rng.integers(0, 4, clen * contigs, dtype=np.uint8)]
    homolog = rng.random(n_reads) < homolog_frac
    ctg = rng.integers(0, contigs, n_reads)
    pos = rng.integers(0, clen - read_len, n_reads)
    minus = rng.random(n_reads) < 0.5
    idx = (s_off[ctg].astype(np.int64) + pos)[:, None] + np.arange(read_len)[None, :]
    fwd = s_res[idx]  # the genome's strand
    sub = rng.integers(0, 65536, fwd.shape, dtype=np.uint16) < np.uint16(sub_rate * 65536)
    fwd[sub] = nt[rng.integers(0, 4, int(sub.sum()))]
    fwd[~homolog] = nt[rng.integers(0, 4, (int((~homolog).sum()), read_len))]
    rc = comp[fwd[:, ::-1]]
    # frame 0 = the read, frame 1 = its reverse complement; a minus-strand read IS rc, so its frame 1 carries the genome's strand
    frames = np.empty((n_reads, 2, read_len), dtype=np.uint8)
    frames[:, 0] = np.where(minus[:, None], rc, fwd)
    frames[:, 1] = np.where(minus[:, None], fwd, rc)
    q_res = frames.reshape(-1)
    q_len = np.full(2 * n_reads, read_len, dtype=np.uint64)
    q_off = np.arange(2 * n_reads, dtype=np.uint64) * np.uint64(read_len)
    q_orig_len = np.full(n_reads, read_len, dtype=np.uint64)
    L = 10
    hr = np.nonzero(homolog)[0]
    k = seeds_per_read
    qs = rng.integers(0, read_len - L, (len(hr), k))
    off_diag = (rng.random((len(hr), k)) < 0.1) * rng.integers(-1, 2, (len(hr), k))
    m1 = np.zeros(len(hr) * k, dtype=MATCH_DTYPE)
    m1["qryId"] = np.repeat(2 * hr + minus[hr], k)
    m1["subjId"] = np.repeat(ctg[hr], k)
    m1["qryStart"] = qs.reshape(-1)
    m1["qryEnd"] = m1["qryStart"] + L
    m1["subjStart"] = np.clip(np.repeat(pos[hr], k) + qs.reshape(-1) + off_diag.reshape(-1), 0, clen - L)
    m1["subjEnd"] = m1["subjStart"] + L
    ns = int(n_reads * spurious_per_read)
    m2 = np.zeros(ns, dtype=MATCH_DTYPE)
    m2["qryId"] = rng.integers(0, 2 * n_reads, ns)
    m2["subjId"] = rng.integers(0, contigs, ns)
    m2["qryStart"] = rng.integers(0, read_len - L, ns)
    m2["qryEnd"] = m2["qryStart"] + L
    m2["subjStart"] = rng.integers(0, clen - L, ns)
    m2["subjEnd"] = m2["subjStart"] + L
    m = np.concatenate([m1, m2])
    # lanes own reads and finish in no particular order: blocks of a few thousand reads' matches, shuffled
    blk = np.argsort(rng.permutation(len(m) // 4096 + 1).repeat(4096)[:len(m)] * (1 << 32) + m["qryId"] // 128, kind="stable")
    return q_res, q_off, q_len, q_orig_len, s_res, s_off, s_len, m[blk]


def make_protein_seed_list_np(n_queries: int, seed: int, lq: int = 150, homologs: int = 16, spurious: int = 16, seeds_per_homolog: int = 4,
                              alphabet: np.ndarray = STD20, sub_rate: float = 0.13, indel_rate: float = 0.02):
    """A seed list of the kind `lambda3 searchp` hands to iterateMatches (/root/reference/src/search_algo.hpp:1364-1385) at BASELINE
    configs[1]'s size -- make_seed_list_np's protein twin: queries of `lq` residues in families of `homologs` members (an ancestor
    with `sub_rate` substitutions each); a subject set of as many proteins (lengths log-normal around 300), each
    holding one family member with substitutions AND indels (query against region: ~25 % substitutions, 2 % indels) between random
    flanks.  Per query: `seeds_per_homolog` seed hits on the diagonal of each of its family's `homologs` regions (shifted where an
    indel lies before them) and `spurious` single hits anywhere -- after _widenAndPreprocessMatches (:1136-1175) about homologs +
    spurious = 32 windows per query, half of them homologous: the list the headline batch stands for.  Returns q_res, q_off, q_len,
    q_orig_len, s_res, s_off, s_len, matches (MATCH_DTYPE, in the arbitrary order a seeding kernel's lanes emit them)."""
    from .capi import MATCH_DTYPE

    rng = np.random.default_rng(seed)
    na = len(alphabet)
    nf = (n_queries + homologs - 1) // homologs
    anc = rng.integers(0, na, (nf, lq), dtype=np.uint8)
    fam = np.arange(n_queries) // homologs
    # queries: substitutions only (they keep the length)
    q_idx = anc[fam]
    sub = rng.integers(0, 65536, q_idx.shape, dtype=np.uint16) < np.uint16(sub_rate * 65536)
    q_idx[sub] = rng.integers(0, na, int(sub.sum()), dtype=np.uint8)
    q_res = alphabet[q_idx].reshape(-1).astype(np.uint8)
    q_len = np.full(n_queries, lq, dtype=np.uint64)
    q_off = np.arange(n_queries, dtype=np.uint64) * np.uint64(lq)
    # subjects: one region each (region j belongs to family j // homologs), random elsewhere
    n_subj = nf * homologs
    lr = lq + 8  # region slots: room for the net shift of the indels
    s_len = np.clip(rng.lognormal(math.log(300.0), 0.6, n_subj), lr + 24, 2000).astype(np.uint64)
    s_off = np.concatenate([[0], np.cumsum(s_len)[:-1]]).astype(np.uint64)
    s_idx = rng.integers(0, na, int(s_len.sum()), dtype=np.uint8)
    pos = (rng.random(n_subj) * (s_len - lr).astype(np.float64)).astype(np.int64)  # where the region begins in its subject
    src_all = np.empty((n_subj, lr), dtype=np.int16)  # ancestor position copied to each region position (-1: none)
    for lo in range(0, n_subj, 1 << 16):
        hi = min(n_subj, lo + (1 << 16))
        ev = rng.random((hi - lo, lr))
        dele = ev < indel_rate / 2
        ins = (ev >= indel_rate / 2) & (ev < indel_rate)
        src = np.arange(lr)[None, :] + np.cumsum(dele, axis=1) - np.cumsum(ins, axis=1)
        inside = (src >= 0) & (src < lq) & ~ins
        copied = anc[(np.arange(lo, hi) // homologs)[:, None], np.clip(src, 0, lq - 1)]
        keep = inside & (rng.integers(0, 65536, src.shape, dtype=np.uint16) >= np.uint16(sub_rate * 65536))
        at = (s_off[lo:hi].astype(np.int64) + pos[lo:hi])[:, None] + np.arange(lr)[None, :]
        s_idx[at[keep]] = copied[keep]
        src_all[lo:hi] = np.where(inside, src, -1)
    s_res = alphabet[s_idx].astype(np.uint8)
    # seed hits: query q against region 16 fam(q) + h, at region positions whose ancestor position is known
    L = 10
    k = seeds_per_homolog
    reg = (fam[:, None] * homologs + np.arange(homologs)[None, :]).reshape(-1)  # [n_queries * homologs]
    qq = np.repeat(np.arange(n_queries), homologs)
    m1 = np.zeros(len(reg) * k, dtype=MATCH_DTYPE)
    rp = rng.integers(4, lr - L - 4, (len(reg), k))
    qs = src_all[reg[:, None], rp].astype(np.int64)
    ok = ((qs >= 0) & (qs <= lq - L)).reshape(-1)
    m1["qryId"] = np.repeat(qq, k)
    m1["subjId"] = np.repeat(reg, k)
    m1["qryStart"] = np.clip(qs.reshape(-1), 0, lq - L)
    m1["qryEnd"] = m1["qryStart"] + L
    m1["subjStart"] = np.repeat(pos[reg], k) + rp.reshape(-1)
    m1["subjEnd"] = m1["subjStart"] + L
    m1 = m1[ok]
    ns = n_queries * spurious
    m2 = np.zeros(ns, dtype=MATCH_DTYPE)
    m2["qryId"] = np.repeat(np.arange(n_queries), spurious)
    m2["subjId"] = rng.integers(0, n_subj, ns)
    m2["qryStart"] = rng.integers(0, lq - L, ns)
    m2["qryEnd"] = m2["qryStart"] + L
    m2["subjStart"] = (rng.random(ns) * (s_len[m2["subjId"]] - L).astype(np.float64)).astype(np.uint64)
    m2["subjEnd"] = m2["subjStart"] + L
    m = np.concatenate([m1, m2])
    # lanes own queries and finish in no particular order: blocks of a few thousand matches, shuffled
    blk = np.argsort(rng.permutation(len(m) // 4096 + 1).repeat(4096)[:len(m)] * (1 << 32) + m["qryId"] // 128, kind="stable")
    return q_res, q_off, q_len, q_len.copy(), s_res, s_off, s_len, m[blk]
