"""CPU tests of row N2: _writeRecord semantics (src/search_algo.hpp:820-913) and the BLAST-tabular / SAM writers
(src/search_output.hpp:463-733).  No GPU needed: the inputs are finished HSP records."""
import numpy as np
import pytest

from lambda_amd import capi


def rec(n_qid, n_sid, qs, qe, ss, se, bits, score=50, alen=None, nm=None, ops_off=0, n_ops=0, ev=1e-5, ident=80.0, frame=0):
    r = np.zeros(1, dtype=capi.BLAST_MATCH_DTYPE)[0]
    r["n_qid"], r["qry_id"], r["n_sid"], r["subj_id"] = n_qid, n_qid, n_sid, n_sid
    r["q_start"], r["q_end"], r["s_start"], r["s_end"] = qs, qe, ss, se
    r["bit_score"], r["score"], r["e_value"], r["identity"] = bits, score, ev, ident
    r["alignment_length"] = alen if alen is not None else qe - qs
    r["num_matches"] = nm if nm is not None else (qe - qs) * 8 // 10
    r["num_mismatches"] = r["alignment_length"] - r["num_matches"]
    r["ops_off"], r["n_ops"] = ops_off, n_ops
    r["q_frame"] = frame
    return r


def test_postprocess_sort_dedupe_topn():
    m = np.array([
        rec(0, 5, 0, 50, 10, 60, 40.0),
        rec(0, 2, 0, 50, 10, 60, 90.0),
        rec(0, 5, 0, 50, 10, 60, 70.0),   # duplicate coordinates of the first, better score -> this one is kept
        rec(0, 3, 5, 40, 100, 135, 90.0),  # ties with subject 2 on bit score: stable order after the coordinate sort
        rec(0, 9, 0, 10, 0, 10, 10.0),
        rec(1, 1, 0, 20, 0, 20, 30.0),
        rec(1, 1, 0, 20, 0, 20, 30.0),     # exact duplicate
    ], dtype=capi.BLAST_MATCH_DTYPE)
    out, st = capi.postprocess_records(m, max_matches=3)
    assert (st.qrys_with_hit, st.hits_duplicate2, st.hits_abundant, st.hits_final, st.pairs) == (2, 2, 1, 4, 4)
    q0 = out[out["n_qid"] == 0]
    assert list(q0["bit_score"]) == [90.0, 90.0, 70.0]
    assert list(q0["n_sid"]) == [2, 3, 5]  # equal bit scores keep the (n_sid, ...) order of the first sort
    q1 = out[out["n_qid"] == 1]
    assert len(q1) == 1


def test_blast_tab_writer(tmp_path):
    ops = b"M" * 30
    m = np.array([rec(0, 1, 4, 34, 99, 129, 61.23, alen=30, nm=27, n_ops=30, ev=3.2e-12, ident=90.0),
                  rec(1, 0, 0, 30, 0, 30, 55.5, alen=30, nm=30, n_ops=30, ev=1.0e-9, ident=100.0)],
                 dtype=capi.BLAST_MATCH_DTYPE)
    m["num_gap_opens"] = [0, 0]
    p = tmp_path / "out.m8"
    capi.write_records(p, capi.LX_OUT_BLAST_TAB, m, ops, ["q0 desc", "q1"], [40, 30], ["s0", "s1 something"], [300, 400])
    lines = p.read_text().splitlines()
    assert lines[0] == "q0\ts1\t90.00\t30\t3\t0\t5\t34\t100\t129\t3.2e-12\t61.2"
    assert lines[1] == "q1\ts0\t100.00\t30\t0\t0\t1\t30\t1\t30\t1.0e-09\t55.5"
    p9 = tmp_path / "out.m9"
    capi.write_records(p9, capi.LX_OUT_BLAST_TAB_COMMENTS, m, ops, ["q0 desc", "q1"], [40, 30], ["s0", "s1"], [300, 400])
    txt = p9.read_text()
    assert txt.count("# BLASTP 2.2.26+") == 2 and "# Query: q0 desc" in txt and "# 1 hits found" in txt


def test_sam_writer_blastn_cigar_and_tags(tmp_path):
    # 3 leading query bases unaligned, 10 M, 2 D (gap in the query row), 5 M, 1 I, 4 M, 5 trailing unaligned
    ops = b"M" * 10 + b"D" * 2 + b"M" * 5 + b"I" + b"M" * 4
    qlen = 3 + 10 + 5 + 1 + 4 + 5
    m = np.array([rec(0, 0, 3, 23, 50, 71, 40.9, alen=len(ops), nm=17, n_ops=len(ops), ev=2e-4, ident=77.3, frame=1),
                  rec(0, 1, 3, 23, 10, 31, 30.0, alen=len(ops), nm=15, n_ops=len(ops), ev=2e-2, ident=70.0, frame=1)],
                 dtype=capi.BLAST_MATCH_DTYPE)
    read = b"ACGTACGTACGTACGTACGTACGTACGT"[:qlen]
    p = tmp_path / "out.sam"
    soft = capi.output_options(sam_hard_clip=0)  # --sam-bam-clip soft
    capi.write_records(p, capi.LX_OUT_SAM, m, ops, ["read1 x"], [qlen], ["chr1", "chr2"], [1000, 1000], program="blastn",
                       q_ascii=read, q_ascii_off=[0], options=soft)
    lines = [l for l in p.read_text().splitlines() if not l.startswith("@")]
    f0, f1 = lines[0].split("\t"), lines[1].split("\t")
    assert f0[:9] == ["read1", "0", "chr1", "51", "255", "3S10M2D5M1I4M5S", "*", "0", "0"]
    assert f0[9] == read.decode() and f0[10] == "*"
    assert f0[11:] == ["ae:f:0.0002", "AS:i:40", "ai:i:77", "qf:i:1", f"NM:i:{len(ops) - 17}"]
    assert f1[1] == "256" and f1[2] == "chr2" and f1[9] == "*"  # secondary; same query region -> sequence not repeated
    assert p.read_text().startswith("@HD\tVN:1.4\tGO:query\n")
    # the reference's default is hard clips (src/search_options.hpp:360, :812): the clipped bases leave CIGAR and SEQ
    capi.write_records(p, capi.LX_OUT_SAM, m, ops, ["read1 x"], [qlen], ["chr1", "chr2"], [1000, 1000], program="blastn",
                       q_ascii=read, q_ascii_off=[0])
    h0, h1 = [l.split("\t") for l in p.read_text().splitlines() if not l.startswith("@")]
    assert h0[5] == "3H10M2D5M1I4M5H" and h0[9] == read[3:23].decode() and h1[9] == "*"


def test_sam_writer_blastp_has_no_cigar(tmp_path):
    m = np.array([rec(0, 0, 0, 20, 5, 25, 50.0, n_ops=20)], dtype=capi.BLAST_MATCH_DTYPE)
    p = tmp_path / "p.sam"
    capi.write_records(p, capi.LX_OUT_SAM, m, b"M" * 20, ["q"], [20], ["s"], [100])
    f = [l for l in p.read_text().splitlines() if not l.startswith("@")][0].split("\t")
    assert f[5] == "*" and f[9] == "*" and "qf:i:0" in f  # src/search_output.hpp:527-531: no DNA cigar for BLASTP


def test_sam_writer_translated_query_cigar_clips_and_sequence(tmp_path):
    """BLASTX / TBLASTX records in SAM (src/search_output.hpp:115-194, :84-109, :493-499): nucleotide-space CIGAR (runs x 3,
    the nucleotides outside the frame as hard clips, reversed on the minus strand), the covered part of the untranslated
    read as SEQ, translated subject positions x 3 + frame offset."""
    read = b"ACGTTGCAAGGCTTAACCGGTTAAGGCCTTAG"  # 32 nt
    qlen = len(read)
    ops = b"MMMMDMMM"  # protein space: 4 M, one gap in the query row, 3 M -> query columns [2, 9)
    # frame +2: one nucleotide in front of the frame, (32 - 1) % 3 = 1 behind it, frame length 10 codons
    plus = rec(0, 0, 2, 9, 5, 13, 40.0, alen=len(ops), nm=6, n_ops=len(ops), frame=2)
    # frame -3: two nucleotides clipped in front (of the reverse strand), none behind, frame length 10 codons
    minus = rec(0, 1, 2, 9, 5, 13, 39.0, alen=len(ops), nm=6, n_ops=len(ops), frame=-3)
    plus["s_frame"], minus["s_frame"] = 3, -1
    m = np.array([plus, minus], dtype=capi.BLAST_MATCH_DTYPE)
    p = tmp_path / "x.sam"
    soft = capi.output_options(sam_hard_clip=0)
    capi.write_records(p, capi.LX_OUT_SAM, m, ops, ["read1"], [qlen], ["s0", "s1"], [900, 900], program="tblastx",
                       q_ascii=read, q_ascii_off=[0], options=soft)
    f0, f1 = [l.split("\t") for l in p.read_text().splitlines() if not l.startswith("@")]
    assert f0[1] == "0" and f0[5] == "1H6S12M3D9M3S1H"
    assert f0[9] == read[1:31].decode()
    assert f0[3] == str(5 * 3 + 2 + 1)  # s_frame +3: 3 s_start + |frame| - 1, 1-based in the file
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    assert f1[1] == "272" and f1[5] == "3S9M3D12M6S2H"  # secondary + reverse strand; element list reversed
    assert f1[9] == "".join(comp[c] for c in reversed(read[0:30].decode()))
    assert f1[3] == str(qlen - (5 * 3 + 0) + 1)  # the reference's minus-strand branch, restated as it stands
    assert "qf:i:2" in f0 and "qf:i:-3" in f1
    # BLASTX: the subject is protein -- plain position
    capi.write_records(p, capi.LX_OUT_SAM, m[:1], ops, ["read1"], [qlen], ["s0", "s1"], [900, 900], program="blastx",
                       q_ascii=read, q_ascii_off=[0], options=soft)
    g = [l.split("\t") for l in p.read_text().splitlines() if not l.startswith("@")][0]
    assert g[3] == "6" and g[5] == "1H6S12M3D9M3S1H"
    # hard clips (the default): frame clip and local clip merge into one H, SEQ is the matched codons only (:84-109 with qStart, qEnd)
    capi.write_records(p, capi.LX_OUT_SAM, m, ops, ["read1"], [qlen], ["s0", "s1"], [900, 900], program="tblastx",
                       q_ascii=read, q_ascii_off=[0])
    h0, h1 = [l.split("\t") for l in p.read_text().splitlines() if not l.startswith("@")]
    assert h0[5] == "7H12M3D9M4H" and h0[9] == read[1 + 6: 1 + 27].decode()
    assert h1[5] == "3H9M3D12M8H" and h1[9] == "".join(comp[c] for c in reversed(read[qlen - (27 + 2): qlen - (6 + 2)].decode()))
    # TBLASTN: protein query -- no DNA cigar, no sequence, translated subject position
    t = rec(0, 0, 2, 9, 5, 13, 40.0, alen=len(ops), nm=6, n_ops=len(ops), frame=0)
    t["s_frame"] = 2
    capi.write_records(p, capi.LX_OUT_SAM, np.array([t], dtype=capi.BLAST_MATCH_DTYPE), ops, ["prot1"], [10], ["s0"], [900],
                       program="tblastn")
    g = [l.split("\t") for l in p.read_text().splitlines() if not l.startswith("@")][0]
    assert g[5] == "*" and g[9] == "*" and g[3] == str(5 * 3 + 1 + 1)


def test_sam_writer_blastn_minus_strand_cigar_is_reversed(tmp_path):
    """blastMatchOneCigar reverses the element list whenever qFrameShift < 0 (src/search_output.hpp:192-193) -- BLASTN hits
    of the reverse-complemented query frame included, not only translated queries.  Asymmetric alignment so that the
    reversal shows: 2 leading unaligned, 6 M, 1 I, 9 M, 3 D, 4 M, 7 trailing unaligned."""
    ops = b"M" * 6 + b"I" + b"M" * 9 + b"D" * 3 + b"M" * 4
    qlen = 2 + 6 + 1 + 9 + 4 + 7
    fwd = rec(0, 0, 2, 22, 40, 62, 41.0, alen=len(ops), nm=18, n_ops=len(ops), frame=1)
    rev = rec(0, 1, 2, 22, 40, 62, 40.0, alen=len(ops), nm=18, n_ops=len(ops), frame=-1)
    m = np.array([fwd, rev], dtype=capi.BLAST_MATCH_DTYPE)
    read = b"ACGGTCATTGCAAGCTTAGGCATCGATAC"[:qlen]
    p = tmp_path / "rc.sam"
    capi.write_records(p, capi.LX_OUT_SAM, m, ops, ["r"], [qlen], ["c1", "c2"], [500, 500], program="blastn",
                       q_ascii=read, q_ascii_off=[0], options=capi.output_options(sam_hard_clip=0))
    f0, f1 = [l.split("\t") for l in p.read_text().splitlines() if not l.startswith("@")]
    assert f0[1] == "0" and f0[5] == "2S6M1I9M3D4M7S"
    assert f1[1] == str(256 | 16) and f1[5] == "7S4M3D9M1I6M2S"  # the same elements, back to front
    assert "qf:i:-1" in f1


def _lca_reference(parents, heights, n1, n2):
    """computeLCA as the reference writes it (/root/reference/src/search_misc.hpp:86-112), restated for the test."""
    if n1 == n2:
        return n1
    i = heights[n1]
    while i > heights[n2]:
        n1 = parents[n1]
        i -= 1
    i = heights[n2]
    while i > heights[n1]:
        n2 = parents[n2]
        i -= 1
    while n1 != 0 and n2 != 0:
        if n1 == n2:
            return n1
        n1, n2 = parents[n1], parents[n2]
    raise RuntimeError("LCA-computation error")


def test_lca_of_a_query_follows_write_record():
    """The LCA step of _writeRecord (/root/reference/src/search_algo.hpp:884-907) on a random taxonomy: start from the first match
    whose subject's first taxon is assigned, fold in every assigned taxon of every match; unassigned taxa (parent 0) are ignored,
    a query without an assigned subject reports 0."""
    rng = np.random.default_rng(5)
    n_taxa = 400
    parents = np.zeros(n_taxa, dtype=np.uint32)
    heights = np.zeros(n_taxa, dtype=np.uint32)
    parents[1] = 1  # the root is its own parent in NCBI's dump; height 0
    for t in range(2, n_taxa):
        if rng.random() < 0.05:
            continue  # unassigned: parent 0
        par = int(rng.integers(1, t))
        while par != 1 and parents[par] == 0:
            par = int(rng.integers(1, t))
        parents[t] = par
        heights[t] = heights[par] + 1
    n_s = 300
    ntax = rng.choice([0, 1, 1, 1, 2, 3], n_s)
    s_tax_off = np.concatenate([[0], np.cumsum(ntax)]).astype(np.uint64)
    s_tax_ids = rng.integers(2, n_taxa, int(ntax.sum())).astype(np.uint32)
    bms = np.zeros(0, dtype=capi.BLAST_MATCH_DTYPE)
    groups = []
    for qid in range(60):
        k = int(rng.integers(1, 12))
        g = np.zeros(k, dtype=capi.BLAST_MATCH_DTYPE)
        g["n_qid"] = qid * 3
        g["n_sid"] = rng.integers(0, 20 if qid % 7 == 0 else n_s, k)
        groups.append(g)
    bms = np.concatenate(groups)
    qids, lcas = capi.compute_lca(bms, parents, heights, s_tax_off, s_tax_ids)
    assert list(qids) == [g["n_qid"][0] for g in groups]
    nonzero = 0
    for g, got in zip(groups, lcas):
        want = 0
        for sid in g["n_sid"]:
            taxa = s_tax_ids[int(s_tax_off[sid]): int(s_tax_off[sid + 1])]
            if len(taxa) and parents[taxa[0]] != 0:
                want = int(taxa[0])
                break
        if want:
            for sid in g["n_sid"]:
                for tax in s_tax_ids[int(s_tax_off[sid]): int(s_tax_off[sid + 1])]:
                    if parents[tax] != 0:
                        want = _lca_reference(parents, heights, int(tax), want)
        assert int(got) == want
        nonzero += want != 0
    assert nonzero > 40
    # two trees that never meet: the reference throws, the library reports the error
    parents2, heights2 = parents.copy(), heights.copy()
    parents2[1] = 0  # cut the root's self-loop: the climb now runs into 0 before the paths meet ...
    a, b = 2, 3
    parents2[a], parents2[b], heights2[a], heights2[b] = 4, 5, 1, 1
    parents2[4], parents2[5], heights2[4], heights2[5] = 0, 0, 0, 0  # ... for these two: roots of their own
    one = np.zeros(2, dtype=capi.BLAST_MATCH_DTYPE)
    one["n_sid"] = [0, 1]
    with pytest.raises(capi.LambdaExtError):
        capi.compute_lca(one, parents2, heights2, np.array([0, 1, 2], dtype=np.uint64), np.array([a, b], dtype=np.uint32))


def test_sam_tags_sequence_modes_and_reference_header(tmp_path):
    """--sam-bam-tags / --sam-bam-seq / --sam-with-refheader / --version-to-outputfile (src/search_options.hpp:276-379, :762-812): every
    optional tag of SamBamExtraTags (src/search_output.hpp:29-76) in the order myWriteRecord appends them (:601-719), with the widths it
    casts to; the header line lists the chosen tags in the order of the enum."""
    ops = b"M" * 10 + b"D" * 2 + b"M" * 8
    qlen = 4 + 18 + 3
    a = rec(0, 0, 4, 22, 50, 70, 300.7, score=300, alen=len(ops), nm=15, n_ops=len(ops), ev=2e-30, ident=75.0)
    b = rec(0, 1, 4, 22, 10, 30, 40.2, score=44, alen=len(ops), nm=12, n_ops=len(ops), ev=2e-3, ident=60.0)
    m = np.array([a, b], dtype=capi.BLAST_MATCH_DTYPE)
    m["num_positives"] = [17, 14]
    prot = b"MKVLAAGIVGLLLAQWERTYCDEFG"[:qlen]
    p = tmp_path / "t.sam"
    o = capi.output_options(sam_tags="lt ls st sf qs OC IH ar ap AS NM ae ai qf", sam_seq=capi.LX_SAM_SEQ_ALWAYS, sam_with_ref_header=1,
                            version_to_output=1, version="3.0.0", command_line="lambda3 searchp -q q.fa")
    capi.write_records(p, capi.LX_OUT_SAM, m, ops, ["q1 some description"], [qlen], ["s0", "s1 x"], [500, 600], program="blastp",
                       q_ascii=prot, q_ascii_off=[0], options=o)
    txt = p.read_text().splitlines()
    assert txt[0] == "@HD\tVN:1.4\tGO:query" and txt[1:3] == ["@SQ\tSN:s0\tLN:500", "@SQ\tSN:s1\tLN:600"]
    assert txt[3] == "@PG\tID:lambda\tPN:lambda\tVN:3.0.0\tCL:lambda3 searchp -q q.fa"
    co = [l for l in txt if l.startswith("@CO\tOptional tags as follow")][0].split("\t")[2:]
    assert [c.split(":")[0] for c in co] == ["AS", "OC", "NM", "IH", "ar", "ae", "ai", "ap", "qf", "qs", "sf", "st", "ls", "lt"]
    f0, f1 = [l.split("\t") for l in txt if not l.startswith("@")]
    # BLASTP: no DNA cigar, no SEQ; the protein cigar and sequence travel in OC / qs (hard clips: the matched part only)
    assert f0[5] == "*" and f0[9] == "*"
    assert f0[11:] == ["ae:f:2e-30", "AS:i:300", f"ar:i:{300 % 256}", "ai:i:75", f"ap:i:{int(100 * 17 / len(ops))}", "qf:i:0", "sf:i:0", "st:Z:*",
                       "ls:Z:*", "lt:i:0", "qs:Z:" + prot[4:22].decode(), "OC:Z:4H10M2D8M3H", f"NM:i:{len(ops) - 15}", "IH:i:2"]
    assert f1[1] == "256" and "qs:Z:" + prot[4:22].decode() in f1  # --sam-bam-seq always
    # uniq: the second record covers the same query range -> "*"; never: both
    capi.write_records(p, capi.LX_OUT_SAM, m, ops, ["q1"], [qlen], ["s0", "s1"], [500, 600], program="blastp", q_ascii=prot, q_ascii_off=[0],
                       options=capi.output_options(sam_tags="qs", sam_hard_clip=0))
    g0, g1 = [l.split("\t") for l in p.read_text().splitlines() if not l.startswith("@")]
    assert g0[11:] == ["qs:Z:" + prot.decode()] and g1[11:] == ["qs:Z:*"]  # soft clips: the whole query
    capi.write_records(p, capi.LX_OUT_SAM, m, ops, ["q1"], [qlen], ["s0", "s1"], [500, 600], program="blastp", q_ascii=prot, q_ascii_off=[0],
                       options=capi.output_options(sam_tags="qs", sam_seq=capi.LX_SAM_SEQ_NEVER))
    assert all(l.endswith("qs:Z:*") for l in p.read_text().splitlines() if not l.startswith("@"))
    with pytest.raises(capi.LambdaExtError, match="Unknown column specifier \"zz\""):
        capi.write_records(p, capi.LX_OUT_SAM, m, ops, ["q1"], [qlen], ["s0", "s1"], [500, 600], options=capi.output_options(sam_tags="AS zz"))


def test_sam_protein_cigar_and_sequence_of_a_translated_query(tmp_path):
    """Tags OC / qs for BLASTX: the protein half of blastMatchTwoCigar (src/search_output.hpp:197-298: clips in protein space, never
    reversed) and the frame's translation (hard clips: the matched residues)."""
    read = b"ACGTTGCAAGGCTTAACCGGTTAAGGCCTTAG"  # 32 nt
    ops = b"MMMMDMMM"
    minus = rec(0, 0, 2, 9, 5, 13, 39.0, alen=len(ops), nm=6, n_ops=len(ops), frame=-3)
    m = np.array([minus], dtype=capi.BLAST_MATCH_DTYPE)
    p = tmp_path / "x.sam"
    capi.write_records(p, capi.LX_OUT_SAM, m, ops, ["r"], [len(read)], ["s0"], [900], program="blastx", q_ascii=read, q_ascii_off=[0],
                       options=capi.output_options(sam_tags="OC qs"))
    f = [l.split("\t") for l in p.read_text().splitlines() if not l.startswith("@")][0]
    assert f[5] == "3H9M3D12M8H"          # DNA cigar: reversed on the minus strand
    assert f[12] == "OC:Z:2H4M1D3M1H"     # protein cigar: as the alignment runs
    # frame -3 = the reverse complement read from its third base
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = "".join(comp[c] for c in reversed(read.decode()))
    from tests.test_translate import dna, prot
    frame = prot(capi.translate_six_frames(dna(rc[2:]))[0])
    assert f[11] == "qs:Z:" + frame[2:9]


def test_tabular_columns_version_line_and_footer(tmp_path):
    """--output-columns (src/search_options.hpp:224-232, :716-760): the NCBI specifiers, "std" = the twelve standard ones; the .m9
    comment block names them; the version line carries lambda's tag only with --version-to-outputfile (src/search_output.hpp:313-345);
    myWriteFooter closes an .m9 (:739-750)."""
    ops = b"M" * 12 + b"I" * 3 + b"M" * 15
    m = np.array([rec(0, 1, 4, 34, 99, 126, 61.23, score=148, alen=30, nm=24, n_ops=30, ev=3.2e-12, ident=80.0, frame=1)], dtype=capi.BLAST_MATCH_DTYPE)
    m["num_gap_opens"], m["num_gap_extensions"], m["num_positives"], m["s_frame"] = 1, 2, 26, 0
    p = tmp_path / "c.m9"
    o = capi.output_options(columns="qseqid qlen sseqid slen score nident positive gaps ppos frames qframe sframe staxids lcataxid std", db_name="db.lba")
    capi.write_records(p, capi.LX_OUT_BLAST_TAB_COMMENTS, m, ops, ["q0 desc"], [40], ["s0", "s1 t"], [300, 400], options=o)
    capi.write_footer(p, capi.LX_OUT_BLAST_TAB_COMMENTS, 1)
    lines = p.read_text().splitlines()
    assert lines[0] == "# BLASTP 2.2.26+" and lines[2] == "# Database: db.lba"
    assert lines[3].startswith("# Fields: query id, query length, subject id, subject length, score, identical, positives, gaps, % positives, "
                               "query/sbjct frames, query frame, sbjct frame, subject tax ids, lowest common ancestor taxonomy ID, query id, subject id, % identity")
    assert lines[5].split("\t") == ["q0", "40", "s1", "400", "148", "24", "26", "3", "86.67", "1/0", "1", "0", "*", "0",
                                    "q0", "s1", "80.00", "30", "6", "1", "5", "34", "100", "126", "3.2e-12", "61.2"]
    assert lines[-1] == "# BLAST processed 1 queries"
    capi.write_records(p, capi.LX_OUT_BLAST_TAB_COMMENTS, m, ops, ["q0"], [40], ["s0", "s1"], [300, 400],
                       options=capi.output_options(version_to_output=1, version="3.0.0"))
    assert p.read_text().splitlines()[0] == ("# BLASTP 2.2.26+ [created by LAMBDA-3.0.0, see http://seqan.de/lambda and please cite correctly "
                                             "in your academic work]")
    with pytest.raises(capi.LambdaExtError, match="Unknown column specifier \"qseq\""):
        capi.write_records(p, capi.LX_OUT_BLAST_TAB, m, ops, ["q0"], [40], ["s0", "s1"], [300, 400], options=capi.output_options(columns="std qseq"))


def test_large_table_written_on_several_threads_equals_small_pieces(tmp_path):
    """A plain table of many records is formatted on the library's host threads (pieces written in order): the file must be the
    concatenation of what the same records give written in small pieces (below the size where the threads come in), both strands and
    odd values included."""
    rng = np.random.default_rng(3)
    n, nq, ns = 70_000, 9_000, 50
    qids = np.sort(rng.integers(0, nq, n))
    recs = []
    for k in range(n):
        qs = int(rng.integers(0, 30))
        recs.append(rec(int(qids[k]), int(rng.integers(0, ns)), qs, qs + int(rng.integers(5, 60)), int(rng.integers(0, 900)), int(rng.integers(900, 1000)),
                        float(rng.integers(200, 9000)) / 10, score=int(rng.integers(20, 400)), ev=float(10.0 ** -rng.integers(1, 80)),
                        ident=float(rng.integers(2000, 10000)) / 100, frame=int(rng.choice([1, -1]))))
    m = np.array(recs, dtype=capi.BLAST_MATCH_DTYPE)
    q_ids, s_ids = [f"q{i} d" for i in range(nq)], [f"s{i}" for i in range(ns)]
    q_lens, s_lens = [100] * nq, [1000] * ns
    whole = tmp_path / "whole.m8"
    capi.write_records(whole, capi.LX_OUT_BLAST_TAB, m, b"", q_ids, q_lens, s_ids, s_lens, program="blastn")
    pieces = tmp_path / "pieces.m8"
    for a in range(0, n, 10_000):
        capi.write_records(pieces, capi.LX_OUT_BLAST_TAB, m[a:a + 10_000], b"", q_ids, q_lens, s_ids, s_lens, program="blastn", write_header=(a == 0))
    assert whole.read_bytes() == pieces.read_bytes() and len(whole.read_text().splitlines()) == n


def test_output_options_are_checked_without_a_file_and_io_errors_are_reported(tmp_path):
    """lx_check_output_options: what a front end calls while it parses its options (the reference throws there,
    /root/reference/src/search_options.hpp:755-758, :803-806) -- and a writer that cannot write says so instead of returning a truncated
    file with LX_OK."""
    import ctypes as C
    import os

    lib = capi.load()
    lib.lx_check_output_options.argtypes = [C.c_int32, C.c_void_p]
    ok = capi.output_options(columns="qseqid sseqid bitscore")
    assert lib.lx_check_output_options(capi.LX_OUT_BLAST_TAB, C.byref(ok)) == capi.LX_OK
    assert lib.lx_check_output_options(capi.LX_OUT_SAM, C.byref(capi.output_options(sam_tags="AS NM"))) == capi.LX_OK
    assert lib.lx_check_output_options(capi.LX_OUT_BLAST_TAB, None) == capi.LX_OK
    bad = capi.output_options(columns="qseqid nosuchcolumn")
    assert lib.lx_check_output_options(capi.LX_OUT_BLAST_TAB_COMMENTS, C.byref(bad)) == capi.LX_EINVAL
    assert "nosuchcolumn" in capi.last_output_error()
    assert lib.lx_check_output_options(capi.LX_OUT_SAM, C.byref(capi.output_options(sam_tags="AS zz"))) == capi.LX_EINVAL
    assert lib.lx_check_output_options(99, None) == capi.LX_EINVAL
    bms = np.array([rec(0, 0, 0, 50, 0, 50, 60.0, n_ops=0)], dtype=capi.BLAST_MATCH_DTYPE)
    if os.path.exists("/dev/full"):
        with pytest.raises(capi.LambdaExtError) as ei:
            capi.write_records("/dev/full", capi.LX_OUT_BLAST_TAB, np.repeat(bms, 5000), b"", ["q"], [50], ["s"], [50])
        assert "error while writing" in str(ei.value)
    with pytest.raises(capi.LambdaExtError) as ei:
        capi.write_records(tmp_path / "no" / "such" / "dir.m8", capi.LX_OUT_BLAST_TAB, bms, b"", ["q"], [50], ["s"], [50])
    assert "cannot open" in str(ei.value)
