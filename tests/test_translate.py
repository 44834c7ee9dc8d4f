"""CPU tests of the frame bookkeeping and the six-frame translation (row N4): the
host functions behind include/lambda_ext.h against the oracle restatement and against hand-checked cases."""
import numpy as np
import pytest

from lambda_amd import capi

AA = "ABCDEFGHIJKLMNOPQRSTUVWYZX*"  # SeqAn AminoAcid rank order
DNA = "ACGNT"                        # BioC++ dna5 rank order


def dna(s):
    return np.array([DNA.index(c) for c in s], dtype=np.uint8)


def prot(r):
    return "".join(AA[x] for x in r)


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))


def test_known_translations():
    seq = "ATGGCCATTGTAATGGGCCGCTGAAAGGGTGCCCGATAG"  # textbook ORF: MAIVMGR*KGAR*
    f = capi.translate_six_frames(dna(seq))
    assert prot(f[0]) == "MAIVMGR*KGAR*"
    assert [len(x) for x in f] == [13, 12, 12, 13, 12, 12]
    # the reverse frames are the forward frames of the reverse complement
    g = capi.translate_six_frames(dna(revcomp(seq)))
    assert [prot(x) for x in f[3:]] == [prot(x) for x in g[:3]]
    # all 64 codons against the code written out by amino acid
    code = {"F": "TTT TTC", "L": "TTA TTG CTT CTC CTA CTG", "I": "ATT ATC ATA", "M": "ATG", "V": "GTT GTC GTA GTG",
            "S": "TCT TCC TCA TCG AGT AGC", "P": "CCT CCC CCA CCG", "T": "ACT ACC ACA ACG", "A": "GCT GCC GCA GCG",
            "Y": "TAT TAC", "*": "TAA TAG TGA", "H": "CAT CAC", "Q": "CAA CAG", "N": "AAT AAC", "K": "AAA AAG",
            "D": "GAT GAC", "E": "GAA GAG", "C": "TGT TGC", "W": "TGG", "R": "CGT CGC CGA CGG AGA AGG",
            "G": "GGT GGC GGA GGG"}
    seen = 0
    for aa, codons in code.items():
        for c in codons.split():
            assert prot(capi.translate_six_frames(dna(c))[0]) == aa, c
            seen += 1
    assert seen == 64


def test_ambiguous_codons():
    # N in the wobble position of a four-fold degenerate family keeps the amino acid; otherwise X
    for codon, aa in (("CTN", "L"), ("GCN", "A"), ("TCN", "S"), ("CGN", "R"), ("TAN", "X"), ("NNN", "X"), ("ANG", "X"),
                      ("ATN", "X"), ("NTG", "X"), ("TTN", "X")):
        assert prot(capi.translate_six_frames(dna(codon))[0]) == aa, codon


def test_translation_matches_oracle(oracle):
    rng = np.random.default_rng(5)
    for n in (0, 1, 2, 3, 4, 5, 6, 7, 100, 301, 1000):
        d = rng.choice(np.array([0, 1, 2, 3, 4], dtype=np.uint8), size=n, p=[0.24, 0.24, 0.24, 0.04, 0.24])
        got = capi.translate_six_frames(d)
        for k, frame in enumerate((1, 2, 3, -1, -2, -3)):
            want = oracle.translate_frame(d, frame)
            assert len(got[k]) == len(want) == max(0, (n - (abs(frame) - 1)) // 3)
            assert (got[k] == want).all(), (n, frame)
    with pytest.raises(capi.LambdaExtError):
        capi.translate_six_frames(np.array([0, 1, 5], dtype=np.uint8))
    with pytest.raises(capi.LambdaExtError):
        capi.translate_six_frames(dna("ATGGCC"), genetic_code=7)  # NCBI / bio::alphabet::genetic_code define no table 7


def test_other_genetic_codes():
    """--genetic-code of the reference is a bio::alphabet::genetic_code id = an NCBI translation table
    (src/search_options.hpp:170, :628).  The tables are pinned by their published differences from table 1 (gc.prt)."""
    diffs = {
        2: {"AGA": "*", "AGG": "*", "ATA": "M", "TGA": "W"},                                  # vertebrate mitochondrial
        3: {"ATA": "M", "CTT": "T", "CTC": "T", "CTA": "T", "CTG": "T", "TGA": "W"},          # yeast mitochondrial
        4: {"TGA": "W"},                                                                      # mold mitochondrial / Mycoplasma
        5: {"AGA": "S", "AGG": "S", "ATA": "M", "TGA": "W"},                                  # invertebrate mitochondrial
        6: {"TAA": "Q", "TAG": "Q"},                                                          # ciliate
        9: {"AAA": "N", "AGA": "S", "AGG": "S", "TGA": "W"},                                  # echinoderm mitochondrial
        10: {"TGA": "C"},                                                                     # euplotid
        11: {},                                                                               # bacterial: as table 1
        12: {"CTG": "S"},                                                                     # alternative yeast
        13: {"AGA": "G", "AGG": "G", "ATA": "M", "TGA": "W"},                                 # ascidian mitochondrial
        14: {"AAA": "N", "AGA": "S", "AGG": "S", "TAA": "Y", "TGA": "W"},                     # alternative flatworm mitochondrial
        15: {"TAG": "Q"},                                                                     # Blepharisma
        16: {"TAG": "L"},                                                                     # chlorophycean mitochondrial
        21: {"TGA": "W", "ATA": "M", "AGA": "S", "AGG": "S", "AAA": "N"},                     # trematode mitochondrial
        22: {"TCA": "*", "TAG": "L"},                                                         # Scenedesmus mitochondrial
        23: {"TTA": "*"},                                                                     # Thraustochytrium mitochondrial
        24: {"AGA": "S", "AGG": "K", "TGA": "W"},                                             # Pterobranchia mitochondrial
        25: {"TGA": "G"},                                                                     # Gracilibacteria
    }
    bases = "TCAG"
    codons = [a + b + c for a in bases for b in bases for c in bases]
    std = {c: prot(capi.translate_six_frames(dna(c))[0]) for c in codons}
    for code, d in diffs.items():
        for c in codons:
            assert prot(capi.translate_six_frames(dna(c), genetic_code=code)[0]) == d.get(c, std[c]), (code, c)
    # the reverse frames use the same table: TCA is the reverse complement of TGA
    f = capi.translate_six_frames(dna("TCA"), genetic_code=4)
    assert prot(f[0]) == "S" and prot(f[3]) == "W"
    # ambiguity under another table: TGN is C/C/W/W in table 4 -> X, ATN is I/I/I/M in both -> X, AGN: S/S/R/R vs S/S/S/S in table 5
    assert prot(capi.translate_six_frames(dna("AGN"), genetic_code=5)[0]) == "S"
    assert prot(capi.translate_six_frames(dna("AGN"), genetic_code=1)[0]) == "X"


def test_frames_match_oracle_and_roundtrip(oracle):
    lib = capi.load()
    for mode, nframes_q, nframes_s in ((capi.LX_FRAMES_NONE, 1, 1), (capi.LX_FRAMES_REVCOMP, 2, 2),
                                       (capi.LX_FRAMES_TRANSLATED, 6, 6), (capi.LX_FRAMES_BISULFITE, 4, 2)):
        for ident in range(0, 48):
            qf, sf = capi.set_frames(mode, mode, ident, ident)
            assert qf == oracle.frame_of(mode, ident, False) and sf == oracle.frame_of(mode, ident, True)
            uq = lib.lx_untrue_qry_id(mode, ident // nframes_q, qf)
            us = lib.lx_untrue_subj_id(mode, ident // nframes_s, sf)
            assert uq == oracle.untrue_id(mode, ident // nframes_q, qf, False)
            assert us == oracle.untrue_id(mode, ident // nframes_s, sf, True)
            if mode != capi.LX_FRAMES_BISULFITE:
                assert uq == ident and us == ident  # _untrue*Id inverts _setFrames
            else:
                # the bisulfite duplicates are identical sequences: mapped onto the first copy of their strand
                assert uq == ident - ident % 2 and us == ident - ident % 2
    # the documented frame values
    assert [capi.set_frames(2, 0, i, 0)[0] for i in range(6)] == [1, 2, 3, -1, -2, -3]
    assert [capi.set_frames(3, 3, i, i) for i in range(4)] == [(1, 1), (2, 2), (-1, 1), (-2, 2)]
    assert [capi.set_frames(1, 1, i, i) for i in range(2)] == [(1, 1), (-1, -1)]
