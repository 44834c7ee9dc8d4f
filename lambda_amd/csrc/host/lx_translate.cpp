// lx_translate.cpp -- frame bookkeeping and six-frame translation.
//
// Host-side C++ mirror of
//   _setFrames                         /root/reference/src/search_algo.hpp:768-814
//   _untrueQryId / _untrueSubjId       src/search_algo.hpp:940-996
//   qryTransView / sbjTransView        src/shared_definitions.hpp:246-281   (which expansion a program uses)
// The translation itself is BioC++'s (bio::views::translate_join, library absent here): frames in the order
// +1 +2 +3 -1 -2 -3, any of NCBI's genetic codes (table below).  [UPSTREAM-RECALL] a codon containing an ambiguous nucleotide is
// translated to the amino acid all of its completions share, otherwise to X.
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "../../../include/lambda_ext.h"

namespace
{

// NCBI translation tables (gc.prt; the ids are bio::alphabet::genetic_code's, the enum the reference casts its
// --genetic-code option to: /root/reference/src/search_options.hpp:170, :628, src/mkindex_options.hpp:240), codon index
// = 16 b1 + 4 b2 + b3 with T = 0, C = 1, A = 2, G = 3.  Table 11 differs from 1 in its start codons only.
struct GeneticCode
{
    int          id;
    char const * aa;
};
constexpr GeneticCode kGeneticCodes[] = {
    {1, "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},  // canonical
    {2, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSS**VVVVAAAADDEEGGGG"},  // vertebrate mitochondrial
    {3, "FFLLSSSSYY**CCWWTTTTPPPPHHQQRRRRIIMMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},  // yeast mitochondrial
    {4, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},  // mold / protozoan / coelenterate mitochondrial, Mycoplasma
    {5, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSSSVVVVAAAADDEEGGGG"},  // invertebrate mitochondrial
    {6, "FFLLSSSSYYQQCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},  // ciliate
    {9, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG"},  // echinoderm / flatworm mitochondrial
    {10, "FFLLSSSSYY**CCCWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"}, // euplotid
    {11, "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"}, // bacterial, archaeal, plant plastid
    {12, "FFLLSSSSYY**CC*WLLLSPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"}, // alternative yeast
    {13, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSGGVVVVAAAADDEEGGGG"}, // ascidian mitochondrial
    {14, "FFLLSSSSYYY*CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG"}, // alternative flatworm mitochondrial
    {15, "FFLLSSSSYY*QCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"}, // Blepharisma
    {16, "FFLLSSSSYY*LCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"}, // chlorophycean mitochondrial
    {21, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNNKSSSSVVVVAAAADDEEGGGG"}, // trematode mitochondrial
    {22, "FFLLSS*SYY*LCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"}, // Scenedesmus obliquus mitochondrial
    {23, "FF*LSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"}, // Thraustochytrium mitochondrial
    {24, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSSKVVVVAAAADDEEGGGG"}, // Pterobranchia mitochondrial
    {25, "FFLLSSSSYY**CCGWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"}, // candidate division SR1, Gracilibacteria
};
constexpr int kNumGeneticCodes = sizeof(kGeneticCodes) / sizeof(GeneticCode);
// SeqAn AminoAcid rank order (the order of the scoring tables, src/seqan2_to_biocpp.hpp:352-366)
constexpr char kSeqanAa[28] = "ABCDEFGHIJKLMNOPQRSTUVWYZX*";
constexpr uint8_t kRankX = 25;

// BioC++ dna5 rank (A, C, G, N, T) -> TCAG index, N = 4
constexpr uint8_t kToTcag[5] = {2, 1, 3, 4, 0};
// complement in dna5 ranks: A<->T, C<->G, N stays
constexpr uint8_t kComplement[5] = {4, 2, 1, 3, 0};

struct CodonTable
{
    uint8_t aa[125]; // [b1][b2][b3] in dna5 ranks -> SeqAn aa rank
    explicit CodonTable(char const * kCanonical)
    {
        auto rankOf = [](char c) -> uint8_t { return (uint8_t)(std::strchr(kSeqanAa, c) - kSeqanAa); };
        for (int b1 = 0; b1 < 5; ++b1)
            for (int b2 = 0; b2 < 5; ++b2)
                for (int b3 = 0; b3 < 5; ++b3)
                {
                    int const t[3] = {kToTcag[b1], kToTcag[b2], kToTcag[b3]};
                    int       first = -1;
                    bool      same  = true;
                    // every completion of the ambiguous positions
                    for (int x1 = (t[0] == 4 ? 0 : t[0]); x1 <= (t[0] == 4 ? 3 : t[0]); ++x1)
                        for (int x2 = (t[1] == 4 ? 0 : t[1]); x2 <= (t[1] == 4 ? 3 : t[1]); ++x2)
                            for (int x3 = (t[2] == 4 ? 0 : t[2]); x3 <= (t[2] == 4 ? 3 : t[2]); ++x3)
                            {
                                int const a = kCanonical[16 * x1 + 4 * x2 + x3];
                                if (first < 0)
                                    first = a;
                                else if (a != first)
                                    same = false;
                            }
                    aa[25 * b1 + 5 * b2 + b3] = same ? rankOf((char)first) : kRankX;
                }
    }
};

// the table of a genetic code, nullptr for an id NCBI / BioC++ do not define
CodonTable const * codonTable(int genetic_code)
{
    static CodonTable const tabs[kNumGeneticCodes] = {
        CodonTable(kGeneticCodes[0].aa),  CodonTable(kGeneticCodes[1].aa),  CodonTable(kGeneticCodes[2].aa),  CodonTable(kGeneticCodes[3].aa),
        CodonTable(kGeneticCodes[4].aa),  CodonTable(kGeneticCodes[5].aa),  CodonTable(kGeneticCodes[6].aa),  CodonTable(kGeneticCodes[7].aa),
        CodonTable(kGeneticCodes[8].aa),  CodonTable(kGeneticCodes[9].aa),  CodonTable(kGeneticCodes[10].aa), CodonTable(kGeneticCodes[11].aa),
        CodonTable(kGeneticCodes[12].aa), CodonTable(kGeneticCodes[13].aa), CodonTable(kGeneticCodes[14].aa), CodonTable(kGeneticCodes[15].aa),
        CodonTable(kGeneticCodes[16].aa), CodonTable(kGeneticCodes[17].aa), CodonTable(kGeneticCodes[18].aa)};
    static_assert(kNumGeneticCodes == 19, "one table per genetic code");
    for (int i = 0; i < kNumGeneticCodes; ++i)
        if (kGeneticCodes[i].id == genetic_code)
            return &tabs[i];
    return nullptr;
}

int32_t frameOf(int mode, uint64_t id, bool subject)
{
    switch (mode)
    {
        case LX_FRAMES_TRANSLATED: // :772-776, :795-799
        {
            int32_t f = (int32_t)(id % 3) + 1;
            return (id % 6 > 2) ? -f : f;
        }
        case LX_FRAMES_BISULFITE: // :778-782 (query), :801-803 (subject: never negative)
        {
            int32_t f = (int32_t)(id % 2) + 1;
            return (!subject && id % 4 > 1) ? -f : f;
        }
        case LX_FRAMES_REVCOMP: // :784-788, :805-809
            return (id % 2) ? -1 : 1;
        default:
            return 0;
    }
}

} // namespace

extern "C" {

void lx_set_frames(int q_mode, int s_mode, uint64_t qry_id, uint64_t subj_id, int32_t * q_frame, int32_t * s_frame)
{
    if (q_frame)
        *q_frame = frameOf(q_mode, qry_id, false);
    if (s_frame)
        *s_frame = frameOf(s_mode, subj_id, true);
}

uint64_t lx_untrue_qry_id(int q_mode, uint64_t n_qid, int32_t q_frame)
{
    switch (q_mode)
    {
        case LX_FRAMES_TRANSLATED: return q_frame > 0 ? n_qid * 6 + (uint64_t)(q_frame - 1) : n_qid * 6 + (uint64_t)(2 - q_frame); // :945-950
        case LX_FRAMES_BISULFITE: return q_frame > 0 ? n_qid * 4 : n_qid * 4 + 2;                                                  // :952-958
        case LX_FRAMES_REVCOMP: return q_frame > 0 ? n_qid * 2 : n_qid * 2 + 1;                                                    // :959-965
        default: return n_qid;
    }
}

uint64_t lx_untrue_subj_id(int s_mode, uint64_t n_sid, int32_t s_frame)
{
    switch (s_mode)
    {
        case LX_FRAMES_TRANSLATED: return s_frame > 0 ? n_sid * 6 + (uint64_t)(s_frame - 1) : n_sid * 6 + (uint64_t)(2 - s_frame); // :977-982
        case LX_FRAMES_BISULFITE:
        case LX_FRAMES_REVCOMP: return s_frame > 0 ? n_sid * 2 : n_sid * 2 + 1;                                                    // :984-990
        default: return n_sid;
    }
}

int lx_translate_six_frames(uint8_t const * dna5, uint64_t n, int genetic_code, uint8_t * out, uint64_t out_capacity,
                            uint64_t * frame_off, uint64_t * frame_len)
{
    CodonTable const * const tabp = codonTable(genetic_code);
    if (!tabp || !frame_off || !frame_len || (!dna5 && n) || (!out && n >= 3))
        return LX_EINVAL;
    uint64_t total = 0;
    for (int f = 0; f < 6; ++f)
    {
        uint64_t const shift = (uint64_t)(f % 3);
        frame_off[f] = total;
        frame_len[f] = n >= shift ? (n - shift) / 3 : 0;
        total += frame_len[f];
    }
    if (total > out_capacity)
        return LX_EINVAL;
    for (uint64_t i = 0; i < n; ++i)
        if (dna5[i] > 4)
            return LX_EINVAL;
    CodonTable const & tab = *tabp;
    for (int f = 0; f < 3; ++f)
    {
        uint8_t * fwd = out + frame_off[f];
        uint8_t * rev = out + frame_off[f + 3];
        for (uint64_t k = 0; k < frame_len[f]; ++k)
        {
            uint64_t const p = (uint64_t)f + 3 * k;
            fwd[k] = tab.aa[25 * dna5[p] + 5 * dna5[p + 1] + dna5[p + 2]];
            // codon k of frame -(f+1): positions p, p+1, p+2 of the reverse complement = n-1-p, n-2-p, n-3-p complemented
            rev[k] = tab.aa[25 * kComplement[dna5[n - 1 - p]] + 5 * kComplement[dna5[n - 2 - p]] + kComplement[dna5[n - 3 - p]]];
        }
    }
    return LX_OK;
}

} // extern "C"
