// lx_output.cpp -- record post-processing and BLAST-tabular / SAM writers.
//
// Host-side C++ mirror of
//   _writeRecord          /root/reference/src/search_algo.hpp:820-913   (sort, dedupe, bit-score order, top-N)
//   myWriteHeader         src/search_output.hpp:305-461
//   myWriteRecord         src/search_output.hpp:463-733                  (tabular via seqan::writeRecord; SAM records)
//   blastMatchOneCigar    src/search_output.hpp:115-194                  (hard or soft clips, frame clips always hard)
//   blastMatchTwoCigar    src/search_output.hpp:197-298                  (the protein-space cigar of tag OC)
//   the output options    src/search_options.hpp:224-379, :716-816       (lx_output_options: columns, SAM tags, sequence, clipping)
// Scope: every program in the tabular formats and in SAM; not BAM and not the pairwise .m0 report (seqan's writers, absent).  The
// number formats of the tabular columns are SeqAn2's (source absent): [UPSTREAM-RECALL] pident %.2f, evalue %.1e, bitscore %.1f,
// 1-based inclusive positions.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include <functional>

#include "../../../include/lambda_ext.h"
#include "scoring_tables.hpp"

#include "../lx_host_pool.h" // the library's host threads

namespace
{

std::string firstWord(char const * id)
{
    std::string s(id ? id : "");
    size_t      p = s.find_first_of(" \t");
    return p == std::string::npos ? s : s.substr(0, p);
}

// blastMatchOneCigar, src/search_output.hpp:115-194.  qLen = length of the untranslated query; a translated query
// (qTrans) is reported in nucleotide space: every run x 3 (:124), the nucleotides in front of the frame and the
// incomplete codon behind it as hard clips (:126-127), the whole element list reversed on the minus strand (:192-193).
std::string cigarOf(lx_blast_match const & m, uint8_t const * ops, uint64_t qLen, bool hardClip, bool qTrans = false)
{
    std::vector<std::pair<char, uint64_t>> el;
    uint64_t const transFac       = qTrans ? 3 : 1;
    uint64_t const leftFrameClip  = (uint64_t)std::abs((int)m.q_frame) - (m.q_frame != 0 ? 1 : 0);
    uint64_t const rightFrameClip = qTrans ? (qLen - leftFrameClip) % 3 : 0;
    uint64_t const frameLen       = qTrans ? (qLen - leftFrameClip) / 3 : qLen; // length(source(alignRow0))
    uint64_t const leftClip = m.q_start * transFac, rightClip = (frameLen - m.q_end) * transFac;
    if (hardClip)
    {
        if (leftFrameClip + leftClip > 0)
            el.emplace_back('H', leftFrameClip + leftClip);
    }
    else
    {
        // (:126 takes |qFrameShift| - 1 for every program: 0 for the frames +-1 of BLASTN, 1 for the second bisulfite
        // duplicate of a strand, frames +-2 of :778-782)
        if (leftFrameClip > 0)
            el.emplace_back('H', leftFrameClip);
        if (leftClip > 0)
            el.emplace_back('S', leftClip);
    }
    uint8_t const * o = ops + m.ops_off;
    for (uint32_t i = 0; i < m.n_ops;)
    {
        uint32_t cnt = 0;
        while (i < m.n_ops && o[i] == 'D') // gap in row0 = deletion in query
        {
            ++cnt;
            ++i;
        }
        if (cnt)
            el.emplace_back('D', cnt * transFac);
        cnt = 0;
        while (i < m.n_ops && o[i] == 'I')
        {
            ++cnt;
            ++i;
        }
        if (cnt)
            el.emplace_back('I', cnt * transFac);
        cnt = 0;
        while (i < m.n_ops && o[i] == 'M')
        {
            ++cnt;
            ++i;
        }
        if (cnt)
            el.emplace_back('M', cnt * transFac);
        if (i < m.n_ops && o[i] != 'D' && o[i] != 'I' && o[i] != 'M') // (not an op byte: the caller's offsets are wrong -- no endless loop)
            return "*";
    }
    if (hardClip)
    {
        if (rightFrameClip + rightClip > 0)
            el.emplace_back('H', rightFrameClip + rightClip);
    }
    else
    {
        if (rightClip > 0)
            el.emplace_back('S', rightClip);
        if (rightFrameClip > 0)
            el.emplace_back('H', rightFrameClip);
    }
    if (m.q_frame < 0) // every program, BLASTN's reverse-complement query frame included (src/search_output.hpp:192-193)
        std::reverse(el.begin(), el.end());
    std::string c;
    for (auto const & e : el)
        c += std::to_string(e.second) + e.first;
    return c;
}

// The protein-space cigar of tag OC: blastMatchOneCigar for a protein query (transFac 1, no frame clips: BLASTP / TBLASTN,
// src/search_output.hpp:517-521) and the protCigar half of blastMatchTwoCigar for a translated one (:197-298: clips in
// protein space, hard or soft like the DNA cigar, never reversed).
std::string protCigarOf(lx_blast_match const & m, uint8_t const * ops, uint64_t qLen, bool hardClip, bool qTrans)
{
    uint64_t const leftFrameClip = (uint64_t)std::abs((int)m.q_frame) - (m.q_frame != 0 ? 1 : 0);
    uint64_t const frameLen      = qTrans ? (qLen - leftFrameClip) / 3 : qLen;
    uint64_t const leftClip = m.q_start, rightClip = frameLen >= m.q_end ? frameLen - m.q_end : 0;
    std::string    c;
    auto           add = [&](char op, uint64_t cnt)
    {
        if (cnt)
            c += std::to_string(cnt) + op;
    };
    add(hardClip ? 'H' : 'S', leftClip);
    uint8_t const * o = ops + m.ops_off;
    for (uint32_t i = 0; i < m.n_ops;)
    {
        if (o[i] != 'D' && o[i] != 'I' && o[i] != 'M')
            return "*";
        uint32_t const i0 = i;
        while (i < m.n_ops && o[i] == o[i0])
            ++i;
        add((char)o[i0], i - i0);
    }
    add(hardClip ? 'H' : 'S', rightClip);
    return c.empty() ? "*" : c;
}

thread_local std::string g_output_error;

// SamBamExtraTags, src/search_output.hpp:29-76: keys and descriptions in the order of the enum (= the order of the header line)
struct SamTag
{
    char const * key, * desc;
};
enum
{
    kTagBitScore, kTagProtCigar, kTagEditDistance, kTagMatchCount, kTagScore, kTagEValue, kTagPIdent, kTagPPos, kTagQFrame, kTagProtSeq,
    kTagSFrame, kTagSTaxIds, kTagLcaId, kTagLcaTaxId, kNumSamTags
};
constexpr SamTag kSamTags[kNumSamTags] = {
    {"AS", "bit score"},
    {"OC", "query protein cigar (* for BLASTN)"},
    {"NM", "edit distance (in protein space unless BLASTN)"},
    {"IH", "number of matches this query has"},
    {"ar", "raw score"},
    {"ae", "expect value"},
    {"ai", "% identity (in protein space unless BLASTN) "},
    {"ap", "% positive (in protein space unless BLASTN)"},
    {"qf", "query frame"},
    {"qs", "query protein sequence (* for BLASTN)"},
    {"sf", "subject frame"},
    {"st", "subject taxonomy IDs (* if n/a)"},
    {"ls", "lowest common ancestor scientific name"},
    {"lt", "lowest common ancestor taxonomy ID"},
};

std::vector<std::string> splitWords(char const * text)
{
    std::vector<std::string> out;
    std::string              cur;
    for (char const * p = text; p && *p; ++p)
    {
        if (std::isspace((unsigned char)*p))
        {
            if (!cur.empty())
                out.push_back(cur);
            cur.clear();
        }
        else
            cur += *p;
    }
    if (!cur.empty())
        out.push_back(cur);
    return out;
}

// --sam-bam-tags (src/search_options.hpp:772-806): every key must be one of the fourteen; the order given is not preserved
bool resolveTags(char const * text, bool (&tags)[kNumSamTags])
{
    for (std::string const & w : splitWords(text ? text : "AS NM ae ai qf")) // :351
    {
        int t = 0;
        while (t < kNumSamTags && w != kSamTags[t].key)
            ++t;
        if (t == kNumSamTags)
        {
            g_output_error = "Unknown column specifier \"" + w + "\". Please see \"--sam-bam-tags help\" for valid options.";
            return false;
        }
        tags[t] = true;
    }
    return true;
}

// --output-columns (src/search_options.hpp:716-760): NCBI's specifiers as seqan::BlastMatchField labels them; "std" = the twelve
// standard columns.  The descriptions are the "# Fields:" labels of .m9 (NCBI's).  Not offered: the columns that need sequence
// data or database titles the writer is not given (qseq, sseq, btop, stitle, ...), and the GI / accession variants.
struct Column
{
    char const * label, * desc;
};
enum
{
    kColQSeqId, kColSSeqId, kColPIdent, kColLength, kColMismatch, kColGapOpen, kColQStart, kColQEnd, kColSStart, kColSEnd, kColEValue,
    kColBitScore, kColQLen, kColSLen, kColScore, kColNIdent, kColPositive, kColGaps, kColPPos, kColFrames, kColQFrame, kColSFrame,
    kColSTaxIds, kColLcaTaxId, kNumColumns
};
constexpr Column kColumns[kNumColumns] = {
    {"qseqid", "query id"},       {"sseqid", "subject id"},      {"pident", "% identity"},  {"length", "alignment length"},
    {"mismatch", "mismatches"},   {"gapopen", "gap opens"},      {"qstart", "q. start"},    {"qend", "q. end"},
    {"sstart", "s. start"},       {"send", "s. end"},            {"evalue", "evalue"},      {"bitscore", "bit score"},
    {"qlen", "query length"},     {"slen", "subject length"},    {"score", "score"},        {"nident", "identical"},
    {"positive", "positives"},    {"gaps", "gaps"},              {"ppos", "% positives"},   {"frames", "query/sbjct frames"},
    {"qframe", "query frame"},    {"sframe", "sbjct frame"},     {"staxids", "subject tax ids"}, {"lcataxid", "lowest common ancestor taxonomy ID"},
};

bool resolveColumns(char const * text, std::vector<int> & cols)
{
    for (std::string const & w : splitWords(text ? text : "std"))
    {
        if (w == "std")
        {
            for (int c = kColQSeqId; c <= kColBitScore; ++c)
                cols.push_back(c);
            continue;
        }
        int c = 0;
        while (c < kNumColumns && w != kColumns[c].label)
            ++c;
        if (c == kNumColumns)
        {
            g_output_error = "Unknown column specifier \"" + w + "\". Please see -oc help for valid options.";
            return false;
        }
        cols.push_back(c);
    }
    return !cols.empty();
}

void reverseComplementAscii(std::string & seq)
{
    std::reverse(seq.begin(), seq.end());
    for (char & ch : seq)
        switch (std::toupper((unsigned char)ch))
        {
            case 'A': ch = 'T'; break;
            case 'C': ch = 'G'; break;
            case 'G': ch = 'C'; break;
            case 'T': case 'U': ch = 'A'; break;
            default: ch = 'N';
        }
}

} // namespace

extern "C" {

uint64_t lx_postprocess_records(lx_blast_match * m, uint64_t n, uint64_t max_matches, lx_record_stats * stats)
{
    lx_record_stats st{};
    uint64_t        w = 0;
    for (uint64_t lo = 0; lo < n;)
    {
        uint64_t hi = lo + 1;
        while (hi < n && m[hi].n_qid == m[lo].n_qid)
            ++hi;
        ++st.qrys_with_hit; // :826
        std::vector<lx_blast_match> rec(m + lo, m + hi);
        // order by subject, coordinates and frames; among equal keys the better bit score comes first (b and a swapped in the
        // last tie component), so that std::unique below keeps it (:832-853)
        std::stable_sort(rec.begin(), rec.end(),
                         [](lx_blast_match const & a, lx_blast_match const & b)
                         {
                             return std::tie(a.n_sid, a.q_start, a.q_end, a.s_start, a.s_end, a.q_frame, a.s_frame, b.bit_score) <
                                    std::tie(b.n_sid, b.q_start, b.q_end, b.s_start, b.s_end, b.q_frame, b.s_frame, a.bit_score);
                         });
        // one record per (subject, coordinates, frames): the first of each group survives (:856-862)
        auto const before = rec.size();
        rec.erase(std::unique(rec.begin(), rec.end(),
                              [](lx_blast_match const & a, lx_blast_match const & b)
                              {
                                  return std::tie(a.n_sid, a.q_start, a.q_end, a.s_start, a.s_end, a.q_frame, a.s_frame) ==
                                         std::tie(b.n_sid, b.q_start, b.q_end, b.s_start, b.s_end, b.q_frame, b.s_frame);
                              }),
                  rec.end());
        st.hits_duplicate2 += before - rec.size();
        // output order: best bit score (= smallest e-value) first; stable like the reference's list sort (:865)
        std::stable_sort(rec.begin(), rec.end(),
                         [](lx_blast_match const & a, lx_blast_match const & b) { return a.bit_score > b.bit_score; });
        // at most max_matches records per query, the rest is counted (:867-872)
        if (rec.size() > max_matches)
        {
            st.hits_abundant += rec.size() - max_matches;
            rec.resize(max_matches);
        }
        st.hits_final += rec.size();
        std::set<uint64_t> uniq; // :876-882
        for (auto const & r : rec)
            uniq.insert(r.n_sid);
        st.pairs += uniq.size();
        for (auto const & r : rec)
            m[w++] = r;
        lo = hi;
    }
    if (stats)
        *stats = st;
    return w;
}

// computeLCA (src/search_misc.hpp:86-112): level the two nodes, then climb together; 0 = the paths never met
static bool lca_of(lx_tax_tree const & t, uint32_t n1, uint32_t n2, uint32_t & out)
{
    if (n1 == n2)
    {
        out = n1;
        return true;
    }
    if (n1 >= t.n_taxa || n2 >= t.n_taxa)
        return false;
    for (uint32_t i = t.heights[n1]; i > t.heights[n2]; --i)
    {
        n1 = t.parents[n1];
        if (n1 >= t.n_taxa)
            return false;
    }
    for (uint32_t i = t.heights[n2]; i > t.heights[n1]; --i)
    {
        n2 = t.parents[n2];
        if (n2 >= t.n_taxa)
            return false;
    }
    while (n1 != 0 && n2 != 0)
    {
        if (n1 == n2)
        {
            out = n1;
            return true;
        }
        n1 = t.parents[n1];
        n2 = t.parents[n2];
        if (n1 >= t.n_taxa || n2 >= t.n_taxa)
            return false;
    }
    return false; // "LCA-computation error: One of the paths didn't lead to root."
}

int lx_compute_lca(lx_blast_match const * m, uint64_t n, lx_tax_tree const * tree, uint64_t * out_qid, uint32_t * out_lca,
                   uint64_t * out_n)
{
    if ((!m && n) || !tree || !out_qid || !out_lca || !out_n || !tree->parents || !tree->heights || !tree->s_tax_off ||
        (!tree->s_tax_ids && tree->n_s && tree->s_tax_off[tree->n_s]))
        return LX_EINVAL;
    lx_tax_tree const & t = *tree;
    uint64_t            w = 0;
    for (uint64_t lo = 0; lo < n;)
    {
        uint64_t hi = lo + 1;
        while (hi < n && m[hi].n_qid == m[lo].n_qid)
            ++hi;
        for (uint64_t k = lo; k < hi; ++k)
            if (m[k].n_sid >= t.n_s)
                return LX_EINVAL;
        // the first match whose subject's first taxon is assigned (has a parent) starts the fold (:887-896)
        uint32_t lca = 0;
        for (uint64_t k = lo; k < hi && lca == 0; ++k)
        {
            uint64_t const a = t.s_tax_off[m[k].n_sid], b = t.s_tax_off[m[k].n_sid + 1];
            if (b > a)
            {
                uint32_t const first = t.s_tax_ids[a];
                if (first >= t.n_taxa)
                    return LX_EINVAL;
                if (t.parents[first] != 0)
                    lca = first;
            }
        }
        if (lca != 0) // every assigned taxon of every match (:898-906); unassigned ones are ignored
            for (uint64_t k = lo; k < hi; ++k)
                for (uint64_t x = t.s_tax_off[m[k].n_sid]; x < t.s_tax_off[m[k].n_sid + 1]; ++x)
                {
                    uint32_t const tax = t.s_tax_ids[x];
                    if (tax >= t.n_taxa)
                        return LX_EINVAL;
                    if (t.parents[tax] != 0 && !lca_of(t, tax, lca, lca))
                        return LX_EINVAL;
                }
        out_qid[w] = m[lo].n_qid;
        out_lca[w] = lca;
        ++w;
        lo = hi;
    }
    *out_n = w;
    return LX_OK;
}

void lx_output_options_default(lx_output_options * o)
{
    if (!o)
        return;
    *o               = lx_output_options{};
    o->sam_seq       = LX_SAM_SEQ_UNIQ; // src/search_options.hpp:339
    o->sam_hard_clip = 1;               // :360
    o->genetic_code  = 1;               // :170
}

char const * lx_last_output_error(void)
{
    return g_output_error.c_str();
}

int lx_write_footer(char const * path, int format, uint64_t n_records)
{
    if (!path)
        return LX_EINVAL;
    if (format != LX_OUT_BLAST_TAB_COMMENTS)
        return LX_OK;
    std::FILE * f = std::fopen(path, "a");
    if (!f)
        return LX_EINVAL;
    bool ok = std::fprintf(f, "# BLAST processed %llu queries\n", (unsigned long long)n_records) >= 0; // [UPSTREAM-RECALL] seqan::writeFooter
    ok      = (std::fclose(f) == 0) && ok;
    if (!ok)
        g_output_error = std::string("error while writing ") + path;
    return ok ? LX_OK : LX_EINVAL;
}

// what lx_write_records_ex would refuse before it touches a file: the format, the column specifiers / SAM tags
int lx_check_output_options(int format, lx_output_options const * opt_in)
{
    g_output_error.clear();
    if (format != LX_OUT_BLAST_TAB && format != LX_OUT_BLAST_TAB_COMMENTS && format != LX_OUT_SAM)
    {
        g_output_error = "unknown output format";
        return LX_EINVAL;
    }
    lx_output_options opt;
    lx_output_options_default(&opt);
    if (opt_in)
        opt = *opt_in;
    std::vector<int> cols;
    bool             tags[kNumSamTags] = {};
    if (format == LX_OUT_SAM)
        return resolveTags(opt.sam_tags, tags) ? LX_OK : LX_EINVAL;
    return resolveColumns(opt.columns, cols) ? LX_OK : LX_EINVAL;
}

int lx_write_records(char const * path, int format, int write_header, char const * program, lx_blast_match const * m,
                     uint64_t n, uint8_t const * ops, lx_seq_names const * names, uint8_t const * q_res_ascii,
                     uint64_t const * q_ascii_off)
{
    return lx_write_records_ex(path, format, write_header, program, m, n, ops, names, q_res_ascii, q_ascii_off, nullptr);
}

int lx_write_records_ex(char const * path, int format, int write_header, char const * program, lx_blast_match const * m,
                        uint64_t n, uint8_t const * ops, lx_seq_names const * names, uint8_t const * q_res_ascii,
                        uint64_t const * q_ascii_off, lx_output_options const * opt_in)
{
    g_output_error.clear();
    if (!path || !names || (!m && n) || !program)
        return LX_EINVAL;
    lx_output_options opt;
    lx_output_options_default(&opt);
    if (opt_in)
        opt = *opt_in;
    bool const isN    = std::strcmp(program, "blastn") == 0;
    bool const qTrans = std::strcmp(program, "blastx") == 0 || std::strcmp(program, "tblastx") == 0;  // qIsTranslated
    bool const sTrans = std::strcmp(program, "tblastn") == 0 || std::strcmp(program, "tblastx") == 0; // sIsTranslated
    if (!isN && !qTrans && !sTrans && std::strcmp(program, "blastp") != 0)
        return LX_EINVAL;
    // what the caller asked for is resolved before the file is touched (the reference fails while parsing its options)
    std::vector<int> cols;
    bool             tags[kNumSamTags] = {};
    if (format == LX_OUT_SAM)
    {
        if (!resolveTags(opt.sam_tags, tags))
            return LX_EINVAL;
    }
    else if (!resolveColumns(opt.columns, cols))
        return LX_EINVAL;
    std::FILE * f = std::fopen(path, write_header ? "w" : "a");
    if (!f)
    {
        g_output_error = std::string("cannot open ") + path;
        return LX_EINVAL;
    }
    bool        io_ok = true;
    std::string upper(program);
    for (char & c : upper)
        c = (char)std::toupper((unsigned char)c);
    // taxonomy of a subject / of a query's record (NULL trees: "*" and 0, as for an index without taxonomy)
    auto taxIdsOf = [&](uint64_t n_sid) -> std::string
    {
        if (!opt.tax || n_sid >= opt.tax->n_s || opt.tax->s_tax_off[n_sid + 1] == opt.tax->s_tax_off[n_sid])
            return "*";
        std::string out;
        for (uint64_t x = opt.tax->s_tax_off[n_sid]; x < opt.tax->s_tax_off[n_sid + 1]; ++x)
            out += (out.empty() ? "" : ";") + std::to_string(opt.tax->s_tax_ids[x]);
        return out;
    };
    uint64_t lcaAt = 0; // (the pairs are in list order, like the records)
    auto lcaOf = [&](uint64_t n_qid) -> uint32_t
    {
        while (lcaAt < opt.n_lca && opt.lca_qid && opt.lca_qid[lcaAt] != n_qid)
            ++lcaAt;
        return (lcaAt < opt.n_lca && opt.lca_qid && opt.lca_tax) ? opt.lca_tax[lcaAt] : 0u;
    };
    // the protein sequence of a query frame (tags qs / OC): the query itself (BLASTP, TBLASTN), its translation (BLASTX, TBLASTX)
    uint64_t    protOfQid = ~0ull;
    std::string protFrames[6];
    auto frameProtein = [&](lx_blast_match const & b) -> std::string
    {
        if (!q_res_ascii || !q_ascii_off)
            return std::string();
        char const *   src  = reinterpret_cast<char const *>(q_res_ascii) + q_ascii_off[b.n_qid];
        uint64_t const qLen = names->q_lens[b.n_qid];
        if (!qTrans)
            return std::string(src, qLen);
        if (b.q_frame == 0 || std::abs((int)b.q_frame) > 3)
            return std::string();
        if (protOfQid != b.n_qid)
        {
            std::vector<uint8_t> nt(qLen), aa(2 * qLen + 8);
            for (uint64_t i = 0; i < qLen; ++i)
                switch (std::toupper((unsigned char)src[i]))
                {
                    case 'A': nt[i] = 0; break;
                    case 'C': nt[i] = 1; break;
                    case 'G': nt[i] = 2; break;
                    case 'T': case 'U': nt[i] = 4; break;
                    default: nt[i] = 3;
                }
            uint64_t fo[6], fl[6];
            if (lx_translate_six_frames(nt.data(), qLen, opt.genetic_code, aa.data(), aa.size(), fo, fl) != LX_OK)
                return std::string();
            for (int fr = 0; fr < 6; ++fr)
            {
                protFrames[fr].resize(fl[fr]);
                for (uint64_t i = 0; i < fl[fr]; ++i)
                    protFrames[fr][i] = lambda_amd::kSeqanOrder[std::min<uint8_t>(aa[fo[fr] + i], 26)];
            }
            protOfQid = b.n_qid;
        }
        return protFrames[b.q_frame > 0 ? b.q_frame - 1 : 2 - b.q_frame];
    };

    if (format == LX_OUT_SAM)
    {
        if (write_header) // src/search_output.hpp:347-460
        {
            std::fprintf(f, "@HD\tVN:1.4\tGO:query\n");
            // --sam-with-refheader: seqan's writeHeader adds one @SQ per subject from the context ([UPSTREAM-RECALL] behind @HD)
            if (opt.sam_with_ref_header)
                for (uint64_t i = 0; i < names->n_s; ++i)
                    std::fprintf(f, "@SQ\tSN:%s\tLN:%llu\n", firstWord(names->s_ids[i]).c_str(), (unsigned long long)names->s_lens[i]);
            if (opt.version_to_output) // :391-399
                std::fprintf(f, "@PG\tID:lambda\tPN:lambda\tVN:%s\tCL:%s\n", opt.version ? opt.version : "", opt.command_line ? opt.command_line : "");
            std::fprintf(f, "@CO\tLambda is a high performance BLAST compatible local aligner, please see http://seqan.de/lambda "
                            "for more information.\n");
            std::fprintf(f, "@CO\tSAM/BAM dialect documentation is available here: https://github.com/seqan/lambda/wiki/Output-Formats\n");
            std::fprintf(f, "@CO\tIf you use any results found by Lambda, please cite Hauswedell et al. (2014) doi: "
                            "10.1093/bioinformatics/btu439\n");
            std::string tagLine = "Optional tags as follow"; // :424-437
            for (int t = 0; t < kNumSamTags; ++t)
                if (tags[t])
                    tagLine += std::string("\t") + kSamTags[t].key + ":" + kSamTags[t].desc;
            std::fprintf(f, "@CO\t%s\n", tagLine.c_str());
        }
        for (uint64_t lo = 0; lo < n;)
        {
            uint64_t hi = lo + 1;
            while (hi < n && m[hi].n_qid == m[lo].n_qid)
                ++hi;
            uint32_t const lca = (tags[kTagLcaTaxId] || tags[kTagLcaId]) ? lcaOf(m[lo].n_qid) : 0u;
            for (uint64_t k = lo; k < hi; ++k)
            {
                lx_blast_match const & b = m[k];
                if (b.n_qid >= names->n_q || b.n_sid >= names->n_s)
                {
                    std::fclose(f);
                    return LX_EINVAL;
                }
                int const         flag = ((k == lo) ? 0 : 256) | (b.q_frame < 0 ? 16 : 0); // secondary (:505, :723), RC (:506-507)
                std::string const qn = firstWord(names->q_ids[b.n_qid]), sn = firstWord(names->s_ids[b.n_sid]);
                bool const        hard = opt.sam_hard_clip != 0;
                // the DNA cigar: every program but BLASTP / TBLASTN, whose query is protein (:513-531)
                std::string cigar = "*", seq = "*", protCigar = "*";
                if (isN || qTrans)
                    cigar = cigarOf(b, ops, names->q_lens[b.n_qid], hard, qTrans);
                if (tags[kTagProtCigar] && !isN) // OC: protein-space cigar, clips like the DNA one, never reversed (:197-298)
                    protCigar = protCigarOf(b, ops, names->q_lens[b.n_qid], hard, qTrans);
                // --sam-bam-seq (:533-553): always, never, or once per query region and frame
                bool writeSeq = opt.sam_seq >= LX_SAM_SEQ_ALWAYS;
                if (opt.sam_seq == LX_SAM_SEQ_UNIQ)
                    writeSeq = (k == lo) || b.q_frame != m[k - 1].q_frame || b.q_start != m[k - 1].q_start || b.q_end != m[k - 1].q_end;
                uint64_t const qLen = names->q_lens[b.n_qid];
                if (isN && writeSeq && q_res_ascii && q_ascii_off)
                {
                    // the aligned sequence is the frame's (the reverse complement on the minus strand); hard clips keep only the
                    // matched part of it (:555-582)
                    seq.assign(reinterpret_cast<char const *>(q_res_ascii) + q_ascii_off[b.n_qid], qLen);
                    if (b.q_frame < 0)
                        reverseComplementAscii(seq);
                    if (hard)
                        seq = b.q_end <= qLen && b.q_start <= b.q_end ? seq.substr(b.q_start, b.q_end - b.q_start) : std::string("*");
                }
                else if (qTrans && writeSeq && q_res_ascii && q_ascii_off && b.q_frame != 0)
                {
                    // _untranslateSequence (:84-109, :583-598): protein [a, e) of frame f covers the nucleotides
                    // [3a + |f| - 1, 3e + |f| - 1) of the strand that was read -- from the read's start on the plus strand,
                    // from its end (then reverse-complemented) on the minus strand; soft clips keep the whole frame
                    uint64_t const fc = (uint64_t)std::abs((int)b.q_frame) - 1, L = (qLen - fc) / 3;
                    uint64_t const a = hard ? b.q_start : 0, e = hard ? b.q_end : L;
                    char const *   src = reinterpret_cast<char const *>(q_res_ascii) + q_ascii_off[b.n_qid];
                    if (e > L || a > e)
                        seq = "*";
                    else if (b.q_frame > 0)
                        seq.assign(src + 3 * a + fc, 3 * (e - a));
                    else
                    {
                        seq.assign(src + (qLen - (3 * e + fc)), 3 * (e - a));
                        reverseComplementAscii(seq);
                    }
                } // BLASTP / TBLASTN: the query is protein -- no SEQ (:599)
                if (seq.empty())
                    seq = "*";
                // POS: a translated subject is reported in nucleotide space (:493-499; the minus-strand branch there
                // subtracts from the QUERY length -- restated as it stands)
                uint64_t pos = b.s_start;
                if (sTrans)
                {
                    pos = b.s_start * 3 + (uint64_t)std::abs((int)b.s_frame) - (b.s_frame != 0 ? 1 : 0);
                    if (b.s_frame < 0)
                        pos = qLen - pos;
                }
                std::fprintf(f, "%s\t%d\t%s\t%llu\t255\t%s\t*\t0\t0\t%s\t*", qn.c_str(), flag, sn.c_str(),
                             (unsigned long long)(pos + 1), cigar.c_str(), seq.c_str());
                // tags in the order myWriteRecord appends them (:601-719), with the widths it casts to
                if (tags[kTagEValue])
                    std::fprintf(f, "\tae:f:%g", (double)(float)b.e_value);
                if (tags[kTagBitScore])
                    std::fprintf(f, "\tAS:i:%u", (unsigned)(uint16_t)b.bit_score);
                if (tags[kTagScore])
                    std::fprintf(f, "\tar:i:%u", (unsigned)(uint8_t)b.score); // (uint8_t in the reference, :612-616: scores beyond 255 wrap)
                if (tags[kTagPIdent])
                    std::fprintf(f, "\tai:i:%u", (unsigned)(uint8_t)b.identity);
                if (tags[kTagPPos])
                    std::fprintf(f, "\tap:i:%u", (unsigned)(uint16_t)(b.alignment_length ? 100.0 * b.num_positives / b.alignment_length : 0.0));
                if (tags[kTagQFrame])
                    std::fprintf(f, "\tqf:i:%d", (int)(int8_t)b.q_frame);
                if (tags[kTagSFrame])
                    std::fprintf(f, "\tsf:i:%d", (int)(int8_t)b.s_frame);
                if (tags[kTagSTaxIds])
                    std::fprintf(f, "\tst:Z:%s", taxIdsOf(b.n_sid).c_str());
                if (tags[kTagLcaId])
                    std::fprintf(f, "\tls:Z:%s", (opt.tax_names && opt.tax && lca < opt.tax->n_taxa && opt.tax_names[lca]) ? opt.tax_names[lca] : "*");
                if (tags[kTagLcaTaxId])
                    std::fprintf(f, "\tlt:i:%u", lca);
                if (tags[kTagProtSeq]) // :669-689
                {
                    std::string prot = (isN || !writeSeq) ? std::string() : frameProtein(b);
                    if (!prot.empty() && hard)
                        prot = b.q_end <= prot.size() && b.q_start <= b.q_end ? prot.substr(b.q_start, b.q_end - b.q_start) : std::string();
                    std::fprintf(f, "\tqs:Z:%s", prot.empty() ? "*" : prot.c_str());
                }
                if (tags[kTagProtCigar])
                    std::fprintf(f, "\tOC:Z:%s", protCigar.c_str());
                if (tags[kTagEditDistance])
                    std::fprintf(f, "\tNM:i:%u", (unsigned)(b.alignment_length - b.num_matches));
                if (tags[kTagMatchCount])
                    std::fprintf(f, "\tIH:i:%u", (unsigned)(hi - lo));
                std::fputc('\n', f);
            }
            lo = hi;
        }
    }
    else
    {
        bool const comments = format == LX_OUT_BLAST_TAB_COMMENTS;
        // "# <PROGRAM> 2.2.26+", followed by lambda's tag when the version goes to the output file (src/search_output.hpp:313-345)
        std::string versionLine = upper + " 2.2.26+";
        if (opt.version_to_output)
            versionLine += std::string(" [created by LAMBDA") + (opt.version ? std::string("-") + opt.version : std::string()) +
                           ", see http://seqan.de/lambda and please cite correctly in your academic work]";
        std::string fields;
        for (int c : cols)
            fields += (fields.empty() ? "" : ", ") + std::string(kColumns[c].desc);
        for (uint64_t k = 0; k < n; ++k)
            if (m[k].n_qid >= names->n_q || m[k].n_sid >= names->n_s)
            {
                std::fclose(f);
                return LX_EINVAL;
            }
        for (uint64_t lo = 0; lo < n;)
        {
            uint64_t hi = lo + 1;
            while (hi < n && m[hi].n_qid == m[lo].n_qid)
                ++hi;
            if (comments)
            {
                std::fprintf(f, "# %s\n", versionLine.c_str());
                std::fprintf(f, "# Query: %s\n# Database: %s\n", names->q_ids[m[lo].n_qid], opt.db_name ? opt.db_name : "lambda_ext");
                std::fprintf(f, "# Fields: %s\n# %llu hits found\n", fields.c_str(), (unsigned long long)(hi - lo));
            }
            uint32_t lca = 0;
            for (int c : cols)
                if (c == kColLcaTaxId)
                    lca = lcaOf(m[lo].n_qid);
            // one record -> one line, appended to `out` (the same text whichever thread makes it)
            auto formatRecord = [&](lx_blast_match const & b, uint32_t lca, std::string & out)
            {
                char tmp[64];
                auto put = [&](char const * fmt, auto... args) { out.append(tmp, (size_t)std::snprintf(tmp, sizeof(tmp), fmt, args...)); };
                // 1-based inclusive; a hit of the reverse-complemented query is reported BLAST-style on the forward query
                // strand with descending subject coordinates
                unsigned long long qs = b.q_start + 1, qe = b.q_end, ss = b.s_start + 1, se = b.s_end;
                // a translated side is reported in nucleotide positions of the original sequence ([UPSTREAM-RECALL]
                // seqan blast module, _untranslatePositions): protein [a, e) of frame f covers the nucleotides
                // [3a + |f| - 1, 3e + |f| - 1) of the strand that was read; on the minus strand the positions are mirrored
                // and start > end
                auto untranslate = [](uint64_t a, uint64_t e, int frame, uint64_t len, unsigned long long & first, unsigned long long & last)
                {
                    uint64_t const shift = (uint64_t)std::abs(frame) - 1;
                    uint64_t const lo = 3 * a + shift, hi = 3 * e + shift;
                    if (frame >= 0)
                    {
                        first = lo + 1;
                        last  = hi;
                    }
                    else
                    {
                        first = len - lo;
                        last  = len - hi + 1;
                    }
                };
                if (qTrans)
                    untranslate(b.q_start, b.q_end, b.q_frame, names->q_lens[b.n_qid], qs, qe);
                else if (b.q_frame < 0)
                {
                    unsigned long long const ql = names->q_lens[b.n_qid];
                    qs = ql - b.q_end + 1;
                    qe = ql - b.q_start;
                    std::swap(ss, se);
                }
                if (sTrans)
                    untranslate(b.s_start, b.s_end, b.s_frame, names->s_lens[b.n_sid], ss, se);
                bool first = true;
                for (int c : cols)
                {
                    if (!first)
                        out.push_back('\t');
                    first = false;
                    switch (c)
                    {
                        case kColQSeqId: out += firstWord(names->q_ids[b.n_qid]).c_str(); break;
                        case kColSSeqId: out += firstWord(names->s_ids[b.n_sid]).c_str(); break;
                        case kColQLen: put("%llu", (unsigned long long)names->q_lens[b.n_qid]); break;
                        case kColSLen: put("%llu", (unsigned long long)names->s_lens[b.n_sid]); break;
                        case kColQStart: put("%llu", qs); break;
                        case kColQEnd: put("%llu", qe); break;
                        case kColSStart: put("%llu", ss); break;
                        case kColSEnd: put("%llu", se); break;
                        case kColEValue: put("%.1e", b.e_value); break;
                        case kColBitScore: put("%.1f", b.bit_score); break;
                        case kColScore: put("%d", b.score); break;
                        case kColLength: put("%d", b.alignment_length); break;
                        case kColPIdent: put("%.2f", (double)b.identity); break;
                        case kColNIdent: put("%d", b.num_matches); break;
                        case kColMismatch: put("%d", b.num_mismatches); break;
                        case kColPositive: put("%d", b.num_positives); break;
                        case kColGapOpen: put("%d", b.num_gap_opens); break;
                        case kColGaps: put("%d", b.num_gap_opens + b.num_gap_extensions); break; // every gap character
                        case kColPPos: put("%.2f", b.alignment_length ? 100.0 * b.num_positives / b.alignment_length : 0.0); break;
                        case kColFrames: put("%d/%d", (int)b.q_frame, (int)b.s_frame); break;
                        case kColQFrame: put("%d", (int)b.q_frame); break;
                        case kColSFrame: put("%d", (int)b.s_frame); break;
                        case kColSTaxIds: out += taxIdsOf(b.n_sid).c_str(); break;
                        case kColLcaTaxId: put("%u", lca); break;
                        default: break;
                    }
                }
                out.push_back('\n');
            };
            bool hasLca = false;
            for (int c : cols)
                hasLca = hasLca || c == kColLcaTaxId;
            if (!comments && !hasLca && lo == 0 && n >= 65536 && lxi::pool_width() > 1)
            {
                // a plain table of many records: the lines are made on the library's host threads, piece by piece, and written in order
                unsigned const           nt = lxi::pool_width();
                std::vector<std::string> piece(nt);
                std::vector<uint8_t>     failed(nt, 0);
                uint64_t const           step = (n + nt - 1) / nt;
                lxi::pool_run(nt,
                              [&](unsigned t)
                              {
                                  try // (an exception must not leave a pool thread: out of memory is reported below)
                                  {
                                      std::string & out = piece[t];
                                      out.reserve((size_t)(std::min(n, (t + 1) * step) - std::min(n, t * step)) * 96);
                                      for (uint64_t k = std::min(n, t * step); k < std::min(n, (t + 1) * step); ++k)
                                          formatRecord(m[k], 0u, out);
                                  }
                                  catch (...)
                                  {
                                      failed[t] = 1;
                                  }
                              });
                for (unsigned t = 0; t < nt; ++t)
                {
                    if (failed[t])
                    {
                        std::fclose(f);
                        g_output_error = "out of memory while formatting the records";
                        return LX_ENOMEM;
                    }
                    io_ok = (std::fwrite(piece[t].data(), 1, piece[t].size(), f) == piece[t].size()) && io_ok;
                }
                break;
            }
            std::string line;
            for (uint64_t k = lo; k < hi; ++k)
            {
                line.clear();
                formatRecord(m[k], lca, line);
                io_ok = (std::fwrite(line.data(), 1, line.size(), f) == line.size()) && io_ok;
            }
            lo = hi;
        }
    }
    // (the fprintf paths set the stream's error indicator: one look at the end covers them)
    io_ok = !std::ferror(f) && io_ok;
    io_ok = (std::fclose(f) == 0) && io_ok;
    if (!io_ok)
    {
        g_output_error = std::string("error while writing ") + path + " (disk full?)";
        return LX_EINVAL;
    }
    return LX_OK;
}

} // extern "C"
