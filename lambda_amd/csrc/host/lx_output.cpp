// lx_output.cpp -- record post-processing and BLAST-tabular / SAM writers.
//
// Host-side C++ mirror of
//   _writeRecord          /root/reference/src/search_algo.hpp:820-913   (sort, dedupe, bit-score order, top-N)
//   myWriteHeader         src/search_output.hpp:305-461
//   myWriteRecord         src/search_output.hpp:463-733                  (tabular via seqan::writeRecord; SAM records)
//   blastMatchOneCigar    src/search_output.hpp:115-194                  (soft clips, no frame clips for untranslated)
// Scope: BLASTP, BLASTN in all formats; BLASTX / TBLASTN / TBLASTX in the tabular formats (nucleotide coordinates).  The number formats of the tabular columns are SeqAn2's
// (source absent): [UPSTREAM-RECALL] pident %.2f, evalue %.1e, bitscore %.1f, 1-based inclusive positions.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "../../../include/lambda_ext.h"

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

int lx_write_records(char const * path, int format, int write_header, char const * program, lx_blast_match const * m,
                     uint64_t n, uint8_t const * ops, lx_seq_names const * names, uint8_t const * q_res_ascii,
                     uint64_t const * q_ascii_off)
{
    if (!path || !names || (!m && n) || !program)
        return LX_EINVAL;
    bool const isN    = std::strcmp(program, "blastn") == 0;
    bool const qTrans = std::strcmp(program, "blastx") == 0 || std::strcmp(program, "tblastx") == 0;  // qIsTranslated
    bool const sTrans = std::strcmp(program, "tblastn") == 0 || std::strcmp(program, "tblastx") == 0; // sIsTranslated
    if (!isN && !qTrans && !sTrans && std::strcmp(program, "blastp") != 0)
        return LX_EINVAL;
    std::FILE * f = std::fopen(path, write_header ? "w" : "a");
    if (!f)
        return LX_EINVAL;
    std::string upper(program);
    for (char & c : upper)
        c = (char)std::toupper((unsigned char)c);

    if (format == LX_OUT_SAM)
    {
        if (write_header) // src/search_output.hpp:382-460 (SAM without reference header records)
        {
            std::fprintf(f, "@HD\tVN:1.4\tGO:query\n");
            std::fprintf(f, "@CO\tLambda is a high performance BLAST compatible local aligner, please see http://seqan.de/lambda "
                            "for more information.\n");
            std::fprintf(f, "@CO\tSAM/BAM dialect documentation is available here: https://github.com/seqan/lambda/wiki/Output-Formats\n");
            std::fprintf(f, "@CO\tIf you use any results found by Lambda, please cite Hauswedell et al. (2014) doi: "
                            "10.1093/bioinformatics/btu439\n");
            std::fprintf(f, "@CO\tOptional tags as follow\tAS:bit score\tNM:edit distance (in protein space unless BLASTN)\tae:expect "
                            "value\tai:%% identity (in protein space unless BLASTN) \tqf:query frame\n");
        }
        for (uint64_t lo = 0; lo < n;)
        {
            uint64_t hi = lo + 1;
            while (hi < n && m[hi].n_qid == m[lo].n_qid)
                ++hi;
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
                // BLASTP: no DNA cigar and no SEQ ("*"); BLASTN: cigar with soft clips and, for the first record of a
                // query region, the read sequence (samBamSeq = uniq, :536-553)
                std::string cigar = "*", seq = "*";
                // samBamSeq = uniq (:536-553): the sequence once per query region and frame
                bool const writeSeq = (k == lo) || b.q_frame != m[k - 1].q_frame || b.q_start != m[k - 1].q_start ||
                                      b.q_end != m[k - 1].q_end;
                uint64_t const qLen = names->q_lens[b.n_qid];
                if (isN)
                {
                    cigar = cigarOf(b, ops, qLen, false);
                    if (writeSeq && q_res_ascii && q_ascii_off)
                    {
                        seq.assign(reinterpret_cast<char const *>(q_res_ascii) + q_ascii_off[b.n_qid], qLen);
                        if (b.q_frame < 0) // the aligned sequence is the reverse-complement frame
                            reverseComplementAscii(seq);
                    }
                }
                else if (qTrans)
                {
                    // nucleotide-space CIGAR (:528-531) and the part of the untranslated read that the frame covers,
                    // soft-clip mode of _untranslateSequence (:84-109, :590-598): [|f| - 1, 3 L + |f| - 1) from the read's
                    // start on the plus strand, from its end (then reverse-complemented) on the minus strand
                    cigar = cigarOf(b, ops, qLen, false, true);
                    if (writeSeq && q_res_ascii && q_ascii_off && b.q_frame != 0)
                    {
                        uint64_t const fc = (uint64_t)std::abs((int)b.q_frame) - 1, L = (qLen - fc) / 3;
                        char const *   src = reinterpret_cast<char const *>(q_res_ascii) + q_ascii_off[b.n_qid];
                        if (b.q_frame > 0)
                            seq.assign(src + fc, 3 * L);
                        else
                        {
                            seq.assign(src + (qLen - (3 * L + fc)), 3 * L);
                            reverseComplementAscii(seq);
                        }
                    }
                } // BLASTP / TBLASTN: the query is protein -- no DNA cigar (:526-531), no SEQ (:599)
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
                // tags in the order of myWriteRecord: ae, AS, ai, qf, NM (:611-716)
                std::fprintf(f, "\tae:f:%g\tAS:i:%u\tai:i:%u\tqf:i:%d\tNM:i:%u\n", (double)(float)b.e_value,
                             (unsigned)(uint16_t)b.bit_score, (unsigned)(uint8_t)b.identity, (int)b.q_frame,
                             (unsigned)(b.alignment_length - b.num_matches));
            }
            lo = hi;
        }
    }
    else
    {
        bool const comments = format == LX_OUT_BLAST_TAB_COMMENTS;
        for (uint64_t lo = 0; lo < n;)
        {
            uint64_t hi = lo + 1;
            while (hi < n && m[hi].n_qid == m[lo].n_qid)
                ++hi;
            if (comments)
            {
                std::fprintf(f, "# %s 2.2.26+ [created by LAMBDA, see http://seqan.de/lambda and please cite correctly in your academic work]\n",
                             upper.c_str());
                std::fprintf(f, "# Query: %s\n# Database: %s\n", names->q_ids[m[lo].n_qid], "lambda_ext");
                std::fprintf(f, "# Fields: query id, subject id, %% identity, alignment length, mismatches, gap opens, q. start, "
                                "q. end, s. start, s. end, evalue, bit score\n# %llu hits found\n",
                             (unsigned long long)(hi - lo));
            }
            for (uint64_t k = lo; k < hi; ++k)
            {
                lx_blast_match const & b = m[k];
                if (b.n_qid >= names->n_q || b.n_sid >= names->n_s)
                {
                    std::fclose(f);
                    return LX_EINVAL;
                }
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
                std::fprintf(f, "%s\t%s\t%.2f\t%d\t%d\t%d\t%llu\t%llu\t%llu\t%llu\t%.1e\t%.1f\n",
                             firstWord(names->q_ids[b.n_qid]).c_str(), firstWord(names->s_ids[b.n_sid]).c_str(),
                             (double)b.identity, b.alignment_length, b.num_mismatches, b.num_gap_opens, qs, qe, ss, se,
                             b.e_value, b.bit_score);
            }
            lo = hi;
        }
    }
    std::fclose(f);
    return LX_OK;
}

} // extern "C"
