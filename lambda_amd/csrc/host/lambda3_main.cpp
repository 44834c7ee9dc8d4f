// lambda3_main.cpp -- minimal `lambda3 searchp|searchn|searchbs` front end around liblambda_ext (SURVEY.md section 7 step 9,
// section 8f rows N2/N3): plumbing for BASELINE.json configs[0], not a port of the reference's driver.
//
//   lambda3 searchp -q queries.fasta -d db.fasta -o out.m8 [-e 1e-2] [-n 25] [--seed-length 10] [--seed-offset 5]
//                   [--devices 0,1,...] [-t THREADS] [--table gpu|host|auto] [--seeding gpu|host]
//
// What it mirrors from the reference, and what it does not:
//   * subcommand split and the search command line (src/lambda.cpp:30-118; src/search_options.hpp:143-816): -q, -o (format from
//     the extension, :684-710), -e, --bit-score, --percent-identity, -n, the seeding options (--seed-length/-offset/-delta[0],
//     --search0, --adaptive-seeding, --seed-half-exact), --pre-scoring[-threshold], the scoring options (-s 45|62|80 for
//     proteins, --score-match/-mismatch for nucleotides, --score-gap, --score-gap-open), the profiles (-p fast | sensitive |
//     pairs-default | pairs-sensitive: they overwrite the seeding options, :634-681), the output options (--output-columns,
//     --sam-bam-tags/-seq/-clip, --sam-with-refheader, --version-to-outputfile), -g, --input-alphabet;
//   * the thread split of realMain (src/search.cpp:379-385): one worker per entry of --devices (default: all visible devices), each
//     with its own handle -- one LocalDataHolder per thread there, one lx_handle = device + stream here --, the queries dealt to
//     them in contiguous ranges, the records concatenated in range order before _writeRecord; -t host threads (default: what
//     the machine grants) build the word table and are shared out among the workers for the seeding of their reads;
//   * searchbs (src/lambda.cpp:103; domain_t::bisulfite, src/search_options.hpp:127, :261-264, :328): four query frames
//     (strand x bisulfite duplicate), two subject frames, the 6-letter reduction of src/view_reduce_to_bisulfite.hpp for
//     seeding, both scoring schemes (src/bisulfite_scoring.hpp:67-93), iterateMatches' bisulfite branch (:1367-1379);
//   * the per-batch flow of realMain (src/search.cpp:389-459): seeding -> seedLooksPromising -> iterateMatches ->
//     writeRecords, with the three middle stages on the GPU through the C ABI;
//   * mkindexp | mkindexn | mkindexbs (src/lambda.cpp:86-88, src/mkindex_options.hpp:96-262: -d, -i with the .lba / .lta name and the
//     refusal to overwrite, -r, -g, --input-alphabet, --truncate-ids, --db-index-type, -t) and `search* -i INDEX`: the index file
//     holds what the reference's holds (the options that fix the types, ids, sequences in the translated alphabet; no taxonomy)
//     with this front end's word table in the FM-index's place -- its own format, not the reference's cereal archive; the
//     searches take the domain, the reduction and the subjects' genetic code from it and refuse an index of another domain
//     with the reference's messages (src/search.cpp:157-207);
//   * NOT the FM-index: the reference searches an index built by `lambda3 mkindexp` (fmindex-collection, absent here,
//     out of scope).  This front end answers the seeding stage's questions from a
//     sorted table of reduced words (host/lx_seeding.hpp), read from its index file (-i) or made in memory from FASTA (-d) -- with the reference's seeding semantics: Li-10 reduction,
//     exact seeds 10/5 first, then (queries without a result) half-exact seeds 11/3 with one substitution in the second
//     half, adaptive elongation, over-abundant seeds dropped, seedLooksPromising per hit (src/search_algo.hpp:426-762,
//     :1391-1457; defaults src/search_options.hpp:309-337).  Everything after seeding follows the reference too.
#include <sched.h>

#include <algorithm>
#include <array>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "blast_stats.hpp"
#include "lambda_ext.hpp"
#include "lx_seeding.hpp"
#include "lx_seeding_gpu.hpp"
#include "scoring_tables.hpp"

namespace
{

struct SeqSet
{
    std::vector<std::string> ids;
    std::vector<uint8_t>     res;   // ranks, concatenated (frame-expanded for nucleotide queries)
    std::vector<uint64_t>    off, len;
    std::string              ascii; // original letters of the untranslated sequences, concatenated
    std::vector<uint64_t>    ascii_off, orig_len;
};

uint8_t aaRank(char c) // SeqAn2 AminoAcid rank (src/seqan2_to_biocpp.hpp:352-366)
{
    static auto const table = []()
    {
        std::array<uint8_t, 256> t;
        for (int ch = 0; ch < 256; ++ch)
        {
            char const * p = std::strchr(lambda_amd::kSeqanOrder, std::toupper(ch));
            t[(size_t)ch]  = (ch != 0 && p && *p) ? (uint8_t)(p - lambda_amd::kSeqanOrder) : 25; // unknown -> X
        }
        return t;
    }();
    return table[(unsigned char)c];
}

uint8_t dnaRank(char c) // BioC++ dna5 rank A,C,G,N,T (Simple-scored alphabets pass it through, :392-393)
{
    static auto const table = []()
    {
        std::array<uint8_t, 256> t;
        t.fill(3);
        t['A'] = t['a'] = 0, t['C'] = t['c'] = 1, t['G'] = t['g'] = 2, t['T'] = t['t'] = t['U'] = t['u'] = 4;
        return t;
    }();
    return table[(unsigned char)c];
}

uint8_t dnaRankSeqan(char c) // SeqAn Dna5 rank A,C,G,T,N: the bisulfite schemes are matrices over it (src/bisulfite_scoring.hpp:54-93)
{
    static auto const table = []()
    {
        std::array<uint8_t, 256> t;
        t.fill(4);
        t['A'] = t['a'] = 0, t['C'] = t['c'] = 1, t['G'] = t['g'] = 2, t['T'] = t['t'] = t['U'] = t['u'] = 3;
        return t;
    }();
    return table[(unsigned char)c];
}

// translate = six protein frames per nucleotide sequence (BLASTX queries: qryNumFrames = 6, translate_join)
// bsFrames: 0 = none; 1 = bisulfite subjects (every sequence twice: views::duplicate, src/shared_definitions.hpp:249-250);
// 2 = bisulfite queries (strand, strand, reverse complement, reverse complement: add_reverse_complement | duplicate, :260-261)
void readFasta(std::string const & path, bool protein, bool addRevComp, SeqSet & out, bool translate = false, int geneticCode = 1,
               int bsFrames = 0)
{
    std::ifstream in(path);
    if (!in)
        throw std::runtime_error("cannot open " + path);
    std::string line, cur;
    auto flush = [&]()
    {
        if (out.ids.size() == out.orig_len.size())
            return;
        out.ascii_off.push_back(out.ascii.size());
        out.orig_len.push_back(cur.size());
        out.ascii += cur;
        if (translate)
        {
            std::vector<uint8_t> nt(cur.size());
            for (size_t i = 0; i < cur.size(); ++i)
                nt[i] = dnaRank(cur[i]);
            std::vector<uint8_t> aa(2 * cur.size() + 8);
            uint64_t             fo[6], fl[6];
            if (lx_translate_six_frames(nt.data(), nt.size(), geneticCode, aa.data(), aa.size(), fo, fl) != LX_OK)
                throw std::runtime_error("translation failed for " + out.ids.back() + " (genetic code " + std::to_string(geneticCode) + ")");
            for (int f = 0; f < 6; ++f)
            {
                out.off.push_back(out.res.size());
                out.len.push_back(fl[f]);
                out.res.insert(out.res.end(), aa.begin() + fo[f], aa.begin() + fo[f] + fl[f]);
            }
            cur.clear();
            return;
        }
        if (bsFrames)
        {
            auto push = [&](bool rc)
            {
                out.off.push_back(out.res.size());
                out.len.push_back(cur.size());
                static uint8_t const comp[5] = {3, 2, 1, 0, 4};
                size_t const at = out.res.size(), n = cur.size();
                out.res.resize(at + n);
                if (!rc)
                    for (size_t k = 0; k < n; ++k)
                        out.res[at + k] = dnaRankSeqan(cur[k]);
                else
                    for (size_t k = 0; k < n; ++k)
                        out.res[at + k] = comp[dnaRankSeqan(cur[n - 1 - k])];
            };
            push(false);
            push(false);
            if (bsFrames == 2)
            {
                push(true);
                push(true);
            }
            cur.clear();
            return;
        }
        out.off.push_back(out.res.size());
        out.len.push_back(cur.size());
        {
            size_t const at = out.res.size();
            out.res.resize(at + cur.size());
            if (protein)
                for (size_t k = 0; k < cur.size(); ++k)
                    out.res[at + k] = aaRank(cur[k]);
            else
                for (size_t k = 0; k < cur.size(); ++k)
                    out.res[at + k] = dnaRank(cur[k]);
        }
        if (addRevComp) // qryNumFrames = 2 for BLASTN (src/search_datastructures.hpp:380-385)
        {
            out.off.push_back(out.res.size());
            out.len.push_back(cur.size());
            static uint8_t const comp[5] = {4, 2, 1, 3, 0};
            size_t const at = out.res.size(), n = cur.size();
            out.res.resize(at + n);
            for (size_t k = 0; k < n; ++k) // (the forward frame's ranks stand right before)
                out.res[at + k] = comp[out.res[at - 1 - k]];
        }
        cur.clear();
    };
    while (std::getline(in, line))
    {
        if (!line.empty() && line.back() == '\r')
            line.pop_back();
        if (line.empty())
            continue;
        if (line[0] == '>')
        {
            flush();
            out.ids.push_back(line.substr(1));
        }
        else if (line.find_first_of(" \t\v\f") == std::string::npos)
            cur += line;
        else
            for (char c : line)
                if (!std::isspace((unsigned char)c))
                    cur.push_back(c);
    }
    flush();
}

struct Options
{
    std::string cmd, query, db, output = "output.m8";
    double      maxEValue   = 1e-2; // src/search_options.hpp:97
    uint64_t    maxMatches  = 25;   // :99
    int         seedLength  = 0, seedOffset = 0, seedDelta = 1;     // searchOpts  (:309-337)
    int         seedLength0 = 0, seedOffset0 = 0;                   // searchOpts0: the exact pre-search
    bool        search0 = true;       // iterativeSearch (:107, --search0)
    bool        adaptive = true, halfExact = true; // :75-76
    std::string reduction = "li10";   // mkindexp -r (src/mkindex_options.hpp:182-205)
    int         geneticCode = 1;      // :170
    int         preScoring  = 2;    // :104
    double      preScoringThresh = 2.0;
    int         idCutOff    = 0;
    int         minBitScore = -1;   // :96, --bit-score
    int         seedDelta0  = 0;    // searchOpts0.maxSeedDist (:314, :322, :331)
    std::string profile     = "none"; // :107, -p
    int         scoringMethod = 62; // :89, -s (searchp)
    int         match = 2, misMatch = -3; // :93-94 (searchn / searchbs)
    int         gapOpen = -11, gapExtend = -1; // :91-92; -5 / -2 outside the protein domain (:293-294)
    // output (:224-379): columns of .m8 / .m9, tags / sequence / clipping of .sam
    std::string outputColumns = "std", samTags = "AS NM ae ai qf", samSeq = "uniq", samClip = "hard";
    bool        samWithRefHeader = false, versionToOutput = true;
    std::string commandLine;
    std::vector<int> devices;         // --devices (default: every visible device)
    std::string table       = "auto"; // --table gpu | host | auto: where the word table is made (auto: search* on the GPU, mkindex* on the host)
    std::string seeding     = "gpu";  // --seeding gpu | host: where search() runs (host/lx_seeding_gpu.hpp -- one lane per read, reads the
                                      // device declines go to the host --, host/lx_seeding.hpp on the -t threads)
    int         threads     = 0;    // -t host threads for the word table and the seeding (default: what the machine grants)
    // mkindex* (src/mkindex_options.hpp:96-262): -d the database (FASTA), -i the index file to write
    std::string index;                // -i of mkindex* (default: DATABASE.lba, :132, :249-250)
    std::string dbIndexType = "fm";   // --db-index-type fm | bifm (:157-165; recorded, the word table answers both)
    bool        truncateIds = false;  // --truncate-ids (:167-173)
    bool        geneticCodeGiven = false;
    std::string qryAlphabet = "auto"; // searchp: "aminoacid" = BLASTP, "dna5" = BLASTX, "auto" = decide from the letters
    std::string dbAlphabet  = "auto"; // searchp: "dna5" = six-frame translated subjects (TBLASTN / TBLASTX)
};

// the reference lets BioC++ detect the query alphabet (src/search_options.hpp); here: nucleotide if >= 90 % of the
// letters of the first sequences are ACGTUN
bool looksLikeDna(std::string const & path)
{
    std::ifstream in(path);
    std::string   line;
    uint64_t      nuc = 0, all = 0;
    while (std::getline(in, line) && all < 100000)
    {
        if (line.empty() || line[0] == '>')
            continue;
        for (char c : line)
        {
            if (std::isspace((unsigned char)c))
                continue;
            ++all;
            nuc += std::strchr("ACGTUNacgtun", c) != nullptr;
        }
    }
    return all > 0 && nuc * 10 >= all * 9;
}

Options parse(int argc, char ** argv)
{
    Options o;
    if (argc < 2)
        throw std::runtime_error("usage: lambda3 searchp|searchn|searchbs -q QUERY.fasta (-i DB.lba | -d DB.fasta) -o OUT.{m8,m9,sam} [-e EVALUE] [-n N] "
                                 "[--devices 0,1,...] [-t THREADS]\n       lambda3 mkindexp|mkindexn|mkindexbs -d DB.fasta [-i DB.lba] [-r li10|murphy10|none] [-g CODE] [-t THREADS]");
    o.cmd = argv[1];
    bool const mk = o.cmd == "mkindexp" || o.cmd == "mkindexn" || o.cmd == "mkindexbs";
    if (o.cmd != "searchp" && o.cmd != "searchn" && o.cmd != "searchbs" && !mk)
        throw std::runtime_error("unknown subcommand '" + o.cmd + "' (searchp, searchn, searchbs, mkindexp, mkindexn, mkindexbs: src/lambda.cpp:86-88)");
    bool const prot = o.cmd == "searchp" || o.cmd == "mkindexp", bs = o.cmd == "searchbs" || o.cmd == "mkindexbs";
    // per-domain defaults, src/search_options.hpp:309-337 (bisulfite: :261-264, :328-336)
    o.seedLength0      = prot ? 10 : bs ? 17 : 14;
    o.seedOffset0      = prot ? 5 : bs ? 10 : 9;
    o.seedLength       = prot ? 11 : bs ? 17 : 14;
    o.seedOffset       = prot ? 3 : bs ? 10 : 7;
    o.preScoringThresh = prot ? 2.0 : bs ? 1.5 : 1.4;
    if (bs)
        o.maxEValue = 1e-9;
    if (!prot)
        o.gapOpen = -5, o.gapExtend = -2;
    for (int i = 1; i < argc; ++i) // (the reference keeps the command line from the subcommand on, :106-109)
        o.commandLine += std::string(i > 1 ? " " : "") + argv[i];
    auto onOff = [](std::string const & v) // sharg's bool options take 0 / 1 / true / false; the help pages write ON / OFF
    {
        std::string u;
        for (char c : v)
            u += (char)std::toupper((unsigned char)c);
        if (u == "1" || u == "TRUE" || u == "ON")
            return true;
        if (u == "0" || u == "FALSE" || u == "OFF")
            return false;
        throw std::runtime_error("expected 0 / 1 / true / false / ON / OFF, got " + v);
    };
    auto inRange = [](std::string const & name, double v, double lo, double hi)
    {
        if (v < lo || v > hi)
            throw std::runtime_error("Value " + std::to_string(v) + " of option " + name + " is not in range [" + std::to_string(lo) + "," + std::to_string(hi) + "].");
    };
    for (int i = 2; i < argc; ++i)
    {
        std::string a = argv[i];
        auto        val = [&]() -> std::string
        {
            if (i + 1 >= argc)
                throw std::runtime_error("missing value for " + a);
            return argv[++i];
        };
        if (a == "-q" || a == "--query")
            o.query = val();
        else if (mk && (a == "-i" || a == "--index"))
            o.index = val();
        else if (mk && a == "--db-index-type")
        {
            o.dbIndexType = val();
            if (o.dbIndexType != "fm" && o.dbIndexType != "bifm")
                throw std::runtime_error("--db-index-type takes fm or bifm");
        }
        else if (mk && a == "--truncate-ids")
            o.truncateIds = true;
        else if (mk && (a == "-m" || a == "--acc-tax-map" || a == "-x" || a == "--tax-dump-dir"))
            throw std::runtime_error(a + ": the taxonomy part of the index (src/mkindex_options.hpp:113-128) is not built by this front end");
        else if (a == "-d" || a == "--database" || a == "-i" || a == "--index")
            o.db = val();
        else if (a == "-o" || a == "--output")
            o.output = val();
        else if (a == "-e" || a == "--e-value")
            o.maxEValue = std::stod(val());
        else if (a == "-n" || a == "--num-matches")
            o.maxMatches = std::stoull(val());
        else if (a == "--seed-length")
            o.seedLength = std::stoi(val());
        else if (a == "--seed-offset")
            o.seedOffset = std::stoi(val());
        else if (a == "--seed-delta")
            o.seedDelta = std::stoi(val());
        else if (a == "--seed-length0")
            o.seedLength0 = std::stoi(val());
        else if (a == "--seed-offset0")
            o.seedOffset0 = std::stoi(val());
        else if (a == "--seed-delta0")
            o.seedDelta0 = std::stoi(val());
        else if (a == "--search0")
            o.search0 = onOff(val());
        else if (a == "--adaptive-seeding")
            o.adaptive = onOff(val());
        else if (a == "--seed-half-exact")
            o.halfExact = onOff(val());
        else if (a == "--pre-scoring") // (the validator's range is 1..10, :488-497; 0 = off is what the description offers)
        {
            o.preScoring = std::stoi(val());
            inRange(a, o.preScoring, 0, 10);
        }
        else if (a == "--pre-scoring-threshold")
        {
            o.preScoringThresh = std::stod(val());
            inRange(a, o.preScoringThresh, 0, 20);
        }
        else if (a == "--bit-score")
        {
            o.minBitScore = std::stoi(val());
            inRange(a, o.minBitScore, -1, 1000);
        }
        else if ((a == "-s" || a == "--scoring-scheme") && prot) // :510-521
        {
            o.scoringMethod = std::stoi(val());
            if (o.scoringMethod != 45 && o.scoringMethod != 62 && o.scoringMethod != 80)
                throw std::runtime_error("--scoring-scheme takes 45, 62 or 80");
        }
        else if (a == "--score-match" && !prot) // :523-540
            o.match = std::stoi(val());
        else if (a == "--score-mismatch" && !prot)
            o.misMatch = std::stoi(val());
        else if (a == "--score-gap")
            o.gapExtend = std::stoi(val());
        else if (a == "--score-gap-open")
            o.gapOpen = std::stoi(val());
        else if (a == "-p" || a == "--profile")
        {
            o.profile = val();
            if (o.profile != "none" && o.profile != "fast" && o.profile != "sensitive" && o.profile != "pairs-default" && o.profile != "pairs-sensitive")
                throw std::runtime_error("--profile takes none, fast, sensitive, pairs-default or pairs-sensitive");
        }
        else if (a == "--output-columns")
            o.outputColumns = val();
        else if (a == "--sam-bam-tags")
            o.samTags = val();
        else if (a == "--sam-bam-seq")
        {
            o.samSeq = val();
            if (o.samSeq != "always" && o.samSeq != "uniq" && o.samSeq != "never")
                throw std::runtime_error("--sam-bam-seq takes always, uniq or never");
        }
        else if (a == "--sam-bam-clip")
        {
            o.samClip = val();
            if (o.samClip != "hard" && o.samClip != "soft")
                throw std::runtime_error("--sam-bam-clip takes hard or soft");
        }
        else if (a == "--sam-with-refheader")
            o.samWithRefHeader = onOff(val());
        else if (a == "--version-to-outputfile")
            o.versionToOutput = onOff(val());
        else if (a == "-r" || a == "--alphabet-reduction")
        {
            o.reduction = val();
            if (o.reduction != "none" && o.reduction != "murphy10" && o.reduction != "li10")
                throw std::runtime_error("--alphabet-reduction takes none, murphy10 or li10");
        }
        else if (a == "-g" || a == "--genetic-code")
        {
            o.geneticCode      = std::stoi(val());
            o.geneticCodeGiven = true;
        }
        else if (a == "--percent-identity")
            o.idCutOff = std::stoi(val());
        else if (a == "--device" || a == "--devices")
        {
            o.devices.clear();
            std::string const v = val();
            for (size_t at = 0; at <= v.size();)
            {
                size_t const e = std::min(v.find(',', at), v.size());
                if (e > at)
                    o.devices.push_back(std::stoi(v.substr(at, e - at)));
                at = e + 1;
            }
            if (o.devices.empty())
                throw std::runtime_error("--devices takes a comma-separated list of device numbers");
        }
        else if (a == "-t" || a == "--threads")
            o.threads = std::stoi(val());
        else if (a == "--table")
        {
            o.table = val();
            if (o.table != "gpu" && o.table != "host" && o.table != "auto")
                throw std::runtime_error("--table takes gpu, host or auto");
        }
        else if (a == "--seeding")
        {
            o.seeding = val();
            if (o.seeding != "host" && o.seeding != "gpu")
                throw std::runtime_error("--seeding takes host or gpu");
        }
        else if (a == "--db-alphabet")
        {
            o.dbAlphabet = val();
            if (o.dbAlphabet != "auto" && o.dbAlphabet != "dna5" && o.dbAlphabet != "aminoacid")
                throw std::runtime_error("--db-alphabet takes auto, dna5 or aminoacid");
        }
        else if (a == "-a" || a == "--query-alphabet" || a == "--input-alphabet") // (the reference's name is --input-alphabet, :172-185)
        {
            std::string & which = mk ? o.dbAlphabet : o.qryAlphabet; // (mkindexp: the database's, src/mkindex_options.hpp:189-197)
            which               = val();
            if (which != "auto" && which != "dna5" && which != "aminoacid")
                throw std::runtime_error("--input-alphabet takes auto, dna5 or aminoacid");
        }
        else if (a == "-v" || a == "--verbosity" || a == "--lazy-query")
            (void)val(); // accepted for command-line compatibility, no effect here
        else
            throw std::runtime_error("unknown option " + a);
    }
    if (mk)
    {
        if (o.db.empty())
            throw std::runtime_error("-d is required");
        if (o.index.empty())
            o.index = o.db + ".lba"; // :249-250
        auto ends = [&](char const * suf)
        { return o.index.size() >= std::strlen(suf) && o.index.compare(o.index.size() - std::strlen(suf), std::string::npos, suf) == 0; };
        if (!ends(".lba") && !ends(".lta")) // :133, :145
            throw std::runtime_error("the index file name must end in .lba or .lta");
        if (std::ifstream(o.index).good()) // :252-256
            throw std::runtime_error("ERROR: An output file already exists at " + o.index + "\n       Remove it, or choose a different location.");
        return o;
    }
    if (o.query.empty() || o.db.empty())
        throw std::runtime_error("-q and -d (a FASTA file) or -i (an index made by lambda3 mkindex*) are required");
    // "Setting a profile other than none always overwrites manually given command line arguments" (:563-566, applied :634-681)
    if (o.profile == "fast")
    {
        if (!prot)
        {
            o.search0   = false;
            o.seedDelta = 0;
            if (!bs)
                o.seedOffset = 9;
        }
        else
            o.seedLength0 = 12, o.seedOffset0 = 8, o.seedLength = 10, o.seedOffset = 5, o.seedDelta = 0;
    }
    else if (o.profile != "none") // sensitive, pairs-default, pairs-sensitive
    {
        if (prot)
            o.seedLength0 = 9, o.seedOffset0 = 4, o.seedLength = 8, o.seedOffset = 3, o.preScoring = 3, o.preScoringThresh = 1.9;
        else if (bs)
            o.seedLength0 = 16, o.seedOffset0 = 8, o.seedLength = 15, o.seedOffset = 10;
        else
            o.seedOffset0 = 3, o.seedOffset = 3;
        if (o.profile.rfind("pairs", 0) == 0)
            o.search0 = false;
        if (o.profile == "pairs-sensitive")
            --o.seedLength;
    }
    if (o.seedLength < 2 || o.seedLength0 < 2 || o.seedOffset < 1 || o.seedOffset0 < 1 || o.seedDelta < 0 || o.seedDelta > 3 || o.seedDelta0 < 0 ||
        o.seedDelta0 > 5)
        throw std::runtime_error("seed length / offset / delta out of range");
    {
        // the output format and what the writers are asked for, before anything is read or searched (:684-816: the reference fails
        // while it parses its options)
        auto ends = [&](char const * suf)
        { return o.output.size() >= std::strlen(suf) && o.output.compare(o.output.size() - std::strlen(suf), std::string::npos, suf) == 0; };
        int const fmt = ends(".m9") ? LX_OUT_BLAST_TAB_COMMENTS : ends(".sam") ? LX_OUT_SAM : ends(".m8") ? LX_OUT_BLAST_TAB : -1;
        if (fmt < 0)
            throw std::runtime_error("output format is chosen by the extension: .m8, .m9 or .sam"); // :684-710
        lx_output_options oo;
        lx_output_options_default(&oo);
        oo.columns  = o.outputColumns.c_str();
        oo.sam_tags = o.samTags.c_str();
        if (lx_check_output_options(fmt, &oo) != LX_OK)
            throw std::runtime_error(lx_last_output_error());
    }
    return o;
}

// ---- the index file of `lambda3 mkindex*`.  It holds what the reference's index_file holds (src/shared_definitions.hpp:343-379:
// the options that fix the types, the ids, the sequences in the translated alphabet) except the taxonomy, and this front end's
// word table in place of the FM-index.  NOT the reference's on-disk format: that is a cereal archive of fmindex-collection
// objects, neither of which is available here; a file of the reference is recognised by the missing magic and refused.
constexpr char kIndexMagic[8] = {'L', 'X', 'I', 'N', 'D', 'E', 'X', '1'};
// AlphabetEnum / DbIndexType with the reference's numbering (src/shared_definitions.hpp:62-66, :127-136)
enum : uint8_t { kAlphUndefined = 0, kAlphDna3Bs = 1, kAlphDna4 = 2, kAlphDna5 = 3, kAlphAminoAcid = 4, kAlphMurphy10 = 5, kAlphLi10 = 6 };
struct IndexFileOptions // index_file_options, :318-341
{
    uint64_t generation = 0; // supportedIndexGeneration, :316
    uint8_t  indexType = 0, origAlph = kAlphUndefined, transAlph = kAlphUndefined, redAlph = kAlphUndefined, geneticCode = 1, pad[3] = {0, 0, 0};
};
char const * alphName(uint8_t a) // _alphabetEnumToName, :138-160
{
    static char const * const names[] = {"UNDEFINED", "dna3bs", "dna4", "dna5", "aminoacid", "murphy10", "li10"};
    return a < 7 ? names[a] : "?";
}

bool isIndexFile(std::string const & path)
{
    std::ifstream f(path, std::ios::binary);
    char          m[8] = {0};
    return f.read(m, 8) && std::memcmp(m, kIndexMagic, 8) == 0;
}

void writeIndexFile(std::string const & path, IndexFileOptions const & io, SeqSet const & db, lambda_amd::ReducedIndex const & ix)
{
    FILE * f = std::fopen(path.c_str(), "wb");
    if (!f)
        throw std::runtime_error("cannot write " + path);
    bool ok    = true;
    auto write = [&](void const * p, size_t bytes) { ok = ok && (bytes == 0 || std::fwrite(p, 1, bytes, f) == bytes); };
    auto u64   = [&](uint64_t v) { write(&v, sizeof(v)); };
    write(kIndexMagic, 8);
    write(&io, sizeof(io));
    u64(db.ids.size());
    for (std::string const & id : db.ids)
    {
        u64(id.size());
        write(id.data(), id.size());
    }
    u64(db.off.size());
    write(db.off.data(), db.off.size() * sizeof(uint64_t));
    write(db.len.data(), db.len.size() * sizeof(uint64_t));
    u64(db.orig_len.size());
    write(db.orig_len.data(), db.orig_len.size() * sizeof(uint64_t));
    u64(db.res.size());
    write(db.res.data(), db.res.size());
    ix.save(write);
    ok = (std::fclose(f) == 0) && ok;
    if (!ok)
    {
        std::remove(path.c_str());
        throw std::runtime_error("error while writing " + path);
    }
}

// everything but the word table (which wants the reduced residues first); the file stays open in *fp
void readIndexHead(std::string const & path, IndexFileOptions & io, SeqSet & db, FILE ** fp)
{
    FILE * f = std::fopen(path.c_str(), "rb");
    if (!f)
        throw std::runtime_error("cannot open " + path);
    auto bad = [&](char const * what) -> std::runtime_error
    {
        std::fclose(f);
        return std::runtime_error("index file " + path + ": " + what);
    };
    auto read = [&](void * p, size_t bytes) { return bytes == 0 || std::fread(p, 1, bytes, f) == bytes; };
    char m[8];
    if (!read(m, 8) || std::memcmp(m, kIndexMagic, 8) != 0)
        throw bad("not an index of this front end (an index of the reference -- a cereal archive of an fmindex-collection FM-index -- cannot be "
                  "read here: run lambda3 mkindexp|mkindexn|mkindexbs on the FASTA file)");
    if (!read(&io, sizeof(io)))
        throw bad("truncated");
    if (io.generation != 0) // :316 "bump this on incompatible changes"; src/search.cpp readIndexOptions
        throw bad("unsupported index generation");
    uint64_t n = 0;
    auto     count = [&](uint64_t limit) -> uint64_t
    {
        uint64_t v = 0;
        if (!read(&v, sizeof(v)) || v > limit)
            throw bad("truncated or corrupt");
        return v;
    };
    n = count(1ull << 40);
    db.ids.resize(n);
    for (std::string & id : db.ids)
    {
        id.resize(count(1ull << 24));
        if (!read(id.data(), id.size()))
            throw bad("truncated");
    }
    n = count(1ull << 40);
    db.off.resize(n);
    db.len.resize(n);
    if (!read(db.off.data(), n * sizeof(uint64_t)) || !read(db.len.data(), n * sizeof(uint64_t)))
        throw bad("truncated");
    n = count(1ull << 40);
    db.orig_len.resize(n);
    if (!read(db.orig_len.data(), n * sizeof(uint64_t)))
        throw bad("truncated");
    n = count(1ull << 46);
    db.res.resize(n);
    if (!read(db.res.data(), n))
        throw bad("truncated");
    if (db.ids.empty() || db.orig_len.size() != db.ids.size() || db.off.empty() || db.off.size() % db.ids.size() != 0)
        throw bad("inconsistent sequence tables");
    for (size_t i = 0; i < db.off.size(); ++i)
        if (db.len[i] > db.res.size() || db.off[i] > db.res.size() - db.len[i])
            throw bad("a sequence lies outside the residues");
    {
        // the residues index the scoring and reduction tables: none beyond the translated alphabet (aa27: 27 ranks, dna5: 5)
        unsigned const size = io.transAlph == kAlphAminoAcid ? 27u : io.transAlph == kAlphDna5 ? 5u : 0u;
        uint8_t        top  = 0;
        for (uint8_t r : db.res)
            top = std::max(top, r);
        if (size == 0 || (!db.res.empty() && top >= size))
            throw bad("residues outside the index's alphabet");
    }
    *fp = f;
}

// host threads this process may use: the hardware's, capped by the affinity mask and the cgroup's CPU quota
unsigned grantedThreads()
{
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0)
        n = std::min<unsigned>(n, (unsigned)CPU_COUNT(&set));
    std::ifstream f("/sys/fs/cgroup/cpu.max");
    std::string   quota;
    double        period = 0;
    if (f >> quota >> period && quota != "max" && period > 0)
        n = std::min<unsigned>(n, std::max(1u, (unsigned)(std::stod(quota) / period + 0.5)));
    return std::min(n, 64u);
}

} // namespace

int main(int argc, char ** argv)
{
    try
    {
        Options const opt  = parse(argc, argv);
        auto const    tStart = std::chrono::steady_clock::now();
        auto          msSince = [](std::chrono::steady_clock::time_point a)
        { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
        bool const    mk = opt.cmd.rfind("mkindex", 0) == 0;
        bool const    prot = opt.cmd == "searchp" || opt.cmd == "mkindexp", bs = opt.cmd == "searchbs" || opt.cmd == "mkindexbs";
        // -i / -d names an index made by `lambda3 mkindex*`, or the database as FASTA (the table is then made in memory)
        bool const    fromIndex = !mk && isIndexFile(opt.db);
        SeqSet           qs, db;
        IndexFileOptions ifo;
        struct FileCloser
        {
            void operator()(FILE * f) const
            {
                if (f)
                    std::fclose(f);
            }
        };
        std::unique_ptr<FILE, FileCloser> indexFileOwner; // (closed on every way out of main)
        FILE *                            indexFile = nullptr;
        if (fromIndex)
        {
            readIndexHead(opt.db, ifo, db, &indexFile);
            indexFileOwner.reset(indexFile);
            // the index fixes the domain (src/search.cpp:189-207)
            {
                if (prot && ifo.transAlph != kAlphAminoAcid)
                    throw std::runtime_error("Attempting to use nucleotide or bisulfite index for protein search.");
                if (!prot && ifo.transAlph != kAlphDna5)
                    throw std::runtime_error(bs ? "Attempting to use protein index for bisulfite search." : "Attempting to use protein index for nucleotide search.");
                if (!prot && !bs && ifo.redAlph != kAlphDna4)
                    throw std::runtime_error("Attempting to use bisulfite index for nucleotide search.");
                if (bs && ifo.redAlph != kAlphDna3Bs)
                    throw std::runtime_error("Attempting to use nucleotid index for bisulfite search.");
            }
        }
        std::string const reduction = !fromIndex ? opt.reduction : ifo.redAlph == kAlphLi10 ? "li10" : ifo.redAlph == kAlphMurphy10 ? "murphy10" : "none";
        // searchp with nucleotide queries is BLASTX: six translated frames per query against the protein database
        bool const    blastx = !mk && prot && (opt.qryAlphabet == "dna5" || (opt.qryAlphabet == "auto" && looksLikeDna(opt.query)));
        // searchp against a nucleotide database translates the subjects instead (TBLASTN), or both sides (TBLASTX)
        bool const    sTrans  = fromIndex ? (prot && ifo.origAlph != ifo.transAlph)
                                          : prot && (opt.dbAlphabet == "dna5" || (opt.dbAlphabet == "auto" && looksLikeDna(opt.db)));
        // genetic code: the subjects' is the index's; the queries' is -g, else the index's (src/search.cpp:157-178)
        int const geneticCodeDb  = fromIndex ? (int)ifo.geneticCode : opt.geneticCode;
        int const geneticCodeQry = (opt.geneticCodeGiven || !fromIndex || !sTrans) ? opt.geneticCode : (int)ifo.geneticCode;
        if (fromIndex && sTrans && geneticCodeQry != geneticCodeDb)
            std::cerr << "WARNING: The genetic code used when creating the index: " << geneticCodeDb
                      << "\n         is not the same as now selected for the query sequences: " << geneticCodeQry
                      << "\n         Are you sure this is what you want?\n";
        // frames per sequence, src/search_datastructures.hpp:380-385
        int const     qFrames = bs ? 4 : blastx ? 6 : prot ? 1 : 2;
        int const     sFrames = bs ? 2 : sTrans ? 6 : 1;
        char const *  program = blastx ? (sTrans ? "tblastx" : "blastx") : sTrans ? "tblastn" : prot ? "blastp" : "blastn";

        // the queries are read beside the database (two files, two threads; -t 1 keeps it to one)
        std::string qryError;
        std::thread qryReader;
        auto        readQueries = [&]()
        {
            try
            {
                readFasta(opt.query, prot, !prot, qs, blastx, geneticCodeQry, bs ? 2 : 0);
            }
            catch (std::exception const & e)
            {
                qryError = e.what();
            }
        };
        struct Joiner // (joined on every way out)
        {
            std::thread & t;
            ~Joiner()
            {
                if (t.joinable())
                    t.join();
            }
        } joiner{qryReader};
        if (!mk && opt.threads != 1)
            qryReader = std::thread(readQueries);
        else if (!mk)
            readQueries();
        if (!fromIndex)
        {
            std::ifstream probe(opt.db, std::ios::binary);
            char          c = 0;
            while (probe.get(c) && std::isspace((unsigned char)c))
                ;
            if (probe && c != '>')
                throw std::runtime_error(opt.db + " is neither a FASTA file nor an index of this front end (an index written by the reference -- a cereal "
                                         "archive of an fmindex-collection FM-index -- cannot be read here: run lambda3 mkindexp|mkindexn|mkindexbs on the FASTA file)");
            readFasta(opt.db, prot, false, db, sTrans, geneticCodeDb, bs ? 1 : 0);
        }
        if (qryReader.joinable())
            qryReader.join();
        if (!qryError.empty())
            throw std::runtime_error(qryError);
        if ((!mk && qs.ids.empty()) || db.ids.empty())
            throw std::runtime_error("empty query or database file");
        if (fromIndex && db.off.size() != db.ids.size() * (size_t)sFrames)
            throw std::runtime_error("index file " + opt.db + ": the frames of its sequences do not fit its alphabets");
        if (mk && opt.truncateIds) // src/mkindex_algo.hpp:123
            for (std::string & id : db.ids)
                id.resize(std::min(id.size(), id.find_first_of(" \t")));
        double const msRead = msSince(tStart);
        unsigned const nThreads = opt.threads > 0 ? (unsigned)opt.threads : grantedThreads();

        // ---- the reduced alphabet of the seeding stage and the word table over the reduced database (the FM-index's place)
        uint8_t const * redTab = nullptr;
        int             alph   = 27;
        if (prot && reduction == "li10")
            redTab = lambda_amd::kLi10, alph = 10;
        else if (prot && reduction == "murphy10")
            redTab = lambda_amd::kMurphy10, alph = 10;
        else if (bs)
            alph = 6;
        else if (!prot)
            redTab = lambda_amd::kDna4, alph = 4;
        // bisulfite: the reduction alternates with the frame (even: forward, odd: reverse; src/view_reduce_to_bisulfite.hpp:132-136)
        auto reduce = [&](SeqSet const & set)
        {
            std::vector<uint8_t> red(set.res.size());
            if (bs)
            {
                for (size_t f = 0; f < set.off.size(); ++f)
                    for (uint64_t i = 0; i < set.len[f]; ++i)
                        red[set.off[f] + i] = ((f & 1) ? lambda_amd::kBsRev : lambda_amd::kBsFwd)[std::min<uint8_t>(set.res[set.off[f] + i], 4)];
                return red;
            }
            for (size_t i = 0; i < set.res.size(); ++i)
                red[i] = redTab ? redTab[set.res[i] < (prot ? 27 : 5) ? set.res[i] : 0] : set.res[i];
            return red;
        };
        if (!mk || opt.table == "gpu") // (before the first device is touched: the library's own wording)
            for (int d : opt.devices)
                if (d < 0 || d >= lx_device_count())
                    throw std::runtime_error("device_id " + std::to_string(d) + " out of range [0," + std::to_string(lx_device_count()) + ")");
        auto const                 tIndex = std::chrono::steady_clock::now();
        std::vector<uint8_t> const dbRed = reduce(db);
        lambda_amd::ReducedIndex   ix;
        bool                       tableOnGpu = false;
        lambda_amd::DeviceTable    deviceTable; // (the table where the GPU builder left it, for the seeding stage on that device)
        if (fromIndex)
        {
            bool const ok = ix.load([&](void * p, size_t bytes) { return bytes == 0 || std::fread(p, 1, bytes, indexFile) == bytes; }, dbRed, db.off, db.len) &&
                            ix.alphabet() == alph;
            indexFileOwner.reset();
            if (!ok)
                throw std::runtime_error("index file " + opt.db + ": the word table is truncated or does not fit the sequences");
        }
        else
        {
            // on the GPU (keys, one radix sort, prefix table: host/lx_seeding_gpu.hpp) where there is one to take it, else on the -t
            // host threads; the same table either way
            bool const wantGpu = opt.table == "gpu" || (opt.table == "auto" && !mk);
            if (wantGpu && lx_device_count() > 0)
            {
                try
                {
                    tableOnGpu = lambda_amd::buildTableOnGpu(opt.devices.empty() ? 0 : opt.devices[0], ix, dbRed, db.off, db.len, alph,
                                                             (!mk && opt.seeding == "gpu") ? &deviceTable : nullptr);
                }
                catch (std::exception const & e) // (e.g. the device ran out of memory after all: the host threads make the table)
                {
                    if (opt.table == "gpu")
                        throw;
                    std::cerr << "WARNING: the word table is made on the host threads (" << e.what() << ")\n";
                    tableOnGpu = false;
                }
            }
            else if (opt.table == "gpu")
                throw std::runtime_error("--table gpu: no HIP device available");
            if (!tableOnGpu)
                ix.build(dbRed, db.off, db.len, alph, nThreads);
        }
        double const msIndex = msSince(tIndex);
        if (mk)
        {
            // mkindexp | mkindexn | mkindexbs (src/mkindex.cpp): the options that fix the types, ids, sequences, table -> one file
            IndexFileOptions out;
            out.indexType   = opt.dbIndexType == "bifm" ? 1 : 0;
            out.origAlph    = prot ? (sTrans ? kAlphDna5 : kAlphAminoAcid) : kAlphDna5; // :206-224, :235
            out.transAlph   = prot ? kAlphAminoAcid : kAlphDna5;
            out.redAlph     = bs ? kAlphDna3Bs : !prot ? kAlphDna4 : reduction == "li10" ? kAlphLi10 : reduction == "murphy10" ? kAlphMurphy10 : kAlphAminoAcid;
            out.geneticCode = (uint8_t)geneticCodeDb;
            auto const tWrite = std::chrono::steady_clock::now();
            writeIndexFile(opt.index, out, db, ix);
            uint64_t residues = 0;
            for (auto l : db.len)
                residues += l;
            std::fprintf(stderr,
                         "lambda3 %s: %zu sequences (%llu residues in %d frame(s); original alphabet %s, translated %s, reduced %s, genetic code %d) -> %s\n"
                         "lambda3 times [ms]: read %.0f, reduce + word table %.0f (%s), write %.0f, total %.0f\n",
                         opt.cmd.c_str(), db.ids.size(), (unsigned long long)residues, sFrames, alphName(out.origAlph), alphName(out.transAlph),
                         alphName(out.redAlph), (int)out.geneticCode, opt.index.c_str(), msRead, msIndex, tableOnGpu ? "on the GPU" : (std::to_string(nThreads) + " host thread(s)").c_str(),
                         msSince(tWrite), msSince(tStart));
            return 0;
        }
        std::vector<uint8_t> const qRed = reduce(qs);

        // ---- scoring + statistics (prepareScoring, src/search_algo.hpp:166-234): bisulfite = two matrices over SeqAn Dna5
        // (forward: slot 0, reverse: slot 1, :176-186), statistics from the match / mismatch scheme
        lx_scoring sc, scRev;
        int const  gapOpen = opt.gapOpen, gapExtend = opt.gapExtend;
        lambda_amd::builtinScoring(prot ? opt.scoringMethod : bs ? -1 : 0, opt.match, opt.misMatch, gapOpen, gapExtend, sc);
        if (bs)
            lambda_amd::builtinScoring(-2, opt.match, opt.misMatch, gapOpen, gapExtend, scRev);
        lx_karlin ka;
        if (!lambda_amd::karlinParams(prot ? opt.scoringMethod : 0, opt.match, opt.misMatch, gapOpen, gapExtend, ka))
            throw std::runtime_error("Could not compute Karlin-Altschul-Values for Scoring Scheme."); // :232-233

        // ---- devices and worker threads (src/search.cpp:379-385: one LocalDataHolder per thread; here one handle per thread)
        std::vector<int> devices = opt.devices;
        if (devices.empty())
            for (int d = 0; d < lx_device_count(); ++d)
                devices.push_back(d);
        if (devices.empty())
            throw std::runtime_error("no HIP device available (this front end has no CPU path)");
        // one worker = one handle per entry of --devices (a device may be listed twice); -t host threads in all, shared out among
        // the workers for the seeding of their reads (the reference: -t OpenMP threads, each with its own LocalDataHolder; a GPU
        // wants few, large extension batches, the seeding wants every core)
        size_t const   nWorkers = std::max<size_t>(1, std::min<size_t>(devices.size(), qs.ids.size()));
        unsigned const seedThreads = std::max(1u, nThreads / (unsigned)nWorkers);

        // ---- seeding (search(), src/search_algo.hpp:611-762) over the sorted table of reduced words
        lambda_amd::SeedingInput sin{};
        sin.qRes = qs.res.data(), sin.qRed = qRed.data(), sin.qOff = qs.off.data(), sin.qLen = qs.len.data(), sin.nQSeq = qs.off.size();
        sin.qNumFrames       = qFrames;
        sin.unknownRank      = prot ? 25 : bs ? 4 : 3; // 'X' / 'N'
        sin.sRes = db.res.data(), sin.sOff = db.off.data(), sin.sLen = db.len.data();
        sin.alph             = alph;
        sin.matrix           = sc.matrix;
        sin.matrixRev        = bs ? scRev.matrix : nullptr;
        sin.maxMatches       = opt.maxMatches;
        sin.halfExact        = opt.halfExact;
        sin.adaptive         = opt.adaptive;
        sin.preScoring       = opt.preScoring;
        sin.preScoringThresh = opt.preScoringThresh;

        // dbTotalLength = sum of the (frame-expanded) subject lengths, src/search_algo.hpp:317-319
        uint64_t dbTotal = 0;
        for (auto l : db.len)
            dbTotal += l;
        lx_search_params sp{};
        sp.max_evalue       = opt.maxEValue;
        sp.min_bitscore     = opt.minBitScore;
        sp.id_cutoff        = opt.idCutOff;
        sp.db_total_length  = dbTotal;
        sp.query_translated = blastx ? 1 : 0;
        sp.qry_num_frames   = qFrames;
        sp.sbj_num_frames   = sFrames;
        sp.bisulfite        = bs ? 1 : 0;
        sp.q_frame_mode     = bs ? LX_FRAMES_BISULFITE : blastx ? LX_FRAMES_TRANSLATED : prot ? LX_FRAMES_NONE : LX_FRAMES_REVCOMP; // _setFrames, :768-814
        sp.s_frame_mode     = bs ? LX_FRAMES_BISULFITE : sTrans ? LX_FRAMES_TRANSLATED : LX_FRAMES_NONE;
        sp.karlin           = ka;
        // only the SAM writer reads alignment columns (its CIGAR); the tables are made from the counts
        bool const wantOps  = opt.output.size() >= 4 && opt.output.compare(opt.output.size() - 4, std::string::npos, ".sam") == 0;
        sp.flags            = wantOps ? 0 : LX_ITERATE_NO_OPS;

        lambda_amd::SeedParams const so1{opt.seedLength, opt.seedOffset, opt.seedDelta}, so0{opt.seedLength0, opt.seedOffset0, opt.seedDelta0};
        // what a worker keeps for its range of reads
        struct Part
        {
            std::vector<lx_blast_match> bms;
            std::vector<uint8_t>        ops;
            lx_iterate_stats            ist{};
            lambda_amd::SeedingStats    sst{};
            size_t                      nPromising = 0;
            double                      msSeed = 0, msExtend = 0;
            bool                        gpuSeeding = false; // the seeding stage of this worker ran on its device
            size_t                      nDeclined = 0, nPassesOnHost = 0; // reads the GPU seeding stage left to the host; passes whose match buffer was full
            std::string                 error;
        };
        std::vector<Part> parts(nWorkers);
        uint64_t const    nReads = qs.ids.size();
        auto worker = [&](size_t w)
        {
            Part & pt = parts[w];
            try
            {
                // contiguous range of reads, as `omp for schedule(dynamic)` over whole batches would hand a thread (:384-385)
                uint64_t const rLo = nReads * w / nWorkers, rHi = nReads * (w + 1) / nWorkers;
                if (rLo == rHi)
                    return;
                lambda_amd::Engine eng(devices[w % devices.size()]);
                eng.setScoring(sc, 0);
                if (bs)
                    eng.setScoring(scRev, 1);
                // the database stays on the GPU for the whole run (the reference keeps it in the index file it maps at start-up)
                eng.check(lx_set_subjects(eng.raw(), db.res.data(), db.res.size()));
                // one pass of the batch loop of realMain (src/search.cpp:426-459): seed, extend (GPU), collect
                std::unique_ptr<lambda_amd::GpuSeeder> gpuSeeder;
                if (opt.seeding == "gpu" && lambda_amd::GpuSeeder::canTake(ix))
                    try
                    {
                        gpuSeeder.reset(new lambda_amd::GpuSeeder(devices[w % devices.size()], ix, sin, dbRed, db.off.size(), db.res.size(), qs.res.size(), &deviceTable));
                    }
                    catch (std::exception const & e) // (the table and the sequences did not fit beside the extension's buffers: host threads)
                    {
                        std::cerr << "WARNING: seeding on the host threads (" << e.what() << ")\n";
                        gpuSeeder.reset();
                    }
                pt.gpuSeeding = gpuSeeder != nullptr;
                // with the seeding on the device its matches stay there: the sequence sets become resident for the Level-2 kernels
                // (LAMBDA3_HOST_LIST=1: the matches come down and go through lx_iterate_matches, the A/B switch of the tests)
                bool const deviceList = gpuSeeder && !std::getenv("LAMBDA3_HOST_LIST");
                if (deviceList)
                {
                    eng.check(lx_set_subject_seqs(eng.raw(), db.off.data(), db.len.data(), db.off.size()));
                    eng.check(lx_set_queries(eng.raw(), qs.res.data(), qs.res.size(), qs.off.data(), qs.len.data(), qs.off.size(), qs.orig_len.data(), qFrames));
                    // this worker makes ONE Level-2 call per seeding pass: what that call would allocate and touch inside itself is asked
                    // for here, before the seeding (a dozen promising seeds per read, a window per seven of them, a record per twelve --
                    // what the read sets of tools/cli_scale_nucl.py come to; a short estimate only moves an allocation back into the call)
                    uint64_t const est = std::min<uint64_t>(12 * (rHi - rLo), 0x7ffffff0ull);
                    uint64_t       len = 0;
                    for (uint64_t i = rLo * (uint64_t)qFrames; i < rHi * (uint64_t)qFrames && i < qs.len.size(); ++i)
                        len = std::max<uint64_t>(len, qs.len[i]);
                    if (est >= 100000)
                        eng.check(lx_reserve(eng.raw(), est, est / 7, est / 12, wantOps ? est / 12 * (len + len / 16) : 0));
                }
                auto pass = [&](lambda_amd::SeedParams const & so, std::vector<uint64_t> const & which)
                {
                    std::vector<lx_match> matches;
                    if (std::getenv("LAMBDA3_TRACE"))
                        std::fprintf(stderr, "[worker %zu] seeding %zu frame sequences (seed %d/%d, delta %d)\n", w, which.size(), so.seedLength, so.seedOffset, so.maxSeedDist);
                    auto const tSeed = std::chrono::steady_clock::now();
                    bool     onHost   = !gpuSeeder;
                    uint64_t onDevice = 0; // matches the seeding kernel left in device memory (a pass of one launch, nothing declined)
                    if (gpuSeeder)
                    {
                        // the device takes the reads; those it declines (words far beyond the table's keys with many occurrences) and
                        // those of a launch whose match buffer filled up are seeded here
                        std::vector<uint64_t>          declined;
                        lambda_amd::SeedingStats const sstBefore = pt.sst;
                        try
                        {
                            pt.nPassesOnHost += gpuSeeder->seed(so, which, matches, pt.sst, declined, deviceList ? &onDevice : nullptr);
                        }
                        catch (std::exception const & e)
                        {
                            // a HIP error in the seeding stage (its match buffer did not fit, ...): this pass and the following ones are
                            // seeded on the host threads, from a clean slate
                            std::cerr << "WARNING: seeding on the host threads from here on (" << e.what() << ")\n";
                            gpuSeeder.reset();
                            pt.gpuSeeding = false;
                            matches.clear();
                            declined.clear();
                            onDevice = 0;
                            pt.sst   = sstBefore;
                            onHost   = true;
                        }
                        if (!declined.empty())
                        {
                            std::sort(declined.begin(), declined.end());
                            std::vector<uint64_t> rest;
                            for (uint64_t rd : declined)
                                for (int f = 0; f < qFrames && rd + (uint64_t)f < qs.off.size(); ++f)
                                    rest.push_back(rd + (uint64_t)f);
                            lambda_amd::seedQueriesParallel(ix, sin, so, rest, matches, pt.sst, seedThreads);
                            pt.nDeclined += declined.size();
                        }
                    }
                    if (onHost)
                        lambda_amd::seedQueriesParallel(ix, sin, so, which, matches, pt.sst, seedThreads);
                    pt.msSeed += msSince(tSeed);
                    pt.nPromising += matches.size() + onDevice;
                    if (std::getenv("LAMBDA3_TRACE"))
                        std::fprintf(stderr, "[worker %zu] %zu promising seeds -> extension\n", w, matches.size() + (size_t)onDevice);
                    if (matches.empty() && onDevice == 0)
                        return;
                    lx_iterate_result * res = nullptr;
                    auto const          tExt = std::chrono::steady_clock::now();
                    if (onDevice) // the Level-2 driver on the list where the seeding kernel left it (include/lambda_ext.h)
                        eng.check(lx_iterate_matches_dev(eng.raw(), 0, gpuSeeder->devMatches(), onDevice, &sp, &res));
                    else
                        eng.check(lx_iterate_matches(eng.raw(), 0, qs.res.data(), qs.res.size(), qs.off.data(), qs.len.data(), qs.off.size(),
                                                     qs.orig_len.data(), nullptr, 0, db.off.data(), db.len.data(), db.off.size(), matches.data(),
                                                     matches.size(), &sp, &res));
                    pt.msExtend += msSince(tExt);
                    uint64_t const         n  = lx_iterate_result_count(res);
                    lx_blast_match const * bm = lx_iterate_result_matches(res);
                    uint64_t const         ob = pt.ops.size();
                    uint64_t opsEnd = 0; // (the records are ordered by query, their ops as the passes produced them: bisulfite runs two)
                    for (uint64_t k = 0; k < n && wantOps; ++k)
                        opsEnd = std::max<uint64_t>(opsEnd, bm[k].ops_off + bm[k].n_ops);
                    if (opsEnd)
                        pt.ops.insert(pt.ops.end(), lx_iterate_result_ops(res), lx_iterate_result_ops(res) + opsEnd);
                    for (uint64_t k = 0; k < n; ++k)
                    {
                        pt.bms.push_back(bm[k]);
                        pt.bms.back().ops_off += ob;
                    }
                    lx_iterate_stats const st = lx_iterate_result_stats(res);
                    pt.ist.hits_duplicate += st.hits_duplicate, pt.ist.failed_bitscore += st.failed_bitscore, pt.ist.failed_evalue += st.failed_evalue;
                    pt.ist.failed_identity += st.failed_identity, pt.ist.num_ext_score += st.num_ext_score, pt.ist.num_ext_ali += st.num_ext_ali;
                    lx_iterate_result_free(res);
                };
                std::vector<uint64_t> all;
                for (uint64_t i = rLo * (uint64_t)qFrames; i < rHi * (uint64_t)qFrames; ++i)
                    all.push_back(i);
                if (opt.search0) // iterativeSearch (:1391-1457): the exact pre-search first, the default parameters for reads without a result
                {
                    pass(so0, all);
                    std::vector<uint8_t> successful(nReads, 0);
                    for (auto const & bm : pt.bms)
                        successful[bm.n_qid] = 1;
                    std::vector<uint64_t> rest;
                    for (uint64_t i : all)
                        if (!successful[i / (uint64_t)qFrames])
                            rest.push_back(i);
                    if (!rest.empty())
                        pass(so1, rest);
                    // (the reference writes phase 1's records of a batch before phase 2's; here both lists are written together:
                    // order by read, as one batch holding every read would give)
                    std::stable_sort(pt.bms.begin(), pt.bms.end(), [](lx_blast_match const & a, lx_blast_match const & b) { return a.n_qid < b.n_qid; });
                }
                else
                    pass(so1, all);
            }
            catch (std::exception const & e)
            {
                pt.error = e.what();
            }
        };
        auto const tSearch = std::chrono::steady_clock::now();
        {
            std::vector<std::thread> pool;
            for (size_t w = 1; w < nWorkers; ++w)
                pool.emplace_back(worker, w);
            worker(0);
            for (auto & t : pool)
                t.join();
        }
        double const msSearch = msSince(tSearch);
        auto const   tOut     = std::chrono::steady_clock::now();
        // the ranges are disjoint and ascending: concatenation in range order is the order one thread would have produced
        std::vector<lx_blast_match> bms;
        std::vector<uint8_t>        ops;
        lx_iterate_stats            ist{};
        lambda_amd::SeedingStats    sst{};
        size_t                      nPromising = 0;
        double                      msSeedMax = 0, msExtendMax = 0;
        size_t                      nDeclined = 0, nPassesOnHost = 0;
        bool                        anyGpuSeeding = false;
        for (Part & pt : parts)
        {
            if (!pt.error.empty())
                throw std::runtime_error(pt.error);
            uint64_t const ob = ops.size();
            ops.insert(ops.end(), pt.ops.begin(), pt.ops.end());
            for (lx_blast_match m : pt.bms)
            {
                m.ops_off += ob;
                bms.push_back(m);
            }
            ist.hits_duplicate += pt.ist.hits_duplicate, ist.failed_bitscore += pt.ist.failed_bitscore, ist.failed_evalue += pt.ist.failed_evalue;
            ist.failed_identity += pt.ist.failed_identity, ist.num_ext_score += pt.ist.num_ext_score, ist.num_ext_ali += pt.ist.num_ext_ali;
            sst.hitsAfterSeeding += pt.sst.hitsAfterSeeding, sst.hitsFailedPreExtendTest += pt.sst.hitsFailedPreExtendTest;
            nPromising += pt.nPromising;
            nDeclined += pt.nDeclined, nPassesOnHost += pt.nPassesOnHost;
            anyGpuSeeding = anyGpuSeeding || pt.gpuSeeding;
            msSeedMax   = std::max(msSeedMax, pt.msSeed);
            msExtendMax = std::max(msExtendMax, pt.msExtend);
        }
        uint64_t const nHsp   = bms.size();
        size_t const   nSeeds = (size_t)sst.hitsAfterSeeding;

        // ---- _writeRecord + writer
        lx_record_stats rst{};
        uint64_t const  nOut = lx_postprocess_records(bms.data(), bms.size(), opt.maxMatches, &rst);
        std::vector<char const *> qid, sid;
        for (auto const & s : qs.ids)
            qid.push_back(s.c_str());
        for (auto const & s : db.ids)
            sid.push_back(s.c_str());
        lx_seq_names names{qid.data(), qs.orig_len.data(), sid.data(), db.orig_len.data(), qid.size(), sid.size()};
        int          fmt = LX_OUT_BLAST_TAB;
        auto         ends = [&](char const * suf)
        { return opt.output.size() >= std::strlen(suf) && opt.output.compare(opt.output.size() - std::strlen(suf), std::string::npos, suf) == 0; };
        if (ends(".m9"))
            fmt = LX_OUT_BLAST_TAB_COMMENTS;
        else if (ends(".sam"))
            fmt = LX_OUT_SAM;
        else if (!ends(".m8"))
            throw std::runtime_error("output format is chosen by the extension: .m8, .m9 or .sam"); // :684-710
        {
            lx_output_options oo;
            lx_output_options_default(&oo);
            oo.columns             = opt.outputColumns.c_str();
            oo.sam_tags            = opt.samTags.c_str();
            oo.sam_seq             = opt.samSeq == "never" ? LX_SAM_SEQ_NEVER : opt.samSeq == "uniq" ? LX_SAM_SEQ_UNIQ : LX_SAM_SEQ_ALWAYS;
            oo.sam_hard_clip       = opt.samClip == "hard";
            oo.sam_with_ref_header = opt.samWithRefHeader;
            oo.version_to_output   = opt.versionToOutput;
            oo.version             = "3.0.0-lx"; // (the reference release this front end follows, src/CMakeLists.txt:14-19)
            oo.command_line        = opt.commandLine.c_str();
            oo.db_name             = opt.db.c_str(); // the index path there (src/search_algo.hpp:320)
            oo.genetic_code        = geneticCodeQry;
            int const rcw = lx_write_records_ex(opt.output.c_str(), fmt, 1, program, bms.data(), nOut, ops.data(), &names,
                                                reinterpret_cast<uint8_t const *>(qs.ascii.data()), qs.ascii_off.data(), &oo);
            if (rcw != LX_OK)
                throw std::runtime_error(*lx_last_output_error() ? lx_last_output_error() : ("cannot write " + opt.output).c_str());
            if (lx_write_footer(opt.output.c_str(), fmt, rst.qrys_with_hit) != LX_OK) // myWriteFooter, src/search.cpp
                throw std::runtime_error("cannot write " + opt.output);
        }

        std::fprintf(stderr,
                     "lambda3 %s (%s, %u host thread(s), %zu handle(s) on %zu device(s)): %zu queries, %zu subjects (%llu residues); seeds %zu -> promising %zu -> "
                     "windows %llu -> traced %llu -> HSPs %llu -> written %llu (queries with hit: %llu)\n",
                     opt.cmd.c_str(), program, nThreads, nWorkers, devices.size(), qs.ids.size(), db.ids.size(), (unsigned long long)dbTotal, nSeeds, nPromising,
                     (unsigned long long)(ist.num_ext_score - ist.hits_duplicate), (unsigned long long)ist.num_ext_ali,
                     (unsigned long long)nHsp, (unsigned long long)nOut, (unsigned long long)rst.qrys_with_hit);
        // where the wall clock went (the reference prints its own at verbosity 2, src/search.cpp): per worker the slowest counts
        std::fprintf(stderr,
                     "lambda3 times [ms]: read %.0f, reduce + word table %s%.0f, search %.0f (seeding on the %s %.0f [%zu read(s) and %zu launch(es) left to "
                     "the host] + extension on the GPU incl. widen / merge / statistics %.0f on the slowest worker), records + output %.0f, total %.0f\n",
                     msRead, fromIndex ? "(read from the index) " : tableOnGpu ? "(on the GPU) " : "", msIndex, msSearch, anyGpuSeeding ? "GPU" : "host", msSeedMax, nDeclined,
                     nPassesOnHost, msExtendMax, msSince(tOut), msSince(tStart));
        return 0;
    }
    catch (std::exception const & e)
    {
        std::cerr << "\nERROR: " << e.what() << "\n";
        return -1; // src/search.cpp:98-125
    }
}
