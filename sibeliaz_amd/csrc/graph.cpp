// graph.cpp — loads a TwoPaCo junction file + FASTA into structure-of-arrays tables.
//
// Replaces JunctionStorage::Init (junctionstorage.h:572-650), JunctionPositionReader
// (common/junctionapi.h:80-98) and StreamFastaParser (common/streamfastaparser.cpp:28-92).
// Same observable semantics (abundance filter, idx after filtering, ch/revCh definition, FASTA
// header token, upper-casing, validation and error texts), different construction: both inputs
// are read with one bulk read each and then worked on by all host threads (record passes cut into blocks, FASTA records
// parsed concurrently), the per-vertex occurrence lists become a CSR, and the nested vectors become flat arrays indexed by
// g = chrStart[chr] + idx.
//
// Inputs the reference silently mis-handles (out-of-bounds access) are rejected loudly here:
// a chromosome without junctions, a FASTA record count different from the chromosome count, a
// junction beyond the end of its sequence, positions that do not increase.
#include <omp.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "lcb_host.h"

namespace {

bool readAll(const std::string& file, std::vector<char>& buf)
{
    FILE* f = fopen(file.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize(sz > 0 ? (size_t)sz : 0);
    size_t got = buf.empty() ? 0 : fread(buf.data(), 1, buf.size(), f);
    fclose(f);
    buf.resize(got);
    return true;
}

// common/dnachar.cpp:52-70,82-85
inline char reverseChar(char c)
{
    switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C'; }
    return 'N';
}

struct ValidTable {                    // common/dnachar.cpp:11 VALID_CHARS
    bool v[256];
    ValidTable() { memset(v, 0, sizeof(v)); for (const char* p = "ACGTURYKMSWBDHWNXV"; *p; ++p) v[(unsigned char)*p] = true; }
};
const ValidTable kValid;

// One FASTA file: the records are located first (a '>' anywhere starts one, streamfastaparser.cpp:28-58), then parsed by all
// threads. `header` carries over between records and files like the reference's currentHeader_ (a header line without a
// newline at the end of the file leaves it unchanged).
void parseFasta(const std::string& file, lcb_graph& g, size_t& record, std::string& header, int threads)
{
    std::vector<char> buf;
    if (!readAll(file, buf)) throw LcbError("Can't open file " + file);                 // streamfastaparser.cpp:24
    const size_t n = buf.size();
    if (n && buf[0] != '>')                                                              // streamfastaparser.cpp:33-36
        throw LcbError("The FASTA header should start with a '>', started with '" + std::string(1, buf[0]) + "'");
    std::vector<size_t> starts;                                                          // offsets of the '>' characters
    {
        const int T = threads;
        std::vector<std::vector<size_t>> part((size_t)T);
#pragma omp parallel for num_threads(threads) schedule(static)
        for (int t = 0; t < T; t++) {
            const size_t lo = n * (size_t)t / T, hi = n * (size_t)(t + 1) / T;
            for (const char* q = (const char*)memchr(buf.data() + lo, '>', hi - lo); q; q = (const char*)memchr(q + 1, '>', (size_t)(buf.data() + hi - q - 1)))
                part[(size_t)t].push_back((size_t)(q - buf.data()));
        }
        for (auto& v : part) starts.insert(starts.end(), v.begin(), v.end());
    }
    // a '>' inside a header line belongs to that header
    std::vector<size_t> recStart, seqStart;
    std::vector<std::string> names;
    for (size_t q = 0; q < starts.size();) {
        size_t i = starts[q] + 1, e = i;
        while (e < n && buf[e] != '\n') e++;
        if (e < n) {                                                                     // `ss >> currentHeader_` only on '\n'
            size_t s0 = i;
            while (s0 < e && isspace((unsigned char)buf[s0])) s0++;
            size_t t0 = s0;
            while (t0 < e && !isspace((unsigned char)buf[t0])) t0++;
            if (t0 > s0) header.assign(buf.data() + s0, t0 - s0);
            i = e + 1;
        } else i = n;
        recStart.push_back(starts[q]); seqStart.push_back(i); names.push_back(header);
        q++;
        while (q < starts.size() && starts[q] < i) q++;                                  // '>' characters of the header line itself
    }
    const size_t nr = recStart.size();
    if (record + nr > g.seq.size()) throw LcbError("the FASTA input has more records than the junction file has chromosomes");
    for (size_t r = 0; r < nr; r++) g.chrName.push_back(names[r]);
    std::vector<std::string> errs(nr);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (int64_t r = 0; r < (int64_t)nr; r++) {
        const size_t end = (size_t)r + 1 < nr ? recStart[(size_t)r + 1] : n;              // a '>' anywhere ends the record
        std::string& seq = g.seq[record + (size_t)r];
        seq.resize(end > seqStart[(size_t)r] ? end - seqStart[(size_t)r] : 0);
        size_t w = 0;
        for (size_t i = seqStart[(size_t)r]; i < end; i++) {
            const unsigned char c = (unsigned char)buf[i];
            if (isspace(c)) continue;
            const unsigned char u = (unsigned char)toupper(c);
            if (!kValid.v[u]) {                                                          // streamfastaparser.cpp:80-83
                if (errs[(size_t)r].empty()) errs[(size_t)r] = "Found an invalid character '" + std::string(1, (char)c) + "' in sequence " + names[(size_t)r];
                break;
            }
            seq[w++] = (char)u;
        }
        seq.resize(w);
    }
    for (auto& e : errs) if (!e.empty()) throw LcbError(e);
    record += nr;
}

}  // namespace

lcb_graph* lcb_graph_load_impl(const char* junctionFile, const std::vector<std::string>& fasta, int k, int abundance, int threads)
{
    if (threads < 1) threads = 1;
    std::vector<char> raw;
    if (!readAll(junctionFile, raw)) throw LcbError("Can't read the input file");       // junctionapi.h:46-49
    const size_t nRec = raw.size() / 12;
    auto* g = new lcb_graph();
    try {
        g->k = k;
        omp_set_dynamic(0);
        const char* rawp = raw.data();
        auto recOf = [rawp](size_t r, uint32_t& pos, int64_t& id) { memcpy(&pos, rawp + r * 12, 4); memcpy(&id, rawp + r * 12 + 4, 8); };
        auto isSep = [](uint32_t pos, int64_t id) { return pos == 0xFFFFFFFFu || id == INT64_MAX; };   // junctionapi.h:93
        // chromosomes = runs of junction records between separator records (junctionstorage.h:576-594), located by all threads
        std::vector<size_t> seps;
        int64_t maxAbs = -1;
        {
            const int T = threads;
            std::vector<std::vector<size_t>> part((size_t)T);
            std::vector<int64_t> pmax((size_t)T, -1);
#pragma omp parallel for num_threads(threads) schedule(static)
            for (int t = 0; t < T; t++) {
                const size_t lo = nRec * (size_t)t / T, hi = nRec * (size_t)(t + 1) / T;
                int64_t mx = -1;
                for (size_t r = lo; r < hi; r++) {
                    uint32_t pos; int64_t id; recOf(r, pos, id);
                    if (isSep(pos, id)) { part[(size_t)t].push_back(r); continue; }
                    const int32_t id32 = (int32_t)id;                                     // junctionstorage.h:129,148
                    const int64_t a = id32 < 0 ? -(int64_t)id32 : id32;
                    if (a > mx) mx = a;
                }
                pmax[(size_t)t] = mx;
            }
            for (int t = 0; t < T; t++) { seps.insert(seps.end(), part[(size_t)t].begin(), part[(size_t)t].end()); maxAbs = std::max(maxAbs, pmax[(size_t)t]); }
        }
        // record ranges of the chromosomes; an empty run in front of a later junction is what the reference cannot represent
        std::vector<std::pair<size_t, size_t>> chrRec;
        {
            size_t from = 0;
            bool pendingEmpty = false;
            for (size_t q = 0; q <= seps.size(); q++) {
                const size_t to = q < seps.size() ? seps[q] : nRec;
                if (to > from) {
                    if (pendingEmpty) throw LcbError("the junction file has a chromosome without junctions (unsupported)");
                    chrRec.emplace_back(from, to);
                } else if (q < seps.size()) pendingEmpty = true;
                from = to + 1;
            }
        }
        g->nVertex = (uint32_t)(maxAbs + 1);
        const size_t C = chrRec.size();
        // abundance per |id| (pass 1), then the records kept per chromosome (pass 2, junctionstorage.h:597-617: abundance < threshold)
        std::vector<uint32_t> abund((size_t)g->nVertex, 0);
#pragma omp parallel for num_threads(threads) schedule(static)
        for (int64_t r = 0; r < (int64_t)nRec; r++) {
            uint32_t pos; int64_t id; recOf((size_t)r, pos, id);
            if (isSep(pos, id)) continue;
            const int32_t id32 = (int32_t)id;
            const int64_t a = id32 < 0 ? -(int64_t)id32 : id32;
#pragma omp atomic
            abund[(size_t)a]++;
        }
        // chromosomes are cut into blocks of records so that all threads share a long one
        struct Block { uint32_t chr; size_t from, to; uint64_t kept; };
        std::vector<Block> blocks;
        for (size_t c = 0; c < C; c++)
            for (size_t a = chrRec[c].first; a < chrRec[c].second; a += (1u << 20)) blocks.push_back(Block{(uint32_t)c, a, std::min(chrRec[c].second, a + (1u << 20)), 0});
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
        for (int64_t bI = 0; bI < (int64_t)blocks.size(); bI++) {
            Block& bl = blocks[(size_t)bI];
            uint64_t kept = 0;
            for (size_t r = bl.from; r < bl.to; r++) {
                uint32_t pos; int64_t id; recOf(r, pos, id);
                const int32_t id32 = (int32_t)id;
                const int64_t a = id32 < 0 ? -(int64_t)id32 : id32;
                if (abund[(size_t)a] < (uint32_t)abundance) kept++;
            }
            bl.kept = kept;
        }
        g->chrStart.assign(C + 1, 0);
        std::vector<uint64_t> blockAt(blocks.size() + 1, 0);
        for (size_t bI = 0; bI < blocks.size(); bI++) { blockAt[bI + 1] = blockAt[bI] + blocks[bI].kept; g->chrStart[blocks[bI].chr + 1] = blockAt[bI + 1]; }
        for (size_t c = 0; c < C; c++) if (g->chrStart[c + 1] < g->chrStart[c]) g->chrStart[c + 1] = g->chrStart[c];   // (a chromosome whose junctions were all filtered)
        const uint64_t P = blockAt[blocks.size()];
        // The reference's limit is per chromosome (uint32_t idx / pos, junctionstorage.h:120-151; README.md:25-26), not on the input:
        // positions are (segment, 32-bit offset) pairs on the device (lcb_segments.h), the host tables are 64-bit
        for (size_t c = 0; c < C; c++)
            if (g->chrStart[c + 1] - g->chrStart[c] >= LCB_SEG_POSITIONS) throw LcbError("a chromosome with 2^32 - 2^20 or more junctions is not supported (the reference's own limit is 2^32 bp, hence fewer than 2^32 junctions, per chromosome)");
        g->posId.resize(P); g->posPos.resize(P);
        bool unordered = false;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
        for (int64_t bI = 0; bI < (int64_t)blocks.size(); bI++) {
            const Block& bl = blocks[(size_t)bI];
            uint64_t at = blockAt[(size_t)bI];
            for (size_t r = bl.from; r < bl.to; r++) {
                uint32_t pos; int64_t id; recOf(r, pos, id);
                const int32_t id32 = (int32_t)id;
                const int64_t a = id32 < 0 ? -(int64_t)id32 : id32;
                if (abund[(size_t)a] < (uint32_t)abundance) { g->posId[at] = id32; g->posPos[at] = pos; at++; }
            }
        }
#pragma omp parallel for num_threads(threads) schedule(static) reduction(|| : unordered)
        for (int64_t c = 0; c < (int64_t)C; c++)
            for (uint64_t i = g->chrStart[c] + 1; i < g->chrStart[c + 1]; i++) if (g->posPos[i] <= g->posPos[i - 1]) unordered = true;
        if (unordered) throw LcbError("junction positions must strictly increase within a chromosome");
        { std::vector<uint32_t>().swap(abund); }
        raw.clear(); raw.shrink_to_fit();
        // FASTA (junctionstorage.h:620-633)
        g->seq.resize(C);
        size_t record = 0;
        std::string header;
        for (auto& f : fasta) parseFasta(f, *g, record, header, threads);
        if (record != C) throw LcbError("the FASTA input has fewer records than the junction file has chromosomes");
        // ch / revCh per occurrence (junctionstorage.h:635-644)
        g->posCh.resize(P); g->posRevCh.resize(P);
        bool bad = false;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1) reduction(|| : bad)
        for (int64_t c = 0; c < (int64_t)C; c++) {
            const std::string& s = g->seq[c];
            for (uint64_t i = g->chrStart[c]; i < g->chrStart[c + 1]; i++) {
                const uint64_t p = g->posPos[i];
                if (p + (uint64_t)k > s.size()) { bad = true; g->posCh[i] = 0; g->posRevCh[i] = 'N'; continue; }
                g->posCh[i] = (uint8_t)(p + k < s.size() ? s[p + k] : '\0');             // std::string terminator
                g->posRevCh[i] = (uint8_t)(p > 0 ? reverseChar(s[p - 1]) : 'N');
            }
        }
        if (bad) throw LcbError("a junction lies beyond the end of its sequence (junction file and FASTA do not match)");
        // CSR over |id| (replaces vertex_ + the per-vertex std::sort, junctionstorage.h:646-649): counts and slots are handed
        // out with atomics by all threads, then every vertex's short list is sorted by g, i.e. by (chr, idx).
        g->occStart.assign((size_t)g->nVertex + 1, 0);
#pragma omp parallel for num_threads(threads) schedule(static)
        for (int64_t i = 0; i < (int64_t)P; i++) {
            const int32_t id = g->posId[i];
#pragma omp atomic
            g->occStart[(size_t)(id < 0 ? -(int64_t)id : id) + 1]++;
        }
        for (size_t v = 0; v < g->nVertex; v++) g->occStart[v + 1] += g->occStart[v];
        g->occG.resize(P); g->occChr.resize(P);
        {
            std::vector<uint64_t> cursor(g->occStart.begin(), g->occStart.end() - 1);
#pragma omp parallel for num_threads(threads) schedule(static)
            for (int64_t i = 0; i < (int64_t)P; i++) {
                const int32_t id = g->posId[i];
                uint64_t at;
#pragma omp atomic capture
                at = cursor[(size_t)(id < 0 ? -(int64_t)id : id)]++;
                g->occG[at] = (uint64_t)i;
            }
        }
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4096)
        for (int64_t v = 0; v < (int64_t)g->nVertex; v++) {
            const uint64_t a = g->occStart[(size_t)v], b2 = g->occStart[(size_t)v + 1];
            if (b2 - a > 1) std::sort(g->occG.begin() + (ptrdiff_t)a, g->occG.begin() + (ptrdiff_t)b2);
            for (uint64_t j = a; j < b2; j++)
                g->occChr[j] = (uint32_t)(std::upper_bound(g->chrStart.begin(), g->chrStart.end(), g->occG[j]) - g->chrStart.begin() - 1);
        }
    } catch (...) {
        delete g;
        throw;
    }
    return g;
}
