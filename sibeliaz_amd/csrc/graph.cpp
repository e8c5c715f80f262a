// graph.cpp — loads a TwoPaCo junction file + FASTA into structure-of-arrays tables.
//
// Replaces JunctionStorage::Init (junctionstorage.h:572-650), JunctionPositionReader
// (common/junctionapi.h:80-98) and StreamFastaParser (common/streamfastaparser.cpp:28-92).
// Same observable semantics (abundance filter, idx after filtering, ch/revCh definition, FASTA
// header token, upper-casing, validation and error texts), different construction: both inputs
// are read with one bulk read each, the per-vertex occurrence lists become a CSR built by a
// counting sort, and the nested vectors become flat arrays indexed by g = chrStart[chr] + idx.
//
// Inputs the reference silently mis-handles (out-of-bounds access) are rejected loudly here:
// a chromosome without junctions, a FASTA record count different from the chromosome count, a
// junction beyond the end of its sequence, positions that do not increase.
#include <omp.h>

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

void parseFasta(const std::string& file, lcb_graph& g, size_t& record, std::string& header)
{
    std::vector<char> buf;
    if (!readAll(file, buf)) throw LcbError("Can't open file " + file);                 // streamfastaparser.cpp:24
    size_t i = 0;
    const size_t n = buf.size();
    while (i < n) {
        if (buf[i] != '>')                                                               // streamfastaparser.cpp:33-36
            throw LcbError("The FASTA header should start with a '>', started with '" + std::string(1, buf[i]) + "'");
        i++;
        size_t e = i;
        while (e < n && buf[e] != '\n') e++;
        if (e < n) {                                                                     // `ss >> currentHeader_` only on '\n'
            size_t s = i;
            while (s < e && isspace((unsigned char)buf[s])) s++;
            size_t t = s;
            while (t < e && !isspace((unsigned char)buf[t])) t++;
            if (t > s) header.assign(buf.data() + s, t - s);
            i = e + 1;
        } else i = n;
        if (record >= g.seq.size()) throw LcbError("the FASTA input has more records than the junction file has chromosomes");
        g.chrName.push_back(header);
        std::string& seq = g.seq[record];
        size_t end = i;
        while (end < n && buf[end] != '>') end++;                                        // a '>' anywhere ends the record
        seq.reserve(end - i);
        for (; i < end; i++) {
            const unsigned char c = (unsigned char)buf[i];
            if (isspace(c)) continue;
            const unsigned char u = (unsigned char)toupper(c);
            if (!kValid.v[u])                                                            // streamfastaparser.cpp:80-83
                throw LcbError("Found an invalid character '" + std::string(1, (char)c) + "' in sequence " + header);
            seq.push_back((char)u);
        }
        record++;
    }
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
        // pass 1 (junctionstorage.h:576-594): chromosome of every record, abundance per |id|, V
        std::vector<uint32_t> abund;
        std::vector<uint64_t> chrCount;
        uint32_t chr = 0;
        int64_t maxAbs = -1;
        for (size_t r = 0; r < nRec; r++) {
            uint32_t pos; int64_t id;
            memcpy(&pos, raw.data() + r * 12, 4);
            memcpy(&id, raw.data() + r * 12 + 4, 8);
            if (pos == 0xFFFFFFFFu || id == INT64_MAX) { chr++; continue; }               // junctionapi.h:93
            const int32_t id32 = (int32_t)id;                                             // junctionstorage.h:129,148
            const int64_t a = id32 < 0 ? -(int64_t)id32 : id32;
            if (a > maxAbs) { maxAbs = a; abund.resize((size_t)a + 1, 0); }
            abund[(size_t)a]++;
            if (chr >= chrCount.size()) {
                if (chr > chrCount.size()) throw LcbError("the junction file has a chromosome without junctions (unsupported)");
                chrCount.push_back(0);
            }
        }
        g->nVertex = (uint32_t)(maxAbs + 1);
        // pass 2 (junctionstorage.h:597-617): keep abundance < threshold
        chr = 0;
        for (size_t r = 0; r < nRec; r++) {
            uint32_t pos; int64_t id;
            memcpy(&pos, raw.data() + r * 12, 4);
            memcpy(&id, raw.data() + r * 12 + 4, 8);
            if (pos == 0xFFFFFFFFu || id == INT64_MAX) { chr++; continue; }
            const int32_t id32 = (int32_t)id;
            const int64_t a = id32 < 0 ? -(int64_t)id32 : id32;
            if (abund[(size_t)a] < (uint32_t)abundance) chrCount[chr]++;
        }
        const size_t C = chrCount.size();
        g->chrStart.assign(C + 1, 0);
        for (size_t c = 0; c < C; c++) g->chrStart[c + 1] = g->chrStart[c] + chrCount[c];
        const uint64_t P = g->chrStart[C];
        if (P >= 0xFFFFFF00ull) throw LcbError("more than 2^32 junction occurrences are not supported");
        g->posId.resize(P); g->posPos.resize(P);
        {
            chr = 0;
            uint64_t at = 0;
            for (size_t r = 0; r < nRec; r++) {
                uint32_t pos; int64_t id;
                memcpy(&pos, raw.data() + r * 12, 4);
                memcpy(&id, raw.data() + r * 12 + 4, 8);
                if (pos == 0xFFFFFFFFu || id == INT64_MAX) { chr++; continue; }
                const int32_t id32 = (int32_t)id;
                const int64_t a = id32 < 0 ? -(int64_t)id32 : id32;
                if (abund[(size_t)a] < (uint32_t)abundance) {
                    if (at > g->chrStart[chr] && pos <= g->posPos[at - 1])
                        throw LcbError("junction positions must strictly increase within a chromosome");
                    g->posId[at] = id32; g->posPos[at] = pos; at++;
                }
            }
        }
        raw.clear(); raw.shrink_to_fit();
        // FASTA (junctionstorage.h:620-633)
        g->seq.resize(C);
        size_t record = 0;
        std::string header;
        for (auto& f : fasta) parseFasta(f, *g, record, header);
        if (record != C) throw LcbError("the FASTA input has fewer records than the junction file has chromosomes");
        // ch / revCh per occurrence (junctionstorage.h:635-644)
        g->posCh.resize(P); g->posRevCh.resize(P);
        bool bad = false;
#pragma omp parallel for num_threads(threads) schedule(static)
        for (int64_t c = 0; c < (int64_t)C; c++) {
            const std::string& s = g->seq[c];
            for (uint64_t i = g->chrStart[c]; i < g->chrStart[c + 1]; i++) {
                const uint64_t p = g->posPos[i];
                if (p + (uint64_t)k > s.size()) { bad = true; continue; }
                g->posCh[i] = (uint8_t)(p + k < s.size() ? s[p + k] : '\0');             // std::string terminator
                g->posRevCh[i] = (uint8_t)(p > 0 ? reverseChar(s[p - 1]) : 'N');
            }
        }
        if (bad) throw LcbError("a junction lies beyond the end of its sequence (junction file and FASTA do not match)");
        // CSR over |id| (replaces vertex_ + the per-vertex std::sort, junctionstorage.h:646-649): a counting
        // sort over ascending g keeps every list ordered by (chr, idx).
        g->occStart.assign((size_t)g->nVertex + 1, 0);
        for (uint64_t i = 0; i < P; i++) {
            const int32_t id = g->posId[i];
            g->occStart[(size_t)(id < 0 ? -(int64_t)id : id) + 1]++;
        }
        for (size_t v = 0; v < g->nVertex; v++) g->occStart[v + 1] += g->occStart[v];
        g->occG.resize(P); g->occChr.resize(P);
        std::vector<uint32_t> cursor(g->occStart.begin(), g->occStart.end() - 1);
        for (size_t c = 0; c < C; c++) {
            for (uint64_t i = g->chrStart[c]; i < g->chrStart[c + 1]; i++) {
                const int32_t id = g->posId[i];
                const uint32_t at = cursor[(size_t)(id < 0 ? -(int64_t)id : id)]++;
                g->occG[at] = (uint32_t)i; g->occChr[at] = (uint32_t)c;
            }
        }
    } catch (...) {
        delete g;
        throw;
    }
    return g;
}
