// bundles.cpp — seed (bundle) enumeration and ordering.
//
// Replaces the single-threaded std::set/std::map loop of BlocksFinder::FindBlocks
// (blocksfinder.h:461-503) and the std::sort at blocksfinder.h:517. Every signed vertex v is
// independent, so vertices are processed by an OpenMP team over the CSR occurrence lists; the
// sort key (count desc, rank asc, resolve asc — Bundle::operator<, blocksfinder.h:195-208) is
// a total order (distinct resolve per bundle), so the result does not depend on the algorithm: the
// sort runs on all threads too (sorted runs + pairwise merges).
#include <omp.h>

#include <algorithm>
#include <vector>

#include "lcb_host.h"

namespace {

inline bool seedLess(const lcb_seed& a, const lcb_seed& b)
{
    if (a.count != b.count) return a.count > b.count;
    if (a.rank != b.rank) return a.rank < b.rank;
    if (a.resolve_pos != b.resolve_pos) return a.resolve_pos < b.resolve_pos;
    return a.resolve_chr < b.resolve_chr;
}

}  // namespace

void lcb_enumerate_seeds_impl(const lcb_graph& g, int threads, std::vector<lcb_seed>& out)
{
    if (threads < 1) threads = 1;
    const int64_t V = g.nVertex;
    std::vector<std::vector<lcb_seed>> part((size_t)threads);
#pragma omp parallel num_threads(threads)
    {
        std::vector<lcb_seed>& mine = part[(size_t)omp_get_thread_num()];
#pragma omp for schedule(dynamic, 4096)
        for (int64_t v = -V + 1; v < V; v++) {
            const uint32_t av = (uint32_t)(v < 0 ? -v : v);
            const uint64_t o0 = g.occStart[av], o1 = g.occStart[av + 1];
            if (o1 - o0 < 2) continue;                                   // count > 1 needs two occurrences
            uint32_t count[256] = {0};
            bool good[256] = {false};
            unsigned char seen[8]; int nSeen = 0;                        // distinct characters, tiny
            unsigned char overflow[256]; int nOver = 0;
            for (uint64_t j = o0; j < o1; j++) {
                const uint64_t p = g.occG[j];
                const bool positive = g.posId[p] == (int32_t)v;          // JunctionIterator::IsPositiveStrand
                const unsigned char ch = positive ? g.posCh[p] : g.posRevCh[p];
                if (count[ch]++ == 0) { if (nSeen < 8) seen[nSeen++] = ch; else overflow[nOver++] = ch; }
                if (positive) good[ch] = true;
            }
            for (int s = 0; s < nSeen + nOver; s++) {
                const unsigned char ch = s < nSeen ? seen[s] : overflow[s - nSeen];
                if (count[ch] > 1 && good[ch]) {
                    lcb_seed b;
                    b.vid = (int32_t)v; b.ch = (int32_t)(signed char)ch; b.count = count[ch];
                    b.rank = 0; b.resolve_pos = UINT64_MAX; b.resolve_chr = UINT64_MAX;
                    uint64_t base = 1;
                    for (uint64_t j = o0; j < o1; j++) {
                        const uint64_t p = g.occG[j];
                        const bool positive = g.posId[p] == (int32_t)v;
                        const unsigned char c2 = positive ? g.posCh[p] : g.posRevCh[p];
                        if (c2 != ch) continue;
                        b.rank += (uint64_t)g.occChr[j] * base;          // size_t wrap-around as in the reference
                        base *= 31;
                        if (positive) {
                            const uint64_t rp = g.posPos[p], rc = g.occChr[j];
                            if (rp < b.resolve_pos || (rp == b.resolve_pos && rc < b.resolve_chr)) { b.resolve_pos = rp; b.resolve_chr = rc; }
                        }
                    }
                    mine.push_back(b);
                }
            }
        }
    }
    size_t total = 0;
    for (auto& p : part) total += p.size();
    out.clear();
    out.reserve(total);
    for (auto& p : part) out.insert(out.end(), p.begin(), p.end());
    // blocksfinder.h:517 on all threads: sorted runs, then pairwise merges (the order is total, so the result is std::sort's)
    const size_t n = out.size();
    int runs = 1;
    while (runs * 2 <= threads && n / (size_t)(runs * 2) >= 65536) runs *= 2;
    auto cut = [&](int r) { return out.begin() + (ptrdiff_t)(n * (size_t)r / (size_t)runs); };
#pragma omp parallel for num_threads(threads) schedule(static, 1)
    for (int r = 0; r < runs; r++) std::sort(cut(r), cut(r + 1), seedLess);
    for (int w = 1; w < runs; w *= 2) {
#pragma omp parallel for num_threads(threads) schedule(static, 1)
        for (int r = 0; r < runs; r += 2 * w) std::inplace_merge(cut(r), cut(r + w), cut(r + 2 * w), seedLess);
    }
}
