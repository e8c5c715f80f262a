// lcb_segments.h — how the 64-bit flat positions of the host tables map to the (segment, 32-bit offset) positions of the device code.
//
// The reference bounds a CHROMOSOME by 2^32 (uint32_t idx / pos per chromosome, junctionstorage.h:120-151; README.md:25-26) and the
// input not at all. The kernels keep 32-bit arithmetic for everything a path instance does inside its chromosome (lcb_kernel.h) by
// cutting the flat position space into SEGMENTS: runs of whole chromosomes with at most `cap` positions each. A position on the
// device is (segment, g) with flat index segDev[segment] + g; inputs below 2^32 positions are one segment with base 0.
//
// `gap` (tests only) puts that many unused positions between two segments of the DEVICE tables, so that the flat indices of a small
// input exceed 2^32 and every 64-bit address computation of the kernels runs on the goldens: the host keeps its dense positions,
// toDev / toHost translate at the boundary (marks and predicted marks down, footprints up).
#ifndef LCB_SEGMENTS_H
#define LCB_SEGMENTS_H

#include <algorithm>
#include <cstdint>
#include <vector>

#include "lcb_host.h"
#include "lcb_kernel_limits.h"

struct LcbSegPlan {
    std::vector<uint64_t> segStart;      // [nSeg + 1] host: flat index of the first position of every segment (and the end of the last)
    std::vector<uint64_t> segDev;        // [LCB_MAX_SEG] device: flat index of g = 0 of every segment (unused entries 0)
    std::vector<uint32_t> chrWord;       // [C] (segment << LCB_SEG_SHIFT) | chromosome
    std::vector<uint32_t> chrLo, chrHi;  // [C] the chromosome's positions inside its segment: g in [lo, hi)
    std::vector<uint64_t> chrDev;        // [C] device flat index of the chromosome's first position
    uint64_t gap = 0;
    uint64_t devPositions = 0;           // length of the device tables
    uint32_t nSeg() const { return (uint32_t)segStart.size() - 1; }
    uint32_t segOfHost(uint64_t f) const { return (uint32_t)(std::upper_bound(segStart.begin() + 1, segStart.end() - 1, f) - (segStart.begin() + 1)); }
    // a position of the host tables (or the end of a range that starts in the same segment: pass the segment of its start)
    uint64_t toDev(uint64_t f) const { return gap ? f + (uint64_t)segOfHost(f) * gap : f; }
    void rangeToDev(uint64_t lo, uint64_t hi, uint64_t& dlo, uint64_t& dhi) const { dlo = toDev(lo); dhi = dlo + (hi - lo); }
    uint64_t toHost(uint64_t d) const
    {
        if (!gap) return d;
        uint32_t s = nSeg() - 1;
        while (s > 0 && segStart[s] + (uint64_t)s * gap > d) s--;
        return d - (uint64_t)s * gap;
    }
};

// cap: most positions of a segment (0 = LCB_SEG_POSITIONS); a chromosome longer than cap gets a segment of its own
inline LcbSegPlan lcb_plan_segments(const lcb_graph& g, uint64_t cap, uint64_t gap)
{
    LcbSegPlan p;
    const size_t C = g.nChr();
    if (!cap || cap > LCB_SEG_POSITIONS) cap = LCB_SEG_POSITIONS;
    if (C >= (1u << LCB_SEG_SHIFT)) throw LcbError("more than 2^24 chromosomes are not supported by the device tables");
    // (test hook: a gap leaves the last bitmap word of a segment and the first one of the next to themselves - lcb_device_set_used copies whole words per segment)
    if (gap && gap < 64) throw LcbError("lcb_device_opts.seg_gap: a gap between the segments of the device tables has at least 64 positions");
    p.gap = gap;
    p.chrWord.resize(C); p.chrLo.resize(C); p.chrHi.resize(C); p.chrDev.resize(C);
    p.segStart.assign(1, 0);
    uint64_t segBegin = 0;
    for (size_t c = 0; c < C; c++) {
        const uint64_t a = g.chrStart[c], b = g.chrStart[c + 1];
        if (b - a >= LCB_SEG_POSITIONS) throw LcbError("a chromosome with 2^32 - 2^20 or more junctions is not supported (the reference's own limit is 2^32 bp, hence fewer than 2^32 junctions, per chromosome)");
        if (b - segBegin > cap && a > segBegin) { p.segStart.push_back(a); segBegin = a; }
        const uint32_t s = (uint32_t)p.segStart.size() - 1;
        if (s >= LCB_MAX_SEG) throw LcbError("more than 32 segments of 2^32 junction occurrences (2^37 in total) are not supported by the device tables");
        p.chrWord[c] = (s << LCB_SEG_SHIFT) | (uint32_t)c;
        p.chrLo[c] = (uint32_t)(a - segBegin); p.chrHi[c] = (uint32_t)(b - segBegin);
        p.chrDev[c] = a + (uint64_t)s * gap;
    }
    p.segStart.push_back(g.nPos());
    p.segDev.assign(LCB_MAX_SEG, 0);
    for (uint32_t s = 0; s < p.nSeg(); s++) p.segDev[s] = p.segStart[s] + (uint64_t)s * gap;
    p.devPositions = g.nPos() + (uint64_t)(p.nSeg() - 1) * gap;
    if (p.devPositions >= (1ull << 37) - (1ull << 30)) throw LcbError("the device tables hold fewer than 2^37 positions");
    return p;
}

// The look-ahead window of every position (round 6): MostPopularVertex walks from a voter's end while `step < lookingDepth ||
// |pos - pos0| <= maxBranchSize` (blocksfinder.h:722-727). Positions ascend strictly inside a chromosome, so the steps that satisfy the
// distance bound are a prefix: win[q] = (steps backward << 16) | steps forward that stay inside q's chromosome and within maxBranch bp
// of pos[q]. With it a voter's window length is known before anything is walked - min(steps to the chromosome end, max(depth - 1,
// win)) - and the walk needs neither Position::pos nor a step-by-step loop test (lcb_vote_walk). A property of (graph, maxBranch);
// host positions, uploaded like the other per-position tables.
inline std::vector<uint32_t> lcb_window_table(const lcb_graph& g, uint32_t maxBranch)
{
    const uint64_t P = g.nPos();
    std::vector<uint32_t> win((size_t)P);
    const int64_t C = (int64_t)g.nChr();
    const uint64_t BLK = 1u << 16;
    const int64_t nBlk = (int64_t)((P + BLK - 1) / BLK);
    #pragma omp parallel for schedule(dynamic, 4)
    for (int64_t bk = 0; bk < nBlk; bk++) {
        const uint64_t q0 = (uint64_t)bk * BLK, q1 = std::min(P, q0 + BLK);
        int64_t c = (int64_t)(std::upper_bound(g.chrStart.begin(), g.chrStart.end(), q0) - g.chrStart.begin()) - 1;
        for (uint64_t q = q0; q < q1; q++) {
            while (c + 1 < C && g.chrStart[(size_t)c + 1] <= q) c++;
            const uint64_t s = g.chrStart[(size_t)c], e = g.chrStart[(size_t)c + 1];
            const uint64_t pq = g.posPos[q];
            // forward: the last j in [q, e) with pos[j] <= pos[q] + maxBranch (positions differ by at least one per step)
            const uint64_t fe = std::min(e, q + (uint64_t)maxBranch + 1);
            const uint64_t f = (uint64_t)(std::upper_bound(g.posPos.begin() + q, g.posPos.begin() + fe, (uint32_t)std::min<uint64_t>(pq + maxBranch, 0xFFFFFFFFull)) - (g.posPos.begin() + q)) - 1;
            // backward: the first j in [s, q] with pos[j] >= pos[q] - maxBranch
            const uint64_t bs = q - s > (uint64_t)maxBranch ? q - maxBranch : s;
            const uint64_t b = q - (uint64_t)(std::lower_bound(g.posPos.begin() + bs, g.posPos.begin() + q + 1, (uint32_t)(pq > maxBranch ? pq - maxBranch : 0)) - g.posPos.begin());
            win[(size_t)q] = (uint32_t)std::min<uint64_t>(f, 0xFFFF) | ((uint32_t)std::min<uint64_t>(b, 0xFFFF) << 16);
        }
    }
    return win;
}

#endif
