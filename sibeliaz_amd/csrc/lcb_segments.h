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
    p.gap = gap;
    p.chrWord.resize(C); p.chrLo.resize(C); p.chrHi.resize(C); p.chrDev.resize(C);
    p.segStart.assign(1, 0);
    uint64_t segBegin = 0;
    for (size_t c = 0; c < C; c++) {
        const uint64_t a = g.chrStart[c], b = g.chrStart[c + 1];
        if (b - a >= LCB_SEG_POSITIONS) throw LcbError("a chromosome with 2^32 or more junctions is not supported");
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

#endif
