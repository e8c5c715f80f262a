// commit.cpp — ordered commit of one phase of per-seed results.
//
// Restates the thread-0 section of ProcessVertex::operator() (blocksfinder.h:372-427) and
// ProcessVertex::Finalize (blocksfinder.h:312-332) over the flat tables: a result is a list of
// instances (chr, front idx, back idx, strand); committing it assigns ++blocksFound_, appends
// one BlockInstance per instance and sets the `used` bits of [Front, Back). On both strands that
// range is the bit range [min(front,back), max(front,back)) over g, because `used` marks the
// edge idx -> idx+1 (junctionstorage.h:270-295).
//
// The weak conflict check is kept exactly: only chromosomes committed to
// earlier in the SAME phase (invalidChr_) are examined; a conflicting seed is re-processed
// against the live state through the callback (on the GPU — there is no CPU fallback here).
#include <cstring>
#include <vector>

#include "lcb_host.h"

lcb_committer::lcb_committer(const lcb_graph* graph, const lcb_params& prm) : g(graph), p(prm)
{
    used.assign((size_t)(g->nPos() / 32 + 2), 0u);
    invalidChr.assign(g->nChr() + 1, 0);
}

bool lcb_committer::anyUsed(uint64_t lo, uint64_t hi) const
{
    if (hi <= lo) return false;
    const uint64_t w0 = lo >> 5, w1 = (hi - 1) >> 5;
    for (uint64_t w = w0; w <= w1; w++) {
        uint32_t mask = 0xFFFFFFFFu;
        if (w == w0) mask &= 0xFFFFFFFFu << (lo & 31);
        if (w == w1) mask &= 0xFFFFFFFFu >> (31 - ((hi - 1) & 31));
        if (used[w] & mask) return true;
    }
    return false;
}

bool lcb_committer::allUsed(uint64_t lo, uint64_t hi) const
{
    if (hi <= lo) return true;
    const uint64_t w0 = lo >> 5, w1 = (hi - 1) >> 5;
    for (uint64_t w = w0; w <= w1; w++) {
        uint32_t mask = 0xFFFFFFFFu;
        if (w == w0) mask &= 0xFFFFFFFFu << (lo & 31);
        if (w == w1) mask &= 0xFFFFFFFFu >> (31 - ((hi - 1) & 31));
        if ((used[w] & mask) != mask) return false;
    }
    return true;
}

void lcb_committer::finalize(const lcb_instance* inst, uint64_t n)                       // blocksfinder.h:312-332
{
    const int64_t currentBlock = ++blocksFound;
    for (uint64_t i = 0; i < n; i++) {
        const lcb_instance& in = inst[i];
        if (!invalidChr[in.chr]) { invalidChr[in.chr] = 1; invalidList.push_back(in.chr); }
        const uint64_t base = g->chrStart[in.chr];
        const uint64_t fp = g->posPos[base + in.front_idx], bp = g->posPos[base + in.back_idx];
        lcb_block b;
        b.chr = in.chr;
        if (in.positive) { b.id = (int32_t)currentBlock; b.start = fp; b.end = bp + (uint64_t)p.k; }        // [Front.pos, Back.pos + k)
        else { b.id = (int32_t)-currentBlock; b.start = bp; b.end = fp + (uint64_t)p.k; }                     // [Back.GetPosition() - k, Front.GetPosition())
        blocks.push_back(b);
        const uint64_t lo = base + (in.front_idx < in.back_idx ? in.front_idx : in.back_idx);
        const uint64_t hi = base + (in.front_idx < in.back_idx ? in.back_idx : in.front_idx);
        if (hi > lo) {
            const uint64_t w0 = lo >> 5, w1 = (hi - 1) >> 5;                // word-wise: blocks are thousands of positions long
            for (uint64_t w = w0; w <= w1; w++) {
                uint32_t mask = 0xFFFFFFFFu;
                if (w == w0) mask &= 0xFFFFFFFFu << (lo & 31);
                if (w == w1) mask &= 0xFFFFFFFFu >> (31 - ((hi - 1) & 31));
                used[w] |= mask;
            }
            marks.push_back(lo); marks.push_back(hi);
        }
    }
}

bool lcb_committer::conflicts(const lcb_instance* r, uint64_t cnt) const                  // blocksfinder.h:377-398
{
    for (uint64_t i = 0; i < cnt; i++) {
        if (!invalidChr[r[i].chr]) continue;
        const uint64_t base = g->chrStart[r[i].chr];
        const uint64_t lo = base + (r[i].front_idx < r[i].back_idx ? r[i].front_idx : r[i].back_idx);
        const uint64_t hi = base + (r[i].front_idx < r[i].back_idx ? r[i].back_idx : r[i].front_idx);
        if (anyUsed(lo, hi)) return true;
    }
    return false;
}

void lcb_committer::endPhase()                                                           // blocksfinder.h:416
{
    for (uint32_t c : invalidList) invalidChr[c] = 0;
    invalidList.clear();
}

void lcb_committer::commitPhase(const lcb_seed* seeds, int64_t n, const uint64_t* offsets, const lcb_instance* inst,
                                lcb_reprocess_fn fn, void* user)
{
    std::vector<lcb_instance> redo;
    for (int64_t s = 0; s < n; s++) {
        const uint64_t cnt = offsets[s + 1] - offsets[s];
        if (cnt <= 1) continue;                                                          // blocksfinder.h:375
        const lcb_instance* r = inst + offsets[s];
        const bool isGood = !conflicts(r, cnt);
        if (isGood) finalize(r, cnt);
        else {                                                                           // blocksfinder.h:404-412
            failures++;
            if (!fn) throw LcbError("commit conflict but no re-process callback was supplied");
            uint64_t cap = redo.size() < 1024 ? 1024 : redo.size(), got = 0;
            for (;;) {
                redo.resize(cap);
                const int rc = fn(user, &seeds[s], redo.data(), cap, &got);
                if (rc != 0) throw LcbError(std::string("re-process callback failed: ") + lcb_last_error());
                if (got <= cap) break;
                cap = got;
            }
            if (got > 1) finalize(redo.data(), got);
        }
    }
    endPhase();
}
