// commit_check — TEST-ONLY: the device-resident commit kernel body (lcb_commit_body, lcb_kernel.h) on the wavefront emulator against a
// plain sequential restatement of what it has to do (the thread-0 section of ProcessVertex::operator(), blocksfinder.h:372-414, over
// a round's results: phase-start validation against the round's marks, weak conflict check, MarkUsed), on random rounds:
//   commit_check [cases] [seed]
// Random chromosomes, pre-marked bits, live seeds with 0-5 instances and footprints around them, dead seeds, tiny phases (so that a
// case has many phase boundaries, conflicts inside phases and void phase-start results). The results of a round arrive in 1-4
// "launches" (a random share of the seeds gets its final result only from a later one, like seeds that overflowed their kernel
// variant) and the kernel body is invoked behind each, carrying its state from one invocation to the next; the outcome must be the
// one of the sequential walk over the complete round. Compared: the committed list, where and why it stopped, the live bitmap and the
// list of the ranges the round marked (what a later invocation rebuilds its summary from).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "emu_runtime.h"
#include "lcb_kernel.h"

namespace {

bool anyBit(const std::vector<uint32_t>& b, uint32_t lo, uint32_t hi) { for (uint32_t q = lo; q < hi; q++) if ((b[q >> 5] >> (q & 31)) & 1u) return true; return false; }
void setBits(std::vector<uint32_t>& b, uint32_t lo, uint32_t hi) { for (uint32_t q = lo; q < hi; q++) b[q >> 5] |= 1u << (q & 31); }

}  // namespace

int main(int argc, char** argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 300;
    std::mt19937 rng(argc > 2 ? (unsigned)atoi(argv[2]) : 12345u);
    auto rnd = [&](uint32_t n) { return n ? (uint32_t)(rng() % n) : 0u; };
    int bad = 0;
    long commits = 0, stops1 = 0, stops2 = 0, clean = 0, waits = 0;
    for (int c = 0; c < cases; c++) {
        const uint32_t nChr = 1 + rnd(6), phase = 1u << rnd(4);                       // 1 .. 8 seeds per phase
        std::vector<uint32_t> chrStart(1, 0);
        for (uint32_t i = 0; i < nChr; i++) chrStart.push_back(chrStart.back() + 40 + rnd(600));
        const uint32_t nPos = chrStart.back(), words = nPos / 32 + 2;
        const std::vector<uint64_t> chrBase(chrStart.begin(), chrStart.end() - 1);
        std::vector<uint32_t> used(words, 0), stamp(nChr + 1, 0);
        for (uint32_t k = rnd(4); k > 0; k--) { const uint32_t a = rnd(nPos); setBits(used, a, std::min(nPos, a + 1 + rnd(30))); }
        const uint32_t nRound = 8 + rnd(120);
        std::vector<uint32_t> seedIdx, off(1, 0), fpOff(1, 0);
        std::vector<uint4> inst;
        std::vector<LcbFpOut> fp;
        const uint32_t density = 1 + rnd(4);
        for (uint32_t s = 0; s < nRound; s++) {
            if (rnd(4) >= density) continue;                                          // a dead seed: not in the live list
            const uint32_t cnt = rnd(6);
            for (uint32_t k = 0; k < cnt; k++) {
                const uint32_t chr = rnd(nChr), len = chrStart[chr + 1] - chrStart[chr], a = rnd(len), b = std::min(len - 1, a + rnd(40));
                const bool pos = rnd(2) != 0;
                inst.push_back(uint4{chr, pos ? a : b, pos ? b : a, pos ? 1u : 0u});
                const uint32_t lo = chrStart[chr] + a, hi = chrStart[chr] + b;
                fp.push_back(LcbFpOut{lo > 5 ? lo - rnd(6) : lo, std::min(nPos - 1, hi + rnd(6))});
            }
            for (uint32_t k = rnd(3); k > 0; k--) { const uint32_t a = rnd(nPos); fp.push_back(LcbFpOut{a, std::min(nPos - 1, a + rnd(12))}); }
            if (rnd(8) == 0) fp.push_back(LcbFpOut{rnd(nPos), 0xFFFFFFFFFFF0ull});             // an over-wide interval (clamped by the kernel)
            seedIdx.push_back(s); off.push_back((uint32_t)inst.size()); fpOff.push_back((uint32_t)fp.size());
        }
        const uint32_t nLive = (uint32_t)seedIdx.size();
        if (!nLive) { c--; continue; }
        if (inst.empty()) inst.push_back(uint4{0, 0, 0, 0});
        if (fp.empty()) fp.push_back(LcbFpOut{0, 0});
        // ---- the sequential restatement (phase-start validation as the kernel does it: a footprint interval that touches a PAGE with a
        // mark of this round hands the phase over, stop kind 3 - a superset of the intervals that hold a marked bit)
        const uint32_t pageShift = 1u + (uint32_t)(c % 7);               // tiny pages: intervals span many of them
        std::vector<uint32_t> rUsed = used, rDelta(words, 0), rStamp(nChr + 1, 0), rCommitted;
        std::vector<uint8_t> rPage((nPos >> pageShift) + 2, 0);
        uint32_t rStop = nLive, rKind = 0;
        for (uint32_t lq = 0; lq < nLive && !rKind;) {
            const uint32_t ph = seedIdx[lq] / phase;
            uint32_t lqEnd = lq;
            while (lqEnd < nLive && seedIdx[lqEnd] / phase == ph) lqEnd++;
            for (uint32_t q = lq; q < lqEnd && !rKind; q++)
                for (uint32_t k = fpOff[q]; k < fpOff[q + 1]; k++)
                    { bool pg = false; for (uint32_t q2 = (uint32_t)(fp[k].lo >> pageShift); q2 <= (uint32_t)(std::min<uint64_t>(fp[k].hi, nPos - 1) >> pageShift); q2++) pg = pg || rPage[q2]; if (pg) { rStop = lq; rKind = 3; break; } }
            if (rKind) break;
            for (uint32_t q = lq; q < lqEnd; q++) {
                if (off[q + 1] - off[q] <= 1) continue;
                bool conflict = false;
                for (uint32_t k = off[q]; k < off[q + 1]; k++) {
                    const uint4 in = inst[k];
                    if (rStamp[in.x] != ph + 1) continue;
                    const uint32_t base = chrStart[in.x], lo = base + std::min(in.y, in.z), hi = base + std::max(in.y, in.z);
                    if (anyBit(rUsed, lo, hi)) conflict = true;
                }
                if (conflict) { rStop = q; rKind = 2; break; }
                for (uint32_t k = off[q]; k < off[q + 1]; k++) {
                    const uint4 in = inst[k];
                    rStamp[in.x] = ph + 1;
                    const uint32_t base = chrStart[in.x], lo = base + std::min(in.y, in.z), hi = base + std::max(in.y, in.z);
                    setBits(rUsed, lo, hi); setBits(rDelta, lo, hi);
                    if (hi > lo) for (uint32_t q2 = lo >> pageShift; q2 <= (hi - 1) >> pageShift; q2++) rPage[q2] = 1;
                }
                rCommitted.push_back(q);
            }
            lq = lqEnd;
        }
        // ---- the kernel body under the emulator (2, 4, 8 or 16 wavefronts), behind each of 1-4 launches of the round
        std::vector<uint32_t> committed(nRound, 0xFFFFFFFFu), state(LCB_CS_WORDS, 0u), roundState(nRound, LCB_RS_NONE);
        std::vector<LcbSeedOut> roundOut(nRound);
        std::vector<LcbFpOut> deltaList(64 + rnd(2000));
        uint32_t deltaCount = 0;
        const uint32_t nLaunch = 1 + rnd(4);
        std::vector<uint32_t> launchOf(nRound);                  // the launch that brings a seed's final result
        for (uint32_t i = 0; i < nRound; i++) launchOf[i] = rnd(3) == 0 ? rnd(nLaunch) : 0u;
        std::vector<int32_t> liveOf(nRound, -1);
        for (uint32_t q = 0; q < nLive; q++) liveOf[seedIdx[q]] = (int32_t)q;
        const int nw = c % 4;
        uint32_t waited = 0;
        for (uint32_t l = 0; l < nLaunch; l++) {
            for (uint32_t i = 0; i < nRound; i++) {
                if (launchOf[i] != l) continue;
                if (liveOf[i] < 0) { roundState[i] = LCB_RS_DEAD; continue; }
                const uint32_t q = (uint32_t)liveOf[i];
                LcbSeedOut o; memset(&o, 0, sizeof(o));
                o.nInst = off[q + 1] - off[q]; o.arenaOff = off[q]; o.nFp = fpOff[q + 1] - fpOff[q]; o.fpOff = fpOff[q];
                roundOut[i] = o; roundState[i] = LCB_RS_DONE;
            }
            LcbCommitArgs A;
            A.chrBase = chrBase.data(); A.used = used.data(); A.chrStamp = stamp.data();
            A.roundState = roundState.data(); A.roundOut = roundOut.data(); A.arena = inst.data(); A.fpArena = fp.data();
            A.n = nRound; A.phase = phase; A.nPos = nPos; A.state = state.data(); A.committed = committed.data();
            A.deltaList = deltaList.data(); A.deltaCount = &deltaCount; A.deltaCap = (uint32_t)deltaList.size();
            A.pageShift = pageShift;
            if (nw == 0) emu_run_block(0, 2, [&]() { lcb_commit_body<2>(A); });
            else if (nw == 1) emu_run_block(0, 4, [&]() { lcb_commit_body<4>(A); });
            else if (nw == 2) emu_run_block(0, 8, [&]() { lcb_commit_body<8>(A); });
            else emu_run_block(0, 16, [&]() { lcb_commit_body<16>(A); });
            if (l + 1 < nLaunch && !state[LCB_CS_STOPKIND] && state[LCB_CS_NEXT] < nRound) waited++;
        }
        // the restatement speaks of live indices, the kernel of indices in the round
        const uint32_t kStop = state[LCB_CS_STOPKIND] ? state[LCB_CS_STOPAT] : nRound, kKind = state[LCB_CS_STOPKIND];
        const uint32_t rStopSeed = rKind == 0 ? nRound : (rKind == 3 ? (seedIdx[rStop] / phase) * phase : seedIdx[rStop]);
        bool ok = state[LCB_CS_NCOMMITTED] == rCommitted.size() && kStop == rStopSeed && kKind == rKind && used == rUsed;
        if (ok && rKind == 0) ok = state[LCB_CS_NEXT] == nRound;
        for (size_t i = 0; ok && i < rCommitted.size(); i++) ok = committed[i] == seedIdx[rCommitted[i]];
        if (ok && deltaCount <= deltaList.size()) {              // the list of the round's marked ranges covers exactly what the round marked
            std::vector<uint32_t> cover(words, 0);
            for (uint32_t r = 0; r < deltaCount; r++) setBits(cover, (uint32_t)deltaList[r].lo, (uint32_t)deltaList[r].hi);
            ok = cover == rDelta;
        }
        waits += waited;
        if (!ok) {
            bad++;
            if (bad <= 5) fprintf(stderr, "case %d (%u seeds, %u live, phase %u, %d waves, %u launches): kernel committed %u stop %u kind %u | expected %zu stop %u kind %u | bitmap %s delta %s\n", c, nRound, nLive, phase,
                                  2 << nw, nLaunch, state[LCB_CS_NCOMMITTED], kStop, kKind, rCommitted.size(), rStopSeed, rKind, used == rUsed ? "equal" : "DIFFERENT", "-");
        }
        commits += (long)rCommitted.size(); stops1 += rKind == 3; stops2 += rKind == 2; clean += rKind == 0;
    }
    fprintf(stderr, "commit_check: %d cases, %ld commits, %ld rounds committed completely, %ld phases handed over to the host, %ld stops at a conflict, %ld invocations that had to wait for a later launch, %d mismatches\n", cases, commits, clean, stops1, stops2, waits, bad);
    return bad ? 1 : 0;
}
