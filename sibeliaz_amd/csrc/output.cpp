// output.cpp — overlap trimming, blocks_coords.gff and the <i>.tmp block-sequence chunks.
//
// Restates BlocksFinder::GenerateOutput (blocksfinder.h:605-670), ListBlocksIndicesGFF
// (blocksfinder.cpp:141-174) and ListBlocksSequences (blocksfinder.h:533-582). The reference
// orders instances inside a block by whatever permutation libstdc++'s introsort produces for
// comparators that only look at block-level keys, so the same std::sort calls
// with equivalent comparators are made here on the same sequences (the permutation depends only
// on the comparison results, not on the element layout).
#include <sys/stat.h>
#include <sys/types.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <map>
#include <string>
#include <vector>

#include "lcb_host.h"

namespace {

inline int blockId(const lcb_block& b) { return abs(b.id); }                             // BlockInstance::GetBlockId, blocksfinder.cpp:64

struct SortByMultiplicity {                                                              // blocksfinder.h:584-603
    const std::vector<int>& multiplicity;
    bool operator()(const lcb_block& a, const lcb_block& b) const
    {
        const int m1 = multiplicity[blockId(a)], m2 = multiplicity[blockId(b)];
        if (m1 != m2) return m1 > m2;
        return blockId(a) < blockId(b);
    }
};

inline bool blockLess(const lcb_block& a, const lcb_block& b)                            // BlockInstance::operator<, blocksfinder.cpp:104-107
{
    if (blockId(a) != blockId(b)) return blockId(a) < blockId(b);
    if (a.chr != b.chr) return a.chr < b.chr;
    return a.start < b.start;
}

inline bool compareById(const lcb_block& a, const lcb_block& b) { return blockId(a) < blockId(b); }   // blocksfinder.cpp:32-35

inline char reverseChar(char c)
{
    switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C'; }
    return 'N';
}

// The bases of one chromosome held by the blocks visited so far: disjoint runs [first, second) that never touch (hold() merges
// neighbours), so that both ends of a run border free positions.
class HeldRuns {
    std::map<uint64_t, uint64_t> run;
    typedef std::map<uint64_t, uint64_t>::iterator It;
    It covering(uint64_t p, bool& inside)                                                // the run with the largest first <= p
    {
        It it = run.upper_bound(p);
        if (it == run.begin()) { inside = false; return it; }
        --it;
        inside = it->second > p;
        return it;
    }

public:
    uint64_t freeAtOrAfter(uint64_t p)                                                   // smallest free position >= p
    {
        bool inside;
        It it = covering(p, inside);
        return inside ? it->second : p;
    }
    bool freeAtOrBefore(uint64_t p, uint64_t& q)                                         // largest free position <= p; false: none
    {
        bool inside;
        It it = covering(p, inside);
        if (!inside) { q = p; return true; }
        if (it->first == 0) return false;
        q = it->first - 1;
        return true;
    }
    void hold(uint64_t lo, uint64_t hi)
    {
        if (lo >= hi) return;
        It it = run.upper_bound(lo);
        if (it != run.begin()) {
            It before = std::prev(it);
            if (before->second >= lo) { lo = before->first; hi = std::max(hi, before->second); it = run.erase(before); }
        }
        while (it != run.end() && it->first <= hi) { hi = std::max(hi, it->second); it = run.erase(it); }
        run.emplace_hint(it, lo, hi);
    }
    void release(uint64_t lo, uint64_t hi)
    {
        if (lo >= hi) return;
        It it = run.upper_bound(lo);
        if (it != run.begin()) {
            It before = std::prev(it);
            const uint64_t end = before->second;
            if (end > lo) {
                if (before->first == lo) run.erase(before); else before->second = lo;
                if (end > hi) { run.emplace(hi, end); return; }
            }
        }
        while (it != run.end() && it->first < hi) {
            const uint64_t end = it->second;
            it = run.erase(it);
            if (end > hi) { run.emplace_hint(it, hi, end); return; }
        }
    }
};

void openOrThrow(const std::string& fileName, std::ofstream& stream)                     // TryOpenFile, blocksfinder.cpp:176-183
{
    stream.open(fileName.c_str());
    if (!stream) throw LcbError("Cannot open file " + fileName);
}

void writeGff(const lcb_graph& g, const std::vector<lcb_block>& blockList, const std::string& fileName)
{
    std::ofstream out;
    openOrThrow(fileName, out);
    std::vector<lcb_block> block(blockList);
    std::sort(block.begin(), block.end(), compareById);                                  // blocksfinder.cpp:146
    std::string text = "##gff-version 3.1.26\n";
    for (uint32_t i = 0; i < g.nChr(); i++)
        text += "##sequence-region " + g.chrName[i] + " 1 " + std::to_string(g.seq[i].size()) + "\n";
    for (const auto& b : block) {
        text += g.chrName[b.chr];
        text += "\tSibeliaZ\tSO:0000856\t";
        text += std::to_string(b.start + 1);
        text += '\t';
        text += std::to_string(b.end);
        text += "\t.\t";
        text += b.id > 0 ? '+' : '-';
        text += "\t.\tID=";
        text += std::to_string(blockId(b));
        text += '\n';
        if (text.size() > (1u << 20)) { out << text; text.clear(); }
    }
    out << text;
    out.flush();
    if (!out) throw LcbError("Cannot write file " + fileName);
}

void writeSequences(const lcb_graph& g, const std::vector<lcb_block>& block, const std::string& prefix, size_t chunks)
{
    if (chunks == 0) throw LcbError("--chunks must be positive unless --noseq is given");   // the reference divides by zero (blocksfinder.h:580)
    std::vector<lcb_block> blockList(block);
    std::vector<std::ofstream> chunkOut(chunks);
    for (size_t i = 0; i < chunks; i++) openOrThrow(prefix + std::to_string(i) + ".tmp", chunkOut[i]);
    size_t nowChunk = 0;
    std::sort(blockList.begin(), blockList.end(), compareById);                          // GroupBy, blocksfinder.h:100-110,546
    std::string line;
    for (size_t now = 0; now < blockList.size();) {
        const size_t prev = now;
        for (; now < blockList.size() && !compareById(blockList[prev], blockList[now]); now++)
            ;
        line.clear();
        for (size_t b = prev; b < now; b++) {
            const lcb_block& bl = blockList[b];
            const size_t length = bl.end - bl.start;
            const std::string& seq = g.seq[bl.chr];
            const size_t chrSize = seq.size();
            line += "> " + g.chrName[bl.chr] + ";";
            if (bl.id > 0) {
                line += std::to_string(bl.start) + ";" + std::to_string(length) + ";+;" + std::to_string(chrSize) + "@";
                line.append(seq, bl.start, length);
            } else {
                line += std::to_string(chrSize - bl.end) + ";" + std::to_string(length) + ";-;" + std::to_string(chrSize) + "@";
                for (size_t i = 0; i < length; i++) line += reverseChar(seq[bl.end - 1 - i]);
            }
            line += '@';
        }
        line += '\n';
        chunkOut[nowChunk] << line;
        nowChunk = (nowChunk + 1) % chunks;
    }
    for (auto& o : chunkOut) { o.flush(); if (!o) throw LcbError("Cannot write block sequence chunk"); }
}

}  // namespace

void lcb_generate_output_impl(const lcb_graph& g, int64_t minBlock, const lcb_block* blocks, int64_t nBlocks, int64_t blocksFound,
                              const std::string& outDir, bool genSeq, int64_t chunks, int64_t* nTrimmed, double* coverage)
{
    // Overlap trimming (blocksfinder.h:605-656): blocks are visited by falling number of copies (ties: rising id, the GroupBy order of
    // blocksfinder.h:623), and an instance gives up the bases at its two ends that a block visited before it already holds. What the
    // reference keeps as one flag per base (plus one past the end) is kept here as runs of held bases per chromosome - memory and time
    // follow the number of block instances, not the genome size - with the flag array's exact outcomes:
    //  * left end: the first base at or after the start that nobody holds, at most the end;
    //  * right end: the reference tests the flag AT the end position, not before it: an instance whose following base is free keeps its
    //    right end even where its last bases are held, otherwise the end moves down to the nearest free position;
    //  * a block left with a single instance returns the whole range of that instance, bases held before it included.
    std::vector<HeldRuns> held(g.nChr());
    int64_t nextId = 1;
    std::vector<lcb_block> inst(blocks, blocks + nBlocks), kept, trimmed;
    std::vector<int> copies((size_t)blocksFound + 1, 0);
    for (const auto& b : inst) copies[blockId(b)]++;
    SortByMultiplicity pred{copies};
    std::sort(inst.begin(), inst.end(), pred);                                           // the permutation of equal keys is part of the result
    for (size_t from = 0, to; from < inst.size(); from = to) {
        for (to = from + 1; to < inst.size() && !pred(inst[from], inst[to]); to++) {}
        kept.clear();
        for (size_t i = from; i < to; i++) {
            HeldRuns& runs = held[inst[i].chr];
            const uint64_t lo = std::min<uint64_t>(inst[i].end, runs.freeAtOrAfter(inst[i].start));
            uint64_t hi;
            if (!runs.freeAtOrBefore(inst[i].end, hi) || hi < lo) hi = lo;
            if ((int64_t)(hi - lo) < minBlock) continue;
            lcb_block t;
            t.id = (int32_t)(inst[i].id > 0 ? nextId : -nextId); t.chr = inst[i].chr; t.start = lo; t.end = hi;
            kept.push_back(t);
            runs.hold(lo, hi);
        }
        if (kept.size() > 1) {
            nextId++;
            trimmed.insert(trimmed.end(), kept.begin(), kept.end());
        } else {
            for (const auto& t : kept) held[t.chr].release(t.start, t.end);
        }
    }
    uint64_t total = 0, totalBlock = 0;                                                  // CalculateCoverage, blocksfinder.cpp:109-124
    for (uint32_t i = 0; i < g.nChr(); i++) total += g.seq[i].size();
    for (const auto& b : trimmed) totalBlock += b.end - b.start;
    if (coverage) *coverage = (double)totalBlock / (double)total;
    if (nTrimmed) *nTrimmed = nextId - 1;
    std::sort(trimmed.begin(), trimmed.end(), blockLess);                                // blocksfinder.h:662
    if (mkdir(outDir.c_str(), 0755) != 0 && errno != EEXIST) throw LcbError("Cannot create dir " + outDir);   // blocksfinder.cpp:15-27
    writeGff(g, trimmed, outDir + "/" + "blocks_coords.gff");
    if (genSeq) writeSequences(g, trimmed, outDir + "/", (size_t)chunks);
}
