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
    std::vector<std::vector<bool>> covered(g.nChr());                                    // blocksfinder.h:607-611
    for (size_t i = 0; i < covered.size(); i++) covered[i].assign(g.seq[i].size() + 1, false);
    int64_t trimmedId = 1;
    std::vector<lcb_block> inst(blocks, blocks + nBlocks), buffer, trimmed;
    std::vector<int> copies((size_t)blocksFound + 1, 0);
    for (const auto& b : inst) copies[blockId(b)]++;
    SortByMultiplicity pred{copies};
    std::sort(inst.begin(), inst.end(), pred);                                           // GroupBy, blocksfinder.h:623
    for (size_t now = 0; now < inst.size();) {
        const size_t prev = now;
        for (; now < inst.size() && !pred(inst[prev], inst[now]); now++)
            ;
        buffer.clear();
        for (size_t i = prev; i < now; i++) {                                            // blocksfinder.h:627-639
            const size_t chr = inst[i].chr;
            size_t start = inst[i].start, end = inst[i].end;
            for (; covered[chr][start] && start < end; start++)
                ;
            for (; covered[chr][end] && end > start; end--)
                ;
            if ((int64_t)(end - start) >= minBlock) {
                lcb_block t;
                t.id = (int32_t)((inst[i].id > 0 ? 1 : -1) * trimmedId); t.chr = (uint32_t)chr; t.start = start; t.end = end;
                buffer.push_back(t);
                std::fill(covered[chr].begin() + start, covered[chr].begin() + end, true);
            }
        }
        if (buffer.size() > 1) {
            trimmedId++;
            trimmed.insert(trimmed.end(), buffer.begin(), buffer.end());
        } else {
            for (const auto& it : buffer) std::fill(covered[it.chr].begin() + it.start, covered[it.chr].begin() + it.end, false);
        }
    }
    uint64_t total = 0, totalBlock = 0;                                                  // CalculateCoverage, blocksfinder.cpp:109-124
    for (uint32_t i = 0; i < g.nChr(); i++) total += g.seq[i].size();
    for (const auto& b : trimmed) totalBlock += b.end - b.start;
    if (coverage) *coverage = (double)totalBlock / (double)total;
    if (nTrimmed) *nTrimmed = trimmedId - 1;
    std::sort(trimmed.begin(), trimmed.end(), blockLess);                                // blocksfinder.h:662
    if (mkdir(outDir.c_str(), 0755) != 0 && errno != EEXIST) throw LcbError("Cannot create dir " + outDir);   // blocksfinder.cpp:15-27
    writeGff(g, trimmed, outDir + "/" + "blocks_coords.gff");
    if (genSeq) writeSequences(g, trimmed, outDir + "/", (size_t)chunks);
}
