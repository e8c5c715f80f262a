// ref_dump: intermediate-golden dumper, compiled ONLY in the build container against the
// unmodified reference headers under /root/reference (never shipped, output goes to oracle/_ref/).
//
// It is a driver, not a copy: it includes the reference's own blocksfinder.h (with `private`
// opened up so the driver can read BlocksFinder::bundle_ / blocksInstance_ and call
// ProcessVertex::Process directly) and prints
//   bundles.tsv      sorted seed list after BlocksFinder::FindBlocks (blocksfinder.h:517)
//   pretrim.tsv      blocksInstance_ in commit order, before GenerateOutput re-sorts it (blocksfinder.h:623)
//   seeds_final.tsv  per-seed Process() result against the FINAL `used` state of storage 1
//   seeds_init.tsv   per-seed Process() result against an all-unused storage 2
//   summary.txt      blocksFound_, failure_, counts
// The per-seed dumps exercise the hot path (blocksfinder.h:228-310) as a pure function of
// (tables, used bits, seed), which is exactly what the HIP kernel implements.
#include <omp.h>
#include <set>
#include <map>
#include <list>
#include <ctime>
#include <queue>
#include <atomic>
#include <string>
#include <vector>
#include <memory>
#include <cstdint>
#include <climits>
#include <cassert>
#include <numeric>
#include <sstream>
#include <fstream>
#include <iostream>
#include <iterator>
#include <stdexcept>
#include <algorithm>
#include <functional>
#include <unordered_map>
#include <unordered_set>

#define private public
#include "blocksfinder.h"
#undef private

using namespace Sibelia;

static void dumpSeeds(BlocksFinder& finder, const std::string& file) {
    std::ofstream out(file.c_str());
    BlocksFinder::ProcessVertex pv(finder);
    int64_t bestScore;
    std::vector<int64_t> logPath;
    std::vector<size_t> data;
    std::vector<uint32_t> count(finder.storage_.GetVerticesNumber() * 2 + 1, 0);
    Path currentPath(finder.storage_, finder.maxBranchSize_, finder.minBlockSize_, finder.minBlockSize_, finder.maxFlankingSize_, true);
    BlocksFinder::InstanceVector inst;
    for (size_t i = 0; i < finder.bundle_.size(); i++) {
        pv.Process(finder.bundle_[i], currentPath, data, count, inst, logPath, bestScore);
        if (inst.empty()) continue;
        out << i << '\t' << bestScore << '\t' << inst.size();
        for (auto& it : inst) {
            out << '\t' << (it.Front().IsPositiveStrand() ? '+' : '-') << ',' << it.Front().GetChrId() << ','
                << it.Front().GetIndex() << ',' << it.Back().GetIndex();
        }
        out << '\n';
    }
}

int main(int argc, char** argv) {
    if (argc < 8) {
        std::cerr << "usage: ref_dump <graph.bin> <k> <b> <m> <a> <outdir> <fasta...>" << std::endl;
        return 2;
    }
    std::string graph = argv[1];
    int k = atoi(argv[2]), b = atoi(argv[3]), m = atoi(argv[4]), a = atoi(argv[5]);
    std::string outDir = argv[6];
    std::vector<std::string> fasta(argv + 7, argv + argc);
    try {
        CreateOutDirectory(outDir);
        JunctionStorage storage(graph, fasta, k, 1, a, 0);
        BlocksFinder finder(storage, k);
        std::streambuf* old = std::cout.rdbuf(nullptr);   // silence the progress bar
        finder.FindBlocks(m, b, b, 8, 0, 1, outDir + "/paths.txt");
        std::cout.rdbuf(old);
        {
            std::ofstream out((outDir + "/bundles.tsv").c_str());
            for (auto& bd : finder.bundle_)
                out << bd.vid << '\t' << int(bd.ch) << '\t' << bd.count << '\t' << bd.rank << '\t'
                    << bd.resolve.first << '\t' << bd.resolve.second << '\n';
        }
        {
            std::ofstream out((outDir + "/pretrim.tsv").c_str());
            for (auto& bi : finder.blocksInstance_)
                out << bi.GetSignedBlockId() << '\t' << bi.GetChrId() << '\t' << bi.GetStart() << '\t' << bi.GetEnd() << '\n';
        }
        {
            std::ofstream out((outDir + "/summary.txt").c_str());
            out << "blocksFound\t" << finder.blocksFound_ << "\nfailure\t" << finder.failure_
                << "\nbundles\t" << finder.bundle_.size() << "\nvertices\t" << storage.GetVerticesNumber()
                << "\nchromosomes\t" << storage.GetChrNumber() << '\n';
            size_t p = 0;
            for (int64_t c = 0; c < storage.GetChrNumber(); c++) p += storage.GetChrVerticesCount(c);
            out << "positions\t" << p << '\n';
        }
        dumpSeeds(finder, outDir + "/seeds_final.tsv");

        JunctionStorage fresh(graph, fasta, k, 1, a, 0);   // re-points JunctionStorage::this_ at an unused table
        BlocksFinder finder2(fresh, k);
        finder2.bundle_ = finder.bundle_;
        finder2.failure_ = 0;
        finder2.lookingDepth_ = 8;
        finder2.minBlockSize_ = m;
        finder2.maxBranchSize_ = b;
        finder2.maxFlankingSize_ = b;
        dumpSeeds(finder2, outDir + "/seeds_init.tsv");
    } catch (std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
