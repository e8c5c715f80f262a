// lcb_host.h — host-side internals behind the C ABI in include/lcb.h.
#ifndef LCB_HOST_H
#define LCB_HOST_H

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "lcb.h"

// Structure-of-arrays replacement of Sibelia::JunctionStorage (junctionstorage.h:116-698).
// Every junction occurrence kept by the abundance filter has a FLAT index g = chrStart[chr] + idx.
struct lcb_graph {
    int k = 0;
    std::vector<uint64_t> chrStart;        // [C+1]
    std::vector<int32_t> posId;            // [P]  Position::id      (junctionstorage.h:142)
    std::vector<uint32_t> posPos;          // [P]  Position::pos     (junctionstorage.h:143)
    std::vector<uint8_t> posCh;            // [P]  Vertex::ch    = seq[pos+k]                     (junctionstorage.h:641)
    std::vector<uint8_t> posRevCh;         // [P]  Vertex::revCh = ReverseChar(seq[pos-1]) or 'N' (junctionstorage.h:642)
    uint32_t nVertex = 0;                  // vertex_.size() = max|id| + 1
    std::vector<uint32_t> occStart;        // [V+1] CSR over |id|, replaces vertex_ (junctionstorage.h:695)
    std::vector<uint32_t> occG;            // [P]  flat position of each occurrence, sorted by (chr, idx)
    std::vector<uint32_t> occChr;          // [P]
    std::vector<std::string> chrName;      // sequenceDescription_
    std::vector<std::string> seq;          // sequence_ (host only: chars above + block sequences for output)
    uint64_t nPos() const { return posId.size(); }
    uint32_t nChr() const { return (uint32_t)chrName.size(); }
};

struct LcbError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

void lcb_set_error(const std::string& msg);

// graph.cpp
lcb_graph* lcb_graph_load_impl(const char* junctionFile, const std::vector<std::string>& fasta, int k, int abundance, int threads);
// bundles.cpp
void lcb_enumerate_seeds_impl(const lcb_graph& g, int threads, std::vector<lcb_seed>& out);
// output.cpp
void lcb_generate_output_impl(const lcb_graph& g, int64_t minBlock, const lcb_block* blocks, int64_t nBlocks, int64_t blocksFound,
                              const std::string& outDir, bool genSeq, int64_t chunks, int64_t* nTrimmed, double* coverage);

// commit.cpp — ordered commit (thread-0 section of ProcessVertex::operator(), blocksfinder.h:372-427)
struct lcb_committer {
    const lcb_graph* g;
    lcb_params p;
    std::vector<uint32_t> used;            // bitmap over g (Position::used, junctionstorage.h:144)
    std::vector<lcb_block> blocks;         // blocksInstance_ (blocksfinder.h:921)
    std::vector<uint64_t> marks;           // (lo, hi) pairs not yet taken by the device
    std::vector<uint8_t> invalidChr;       // invalidChr_ (blocksfinder.h:923)
    std::vector<uint32_t> invalidList;
    int64_t blocksFound = 0, failures = 0;
    lcb_committer(const lcb_graph* graph, const lcb_params& prm);
    bool anyUsed(uint64_t lo, uint64_t hi) const;
    void finalize(const lcb_instance* inst, uint64_t n);
    void commitPhase(const lcb_seed* seeds, int64_t n, const uint64_t* offsets, const lcb_instance* inst,
                     lcb_reprocess_fn fn, void* user);
};

#endif
