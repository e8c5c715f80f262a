// lcb-mkgraph: exact junction finder that writes the TwoPaCo junction-list format.
//
// `twopaco` (the stage upstream of sibeliaz-lcb, reference SibeliaZ-LCB/sibeliaz:145) is an empty
// submodule in the reference checkout, so graphs are produced here. The output format is the one
// the reference reads in common/junctionapi.h:80-98 and writes in :106-136: packed little-endian
// records {uint32 pos, int64 id}; a record with pos == 0xFFFFFFFF and id == INT64_MAX separates
// chromosomes.
//
// Junction definition (compacted de Bruijn graph on canonical k-mers, k odd): a k-mer occurrence
// is a junction iff its canonical k-mer has >= 2 distinct successor characters or >= 2 distinct
// predecessor characters over both strands of the whole input, or the occurrence class touches a
// sequence end / a non-ACGT character. id = 1-based rank of the canonical k-mer in order of first
// appearance among junctions; the sign is + iff the occurrence spells the canonical form.
// The k-mer table is filled by all cores (OpenMP, lock-free open addressing); the output does not depend on the thread count.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <string>
#include <vector>
#include <stdexcept>
#include <algorithm>
#include <atomic>
#include <memory>
#include <chrono>

namespace {

double nowS() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
const bool kVerbose = getenv("LCB_MKGRAPH_VERBOSE") != nullptr;
#define PHASE(name) do { if (kVerbose) { const double t_ = nowS(); fprintf(stderr, "lcb-mkgraph: %-28s %.2f s\n", name, t_ - tPhase); tPhase = t_; } } while (0)


struct Record { std::string name; std::string seq; };

void readFasta(const std::string& file, std::vector<Record>& out) {
    FILE* f = fopen(file.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + file);
    std::vector<char> buf(1 << 20);
    bool inHeader = false;
    std::string header;
    size_t n;
    while ((n = fread(buf.data(), 1, buf.size(), f)) > 0) {
        for (size_t i = 0; i < n; i++) {
            char c = buf[i];
            if (inHeader) {
                if (c == '\n') {
                    inHeader = false;
                    size_t e = 0;
                    while (e < header.size() && !isspace((unsigned char)header[e])) e++;
                    out.push_back({header.substr(0, e), std::string()});
                } else header.push_back(c);
            } else if (c == '>') { inHeader = true; header.clear(); }
            else if (!isspace((unsigned char)c) && !out.empty()) out.back().seq.push_back((char)toupper((unsigned char)c));
        }
    }
    fclose(f);
}

inline int code(char c) {
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; }
    return -1;
}

// Open-addressing map canonical k-mer -> {succ mask(4) | pred mask(4) << 4 | forced << 8}, junction id. Fixed capacity,
// filled concurrently: a slot is claimed with a compare-and-swap on its key, the masks are OR-ed in atomically.
struct KmerTable {
    std::vector<std::atomic<uint64_t>> key;   // code + 1, 0 = empty
    std::vector<std::atomic<uint32_t>> val;   // the masks; after the junctions are known the same word holds the k-mer's id (bit 31 set)
    std::atomic<size_t> used{0};
    size_t mask = 0;
    explicit KmerTable(size_t cap) : key(cap), val(cap), mask(cap - 1)
    {
        #pragma omp parallel for schedule(static)
        for (size_t i = 0; i < cap; i++) { key[i].store(0, std::memory_order_relaxed); val[i].store(0, std::memory_order_relaxed); }
    }
    static uint64_t mix(uint64_t x) {
        x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
        return x;
    }
    size_t find(uint64_t kmer) const {
        for (size_t h = mix(kmer) & mask;; h = (h + 1) & mask) {
            const uint64_t k = key[h].load(std::memory_order_relaxed);
            if (k == kmer + 1) return h;
            if (k == 0) return SIZE_MAX;
        }
    }
    // false: the table is too full (the caller starts over with a larger one). `full` is shared by all threads: once it is
    // set nobody inserts any more, and a probe sequence never runs past one sweep of the table, so a table that filled up under
    // the feet of the in-flight chunks cannot spin.
    std::atomic<bool> full{false};
    bool add(uint64_t kmer, uint32_t bits) {
        if (full.load(std::memory_order_relaxed)) return false;
        size_t h = mix(kmer) & mask;
        for (size_t probes = 0; probes <= mask; probes++, h = (h + 1) & mask) {
            uint64_t k = key[h].load(std::memory_order_relaxed);
            if (k == 0) {
                if (key[h].compare_exchange_strong(k, kmer + 1, std::memory_order_relaxed)) {
                    // (the fill level is counted in batches per thread: one shared counter bumped for every new k-mer is a cache line that
                    // hundreds of threads fight over - the GPU box's 256 threads generated a 4-Gbp graph SLOWER than 8 threads here)
                    static thread_local uint32_t mine = 0;
                    if (++mine == 1024) {
                        mine = 0;
                        if ((used.fetch_add(1024, std::memory_order_relaxed) + 1024 + 1024 * 512) * 10 > (mask + 1) * 9) { full.store(true, std::memory_order_relaxed); return false; }   // (load factor 0.9: a table of 2^31 slots is 24 GB)
                    }
                    k = kmer + 1;
                }
            }
            if (k == kmer + 1) {
                if ((val[h].load(std::memory_order_relaxed) & bits) != bits) val[h].fetch_or(bits, std::memory_order_relaxed);   // (most occurrences of a k-mer repeat what is known: a read, not a read-modify-write)
                return true;
            }
            if ((probes & 1023) == 1023 && full.load(std::memory_order_relaxed)) return false;
        }
        full.store(true, std::memory_order_relaxed);
        return false;
    }
};

inline bool isJunction(uint32_t v) {
    return (v & 0x100) || __builtin_popcount(v & 0xF) >= 2 || __builtin_popcount((v >> 4) & 0xF) >= 2;
}

// Calls fn(pos, fwdCode, rcCode) for every k-mer window of seq made of ACGT only whose start lies in [from, to).
template <class F>
void forEachKmer(const std::string& seq, int k, size_t from, size_t to, F fn) {
    const uint64_t m = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    uint64_t fwd = 0, rc = 0;
    int valid = 0;
    const size_t stop = std::min(seq.size(), to + (size_t)k - 1);
    for (size_t i = from; i < stop; i++) {
        int c = code(seq[i]);
        if (c < 0) { valid = 0; fwd = rc = 0; continue; }
        fwd = ((fwd << 2) | (uint64_t)c) & m;
        rc = (rc >> 2) | ((uint64_t)(3 - c) << (2 * (k - 1)));
        if (++valid >= k) fn(i + 1 - k, fwd, rc);
    }
}

}  // namespace

int main(int argc, char** argv) {
    int k = 25;
    std::string out;
    std::vector<std::string> fasta;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "-k" && i + 1 < argc) k = atoi(argv[++i]);
        else if (a == "-o" && i + 1 < argc) out = argv[++i];
        else fasta.push_back(a);
    }
    if (out.empty() || fasta.empty() || k < 3 || k > 31 || k % 2 == 0) {
        fprintf(stderr, "usage: lcb-mkgraph -k <odd 3..31> -o junctions.bin <fasta...>\n");
        return 2;
    }
    try {
        double tPhase = nowS();
        std::vector<Record> rec;
        for (auto& f : fasta) readFasta(f, rec);
        PHASE("read FASTA");

        // work units: chunks of sequence starts, so that all cores share one long chromosome
        struct Chunk { uint32_t rec; size_t from, to; };
        std::vector<Chunk> chunks;
        size_t nWindows = 0;
        for (size_t r = 0; r < rec.size(); r++) {
            const size_t n = rec[r].seq.size() >= (size_t)k ? rec[r].seq.size() - k + 1 : 0;
            nWindows += n;
            for (size_t a = 0; a < n; a += (1u << 20)) chunks.push_back(Chunk{(uint32_t)r, a, std::min(n, a + (1u << 20))});
        }
        size_t cap = 1 << 20;
        while (cap < nWindows / 2) cap <<= 1;
        std::unique_ptr<KmerTable> tp;
        for (;; cap <<= 1) {
            tp.reset(new KmerTable(cap));
            KmerTable& table = *tp;
            #pragma omp parallel for schedule(dynamic, 1)
            for (size_t c = 0; c < chunks.size(); c++) {
                if (table.full.load(std::memory_order_relaxed)) continue;
                const std::string& s = rec[chunks[c].rec].seq;
                forEachKmer(s, k, chunks[c].from, chunks[c].to, [&](size_t p, uint64_t fwd, uint64_t rc) {
                    const int nx = p + k < s.size() ? code(s[p + k]) : -1;
                    const int pv = p > 0 ? code(s[p - 1]) : -1;
                    uint32_t bits = (nx < 0 || pv < 0) ? 0x100 : 0;
                    if (fwd < rc) {
                        if (nx >= 0) bits |= 1u << nx;
                        if (pv >= 0) bits |= 1u << (4 + pv);
                        table.add(fwd, bits);         // (a refused insert sets table.full: this pass is abandoned)
                    } else {
                        if (pv >= 0) bits |= 1u << (3 - pv);
                        if (nx >= 0) bits |= 1u << (4 + 3 - nx);
                        table.add(rc, bits);
                    }
                });
            }
            if (!table.full.load()) break;
        }
        KmerTable& table = *tp;
        PHASE("k-mer table");

        // junction occurrences of every chunk (parallel), then ids in order of first appearance and the records (in order)
        struct Occ { uint32_t pos; uint32_t slotLo; uint8_t slotHi; uint8_t fwd; };
        std::vector<std::vector<Occ>> found(chunks.size());
        #pragma omp parallel for schedule(dynamic, 1)
        for (size_t c = 0; c < chunks.size(); c++) {
            forEachKmer(rec[chunks[c].rec].seq, k, chunks[c].from, chunks[c].to, [&](size_t p, uint64_t fwd, uint64_t rc) {
                const bool isFwd = fwd < rc;
                const size_t h = table.find(isFwd ? fwd : rc);
                if (!isJunction(table.val[h].load(std::memory_order_relaxed))) return;
                found[c].push_back(Occ{(uint32_t)p, (uint32_t)h, (uint8_t)(h >> 32), (uint8_t)isFwd});
            });
        }
        PHASE("junction occurrences");
        uint32_t nextId = 1;
        uint64_t written = 0;
        FILE* f = fopen(out.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot create " + out);
        std::vector<unsigned char> buf;
        auto flushBuf = [&]() { if (!buf.empty() && fwrite(buf.data(), 1, buf.size(), f) != buf.size()) throw std::runtime_error("cannot write " + out); buf.clear(); };
        auto put = [&](uint32_t pos, int64_t id) {
            const size_t at = buf.size();
            buf.resize(at + 12);
            memcpy(&buf[at], &pos, 4); memcpy(&buf[at + 4], &id, 8);
            if (buf.size() >= (64u << 20)) flushBuf();
        };
        size_t c = 0;
        for (size_t r = 0; r < rec.size(); r++) {
            for (; c < chunks.size() && chunks[c].rec == r; c++) {
                for (const Occ& o : found[c]) {
                    const size_t h = ((size_t)o.slotHi << 32) | o.slotLo;
                    uint32_t v = table.val[h].load(std::memory_order_relaxed);
                    if (!(v & 0x80000000u)) { v = 0x80000000u | nextId++; table.val[h].store(v, std::memory_order_relaxed); }     // (the masks are no longer needed)
                    const int64_t id = (int64_t)(v & 0x7FFFFFFFu);
                    put(o.pos, o.fwd ? id : -id);
                    written++;
                }
                std::vector<Occ>().swap(found[c]);
            }
            put(0xFFFFFFFFu, INT64_MAX);
        }
        flushBuf();
        fclose(f);
        PHASE("ids + records");
        fprintf(stderr, "lcb-mkgraph: %zu records, %llu junction occurrences, %u junction k-mers\n",
                rec.size(), (unsigned long long)written, nextId - 1);
    } catch (std::exception& e) {
        fprintf(stderr, "lcb-mkgraph: error: %s\n", e.what());
        return 1;
    }
    return 0;
}
