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
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <string>
#include <vector>
#include <stdexcept>

namespace {

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

// Open-addressing map canonical k-mer -> {succ mask(4) | pred mask(4) << 4 | forced << 8}, junction id.
struct KmerTable {
    std::vector<uint64_t> key;   // code + 1, 0 = empty
    std::vector<uint16_t> val;
    std::vector<uint32_t> id;
    size_t used = 0, mask = 0;
    explicit KmerTable(size_t cap) { resize(cap); }
    static uint64_t mix(uint64_t x) {
        x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
        return x;
    }
    void resize(size_t cap) {
        std::vector<uint64_t> ok; ok.swap(key);
        std::vector<uint16_t> ov; ov.swap(val);
        key.assign(cap, 0); val.assign(cap, 0); mask = cap - 1; used = 0;
        for (size_t i = 0; i < ok.size(); i++) if (ok[i]) val[slot(ok[i] - 1, true)] = ov[i];
    }
    size_t slot(uint64_t kmer, bool insert) {
        size_t h = mix(kmer) & mask;
        for (;; h = (h + 1) & mask) {
            if (key[h] == kmer + 1) return h;
            if (key[h] == 0) {
                if (!insert) return SIZE_MAX;
                key[h] = kmer + 1; used++;
                return h;
            }
        }
    }
    void add(uint64_t kmer, uint16_t bits) {
        if ((used + 1) * 10 > (mask + 1) * 6) resize((mask + 1) * 2);
        val[slot(kmer, true)] |= bits;
    }
};

inline bool isJunction(uint16_t v) {
    return (v & 0x100) || __builtin_popcount(v & 0xF) >= 2 || __builtin_popcount((v >> 4) & 0xF) >= 2;
}

// Calls fn(pos, fwdCode, rcCode) for every k-mer window of seq made of ACGT only.
template <class F>
void forEachKmer(const std::string& seq, int k, F fn) {
    const uint64_t m = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    uint64_t fwd = 0, rc = 0;
    int valid = 0;
    for (size_t i = 0; i < seq.size(); i++) {
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
        std::vector<Record> rec;
        for (auto& f : fasta) readFasta(f, rec);

        KmerTable table(1 << 20);
        for (auto& r : rec) {
            const std::string& s = r.seq;
            forEachKmer(s, k, [&](size_t p, uint64_t fwd, uint64_t rc) {
                const int nx = p + k < s.size() ? code(s[p + k]) : -1;
                const int pv = p > 0 ? code(s[p - 1]) : -1;
                uint16_t bits = (nx < 0 || pv < 0) ? 0x100 : 0;
                if (fwd < rc) {
                    if (nx >= 0) bits |= 1u << nx;
                    if (pv >= 0) bits |= 1u << (4 + pv);
                    table.add(fwd, bits);
                } else {
                    if (pv >= 0) bits |= 1u << (3 - pv);
                    if (nx >= 0) bits |= 1u << (4 + 3 - nx);
                    table.add(rc, bits);
                }
            });
        }

        table.id.assign(table.key.size(), 0);
        uint32_t nextId = 1;
        uint64_t written = 0;
        FILE* f = fopen(out.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot create " + out);
        auto put = [&](uint32_t pos, int64_t id) {
            unsigned char b[12];
            memcpy(b, &pos, 4); memcpy(b + 4, &id, 8);
            if (fwrite(b, 1, 12, f) != 12) throw std::runtime_error("cannot write " + out);
        };
        for (auto& r : rec) {
            forEachKmer(r.seq, k, [&](size_t p, uint64_t fwd, uint64_t rc) {
                const bool isFwd = fwd < rc;
                const size_t h = table.slot(isFwd ? fwd : rc, false);
                if (!isJunction(table.val[h])) return;
                if (!table.id[h]) table.id[h] = nextId++;
                put((uint32_t)p, isFwd ? (int64_t)table.id[h] : -(int64_t)table.id[h]);
                written++;
            });
            put(0xFFFFFFFFu, INT64_MAX);
        }
        fclose(f);
        fprintf(stderr, "lcb-mkgraph: %zu records, %llu junction occurrences, %u junction k-mers\n",
                rec.size(), (unsigned long long)written, nextId - 1);
    } catch (std::exception& e) {
        fprintf(stderr, "lcb-mkgraph: error: %s\n", e.what());
        return 1;
    }
    return 0;
}
