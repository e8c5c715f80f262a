// lcb-synth: deterministic synthetic multi-strain genome generator.
//
// The reference checkout ships no genomes (examples/genome{1,2}.fa are missing) and there is no
// network, so every BASELINE.json config is restated as a synthetic pangenome:
// an ancestor made of segments; every strain keeps a segment with probability `keep`, swaps a
// fraction of adjacent pairs, inverts a fraction of segments, gets per-base substitutions and
// indels, strain-private filler sequence, dispersed repeat families and (optionally) short tandem
// duplications. Output is one FASTA, one record per strain chromosome.
//
// Everything is driven by a splitmix64/xoshiro256** stream seeded from --seed, so a given command
// line yields byte-identical output on every platform.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

namespace {

struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t& x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) { for (auto& v : s) v = splitmix(seed); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return n ? next() % n : 0; }
    uint64_t range(uint64_t lo, uint64_t hi) { return lo + below(hi - lo + 1); }  // inclusive
    bool chance(double p) { return uniform() < p; }
    char base() { return "ACGT"[next() & 3]; }
};

std::string randomSeq(Rng& rng, size_t n) {
    std::string s(n, 'A');
    for (auto& c : s) c = rng.base();
    return s;
}

char comp(char c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; }
    return 'N';
}

std::string revComp(const std::string& s) {
    std::string r(s.rbegin(), s.rend());
    for (auto& c : r) c = comp(c);
    return r;
}

struct Opt {
    int strains = 4, segments = 12, chromosomes = 1, repeatFamilies = 2, repeatCopies = 6;
    long segMin = 500, segMax = 8000, fillerMin = 200, fillerMax = 3000, repeatLen = 700;
    double keep = 0.8, swapFrac = 0.05, invert = 0.10, sub = 0.02, indel = 0.002, fillerFrac = 0.25, tandem = 0.0, nrun = 0.0;
    uint64_t seed = 1;
    std::string out;
};

double argD(const char* v) { return atof(v); }

void usage() {
    fprintf(stderr,
        "usage: lcb-synth -o out.fa [--strains N] [--segments M] [--seg-min L] [--seg-max L] [--keep P]\n"
        "       [--swap P] [--invert P] [--sub P] [--indel P] [--filler-frac P] [--filler-min L] [--filler-max L]\n"
        "       [--repeat-families F] [--repeat-copies C] [--repeat-len L] [--tandem P] [--nrun P]\n"
        "       [--chromosomes C] [--seed S]\n");
}

}  // namespace

int main(int argc, char** argv) {
    Opt o;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](void) -> const char* { if (i + 1 >= argc) { usage(); exit(2); } return argv[++i]; };
        if (a == "-o") o.out = need();
        else if (a == "--strains") o.strains = atoi(need());
        else if (a == "--segments") o.segments = atoi(need());
        else if (a == "--chromosomes") o.chromosomes = atoi(need());
        else if (a == "--seg-min") o.segMin = atol(need());
        else if (a == "--seg-max") o.segMax = atol(need());
        else if (a == "--filler-min") o.fillerMin = atol(need());
        else if (a == "--filler-max") o.fillerMax = atol(need());
        else if (a == "--repeat-families") o.repeatFamilies = atoi(need());
        else if (a == "--repeat-copies") o.repeatCopies = atoi(need());
        else if (a == "--repeat-len") o.repeatLen = atol(need());
        else if (a == "--keep") o.keep = argD(need());
        else if (a == "--swap") o.swapFrac = argD(need());
        else if (a == "--invert") o.invert = argD(need());
        else if (a == "--sub") o.sub = argD(need());
        else if (a == "--indel") o.indel = argD(need());
        else if (a == "--filler-frac") o.fillerFrac = argD(need());
        else if (a == "--tandem") o.tandem = argD(need());
        else if (a == "--nrun") o.nrun = argD(need());
        else if (a == "--seed") o.seed = strtoull(need(), nullptr, 10);
        else { usage(); return 2; }
    }
    if (o.out.empty() || o.strains < 1 || o.segments < 1 || o.chromosomes < 1) { usage(); return 2; }

    Rng rng(o.seed);
    // Ancestor segments.
    std::vector<std::string> anc(o.segments);
    for (auto& s : anc) s = randomSeq(rng, rng.range(o.segMin, o.segMax));
    // Dispersed repeat families pasted into random ancestor segments.
    for (int f = 0; f < o.repeatFamilies; f++) {
        std::string rep = randomSeq(rng, o.repeatLen);
        for (int c = 0; c < o.repeatCopies; c++) {
            std::string copy = rep;
            for (auto& ch : copy) if (rng.chance(0.01)) ch = rng.base();
            if (rng.chance(0.3)) copy = revComp(copy);
            std::string& seg = anc[rng.below(anc.size())];
            seg.insert(rng.below(seg.size() + 1), copy);
        }
    }

    FILE* out = fopen(o.out.c_str(), "w");
    if (!out) { fprintf(stderr, "lcb-synth: cannot open %s\n", o.out.c_str()); return 1; }

    for (int st = 0; st < o.strains; st++) {
        // Segment order for this strain: presence/absence, adjacent swaps, inversions.
        std::vector<std::pair<int, bool>> order;
        for (int s = 0; s < o.segments; s++) if (rng.chance(o.keep)) order.push_back({s, false});
        if (order.empty()) order.push_back({0, false});
        for (size_t i = 0; i + 1 < order.size(); i++) if (rng.chance(o.swapFrac)) { std::swap(order[i], order[i + 1]); i++; }
        for (auto& e : order) if (rng.chance(o.invert)) e.second = true;

        // Chromosome boundaries fall between segments.
        std::vector<std::string> chrSeq(o.chromosomes);
        for (size_t i = 0; i < order.size(); i++) {
            std::string& dst = chrSeq[std::min<size_t>(o.chromosomes - 1, i * o.chromosomes / order.size())];
            const std::string src = order[i].second ? revComp(anc[order[i].first]) : anc[order[i].first];
            const size_t segStart = dst.size();
            for (size_t p = 0; p < src.size(); p++) {
                if (rng.chance(o.indel * 0.5)) continue;                    // deletion
                if (rng.chance(o.indel * 0.5)) dst.push_back(rng.base());   // insertion
                char c = src[p];
                if (rng.chance(o.sub)) { char n; do { n = rng.base(); } while (n == c); c = n; }
                dst.push_back(c);
            }
            if (o.tandem > 0 && rng.chance(o.tandem) && dst.size() - segStart > 600) {
                // short tandem duplication inside the segment just emitted
                size_t len = rng.range(60, 400), at = segStart + rng.below(dst.size() - segStart - len);
                dst.insert(at + len, dst.substr(at, len));
            }
            if (o.nrun > 0 && rng.chance(o.nrun)) dst.append(rng.range(1, 40), 'N');
            if (rng.chance(o.fillerFrac)) dst += randomSeq(rng, rng.range(o.fillerMin, o.fillerMax));
        }
        for (int c = 0; c < o.chromosomes; c++) {
            if (chrSeq[c].size() < 64) chrSeq[c] += randomSeq(rng, 64);   // never emit a record shorter than any k
            if (o.chromosomes == 1) fprintf(out, ">s%d\n", st);
            else fprintf(out, ">s%d.c%d\n", st, c);
            for (size_t p = 0; p < chrSeq[c].size(); p += 80) {
                fwrite(chrSeq[c].data() + p, 1, std::min<size_t>(80, chrSeq[c].size() - p), out);
                fputc('\n', out);
            }
        }
    }
    fclose(out);
    return 0;
}
