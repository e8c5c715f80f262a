// main.cpp — `sibeliaz-lcb`: drop-in replacement of the reference executable (sibeliaz.cpp:37-157).
//
// Same flags, defaults, stdout banners, output files and exit codes, so line 146 of the reference's
// `sibeliaz` wrapper script keeps working verbatim:
//   sibeliaz-lcb --graph <junctions> <fasta...> -k <odd> -b <int> -o <dir> -m <int> -t <int> --abundance <int> [--noseq] --chunks <int>
// `-t` stays "host threads" (loading, seed enumeration); the GPU is chosen by the environment
// (LCB_DEVICE, default 0; HIP_VISIBLE_DEVICES also applies), never by a new flag.
// The block finder itself runs on the MI355X through the C ABI in include/lcb.h.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "lcb.h"

namespace {

struct Args {
    unsigned k = 25, b = 200, m = 200, t = 1, a = 150, chunks = 0;   // sibeliaz.cpp:45-110
    std::string graph, outDir;
    bool noSeq = false;
    std::vector<std::string> fasta;
};

[[noreturn]] void parseError(const std::string& arg, const std::string& msg)
{
    std::cerr << "PARSE ERROR: Argument: " << arg << "\n             " << msg << "\n\n"
              << "Brief USAGE: \n   sibeliaz-lcb  [--chunks <integer>] [--noseq] [-o <directory name>] --graph <file name> [-a <integer>] "
                 "[-t <integer>] [-m <integer>] [-b <integer>] [-k <oddc>] [--] [--version] [-h] <fasta files with genomes> ...\n\n"
              << "For complete USAGE and HELP type: \n   sibeliaz-lcb --help\n" << std::endl;
    exit(1);
}

unsigned toUnsigned(const std::string& flag, const char* v)
{
    char* end = nullptr;
    if (!v || !*v || *v == '-') parseError(flag, "Couldn't read argument value from string '" + std::string(v ? v : "") + "'");
    const unsigned long x = strtoul(v, &end, 10);
    if (*end) parseError(flag, "Couldn't read argument value from string '" + std::string(v) + "'");
    return (unsigned)x;
}

void usage()
{
    std::cout << "\nUSAGE: \n\n   sibeliaz-lcb  [--chunks <integer>] [--noseq] [-o <directory name>] --graph <file name>\n"
                 "                 [-a <integer>] [-t <integer>] [-m <integer>] [-b <integer>] [-k <oddc>] [--] [--version] [-h]\n"
                 "                 <fasta files with genomes> ...\n\n"
                 "Where: \n\n"
                 "   --chunks <integer>\n     Split blocks for alignment into a number of chunks\n\n"
                 "   --noseq\n     Do not output blocks sequences\n\n"
                 "   -o <directory name>,  --outdir <directory name>\n     Output dir for blocks sequences\n\n"
                 "   --graph <file name>\n     (required)  Binary file containing the graph\n\n"
                 "   -a <integer>,  --abundance <integer>\n     Max abundance of a junction\n\n"
                 "   -t <integer>,  --threads <integer>\n     Number of worker threads\n\n"
                 "   -m <integer>,  --blocksize <integer>\n     Minimum block size\n\n"
                 "   -b <integer>,  --branchsize <integer>\n     Maximum branch size\n\n"
                 "   -k <oddc>,  --kvalue <oddc>\n     Value of k\n\n"
                 "   --,  --ignore_rest\n     Ignores the rest of the labeled arguments following this flag.\n\n"
                 "   --version\n     Displays version information and exits.\n\n"
                 "   -h,  --help\n     Displays usage information and exits.\n\n"
                 "   <fasta files with genomes>  (accepted multiple times)\n     (required)  FASTA file(s) with nucleotide sequences.\n\n\n"
                 "   SibeliaZ-LCB, a program for construction of locally-collinear blocks from\n   complete genomes (MI355X build)\n" << std::endl;
}

Args parse(int argc, char** argv)
{
    Args a;
    bool rest = false, haveGraph = false;
    for (int i = 1; i < argc; i++) {
        std::string s = argv[i];
        if (rest || s.empty() || s[0] != '-' || s == "-") { a.fasta.push_back(s); continue; }
        if (s == "--" || s == "--ignore_rest") { rest = true; continue; }
        if (s == "-h" || s == "--help") { usage(); exit(0); }
        if (s == "--version") { std::cout << "\nsibeliaz-lcb  version: 1.2.7 (" << lcb_version() << ")\n" << std::endl; exit(0); }
        if (s == "--noseq") { a.noSeq = true; continue; }
        std::string flag = s, val;
        bool hasVal = false;
        const size_t eq = s.find('=');
        if (eq != std::string::npos) { flag = s.substr(0, eq); val = s.substr(eq + 1); hasVal = true; }
        auto value = [&]() -> const char* {
            if (hasVal) return val.c_str();
            if (i + 1 >= argc) parseError(flag, "Missing a value for this argument!");
            return argv[++i];
        };
        if (flag == "-k" || flag == "--kvalue") {
            a.k = toUnsigned(flag, value());
            if (a.k % 2 != 1) parseError("-k (--kvalue)", "Value '" + std::to_string(a.k) + "' does not meet constraint: value of K must be odd");
        } else if (flag == "-b" || flag == "--branchsize") a.b = toUnsigned(flag, value());
        else if (flag == "-m" || flag == "--blocksize") a.m = toUnsigned(flag, value());
        else if (flag == "-t" || flag == "--threads") a.t = toUnsigned(flag, value());
        else if (flag == "-a" || flag == "--abundance") a.a = toUnsigned(flag, value());
        else if (flag == "--chunks") a.chunks = toUnsigned(flag, value());
        else if (flag == "--graph") { a.graph = value(); haveGraph = true; }
        else if (flag == "-o" || flag == "--outdir") a.outDir = value();
        else parseError(s, "Couldn't find match for argument");
    }
    if (!haveGraph) parseError("(--graph)", "Required argument missing: graph");
    if (a.fasta.empty()) parseError("(--filenames)", "Required argument missing: filenames");
    return a;
}

}  // namespace

int main(int argc, char** argv)
{
    const Args a = parse(argc, argv);
    lcb_graph* g = nullptr;
    lcb_device* dev = nullptr;
    lcb_seed* seeds = nullptr;
    lcb_block* blocks = nullptr;
    int rc = 0;
    auto fail = [&]() { std::cerr << "error: " << lcb_last_error() << std::endl; rc = 1; };   // sibeliaz.cpp:150-154
    do {
        std::cout << "Loading the graph..." << std::endl;                                      // sibeliaz.cpp:124
        std::vector<const char*> fa;
        for (auto& f : a.fasta) fa.push_back(f.c_str());
        g = lcb_graph_load(a.graph.c_str(), fa.data(), (int)fa.size(), (int)a.k, (int)a.a, (int)a.t);
        if (!g) { fail(); break; }
        std::cout << "Analyzing the graph..." << std::endl;                                    // sibeliaz.cpp:132
        lcb_params p;
        p.k = (int)a.k; p.min_block = (int)a.m; p.max_branch = (int)a.b; p.max_flank = (int)a.b;   // sibeliaz.cpp:133-138
        p.looking_depth = 8; p.phase_size = 256;
        const int64_t nSeeds = lcb_enumerate_seeds(g, (int)a.t, &seeds);
        if (nSeeds < 0) { fail(); break; }
        // GPU selection comes from the environment, never from argv (the wrapper's command line, sibeliaz:146, stays as it is):
        // LCB_GPUS=N uses HIP devices 0..N-1 (as filtered by HIP_VISIBLE_DEVICES), LCB_DEVICE=i uses device i; default one GPU.
        const char* devEnv = getenv("LCB_DEVICE");
        const char* gpusEnv = getenv("LCB_GPUS");
        const int nGpus = gpusEnv && *gpusEnv ? atoi(gpusEnv) : 1;
        int64_t nBlocks = 0;
        lcb_stats st;
        if (nGpus > 1) {
            std::vector<int> ord;
            for (int i = 0; i < nGpus; i++) ord.push_back(i);
            lcb_hooks hk;
            memset(&hk, 0, sizeof(hk));
            hk.abi = LCB_ABI_VERSION; hk.world = 1; hk.progress = 1;
            // the same handle bench.py --gpus N times: devices + tables + RCCL once, then one pass
            lcb_gpus* set = lcb_gpus_create(g, ord.data(), nGpus, &p, nullptr, 0);
            if (!set) { fail(); break; }
            const int rc = lcb_gpus_find_blocks(set, seeds, nSeeds, &hk, &blocks, &nBlocks, &st);
            lcb_gpus_destroy(set);
            if (rc != LCB_OK) { fail(); break; }
        } else {
            dev = lcb_device_create(g, &p, devEnv && *devEnv ? atoi(devEnv) : 0);
            if (!dev) { fail(); break; }
            if (lcb_find_blocks(g, dev, &p, seeds, nSeeds, 1, &blocks, &nBlocks, &st) != LCB_OK) { fail(); break; }
        }
        if (getenv("LCB_VERBOSE"))
            std::cerr << "lcb: seeds=" << st.seeds << " blocks=" << st.blocks_found << " conflicts=" << st.failures << " launches=" << st.launches
                      << " big_retries=" << st.big_retries << " kernel_ms=" << st.kernel_ms << " loop_ms=" << st.wall_ms << " | rounds=" << st.rounds
                      << " job_launches=" << st.recompute_launches << " (" << st.conflict_launches << " at a conflict) jobs=" << st.recomputed_seeds << " used="
                      << st.jobs_used << " views=" << st.views_built << " over_predicted=" << st.over_predicted << " process_ms=" << st.process_ms << " plan_ms=" << st.plan_ms << std::endl;
        std::cout << "Generating the output..." << std::endl;                                  // sibeliaz.cpp:142
        // Blocks found / Coverage are printed by GenerateOutput before the files are written (blocksfinder.h:658-661);
        // the counts are only known after trimming, which lcb_generate_output does, so print right after it.
        int64_t nTrimmed = 0;
        double coverage = 0;
        const int orc = lcb_generate_output(g, a.m, blocks, nBlocks, st.blocks_found, a.outDir.c_str(), a.noSeq ? 0 : 1, a.chunks, &nTrimmed, &coverage);
        char buf[64];
        snprintf(buf, sizeof(buf), "%.2f", coverage);
        if (orc != LCB_OK) { fail(); break; }
        std::cout << "Blocks found: " << nTrimmed << std::endl << "Coverage: " << buf << std::endl;
    } while (false);
    lcb_free(blocks);
    lcb_free(seeds);
    lcb_device_destroy(dev);
    lcb_graph_free(g);
    return rc;
}
