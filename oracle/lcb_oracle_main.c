/* lcb_oracle: command-line front end of the CPU restatement (TEST INFRASTRUCTURE ONLY).
 * Accepts the sibeliaz-lcb flags that matter for parity (sibeliaz.cpp:45-120) and, with --dump <dir>,
 * writes the same intermediate files as oracle/ref_dump.cpp so the two can be diffed byte for byte. */
#include "lcb_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

static void dump_seeds(orc_graph* g, const orc_params* p, const char* file)
{
    FILE* out = fopen(file, "w");
    int64_t S = orc_build_bundles(g);
    orc_inst* inst = (orc_inst*)malloc(sizeof(orc_inst) * 65536);
    for (int64_t i = 0; i < S; i++) {
        int64_t vid; int32_t ch; uint64_t count, rank, rp, rc;
        orc_get_bundle(g, i, &vid, &ch, &count, &rank, &rp, &rc);
        int64_t bs = 0;
        int64_t n = orc_process_seed(g, p, vid, ch, inst, 65536, &bs, NULL);
        if (n == 0) continue;
        fprintf(out, "%lld\t%lld\t%lld", (long long)i, (long long)bs, (long long)n);
        for (int64_t j = 0; j < n; j++)
            fprintf(out, "\t%c,%u,%u,%u", inst[j].positive ? '+' : '-', inst[j].chr, inst[j].front_idx, inst[j].back_idx);
        fputc('\n', out);
    }
    free(inst);
    fclose(out);
}

int main(int argc, char** argv)
{
    const char* graph = NULL; const char* outDir = ""; const char* dump = NULL;
    const char* fasta[4096]; int nFasta = 0;
    orc_params p = { 25, 200, 200, 200, 8 };
    int64_t abundance = 150;
    for (int i = 1; i < argc; i++) {
        const char* a = argv[i];
        if (!strcmp(a, "--graph") && i + 1 < argc) graph = argv[++i];
        else if (!strcmp(a, "-k") && i + 1 < argc) p.k = atoll(argv[++i]);
        else if (!strcmp(a, "-b") && i + 1 < argc) p.max_branch = p.max_flank = atoll(argv[++i]);
        else if (!strcmp(a, "-m") && i + 1 < argc) p.min_block = atoll(argv[++i]);
        else if ((!strcmp(a, "-a") || !strcmp(a, "--abundance")) && i + 1 < argc) abundance = atoll(argv[++i]);
        else if (!strcmp(a, "-o") && i + 1 < argc) outDir = argv[++i];
        else if (!strcmp(a, "-t") && i + 1 < argc) ++i;
        else if (!strcmp(a, "--chunks") && i + 1 < argc) ++i;
        else if (!strcmp(a, "--noseq")) {}
        else if (!strcmp(a, "--dump") && i + 1 < argc) dump = argv[++i];
        else if (nFasta < 4096) fasta[nFasta++] = a;
    }
    if (!graph || nFasta == 0) { fprintf(stderr, "usage: lcb_oracle --graph g.bin [-k -b -m -a -o dir --dump dir] fasta...\n"); return 1; }
    char err[512];
    printf("Loading the graph...\n");
    orc_graph* g = orc_load(graph, fasta, nFasta, p.k, abundance, err, sizeof(err));
    if (!g) { fprintf(stderr, "error: %s\n", err); return 1; }
    printf("Analyzing the graph...\n");
    orc_block* blocks = NULL; orc_stats st; orc_counters ctr; memset(&ctr, 0, sizeof(ctr));
    struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    int64_t S = orc_build_bundles(g);
    int64_t n = orc_find_blocks(g, &p, &blocks, &st, &ctr);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    printf("Generating the output...\n");
    double cov = 0;
    int64_t nb = orc_generate_output(g, p.min_block, blocks, n, st.blocks_found, outDir, &cov, err, sizeof(err));
    if (nb < 0) { fprintf(stderr, "error: %s\n", err); return 1; }
    printf("Blocks found: %lld\nCoverage: %.2f\n", (long long)nb, cov);
    fprintf(stderr, "oracle: seeds=%lld analyze_s=%.3f seeds_per_s=%.1f walk=%llu occ=%llu compat_call=%llu compat_step=%llu inst_out=%llu vote=%llu push=%llu process=%llu failures=%lld\n",
            (long long)S, sec, (double)S / sec, (unsigned long long)ctr.n_walk, (unsigned long long)ctr.n_occ,
            (unsigned long long)ctr.n_compat_call, (unsigned long long)ctr.n_compat_step, (unsigned long long)ctr.n_inst_out,
            (unsigned long long)ctr.n_vote, (unsigned long long)ctr.n_push, (unsigned long long)ctr.n_process, (long long)st.failures);
    if (dump) {
        mkdir(dump, 0755);
        char fn[4096];
        snprintf(fn, sizeof(fn), "%s/bundles.tsv", dump);
        FILE* f = fopen(fn, "w");
        for (int64_t i = 0; i < S; i++) {
            int64_t vid; int32_t ch; uint64_t count, rank, rp, rc;
            orc_get_bundle(g, i, &vid, &ch, &count, &rank, &rp, &rc);
            fprintf(f, "%lld\t%d\t%llu\t%llu\t%llu\t%llu\n", (long long)vid, ch, (unsigned long long)count,
                    (unsigned long long)rank, (unsigned long long)rp, (unsigned long long)rc);
        }
        fclose(f);
        snprintf(fn, sizeof(fn), "%s/pretrim.tsv", dump);
        f = fopen(fn, "w");
        for (int64_t i = 0; i < n; i++)
            fprintf(f, "%d\t%llu\t%llu\t%llu\n", blocks[i].id, (unsigned long long)blocks[i].chr,
                    (unsigned long long)blocks[i].start, (unsigned long long)blocks[i].end);
        fclose(f);
        snprintf(fn, sizeof(fn), "%s/summary.txt", dump);
        f = fopen(fn, "w");
        int64_t P = 0;
        for (int64_t c = 0; c < orc_n_chr(g); c++) P += orc_chr_n_pos(g, c);
        fprintf(f, "blocksFound\t%lld\nfailure\t%lld\nbundles\t%lld\nvertices\t%lld\nchromosomes\t%lld\npositions\t%lld\n",
                (long long)st.blocks_found, (long long)st.failures, (long long)S, (long long)orc_n_vertices(g),
                (long long)orc_n_chr(g), (long long)P);
        fclose(f);
        snprintf(fn, sizeof(fn), "%s/seeds_final.tsv", dump);
        dump_seeds(g, &p, fn);
        orc_reset_used(g);
        snprintf(fn, sizeof(fn), "%s/seeds_init.tsv", dump);
        dump_seeds(g, &p, fn);
    }
    orc_free_blocks(blocks);
    orc_free(g);
    return 0;
}
