/* lcb_oracle.c — CPU restatement (plain C) of the SibeliaZ-LCB block finder.
 *
 * TEST INFRASTRUCTURE ONLY — see lcb_oracle.h. Parity is pinned against the compiled reference
 * (oracle/_ref) through the goldens in tests/golden/ (tests/test_oracle_golden.py).
 *
 * Every function cites the reference file:line it follows. Abbreviations:
 *   BF  = SibeliaZ-LCB/blocksfinder.h      BFC = SibeliaZ-LCB/blocksfinder.cpp
 *   PH  = SibeliaZ-LCB/path.h              JS  = SibeliaZ-LCB/junctionstorage.h
 *   DK  = SibeliaZ-LCB/distancekeeper.h    JA  = SibeliaZ-LCB/common/junctionapi.h
 *   SFP = SibeliaZ-LCB/common/streamfastaparser.cpp   DC = SibeliaZ-LCB/common/dnachar.cpp
 *
 * The structure deliberately mirrors the reference (dense DistanceKeeper, dense vote array,
 * per-chromosome ordered instance sets) so that it is easy to audit; it is written for
 * obviousness, not speed.
 */
#include "lcb_oracle.h"

#include <ctype.h>
#include <errno.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>

/* ------------------------------------------------------------------------------------------- */
/* small helpers                                                                                 */

/* Q16: inside namespace Sibelia the unqualified abs(int64_t) calls resolve to
 * ::abs(int) under g++/libstdc++, i.e. the argument is truncated to 32 bits first. */
static int64_t abs32(int64_t x)
{
    int t = (int)x;
    return t < 0 ? -(int64_t)t : (int64_t)t;
}

static void* xmalloc(size_t n)
{
    void* p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "lcb_oracle: out of memory\n"); abort(); }
    return p;
}

static void* xrealloc(void* q, size_t n)
{
    void* p = realloc(q, n ? n : 1);
    if (!p) { fprintf(stderr, "lcb_oracle: out of memory\n"); abort(); }
    return p;
}

#define GROW(ptr, n, cap, T)                                           \
    do {                                                                \
        if ((n) >= (cap)) {                                             \
            (cap) = (cap) ? (cap) * 2 : 16;                             \
            (ptr) = (T*)xrealloc((ptr), (size_t)(cap) * sizeof(T));     \
        }                                                               \
    } while (0)

/* DC:52-70,82-85 */
static char reverse_char(char ch)
{
    switch (ch) {
    case 'A': return 'T';
    case 'T': return 'A';
    case 'C': return 'G';
    case 'G': return 'C';
    }
    return 'N';
}

/* DC:11 VALID_CHARS */
static int is_valid_char(int ch)
{
    return ch != 0 && strchr("ACGTURYKMSWBDHWNXV", ch) != NULL;
}

/* ------------------------------------------------------------------------------------------- */
/* graph storage: JS:116-698                                                                     */

typedef struct { int32_t id; uint32_t pos; uint8_t used; } Position;                 /* JS:140-151 */
typedef struct { int32_t id; uint32_t chr, idx, pos; char ch, revCh; } Vertex;       /* JS:120-138 */

typedef struct {                  /* BF:182-209 */
    int64_t vid;
    char ch;
    uint64_t count, rank, resolve_first, resolve_second;
} Bundle;

struct orc_graph {
    int64_t k;
    int64_t nChr;
    Position** position;          /* position_[chr] */
    int64_t* nPos;
    int64_t* capPos;
    int64_t nVertex;              /* vertex_.size() */
    Vertex** vertex;              /* vertex_[absId] */
    int64_t* nOcc;
    int64_t* capOcc;
    char** seq;                   /* sequence_[chr], NUL terminated like std::string */
    int64_t* seqLen;
    char** desc;                  /* sequenceDescription_ */
    int64_t nDesc;
    Bundle* bundle;
    int64_t nBundle;
};

/* JunctionSequentialIterator (JS:158-396): chrId_ = +-(chr+1), idx_ */
typedef struct { int64_t chrId; int64_t idx; } SeqIt;

static SeqIt it_make(int64_t chr, int64_t idx, int positive)      /* JS:388 */
{
    SeqIt r; r.idx = idx; r.chrId = positive ? chr + 1 : -(chr + 1); return r;
}
static int it_positive(SeqIt it) { return it.chrId > 0; }                          /* JS:166 */
static int64_t it_chr(SeqIt it) { return abs32(it.chrId) - 1; }                    /* JS:260 */
static int64_t it_vid(const orc_graph* g, SeqIt it)                                /* JS:171 */
{
    int32_t id = g->position[it_chr(it)][it.idx].id;
    return it_positive(it) ? (int64_t)id : -(int64_t)id;
}
static int64_t it_position(const orc_graph* g, SeqIt it)                           /* JS:176 */
{
    int64_t p = g->position[it_chr(it)][it.idx].pos;
    return it_positive(it) ? p : p + g->k;
}
static char it_char(const orc_graph* g, SeqIt it)                                  /* JS:234 */
{
    int64_t chr = it_chr(it);
    int64_t pos = g->position[chr][it.idx].pos;
    if (it_positive(it)) return g->seq[chr][pos + g->k];
    /* pos == 0 reads sequence_[chr][-1] in the reference (an out-of-bounds byte that is 0 in
     * practice -> ReverseChar -> 'N'); defined here as 'N' (JS:642). */
    return pos > 0 ? reverse_char(g->seq[chr][pos - 1]) : 'N';
}
static int it_valid(const orc_graph* g, SeqIt it)                                  /* JS:265 */
{
    return it.idx >= 0 && it.idx < g->nPos[it_chr(it)];
}
/* Diagnostics for tests/emu/engine_model (pricing of resumable seeds; not part of the reference): a `used` byte with bit 1 set besides
 * bit 0 is WATCHED - still used, as far as the algorithm is concerned -, and between orc_watch_begin() and orc_watch_end() the
 * number of pushes made before the first read of a watched byte is recorded (pushes of the replay, BF:271-284, do not count:
 * the device restarts at a checkpoint instead). Per thread. */
static __thread int tl_watchOn = 0, tl_inReplay = 0;
static __thread int64_t tl_pushes = 0, tl_watchFirst = -1;
void orc_watch_begin(void) { tl_watchOn = 1; tl_pushes = 0; tl_watchFirst = -1; tl_inReplay = 0; }
int64_t orc_watch_end(int64_t* pushes) { tl_watchOn = 0; if (pushes) *pushes = tl_pushes; return tl_watchFirst; }
static int used_byte(uint8_t u)
{
    if ((u & 2) && tl_watchOn && tl_watchFirst < 0) tl_watchFirst = tl_pushes;
    return u;
}
static int it_used(const orc_graph* g, SeqIt it)                                   /* JS:270 */
{
    if (it_positive(it)) return used_byte(g->position[it_chr(it)][it.idx].used);
    if (it.idx > 0) return used_byte(g->position[it_chr(it)][it.idx - 1].used);
    return 0;
}
static void it_mark_used(orc_graph* g, SeqIt it)                                   /* JS:285 */
{
    if (it_positive(it)) g->position[it_chr(it)][it.idx].used = 1;
    else if (it.idx > 0) g->position[it_chr(it)][it.idx - 1].used = 1;
}
static SeqIt it_next(SeqIt it) { it.idx += it_positive(it) ? 1 : -1; return it; }  /* JS:376 */
static SeqIt it_prev(SeqIt it) { it.idx += it_positive(it) ? -1 : 1; return it; }  /* JS:382 */
static int it_eq(SeqIt a, SeqIt b) { return a.chrId == b.chrId && a.idx == b.idx; }/* JS:364 */
static int it_less(SeqIt a, SeqIt b)                                               /* JS:349 */
{
    if (it_positive(a) != it_positive(b)) return it_positive(a) < it_positive(b);
    if (it_chr(a) != it_chr(b)) return it_chr(a) < it_chr(b);
    return (uint64_t)a.idx < (uint64_t)b.idx;
}

typedef struct {                  /* Edge, JS:21-114 (only the consumed fields) */
    int64_t startVertex, endVertex, length;
    char ch;
} Edge;

static Edge it_outgoing_edge(const orc_graph* g, SeqIt it)                         /* JS:191 */
{
    Edge e;
    int64_t chr = it_chr(it);
    const Position* now = &g->position[chr][it.idx];
    if (it_positive(it)) {
        const Position* next = &g->position[chr][it.idx + 1];
        e.ch = g->seq[chr][now->pos + g->k];
        e.startVertex = now->id; e.endVertex = next->id;
        e.length = (int64_t)(uint32_t)(next->pos - now->pos);
    } else {
        const Position* next = &g->position[chr][it.idx - 1];
        e.ch = now->pos > 0 ? reverse_char(g->seq[chr][now->pos - 1]) : 'N';
        e.startVertex = -(int64_t)now->id; e.endVertex = -(int64_t)next->id;
        e.length = (int64_t)(uint32_t)(now->pos - next->pos);
    }
    return e;
}

static Edge it_ingoing_edge(const orc_graph* g, SeqIt it)                          /* JS:210 */
{
    Edge e;
    int64_t chr = it_chr(it);
    const Position* now = &g->position[chr][it.idx];
    if (it_positive(it)) {
        const Position* prev = &g->position[chr][it.idx - 1];
        e.ch = g->seq[chr][prev->pos + g->k];
        e.startVertex = prev->id; e.endVertex = now->id;
        e.length = (int64_t)(uint32_t)(now->pos - prev->pos);
    } else {
        const Position* prev = &g->position[chr][it.idx + 1];
        e.ch = prev->pos > 0 ? reverse_char(g->seq[chr][prev->pos - 1]) : 'N';
        e.startVertex = -(int64_t)prev->id; e.endVertex = -(int64_t)now->id;
        e.length = (int64_t)(uint32_t)(prev->pos - now->pos);
    }
    return e;
}

/* --- junction file reader: JA:80-98 --- */
typedef struct { FILE* f; uint32_t nowChr; } JReader;

static int jreader_next(JReader* r, uint32_t* chr, uint32_t* pos, int64_t* id)
{
    for (;; r->nowChr++) {
        unsigned char b[12];
        if (fread(b, 1, 4, r->f) != 4) return 0;
        if (fread(b + 4, 1, 8, r->f) != 8) return 0;
        memcpy(pos, b, 4);
        memcpy(id, b + 4, 8);
        *chr = r->nowChr;
        if (*pos != 0xFFFFFFFFu && *id != INT64_MAX) return 1;
    }
}

static void set_err(char* err, size_t n, const char* a, const char* b)
{
    if (err && n) snprintf(err, n, "%s%s", a, b ? b : "");
}

/* --- FASTA: SFP:28-92 --- */
static int load_fasta(orc_graph* g, const char* file, int64_t* record, char* err, size_t errLen)
{
    FILE* f = fopen(file, "rb");
    if (!f) { set_err(err, errLen, "Can't open file ", file); return -1; }
    int c = fgetc(f);
    while (c != EOF) {
        if (c != '>') {                                            /* SFP:33-36 */
            char m[2] = { (char)c, 0 };
            set_err(err, errLen, "The FASTA header should start with a '>', started with ", m);
            fclose(f); return -1;
        }
        /* header = first whitespace-delimited token of the line (SFP:41-55) */
        char* line = NULL; int64_t n = 0, cap = 0;
        while ((c = fgetc(f)) != EOF && c != '\n') { GROW(line, n, cap, char); line[n++] = (char)c; }
        GROW(line, n, cap, char); line[n] = 0;
        char* s = line; while (*s && isspace((unsigned char)*s)) s++;
        char* e = s; while (*e && !isspace((unsigned char)*e)) e++;
        *e = 0;
        g->desc = (char**)xrealloc(g->desc, (size_t)(g->nDesc + 1) * sizeof(char*));
        g->desc[g->nDesc++] = strdup(s);
        free(line);
        if (*record >= g->nChr) { set_err(err, errLen, "more FASTA records than chromosomes in the graph", NULL); fclose(f); return -1; }
        /* sequence: skip whitespace, stop at '>', validate, upper-case (SFP:60-92) */
        char* seq = NULL; int64_t sn = 0, scap = 0;
        if (c != EOF) c = fgetc(f);
        while (c != EOF && c != '>') {
            if (!isspace(c)) {
                int u = toupper(c);
                if (!is_valid_char(u)) {
                    char m[2] = { (char)c, 0 };
                    set_err(err, errLen, "Found an invalid character in sequence: ", m);
                    free(seq); fclose(f); return -1;
                }
                GROW(seq, sn, scap, char); seq[sn++] = (char)u;
            }
            c = fgetc(f);
        }
        GROW(seq, sn, scap, char); seq[sn] = 0;
        g->seq[*record] = seq; g->seqLen[*record] = sn;
        (*record)++;
    }
    fclose(f);
    return 0;
}

static int vertex_cmp(const void* a, const void* b)                /* JS:134-137 */
{
    const Vertex* x = (const Vertex*)a; const Vertex* y = (const Vertex*)b;
    if (x->chr != y->chr) return x->chr < y->chr ? -1 : 1;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return 0;
}

/* JunctionStorage::Init, JS:572-650 */
orc_graph* orc_load(const char* graph_file, const char* const* fasta, int n_fasta, int64_t k,
                    int64_t abundance, char* err, size_t err_len)
{
    orc_graph* g = (orc_graph*)calloc(1, sizeof(orc_graph));
    g->k = k;
    size_t* abund = NULL; int64_t abundCap = 0;
    {   /* pass 1, JS:576-594 */
        JReader r = { fopen(graph_file, "rb"), 0 };
        if (!r.f) { set_err(err, err_len, "Can't read the input file", NULL); free(g); return NULL; }   /* JA:46-49 */
        uint32_t chr, pos; int64_t id;
        while (jreader_next(&r, &chr, &pos, &id)) {
            if ((int64_t)chr >= g->nChr) {                         /* JS:580-583: at most one push per record */
                g->position = (Position**)xrealloc(g->position, (size_t)(g->nChr + 1) * sizeof(Position*));
                g->nPos = (int64_t*)xrealloc(g->nPos, (size_t)(g->nChr + 1) * sizeof(int64_t));
                g->capPos = (int64_t*)xrealloc(g->capPos, (size_t)(g->nChr + 1) * sizeof(int64_t));
                g->position[g->nChr] = NULL; g->nPos[g->nChr] = 0; g->capPos[g->nChr] = 0;
                g->nChr++;
            }
            if ((int64_t)chr >= g->nChr) { set_err(err, err_len, "junction file has a chromosome without junctions", NULL); fclose(r.f); orc_free(g); free(abund); return NULL; }
            int64_t absId = abs32(id);                             /* JS:585 */
            while (absId >= g->nVertex) {
                if (g->nVertex >= abundCap) {
                    abundCap = abundCap ? abundCap * 2 : 1024;
                    g->vertex = (Vertex**)xrealloc(g->vertex, (size_t)abundCap * sizeof(Vertex*));
                    g->nOcc = (int64_t*)xrealloc(g->nOcc, (size_t)abundCap * sizeof(int64_t));
                    g->capOcc = (int64_t*)xrealloc(g->capOcc, (size_t)abundCap * sizeof(int64_t));
                    abund = (size_t*)xrealloc(abund, (size_t)abundCap * sizeof(size_t));
                }
                g->vertex[g->nVertex] = NULL; g->nOcc[g->nVertex] = 0; g->capOcc[g->nVertex] = 0; abund[g->nVertex] = 0;
                g->nVertex++;
            }
            ++abund[absId];
        }
        fclose(r.f);
    }
    {   /* pass 2, JS:597-617 */
        size_t chrNow = 0; uint32_t idx = 0;
        JReader r = { fopen(graph_file, "rb"), 0 };
        uint32_t chr, pos; int64_t id;
        while (jreader_next(&r, &chr, &pos, &id)) {
            if (chr > chrNow) { chrNow++; idx = 0; }
            int64_t absId = abs32(id);
            if (abund[absId] < (size_t)abundance) {
                Position p; p.id = (int32_t)id; p.pos = pos; p.used = 0;
                GROW(g->position[chr], g->nPos[chr], g->capPos[chr], Position);
                g->position[chr][g->nPos[chr]++] = p;
                Vertex v; v.id = (int32_t)id; v.chr = chr; v.pos = pos; v.idx = idx++; v.ch = 0; v.revCh = 0;
                GROW(g->vertex[absId], g->nOcc[absId], g->capOcc[absId], Vertex);
                g->vertex[absId][g->nOcc[absId]++] = v;
            }
        }
        fclose(r.f);
    }
    free(abund);
    /* FASTA, JS:620-633 */
    g->seq = (char**)calloc((size_t)g->nChr + 1, sizeof(char*));
    g->seqLen = (int64_t*)calloc((size_t)g->nChr + 1, sizeof(int64_t));
    int64_t record = 0;
    for (int i = 0; i < n_fasta; i++) {
        if (load_fasta(g, fasta[i], &record, err, err_len) != 0) { orc_free(g); return NULL; }
    }
    if (record != g->nChr) { set_err(err, err_len, "number of FASTA records differs from chromosomes in the graph", NULL); orc_free(g); return NULL; }
    /* ch / revCh, JS:635-644 */
    for (int64_t i = 0; i < g->nVertex; i++) {
        for (int64_t j = 0; j < g->nOcc[i]; j++) {
            Vertex* v = &g->vertex[i][j];
            v->ch = g->seq[v->chr][(int64_t)v->pos + g->k];
            v->revCh = v->pos > 0 ? reverse_char(g->seq[v->chr][v->pos - 1]) : 'N';
        }
        qsort(g->vertex[i], (size_t)g->nOcc[i], sizeof(Vertex), vertex_cmp);   /* JS:646-649, keys unique */
    }
    return g;
}

void orc_free(orc_graph* g)
{
    if (!g) return;
    for (int64_t i = 0; i < g->nChr; i++) { free(g->position[i]); if (g->seq) free(g->seq[i]); }
    for (int64_t i = 0; i < g->nVertex; i++) free(g->vertex[i]);
    for (int64_t i = 0; i < g->nDesc; i++) free(g->desc[i]);
    free(g->position); free(g->nPos); free(g->capPos); free(g->vertex); free(g->nOcc); free(g->capOcc);
    free(g->seq); free(g->seqLen); free(g->desc); free(g->bundle);
    free(g);
}

int64_t orc_n_chr(const orc_graph* g) { return g->nChr; }
int64_t orc_n_vertices(const orc_graph* g) { return g->nVertex; }
int64_t orc_chr_len(const orc_graph* g, int64_t chr) { return g->seqLen[chr]; }
int64_t orc_chr_n_pos(const orc_graph* g, int64_t chr) { return g->nPos[chr]; }
const char* orc_chr_name(const orc_graph* g, int64_t chr) { return g->desc[chr]; }
void orc_chr_positions(const orc_graph* g, int64_t chr, int32_t* id, uint32_t* pos)
{
    for (int64_t i = 0; i < g->nPos[chr]; i++) { id[i] = g->position[chr][i].id; pos[i] = g->position[chr][i].pos; }
}
static uint8_t* g_usedScratch = NULL;
uint8_t* orc_chr_used(orc_graph* g, int64_t chr)
{
    /* Position is an AoS; expose a packed copy-in/copy-out view through a scratch buffer is not
     * possible for writes, so hand out the address of the first `used` byte with its stride. */
    (void)g_usedScratch;
    return &g->position[chr][0].used;   /* stride = sizeof(Position) = 12, see orc_used_stride() */
}
size_t orc_used_stride(void) { return sizeof(Position); }
void orc_reset_used(orc_graph* g)
{
    for (int64_t c = 0; c < g->nChr; c++)
        for (int64_t i = 0; i < g->nPos[c]; i++) g->position[c][i].used = 0;
}

/* ------------------------------------------------------------------------------------------- */
/* JunctionIterator accessors over vertex_[abs(vid)][j] (JS:400-520)                             */

static int64_t vabs(int64_t vid) { return abs32(vid); }
static int jit_positive(const orc_graph* g, int64_t vid, int64_t j) { return g->vertex[vabs(vid)][j].id == vid; }   /* JS:408 */
static char jit_char(const orc_graph* g, int64_t vid, int64_t j)                                                    /* JS:423 */
{
    const Vertex* v = &g->vertex[vabs(vid)][j];
    return jit_positive(g, vid, j) ? v->ch : v->revCh;
}
static SeqIt jit_seq(const orc_graph* g, int64_t vid, int64_t j)                                                    /* JS:433 */
{
    const Vertex* v = &g->vertex[vabs(vid)][j];
    return it_make(v->chr, v->idx, jit_positive(g, vid, j));
}

/* ------------------------------------------------------------------------------------------- */
/* seed enumeration: BF:461-503,517                                                              */

static int bundle_less(const Bundle* a, const Bundle* b)           /* BF:195-208 */
{
    if (a->count != b->count) return a->count > b->count;
    if (a->rank != b->rank) return a->rank < b->rank;
    if (a->resolve_first != b->resolve_first) return a->resolve_first < b->resolve_first;
    return a->resolve_second < b->resolve_second;
}
static int bundle_qcmp(const void* a, const void* b)
{
    /* the order is total (distinct resolve per bundle), so any sort gives std::sort's result */
    if (bundle_less((const Bundle*)a, (const Bundle*)b)) return -1;
    if (bundle_less((const Bundle*)b, (const Bundle*)a)) return 1;
    return 0;
}

int64_t orc_build_bundles(orc_graph* g)
{
    free(g->bundle); g->bundle = NULL; g->nBundle = 0;
    int64_t cap = 0;
    for (int64_t v = -g->nVertex + 1; v < g->nVertex; v++) {
        /* std::set<char> good; std::map<char,size_t> count — iterated in ascending char order */
        int good[256]; size_t count[256];
        memset(good, 0, sizeof(good)); memset(count, 0, sizeof(count));
        int64_t n = g->nOcc[vabs(v)];
        for (int64_t j = 0; j < n; j++) {
            unsigned char ch = (unsigned char)jit_char(g, v, j);
            if (jit_positive(g, v, j)) good[ch] = 1;
            count[ch] += 1;
        }
        /* std::map<char,...> orders by (signed) char */
        for (int sc = -128; sc < 128; sc++) {
            unsigned char ch = (unsigned char)(char)sc;
            if (count[ch] == 0) continue;
            if (count[ch] > 1 && good[ch]) {
                Bundle b; b.vid = v; b.ch = (char)sc; b.count = count[ch]; b.rank = 0;
                b.resolve_first = SIZE_MAX; b.resolve_second = SIZE_MAX;
                uint64_t base = 1;
                for (int64_t j = 0; j < n; j++) {
                    if (jit_char(g, v, j) == b.ch) {
                        const Vertex* vx = &g->vertex[vabs(v)][j];
                        b.rank += (uint64_t)vx->chr * base;
                        base *= 31;
                        if (jit_positive(g, v, j)) {
                            uint64_t rf = vx->pos, rs = vx->chr;
                            if (rf < b.resolve_first || (rf == b.resolve_first && rs < b.resolve_second)) {
                                b.resolve_first = rf; b.resolve_second = rs;
                            }
                        }
                    }
                }
                GROW(g->bundle, g->nBundle, cap, Bundle);
                g->bundle[g->nBundle++] = b;
            }
        }
    }
    qsort(g->bundle, (size_t)g->nBundle, sizeof(Bundle), bundle_qcmp);   /* BF:517 */
    return g->nBundle;
}

void orc_get_bundle(const orc_graph* g, int64_t i, int64_t* vid, int32_t* ch, uint64_t* count,
                    uint64_t* rank, uint64_t* resolve_pos, uint64_t* resolve_chr)
{
    const Bundle* b = &g->bundle[i];
    *vid = b->vid; *ch = (int32_t)b->ch; *count = b->count; *rank = b->rank;
    *resolve_pos = b->resolve_first; *resolve_chr = b->resolve_second;
}

/* ------------------------------------------------------------------------------------------- */
/* Path: PH:12-697                                                                               */

typedef struct {                  /* Path::Instance, PH:53-181 */
    int backFinished, frontFinished;
    int64_t compareIdx, frontDistance, backDistance;
    SeqIt front, back;
} Instance;

typedef struct { Edge edge; int64_t startDistance; } Point;        /* PH:185-218 */

typedef struct {
    const orc_graph* g;
    int64_t maxBranchSize, minBlockSize, maxFlankingSize;
    int64_t origin, leftBodyFlank, rightBodyFlank;
    /* DistanceKeeper, DK:9-41 */
    int64_t vertices;
    int* distance;
    /* bodies */
    Point* leftBody; int64_t nLeft, capLeft;
    Point* rightBody; int64_t nRight, capRight;
    /* instance_[chr]: ordered multiset of instances by compareIdx_; stored as arrays of pool indices */
    Instance* pool; int64_t nPool, capPool;
    int** set; int64_t* setN; int64_t* setCap;
    int* allInstance; int64_t nAll, capAll;          /* PH:684, insertion ordered */
    int* goodInstance; int64_t nGood, capGood;       /* PH:685 */
    orc_counters* ctr;
    /* Optional footprints (orc_worker_process; not part of the reference): per pool entry ever created, the range of
     * positions whose `used` bit the computation read as 0 — the instance's span plus every look-ahead window walked from its
     * ends. Pool entries are re-created in the same order by the replay (BF:271-284), so the arrays survive path_clear. */
    int fpOn; int64_t* fpLo; int64_t* fpHi; int64_t nFp, capFp;   /* flat positions: chrBase[chr] + idx */
    int64_t* chrBase;
    /* The replay (BF:271-284) re-creates pool entries 0 .. n0-1 exactly as before; what the backward extension creates after it
     * re-uses the pool indices of the entries the forward extension had created beyond the best point. Those are different
     * instances: they get footprint slots of their own (index + fpShift for pool indices >= fpSplit). */
    int64_t fpSplit, fpShift;
    int fpMode;                                     /* 2: as described; 1: a re-used index merges into the old slot; 0: ... and the creating read is not recorded (the round-2 kernel) */
    int64_t* seenV; int64_t nSeenV, capSeenV;       /* with fpOn: every vertex that was ever part of the path (with repeats) */
} Path;

static int64_t fp_slot(const Path* p, int id) { return id < p->fpSplit ? id : id + p->fpShift; }
static void fp_touch(Path* p, int id, SeqIt it)
{
    if (!p->fpOn) return;
    const int64_t f = fp_slot(p, id), x = p->chrBase[it.chrId > 0 ? it.chrId - 1 : -it.chrId - 1] + it.idx;
    if (x < p->fpLo[f]) p->fpLo[f] = x;
    if (x > p->fpHi[f]) p->fpHi[f] = x;
}

static FILE* g_trace = NULL;
void orc_set_trace(const char* file)
{
    if (g_trace) { fclose(g_trace); g_trace = NULL; }
    if (file && *file) g_trace = fopen(file, "w");
}

static int dk_is_set(const Path* p, int64_t v) { return p->distance[v + p->vertices] != INT_MAX; }     /* DK:17 */
static void dk_set(Path* p, int64_t v, int d)                                                          /* DK:22 */
{
    p->distance[v + p->vertices] = d;
    if (p->fpOn) { GROW(p->seenV, p->nSeenV, p->capSeenV, int64_t); p->seenV[p->nSeenV++] = v; }
}
static int dk_get(const Path* p, int64_t v) { return p->distance[v + p->vertices]; }                   /* DK:27 */
static void dk_unset(Path* p, int64_t v) { p->distance[v + p->vertices] = INT_MAX; }                   /* DK:32 */

static Path* path_new(const orc_graph* g, const orc_params* prm, orc_counters* ctr)                    /* PH:15-31 */
{
    Path* p = (Path*)calloc(1, sizeof(Path));
    p->g = g; p->ctr = ctr;
    p->maxBranchSize = prm->max_branch; p->minBlockSize = prm->min_block; p->maxFlankingSize = prm->max_flank;
    p->vertices = g->nVertex;
    p->distance = (int*)xmalloc((size_t)(g->nVertex * 2 + 2) * sizeof(int));
    for (int64_t i = 0; i < g->nVertex * 2 + 2; i++) p->distance[i] = INT_MAX;
    p->set = (int**)calloc((size_t)g->nChr + 1, sizeof(int*));
    p->setN = (int64_t*)calloc((size_t)g->nChr + 1, sizeof(int64_t));
    p->setCap = (int64_t*)calloc((size_t)g->nChr + 1, sizeof(int64_t));
    return p;
}

static void path_free(Path* p)
{
    for (int64_t c = 0; c < p->g->nChr; c++) free(p->set[c]);
    free(p->set); free(p->setN); free(p->setCap); free(p->distance); free(p->leftBody); free(p->rightBody);
    free(p->pool); free(p->allInstance); free(p->goodInstance); free(p->fpLo); free(p->fpHi); free(p->chrBase); free(p->seenV); free(p);
}

/* multiset::upper_bound(Instance(seqIt, 0)) under operator< on compareIdx_ (PH:177-180) */
static int64_t set_upper_bound(const Path* p, int64_t chr, int64_t idx)
{
    int64_t lo = 0, hi = p->setN[chr];
    while (lo < hi) {
        int64_t mid = (lo + hi) / 2;
        if (idx < p->pool[p->set[chr][mid]].compareIdx) hi = mid; else lo = mid + 1;
    }
    return lo;
}

/* multiset::insert(Instance(it, distance)) — goes after all equal keys; returns the pool index,
 * which plays the role of the stable InstanceSet::iterator kept in allInstance_/goodInstance_. */
static int set_insert(Path* p, SeqIt it, int64_t distance)                          /* PH:82-91 */
{
    Instance in;
    in.front = it; in.back = it; in.frontDistance = distance; in.backDistance = distance;
    in.compareIdx = it.idx; in.backFinished = 0; in.frontFinished = 0;
    GROW(p->pool, p->nPool, p->capPool, Instance);
    int id = (int)p->nPool;
    p->pool[p->nPool++] = in;
    int64_t chr = it_chr(it);
    if (p->fpOn) {
        const int64_t f = fp_slot(p, id);
        if (f >= p->nFp) {
            while (f >= p->capFp) {
                p->capFp = p->capFp ? p->capFp * 2 : 1024;
                p->fpLo = (int64_t*)realloc(p->fpLo, (size_t)p->capFp * sizeof(int64_t));
                p->fpHi = (int64_t*)realloc(p->fpHi, (size_t)p->capFp * sizeof(int64_t));
            }
            p->fpLo[f] = p->fpHi[f] = p->chrBase[chr] + it.idx; p->nFp = f + 1;
        } else if (p->fpMode != 0) fp_touch(p, id, it);
    }
    int64_t at = set_upper_bound(p, chr, it.idx);
    GROW(p->set[chr], p->setN[chr], p->setCap[chr], int);
    memmove(p->set[chr] + at + 1, p->set[chr] + at, (size_t)(p->setN[chr] - at) * sizeof(int));
    p->set[chr][at] = id;
    p->setN[chr]++;
    return id;
}

static int64_t inst_real_length(const orc_graph* g, const Instance* in)             /* PH:165-168 */
{
    return abs32(it_position(g, in->front) - it_position(g, in->back));
}
static int inst_within(const Instance* in, uint64_t idx)                            /* PH:170-175 */
{
    uint64_t a = (uint64_t)in->front.idx, b = (uint64_t)in->back.idx;
    uint64_t left = a < b ? a : b, right = a < b ? b : a;
    return idx >= left && idx <= right;
}
static int path_is_good(const Path* p, const Instance* in)                          /* PH:645-648 */
{
    return inst_real_length(p->g, in) >= p->minBlockSize;
}
static void inst_change_front(Instance* in, SeqIt it, int64_t distance)             /* PH:113-122 */
{
    in->front = it; in->frontDistance = distance;
    if (!it_positive(in->back)) in->compareIdx = in->front.idx;
}
static void inst_change_back(Instance* in, SeqIt it, int64_t distance)              /* PH:124-133 */
{
    in->back = it; in->backDistance = distance;
    if (it_positive(in->back)) in->compareIdx = in->back.idx;
}

static void path_init(Path* p, int64_t vid, char ch)                                /* PH:33-46 */
{
    const orc_graph* g = p->g;
    p->origin = vid;
    dk_set(p, vid, 0);
    p->leftBodyFlank = p->rightBodyFlank = 0;
    int64_t n = g->nOcc[vabs(vid)];
    for (int64_t j = 0; j < n; j++) {
        if (p->ctr) p->ctr->n_occ++;
        SeqIt seqIt = jit_seq(g, vid, j);
        if (!it_used(g, seqIt) && ch == it_char(g, seqIt)) {
            int id = set_insert(p, seqIt, 0);
            GROW(p->allInstance, p->nAll, p->capAll, int);
            p->allInstance[p->nAll++] = id;
        }
    }
}

static int64_t path_left_distance(const Path* p) { return -p->leftBodyFlank; }      /* PH:235 */
static int64_t path_right_distance(const Path* p) { return p->rightBodyFlank; }     /* PH:240 */
static int64_t path_middle_length(const Path* p) { return path_left_distance(p) + path_right_distance(p); }   /* PH:245 */
static int64_t path_right_vertex(const Path* p)                                     /* PH:285 */
{
    return p->nRight == 0 ? p->origin : p->rightBody[p->nRight - 1].edge.endVertex;
}
static int64_t path_left_vertex(const Path* p)                                      /* PH:320 */
{
    return p->nLeft == 0 ? p->origin : p->leftBody[p->nLeft - 1].edge.startVertex;
}

static int path_compatible(const Path* p, SeqIt start, SeqIt end, const Edge* e)    /* PH:380-428 */
{
    const orc_graph* g = p->g;
    if (p->ctr) p->ctr->n_compat_call++;
    if (it_positive(start) != it_positive(end)) return 0;
    for (SeqIt it = start; !it_eq(it, end); it = it_next(it)) {
        if (p->ctr) p->ctr->n_compat_step++;
        if (it_used(g, it)) return 0;
    }
    int64_t realDiff = it_position(g, end) - it_position(g, start);
    int64_t ancestralDiff = (int64_t)dk_get(p, it_vid(g, end)) - (int64_t)dk_get(p, it_vid(g, start));
    if (it_positive(start)) {
        if (realDiff < 0) return 0;
        SeqIt start1 = it_next(start);
        if ((realDiff > p->maxBranchSize || ancestralDiff > p->maxBranchSize) &&
            (!it_valid(g, start1) || it_char(g, start) != e->ch || !it_eq(end, start1) || it_vid(g, start1) != e->endVertex))
            return 0;
    } else {
        if (-realDiff < 0) return 0;
        SeqIt start1 = it_next(start);
        if ((-realDiff > p->maxBranchSize || ancestralDiff > p->maxBranchSize) &&
            (!it_valid(g, start1) || it_char(g, start) != e->ch || !it_eq(end, start1) || it_vid(g, start1) != e->endVertex))
            return 0;
    }
    return 1;
}

static void push_good(Path* p, int id)
{
    GROW(p->goodInstance, p->nGood, p->capGood, int);
    p->goodInstance[p->nGood++] = id;
}
static void push_all(Path* p, int id)
{
    GROW(p->allInstance, p->nAll, p->capAll, int);
    p->allInstance[p->nAll++] = id;
}

/* PointPushFrontWorker::operator(), PH:444-495 (failFlag is never set; complete_ is true, BF:342) */
static void point_push_front_worker(Path* p, int64_t vertex, int64_t distance, const Edge* e)
{
    const orc_graph* g = p->g;
    int64_t n = g->nOcc[vabs(vertex)];
    for (int64_t j = 0; j < n; j++) {
        if (p->ctr) p->ctr->n_occ++;
        int newInstance = 1;
        const Vertex* vx = &g->vertex[vabs(vertex)][j];
        int positive = jit_positive(g, vertex, j);
        SeqIt seqIt = it_make(vx->chr, vx->idx, positive);
        int64_t chr = vx->chr;
        int64_t end = p->setN[chr];
        int64_t inst = set_upper_bound(p, chr, seqIt.idx);
        if (inst != end && inst_within(&p->pool[p->set[chr][inst]], vx->idx)) continue;
        if (positive) {
            if (inst != end && path_compatible(p, seqIt, p->pool[p->set[chr][inst]].front, e)) newInstance = 0;
        } else {
            if (inst != 0 && path_compatible(p, seqIt, p->pool[p->set[chr][--inst]].front, e)) newInstance = 0;
        }
        if (!newInstance && it_vid(g, p->pool[p->set[chr][inst]].front) != vertex) {
            Instance* in = &p->pool[p->set[chr][inst]];
            if (!in->frontFinished) {
                int prevGood = path_is_good(p, in);
                inst_change_front(in, seqIt, distance);
                fp_touch(p, p->set[chr][inst], seqIt);
                if (!prevGood && path_is_good(p, in)) push_good(p, p->set[chr][inst]);
                if (it_used(g, seqIt)) in->frontFinished = 1;
            }
        } else if (!it_used(g, seqIt)) {
            push_all(p, set_insert(p, seqIt, distance));
        }
    }
}

/* PointPushBackWorker::operator(), PH:513-565 */
static void point_push_back_worker(Path* p, int64_t vertex, int64_t distance, const Edge* e)
{
    const orc_graph* g = p->g;
    int64_t n = g->nOcc[vabs(vertex)];
    for (int64_t j = 0; j < n; j++) {
        if (p->ctr) p->ctr->n_occ++;
        int newInstance = 1;
        const Vertex* vx = &g->vertex[vabs(vertex)][j];
        int positive = jit_positive(g, vertex, j);
        SeqIt seqIt = it_make(vx->chr, vx->idx, positive);
        int64_t chr = vx->chr;
        int64_t end = p->setN[chr];
        int64_t inst = set_upper_bound(p, chr, seqIt.idx);
        if (inst != end && inst_within(&p->pool[p->set[chr][inst]], vx->idx)) continue;
        if (positive) {
            if (inst != 0 && path_compatible(p, p->pool[p->set[chr][--inst]].back, seqIt, e)) newInstance = 0;
        } else {
            if (inst != end && path_compatible(p, p->pool[p->set[chr][inst]].back, seqIt, e)) newInstance = 0;
        }
        if (!newInstance && it_vid(g, p->pool[p->set[chr][inst]].back) != vertex) {
            Instance* in = &p->pool[p->set[chr][inst]];
            if (!in->backFinished) {
                int prevGood = path_is_good(p, in);
                inst_change_back(in, seqIt, distance);
                fp_touch(p, p->set[chr][inst], seqIt);
                if (!prevGood && path_is_good(p, in)) push_good(p, p->set[chr][inst]);
                if (it_used(g, seqIt)) in->backFinished = 1;
            }
        } else if (!it_used(g, seqIt)) {
            push_all(p, set_insert(p, seqIt, distance));
        }
    }
}

static int path_point_push_back(Path* p, const Edge* e)                             /* PH:568-584 */
{
    int64_t vertex = e->endVertex;
    if (dk_is_set(p, vertex)) return 0;
    int64_t startVertexDistance = p->rightBodyFlank;
    int64_t endVertexDistance = startVertexDistance + e->length;
    dk_set(p, e->endVertex, (int)endVertexDistance);
    point_push_back_worker(p, vertex, endVertexDistance, e);
    GROW(p->rightBody, p->nRight, p->capRight, Point);
    p->rightBody[p->nRight].edge = *e; p->rightBody[p->nRight].startDistance = startVertexDistance;
    p->nRight++;
    p->rightBodyFlank = startVertexDistance + e->length;                            /* Point::EndDistance, PH:204 */
    if (p->ctr) p->ctr->n_push++;
    if (!tl_inReplay) tl_pushes++;
    return 1;
}

static int path_point_push_front(Path* p, const Edge* e)                            /* PH:586-602 */
{
    int64_t vertex = e->startVertex;
    if (dk_is_set(p, vertex)) return 0;
    int64_t endVertexDistance = p->leftBodyFlank;
    int64_t startVertexDistance = endVertexDistance - e->length;
    dk_set(p, e->startVertex, (int)startVertexDistance);
    point_push_front_worker(p, vertex, startVertexDistance, e);
    GROW(p->leftBody, p->nLeft, p->capLeft, Point);
    p->leftBody[p->nLeft].edge = *e; p->leftBody[p->nLeft].startDistance = startVertexDistance;
    p->nLeft++;
    p->leftBodyFlank = startVertexDistance;
    if (p->ctr) p->ctr->n_push++;
    if (!tl_inReplay) tl_pushes++;
    return 1;
}

static int64_t path_score(const Path* p)                                            /* PH:604-628 */
{
    int64_t ret = 0;
    for (int64_t i = 0; i < p->nGood; i++) {
        const Instance* in = &p->pool[p->goodInstance[i]];
        int64_t score = inst_real_length(p->g, in);
        int64_t rightPenalty = path_right_distance(p) - in->backDistance;
        int64_t leftPenalty = path_left_distance(p) + in->frontDistance;
        if (leftPenalty >= p->maxFlankingSize || rightPenalty >= p->maxFlankingSize) {
            ret = -INT32_MAX;
            break;
        } else {
            score -= (rightPenalty + leftPenalty) * (rightPenalty + leftPenalty);
        }
        ret += score;
    }
    return ret;
}

static void path_clear(Path* p)                                                     /* PH:650-677 */
{
    for (int64_t i = 0; i < p->nLeft; i++) dk_unset(p, p->leftBody[i].edge.startVertex);
    for (int64_t i = 0; i < p->nRight; i++) dk_unset(p, p->rightBody[i].edge.endVertex);
    p->nLeft = 0; p->nRight = 0;
    dk_unset(p, p->origin);
    for (int64_t i = 0; i < p->nAll; i++) p->setN[it_chr(p->pool[p->allInstance[i]].front)] = 0;
    p->nAll = 0; p->nGood = 0; p->nPool = 0;
}

/* ------------------------------------------------------------------------------------------- */
/* BlocksFinder: BF:178-929                                                                      */

typedef struct {
    orc_graph* g;
    orc_params prm;
    Path* path;
    uint32_t* count;              /* BF:341, size 2V+1 */
    int64_t* data; int64_t nData, capData;
    Instance* best; int64_t nBest, capBest;      /* bestInstance */
    orc_counters* ctr;
} Finder;

typedef struct { int64_t count; SeqIt origin; } NextVertex;        /* BF:692-706 (diff is dead) */

/* MostPopularVertex, BF:708-768 */
static int64_t most_popular_vertex(Finder* f, int forward, int tryUsed, NextVertex* out)
{
    const orc_graph* g = f->g;
    Path* p = f->path;
    NextVertex ret; ret.count = 0; ret.origin.chrId = 0; ret.origin.idx = 0;
    int64_t bestVid = 0;
    int64_t startVid = forward ? path_right_vertex(p) : path_left_vertex(p);
    const int* instList = p->nGood >= 2 ? p->goodInstance : p->allInstance;
    int64_t nList = p->nGood >= 2 ? p->nGood : p->nAll;
    if (f->ctr) f->ctr->n_vote++;
    for (int64_t i = 0; i < nList; i++) {
        const Instance* inst = &p->pool[instList[i]];
        int64_t nowVid = forward ? it_vid(g, inst->back) : it_vid(g, inst->front);
        if (nowVid == startVid) {
            int64_t weight = abs32(it_position(g, inst->front) - it_position(g, inst->back)) + 1;
            SeqIt origin = forward ? inst->back : inst->front;
            SeqIt it = forward ? it_next(origin) : it_prev(origin);
            for (size_t d = 1; it_valid(g, it) && (d < (size_t)f->prm.looking_depth ||
                               abs32(it_position(g, it) - it_position(g, origin)) <= f->prm.max_branch); d++) {
                if (f->ctr) f->ctr->n_walk++;
                fp_touch(p, instList[i], it);
                int64_t vid = it_vid(g, it);
                if (!dk_is_set(p, vid) && (!it_used(g, it) || tryUsed)) {
                    int64_t adjVid = vid + g->nVertex;
                    if (f->count[adjVid] == 0) {
                        GROW(f->data, f->nData, f->capData, int64_t);
                        f->data[f->nData++] = adjVid;
                    }
                    f->count[adjVid] += (uint32_t)weight;
                    if ((int64_t)f->count[adjVid] > ret.count ||
                        ((int64_t)f->count[adjVid] == ret.count && it_less(origin, ret.origin))) {
                        ret.origin = origin;
                        ret.count = f->count[adjVid];
                        bestVid = vid;
                    }
                } else {
                    break;
                }
                it = forward ? it_next(it) : it_prev(it);
            }
        }
    }
    for (int64_t i = 0; i < f->nData; i++) f->count[f->data[i]] = 0;
    f->nData = 0;
    *out = ret;
    return bestVid;
}

static void snapshot_best(Finder* f)                               /* BF:820-824, 883-887 */
{
    Path* p = f->path;
    f->nBest = 0;
    for (int64_t i = 0; i < p->nGood; i++) {
        GROW(f->best, f->nBest, f->capBest, Instance);
        f->best[f->nBest++] = p->pool[p->goodInstance[i]];
    }
}

/* ExtendPathForward, BF:770-832 */
static int extend_path_forward(Finder* f, size_t* bestRightSize, int64_t* bestScore, int64_t* nowScore)
{
    const orc_graph* g = f->g;
    Path* p = f->path;
    int success = 0;
    NextVertex nv;
    int64_t next = most_popular_vertex(f, 1, 0, &nv);
    if (next == 0) next = most_popular_vertex(f, 1, 1, &nv);
    if (g_trace) fprintf(g_trace, "VF %lld %lld %lld %lld\n", (long long)next, (long long)nv.count, (long long)(next ? nv.origin.chrId : 0), (long long)(next ? nv.origin.idx : 0));
    if (next != 0) {
        for (SeqIt it = nv.origin; it_vid(g, it) != next; it = it_next(it)) {
            Edge e = it_outgoing_edge(g, it);
            success = path_point_push_back(p, &e);
            if (success) {
                *nowScore = path_score(p);
                if (g_trace) fprintf(g_trace, "PB %lld %lld %lld %lld %lld\n", (long long)e.endVertex, (long long)p->rightBodyFlank, (long long)p->nAll, (long long)p->nGood, (long long)*nowScore);
                if (*nowScore > *bestScore) {
                    *bestScore = *nowScore;
                    *bestRightSize = (size_t)p->nRight + 1;
                    if (*nowScore > 0) snapshot_best(f);
                }
            } else if (g_trace) fprintf(g_trace, "PB- %lld\n", (long long)e.endVertex);
        }
    }
    return success;
}

/* ExtendPathBackward, BF:834-895 (no tryUsed retry: BF:845-848 is commented out) */
static int extend_path_backward(Finder* f, size_t* bestLeftSize, int64_t* bestScore, int64_t* nowScore)
{
    const orc_graph* g = f->g;
    Path* p = f->path;
    int success = 0;
    NextVertex nv;
    int64_t next = most_popular_vertex(f, 0, 0, &nv);
    if (g_trace) fprintf(g_trace, "VB %lld %lld %lld %lld\n", (long long)next, (long long)nv.count, (long long)(next ? nv.origin.chrId : 0), (long long)(next ? nv.origin.idx : 0));
    if (next != 0) {
        for (SeqIt it = nv.origin; it_vid(g, it) != next; it = it_prev(it)) {
            Edge e = it_ingoing_edge(g, it);
            success = path_point_push_front(p, &e);
            if (success) {
                *nowScore = path_score(p);
                if (g_trace) fprintf(g_trace, "PF %lld %lld %lld %lld %lld\n", (long long)e.startVertex, (long long)p->leftBodyFlank, (long long)p->nAll, (long long)p->nGood, (long long)*nowScore);
                if (*nowScore > *bestScore) {
                    *bestScore = *nowScore;
                    *bestLeftSize = (size_t)p->nLeft + 1;
                    if (*nowScore > 0) snapshot_best(f);
                }
            } else if (g_trace) fprintf(g_trace, "PF- %lld\n", (long long)e.startVertex);
        }
    }
    return success;
}

/* ProcessVertex::Process, BF:228-310 */
static void process(Finder* f, int64_t vid, char initChar, int64_t* bestScoreOut)
{
    Path* p = f->path;
    int64_t score = 0;            /* uninitialised in the reference; only read after a successful push wrote it (Q1) */
    f->nBest = 0;
    if (f->ctr) f->ctr->n_process++;
    if (g_trace) fprintf(g_trace, "S %lld %d\n", (long long)vid, (int)initChar);
    path_init(p, vid, initChar);
    int64_t bestScore = 0;
    size_t bestRightSize = (size_t)p->nRight + 1;
    size_t bestLeftSize = (size_t)p->nLeft + 1;
    int64_t minRun = f->prm.max_branch * 2;
    for (;;) {                                                     /* BF:255-269 */
        int ret = 1, positive = 0;
        int64_t prevLength = path_middle_length(p);
        while ((ret = extend_path_forward(f, &bestRightSize, &bestScore, &score)) && path_middle_length(p) - prevLength <= minRun)
            positive = positive || (score > 0);
        if (!ret || !positive) break;
    }
    {                                                              /* BF:271-284 */
        int64_t nEdge = (int64_t)bestRightSize - 1;
        Edge* bestEdge = (Edge*)xmalloc((size_t)(nEdge > 0 ? nEdge : 1) * sizeof(Edge));
        for (int64_t i = 0; i < nEdge; i++) bestEdge[i] = p->rightBody[i].edge;
        path_clear(p);
        path_init(p, vid, initChar);
        tl_inReplay = 1;
        for (int64_t i = 0; i < nEdge; i++) path_point_push_back(p, &bestEdge[i]);
        tl_inReplay = 0;
        free(bestEdge);
        if (p->fpOn && p->fpMode == 2) { p->fpSplit = p->nPool; p->fpShift = p->nFp - p->nPool; }
    }
    if (g_trace) fprintf(g_trace, "R %lld %lld %lld\n", (long long)bestRightSize, (long long)p->nAll, (long long)p->nGood);
    for (;;) {                                                     /* BF:292-306; note the stray ';' at BF:297 (Q1) */
        int ret = 1, positive = 0;
        int64_t prevLength = path_middle_length(p);
        while ((ret = extend_path_backward(f, &bestLeftSize, &bestScore, &score)) && path_middle_length(p) - prevLength <= minRun)
            ;
        positive = positive || (score > 0);
        if (!ret || !positive) break;
    }
    path_clear(p);
    *bestScoreOut = bestScore;
    if (f->ctr) f->ctr->n_inst_out += (uint64_t)f->nBest;
}

static Finder* finder_new(orc_graph* g, const orc_params* prm, orc_counters* ctr)
{
    Finder* f = (Finder*)calloc(1, sizeof(Finder));
    f->g = g; f->prm = *prm; f->ctr = ctr;
    f->path = path_new(g, prm, ctr);
    f->count = (uint32_t*)calloc((size_t)(g->nVertex * 2 + 1), sizeof(uint32_t));
    return f;
}
static void finder_free(Finder* f)
{
    path_free(f->path); free(f->count); free(f->data); free(f->best); free(f);
}

int64_t orc_process_seed(orc_graph* g, const orc_params* p, int64_t vid, int32_t ch,
                         orc_inst* out, int64_t cap, int64_t* best_score, orc_counters* ctr)
{
    Finder* f = finder_new(g, p, ctr);
    int64_t bs = 0;
    process(f, vid, (char)ch, &bs);
    int64_t n = f->nBest;
    for (int64_t i = 0; i < n && i < cap; i++) {
        out[i].positive = it_positive(f->best[i].front);
        out[i].chr = (uint32_t)it_chr(f->best[i].front);
        out[i].front_idx = (uint32_t)f->best[i].front.idx;
        out[i].back_idx = (uint32_t)f->best[i].back.idx;
    }
    if (best_score) *best_score = bs;
    finder_free(f);
    return n;
}

/* A reusable finder (the per-seed scratch of 2V counters is allocated once) that can also report the footprints of a seed:
 * model / test support for the round engine, which needs them to validate speculative results. */
struct orc_worker { Finder* f; };
orc_worker* orc_worker_new(orc_graph* g, const orc_params* p)
{
    orc_worker* w = (orc_worker*)calloc(1, sizeof(orc_worker));
    w->f = finder_new(g, p, NULL);
    Path* pa = w->f->path;
    pa->fpOn = 1;
    pa->fpMode = getenv("ORC_FP_MODE") ? atoi(getenv("ORC_FP_MODE")) : 2;
    pa->chrBase = (int64_t*)xmalloc((size_t)(g->nChr + 1) * sizeof(int64_t));
    pa->chrBase[0] = 0;
    for (int64_t c = 0; c < g->nChr; c++) pa->chrBase[c + 1] = pa->chrBase[c] + g->nPos[c];
    return w;
}
void orc_worker_free(orc_worker* w) { if (w) { finder_free(w->f); free(w); } }
int64_t orc_worker_process(orc_worker* w, int64_t vid, int32_t ch, orc_inst* out, int64_t cap, int64_t* best_score, orc_counters* ctr,
                           orc_fp* fp, int64_t fp_cap, int64_t* n_fp)
{
    Finder* f = w->f;
    f->ctr = ctr; f->path->ctr = ctr;
    f->path->nFp = 0; f->path->nSeenV = 0; f->path->fpSplit = INT64_MAX; f->path->fpShift = 0;
    int64_t bs = 0;
    process(f, vid, (char)ch, &bs);
    int64_t n = f->nBest;
    for (int64_t i = 0; i < n && i < cap; i++) {
        out[i].positive = it_positive(f->best[i].front);
        out[i].chr = (uint32_t)it_chr(f->best[i].front);
        out[i].front_idx = (uint32_t)f->best[i].front.idx;
        out[i].back_idx = (uint32_t)f->best[i].back.idx;
    }
    if (best_score) *best_score = bs;
    if (n_fp) *n_fp = f->path->nFp;
    for (int64_t i = 0; fp && i < f->path->nFp && i < fp_cap; i++) { fp[i].chr = -1; fp[i].lo = f->path->fpLo[i]; fp[i].hi = f->path->fpHi[i]; }   /* chr -1: flat positions */
    return n;
}

/* the vertices of the last orc_worker_process call's path (every vertex that was pushed at any time, and the seed vertex): |id|, with repeats */
int64_t orc_worker_path_vertices(const orc_worker* w, int64_t* out, int64_t cap)
{
    const Path* p = w->f->path;
    for (int64_t i = 0; i < p->nSeenV && i < cap; i++) out[i] = p->seenV[i] < 0 ? -p->seenV[i] : p->seenV[i];
    return p->nSeenV;
}

/* --- phase loop + ordered commit: BF:312-433 --- */

typedef struct { orc_block* b; int64_t n, cap; } BlockVec;

static void finalize(Finder* f, const Instance* inst, int64_t n, int64_t* blocksFound, BlockVec* bv, uint8_t* invalidChr)   /* BF:312-332 */
{
    orc_graph* g = f->g;
    int64_t currentBlock = ++(*blocksFound);
    for (int64_t i = 0; i < n; i++) {
        const Instance* jt = &inst[i];
        invalidChr[it_chr(jt->front)] = 1;
        orc_block b;
        b.chr = (uint64_t)it_chr(jt->front);
        if (it_positive(jt->front)) {
            b.id = (int32_t)currentBlock;
            b.start = (uint64_t)it_position(g, jt->front);
            b.end = (uint64_t)(it_position(g, jt->back) + g->k);
        } else {
            b.id = (int32_t)-currentBlock;
            b.start = (uint64_t)(it_position(g, jt->back) - g->k);
            b.end = (uint64_t)it_position(g, jt->front);
        }
        GROW(bv->b, bv->n, bv->cap, orc_block);
        bv->b[bv->n++] = b;
        for (SeqIt it = jt->front; !it_eq(it, jt->back); it = it_next(it)) it_mark_used(g, it);
    }
}

int64_t orc_find_blocks(orc_graph* g, const orc_params* p, orc_block** out, orc_stats* st, orc_counters* ctr)
{
    const int64_t phaseSize = 256;                                 /* BF:519 */
    if (!g->bundle) orc_build_bundles(g);
    Finder* f = finder_new(g, p, ctr);
    BlockVec bv = { NULL, 0, 0 };
    int64_t blocksFound = 0, failure = 0;
    uint8_t* invalidChr = (uint8_t*)calloc((size_t)g->nChr + 1, 1);
    typedef struct { Instance* v; int64_t n; } Result;
    Result* result = (Result*)calloc((size_t)phaseSize, sizeof(Result));
    int64_t S = g->nBundle;
    for (int64_t phase = 0; phase < S; phase += phaseSize) {
        int64_t limit = S < phase + phaseSize ? S : phase + phaseSize;
        /* every seed of the phase sees the `used` bits as of the start of the phase (BF:345-367):
         * workers only read shared state, so a serial pass without commits is equivalent. */
        for (int64_t idx = phase; idx < limit; idx++) {
            int64_t bs;
            process(f, g->bundle[idx].vid, g->bundle[idx].ch, &bs);
            Result* r = &result[idx - phase];
            r->v = (Instance*)xrealloc(r->v, (size_t)(f->nBest ? f->nBest : 1) * sizeof(Instance));
            memcpy(r->v, f->best, (size_t)f->nBest * sizeof(Instance));
            r->n = f->nBest;
        }
        /* thread 0, BF:372-414 */
        for (int64_t idx = phase; idx < limit; idx++) {
            Result* r = &result[idx - phase];
            if (r->n > 1) {
                int isGood = 1;
                for (int64_t i = 0; i < r->n && isGood; i++) {
                    const Instance* inst = &r->v[i];
                    if (!invalidChr[it_chr(inst->front)]) continue;
                    for (SeqIt it = inst->front; !it_eq(it, inst->back); it = it_next(it)) {
                        if (it_used(g, it)) { isGood = 0; break; }
                    }
                }
                if (isGood) {
                    finalize(f, r->v, r->n, &blocksFound, &bv, invalidChr);
                } else {
                    failure++;
                    int64_t bs;
                    process(f, g->bundle[idx].vid, g->bundle[idx].ch, &bs);
                    if (f->nBest > 1) finalize(f, f->best, f->nBest, &blocksFound, &bv, invalidChr);
                }
            }
        }
        memset(invalidChr, 0, (size_t)g->nChr + 1);                 /* BF:416 */
    }
    for (int64_t i = 0; i < phaseSize; i++) free(result[i].v);
    free(result); free(invalidChr);
    finder_free(f);
    if (st) { st->blocks_found = blocksFound; st->failures = failure; st->seeds = S; }
    *out = bv.b;
    return bv.n;
}

void orc_free_blocks(orc_block* b) { free(b); }

/* ------------------------------------------------------------------------------------------- */
/* libstdc++ std::sort restated (bits/stl_algo.h: __introsort_loop, __final_insertion_sort,    */
/* bits/stl_heap.h). The reference relies on its permutation of equal keys.      */

typedef int (*less_fn)(const void*, const void*, void*);
typedef struct { char* base; size_t size; less_fn less; void* ctx; char* tmp; char* tmp2; } Sorter;

#define EL(s, i) ((s)->base + (size_t)(i) * (s)->size)
static void el_swap(Sorter* s, int64_t a, int64_t b)
{
    if (a == b) return;
    memcpy(s->tmp2, EL(s, a), s->size); memcpy(EL(s, a), EL(s, b), s->size); memcpy(EL(s, b), s->tmp2, s->size);
}

static void adjust_heap(Sorter* s, int64_t first, int64_t holeIndex, int64_t len, const char* value)   /* __adjust_heap */
{
    const int64_t topIndex = holeIndex;
    int64_t secondChild = holeIndex;
    while (secondChild < (len - 1) / 2) {
        secondChild = 2 * (secondChild + 1);
        if (s->less(EL(s, first + secondChild), EL(s, first + secondChild - 1), s->ctx)) secondChild--;
        memcpy(EL(s, first + holeIndex), EL(s, first + secondChild), s->size);
        holeIndex = secondChild;
    }
    if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
        secondChild = 2 * (secondChild + 1);
        memcpy(EL(s, first + holeIndex), EL(s, first + secondChild - 1), s->size);
        holeIndex = secondChild - 1;
    }
    /* __push_heap */
    int64_t parent = (holeIndex - 1) / 2;
    while (holeIndex > topIndex && s->less(EL(s, first + parent), value, s->ctx)) {
        memcpy(EL(s, first + holeIndex), EL(s, first + parent), s->size);
        holeIndex = parent;
        parent = (holeIndex - 1) / 2;
    }
    memcpy(EL(s, first + holeIndex), value, s->size);
}

static void heap_sort(Sorter* s, int64_t first, int64_t last)      /* __partial_sort(first,last,last) */
{
    int64_t len = last - first;
    char* value = (char*)xmalloc(s->size);
    if (len >= 2) {                                                /* __make_heap */
        int64_t parent = (len - 2) / 2;
        for (;;) {
            memcpy(value, EL(s, first + parent), s->size);
            adjust_heap(s, first, parent, len, value);
            if (parent == 0) break;
            parent--;
        }
    }
    while (last - first > 1) {                                     /* __sort_heap / __pop_heap */
        --last;
        memcpy(value, EL(s, last), s->size);
        memcpy(EL(s, last), EL(s, first), s->size);
        adjust_heap(s, first, 0, last - first, value);
    }
    free(value);
}

static void move_median_to_first(Sorter* s, int64_t result, int64_t a, int64_t b, int64_t c)
{
    if (s->less(EL(s, a), EL(s, b), s->ctx)) {
        if (s->less(EL(s, b), EL(s, c), s->ctx)) el_swap(s, result, b);
        else if (s->less(EL(s, a), EL(s, c), s->ctx)) el_swap(s, result, c);
        else el_swap(s, result, a);
    } else if (s->less(EL(s, a), EL(s, c), s->ctx)) el_swap(s, result, a);
    else if (s->less(EL(s, b), EL(s, c), s->ctx)) el_swap(s, result, c);
    else el_swap(s, result, b);
}

static int64_t unguarded_partition(Sorter* s, int64_t first, int64_t last, int64_t pivot)
{
    for (;;) {
        while (s->less(EL(s, first), EL(s, pivot), s->ctx)) ++first;
        --last;
        while (s->less(EL(s, pivot), EL(s, last), s->ctx)) --last;
        if (!(first < last)) return first;
        el_swap(s, first, last);
        ++first;
    }
}

static void introsort_loop(Sorter* s, int64_t first, int64_t last, int64_t depth)
{
    while (last - first > 16) {
        if (depth == 0) { heap_sort(s, first, last); return; }
        --depth;
        int64_t mid = first + (last - first) / 2;
        move_median_to_first(s, first, first + 1, mid, last - 1);
        int64_t cut = unguarded_partition(s, first + 1, last, first);
        introsort_loop(s, cut, last, depth);
        last = cut;
    }
}

static void unguarded_linear_insert(Sorter* s, int64_t last)
{
    memcpy(s->tmp, EL(s, last), s->size);
    int64_t next = last - 1;
    while (s->less(s->tmp, EL(s, next), s->ctx)) {
        memcpy(EL(s, last), EL(s, next), s->size);
        last = next;
        --next;
    }
    memcpy(EL(s, last), s->tmp, s->size);
}

static void insertion_sort(Sorter* s, int64_t first, int64_t last)
{
    if (first == last) return;
    for (int64_t i = first + 1; i != last; ++i) {
        if (s->less(EL(s, i), EL(s, first), s->ctx)) {
            memcpy(s->tmp, EL(s, i), s->size);
            memmove(EL(s, first + 1), EL(s, first), (size_t)(i - first) * s->size);
            memcpy(EL(s, first), s->tmp, s->size);
        } else unguarded_linear_insert(s, i);
    }
}

void orc_introsort(void* base, size_t n, size_t size, less_fn less, void* ctx)
{
    if (n == 0) return;
    Sorter s; s.base = (char*)base; s.size = size; s.less = less; s.ctx = ctx;
    s.tmp = (char*)xmalloc(size); s.tmp2 = (char*)xmalloc(size);
    int64_t lg = 0; for (size_t t = n; t > 1; t >>= 1) lg++;       /* std::__lg */
    introsort_loop(&s, 0, (int64_t)n, lg * 2);
    if ((int64_t)n > 16) {                                         /* __final_insertion_sort */
        insertion_sort(&s, 0, 16);
        for (int64_t i = 16; i < (int64_t)n; ++i) unguarded_linear_insert(&s, i);
    } else insertion_sort(&s, 0, (int64_t)n);
    free(s.tmp); free(s.tmp2);
}

/* ------------------------------------------------------------------------------------------- */
/* output: BF:605-670, BFC:141-174                                                               */

static int64_t block_id(const orc_block* b) { return abs32(b->id); }                /* BFC:64 */

static int less_multiplicity(const void* a, const void* b, void* ctx)               /* BF:584-603 */
{
    const int* mult = (const int*)ctx;
    const orc_block* x = (const orc_block*)a; const orc_block* y = (const orc_block*)b;
    int m1 = mult[block_id(x)], m2 = mult[block_id(y)];
    if (m1 != m2) return m1 > m2;
    return block_id(x) < block_id(y);
}
static int less_block(const void* a, const void* b, void* ctx)                      /* BFC:104-107 */
{
    (void)ctx;
    const orc_block* x = (const orc_block*)a; const orc_block* y = (const orc_block*)b;
    if (block_id(x) != block_id(y)) return block_id(x) < block_id(y);
    if (x->chr != y->chr) return x->chr < y->chr;
    return x->start < y->start;
}
static int less_by_id(const void* a, const void* b, void* ctx)                      /* BFC:32-35 */
{
    (void)ctx;
    return block_id((const orc_block*)a) < block_id((const orc_block*)b);
}

int64_t orc_generate_output(const orc_graph* g, int64_t min_block, const orc_block* blocks, int64_t n,
                            int64_t blocks_found, const char* out_dir, double* coverage, char* err, size_t err_len)
{
    uint8_t** covered = (uint8_t**)calloc((size_t)g->nChr + 1, sizeof(uint8_t*));
    for (int64_t i = 0; i < g->nChr; i++) covered[i] = (uint8_t*)calloc((size_t)g->seqLen[i] + 1, 1);   /* BF:607-611 */
    int64_t trimmedId = 1;
    orc_block* inst = (orc_block*)xmalloc((size_t)(n ? n : 1) * sizeof(orc_block));
    memcpy(inst, blocks, (size_t)n * sizeof(orc_block));
    int* copies = (int*)calloc((size_t)blocks_found + 2, sizeof(int));
    for (int64_t i = 0; i < n; i++) copies[block_id(&inst[i])]++;
    orc_introsort(inst, (size_t)n, sizeof(orc_block), less_multiplicity, copies);                       /* GroupBy, BF:100-110,623 */
    orc_block* trimmed = (orc_block*)xmalloc((size_t)(n ? n : 1) * sizeof(orc_block));
    orc_block* buffer = (orc_block*)xmalloc((size_t)(n ? n : 1) * sizeof(orc_block));
    int64_t nTrim = 0;
    for (int64_t now = 0; now < n;) {
        int64_t prev = now;
        for (; now < n && !less_multiplicity(&inst[prev], &inst[now], copies); now++)
            ;
        int64_t nBuf = 0;
        for (int64_t i = prev; i < now; i++) {                     /* BF:627-639 */
            uint64_t chr = inst[i].chr, start = inst[i].start, end = inst[i].end;
            for (; covered[chr][start] && start < end; start++)
                ;
            for (; covered[chr][end] && end > start; end--)
                ;
            if ((int64_t)(end - start) >= min_block) {
                orc_block t; t.id = (int32_t)((inst[i].id > 0 ? 1 : -1) * trimmedId); t.chr = chr; t.start = start; t.end = end;
                buffer[nBuf++] = t;
                memset(covered[chr] + start, 1, (size_t)(end - start));
            }
        }
        if (nBuf > 1) {
            trimmedId++;
            for (int64_t i = 0; i < nBuf; i++) trimmed[nTrim++] = buffer[i];
        } else {
            for (int64_t i = 0; i < nBuf; i++) memset(covered[buffer[i].chr] + buffer[i].start, 0, (size_t)(buffer[i].end - buffer[i].start));
        }
    }
    uint64_t total = 0, totalBlock = 0;                            /* BFC:109-124 */
    for (int64_t i = 0; i < g->nChr; i++) total += (uint64_t)g->seqLen[i];
    for (int64_t i = 0; i < nTrim; i++) totalBlock += trimmed[i].end - trimmed[i].start;
    if (coverage) *coverage = (double)totalBlock / (double)total;
    orc_introsort(trimmed, (size_t)nTrim, sizeof(orc_block), less_block, NULL);                          /* BF:662 */
    int rc = mkdir(out_dir, 0755);                                 /* BFC:15-27 */
    if (rc != 0 && errno != EEXIST) { set_err(err, err_len, "Cannot create dir ", out_dir); return -1; }
    char* fn = (char*)xmalloc(strlen(out_dir) + 32);
    sprintf(fn, "%s/blocks_coords.gff", out_dir);
    FILE* out = fopen(fn, "w");
    if (!out) { set_err(err, err_len, "Cannot open file ", fn); free(fn); return -1; }
    orc_introsort(trimmed, (size_t)nTrim, sizeof(orc_block), less_by_id, NULL);                          /* BFC:146 */
    fprintf(out, "##gff-version 3.1.26\n");
    for (int64_t i = 0; i < g->nChr; i++) fprintf(out, "##sequence-region %s 1 %lld\n", g->desc[i], (long long)g->seqLen[i]);
    for (int64_t i = 0; i < nTrim; i++) {
        fprintf(out, "%s\tSibeliaZ\tSO:0000856\t%llu\t%llu\t.\t%s\t.\tID=%lld\n", g->desc[trimmed[i].chr],
                (unsigned long long)(trimmed[i].start + 1), (unsigned long long)trimmed[i].end,
                trimmed[i].id > 0 ? "+" : "-", (long long)block_id(&trimmed[i]));
    }
    fclose(out);
    free(fn);
    for (int64_t i = 0; i < g->nChr; i++) free(covered[i]);
    free(covered); free(inst); free(copies); free(trimmed); free(buffer);
    return trimmedId - 1;
}
