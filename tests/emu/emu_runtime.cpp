// emu_runtime.cpp — TEST-ONLY lockstep wavefront emulator (see tests/emu/hip/hip_runtime.h).
// A workgroup of nWaves x 64 coroutine lanes on one OS thread (state is thread-local, so several OS threads can each
// emulate their own workgroup); x86-64 System V only.
//
// Wave-level cross-lane operations (__ballot/__shfl/wave barrier) complete when all 64 lanes of THAT wave have
// arrived at the same call site; __syncthreads completes when every live lane of the workgroup has arrived.
// Scheduling is deliberately maximally skewed: wave 0 runs until it is blocked at a workgroup barrier (or done) before
// wave 1 gets to run at all, and so on — which is the schedule most likely to expose a missing barrier.
#include "emu_runtime.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hip/hip_runtime.h"

extern "C" void emu_switch(void** fromSp, void* toSp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");

namespace {

constexpr int kWave = 64;
constexpr size_t kStack = 512 * 1024;

struct Lane {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = true, parked = false;
    int kind = 0, arg = 0, line = 0;
    const char* file = nullptr;
    uint64_t value = 0, result = 0;
};

struct Block {
    std::vector<Lane> lane;
    void* schedSp = nullptr;
    int current = 0, nWaves = 1;
    uint32_t block = 0;
    std::function<void()> body;
    uint64_t collectives = 0;
};

thread_local Block g_blk;   // one emulated workgroup per OS thread (emu_main runs independent seeds on several threads)

void laneEntry()
{
    g_blk.body();
    Lane& l = g_blk.lane[g_blk.current];
    l.done = true;
    emu_switch(&l.sp, g_blk.schedSp);
    abort();   // a finished lane is never resumed
}

// Runs every runnable lane of wave w until it parks or finishes. Returns true if any lane ran.
bool sweepWave(int w)
{
    bool ran = false;
    for (int i = w * kWave; i < (w + 1) * kWave; i++) {
        Lane& l = g_blk.lane[i];
        if (l.done || l.parked) continue;
        g_blk.current = i;
        emu_switch(&g_blk.schedSp, l.sp);
        ran = true;
    }
    return ran;
}

// Completes wave w's pending wave-level operation if all its live lanes are parked at it.
// Returns: 0 nothing to do, 1 completed an op, 2 the wave is parked at a workgroup barrier, 3 the wave is done.
int settleWave(int w)
{
    int first = -1, live = 0;
    for (int i = w * kWave; i < (w + 1) * kWave; i++) {
        Lane& l = g_blk.lane[i];
        if (l.done) continue;
        live++;
        if (!l.parked) return 0;
        if (first < 0) first = i;
        else if (l.kind != g_blk.lane[first].kind || l.line != g_blk.lane[first].line || l.file != g_blk.lane[first].file) {
            fprintf(stderr, "emu: divergent cross-lane operation in wave %d: lane %d at %s:%d (kind %d) vs lane %d at %s:%d (kind %d)\n", w, first,
                    g_blk.lane[first].file, g_blk.lane[first].line, g_blk.lane[first].kind, i, l.file, l.line, l.kind);
            abort();
        }
    }
    if (!live) return 3;
    if (live != kWave) {
        fprintf(stderr, "emu: wave %d: %d lanes exited before a cross-lane operation at %s:%d\n", w, kWave - live, g_blk.lane[first].file, g_blk.lane[first].line);
        abort();
    }
    const int kind = g_blk.lane[first].kind;
    if (kind == EMU_BARRIER) return 2;
    g_blk.collectives++;
    Lane* L = &g_blk.lane[w * kWave];
    if (kind == EMU_BALLOT) {
        uint64_t m = 0;
        for (int i = 0; i < kWave; i++) if (L[i].value) m |= 1ull << i;
        for (int i = 0; i < kWave; i++) L[i].result = m;
    } else if (kind == EMU_SHFL) {
        for (int i = 0; i < kWave; i++) L[i].result = L[L[i].arg & 63].value;
    } else if (kind == EMU_DPP) {
        // v_mov_b32_dpp: value = (old << 32) | src; arg = ctrl | rowMask << 12 | bankMask << 16 | boundCtrl << 20
        for (int i = 0; i < kWave; i++) {
            const int a = L[i].arg, ctrl = a & 0xFFF, rowMask = (a >> 12) & 0xF, bankMask = (a >> 16) & 0xF;
            const bool boundCtrl = (a >> 20) & 1;
            const uint32_t old = (uint32_t)(L[i].value >> 32);
            const int row = i / 16, pos = i % 16;
            int src = -1;
            if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl - 0x110; if (pos >= n) src = i - n; }            // row_shr:n
            else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl - 0x100; if (pos + n < 16) src = i + n; }   // row_shl:n
            else if (ctrl == 0x142) { if (row >= 1) src = (row - 1) * 16 + 15; }                                       // row_bcast:15
            else if (ctrl == 0x143) { if (row >= 2) src = 31; }                                                        // row_bcast:31
            else { fprintf(stderr, "emu: DPP control 0x%x is not emulated\n", ctrl); abort(); }
            const bool enabled = ((rowMask >> row) & 1) && ((bankMask >> (pos / 4)) & 1);
            if (!enabled) L[i].result = old;
            else if (src < 0) L[i].result = boundCtrl ? 0u : old;
            else L[i].result = (uint32_t)L[src].value;
        }
    } else {
        for (int i = 0; i < kWave; i++) L[i].result = 0;
    }
    for (int i = 0; i < kWave; i++) L[i].parked = false;
    return 1;
}

}  // namespace

EmuDim3 emu_thread_idx() { return EmuDim3{(uint32_t)g_blk.current, 0, 0}; }
EmuDim3 emu_block_idx() { return EmuDim3{g_blk.block, 0, 0}; }

uint64_t emu_collective(int kind, uint64_t value, int arg, const char* file, int line)
{
    Lane& l = g_blk.lane[g_blk.current];
    l.kind = kind; l.value = value; l.arg = arg; l.file = file; l.line = line; l.parked = true;
    emu_switch(&l.sp, g_blk.schedSp);
    return l.result;
}

uint64_t emu_collective_count() { return g_blk.collectives; }

void emu_run_block(uint32_t blockId, int nWaves, const std::function<void()>& body)
{
    Block& b = g_blk;
    b.body = body; b.block = blockId; b.nWaves = nWaves;
    if ((int)b.lane.size() < nWaves * kWave) b.lane.resize((size_t)nWaves * kWave);
    for (int i = 0; i < nWaves * kWave; i++) {
        Lane& l = b.lane[i];
        if (!l.stack) l.stack = (char*)aligned_alloc(64, kStack);
        uint64_t* top = (uint64_t*)(((uintptr_t)(l.stack + kStack) & ~(uintptr_t)15) - 64);
        memset(top, 0, 64);
        top[6] = (uint64_t)(uintptr_t)&laneEntry;    // popped by `ret` after the six callee-saved registers
        l.sp = top; l.done = false; l.parked = false; l.kind = 0;
    }
    // EMU_SCHED=rr: the waves take turns at every wave-level operation (interleaved progress: what work the waves draw from shared
    // counters depends on it); EMU_SCHED=rev: maximal skew with the LAST wave first. Default: maximal skew, wave 0 first.
    static const char* schedEnv = getenv("EMU_SCHED");
    const bool rr = schedEnv && !strcmp(schedEnv, "rr"), rev = schedEnv && !strcmp(schedEnv, "rev");
    for (;;) {
        // run each wave as far as it can go on its own (maximal skew between waves)
        int atBarrier = 0, done = 0;
        if (rr) {
            for (bool progress = true; progress;) {
                progress = false;
                for (int w = 0; w < nWaves; w++) { sweepWave(w); if (settleWave(w) == 1) progress = true; }
            }
            for (int w = 0; w < nWaves; w++) { const int r = settleWave(w); if (r == 2) atBarrier++; if (r == 3) done++; }
        } else
        for (int q = 0; q < nWaves; q++) {
            const int w = rev ? nWaves - 1 - q : q;
            for (;;) {
                sweepWave(w);
                const int r = settleWave(w);
                if (r == 1) continue;          // a wave-level op completed: keep running this wave
                if (r == 2) atBarrier++;
                if (r == 3) done++;
                break;                          // 0 cannot happen after a full sweep; 2/3: blocked or finished
            }
        }
        if (done == nWaves) break;
        if (atBarrier + done == nWaves && atBarrier > 0) {
            // __syncthreads: every live wave has arrived (exited waves no longer take part)
            b.collectives++;
            for (int i = 0; i < nWaves * kWave; i++) if (!b.lane[i].done) { b.lane[i].parked = false; b.lane[i].result = 0; }
            continue;
        }
        fprintf(stderr, "emu: deadlock (%d waves at a workgroup barrier, %d done, %d total)\n", atBarrier, done, nWaves);
        abort();
    }
}

void emu_run_wave(uint32_t blockId, const std::function<void()>& body) { emu_run_block(blockId, 1, body); }
