// emu_runtime.cpp — TEST-ONLY lockstep wavefront emulator (see tests/emu/hip/hip_runtime.h).
// 64 coroutine lanes per wavefront on one OS thread; x86-64 System V only.
#include "emu_runtime.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

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

constexpr int kLanes = 64;
constexpr size_t kStack = 512 * 1024;

struct Lane {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = true;
    int kind = 0, arg = 0, line = 0;
    const char* file = nullptr;
    uint64_t value = 0, result = 0;
};

struct Wave {
    Lane lane[kLanes];
    void* schedSp = nullptr;
    int current = 0;
    uint32_t block = 0;
    std::function<void()> body;
    uint64_t collectives = 0;
};

Wave g_wave;

void laneEntry()
{
    g_wave.body();
    Lane& l = g_wave.lane[g_wave.current];
    l.done = true;
    emu_switch(&l.sp, g_wave.schedSp);
    abort();   // a finished lane is never resumed
}

}  // namespace

EmuDim3 emu_thread_idx() { return EmuDim3{(uint32_t)g_wave.current, 0, 0}; }
EmuDim3 emu_block_idx() { return EmuDim3{g_wave.block, 0, 0}; }

uint64_t emu_collective(int kind, uint64_t value, int arg, const char* file, int line)
{
    Lane& l = g_wave.lane[g_wave.current];
    l.kind = kind; l.value = value; l.arg = arg; l.file = file; l.line = line;
    emu_switch(&l.sp, g_wave.schedSp);
    return l.result;
}

uint64_t emu_collective_count() { return g_wave.collectives; }

void emu_run_wave(uint32_t blockId, const std::function<void()>& body)
{
    Wave& w = g_wave;
    w.body = body; w.block = blockId;
    for (int i = 0; i < kLanes; i++) {
        Lane& l = w.lane[i];
        if (!l.stack) l.stack = (char*)aligned_alloc(64, kStack);
        uint64_t* top = (uint64_t*)(((uintptr_t)(l.stack + kStack) & ~(uintptr_t)15) - 64);
        memset(top, 0, 64);
        top[6] = (uint64_t)(uintptr_t)&laneEntry;    // popped by `ret` after the six callee-saved registers
        l.sp = top; l.done = false; l.kind = 0;
    }
    for (;;) {
        int live = 0;
        for (int i = 0; i < kLanes; i++) {
            if (w.lane[i].done) continue;
            w.current = i;
            emu_switch(&w.schedSp, w.lane[i].sp);
            if (!w.lane[i].done) live++;
        }
        if (!live) break;
        // all live lanes are parked at a cross-lane operation: it must be the same one
        int first = -1;
        for (int i = 0; i < kLanes; i++) {
            if (w.lane[i].done) continue;
            if (first < 0) first = i;
            else if (w.lane[i].kind != w.lane[first].kind || w.lane[i].line != w.lane[first].line || w.lane[i].file != w.lane[first].file) {
                fprintf(stderr, "emu: divergent cross-lane operation: lane %d at %s:%d (kind %d) vs lane %d at %s:%d (kind %d)\n", first,
                        w.lane[first].file, w.lane[first].line, w.lane[first].kind, i, w.lane[i].file, w.lane[i].line, w.lane[i].kind);
                abort();
            }
        }
        if (live != kLanes) {
            // a lane returned from the kernel while others still communicate: the kernels under test never do that
            fprintf(stderr, "emu: %d lanes exited before a cross-lane operation at %s:%d\n", kLanes - live, w.lane[first].file, w.lane[first].line);
            abort();
        }
        w.collectives++;
        const int kind = w.lane[first].kind;
        if (kind == EMU_BALLOT) {
            uint64_t m = 0;
            for (int i = 0; i < kLanes; i++) if (w.lane[i].value) m |= 1ull << i;
            for (int i = 0; i < kLanes; i++) w.lane[i].result = m;
        } else if (kind == EMU_SHFL) {
            for (int i = 0; i < kLanes; i++) w.lane[i].result = w.lane[w.lane[i].arg & 63].value;
        } else {
            for (int i = 0; i < kLanes; i++) w.lane[i].result = 0;
        }
    }
}
