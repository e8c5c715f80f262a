// tests/emu/hip/hip_runtime.h — TEST-ONLY stand-in for <hip/hip_runtime.h>.
//
// Lets the unmodified device code in sibeliaz_amd/csrc/lcb_kernel.h be compiled with g++ and
// executed on the CPU by a lockstep wavefront emulator (tests/emu/emu_runtime.cpp): the 64 lanes
// of a wavefront are coroutines; a lane runs until it reaches a cross-lane operation
// (__ballot/__shfl/__shfl_xor/wave barrier), where it parks until all live lanes have arrived at
// the same call site. This is a debugging and CPU-CI aid for the kernel LOGIC (it also asserts that
// every cross-lane operation is reached convergently); it is never part of the product and says
// nothing about performance. The include path tests/emu is only ever given to the emulator build.
#ifndef LCB_EMU_HIP_RUNTIME_H
#define LCB_EMU_HIP_RUNTIME_H

#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static thread_local
#define __launch_bounds__(...)

struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct EmuDim3 { uint32_t x, y, z; };

EmuDim3 emu_thread_idx();
EmuDim3 emu_block_idx();
#define threadIdx (emu_thread_idx())
#define blockIdx (emu_block_idx())

enum EmuKind { EMU_BALLOT = 1, EMU_SHFL = 2, EMU_SYNC = 3, EMU_BARRIER = 4, EMU_DPP = 5 };
uint64_t emu_collective(int kind, uint64_t value, int arg, const char* file, int line);

#define __ballot(p) ((unsigned long long)emu_collective(EMU_BALLOT, (p) ? 1u : 0u, 0, __FILE__, __LINE__))
#define __shfl(v, src) ((int)emu_collective(EMU_SHFL, (uint64_t)(uint32_t)(v), (int)((src) & 63), __FILE__, __LINE__))
#define __shfl_xor(v, m) ((int)emu_collective(EMU_SHFL, (uint64_t)(uint32_t)(v), (int)(((emu_thread_idx().x & 63u) ^ (uint32_t)(m)) & 63), __FILE__, __LINE__))
// scalar broadcasts and the DPP row operations the kernel's wave-wide max/min reductions are built from
#define __builtin_amdgcn_readfirstlane(v) ((int)emu_collective(EMU_SHFL, (uint64_t)(uint32_t)(v), 0, __FILE__, __LINE__))
#define __builtin_amdgcn_readlane(v, l) ((int)emu_collective(EMU_SHFL, (uint64_t)(uint32_t)(v), (int)((l) & 63), __FILE__, __LINE__))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rowMask, bankMask, boundCtrl) \
    ((int)emu_collective(EMU_DPP, ((uint64_t)(uint32_t)(old) << 32) | (uint64_t)(uint32_t)(src), (int)((ctrl) | ((rowMask) << 12) | ((bankMask) << 16) | ((boundCtrl) ? 1 << 20 : 0)), __FILE__, __LINE__))
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)emu_collective(EMU_SYNC, 0, 0, __FILE__, __LINE__))
#define __syncthreads() ((void)emu_collective(EMU_BARRIER, 0, 0, __FILE__, __LINE__))

#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __HIP_MEMORY_SCOPE_AGENT 1
template <class T_> static inline void __hip_atomic_store(T_* p, T_ v, int, int) { *p = v; }
template <class T_> static inline T_ __hip_atomic_load(const T_* p, int, int) { return *p; }
static inline void __threadfence_system() {}
static inline uint64_t wall_clock64() { return 0; }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }

// Lanes are interleaved only at cross-lane operations, so plain read-modify-write is atomic here.
static inline int atomicCAS(int* a, int cmp, int val) { int old = *a; if (old == cmp) *a = val; return old; }
static inline uint32_t atomicAdd(uint32_t* a, uint32_t v) { uint32_t old = *a; *a = old + v; return old; }
static inline unsigned long long atomicAdd(unsigned long long* a, unsigned long long v) { unsigned long long old = *a; *a = old + v; return old; }
static inline unsigned long long atomicMax(unsigned long long* a, unsigned long long v) { unsigned long long old = *a; if (v > old) *a = v; return old; }
static inline unsigned long long atomicAdd(unsigned long long* a, long long v) { unsigned long long old = *a; *a = old + (unsigned long long)v; return old; }
static inline uint32_t atomicOr(uint32_t* a, uint32_t v) { uint32_t old = *a; *a = old | v; return old; }
static inline uint32_t atomicMax(uint32_t* a, uint32_t v) { uint32_t old = *a; if (v > old) *a = v; return old; }
static inline uint32_t atomicMin(uint32_t* a, uint32_t v) { uint32_t old = *a; if (v < old) *a = v; return old; }

#endif
