// TEST-ONLY lockstep wavefront emulator API (see hip/hip_runtime.h in this directory).
#ifndef LCB_EMU_RUNTIME_H
#define LCB_EMU_RUNTIME_H
#include <cstdint>
#include <functional>
// Runs `body` once per lane (64 lanes, lockstep at cross-lane operations) as workgroup blockIdx.
void emu_run_wave(uint32_t blockId, const std::function<void()>& body);
// Same for a workgroup of nWaves wavefronts (maximally skewed schedule between waves).
void emu_run_block(uint32_t blockId, int nWaves, const std::function<void()>& body);
uint64_t emu_collective_count();
#endif
