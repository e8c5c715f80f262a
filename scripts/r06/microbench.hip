// Round 6 calibration: what do the dependent steps of the seed's wavefront cost on this chip? One wavefront per workgroup (the process
// kernels' situation in the wide / big variants), each test a chain of N dependent operations; ns per operation from wall_clock64
// (100 MHz) and cycles from s_memtime / clock64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define N 4096
__global__ __launch_bounds__(64) void chase(const uint32_t* gbuf, uint32_t gmask, uint32_t* out, unsigned long long* res)
{
    __shared__ uint32_t lds[4096];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 4096; i += 64) lds[i] = (i * 1664525u + 1013904223u) & 4095u;
    __syncthreads();
    unsigned long long t[16]; unsigned long long c[16];
    uint32_t x = lane;
    // 1: dependent LDS reads
    t[0] = wall_clock64(); c[0] = clock64();
    for (int i = 0; i < N; i++) x = lds[x];
    t[1] = wall_clock64(); c[1] = clock64();
    // 2: dependent LDS atomic CAS with return (never succeeds)
    uint32_t y = x;
    for (int i = 0; i < N; i++) y = atomicCAS(&lds[y & 4095u], 0xFFFFFFFFu, 0u) & 4095u;
    t[2] = wall_clock64(); c[2] = clock64();
    // 3: dependent global loads, small footprint (64 KB: L1/L2 resident)
    uint32_t z = (y + lane) & 16383u;
    for (int i = 0; i < N; i++) z = gbuf[z & 16383u];
    t[3] = wall_clock64(); c[3] = clock64();
    // 4: dependent global loads over the whole buffer (256 MB: HBM / MALL)
    uint32_t w = (z * 2654435761u) & gmask;
    for (int i = 0; i < N; i++) w = gbuf[w & gmask];
    t[4] = wall_clock64(); c[4] = clock64();
    // 5: dependent VALU adds (one instruction each)
    uint32_t a = w;
    #pragma unroll 16
    for (int i = 0; i < N * 16; i++) a = a * 3u + (uint32_t)i;
    t[5] = wall_clock64(); c[5] = clock64();
    // 6: readfirstlane + scalar op round trips (VALU -> SGPR -> VALU)
    uint32_t b = a;
    #pragma unroll 16
    for (int i = 0; i < N * 4; i++) { uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)b); b = b + s + lane; }
    t[6] = wall_clock64(); c[6] = clock64();
    // 7: ballot + ffs + readlane
    uint32_t d = b;
    #pragma unroll 8
    for (int i = 0; i < N * 4; i++) { unsigned long long m = __ballot((d & 3u) != 0); uint32_t f = m ? (uint32_t)__ffsll((long long)m) - 1u : 0u; d = d + (uint32_t)__builtin_amdgcn_readlane((int)d, (int)f) + 1u; }
    t[7] = wall_clock64(); c[7] = clock64();
    // 8: LDS write then read of another lane's value (wave-level exchange through LDS)
    uint32_t e = d;
    for (int i = 0; i < N; i++) { lds[lane] = e; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); e = lds[(lane + 1) & 63] + 1; }
    t[8] = wall_clock64(); c[8] = clock64();
    // 9: 8 independent global loads (whole buffer) then a wait: memory-level parallelism
    uint32_t g = e;
    for (int i = 0; i < N / 8; i++) {
        uint32_t q = (g * 2654435761u + lane) & gmask;
        uint32_t r0 = gbuf[q], r1 = gbuf[(q + 4099u) & gmask], r2 = gbuf[(q + 8219u) & gmask], r3 = gbuf[(q + 16411u) & gmask];
        uint32_t r4 = gbuf[(q + 32771u) & gmask], r5 = gbuf[(q + 65537u) & gmask], r6 = gbuf[(q + 131101u) & gmask], r7 = gbuf[(q + 262147u) & gmask];
        g = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    }
    t[9] = wall_clock64(); c[9] = clock64();
    if (lane == 0 && blockIdx.x == 0) { for (int i = 0; i < 10; i++) { res[i] = t[i]; res[16 + i] = c[i]; } }
    out[blockIdx.x * 64 + lane] = x + y + z + w + a + b + d + e + g;
}

int main(int argc, char** argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 1;
    const uint32_t words = 1u << 26;      // 256 MB
    std::vector<uint32_t> h(words);
    uint32_t s = 12345;
    for (uint32_t i = 0; i < words; i++) { s = s * 1664525u + 1013904223u; h[i] = (s >> 4) & (words - 1); }
    for (uint32_t i = 0; i < 16384; i++) h[i] = (h[i] * 7u + 1u) & 16383u;   // the small region chases inside itself... (values also used by the big chase: fine)
    uint32_t *g, *out; unsigned long long* res;
    hipMalloc(&g, (size_t)words * 4); hipMalloc(&out, (size_t)blocks * 64 * 4); hipMalloc(&res, 32 * 8);
    hipMemcpy(g, h.data(), (size_t)words * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, g, words - 1, out, res);
        hipDeviceSynchronize();
    }
    unsigned long long r[32];
    hipMemcpy(r, res, sizeof(r), hipMemcpyDeviceToHost);
    const char* name[] = {"LDS read (dependent)", "LDS atomic CAS rtn (dependent)", "global load, 64 KB footprint (dependent)", "global load, 256 MB footprint (dependent)",
                          "VALU mad (dependent)", "readfirstlane -> VALU (dependent)", "ballot + ffs + readlane (dependent)", "LDS write / wave sync / read neighbour", "8 independent global loads, 256 MB (per group of 8)"};
    const double ops[] = {N, N, N, N, N * 16.0, N * 4.0, N * 4.0, N, N / 8.0};
    printf("blocks %d (one wavefront each)\n", blocks);
    for (int i = 0; i < 9; i++) printf("%-55s %8.1f ns  %8.1f clock64 ticks per op\n", name[i], (r[i + 1] - r[i]) * 10.0 / ops[i], (double)(r[16 + i + 1] - r[16 + i]) / ops[i]);
    return 0;
}
