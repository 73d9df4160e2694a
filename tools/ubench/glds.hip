// Microbenchmark / probe: loads that land in LDS (global_load_lds_*) with per-lane source addresses, as the trip
// loop of the batch kernel uses them: (1) a 16-byte load from an address of any alignment, (2) a 4-byte load,
// (3) the same instruction issued with no lane enabled still passes through the memory counter, (4) a counted wait
// (vmcnt(2)) behind three loads guarantees the first, (5) cycles per trip of a dependent chain "load -> 150
// instructions" when the wait for the load sits before or behind the instructions.
// build: hipcc -O2 --offload-arch=gfx950 -o glds glds.hip ; run: ./glds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LAS __attribute__((address_space(3)))

// one lane-linear slot per wave: lane i's bytes land at base + i * size
__device__ __forceinline__ void glds16(const void *src, uint32_t lds_base, uint64_t mask)
{
    uint64_t keep;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %3\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b64 exec, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_base), "s"(mask) : "memory");
}
__device__ __forceinline__ void glds4(const void *src, uint32_t lds_base, uint64_t mask)
{
    uint64_t keep;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %3\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dword %1, off\n\ts_mov_b64 exec, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_base), "s"(mask) : "memory");
}

__global__ __launch_bounds__(64) void probe(const uint8_t *buf, uint32_t *out, int phase)
{
    __shared__ __attribute__((aligned(16))) uint8_t slot16[1024];
    __shared__ uint32_t slot4[64];
    __shared__ __attribute__((aligned(16))) uint8_t slotb[1024];
    uint32_t lane = threadIdx.x;
    uint32_t b16 = (uint32_t)(uintptr_t)(LAS uint8_t *)slot16, b4 = (uint32_t)(uintptr_t)(LAS uint32_t *)slot4,
             bb = (uint32_t)(uintptr_t)(LAS uint8_t *)slotb;
    b16 = __builtin_amdgcn_readfirstlane(b16); b4 = __builtin_amdgcn_readfirstlane(b4); bb = __builtin_amdgcn_readfirstlane(bb);
    for (int i = lane; i < 256; i += 64) { ((uint32_t *)slot16)[i] = 0xdeadbeef; ((uint32_t *)slotb)[i] = 0xdeadbeef; }
    slot4[lane] = 0xdeadbeef;
    __syncthreads();
    const uint8_t *p16 = buf + lane * 37 + 1;          // any alignment
    const uint8_t *p4 = buf + 4096 + lane * 6;         // 2-byte aligned, half of them not 4-byte aligned
    uint64_t all = __ballot(1);
    if (phase == 0) {
        glds16(p16, b16, all);
        glds4(p4, b4, all);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (phase == 1) { // odd lanes only, then an instruction with no lane enabled
        glds16(p16, b16, 0xaaaaaaaaaaaaaaaaull);
        glds16(p16, bb, 0);
        glds4(p4, b4, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else { // counted wait: the first of three loads is there at vmcnt(2)
        glds4(p4, b4, all);
        glds16(p16 + (1 << 20), b16, all);
        glds16(p16 + (2 << 20), bb, all);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        out[64 * 6 + lane] = slot4[lane];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    u32x4 v = *(u32x4 *)(slot16 + lane * 16);
    out[lane * 4 + 0] = v.x; out[lane * 4 + 1] = v.y; out[lane * 4 + 2] = v.z; out[lane * 4 + 3] = v.w;
    out[256 + lane] = slot4[lane];
    out[320 + lane] = ((uint32_t *)slotb)[lane * 4];
}

// chain: each trip loads a 4-byte value whose address depends on the last one (a random walk over a table), and does
// `work` dependent ALU instructions that do NOT depend on the load.  early: load -> wait -> work;  late: load -> work -> wait
template <int LATE>
__global__ __launch_bounds__(256) void chain(const uint32_t *tab, uint32_t mask, int trips, int work, uint64_t *cyc, uint32_t *sink)
{
    __shared__ uint32_t slot4[4][64];
    uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t b4 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(LAS uint32_t *)slot4[w]);
    uint32_t idx = (blockIdx.x * 256 + threadIdx.x) * 2654435761u & mask, acc = lane;
    uint64_t all = __ballot(1);
    uint64_t t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int t = 0; t < trips; t++) {
        glds4(tab + idx, b4, all);
        if (!LATE)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int i = 0; i < work; i++)
            acc = acc * 5 + 1;
        asm volatile("" : "+v"(acc));
        if (LATE)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        idx = (slot4[w][lane] + acc * 0) & mask;
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    sink[blockIdx.x * 256 + threadIdx.x] = acc + idx;
    if (lane == 0)
        cyc[blockIdx.x * 4 + w] = t1 - t0;
}

int main()
{
    uint8_t *buf; uint32_t *out;
    const size_t nb = 4u << 20;
    CK(hipMalloc(&buf, nb)); CK(hipMalloc(&out, 4096));
    uint8_t *h = (uint8_t *)malloc(nb);
    for (size_t i = 0; i < nb; i++) h[i] = (uint8_t)(i * 131 + (i >> 8) * 7 + (i >> 16));
    CK(hipMemcpy(buf, h, nb, hipMemcpyHostToDevice));
    uint32_t ho[1024];
    for (int phase = 0; phase < 3; phase++) {
        CK(hipMemset(out, 0, 4096));
        probe<<<1, 64>>>(buf, out, phase);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ho, out, 4096, hipMemcpyDeviceToHost));
        int bad16 = 0, bad4 = 0, badn = 0, keep16 = 0, early = 0;
        for (int l = 0; l < 64; l++) {
            const uint8_t *e16 = h + l * 37 + 1 + (phase == 2 ? (1 << 20) : 0), *e4 = h + 4096 + l * 6;
            uint32_t w4; memcpy(&w4, e4, 4);
            bool on = phase != 1 || (l & 1);
            if (on ? memcmp(&ho[l * 4], e16, 16) != 0 : ho[l * 4] != 0xdeadbeef) bad16++;
            if (!on && ho[l * 4] == 0xdeadbeef) keep16++;
            if (phase != 1 ? ho[256 + l] != w4 : ho[256 + l] != 0xdeadbeef) bad4++;
            if (phase == 1 && ho[320 + l] != 0xdeadbeef) badn++;
            if (phase == 2 && ho[384 + l] != w4) early++;
        }
        printf("phase %d: 16-byte lanes wrong %d (masked lanes untouched %d), 4-byte lanes wrong %d, no-lane load wrote %d, not there at vmcnt(2): %d\n",
               phase, bad16, keep16, bad4, badn, early);
    }
    // chain timing
    uint32_t *tab; uint64_t *cyc; uint32_t *sink;
    const uint32_t entries = 1u << 26; // 256 MiB
    CK(hipMalloc(&tab, (size_t)entries * 4)); CK(hipMalloc(&cyc, 8192 * 8)); CK(hipMalloc(&sink, 2048 * 256 * 4));
    uint32_t *ht = (uint32_t *)malloc((size_t)entries * 4);
    uint32_t x = 12345;
    for (uint32_t i = 0; i < entries; i++) { x = x * 1664525u + 1013904223u; ht[i] = x >> 4; }
    CK(hipMemcpy(tab, ht, (size_t)entries * 4, hipMemcpyHostToDevice));
    uint64_t hc[8192];
    for (int blocks : {256, 2048})
        for (int work : {0, 100, 200, 400})
            for (int late = 0; late < 2; late++) {
                const int trips = 2000;
                for (int rep = 0; rep < 2; rep++) {
                    if (late) chain<1><<<blocks, 256>>>(tab, entries - 1, trips, work, cyc, sink);
                    else chain<0><<<blocks, 256>>>(tab, entries - 1, trips, work, cyc, sink);
                    CK(hipGetLastError());
                    CK(hipDeviceSynchronize());
                }
                CK(hipMemcpy(hc, cyc, blocks * 4 * 8, hipMemcpyDeviceToHost));
                double s = 0;
                for (int i = 0; i < blocks * 4; i++) s += (double)hc[i];
                printf("chain: %4d blocks x 4 waves, %3d instructions per trip, wait %s them: %.0f cycles per trip\n", blocks, work,
                       late ? "behind" : "before", s / (blocks * 4) / trips);
            }
    return 0;
}
