// Microbenchmark: what one load instruction costs the issuing wavefront (cycles until the next instruction), lines
// resident in L2: an ordinary 4-byte load into a register, the same load landing in LDS (global_load_lds_dword) with
// and without the exec/m0 set-up of the trip loop, 16-byte forms, 32 of 64 lanes live, one wavefront per SIMD.
// build: hipcc -O2 --offload-arch=gfx950 -o glds_issue glds_issue.hip ; run: ./glds_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define LAS __attribute__((address_space(3)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint8_t *buf, int iters, int per, uint64_t *cyc, uint32_t *sink)
{
    __shared__ __attribute__((aligned(16))) uint8_t land[4][1024];
    uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(LAS uint8_t *)land[w]);
    const uint8_t *p = buf + (size_t)(blockIdx.x * 256 + threadIdx.x) * 192 + 2;
    const uint8_t *sbase = (const uint8_t *)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)buf >> 32)) << 32) |
                                             (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)buf));
    uint32_t voff = (blockIdx.x * 256 + threadIdx.x) * 192 + 2;
    uint32_t acc = 0;
    uint64_t t0 = 0, t1 = 0;
    if (lane < 32) {
        uint64_t mask = __ballot(1), keep;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int i = 0; i < iters; i++) {
            for (int j = 0; j < per; j++) {
                if (MODE == 0) { uint32_t v; asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory"); acc += 0; }
                if (MODE == 1) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(p), "s"(lds) : "memory");
                if (MODE == 2) asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %3\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b64 exec, %0"
                                            : "=&s"(keep) : "v"(p), "s"(lds), "s"(mask) : "memory");
                if (MODE == 3) asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b64 exec, %0"
                                            : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds), "s"(mask) : "memory");
                if (MODE == 4) { u32x4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); }
                if (MODE == 5) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(p), "s"(lds) : "memory");
                if (MODE == 6) asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %3\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b64 exec, %0"
                                            : "=&s"(keep) : "v"(p), "s"(lds), "s"(0ull) : "memory");
                for (int a = 0; a < 20; a++) /* twenty instructions between loads, as in the loop */
                    asm volatile("v_add_u32 %0, %0, 1" : "+v"(acc));
            }
            if ((i & 15) == 15)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
    if (lane == 0)
        cyc[blockIdx.x * 4 + w] = t1 - t0;
}

int main()
{
    const int blocks = 256, iters = 2000, per = 3;
    uint8_t *buf; uint64_t *cyc; uint32_t *sink;
    CK(hipMalloc(&buf, (size_t)blocks * 256 * 192 + 4096)); CK(hipMemset(buf, 1, (size_t)blocks * 256 * 192 + 4096));
    CK(hipMalloc(&cyc, blocks * 4 * 8)); CK(hipMalloc(&sink, blocks * 256 * 4));
    uint64_t h[1024];
    const char *names[7] = {"4-byte load into a register", "4-byte load into LDS (m0 set)", "the same with exec saved, set, restored",
                            "the same, base in scalar registers", "16-byte load into registers", "16-byte load into LDS (m0 set)",
                            "16-byte load into LDS, no lane enabled"};
    for (int mode = 0; mode < 7; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            if (mode == 0) k<0><<<blocks, 256>>>(buf, iters, per, cyc, sink);
            if (mode == 1) k<1><<<blocks, 256>>>(buf, iters, per, cyc, sink);
            if (mode == 2) k<2><<<blocks, 256>>>(buf, iters, per, cyc, sink);
            if (mode == 3) k<3><<<blocks, 256>>>(buf, iters, per, cyc, sink);
            if (mode == 4) k<4><<<blocks, 256>>>(buf, iters, per, cyc, sink);
            if (mode == 5) k<5><<<blocks, 256>>>(buf, iters, per, cyc, sink);
            if (mode == 6) k<6><<<blocks, 256>>>(buf, iters, per, cyc, sink);
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h, cyc, blocks * 4 * 8, hipMemcpyDeviceToHost));
        double s = 0;
        for (int i = 0; i < blocks * 4; i++) s += (double)h[i];
        printf("%-46s %7.0f cycles per group of three loads and 60 adds (a wait every 16 groups)\n", names[mode], s / (blocks * 4) / iters);
    }
    return 0;
}
