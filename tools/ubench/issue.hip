// Microbenchmark: what a wave64 vector instruction costs on gfx950 by waves per SIMD and by the
// share of live lanes (the question behind half-filled wavefronts: DESIGN.md section 5).
// build: hipcc -O2 --offload-arch=gfx950 -o issue issue.hip ; run: ./issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// DEP: one dependent chain; else four independent chains
template <bool DEP>
__global__ __launch_bounds__(256) void k_issue(uint32_t *sink, int iters, int live, uint64_t *cycles)
{
    uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    uint64_t t0 = 0, t1 = 0;
    if ((int)(threadIdx.x & 63) < live) {
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
            if (DEP)
                asm volatile(REP64("v_add_u32 %0, %0, %0\n v_xor_b32 %0, %0, %1\n") : "+v"(a) : "v"(b));
            else
                asm volatile(REP64("v_add_u32 %0, %0, %0\n v_xor_b32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_xor_b32 %3, %3, %3\n")
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        }
        t1 = __builtin_readcyclecounter();
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
    if ((threadIdx.x & 63) == 0)
        cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// scalar instruction chain
__global__ __launch_bounds__(256) void k_salu(uint32_t *sink, int iters, uint64_t *cycles)
{
    uint32_t a = __builtin_amdgcn_readfirstlane(blockIdx.x);
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++)
        asm volatile(REP64("s_add_u32 %0, %0, %0\n s_xor_b32 %0, %0, 5\n") : "+s"(a));
    uint64_t t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if ((threadIdx.x & 63) == 0)
        cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// compare + select chains (v_cmp writes vcc, v_cndmask reads it): the staple of predicated code
__global__ __launch_bounds__(256) void k_sel(uint32_t *sink, int iters, uint64_t *cycles)
{
    uint32_t a = threadIdx.x, b = a * 7;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++)
        asm volatile(REP64("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %1, %0, vcc\n v_add_u32 %0, %0, 3\n") : "+v"(a) : "v"(b) : "vcc");
    uint64_t t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if ((threadIdx.x & 63) == 0)
        cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// LDS dependent read chain
__global__ __launch_bounds__(256) void k_lds(uint32_t *sink, int iters, uint64_t *cycles)
{
    __shared__ uint32_t tab[4096];
    for (int i = threadIdx.x; i < 4096; i += 256)
        tab[i] = (i * 37 + 11) & 4095;
    __syncthreads();
    uint32_t a = threadIdx.x;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters * 16; i++)
        a = tab[a];
    uint64_t t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if ((threadIdx.x & 63) == 0)
        cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

int main()
{
    uint32_t *sink; uint64_t *cyc;
    const int cus = 256;
    CK(hipMalloc(&sink, cus * 8 * 256 * 4));
    CK(hipMalloc(&cyc, cus * 8 * 4 * 8));
    uint64_t *h = (uint64_t *)malloc(cus * 8 * 4 * 8);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    auto report = [&](const char *what, int blocks, double instr, float ms) {
        CK(hipMemcpy(h, cyc, blocks * 4 * 8, hipMemcpyDeviceToHost));
        double s = 0; for (int i = 0; i < blocks * 4; i++) s += (double)h[i];
        s /= blocks * 4;
        printf("%-44s waves/SIMD %d: %.2f cycles per instruction per wave (wall %.3f ms, %.2f GHz-equivalent)\n", what, blocks / cus,
               s / instr, ms, s / (ms * 1e6));
    };
    for (int wps = 1; wps <= 8; wps *= 2) {        // blocks per CU == waves per SIMD (256-thread blocks)
        int blocks = cus * wps;
        int lives[3] = {64, 32, 16};
        for (int li = 0; li < 3; li++) {
            int live = lives[li];
            for (int dep = 0; dep < 2; dep++) {
                float ms = 0;
                for (int rep = 0; rep < 2; rep++) {
                    CK(hipEventRecord(e0));
                    if (dep) k_issue<true><<<blocks, 256>>>(sink, iters, live, cyc);
                    else k_issue<false><<<blocks, 256>>>(sink, iters, live, cyc);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                }
                char what[80];
                snprintf(what, sizeof what, "VALU %s, %d live lanes", dep ? "dependent" : "independent", live);
                report(what, blocks, (double)iters * 64 * (dep ? 2 : 4), ms);
            }
        }
        float ms;
        CK(hipEventRecord(e0)); k_salu<<<blocks, 256>>>(sink, iters, cyc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        report("SALU dependent", blocks, (double)iters * 64 * 2, ms);
        CK(hipEventRecord(e0)); k_sel<<<blocks, 256>>>(sink, iters, cyc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        report("v_cmp + v_cndmask + v_add dependent", blocks, (double)iters * 64 * 3, ms);
        CK(hipEventRecord(e0)); k_lds<<<blocks, 256>>>(sink, iters, cyc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        report("LDS dependent ds_read_b32 (per read)", blocks, (double)iters * 16, ms);
    }
    return 0;
}
