// Microbenchmark: what a 16-byte per-lane load or store costs the issuing wavefront on gfx950 when the lanes'
// addresses are scattered (one stream per lane) and not 16-byte aligned.
// build: hipcc -O2 --offload-arch=gfx950 -o unaligned unaligned.hip ; run: ./unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4u;
typedef uint64_t __attribute__((aligned(1))) u64u;

// MODE 0: 16-byte stores, 1: 16-byte loads, 2: 8-byte stores, 3: byte stores, 4: 2-byte loads
template <int MODE>
__global__ __launch_bounds__(256) void k(uint8_t *buf, size_t stride, int mis, int live, int iters, int span, uint64_t *cyc, uint32_t *sink)
{
    uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    uint8_t *p = buf + (size_t)lane * stride + mis;
    u32x4 v = {lane, lane + 1, lane + 2, lane + 3};
    uint32_t acc = 0;
    uint64_t t0 = 0, t1 = 0;
    if ((int)(threadIdx.x & 63) < live) {
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
            uint8_t *q = p + (size_t)(i & span) * 48;
            if (MODE == 0) *(u32x4u *)q = v;
            if (MODE == 1) { u32x4 r = __builtin_nontemporal_load((const u32x4u *)q); acc += r.x + r.w; }
            if (MODE == 2) *(u64u *)q = (uint64_t)lane;
            if (MODE == 3) *q = (uint8_t)i;
            if (MODE == 4) acc += *(const uint16_t __attribute__((aligned(1))) *)q;
            v.x += acc;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t1 = __builtin_readcyclecounter();
    }
    sink[lane] = acc + v.x;
    if ((threadIdx.x & 63) == 0)
        cyc[lane >> 6] = t1 - t0;
}

int main()
{
    const int cus = 256, iters = 2000;
    const size_t stride = 65536 + 192;
    uint8_t *buf; uint64_t *cyc; uint32_t *sink;
    CK(hipMalloc(&buf, (size_t)cus * 8 * 256 * stride / 4 + (1 << 20)));   // up to 2 blocks per CU at 32 live lanes... sized for 512 blocks
    CK(hipMalloc(&cyc, 4096 * 8)); CK(hipMalloc(&sink, 4096 * 64 * 4));
    uint64_t *h = (uint64_t *)malloc(4096 * 8);
    const char *names[5] = {"16-byte store", "16-byte load (nt)", "8-byte store", "byte store", "2-byte load"};
    for (int span = 3; span <= 255; span += 252)
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; blocks_per_cu++)
        for (int mode = 0; mode < 5; mode++)
            for (int mis = 0; mis <= 5; mis += (mis == 0 ? 1 : 4)) {   // 0, 1, 5 ; plus 4 below
                for (int pass = 0; pass < 2; pass++) {
                    int m = pass == 0 ? mis : 4;
                    if (pass == 1 && mis != 0) continue;
                    int blocks = cus * blocks_per_cu;
                    for (int rep = 0; rep < 2; rep++) {
                        if (mode == 0) k<0><<<blocks, 256>>>(buf, stride / 2, m, 32, iters, span, cyc, sink);
                        if (mode == 1) k<1><<<blocks, 256>>>(buf, stride / 2, m, 32, iters, span, cyc, sink);
                        if (mode == 2) k<2><<<blocks, 256>>>(buf, stride / 2, m, 32, iters, span, cyc, sink);
                        if (mode == 3) k<3><<<blocks, 256>>>(buf, stride / 2, m, 32, iters, span, cyc, sink);
                        if (mode == 4) k<4><<<blocks, 256>>>(buf, stride / 2, m, 32, iters, span, cyc, sink);
                        CK(hipDeviceSynchronize());
                    }
                    CK(hipMemcpy(h, cyc, blocks * 4 * 8, hipMemcpyDeviceToHost));
                    double s = 0; for (int i = 0; i < blocks * 4; i++) s += (double)h[i];
                    printf("%-18s footprint %5d B per lane, misaligned by %d, %d waves per CU (32 live lanes each): %.0f cycles per instruction\n", names[mode], (span + 1) * 48, m,
                           blocks_per_cu * 4, s / (blocks * 4) / iters);
                }
            }
    return 0;
}
