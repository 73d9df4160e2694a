// Microbenchmark: 16-byte LDS reads and writes at byte offsets that are not multiples of 16 (one 148-byte area per
// lane, 32 lanes live): are they right, and what do they cost the issuing wavefront against aligned ones?
// build: hipcc -O2 --offload-arch=gfx950 -o lds_unaligned lds_unaligned.hip ; run: ./lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define LAS __attribute__((address_space(3)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4u;
#define STRIDE 148

// correctness: lane l writes bytes (l * 7 + i) at offset wo(l) of its area, then reads 16 bytes at offset ro(l)
__global__ __launch_bounds__(64) void k_check(uint8_t *out, int wbase, int rbase)
{
    __shared__ __attribute__((aligned(16))) uint8_t area[64 * STRIDE + 64];
    uint32_t l = threadIdx.x;
    LAS uint8_t *a = (LAS uint8_t *)area + l * STRIDE;
    for (int i = 0; i < STRIDE; i++)
        a[i] = (uint8_t)(0x80 + i);
    __syncthreads();
    u32x4 v;
    uint8_t b[16];
    for (int i = 0; i < 16; i++)
        b[i] = (uint8_t)(l * 7 + i);
    memcpy(&v, b, 16);
    uint32_t wo = (wbase + l * 5) % 100, ro = (rbase + l * 3) % 100;
    *(LAS u32x4u *)(a + wo) = v;
    u32x4 r = *(LAS u32x4u *)(a + ro);
    memcpy(out + l * 16, &r, 16);
}

// cost: MODE 0 aligned write, 1 unaligned write, 2 aligned read, 3 unaligned read (dependent chain through the address)
template <int MODE>
__global__ __launch_bounds__(256) void k_cost(int iters, uint64_t *cyc, uint32_t *sink)
{
    __shared__ __attribute__((aligned(16))) uint8_t area[256 * STRIDE + 64];
    uint32_t l = threadIdx.x;
    LAS uint8_t *a = (LAS uint8_t *)area + l * STRIDE;
    u32x4 v = {l, l + 1, l + 2, l + 3};
    uint32_t off = (MODE & 1) ? 5 + (l & 3) : 16;
    uint32_t acc = 0;
    uint64_t t0 = 0, t1 = 0;
    if ((l & 63) < 32) {
        *(LAS u32x4u *)(a + off) = v;
        *(LAS u32x4u *)(a + off + 16) = v;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
            if (MODE < 2) {
                *(LAS u32x4u *)(a + off + (i & 1) * 16) = v;
                v.x += i;
            } else {
                u32x4 r = *(LAS u32x4u *)(a + off + (acc & 16));
                acc += r.x & 16;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t1 = __builtin_readcyclecounter();
    }
    if ((l & 63) == 0)
        cyc[blockIdx.x * 4 + (l >> 6)] = t1 - t0;
    if (acc == 0x12345678)
        sink[0] = acc + v.x;
}

int main()
{
    uint8_t *out;
    CK(hipMalloc(&out, 64 * 16));
    int bad = 0;
    for (int wb = 0; wb < 40; wb += 3)
        for (int rb = 0; rb < 40; rb += 7) {
            hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, out, wb, rb);
            uint8_t h[64 * 16];
            CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
            for (int l = 0; l < 64; l++) {
                uint8_t m[STRIDE + 32];
                for (int i = 0; i < STRIDE + 32; i++)
                    m[i] = (uint8_t)(0x80 + i);
                int wo = (wb + l * 5) % 100, ro = (rb + l * 3) % 100;
                for (int i = 0; i < 16; i++)
                    m[wo + i] = (uint8_t)(l * 7 + i);
                if (memcmp(m + ro, h + l * 16, 16))
                    bad++;
            }
        }
    printf("unaligned 16-byte LDS write then read, 64 lanes x 84 offset pairs: %d wrong\n", bad);
    uint64_t *cyc;
    uint32_t *sink;
    CK(hipMalloc(&cyc, 4096));
    CK(hipMalloc(&sink, 64));
    const int iters = 20000;
    const char *names[4] = {"aligned write", "unaligned write", "aligned read (dependent)", "unaligned read (dependent)"};
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            if (mode == 0) hipLaunchKernelGGL(k_cost<0>, dim3(1), dim3(256), 0, 0, iters, cyc, sink);
            if (mode == 1) hipLaunchKernelGGL(k_cost<1>, dim3(1), dim3(256), 0, 0, iters, cyc, sink);
            if (mode == 2) hipLaunchKernelGGL(k_cost<2>, dim3(1), dim3(256), 0, 0, iters, cyc, sink);
            if (mode == 3) hipLaunchKernelGGL(k_cost<3>, dim3(1), dim3(256), 0, 0, iters, cyc, sink);
            CK(hipDeviceSynchronize());
        }
        uint64_t h[4];
        CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
        printf("%-28s %6.1f cycles per operation (one wavefront per SIMD, 32 lanes)\n", names[mode], (double)h[0] / iters);
    }
    return bad != 0;
}
