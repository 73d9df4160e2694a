// Microbenchmark: cycles a wavefront spends per scattered vector-memory instruction (every lane its own cache
// line, lines resident in L2/L1) by instruction kind, live lanes and wavefronts per CU.
// build: hipcc -O2 --offload-arch=gfx950 -o vmem_issue vmem_issue.hip ; run: ./vmem_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4u;

// MODE 0: byte store, 1: 16-byte store, 2: 2-byte load, 3: 16-byte load, 4: byte store + 16-byte store + 2-byte load + 16-byte load
template <int MODE>
__global__ __launch_bounds__(64) void k(uint8_t *buf, size_t stride, int live, int iters, uint64_t *cyc, uint32_t *sink)
{
    uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    uint8_t *p = buf + (size_t)lane * stride + 3;
    u32x4 v = {lane, lane + 1, lane + 2, lane + 3};
    uint32_t acc = 0;
    uint64_t t0 = 0, t1 = 0;
    if ((int)(threadIdx.x & 63) < live) {
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
            uint8_t *q = p + (size_t)(i & 3) * 48;
            if (MODE == 0 || MODE == 4) *q = (uint8_t)i;
            if (MODE == 1 || MODE == 4) *(u32x4u *)(q + 64) = v;
            if (MODE == 2 || MODE == 4) acc += *(const uint16_t __attribute__((aligned(1))) *)(q + 17);
            if (MODE == 3 || MODE == 4) { u32x4 r = *(const u32x4u *)(q + 33); acc += r.x + r.w; }
            if (MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
    const int cus = 256, iters = 4000;
    const size_t stride = 4096;
    uint8_t *buf; uint64_t *cyc; uint32_t *sink;
    CK(hipMalloc(&buf, (size_t)cus * 8 * 64 * stride + (1 << 20)));
    CK(hipMalloc(&cyc, 4096 * 8)); CK(hipMalloc(&sink, 4096 * 64 * 4));
    uint64_t *h = (uint64_t *)malloc(4096 * 8);
    const char *names[5] = {"byte store", "16-byte store", "2-byte load", "16-byte load", "the four, then a wait"};
    int wavesl[3] = {1, 4, 8};
    int lives[3] = {8, 32, 64};
    for (int wi = 0; wi < 3; wi++)
        for (int li = 0; li < 3; li++)
            for (int mode = 0; mode < 5; mode++) {
                int blocks = cus * wavesl[wi], live = lives[li];
                for (int rep = 0; rep < 2; rep++) {
                    if (mode == 0) k<0><<<blocks, 64>>>(buf, stride, live, iters, cyc, sink);
                    if (mode == 1) k<1><<<blocks, 64>>>(buf, stride, live, iters, cyc, sink);
                    if (mode == 2) k<2><<<blocks, 64>>>(buf, stride, live, iters, cyc, sink);
                    if (mode == 3) k<3><<<blocks, 64>>>(buf, stride, live, iters, cyc, sink);
                    if (mode == 4) k<4><<<blocks, 64>>>(buf, stride, live, iters, cyc, sink);
                    CK(hipDeviceSynchronize());
                }
                CK(hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost));
                double s = 0; for (int i = 0; i < blocks; i++) s += (double)h[i];
                printf("%-22s %d wavefronts per CU, %2d live lanes: %.0f cycles per iteration\n", names[mode], wavesl[wi], live, s / blocks / iters);
            }
    return 0;
}
