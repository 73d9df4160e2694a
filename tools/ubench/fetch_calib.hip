// Calibration of rocprofv3's FETCH_SIZE for this library's access patterns: every 64-byte line of a
// 1 GiB buffer is touched exactly once, so the bytes HBM must deliver are known, by
//   mode 0: one scattered 2-byte load per line (a table lookup), lines in a scrambled order
//   mode 1: one scattered 16-byte load per line at a 4-byte-aligned offset inside it (a copy's piece)
//   mode 2: a wide streaming read, 16 bytes per lane, consecutive lanes consecutive addresses (the guide's case)
// build: hipcc -O2 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip ; run under rocprofv3 --pmc FETCH_SIZE (or
// TCC_MISS_sum) -- ./fetch_calib <mode>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LINES (1u << 24) /* 1 GiB of 64-byte lines */

__global__ __launch_bounds__(256) void k_lines(const uint8_t *buf, uint32_t *sink, int mode)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    if (mode == 2) {
        for (uint32_t k = 0; k < 4; k++) { /* 64 bytes per thread in all, 16 at a time, coalesced */
            u32x4 v = *(const u32x4 *)(buf + ((size_t)k * LINES + i) * 16);
            acc += v.x + v.y + v.z + v.w;
        }
    } else {
        uint32_t line = (i * 2654435761u + 12345u) & (LINES - 1); /* odd multiplier: a bijection of the 2^24 lines */
        const uint8_t *p = buf + (size_t)line * 64;
        if (mode == 0)
            acc = *(const uint16_t *)(p + ((i >> 3) & 62));
        else {
            typedef u32x4 __attribute__((aligned(4))) u32x4a;
            u32x4 v = *(const u32x4a *)(p + ((i >> 5) & 44)); /* offsets 0..44, multiples of 4: inside the line */
            acc = v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 0x12345678u)
        sink[0] = acc;
}

int main(int argc, char **argv)
{
    int mode = argc > 1 ? atoi(argv[1]) : 0;
    uint8_t *buf; uint32_t *sink;
    CK(hipMalloc(&buf, (size_t)LINES * 64));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, (size_t)LINES * 64));
    // push the buffer out of L2 and the Infinity Cache: write 1 GiB elsewhere
    uint8_t *other; CK(hipMalloc(&other, (size_t)1 << 30)); CK(hipMemset(other, 2, (size_t)1 << 30)); CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    k_lines<<<LINES / 256, 256>>>(buf, sink, mode);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("mode %d: %u lines of 64 bytes touched once = %.0f bytes; %.3f ms = %.1f G lines/s\n", mode, LINES, (double)LINES * 64, ms,
           LINES / (ms * 1e6));
    return 0;
}
