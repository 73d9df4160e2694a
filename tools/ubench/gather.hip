// Microbenchmark: what does one dependent, fully divergent gather round trip cost with the batch
// kernel's footprint (one lane per stream, per-stream regions tens of KB apart, 1 wave per SIMD)?
// build: hipcc -O2 --offload-arch=gfx950 -o gather gather.hip ; run: ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256, 1) void k_chase(const uint16_t *tab, uint8_t *out, uint32_t *sink, size_t stride_tab,
                                                  size_t stride_out, uint32_t entries, int steps)
{
    size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint16_t *t = (const uint16_t *)((const uint8_t *)tab + lane * stride_tab);
    uint8_t *o = out + lane * stride_out;
    uint32_t idx = threadIdx.x, acc = 0;
    for (int k = 0; k < steps; k++) {
        uint32_t e = t[idx % entries];
        if (MODE == 2) acc += t[(idx * 7 + 13) % entries] + t[(idx * 11 + 5) % entries]; // two more independent loads
        idx = idx * 1664525u + e + 1013904223u + k;
        if (MODE >= 1) o[k & 0xffff] = (uint8_t)idx;  // one byte store per step, like a literal
    }
    sink[lane] = idx + acc;
}

int main(int argc, char **argv)
{
    int nlanes = argc > 1 ? atoi(argv[1]) : 65536;
    size_t stride_tab = (argc > 2 ? atoi(argv[2]) : 48) * 1024, stride_out = 65536;
    uint32_t entries = (argc > 3 ? atoi(argv[3]) : 5120);   // u16 entries actually touched per stream (10 KB)
    int steps = 4000;
    uint16_t *tab; uint8_t *out; uint32_t *sink;
    CK(hipMalloc(&tab, (size_t)nlanes * stride_tab));
    CK(hipMalloc(&out, (size_t)nlanes * stride_out));
    CK(hipMalloc(&sink, nlanes * 4));
    CK(hipMemset(tab, 1, (size_t)nlanes * stride_tab));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(a));
            if (mode == 0) k_chase<0><<<nlanes / 256, 256>>>(tab, out, sink, stride_tab, stride_out, entries, steps);
            if (mode == 1) k_chase<1><<<nlanes / 256, 256>>>(tab, out, sink, stride_tab, stride_out, entries, steps);
            if (mode == 2) k_chase<2><<<nlanes / 256, 256>>>(tab, out, sink, stride_tab, stride_out, entries, steps);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep) printf("lanes %d stride %zu KB entries %u mode %d (%s): %.1f ns per dependent step\n", nlanes, stride_tab >> 10, entries, mode,
                            mode == 0 ? "load" : mode == 1 ? "load+byte store" : "3 loads+byte store", ms * 1e6 / steps);
        }
    }
    return 0;
}
