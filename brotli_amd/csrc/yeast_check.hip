/*
 * yeast_check.hip -- K4: output check values on MI355X (gfx950).
 *
 * Replaces, for outputs resident in HBM, the reference's check functions
 * (paths relative to /root/reference): crc32c() crc32c.c:258 (with the zeros
 * operator of crc32c.c:82-134 used to combine per-lane slices), XXH32()
 * xxhash.c:288-391 and XXH64() xxhash.c:393-531; call sites brand.c:245-248 and
 * broad.c:278-281 (seed / initial value 0).
 */
#include <hip/hip_runtime.h>
#include "yeast_common.h"

#define DEV __device__ __forceinline__

__constant__ u32 c_x2n[68]; /* x^(2^k) mod P for the CRC-32C combine */

#define P32_1 2654435761u
#define P32_2 2246822519u
#define P32_3 3266489917u
#define P32_4 668265263u
#define P32_5 374761393u
#define P64_1 11400714785074694791ull
#define P64_2 14029467366897019727ull
#define P64_3 1609587929392839161ull
#define P64_4 9650029242287828579ull
#define P64_5 2870177450012600261ull

DEV u32 rol32(u32 x, int r) { return (x << r) | (x >> (32 - r)); }
DEV u64 rol64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
DEV u32 ld32(const u8 *p) { u32 v; __builtin_memcpy(&v, p, 4); return v; }
DEV u64 ld64(const u8 *p) { u64 v; __builtin_memcpy(&v, p, 8); return v; }
DEV u64 xxh64_round(u64 acc, u64 in) { return rol64(acc + in * P64_2, 31) * P64_1; }
DEV u64 xxh64_merge(u64 h, u64 v) { return (h ^ xxh64_round(0, v)) * P64_1 + P64_4; }

/* lanes 0..3 of the wave each own one accumulator; lane 0 finishes */
DEV u64 wave_xxh64(const u8 *p, u64 len, u64 seed, u32 lane)
{
    u64 h = 0;
    u64 nstripe = len >> 5;
    u64 acc = lane == 0 ? seed + P64_1 + P64_2 : lane == 1 ? seed + P64_2 : lane == 2 ? seed : seed - P64_1;
    if (lane < 4)
        for (u64 i = 0; i < nstripe; i++)
            acc = xxh64_round(acc, ld64(p + (i << 5) + (lane << 3)));
    u64 v1 = __shfl(acc, 0), v2 = __shfl(acc, 1), v3 = __shfl(acc, 2), v4 = __shfl(acc, 3);
    if (len >= 32) {
        h = rol64(v1, 1) + rol64(v2, 7) + rol64(v3, 12) + rol64(v4, 18);
        h = xxh64_merge(h, v1);
        h = xxh64_merge(h, v2);
        h = xxh64_merge(h, v3);
        h = xxh64_merge(h, v4);
    } else
        h = seed + P64_5;
    h += len;
    const u8 *q = p + (nstripe << 5), *end = p + len;
    for (; q + 8 <= end; q += 8)
        h = rol64(h ^ xxh64_round(0, ld64(q)), 27) * P64_1 + P64_4;
    if (q + 4 <= end) {
        h = rol64(h ^ (u64)ld32(q) * P64_1, 23) * P64_2 + P64_3;
        q += 4;
    }
    for (; q < end; q++)
        h = rol64(h ^ (u64)*q * P64_5, 11) * P64_1;
    h ^= h >> 33; h *= P64_2; h ^= h >> 29; h *= P64_3; h ^= h >> 32;
    return h;
}

DEV u32 wave_xxh32(const u8 *p, u64 len, u32 seed, u32 lane)
{
    u32 h;
    u64 nstripe = len >> 4;
    u32 acc = lane == 0 ? seed + P32_1 + P32_2 : lane == 1 ? seed + P32_2 : lane == 2 ? seed : seed - P32_1;
    if (lane < 4)
        for (u64 i = 0; i < nstripe; i++)
            acc = rol32(acc + ld32(p + (i << 4) + (lane << 2)) * P32_2, 13) * P32_1;
    u32 v1 = __shfl(acc, 0), v2 = __shfl(acc, 1), v3 = __shfl(acc, 2), v4 = __shfl(acc, 3);
    if (len >= 16)
        h = rol32(v1, 1) + rol32(v2, 7) + rol32(v3, 12) + rol32(v4, 18);
    else
        h = seed + P32_5;
    h += (u32)len;
    const u8 *q = p + (nstripe << 4), *end = p + len;
    for (; q + 4 <= end; q += 4)
        h = rol32(h + ld32(q) * P32_3, 17) * P32_4;
    for (; q < end; q++)
        h = rol32(h + (u32)*q * P32_5, 11) * P32_1;
    h ^= h >> 15; h *= P32_2; h ^= h >> 13; h *= P32_3; h ^= h >> 16;
    return h;
}

#define CRC_POLY 0x82f63b78u
/* a(x) * b(x) mod P in the reflected representation (x^0 at bit 31) */
DEV u32 crc_mul(u32 a, u32 b)
{
    u32 p = 0;
    for (int i = 0; i < 32; i++) {
        p ^= (a & 0x80000000u) ? b : 0;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1) ? CRC_POLY : 0);
    }
    return p;
}
/* x^(8n) mod P */
DEV u32 crc_x8n(u64 n)
{
    u32 p = 0x80000000u;
    u32 k = 3;
    while (n) {
        if (n & 1)
            p = crc_mul(c_x2n[k], p);
        n >>= 1;
        k++;
    }
    return p;
}

/* each lane takes a contiguous slice, then the slices are combined with the
   zeros operator (crc32c_zeros_op/crc32c_shift, crc32c.c:82-134) in its
   polynomial form */
DEV u32 wave_crc32c(const u32 *tab /*LDS, 256*/, const u8 *p, u64 len, u32 init, u32 lane)
{
    u64 per = (len + 63) >> 6;
    u64 a = per * lane, e = a + per;
    if (a > len) a = len;
    if (e > len) e = len;
    u32 crc = lane == 0 ? ~init : 0xffffffffu;
    for (u64 i = a; i < e; i++)
        crc = tab[(crc ^ p[i]) & 255] ^ (crc >> 8);
    crc = ~crc;
    /* shift by the bytes that follow this slice and fold */
    u32 part = crc_mul(crc_x8n(len - e), crc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        part ^= __shfl_xor(part, o);
    return part;
}

/* kind: 1 = CRC-32C, 2 = XXH32, 3 = XXH64 (seed / initial value 0 at every
   reference call site: brand.c:245-248, broad.c:278-281) */
__global__ __launch_bounds__(256) void k_check(const u8 *dst, const u64 *dst_off, ys_record *rec, u32 n,
                                               u32 kind, u64 seed, int only_ok)
{
    __shared__ u32 s_tab[256];
    for (u32 i = threadIdx.x; i < 256; i += blockDim.x) {
        u32 c = i;
        for (int k = 0; k < 8; k++)
            c = (c & 1) ? (c >> 1) ^ CRC_POLY : c >> 1;
        s_tab[i] = c;
    }
    __syncthreads();
    u32 lane = threadIdx.x & 63;
    u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    u32 nwave = (gridDim.x * blockDim.x) >> 6;
    for (u32 j = wave; j < n; j += nwave) {
        ys_record r = rec[j];
        if (only_ok && r.rc != YS_OK && r.rc != YS_MISMATCH && r.rc != YS_EOF && r.rc != YS_INVALID)
            continue;
        const u8 *p = dst + dst_off[j];
        u64 v;
        if (kind == 1)
            v = wave_crc32c(s_tab, p, r.got, (u32)seed, lane);
        else if (kind == 2)
            v = wave_xxh32(p, r.got, (u32)seed, lane);
        else
            v = wave_xxh64(p, r.got, seed, lane);
        if (lane == 0)
            rec[j].check = v;
    }
}

/* plain buffers: one wave per (ptr,len) pair */
__global__ __launch_bounds__(64) void k_check_buf(const u8 *p, u64 len, u32 kind, u64 seed, u64 *out)
{
    __shared__ u32 s_tab[256];
    for (u32 i = threadIdx.x; i < 256; i += blockDim.x) {
        u32 c = i;
        for (int k = 0; k < 8; k++)
            c = (c & 1) ? (c >> 1) ^ CRC_POLY : c >> 1;
        s_tab[i] = c;
    }
    __syncthreads();
    u32 lane = threadIdx.x & 63;
    u64 v;
    if (kind == 1)
        v = wave_crc32c(s_tab, p, len, (u32)seed, lane);
    else if (kind == 2)
        v = wave_xxh32(p, len, (u32)seed, lane);
    else
        v = wave_xxh64(p, len, seed, lane);
    if (lane == 0)
        *out = v;
}

/* ---------------------------------------------------------------- launchers */

extern "C" int ys_upload_check_constants(const u32 *x2n)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(c_x2n), x2n, 68 * sizeof(u32));
}

extern "C" int ys_launch_check(const u8 *dst, const u64 *dst_off, ys_record *rec, unsigned n, unsigned kind,
                               u64 seed, void *stream)
{
    if (n == 0) return 0;
    u32 waves = n < 256 * 16 ? n : 256 * 16;
    u32 blocks = (waves + 3) / 4;
    hipLaunchKernelGGL(k_check, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dst, dst_off, rec, n, kind,
                       seed, 1);
    return (int)hipGetLastError();
}

extern "C" int ys_launch_check_buf(const u8 *p, u64 len, unsigned kind, u64 seed, u64 *out, void *stream)
{
    hipLaunchKernelGGL(k_check_buf, dim3(1), dim3(64), 0, (hipStream_t)stream, p, len, kind, seed, out);
    return (int)hipGetLastError();
}
