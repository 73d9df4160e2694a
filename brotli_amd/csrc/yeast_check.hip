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
#define GAS __attribute__((address_space(1)))
typedef u32 __attribute__((aligned(1))) u32u;
typedef u64 __attribute__((aligned(1))) u64u;
#include "yeast_bounds.h"
DEV u32 ld32(const u8 *p) { return YSB(p, 4) ? *(const GAS u32u *)p : 0; }
DEV u64 ld64(const u8 *p) { return YSB(p, 8) ? *(const GAS u64u *)p : 0; }
DEV u32 ld8(const u8 *p) { return YSB(p, 1) ? *(const GAS u8 *)p : 0; }
DEV u64 xxh64_round(u64 acc, u64 in) { return rol64(acc + in * P64_2, 31) * P64_1; }
DEV u64 xxh64_merge(u64 h, u64 v) { return (h ^ xxh64_round(0, v)) * P64_1 + P64_4; }

/* Four lanes per buffer (the hash has four independent accumulators and no more): lane `sub` of
   the quad owns accumulator sub; every lane of the quad finishes the hash redundantly.  The
   stripe loop keeps eight loads in flight per lane. */
DEV u64 quad_xxh64(const u8 *p, u64 len, u64 seed, u32 sub)
{
    u64 h = 0;
    u64 nstripe = len >> 5;
    u64 acc = sub == 0 ? seed + P64_1 + P64_2 : sub == 1 ? seed + P64_2 : sub == 2 ? seed : seed - P64_1;
    const u8 *q = p + (sub << 3);
    u64 i = 0;
    for (; i + 8 <= nstripe; i += 8, q += 256) {
        u64 x[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            x[k] = ld64(q + (k << 5));
#pragma unroll
        for (int k = 0; k < 8; k++)
            acc = xxh64_round(acc, x[k]);
    }
    for (; i < nstripe; i++, q += 32)
        acc = xxh64_round(acc, ld64(q));
    u64 v1 = __shfl(acc, 0, 4), v2 = __shfl(acc, 1, 4), v3 = __shfl(acc, 2, 4), v4 = __shfl(acc, 3, 4);
    if (len >= 32) {
        h = rol64(v1, 1) + rol64(v2, 7) + rol64(v3, 12) + rol64(v4, 18);
        h = xxh64_merge(h, v1);
        h = xxh64_merge(h, v2);
        h = xxh64_merge(h, v3);
        h = xxh64_merge(h, v4);
    } else
        h = seed + P64_5;
    h += len;
    const u8 *end = p + len;
    q = p + (nstripe << 5);
    for (; q + 8 <= end; q += 8)
        h = rol64(h ^ xxh64_round(0, ld64(q)), 27) * P64_1 + P64_4;
    if (q + 4 <= end) {
        h = rol64(h ^ (u64)ld32(q) * P64_1, 23) * P64_2 + P64_3;
        q += 4;
    }
    for (; q < end; q++)
        h = rol64(h ^ (u64)ld8(q) * P64_5, 11) * P64_1;
    h ^= h >> 33; h *= P64_2; h ^= h >> 29; h *= P64_3; h ^= h >> 32;
    return h;
}

DEV u32 xxh32_round(u32 acc, u32 in) { return rol32(acc + in * P32_2, 13) * P32_1; }

DEV u32 quad_xxh32(const u8 *p, u64 len, u32 seed, u32 sub)
{
    u32 h;
    u64 nstripe = len >> 4;
    u32 acc = sub == 0 ? seed + P32_1 + P32_2 : sub == 1 ? seed + P32_2 : sub == 2 ? seed : seed - P32_1;
    const u8 *q = p + (sub << 2);
    u64 i = 0;
    for (; i + 8 <= nstripe; i += 8, q += 128) {
        u32 x[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            x[k] = ld32(q + (k << 4));
#pragma unroll
        for (int k = 0; k < 8; k++)
            acc = xxh32_round(acc, x[k]);
    }
    for (; i < nstripe; i++, q += 16)
        acc = xxh32_round(acc, ld32(q));
    u32 v1 = __shfl(acc, 0, 4), v2 = __shfl(acc, 1, 4), v3 = __shfl(acc, 2, 4), v4 = __shfl(acc, 3, 4);
    if (len >= 16)
        h = rol32(v1, 1) + rol32(v2, 7) + rol32(v3, 12) + rol32(v4, 18);
    else
        h = seed + P32_5;
    h += (u32)len;
    const u8 *end = p + len;
    q = p + (nstripe << 4);
    for (; q + 4 <= end; q += 4)
        h = rol32(h + ld32(q) * P32_3, 17) * P32_4;
    for (; q < end; q++)
        h = rol32(h + ld8(q) * P32_5, 11) * P32_1;
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

/* four bytes of the message at once: slicing-by-4 over the tables tab[k][b] = CRC of byte b
   followed by k zero bytes (tab[0] is the byte table of crc32c.c:25-42) */
DEV u32 crc_word(const u32 *tab, u32 crc, u32 w)
{
    crc ^= w;
    return tab[768 + (crc & 255)] ^ tab[512 + ((crc >> 8) & 255)] ^ tab[256 + ((crc >> 16) & 255)] ^ tab[crc >> 24];
}

/* each lane takes a contiguous slice (a multiple of 16 bytes, read 16 bytes at a time), then the
   slices are combined with the zeros operator (crc32c_zeros_op/crc32c_shift, crc32c.c:82-134) in
   its polynomial form */
template <u32 SLICES>
DEV u32 slice_crc32c(const u32 *tab /*LDS, 4 x 256*/, const u8 *p, u64 len, u32 init, u32 slice)
{
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    typedef u32x4 __attribute__((aligned(1))) u32x4u;
    u64 per = (((len + SLICES - 1) / SLICES) + 15) & ~15ull;
    u64 a = per * slice, e = a + per;
    if (a > len) a = len;
    if (e > len) e = len;
    u32 crc = slice == 0 ? ~init : 0xffffffffu;
    u64 i = a;
    for (; i + 32 <= e; i += 32) {
        u32x4 x = YSB(p + i, 32) ? *(const GAS u32x4u *)(p + i) : (u32x4)0, y = YSB(p + i, 32) ? *(const GAS u32x4u *)(p + i + 16) : (u32x4)0;
        crc = crc_word(tab, crc, x.x);
        crc = crc_word(tab, crc, x.y);
        crc = crc_word(tab, crc, x.z);
        crc = crc_word(tab, crc, x.w);
        crc = crc_word(tab, crc, y.x);
        crc = crc_word(tab, crc, y.y);
        crc = crc_word(tab, crc, y.z);
        crc = crc_word(tab, crc, y.w);
    }
    for (; i < e; i++)
        crc = tab[(crc ^ ld8(p + i)) & 255] ^ (crc >> 8);
    crc = ~crc;
    /* shift by the bytes that follow this slice; the CRC of the whole is the XOR of these over the slices */
    return a == e && slice ? 0 : crc_mul(crc_x8n(len - e), crc);
}

DEV u32 wave_xor(u32 part)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        part ^= __shfl_xor(part, o);
    return part;
}

DEV u32 wave_crc32c(const u32 *tab, const u8 *p, u64 len, u32 init, u32 lane)
{
    return wave_xor(slice_crc32c<64>(tab, p, len, init, lane));
}

/* CRC-32C of one buffer by a whole block of 1,024 lanes: as many slices, folded through LDS (a single wavefront
   reads 3.7 GB/s: 4.6 ms for the 18 MB output of BASELINE config 5) */
DEV u32 block_crc32c(const u32 *tab, u32 *fold /*LDS, 16*/, const u8 *p, u64 len, u32 init)
{
    u32 part = wave_xor(slice_crc32c<1024>(tab, p, len, init, threadIdx.x));
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        fold[threadIdx.x >> 6] = part;
    __syncthreads();
    u32 v = 0;
#pragma unroll
    for (int k = 0; k < 16; k++)
        v ^= fold[k];
    return v;
}

DEV void crc_tables(u32 *tab /*LDS, 4 x 256*/)
{
    for (u32 i = threadIdx.x; i < 256; i += blockDim.x) {
        u32 c = i;
        for (int k = 0; k < 8; k++)
            c = (c & 1) ? (c >> 1) ^ CRC_POLY : c >> 1;
        tab[i] = c;
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < 256; i += blockDim.x) {
        u32 c = tab[i];
        for (int k = 1; k < 4; k++) {
            c = tab[c & 255] ^ (c >> 8);
            tab[k * 256 + i] = c;
        }
    }
    __syncthreads();
}

/* kind: 1 = CRC-32C, 2 = XXH32, 3 = XXH64 (seed / initial value 0 at every
   reference call site: brand.c:245-248, broad.c:278-281) */
__global__ __launch_bounds__(256) void k_check(const u8 *dst, const u64 *dst_off, ys_record *rec, u32 n,
                                               u32 kind, u64 seed, int only_ok)
{
    __shared__ u32 s_tab[1024];
    crc_tables(s_tab);
    /* CRC-32C: one wavefront per stream (64 slices); the xxHashes: four lanes per stream */
    const u32 gsz = kind == 1 ? 64 : 4;
    u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    u32 sub = tid & (gsz - 1);
    u32 ngroup = (gridDim.x * blockDim.x) / gsz;
    for (u32 j = tid / gsz; j < n; j += ngroup) {
        ys_record r = rec[j];
        if (only_ok && r.rc != YS_OK && r.rc != YS_MISMATCH && r.rc != YS_EOF && r.rc != YS_INVALID)
            continue;
        const u8 *p = dst + dst_off[j];
        u64 v;
        if (kind == 1)
            v = wave_crc32c(s_tab, p, r.got, (u32)seed, sub);
        else if (kind == 2)
            v = quad_xxh32(p, r.got, (u32)seed, sub);
        else
            v = quad_xxh64(p, r.got, seed, sub);
        if (sub == 0)
            rec[j].check = v;
    }
}

/* CRC-32C over few streams: one block of sixteen wavefronts per stream */
__global__ __launch_bounds__(1024) void k_crc_wide(const u8 *dst, const u64 *dst_off, ys_record *rec, u32 n, u32 init,
                                                   int only_ok)
{
    __shared__ u32 s_tab[1024];
    __shared__ u32 s_fold[16];
    crc_tables(s_tab);
    for (u32 j = blockIdx.x; j < n; j += gridDim.x) {
        ys_record r = rec[j];
        if (only_ok && r.rc != YS_OK && r.rc != YS_MISMATCH && r.rc != YS_EOF && r.rc != YS_INVALID)
            continue; /* the same for the whole block */
        u32 v = block_crc32c(s_tab, s_fold, dst + dst_off[j], r.got, init);
        if (threadIdx.x == 0)
            rec[j].check = v;
    }
}

/* plain buffers: one wave per (ptr,len) pair (CRC-32C of a long one: sixteen) */
__global__ __launch_bounds__(1024) void k_check_buf(const u8 *p, u64 len, u32 kind, u64 seed, u64 *out)
{
    __shared__ u32 s_tab[1024];
    __shared__ u32 s_fold[16];
    crc_tables(s_tab);
    u32 lane = threadIdx.x & 63;
    u64 v;
    if (blockDim.x == 1024) /* CRC-32C only */
        v = block_crc32c(s_tab, s_fold, p, len, (u32)seed);
    else if (kind == 1)
        v = wave_crc32c(s_tab, p, len, (u32)seed, lane);
    else if (kind == 2)
        v = quad_xxh32(p, len, (u32)seed, lane & 3);
    else
        v = quad_xxh64(p, len, seed, lane & 3);
    if (threadIdx.x == 0)
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
    if (kind == 1 && n <= 2048) { /* few streams: the wavefronts that would idle share each stream */
        hipLaunchKernelGGL(k_crc_wide, dim3(n), dim3(1024), 0, (hipStream_t)stream, dst, dst_off, rec, n, (u32)seed, 1);
        return (int)hipGetLastError();
    }
    u64 per_block = kind == 1 ? 4 : 64; /* streams a block of 256 lanes takes at a time */
    u64 want = (n + per_block - 1) / per_block;
    u32 blocks = (u32)(want < 256 * 8 ? want : 256 * 8);
    hipLaunchKernelGGL(k_check, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dst, dst_off, rec, n, kind,
                       seed, 1);
    return (int)hipGetLastError();
}

/* a launch over no streams: the first launch of a kernel is when the runtime loads its code object and reserves
   what it needs on the device, and yeast_init() wants that behind it */
extern "C" int ys_warm_check(void *stream)
{
    hipLaunchKernelGGL(k_check, dim3(1), dim3(256), 0, (hipStream_t)stream, (const u8 *)nullptr, (const u64 *)nullptr,
                       (ys_record *)nullptr, 0u, 3u, (u64)0, 1);
    return (int)hipGetLastError();
}

extern "C" int ys_launch_check_buf(const u8 *p, u64 len, unsigned kind, u64 seed, u64 *out, void *stream)
{
    hipLaunchKernelGGL(k_check_buf, dim3(1), dim3(kind == 1 && len >= 65536 ? 1024 : 64), 0, (hipStream_t)stream, p, len,
                       kind, seed, out);
    return (int)hipGetLastError();
}

/* ---- SHA-256 (FIPS 180-4): the fourth check of the .br format (br-format-v3.txt:153-178; the
   reference takes it from OpenSSL, broad.c:266, brand.c:241).  The blocks of one message chain, so a
   message is one lane's work; messages run side by side. */
static __constant__ u32 c_sha_k[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98,
    0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786,
    0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8,
    0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819,
    0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a,
    0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7,
    0xc67178f2};

__device__ __forceinline__ u32 sha_ror(u32 x, int r) { return (x >> r) | (x << (32 - r)); }

/* message i is p[off[i] .. off[i] + len[i]); digest i goes to out[32 * i ..] */
__global__ __launch_bounds__(64) void k_sha256(const u8 *p, const u64 *off, const u64 *len, u32 n, u8 *out)
{
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const u8 *m = p + off[i];
    const u64 bytes = len[i], blocks = (bytes + 9 + 63) / 64;
    u32 h0 = 0x6a09e667, h1 = 0xbb67ae85, h2 = 0x3c6ef372, h3 = 0xa54ff53a, h4 = 0x510e527f, h5 = 0x9b05688c,
        h6 = 0x1f83d9ab, h7 = 0x5be0cd19;
    for (u64 b = 0; b < blocks; b++) {
        u32 w[16];
        for (u32 k = 0; k < 16; k++) {
            u32 v = 0;
            for (u32 j = 0; j < 4; j++) {
                u64 at = b * 64 + k * 4 + j;
                u32 c = at < bytes ? m[at] : at == bytes ? 0x80u : 0u;
                v = (v << 8) | c;
            }
            w[k] = v;
        }
        if (b == blocks - 1) {
            w[14] = (u32)((bytes * 8) >> 32);
            w[15] = (u32)(bytes * 8);
        }
        u32 a = h0, bb = h1, c = h2, d = h3, e = h4, f = h5, g = h6, hh = h7;
#pragma unroll
        for (u32 t = 0; t < 64; t++) {
            if (t >= 16) {
                u32 w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
                w[t & 15] += (sha_ror(w15, 7) ^ sha_ror(w15, 18) ^ (w15 >> 3)) + w[(t + 9) & 15] +
                             (sha_ror(w2, 17) ^ sha_ror(w2, 19) ^ (w2 >> 10));
            }
            u32 t1 = hh + (sha_ror(e, 6) ^ sha_ror(e, 11) ^ sha_ror(e, 25)) + ((e & f) ^ (~e & g)) + c_sha_k[t] + w[t & 15];
            u32 t2 = (sha_ror(a, 2) ^ sha_ror(a, 13) ^ sha_ror(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
        h0 += a; h1 += bb; h2 += c; h3 += d; h4 += e; h5 += f; h6 += g; h7 += hh;
    }
    u32 hs[8] = {h0, h1, h2, h3, h4, h5, h6, h7};
    for (u32 k = 0; k < 8; k++)
        for (u32 j = 0; j < 4; j++)
            out[32 * (u64)i + 4 * k + j] = (u8)(hs[k] >> (24 - 8 * j));
}

extern "C" int ys_launch_sha256(const u8 *p, const u64 *off, const u64 *len, unsigned n, u8 *out, void *stream)
{
    if (n == 0)
        return 0;
    hipLaunchKernelGGL(k_sha256, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, p, off, len, n, out);
    return (int)hipGetLastError();
}

/* ---- streaming XXH32 / XXH64 (xxhash.c:588-960): the accumulators live with the caller between
   updates; an update advances them over whole stripes, the digest folds them with the buffered tail.
   v: four accumulators in u64 slots (XXH32 uses the low halves). */
__global__ __launch_bounds__(64) void k_xxh_update(u32 is64, u64 *v, const u8 *p, u64 nstripe)
{
    u32 sub = threadIdx.x;
    if (sub >= 4)
        return;
    if (is64) {
        u64 acc = v[sub];
        const u8 *q = p + (sub << 3);
        for (u64 i = 0; i < nstripe; i++, q += 32)
            acc = xxh64_round(acc, ld64(q));
        v[sub] = acc;
    } else {
        u32 acc = (u32)v[sub];
        const u8 *q = p + (sub << 2);
        for (u64 i = 0; i < nstripe; i++, q += 16)
            acc = xxh32_round(acc, ld32(q));
        v[sub] = acc;
    }
}

/* XXH32_digest() xxhash.c:716-752, XXH64_digest() xxhash.c:870-934; mem: the buffered tail (memsize bytes) */
__global__ void k_xxh_digest(u32 is64, const u64 *v, u64 total, u64 seed, const u8 *mem, u32 memsize, u64 *out)
{
    if (threadIdx.x)
        return;
    const u8 *q = mem, *end = mem + memsize;
    if (is64) {
        u64 h;
        if (total >= 32) {
            h = rol64(v[0], 1) + rol64(v[1], 7) + rol64(v[2], 12) + rol64(v[3], 18);
            h = xxh64_merge(h, v[0]);
            h = xxh64_merge(h, v[1]);
            h = xxh64_merge(h, v[2]);
            h = xxh64_merge(h, v[3]);
        } else
            h = seed + P64_5;
        h += total;
        for (; q + 8 <= end; q += 8)
            h = rol64(h ^ xxh64_round(0, ld64(q)), 27) * P64_1 + P64_4;
        if (q + 4 <= end) {
            h = rol64(h ^ (u64)ld32(q) * P64_1, 23) * P64_2 + P64_3;
            q += 4;
        }
        for (; q < end; q++)
            h = rol64(h ^ (u64)ld8(q) * P64_5, 11) * P64_1;
        h ^= h >> 33; h *= P64_2; h ^= h >> 29; h *= P64_3; h ^= h >> 32;
        *out = h;
    } else {
        u32 h;
        if (total >= 16)
            h = rol32((u32)v[0], 1) + rol32((u32)v[1], 7) + rol32((u32)v[2], 12) + rol32((u32)v[3], 18);
        else
            h = (u32)seed + P32_5;
        h += (u32)total;
        for (; q + 4 <= end; q += 4)
            h = rol32(h + ld32(q) * P32_3, 17) * P32_4;
        for (; q < end; q++)
            h = rol32(h + ld8(q) * P32_5, 11) * P32_1;
        h ^= h >> 15; h *= P32_2; h ^= h >> 13; h *= P32_3; h ^= h >> 16;
        *out = h;
    }
}

extern "C" int ys_launch_xxh_update(unsigned is64, u64 *v, const u8 *p, u64 nstripe, void *stream)
{
    hipLaunchKernelGGL(k_xxh_update, dim3(1), dim3(64), 0, (hipStream_t)stream, is64, v, p, nstripe);
    return (int)hipGetLastError();
}
extern "C" int ys_launch_xxh_digest(unsigned is64, const u64 *v, u64 total, u64 seed, const u8 *mem, unsigned memsize,
                                    u64 *out, void *stream)
{
    hipLaunchKernelGGL(k_xxh_digest, dim3(1), dim3(64), 0, (hipStream_t)stream, is64, v, total, seed, mem, memsize, out);
    return (int)hipGetLastError();
}

YSB_HOST(check) /* bounds-checked debug build only: yeast_bounds.h */
