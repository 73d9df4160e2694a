/*
 * yeast_lane.hip -- MI355X (gfx950) batch Brotli decode, one stream per lane: the kernel of large
 * batches (BASELINE config 2).  The per-stream decoder is yeast_stream.h, its command loop
 * yeast_trip.h; this file holds the kernel, the kernel that expands the static dictionary, and
 * their launchers.
 */
#include <hip/hip_runtime.h>
#include "yeast_stream.h"

/* Every stream of a wavefront starts at the same time, so header parsing, the command loop and the
   record write-out run converged across the S streams of the wavefront; a wavefront takes the
   next S streams when all of its streams are done.  S = 64 fills the wavefront; S = 32 leaves the
   upper half idle so that two wavefronts fit each SIMD and one's memory latency hides behind the
   other's work. */
#ifndef YS_MAX_OCC
#define YS_MAX_OCC 4 /* blocks per CU the register budget is cut for (experiments: 2 leaves S = 16 unspilled) */
#endif
/* UNIT: the form of the loop's stores (yeast_trip.h).  0: a store per piece and per literal; 1: a copy's last piece and the
   literals behind it as one unit; 2: chunk write-behind through a ring in LDS, for batches that keep every SIMD busy with
   two wavefronts */
template <int S, int UNIT = 0>
__global__ __launch_bounds__(256, (64 / S) > YS_MAX_OCC ? YS_MAX_OCC : 64 / S) void k_decode_lean(ys_params p)
{
    __shared__ TripShared tsh;
    __shared__ TripLds<S> wlds[4];
    trip_shared_fill(tsh, p.xforms, p.xdict_base);
    const Shared *shp = nullptr; /* the header parser of the batch kernel reads none of it */
    const TripShared *tsp = &tsh;
    u32 lane = threadIdx.x;
    u32 wave = (blockIdx.x * blockDim.x + lane) >> 6;
    u32 slot = lane & 63;
    /* the wavefront's workspaces: tables are addressed from wbase with 32-bit offsets (S * ws_stride <= 64 MiB) */
    u8 *const wbase = p.ws + (u64)(wave * S) * p.ws_stride;
    u8 *ws = wbase + (u64)(slot < S ? slot : 0) * p.ws_stride;
    if (p.debug & 1) /* every stream of the wavefront uses one workspace: only meaningful when they are copies of one stream */
        ws = wbase;
    const u32 limit = p.retry ? *p.retry_count : p.n;
    /* Pairing (p.debug bit 1; the host sets it when every block is resident at once, two per CU): the streams are handed
       out longest first, and blocks b and b + gridDim.x / 2 land on the same CU (observed, not promised: it only decides
       who shares a SIMD, never what is decoded).  The first half of the blocks takes the order as it is, the second half
       takes it from the end, so the wavefront with the longest streams shares its SIMD with the one that has the shortest
       and has it to itself soonest.  Later rounds, if any, take what is left in order. */
    bool first = (p.debug & 2) != 0 && !p.retry;
    bool delayed = first;
    const u32 nblk = gridDim.x;
    for (;;) {
        u32 base = 0;
        if (first) {
            const u32 b = blockIdx.x, half = nblk >> 1;
            const u32 r = b < half ? b : nblk - 1 - (b - half);
            base = (r * 4 + (lane >> 6)) * S;
            first = false;
        } else {
            if ((lane & 63) == 0)
                base = atomicAdd(p.counter, (u32)S) + ((p.debug & 2) && !p.retry ? nblk * 4 * S : 0);
            base = __builtin_amdgcn_readfirstlane(base);
        }
        if (base >= limit)
            break;
        u32 job = base + slot;
        const bool mine = slot < S && job < limit;
        if (mine) {
            if (p.retry)
                job = p.retry_list[job];
            else if (p.order)
                job = p.order[job];
        }
        /* Delayed start (one round, every block resident).  The launch lasts as long as the wavefronts with the longest
           streams; the wavefronts that share their SIMDs hold the shortest and have time to spare at the end.  They spend
           a little of it here, before they begin: the longest then parse their headers (3.3 ms alone, 3.8 ms beside
           another header on the SIMD) and make their first trips with the SIMD and the memory system less busy.  A stream's
           work follows its compressed size, so the wait is proportional to what the wavefront's longest stream lacks to
           the batch's longest -- nothing for a batch of equal streams -- and never more than 4 ms, about a header's
           time: a wavefront whose size did not predict its work extends the launch by that at most.  It decides when a
           stream is decoded, never how (44.2 -> 43.7 ms at 65,536 streams; profiles/r03/pace_sweep.txt). */
        if (delayed && p.delay) {
            u64 len = mine ? p.src_off[job + 1] - p.src_off[job] : 0;
            u32 l32 = len > 0x7fffffffull ? 0x7fffffffu : (u32)len, lw = 0;
            for (u64 todo = __ballot(1); todo; todo &= todo - 1) {
                u32 v = (u32)__builtin_amdgcn_readlane((int)l32, (int)__builtin_ctzll(todo));
                lw = v > lw ? v : lw;
            }
            const u32 lmax = __builtin_amdgcn_readfirstlane(*p.maxlen);
            if (lmax > lw) {
                u64 ticks = (u64)(lmax - lw) * p.delay / 10; /* s_memrealtime counts 100 MHz */
                ticks = ticks > 400000 ? 400000 : ticks;
                const u64 t0 = __builtin_amdgcn_s_memrealtime();
                while (__builtin_amdgcn_s_memrealtime() - t0 < ticks)
                    __builtin_amdgcn_s_sleep(32);
            }
        }
        delayed = false;
        if (mine) {
            decode_stream<1, false, S | (UNIT << 8)>(p, job, ws, shp, 0, &wlds[lane >> 6], tsp, wbase);
        }
    }
}

/* The expanded dictionary: entry (transform t, word length L, index i) is dict_word() of
   yeast.c:908-949 for id = t << NDBITS[L] | i, written with the same code the decode kernels use.
   One thread per entry; entries of one (t, L) lie back to back from base[t][L - 4]. */
__global__ __launch_bounds__(256) void k_expand_dict(const u8 *dict, const u8 *xforms, const u32 *base, u8 *xdict)
{
    __shared__ Shared sh;
    shared_fill(sh, xforms, base);
    u32 t = blockIdx.y;
    Ctx c;
    c.dict = dict;
    c.xf = xforms;
    c.sh = &sh;
    for (u32 L = 4; L <= 24; L++) {
        u32 nd = sh.dict[L] >> 24;
        for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < (1u << nd); i += gridDim.x * blockDim.x) {
            DictRef d;
            if (!dict_prepare(c, L, ((u64)t << nd) | i, d))
                continue;
            dict_write_exact(d, xdict + sh.xbase[t * YS_XD_LENS + L - 4] + i * d.total);
        }
    }
}

/* Hand-out order of the batch kernel.  The streams of a wavefront run until the longest of them is
   done, and the work of a stream follows its compressed size closely (correlation 0.999 on the
   text batch), so streams are handed out by size class, longest first: a wavefront gets streams
   of one class (5 bits of mantissa: classes are 3 % wide) and the last wavefronts get the
   shortest.  Three small launches: class histogram, scan, scatter.  The order inside a class is
   whatever the atomics give; results do not depend on it. */
#define YS_ORDER_BINS 1024
DEV u32 order_bin(u64 len)
{
    u32 l = len > 0xffffffffull ? 0xffffffffu : (u32)len;
    u32 key = l;
    if (l >= 32) {
        u32 e = 31 - __builtin_clz(l);
        key = ((e - 4) << 5) | ((l >> (e - 5)) & 31);
    }
    return YS_ORDER_BINS - 1 - key; /* key <= 27 * 32 + 31 */
}
/* both counting kernels count inside the block first (LDS atomics) and touch a global counter once
   per block and class: 65,536 streams fall into a few dozen classes */
__global__ __launch_bounds__(256) void k_order_hist(const u64 *src_off, u32 n, u32 *bins, u32 *maxlen)
{
    __shared__ u32 cnt[YS_ORDER_BINS];
    __shared__ u32 longest;
    if (threadIdx.x == 0)
        longest = 0;
    for (u32 k = threadIdx.x; k < YS_ORDER_BINS; k += 256)
        cnt[k] = 0;
    __syncthreads();
    u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        u64 len = src_off[i + 1] - src_off[i];
        atomicAdd(cnt + order_bin(len), 1u);
        atomicMax(&longest, len > 0x7fffffffull ? 0x7fffffffu : (u32)len);
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < YS_ORDER_BINS; k += 256)
        if (cnt[k])
            atomicAdd(bins + k, cnt[k]);
    if (threadIdx.x == 0)
        atomicMax(maxlen, longest);
}
__global__ __launch_bounds__(YS_ORDER_BINS) void k_order_scan(u32 *bins)
{
    __shared__ u32 t[YS_ORDER_BINS];
    u32 i = threadIdx.x, v = bins[i];
    t[i] = v;
    __syncthreads();
    for (u32 o = 1; o < YS_ORDER_BINS; o <<= 1) {
        u32 a = i >= o ? t[i - o] : 0;
        __syncthreads();
        t[i] += a;
        __syncthreads();
    }
    bins[i] = t[i] - v; /* exclusive */
}
__global__ __launch_bounds__(256) void k_order_scatter(const u64 *src_off, u32 n, u32 *bins, u32 *order)
{
    __shared__ u32 cnt[YS_ORDER_BINS];
    for (u32 k = threadIdx.x; k < YS_ORDER_BINS; k += 256)
        cnt[k] = 0;
    __syncthreads();
    u32 i = blockIdx.x * 256 + threadIdx.x, b = 0, rank = 0;
    if (i < n) {
        b = order_bin(src_off[i + 1] - src_off[i]);
        rank = atomicAdd(cnt + b, 1u);
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < YS_ORDER_BINS; k += 256)
        if (cnt[k])
            cnt[k] = atomicAdd(bins + k, cnt[k]); /* the block's first place in the class */
    __syncthreads();
    if (i < n)
        order[cnt[b] + rank] = i;
}

/* bins: YS_ORDER_BINS zeroed counters; maxlen: a zeroed counter (the longest stream's bytes); order: n entries */
extern "C" int ysl_order(const u64 *src_off, unsigned n, u32 *bins, u32 *maxlen, u32 *order, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    unsigned blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_order_hist, dim3(blocks), dim3(256), 0, st, src_off, n, bins, maxlen);
    hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(YS_ORDER_BINS), 0, st, bins);
    hipLaunchKernelGGL(k_order_scatter, dim3(blocks), dim3(256), 0, st, src_off, n, bins, order);
    return (int)hipGetLastError();
}

extern "C" int ysl_upload_constants(const u8 *lut768) { return ys_upload_constants_impl(lut768); }

extern "C" int ysl_expand_dict(const u8 *dict, const u8 *xforms, const u32 *base, u8 *xdict, void *stream)
{
    hipLaunchKernelGGL(k_expand_dict, dim3(4, 121), dim3(256), 0, (hipStream_t)stream, dict, xforms, base, xdict);
    return (int)hipGetLastError();
}

template <int S, int UNIT = 0>
static int launch_lean_t(const ys_params &p, u32 streams, hipStream_t st)
{
    u32 per_block = 4 * S;
    u32 blocks = (streams + per_block - 1) / per_block;
    hipLaunchKernelGGL((k_decode_lean<S, UNIT>), dim3(blocks), dim3(256), 0, st, p);
    return (int)hipGetLastError();
}

/* streams = resident streams (each owns one workspace slot); half != 0: half-filled wavefronts */
extern "C" int ysl_launch(const ys_params *p, int half, int merge, unsigned streams, void *stream)
{
    if (half == 1 && merge == 3)
        return launch_lean_t<32, 3>(*p, streams, (hipStream_t)stream);
    if (half == 0 && merge == 3)
        return launch_lean_t<64, 3>(*p, streams, (hipStream_t)stream);
    if (half == 1 && merge == 4)
        return launch_lean_t<32, 4>(*p, streams, (hipStream_t)stream);
    if (half == 0 && merge == 4)
        return launch_lean_t<64, 4>(*p, streams, (hipStream_t)stream);
    if (half == 1 && merge == 2)
        return launch_lean_t<32, 2>(*p, streams, (hipStream_t)stream);
    if (half == 0 && merge == 2)
        return launch_lean_t<64, 2>(*p, streams, (hipStream_t)stream);
    if (half == 1 && merge)
        return launch_lean_t<32, 1>(*p, streams, (hipStream_t)stream);
    if (half == 2) /* quarter-filled wavefronts (a test and measurement shape: the kernel needs ~196 registers with its
                      out-of-line header code, so two wavefronts per SIMD are resident whatever the shape) */
        return launch_lean_t<16>(*p, streams, (hipStream_t)stream);
    return half ? launch_lean_t<32>(*p, streams, (hipStream_t)stream) : launch_lean_t<64>(*p, streams, (hipStream_t)stream);
}

YSB_HOST(lane) /* bounds-checked debug build only: yeast_bounds.h */
