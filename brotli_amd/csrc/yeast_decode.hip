/*
 * yeast_decode.hip -- MI355X (gfx950) Brotli decode kernel.
 *
 * What this replaces in the reference (paths relative to /root/reference):
 *   yeast.c:163-179   bits()            -> Rd / take() (64-bit window, prefetched dword refills)
 *   yeast.c:204-219   decode()          -> dec_sym() (two-level LUT, 8-bit root)
 *   yeast.c:246-348   construct/simple  -> build_table()
 *   yeast.c:361-542   prefix()          -> read_code()
 *   yeast.c:550-647   block_length/block_types/context_map
 *   yeast.c:652-789   insert/copy lengths, context ids, distance()
 *   yeast.c:795-949   dict_word() + xforms.h + dict.h -> emit_dict()
 *   yeast.c:971-1308  metablock()       -> metablock()
 *   yeast.c:1353-1400 yeast()           -> decode_stream()
 *
 * Execution model: a *group* of G lanes (G = 64 ... 1, a power of two)
 * decodes one stream.  Everything that is a function of the bit stream alone
 * is computed redundantly by every lane of the group (no cross-lane traffic,
 * no branches on the lane id); the lanes differ only when bytes are moved:
 * LZ77 window copies, dictionary words and stored blocks are spread over the
 * G lanes.  Groups pull streams from an atomic work queue, so a launch is a
 * fixed-size grid of resident groups.
 *
 * Errors are sticky: a failed read marks the reader (rc set, no further bits
 * are handed out, the position freezes) and decoding winds down at the next
 * loop head.  That keeps the hot loops free of per-read exit paths.
 */
#include <hip/hip_runtime.h>
#include "yeast_stream.h"


template <int G, bool CMP>
__global__ __launch_bounds__(256, YS_MIN_WAVES) void k_decode(ys_params p)
{
    __shared__ Shared sh;
    shared_fill(sh, p.xforms);
    u32 lane = threadIdx.x;
    u32 sub = lane & (G - 1);
    u32 gid = (blockIdx.x * blockDim.x + lane) / G;
    u8 *ws = p.ws + (u64)gid * p.ws_stride;
    for (;;) {
        u32 job = 0;
        if (sub == 0)
            job = atomicAdd(p.counter, 1u);
        if (G > 1)
            job = __shfl(job, 0, G);
        if (job >= (p.retry ? *p.retry_count : p.n))
            break;
        if (p.retry)
            job = p.retry_list[job];
        decode_stream<G, CMP, 0>(p, job, ws, &sh, sub);
    }
}

/* ---------------------------------------------------------------- launchers */

extern "C" int ys_upload_constants(const u8 *lut768) { return ys_upload_constants_impl(lut768); }

template <int G, bool CMP>
static int launch_decode_t(const ys_params &p, u32 groups, hipStream_t st)
{
    u32 per_block = 256 / G;
    u32 blocks = (groups + per_block - 1) / per_block;
    hipLaunchKernelGGL((k_decode<G, CMP>), dim3(blocks), dim3(256), 0, st, p);
    return (int)hipGetLastError();
}

/* groups = number of resident groups (each owns one workspace slot) */
extern "C" int ys_launch_decode(const ys_params *p, int lanes_per_stream, int cmp, unsigned groups, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (cmp)
        return launch_decode_t<64, true>(*p, groups, st);
    switch (lanes_per_stream) {
    case 64: return launch_decode_t<64, false>(*p, groups, st);
    case 16: return launch_decode_t<16, false>(*p, groups, st);
    case 8: return launch_decode_t<8, false>(*p, groups, st);
    }
    return -1;
}

/* ---------------------------------------------------------------- the entry point that allocates nothing
 * yeast_batch_device_static() (include/yeast_batch.h) decodes with the cooperative kernel, one wavefront per
 * stream, on YS_STATIC_GROUPS resident wavefronts whose workspace slots -- worst-case size each, so no stream
 * ever needs the retry pass -- lie in this array instead of in memory the library allocates.  The array's name
 * is a host-side stand-in: the device address is asked for once per device (ys_static_init) and used from
 * then on.  Layout: 256 bytes of counters, the retry list (never filled: see above), the slots.
 */
#define YS_STATIC_GROUPS 16
#define YS_STATIC_MAX_STREAMS 16384
#define YS_STATIC_SLOT ((YS_WS_MAX + 255) & ~(size_t)255)
#define YS_STATIC_LIST (256 + (size_t)YS_STATIC_MAX_STREAMS * 4)
static __device__ __align__(256) unsigned char scratch[YS_STATIC_LIST + YS_STATIC_GROUPS * YS_STATIC_SLOT];
static unsigned char *g_scratch_ptr[64]; /* per device: each GPU holds its own copy of the array */

extern "C" int ys_static_init(int device)
{
    if (device < 0 || device >= 64)
        return -1;
    if (g_scratch_ptr[device])
        return 0;
    void *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr)
        return e != hipSuccess ? (int)e : -1;
    g_scratch_ptr[device] = (unsigned char *)ptr;
    return 0;
}
extern "C" unsigned ys_static_max_streams(void) { return YS_STATIC_MAX_STREAMS; }
extern "C" size_t ys_static_bytes(void) { return sizeof(scratch); }
extern "C" const void *ys_static_ptr(int device) { return device >= 0 && device < 64 ? g_scratch_ptr[device] : nullptr; }

__global__ void k_static_reset(u32 *counters) { counters[threadIdx.x] = 0; } /* the 256 bytes of counters */

/* p: everything but the workspace and the counters filled in by the caller; n <= YS_STATIC_MAX_STREAMS */
extern "C" int ys_launch_decode_static(const ys_params *p_in, int device, void *stream)
{
    if (device < 0 || device >= 64 || !g_scratch_ptr[device])
        return -1;
    unsigned char *ptr = g_scratch_ptr[device];
    hipStream_t st = (hipStream_t)stream;
    /* the counters start at zero; the slots need no clearing.  By a kernel, not hipMemsetAsync: captured into a HIP
       graph the memset node cleared the counters on the first replay only (ROCm 7.0/7.2, 256 bytes), and the second
       replay found the queue head where the first had left it and decoded nothing */
    hipLaunchKernelGGL(k_static_reset, dim3(1), dim3(64), 0, st, (u32 *)ptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return (int)e;
    ys_params p = *p_in;
    p.counter = (u32 *)ptr;
    p.retry_count = (u32 *)ptr + 2;
    p.retry_list = (u32 *)(ptr + 256);
    p.retry = 0;
    p.order = nullptr;
    p.debug = 0;
    p.rootcaps = 0;
    p.ws = ptr + YS_STATIC_LIST;
    p.ws_stride = YS_STATIC_SLOT;
    return launch_decode_t<64, false>(p, YS_STATIC_GROUPS, st);
}

YSB_HOST(decode) /* bounds-checked debug build only: yeast_bounds.h */
