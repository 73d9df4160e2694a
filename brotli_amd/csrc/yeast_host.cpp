/*
 * yeast_host.cpp -- host side of libyeast_mi355x.so: the C-ABI of include/yeast.h and
 * include/yeast_batch.h (and the SHA-256 calls of include/yeast_br.h) on top of the HIP kernels of
 * yeast_lane.hip (one stream per lane), yeast_wave.hip (one wavefront per stream), yeast_decode.hip
 * (cooperative groups, compare mode) and yeast_check.hip.  No decode or check arithmetic lives here:
 * the host stages buffers, sizes output regions, picks the kernel for the batch size, launches, and
 * retries a stream whose output region or table arena turned out too small.
 *
 * Reference counterparts: yeast() yeast.c:1353-1400 (contract yeast.h:7-25); crc32c() crc32c.c:258;
 * XXH32()/XXH64() xxhash.c:365,505 and their streaming forms xxhash.c:588-960.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <atomic>
#include <thread>
#include <vector>

#include "yeast_common.h"
#include "../../include/yeast.h"
#include "../../include/yeast_batch.h"

extern "C" {
extern const unsigned char ys_dictionary[122784];
extern const unsigned char ys_transforms[2420];
int ys_upload_constants(const u8 *lut768);
int ys_upload_check_constants(const u32 *x2n);
int ys_launch_decode(const ys_params *p, int lanes_per_stream, int cmp, unsigned groups, void *stream);
int ys_static_init(int device);
unsigned ys_static_max_streams(void);
size_t ys_static_bytes(void);
int ys_launch_decode_static(const ys_params *p, int device, void *stream);
int ysl_upload_constants(const u8 *lut768);
int ysl_expand_dict(const u8 *dict, const u8 *xforms, const u32 *base, u8 *xdict, void *stream);
int ysl_launch(const ys_params *p, int half, int merge, unsigned streams, void *stream);
int ysl_order(const u64 *src_off, unsigned n, u32 *bins, u32 *maxlen, u32 *order, void *stream);
int ysw_upload_constants(const u8 *lut768);
unsigned ysw_shared_bytes(void);
int ysw_launch(const ys_params *p, unsigned slot_entries, unsigned rootcaps, unsigned blocks, void *stream);
int ys_launch_check(const u8 *dst, const u64 *dst_off, ys_record *rec, unsigned n, unsigned kind, u64 seed,
                    void *stream);
int ys_launch_check_buf(const u8 *p, u64 len, unsigned kind, u64 seed, u64 *out, void *stream);
int ys_warm_check(void *stream);
int ys_launch_xxh_update(unsigned is64, u64 *v, const u8 *p, u64 nstripe, void *stream);
int ys_launch_xxh_digest(unsigned is64, const u64 *v, u64 total, u64 seed, const u8 *mem, unsigned memsize, u64 *out,
                         void *stream);
int ys_launch_sha256(const u8 *p, const u64 *off, const u64 *len, unsigned n, u8 *out, void *stream);
int yeast_verbosity = 0;
}

static_assert(sizeof(ys_record) == 32 && sizeof(yeast_record) == 32, "record layout");

static thread_local char t_err[256] = "";
static thread_local int t_check_failed = 0; /* check-value calls of this thread that could not run since last asked */
static std::mutex g_mu;

static int fail(const char *what, hipError_t e)
{
    snprintf(t_err, sizeof(t_err), "libyeast_mi355x: %s: %s", what, hipGetErrorString(e));
    fprintf(stderr, "%s\n", t_err);
    return -1;
}
#define HIPCK(call)                  \
    do {                             \
        hipError_t e_ = (call);      \
        if (e_ != hipSuccess)        \
            return fail(#call, e_);  \
    } while (0)

extern "C" const char *yeast_last_error(void) { return t_err; }
/* which decode kernel the last batch call of this thread was given (as rocprofv3 prints its name, without the
   argument list): a measurement aid, so that a bench line names the kernel that ran and not one it guesses */
static thread_local char t_kernel[64] = "";
extern "C" const char *yeast_last_kernel(void) { return t_kernel; }
extern "C" int yeast_check_error(void)
{
    int n = t_check_failed;
    t_check_failed = 0;
    return n;
}

extern "C" int yeast_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

/* ---- static tables ------------------------------------------------------- */

/* The three literal-context tables (reference: lut0/lut1/lut2, yeast.c:700-740),
   generated from run-length descriptions of RFC 7932 section 7.1. */
static void make_ctx_luts(u8 *t)
{
    u8 *l0 = t, *l1 = t + 256, *l2 = t + 512;
    static const u8 l0_32[96] = {
        8,  12, 16, 12, 12, 20, 12, 16, 24, 28, 12, 12, 32, 12, 36, 12, 44, 44, 44, 44, 44, 44, 44, 44,
        44, 44, 32, 32, 24, 40, 28, 12, 12, 48, 52, 52, 52, 48, 52, 52, 52, 48, 52, 52, 52, 52, 52, 48,
        52, 52, 52, 52, 52, 48, 52, 52, 52, 52, 52, 24, 12, 28, 12, 12, 12, 56, 60, 60, 60, 56, 60, 60,
        60, 56, 60, 60, 60, 60, 60, 56, 60, 60, 60, 60, 60, 56, 60, 60, 60, 60, 60, 24, 12, 28, 12, 0};
    memset(l0, 0, 32);
    l0[9] = l0[10] = l0[13] = 4;
    memcpy(l0 + 32, l0_32, 96);
    for (int c = 128; c < 256; c++)
        l0[c] = (u8)((c & 1) + (c >= 192 ? 2 : 0));
    /* lut1: class of the second-last byte */
    struct Run { int upto; u8 v; };
    static const Run r1[] = {{33, 0}, {48, 1}, {58, 2}, {65, 1}, {91, 2}, {97, 1}, {123, 3}, {127, 1}, {224, 0}, {256, 2}};
    static const Run r2[] = {{1, 0}, {16, 1}, {64, 2}, {128, 3}, {192, 4}, {240, 5}, {255, 6}, {256, 7}};
    int c = 0;
    for (const Run &r : r1)
        for (; c < r.upto; c++)
            l1[c] = r.v;
    c = 0;
    for (const Run &r : r2)
        for (; c < r.upto; c++)
            l2[c] = r.v;
}

#define CRC_POLY 0x82f63b78u
static u32 crc_mul(u32 a, u32 b)
{
    u32 p = 0;
    for (int i = 0; i < 32; i++) {
        if (a & 0x80000000u)
            p ^= b;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1) ? CRC_POLY : 0);
    }
    return p;
}

/* ---- per-device state ------------------------------------------------------ */

struct Buf {
    void *p = nullptr;
    size_t cap = 0;
};

struct Device {
    bool ready = false;
    u8 *d_dict = nullptr, *d_xf = nullptr;
    u8 *d_xdict = nullptr;   /* every transform of every dictionary word, ~19 MB (yeast_lane.hip k_expand_dict) */
    u32 *d_xbase = nullptr;  /* [121][21] offsets into it */
    size_t xdict_bytes = 0;
    int cus = 256;
    Buf src, dst, soff, doff, rec, ws, ws_big, ctr, scratch;
    Buf pin;    /* pinned host staging for the outputs of a large host-pointer batch */
    Buf pin_in; /* and for its packed inputs */
    hipEvent_t last_ev = nullptr; /* end of the last enqueue(): the per-device counters and workspaces are shared */
    hipStream_t last_st = nullptr;
    bool have_ev = false;
};

static Device g_dev[64];
static u8 g_luts[768];
static bool g_luts_ready = false;

static int grow(Buf &b, size_t need)
{
    if (need <= b.cap && b.p)
        return 0;
    if (b.p)
        (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    size_t cap = need + need / 4 + 4096;
    HIPCK(hipMalloc(&b.p, cap));
    b.cap = cap;
    return 0;
}

static int grow_pinned(Buf &b, size_t need)
{
    if (need <= b.cap && b.p)
        return 0;
    if (b.p)
        (void)hipHostFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    size_t cap = need + need / 4 + 4096;
    HIPCK(hipHostMalloc(&b.p, cap, hipHostMallocDefault));
    b.cap = cap;
    return 0;
}

static int device_get(Device **out)
{
    int id = 0;
    HIPCK(hipGetDevice(&id));
    if (id < 0 || id >= 64) {
        snprintf(t_err, sizeof(t_err), "libyeast_mi355x: device id %d out of range", id);
        fprintf(stderr, "%s\n", t_err);
        return -1;
    }
    Device &d = g_dev[id];
    if (!d.ready) {
        if (!g_luts_ready) {
            make_ctx_luts(g_luts);
            g_luts_ready = true;
        }
        u32 x2n[68];
        x2n[0] = 0x40000000u;
        for (int k = 1; k < 68; k++)
            x2n[k] = crc_mul(x2n[k - 1], x2n[k - 1]);
        int e = ys_upload_constants(g_luts);
        if (!e)
            e = ysl_upload_constants(g_luts);
        if (!e)
            e = ysw_upload_constants(g_luts);
        if (!e)
            e = ys_upload_check_constants(x2n);
        if (e)
            return fail("upload constants (is a gfx950 code object present?)", (hipError_t)e);
        HIPCK(hipMalloc((void **)&d.d_dict, sizeof(ys_dictionary) + 64)); /* words are read 8 bytes at a time */
        HIPCK(hipMemset(d.d_dict, 0, sizeof(ys_dictionary) + 64));
        HIPCK(hipMalloc((void **)&d.d_xf, 4096));
        HIPCK(hipMemcpy(d.d_dict, ys_dictionary, sizeof(ys_dictionary), hipMemcpyHostToDevice));
        HIPCK(hipMemcpy(d.d_xf, ys_transforms, sizeof(ys_transforms), hipMemcpyHostToDevice));
        {
            /* layout of the expanded dictionary: for each transform and word length the entries
               (prefix + transformed word + suffix, a fixed size per pair) lie back to back */
            static const u8 ndbits[25] = {0, 0, 0, 0, 10, 10, 11, 11, 10, 10, 10, 10, 10, 9, 9, 8, 7, 7, 8, 7, 7, 6, 6, 5, 5};
            std::vector<u32> base(121 * 21);
            u64 total = 0;
            for (u32 t = 0; t < 121; t++) {
                const u8 *rec = ys_transforms + 20 * t;
                u32 op = rec[0], omit = rec[1], plen = rec[2], slen = rec[3];
                for (u32 L = 4; L <= 24; L++) {
                    u32 wlen = (op == 3 || op == 4) ? (L > omit ? L - omit : 0) : L;
                    base[t * 21 + L - 4] = (u32)total;
                    total += (u64)(plen + wlen + slen) << ndbits[L];
                }
            }
            HIPCK(hipMalloc((void **)&d.d_xdict, total + 64));
            d.xdict_bytes = total + 64;
            HIPCK(hipMemset(d.d_xdict, 0, total + 64));
            HIPCK(hipMalloc((void **)&d.d_xbase, base.size() * 4));
            HIPCK(hipMemcpy(d.d_xbase, base.data(), base.size() * 4, hipMemcpyHostToDevice));
            e = ysl_expand_dict(d.d_dict, d.d_xf, d.d_xbase, d.d_xdict, nullptr);
            if (e)
                return fail("dictionary expansion", (hipError_t)e);
            HIPCK(hipDeviceSynchronize());
        }
        hipDeviceProp_t prop;
        HIPCK(hipGetDeviceProperties(&prop, id));
        d.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        e = ys_static_init(id); /* the address of the static workspace of yeast_batch_device_static() */
        if (e)
            return fail("static workspace address", (hipError_t)e);
        {
            /* The first launch of a kernel is when the runtime loads its code object and reserves the kernel's private
               memory on the device.  yeast_batch_device_static() must cause no allocation, the runtime's included, so
               its two kernels are launched here once, over no streams. */
            ys_params p0;
            memset(&p0, 0, sizeof(p0));
            p0.dict = d.d_dict;
            p0.xforms = d.d_xf;
            p0.xdict = d.d_xdict;
            p0.xdict_base = d.d_xbase;
            e = ys_launch_decode_static(&p0, id, nullptr);
            if (!e)
                e = ys_warm_check(nullptr);
            if (e)
                return fail("warm-up launch", (hipError_t)e);
            HIPCK(hipDeviceSynchronize());
        }
        HIPCK(hipEventCreateWithFlags(&d.last_ev, hipEventDisableTiming));
        d.ready = true;
    }
    *out = &d;
    return 0;
}

extern "C" int yeast_trim(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    int id = 0;
    HIPCK(hipGetDevice(&id));
    if (id < 0 || id >= 64 || !g_dev[id].ready)
        return 0;
    Device &d = g_dev[id];
    HIPCK(hipDeviceSynchronize());
    for (Buf *b : {&d.src, &d.dst, &d.soff, &d.doff, &d.rec, &d.ws, &d.ws_big, &d.ctr, &d.scratch}) {
        if (b->p)
            HIPCK(hipFree(b->p));
        b->p = nullptr;
        b->cap = 0;
    }
    for (Buf *b : {&d.pin, &d.pin_in}) {
        if (b->p)
            HIPCK(hipHostFree(b->p));
        b->p = nullptr;
        b->cap = 0;
    }
    return 0;
}

extern "C" const unsigned char *yeast_debug_table(int which, size_t *size)
{
    if (!g_luts_ready) {
        make_ctx_luts(g_luts);
        g_luts_ready = true;
    }
    if (which >= 0 && which < 3) {
        *size = 256;
        return g_luts + 256 * which;
    }
    if (which == 3) {
        *size = sizeof(ys_dictionary);
        return ys_dictionary;
    }
    if (which == 4) {
        *size = sizeof(ys_transforms);
        return ys_transforms;
    }
    *size = 0;
    return nullptr;
}

/* ---- launch ------------------------------------------------------------------ */

#define YS_LANES_WAVE 65 /* lanes_per_stream value of the wavefront-per-stream batch kernel (LDS tables) */

/* The YEAST_* environment knobs (measurement and test aids) are read once, when the library is first used, and
   again only on yeast_reload_knobs(): never on the path of a call. */
static const char *const KNOB_NAMES[] = {"YEAST_LANES", "YEAST_WAVES_PER_CU", "YEAST_SLOT_ENTRIES", "YEAST_ROOT_LIT",
                                         "YEAST_ROOT_IAC", "YEAST_ROOT_DIST", "YEAST_HALF", "YEAST_BLOCKS_PER_CU",
                                         "YEAST_WS_KB", "YEAST_LEAN_ROOT_LIT", "YEAST_LEAN_ROOT_IAC",
                                         "YEAST_LEAN_ROOT_DIST", "YEAST_SORT", "YEAST_DEBUG", "YEAST_HOST_THREADS",
                                         "YEAST_FORCE_LANES", "YEAST_PRIO", "YEAST_HOST_BUDGET_MB", "YEAST_MERGE", "YEAST_PAIR",
                                         "YEAST_HOST_FLOOR", "YEAST_D2H_PIN_KB", "YEAST_BOUNDS_SHRINK", "YEAST_CHUNK64_FROM", "YEAST_DELAY"};
enum { KNOBS = sizeof(KNOB_NAMES) / sizeof(KNOB_NAMES[0]) };
static int g_knob_val[KNOBS];
static bool g_knob_set[KNOBS];
static std::atomic<bool> g_knobs_ready{false};
static void load_knobs()
{
    for (int k = 0; k < KNOBS; k++) {
        const char *s = getenv(KNOB_NAMES[k]);
        g_knob_set[k] = s && *s;
        g_knob_val[k] = g_knob_set[k] ? atoi(s) : 0;
    }
    g_knobs_ready = true;
}
static int env_int(const char *name, int dflt)
{
    if (!g_knobs_ready)
        load_knobs();
    for (int k = 0; k < KNOBS; k++)
        if (!strcmp(name, KNOB_NAMES[k]))
            return g_knob_set[k] ? g_knob_val[k] : dflt;
    return dflt;
}
extern "C" void yeast_reload_knobs(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    load_knobs();
}

#ifdef YS_BOUNDS
/* ---- the bounds-checked debug build (yeast_bounds.h): the ranges the kernels of one call may touch, written into
   every code object before its launches; the count of accesses outside them read back after the call ---- */
#include "yeast_bounds.h"
extern "C" {
int ysb_set_check(const ys_bounds_t *), ysb_get_check(ys_bounds_t *), ysb_set_decode(const ys_bounds_t *),
    ysb_get_decode(ys_bounds_t *), ysb_set_wave(const ys_bounds_t *), ysb_get_wave(ys_bounds_t *),
    ysb_set_lane(const ys_bounds_t *), ysb_get_lane(ys_bounds_t *);
const void *ys_static_ptr(int device);
}
static const char *g_ysb_name[YSB_MAX];
static void ysb_add(ys_bounds_t &b, const char *name, const void *p, size_t n)
{
    if (b.n >= YSB_MAX)
        return;
    g_ysb_name[b.n] = name; /* a buffer that does not exist yet keeps its place in the list (an empty range) */
    b.lo[b.n] = p && n ? (unsigned long long)(size_t)p : 0;
    b.hi[b.n] = p && n ? b.lo[b.n] + n : 0;
    b.n++;
}
static int ysb_set_all(const ys_bounds_t &b)
{
    int e = ysb_set_check(&b);
    e = e ? e : ysb_set_decode(&b);
    e = e ? e : ysb_set_wave(&b);
    e = e ? e : ysb_set_lane(&b);
    return e;
}
/* what the launches of the call so far have counted, by code object (writing a new list clears the device's counts) */
static ys_bounds_t g_ysb_got[4];
static int ysb_collect(void)
{
    ys_bounds_t now[4];
    memset(now, 0, sizeof(now));
    int e = ysb_get_check(&now[0]);
    e = e ? e : ysb_get_decode(&now[1]);
    e = e ? e : ysb_get_wave(&now[2]);
    e = e ? e : ysb_get_lane(&now[3]);
    for (int k = 0; k < 4 && !e; k++) {
        if (!now[k].hits)
            continue;
        unsigned before = g_ysb_got[k].hits;
        if (!before)
            g_ysb_got[k] = now[k];
        g_ysb_got[k].hits = before + now[k].hits;
    }
    return e;
}
/* called before each launch of a checked call (the library's buffers may have grown since the last one) */
static int ysb_arm(Device &d, const u8 *src, const u64 *soff, u8 *dst, const u64 *doff, ys_record *rec, size_t n)
{
    u64 send = 0, dend = 0;
    HIPCK(hipMemcpy(&send, soff + n, 8, hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(&dend, doff + n, 8, hipMemcpyDeviceToHost));
    /* self-test of the checker: declare the output arena this many bytes shorter than it is */
    u64 shrink = (u64)env_int("YEAST_BOUNDS_SHRINK", 0);
    dend -= shrink < dend ? shrink : dend;
    int id = 0;
    HIPCK(hipGetDevice(&id));
    ys_bounds_t b;
    memset(&b, 0, sizeof(b));
    ysb_add(b, "dictionary", d.d_dict, sizeof(ys_dictionary) + 64); /* range 0: what a refused LDS-landing load reads instead */
    ysb_add(b, "transforms", d.d_xf, 4096);
    ysb_add(b, "expanded dictionary", d.d_xdict, d.xdict_bytes);
    ysb_add(b, "expanded dictionary index", d.d_xbase, 121 * 21 * 4);
    ysb_add(b, "table workspace", d.ws.p, d.ws.cap);
    ysb_add(b, "retry workspace", d.ws_big.p, d.ws_big.cap);
    ysb_add(b, "counters and lists", d.ctr.p, d.ctr.cap);
    ysb_add(b, "static workspace", ys_static_ptr(id), ys_static_bytes());
    ysb_add(b, "d_src (+ YEAST_SRC_PAD)", src, send + YS_SRC_PAD);
    ysb_add(b, "d_src_off", soff, (n + 1) * 8);
    ysb_add(b, "d_dst", dst, dend);
    ysb_add(b, "d_dst_off", doff, (n + 1) * 8);
    ysb_add(b, "d_rec", rec, n * sizeof(ys_record));
    int e = ysb_collect(); /* waits for the launches so far */
    e = e ? e : ysb_set_all(b);
    if (e)
        return fail("bounds list upload", (hipError_t)e);
    return 0;
}
/* after the call: waits for the device, reads the counts, switches checking off again; -1 if anything was refused */
static int ysb_verdict(void)
{
    HIPCK(hipDeviceSynchronize());
    int e = ysb_collect();
    ys_bounds_t got[4];
    memcpy(got, g_ysb_got, sizeof(got));
    memset(g_ysb_got, 0, sizeof(g_ysb_got));
    ys_bounds_t off;
    memset(&off, 0, sizeof(off));
    e = e ? e : ysb_set_all(off);
    if (e)
        return fail("bounds list read-back", (hipError_t)e);
    static const char *const tu[4] = {"check", "cooperative/retry", "wavefront-per-stream", "batch"};
    for (int k = 0; k < 4; k++) {
        if (!got[k].hits)
            continue;
        /* the nearest range, for the message */
        unsigned best = 0;
        unsigned long long bd = ~0ull;
        for (unsigned i = 0; i < got[k].n; i++) {
            if (!got[k].hi[i])
                continue;
            unsigned long long a = got[k].addr, dd = a < got[k].lo[i] ? got[k].lo[i] - a : a >= got[k].hi[i] ? a - got[k].hi[i] + 1 : 0;
            if (dd < bd) {
                bd = dd;
                best = i;
            }
        }
        snprintf(t_err, sizeof(t_err),
                 "libyeast (bounds build): %u access%s of the %s kernels outside every buffer; the first: %u bytes at %#llx, "
                 "%lld bytes %s %s [%#llx, %#llx)",
                 got[k].hits, got[k].hits == 1 ? "" : "es", tu[k], got[k].size, got[k].addr,
                 (long long)(got[k].addr < got[k].lo[best] ? got[k].lo[best] - got[k].addr : got[k].addr + got[k].size - got[k].hi[best]),
                 got[k].addr < got[k].lo[best] ? "before" : "past the end of", g_ysb_name[best], got[k].lo[best], got[k].hi[best]);
        fprintf(stderr, "%s\n", t_err);
        for (int j = 0; j < 6 && (unsigned)j + 1 < got[k].hits; j++)
            fprintf(stderr, "  then %u bytes at %#llx\n", (unsigned)(got[k].more[j] >> 56), got[k].more[j] & 0xffffffffffffffull);
        return -1;
    }
    return 0;
}
#define YSB_ARM()                                                                                                      \
    do {                                                                                                               \
        if (ysb_arm(d, src, soff, dst, doff, rec, n))                                                                  \
            return -1;                                                                                                 \
    } while (0)
#else
#define YSB_ARM() do { } while (0)
#endif

/* Enqueue decode (+ retry pass for streams whose tables did not fit the
   default arena, + check values) for n streams.  Everything is device
   resident; nothing here synchronises (the bounds-checked debug build does). */
static int enqueue_impl(Device &d, const u8 *src, const u64 *soff, u8 *dst, const u64 *doff, ys_record *rec, size_t n,
                   unsigned check_kind, int lanes, int cmp, hipStream_t st)
{
    if (n == 0)
        return 0;
    if (n > 0x7fffffffu) {
        snprintf(t_err, sizeof(t_err), "libyeast_mi355x: batch too large");
        return -1;
    }
    /* Every call on a device shares its counters and table workspaces: a call on another stream waits for the
       last one's kernels (calls on one stream are ordered anyway). */
    if (d.have_ev && d.last_st != st)
        HIPCK(hipStreamWaitEvent(st, d.last_ev, 0));
    struct Mark { /* whatever the way out, later calls wait for what this one has enqueued */
        Device &d;
        hipStream_t st;
        ~Mark()
        {
            if (hipEventRecord(d.last_ev, st) == hipSuccess) {
                d.last_st = st;
                d.have_ev = true;
            }
        }
    } mark{d, st};
    if (lanes == 0) /* large batches: one stream per lane; small ones: one stream per wavefront, tables in LDS */
        lanes = env_int("YEAST_LANES", n >= 6656 ? 1 : YS_LANES_WAVE); /* crossover: 37 against 41 ms at 6,144 streams of 64 KiB, 43 against 41.5 at 7,168 (tools/series.sh) */
    if (cmp)
        lanes = 64;
    if (lanes != 1 && lanes != 8 && lanes != 16 && lanes != 64 && lanes != YS_LANES_WAVE) {
        snprintf(t_err, sizeof(t_err), "libyeast_mi355x: lanes_per_stream must be 1, 8, 16, 64 or 65");
        return -1;
    }
    /* counters: [0] main queue head, [1] retry queue head, [2] retry count; then the size-class
       counters of the hand-out order, the retry list and the hand-out order */
    const size_t order_bins = 1024;
    if (grow(d.ctr, 16 + order_bins * sizeof(u32) + 2 * n * sizeof(u32)))
        return -1;
    HIPCK(hipMemsetAsync(d.ctr.p, 0, 16 + order_bins * sizeof(u32), st));
    ys_params p;
    memset(&p, 0, sizeof(p));
    p.src = src;
    p.src_off = soff;
    p.dst = dst;
    p.dst_off = doff;
    p.rec = rec;
    p.counter = (u32 *)d.ctr.p;
    p.retry_count = (u32 *)d.ctr.p + 2;
    p.retry_list = (u32 *)d.ctr.p + 4 + order_bins;
    p.retry = 0;
    p.n = (u32)n;
    p.dict = d.d_dict;
    p.xforms = d.d_xf;
    p.xdict = d.d_xdict;
    p.xdict_base = d.d_xbase;
    int e;
    if (lanes == YS_LANES_WAVE) {
        /* one wavefront per stream, tables in LDS: blocks of four wavefronts; the LDS of a CU is
           split between the resident blocks, so the slot size sets how many streams a CU holds */
        unsigned waves_per_cu = (unsigned)env_int("YEAST_WAVES_PER_CU", 16);
        if (waves_per_cu < 4)
            waves_per_cu = 4;
        if (waves_per_cu > 32)
            waves_per_cu = 32;
        waves_per_cu &= ~3u;
        unsigned blocks_per_cu = waves_per_cu / 4;
        unsigned lds_per_block = (160u * 1024u) / blocks_per_cu;
        if (lds_per_block > 64u * 1024u)
            lds_per_block = 64u * 1024u;
        unsigned slot_entries = ((lds_per_block - ysw_shared_bytes() - 64) / 4 / 2) & ~7u;
        slot_entries = (unsigned)env_int("YEAST_SLOT_ENTRIES", (int)slot_entries);
        /* widest root tables: other | literal << 8 | insert-and-copy << 16 | distance << 24 */
        unsigned rootcaps = 8u | ((unsigned)env_int("YEAST_ROOT_LIT", 7) << 8) | ((unsigned)env_int("YEAST_ROOT_IAC", 8) << 16) |
                            ((unsigned)env_int("YEAST_ROOT_DIST", 8) << 24);
        size_t want = (n + 3) / 4, maxb = (size_t)d.cus * blocks_per_cu;
        unsigned blocks = (unsigned)(want < maxb ? want : maxb);
        u64 stride = 32768; /* header scratch per wavefront: yeast_wave.hip w_metablock() */
        if (grow(d.ws, (size_t)blocks * 4 * stride))
            return -1;
        p.ws = (u8 *)d.ws.p;
        p.ws_stride = stride;
        snprintf(t_kernel, sizeof(t_kernel), "k_decode_wave");
        YSB_ARM();
        e = ysw_launch(&p, slot_entries, rootcaps, blocks, st);
        if (e)
            return fail("decode launch", (hipError_t)e);
    } else {
    /* 1 lane: the batch kernel (one stream per lane, yeast_lane.hip);
       8, 16, 64 lanes: the cooperative kernel (wide copies; compare mode) */
    const bool flat = lanes == 1;
    /* one stream per lane: half-filled wavefronts (two per SIMD, the default), full ones (0) or
       quarter-filled ones (2) */
    /* half-filled wavefronts, two per SIMD: 65,536 streams per 256 CUs are one round (with the plain loop larger
       batches were better off in them too: 107 ms against 116 ms in full wavefronts at 131,072 streams) */
    /* with the write-behind form of the loop (YEAST_MERGE 2, below) full wavefronts win as soon as there is more than
       one round: 82.8 against 88.6 ms at 131,072 streams, 162.7 against 173.7 at 262,144 (one round: 49.3 against 47.6) */
    /* YEAST_MERGE: the form of the batch kernel's loop.  3 (the default): write-behind through the LDS ring with the work of a
       trip done per symbol kind (yeast_trip3.h): 41.3 ms at 8,192 streams, 43.0 at 32,768, 46.5 at 65,536 -- as fast as the
       plain loop (0: 41.1 at 8,192) where a wavefront has its SIMD to itself and faster than the earlier write-behind
       loop (2: 45.2 / 46.0 / 47.8) everywhere, so one form serves all batch sizes.  0, 1, 2 stay selectable (tests, A/B) */
    /* 3 and 4 are the same loop with 64-byte and with 16-byte output chunks: whole lines for batches that keep every SIMD
       busy with two wavefronts (a chunk is evicted from L2 before its line is complete: 45.6 -> 44.0 ms at 65,536 streams,
       160.5 -> 151.3 at 262,144), the shorter trip below that (40.1 against 41.2 ms at 8,192; the two meet between 16,384
       and 24,576 streams) */
    const int merge = flat ? env_int("YEAST_MERGE", n > (size_t)d.cus * (size_t)env_int("YEAST_CHUNK64_FROM", 80) ? 3 : 4) : 0;
    const int half = flat ? env_int("YEAST_HALF", merge >= 2 && n > (size_t)d.cus * 256 ? 0 : 1) : 0;
    unsigned per_block = 256u / (unsigned)lanes / (half == 2 ? 4u : half ? 2u : 1u);
    /* full wavefronts with the ring: 110 KB of LDS per block, one block per CU */
    unsigned blocks_per_cu = (unsigned)env_int("YEAST_BLOCKS_PER_CU", flat ? (half == 2 ? 4 : half == 0 && merge >= 2 ? 1 : 2) : 4);
    size_t want_blocks = (n + per_block - 1) / per_block;
    size_t max_blocks = (size_t)d.cus * blocks_per_cu;
    unsigned blocks = (unsigned)(want_blocks < max_blocks ? want_blocks : max_blocks);
    unsigned groups = blocks * per_block;
    u64 stride = (u64)env_int("YEAST_WS_KB", 48) * 1024;
    if (stride < YS_WS_ARENA + 8192)
        stride = YS_WS_ARENA + 8192;
    if (flat && stride > (1u << 20))
        stride = 1u << 20; /* the batch kernel keeps table offsets >> 4 in 16 bits */
    if (grow(d.ws, (size_t)groups * stride + 256))
        return -1;
    p.ws = (u8 *)d.ws.p;
    p.ws_stride = stride;
    if (flat) {
        /* root table widths of the batch kernel's tables */
        unsigned rl = (unsigned)env_int("YEAST_LEAN_ROOT_LIT", 7), ri = (unsigned)env_int("YEAST_LEAN_ROOT_IAC", 8),
                 rd = (unsigned)env_int("YEAST_LEAN_ROOT_DIST", 7);
        auto clamp = [](unsigned v) { return v < 4 ? 4u : v > 8 ? 8u : v; };
        p.rootcaps = 8u | (clamp(rl) << 8) | (clamp(ri) << 16) | (clamp(rd) << 24);
        YSB_ARM();
        if (env_int("YEAST_SORT", 1)) { /* hand the streams out by size class, longest first (yeast_lane.hip) */
            u32 *order = p.retry_list + n;
            e = ysl_order(soff, (unsigned)n, (u32 *)d.ctr.p + 4, (u32 *)d.ctr.p + 3, order, st);
            p.maxlen = (u32 *)d.ctr.p + 3;
            if (e)
                return fail("hand-out order launch", (hipError_t)e);
            p.order = order;
        }
        p.debug = (u32)env_int("YEAST_DEBUG", 0);
        /* two blocks per CU, all resident at once, sorted hand-out: pair long with short on a SIMD (yeast_lane.hip) */
        if (p.order && blocks_per_cu == 2 && blocks == 2 * (unsigned)d.cus && (blocks & 1) == 0 && env_int("YEAST_PAIR", 1)) {
            p.delay = (u32)env_int("YEAST_DELAY", 300); /* ns per byte a wavefront's longest stream lacks to the batch's longest */
            p.debug |= 2;
        }
        /* (issue priority for the wavefronts with the long streams on top of the pairing: no difference, 46.5 ms either way) */
        /* batches that fill the GPU: the loop that gathers its stores (yeast_trip.h; YEAST_MERGE 2: every output byte
           through a ring in LDS and out in complete 16-byte chunks, 47.6 ms against 57.4 at 65,536 streams; 1: one store
           per command, 52.6 ms; 0: the plain loop, faster at 32,768 streams and fewer, where a wavefront has its SIMD to
           itself and every LDS round trip shows: 43.2 against 45.4 ms; the two meet at 40,960 streams) */
        {
            int S = half == 2 ? 16 : half ? 32 : 64, form = (half == 0 || half == 1) && merge >= 2 ? merge : half == 1 && merge ? 1 : 0;
            if (form)
                snprintf(t_kernel, sizeof(t_kernel), "k_decode_lean<%d, %d>", S, form);
            else
                snprintf(t_kernel, sizeof(t_kernel), "k_decode_lean<%d, 0>", S);
        }
        e = ysl_launch(&p, half, merge, groups, st);
        p.order = nullptr;
        p.debug = 0;
        p.delay = 0;
    }
    else {
        snprintf(t_kernel, sizeof(t_kernel), "k_decode<%d, %s>", lanes, cmp ? "true" : "false");
        YSB_ARM();
        e = ys_launch_decode(&p, lanes, cmp, groups, st);
    }
    if (e)
        return fail("decode launch", (hipError_t)e);
    }

    /* retry pass: whole wavefronts, worst-case arena; exits at once when the list is empty */
    const unsigned big_groups = 64;
    if (grow(d.ws_big, (size_t)big_groups * YS_WS_MAX))
        return -1;
    p.ws = (u8 *)d.ws_big.p;
    p.ws_stride = YS_WS_MAX;
    p.counter = (u32 *)d.ctr.p + 1;
    p.retry = 1;
    p.rootcaps = 0;
    YSB_ARM();
    e = ys_launch_decode(&p, 64, cmp, big_groups, st);
    if (e)
        return fail("retry launch", (hipError_t)e);

    if (check_kind) {
        e = ys_launch_check(dst, doff, rec, (unsigned)n, check_kind, 0, st);
        if (e)
            return fail("check launch", (hipError_t)e);
    }
    return 0;
}

static int enqueue(Device &d, const u8 *src, const u64 *soff, u8 *dst, const u64 *doff, ys_record *rec, size_t n,
                   unsigned check_kind, int lanes, int cmp, hipStream_t st)
{
    int e = enqueue_impl(d, src, soff, dst, doff, rec, n, check_kind, lanes, cmp, st);
#ifdef YS_BOUNDS
    if (n) {
        int v = ysb_verdict();
        e = e ? e : v;
    }
#endif
    return e;
}

extern "C" int yeast_batch_device(const void *d_src, const uint64_t *d_src_off, void *d_dst,
                                  const uint64_t *d_dst_off, yeast_record *d_rec, size_t n, unsigned check_kind,
                                  int lanes_per_stream, void *hip_stream)
{
    std::lock_guard<std::mutex> lk(g_mu);
    Device *d;
    if (device_get(&d))
        return -1;
    if (check_kind > 3) {
        snprintf(t_err, sizeof(t_err), "libyeast_mi355x: unknown check kind %u", check_kind);
        return -1;
    }
    return enqueue(*d, (const u8 *)d_src, d_src_off, (u8 *)d_dst, d_dst_off, (ys_record *)d_rec, n, check_kind,
                   lanes_per_stream, 0, (hipStream_t)hip_stream);
}

extern "C" int yeast_init(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    Device *d;
    return device_get(&d) ? -1 : 0;
}

/* The same decode and check as yeast_batch_device(), for callers that cannot allocate: nothing here calls
   hipMalloc, hipFree or anything that does (the device must have been set up by yeast_init() or an earlier call
   of any other entry point, which may allocate).  The kernels' workspace is a static device array. */
extern "C" int yeast_batch_device_static(const void *d_src, const uint64_t *d_src_off, void *d_dst,
                                         const uint64_t *d_dst_off, yeast_record *d_rec, size_t n,
                                         unsigned check_kind, void *hip_stream)
{
    std::lock_guard<std::mutex> lk(g_mu);
    int id = 0;
    HIPCK(hipGetDevice(&id));
    if (id < 0 || id >= 64 || !g_dev[id].ready) {
        snprintf(t_err, sizeof(t_err), "libyeast_mi355x: yeast_batch_device_static() before yeast_init() on this device");
        return -1;
    }
    if (check_kind > 3) {
        snprintf(t_err, sizeof(t_err), "libyeast_mi355x: unknown check kind %u", check_kind);
        return -1;
    }
    if (n > ys_static_max_streams()) {
        snprintf(t_err, sizeof(t_err), "libyeast_mi355x: yeast_batch_device_static() takes at most %u streams per call (%zu asked for)",
                 ys_static_max_streams(), n);
        return -1;
    }
    if (n == 0)
        return 0;
    Device &d = g_dev[id];
    hipStream_t st = (hipStream_t)hip_stream;
    /* Calls on one device share the static workspace, so a call on another stream is put behind the last one.
       Not while the stream is being captured into a HIP graph: an event recorded outside the capture cannot be
       waited for inside it (the capture would be invalidated), and one recorded inside belongs to the graph.  A
       captured call is ordered by whoever launches the graph: replays must not overlap other calls on the device. */
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st != nullptr && hipStreamIsCapturing(st, &cap) != hipSuccess)
        cap = hipStreamCaptureStatusNone;
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (!capturing && d.have_ev && d.last_st != st)
        HIPCK(hipStreamWaitEvent(st, d.last_ev, 0));
    ys_params p;
    memset(&p, 0, sizeof(p));
    p.src = (const u8 *)d_src;
    p.src_off = d_src_off;
    p.dst = (u8 *)d_dst;
    p.dst_off = d_dst_off;
    p.rec = (ys_record *)d_rec;
    p.n = (u32)n;
    p.dict = d.d_dict;
    p.xforms = d.d_xf;
    p.xdict = d.d_xdict;
    p.xdict_base = d.d_xbase;
    int e = ys_launch_decode_static(&p, id, st);
    if (!e && check_kind)
        e = ys_launch_check((const u8 *)d_dst, d_dst_off, (ys_record *)d_rec, (unsigned)n, check_kind, 0, st);
    if (!capturing && hipEventRecord(d.last_ev, st) == hipSuccess) {
        d.last_st = st;
        d.have_ev = true;
    }
    if (e)
        return fail("static decode launch", (hipError_t)e);
    return 0;
}

extern "C" int yeast_check_device(const void *d_dst, const uint64_t *d_dst_off, yeast_record *d_rec, size_t n,
                                  unsigned check_kind, void *hip_stream)
{
    std::lock_guard<std::mutex> lk(g_mu);
    Device *d;
    if (device_get(&d))
        return -1;
    if (check_kind < 1 || check_kind > 3) {
        snprintf(t_err, sizeof(t_err), "libyeast_mi355x: unknown check kind %u", check_kind);
        return -1;
    }
    int e = ys_launch_check((const u8 *)d_dst, d_dst_off, (ys_record *)d_rec, (unsigned)n, check_kind, 0,
                            hip_stream);
    if (e)
        return fail("check launch", (hipError_t)e);
    return 0;
}

/* ---- host-pointer paths ---------------------------------------------------- */

/* Read WBITS and the first meta-block header on the host, only to size the
   output region (yeast.c:1372-1378, 990-1038).  Returns a guess of the output
   size; never an error: the kernels do the real parse. */
static u64 first_mlen_guess(const u8 *s, size_t len)
{
    u64 pos = 0, nbits = (u64)len * 8;
    auto bits = [&](unsigned n, u32 &v) -> bool {
        if (pos + n > nbits)
            return false;
        v = 0;
        for (unsigned i = 0; i < n; i++, pos++)
            v |= (u32)((s[pos >> 3] >> (pos & 7)) & 1) << i;
        return true;
    };
    u32 v, n;
    if (!bits(1, v))
        return 0;
    if (v) {
        if (!bits(3, v))
            return 0;
        if (!v) {
            if (!bits(3, v))
                return 0;
        }
    }
    u32 last;
    if (!bits(1, last))
        return 0;
    if (last) {
        if (!bits(1, v) || v)
            return 0;
    }
    if (!bits(2, n) || n == 3)
        return 0;
    if (!bits(16, v))
        return 0;
    u64 mlen = 1 + (u64)v;
    if (n) {
        if (!bits(n << 2, v))
            return 0;
        mlen += (u64)v << 16;
    }
    return mlen;
}

static size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

/* Packing the inputs and cutting up the outputs of a large host-pointer batch is plain memcpy: a few
   host threads (YEAST_HOST_THREADS, default 8) share it, fn(first, last) each over its range. */
template <class F> static void host_parallel(size_t n, u64 bytes, F fn)
{
    unsigned t = (unsigned)env_int("YEAST_HOST_THREADS", 8), hw = std::thread::hardware_concurrency();
    if (hw && t > hw)
        t = hw;
    if (bytes < ((u64)8 << 20) || n < 64 || t < 2) {
        fn((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    unsigned started = 0;
    try {
        for (; started < t; started++)
            th.emplace_back(fn, n * started / t, n * (started + 1) / t);
    } catch (...) { /* no more threads to be had: this one does the rest */
    }
    for (unsigned k = started; k < t; k++)
        fn(n * k / t, n * (k + 1) / t);
    for (std::thread &x : th)
        x.join();
}

/* one large memcpy shared by the same threads (pieces of whole pages) */
static void host_parallel_bytes(u8 *to, const u8 *from, size_t n)
{
    unsigned t = (unsigned)env_int("YEAST_HOST_THREADS", 8), hw = std::thread::hardware_concurrency();
    if (hw && t > hw)
        t = hw;
    if (n < ((size_t)2 << 20) || t < 2) {
        memcpy(to, from, n);
        return;
    }
    size_t piece = round_up((n + t - 1) / t, 4096);
    std::vector<std::thread> th;
    size_t at = piece; /* this thread takes the first piece */
    try {
        for (; at < n; at += piece) {
            size_t k = n - at < piece ? n - at : piece;
            th.emplace_back([=] { memcpy(to + at, from + at, k); });
        }
    } catch (...) {
        if (at < n)
            memcpy(to + at, from + at, n - at);
    }
    memcpy(to, from, piece < n ? piece : n);
    for (std::thread &x : th)
        x.join();
}

/* Decode n host streams; fills rec[] and, for cmp == 0, out[i] (malloc'd).
   expected/have: compare mode inputs (cmp != 0, n == 1 in practice). */
static int run_host_batch_impl(size_t n, void const *const *src, const size_t *len, std::vector<ys_record> &rec,
                               void **out, unsigned check_kind, int cmp, const void *expected, size_t have);

/* On failure nothing is left with the caller: copies in flight are waited for, the buffers handed out so far freed. */
static int run_host_batch(size_t n, void const *const *src, const size_t *len, std::vector<ys_record> &rec,
                          void **out, unsigned check_kind, int cmp, const void *expected, size_t have)
{
    /* out[] is a pure output: whatever the caller left in it is never looked at (and never freed) */
    if (out)
        for (size_t i = 0; i < n; i++)
            out[i] = nullptr;
    int e = run_host_batch_impl(n, src, len, rec, out, check_kind, cmp, expected, have);
    if (e && out) {
        (void)hipStreamSynchronize(nullptr);
        for (size_t i = 0; i < n; i++) {
            free(out[i]);
            out[i] = nullptr;
        }
    }
    return e;
}

static int run_host_batch_impl(size_t n, void const *const *src, const size_t *len, std::vector<ys_record> &rec,
                               void **out, unsigned check_kind, int cmp, const void *expected, size_t have)
{
    Device *dp;
    if (device_get(&dp))
        return -1;
    Device &d = *dp;
    hipStream_t st = nullptr;
    rec.assign(n, ys_record{});
    std::vector<u64> soff(n + 1), doff(n + 1);
    std::vector<size_t> todo(n);
    std::vector<u64> cap(n), need(n, 0); /* need: what a stream was last found to need at least */
    for (size_t i = 0; i < n; i++) {
        todo[i] = i;
        if (cmp)
            cap[i] = have;
        else {
            /* the first meta-block's length, or four times the compressed size (text compresses about
               3:1) if that is more: streams of several meta-blocks then rarely need a second pass */
            u64 g = first_mlen_guess((const u8 *)src[i], len[i]);
            u64 c = g ? g : 65536;
            if (c < 4 * (u64)len[i])
                c = 4 * (u64)len[i];
            if (c < 4096)
                c = 4096;
            cap[i] = c;
        }
        if (out)
            out[i] = nullptr;
    }
    int lanes = env_int("YEAST_FORCE_LANES", 0); /* enqueue() chooses by n; the override is for the tests */

    while (!todo.empty()) {
        /* One pass takes as many of the streams still to do as fit a share of the device memory that is free: the
           capacities come from unverified headers (up to 16 MiB per meta-block), and one arena for a whole batch of
           such claims may not exist.  A stream that does not fit by itself is out of memory (code 1), as it would
           be for the reference's realloc(); the others are decoded. */
        size_t m = todo.size();
        std::vector<size_t> later;
        if (!cmp) {
            size_t freeb = 0, totalb = 0;
            u64 budget = (u64)16 << 30;
            if (hipMemGetInfo(&freeb, &totalb) == hipSuccess) {
                u64 b = ((u64)freeb + d.dst.cap) / 5 * 3;
                budget = b < budget ? b : budget;
            }
            if (budget < ((u64)64 << 20))
                budget = (u64)64 << 20;
            if (env_int("YEAST_HOST_BUDGET_MB", 0) > 0) /* for the tests */
                budget = (u64)env_int("YEAST_HOST_BUDGET_MB", 0) << 20;
            /* a few streams: each gets a generous region whatever its header says (a stream of many meta-blocks,
               or one that copies itself up from a small start, otherwise comes back for pass after pass, each
               decoding from the beginning; address space in HBM costs nothing until it is written) */
            if (m < 64) {
                u64 fl = ((u64)256 << 20) / m, b2 = budget / (2 * m);
                fl = fl > ((u64)32 << 20) ? (u64)32 << 20 : fl;
                fl = (fl > b2 ? b2 : fl) & ~(u64)15;
                if (env_int("YEAST_HOST_FLOOR", 1))
                    for (size_t k = 0; k < m; k++)
                        if (cap[todo[k]] < fl)
                            cap[todo[k]] = fl;
            }
            u64 sum = 0;
            size_t take = 0;
            std::vector<size_t> now;
            for (size_t k = 0; k < m; k++) {
                size_t i = todo[k];
                u64 c = round_up(cap[i], 16);
                if (c > budget && need[i] && need[i] <= (budget & ~(u64)15)) /* the doubling overshot: what fits */
                    c = cap[i] = budget & ~(u64)15;
                if (c > budget) { /* not even alone */
                    rec[i] = ys_record{};
                    rec[i].rc = YS_NOMEM;
                    continue;
                }
                if (sum + c > budget) {
                    later.push_back(i);
                    continue;
                }
                sum += c;
                now.push_back(i);
                take++;
            }
            todo.swap(now);
            m = take;
            if (m == 0) {
                todo.swap(later);
                continue;
            }
        }
        u64 stot = 0, dtot = 0;
        for (size_t k = 0; k < m; k++) {
            soff[k] = stot;
            stot += len[todo[k]];
            doff[k] = dtot;
            dtot += cmp ? cap[todo[k]] : round_up(cap[todo[k]], 16);
        }
        soff[m] = stot;
        doff[m] = dtot;
        if (cmp)
            doff[m] = doff[0] + have;
        if (grow(d.src, stot + YS_SRC_PAD + 64) || grow(d.dst, dtot + 64) || grow(d.soff, (m + 1) * 8) ||
            grow(d.doff, (m + 1) * 8) || grow(d.rec, m * sizeof(ys_record)))
            return -1;
        /* the inputs go over back to back: packed into pinned memory when there is much of them */
        std::vector<u8> pack_small;
        u8 *pack;
        if (stot >= ((u64)8 << 20) && grow_pinned(d.pin_in, stot + YS_SRC_PAD) == 0)
            pack = (u8 *)d.pin_in.p;
        else {
            pack_small.resize(stot + YS_SRC_PAD);
            pack = pack_small.data();
        }
        host_parallel(m, stot, [&](size_t k0, size_t k1) {
            for (size_t k = k0; k < k1; k++)
                if (len[todo[k]])
                    memcpy(pack + soff[k], src[todo[k]], len[todo[k]]);
        });
        memset(pack + stot, 0, YS_SRC_PAD);
        HIPCK(hipMemcpyAsync(d.src.p, pack, stot + YS_SRC_PAD, hipMemcpyHostToDevice, st));
        HIPCK(hipMemcpyAsync(d.soff.p, soff.data(), (m + 1) * 8, hipMemcpyHostToDevice, st));
        HIPCK(hipMemcpyAsync(d.doff.p, doff.data(), (m + 1) * 8, hipMemcpyHostToDevice, st));
        if (cmp && have)
            HIPCK(hipMemcpyAsync(d.dst.p, expected, have, hipMemcpyHostToDevice, st));
        if (enqueue(d, (const u8 *)d.src.p, (const u64 *)d.soff.p, (u8 *)d.dst.p, (const u64 *)d.doff.p,
                    (ys_record *)d.rec.p, m, check_kind, lanes, cmp, st))
            return -1;
        std::vector<ys_record> r(m);
        HIPCK(hipMemcpyAsync(r.data(), d.rec.p, m * sizeof(ys_record), hipMemcpyDeviceToHost, st));
        HIPCK(hipStreamSynchronize(st));

        /* many outputs: the whole output arena comes back in one copy through pinned memory and is
           cut up on the host (one copy per stream costs more in calls than in bytes) */
        const u8 *bulk = nullptr;
        if (!cmp && out && m >= 64 && dtot <= ((u64)8 << 30) && grow_pinned(d.pin, dtot) == 0) {
            HIPCK(hipMemcpyAsync(d.pin.p, d.dst.p, dtot, hipMemcpyDeviceToHost, st));
            HIPCK(hipStreamSynchronize(st));
            bulk = (const u8 *)d.pin.p;
        }
        std::vector<size_t> again;
        for (size_t k = 0; k < m; k++) {
            size_t i = todo[k];
            if (r[k].rc == YS_CAPACITY) {
                need[i] = r[k].check;
                u64 c2 = cap[i] * 2;
                cap[i] = need[i] > c2 ? need[i] : c2;
                again.push_back(i);
                continue;
            }
            if (r[k].rc == YS_WORKSPACE) { /* cannot happen after the worst-case retry pass */
                snprintf(t_err, sizeof(t_err), "libyeast_mi355x: internal: table arena exhausted");
                fprintf(stderr, "%s\n", t_err);
                return -1;
            }
            rec[i] = r[k];
            if (!cmp && out && r[k].got && !bulk) {
                out[i] = malloc(r[k].got);
                if (!out[i])
                    return 1;
                const u64 big = (u64)env_int("YEAST_D2H_PIN_KB", 1024) << 10;
                if (big && r[k].got >= big && grow_pinned(d.pin, r[k].got) == 0) {
                    /* a large output: over the link into pinned memory at the link's rate, then into the caller's
                       fresh pages by several threads (a copy straight into pageable memory has the runtime pin
                       those pages first, one fault after the other) */
                    HIPCK(hipMemcpyAsync(d.pin.p, (u8 *)d.dst.p + doff[k], r[k].got, hipMemcpyDeviceToHost, st));
                    HIPCK(hipStreamSynchronize(st));
                    host_parallel_bytes((u8 *)out[i], (const u8 *)d.pin.p, r[k].got);
                } else
                    HIPCK(hipMemcpyAsync(out[i], (u8 *)d.dst.p + doff[k], r[k].got, hipMemcpyDeviceToHost, st));
            }
        }
        if (bulk) { /* one malloc'd buffer per stream, as the single call returns it */
            std::atomic<int> nomem{0};
            host_parallel(m, dtot, [&](size_t k0, size_t k1) {
                for (size_t k = k0; k < k1; k++) {
                    if (r[k].rc == YS_CAPACITY || !r[k].got)
                        continue;
                    void *o = malloc(r[k].got);
                    if (!o) {
                        nomem = 1;
                        return;
                    }
                    memcpy(o, bulk + doff[k], r[k].got);
                    out[todo[k]] = o;
                }
            });
            if (nomem)
                return 1;
        }
        HIPCK(hipStreamSynchronize(st));
        again.insert(again.end(), later.begin(), later.end());
        todo.swap(again);
    }
    return 0;
}

extern "C" int yeast(void **dest, size_t *got, void const *source, size_t *len, int cmp)
{
    std::lock_guard<std::mutex> lk(g_mu);
    std::vector<ys_record> rec;
    void *out = nullptr;
    const void *srcs[1] = {source};
    size_t lens[1] = {*len};
    const void *expected = cmp && *got ? *dest : nullptr;
    size_t have = cmp ? *got : 0;
    int e = run_host_batch(1, srcs, lens, rec, cmp ? nullptr : &out, 0, cmp ? 1 : 0, expected, have);
    if (e) {
        free(out);
        if (!cmp)
            *dest = nullptr;
        *got = 0;
        *len = 0;
        return e;
    }
    *len = (size_t)rec[0].used;
    *got = (size_t)rec[0].got;
    if (!cmp)
        *dest = out;
    if (yeast_verbosity > 0)
        fprintf(stderr, "yeast: window bits %u, %u meta-block%s, %zu bytes out, %zu bytes used%s\n",
                rec[0].flags & 255, rec[0].flags >> 8, (rec[0].flags >> 8) == 1 ? "" : "s", *got, *len,
                rec[0].rc ? " -- error: aborting" : "");
    return rec[0].rc;
}

extern "C" int yeast_batch(size_t n, void const *const *src, size_t *len, void **dest, size_t *got, int *rc,
                           unsigned check_kind, uint64_t *check)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (check_kind > 3) {
        snprintf(t_err, sizeof(t_err), "libyeast_mi355x: unknown check kind %u", check_kind);
        return -1;
    }
    std::vector<ys_record> rec;
    int e = run_host_batch(n, src, len, rec, dest, check_kind, 0, nullptr, 0);
    if (e)
        return e;
    for (size_t i = 0; i < n; i++) {
        len[i] = (size_t)rec[i].used;
        got[i] = (size_t)rec[i].got;
        rc[i] = rec[i].rc;
        if (check && check_kind)
            check[i] = rec[i].check;
    }
    return 0;
}

/* ---- check functions over host buffers --------------------------------------- */

static u64 host_check(const void *buf, size_t len, unsigned kind, u64 seed)
{
    std::lock_guard<std::mutex> lk(g_mu);
    Device *d;
    u64 v = 0;
    bool ok = device_get(&d) == 0 && grow(d->scratch, len + 64) == 0 && grow(d->rec, 64) == 0;
    if (ok && len)
        ok = hipMemcpy(d->scratch.p, buf, len, hipMemcpyHostToDevice) == hipSuccess;
    if (ok)
        ok = ys_launch_check_buf((const u8 *)d->scratch.p, len, kind, seed, (u64 *)d->rec.p, nullptr) == 0;
    if (ok)
        ok = hipMemcpy(&v, d->rec.p, 8, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) { /* no in-band error value in the reference's signatures: 0, and yeast_check_error() tells */
        fprintf(stderr, "libyeast_mi355x: check value could not be computed on the GPU (%s)\n", t_err);
        t_check_failed++;
        return 0;
    }
    return v;
}

/* SHA-256 of n host buffers in one launch (one lane per buffer); digests to out[32 * i ..].
   Returns 0, or -1 when the GPU path is unavailable. */
extern "C" int yeast_sha256_batch(size_t n, void const *const *buf, const size_t *len, unsigned char *out)
{
    std::lock_guard<std::mutex> lk(g_mu);
    Device *d;
    if (n == 0)
        return 0;
    if (n > 0x7fffffffu || device_get(&d))
        return -1;
    std::vector<u64> meta(2 * n); /* offsets, then lengths */
    u64 tot = 0;
    for (size_t i = 0; i < n; i++) {
        meta[i] = tot;
        meta[n + i] = len[i];
        tot += len[i];
    }
    size_t meta_at = (tot + 63) & ~(size_t)63, out_at = meta_at + 16 * n;
    if (grow(d->scratch, out_at + 32 * n))
        return -1;
    u8 *base = (u8 *)d->scratch.p;
    for (size_t i = 0; i < n; i++)
        if (len[i])
            HIPCK(hipMemcpyAsync(base + meta[i], buf[i], len[i], hipMemcpyHostToDevice, nullptr));
    HIPCK(hipMemcpyAsync(base + meta_at, meta.data(), 16 * n, hipMemcpyHostToDevice, nullptr));
    int e = ys_launch_sha256(base, (const u64 *)(base + meta_at), (const u64 *)(base + meta_at) + n, (unsigned)n,
                             base + out_at, nullptr);
    if (e)
        return fail("sha256 launch", (hipError_t)e);
    HIPCK(hipMemcpy(out, base + out_at, 32 * n, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int yeast_sha256(const void *buf, size_t len, unsigned char out[32])
{
    const void *b[1] = {buf};
    size_t l[1] = {len};
    return yeast_sha256_batch(1, b, l, out);
}

extern "C" uint32_t crc32c(uint32_t crc, void const *buf, size_t len)
{
    return (uint32_t)host_check(buf, len, YEAST_CHECK_CRC32C, crc);
}
extern "C" unsigned int XXH32(const void *input, size_t len, unsigned int seed)
{
    return (unsigned int)host_check(input, len, YEAST_CHECK_XXH32, seed);
}
extern "C" unsigned long long XXH64(const void *input, size_t len, unsigned long long seed)
{
    return host_check(input, len, YEAST_CHECK_XXH64, seed);
}

/* ---- streaming XXH32 / XXH64: xxhash.h:139-167, xxhash.c:588-960 ------------------------------

   The state is the caller's (same sizes as the reference's opaque types: 6 and 11 long longs); it holds
   the four accumulators, the bytes of an unfinished stripe and the totals.  Bytes are only moved on the
   host; the accumulators are advanced and the digest is folded on the GPU. */
struct XxhState32 { /* 48 bytes */
    u64 total;
    u32 seed, v[4];
    u8 mem[16];
    u32 memsize;
};
struct XxhState64 { /* 88 bytes */
    u64 total, seed, v[4];
    u8 mem[32];
    u32 memsize, pad;
};
static_assert(sizeof(XxhState32) == sizeof(XXH32_state_t) && sizeof(XxhState64) == sizeof(XXH64_state_t), "streaming state layout");

extern "C" XXH32_state_t *XXH32_createState(void) { return (XXH32_state_t *)malloc(sizeof(XxhState32)); }
extern "C" XXH64_state_t *XXH64_createState(void) { return (XXH64_state_t *)malloc(sizeof(XxhState64)); }
extern "C" int XXH32_freeState(XXH32_state_t *s) { free(s); return 0; }
extern "C" int XXH64_freeState(XXH64_state_t *s) { free(s); return 0; }

extern "C" int XXH32_reset(XXH32_state_t *sp, unsigned seed) /* xxhash.c:588-599 */
{
    XxhState32 *s = (XxhState32 *)sp;
    s->seed = seed;
    s->v[0] = seed + 2654435761u + 2246822519u;
    s->v[1] = seed + 2246822519u;
    s->v[2] = seed;
    s->v[3] = seed - 2654435761u;
    s->total = 0;
    s->memsize = 0;
    return 0;
}
extern "C" int XXH64_reset(XXH64_state_t *sp, unsigned long long seed) /* xxhash.c:601-612 */
{
    XxhState64 *s = (XxhState64 *)sp;
    s->seed = seed;
    s->v[0] = seed + 11400714785074694791ull + 14029467366897019727ull;
    s->v[1] = seed + 14029467366897019727ull;
    s->v[2] = seed;
    s->v[3] = seed - 11400714785074694791ull;
    s->total = 0;
    s->memsize = 0;
    return 0;
}

/* advance acc[4] over the whole stripes of mem[0..memsize) ++ in[0..len); the rest goes back to mem */
static int xxh_update(unsigned is64, u64 *acc, u8 *mem, u32 *memsize, const void *in, size_t len)
{
    const size_t stripe = is64 ? 32 : 16;
    if (*memsize + len < stripe) { /* xxhash.c:627-632, 781-786 */
        if (len)
            memcpy(mem + *memsize, in, len);
        *memsize += (u32)len;
        return 0;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    Device *d;
    if (device_get(&d))
        return 1;
    size_t have = *memsize + len, nstripe = have / stripe, rest = have - nstripe * stripe;
    size_t data_at = 64;
    if (grow(d->scratch, data_at + have + 64))
        return 1;
    u8 *base = (u8 *)d->scratch.p;
    HIPCK(hipMemcpyAsync(base, acc, 32, hipMemcpyHostToDevice, nullptr));
    if (*memsize)
        HIPCK(hipMemcpyAsync(base + data_at, mem, *memsize, hipMemcpyHostToDevice, nullptr));
    HIPCK(hipMemcpyAsync(base + data_at + *memsize, in, len, hipMemcpyHostToDevice, nullptr));
    int e = ys_launch_xxh_update(is64, (u64 *)base, base + data_at, nstripe, nullptr);
    if (e)
        return fail("xxh update launch", (hipError_t)e), 1;
    HIPCK(hipMemcpy(acc, base, 32, hipMemcpyDeviceToHost));
    if (rest)
        memcpy(mem, (const u8 *)in + (len - rest), rest);
    *memsize = (u32)rest;
    return 0;
}
static int xxh_digest(unsigned is64, const u64 *acc, u64 total, u64 seed, const u8 *mem, u32 memsize, u64 *out)
{
    std::lock_guard<std::mutex> lk(g_mu);
    Device *d;
    if (device_get(&d) || grow(d->scratch, 256))
        return 1;
    u8 *base = (u8 *)d->scratch.p;
    HIPCK(hipMemcpyAsync(base, acc, 32, hipMemcpyHostToDevice, nullptr));
    HIPCK(hipMemcpyAsync(base + 64, mem, 32, hipMemcpyHostToDevice, nullptr));
    int e = ys_launch_xxh_digest(is64, (const u64 *)base, total, seed, base + 64, memsize, (u64 *)(base + 128), nullptr);
    if (e)
        return fail("xxh digest launch", (hipError_t)e), 1;
    HIPCK(hipMemcpy(out, base + 128, 8, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int XXH32_update(XXH32_state_t *sp, const void *input, size_t len) /* xxhash.c:614-702 */
{
    XxhState32 *s = (XxhState32 *)sp;
    if (input == nullptr)
        return 1;
    s->total += len;
    u64 acc[4] = {s->v[0], s->v[1], s->v[2], s->v[3]};
    int e = xxh_update(0, acc, s->mem, &s->memsize, input, len);
    for (int k = 0; k < 4; k++)
        s->v[k] = (u32)acc[k];
    return e ? 1 : 0;
}
extern "C" int XXH64_update(XXH64_state_t *sp, const void *input, size_t len) /* xxhash.c:768-856 */
{
    XxhState64 *s = (XxhState64 *)sp;
    if (input == nullptr)
        return 1;
    s->total += len;
    return xxh_update(1, s->v, s->mem, &s->memsize, input, len) ? 1 : 0;
}
extern "C" unsigned int XXH32_digest(const XXH32_state_t *sp) /* xxhash.c:706-765 */
{
    const XxhState32 *s = (const XxhState32 *)sp;
    u64 acc[4] = {s->v[0], s->v[1], s->v[2], s->v[3]}, out = 0;
    u8 mem[32] = {0};
    memcpy(mem, s->mem, 16);
    if (xxh_digest(0, acc, s->total, s->seed, mem, s->memsize, &out)) {
        fprintf(stderr, "libyeast_mi355x: XXH32_digest could not run on the GPU (%s)\n", t_err);
        t_check_failed++;
        return 0;
    }
    return (unsigned int)out;
}
extern "C" unsigned long long XXH64_digest(const XXH64_state_t *sp) /* xxhash.c:860-940 */
{
    const XxhState64 *s = (const XxhState64 *)sp;
    u64 out = 0;
    if (xxh_digest(1, s->v, s->total, s->seed, s->mem, s->memsize, &out)) {
        fprintf(stderr, "libyeast_mi355x: XXH64_digest could not run on the GPU (%s)\n", t_err);
        t_check_failed++;
        return 0;
    }
    return out;
}
