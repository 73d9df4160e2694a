/*
 * yeast_stream.h -- per-stream decode shared by the kernels of yeast_decode.hip and yeast_lane.hip:
 * constant tables, decoder state, byte movers, the batch command loops, meta-block and stream
 * drivers.  Reference lines are cited per function.
 */
#ifndef YEAST_STREAM_H
#define YEAST_STREAM_H
#include "yeast_dev.h"

#ifndef YS_MIN_WAVES
#define YS_MIN_WAVES 4
#endif

static __constant__ u8 c_ctx_lut[768]; /* lut0 | lut1 | lut2 of yeast.c:700-740, filled by the host */
static __constant__ u64 c_cmd_tab[704]; /* per insert-and-copy symbol: insert base | extra << 16 | copy base << 24 |
                                    copy extra << 40  (insert_length()/copy_length(), yeast.c:652-693) */
static __constant__ u32 c_dict_tab[32]; /* per word length: DOFFSET | NDBITS << 24 (yeast.c:911-919) */

/* LDS image shared by the groups of a block */
struct Shared {
    u64 cmd[704];
    u8 lut[768];
    u32 dict[32];
    u32 xf[128]; /* first word of each transform record: op | omit << 8 | prefix len << 16 | suffix len << 24 */
    u64 xpre[128], xsuf[128]; /* prefix and suffix bytes of each transform, little endian */
    u32 xbase[121 * 21];      /* expanded dictionary: offset of (transform, word length) */
};

DEV void shared_fill(Shared &sh, const u8 *xforms, const u32 *xbase = nullptr)
{
    if (xbase)
        for (u32 i = threadIdx.x; i < 121 * 21; i += blockDim.x)
            sh.xbase[i] = xbase[i];
    for (u32 i = threadIdx.x; i < 704; i += blockDim.x)
        sh.cmd[i] = c_cmd_tab[i];
    for (u32 i = threadIdx.x; i < 768; i += blockDim.x)
        sh.lut[i] = c_ctx_lut[i];
    if (threadIdx.x < 32)
        sh.dict[threadIdx.x] = c_dict_tab[threadIdx.x];
    if (threadIdx.x < 128) {
        u32 i = threadIdx.x;
        const u32 *rec = (const u32 *)(xforms + 20 * i);
        sh.xf[i] = i < 121 ? rec[0] : 0;
        sh.xpre[i] = i < 121 ? (u64)rec[1] | ((u64)rec[2] << 32) : 0;
        sh.xsuf[i] = i < 121 ? (u64)rec[3] | ((u64)rec[4] << 32) : 0;
    }
    __syncthreads();
}

/* ------------------------------------------------------------------ state */

/* per-wavefront LDS of the batch kernel: the copy queues, the literal table map and the output
   staging rings.  Every array is [index][stream] so that the 64 streams of a wavefront touch 64
   different banks whatever their indices are. */
#define FQ 8
#ifndef YS_HOLD
#define YS_HOLD 12 /* trips a literal may wait for its context bytes before a drain is forced */
#endif
#define RB 256        /* bytes of output staged per stream (the newest RB bytes of its output) */
#define RU (RB / 8)   /* the ring is kept in 8-byte units */
#define RB_SLACK 48   /* a drain is forced when fewer bytes than this are free */
template <int S> /* S streams share the wavefront */
struct WaveLds {
    u32 qa[FQ * S]; /* destination, bytes from the start of the meta-block's output */
    u32 qb[FQ * S]; /* bytes to produce | kind << 28 */
    u32 qc[FQ * S]; /* copy: distance; dictionary: transform << 25 | word bytes << 20 | word offset */
    u16 lt[64 * S]; /* [context][stream]: byte offset of the literal table >> 4 */
    u64 ring[RU * S]; /* [unit][stream] */
};

/* byte offset of output address a (low bits) in the ring array of a wavefront with S streams */
template <int S>
DEV u32 ring_at(u32 slot, u32 a) { return ((((a >> 3) & (RU - 1)) * S + slot) << 3) | (a & 7); }

/* n (1..8) bytes of v to output address a, staged: a read-modify-write of the one or two units the
   bytes fall in (bytes of v beyond n are ignored; neighbouring bytes in the units are kept) */
template <int S>
DEV void ring_put(LAS u8 *rb, u32 slot, u32 a, u64 v, u32 n)
{
    u32 s = (a & 7) * 8;
    u64 m = n >= 8 ? ~0ull : ~(~0ull << (8 * n));
    u32 u0 = (a >> 3) & (RU - 1), u1 = (u0 + 1) & (RU - 1);
    LAS u64 *w0 = (LAS u64 *)(rb + ((u0 * S + slot) << 3));
    u64 m0 = m << s;
    *w0 = (*w0 & ~m0) | ((v << s) & m0);
    u64 m1 = s ? m >> (64 - s) : 0;
    if (m1) {
        LAS u64 *w1 = (LAS u64 *)(rb + ((u1 * S + slot) << 3));
        *w1 = (*w1 & ~m1) | ((v >> (64 - s)) & m1);
    }
}

/* the 8 staged bytes from output address a on */
template <int S>
DEV u64 ring_get(LAS u8 *rb, u32 slot, u32 a)
{
    u32 s = (a & 7) * 8;
    u32 u0 = (a >> 3) & (RU - 1), u1 = (u0 + 1) & (RU - 1);
    u64 w0 = *(LAS u64 *)(rb + ((u0 * S + slot) << 3)), w1 = *(LAS u64 *)(rb + ((u1 * S + slot) << 3));
    return s ? (w0 >> s) | (w1 << (64 - s)) : w0;
}

/* one complete 64-byte line from the ring to its place in the output region */
template <int S>
DEV void ring_flush_line(LAS u8 *rb, u32 slot, u8 *line)
{
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
    u32 u = ((u32)(u64)line >> 3) & (RU - 1);
    u64 d[8];
#pragma unroll
    for (u32 j = 0; j < 8; j++)
        d[j] = *(LAS u64 *)(rb + (((u + j) * S + slot) << 3));
#pragma unroll
    for (u32 j = 0; j < 4; j++) {
        u64x2 v = {d[2 * j], d[2 * j + 1]};
        *(GAS u64x2 *)(line + 16 * j) = v;
    }
}

struct Blk { /* one block category: yeast.c:112-123 */
    u32 num, type, prev, left;
    u32 t_type, t_count; /* byte offsets of the two block-switch tables from ws */
};

struct Ctx {
    Rd r;
    const u8 *src;
    u8 *out;  /* start of this stream's output region */
    u8 *outp; /* next output byte: out + got */
    u64 cap;
    u32 wsize;
    u32 r0, r1, r2, r3; /* last four distances, r0 most recent (ring[] yeast.c:126) */
    u32 postfix, direct;
    Blk L, I, D;
    u32 lit_codes, dist_codes;
    u8 *ws;
    u32 arena_size;
    u32 p1, p2;
    u32 nmeta;
    u64 need; /* output bytes required when rc == YS_CAPACITY */
    const u8 *dict, *xf, *xdict;
    u32 rootcaps; /* root table widths (Hdr.rootcaps) */
    const Shared *sh;
    void *wl; /* WaveLds<S> of this wavefront (batch kernel only) */
#ifdef YS_PROFILE
    u64 t_hdr, t_loop, t_drain, t_look, t_dload;
    u64 t_s0, t_s2, t_s3, t_s4, t_s5, t_mark;
    u32 n_trips, n_drains, n_need, n_qfull, n_rfull, n_ents;
#endif
};

DEV const u16 *tab_at(const Ctx &c, u32 off) { return (const u16 *)(c.ws + off); }
DEV u64 got_of(const Ctx &c) { return (u64)(c.outp - c.out); }

/* one category's NBLTYPES header: yeast.c:1079-1118 */
DEV void read_block_header(Hdr &h, Blk &k)
{
    k.prev = 1;
    k.type = 0;
    k.left = 0xffffffffu;
    k.t_type = k.t_count = YS_WS_ARENA;
    k.num = read_count256(h.r);
    if (k.num > 1) {
        h = read_code(h, k.num + 2);
        k.t_type = h.tab;
        h = read_code(h, 26);
        k.t_count = h.tab;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        k.left = read_block_count(h.r, (const u16 *)(h.ws + k.t_count));
    }
}

/* block switch: yeast.c:1168-1180 */
DEV void switch_block(Ctx &c, Blk &k)
{
    u32 n = dec_sym(c.r, tab_at(c, k.t_type));
    n = n > 1 ? n - 2 : n ? (k.type + 1 == k.num ? 0 : k.type + 1) : k.prev;
    if (n >= k.num)
        n = 0; /* only reachable after a read failure; keeps table indices in range */
    k.prev = k.type;
    k.type = n;
    k.left = read_block_count(c.r, tab_at(c, k.t_count));
}

/* context_id(): yeast.c:698-755 */
DEV u32 context_id(const Ctx &c, u32 p1, u32 p2, u32 mode)
{
    const u8 *lut = c.sh->lut;
    switch (mode) {
    case 0: return p1 & 0x3f;
    case 1: return p1 >> 2;
    case 2: return lut[p1] | lut[256 + p2];
    default: return ((u32)lut[512 + p1] << 3) | lut[512 + p2];
    }
}

/* ------------------------------------------------------------ byte movers */

template <int G>
DEV u32 group_min(u32 v)
{
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
        u32 t = __shfl_xor(v, o, G);
        v = t < v ? t : v;
    }
    return v;
}

/* LZ77 copy of n bytes from distance d (yeast.c:1285-1293): byte-serial
   semantics, so an overlapping copy repeats its first d bytes.  ctx: also
   fetch the last two bytes of the copy (from its source) for the literal
   context of the next command. */
template <int G, bool CMP>
DEV void emit_copy(Ctx &c, u32 n, u32 d, u32 sub, bool ctx)
{
    u8 *dst = c.outp;
    const u8 *from = dst - d;
    if (CMP) {
        /* every byte is compared with the expected data itself d bytes back */
        u32 bad = 0xffffffffu;
        for (u32 i = sub; i < n; i += G)
            if (gld8(dst + i) != gld8(from + i) && i < bad)
                bad = i;
        bad = group_min<G>(bad);
        if (bad != 0xffffffffu) {
            c.outp += bad;
            rd_fail(c.r, YS_MISMATCH);
            return;
        }
        c.outp += n;
        if (ctx) {
            c.p1 = gld8(c.outp - 1);
            c.p2 = gld8(c.outp - 2);
        }
        return;
    }
    if (d >= n || (G == 1 && d >= 8)) {
        /* disjoint (or, for a single lane, far enough apart to go in order):
           8-byte units spread over the lanes, then the tail */
        u32 i1 = n - 1, i2 = n - 2;
        if (G == 1 && d < n) {
            i1 %= d;
            i2 %= d;
        }
        if (ctx) {
            c.p1 = gld8(from + i1);
            c.p2 = gld8(from + i2);
        }
        u32 n8 = n >> 3;
        for (u32 j = sub; j < n8; j += G)
            gst64u(dst + 8 * j, gld64u(from + 8 * j));
        for (u32 i = (n8 << 3) + sub; i < n; i += G)
            gst8(dst + i, gld8(from + i));
    } else { /* overlapping: byte i repeats byte i mod d of the source */
        if (ctx) {
            c.p1 = gld8(from + (n - 1) % d);
            c.p2 = gld8(from + (n - 2) % d);
        }
        u32 m = sub % d, step = G % d;
        for (u32 i = sub; i < n; i += G) {
            gst8(dst + i, gld8(from + m));
            m += step;
            if (m >= d)
                m -= d;
        }
    }
    c.outp += n;
}

/* The same copy for the flat loop when there is slack behind the destination:
   whole 8-byte units only; up to 7 bytes beyond the end are scribbled on and
   overwritten by whatever is decoded next (never read before that). */
template <int G>
DEV void emit_copy_wild(Ctx &c, u32 n, u32 d, u32 sub, bool ctx)
{
    u8 *dst = c.outp;
    const u8 *from = dst - d;
    if (d >= n || (G == 1 && d >= 8)) {
        if (ctx) {
            if (d >= n) {
                u32 v = gld8(from + n - 2) | (gld8(from + n - 1) << 8);
                c.p2 = v & 255;
                c.p1 = v >> 8;
            } else {
                c.p1 = gld8(from + (n - 1) % d);
                c.p2 = gld8(from + (n - 2) % d);
            }
        }
        for (u32 j = sub * 8; j < n; j += G * 8)
            gst64u(dst + j, gld64u(from + j));
    } else {
        if (ctx) {
            c.p1 = gld8(from + (n - 1) % d);
            c.p2 = gld8(from + (n - 2) % d);
        }
        u32 m = sub % d, step = G % d;
        for (u32 i = sub; i < n; i += G) {
            gst8(dst + i, gld8(from + m));
            m += step;
            if (m >= d)
                m -= d;
        }
    }
    c.outp += n;
}

/* A static dictionary reference resolved to pointers and lengths
   (dict_word(): yeast.c:908-949). */
struct DictRef {
    const u8 *w, *t;
    u32 op, plen, slen, wlen, wskip, copy, total, xform;
};

/* validation part of dict_word(); false = invalid reference (code 3) */
DEV bool dict_prepare(const Ctx &c, u32 copy, u64 id, DictRef &d)
{
    if (copy > 24 || copy < 4)
        return false; /* :921-924 */
    u32 de = c.sh->dict[copy];
    u32 nd = de >> 24;
    u64 xform = id >> nd;
    if (xform >= 121)
        return false; /* :929-930 */
    u32 index = (u32)id & ((1u << nd) - 1);
    d.w = c.dict + (de & 0xffffff) + index * copy;
    d.t = c.xf + 20 * (u32)xform;
    d.xform = (u32)xform;
    u32 hdr = c.sh->xf[(u32)xform];
    u32 omit = (hdr >> 8) & 255;
    d.op = hdr & 255;
    d.plen = (hdr >> 16) & 255;
    d.slen = hdr >> 24;
    d.copy = copy;
    d.wlen = copy;
    d.wskip = 0;
    if (d.op == 3) { /* omit first */
        d.wlen = copy > omit ? copy - omit : 0;
        d.wskip = d.wlen ? omit : 0;
    } else if (d.op == 4)
        d.wlen = copy > omit ? copy - omit : 0;
    d.total = d.plen + d.wlen + d.slen;
    return true;
}

/* uppercase masks (UPPER(): yeast.c:809-827): bit j set -> word byte j is xor-ed with 32 / with 5 */
DEV void dict_masks(const DictRef &d, u32 &x32, u32 &x5)
{
    x32 = x5 = 0;
    if (d.op == 1 || d.op == 2) {
        u32 j = 0;
        do {
            u32 ch = gld8(d.w + j);
            if (ch < 192) {
                if (ch >= 97 && ch <= 122)
                    x32 |= 1u << j;
                j += 1;
            } else if (ch < 224) {
                if (j + 1 < d.copy)
                    x32 |= 1u << (j + 1);
                j += 2;
            } else {
                if (j + 2 < d.copy)
                    x5 |= 1u << (j + 2);
                j += 3;
            }
        } while (d.op == 2 && j < d.copy);
    }
}

DEV u32 dict_byte(const DictRef &d, u32 x32, u32 x5, u32 k)
{
    u32 v;
    if (k < d.plen)
        v = gld8(d.t + 4 + k);
    else if (k < d.plen + d.wlen) {
        u32 j = k - d.plen + d.wskip;
        v = gld8(d.w + j);
        v ^= ((x32 >> j) & 1) << 5;
        v ^= ((x5 >> j) & 1) * 5;
    } else
        v = gld8(d.t + 12 + k - d.plen - d.wlen);
    return v;
}

/* exactly n (0..8) bytes of v to p, with at most four stores and no loop */
DEV void put_le(u8 *p, u64 v, u32 n)
{
    if (n >= 8) {
        gst64u(p, v);
        return;
    }
    u32 o = 0;
    if (n & 4) {
        typedef u32 __attribute__((aligned(1))) u32u;
        *(GAS u32u *)p = (u32)v;
        o = 4;
    }
    if (n & 2) {
        typedef u16 __attribute__((aligned(1))) u16u;
        *(GAS u16u *)(p + o) = (u16)(v >> (8 * o));
        o += 2;
    }
    if (n & 1)
        gst8(p + o, (u32)(v >> (8 * o)) & 255);
}

/* Write a dictionary word with no byte loop and without touching a byte beyond its end.
   Every lane of the group issues the same stores. */
DEV void dict_write_exact(const DictRef &d, u8 *dst)
{
    u32 x32, x5;
    dict_masks(d, x32, x5);
    const u8 *w = d.w + d.wskip;
    u64 pre = gld64u(d.t + 4), suf = gld64u(d.t + 12);
    u64 w0 = gld64u(w), w1 = gld64u(w + 8), w2 = gld64u(w + 16);
    if (x32 | x5) { /* spread the per-byte masks (wskip is 0 for the uppercase transforms) */
        u64 m0 = 0, m1 = 0, m2 = 0;
        for (u32 j = 0; j < 8; j++) {
            m0 |= (u64)((((x32 >> j) & 1) << 5) | (((x5 >> j) & 1) * 5)) << (8 * j);
            m1 |= (u64)((((x32 >> (j + 8)) & 1) << 5) | (((x5 >> (j + 8)) & 1) * 5)) << (8 * j);
            m2 |= (u64)((((x32 >> (j + 16)) & 1) << 5) | (((x5 >> (j + 16)) & 1) * 5)) << (8 * j);
        }
        w0 ^= m0;
        w1 ^= m1;
        w2 ^= m2;
    }
    put_le(dst, pre, d.plen);
    u8 *wd = dst + d.plen;
    u32 wl = d.wlen;
    put_le(wd, w0, wl);
    if (wl > 8)
        put_le(wd + 8, w1, wl - 8);
    if (wl > 16)
        put_le(wd + 16, w2, wl - 16);
    put_le(wd + wl, suf, d.slen);
}

/* LZ77 copy with no byte tail loop and without touching a byte beyond dst + n. */
template <int G>
DEV void copy_exact(u8 *dst, u32 n, u32 d, u32 sub)
{
    const u8 *from = dst - d;
    if (d >= n || (G == 1 && d >= 8)) {
        if (n >= 8) {
            for (u32 j = sub * 8; j + 8 <= n; j += G * 8)
                gst64u(dst + j, gld64u(from + j));
            if ((n & 7) && (G == 1 || sub == ((n >> 3) & (G - 1))))
                gst64u(dst + n - 8, gld64u(from + n - 8));
        } else if (G == 1 || sub == 0) {
            u32 o = 0;
            if (n & 4) {
                typedef u32 __attribute__((aligned(1))) u32u;
                *(GAS u32u *)dst = *(const GAS u32u *)from;
                o = 4;
            }
            if (n & 2) {
                typedef u16 __attribute__((aligned(1))) u16u;
                *(GAS u16u *)(dst + o) = *(const GAS u16u *)(from + o);
                o += 2;
            }
            if (n & 1)
                gst8(dst + o, gld8(from + o));
        }
    } else { /* overlapping: byte i repeats byte i mod d of the source */
        u32 m = sub % d, step = G % d;
        for (u32 i = sub; i < n; i += G) {
            gst8(dst + i, gld8(from + m));
            m += step;
            if (m >= d)
                m -= d;
        }
    }
}

/* dict_word(): yeast.c:908-949 with the elementary transforms :799-879.
   Byte k of the result is computed independently by whichever lane owns it. */
template <int G, bool CMP>
DEV void emit_dict(Ctx &c, u32 copy, u64 id, u32 &mlen, u32 sub, bool ctx)
{
    if (copy > 24 || copy < 4) {
        rd_fail(c.r, YS_INVALID); /* :921-924 */
        return;
    }
    u32 de = c.sh->dict[copy];
    u32 nd = de >> 24;
    u64 xform = id >> nd;
    if (xform >= 121) {
        rd_fail(c.r, YS_INVALID); /* :929-930 */
        return;
    }
    u32 index = (u32)id & ((1u << nd) - 1);
    const u8 *w = c.dict + (de & 0xffffff) + index * copy;
    const u8 *t = c.xf + 20 * (u32)xform;
    u32 hdr = gld32(t);
    u32 op = hdr & 255, omit = (hdr >> 8) & 255, plen = (hdr >> 16) & 255, slen = hdr >> 24;
    u32 wlen = copy, wskip = 0;
    if (op == 3) { /* omit first */
        wlen = copy > omit ? copy - omit : 0;
        wskip = wlen ? omit : 0;
    } else if (op == 4)
        wlen = copy > omit ? copy - omit : 0;
    u32 total = plen + wlen + slen;
    if (total > mlen) {
        rd_fail(c.r, YS_INVALID); /* :1269 */
        return;
    }
    /* uppercase masks: bit j set -> word byte j is xor-ed with 32 / with 5 */
    u32 x32 = 0, x5 = 0;
    if (op == 1 || op == 2) {
        u32 j = 0;
        do {
            u32 ch = gld8(w + j);
            if (ch < 192) {
                if (ch >= 97 && ch <= 122)
                    x32 |= 1u << j;
                j += 1;
            } else if (ch < 224) {
                if (j + 1 < copy)
                    x32 |= 1u << (j + 1);
                j += 2;
            } else {
                if (j + 2 < copy)
                    x5 |= 1u << (j + 2);
                j += 3;
            }
        } while (op == 2 && j < copy);
    }
    auto byte_at = [&](u32 k) -> u32 {
        u32 v;
        if (k < plen)
            v = gld8(t + 4 + k);
        else if (k < plen + wlen) {
            u32 j = k - plen + wskip;
            v = gld8(w + j);
            v ^= ((x32 >> j) & 1) << 5;
            v ^= ((x5 >> j) & 1) * 5;
        } else
            v = gld8(t + 12 + k - plen - wlen);
        return v;
    };
    u8 *dst = c.outp;
    u32 bad = 0;
    for (u32 k = sub; k < total; k += G) {
        u32 v = byte_at(k);
        if (CMP)
            bad |= gld8(dst + k) != v;
        else
            gst8(dst + k, v);
    }
    if (CMP) {
        if (__any(bad)) { /* whole-wave vote: CMP is only instantiated with G == 64 */
            rd_fail(c.r, YS_MISMATCH); /* memcmp semantics: got stays at the start of the word :1271 */
            return;
        }
    }
    if (ctx) {
        if (total >= 2) {
            c.p1 = byte_at(total - 1);
            c.p2 = byte_at(total - 2);
        } else if (total == 1) {
            c.p2 = c.p1;
            c.p1 = byte_at(0);
        }
    }
    c.outp += total;
    mlen -= total;
}

/* stored block (yeast.c:1061-1070) */
template <int G, bool CMP>
DEV void emit_raw(Ctx &c, const u8 *from, u32 n, u32 sub)
{
    u8 *dst = c.outp;
    if (CMP) {
        u32 bad = 0;
        for (u32 i = sub; i < n; i += G)
            bad |= gld8(dst + i) != gld8(from + i);
        if (__any(bad)) {
            rd_fail(c.r, YS_MISMATCH); /* memcmp over the block: got unchanged :1063 */
            return;
        }
    } else {
        u32 n8 = n >> 3;
        for (u32 j = sub; j < n8; j += G)
            gst64u(dst + 8 * j, gld64u(from + 8 * j));
        for (u32 i = (n8 << 3) + sub; i < n; i += G)
            gst8(dst + i, gld8(from + i));
    }
    c.outp += n;
    /* literal context for what follows: the last two bytes of the block (n >= 1) */
    c.p2 = n >= 2 ? gld8(from + n - 2) : c.p1;
    c.p1 = gld8(from + n - 1);
}

/* the bytes a group has just written become visible to its own later loads */
DEV void group_sync_mem() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

/* ------------------------------------------------ batch kernel command loop */
#ifdef YS_PROFILE
#define YS_MARK(field) do { u64 now_ = __builtin_readcyclecounter(); c.field += now_ - c.t_mark; c.t_mark = now_; } while (0)
#else
#define YS_MARK(field) do { } while (0)
#endif


struct FlatEnv {
    const u32 *itab, *dtab, *lit_tab;
    const u8 *mode;
    u32 offL0, offD0, cur_mode;
    bool ctx;
};

/* spread the low 8 bits of b to the low bit of 8 bytes */
DEV u64 spread8(u32 b)
{
    u64 x = b & 255;
    x = (x | (x << 28)) & 0x0000000f0000000full;
    x = (x | (x << 14)) & 0x0003000300030003ull;
    x = (x | (x << 7)) & 0x0101010101010101ull;
    return x;
}

/* The command loop of yeast.c:1166-1295 as a flat state machine.  Every trip performs exactly
 * one table lookup per stream (an insert-and-copy, literal or distance symbol, or the second
 * level of a long code), so the streams that share a wavefront stay converged on the expensive
 * part whatever mix of commands they are in.  The body is written as straight-line code with
 * selects: in a wavefront of 64 streams every kind of trip occurs on every pass anyway, so
 * branching around them would only add mask bookkeeping.  Window copies and dictionary words are
 * validated and accounted for where they are decoded, but their bytes are moved later: they go to
 * a short per-stream queue in LDS and the whole wavefront drains its queues together (the "second
 * pass" that resolves LZ77 and dictionary references).
 * SIMPLE: no block switching, one distance code (checked for the whole wavefront). */
template <int G, int S, bool SIMPLE>
DEV void run_flat(Ctx &c, u32 &mlen, u32 sub, const FlatEnv &fe)
{
    enum { ST_CMD = 0, ST_LIT = 1, ST_DIST = 2, ST_END = 3, ACT_NONE = 4, ST_COPY = 5 };
    enum { K_FAR = 0, K_BYTES = 1, K_SLOW = 2, K_DICT = 3 };
    Rd &r = c.r;
    WaveLds<S> *wl = (WaveLds<S> *)c.wl;
    const u32 slot = (threadIdx.x & 63) / G;
    LAS u8 *const rb = (LAS u8 *)wl->ring;
    const bool ctx = fe.ctx;
    const u64 *cmd_tab = c.sh->cmd;
    u32 cur_mode = fe.cur_mode;
    u32 st = ST_CMD;
    u32 offI = gld32(fe.itab), offL = fe.offL0, offD = fe.offD0;
    u32 off = offI;
    u32 insert = 0, copy = 0, implicit = 0;
    u32 link = 0;             /* root entry of a long code whose second lookup is this trip's work */
    /* last two bytes of the newest copy: raw load results still in flight (nothing is computed
       from them until the next lookup has returned), 1 = two bytes, 2 = one byte + pn_c */
    u32 pn = 0, pn_c = 0, pn_valid = 0;
    u8 *const mb_base = c.outp; /* queue entries hold offsets from here (mlen < 2^25) */
    /* Output is staged: bytes below fl are in the output region (final), bytes from fl up to
       outp live in the ring -- literals at once, queued copies when the queue is drained.
       outp - fl never exceeds RB. */
    u8 *fl = c.outp;
    u32 qn = 0, pend_rel = 0, stale = 0, held = 0;
    /* max distance into the window = min(bytes produced so far, window size), kept incrementally */
    u64 got0 = got_of(c);
    u32 maxd = got0 > c.wsize ? c.wsize : (u32)got0;
    const bool lds_lt = SIMPLE ? ctx : (ctx && c.L.num == 1);
    if (lds_lt) { /* the 64 table offsets of the only literal block type move to LDS */
        for (u32 n = sub; n < 64; n += G)
            wl->lt[n * S + slot] = (u16)(gld32(fe.lit_tab + n) >> 4);
    }
    u32 nxt_empty = 0;
    /* a refill in the middle of a trip: the prefetched word is in a register (pinned after the
       lookup); no new load is started here, so the top of the next trip never waits */
    auto refill_mid = [&]() {
        if (r.have <= 32) {
            if (nxt_empty) {
                r.nxt = gld32(r.p + r.idx);
                asm volatile("" : "+v"(r.nxt));
            }
            r.w |= (u64)r.nxt << r.have;
            u32 add = r.rem < 32 ? r.rem : 32;
            r.have += add;
            r.rem -= add;
            r.idx += add ? 1 : 0;
            nxt_empty = 1;
        }
    };
    auto next_lit_table = [&]() -> u32 {
        u32 id = (c.L.type << 6) + context_id(c, c.p1, c.p2, cur_mode);
        if (SIMPLE)
            return (u32)wl->lt[id * S + slot] << 4;
        return lds_lt ? (u32)wl->lt[id * S + slot] << 4 : gld32(fe.lit_tab + id);
    };
    auto staged = [&](const u8 *q) -> u32 { return rb[ring_at<S>(slot, (u32)(u64)q)]; };
    /* ring -> output region: whole 64-byte lines below lim; exact: every byte below lim */
    auto flush = [&](u8 *lim, bool exact) {
        u8 *lim64 = exact ? lim : (u8 *)((u64)lim & ~63ull);
        while (fl < lim64 && ((u64)fl & 63)) { /* ragged start: stream start, after a slow copy */
            if (G == 1 || sub == 0)
                gst8(fl, staged(fl));
            fl++;
        }
        while (fl + 64 <= lim64) {
            if (G == 1 || sub == 0)
                ring_flush_line<S>(rb, slot, fl);
            fl += 64;
        }
        if (exact)
            while (fl < lim) {
                if (G == 1 || sub == 0)
                    gst8(fl, staged(fl));
                fl++;
            }
    };
    /* the last two bytes of the output so far, when nothing is queued */
    auto last_two = [&]() {
        group_sync_mem(); /* bytes below fl are read back from the output region */
        u64 got = got_of(c);
        u32 b1 = 0, b2 = 0;
        if (got >= 1)
            b1 = c.outp - 1 >= fl ? staged(c.outp - 1) : gld8(c.outp - 1);
        if (got >= 2)
            b2 = c.outp - 2 >= fl ? staged(c.outp - 2) : gld8(c.outp - 2);
        c.p1 = b1;
        c.p2 = b2;
    };
    /* second pass over one stream's queue, entries in order, four source fetches per round trip */
    constexpr u32 DB = 4;
    auto drain = [&]() {
        for (u32 k0 = 0; k0 < qn; k0 += DB) {
            u32 ea[DB], eb[DB], ec[DB];
            u64 d0[DB], d1[DB], d2[DB];
#ifdef YS_PROFILE
            u64 tb0 = __builtin_readcyclecounter();
#endif
#pragma unroll
            for (u32 j = 0; j < DB; j++) {
                u32 k = k0 + j;
                ea[j] = eb[j] = ec[j] = 0;
                d0[j] = d1[j] = d2[j] = 0;
                if (k < qn) {
                    ea[j] = wl->qa[k * S + slot];
                    eb[j] = wl->qb[k * S + slot];
                    ec[j] = wl->qc[k * S + slot];
                    u32 kind = eb[j] >> 28, len = 0;
                    const u8 *src = c.dict;
                    if (kind == K_FAR) {
                        src = mb_base + ea[j] - ec[j];
                        len = eb[j] & 0xfffffffu;
                    } else if (kind == K_DICT) {
                        src = c.dict + (ec[j] & 0xfffffu);
                        len = (ec[j] >> 20) & 31;
                    }
                    if (len)
                        d0[j] = gld64u(src);
                    if (len > 8)
                        d1[j] = gld64u(src + 8);
                    if (len > 16)
                        d2[j] = gld64u(src + 16);
                }
            }
#ifdef YS_PROFILE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            c.t_dload += __builtin_readcyclecounter() - tb0;
#endif
#pragma unroll
            for (u32 j = 0; j < DB; j++) {
                u32 k = k0 + j;
                if (k < qn) {
                    u32 kind = eb[j] >> 28, n = eb[j] & 0xfffffffu;
                    u8 *dst = mb_base + ea[j];
                    u32 a = (u32)(u64)dst;
                    if (kind == K_FAR || kind == K_DICT) {
                        u32 plen = 0, slen = 0, len = n;
                        u64 pre = 0, suf = 0, w0 = d0[j], w1 = d1[j], w2 = d2[j];
                        if (kind == K_DICT) {
                            u32 xform = ec[j] >> 25;
                            u32 hdr = c.sh->xf[xform], op = hdr & 255;
                            plen = (hdr >> 16) & 255;
                            slen = hdr >> 24;
                            len = (ec[j] >> 20) & 31;
                            pre = c.sh->xpre[xform];
                            suf = c.sh->xsuf[xform];
                            if (op == 1 || op == 2) { /* UPPER(): yeast.c:809-827, on the fetched word */
                                u32 x32 = 0, x5 = 0, i = 0;
#ifdef YS_PROFILE
                                c.n_rfull++;
#endif
                                do {
                                    u64 w = i < 8 ? w0 : i < 16 ? w1 : w2;
                                    u32 ch = (u32)(w >> (8 * (i & 7))) & 255;
                                    if (ch < 192) {
                                        if (ch >= 97 && ch <= 122)
                                            x32 |= 1u << i;
                                        i += 1;
                                    } else if (ch < 224) {
                                        if (i + 1 < len)
                                            x32 |= 1u << (i + 1);
                                        i += 2;
                                    } else {
                                        if (i + 2 < len)
                                            x5 |= 1u << (i + 2);
                                        i += 3;
                                    }
                                } while (op == 2 && i < len);
                                w0 ^= (spread8(x32) << 5) | (spread8(x5) * 5);
                                w1 ^= (spread8(x32 >> 8) << 5) | (spread8(x5 >> 8) * 5);
                                w2 ^= (spread8(x32 >> 16) << 5) | (spread8(x5 >> 16) * 5);
                            }
                        }
                        if (plen)
                            ring_put<S>(rb, slot, a, pre, plen);
                        a += plen;
                        if (len)
                            ring_put<S>(rb, slot, a, w0, len);
                        if (len > 8)
                            ring_put<S>(rb, slot, a + 8, w1, len - 8);
                        if (len > 16)
                            ring_put<S>(rb, slot, a + 16, w2, len - 16);
                        if (len > 24) { /* a long copy that fits the ring (never a dictionary word) */
                            const u8 *src = dst - ec[j];
                            for (u32 i = 24; i < len; i += 8)
                                ring_put<S>(rb, slot, a + i, gld64u(src + i), len - i);
                        }
                        if (slen)
                            ring_put<S>(rb, slot, a + len, suf, slen);
                    } else if (kind == K_BYTES) {
                        /* near or overlapping copy (byte-serial in yeast.c:1285-1293): 8 bytes at a
                           time, each group from wherever its source is now.  A distance below 8 is
                           a repeating pattern: its first 8 bytes are built in a register, and from
                           there on the same bytes are found a multiple of the distance back. */
                        u32 d = ec[j], i = 0;
                        if (d < 8) {
                            const u8 *src = dst - d;
                            u64 pat = 0;
                            if (src >= fl)
                                pat = ring_get<S>(rb, slot, (u32)(u64)src) & ~(~0ull << (8 * d));
                            else
                                for (u32 b = 0; b < d; b++)
                                    pat |= (u64)(src + b < fl ? gld8(src + b) : staged(src + b)) << (8 * b);
                            for (u32 w = d; w < 8; w <<= 1)
                                pat |= pat << (8 * w);
                            ring_put<S>(rb, slot, a, pat, n);
                            d *= d >= 4 ? 2 : d == 3 ? 3 : d == 2 ? 4 : 8;
                            i = 8;
                        }
                        for (; i < n; i += 8) {
                            const u8 *src = dst + i - d;
                            u64 v;
                            if (src >= fl)
                                v = ring_get<S>(rb, slot, (u32)(u64)src);
                            else if (src + 8 <= fl)
                                v = gld64u(src);
                            else {
                                v = 0;
                                for (u32 b = 0; b < 8; b++)
                                    v |= (u64)(src + b < fl ? gld8(src + b) : staged(src + b)) << (8 * b);
                            }
                            ring_put<S>(rb, slot, a + i, v, n - i);
                        }
                    } else {
                        /* too long for the ring: bring the output region up to date and copy there;
                           this is always the newest entry and the drain follows it at once */
                        flush(dst, true);
                        group_sync_mem();
                        copy_exact<G>(dst, n, ec[j], sub);
                        group_sync_mem();
                        fl = dst + n;
                    }
                }
            }
        }
        qn = 0;
        flush(c.outp, false);
    };
    /* The distance and copy code is the longest part of a trip and only about a third of the
       streams are at a distance symbol on any one trip, so it runs on odd trips only: a stream
       that reaches a distance on an even trip sits that trip out, and the code runs half as
       often with about twice as many streams ready for it. */
    u32 odd = 1;
#ifdef YS_PROFILE
    c.t_mark = __builtin_readcyclecounter();
#endif
    while (st != ST_END) {
        odd ^= 1;
        /* sits out: a literal whose context bytes are still queued (it waits for the next
           drain), a distance or an implicit-distance copy on an even trip */
        const bool hold = (st == ST_LIT && ctx && stale) || ((st == ST_DIST || st == ST_COPY) && !odd);
        if (!SIMPLE) {
            if (!link && !hold) {
                if (st == ST_CMD) {
                    if (c.I.num > 1) {
                        if (c.I.left == 0) {
                            switch_block(c, c.I);
                            offI = gld32(fe.itab + c.I.type);
                        }
                        c.I.left--;
                    }
                } else if (st == ST_LIT) {
                    if (c.L.num > 1) {
                        if (c.L.left == 0) {
                            switch_block(c, c.L);
                            cur_mode = gld8(fe.mode + c.L.type);
                            if (ctx)
                                offL = next_lit_table();
                        }
                        c.L.left--;
                    }
                } else if (st == ST_DIST) {
                    if (c.D.num > 1) {
                        if (c.D.left == 0)
                            switch_block(c, c.D);
                        c.D.left--;
                    }
                    offD = fe.offD0;
                    if (c.dist_codes > 1)
                        offD = gld32(fe.dtab + (c.D.type << 2) + (copy > 4 ? 3 : copy - 2));
                }
            }
        }
        if (!link)
            off = st == ST_CMD ? offI : st == ST_LIT ? offL : offD;

        /* ---- the one table lookup of this trip ---- */
        if (nxt_empty) { /* rare: a second refill in the previous trip used up the prefetched word */
            r.nxt = gld32(r.p + r.idx);
            asm volatile("" : "+v"(r.nxt));
            nxt_empty = 0;
        }
        rd_refill(r); /* consumes a word that is in a register already; starts the next prefetch */
        u32 peek = (u32)r.w;
        u32 idx = link ? (link >> 4) + ((peek >> ROOT) & ((1u << ((link & 15) - ROOT)) - 1)) : (peek & 255);
        YS_MARK(t_s0);
        u32 e = gld16((const u16 *)(c.ws + off) + idx);
#ifdef YS_PROFILE
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(e));
        YS_MARK(t_look);
        c.n_trips++;
#endif
        /* what this trip does with the entry; ACT_NONE = first half of a long code, or failed */
        u32 l = e & 15;
        /* everything issued before the lookup has arrived with it: the prefetched input word
           (pinned here so that the next refill finds it in a register, not in flight) and the
           context bytes of the newest copy */
        asm volatile("" : "+v"(r.nxt));
        if (pn_valid) {
            c.p1 = pn_valid == 1 ? pn >> 8 : pn_c;
            c.p2 = pn & 255;
            pn_valid = 0;
        }
        bool first_half = !link && l > ROOT;
        const bool nolook = hold || st == ST_COPY; /* nothing is decoded for this stream on this trip */
        first_half = first_half && !nolook;
        link = first_half ? e : hold ? link : 0; /* the second half of a long distance code may have to wait */
        l = first_half || nolook ? 0 : l;
        u32 act = first_half || hold ? (u32)ACT_NONE : st;
        if (l > r.have) {
            rd_eof(r);
            l = 0;
            act = ACT_NONE;
        }
        r.w >>= l;
        r.have -= l;
        u32 sym = e >> 4;

        /* ---- extra bits: insert + copy (command), distance extra, none (literal) ---- */
        u64 ce = cmd_tab[sym < 704 ? sym : 0];
        u32 iext = ((u32)ce >> 16) & 0xff, cext = (u32)(ce >> 40);
        u32 dn = sym - c.direct - 16; /* distance symbols past the ring and direct codes */
        bool dlong = act == ST_DIST && sym >= 16 + c.direct;
        u32 dx = 1 + (dn >> (c.postfix + 1));
        dx = dx > 25 ? 25 : dx; /* only after a read failure */
        /* two fields at most: insert then copy extra bits, or the distance extra bits */
        u32 n1 = act == ST_CMD ? iext : dlong ? dx : 0, n2 = act == ST_CMD ? cext : 0;
        if (n1 > r.have)
            refill_mid();
        if (n1 > r.have) {
            rd_eof(r);
            n1 = n2 = 0;
            act = ACT_NONE;
        }
        u32 x1 = (u32)r.w & ((1u << n1) - 1);
        r.w >>= n1;
        r.have -= n1;
        if (n2 > r.have)
            refill_mid();
        if (n2 > r.have) {
            rd_eof(r);
            n2 = 0;
            act = ACT_NONE;
        }
        u32 x2 = (u32)r.w & ((1u << n2) - 1);
        r.w >>= n2;
        r.have -= n2;

        YS_MARK(t_s2);
        bool do_copy = false;
        i64 dist = c.r0;
        if (act == ST_CMD) {
            insert = ((u32)ce & 0xffff) + x1;
            copy = ((u32)(ce >> 24) & 0xffff) + x2;
            if (insert > mlen) {
                rd_fail(r, YS_INVALID); /* :1189 */
                insert = 0;
            }
            mlen -= insert;
            implicit = sym < 128;
            st = insert ? (u32)ST_LIT : implicit ? (u32)ST_COPY : (u32)ST_DIST;
        } else if (act == ST_LIT) {
            rb[ring_at<S>(slot, (u32)(u64)c.outp)] = (u8)sym;
            c.outp++;
            maxd = maxd < c.wsize ? maxd + 1 : maxd;
            c.p2 = c.p1;
            c.p1 = sym;
            if (--insert == 0) {
                st = mlen == 0 ? (u32)ST_END : implicit ? (u32)ST_COPY : (u32)ST_DIST; /* :1227 */
            }
        }
        YS_MARK(t_s3);
        if (odd) {
        if (act == ST_COPY)
            do_copy = true; /* the distance is the newest one of the ring */
        else if (act == ST_DIST) { /* distance(): yeast.c:761-789 */
            u32 ds = sym;
            if (ds < 16) {
                u32 back = ds < 4 ? ds : ds < 10 ? 0 : 1;
                u32 rv = back == 0 ? c.r0 : back == 1 ? c.r1 : back == 2 ? c.r2 : c.r3;
                u32 k = ds < 10 ? ds - 4 : ds - 10; /* -1 +1 -2 +2 -3 +3 */
                int dl = (int)(k >> 1) + 1;
                dl = (k & 1) ? dl : -dl;
                dl = ds < 4 ? 0 : dl;
                dist = (i64)rv + dl;
            } else if (!dlong)
                dist = ds - 15;
            else {
                u64 o2 = ((u64)(2 + ((dn >> c.postfix) & 1)) << dx) - 4;
                dist = (i64)(((o2 + x1) << c.postfix) + (dn & ((1u << c.postfix) - 1)) + c.direct + 1);
            }
            if (ds && dist > 0 && dist <= (i64)maxd) {
                c.r3 = c.r2; c.r2 = c.r1; c.r1 = c.r0; c.r0 = (u32)dist;
            }
            do_copy = true;
        }
        if (do_copy && !r.rc) {
            /* validate and account for the copy now; queue the byte moving */
            u32 rel = (u32)(c.outp - mb_base), nout = 0, ref = 0, kind = K_FAR;
            if (dist <= 0) {
                rd_fail(r, YS_INVALID); /* zero: see DESIGN.md; negative: the reference's code 3 */
            } else if (dist > (i64)maxd) {
                DictRef dr;
                u64 id = (u64)dist - maxd - 1;
                if (!dict_prepare(c, copy, id, dr) || dr.total > mlen) {
                    rd_fail(r, YS_INVALID); /* :921-930, :1269 */
                } else {
                    nout = dr.total;
                    kind = K_DICT;
                    ref = (dr.xform << 25) | (dr.wlen << 20) | (u32)(dr.w + dr.wskip - c.dict);
                    if (ctx && nout) {
                        /* last two bytes of the transformed word: suffix and prefix bytes are in
                           LDS; raw dictionary bytes are fetched (arriving with the next lookup);
                           the few cases that need the uppercase rules wait for the drain */
                        if (dr.slen >= 2) {
                            u64 sf = c.sh->xsuf[dr.xform] >> (8 * (dr.slen - 2));
                            c.p2 = (u32)sf & 255;
                            c.p1 = (u32)(sf >> 8) & 255;
                            stale = 0;
                        } else if (dr.wlen >= 2 + (1 - dr.slen) && dr.op != 2 && !(dr.op == 1 && dr.wlen < 5)) {
                            const u8 *we = dr.w + dr.wskip + dr.wlen;
                            if (dr.slen == 1) {
                                pn_c = (u32)c.sh->xsuf[dr.xform] & 255;
                                pn = gld8(we - 1);
                                pn_valid = 2;
                            } else {
                                pn = gld16u(we - 2);
                                pn_valid = 1;
                            }
                            stale = 0;
                        } else
                            stale = 1;
                    }
                }
            } else if (copy > mlen) {
                rd_fail(r, YS_INVALID); /* :1283 */
            } else {
                nout = copy;
                ref = (u32)dist;
                const u8 *s1 = c.outp - dist + copy; /* end of the source */
                bool fits = (u32)(c.outp - fl) + copy <= RB;
                kind = !fits ? (u32)K_SLOW : s1 <= fl ? (u32)K_FAR : (u32)K_BYTES;
                if (ctx) {
                    /* the last two bytes of the copy, straight from their sources when those are
                       final already (not behind a queued write): fetched from the output region
                       (arriving with the next lookup) or read from the ring */
                    const u8 *first = qn ? mb_base + pend_rel : c.outp;
                    u32 i1 = copy - 1, i2 = copy - 2;
                    if ((u32)dist < copy) { /* overlapping: byte i repeats byte i mod dist */
                        i1 %= (u32)dist;
                        i2 %= (u32)dist;
                    }
                    const u8 *q1 = c.outp - dist + i1, *q2 = c.outp - dist + i2;
                    if (q1 >= first || q2 >= first)
                        stale = 1;
                    else if (q2 + 1 == q1 && q1 < fl) {
                        pn = gld16u(q2);
                        pn_valid = 1;
                        stale = 0;
                    } else {
                        c.p1 = q1 >= fl ? staged(q1) : gld8(q1);
                        c.p2 = q2 >= fl ? staged(q2) : gld8(q2);
                        stale = 0;
                    }
                }
            }
            mlen -= nout;
            if (nout) {
                pend_rel = qn ? pend_rel : rel;
                wl->qa[qn * S + slot] = rel;
                wl->qb[qn * S + slot] = nout | (kind << 28);
                wl->qc[qn * S + slot] = ref;
                qn++;
            }
            c.outp += nout;
            u32 room = c.wsize - maxd;
            maxd += nout < room ? nout : room;
            st = mlen ? (u32)ST_CMD : (u32)ST_END;
        }
        }
        st = r.rc ? (u32)ST_END : st;
        YS_MARK(t_s4);
        /* second pass, the whole wavefront at once: when some queue is full, some ring is nearly
           full, or some stream needs bytes that are still queued */
        bool need = st == ST_LIT && ctx && stale;
        held = need ? held + 1 : 0;
        if (__any(qn == FQ || held > YS_HOLD || (u64)(c.outp - fl) > RB - RB_SLACK)) {
#ifdef YS_PROFILE
            u64 td0 = __builtin_readcyclecounter();
            c.n_need += need;
            c.n_qfull += qn == FQ;
#endif
            drain();
            if (need) {
                last_two();
                stale = 0;
                held = 0;
            }
#ifdef YS_PROFILE
            c.t_drain += __builtin_readcyclecounter() - td0;
            c.n_drains++;
            c.t_mark = __builtin_readcyclecounter();
#endif
        }
        if (st == ST_LIT && ctx)
            offL = next_lit_table();
        YS_MARK(t_s5);
    }
    /* metablock end (or failure): drain what is queued -- everything queued was valid -- and
       bring the output region up to date */
    drain();
    flush(c.outp, true);
    group_sync_mem();
    if (pn_valid) {
        c.p1 = pn_valid == 1 ? pn >> 8 : pn_c;
        c.p2 = pn & 255;
    }
    if (stale) /* context for the next meta-block */
        last_two();
}

#include "yeast_lean.h"

/* ------------------------------------------------------------- meta-block */

/* metablock(): yeast.c:971-1308.  Returns ISLAST. */
template <int G, bool CMP, int FLAT> /* FLAT: 0, or the streams per wavefront of the batch kernel */
DEV u32 metablock(Ctx &c, u32 sub)
{
    Rd &r = c.r;
    u32 last = take(r, 1);
    if (last && take(r, 1)) { /* ISLASTEMPTY :993 */
        if (!r.rc && tail_nonzero(r))
            rd_fail(r, YS_INVALID);
        return 1;
    }
    u32 n = take(r, 2); /* MNIBBLES - 4 */
    if (r.rc)
        return 1;
    if (n == 3) { /* metadata :1008-1032 */
        if (take(r, 1)) {
            rd_fail(r, YS_INVALID);
            return 1;
        }
        n = take(r, 2);
        u32 skip = n ? take(r, n << 3) + 1 : 0;
        if (r.rc)
            return 1;
        if ((n > 1 && (skip >> ((n - 1) << 3)) == 0) || tail_nonzero(r)) {
            rd_fail(r, YS_INVALID);
            return 1;
        }
        u32 pos = rd_align(r);
        if (skip > ((r.total - pos) >> 3)) {
            rd_fail(r, YS_EOF); /* :1025, nothing more consumed */
            return 1;
        }
        rd_seek(r, c.src, pos + skip * 8);
        return last;
    }
    u32 mlen = 1 + take(r, 16);
    if (n) {
        mlen += take(r, n << 2) << 16;
        if (!r.rc && (mlen >> ((n + 3) << 2)) == 0)
            rd_fail(r, YS_INVALID); /* :1037 */
    }
    if (r.rc)
        return 1;
    if (got_of(c) + mlen > c.cap) { /* :1040-1047 */
        c.need = got_of(c) + mlen;
        rd_fail(r, CMP ? YS_MISMATCH : YS_CAPACITY);
        return 1;
    }
    if (!last && take(r, 1)) { /* ISUNCOMPRESSED :1050 */
        if (tail_nonzero(r)) {
            rd_fail(r, YS_INVALID);
            return 1;
        }
        u32 pos = rd_align(r);
        if (mlen > ((r.total - pos) >> 3)) {
            rd_fail(r, YS_EOF); /* :1059 */
            return 1;
        }
        emit_raw<G, CMP>(c, c.src + (pos >> 3), mlen, sub);
        if (r.rc)
            return 1;
        rd_seek(r, c.src, pos + mlen * 8);
        return last;
    }
    if (r.rc)
        return 1;

    /* ---- header: yeast.c:1078-1159 ---- */
#ifdef YS_PROFILE
    u64 tp0 = __builtin_readcyclecounter();
#endif
    Hdr h;
    h.r = r;
    h.ws = c.ws;
    h.arena_used = 0;
    h.arena_size = c.arena_size;
    h.tab = YS_WS_ARENA;
    h.sub = sub;
    h.glog = 31 - __builtin_clz((u32)G);
    h.lds = nullptr; /* tables in the global arena */
    h.lds_used = h.lds_size = 0;
    h.rootcaps = c.rootcaps;
    h.kind = YS_KIND_OTHER;
    read_block_header(h, c.L);
    read_block_header(h, c.I);
    read_block_header(h, c.D);
    c.postfix = take(h.r, 2);
    c.direct = take(h.r, 4) << c.postfix;
    u32 dists = 16 + c.direct + (48u << c.postfix);
    u8 *mode = c.ws + YS_WS_MODE;
    for (n = 0; n < c.L.num; n++)
        gst8(mode + n, take(h.r, 2));
    c.lit_codes = read_count256(h.r);
    /* literal context map (u8 per entry) and literal table map (u32 per entry) */
    u32 nmap = c.lit_codes > 1 ? (c.L.num << 6) : 0;
    if (h.arena_used + nmap * 5 + 32 > h.arena_size)
        rd_fail(h.r, YS_WORKSPACE);
    if (h.r.rc) {
        r = h.r;
        return 1;
    }
    u8 *lit_map = c.ws + YS_WS_ARENA + h.arena_used;
    h.arena_used += (nmap + 15) & ~15u;
    u32 *lit_tab = (u32 *)(c.ws + YS_WS_ARENA + h.arena_used);
    h.arena_used += nmap * 4;
    if (c.lit_codes > 1)
        h = read_context_map(h, lit_map, nmap, c.lit_codes);
    c.dist_codes = read_count256(h.r);
    u8 *dist_map = c.ws + YS_WS_DMAP;
    if (c.dist_codes > 1 && !h.r.rc)
        h = read_context_map(h, dist_map, c.D.num << 2, c.dist_codes);

    u32 *ldir = (u32 *)(c.ws + YS_WS_LDIR), *ddir = (u32 *)(c.ws + YS_WS_DDIR);
    u32 *itab = (u32 *)(c.ws + YS_WS_ITAB), *dtab = (u32 *)(c.ws + YS_WS_DTAB);
    h.kind = YS_KIND_LIT;
    for (n = 0; n < c.lit_codes && !h.r.rc; n++) {
        h = read_code(h, 256);
        gst32(ldir + n, h.tab);
    }
    h.kind = YS_KIND_IAC;
    for (n = 0; n < c.I.num && !h.r.rc; n++) {
        h = read_code(h, 704);
        gst32(itab + n, h.tab);
    }
    h.kind = YS_KIND_DIST;
    for (n = 0; n < c.dist_codes && !h.r.rc; n++) {
        h = read_code(h, dists);
        gst32(ddir + n, h.tab);
    }
    r = h.r;
    if (r.rc)
        return 1;
    group_sync_mem();
    /* resolve the context maps to table offsets once, the entries spread over the lanes */
    const u16 *tL0 = tab_at(c, gld32(ldir)), *tD0 = tab_at(c, gld32(ddir));
    for (n = sub; n < nmap; n += G)
        gst32(lit_tab + n, gld32(ldir + gld8(lit_map + n)));
    if (c.dist_codes > 1)
        for (n = sub; n < (c.D.num << 2); n += G)
            gst32(dtab + n, gld32(ddir + gld8(dist_map + n)));
    group_sync_mem();
    const u16 *tI = tab_at(c, gld32(itab));
    u32 cur_mode = gld8(mode);
    const bool ctx = c.lit_codes > 1;
    const u64 *cmd_tab = c.sh->cmd;

    if (!FLAT) {
        /* ---- command loop: yeast.c:1166-1295 ---- */
        do {
            if (c.I.num > 1) {
                if (c.I.left == 0) {
                    switch_block(c, c.I);
                    tI = tab_at(c, gld32(itab + c.I.type));
                }
                c.I.left--;
            }
            u32 cmd = dec_sym(r, tI);
            u64 ce = cmd_tab[cmd < 704 ? cmd : 0];
            u32 insert = ((u32)ce & 0xffff) + take(r, ((u32)ce >> 16) & 0xff);
            u32 copy = ((u32)(ce >> 24) & 0xffff) + take(r, (u32)(ce >> 40));
            if (r.rc)
                break;
            if (insert > mlen) {
                rd_fail(r, YS_INVALID); /* :1189 */
                break;
            }
            mlen -= insert;
            if (insert) {
                do {
                    if (c.L.num > 1) {
                        if (c.L.left == 0) {
                            switch_block(c, c.L);
                            cur_mode = gld8(mode + c.L.type);
                        }
                        c.L.left--;
                    }
                    const u16 *tl = tL0;
                    if (ctx)
                        tl = tab_at(c, gld32(lit_tab + (c.L.type << 6) + context_id(c, c.p1, c.p2, cur_mode)));
                    u32 s = dec_sym(r, tl);
                    if (r.rc)
                        break;
                    if (CMP) {
                        u32 e = gld8(c.outp);
                        c.outp++; /* yeast.c:1217: got advances before the compare throws */
                        if (e != s) {
                            rd_fail(r, YS_MISMATCH);
                            break;
                        }
                    } else {
                        gst8(c.outp, s);
                        c.outp++;
                    }
                    c.p2 = c.p1;
                    c.p1 = s;
                } while (--insert);
                if (r.rc)
                    break;
            }
            if (mlen == 0)
                break;

            u64 got = got_of(c);
            u32 max = got > c.wsize ? c.wsize : (u32)got;
            i64 dist;
            if (cmd < 128)
                dist = c.r0;
            else {
                if (c.D.num > 1) {
                    if (c.D.left == 0)
                        switch_block(c, c.D);
                    c.D.left--;
                }
                const u16 *td = tD0;
                if (c.dist_codes > 1)
                    td = tab_at(c, gld32(dtab + (c.D.type << 2) + (copy > 4 ? 3 : copy - 2)));
                u32 ds = dec_sym(r, td);
                /* distance(): yeast.c:761-789 */
                if (ds < 16) {
                    u32 back = ds < 4 ? ds : ds < 10 ? 0 : 1;
                    u32 rv = back == 0 ? c.r0 : back == 1 ? c.r1 : back == 2 ? c.r2 : c.r3;
                    int dl = 0;
                    if (ds >= 4) {
                        u32 k = ds < 10 ? ds - 4 : ds - 10; /* -1 +1 -2 +2 -3 +3 */
                        dl = (int)(k >> 1) + 1;
                        if (!(k & 1))
                            dl = -dl;
                    }
                    dist = (i64)rv + dl;
                } else if (ds < 16 + c.direct)
                    dist = ds - 15;
                else {
                    u32 nn = ds - c.direct - 16;
                    u32 x = 1 + (nn >> (c.postfix + 1));
                    if (x > 25)
                        x = 25; /* only after a read failure */
                    u64 off = ((u64)(2 + ((nn >> c.postfix) & 1)) << x) - 4;
                    dist = (i64)(((off + take(r, x)) << c.postfix) + (nn & ((1u << c.postfix) - 1)) + c.direct + 1);
                }
                if (r.rc)
                    break;
                if (ds && dist > 0 && dist <= (i64)max) {
                    c.r3 = c.r2; c.r2 = c.r1; c.r1 = c.r0; c.r0 = (u32)dist;
                }
            }
            if (dist <= 0) {
                rd_fail(r, YS_INVALID); /* zero: see DESIGN.md; negative: the reference's code 3 */
                break;
            }
            if (dist > (i64)max) {
                emit_dict<G, CMP>(c, copy, (u64)dist - max - 1, mlen, sub, ctx);
            } else {
                if (copy > mlen) {
                    rd_fail(r, YS_INVALID); /* :1283 */
                    break;
                }
                mlen -= copy;
                emit_copy<G, CMP>(c, copy, (u32)dist, sub, ctx);
            }
        } while (mlen && !r.rc);
    } else {
#ifdef YS_PROFILE
        u64 tp1 = __builtin_readcyclecounter();
        c.t_hdr += tp1 - tp0;
#endif
        FlatEnv fe;
        fe.itab = itab;
        fe.dtab = dtab;
        fe.lit_tab = lit_tab;
        fe.mode = mode;
        fe.offL0 = (u32)((const u8 *)tL0 - c.ws);
        fe.offD0 = (u32)((const u8 *)tD0 - c.ws);
        fe.cur_mode = cur_mode;
        fe.ctx = ctx;
        /* meta-blocks without block switching and with one distance code (the usual case for
           short streams) run a loop that carries none of that machinery; the choice is made
           for the whole wavefront so the streams stay in one loop */
        if constexpr (FLAT >= 1000) { /* the lean loop, one stream per lane */
            bool simple = c.L.num == 1 && c.I.num == 1 && c.D.num == 1;
            if (__all(simple))
                run_lean<FLAT >= 1000 ? FLAT - 1000 : 64, true>(c, mlen, fe);
            else
                run_lean<FLAT >= 1000 ? FLAT - 1000 : 64, false>(c, mlen, fe);
        } else {
            bool simple = c.L.num == 1 && c.I.num == 1 && c.D.num == 1 && c.dist_codes == 1;
            if (__all(simple))
                run_flat<G, FLAT && FLAT < 1000 ? FLAT : 64, true>(c, mlen, sub, fe);
            else
                run_flat<G, FLAT && FLAT < 1000 ? FLAT : 64, false>(c, mlen, sub, fe);
        }
#ifdef YS_PROFILE
        c.t_loop += __builtin_readcyclecounter() - tp1;
#endif
    }
    if (!r.rc && last && tail_nonzero(r))
        rd_fail(r, YS_INVALID); /* :1305 */
    return last;
}

/* yeast(): yeast.c:1353-1400 for one stream of the batch */
template <int G, bool CMP, int FLAT>
DEV void decode_stream(const ys_params &p, u32 job, u8 *ws, const Shared *sh, u32 sub, void *wl = nullptr)
{
    Ctx c;
    u64 s0 = p.src_off[job], s1 = p.src_off[job + 1];
    u64 d0 = p.dst_off[job], d1 = p.dst_off[job + 1];
    c.src = p.src + s0;
    c.out = p.dst + d0;
    c.outp = c.out;
    c.cap = d1 - d0;
    c.r.rc = YS_OK;
    c.r.dry = 0;
    c.r.fpos = 0;
    c.r.total = (u32)(s1 - s0) * 8;
    c.r0 = 4; c.r1 = 11; c.r2 = 15; c.r3 = 16; /* :1327-1331, most recent first */
    c.ws = ws;
    c.arena_size = (u32)(p.ws_stride - YS_WS_ARENA);
    c.p1 = c.p2 = 0;
    c.nmeta = 0;
    c.need = 0;
#ifdef YS_PROFILE
    c.t_hdr = c.t_loop = c.t_drain = c.t_look = c.t_dload = 0;
    c.t_s0 = c.t_s2 = c.t_s3 = c.t_s4 = c.t_s5 = c.t_mark = 0;
    c.n_trips = c.n_drains = c.n_need = c.n_qfull = c.n_rfull = c.n_ents = 0;
#endif
    c.dict = p.dict;
    c.xf = p.xforms;
    c.xdict = p.xdict;
    c.rootcaps = p.rootcaps ? p.rootcaps : 0x08080808u;
    c.sh = sh;
    c.wl = wl;
    c.L.num = c.I.num = c.D.num = 1;
    rd_seek(c.r, c.src, 0);
    if (s1 - s0 >= (1u << 29))
        rd_fail(c.r, YS_NOMEM); /* the bit cursor is 32 bits wide: streams of 512 MiB and more are refused */

    u32 wbits = 16;
    if (take(c.r, 1)) { /* WBITS :1372-1378 */
        u32 v = take(c.r, 3);
        if (v)
            wbits = v + 17;
        else {
            v = take(c.r, 3);
            wbits = v ? v + 8 : 17;
        }
    }
    if (!c.r.rc && wbits == 9)
        rd_fail(c.r, YS_INVALID);
    c.wsize = (1u << wbits) - 16;
    while (!c.r.rc) {
        u32 last = metablock<G, CMP, FLAT>(c, sub);
        if (c.r.rc)
            break;
        c.nmeta++;
        if (last)
            break;
    }

    ys_record rec;
    rec.rc = (int32_t)c.r.rc;
    rec.flags = wbits | ((c.nmeta > 0xffffff ? 0xffffffu : c.nmeta) << 8);
    rec.got = got_of(c);
    /* bytes pulled: every byte when the bit reader ran dry, else ceil(bits/8) (yeast.c:1388) */
    rec.used = c.r.dry ? (s1 - s0) : (rd_pos(c.r) + 7) >> 3;
    rec.check = c.need;
#ifdef YS_PROFILE
    rec.check = ((c.t_hdr >> 10) & 0xfffff) | (((c.t_loop >> 10) & 0xfffff) << 20) | (((c.t_drain >> 10) & 0xfffff) << 40);
    rec.used = (c.t_look >> 10) | ((u64)c.n_trips << 32);
    rec.flags = c.n_drains;
    if (sub == 0 && c.cap >= 64) {
        u64 *prof = (u64 *)c.out; /* the profile build's output bytes are not meant to be looked at */
        prof[0] = c.t_s0; prof[1] = c.t_look; prof[2] = c.t_s2; prof[3] = c.t_s3; prof[4] = c.t_s4; prof[5] = c.t_s5;
    }
    rec.got = (u64)(c.n_need & 0xfff) | ((u64)(c.n_qfull & 0xfff) << 12) | ((u64)(c.n_rfull & 0xfff) << 24) | ((c.t_dload >> 10) << 36);
#endif
    p.rec[job] = rec;
    if (c.r.rc == YS_WORKSPACE && !p.retry && sub == 0)
        p.retry_list[atomicAdd(p.retry_count, 1u)] = job;
}

/* fills this translation unit's copies of the constant tables */
static int ys_upload_constants_impl(const u8 *lut768)
{
    /* insert_length()/copy_length() tables of yeast.c:655-665 and :678-688, from their structure:
       each base is the previous base plus 1 << previous extra; cell maps per symbol >> 6 */
    static const u8 ins_x[24] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24};
    static const u8 cpy_x[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24};
    static const u8 ins_cell[11] = {0, 0, 0, 0, 8, 8, 0, 16, 8, 16, 16};
    static const u8 cpy_cell[11] = {0, 8, 0, 8, 0, 8, 16, 0, 16, 8, 16};
    u32 ib[24], cb[24];
    u32 b = 0;
    for (int i = 0; i < 24; i++) {
        ib[i] = b;
        b += 1u << ins_x[i];
    }
    b = 2;
    for (int i = 0; i < 24; i++) {
        cb[i] = b;
        b += 1u << cpy_x[i];
    }
    static u64 cmd[704];
    for (u32 s = 0; s < 704; s++) {
        u32 ic = ins_cell[s >> 6] + ((s >> 3) & 7), cc = cpy_cell[s >> 6] + (s & 7);
        cmd[s] = (u64)ib[ic] | ((u64)ins_x[ic] << 16) | ((u64)cb[cc] << 24) | ((u64)cpy_x[cc] << 40);
    }
    /* NDBITS (yeast.c:916-919); DOFFSET[len] = sum over l < len of l << NDBITS[l] (:911-915) */
    static const u8 ndbits[25] = {0, 0, 0, 0, 10, 10, 11, 11, 10, 10, 10, 10, 10, 9, 9, 8, 7, 7, 8, 7, 7, 6, 6, 5, 5};
    u32 dict[32] = {0};
    u32 off = 0;
    for (u32 l = 4; l <= 24; l++) {
        dict[l] = off | ((u32)ndbits[l] << 24);
        off += l << ndbits[l];
    }
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_ctx_lut), lut768, 768);
    if (e != hipSuccess)
        return (int)e;
    e = hipMemcpyToSymbol(HIP_SYMBOL(c_cmd_tab), cmd, sizeof(cmd));
    if (e != hipSuccess)
        return (int)e;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(c_dict_tab), dict, sizeof(dict));
}

#endif
