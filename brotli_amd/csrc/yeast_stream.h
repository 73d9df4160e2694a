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
    u32 kcls[256];            /* per byte value b: lut0[b] | lut1[b] << 8 | lut2[b] << 16 | b << 24 (literal contexts) */
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
    for (u32 i = threadIdx.x; i < 256; i += blockDim.x)
        sh.kcls[i] = (u32)c_ctx_lut[i] | ((u32)c_ctx_lut[256 + i] << 8) | ((u32)c_ctx_lut[512 + i] << 16) | (i << 24);
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
    void *wl; /* per-wavefront LDS of the batch kernel (TripLds<S>) */
    const struct TripShared *tsh; /* per-block LDS of the batch kernel */
    const u8 *ws_base; /* start of the workspace arena (uniform): the batch kernel addresses tables from it */
#ifdef YS_PROFILE
    u64 t_start; /* 100 MHz clock at the start of the stream */
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
    if (G == 64 && n >= 1024 && (d >= n || d >= 1024)) { /* a long copy by the whole wavefront */
        if (ctx) {
            c.p1 = gld8(from + (n - 1) % d);
            c.p2 = gld8(from + (n - 2) % d);
        }
        wave_copy_far(dst, d, n, sub);
    } else if (d >= n || (G == 1 && d >= 8)) {
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
        gst32u(p, (u32)v);
        o = 4;
    }
    if (n & 2) {
        gst16u(p + o, (u32)(v >> (8 * o)));
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
    if (G == 64 && n >= 1024 && (d >= n || d >= 1024)) { /* a long copy by the whole wavefront */
        wave_copy_far(dst, d, n, sub);
    } else if (d >= n || (G == 1 && d >= 8)) {
        if (n >= 8) {
            for (u32 j = sub * 8; j + 8 <= n; j += G * 8)
                gst64u(dst + j, gld64u(from + j));
            if ((n & 7) && (G == 1 || sub == ((n >> 3) & (G - 1))))
                gst64u(dst + n - 8, gld64u(from + n - 8));
        } else if (G == 1 || sub == 0) {
            u32 o = 0;
            if (n & 4) {
                gst32u(dst, gld32u(from));
                o = 4;
            }
            if (n & 2) {
                gst16u(dst + o, gld16u(from + o));
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
    } else if (G == 64 && n >= 1024) {
        wave_copy_disjoint(dst, from, n, sub);
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

/* what the meta-block header hands to the command loop of the batch kernel (yeast_trip.h) */
struct FlatEnv {
    const u32 *itab, *dtab, *lit_tab;
    const u8 *mode;
    u32 offL0, offD0, cur_mode;
    bool ctx;
};

#define YS_XD_LENS 21 /* word lengths 4..24 */
#include "yeast_trip.h"

/* ------------------------------------------------------------- meta-block */

/* metablock(): yeast.c:971-1308.  Returns ISLAST. */
template <int G, bool CMP, int FLAT> /* FLAT: 0, or the streams per wavefront of the batch kernel (G = 1) */
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
    if (FLAT && got_of(c) + mlen > YS_LANE_MAX_OUT) {
        /* a stream this long is one lane's work for too long: it goes to the retry pass, where a whole
           wavefront decodes it with wide copies (k_decode<64>) */
        rd_fail(r, YS_WORKSPACE);
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
#ifdef YS_PROFILE
    u64 hp_[6];
    hp_[0] = __builtin_amdgcn_s_memrealtime();
    h.tp0 = h.tp1 = h.tp2 = h.tp3 = 0;
#endif
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

#ifdef YS_PROFILE
    hp_[1] = __builtin_amdgcn_s_memrealtime();
#endif
    u32 *ldir = (u32 *)(c.ws + YS_WS_LDIR), *ddir = (u32 *)(c.ws + YS_WS_DDIR);
    u32 *itab = (u32 *)(c.ws + YS_WS_ITAB), *dtab = (u32 *)(c.ws + YS_WS_DTAB);
    h.kind = YS_KIND_LIT;
    for (n = 0; n < c.lit_codes && !h.r.rc; n++) {
        h = read_code(h, 256);
        gst32(ldir + n, h.tab);
    }
#ifdef YS_PROFILE
    hp_[2] = __builtin_amdgcn_s_memrealtime();
#endif
    h.kind = YS_KIND_IAC;
    for (n = 0; n < c.I.num && !h.r.rc; n++) {
        h = read_code(h, 704);
        gst32(itab + n, h.tab);
    }
#ifdef YS_PROFILE
    hp_[3] = __builtin_amdgcn_s_memrealtime();
#endif
    h.kind = YS_KIND_DIST;
    for (n = 0; n < c.dist_codes && !h.r.rc; n++) {
        h = read_code(h, dists);
        gst32(ddir + n, h.tab);
    }
#ifdef YS_PROFILE
    hp_[4] = __builtin_amdgcn_s_memrealtime();
#endif
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
#ifdef YS_PROFILE
    hp_[5] = __builtin_amdgcn_s_memrealtime();
    if (FLAT && (threadIdx.x & 63) == 0 && c.cap >= 256) { /* block headers + maps, literal codes, insert-and-copy, distance, map resolution; counts */
        u64 *hp = (u64 *)(c.out + c.cap - 256);
        for (int i = 0; i < 5; i++)
            hp[i] = hp_[i + 1] - hp_[i];
        hp[5] = c.lit_codes | ((u64)c.I.num << 16) | ((u64)c.dist_codes << 32) | ((u64)c.L.num << 48);
        hp[6] = h.tp0; hp[7] = h.tp1; hp[8] = h.tp2; hp[9] = h.tp3;
    }
#endif
    const u16 *tI = tab_at(c, gld32(itab));
    u32 cur_mode = gld8(mode);
    const bool ctx = c.lit_codes > 1;
    const u64 *cmd_tab = c.sh->cmd;

    if (!FLAT) {
        /* The copies of a meta-block WITHOUT a literal context map leave p1/p2 alone (emit_copy, emit_dict), so when such
           a block ends in a copy, the first literals of a following block WITH a map would take their context from stale
           bytes (found by tools/soak_random.py on a damaged stream, tests/golden/soak_3034_770.bin: intact plans end
           every block in literals).  The last two bytes are in memory (compare mode: the expected data, equal so far). */
        if (ctx) {
            u64 got = got_of(c);
            c.p1 = got >= 1 ? gld8(c.outp - 1) : 0;
            c.p2 = got >= 2 ? gld8(c.outp - 2) : 0;
        }
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
        FlatEnv fe;
        fe.itab = itab;
        fe.dtab = dtab;
        fe.lit_tab = lit_tab;
        fe.mode = mode;
        fe.offL0 = (u32)((const u8 *)tL0 - c.ws);
        fe.offD0 = (u32)((const u8 *)tD0 - c.ws);
        fe.cur_mode = cur_mode;
        fe.ctx = ctx;
        run_trip<FLAT>(c, mlen, fe);
    }
    if (!r.rc && last && tail_nonzero(r))
        rd_fail(r, YS_INVALID); /* :1305 */
    return last;
}

/* yeast(): yeast.c:1353-1400 for one stream of the batch */
template <int G, bool CMP, int FLAT>
DEV void decode_stream(const ys_params &p, u32 job, u8 *ws, const Shared *sh, u32 sub, void *wl = nullptr,
                       const struct TripShared *tsh = nullptr, const u8 *ws_base = nullptr)
{
    Ctx c;
#ifdef YS_PROFILE
    c.t_start = __builtin_amdgcn_s_memrealtime();
#endif
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
    c.dict = p.dict;
    c.xf = p.xforms;
    c.xdict = p.xdict;
    c.rootcaps = p.rootcaps ? p.rootcaps : 0x08080808u;
    c.sh = sh;
    c.wl = wl;
    c.tsh = tsh;
    c.ws_base = ws_base ? ws_base : p.ws;
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
