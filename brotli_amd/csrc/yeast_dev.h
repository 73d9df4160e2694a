/*
 * yeast_dev.h -- device code shared by the decode kernels (yeast_decode.hip, yeast_wave.hip):
 * global-memory accessors, the bit reader, two-level LUT decode and the meta-block header
 * parser (prefix-code descriptors, context maps).  Reference lines are cited per function.
 */
#ifndef YEAST_DEV_H
#define YEAST_DEV_H
#include <hip/hip_runtime.h>
#include "yeast_common.h"

#define DEV __device__ __forceinline__
#define NOINL __device__ __noinline__

/* every access below is to global memory: say so, or pointers that travelled through an
   out-of-line call are treated as generic and cost flat_* instructions */
#define GAS __attribute__((address_space(1)))
#include "yeast_bounds.h" /* YSB(p, n): true, unless this is the bounds-checked debug build and the access is out of range */
DEV u32 gld8(const void *p) { return YSB(p, 1) ? *(const GAS u8 *)p : 0; }
DEV u32 gld16(const void *p) { return YSB(p, 2) ? *(const GAS u16 *)p : 0; }
DEV u32 gld32(const void *p) { return YSB(p, 4) ? *(const GAS u32 *)p : 0; }
DEV u64 gld64u(const void *p) /* unaligned */
{
    typedef u64 __attribute__((aligned(1))) u64u;
    return YSB(p, 8) ? *(const GAS u64u *)p : 0;
}
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
DEV u32x4 gld128u(const void *p) /* unaligned */
{
    typedef u32x4 __attribute__((aligned(1))) u32x4u;
    return YSB(p, 16) ? *(const GAS u32x4u *)p : (u32x4)0;
}
/* the same with the streaming hint: the line is not kept in L2 for this access */
DEV u32x4 gld128u_nt(const void *p)
{
    typedef u32x4 __attribute__((aligned(1))) u32x4u;
    return YSB(p, 16) ? __builtin_nontemporal_load((const GAS u32x4u *)p) : (u32x4)0;
}
DEV void gst128u(void *p, u32x4 v)
{
    typedef u32x4 __attribute__((aligned(1))) u32x4u;
    if (YSB(p, 16))
        *(GAS u32x4u *)p = v;
}
DEV u32 gld16u(const void *p) /* unaligned */
{
    typedef u16 __attribute__((aligned(1))) u16u;
    return YSB(p, 2) ? *(const GAS u16u *)p : 0;
}
DEV void gst8(void *p, u32 v) { if (YSB(p, 1)) *(GAS u8 *)p = (u8)v; }
DEV void gst16(void *p, u32 v) { if (YSB(p, 2)) *(GAS u16 *)p = (u16)v; }
DEV void gst32(void *p, u32 v) { if (YSB(p, 4)) *(GAS u32 *)p = v; }
DEV void gst64(void *p, u64 v) { if (YSB(p, 8)) *(GAS u64 *)p = v; }
DEV void gst32u(void *p, u32 v) /* unaligned */
{
    typedef u32 __attribute__((aligned(1))) u32u;
    if (YSB(p, 4))
        *(GAS u32u *)p = v;
}
DEV void gst64u(void *p, u64 v)
{
    typedef u64 __attribute__((aligned(1))) u64u;
    if (YSB(p, 8))
        *(GAS u64u *)p = v;
}
DEV void gst16u(void *p, u32 v) /* unaligned */
{
    typedef u16 __attribute__((aligned(1))) u16u;
    if (YSB(p, 2))
        *(GAS u16u *)p = (u16)v;
}
DEV u32 gld32u(const void *p) /* unaligned */
{
    typedef u32 __attribute__((aligned(1))) u32u;
    return YSB(p, 4) ? *(const GAS u32u *)p : 0;
}

/* A long copy by a whole wavefront (64 lanes): n bytes from d bytes back, d >= n or d >= 1024.  Byte-serial semantics
   (yeast.c:1285-1293) hold as long as a byte is read after it was written, so the copy goes in segments no longer than d
   (a whole segment's source lies behind its destination) with a wait for the stores between segments; inside a
   segment the order is free: 16-byte units, YS_COPY_U loads in flight per lane before the first store (one load and
   its dependent store at a time is one memory round trip per 512 bytes: 28 ms for the 18 MB of BASELINE config 5).
   Out of line: the unit buffer is registers the caller's loop should not be allocated around. */
#ifndef YS_COPY_U
#define YS_COPY_U 16
#endif
static NOINL void wave_copy_units(u8 *t, const u8 *s, u32 units, u32 lane)
{
    for (u32 k0 = lane; k0 < units; k0 += 64 * YS_COPY_U) {
        u32x4 v[YS_COPY_U];
        if (k0 + 64 * (YS_COPY_U - 1) < units) { /* a full batch for this lane */
#pragma unroll
            for (u32 u = 0; u < YS_COPY_U; u++)
                v[u] = gld128u(s + 16 * (size_t)(k0 + u * 64));
#pragma unroll
            for (u32 u = 0; u < YS_COPY_U; u++)
                gst128u(t + 16 * (size_t)(k0 + u * 64), v[u]);
        } else {
#pragma unroll
            for (u32 u = 0; u < YS_COPY_U; u++)
                if (k0 + u * 64 < units)
                    v[u] = gld128u(s + 16 * (size_t)(k0 + u * 64));
#pragma unroll
            for (u32 u = 0; u < YS_COPY_U; u++)
                if (k0 + u * 64 < units)
                    gst128u(t + 16 * (size_t)(k0 + u * 64), v[u]);
        }
    }
}

/* n bytes that do not overlap their source (a stored block: yeast.c:1061-1070) */
DEV void wave_copy_disjoint(u8 *t, const u8 *s, u32 n, u32 lane)
{
    const u32 units = n >> 4;
    wave_copy_units(t, s, units, lane);
    const u32 t0 = units << 4;
    if (lane < (n & 15))
        gst8(t + t0 + lane, gld8(s + t0 + lane));
}

DEV void wave_copy_far(u8 *dst, u32 d, u32 n, u32 lane)
{
    const u8 *from = dst - d;
    u32 done = 0;
    while (done < n) {
        u32 seg = n - done;
        if (seg > d)
            seg = d & ~15u;
        wave_copy_disjoint(dst + done, from + done, seg, lane);
        done += seg;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); /* the next segment reads what this one wrote */
    }
}

/* ------------------------------------------------------------- bit reader */

struct Rd {
    const u32 *p; /* 4-byte aligned base at or before the current seek point */
    u32 idx;      /* dword index (from p) of the word held in nxt */
    u32 nxt;      /* prefetched dword */
    u64 w;        /* window, next bit at bit 0 */
    u32 have;     /* valid bits in w */
    u32 rem;      /* valid bits of the stream not yet moved into w */
    u32 total;    /* 8 * stream length */
    u32 rc;       /* sticky status */
    u32 fpos;     /* bits consumed when rc was set */
    u32 dry;      /* the reader ran out of input: the reference had pulled every byte */
};

DEV u32 rd_pos(const Rd &r) { return r.rc ? r.fpos : r.total - r.rem - r.have; }

DEV void rd_fail(Rd &r, u32 code)
{
    if (!r.rc) {
        r.fpos = r.total - r.rem - r.have;
        r.rc = code;
    }
    r.have = 0;
    r.rem = 0;
}

DEV void rd_seek(Rd &r, const u8 *src, u32 bitpos)
{
    const u8 *at = src + (bitpos >> 3);
    u32 mis = (u32)((uintptr_t)at & 3);
    r.p = (const u32 *)(at - mis);
    u32 sh = mis * 8 + (bitpos & 7);
    u64 lo = gld32(r.p), hi = gld32(r.p + 1);
    r.w = (lo | (hi << 32)) >> sh;
    u32 left = r.total - bitpos;
    r.have = 64 - sh < left ? 64 - sh : left;
    r.rem = left - r.have;
    r.nxt = gld32(r.p + 2);
    r.idx = 2;
}

DEV void rd_refill(Rd &r)
{
    if (r.have <= 32) {
        r.w |= (u64)r.nxt << r.have;
        u32 add = r.rem < 32 ? r.rem : 32;
        r.have += add;
        r.rem -= add;
        r.idx += add ? 1 : 0; /* never runs more than two words past the end */
        r.nxt = gld32(r.p + r.idx);
    }
}

DEV void rd_eof(Rd &r)
{
    if (!r.rc)
        r.dry = 1;
    rd_fail(r, YS_EOF);
}

/* bits(): yeast.c:163-179.  n <= 25.  On exhaustion the reference has pulled
   every input byte (yeast.c:169-175): dry is set and reported as *len. */
DEV u32 take(Rd &r, u32 n)
{
    rd_refill(r);
    if (n > r.have) {
        rd_eof(r);
        n = 0;
    }
    u32 v = (u32)r.w & ((1u << n) - 1);
    r.w >>= n;
    r.have -= n;
    return v;
}

/* "s->left && s->bits": the rest of the current byte is not zero.  Bits enter
   the window in whole bytes, so the bits left in the current byte are have % 8. */
DEV bool tail_nonzero(const Rd &r)
{
    u32 k = r.have & 7;
    return k && ((u32)r.w & ((1u << k) - 1));
}

/* drop the rest of the current byte; returns the new bit position */
DEV u32 rd_align(Rd &r)
{
    u32 k = r.have & 7;
    r.w >>= k;
    r.have -= k;
    return r.total - r.rem - r.have;
}

/* -------------------------------------------------------------- LUT decode
 * Table entry (u16): bits 0..3 = code length, bits 4..15 = symbol.  In the
 * 256-entry root a length above 8 marks a link: (length - 8) index bits,
 * value = offset of the sub-table from the root, in entries.
 */
#define ROOT 8
#define YS_TAB_ENTRIES 1104

DEV u32 dec_sym(Rd &r, const u16 *t)
{
    rd_refill(r);
    u32 peek = (u32)r.w;
    u32 e = gld16(t + (peek & 255));
    u32 l = e & 15;
    if (l > ROOT) {
        e = gld16(t + (e >> 4) + ((peek >> ROOT) & ((1u << (l - ROOT)) - 1)));
        l = e & 15;
    }
    if (l > r.have) {
        rd_eof(r);
        l = 0;
    }
    r.w >>= l;
    r.have -= l;
    return e >> 4;
}

/* packed 16 x 16-bit counters in four 64-bit registers (no indexed arrays) */
struct Cnt16 {
    u64 q0, q1, q2, q3;
};
DEV void cnt_clear(Cnt16 &k) { k.q0 = k.q1 = k.q2 = k.q3 = 0; }
DEV void cnt_add(Cnt16 &k, u32 i, u32 v)
{
    u64 inc = (u64)v << ((i & 3) * 16);
    u32 q = i >> 2;
    k.q0 += q == 0 ? inc : 0;
    k.q1 += q == 1 ? inc : 0;
    k.q2 += q == 2 ? inc : 0;
    k.q3 += q == 3 ? inc : 0;
}
DEV void cnt_dec(Cnt16 &k, u32 i)
{
    u64 inc = (u64)1 << ((i & 3) * 16);
    u32 q = i >> 2;
    k.q0 -= q == 0 ? inc : 0;
    k.q1 -= q == 1 ? inc : 0;
    k.q2 -= q == 2 ? inc : 0;
    k.q3 -= q == 3 ? inc : 0;
}
DEV u32 cnt_get(const Cnt16 &k, u32 i)
{
    u32 q = i >> 2;
    u64 x = q == 0 ? k.q0 : q == 1 ? k.q1 : q == 2 ? k.q2 : k.q3;
    return (u32)(x >> ((i & 3) * 16)) & 0xffff;
}

#define LAS __attribute__((address_space(3)))

/* Header-phase state handed to and from the out-of-line header routines. */
struct Hdr {
    Rd r;
    u8 *ws;
    u32 arena_used, arena_size; /* bytes, relative to ws + YS_WS_ARENA */
    u32 tab;                    /* result: byte offset of the table from ws (LDS mode: table descriptor) */
    u32 sub, glog;              /* lane within the group, log2(lanes per group) */
    /* LDS mode (the wavefront-per-stream kernel; lds != null, 64 lanes): tables are built in the
       wavefront's LDS slot.  Units are u16 entries from the slot base. */
    LAS u16 *lds;
    u32 lds_used, lds_size;
    u32 rootcaps; /* root table width by kind: other | literal << 8 | insert-and-copy << 16 | distance << 24
                     (global tables: exactly that wide; LDS mode: at most that wide) */
    u32 kind;     /* YS_KIND_* of the code read next */
#ifdef YS_PROFILE
    u64 tp0, tp1, tp2, tp3; /* read_code: code length code, lengths, sort, table (10 ns units) */
#endif
};

/* A wavefront that decodes ONE stream holds the same value of all of this in every lane, but a
   value that has travelled through an out-of-line call (or through memory) is a vector value to
   the compiler.  These put it back on the scalar unit. */
DEV u32 uni(u32 x) { return __builtin_amdgcn_readfirstlane(x); }
DEV u64 uni64(u64 x) { return (u64)uni((u32)x) | ((u64)uni((u32)(x >> 32)) << 32); }
DEV void rd_uniform(Rd &r)
{
    r.p = (const u32 *)uni64((u64)r.p);
    r.idx = uni(r.idx);
    r.nxt = uni(r.nxt);
    r.w = uni64(r.w);
    r.have = uni(r.have);
    r.rem = uni(r.rem);
    r.total = uni(r.total);
    r.rc = uni(r.rc);
    r.fpos = uni(r.fpos);
    r.dry = uni(r.dry);
}
DEV void hdr_uniform(Hdr &h)
{
    rd_uniform(h.r);
    h.ws = (u8 *)uni64((u64)h.ws);
    h.arena_used = uni(h.arena_used);
    h.arena_size = uni(h.arena_size);
    h.tab = uni(h.tab);
    h.glog = uni(h.glog);
    h.lds = (LAS u16 *)(size_t)uni((u32)(size_t)h.lds);
    h.lds_used = uni(h.lds_used);
    h.lds_size = uni(h.lds_size);
    h.rootcaps = uni(h.rootcaps);
    h.kind = uni(h.kind);
}


/* Build the decode table of a complete canonical code (construct():
 * yeast.c:246-278) in the global arena from count[] and the symbols already in
 * canonical order at ws+YS_WS_SORT.  The root table is exactly `root` bits wide
 * (the decode loops know the width from what they are decoding, not from the
 * table).  Layout of the sub-tables: see "LDS mode" below -- the prefixes that
 * need a sub-table are the highest ones, grouped by the depth they need. */
DEV void cnt_set(Cnt16 &k, u32 i, u32 v)
{
    u64 m = (u64)0xffff << ((i & 3) * 16), x = (u64)v << ((i & 3) * 16);
    u32 q = i >> 2;
    k.q0 = q == 0 ? (k.q0 & ~m) | x : k.q0;
    k.q1 = q == 1 ? (k.q1 & ~m) | x : k.q1;
    k.q2 = q == 2 ? (k.q2 & ~m) | x : k.q2;
    k.q3 = q == 3 ? (k.q3 & ~m) | x : k.q3;
}

DEV void build_table(Hdr &h, const Cnt16 &count, u32 root)
{
    h.tab = YS_WS_ARENA + h.arena_used;
    u16 *t = (u16 *)(h.ws + h.tab);
    const u16 *sorted = (const u16 *)(h.ws + YS_WS_SORT);
    const u32 sub = h.sub, glog = h.glog;
    const u32 nroot = 1u << root;

    if (cnt_get(count, 0)) { /* one symbol, zero bits */
        if (h.arena_used + nroot * 2 + 16 > h.arena_size) {
            rd_fail(h.r, YS_WORKSPACE);
            h.tab = YS_WS_ARENA;
            return;
        }
        u32 e = gld16(sorted) << 4;
        for (u32 i = sub; i < nroot; i += 1u << glog)
            gst16(t + i, e);
        h.arena_used += (nroot * 2 + 15) & ~15u;
        return;
    }
    u32 maxlen = 1;
    Cnt16 first; /* canonical first code of each length */
    cnt_clear(first);
    {
        u32 code = 0;
        for (u32 l = 1; l <= 15; l++) {
            code <<= 1;
            cnt_set(first, l, code);
            u32 n = cnt_get(count, l);
            code += n;
            maxlen = n ? l : maxlen;
        }
    }
    /* sub-table groups by depth L: sbase = first entry, plow = first prefix */
    Cnt16 sbase, plow;
    cnt_clear(sbase);
    cnt_clear(plow);
    u32 total = nroot;
    {
        u32 pprev = cnt_get(first, root) + cnt_get(count, root); /* prefixes below hold codes up to root bits */
        for (u32 L = root + 1; L <= maxlen; L++) {
            u32 pl = (cnt_get(first, L) + cnt_get(count, L)) >> (L - root);
            cnt_set(plow, L, pprev);
            cnt_set(sbase, L, total);
            total += (pl - pprev) << (L - root);
            pprev = pl;
        }
    }
    if (total > 4096 || h.arena_used + total * 2 + 16 > h.arena_size) {
        rd_fail(h.r, YS_WORKSPACE); /* redone with 8-bit roots and the worst-case arena */
        h.tab = YS_WS_ARENA;
        return;
    }
    u32 nsyms = 0;
    for (u32 l = 1; l <= 15; l++)
        nsyms += cnt_get(count, l);
    u32 grp = root + 1; /* depth group of the current prefix (prefixes only grow) */
    u32 cur_c = 0xffffffffu, cur_sb = 0;
    u32 len = 0, left_len = 0, code = 0;
    /* the symbols come in canonical order, eight to a load (one load per symbol is one memory round trip per symbol) */
    for (u32 s0 = 0; s0 < nsyms; s0 += 8) {
        const u32x4 sv = gld128u(sorted + s0);
        const u32 nk = nsyms - s0 < 8 ? nsyms - s0 : 8;
        for (u32 k = 0; k < nk; k++, code++, left_len--) {
            while (left_len == 0) {
                len++;
                left_len = cnt_get(count, len);
                code = cnt_get(first, len);
            }
            u32 sw = (k >> 1) == 0 ? sv.x : (k >> 1) == 1 ? sv.y : (k >> 1) == 2 ? sv.z : sv.w;
            u32 s = (k & 1) ? sw >> 16 : sw & 0xffff;
            u32 e = (s << 4) | len;
            if (len <= root) {
                /* the replicas of one code are spread over the lanes of the group */
                u32 rev = __builtin_bitreverse32(code) >> (32 - len);
                for (u32 i = rev + (sub << len); i < nroot; i += 1u << (len + glog))
                    gst16(t + i, e);
            } else {
                u32 sl = len - root;
                u32 c = code >> sl;
                if (c != cur_c) { /* first code under this prefix: place its sub-table, write the link */
                    cur_c = c;
                    while (grp < maxlen && c >= ((cnt_get(first, grp) + cnt_get(count, grp)) >> (grp - root)))
                        grp++;
                    cur_sb = cnt_get(sbase, grp) + ((c - cnt_get(plow, grp)) << (grp - root));
                    if (sub == 0)
                        gst16(t + (__builtin_bitreverse32(c) >> (32 - root)), (cur_sb << 4) | grp);
                }
                u32 rev = __builtin_bitreverse32(code & ((1u << sl) - 1)) >> (32 - sl);
                for (u32 i = rev + (sub << sl); i < (1u << (grp - root)); i += 1u << (sl + glog))
                    gst16(t + cur_sb + i, e);
            }
        }
    }
    h.arena_used += (total * 2 + 15) & ~15u;
}

/* ---- LDS mode ---------------------------------------------------------------------------------
 * Table descriptor (u32): bits 0..15 = offset of the table in the slot (u16 entries), bits 16..19
 * = width of its root table in bits.  Entries are as above (length | symbol << 4); an entry of
 * the root table whose length exceeds the root width is a link: length - root index bits, value =
 * offset of the sub-table from the root, in entries.  The root is min(cap, longest code) bits
 * wide, a zero-bit code is a single entry.
 *
 * Sub-tables: in a canonical code the codes are ordered by length, so the root-width prefixes
 * that need a sub-table are the highest ones and the depth they need grows with the prefix.  With
 * E(L) = first code of length L + number of codes of length L, the prefixes below
 * P(L) = E(L) >> (L - root) hold no code longer than L.  All sizes follow from count[] alone, so
 * the table is laid out before a single entry is written, and every lane can place its own
 * symbols (construct(): yeast.c:246-278 gives the canonical order the ranks below reproduce). */
/* nz: the coded symbols as (length << 12 | symbol), ascending symbols within each length */
DEV void build_table_lds(Hdr &h, const Cnt16 &count, const u16 *nz, u32 nnz, u32 rootcap)
{
    const u32 lane = h.sub;
    Rd &r = h.r;
    h.tab = 0; /* a readable table whatever happens: the slot always holds entries at its start */
    if (cnt_get(count, 0)) { /* one symbol, zero bits */
        if (h.lds_used + 2 > h.lds_size) {
            rd_fail(r, YS_WORKSPACE);
            return;
        }
        u32 s = gld16(nz) & 0xfff;
        if (lane == 0)
            h.lds[h.lds_used] = (u16)(s << 4);
        h.tab = h.lds_used;
        h.lds_used += 2;
        return;
    }
    u32 maxlen = 1;
    Cnt16 first; /* canonical first code of each length */
    cnt_clear(first);
    {
        u32 code = 0;
        for (u32 l = 1; l <= 15; l++) {
            code <<= 1;
            cnt_set(first, l, code);
            u32 n = cnt_get(count, l);
            code += n;
            maxlen = n ? l : maxlen;
        }
    }
    const u32 root = rootcap < maxlen ? rootcap : maxlen;
    /* groups of sub-tables by depth: sbase = where the group starts, plow = its first prefix */
    Cnt16 sbase, plow;
    cnt_clear(sbase);
    cnt_clear(plow);
    u32 total = 1u << root;
    const u32 c_link = cnt_get(first, root) + cnt_get(count, root);
    {
        u32 pprev = c_link;
        for (u32 L = root + 1; L <= maxlen; L++) {
            u32 pl = (cnt_get(first, L) + cnt_get(count, L)) >> (L - root);
            cnt_set(plow, L, pprev);
            cnt_set(sbase, L, total);
            total += (pl - pprev) << (L - root);
            pprev = pl;
        }
    }
    if (total > 4096 || h.lds_used + total > h.lds_size) {
        rd_fail(r, YS_WORKSPACE); /* redone by the kernel that keeps its tables in global memory */
        return;
    }
    LAS u16 *t = h.lds + h.lds_used;
    /* depth group of prefix c: the smallest L above root with c < P(L) */
    auto group_of = [&](u32 c) -> u32 {
        u32 L = root + 1;
        while (L < maxlen && c >= ((cnt_get(first, L) + cnt_get(count, L)) >> (L - root)))
            L++;
        return L;
    };
    for (u32 c = c_link + lane; c < (1u << root); c += 64) {
        u32 L = group_of(c);
        u32 sb = cnt_get(sbase, L) + ((c - cnt_get(plow, L)) << (L - root));
        t[__builtin_bitreverse32(c) >> (32 - root)] = (u16)((sb << 4) | L);
    }
    Cnt16 offs; /* symbols of each length placed so far */
    cnt_clear(offs);
    for (u32 i0 = 0; i0 < nnz; i0 += 64) {
        u32 i = i0 + lane;
        bool active = i < nnz;
        u32 e = active ? gld16(nz + i) : 0;
        u32 len = e >> 12, sym = e & 0xfff;
        u32 rank = 0;
        for (u32 l = 1; l <= maxlen; l++) {
            if (!cnt_get(count, l))
                continue;
            u64 m = __ballot(active && len == l);
            u32 before = __popcll(m & ((1ull << lane) - 1));
            rank = len == l ? cnt_get(offs, l) + before : rank;
            cnt_add(offs, l, __popcll(m));
        }
        if (active) {
            u32 code = cnt_get(first, len) + rank;
            u32 ent = (sym << 4) | len;
            if (len <= root) {
                u32 rev = __builtin_bitreverse32(code) >> (32 - len);
                for (u32 k = rev; k < (1u << root); k += 1u << len)
                    t[k] = (u16)ent;
            } else {
                u32 sl = len - root;
                u32 c = code >> sl;
                u32 L = group_of(c);
                u32 sb = cnt_get(sbase, L) + ((c - cnt_get(plow, L)) << (L - root));
                u32 rev = __builtin_bitreverse32(code & ((1u << sl) - 1)) >> (32 - sl);
                for (u32 k = rev; k < (1u << (L - root)); k += 1u << sl)
                    t[sb + k] = (u16)ent;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    h.tab = h.lds_used | (root << 16);
    h.lds_used += (total + 1) & ~1u;
}

/* dec_sym() for a table in the LDS slot */
DEV u32 dec_sym_lds(Rd &r, LAS const u16 *slot, u32 desc)
{
    rd_refill(r);
    u32 peek = (u32)r.w;
    u32 root = desc >> 16;
    LAS const u16 *t = slot + (desc & 0xffff);
    u32 e = __builtin_amdgcn_readfirstlane(t[peek & ((1u << root) - 1)]);
    u32 l = e & 15;
    if (l > root) {
        e = __builtin_amdgcn_readfirstlane(t[(e >> 4) + ((peek >> root) & ((1u << (l - root)) - 1))]);
        l = e & 15;
    }
    if (l > r.have) {
        rd_eof(r);
        l = 0;
    }
    r.w >>= l;
    r.have -= l;
    return e >> 4;
}

/* root width of the table about to be built: h.kind says what the code is for */
enum { YS_KIND_OTHER = 0, YS_KIND_LIT = 1, YS_KIND_IAC = 2, YS_KIND_DIST = 3 };
DEV u32 rootcap_of(const Hdr &h)
{
    return (h.rootcaps >> (8 * h.kind)) & 255;
}

/* read_code(): prefix() yeast.c:361-542.  num = alphabet size.  Leaves the
   table offset in h.tab (always a readable table, even after a failure). */
template <bool UNI> /* UNI: the whole wavefront decodes this one stream */
NOINL Hdr read_code_t(Hdr h, u32 num)
{
    if (UNI) {
        hdr_uniform(h);
        num = uni(num);
    }
    Rd &r = h.r;
    u16 *sorted = (u16 *)(h.ws + YS_WS_SORT);
    h.tab = YS_WS_ARENA;
#ifdef YS_PROFILE
    u64 tq0 = __builtin_amdgcn_s_memrealtime();
#endif
    u32 hskip = take(r, 2);
    Cnt16 count;
    cnt_clear(count);

    if (hskip == 1) { /* simple code, yeast.c:372-403 and simple() :307-348 */
        u32 abits = 32 - __builtin_clz(num - 1);
        u32 nsym = take(r, 2) + 1;
        u32 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        for (u32 k = 0; k < nsym; k++) {
            u32 v = take(r, abits);
            if (v >= num) {
                rd_fail(r, YS_INVALID); /* :393 */
                return h;
            }
            if (k == 0) s0 = v;
            else if (k == 1) s1 = v;
            else if (k == 2) s2 = v;
            else s3 = v;
        }
        u32 shape = nsym;
        if (nsym == 4)
            shape += take(r, 1);
        if (r.rc)
            return h;
#define ORD(a, b) do { if (a > b) { u32 t_ = a; a = b; b = t_; } } while (0)
        switch (shape) {
        case 1: cnt_add(count, 0, 1); break;
        case 2: cnt_add(count, 1, 2); ORD(s0, s1); break;
        case 3: cnt_add(count, 1, 1); cnt_add(count, 2, 2); ORD(s1, s2); break;
        case 4:
            cnt_add(count, 2, 4);
            ORD(s0, s1); ORD(s2, s3); ORD(s0, s2); ORD(s1, s3); ORD(s1, s2);
            break;
        default:
            cnt_add(count, 1, 1); cnt_add(count, 2, 1); cnt_add(count, 3, 2);
            ORD(s2, s3);
        }
#undef ORD
        if (h.lds) { /* canonical order is length, then symbol: exactly s0..s3 after the sorting above */
            u16 *nz = (u16 *)(h.ws + YS_WS_ARENA);
            u32 sl = shape == 1 ? 0x0000u : shape == 2 ? 0x0011u : shape == 3 ? 0x0221u : shape == 4 ? 0x2222u : 0x3321u;
            gst16(nz, ((sl & 15) << 12) | s0);
            gst16(nz + 1, (((sl >> 4) & 15) << 12) | s1);
            gst16(nz + 2, (((sl >> 8) & 15) << 12) | s2);
            gst16(nz + 3, (((sl >> 12) & 15) << 12) | s3);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            build_table_lds(h, count, nz, shape > 4 ? 4 : shape, rootcap_of(h));
            return h;
        }
        gst16(sorted, s0); gst16(sorted + 1, s1); gst16(sorted + 2, s2); gst16(sorted + 3, s3);
        build_table(h, count, rootcap_of(h));
        return h;
    }

    /* ---- complex code: first the code length code, yeast.c:429-461 ---- */
    /* fixed code for the code length code lengths (yeast.c:422), indexed by 4 stream bits */
    const u64 FIX_SYM = 0x5340234013402340ull; /* entry i at bits 4i: symbol */
    const u64 FIX_LEN = 0x4222322242223222ull; /* entry i at bits 4i: length */
    u64 clen = 0; /* 18 x 3-bit code length code lengths */
    int space = 32;
    u32 nonzero = 0, lastsym = 0;
    u32 nsym = hskip;
    /* order[] of yeast.c:415-417 packed 5 bits each */
    const u64 ORD_LO = (1ull) | (2ull << 5) | (3ull << 10) | (4ull << 15) | (0ull << 20) | (5ull << 25) |
                       (17ull << 30) | (6ull << 35) | (16ull << 40) | (7ull << 45) | (8ull << 50) | (9ull << 55);
    const u64 ORD_HI = (10ull) | (11ull << 5) | (12ull << 10) | (13ull << 15) | (14ull << 20) | (15ull << 25);
    while (nsym < 18) {
        rd_refill(r);
        u32 pk = (u32)r.w & 15;
        u32 l = (u32)(FIX_LEN >> (pk * 4)) & 15;
        u32 s = (u32)(FIX_SYM >> (pk * 4)) & 15;
        if (l > r.have) {
            rd_eof(r);
            return h;
        }
        r.w >>= l; r.have -= l;
        u32 k = nsym < 12 ? (u32)(ORD_LO >> (nsym * 5)) & 31 : (u32)(ORD_HI >> ((nsym - 12) * 5)) & 31;
        nsym++;
        clen |= (u64)s << (k * 3);
        if (s) {
            nonzero++;
            lastsym = k;
            space -= 32 >> s;
            if (space <= 0)
                break;
        }
    }
    if (space < 0 || (space && nonzero != 1)) {
        rd_fail(r, YS_INVALID); /* over-subscribed :451 / incomplete :453 code length code */
        return h;
    }

    /* decode table of the code length code: 32 entries x 8 bits (sym<<3 | len) in registers */
    u64 q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    if (space) { /* single symbol, zero bits :456-459 */
        u64 e = (u64)(lastsym << 3) * 0x0101010101010101ull;
        q0 = q1 = q2 = q3 = e;
    } else {
        u32 code = 0;
        for (u32 len = 1; len <= 5; len++) {
            code <<= 1;
            for (u32 s = 0; s < 18; s++) {
                if (((u32)(clen >> (s * 3)) & 7) != len)
                    continue;
                u32 rev = __builtin_bitreverse32(code) >> (32 - len);
                u64 e = (u64)((s << 3) | len);
                for (u32 i = rev; i < 32; i += 1u << len) {
                    u64 m = e << ((i & 7) * 8);
                    u32 q = i >> 3;
                    q0 |= q == 0 ? m : 0;
                    q1 |= q == 1 ? m : 0;
                    q2 |= q == 2 ? m : 0;
                    q3 |= q == 3 ? m : 0;
                }
                code++;
            }
        }
    }

#ifdef YS_PROFILE
    u64 tq1 = __builtin_amdgcn_s_memrealtime();
#endif
    /* ---- the code lengths themselves, yeast.c:464-518 ---- */
    if (!h.lds && h.arena_used + 704 * 2 + 16 + YS_TAB_ENTRIES * 2 > h.arena_size) {
        rd_fail(r, YS_WORKSPACE);
        return h;
    }
    /* coded symbols as (length << 12 | symbol), in symbol order; scratch above the table's place
       (LDS mode: at the start of the global arena, which holds nothing else) */
    u16 *nz = (u16 *)(h.ws + YS_WS_ARENA + (h.lds ? 0 : h.arena_used + YS_TAB_ENTRIES * 2));
    u32 nnz = 0;
    int left = 1 << 15;
    u32 prev = 8, rep = 0, zeros = 0, n;
    nsym = 0;
    do {
        rd_refill(r);
        u32 pk = (u32)r.w & 31;
        u32 q = pk >> 3;
        u64 x = q == 0 ? q0 : q == 1 ? q1 : q == 2 ? q2 : q3;
        u32 e = (u32)(x >> ((pk & 7) * 8)) & 255;
        u32 l = e & 7;
        if (l > r.have) {
            rd_eof(r);
            return h;
        }
        r.w >>= l; r.have -= l;
        u32 len = e >> 3;
        if (len < 16) {
            if (len) {
                gst16(nz + nnz++, (len << 12) | nsym);
                left -= (1 << 15) >> len;
                prev = len;
                cnt_add(count, len, 1);
            }
            nsym++;
            rep = 0;
            zeros = 0;
        } else if (len == 16) {
            n = rep;
            rep = (rep ? (rep - 2) << 2 : 0) + 3 + take(r, 2);
            n = rep - n;
            if (r.rc)
                return h;
            if (nsym + n > num) {
                rd_fail(r, YS_INVALID); /* too many repeats :491 */
                return h;
            }
            left -= (int)n * ((1 << 15) >> prev);
            if (left < 0)
                break;
            cnt_add(count, prev, n);
            for (u32 k = 0; k < n; k++)
                gst16(nz + nnz++, (prev << 12) | nsym++);
            zeros = 0;
        } else {
            n = zeros;
            zeros = (zeros ? (zeros - 2) << 3 : 0) + 3 + take(r, 3);
            n = zeros - n;
            if (r.rc)
                return h;
            if (nsym + n > num) {
                rd_fail(r, YS_INVALID); /* too many repeats :508 */
                return h;
            }
            nsym += n;
            rep = 0;
        }
    } while (left > 0 && nsym < num);
    if (left != 0) {
        rd_fail(r, YS_INVALID); /* incomplete :516 / over-subscribed :518 code */
        return h;
    }

    if (h.lds) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        build_table_lds(h, count, nz, nnz, rootcap_of(h));
        return h;
    }
#ifdef YS_PROFILE
    u64 tq2 = __builtin_amdgcn_s_memrealtime();
#endif
    /* counting sort into canonical order (construct() yeast.c:266-277) */
    Cnt16 offs;
    cnt_clear(offs);
    {
        u32 acc = 0;
        for (u32 l = 1; l <= 15; l++) {
            cnt_add(offs, l, acc);
            acc += cnt_get(count, l);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (u32 i0 = 0; i0 < nnz; i0 += 8) { /* eight entries to a load */
        const u32x4 nv = gld128u(nz + i0);
        const u32 nk = nnz - i0 < 8 ? nnz - i0 : 8;
        for (u32 k = 0; k < nk; k++) {
            u32 w = (k >> 1) == 0 ? nv.x : (k >> 1) == 1 ? nv.y : (k >> 1) == 2 ? nv.z : nv.w;
            u32 e = (k & 1) ? w >> 16 : w & 0xffff;
            u32 l = e >> 12;
            gst16(sorted + cnt_get(offs, l), e & 0xfff);
            cnt_add(offs, l, 1);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#ifdef YS_PROFILE
    u64 tq3 = __builtin_amdgcn_s_memrealtime();
#endif
    build_table(h, count, rootcap_of(h));
#ifdef YS_PROFILE
    h.tp0 += tq1 - tq0; h.tp1 += tq2 - tq1; h.tp2 += tq3 - tq2; h.tp3 += __builtin_amdgcn_s_memrealtime() - tq3;
#endif
    return h;
}

/* block_types(): yeast.c:576-584 */
DEV u32 read_count256(Rd &r)
{
    if (!take(r, 1))
        return 1;
    u32 k = take(r, 3);
    return 1 + (1u << k) + take(r, k);
}

/* block_length(): yeast.c:550-567; base[]/extra[] from their closed form */
DEV u32 read_block_count(Rd &r, const u16 *t)
{
    u32 s = dec_sym(r, t);
    u32 base, extra;
    if (s < 4) { base = 1 + 4 * s; extra = 2; }
    else if (s < 8) { base = 17 + 8 * (s - 4); extra = 3; }
    else if (s < 12) { base = 49 + 16 * (s - 8); extra = 4; }
    else if (s < 16) { base = 113 + 32 * (s - 12); extra = 5; }
    else if (s < 18) { base = 241 + 64 * (s - 16); extra = 6; }
    else if (s >= 25) { base = 16625; extra = 24; }
    else { extra = s - 11; base = 369 + ((1u << (s - 11)) - 128); }
    return base + take(r, extra);
}

/* context_map(): yeast.c:589-647.  map has room for len bytes. */
DEV Hdr read_code(Hdr h, u32 num) { return read_code_t<false>(h, num); }

template <bool UNI>
NOINL Hdr read_context_map_t(Hdr h, u8 *map, u32 len, u32 trees)
{
    if (UNI) {
        hdr_uniform(h);
        map = (u8 *)uni64((u64)map);
        len = uni(len);
        trees = uni(trees);
    }
    Rd &r = h.r;
    u32 rlemax = 0;
    if (take(r, 1))
        rlemax = 1 + take(r, 4);
    u32 save = h.arena_used, save_lds = h.lds_used;
    h = read_code_t<UNI>(h, rlemax + trees);
    if (UNI)
        hdr_uniform(h);
    const u16 *t = (const u16 *)(h.ws + h.tab);
    const u32 desc = h.tab;
    h.arena_used = save; /* the map's own code is dead once the map is read */
    h.lds_used = save_lds;
    if (r.rc)
        return h;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    u32 n = 0;
    do {
        u32 s = h.lds ? dec_sym_lds(r, h.lds, desc) : dec_sym(r, t);
        if (r.rc)
            return h;
        if (s == 0)
            gst8(map + n++, 0);
        else if (s <= rlemax) {
            u32 z = (1u << s) + take(r, s);
            if (r.rc)
                return h;
            if (n + z > len) {
                rd_fail(r, YS_INVALID); /* run length too long :616 */
                return h;
            }
            for (u32 k = 0; k < z; k++)
                gst8(map + n++, 0);
        } else
            gst8(map + n++, s - rlemax);
    } while (n < len);
    if (take(r, 1)) { /* inverse move-to-front :629-646 */
        u8 *tab = h.ws + YS_WS_MTF;
        for (u32 k = 0; k < trees; k++)
            gst8(tab + k, k);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (n = 0; n < len; n++) {
            u32 s = gld8(map + n);
            if (s >= trees)
                s = 0; /* unreachable for a valid map; keeps the list index in range */
            if (s) {
                u32 front = gld8(tab + s);
                for (u32 k = s; k; k--)
                    gst8(tab + k, gld8(tab + k - 1));
                gst8(tab, front);
                gst8(map + n, front);
            } else
                gst8(map + n, gld8(tab));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return h;
}

DEV Hdr read_context_map(Hdr h, u8 *map, u32 len, u32 trees) { return read_context_map_t<false>(h, map, len, trees); }


#endif
