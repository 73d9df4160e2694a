/*
 * yeast_kernels.hip -- MI355X (gfx950) Brotli decode kernels.
 *
 * What this replaces in the reference (paths relative to /root/reference):
 *   yeast.c:163-179   bits()            -> Bits (64-bit window, prefetched dword refills)
 *   yeast.c:204-219   decode()          -> dec_sym() (two-level LUT, 8-bit root)
 *   yeast.c:246-348   construct/simple  -> build_table()
 *   yeast.c:361-542   prefix()          -> read_code()
 *   yeast.c:550-647   block_length/block_types/context_map
 *   yeast.c:652-789   insert/copy lengths, context ids, distance()
 *   yeast.c:795-949   dict_word() + xforms.h + dict.h -> emit_dict()
 *   yeast.c:971-1308  metablock()       -> metablock()
 *   yeast.c:1353-1400 yeast()           -> decode_stream()
 *
 * Execution model: a *group* of G lanes (G = 64, 16, 8, 4 ... a power of two)
 * decodes one stream.  Everything that is a function of the bit stream alone
 * is computed redundantly by every lane of the group (so it needs no
 * cross-lane traffic and no branches on the lane id); the lanes differ only
 * when bytes are moved: LZ77 window copies, dictionary words and stored
 * blocks are spread over the G lanes.  Groups pull streams from an atomic
 * work queue, so a launch is a fixed-size grid of resident groups.
 */
#include <hip/hip_runtime.h>
#include "yeast_common.h"

#define DEV __device__ __forceinline__

__constant__ u8 c_ctx_lut[768]; /* lut0 | lut1 | lut2 of yeast.c:700-740, filled by the host */
__constant__ u32 c_x2n[68];     /* x^(2^k) mod P for the CRC-32C combine */

/* ------------------------------------------------------------------ state */

struct Bits {
    const u32 *p; /* 4-byte aligned base at or before the stream */
    u32 idx;      /* dword index of the next prefetch */
    u32 nxt;      /* prefetched dword */
    u64 w;        /* window, next bit at bit 0 */
    u32 have;     /* bits in w (may extend past the end of the stream) */
    u64 pos;      /* bits consumed since the start of the stream */
    u64 nbits;    /* 8 * stream length */
};

struct Blk { /* one block category: yeast.c:112-123 */
    u32 num, type, prev, left;
    const u16 *t_type, *t_count;
};

struct Ctx {
    Bits b;
    const u8 *src;
    u8 *out;
    u64 got, cap;
    u32 rc;
    u32 wsize;
    u32 r0, r1, r2, r3; /* last four distances, r0 most recent (ring[] yeast.c:126) */
    u32 postfix, direct;
    Blk L, I, D;
    u32 lit_codes, dist_codes;
    u8 *ws;
    u32 arena_used, arena_size; /* bytes */
    u32 p1, p2;
    u32 nmeta;
    u32 dry;  /* the bit reader ran out of input: every byte was pulled */
    u64 need; /* output bytes required when rc == YS_CAPACITY */
    const u8 *dict, *xf;
    const u8 *lut; /* LDS copy of the three context tables */
};

#define FAIL(c, code)       \
    do {                    \
        (c).rc = (code);    \
        return false;       \
    } while (0)
#define CK(x)             \
    do {                  \
        if (!(x))         \
            return false; \
    } while (0)

/* ------------------------------------------------------------- bit reader */

DEV void bits_seek(Bits &b, const u8 *src, u64 bitpos)
{
    const u8 *at = src + (bitpos >> 3);
    u32 mis = (u32)((uintptr_t)at & 3);
    b.p = (const u32 *)(at - mis);
    u32 sh = mis * 8 + (u32)(bitpos & 7);
    u64 lo = b.p[0], hi = b.p[1];
    b.w = (lo | (hi << 32)) >> sh;
    b.have = 64 - sh;
    b.nxt = b.p[2];
    b.idx = 3;
    b.pos = bitpos;
}

DEV void bits_refill(Bits &b)
{
    if (b.have <= 32) {
        b.w |= (u64)b.nxt << b.have;
        b.have += 32;
        b.nxt = b.p[b.idx++];
    }
}

/* bits(): yeast.c:163-179.  n <= 25.  On exhaustion the reference has pulled
   every input byte (yeast.c:169-175), which decode_stream() reports. */
DEV bool take(Ctx &c, u32 n, u32 &v)
{
    Bits &b = c.b;
    if (b.pos + n > b.nbits) {
        c.dry = 1;
        FAIL(c, YS_EOF);
    }
    bits_refill(b);
    v = (u32)b.w & ((1u << n) - 1);
    b.w >>= n;
    b.have -= n;
    b.pos += n;
    return true;
}

/* "s->left && s->bits": the rest of the current byte is not zero */
DEV bool tail_nonzero(const Ctx &c)
{
    u32 k = (u32)(c.b.pos & 7);
    return k && ((u32)c.b.w & ((1u << (8 - k)) - 1));
}

/* -------------------------------------------------------------- LUT decode
 * Table entry (u16): bits 0..3 = code length, bits 4..15 = symbol.  In the
 * 256-entry root a length above 8 marks a link: (length - 8) index bits,
 * value = offset of the sub-table from the root, in entries.
 */
#define ROOT 8

DEV bool dec_sym(Ctx &c, const u16 *t, u32 &sym)
{
    Bits &b = c.b;
    bits_refill(b);
    u32 peek = (u32)b.w;
    u32 e = t[peek & 255];
    u32 l = e & 15;
    if (l > ROOT) {
        e = t[(e >> 4) + ((peek >> ROOT) & ((1u << (l - ROOT)) - 1))];
        l = e & 15;
    }
    if (b.pos + l > b.nbits) {
        c.dry = 1;
        FAIL(c, YS_EOF);
    }
    b.w >>= l;
    b.have -= l;
    b.pos += l;
    sym = e >> 4;
    return true;
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

DEV const u16 *arena_tables(const Ctx &c) { return (const u16 *)(c.ws + YS_WS_ARENA); }

/* Build the decode table of a complete canonical code (construct():
 * yeast.c:246-278) from count[] and the symbols already in canonical order at
 * ws+YS_WS_SORT.  Returns the table (in the arena) through *out.  The sub-table
 * width rule is the usual one: the smallest width that the remaining counts can
 * fill. */
#define YS_TAB_ENTRIES 1104
DEV bool build_table(Ctx &c, const Cnt16 &count, const u16 **out)
{
    if (c.arena_used + YS_TAB_ENTRIES * 2 > c.arena_size)
        FAIL(c, YS_WORKSPACE);
    u16 *t = (u16 *)(c.ws + YS_WS_ARENA + c.arena_used);
    const u16 *sorted = (const u16 *)(c.ws + YS_WS_SORT);
    *out = t;

    if (cnt_get(count, 0)) { /* one symbol, zero bits */
        u32 e = (u32)sorted[0] << 4;
        for (u32 i = 0; i < 256; i++)
            t[i] = (u16)e;
        c.arena_used += 512;
        return true;
    }
    Cnt16 left = count; /* codes not yet placed, per length */
    u32 code = 0;       /* canonical code, MSB first, of the current length */
    u32 si = 0;
    u32 total = 256; /* entries used */
    u32 sub_base = 0, sub_bits = 0, sub_key = 0xffffffffu;
    for (u32 len = 1; len <= 15; len++) {
        code <<= 1;
        u32 n = cnt_get(count, len);
        for (u32 k = 0; k < n; k++, si++, code++) {
            u32 s = sorted[si];
            u32 rev = __builtin_bitreverse32(code) >> (32 - len); /* stream bit order */
            u32 e = (s << 4) | len;
            if (len <= ROOT) {
                for (u32 i = rev; i < 256; i += 1u << len)
                    t[i] = (u16)e;
            } else {
                u32 key = rev & 255;
                if (key != sub_key) { /* open the next sub-table */
                    sub_key = key;
                    sub_base = total;
                    u32 l2 = len;
                    i64 room = (i64)1 << (l2 - ROOT);
                    /* codes of this length still to come, this one included */
                    Cnt16 tmp = left;
                    while (l2 < 15) {
                        room -= cnt_get(tmp, l2);
                        if (room <= 0)
                            break;
                        l2++;
                        room <<= 1;
                    }
                    sub_bits = l2 - ROOT;
                    total += 1u << sub_bits;
                    if (total > YS_TAB_ENTRIES)
                        FAIL(c, YS_WORKSPACE);
                    t[key] = (u16)((sub_base << 4) | (ROOT + sub_bits));
                }
                for (u32 i = rev >> ROOT; i < (1u << sub_bits); i += 1u << (len - ROOT))
                    t[sub_base + i] = (u16)e;
            }
            cnt_dec(left, len);
        }
    }
    c.arena_used += (total * 2 + 15) & ~15u;
    return true;
}

/* read_code(): prefix() yeast.c:361-542.  num = alphabet size. */
DEV bool read_code(Ctx &c, u32 num, const u16 **out)
{
    u16 *sorted = (u16 *)(c.ws + YS_WS_SORT);
    u32 hskip, v;
    CK(take(c, 2, hskip));
    Cnt16 count;
    cnt_clear(count);

    if (hskip == 1) { /* simple code, yeast.c:372-403 and simple() :307-348 */
        u32 abits = 32 - __builtin_clz(num - 1);
        if (abits == 0)
            abits = 1;
        u32 nsym;
        CK(take(c, 2, nsym));
        nsym += 1;
        u32 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        for (u32 k = 0; k < nsym; k++) {
            CK(take(c, abits, v));
            if (v >= num)
                FAIL(c, YS_INVALID);
            if (k == 0) s0 = v;
            else if (k == 1) s1 = v;
            else if (k == 2) s2 = v;
            else s3 = v;
        }
        u32 shape = nsym;
        if (nsym == 4) {
            CK(take(c, 1, v));
            shape += v;
        }
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
        sorted[0] = (u16)s0; sorted[1] = (u16)s1; sorted[2] = (u16)s2; sorted[3] = (u16)s3;
        return build_table(c, count, out);
    }

    /* ---- complex code: first the code length code, yeast.c:429-461 ---- */
    /* fixed code for the code length code lengths (yeast.c:422), indexed by 4 stream bits:
       value nibble and length nibble per entry */
    const u64 FIX_SYM = 0x5340234013402340ull; /* entry i at bits 4i: symbol */
    const u64 FIX_LEN = 0x4222322242223222ull; /* entry i at bits 4i: length */
    u64 clen = 0; /* 18 x 3-bit code length code lengths */
    i64 space = 32;
    u32 nonzero = 0, lastsym = 0;
    u32 nsym = hskip;
    /* order[] of yeast.c:415-417 packed 5 bits each */
    const u64 ORD_LO = (1ull) | (2ull << 5) | (3ull << 10) | (4ull << 15) | (0ull << 20) | (5ull << 25) |
                       (17ull << 30) | (6ull << 35) | (16ull << 40) | (7ull << 45) | (8ull << 50) | (9ull << 55);
    const u64 ORD_HI = (10ull) | (11ull << 5) | (12ull << 10) | (13ull << 15) | (14ull << 20) | (15ull << 25);
    while (nsym < 18) {
        Bits &b = c.b;
        bits_refill(b);
        u32 pk = (u32)b.w & 15;
        u32 l = (u32)(FIX_LEN >> (pk * 4)) & 15;
        u32 s = (u32)(FIX_SYM >> (pk * 4)) & 15;
        if (b.pos + l > b.nbits) {
            c.dry = 1;
            FAIL(c, YS_EOF);
        }
        b.w >>= l; b.have -= l; b.pos += l;
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
    if (space < 0)
        FAIL(c, YS_INVALID); /* oversubscribed code length code :451 */
    if (space && nonzero != 1)
        FAIL(c, YS_INVALID); /* incomplete code length code :453 */

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

    /* ---- the code lengths themselves, yeast.c:464-518; counting sort on the fly ---- */
    u8 *lens = (u8 *)(c.ws + YS_WS_ARENA + c.arena_used); /* scratch just above the arena top */
    if (c.arena_used + 704 + 16 + YS_TAB_ENTRIES * 2 > c.arena_size)
        FAIL(c, YS_WORKSPACE);
    i64 left = 1 << 15;
    u32 prev = 8, rep = 0, zeros = 0, n;
    nsym = 0;
    do {
        Bits &b = c.b;
        bits_refill(b);
        u32 pk = (u32)b.w & 31;
        u32 q = pk >> 3;
        u64 x = q == 0 ? q0 : q == 1 ? q1 : q == 2 ? q2 : q3;
        u32 e = (u32)(x >> ((pk & 7) * 8)) & 255;
        u32 l = e & 7;
        if (b.pos + l > b.nbits) {
            c.dry = 1;
            FAIL(c, YS_EOF);
        }
        b.w >>= l; b.have -= l; b.pos += l;
        u32 len = e >> 3;
        if (len < 16) {
            lens[nsym++] = (u8)len;
            if (len) {
                left -= (1 << 15) >> len;
                prev = len;
                cnt_add(count, len, 1);
            }
            rep = 0;
            zeros = 0;
        } else if (len == 16) {
            n = rep;
            CK(take(c, 2, v));
            rep = (rep ? (rep - 2) << 2 : 0) + 3 + v;
            n = rep - n;
            if (nsym + n > num)
                FAIL(c, YS_INVALID); /* too many repeats :491 */
            left -= (i64)n * ((1 << 15) >> prev);
            if (left < 0)
                break;
            cnt_add(count, prev, n);
            for (u32 k = 0; k < n; k++)
                lens[nsym++] = (u8)prev;
            zeros = 0;
        } else {
            n = zeros;
            CK(take(c, 3, v));
            zeros = (zeros ? (zeros - 2) << 3 : 0) + 3 + v;
            n = zeros - n;
            if (nsym + n > num)
                FAIL(c, YS_INVALID); /* too many repeats :508 */
            for (u32 k = 0; k < n; k++)
                lens[nsym++] = 0;
            rep = 0;
        }
    } while (left > 0 && nsym < num);
    if (left > 0)
        FAIL(c, YS_INVALID); /* incomplete code :516 */
    if (left < 0)
        FAIL(c, YS_INVALID); /* oversubscribed code :518 */

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
    for (u32 s = 0; s < nsym; s++) {
        u32 l = lens[s];
        if (l) {
            sorted[cnt_get(offs, l)] = (u16)s;
            cnt_add(offs, l, 1);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    return build_table(c, count, out);
}

/* block_types(): yeast.c:576-584 */
DEV bool read_count256(Ctx &c, u32 &n)
{
    u32 v, k;
    CK(take(c, 1, v));
    if (!v) {
        n = 1;
        return true;
    }
    CK(take(c, 3, k));
    CK(take(c, k, v));
    n = 1 + (1u << k) + v;
    return true;
}

/* block_length(): yeast.c:550-567 */
DEV bool read_block_count(Ctx &c, const u16 *t, u32 &cnt)
{
    /* base[] and extra[] of the 26 block count codes, from their closed form */
    u32 s, v;
    CK(dec_sym(c, t, s));
    u32 base, extra;
    if (s < 4) { base = 1 + 4 * s; extra = 2; }
    else if (s < 8) { base = 17 + 8 * (s - 4); extra = 3; }
    else if (s < 12) { base = 49 + 16 * (s - 8); extra = 4; }
    else if (s < 16) { base = 113 + 32 * (s - 12); extra = 5; }
    else if (s < 18) { base = 241 + 64 * (s - 16); extra = 6; }
    else {
        /* 369/7 497/8 753/9 1265/10 2289/11 4337/12 8433/13 16625/24 */
        extra = s == 25 ? 24 : s - 11;
        base = 369 + ((1u << (s - 11)) - 128);
        if (s == 25) base = 16625;
    }
    CK(take(c, extra, v));
    cnt = base + v;
    return true;
}

DEV bool read_block_header(Ctx &c, Blk &k, u32 dir_type, u32 dir_count)
{
    (void)dir_type; (void)dir_count;
    k.prev = 1;
    k.type = 0;
    CK(read_count256(c, k.num));
    if (k.num > 1) {
        CK(read_code(c, k.num + 2, &k.t_type));
        CK(read_code(c, 26, &k.t_count));
        CK(read_block_count(c, k.t_count, k.left));
    } else
        k.left = 0xffffffffu;
    return true;
}

/* block switch: yeast.c:1168-1180 */
DEV bool switch_block(Ctx &c, Blk &k)
{
    u32 n;
    CK(dec_sym(c, k.t_type, n));
    n = n > 1 ? n - 2 : n ? (k.type + 1 == k.num ? 0 : k.type + 1) : k.prev;
    k.prev = k.type;
    k.type = n;
    return read_block_count(c, k.t_count, k.left);
}

/* context_map(): yeast.c:589-647 */
DEV bool read_context_map(Ctx &c, u8 *map, u32 len, u32 trees)
{
    u32 v, rlemax = 0;
    CK(take(c, 1, v));
    if (v) {
        CK(take(c, 4, v));
        rlemax = 1 + v;
    }
    u32 save = c.arena_used;
    const u16 *t;
    CK(read_code(c, rlemax + trees, &t));
    u32 n = 0;
    do {
        u32 s;
        CK(dec_sym(c, t, s));
        if (s == 0)
            map[n++] = 0;
        else if (s <= rlemax) {
            CK(take(c, s, v));
            u32 z = (1u << s) + v;
            if (n + z > len)
                FAIL(c, YS_INVALID); /* run length too long :616 */
            for (u32 k = 0; k < z; k++)
                map[n++] = 0;
        } else
            map[n++] = (u8)(s - rlemax);
    } while (n < len);
    c.arena_used = save; /* the map's own code is dead now */
    CK(take(c, 1, v));
    if (v) { /* inverse move-to-front :629-646 */
        u8 *tab = c.ws + YS_WS_MTF;
        for (u32 k = 0; k < trees; k++)
            tab[k] = (u8)k;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        for (n = 0; n < len; n++) {
            u32 s = map[n];
            if (s) {
                u8 front = tab[s];
                for (u32 k = s; k; k--)
                    tab[k] = tab[k - 1];
                tab[0] = front;
                map[n] = front;
            } else
                map[n] = tab[0];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    return true;
}

/* insert_length()/copy_length(): yeast.c:652-693, tables packed in registers */
DEV void len_code(u32 code, bool is_copy, u32 &base, u32 &extra)
{
    /* extra bits per code, 5 bits each: codes 0..11 and 12..23 */
    const u64 INS_X0 = 0ull | (0ull << 5) | (0ull << 10) | (0ull << 15) | (0ull << 20) | (0ull << 25) |
                       (1ull << 30) | (1ull << 35) | (2ull << 40) | (2ull << 45) | (3ull << 50) | (3ull << 55);
    const u64 INS_X1 = 4ull | (4ull << 5) | (5ull << 10) | (5ull << 15) | (6ull << 20) | (7ull << 25) |
                       (8ull << 30) | (9ull << 35) | (10ull << 40) | (12ull << 45) | (14ull << 50) | (24ull << 55);
    const u64 CPY_X0 = 0ull | (0ull << 5) | (0ull << 10) | (0ull << 15) | (0ull << 20) | (0ull << 25) |
                       (0ull << 30) | (0ull << 35) | (1ull << 40) | (1ull << 45) | (2ull << 50) | (2ull << 55);
    const u64 CPY_X1 = 3ull | (3ull << 5) | (4ull << 10) | (4ull << 15) | (5ull << 20) | (5ull << 25) |
                       (6ull << 30) | (7ull << 35) | (8ull << 40) | (9ull << 45) | (10ull << 50) | (24ull << 55);
    u64 x = is_copy ? (code < 12 ? CPY_X0 : CPY_X1) : (code < 12 ? INS_X0 : INS_X1);
    extra = (u32)(x >> ((code < 12 ? code : code - 12) * 5)) & 31;
    /* bases: running sums of 1<<extra from 0 (insert) or 2 (copy); closed forms below */
    if (!is_copy) {
        /* 0,1,2,3,4,5,6,8,10,14,18,26,34,50,66,98,130,194,322,578,1090,2114,6210,22594 */
        if (code < 6) base = code;
        else if (code < 8) base = 6 + 2 * (code - 6);
        else if (code < 10) base = 10 + 4 * (code - 8);
        else if (code < 12) base = 18 + 8 * (code - 10);
        else if (code < 14) base = 34 + 16 * (code - 12);
        else if (code < 16) base = 66 + 32 * (code - 14);
        else if (code == 16) base = 130;
        else if (code == 17) base = 194;
        else if (code == 18) base = 322;
        else if (code == 19) base = 578;
        else if (code == 20) base = 1090;
        else if (code == 21) base = 2114;
        else if (code == 22) base = 6210;
        else base = 22594;
    } else {
        /* 2,3,4,5,6,7,8,9,10,12,14,18,22,30,38,54,70,102,134,198,326,582,1094,2118 */
        if (code < 8) base = 2 + code;
        else if (code < 10) base = 10 + 2 * (code - 8);
        else if (code < 12) base = 14 + 4 * (code - 10);
        else if (code < 14) base = 22 + 8 * (code - 12);
        else if (code < 16) base = 38 + 16 * (code - 14);
        else if (code < 18) base = 70 + 32 * (code - 16);
        else if (code == 18) base = 134;
        else if (code == 19) base = 198;
        else if (code == 20) base = 326;
        else if (code == 21) base = 582;
        else if (code == 22) base = 1094;
        else base = 2118;
    }
}

/* context_id(): yeast.c:698-755 */
DEV u32 context_id(const Ctx &c, u32 p1, u32 p2, u32 mode)
{
    switch (mode) {
    case 0: return p1 & 0x3f;
    case 1: return p1 >> 2;
    case 2: return c.lut[p1] | c.lut[256 + p2];
    default: return ((u32)c.lut[512 + p1] << 3) | c.lut[512 + p2];
    }
}

/* ------------------------------------------------------------ byte movers */

/* Store (or, in compare mode, check) one byte that every lane of the group
   holds.  Returns false on mismatch. */
template <int G, bool CMP>
DEV bool emit_byte(Ctx &c, u32 v)
{
    if (CMP) {
        u32 e = c.out[c.got];
        c.got++; /* yeast.c:1217: got advances before the compare throws */
        if (e != v)
            FAIL(c, YS_MISMATCH);
    } else {
        c.out[c.got] = (u8)v;
        c.got++;
    }
    return true;
}

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
   semantics, so an overlapping copy repeats its first d bytes. */
template <int G, bool CMP>
DEV bool emit_copy(Ctx &c, u32 n, u64 d, u32 sub)
{
    u8 *dst = c.out + c.got;
    const u8 *from = dst - d;
    if (CMP) {
        /* every byte is compared with the expected data itself d bytes back */
        u32 bad = 0xffffffffu;
        for (u32 i = sub; i < n; i += G)
            if (dst[i] != from[i] && i < bad)
                bad = i;
        bad = group_min<G>(bad);
        if (bad != 0xffffffffu) {
            c.got += bad;
            FAIL(c, YS_MISMATCH);
        }
        c.got += n;
        return true;
    }
    if (d >= n) { /* disjoint: 8-byte units spread over the lanes, then the tail */
        u32 n8 = n >> 3;
        for (u32 j = sub; j < n8; j += G) {
            u64 v;
            __builtin_memcpy(&v, from + 8 * j, 8);
            __builtin_memcpy(dst + 8 * j, &v, 8);
        }
        for (u32 i = (n8 << 3) + sub; i < n; i += G)
            dst[i] = from[i];
    } else if (G == 1 && d >= 8) {
        u32 i = 0;
        for (; i + 8 <= n; i += 8) {
            u64 v;
            __builtin_memcpy(&v, from + i, 8);
            __builtin_memcpy(dst + i, &v, 8);
        }
        for (; i < n; i++)
            dst[i] = from[i];
    } else { /* overlapping: byte i repeats byte i mod d of the source */
        u32 dd = (u32)d;
        u32 m = sub % dd, step = G % dd;
        for (u32 i = sub; i < n; i += G) {
            dst[i] = from[m];
            m += step;
            if (m >= dd)
                m -= dd;
        }
    }
    c.got += n;
    return true;
}

/* dict_word(): yeast.c:908-949 with the elementary transforms :799-879.
   Byte k of the result is computed independently by whichever lane owns it. */
template <int G, bool CMP>
DEV bool emit_dict(Ctx &c, u32 copy, u64 id, u32 &mlen, u32 sub)
{
    if (copy > 24 || copy < 4)
        FAIL(c, YS_INVALID);
    /* NDBITS / DOFFSET (yeast.c:911-919) */
    const u64 NDB_LO = (10ull << 16) | (10ull << 20) | (11ull << 24) | (11ull << 28) | (10ull << 32) |
                       (10ull << 36) | (10ull << 40) | (10ull << 44) | (10ull << 48) | (9ull << 52) |
                       (9ull << 56) | (8ull << 60);                /* lengths 4..15 at bits 4*len */
    const u64 NDB_HI = 7ull | (7ull << 4) | (8ull << 8) | (7ull << 12) | (7ull << 16) | (6ull << 20) |
                       (6ull << 24) | (5ull << 28) | (5ull << 32); /* lengths 16..24 */
    u32 nd = copy < 16 ? (u32)(NDB_LO >> (copy * 4)) & 15 : (u32)(NDB_HI >> ((copy - 16) * 4)) & 15;
    u64 xform = id >> nd;
    if (xform >= 121)
        FAIL(c, YS_INVALID);
    u32 index = (u32)id & ((1u << nd) - 1);
    /* DOFFSET[len] = sum over l<len of l << NDBITS[l] */
    u32 doff = 0;
    for (u32 l = 4; l < copy; l++) {
        u32 b = l < 16 ? (u32)(NDB_LO >> (l * 4)) & 15 : (u32)(NDB_HI >> ((l - 16) * 4)) & 15;
        doff += l << b;
    }
    const u8 *w = c.dict + doff + index * copy;
    const u8 *t = c.xf + 20 * (u32)xform;
    u32 hdr;
    __builtin_memcpy(&hdr, t, 4);
    u32 op = hdr & 255, omit = (hdr >> 8) & 255, plen = (hdr >> 16) & 255, slen = hdr >> 24;
    u32 wlen = copy, wskip = 0;
    if (op == 3) { /* omit first */
        wskip = omit;
        wlen = copy > omit ? copy - omit : 0;
        if (wlen == 0) wskip = 0;
    } else if (op == 4)
        wlen = copy > omit ? copy - omit : 0;
    u32 total = plen + wlen + slen;
    if (total > mlen)
        FAIL(c, YS_INVALID); /* :1269 */

    /* uppercase masks: bit j set -> word byte j is xor-ed with 32 / with 5 */
    u32 x32 = 0, x5 = 0;
    if (op == 1 || op == 2) {
        u32 j = 0;
        do {
            u32 ch = w[j];
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
    u8 *dst = c.out + c.got;
    u32 bad = 0;
    for (u32 k = sub; k < total; k += G) {
        u32 v;
        if (k < plen)
            v = t[4 + k];
        else if (k < plen + wlen) {
            u32 j = k - plen + wskip;
            v = w[j];
            v ^= ((x32 >> j) & 1) << 5;
            v ^= ((x5 >> j) & 1) * 5;
        } else
            v = t[12 + k - plen - wlen];
        if (CMP)
            bad |= dst[k] != v;
        else
            dst[k] = (u8)v;
    }
    if (CMP) {
        if (G > 1)
            bad = __any(bad) ? 1 : 0; /* whole-wave vote: CMP is only instantiated with G == 64 */
        if (bad)
            FAIL(c, YS_MISMATCH); /* memcmp semantics: got stays at the start of the word :1271 */
    }
    c.got += total;
    mlen -= total;
    return true;
}

/* stored block / any plain byte run from the input */
template <int G, bool CMP>
DEV bool emit_raw(Ctx &c, const u8 *from, u32 n, u32 sub)
{
    u8 *dst = c.out + c.got;
    if (CMP) {
        u32 bad = 0;
        for (u32 i = sub; i < n; i += G)
            bad |= dst[i] != from[i];
        if (G > 1)
            bad = __any(bad) ? 1 : 0;
        if (bad)
            FAIL(c, YS_MISMATCH); /* memcmp over the block: got unchanged :1063 */
    } else {
        u32 n8 = n >> 3;
        for (u32 j = sub; j < n8; j += G) {
            u64 v;
            __builtin_memcpy(&v, from + 8 * j, 8);
            __builtin_memcpy(dst + 8 * j, &v, 8);
        }
        for (u32 i = (n8 << 3) + sub; i < n; i += G)
            dst[i] = from[i];
    }
    c.got += n;
    return true;
}

/* make the bytes a group has just written visible to its own later loads */
DEV void group_sync_mem() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

template <int G, bool CMP>
DEV void reload_context(Ctx &c)
{
    group_sync_mem();
    c.p1 = c.got ? c.out[c.got - 1] : 0;
    c.p2 = c.got > 1 ? c.out[c.got - 2] : 0;
}

/* ------------------------------------------------------------- meta-block */

/* metablock(): yeast.c:971-1308.  *last = ISLAST. */
template <int G, bool CMP>
DEV bool metablock(Ctx &c, u32 &last, u32 sub)
{
    u32 v, n, mlen;
    CK(take(c, 1, last));
    if (last) {
        CK(take(c, 1, v)); /* ISLASTEMPTY :993 */
        if (v) {
            if (tail_nonzero(c))
                FAIL(c, YS_INVALID);
            return true;
        }
    }
    CK(take(c, 2, n)); /* MNIBBLES - 4 */
    if (n == 3) {      /* metadata :1008-1032 */
        CK(take(c, 1, v));
        if (v)
            FAIL(c, YS_INVALID);
        CK(take(c, 2, n));
        u32 skip = 0;
        if (n) {
            CK(take(c, n << 3, v));
            skip = v + 1;
        }
        if (n > 1 && (skip >> ((n - 1) << 3)) == 0)
            FAIL(c, YS_INVALID);
        if (tail_nonzero(c))
            FAIL(c, YS_INVALID);
        u64 pos = (c.b.pos + 7) & ~(u64)7;
        if ((u64)skip * 8 > c.b.nbits - pos) {
            c.b.pos = pos;
            FAIL(c, YS_EOF); /* :1025, nothing more consumed */
        }
        bits_seek(c.b, c.src, pos + (u64)skip * 8);
        return true;
    }
    CK(take(c, 16, v));
    mlen = 1 + v;
    if (n) {
        CK(take(c, n << 2, v));
        mlen += v << 16;
        if ((mlen >> ((n + 3) << 2)) == 0)
            FAIL(c, YS_INVALID); /* :1037 */
    }
    if (c.got + mlen > c.cap) { /* :1040-1047 */
        c.need = c.got + mlen;
        FAIL(c, CMP ? YS_MISMATCH : YS_CAPACITY);
    }

    if (!last) {
        CK(take(c, 1, v)); /* ISUNCOMPRESSED :1050 */
        if (v) {
            if (tail_nonzero(c))
                FAIL(c, YS_INVALID);
            u64 pos = (c.b.pos + 7) & ~(u64)7;
            c.b.pos = pos;
            if ((u64)mlen * 8 > c.b.nbits - pos)
                FAIL(c, YS_EOF); /* :1059 */
            CK((emit_raw<G, CMP>(c, c.src + (pos >> 3), mlen, sub)));
            bits_seek(c.b, c.src, pos + (u64)mlen * 8);
            reload_context<G, CMP>(c);
            return true;
        }
    }

    /* header: yeast.c:1078-1159 */
    c.arena_used = 0;
    CK(read_block_header(c, c.L, 0, 1));
    CK(read_block_header(c, c.I, 2, 3));
    CK(read_block_header(c, c.D, 4, 5));
    CK(take(c, 2, c.postfix));
    CK(take(c, 4, v));
    c.direct = v << c.postfix;
    u32 dists = 16 + c.direct + (48u << c.postfix);
    u8 *mode = c.ws + YS_WS_MODE;
    for (n = 0; n < c.L.num; n++) {
        CK(take(c, 2, v));
        mode[n] = (u8)v;
    }
    CK(read_count256(c, c.lit_codes));
    /* the literal map sits at the bottom of the arena, above the block-switch tables */
    u32 lmap_off = c.arena_used;
    u8 *lit_map = c.ws + YS_WS_ARENA + lmap_off;
    if (c.lit_codes > 1) {
        if (c.arena_used + (c.L.num << 6) + 16 > c.arena_size)
            FAIL(c, YS_WORKSPACE);
        c.arena_used += ((c.L.num << 6) + 15) & ~15u;
        CK(read_context_map(c, lit_map, c.L.num << 6, c.lit_codes));
    }
    CK(read_count256(c, c.dist_codes));
    u8 *dist_map = c.ws + YS_WS_DMAP;
    if (c.dist_codes > 1)
        CK(read_context_map(c, dist_map, c.D.num << 2, c.dist_codes));

    u32 *dir = (u32 *)(c.ws + YS_WS_DIR);
    const u16 *base = arena_tables(c);
    const u16 *tL0 = nullptr, *tD0 = nullptr, *t;
    for (n = 0; n < c.lit_codes; n++) {
        CK(read_code(c, 256, &t));
        dir[6 + n] = (u32)(t - base);
        if (n == 0) tL0 = t;
    }
    for (n = 0; n < c.I.num; n++) {
        CK(read_code(c, 704, &t));
        dir[262 + n] = (u32)(t - base);
    }
    for (n = 0; n < c.dist_codes; n++) {
        CK(read_code(c, dists, &t));
        dir[518 + n] = (u32)(t - base);
        if (n == 0) tD0 = t;
    }
    group_sync_mem();
    const u16 *tI = base + dir[262];
    u32 cur_mode = mode[0];

    /* command loop: yeast.c:1166-1295 */
    do {
        if (c.I.num > 1) {
            if (c.I.left == 0) {
                CK(switch_block(c, c.I));
                tI = base + dir[262 + c.I.type];
            }
            c.I.left--;
        }
        u32 cmd, ibase, iext, cbase, cext, insert, copy;
        CK(dec_sym(c, tI, cmd));
        {
            /* cell -> insert/copy code offsets, map[] of yeast.c:655 and :678, 5 bits a cell */
            const u64 ICELL = 0ull | (0ull << 5) | (0ull << 10) | (0ull << 15) | (8ull << 20) | (8ull << 25) |
                              (0ull << 30) | (16ull << 35) | (8ull << 40) | (16ull << 45) | (16ull << 50);
            const u64 CCELL = 0ull | (8ull << 5) | (0ull << 10) | (8ull << 15) | (0ull << 20) | (8ull << 25) |
                              (16ull << 30) | (0ull << 35) | (16ull << 40) | (8ull << 45) | (16ull << 50);
            u32 cell = cmd >> 6;
            u32 ic = ((u32)(ICELL >> (cell * 5)) & 31) + ((cmd >> 3) & 7);
            u32 cc = ((u32)(CCELL >> (cell * 5)) & 31) + (cmd & 7);
            len_code(ic, false, ibase, iext);
            len_code(cc, true, cbase, cext);
        }
        CK(take(c, iext, v));
        insert = ibase + v;
        CK(take(c, cext, v));
        copy = cbase + v;
        if (insert > mlen)
            FAIL(c, YS_INVALID); /* :1189 */
        mlen -= insert;
        while (insert) {
            if (c.L.num > 1) {
                if (c.L.left == 0) {
                    CK(switch_block(c, c.L));
                    cur_mode = mode[c.L.type];
                }
                c.L.left--;
            }
            const u16 *tl = tL0;
            if (c.lit_codes > 1) {
                u32 id = context_id(c, c.p1, c.p2, cur_mode);
                tl = base + dir[6 + lit_map[(c.L.type << 6) + id]];
            }
            u32 s;
            CK(dec_sym(c, tl, s));
            CK((emit_byte<G, CMP>(c, s)));
            c.p2 = c.p1;
            c.p1 = s;
            insert--;
        }
        if (mlen == 0)
            break;

        u64 max = c.got > c.wsize ? c.wsize : c.got;
        i64 dist;
        if (cmd < 128)
            dist = c.r0;
        else {
            if (c.D.num > 1) {
                if (c.D.left == 0)
                    CK(switch_block(c, c.D));
                c.D.left--;
            }
            const u16 *td = tD0;
            if (c.dist_codes > 1)
                td = base + dir[518 + dist_map[(c.D.type << 2) + (copy > 4 ? 3 : copy - 2)]];
            u32 ds;
            CK(dec_sym(c, td, ds));
            /* distance(): yeast.c:761-789 */
            if (ds < 16) {
                u32 back = ds < 4 ? ds : ds < 10 ? 0 : 1;
                u32 rv = back == 0 ? c.r0 : back == 1 ? c.r1 : back == 2 ? c.r2 : c.r3;
                int dl = 0;
                if (ds >= 4) {
                    u32 k = (ds - 4) % 6; /* -1 +1 -2 +2 -3 +3 */
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
                u64 off = ((u64)(2 + ((nn >> c.postfix) & 1)) << x) - 4;
                CK(take(c, x, v));
                dist = (i64)(((off + v) << c.postfix) + (nn & ((1u << c.postfix) - 1)) + c.direct + 1);
            }
            if (ds && dist > 0 && (u64)dist <= max) {
                c.r3 = c.r2; c.r2 = c.r1; c.r1 = c.r0; c.r0 = (u32)dist;
            }
        }
        if (dist <= 0)
            FAIL(c, YS_INVALID); /* see DESIGN.md: zero rejected, negative is the reference's code 3 */
        if ((u64)dist > max) {
            CK((emit_dict<G, CMP>(c, copy, (u64)dist - max - 1, mlen, sub)));
        } else {
            if (copy > mlen)
                FAIL(c, YS_INVALID); /* :1283 */
            mlen -= copy;
            CK((emit_copy<G, CMP>(c, copy, (u64)dist, sub)));
        }
        reload_context<G, CMP>(c);
    } while (mlen);
    if (last && tail_nonzero(c))
        FAIL(c, YS_INVALID); /* :1305 */
    return true;
}

/* yeast(): yeast.c:1353-1400 for one stream of the batch */
template <int G, bool CMP>
DEV void decode_stream(const ys_params &p, u32 job, u8 *ws, const u8 *lut, u32 sub)
{
    Ctx c;
    u64 s0 = p.src_off[job], s1 = p.src_off[job + 1];
    u64 d0 = p.dst_off[job], d1 = p.dst_off[job + 1];
    c.src = p.src + s0;
    c.out = p.dst + d0;
    c.cap = d1 - d0;
    c.got = 0;
    c.rc = YS_OK;
    c.b.nbits = (s1 - s0) * 8;
    c.r0 = 4; c.r1 = 11; c.r2 = 15; c.r3 = 16; /* :1327-1331, most recent first */
    c.ws = ws;
    c.arena_used = 0;
    c.arena_size = (u32)(p.ws_stride - YS_WS_ARENA);
    c.p1 = c.p2 = 0;
    c.nmeta = 0;
    c.dry = 0;
    c.need = 0;
    c.dict = p.dict;
    c.xf = p.xforms;
    c.lut = lut;
    c.L.num = c.I.num = c.D.num = 1;
    bits_seek(c.b, c.src, 0);

    u32 wbits = 16;
    bool ok = [&]() -> bool {
        u32 v;
        CK(take(c, 1, v)); /* WBITS :1372-1378 */
        if (v) {
            CK(take(c, 3, v));
            if (v)
                wbits = v + 17;
            else {
                CK(take(c, 3, v));
                wbits = v ? v + 8 : 17;
            }
        }
        if (wbits == 9)
            FAIL(c, YS_INVALID);
        c.wsize = (1u << wbits) - 16;
        u32 last = 0;
        do {
            CK((metablock<G, CMP>(c, last, sub)));
            c.nmeta++;
        } while (!last);
        return true;
    }();
    (void)ok;

    ys_record r;
    r.rc = (int32_t)c.rc;
    r.flags = wbits | ((c.nmeta > 0xffffff ? 0xffffffu : c.nmeta) << 8);
    r.got = c.got;
    /* bytes pulled: every byte when the bit reader ran dry in take()/dec_sym(), else ceil(bits/8) */
    r.used = c.dry ? (s1 - s0) : (c.b.pos + 7) >> 3;
    r.check = c.need;
    p.rec[job] = r;
    if (c.rc == YS_WORKSPACE && !p.retry && sub == 0)
        p.retry_list[atomicAdd(p.retry_count, 1u)] = job;
}

template <int G, bool CMP>
__global__ __launch_bounds__(256) void k_decode(ys_params p)
{
    __shared__ u8 s_lut[768];
    for (u32 i = threadIdx.x; i < 768; i += blockDim.x)
        s_lut[i] = c_ctx_lut[i];
    __syncthreads();
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
        decode_stream<G, CMP>(p, job, ws, s_lut, sub);
    }
}

/* ------------------------------------------------------------ check values
 * K4: CRC-32C (crc32c.c:258), XXH32 (xxhash.c:365), XXH64 (xxhash.c:505) of
 * each stream's output, one wavefront per stream.
 */
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

extern "C" int ys_upload_constants(const u8 *lut768, const u32 *x2n)
{
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_ctx_lut), lut768, 768);
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyToSymbol(HIP_SYMBOL(c_x2n), x2n, 68 * sizeof(u32));
    return (int)e;
}

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
    case 32: return launch_decode_t<32, false>(*p, groups, st);
    case 16: return launch_decode_t<16, false>(*p, groups, st);
    case 8: return launch_decode_t<8, false>(*p, groups, st);
    case 4: return launch_decode_t<4, false>(*p, groups, st);
    case 2: return launch_decode_t<2, false>(*p, groups, st);
    case 1: return launch_decode_t<1, false>(*p, groups, st);
    }
    return -1;
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
