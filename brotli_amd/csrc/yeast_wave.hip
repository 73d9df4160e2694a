/*
 * yeast_wave.hip -- MI355X (gfx950) batch Brotli decode, one wavefront per stream.
 *
 * The same path as yeast_decode.hip (reference lines are cited where each step is restated),
 * laid out for the way CDNA4 executes it:
 *
 *   - everything that is a function of the bit stream alone is wave-uniform and is written so
 *     that it lands on the scalar unit: the 64-bit bit window, the symbol just decoded, lengths,
 *     distances, the distance ring, block counts;
 *   - the prefix-code tables of the stream live in the wavefront's LDS slot (built there by the
 *     64 lanes, yeast_dev.h), so a symbol costs one LDS round trip, not a trip to HBM;
 *   - small constant tables live in vector registers, one entry per lane, and are read with
 *     v_readlane: the three literal-context tables (yeast.c:700-740), the insert and copy length
 *     codes (yeast.c:652-693), the literal table of each of the 64 contexts, and the next 512
 *     bytes of input (two registers, refilled by coalesced 256-byte loads);
 *   - the lanes move bytes: a window copy or a transformed dictionary word is one byte per lane,
 *     literals are collected in a register and stored a run at a time;
 *   - a copy's load stays in flight while the next command is decoded; its store is issued when
 *     the bytes are needed (literal context) or before the next copy is started.
 *
 * Streams whose tables do not fit the LDS slot are put on the retry list and redone by
 * k_decode<64> with its tables in global memory.
 */
#include <hip/hip_runtime.h>
#include "yeast_dev.h"
#include <string.h>

#ifndef YSW_MIN_WAVES
#define YSW_MIN_WAVES 4
#endif

struct WShared {
    u32 dict[32]; /* per word length: DOFFSET | NDBITS << 24 (yeast.c:911-919) */
    u32 xf[128];  /* per transform: op | omit << 8 | prefix len << 16 | suffix len << 24 */
    u64 xpre[128], xsuf[128];
};

/* per-lane constants: [0..63] lut0, [64..127] lut1, [128..191] lut2 (four table bytes per lane),
   [192..255] length codes: lanes 0..23 insert base | extra << 16, lanes 32..55 copy base | extra << 16 */
__device__ u32 g_wave_const[256];
__device__ u32 g_wave_dict[32];

struct ysw_cfg {
    u32 slot_entries; /* u16 entries per wavefront slot */
    u32 rootcaps;
};

#define RFL(x) uni(x)
#define UNLIKELY(x) __builtin_expect(!!(x), 0)
DEV u32 rdlane(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
DEV u64 rfl64(u64 x) { return uni64(x); }

struct WBlk { /* one block category: yeast.c:112-123 */
    u32 num, type, prev, left;
    u32 d_type, d_count; /* descriptors of the two block-switch tables */
};

struct WConst {
    u32 l0, l1, l2, cmd;
};

struct WCtx {
    Rd r;
    const u8 *src;
    u32 src_len;
    u8 *out, *outp;
    u64 cap;
    u32 wsize;
    u32 r0, r1, r2, r3;
    u32 postfix, direct;
    WBlk L, I, D;
    u32 lit_codes, dist_codes;
    u8 *ws;
    LAS u16 *slot;
    u32 slot_entries, rootcaps;
    u32 p1, p2;
    u32 nmeta;
    u64 need;
    const u8 *dict;
    LAS const WShared *sh;
    u32 lane;
#ifdef YS_PROFILE
    u64 t[8], t_mark; /* 0 header, 1 command symbol, 2 literals, 3 distance, 4 complete, 5 issue, 6 stream total, 7 setup */
    u32 n[4];         /* 0 commands, 1 literals, 2 copies, 3 dictionary words */
#endif
};

#ifdef YS_PROFILE
#define WMARK(i) do { u64 now_ = __builtin_readcyclecounter(); c.t[i] += now_ - c.t_mark; c.t_mark = now_; } while (0)
#define WCOUNT(i) c.n[i]++
#else
#define WMARK(i) do { } while (0)
#define WCOUNT(i) do { } while (0)
#endif

DEV u64 w_got(const WCtx &c) { return (u64)(c.outp - c.out); }

DEV void blk_uniform(WBlk &k)
{
    k.num = uni(k.num); k.type = uni(k.type); k.prev = uni(k.prev); k.left = uni(k.left);
    k.d_type = uni(k.d_type); k.d_count = uni(k.d_count);
}

/* every field but the lane number is the same in all lanes: keep it on the scalar unit */
DEV void wctx_uniform(WCtx &c)
{
    rd_uniform(c.r);
    c.src = (const u8 *)uni64((u64)c.src);
    c.src_len = uni(c.src_len);
    c.out = (u8 *)uni64((u64)c.out);
    c.outp = (u8 *)uni64((u64)c.outp);
    c.cap = uni64(c.cap);
    c.wsize = uni(c.wsize);
    c.r0 = uni(c.r0); c.r1 = uni(c.r1); c.r2 = uni(c.r2); c.r3 = uni(c.r3);
    c.postfix = uni(c.postfix); c.direct = uni(c.direct);
    blk_uniform(c.L); blk_uniform(c.I); blk_uniform(c.D);
    c.lit_codes = uni(c.lit_codes); c.dist_codes = uni(c.dist_codes);
    c.ws = (u8 *)uni64((u64)c.ws);
    c.slot = (LAS u16 *)(size_t)uni((u32)(size_t)c.slot);
    c.slot_entries = uni(c.slot_entries); c.rootcaps = uni(c.rootcaps);
    c.p1 = uni(c.p1); c.p2 = uni(c.p2);
    c.nmeta = uni(c.nmeta);
    c.need = uni64(c.need);
    c.dict = (const u8 *)uni64((u64)c.dict);
    c.sh = (LAS const WShared *)(size_t)uni((u32)(size_t)c.sh);
}

/* the out-of-line header routines, entered and left with wave-uniform state */
DEV void w_code(Hdr &h, u32 num)
{
    h = read_code_t<true>(h, num);
    hdr_uniform(h);
}
DEV void w_context_map(Hdr &h, u8 *map, u32 len, u32 trees)
{
    h = read_context_map_t<true>(h, map, len, trees);
    hdr_uniform(h);
}

/* block_length(): yeast.c:550-567; base[]/extra[] from their closed form */
DEV void blen_of(u32 s, u32 &base, u32 &extra)
{
    if (s < 4) { base = 1 + 4 * s; extra = 2; }
    else if (s < 8) { base = 17 + 8 * (s - 4); extra = 3; }
    else if (s < 12) { base = 49 + 16 * (s - 8); extra = 4; }
    else if (s < 16) { base = 113 + 32 * (s - 12); extra = 5; }
    else if (s < 18) { base = 241 + 64 * (s - 16); extra = 6; }
    else if (s >= 25) { base = 16625; extra = 24; }
    else { extra = s - 11; base = 369 + ((1u << (s - 11)) - 128); }
}

/* one category's NBLTYPES header: yeast.c:1079-1118 */
DEV void w_block_header(Hdr &h, WBlk &k)
{
    k.prev = 1;
    k.type = 0;
    k.left = 0xffffffffu;
    k.d_type = k.d_count = 0;
    k.num = read_count256(h.r);
    if (k.num > 1) {
        w_code(h, k.num + 2);
        k.d_type = h.tab;
        w_code(h, 26);
        k.d_count = h.tab;
        u32 base, extra;
        blen_of(dec_sym_lds(h.r, h.lds, k.d_count), base, extra);
        k.left = base + take(h.r, extra);
    }
}

/* room for n u32 words in the slot; returns the offset in u16 entries */
DEV u32 slot_words(Hdr &h, u32 n)
{
    u32 at = h.lds_used;
    if (h.lds_used + 2 * n > h.lds_size) {
        rd_fail(h.r, YS_WORKSPACE);
        return 0;
    }
    h.lds_used += 2 * n;
    return at;
}

/* ----------------------------------------------------------- command loop */

struct WEnv {
    u32 o_ltab, o_itab, o_dtab; /* u32 arrays in the slot (u16 entry offsets) */
    u32 descL0, descD0;
    const u8 *mode;
    bool ctx;
};

/* The command loop of yeast.c:1166-1295 for one stream, executed by the whole wavefront. */
DEV void w_commands(WCtx &c, u32 &mlen, const WEnv &env_in, const WConst &K)
{
    wctx_uniform(c);
    mlen = uni(mlen);
    WEnv env;
    env.o_ltab = uni(env_in.o_ltab); env.o_itab = uni(env_in.o_itab); env.o_dtab = uni(env_in.o_dtab);
    env.descL0 = uni(env_in.descL0); env.descD0 = uni(env_in.descD0);
    env.mode = (const u8 *)uni64((u64)env_in.mode);
    env.ctx = uni(env_in.ctx ? 1u : 0u) != 0;
    Rd &r = c.r;
    const u32 lane = c.lane;
    LAS const u16 *const slot = c.slot;
    LAS const u32 *const ltab = (LAS const u32 *)(slot + env.o_ltab);
    LAS const u32 *const itab = (LAS const u32 *)(slot + env.o_itab);
    LAS const u32 *const dtab = (LAS const u32 *)(slot + env.o_dtab);
    const bool ctx = env.ctx;

    /* ---- input: the next 512 bytes of the stream in two registers, a dword per lane ---- */
    const u32 mis0 = (u32)((uintptr_t)c.src & 3);
    const u32 *const inp = (const u32 *)(c.src - mis0);
    const u32 ndw = (mis0 + c.src_len + 3) >> 2;
    auto load_chunk = [&](u32 base) -> u32 {
        u32 i = base + lane;
        return i < ndw ? gld32(inp + i) : 0;
    };
    u32 ipos, cbase, cur, nxt;
    {
        u32 bitpos = rd_pos(r);
        u32 bp = mis0 * 8 + bitpos;
        cbase = bp >> 5;
        u32 sh = bp & 31;
        cur = load_chunk(cbase);
        nxt = load_chunk(cbase + 64);
        asm volatile("" : "+v"(cur), "+v"(nxt)); /* landed: see refill */
        u64 lo = rdlane(cur, 0), hi = rdlane(cur, 1);
        r.w = (lo | (hi << 32)) >> sh;
        u32 left = r.total - bitpos;
        r.have = 64 - sh < left ? 64 - sh : left;
        r.rem = left - r.have;
        ipos = cbase + 2;
    }
    auto refill = [&]() {
        if (r.have <= 32) {
            u32 k = ipos - cbase;
            if (UNLIKELY(k >= 64)) {
                cur = nxt;
                cbase += 64;
                nxt = load_chunk(cbase + 64);
                /* waited for here, once per 256 bytes of input: a register with a load on its way makes the
                   compiler wait for ALL loads wherever the loop copies it, and the copy whose bytes are meant to
                   travel while the next command is decoded would be waited for at the head of every command */
                asm volatile("" : "+v"(nxt));
                k -= 64;
            }
            r.w |= (u64)rdlane(cur, k) << r.have;
            u32 add = r.rem < 32 ? r.rem : 32;
            r.have += add;
            r.rem -= add;
            ipos += add ? 1 : 0;
        }
    };
    auto wtake = [&](u32 n) -> u32 { /* bits(): yeast.c:163-179 */
        refill();
        if (UNLIKELY(n > r.have)) {
            rd_eof(r);
            n = 0;
        }
        u32 v = (u32)r.w & ((1u << n) - 1);
        r.w >>= n;
        r.have -= n;
        return v;
    };
    auto wdec = [&](u32 desc) -> u32 { /* decode(): yeast.c:204-219 */
        refill();
        u32 peek = (u32)r.w;
        u32 root = desc >> 16;
        LAS const u16 *t = slot + (desc & 0xffff);
        u32 e = RFL(t[peek & ((1u << root) - 1)]);
        u32 l = e & 15;
        if (l > root) {
            e = RFL(t[(e >> 4) + ((peek >> root) & ((1u << (l - root)) - 1))]);
            l = e & 15;
        }
        if (UNLIKELY(l > r.have)) {
            rd_eof(r);
            l = 0;
        }
        r.w >>= l;
        r.have -= l;
        return e >> 4;
    };
    auto wswitch = [&](WBlk &k) { /* block switch: yeast.c:1168-1180 */
        u32 n = wdec(k.d_type);
        n = n > 1 ? n - 2 : n ? (k.type + 1 == k.num ? 0 : k.type + 1) : k.prev;
        if (n >= k.num)
            n = 0; /* only reachable after a read failure; keeps table indices in range */
        k.prev = k.type;
        k.type = n;
        u32 base, extra;
        blen_of(wdec(k.d_count), base, extra);
        k.left = base + wtake(extra);
    };
    auto lutb = [&](u32 v, u32 b) -> u32 { return (rdlane(v, b >> 2) >> ((b & 3) * 8)) & 255; };

    /* ---- the copy or dictionary word whose bytes are still on their way ---- */
    u32 pk = 0;      /* 0 none, 1 copy, 2 dictionary word */
    u8 *pdst = c.outp;
    u32 pn = 0;      /* bytes it produces (<= 64) */
    u32 pv = 0;      /* per lane: the byte (copy), or the raw word byte for this position (dictionary) */
    u32 d_plen = 0, d_wlen = 0, d_xform = 0, d_op = 0;
    auto complete = [&]() {
        if (pk == 1) {
            if (lane < pn)
                gst8(pdst + lane, pv);
            u32 a = rdlane(pv, pn - 1);
            u32 b = pn >= 2 ? rdlane(pv, pn >= 2 ? pn - 2 : 0) : c.p1;
            c.p1 = a;
            c.p2 = b;
        } else if (pk == 2) { /* dict_word(): yeast.c:908-949, transforms :799-879 */
            u64 pre = rfl64(c.sh->xpre[d_xform]), suf = rfl64(c.sh->xsuf[d_xform]);
            u32 x32 = 0, x5 = 0; /* bit j: word byte j is xor-ed with 32 / with 5 (UPPER(): :809-827) */
            if (d_op == 1 || d_op == 2) {
                u32 j = 0;
                do {
                    u32 ch = rdlane(pv, (d_plen + j) & 63);
                    if (ch < 192) {
                        if (ch >= 97 && ch <= 122)
                            x32 |= 1u << j;
                        j += 1;
                    } else if (ch < 224) {
                        if (j + 1 < d_wlen)
                            x32 |= 1u << (j + 1);
                        j += 2;
                    } else {
                        if (j + 2 < d_wlen)
                            x5 |= 1u << (j + 2);
                        j += 3;
                    }
                } while (d_op == 2 && j < d_wlen);
            }
            u32 v = pv, k = lane;
            if (k < d_plen)
                v = (u32)(pre >> ((8 * k) & 63)) & 255;
            else if (k < d_plen + d_wlen) {
                u32 j = k - d_plen;
                v ^= ((x32 >> j) & 1) << 5;
                v ^= ((x5 >> j) & 1) * 5;
            } else
                v = (u32)(suf >> ((8 * (k - d_plen - d_wlen)) & 63)) & 255;
            if (k < pn)
                gst8(pdst + k, v);
            if (pn >= 2) {
                u32 a = rdlane(v, pn - 1), b = rdlane(v, pn - 2);
                c.p1 = a;
                c.p2 = b;
            } else if (pn == 1) {
                c.p2 = c.p1;
                c.p1 = rdlane(v, 0);
            }
        }
        pk = 0;
    };

    /* context_id() (yeast.c:698-755) of the current literal block type as two byte tables, four entries per lane:
       id = A[p1] | B[p2] whatever the mode (LSB6: A = b & 63; MSB6: A = b >> 2; UTF8: lut0, lut1; signed: lut2 << 3, lut2) */
    u32 cxa = 0, cxb = 0;
    auto set_mode = [&]() {
        const u32 mode = RFL(gld8(env.mode + c.L.type));
        const u32 b = lane * 4;
        const u32 m0 = (b & 63) | (((b + 1) & 63) << 8) | (((b + 2) & 63) << 16) | (((b + 3) & 63) << 24);
        cxa = mode == 0 ? m0 : mode == 1 ? lane * 0x01010101u : mode == 2 ? K.l0 : K.l2 << 3;
        cxb = mode == 2 ? K.l1 : mode == 3 ? K.l2 : 0u;
    };
    set_mode();
    u32 descI = c.I.num > 1 ? RFL(itab[c.I.type]) : RFL(itab[0]);
    u32 ltv = ctx ? ltab[(c.L.type << 6) + lane] : 0; /* literal table of each context of this block type */
    u32 dtv = c.dist_codes > 1 ? dtab[(c.D.type << 2) + (lane & 3)] : 0; /* distance table of each distance context */
    u32 litv = 0;
    u64 got0 = w_got(c);
    u32 maxd = got0 > c.wsize ? c.wsize : (u32)got0;
    const u64 INS_CELL = 0ull | (0ull << 5) | (0ull << 10) | (0ull << 15) | (8ull << 20) | (8ull << 25) | (0ull << 30) |
                         (16ull << 35) | (8ull << 40) | (16ull << 45) | (16ull << 50);
    const u64 CPY_CELL = 0ull | (8ull << 5) | (0ull << 10) | (8ull << 15) | (0ull << 20) | (8ull << 25) | (16ull << 30) |
                         (0ull << 35) | (16ull << 40) | (8ull << 45) | (16ull << 50);

    WMARK(7);
    do {
        if (c.I.num > 1) {
            if (UNLIKELY(c.I.left == 0)) {
                wswitch(c.I);
                descI = RFL(itab[c.I.type]);
            }
            c.I.left--;
        }
        WCOUNT(0);
        u32 cmd = wdec(descI);
        cmd = cmd < 704 ? cmd : 0;
        u32 cell = cmd >> 6;
        u32 icode = ((u32)(INS_CELL >> (5 * cell)) & 31) + ((cmd >> 3) & 7);
        u32 ccode = ((u32)(CPY_CELL >> (5 * cell)) & 31) + (cmd & 7);
        u32 ie = rdlane(K.cmd, icode), ce = rdlane(K.cmd, 32 + ccode);
        u32 insert = uni((ie & 0xffff) + wtake(ie >> 16));
        u32 copy = uni((ce & 0xffff) + wtake(ce >> 16));
        if (UNLIKELY(r.rc != 0))
            break;
        if (UNLIKELY(insert > mlen)) {
            rd_fail(r, YS_INVALID); /* :1189 */
            break;
        }
        mlen -= insert;
        WMARK(1);
        if (insert) {
            if (pk)
                complete(); /* the literal context needs the bytes before it */
            WMARK(4);
            u32 nl = 0;
            u8 *lbase = c.outp;
            do {
                if (c.L.num > 1) {
                    if (UNLIKELY(c.L.left == 0)) {
                        wswitch(c.L);
                        set_mode();
                        if (ctx)
                            ltv = ltab[(c.L.type << 6) + lane];
                    }
                    c.L.left--;
                }
                u32 desc = env.descL0;
                if (ctx) /* context_id(): yeast.c:698-755 */
                    desc = rdlane(ltv, (lutb(cxa, c.p1) | lutb(cxb, c.p2)) & 63);
                u32 s = wdec(desc);
                if (UNLIKELY(r.rc != 0))
                    break;
                WCOUNT(1);
                litv = lane == (nl & 63) ? s : litv;
                nl++;
                if (UNLIKELY((nl & 63) == 0)) {
                    gst8(lbase + lane, litv);
                    lbase += 64;
                }
                c.p2 = c.p1;
                c.p1 = s;
            } while (--insert);
            if (lane < (nl & 63))
                gst8(lbase + lane, litv);
            c.outp += nl;
            u32 room = c.wsize - maxd;
            maxd += nl < room ? nl : room;
            WMARK(2);
            if (UNLIKELY(r.rc != 0))
                break;
        }
        if (mlen == 0)
            break; /* :1227 */

        i64 dist = c.r0;
        if (cmd >= 128) {
            if (c.D.num > 1) {
                if (UNLIKELY(c.D.left == 0)) {
                    wswitch(c.D);
                    if (c.dist_codes > 1)
                        dtv = dtab[(c.D.type << 2) + (lane & 3)];
                }
                c.D.left--;
            }
            u32 descD = env.descD0;
            if (c.dist_codes > 1)
                descD = rdlane(dtv, copy > 4 ? 3 : copy - 2);
            u32 ds = wdec(descD);
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
                dist = (i64)(((off + wtake(x)) << c.postfix) + (nn & ((1u << c.postfix) - 1)) + c.direct + 1);
            }
            if (UNLIKELY(r.rc != 0))
                break;
            if (ds && dist > 0 && dist <= (i64)maxd) {
                c.r3 = c.r2; c.r2 = c.r1; c.r1 = c.r0; c.r0 = (u32)dist;
            }
        }
        if (UNLIKELY(dist <= 0)) {
            rd_fail(r, YS_INVALID); /* zero: see DESIGN.md; negative: the reference's code 3 */
            break;
        }
        WMARK(3);
        if (pk)
            complete();
        WMARK(4);
        if (dist > (i64)maxd) { /* static dictionary: dict_word() yeast.c:908-949 */
            u64 id = (u64)dist - maxd - 1;
            if (copy > 24 || copy < 4) {
                rd_fail(r, YS_INVALID); /* :921-924 */
                break;
            }
            u32 de = RFL(c.sh->dict[copy]);
            u32 nd = de >> 24;
            u64 xform = id >> nd;
            if (xform >= 121) {
                rd_fail(r, YS_INVALID); /* :929-930 */
                break;
            }
            u32 index = (u32)id & ((1u << nd) - 1);
            const u8 *w = c.dict + (de & 0xffffff) + index * copy;
            u32 hdr = RFL(c.sh->xf[(u32)xform]);
            u32 op = hdr & 255, omit = (hdr >> 8) & 255, plen = (hdr >> 16) & 255, slen = hdr >> 24;
            u32 wlen = copy, wskip = 0;
            if (op == 3) { /* omit first */
                wlen = copy > omit ? copy - omit : 0;
                wskip = wlen ? omit : 0;
            } else if (op == 4)
                wlen = copy > omit ? copy - omit : 0;
            u32 total = plen + wlen + slen;
            if (total > mlen) {
                rd_fail(r, YS_INVALID); /* :1269 */
                break;
            }
            pv = (lane >= plen && lane < plen + wlen) ? gld8(w + wskip + lane - plen) : 0;
            WCOUNT(3);
            pk = 2;
            pdst = c.outp;
            pn = total;
            d_plen = plen;
            d_wlen = wlen;
            d_xform = (u32)xform;
            d_op = op;
            c.outp += total;
            mlen -= total;
            u32 room = c.wsize - maxd;
            maxd += total < room ? total : room;
        } else {
            if (UNLIKELY(copy > mlen)) {
                rd_fail(r, YS_INVALID); /* :1283 */
                break;
            }
            mlen -= copy;
            WCOUNT(2);
            const u32 d = (u32)dist, n = copy;
            u8 *dst = c.outp;
            const u8 *from = dst - d;
            if (n <= 64) {
                /* byte-serial semantics (yeast.c:1285-1293): byte i repeats byte i mod d */
                u32 m = d < n ? lane % d : lane;
                pv = lane < n ? gld8(from + m) : 0;
                pk = 1;
                pdst = dst;
                pn = n;
            } else {
                if (d >= n || d >= 1024) {
                    wave_copy_far(dst, d, n, lane);
                } else if (d >= 64) {
                    /* each 64-byte piece lies wholly behind its source; pieces go in order */
                    for (u32 j = 0; j < n; j += 64) {
                        u32 i = j + lane;
                        if (i < n)
                            gst8(dst + i, gld8(from + i));
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    }
                } else {
                    u32 m = lane % d, step = 64 % d;
                    for (u32 i = lane; i < n; i += 64) {
                        gst8(dst + i, gld8(from + m));
                        m += step;
                        if (m >= d)
                            m -= d;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                c.p1 = RFL(gld8(dst + n - 1));
                c.p2 = RFL(gld8(dst + n - 2));
            }
            c.outp += n;
            u32 room = c.wsize - maxd;
            maxd += n < room ? n : room;
        }
        WMARK(5);
    } while (mlen && !r.rc);
    if (pk)
        complete();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    /* hand the position back to the header reader */
    if (!r.rc) {
        u32 pos = rd_pos(r);
        rd_seek(r, c.src, pos);
    }
}

/* stored block (yeast.c:1061-1070) */
DEV void w_raw(WCtx &c, const u8 *from, u32 n)
{
    u8 *dst = c.outp;
    u32 lane = c.lane;
    if (n >= 1024)
        wave_copy_disjoint(dst, from, n, lane);
    else {
        u32 n8 = n >> 3;
        for (u32 j = lane; j < n8; j += 64)
            gst64u(dst + 8 * j, gld64u(from + 8 * j));
        for (u32 i = (n8 << 3) + lane; i < n; i += 64)
            gst8(dst + i, gld8(from + i));
    }
    c.outp += n;
    c.p2 = n >= 2 ? RFL(gld8(from + n - 2)) : c.p1;
    c.p1 = RFL(gld8(from + n - 1));
}

/* The command loop out of line: as a function of its own it has the scalar registers to itself (inlined into the
   kernel, the stream's whole state competed for them and the loop ran on spilled registers: 648 v_readlane, 225
   v_writelane and 234 scratch accesses in the kernel's code).  State goes in and comes back by value. */
struct WCmdIO {
    WCtx c;
    u32 mlen;
};
NOINL WCmdIO w_commands_fn(WCmdIO io, WEnv env, WConst K)
{
    w_commands(io.c, io.mlen, env, K);
    return io;
}

/* metablock(): yeast.c:971-1308.  Returns ISLAST. */
DEV u32 w_metablock(WCtx &c, const WConst &K)
{
    wctx_uniform(c);
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
    if (w_got(c) + mlen > c.cap) { /* :1040-1047 */
        c.need = w_got(c) + mlen;
        rd_fail(r, YS_CAPACITY);
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
        w_raw(c, c.src + (pos >> 3), mlen);
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
    h.arena_size = 0;
    h.tab = 0;
    h.sub = c.lane;
    h.glog = 6;
    h.lds = c.slot;
    h.lds_used = 0;
    h.lds_size = c.slot_entries;
    h.rootcaps = c.rootcaps;
    h.kind = YS_KIND_OTHER;
    w_block_header(h, c.L);
    w_block_header(h, c.I);
    w_block_header(h, c.D);
    c.postfix = take(h.r, 2);
    c.direct = take(h.r, 4) << c.postfix;
    u32 dists = 16 + c.direct + (48u << c.postfix);
    u8 *mode = c.ws + YS_WS_MODE;
    for (n = 0; n < c.L.num; n++)
        gst8(mode + n, take(h.r, 2));
    c.lit_codes = read_count256(h.r);
    u32 nmap = c.lit_codes > 1 ? (c.L.num << 6) : 0;
    WEnv env;
    env.mode = mode;
    env.ctx = c.lit_codes > 1;
    env.o_ltab = slot_words(h, nmap);
    env.o_itab = slot_words(h, c.I.num);
    if (h.r.rc) {
        r = h.r;
        return 1;
    }
    u8 *lit_map = c.ws + YS_WS_ARENA + 2048; /* behind the code-length scratch of read_code() */
    if (c.lit_codes > 1)
        w_context_map(h, lit_map, nmap, c.lit_codes);
    c.dist_codes = read_count256(h.r);
    env.o_dtab = slot_words(h, c.dist_codes > 1 ? (c.D.num << 2) : 0);
    u8 *dist_map = c.ws + YS_WS_DMAP;
    if (c.dist_codes > 1 && !h.r.rc)
        w_context_map(h, dist_map, c.D.num << 2, c.dist_codes);

    u32 *ldir = (u32 *)(c.ws + YS_WS_LDIR), *ddir = (u32 *)(c.ws + YS_WS_DDIR);
    LAS u32 *ltab = (LAS u32 *)(c.slot + env.o_ltab);
    LAS u32 *itab = (LAS u32 *)(c.slot + env.o_itab);
    LAS u32 *dtab = (LAS u32 *)(c.slot + env.o_dtab);
    env.descL0 = env.descD0 = 0;
    h.kind = YS_KIND_LIT;
    for (n = 0; n < c.lit_codes && !h.r.rc; n++) {
        w_code(h, 256);
        gst32(ldir + n, h.tab);
        env.descL0 = n ? env.descL0 : h.tab;
    }
    h.kind = YS_KIND_IAC;
    for (n = 0; n < c.I.num && !h.r.rc; n++) {
        w_code(h, 704);
        if (c.lane == 0)
            itab[n] = h.tab;
    }
    h.kind = YS_KIND_DIST;
    for (n = 0; n < c.dist_codes && !h.r.rc; n++) {
        w_code(h, dists);
        gst32(ddir + n, h.tab);
        env.descD0 = n ? env.descD0 : h.tab;
    }
    r = h.r;
    if (r.rc)
        return 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    /* resolve the context maps to table descriptors once, the entries spread over the lanes */
    for (n = c.lane; n < nmap; n += 64)
        ltab[n] = gld32(ldir + gld8(lit_map + n));
    if (c.dist_codes > 1)
        for (n = c.lane; n < (c.D.num << 2); n += 64)
            dtab[n] = gld32(ddir + gld8(dist_map + n));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    WMARK(0);
    {
        WCmdIO io;
        io.c = c;
        io.mlen = mlen;
        io = w_commands_fn(io, env, K);
        c = io.c;
        mlen = io.mlen;
        wctx_uniform(c);
    }
    WMARK(5);

    if (!r.rc && last && tail_nonzero(r))
        rd_fail(r, YS_INVALID); /* :1305 */
    return last;
}

/* yeast(): yeast.c:1353-1400 for one stream of the batch */
DEV void w_stream(const ys_params &p, u32 job, WCtx &c, const WConst &K)
{
    u64 s0 = p.src_off[job], s1 = p.src_off[job + 1];
    u64 d0 = p.dst_off[job], d1 = p.dst_off[job + 1];
    c.src = p.src + s0;
    c.src_len = (u32)(s1 - s0);
    c.out = p.dst + d0;
    c.outp = c.out;
    c.cap = d1 - d0;
    c.r.rc = YS_OK;
    c.r.dry = 0;
    c.r.fpos = 0;
    c.r.total = (u32)(s1 - s0) * 8;
    c.r0 = 4; c.r1 = 11; c.r2 = 15; c.r3 = 16; /* :1327-1331, most recent first */
    c.p1 = c.p2 = 0;
    c.nmeta = 0;
    c.need = 0;
    c.L.num = c.I.num = c.D.num = 1;
#ifdef YS_PROFILE
    for (int i = 0; i < 8; i++)
        c.t[i] = 0;
    for (int i = 0; i < 4; i++)
        c.n[i] = 0;
    c.t_mark = __builtin_readcyclecounter();
    u64 t_start = c.t_mark;
#endif
    rd_seek(c.r, c.src, 0);
    if (s1 - s0 >= (1u << 29))
        rd_fail(c.r, YS_NOMEM); /* the bit cursor is 32 bits wide */

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
        u32 last = w_metablock(c, K);
        if (c.r.rc)
            break;
        c.nmeta++;
        if (last)
            break;
    }
    if (c.lane == 0) {
        ys_record rec;
        rec.rc = (int32_t)c.r.rc;
        rec.flags = wbits | ((c.nmeta > 0xffffff ? 0xffffffu : c.nmeta) << 8);
        rec.got = w_got(c);
        rec.used = c.r.dry ? (s1 - s0) : (rd_pos(c.r) + 7) >> 3;
        rec.check = c.need;
        p.rec[job] = rec;
#ifdef YS_PROFILE
        if (c.cap >= 128) { /* the profile build's output bytes are not meant to be looked at */
            u64 *prof = (u64 *)c.out;
            c.t[6] = __builtin_readcyclecounter() - t_start;
            for (int i = 0; i < 8; i++)
                prof[i] = c.t[i];
            for (int i = 0; i < 4; i++)
                prof[8 + i] = c.n[i];
        }
#endif
        if (c.r.rc == YS_WORKSPACE && !p.retry)
            p.retry_list[atomicAdd(p.retry_count, 1u)] = job;
    }
}

__global__ __launch_bounds__(256, YSW_MIN_WAVES) void k_decode_wave(ys_params p, ysw_cfg cfg)
{
    extern __shared__ __attribute__((aligned(16))) u8 dyn[];
    LAS u8 *lds = (LAS u8 *)dyn;
    LAS WShared *sh = (LAS WShared *)lds;
    {
        u32 i = threadIdx.x;
        if (i < 32)
            sh->dict[i] = g_wave_dict[i];
        if (i < 128) {
            const u32 *rec = (const u32 *)(p.xforms + 20 * i);
            sh->xf[i] = i < 121 ? rec[0] : 0;
            sh->xpre[i] = i < 121 ? (u64)rec[1] | ((u64)rec[2] << 32) : 0;
            sh->xsuf[i] = i < 121 ? (u64)rec[3] | ((u64)rec[4] << 32) : 0;
        }
        __syncthreads();
    }
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    WConst K;
    K.l0 = g_wave_const[lane];
    K.l1 = g_wave_const[64 + lane];
    K.l2 = g_wave_const[128 + lane];
    K.cmd = g_wave_const[192 + lane];
    WCtx c;
    c.lane = lane;
    c.sh = sh;
    c.slot = (LAS u16 *)(lds + sizeof(WShared) + (size_t)wave * cfg.slot_entries * 2);
    c.slot_entries = cfg.slot_entries;
    c.rootcaps = cfg.rootcaps;
    c.dict = p.dict;
    u32 gwave = blockIdx.x * (blockDim.x >> 6) + wave;
    c.ws = p.ws + (u64)gwave * p.ws_stride;
    const u32 limit = p.retry ? *p.retry_count : p.n;
    for (;;) {
        u32 job = 0;
        if (lane == 0)
            job = atomicAdd(p.counter, 1u);
        job = RFL(job);
        if (job >= limit)
            break;
        if (p.retry)
            job = p.retry_list[job];
        w_stream(p, job, c, K);
    }
}

/* ---------------------------------------------------------------- launchers */

extern "C" int ysw_upload_constants(const u8 *lut768)
{
    static const u8 ins_x[24] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24};
    static const u8 cpy_x[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24};
    u32 k[256];
    memset(k, 0, sizeof(k));
    memcpy(k, lut768, 768);
    u32 b = 0;
    for (int i = 0; i < 24; i++) { /* each base is the previous base plus 1 << previous extra (yeast.c:655-665) */
        k[192 + i] = b | ((u32)ins_x[i] << 16);
        b += 1u << ins_x[i];
    }
    b = 2;
    for (int i = 0; i < 24; i++) { /* yeast.c:678-688 */
        k[192 + 32 + i] = b | ((u32)cpy_x[i] << 16);
        b += 1u << cpy_x[i];
    }
    static const u8 ndbits[25] = {0, 0, 0, 0, 10, 10, 11, 11, 10, 10, 10, 10, 10, 9, 9, 8, 7, 7, 8, 7, 7, 6, 6, 5, 5};
    u32 dict[32] = {0};
    u32 off = 0;
    for (u32 l = 4; l <= 24; l++) {
        dict[l] = off | ((u32)ndbits[l] << 24);
        off += l << ndbits[l];
    }
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_wave_const), k, sizeof(k));
    if (e != hipSuccess)
        return (int)e;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_dict), dict, sizeof(dict));
}

extern "C" unsigned ysw_shared_bytes(void) { return (unsigned)sizeof(WShared); }

/* blocks of 256 threads (four wavefronts, four LDS slots of slot_entries u16 entries each) */
extern "C" int ysw_launch(const ys_params *p, unsigned slot_entries, unsigned rootcaps, unsigned blocks, void *stream)
{
    ysw_cfg cfg;
    cfg.slot_entries = slot_entries;
    cfg.rootcaps = rootcaps;
    size_t lds = sizeof(WShared) + 4 * (size_t)slot_entries * 2;
    hipError_t e = hipFuncSetAttribute((const void *)k_decode_wave, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess)
        return (int)e;
    hipLaunchKernelGGL(k_decode_wave, dim3(blocks), dim3(256), lds, (hipStream_t)stream, *p, cfg);
    return (int)hipGetLastError();
}

YSB_HOST(wave) /* bounds-checked debug build only: yeast_bounds.h */
