/*
 * yeast_trip3.h -- the command loop of the batch kernel in its third form (included by yeast_trip.h; reference:
 * the loop of yeast.c:1166-1295).  Same trip as run_trip_fn's write-behind form -- one prefix-code symbol per
 * stream and trip, three loads that land in LDS behind counted waits, every output byte through a ring in LDS and out
 * in complete chunks (a 128-byte ring and whole 64-byte lines, or 64 bytes and 16-byte chunks: Trip3Ring below) --
 * written for fewer instructions per trip:
 *
 *   - what a symbol kind needs is done under that kind's lanes only (one of: literal, insert-and-copy symbol,
 *     distance symbol, copy set-up), and each kind says what the stream decodes next (nk), instead of every step
 *     being computed for all lanes and selected through a web of conditions;
 *   - failures (end of input, an insert length beyond the meta-block, a bad distance) are collected and dealt
 *     with behind one wavefront-uniform branch: the other lanes' path carries no checks;
 *   - a copy of several pieces is a state of its own (HOLD): the next command's symbol is asked for with the
 *     copy's last piece, and no entry is held in a register meanwhile;
 *   - the last two bytes (literal context) are one unaligned 16-bit read from the ring, whatever wrote them;
 *   - a copy from close behind the output (less than 112 bytes with the 128-byte ring, 48 with the 64-byte one) reads the
 *     ring when its piece is appended, a trip later (nothing writes the ring in between), so no staging slot and no
 *     branch at set-up;
 *   - streams without a literal or distance context map get maps filled with their one table, so the lookup is
 *     the same for all.
 */
#ifndef YEAST_TRIP3_H
#define YEAST_TRIP3_H

/* rare wavefront-uniform branches, laid out of the common path (46.7 -> 46.2 ms; 8,192 streams 41.6 -> 40.1) */
#define YT_RARE(x) __builtin_expect(!!(x), 0)
#define YT_HOLD 5u /* st: a copy of several pieces is going on */
/* nk, what a stream does next: YT_CMD/YT_LIT/YT_DIST a lookup of that category, YT_GO the copy begins (and the next
   command's symbol is looked up), YT_END the meta-block is over (or the stream has failed), YT_NONE nothing new */
#define YT_GO 4u
#define YT_NONE 7u

typedef u16 __attribute__((aligned(1))) u16u;

/* The stream's ring in LDS: the last RING bytes of its output at offset (position & (RING - 1)), 16 bytes of slack in front,
   and behind it a copy of its first 16 bytes (so that 16 bytes can be read at any offset) and 16 bytes that take the second
   write of an access that needs none.  Output leaves it in complete chunks of CHUNK bytes.  Two settings:
     RING 128, CHUNK 64 (loop form 3): for batches that keep every SIMD busy.  With every stream resident a chunk is evicted
       from L2 before the next one of its line arrives, so a 16-byte chunk costs a 32-byte write of its own (WRITE_SIZE 174 KB
       per stream against 87 KB at 8,192 streams, profiles/r03/write_traffic.txt); a 64-byte chunk is a whole line, written
       once: 45.6 -> 44.0 ms at 65,536 streams, 160.5 -> 151.3 at 262,144;
     RING 64, CHUNK 16 (loop form 4): for smaller batches, where the chunks merge in L2 anyway and the shorter trip counts
       (40.1 against 40.9 ms at 8,192 streams). */
template <u32 RING>
struct Trip3Ring {
    /* RING 128: 16 + RING + 32 bytes; RING 64: 32 bytes of slack, the ring, 52 behind it (every write has its twin 64 bytes
       on or back: one select instead of two).  An odd number of words either way: the streams' rings start in different banks */
    static constexpr u32 STRIDE = RING == 128u ? 180u : YT_RING_STRIDE;
    static constexpr u32 FRONT = RING == 128u ? 16u : 32u;
    static constexpr u32 NEAR = RING - 16u; /* a copy from less than this far back finds its source in the ring; one from further
                                               back in memory, where all but the last CHUNK + 15 bytes are */
};

template <int S, u32 RING, u32 CHUNK>
NOINL TripOut run_trip3_fn(TripIn a)
{
    static_assert(Trip3Ring<RING>::STRIDE <= YT_RING_ALLOC && CHUNK + 32u <= RING && (CHUNK & 15u) == 0, "ring layout");
    constexpr u32 YT3_RING = RING, YT3_CHUNK = CHUNK, YT3_STRIDE = Trip3Ring<RING>::STRIDE, YT3_NEAR = Trip3Ring<RING>::NEAR;
    LAS TripLds<S> *wl = (LAS TripLds<S> *)a.wl;
    LAS const TripShared *ts = a.ts;
    const u32 slot = threadIdx.x & 63;
    const u8 *const wsb = (const u8 *)uni64((u64)a.ws_base);
    const u8 *const xdict = (const u8 *)uni64((u64)a.xdict);
    const u32 wso = a.wso;
    const u32 rootcaps = uni(a.rootcaps);
    /* index widths of the root tables by category (CMD 0, LIT 1, DIST 2), a byte each */
    const u32 rw3 = ((rootcaps >> 16) & 255) | (((rootcaps >> 8) & 255) << 8) | ((rootcaps >> 24) << 16);
    const bool ctx = a.flags & 1;
    const bool dmap = a.flags & 2;
    const u32 postfix = a.postfix, direct = a.direct, wsize = a.wsize;
    const u32 out_lo = (u32)(uintptr_t)a.out;
    u32 mlen = a.mlen;
    u32 rc = 0, dry = 0;
    const u32 lds_ent = uni((u32)(uintptr_t)&wl->ent[0]);
    const u32 lds_top = uni((u32)(uintptr_t)&wl->top[0]);
    const u32 lds_piece = uni((u32)(uintptr_t)&wl->piece[0]);

    /* ---- per-type tables of the current block types (a stream with one table has a map of that table) ---- */
    const u32 w_ltab = a.w_ltab;
    auto load_lt = [&](u32 type) __attribute__((always_inline)) {
#pragma clang loop unroll(disable)
        for (u32 n = 0; n < 64; n += 16) {
            const u32 at = w_ltab + ((type << 6) + n) * 4;
            u32x4 v0 = sld128(wsb, at), v1 = sld128(wsb, at + 16), v2 = sld128(wsb, at + 32), v3 = sld128(wsb, at + 48);
            LAS u16 *to = &wl->lt[n * S + slot];
            to[0 * S] = (u16)(v0.x >> 4); to[1 * S] = (u16)(v0.y >> 4); to[2 * S] = (u16)(v0.z >> 4); to[3 * S] = (u16)(v0.w >> 4);
            to[4 * S] = (u16)(v1.x >> 4); to[5 * S] = (u16)(v1.y >> 4); to[6 * S] = (u16)(v1.z >> 4); to[7 * S] = (u16)(v1.w >> 4);
            to[8 * S] = (u16)(v2.x >> 4); to[9 * S] = (u16)(v2.y >> 4); to[10 * S] = (u16)(v2.z >> 4); to[11 * S] = (u16)(v2.w >> 4);
            to[12 * S] = (u16)(v3.x >> 4); to[13 * S] = (u16)(v3.y >> 4); to[14 * S] = (u16)(v3.z >> 4); to[15 * S] = (u16)(v3.w >> 4);
        }
    };
    auto load_dt = [&](u32 type) __attribute__((always_inline)) {
#pragma clang loop unroll(disable)
        for (u32 n = 0; n < 4; n++)
            wl->dt[n * S + slot] = (u16)(sld32(wsb, wso + YS_WS_DTAB + ((type << 2) + n) * 4) >> 4);
    };
    if (ctx)
        load_lt(0);
    else {
        const u16 v = (u16)((a.offL0 - wso) >> 4);
#pragma clang loop unroll(disable)
        for (u32 n = 0; n < 64; n++)
            wl->lt[n * S + slot] = v;
    }
    if (dmap)
        load_dt(0);
    else {
        const u16 v = (u16)((a.offD0 - wso) >> 4);
        for (u32 n = 0; n < 4; n++)
            wl->dt[n * S + slot] = v;
    }
    u32 offI = wso + sld32(wsb, wso + YS_WS_ITAB);
    u32 kmode = a.cur_mode << 8;
    const u32 num3 = a.num3;
    u32 type3 = 0, prev3 = 0x010101u;
    u32 left0 = a.left0, left1 = a.left1, left2 = a.left2;
    const bool switching = num3 != 0;
    if (switching) {
        wl->bt[0 * S + slot] = a.bt0;
        wl->bt[1 * S + slot] = a.bt1;
        wl->bt[2 * S + slot] = a.bt2;
        wl->bt[3 * S + slot] = a.bt3;
        wl->bt[4 * S + slot] = a.bt4;
        wl->bt[5 * S + slot] = a.bt5;
    }
    const bool any_switching = __any(switching);

    /* ---- input: eight words behind the bit offset bo (< 32 at the head of a trip), as in run_trip_fn ---- */
    u32 d0, d1, d2, d3, d4, d5, d6, d7, nq = 8, bo;
    const u8 *ip;
    u32 endrel;
    {
        u32 bitpos = a.bitpos;
        const u8 *at = a.src + (bitpos >> 3);
        u32 mis = (u32)(uintptr_t)at & 3;
        ip = at - mis;
        bo = mis * 8 + (bitpos & 7);
        endrel = a.total - bitpos + bo;
        u32x4 q0 = gld128u(ip), q1 = gld128u(ip + 16);
        d0 = q0.x; d1 = q0.y; d2 = q0.z; d3 = q0.w; d4 = q1.x; d5 = q1.y; d6 = q1.z; d7 = q1.w;
        ip += 32;
    }

    /* ---- output, ring, copy engine ---- */
    u8 *op = a.outp;
    u32 r0 = a.r0, r1 = a.r1, r2 = a.r2, r3 = a.r3;
    u32 insert = 0, copy = 0, dcx = 0, after = YT_DIST;
    u32 dsym = 0, dxv = 0;
    const u8 *csrc = a.outp;
    u32 crem = 0, cD = 0;
    u32 wn = 0; /* bytes in the ring that are not in memory yet */
    u32 pn = 0; /* bytes of the piece asked for on the last trip */
    u32 po = 0; /* that piece is taken from the ring, at this offset + 1 (0: it lands in wl->piece) */
    const u32 base_lo = (u32)(uintptr_t)a.outp;
    constexpr u32 RM = YT3_RING - 1;
    LAS u8 *const rb = (LAS u8 *)wl->ring + (slot & (S - 1)) * YT3_STRIDE + Trip3Ring<RING>::FRONT;
    /* where the second copy of sixteen bytes written at o goes: behind the ring's end for its first bytes, in front of
       its start for a write that runs over the end, into the spare bytes otherwise */
    auto ring_twin = [&](u32 o) __attribute__((always_inline)) -> u32 {
        if constexpr (RING == 64u)
            return o < 32 ? o + 64 : o - 64;
        else
            return o < 16 ? o + YT3_RING : o > YT3_RING - 16 ? o - YT3_RING : YT3_RING + 16;
    };
    auto ring_put8 = [&](u32 o, u32 v) __attribute__((always_inline)) {
        rb[o] = (u8)v;
        if constexpr (RING == 64u)
            rb[o < 32 ? o + 64 : o - 64] = (u8)v;
        else
            rb[o < 16 ? o + YT3_RING : YT3_RING + 16] = (u8)v;
    };
    {
        ring_put8(RM, a.p1);
        ring_put8(RM - 1, a.p2);
        u64 got0 = (u64)(a.outp - a.out);
        u32 k = got0 < YT3_NEAR ? (u32)got0 : YT3_NEAR;
        for (u32 j = 1; j <= k; j++)
            ring_put8((0u - j) & RM, gld8(a.outp - j));
    }

#define WAITV(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory") /* at most n of the trip's three loads in flight */
    u32 st = YT_CMD, cat = 0;
    u32 la = 0, lw = 0, lk = 0;
    u32 lo = 0, hi = 0;
    u32 treq = 0;

    /* a lookup of category w begins for the lanes in `on`; a used-up block count turns it into a block switch */
    auto enter = [&](bool on, u32 w, u32 off_w, u32 l0, u32 l1, u32 l2) __attribute__((always_inline)) {
        u32 nst = w, nla = off_w, nlw = __builtin_amdgcn_ubfe(rw3, w << 3, 8);
        if (any_switching) {
            u32 left = w == YT_CMD ? l0 : w == YT_LIT ? l1 : l2;
            bool sw = ((num3 >> (8 * w)) & 255) != 0 && left == 0;
            if (YT_RARE(__any(on && sw))) {
                u32 tt = wl->bt[(2 * w) * S + slot];
                nla = sw ? wso + tt : nla;
                nlw = sw ? 8u : nlw;
                nst = sw ? YT_BT : nst;
                cat = on && sw ? w : cat;
            }
        }
        st = on ? nst : st;
        la = on ? nla : la;
        lw = on ? nlw : lw;
    };
    auto advance = [&]() __attribute__((always_inline)) {
        bool adv = bo >= 32;
        d0 = adv ? d1 : d0;
        d1 = adv ? d2 : d1;
        d2 = adv ? d3 : d2;
        d3 = adv ? d4 : d3;
        d4 = adv ? d5 : d4;
        d5 = adv ? d6 : d5;
        d6 = adv ? d7 : d6;
        nq -= adv;
        u32 step = adv ? 32u : 0u;
        endrel -= step;
        bo -= step;
    };
    auto peek = [&]() __attribute__((always_inline)) {
        advance();
        if (YT_RARE(__any(bo >= 32)))
            advance();
        lo = __builtin_amdgcn_alignbit(d1, d0, bo);
        hi = __builtin_amdgcn_alignbit(d2, d1, bo);
    };
    auto request = [&](bool on) __attribute__((always_inline)) {
        lds_load4(wsb, on ? la + (__builtin_amdgcn_ubfe(lo, 0, lw) << 1) : 0u, lds_ent);
        treq = nq <= 4 && st != YT_END && endrel > nq * 32;
#ifdef YT3_TOP_NT /* measurement aid: the input words not kept in L2 */
        lds_load16_nt(treq ? ip : wsb, lds_top);
#else
        lds_load16(treq ? ip : wsb, lds_top);
#endif
    };

    enter(true, YT_CMD, offI, left0, left1, left2);
    peek();
    request(true);
    lds_load16_nt(wsb, lds_piece);
    asm volatile("" : "+v"(offI), "+v"(kmode), "+v"(d0), "+v"(d7));

    u32 budget;
    {
        u64 l64 = 16ull * mlen + 2ull * endrel + 256;
        u32 lim = l64 > 0xffffffffull ? 0xffffffffu : (u32)l64;
        budget = 0;
        for (u64 todo = __ballot(1); todo; todo &= todo - 1) {
            u32 v = (u32)__builtin_amdgcn_readlane((int)lim, (int)__builtin_ctzll(todo));
            budget = v > budget ? v : budget;
        }
    }
#ifdef YS_PROFILE
    u32 ntrip = 0;
    u64 pwait = 0, pwait2 = 0, pwait3 = 0, plast;
    u64 pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(plast)::"memory");
    const u64 t_loop = __builtin_amdgcn_s_memrealtime();
#endif
    while (__any(st != YT_END)) {
        if (--budget == 0) {
            rc = st != YT_END && !rc ? (u32)YS_WORKSPACE : rc;
            break;
        }
        /* ================= 1. the entry arrives: the code, then what the symbol means ================= */
#ifdef YS_PROFILE
        ntrip++;
        u64 ta_, tb_;
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(ta_)::"memory");
#endif
        WAITV(2);
#ifdef YS_PROFILE
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tb_)::"memory");
        pwait += tb_ - ta_;
#endif
        u32 en = wl->ent[slot];
        asm volatile("" : "+v"(en));
        TMARK(0);
#ifdef YT3_PAD_A /* measurement build: what an instruction costs at this place (dependent vector adds) */
        {
            u32 padv = st;
#pragma unroll
            for (int i = 0; i < YT3_PAD_A; i++)
                asm volatile("v_add_u32 %0, %0, 1" : "+v"(padv));
            asm volatile("" ::"v"(padv));
        }
#endif
        const u32 l = en & 15, sym = (en >> 4) & 0xfff;
        const bool active = st <= YT_BC;
        /* a code longer than the table's index: the entry names the second table (lk != 0 there, and no code of
           it is longer than lk + lw) */
        const bool is_link = active && l > lw + lk;
        u32 nb = is_link ? lw : l - lk;
        nb = active ? nb : 0;
        const u64 t64 = (((u64)hi << 32) | lo) >> nb; /* the bits behind the code */
        const u32 t = (u32)t64;
        bo += nb;
        la = is_link ? la + (sym << 1) : la;
        {
            const u32 sub_w = l - lw;
            lk = is_link ? lw : 0;
            lw = is_link ? sub_w : lw;
        }
        const bool sd = active && !is_link && bo <= endrel; /* a symbol, and the input held all of its code */
        u32 nk = YT_NONE;
        bool over = false;

        const bool dL = sd && st == YT_LIT;
        if (dL) { /* a literal: into the ring (yeast.c:1212-1223) */
            ring_put8(((u32)(uintptr_t)op - base_lo) & RM, sym);
            wn++;
            op++;
            insert--;
            nk = insert ? YT_LIT : mlen == 0 ? YT_END : after; /* :1227 */
        }
        const bool dC = sd && st == YT_CMD;
        if (dC) { /* insert_length(), copy_length(): yeast.c:652-693 */
            const u64 ce = ts->cmd[sym < 704 ? sym : 0];
            const u32 clo = (u32)ce, chi = (u32)(ce >> 32);
            const u32 n1 = (clo >> 16) & 0xff, n2 = clo >> 24;
            insert = (clo & 0xffff) + __builtin_amdgcn_ubfe(t, 0, n1);
            copy = (chi & 0xffff) + __builtin_amdgcn_ubfe((u32)(t64 >> n1), 0, n2);
            dcx = chi >> 16;
            after = sym < 128 ? YT_GO : YT_DIST; /* yeast.c:1234-1236 */
            bo += n1 + n2;
            over = insert > mlen; /* :1189 */
            mlen -= insert;
            nk = insert ? YT_LIT : after;
        }
        const bool dD = sd && st == YT_DIST;
        if (dD) { /* the distance symbol's extra bits now, the value with the copy's set-up */
            const u32 dn = sym - direct - 16;
            u32 dx = sym >= 16 + direct ? 1 + (dn >> (postfix + 1)) : 0;
            dx = dx > 24 ? 24 : dx;
            dsym = sym;
            dxv = __builtin_amdgcn_ubfe(t, 0, dx);
            bo += dx;
            nk = YT_GO;
        }
        bool btc = false; /* a block-switch type symbol was read: its count symbol is next */
        if (any_switching) {
            left0 -= dC && switching;
            left1 -= dL && switching;
            left2 -= dD && switching;
            if (YT_RARE(__any(sd && st >= YT_BT))) { /* block switch symbols (rare): yeast.c:1168-1180 */
                const bool dT = sd && st == YT_BT, dN = sd && st == YT_BC;
                u32 sh = 8 * cat;
                u32 k_type = (type3 >> sh) & 255, k_num = ((num3 >> sh) & 255) + 1, k_prev = (prev3 >> sh) & 255;
                u32 n = sym > 1 ? sym - 2 : sym ? (k_type + 1 == k_num ? 0 : k_type + 1) : k_prev;
                n = n >= k_num ? 0 : n;
                prev3 = dT ? (prev3 & ~(255u << sh)) | (k_type << sh) : prev3;
                type3 = dT ? (type3 & ~(255u << sh)) | (n << sh) : type3;
                /* block_length(): yeast.c:550-567 */
                u32 sb = sym, base, extra;
                if (sb < 4) { base = 1 + 4 * sb; extra = 2; }
                else if (sb < 8) { base = 17 + 8 * (sb - 4); extra = 3; }
                else if (sb < 12) { base = 49 + 16 * (sb - 8); extra = 4; }
                else if (sb < 16) { base = 113 + 32 * (sb - 12); extra = 5; }
                else if (sb < 18) { base = 241 + 64 * (sb - 16); extra = 6; }
                else if (sb >= 25) { base = 16625; extra = 24; }
                else { extra = sb - 11; base = 369 + ((1u << (sb - 11)) - 128); }
                u32 cnt = base + __builtin_amdgcn_ubfe(t, 0, extra);
                bo += dN ? extra : 0;
                const bool okN = dN && bo <= endrel;
                left0 = okN && cat == YT_CMD ? cnt : left0;
                left1 = okN && cat == YT_LIT ? cnt : left1;
                left2 = okN && cat == YT_DIST ? cnt : left2;
                if (okN) { /* the new type's tables (ordinary loads: the compiler waits for them, and for all else) */
                    u32 ty = (type3 >> sh) & 255;
                    if (cat == YT_CMD)
                        offI = wso + sld32(wsb, wso + YS_WS_ITAB + ty * 4);
                    else if (cat == YT_LIT) {
                        kmode = sld8(wsb, wso + YS_WS_MODE + ty) << 8;
                        if (ctx)
                            load_lt(ty);
                    } else if (dmap)
                        load_dt(ty);
                    nk = cat; /* the lookup the switch was for */
                }
                if (dT) {
                    la = wso + wl->bt[(2 * cat + 1) * S + slot];
                    lw = 8;
                    st = YT_BC;
                    btc = true;
                }
            }
        }
        asm volatile("" : "+v"(offI), "+v"(kmode));
        /* failures, behind one branch: the input ended inside the symbol or its extra bits (yeast.c:171), the
           insert length exceeds what is left of the meta-block (:1189) */
        {
            const bool eof = active && bo > endrel;
            if (YT_RARE(__any(eof || over))) {
                dry = eof && !rc ? 1 : dry;
                rc = eof && !rc ? (u32)YS_EOF : rc;
                rc = over && !rc ? (u32)YS_INVALID : rc;
                nk = eof || over ? YT_END : nk;
                btc = btc && !eof;
            }
        }

        TMARK(1);
        /* ================= 2. the other two loads of the last trip: input words, the copy's piece ================= */
#ifdef YS_PROFILE
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(ta_)::"memory");
#endif
        WAITV(1);
#ifdef YS_PROFILE
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tb_)::"memory");
        pwait2 += tb_ - ta_;
#endif
        {
            const bool got4 = treq != 0;
            const u32x4 tv = wl->top[slot];
            const bool at3 = nq == 3; /* else 4 */
            d3 = got4 && at3 ? tv.x : d3;
            d4 = got4 ? (at3 ? tv.y : tv.x) : d4;
            d5 = got4 ? (at3 ? tv.z : tv.y) : d5;
            d6 = got4 ? (at3 ? tv.w : tv.z) : d6;
            d7 = got4 && !at3 ? tv.w : d7;
            ip += got4 ? 16 : 0;
            nq += got4 ? 4 : 0;
        }
        peek();
        /* a copy of several pieces: this trip's piece; with the last one the next command's symbol is asked for */
        u32 m = crem < 16 ? crem : 16;
        m = m < cD ? m : cD;
        if (st == YT_HOLD)
            nk = crem != m ? YT_NONE : mlen == 0 ? YT_END : YT_CMD;

        TMARK(2);
#ifdef YS_PROFILE
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(ta_)::"memory");
#endif
        WAITV(0);
#ifdef YS_PROFILE
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tb_)::"memory");
        pwait3 += tb_ - ta_;
#endif
        const u32 ac = (u32)(uintptr_t)op - base_lo; /* bytes produced by this call */
        if (pn) { /* the piece asked for on the last trip: sixteen bytes whatever pn, the excess lies where the next bytes go */
            const LAS u8 *pb = po ? (const LAS u8 *)(rb + (po - 1)) : (const LAS u8 *)&wl->piece[slot];
            const u32x4 pv = *(const LAS u32x4u *)pb;
            const u32 o = (ac - pn) & RM;
            *(LAS u32x4u *)(rb + o) = pv;
            *(LAS u32x4u *)(rb + ring_twin(o)) = pv;
            wn += pn;
            pn = 0;
        }
        if (wn >= YT3_CHUNK) { /* at most YT3_CHUNK - 1 + 16 before: one chunk a trip keeps up */
            const u32 co = (ac - wn) & RM; /* a multiple of YT3_CHUNK */
#pragma unroll
            for (u32 j = 0; j < YT3_CHUNK; j += 16) {
#ifdef YT3_NT_STORE /* measurement aid: the output's lines are not kept in L2 */
                typedef u32x4 __attribute__((aligned(16))) u32x4al;
                __builtin_nontemporal_store(*(const LAS u32x4u *)(rb + co + j), (GAS u32x4al *)(op - wn + j));
#else
                gst128u(op - wn + j, *(const LAS u32x4u *)(rb + co + j));
#endif
            }
            wn -= YT3_CHUNK;
        }

        TMARK(3);
        /* ================= 3. what comes next, and the request for it ================= */
        {
            /* literal context -> table (context_id() yeast.c:698-755 through kctx; lit_map through lt),
               distance context -> table (yeast.c:1253-1257 through dt); the last two bytes are in the ring */
            const u32 pp = *(const LAS u16u *)(rb + ((ac - 2) & RM)); /* p2 | p1 << 8 */
            const u32 f1 = ((const LAS u8 *)ts->kctx)[(kmode + (pp >> 8)) * 2];
            const u32 f2 = ((const LAS u8 *)ts->kctx)[(kmode + (pp & 255)) * 2 + 1];
            const u32 mi = nk == YT_DIST ? 64 * S + dcx * S : (f1 | f2) * S; /* dt lies behind lt */
            const u32 o = (u32)((const LAS u16 *)wl->lt)[mi + slot] << 4;
            const u32 w = nk & 3; /* YT_GO: the next command */
            enter(nk <= YT_GO, w, w == YT_CMD ? offI : wso + o, left0, left1, left2);
        }
        st = nk == YT_END ? YT_END : st;
        TMARK(4);
        request(st <= YT_BC && (nk <= YT_GO || is_link || btc));
        TMARK(5);
#ifdef YT3_PAD_C /* measurement build: what an instruction costs at this place (dependent vector adds) */
        {
            u32 padv = st;
#pragma unroll
            for (int i = 0; i < YT3_PAD_C; i++)
                asm volatile("v_add_u32 %0, %0, 1" : "+v"(padv));
            asm volatile("" ::"v"(padv));
        }
#endif

        /* ================= 4. under those loads: the copy that begins, the next piece ================= */
        if (nk == YT_GO) {
            const u32 got = (u32)(uintptr_t)op - out_lo;
            const u32 maxd = got < wsize ? got : wsize;
            i32 dv = (i32)r0; /* yeast.c:1234-1236: the last distance, not pushed again */
            if (dD) { /* distance(): yeast.c:761-789 */
                const u32 ds = dsym;
                const u32 back = (0x555000e4u >> (2 * (ds & 15))) & 3;
                const i32 dl = (i32)((u32)(0x6051426051423333ull >> (4 * (ds & 15))) & 15) - 3;
                const u32 r01 = (back & 1) ? r1 : r0, r23 = (back & 1) ? r3 : r2;
                const i32 dring = (i32)((back & 2) ? r23 : r01) + dl;
                const u32 dn = ds - direct - 16;
                u32 dx = 1 + (dn >> (postfix + 1));
                dx = dx > 24 ? 24 : dx;
                const u32 o2 = ((2 + ((dn >> postfix) & 1)) << dx) - 4;
                const i32 dcode = (i32)(((o2 + dxv) << postfix) + (dn & ((1u << postfix) - 1)) + direct + 1);
                dv = ds < 16 + direct ? (i32)ds - 15 : dcode;
                dv = ds < 16 ? dring : dv;
                if (ds != 0 && dv > 0 && (u32)dv <= maxd) {
                    r3 = r2;
                    r2 = r1;
                    r1 = r0;
                    r0 = (u32)dv;
                }
            }
            const u32 dist = (u32)dv;
            bool bad = dv <= 0; /* zero: see DESIGN.md; negative: the reference's code 3 */
            /* the copy: from the window, or a static dictionary word (dict_word() yeast.c:908-949) */
            const bool isd = !bad && dist > maxd;
            u32 n = copy;
            const u8 *src = op - dist;
            u32 per = dist;
            if (isd) {
                const u32 id = dist - maxd - 1;
                const u32 ci = copy - 4; /* NDBITS[copy] (yeast.c:916-919), four bits per word length */
                const u32 ndlo = (u32)(0x7877899aaaaabbaaull >> (4 * (ci & 15))) & 15, ndhi = (0x55667u >> (4 * (ci & 7))) & 15;
                const u32 nd = ci < 16 ? ndlo : ndhi;
                const u32 xform = id >> nd;
                bad = ci > 20 || xform >= 121; /* :921-924, :929-930 */
                const u32 xe = ts->xbt[(xform < 121 ? xform : 0) * YS_XD_LENS + (ci > 20 ? 0 : ci)];
                n = xe >> 25;
                src = xdict + (u64)((xe & 0x1ffffffu) + (id & ((1u << nd) - 1)) * n);
                per = 0x40000000u;
            }
            bad = bad || n > mlen; /* :1269, :1283 */
            if (YT_RARE(__any(bad))) {
                rc = bad && !rc ? (u32)YS_INVALID : rc;
                n = bad ? 0 : n;
                st = bad ? YT_END : st;
            }
            csrc = src;
            cD = per;
            crem = n;
            mlen -= n;
            m = n < 16 ? n : 16;
            m = m < per ? m : per;
            /* the meta-block ends with this copy (:1295): the entry asked for above is never looked at */
            st = mlen == 0 ? YT_END : st;
        }
        TMARK(6);
        /* the next piece of the copy: requested now, appended to the ring on the next trip */
        {
            const bool piece = crem != 0;
            const u32 back = (u32)(uintptr_t)op - (u32)(uintptr_t)csrc;
            const bool nearc = piece && cD < 0x40000000u && back < YT3_NEAR; /* a dictionary word is never near */
            lds_load16_nt(piece && !nearc ? csrc : wsb, lds_piece);
            po = nearc ? (((u32)(uintptr_t)csrc - base_lo) & RM) + 1 : 0;
            pn = m;
            op += m;
            crem -= m;
            const bool whole = m == cD; /* a whole period is written: twice as far back is the same data */
            cD = whole ? cD * 2 : cD;
            csrc += whole ? 0 : m;
            /* more pieces to come: the stream stands still meanwhile (and the meta-block may end with the copy) */
            st = crem != 0 ? YT_HOLD : st;
        }
#ifdef YT3_PAD_B /* measurement build: what an instruction costs at this place (dependent vector adds) */
        {
            u32 padv = st;
#pragma unroll
            for (int i = 0; i < YT3_PAD_B; i++)
                asm volatile("v_add_u32 %0, %0, 1" : "+v"(padv));
            asm volatile("" ::"v"(padv));
        }
#endif
        TMARK(7);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef YS_PROFILE
    if ((threadIdx.x & 63) == 0 && a.cap >= 128) { /* the profile build's output bytes are not meant to be looked at */
        u64 *prof = (u64 *)(a.out + a.cap - 128);
        prof[12] = ntrip;
        prof[2] = pwait;
        prof[3] = pwait2;
        prof[4] = pwait3;
        for (int i = 0; i < 7; i++)
            prof[5 + i] = pt[i];
        prof[15] = pt[7];
        prof[13] = t_loop - a.t_start;
        prof[14] = __builtin_amdgcn_s_memrealtime() - t_loop;
    }
#endif
    { /* the piece set up last, then what the ring still holds, byte by byte */
        if (pn) {
            const LAS u8 *pb = po ? (const LAS u8 *)(rb + (po - 1)) : (const LAS u8 *)&wl->piece[slot];
            const u32x4 pv = *(const LAS u32x4u *)pb;
            const u32 o = ((u32)(uintptr_t)op - base_lo - pn) & RM;
            *(LAS u32x4u *)(rb + o) = pv;
            *(LAS u32x4u *)(rb + ring_twin(o)) = pv;
            wn += pn;
            pn = 0;
        }
        const u32 co = (u32)(uintptr_t)op - base_lo - wn;
#pragma clang loop unroll(disable) vectorize(disable)
        for (u32 j = 0; j < wn; j++)
            gst8(op - wn + j, rb[(co + j) & RM]);
    }
    TripOut o;
    o.outp = op;
    o.r0 = r0; o.r1 = r1; o.r2 = r2; o.r3 = r3;
    o.mlen = mlen;
    o.rc = rc;
    o.dry = dry;
    o.pos = a.total - endrel + bo;
    return o;
}

#endif
