/*
 * yeast_trip.h -- the command loop of the batch kernel, one stream per lane (included by
 * yeast_stream.h; reference: the loop of yeast.c:1166-1295).
 *
 * One trip of the loop decodes ONE prefix-code symbol per stream -- whichever the stream needs next:
 * an insert-and-copy symbol, a literal, a distance symbol, a block-switch type or count symbol, or
 * the second half of a code longer than its root table -- so a trip has exactly one table lookup
 * and one wait for memory, and every live lane uses it.  The order inside a trip is chosen for the
 * length of the dependent chain from one lookup to the next:
 *
 *   1. the entry requested on the last trip arrives: consume the code and the extra bits that belong
 *      to it (insert/copy extra bits, distance extra bits) from a 64-bit funnel over the input words;
 *   2. bytes: store the piece loaded on the last trip and this trip's literal; the distance value and
 *      the ring buffer (distance() yeast.c:761-789), the static dictionary reference (dict_word()
 *      yeast.c:908-949, a copy from the expanded dictionary), the copy set-up with its validity checks,
 *      and the load of the copy's next piece;
 *   3. decide what the stream decodes next and where that table is (literal context -> table through
 *      the wavefront's LDS map) and request that entry and the input top-up; the loop back runs
 *      under their latency.
 *
 * Bytes are moved by a per-stream copy engine in pieces of up to 16 bytes: one 16-byte load when the
 * piece is set up, one 16-byte store a trip later; the literal context after a copy comes from the
 * bytes loaded with it.  Overlapping copies (byte-serial semantics, yeast.c:1285-1293) go in pieces no
 * longer than their period, the period doubling with every whole period written.  A copy of up to 16
 * bytes (98 % of the copies of text) costs no trip of its own: the next command's symbol is requested
 * on the trip that sets the copy up.
 *
 * The code is written without divergent branches: every step is computed for all lanes and selected,
 * memory operations are masked, and the rare cases (block switches, the last bytes of an output
 * region, a second input word in one trip) sit behind wavefront-uniform branches.
 */
#ifndef YEAST_TRIP_H
#define YEAST_TRIP_H

/* RING form: bytes of LDS per stream (the ring proper is [0, 64) behind 32 bytes of slack; what is written at
   [0, 16) is written at [64, 80) too and the other way round, so sixteen bytes can be read or written at any offset
   below 64 in one piece; 37 words: the streams' rings start in different banks) */
#define YT_RING_STRIDE 148
#define YT_RING_ALLOC 180 /* bytes per stream the ring array is sized for: the third loop form's layout (yeast_trip3.h) */
typedef u32x4 __attribute__((aligned(1))) u32x4u;

/* per-wavefront LDS: [index][stream], so the streams of a wavefront use different banks */
template <int S>
struct TripLds {
    u16 lt[64 * S]; /* literal table (byte offset >> 4) per context, current literal block type */
    u16 dt[4 * S];  /* distance table per distance context, current distance block type */
    u32 bt[6 * S];  /* block-switch tables (byte offsets): type code, count code of the three categories */
    /* where the loop's loads land, by lane (not by stream slot): the table entry, the input top-up, the copy's piece */
    u32 ent[64];
    u32x4 top[64];
    u32x4 piece[64];
    u32x4 unit[64]; /* UNIT form of the loop: the output unit being put together (a piece and the literals behind it);
                       RING form: where a piece taken from the ring waits for the next trip */
    u8 ring[S * YT_RING_ALLOC + 16]; /* RING form: the last 64 bytes of each stream's output (see run_trip_fn); third form: the
                                        last 128 (yeast_trip3.h) */
};

/* LDS image shared by the wavefronts of a block */
struct TripShared {
    u64 cmd[704];     /* per insert-and-copy symbol: insert base | insert extra << 16 | copy extra << 24, then
                         copy base | distance context << 16  (insert_length()/copy_length() yeast.c:652-693, :1254) */
    u16 kctx[4 * 256]; /* per context mode and byte value: contribution to context_id() (yeast.c:698-755) as the
                          last byte (low byte) and as the byte before it (high byte) */
    u32 xbt[121 * 21]; /* expanded dictionary: offset of (transform, word length) | entry size << 25 */
};

DEV void trip_shared_fill(TripShared &ts, const u8 *xforms, const u32 *xbase)
{
    for (u32 i = threadIdx.x; i < 704; i += blockDim.x) {
        u64 ce = c_cmd_tab[i];
        u32 ib = (u32)ce & 0xffff, n1 = ((u32)ce >> 16) & 0xff, cb = (u32)(ce >> 24) & 0xffff, n2 = (u32)(ce >> 40);
        u32 dctx = cb > 4 ? 3 : cb - 2;
        ts.cmd[i] = (u64)(ib | (n1 << 16) | (n2 << 24)) | ((u64)(cb | (dctx << 16)) << 32);
    }
    for (u32 i = threadIdx.x; i < 1024; i += blockDim.x) {
        u32 mode = i >> 8, b = i & 255;
        u32 l0 = c_ctx_lut[b], l1 = c_ctx_lut[256 + b], l2 = c_ctx_lut[512 + b];
        u32 f1 = mode == 0 ? (b & 63) : mode == 1 ? (b >> 2) : mode == 2 ? l0 : (l2 << 3);
        u32 f2 = mode < 2 ? 0 : mode == 2 ? l1 : l2;
        ts.kctx[i] = (u16)(f1 | (f2 << 8));
    }
    for (u32 i = threadIdx.x; i < 121 * 21; i += blockDim.x) {
        u32 t = i / 21, L = i % 21 + 4;
        const u8 *rec = xforms + 20 * t;
        u32 op = rec[0], omit = rec[1], plen = rec[2], slen = rec[3];
        u32 wlen = (op == 3 || op == 4) ? (L > omit ? L - omit : 0) : L;
        ts.xbt[i] = xbase[i] | ((plen + wlen + slen) << 25);
    }
    __syncthreads();
}

/* global memory through a wavefront-uniform base and a 32-bit lane offset */
DEV u32 sld16(const u8 *base, u32 off) { return YSB(base + (u64)off, 2) ? *(const GAS u16 *)(base + (u64)off) : 0; }
DEV u32 sld32(const u8 *base, u32 off) { return YSB(base + (u64)off, 4) ? *(const GAS u32 *)(base + (u64)off) : 0; }
DEV u32 sld8(const u8 *base, u32 off) { return YSB(base + (u64)off, 1) ? *(const GAS u8 *)(base + (u64)off) : 0; }
DEV u32x4 sld128(const u8 *base, u32 off) /* 4-byte aligned */
{
    typedef u32x4 __attribute__((aligned(4))) u32x4a;
    return YSB(base + (u64)off, 16) ? *(const GAS u32x4a *)(base + (u64)off) : (u32x4)0;
}

#define YT_CMD 0u
#define YT_LIT 1u
#define YT_DIST 2u
#define YT_BT 3u   /* block-switch type symbol */
#define YT_BC 4u   /* block-switch count symbol */
#define YT_END 6u

/* What the loop takes from the stream's state and hands back.  The loop is an out-of-line function so that it
   has the register file to itself: the rest of the per-stream state (Ctx, the header parser's variables, the
   kernel's arguments) waits on the stack instead of crowding the loop into spills. */
struct TripIn {
    const u8 *ws_base, *src, *out, *xdict;
    u8 *outp;
    LAS void *wl;
    LAS const TripShared *ts;
    u32 wso, rootcaps, flags; /* flags: 1 literal context map in use, 2 distance context map in use */
    u32 postfix, direct, wsize, cap, mlen;
    u32 bitpos, total;
    u32 p1, p2, r0, r1, r2, r3;
    u32 w_ltab, offL0, offD0, cur_mode;
    u32 num3, left0, left1, left2;
    u32 bt0, bt1, bt2, bt3, bt4, bt5;
#ifdef YS_PROFILE
    u64 t_start;
#endif
};
struct TripOut {
    u8 *outp;
    u32 r0, r1, r2, r3, mlen, rc, dry, pos;
};

/* Loads that land in the wavefront's LDS instead of registers (lane i's bytes at base + i * size).  The compiler
   keeps no count of them: the loop waits for them itself (s_waitcnt vmcnt below), and for that count to be known
   they are issued on every trip by every lane -- a lane with nothing to load reads the first bytes of the workspace
   (one line for all such lanes) and ignores what lands.  (Switching lanes off in exec instead costs ~50 cycles per
   load, tools/ubench/glds_issue.hip.) */
DEV void lds_load4(const u8 *sbase, u32 voff, u32 lds)
{
    voff = YSB(sbase + (u64)voff, 4) ? voff : 0u; /* bounds-checked build: a load out of range reads the base instead */
#ifdef YT_ENT_HINT /* measurement builds: the table entry's load with a cache hint (nt, sc0, sc1 ...) */
#define YT_STR2(x) #x
#define YT_STR(x) YT_STR2(x)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1 " YT_STR(YT_ENT_HINT) : : "v"(voff), "s"(sbase), "s"(lds) : "memory");
#else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(sbase), "s"(lds) : "memory");
#endif
}
DEV void lds_load16(const u8 *src, u32 lds)
{
#ifdef YS_BOUNDS
    src = YSB(src, 16) ? src : (const u8 *)(size_t)g_ysb.lo[0];
#endif
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds) : "memory");
}
DEV void lds_load16_nt(const u8 *src, u32 lds) /* read once: not worth a place in L2 */
{
#ifdef YS_BOUNDS
    src = YSB(src, 16) ? src : (const u8 *)(size_t)g_ysb.lo[0];
#endif
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" : : "v"(src), "s"(lds) : "memory");
}

/* SF: streams per wavefront, and in bit 8 the form of the loop's stores (UNIT below) */
template <int SF>
NOINL TripOut run_trip_fn(TripIn a)
{
    constexpr int S = SF & 0xff;
    /* UNIT: one 16-byte store per command -- the last piece of a copy is not stored when it lands but kept in LDS, the
       literals that follow are put behind it, and the unit goes out when it is full or when the next copy is about to
       read from the output.  Fewer requests to memory: 6 % faster when both wavefronts of every SIMD are busy (65,536
       streams and more); more LDS round trips per trip: 8 % slower below that -- so the kernel exists in both forms
       (yeast_lane.hip). */
    constexpr bool UNIT = (SF & 0x100) != 0;
    /* RING: chunk write-behind.  Every output byte goes through a 64-byte ring per stream in LDS; a trip stores at most
       one chunk of sixteen bytes and only a complete one, on the 16-byte grid that starts where this call's output
       starts -- so each byte is written to memory once.  A copy from less than 48 bytes back takes its piece from the
       ring (the bytes may not be in memory yet); one from further back reads memory, where everything but the last
       fifteen bytes is.  What is left at the end of the loop goes out byte by byte, so nothing is written beyond the
       output region and the loop needs no special case for the region's last bytes. */
    constexpr bool RING = (SF & 0x200) != 0;
    LAS TripLds<S> *wl = (LAS TripLds<S> *)a.wl;
    LAS const TripShared *ts = a.ts;
    const u32 slot = threadIdx.x & 63;
    const u8 *const wsb = (const u8 *)uni64((u64)a.ws_base);
    const u8 *const xdict = (const u8 *)uni64((u64)a.xdict);
    const u32 wso = a.wso;
    const u32 rootcaps = uni(a.rootcaps);
    const u32 rootL = (rootcaps >> 8) & 255, rootI = (rootcaps >> 16) & 255, rootD = rootcaps >> 24;
    const bool ctx = a.flags & 1;
    const bool dmap = a.flags & 2;
    const u32 postfix = a.postfix, direct = a.direct, wsize = a.wsize;
    const u32 out_lo = (u32)(uintptr_t)a.out;
    const u32 cap = a.cap;
    u32 mlen = a.mlen;
    u32 rc = 0, dry = 0;
    /* the three landing places of this wavefront's loads */
    const u32 lds_ent = uni((u32)(uintptr_t)&wl->ent[0]);
    const u32 lds_top = uni((u32)(uintptr_t)&wl->top[0]);
    const u32 lds_piece = uni((u32)(uintptr_t)&wl->piece[0]);

    /* ---- per-type tables of the current block types ---- */
    const u32 w_ltab = a.w_ltab;
    auto load_lt = [&](u32 type) { /* 64 offsets, sixteen per round of loads */
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
    auto load_dt = [&](u32 type) {
#pragma clang loop unroll(disable)
        for (u32 n = 0; n < 4; n++)
            wl->dt[n * S + slot] = (u16)(sld32(wsb, wso + YS_WS_DTAB + ((type << 2) + n) * 4) >> 4);
    };
    if (ctx)
        load_lt(0);
    if (dmap)
        load_dt(0);
    u32 offI = wso + sld32(wsb, wso + YS_WS_ITAB);
    const u32 offL0 = a.offL0, offD0 = a.offD0;
    u32 kmode = a.cur_mode << 8; /* index of the mode's row in kctx */
    /* block switching of the three categories (CMD 0, LIT 1, DIST 2), one byte each: number of types - 1,
       current type, previous type; the counts of symbols left in the current blocks */
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

    /* ---- input: eight words behind the bit offset bo (< 32 at the head of a trip); 64 bits from bo come out of
       d0..d2 by two funnel shifts.  Topped up by one 16-byte load while four words or fewer are held. */
    u32 d0, d1, d2, d3, d4, d5, d6, d7, nq = 8, bo;
    const u8 *ip;
    u32 endrel; /* bits from bit 0 of d0 to the end of the stream */
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

    /* ---- output and copy engine ---- */
    u8 *op = a.outp;
    u32 p1 = a.p1 & 255, p2 = a.p2 & 255;
    u32 r0 = a.r0, r1 = a.r1, r2 = a.r2, r3 = a.r3;
    u32 insert = 0, copy = 0, dctx = 0;
    u32 dsym = 0, dxv = 0;
    const u8 *csrc = a.outp; /* copy in progress: source of the next piece, bytes left, period limit */
    u32 crem = 0, cD = 0;
    u32 wn = 0;         /* UNIT: bytes in the unit that is being put together; RING: bytes in the ring that are not in memory yet */
    bool pl = false;    /* RING: the piece asked for on the last trip came from the ring and waits in wl->unit */
    const u32 base_lo = (u32)(uintptr_t)a.outp; /* RING: where this call's output starts: byte k of it lives at ring offset k & 63 */
    LAS u8 *const rb = (LAS u8 *)wl->ring + (slot & (S - 1)) * YT_RING_STRIDE + 32;
    if constexpr (RING) { /* the output of earlier meta-blocks that a copy may reach through the ring */
        u64 got0 = (u64)(a.outp - a.out);
        u32 k = got0 < 48 ? (u32)got0 : 48u;
        for (u32 j = 1; j <= k; j++) {
            u32 o = (0u - j) & 63, v = gld8(a.outp - j);
            rb[o] = (u8)v;
            rb[o < 32 ? o + 64 : o - 64] = (u8)v;
        }
    }
    u32 pn = 0; /* bytes of the piece requested on the last trip (it lands in wl->piece and goes to op - pn) */

    /* ---- the lookup in flight (its entry lands in wl->ent) ---- */
    u32 st = YT_CMD, cat = 0;
    u32 la = 0, lw = 0, lk = 0; /* table base (offset from wsb), index width; lk: root width of the code whose
                                   second half is being looked up, else 0 */
    u32 lo = 0, hi = 0;         /* 64 bits of input from bo */
    u32 treq = 0;               /* a top-up is in flight */
    u32 e = 0;                  /* the entry last asked for */
    bool asked = false;         /* an entry was asked for on the last trip */
    u32 after = YT_DIST;        /* what follows the literals of the command: YT_DIST, or YT_START = the copy */
#define YT_START 8u
#ifdef YS_PROFILE
#define TMARK(i) do { u64 now_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); pt[i] += now_ - plast; plast = now_; } while (0)
#else
#define TMARK(i) do { } while (0)
#endif

    /* what a stream decodes next: w = category wanted (0..2).  A used-up block count turns it into a block
       switch first (yeast.c:1168-1180, 1192-1204, 1239-1251). */
    auto enter = [&](bool on, u32 w, u32 off_w, u32 l0, u32 l1, u32 l2) {
        u32 root_w = w == YT_CMD ? rootI : w == YT_LIT ? rootL : rootD;
        u32 nst = w, nla = off_w, nlw = root_w;
        if (any_switching) {
            u32 left = w == YT_CMD ? l0 : w == YT_LIT ? l1 : l2;
            bool sw = ((num3 >> (8 * w)) & 255) != 0 && left == 0;
            if (__any(on && sw)) {
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
    /* move whole words out of the queue while bo >= 32 (once; a second time behind a uniform branch) */
    auto advance = [&]() {
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
    /* The three loads of a trip, always in this order and always all three (with no lane enabled if need be):
       the entry of the lookup described by (la, lw) for the lanes in `on`, the input top-up -- and, by the caller
       when the copy is set up, the copy's next piece.  So the entry has arrived when two loads or fewer are in
       flight, whatever the stores of the trip (issued before it) are doing. */
    auto peek = [&]() {
        advance();
        if (__any(bo >= 32))
            advance();
        lo = __builtin_amdgcn_alignbit(d1, d0, bo);
        hi = __builtin_amdgcn_alignbit(d2, d1, bo);
    };
    auto request = [&](bool on) {
        lds_load4(wsb, on ? la + (__builtin_amdgcn_ubfe(lo, 0, lw) << 1) : 0u, lds_ent);
        asked = on;
        /* no more top-ups once the queue holds the rest of the stream: nothing is read far beyond its end */
        treq = nq <= 4 && st != YT_END && endrel > nq * 32;
        lds_load16(treq ? ip : wsb, lds_top);
    };

    enter(true, YT_CMD, offI, left0, left1, left2);
    peek();
    request(true);
    lds_load16_nt(wsb, lds_piece);
    asm volatile("" : "+v"(offI), "+v"(kmode), "+v"(d0), "+v"(d7));

#ifdef YS_PROFILE
    u32 ntrip = 0;
    u64 pwait = 0, pwait2 = 0, pwait3 = 0, plast;
    u64 pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(plast)::"memory");
    const u64 t_loop = __builtin_amdgcn_s_memrealtime();
#endif
    /* No stream needs more trips than this (a trip that neither writes a byte nor reads a bit is a block switch
       or the symbol of a one-symbol code, a bounded number per byte written): the loop ends whatever the data.
       A stream cut short by it is handed to the retry pass like one whose tables did not fit. */
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
    while (__any(st != YT_END || crem != 0)) {
        if (--budget == 0) {
            rc = (st != YT_END || crem != 0) && !rc ? (u32)YS_WORKSPACE : rc;
            break;
        }
        /* ================= 1. the entry arrives: consume it ================= */
#ifdef YS_PROFILE
        ntrip++;
        u64 ta_, tb_;
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(ta_)::"memory");
#endif
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#ifdef YS_PROFILE
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tb_)::"memory");
        pwait += tb_ - ta_;
#endif
        {
            u32 en = wl->ent[slot];
            asm volatile("" : "+v"(en)); /* read by every lane: a branch around the read costs more than the read */
            e = asked ? en & 0xffff : e; /* a stream in the middle of a copy keeps the entry it holds */
        }
        TMARK(0);
        const bool hold = crem != 0; /* a copy of several pieces is going on: the entry waits */
        const bool active = st <= YT_BC && !hold;
        u32 l = e & 15, sym = e >> 4;
        const bool is_link = active && lk == 0 && l > lw;
        u32 nb = is_link ? lw : l - lk;
        nb = active ? nb : 0;
        const u64 t64 = (((u64)hi << 32) | lo) >> nb; /* the bits behind the code */
        const u32 t = (u32)t64;
        bo += nb;
        bool eof = bo > endrel;
        const bool done = active && !is_link && !eof;
        la = is_link ? la + (sym << 1) : la;
        const u32 lw_link = l - lw;
        lk = is_link ? lw : 0;
        lw = is_link ? lw_link : lw;
        const bool dC = done && st == YT_CMD, dL = done && st == YT_LIT, dD = done && st == YT_DIST;

        /* insert-and-copy symbol: insert_length(), copy_length() yeast.c:652-693 */
        u32 ins_new;
        {
            u64 ce = ts->cmd[sym < 704 ? sym : 0];
            u32 clo = (u32)ce, chi = (u32)(ce >> 32);
            u32 n1 = (clo >> 16) & 0xff, n2 = clo >> 24;
            u32 x1 = __builtin_amdgcn_ubfe(t, 0, n1);
            u32 x2 = __builtin_amdgcn_ubfe((u32)(t64 >> n1), 0, n2);
            ins_new = (clo & 0xffff) + x1;
            copy = dC ? (chi & 0xffff) + x2 : copy;
            dctx = dC ? chi >> 16 : dctx;
            after = dC ? (sym < 128 ? YT_START : YT_DIST) : after; /* yeast.c:1234-1236 */
            /* distance symbol: its extra bits now, the value under the request */
            u32 dn = sym - direct - 16;
            u32 dx = sym >= 16 + direct ? 1 + (dn >> (postfix + 1)) : 0;
            dx = dx > 24 ? 24 : dx; /* cannot be: the alphabet ends at 16 + direct + (48 << postfix) */
            dsym = dD ? sym : dsym;
            dxv = dD ? __builtin_amdgcn_ubfe(t, 0, dx) : dxv;
            u32 ext = dC ? n1 + n2 : dx;
            bo += dC || dD ? ext : 0;
        }
        insert = dC ? ins_new : insert - dL;
        left0 -= dC && switching;
        left1 -= dL && switching;
        left2 -= dD && switching;

        /* block switch symbols (rare) */
        bool redo = false; /* the lookup of the category the switch was for */
        if (any_switching && __any(done && st >= YT_BT)) {
            const bool dT = done && st == YT_BT, dN = done && st == YT_BC;
            u32 sh = 8 * cat;
            u32 k_type = (type3 >> sh) & 255, k_num = ((num3 >> sh) & 255) + 1, k_prev = (prev3 >> sh) & 255;
            u32 n = sym > 1 ? sym - 2 : sym ? (k_type + 1 == k_num ? 0 : k_type + 1) : k_prev;
            n = n >= k_num ? 0 : n; /* only after a read failure; keeps table indices in range */
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
            }
            if (dT) { /* next: the count symbol */
                la = wso + wl->bt[(2 * cat + 1) * S + slot];
                lw = 8;
                st = YT_BC;
            }
            redo = okN;
        }
        asm volatile("" : "+v"(offI), "+v"(kmode));
        eof = eof || (active && bo > endrel);
        TMARK(1);

        /* ================= 2. the other two loads of the last trip: input words, the copy's piece ================= */
#ifdef YS_PROFILE
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(ta_)::"memory");
#endif
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
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
        const bool lits = insert != 0;
        const bool cl = dC || dL;
        { /* the insert length against what is left of the meta-block; end of input */
            const bool over = dC && !eof && insert > mlen; /* :1189 */
            mlen -= dC && !eof && !over ? insert : 0;
            dry = active && eof && !rc ? 1 : dry;
            rc = (active && eof) && !rc ? (u32)YS_EOF : rc;
            rc = over && !rc ? (u32)YS_INVALID : rc;
        }
        const bool ends = (dL && !lits && mlen == 0) || eof; /* :1227 */
        const bool to_after = cl && !lits && !ends;
        const bool start = !rc && ((to_after && after == YT_START) || (dD && !eof)); /* a copy begins on this trip */
        const bool wantL = (cl && lits) || (redo && cat == YT_LIT);
        const bool wantD = (to_after && after == YT_DIST) || (redo && cat == YT_DIST);
        /* the next command's symbol is requested on the trip that sets the copy up, before the copy is known to be
           valid and to leave something of the meta-block: if not, the entry is never looked at */
        const bool wantC = start || (redo && cat == YT_CMD);
        const bool want = (wantL || wantD || wantC) && !rc && !eof;
        TMARK(2);

        /* the piece requested on the last trip (the youngest load: waited for as late as can be) goes out -- a whole
           unit; the excess is overwritten by what follows -- then this trip's literal; the literal context: the last
           two bytes of the piece, or the literal */
#ifdef YS_PROFILE
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(ta_)::"memory");
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef YS_PROFILE
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tb_)::"memory");
        pwait3 += tb_ - ta_;
#endif
        if constexpr (RING) {
            const u32 ac = (u32)(uintptr_t)op - base_lo; /* bytes produced by this call, the piece that landed included */
            if (pn) {
                const LAS u8 *pb = pl ? (const LAS u8 *)&wl->unit[slot] : (const LAS u8 *)&wl->piece[slot];
                u32 b1 = pb[pn - 1], b2 = pb[pn >= 2 ? pn - 2 : 0];
                const u32x4 pv = *(const LAS u32x4 *)pb;
                const u32 o = (ac - pn) & 63;
                *(LAS u32x4u *)(rb + o) = pv; /* sixteen bytes whatever pn: the excess lies where the next bytes go */
                *(LAS u32x4u *)(rb + (o < 32 ? o + 64 : o - 64)) = pv;
                p2 = pn >= 2 ? b2 : p1;
                p1 = b1;
            }
            wn += pn;
            pn = 0;
            if (dL) {
                const u32 o = ac & 63;
                rb[o] = (u8)sym;
                rb[o < 32 ? o + 64 : o - 64] = (u8)sym;
            }
            wn += dL;
            op += dL;
            const bool fl = wn >= 16; /* at most 15 + 16 before: one chunk a trip keeps up */
            if (fl) {
                const u32 co = ((u32)(uintptr_t)op - base_lo - wn) & 63; /* a multiple of 16 */
                gst128u(op - wn, *(const LAS u32x4u *)(rb + co));
            }
            wn -= fl ? 16u : 0u;
        } else if constexpr (UNIT) {
            const bool reads = start | (crem != 0); /* a piece is loaded on this trip: the output must be in memory */
            const bool dead = ends | (rc != 0) | (st == YT_END);
            LAS u8 *ub = (LAS u8 *)&wl->unit[slot];
            if (pn) { /* the piece that landed opens a unit (the one before went out when this piece was asked for) */
                const LAS u8 *pb = (const LAS u8 *)&wl->piece[slot];
                u32 b1 = pb[pn - 1], b2 = pb[pn >= 2 ? pn - 2 : 0];
                wl->unit[slot] = wl->piece[slot];
                p2 = pn >= 2 ? b2 : p1;
                p1 = b1;
            }
            wn = pn ? pn : wn;
            pn = 0;
            if (dL)
                ub[wn] = (u8)sym;
            wn += dL;
            op += dL;
            const bool fl = (wn != 0) & ((wn == 16) | reads | dead);
            const u32 gotw = (u32)(uintptr_t)op - out_lo;
            const bool wnear = fl & (gotw - wn + 16 > cap); /* the last 16 bytes of the region: nothing is written beyond it */
            if (fl & !wnear)
                gst128u(op - wn, wl->unit[slot]);
            if (__any(wnear)) {
                if (wnear) {
#pragma clang loop unroll(disable) vectorize(disable)
                    for (u32 j = 0; j < wn; j++)
                        gst8(op - wn + j, ub[j]);
                }
            }
            wn = fl ? 0 : wn;
        } else {
            /* the piece goes out when it has landed (a whole unit; the excess is overwritten by what follows), then this
               trip's literal */
            if (pn) {
                const LAS u8 *pb = (const LAS u8 *)&wl->piece[slot];
                u32 b1 = pb[pn - 1], b2 = pb[pn >= 2 ? pn - 2 : 0];
                gst128u(op - pn, wl->piece[slot]);
                p2 = pn >= 2 ? b2 : p1;
                p1 = b1;
            }
            pn = 0;
            if (dL)
                gst8(op, sym);
            op += dL;
        }
        p2 = dL ? p1 : p2;
        p1 = dL ? sym : p1;
        TMARK(3);

        /* ================= 3. what comes next, and the request for it ================= */
        /* literal context -> table (context_id() yeast.c:698-755 through kctx; lit_map through lt),
           distance context -> table (yeast.c:1253-1257 through dt) */
        u32 offL = offL0, offD = offD0;
        {
            u32 f1 = ((const LAS u8 *)ts->kctx)[(kmode + p1) * 2];
            u32 f2 = ((const LAS u8 *)ts->kctx)[(kmode + p2) * 2 + 1];
            u32 cid = f1 | f2;
            u32 mi = wantD ? 64 * S + dctx * S : cid * S; /* dt lies behind lt */
            u32 o = (u32)((const LAS u16 *)wl->lt)[mi + slot] << 4;
            offL = ctx ? wso + o : offL;
            offD = dmap ? wso + o : offD;
        }
        {
            u32 w = wantL ? YT_LIT : wantD ? YT_DIST : YT_CMD;
            u32 off_w = wantL ? offL : wantD ? offD : offI;
            enter(want, w, off_w, left0, left1, left2);
        }
        st = rc || ends ? YT_END : st;
        TMARK(4);
        request(st <= YT_BC && (want || is_link || (active && st >= YT_BT)));

        TMARK(5);
#ifdef YT_EXP_PAD /* measurement build: what an instruction costs (dependent vector adds, placed under the loads) */
        {
            u32 padv = st;
#pragma unroll
            for (int i = 0; i < YT_EXP_PAD; i++)
                asm volatile("v_add_u32 %0, %0, 1" : "+v"(padv));
            asm volatile("" :: "v"(padv));
        }
#endif
        /* ================= 4. under those loads: the copy that begins or goes on, its next piece ================= */
        const u32 got = (u32)(uintptr_t)op - out_lo;
        u32 m = crem < 16 ? crem : 16;
        m = m < cD ? m : cD;
        {
            /* distance(): yeast.c:761-789 (every lane computes it; the lanes in `start` use it).  Ring buffer
               entry and offset per symbol below 16 from packed constants: -1 +1 -2 +2 -3 +3 on the last and the
               second-last distance */
            u32 ds = dsym;
            u32 back = (0x555000e4u >> (2 * (ds & 15))) & 3;
            i32 dl = (i32)((u32)(0x6051426051423333ull >> (4 * (ds & 15))) & 15) - 3;
            u32 r01 = (back & 1) ? r1 : r0, r23 = (back & 1) ? r3 : r2;
            u32 rv = (back & 2) ? r23 : r01;
            i32 dring = (i32)rv + dl;
            u32 dn = ds - direct - 16;
            u32 dx = 1 + (dn >> (postfix + 1));
            dx = dx > 24 ? 24 : dx;
            u32 o2 = ((2 + ((dn >> postfix) & 1)) << dx) - 4;
            i32 dcode = (i32)(((o2 + dxv) << postfix) + (dn & ((1u << postfix) - 1)) + direct + 1); /* <= 2^30 + 121 */
            i32 dv = ds < 16 + direct ? (i32)ds - 15 : dcode;
            dv = ds < 16 ? dring : dv;
            dv = dD ? dv : (i32)r0; /* yeast.c:1234-1236: the last distance, not pushed again */
            const u32 dist = (u32)dv;
            bool bad = dv <= 0; /* zero: see DESIGN.md; negative: the reference's code 3 */
            const u32 maxd = got < wsize ? got : wsize;
            const bool push = start && dD && ds != 0 && !bad && dist <= maxd;
            r3 = push ? r2 : r3;
            r2 = push ? r1 : r2;
            r1 = push ? r0 : r1;
            r0 = push ? dist : r0;
            /* the copy: from the window, or a static dictionary word (dict_word() yeast.c:908-949) */
            const bool isd = !bad && dist > maxd;
            u32 id = dist - maxd - 1;
            u32 ci = copy - 4; /* NDBITS[copy] (yeast.c:916-919), four bits per word length */
            u32 ndlo = (u32)(0x7877899aaaaabbaaull >> (4 * (ci & 15))) & 15, ndhi = (0x55667u >> (4 * (ci & 7))) & 15;
            u32 nd = ci < 16 ? ndlo : ndhi;
            u32 xform = id >> nd;
            const bool dbad = isd && (ci > 20 || xform >= 121); /* :921-924, :929-930 */
            u32 xi = (xform < 121 ? xform : 0) * YS_XD_LENS + (ci > 20 ? 0 : ci);
            u32 xe = ts->xbt[xi];
            u32 total = xe >> 25;
            u32 n = isd ? total : copy;
            bad = bad || dbad || n > mlen; /* :1269, :1283 */
            bad = bad && start;
            rc = bad && !rc ? (u32)YS_INVALID : rc;
            const u8 *sd = xdict + (u64)((xe & 0x1ffffffu) + (id & ((1u << nd) - 1)) * total);
            const u8 *sw = op - dist;
            const bool go = start && !bad;
            csrc = go ? (isd ? sd : sw) : csrc;
            cD = go ? (isd ? 0x40000000u : dist) : cD;
            crem = go ? n : crem;
            mlen -= go ? n : 0;
            u32 m2 = crem < 16 ? crem : 16;
            m2 = m2 < cD ? m2 : cD;
            m = start ? m2 : m;
            /* the meta-block ends with this copy (:1295), or the copy is no good: the entry requested above is dropped */
            st = bad || (go && mlen == 0) ? YT_END : st;
        }
        TMARK(6);
        /* the next piece of the copy in progress: requested now, stored on the next trip */
        if constexpr (RING) {
            const bool piece = crem != 0;
            const bool nearc = piece && (u64)(op - csrc) < 48; /* a dictionary word is never that close */
            m = piece ? m : 0;
            lds_load16_nt(piece && !nearc ? csrc : wsb, lds_piece);
            if (__any(nearc)) {
                if (nearc)
                    wl->unit[slot] = *(const LAS u32x4u *)(rb + (((u32)(uintptr_t)csrc - base_lo) & 63));
            }
            pl = nearc;
            pn = m;
            op += m;
            crem -= m;
            const bool whole = m == cD; /* a whole period is written: twice as far back is the same data */
            cD = whole ? cD * 2 : cD;
            csrc += whole ? 0 : m;
        } else {
            const bool piece = crem != 0;
            const bool tail = piece && got + 16 > cap; /* the last bytes of the output region: nothing is written beyond it */
            if (__any(tail)) {
                if (tail) { /* byte by byte, at once */
                    u32 last1 = p1, last2 = p2;
                    for (u32 j = 0; j < m; j++) {
                        u32 v = gld8(csrc + j);
                        gst8(op + j, v);
                        last2 = last1;
                        last1 = v;
                    }
                    p1 = last1;
                    p2 = last2;
                }
            }
            m = piece ? m : 0;
            const bool ld = piece && !tail;
            lds_load16_nt(ld ? csrc : wsb, lds_piece);
            pn = ld ? m : 0;
            op += m;
            crem -= m;
            const bool whole = m == cD; /* a whole period is written: twice as far back is the same data */
            cD = whole ? cD * 2 : cD;
            csrc += whole ? 0 : m;
        }
        TMARK(7);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef YS_PROFILE
    if ((threadIdx.x & 63) == 0 && cap >= 128) { /* the profile build's output bytes are not meant to be looked at */
        u64 *prof = (u64 *)(a.out + cap - 128);
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
    if constexpr (RING) { /* the piece set up last, then what the ring still holds, byte by byte */
        if (pn) {
            const u32x4 pv = pl ? wl->unit[slot] : wl->piece[slot];
            const u32 o = ((u32)(uintptr_t)op - base_lo - pn) & 63;
            *(LAS u32x4u *)(rb + o) = pv;
            *(LAS u32x4u *)(rb + (o < 32 ? o + 64 : o - 64)) = pv;
            wn += pn;
            pn = 0;
        }
        const u32 co = (u32)(uintptr_t)op - base_lo - wn;
#pragma clang loop unroll(disable) vectorize(disable)
        for (u32 j = 0; j < wn; j++)
            gst8(op - wn + j, rb[(co + j) & 63]);
        wn = 0;
    }
    /* meta-block end (or failure): the piece set up last was valid */
    if (pn)
        gst128u(op - pn, wl->piece[slot]);
    if constexpr (UNIT) {
        const LAS u8 *ub = (const LAS u8 *)&wl->unit[slot];
#pragma clang loop unroll(disable) vectorize(disable)
        for (u32 j = 0; j < wn; j++)
            gst8(op - wn + j, ub[j]);
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

#include "yeast_trip3.h"

template <int SF>
DEV void run_trip(Ctx &c, u32 &mlen, const FlatEnv &fe)
{
    Rd &r = c.r;
    if (got_of(c) + mlen >= 0xffffff00ull) { /* the loop counts output bytes in 32 bits: 4 GiB per stream and more are refused */
        rd_fail(r, YS_NOMEM);
        return;
    }
    TripIn a;
    a.ws_base = c.ws_base; a.src = c.src; a.out = c.out; a.xdict = c.xdict;
    a.outp = c.outp;
    a.wl = (LAS void *)c.wl;
    a.ts = (LAS const TripShared *)c.tsh;
    a.wso = (u32)(c.ws - c.ws_base);
    a.rootcaps = c.rootcaps;
    a.flags = (fe.ctx ? 1u : 0u) | (c.dist_codes > 1 ? 2u : 0u);
    a.postfix = c.postfix; a.direct = c.direct; a.wsize = c.wsize;
    a.cap = c.cap > 0xffffffffull ? 0xffffffffu : (u32)c.cap;
    a.mlen = mlen;
    a.bitpos = rd_pos(r); a.total = r.total;
    a.p1 = c.p1; a.p2 = c.p2; a.r0 = c.r0; a.r1 = c.r1; a.r2 = c.r2; a.r3 = c.r3;
    a.w_ltab = a.wso + (u32)((const u8 *)fe.lit_tab - c.ws);
    a.offL0 = a.wso + fe.offL0; a.offD0 = a.wso + fe.offD0;
    a.cur_mode = fe.cur_mode;
    a.num3 = (c.I.num - 1) | ((c.L.num - 1) << 8) | ((c.D.num - 1) << 16);
    a.left0 = c.I.left; a.left1 = c.L.left; a.left2 = c.D.left;
    a.bt0 = c.I.t_type; a.bt1 = c.I.t_count; a.bt2 = c.L.t_type; a.bt3 = c.L.t_count; a.bt4 = c.D.t_type; a.bt5 = c.D.t_count;
#ifdef YS_PROFILE
    a.t_start = c.t_start;
#endif
    TripOut o;
    if constexpr ((SF >> 8) == 3)
        o = run_trip3_fn<(SF & 0xff), 128u, 64u>(a);
    else if constexpr ((SF >> 8) == 4)
        o = run_trip3_fn<(SF & 0xff), 64u, 16u>(a);
    else
        o = run_trip_fn<SF>(a);
    c.outp = o.outp;
    c.r0 = o.r0; c.r1 = o.r1; c.r2 = o.r2; c.r3 = o.r3;
    mlen = o.mlen;
    group_sync_mem();
    { /* literal context for the next meta-block */
        u64 got = got_of(c);
        c.p1 = got >= 1 ? gld8(c.outp - 1) : 0;
        c.p2 = got >= 2 ? gld8(c.outp - 2) : 0;
    }
    /* hand the position back to the header reader */
    if (o.rc) {
        r.rc = o.rc;
        r.dry = o.dry;
        r.fpos = o.dry ? r.total : o.pos;
        r.have = 0;
        r.rem = 0;
    } else
        rd_seek(r, c.src, o.pos);
}

#endif
