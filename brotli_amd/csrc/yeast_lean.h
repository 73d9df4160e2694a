/*
 * yeast_lean.h -- the command loop of the batch kernel, one stream per lane (included by
 * yeast_stream.h; reference: the loop of yeast.c:1166-1295).
 *
 * One pass of the loop takes every stream of the wavefront through the phases of a command in
 * the order the format puts them: (A) the insert-and-copy symbol with its extra bits, (B) a
 * literal, twice, (C) the distance symbol, (D) the copy.  A stream takes part in a phase when it
 * is at that point of its command, so a stream whose command has no literals goes through a whole
 * command per pass and the instructions of a phase always serve most of the lanes; streams in a
 * long run of literals use only the (B) phases, streams in a long copy only (D).  Each decoding
 * phase has one table lookup and waits for memory once.
 *
 *   - bytes are moved by a small per-stream copy engine: phase (D) sets up a piece of up to 16
 *     bytes (98 % of the copies of text streams are no longer), moved by one 16-byte load, issued
 *     at once, and one 16-byte store after the next pass's phase (A) wait.  Literal contexts after
 *     a copy therefore never wait.  Overlapping copies (byte-serial semantics of
 *     yeast.c:1285-1293) go in pieces no longer than their period, the period doubling with every
 *     whole period written;
 *   - static dictionary words are copies too: the source is the table of all 121 transforms of
 *     all words (ys_params.xdict, built on the device at start-up by k_expand_dict from
 *     dict_word(): yeast.c:908-949), so no transform is evaluated here;
 *   - literals go straight to the output region;
 *   - the second half of a code longer than its root table is looked up on the next pass;
 *   - a block switch (type code, count code, reload of the new type's table offsets) is a rare
 *     phase of its own at the head of the pass;
 *   - input arrives through a four-word queue behind the 64-bit window, topped up by one 8-byte
 *     load per pass.
 */
#ifndef YEAST_LEAN_H
#define YEAST_LEAN_H

#ifndef YS_LIT_SLOTS
#define YS_LIT_SLOTS 2 /* literals a stream can decode per pass */
#endif

/* per-wavefront LDS: [index][stream], so the streams of a wavefront use different banks */
template <int S>
struct LeanLds {
    u16 lt[64 * S]; /* literal table (byte offset >> 4) per context, streams with one literal block type */
    u16 dt[4 * S];  /* distance table per distance context, streams with one distance block type */
};

/* expanded dictionary index: per transform and word length, the byte offset of the first entry */
#define YS_XD_LENS 21 /* word lengths 4..24 */
#define YS_XD_INDEX (121 * YS_XD_LENS)

template <int S, bool SIMPLE> /* SIMPLE: no stream of the wavefront switches block types */
DEV void run_lean(Ctx &c, u32 &mlen, const FlatEnv &fe)
{
    enum { ST_CMD = 0, ST_LIT = 1, ST_DIST = 2, ST_END = 3, ST_BSW = 4, ST_COPY = 6, ST_START = 7 };
    Rd &r = c.r;
    LAS LeanLds<S> *wl = (LAS LeanLds<S> *)c.wl;
    const u32 slot = threadIdx.x & 63;
    const bool ctx = fe.ctx;
    const u64 *cmd_tab = c.sh->cmd;
    const u32 *kcls = c.sh->kcls;
    u8 *const ws = c.ws;
    u8 *const out_end = c.out + c.cap;
    u32 cur_mode = fe.cur_mode;
    u32 offI = gld32(fe.itab), offL = fe.offL0, offD = fe.offD0;
    /* the table offsets of the current literal and distance block types live in LDS */
    auto load_lt = [&](u32 type) {
        for (u32 n = 0; n < 64; n++)
            wl->lt[n * S + slot] = (u16)(gld32(fe.lit_tab + (type << 6) + n) >> 4);
    };
    auto load_dt = [&](u32 type) {
        for (u32 n = 0; n < 4; n++)
            wl->dt[n * S + slot] = (u16)(gld32(fe.dtab + (type << 2) + n) >> 4);
    };
    if (ctx)
        load_lt(0);
    if (c.dist_codes > 1)
        load_dt(0);
    /* root table width by what is decoded (the tables were built to match) */
    const u32 rootL = (c.rootcaps >> 8) & 255, rootI = (c.rootcaps >> 16) & 255, rootD = c.rootcaps >> 24;

    /* block switching state of the three categories, indexed by the state they belong to
       (ST_CMD 0, ST_LIT 1, ST_DIST 2), one byte each: number of types - 1, current, previous */
    u32 num3 = 0, type3 = 0, prev3 = 0x010101u, left0 = 0, left1 = 0, left2 = 0;
    if (!SIMPLE) {
        num3 = (c.I.num - 1) | ((c.L.num - 1) << 8) | ((c.D.num - 1) << 16);
        left0 = c.I.left;
        left1 = c.L.left;
        left2 = c.D.left;
    }
    u32 st, cat = 0; /* cat: the state a block switch returns to */
    u32 link = 0;    /* first half of a long code of the state the stream is in */
    u32 insert = 0, copy = 0, implicit = 0;
    i32 dist = 0;
    /* the last two output bytes as their class words (Shared.kcls): everything context_id() needs */
    u32 k1 = kcls[c.p1 & 255], k2 = kcls[c.p2 & 255];
    u64 got0 = got_of(c);
    u32 maxd = got0 > c.wsize ? c.wsize : (u32)got0;
    /* input: the window r.w and, behind it, a queue of up to four words (a0 = word r.idx of the stream,
       then a1, b0, b1), topped up by one 8-byte load per pass: requested beside the phase (A) lookup when
       two words or fewer are queued, merged when that phase's wait is over */
    u32 a0 = r.nxt, a1 = 0, b0 = 0, b1 = 0, nq = 1;
    u64 tq = 0;    /* the request in flight */
    u32 treq = 0;
    /* copy engine */
    const u8 *psrc = c.outp; /* the piece set up on the previous pass: fetched and stored on this one */
    u8 *pdst = c.outp;
    u32 pn = 0;    /* its bytes */
    u32 psafe = 0; /* it lies at the very end of the output region: nothing is read or written beyond it */
    const u8 *csrc = c.outp;
    u32 crem = 0, cD = 0; /* copy in progress: bytes left, period limit */
    u32x4 pv = {0, 0, 0, 0}; /* the piece's bytes */


    /* move the next word into the window (a stream that uses up its queue inside one pass -- more than
       64 bits beyond what the window held -- fetches on the spot and waits) */
    auto request = [&]() {
        treq = nq <= 2;
        if (treq)
            tq = gld64u(r.p + r.idx + nq);
    };
    auto merge = [&]() {
        if (treq) {
            u32 t0 = (u32)tq, t1 = (u32)(tq >> 32);
            a0 = nq == 0 ? t0 : a0;
            a1 = nq == 0 ? t1 : nq == 1 ? t0 : a1;
            b0 = nq == 1 ? t1 : nq == 2 ? t0 : b0;
            b1 = nq == 2 ? t1 : b1;
            nq += 2;
            treq = 0;
        }
    };
    auto refill = [&]() {
        const bool need = r.have <= 32;
        if (need && nq == 0) { /* rare: the queue ran dry inside a pass */
            if (treq) /* the words on their way are the ones wanted */
                merge();
            else {
                u64 t = gld64u(r.p + r.idx);
                a0 = (u32)t;
                a1 = (u32)(t >> 32);
                nq = 2;
            }
        }
        /* the usual case without branches */
        u32 add = r.rem < 32 ? r.rem : 32;
        add = need ? add : 0;
        r.w |= need ? (u64)a0 << (r.have & 63) : 0;
        r.have += add;
        r.rem -= add;
        const bool adv = add != 0;
        r.idx += adv;
        a0 = adv ? a1 : a0;
        a1 = adv ? b0 : a1;
        b0 = adv ? b1 : b0;
        nq -= adv;
    };
    auto bits_mid = [&](u32 n, bool &ok) -> u32 {
        if (n > r.have)
            refill();
        if (n > r.have) {
            rd_eof(r);
            n = 0;
            ok = false;
        }
        u32 v = (u32)r.w & ((1u << n) - 1);
        r.w >>= n;
        r.have -= n;
        return v;
    };
    /* the lookup of a phase: request the entry for the streams that take part */
    auto lookup = [&](bool on, u32 off, u32 root) -> u32 {
        if (on)
            refill();
        u32 e = 0;
        if (on) {
            u32 peek = (u32)r.w;
            u32 idx = link ? (link >> 4) + ((peek >> root) & ((1u << ((link & 15) - root)) - 1)) : (peek & ((1u << root) - 1));
            e = gld16((const u16 *)(ws + off) + idx);
        }
        return e;
    };
    /* consume the code; false: nothing decoded (first half of a long code, or the input ran out) */
    auto decode = [&](bool on, u32 e, u32 root) -> bool {
        u32 l = e & 15;
        bool first_half = on && !link && l > root;
        link = on ? (first_half ? e : 0) : link;
        bool ok = on && !first_half;
        l = ok ? l : 0;
        if (l > r.have) {
            rd_eof(r);
            l = 0;
            ok = false;
        }
        r.w >>= l;
        r.have -= l;
        return ok;
    };
    /* a whole symbol at once, both levels (block switches only) */
    auto decode_now = [&](u32 off) -> u32 {
        refill();
        u32 peek = (u32)r.w;
        u32 e = gld16((const u16 *)(ws + off) + (peek & 255));
        u32 l = e & 15;
        if (l > ROOT) {
            e = gld16((const u16 *)(ws + off) + (e >> 4) + ((peek >> ROOT) & ((1u << (l - ROOT)) - 1)));
            l = e & 15;
        }
        if (l > r.have) {
            rd_eof(r);
            l = 0;
        }
        r.w >>= l;
        r.have -= l;
        return e >> 4;
    };
    /* move to a symbol state; a used-up block count turns it into a block switch first
       (yeast.c:1168-1180, 1192-1204, 1239-1251) */
    auto enter = [&](u32 s) {
        st = s;
        if (!SIMPLE) {
            if ((num3 >> (8 * s)) & 255) {
                u32 left = s == 0 ? left0 : s == 1 ? left1 : left2;
                if (left == 0) {
                    st = ST_BSW;
                    cat = s;
                }
                left0 -= s == 0 && left;
                left1 -= s == 1 && left;
                left2 -= s == 2 && left;
            }
        }
    };

    /* nothing loaded above is still on its way when the loop starts (a wait placed inside the loop
       for the first pass's sake would be paid on every pass) */
    asm volatile("" : "+v"(offI), "+v"(offL), "+v"(offD), "+v"(cur_mode), "+v"(a0));
#ifdef YS_PROFILE
    u64 pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pmark = __builtin_readcyclecounter();
    const u64 t_loop = __builtin_amdgcn_s_memrealtime();
    u32 npass = 0;
#define LMARK(i) do { u64 now_ = __builtin_readcyclecounter(); pt[i] += now_ - pmark; pmark = now_; } while (0)
#define LWAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define LMARK(i) do { } while (0)
#define LWAIT() do { } while (0)
#endif
    enter(ST_CMD);
    /* the phase (A) lookup of a pass is requested at the end of the pass before it, beside the piece
       and the input request, so that the loop back and the head of the pass run under its latency */
    u32 eA = lookup(st == ST_CMD, offI, rootI);
    u32 hadA = st == ST_CMD;
    request();
    while (st != ST_END) {
#ifdef YS_PROFILE
        npass++;
#endif
        LMARK(11);
        /* ---- block switch (rare): type code, count code, the new type's tables ---- */
        if (!SIMPLE && st == ST_BSW) {
            u32 sh = 8 * cat;
            u32 t_type = cat == ST_CMD ? c.I.t_type : cat == ST_LIT ? c.L.t_type : c.D.t_type;
            u32 t_count = cat == ST_CMD ? c.I.t_count : cat == ST_LIT ? c.L.t_count : c.D.t_count;
            u32 sym = decode_now(t_type);
            u32 k_type = (type3 >> sh) & 255, k_num = ((num3 >> sh) & 255) + 1, k_prev = (prev3 >> sh) & 255;
            u32 n = sym > 1 ? sym - 2 : sym ? (k_type + 1 == k_num ? 0 : k_type + 1) : k_prev;
            if (n >= k_num)
                n = 0; /* only reachable after a read failure; keeps table indices in range */
            prev3 = (prev3 & ~(255u << sh)) | (k_type << sh);
            type3 = (type3 & ~(255u << sh)) | (n << sh);
            u32 s = decode_now(t_count), base, extra; /* block_length(): yeast.c:550-567 */
            if (s < 4) { base = 1 + 4 * s; extra = 2; }
            else if (s < 8) { base = 17 + 8 * (s - 4); extra = 3; }
            else if (s < 12) { base = 49 + 16 * (s - 8); extra = 4; }
            else if (s < 16) { base = 113 + 32 * (s - 12); extra = 5; }
            else if (s < 18) { base = 241 + 64 * (s - 16); extra = 6; }
            else if (s >= 25) { base = 16625; extra = 24; }
            else { extra = s - 11; base = 369 + ((1u << (s - 11)) - 128); }
            bool ok = true;
            u32 left = base + bits_mid(extra, ok) - 1; /* the symbol that follows is the block's first */
            if (cat == ST_CMD) {
                left0 = left;
                offI = gld32(fe.itab + n);
            } else if (cat == ST_LIT) {
                left1 = left;
                cur_mode = gld8(fe.mode + n);
                if (ctx)
                    load_lt(n);
            } else {
                left2 = left;
                if (c.dist_codes > 1)
                    load_dt(n);
            }
            st = r.rc ? (u32)ST_END : cat;
            asm volatile("" : "+v"(offI), "+v"(cur_mode));
        }

        LMARK(0);
        /* ---- (A) the piece set up on the last pass, and the insert-and-copy symbol ---- */
        {
            if (pn && psafe) { /* at most 8 bytes, one at a time (the usual piece was requested in (D)) */
                u64 t = 0;
                for (u32 b = 0; b < pn; b++)
                    t |= (u64)gld8(psrc + b) << (8 * b);
                pv.x = (u32)t;
                pv.y = (u32)(t >> 32);
            }
            const bool on = st == ST_CMD;
            u32 e = eA;
            if (!SIMPLE) { /* a stream that has just switched blocks could not be looked up ahead */
                u32 late = lookup(on && !hadA, offI, rootI);
                e = on && !hadA ? late : e;
            }
            LMARK(1);
            LWAIT();
            asm volatile("" : "+v"(e), "+v"(tq), "+v"(pv));
            merge();
            LMARK(2);
            if (pn) {
                if (psafe) { /* the last bytes of the output region: exact */
                    u64 t = (u64)pv.x | ((u64)pv.y << 32);
                    for (u32 b = 0; b < pn; b++)
                        gst8(pdst + b, (u32)(t >> (8 * b)) & 255);
                } else /* a whole unit; the excess is overwritten by what follows */
                    gst128u(pdst, pv);
                /* the last two bytes written: literal context of what follows */
                u32 k = pn >= 2 ? pn - 2 : 0, ki = k >> 2;
                u32 lo = ki == 0 ? pv.x : ki == 1 ? pv.y : ki == 2 ? pv.z : pv.w;
                u32 hi = ki == 0 ? pv.y : ki == 1 ? pv.z : ki == 2 ? pv.w : 0;
                u32 pe = __builtin_amdgcn_alignbyte(hi, lo, k & 3);
                u32 nk2 = kcls[pe & 255], nk1 = kcls[pn >= 2 ? (pe >> 8) & 255 : pv.x & 255];
                k2 = pn >= 2 ? nk2 : k1;
                k1 = nk1;
                pn = 0;
            }
            bool ok = decode(on, e, rootI);
            if (ok) {
                u32 sym = e >> 4;
                u64 ce = cmd_tab[sym < 704 ? sym : 0];
                u32 x1 = bits_mid(((u32)ce >> 16) & 0xff, ok);
                u32 x2 = bits_mid((u32)(ce >> 40), ok);
                if (ok) {
                    insert = ((u32)ce & 0xffff) + x1;
                    copy = ((u32)(ce >> 24) & 0xffff) + x2;
                    implicit = sym < 128;
                    if (insert > mlen) {
                        rd_fail(r, YS_INVALID); /* :1189 */
                        insert = 0;
                    }
                    mlen -= insert;
                    dist = (i32)c.r0;
                    if (insert)
                        enter(ST_LIT);
                    else if (implicit)
                        st = ST_START;
                    else
                        enter(ST_DIST);
                }
            }
            st = r.rc ? (u32)ST_END : st;
        }

        LMARK(3);
        /* ---- (B) a literal, twice ---- */
#pragma unroll
        for (int rep = 0; rep < YS_LIT_SLOTS; rep++) {
            const bool on = st == ST_LIT;
            if (ctx) { /* context_id(): yeast.c:698-755, from the class words, without branches */
                u32 c0 = (k1 >> 24) & 63, c1 = k1 >> 26;
                u32 c2 = (k1 & 63) | ((k2 >> 8) & 3), c3 = ((k1 >> 13) & 0x38) | ((k2 >> 16) & 7);
                u32 cid = cur_mode < 2 ? (cur_mode == 0 ? c0 : c1) : (cur_mode == 2 ? c2 : c3);
                u32 o = (u32)wl->lt[cid * S + slot] << 4;
                offL = on ? o : offL;
            }
            u32 e = lookup(on, offL, rootL);
            LMARK(4);
            LWAIT();
            asm volatile("" : "+v"(e));
            LMARK(5);
            if (decode(on, e, rootL)) {
                u32 sym = e >> 4;
                gst8(c.outp, sym);
                c.outp++;
                maxd = maxd < c.wsize ? maxd + 1 : maxd;
                k2 = k1;
                k1 = kcls[sym & 255];
                if (--insert == 0) {
                    if (mlen == 0)
                        st = ST_END; /* :1227 */
                    else if (implicit)
                        st = ST_START;
                    else
                        enter(ST_DIST);
                } else
                    enter(ST_LIT);
            }
            st = r.rc ? (u32)ST_END : st;
        }

        LMARK(6);
        /* ---- (C) the distance symbol: distance() yeast.c:761-789 ---- */
        {
            const bool on = st == ST_DIST;
            if (on && c.dist_codes > 1)
                offD = (u32)wl->dt[(copy > 4 ? 3 : copy - 2) * S + slot] << 4;
            u32 e = lookup(on, offD, rootD);
            LMARK(7);
            LWAIT();
            asm volatile("" : "+v"(e));
            LMARK(8);
            bool ok = decode(on, e, rootD);
            if (ok) {
                u32 ds = e >> 4;
                if (ds < 16) {
                    u32 back = ds < 4 ? ds : ds < 10 ? 0 : 1;
                    u32 rv = back == 0 ? c.r0 : back == 1 ? c.r1 : back == 2 ? c.r2 : c.r3;
                    u32 k = ds < 10 ? ds - 4 : ds - 10; /* -1 +1 -2 +2 -3 +3 */
                    int dl = (int)(k >> 1) + 1;
                    dl = (k & 1) ? dl : -dl;
                    dl = ds < 4 ? 0 : dl;
                    dist = (i32)rv + dl;
                } else if (ds < 16 + c.direct)
                    dist = (i32)ds - 15;
                else { /* at most 2^30 + 121: 32 bits are enough */
                    u32 dn = ds - c.direct - 16;
                    u32 dx = 1 + (dn >> (c.postfix + 1));
                    dx = dx > 25 ? 25 : dx; /* only after a read failure */
                    u32 x1 = bits_mid(dx, ok);
                    u32 o2 = ((2 + ((dn >> c.postfix) & 1)) << dx) - 4;
                    dist = (i32)(((o2 + x1) << c.postfix) + (dn & ((1u << c.postfix) - 1)) + c.direct + 1);
                }
                if (ok) {
                    if (ds && dist > 0 && dist <= (i32)maxd) {
                        c.r3 = c.r2; c.r2 = c.r1; c.r1 = c.r0; c.r0 = (u32)dist;
                    }
                    st = ST_START;
                }
            }
            st = r.rc ? (u32)ST_END : st;
        }

        LMARK(9);
        /* ---- (D) a copy begins: where its bytes come from ---- */
        if (st == ST_START) {
            if (dist <= 0) {
                rd_fail(r, YS_INVALID); /* zero: see DESIGN.md; negative: the reference's code 3 */
            } else if ((u32)dist > maxd) { /* static dictionary: dict_word() yeast.c:908-949 */
                u32 id = (u32)dist - maxd - 1;
                u32 ci = copy - 4; /* NDBITS[copy] (yeast.c:916-919), four bits per word length */
                u32 nd = ci < 16 ? (u32)(0x7877899aaaaabbaaull >> (4 * (ci & 15))) & 15 : (0x55667u >> (4 * (ci & 7))) & 15;
                u32 xform = id >> nd;
                if (copy > 24 || copy < 4 || xform >= 121) {
                    rd_fail(r, YS_INVALID); /* :921-924, :929-930 */
                } else {
                    u32 hdr = c.sh->xf[xform];
                    u32 op = hdr & 255, omit = (hdr >> 8) & 255;
                    u32 wlen = op == 3 || op == 4 ? (copy > omit ? copy - omit : 0) : copy;
                    u32 total = ((hdr >> 16) & 255) + wlen + (hdr >> 24);
                    if (total > mlen) {
                        rd_fail(r, YS_INVALID); /* :1269 */
                    } else {
                        u32 index = id & ((1u << nd) - 1);
                        csrc = c.xdict + c.sh->xbase[xform * YS_XD_LENS + copy - 4] + index * total;
                        crem = total;
                        cD = 0x40000000u;
                        mlen -= total;
                    }
                }
            } else if (copy > mlen) {
                rd_fail(r, YS_INVALID); /* :1283 */
            } else {
                csrc = c.outp - dist;
                crem = copy;
                cD = (u32)dist;
                mlen -= copy;
            }
            st = r.rc ? (u32)ST_END : (u32)ST_COPY;
            if (st == ST_COPY && crem == 0) { /* a transform that leaves nothing of its word */
                if (mlen)
                    enter(ST_CMD);
                else
                    st = ST_END;
            }
        }
        /* ---- and the next piece of the copy in progress (requested now, stored on the next pass) ---- */
        if (st == ST_COPY) {
            u32 m = crem < 16 ? crem : 16;
            m = m < cD ? m : cD;
            psrc = csrc;
            psafe = c.outp + 24 > out_end;
            m = psafe && m > 8 ? 8 : m;
            if (m == cD)
                cD *= 2; /* a whole period is written: twice as far back is the same data */
            else
                csrc += m;
            pdst = c.outp;
            pn = m;
            if (!psafe) /* requested now, used after the next pass's first wait */
                pv = gld128u_nt(psrc); /* read once, far back in the output: not worth a place in L2 */
            c.outp += m;
            crem -= m;
            u32 room = c.wsize - maxd;
            maxd += m < room ? m : room;
            if (crem == 0) {
                if (mlen)
                    enter(ST_CMD);
                else
                    st = ST_END;
            }
        }
        /* ---- ahead of the next pass: its phase (A) lookup and the input request ---- */
        hadA = st == ST_CMD;
        eA = lookup(hadA, offI, rootI);
        request();
        LMARK(10);
    }
#ifdef YS_PROFILE
    if ((threadIdx.x & 63) == 0 && c.cap >= 128) { /* the profile build's output bytes are not meant to be looked at */
        u64 *prof = (u64 *)(c.out + c.cap - 128);
        for (int i = 0; i < 12; i++)
            prof[i] = pt[i];
        prof[12] = npass;
        prof[13] = t_loop - c.t_start; /* header parsing and table building, 10 ns units */
        prof[14] = __builtin_amdgcn_s_memrealtime() - t_loop; /* the loop */
    }
#endif
    /* meta-block end (or failure): the piece set up last was valid */
    if (pn) {
        for (u32 b = 0; b < pn; b += 8) {
            u32 k = pn - b < 8 ? pn - b : 8;
            u64 v = 0;
            if (psafe || k < 8) {
                for (u32 j = 0; j < k; j++)
                    v |= (u64)gld8(psrc + b + j) << (8 * j);
                for (u32 j = 0; j < k; j++)
                    gst8(pdst + b + j, (u32)(v >> (8 * j)) & 255);
            } else
                gst64u(pdst + b, gld64u(psrc + b));
        }
    }
    group_sync_mem();
    { /* literal context for the next meta-block */
        u64 got = got_of(c);
        c.p1 = got >= 1 ? gld8(c.outp - 1) : 0;
        c.p2 = got >= 2 ? gld8(c.outp - 2) : 0;
    }
    /* hand the position back to the header reader */
    if (!r.rc)
        rd_seek(r, c.src, rd_pos(r));
}

#endif
