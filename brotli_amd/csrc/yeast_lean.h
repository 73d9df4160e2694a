/*
 * yeast_lean.h -- the command loop of the batch kernel, one stream per lane (included by
 * yeast_stream.h; reference: the loop of yeast.c:1166-1295).
 *
 * Every trip of the loop decodes at most one prefix-code symbol per stream, so the streams that
 * share a wavefront meet at the one table lookup of the trip whatever they are decoding:
 * insert-and-copy, literal, distance, block type or block count.  What a trip does besides the
 * lookup is kept short:
 *
 *   - one place where the wavefront waits for memory: the lookup, the two prefetched input words
 *     and the copy bytes requested on the previous trip all arrive there (every table offset the
 *     loop needs is in LDS or in a register; only a block switch reloads some from memory);
 *   - bytes are moved by a small per-stream copy engine: a trip that reaches a copy (or is in the
 *     middle of a long one) sets up a piece of up to 32 bytes; the next trip fetches it together
 *     with its lookup and stores it.  The command symbol that always lies between a copy and the next literal gives that
 *     trip for free, so literal contexts never wait.  Overlapping copies (byte-serial semantics of
 *     yeast.c:1285-1293) go in pieces no longer than their period, the period doubling with every
 *     whole period written; periods below 8 bytes are expanded in registers;
 *   - static dictionary words are copies too: the source is the table of all 121 transforms of
 *     all words (ys_params.xdict, built on the device at start-up by k_expand_dict from
 *     dict_word(): yeast.c:908-949), so no transform is evaluated here;
 *   - literals go straight to the output region.
 */
#ifndef YEAST_LEAN_H
#define YEAST_LEAN_H

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
    enum { ST_CMD = 0, ST_LIT = 1, ST_DIST = 2, ST_END = 3, ST_BT = 4, ST_BC = 5, ST_COPY = 6, ACT_NONE = 7 };
    Rd &r = c.r;
    LeanLds<S> *wl = (LeanLds<S> *)c.wl;
    const u32 slot = threadIdx.x & 63;
    const bool ctx = fe.ctx;
    const u64 *cmd_tab = c.sh->cmd;
    const u8 *lut = c.sh->lut;
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
    u32 off = 0, link = 0;
    u32 insert = 0, copy = 0, implicit = 0;
    u64 got0 = got_of(c);
    u32 maxd = got0 > c.wsize ? c.wsize : (u32)got0;
    /* input: the window r.w and the next two words, requested a trip ahead */
    u32 q0 = r.nxt, q1 = 0, nq = 1; /* q0 = word r.idx of the stream, q1 the one after it */
    /* copy engine */
    const u8 *psrc = c.outp; /* the piece set up on the previous trip: fetched and stored on this one */
    u8 *pdst = c.outp;
    u32 pn = 0, pmode = 0; /* its bytes; period (1..7) of a short-period piece, else 0 */
    u32 psafe = 0;         /* it lies at the very end of the output region: nothing is read beyond it */
    const u8 *csrc = c.outp;
    u32 crem = 0, cD = 0; /* copy in progress: bytes left, period limit */

    /* move the next word into the window (both queued words have arrived by the time this runs,
       except after three refills within two trips: then the word is fetched on the spot) */
    auto refill = [&]() {
        if (r.have <= 32) {
            if (nq == 0) {
                q0 = gld32(r.p + r.idx);
                nq = 1;
            }
            r.w |= (u64)q0 << r.have;
            u32 add = r.rem < 32 ? r.rem : 32;
            r.have += add;
            r.rem -= add;
            if (add) {
                r.idx++;
                q0 = q1;
                nq--;
            }
        }
    };
    auto bits_mid = [&](u32 n, u32 &act) -> u32 {
        if (n > r.have)
            refill();
        if (n > r.have) {
            rd_eof(r);
            n = 0;
            act = ACT_NONE;
        }
        u32 v = (u32)r.w & ((1u << n) - 1);
        r.w >>= n;
        r.have -= n;
        return v;
    };
    /* move to a symbol state; a used-up block count turns it into a block switch first
       (yeast.c:1168-1180, 1192-1204, 1239-1251) */
    auto enter = [&](u32 s) {
        st = s;
        if (!SIMPLE) {
            if ((num3 >> (8 * s)) & 255) {
                u32 left = s == 0 ? left0 : s == 1 ? left1 : left2;
                if (left == 0) {
                    st = ST_BT;
                    cat = s;
                }
                left0 -= s == 0 && left;
                left1 -= s == 1 && left;
                left2 -= s == 2 && left;
            }
        }
    };
    /* the pieces requested on the previous trip go to the output region */
    auto store_pending = [&](u64 a0, u64 a1, u64 a2, u64 a3) {
        if (pmode) { /* period below 8: byte i of the piece is byte i mod period of the source */
            u32 d = pmode;
            u64 pat = a0 & ~(~0ull << (8 * d));
            pat |= d < 8 ? pat << (8 * d) : 0;
            pat |= 2 * d < 8 ? pat << (16 * d) : 0;
            pat |= 4 * d < 8 ? pat << (32 * d) : 0;
            u32 r1 = (0x1230200u >> (4 * (d - 1))) & 15; /* 8 mod d */
            u32 r2 = 2 * r1 >= d ? 2 * r1 - d : 2 * r1;
            u32 r3 = r2 + r1 >= d ? r2 + r1 - d : r2 + r1;
            a0 = pat;
            a1 = (pat >> (8 * r1)) | (pat << (8 * (d - r1)));
            a2 = (pat >> (8 * r2)) | (pat << (8 * (d - r2)));
            a3 = (pat >> (8 * r3)) | (pat << (8 * (d - r3)));
        }
        if (pdst + ((pn + 7) & ~7u) <= out_end) { /* whole units; the excess is overwritten by what follows */
            gst64u(pdst, a0);
            if (pn > 8)
                gst64u(pdst + 8, a1);
            if (pn > 16)
                gst64u(pdst + 16, a2);
            if (pn > 24)
                gst64u(pdst + 24, a3);
        } else { /* the last bytes of the output region: exact */
            put_le(pdst, a0, pn);
            if (pn > 8)
                put_le(pdst + 8, a1, pn - 8);
            if (pn > 16)
                put_le(pdst + 16, a2, pn - 16);
            if (pn > 24)
                put_le(pdst + 24, a3, pn - 24);
        }
        if (ctx) { /* the last two bytes written: literal context of what follows (without
                      context modelling they are read back at the end of the meta-block) */
            u32 k1 = pn - 1, q = k1 >> 3;
            u64 x1 = q == 0 ? a0 : q == 1 ? a1 : q == 2 ? a2 : a3;
            u32 b1 = (u32)(x1 >> (8 * (k1 & 7))) & 255;
            u32 b2 = c.p1;
            if (pn >= 2) {
                u32 k2 = pn - 2;
                q = k2 >> 3;
                u64 x2 = q == 0 ? a0 : q == 1 ? a1 : q == 2 ? a2 : a3;
                b2 = (u32)(x2 >> (8 * (k2 & 7))) & 255;
            }
            c.p1 = b1;
            c.p2 = b2;
        }
        pn = 0;
    };

    /* nothing loaded above is still on its way when the loop starts (a wait placed inside the loop
       for the first trip's sake would be paid on every trip) */
    asm volatile("" : "+v"(offI), "+v"(offL), "+v"(offD), "+v"(cur_mode), "+v"(q0));
    enter(ST_CMD);
    while (st != ST_END) {
        const bool look = st != ST_COPY;
        u32 root = st == ST_LIT ? rootL : st == ST_CMD ? rootI : rootD;
        if (!link) {
            off = st == ST_CMD ? offI : st == ST_LIT ? offL : offD;
            if (!SIMPLE) {
                if (st == ST_BT)
                    off = cat == ST_CMD ? c.I.t_type : cat == ST_LIT ? c.L.t_type : c.D.t_type;
                else if (st == ST_BC)
                    off = cat == ST_CMD ? c.I.t_count : cat == ST_LIT ? c.L.t_count : c.D.t_count;
            }
        }
        if (!SIMPLE)
            root = st == ST_BT || st == ST_BC ? (u32)ROOT : root;
        /* ---- the one table lookup of this trip, and the input words after the queued ones ---- */
        refill();
        u32 peek = (u32)r.w;
        u32 idx = link ? (link >> 4) + ((peek >> root) & ((1u << ((link & 15) - root)) - 1)) : (peek & ((1u << root) - 1));
        /* every load of the trip is issued here, and waited for once */
        u64 pd0 = 0, pd1 = 0, pd2 = 0, pd3 = 0;
        if (pn) {
            if (psafe) {
                for (u32 b = 0; b < pn; b++)
                    pd0 |= (u64)gld8(psrc + b) << (8 * b);
            } else {
                pd0 = gld64u(psrc);
                if (pn > 8 && !pmode)
                    pd1 = gld64u(psrc + 8);
                if (pn > 16 && !pmode)
                    pd2 = gld64u(psrc + 16);
                if (pn > 24 && !pmode)
                    pd3 = gld64u(psrc + 24);
            }
        }
        if (nq == 0)
            q0 = gld32(r.p + r.idx);
        if (nq < 2)
            q1 = gld32(r.p + r.idx + 1);
        nq = 2;
        u32 e = 0;
        if (look)
            e = gld16((const u16 *)(ws + off) + idx);
        asm volatile("" : "+v"(e), "+v"(q0), "+v"(q1), "+v"(pd0), "+v"(pd1), "+v"(pd2), "+v"(pd3));

        if (pn)
            store_pending(pd0, pd1, pd2, pd3);

        u32 l = e & 15;
        bool first_half = look && !link && l > root;
        link = first_half ? e : 0;
        l = first_half || !look ? 0 : l;
        u32 act = first_half ? (u32)ACT_NONE : st;
        if (l > r.have) {
            rd_eof(r);
            l = 0;
            act = ACT_NONE;
        }
        r.w >>= l;
        r.have -= l;
        const u32 sym = e >> 4;

        bool start_copy = false;
        i64 dist = c.r0;
        if (act == ST_LIT) {
            gst8(c.outp, sym);
            c.outp++;
            maxd = maxd < c.wsize ? maxd + 1 : maxd;
            c.p2 = c.p1;
            c.p1 = sym;
            if (--insert == 0) {
                if (mlen == 0)
                    st = ST_END; /* :1227 */
                else if (implicit)
                    start_copy = true;
                else
                    enter(ST_DIST);
            } else
                enter(ST_LIT);
        } else if (act == ST_CMD) {
            u64 ce = cmd_tab[sym < 704 ? sym : 0];
            u32 x1 = bits_mid(((u32)ce >> 16) & 0xff, act);
            u32 x2 = bits_mid((u32)(ce >> 40), act);
            if (act == ST_CMD) {
                insert = ((u32)ce & 0xffff) + x1;
                copy = ((u32)(ce >> 24) & 0xffff) + x2;
                implicit = sym < 128;
                if (insert > mlen) {
                    rd_fail(r, YS_INVALID); /* :1189 */
                    insert = 0;
                }
                mlen -= insert;
                if (insert)
                    enter(ST_LIT);
                else if (implicit)
                    start_copy = true;
                else
                    enter(ST_DIST);
            }
        } else if (act == ST_DIST) { /* distance(): yeast.c:761-789 */
            u32 ds = sym;
            if (ds < 16) {
                u32 back = ds < 4 ? ds : ds < 10 ? 0 : 1;
                u32 rv = back == 0 ? c.r0 : back == 1 ? c.r1 : back == 2 ? c.r2 : c.r3;
                u32 k = ds < 10 ? ds - 4 : ds - 10; /* -1 +1 -2 +2 -3 +3 */
                int dl = (int)(k >> 1) + 1;
                dl = (k & 1) ? dl : -dl;
                dl = ds < 4 ? 0 : dl;
                dist = (i64)rv + dl;
            } else if (ds < 16 + c.direct)
                dist = ds - 15;
            else {
                u32 dn = ds - c.direct - 16;
                u32 dx = 1 + (dn >> (c.postfix + 1));
                dx = dx > 25 ? 25 : dx; /* only after a read failure */
                u32 x1 = bits_mid(dx, act);
                u64 o2 = ((u64)(2 + ((dn >> c.postfix) & 1)) << dx) - 4;
                dist = (i64)(((o2 + x1) << c.postfix) + (dn & ((1u << c.postfix) - 1)) + c.direct + 1);
            }
            if (act == ST_DIST) {
                if (ds && dist > 0 && dist <= (i64)maxd) {
                    c.r3 = c.r2; c.r2 = c.r1; c.r1 = c.r0; c.r0 = (u32)dist;
                }
                start_copy = true;
            }
        } else if (!SIMPLE && act == ST_BT) { /* block switch, type code: yeast.c:1168-1180 */
            u32 sh = 8 * cat;
            u32 k_type = (type3 >> sh) & 255, k_num = ((num3 >> sh) & 255) + 1, k_prev = (prev3 >> sh) & 255;
            u32 n = sym > 1 ? sym - 2 : sym ? (k_type + 1 == k_num ? 0 : k_type + 1) : k_prev;
            if (n >= k_num)
                n = 0; /* only reachable after a read failure; keeps table indices in range */
            prev3 = (prev3 & ~(255u << sh)) | (k_type << sh);
            type3 = (type3 & ~(255u << sh)) | (n << sh);
            st = ST_BC;
        } else if (!SIMPLE && act == ST_BC) { /* and the new block's count: block_length() yeast.c:550-567 */
            u32 s = sym, base, extra;
            if (s < 4) { base = 1 + 4 * s; extra = 2; }
            else if (s < 8) { base = 17 + 8 * (s - 4); extra = 3; }
            else if (s < 12) { base = 49 + 16 * (s - 8); extra = 4; }
            else if (s < 16) { base = 113 + 32 * (s - 12); extra = 5; }
            else if (s < 18) { base = 241 + 64 * (s - 16); extra = 6; }
            else if (s >= 25) { base = 16625; extra = 24; }
            else { extra = s - 11; base = 369 + ((1u << (s - 11)) - 128); }
            u32 x = bits_mid(extra, act);
            if (act == ST_BC) {
                u32 left = base + x - 1; /* the symbol that follows is the block's first */
                u32 type = (type3 >> (8 * cat)) & 255;
                /* the new type's table offsets come from memory: the one place besides the lookup
                   where a trip waits for it (block switches are rare) */
                if (cat == ST_CMD) {
                    left0 = left;
                    offI = gld32(fe.itab + type);
                } else if (cat == ST_LIT) {
                    left1 = left;
                    cur_mode = gld8(fe.mode + type);
                    if (ctx)
                        load_lt(type);
                } else {
                    left2 = left;
                    if (c.dist_codes > 1)
                        load_dt(type);
                }
                st = cat;
            }
        }

        /* ---- a copy begins: where its bytes come from ---- */
        if (start_copy && !r.rc) {
            if (dist <= 0) {
                rd_fail(r, YS_INVALID); /* zero: see DESIGN.md; negative: the reference's code 3 */
            } else if (dist > (i64)maxd) { /* static dictionary: dict_word() yeast.c:908-949 */
                u64 id = (u64)dist - maxd - 1;
                u32 de = c.sh->dict[copy < 32 ? copy : 0];
                u32 nd = de >> 24;
                u64 xform = id >> nd;
                if (copy > 24 || copy < 4 || xform >= 121) {
                    rd_fail(r, YS_INVALID); /* :921-924, :929-930 */
                } else {
                    u32 hdr = c.sh->xf[(u32)xform];
                    u32 op = hdr & 255, omit = (hdr >> 8) & 255;
                    u32 wlen = op == 3 || op == 4 ? (copy > omit ? copy - omit : 0) : copy;
                    u32 total = ((hdr >> 16) & 255) + wlen + (hdr >> 24);
                    if (total > mlen) {
                        rd_fail(r, YS_INVALID); /* :1269 */
                    } else {
                        u32 index = (u32)id & ((1u << nd) - 1);
                        csrc = c.xdict + c.sh->xbase[(u32)xform * YS_XD_LENS + copy - 4] + index * total;
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
            st = ST_COPY;
        }
        /* ---- set up the next piece of the copy in progress (fetched on the next trip) ---- */
        if (st == ST_COPY && !r.rc) {
            u32 m = crem < 32 ? crem : 32;
            psrc = csrc;
            if (c.outp + 40 > out_end && cD < 0x40000000u) {
                /* the last bytes of the output region: short pieces, fetched byte by byte */
                m = m < cD ? m : cD;
                m = m < 8 ? m : 8;
                pmode = 0;
                psafe = 1;
                csrc += m;
            } else if (cD < 8) {
                pmode = cD;
                psafe = 0;
                /* what is left continues one whole number of periods back */
                u32 back = (0x1c1e1e201e2020ull >> (8 * (cD - 1))) & 255; /* largest multiple of the period <= 32 */
                csrc = c.outp + 32 - back;
                cD = back;
            } else {
                pmode = 0;
                psafe = 0;
                m = m < cD ? m : cD;
                if (m == cD)
                    cD *= 2; /* a whole period is written: twice as far back is the same data */
                else
                    csrc += m;
            }
            pdst = c.outp;
            pn = m;
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
        st = r.rc ? (u32)ST_END : st;

        /* ---- tables of the next lookup that depend on what was just decoded ---- */
        if (st == ST_LIT && ctx) { /* context_id(): yeast.c:698-755 */
            u32 cid;
            if (cur_mode == 0)
                cid = c.p1 & 0x3f;
            else if (cur_mode == 1)
                cid = c.p1 >> 2;
            else if (cur_mode == 2)
                cid = lut[c.p1] | lut[256 + c.p2];
            else
                cid = ((u32)lut[512 + c.p1] << 3) | lut[512 + c.p2];
            offL = (u32)wl->lt[cid * S + slot] << 4;
        } else if (st == ST_DIST && c.dist_codes > 1)
            offD = (u32)wl->dt[(copy > 4 ? 3 : copy - 2) * S + slot] << 4;
    }
    /* meta-block end (or failure): the piece set up last was valid */
    if (pn) {
        u64 pd0 = 0, pd1 = 0, pd2 = 0, pd3 = 0;
        if (psafe) {
            for (u32 b = 0; b < pn; b++)
                pd0 |= (u64)gld8(psrc + b) << (8 * b);
        } else {
            pd0 = gld64u(psrc);
            if (pn > 8 && !pmode)
                pd1 = gld64u(psrc + 8);
            if (pn > 16 && !pmode)
                pd2 = gld64u(psrc + 16);
            if (pn > 24 && !pmode)
                pd3 = gld64u(psrc + 24);
        }
        store_pending(pd0, pd1, pd2, pd3);
    }
    group_sync_mem();
    if (!ctx) { /* context for the next meta-block */
        u64 got = got_of(c);
        c.p1 = got >= 1 ? gld8(c.outp - 1) : 0;
        c.p2 = got >= 2 ? gld8(c.outp - 2) : 0;
    }
    /* hand the position back to the header reader */
    if (!r.rc)
        rd_seek(r, c.src, rd_pos(r));
}

#endif
