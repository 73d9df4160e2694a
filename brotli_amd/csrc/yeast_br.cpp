/*
 * yeast_br.cpp -- the .br container reader of libyeast_mi355x.so: yeast_broad() of include/yeast_br.h.
 *
 * Reference counterpart: broad() broad.c:142-353 (format: br-format-v3.txt; constants br.h:6-39).
 * The reference walks the file front to back and calls yeast() once per segment, because only the
 * decoder knows where a brotli stream ends.  Here the walk is the same -- verify() below makes the
 * reference's checks in the reference's order -- but the decoding is taken out of it: when the file
 * carries reverse offsets (br-format-v3.txt:36-48) the headers are found from the end first (the
 * backward walk of braid.c:146-202), every stream is decoded in ONE batch launch (yeast_batch), and the
 * walk then finds each stream's result ready.  A stream the index did not predict, or one that ran
 * into the bound the index gave it, is decoded on the spot through yeast() with everything up to the
 * end of the file, so the verdict never depends on the index.
 *
 * No decode or check arithmetic runs on the host: streams go through yeast()/yeast_batch(), check
 * values through XXH32()/XXH64()/crc32c()/yeast_sha256_batch() of this library (GPU kernels).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <new>
#include <vector>

#include "../../include/yeast.h"
#include "../../include/yeast_batch.h"
#include "../../include/yeast_br.h"

namespace {

/* br.h:6-39 */
enum {
    CONTENT_CHECK = 7, CONTENT_LEN = 8, CONTENT_OFF = 0x10, CONTENT_TRAIL = 0x20, CONTENT_EXTRA = 0x40,
    EXTRA_MOD = 1, EXTRA_NAME = 2, EXTRA_EXTRA = 4, EXTRA_RESERVED = 0x18, EXTRA_CHECK = 0x20, EXTRA_METHOD = 0x40,
    METHOD_MUST_BE_ZERO = 0x47
};

struct Fail {
    int code;
};

/* parity(): broad.c:137-139 */
unsigned parity(unsigned n) { return (0x34cb00 >> ((n ^ (n >> 4)) & 0xf)) & 0x80; }

struct Cursor {
    const uint8_t *buf;
    size_t next, len;
    unsigned get1() /* broad.c:59-64 */
    {
        if (next == len)
            throw Fail{2};
        return buf[next++];
    }
    void skip(uintmax_t n) /* broad.c:46-55 */
    {
        size_t pass = len - next;
        if (pass > n)
            pass = (size_t)n;
        next += pass;
        if (pass < n)
            throw Fail{2};
    }
    uintmax_t getn(unsigned n) /* broad.c:67-76 */
    {
        uintmax_t v = 0;
        for (unsigned sh = 0; n; n--, sh += 8)
            v |= (uintmax_t)get1() << sh;
        return v;
    }
    uintmax_t getvar() /* broad.c:79-89; shifts of 64 and more wrap as they do in the reference's x86-64 build */
    {
        uintmax_t v = 0;
        unsigned ch, sh = 0;
        do {
            ch = get1();
            v |= (uintmax_t)(ch & 0x7f) << (sh & 63);
            sh += 7;
        } while ((ch & 0x80) == 0);
        return v;
    }
    uintmax_t getbvar() /* broad.c:92-104 */
    {
        unsigned ch = get1();
        if ((ch & 0x80) == 0)
            throw Fail{3};
        uintmax_t v = ch & 0x7f;
        unsigned sh = 0;
        do {
            ch = get1();
            sh += 7;
            v |= (uintmax_t)(ch & 0x7f) << (sh & 63);
        } while ((ch & 0x80) == 0);
        return v;
    }
};

/* the check named by the low three bits of a content mask, cut to its stored width
   (broad.c:278-284 for a segment, :128-133 for the check of checks) */
uintmax_t cut_check(unsigned kind, uintmax_t v)
{
    unsigned width = 1u << (kind & 3);
    if ((kind & 7) != 3 && width < 4)
        v &= width == 2 ? 0xffff : 0xff;
    return v;
}
uintmax_t check_of(unsigned kind, const void *p, size_t n)
{
    kind &= 7;
    uintmax_t v = kind < 3 ? XXH32(p, n, 0) : kind == 3 ? XXH64(p, n, 0) : crc32c(0, p, n);
    return cut_check(kind, v);
}
unsigned batch_kind(unsigned mask) /* content mask check -> yeast_batch check kind */
{
    unsigned k = mask & 7;
    return k < 3 ? YEAST_CHECK_XXH32 : k == 3 ? YEAST_CHECK_XXH64 : k < 7 ? YEAST_CHECK_CRC32C : YEAST_CHECK_NONE;
}

/* One header, from its content mask to the first byte of the brotli stream (broad.c:169-238).
   first: no header precedes it.  want_off: the reverse offset it must hold. */
struct Header {
    unsigned mask;
    size_t stream; /* offset of the brotli stream */
};
Header read_header(Cursor &c, bool first, size_t want_off)
{
    Header h;
    size_t start = c.next;
    h.mask = c.get1();
    if (parity(h.mask))
        throw Fail{3};
    if (h.mask & CONTENT_TRAIL) {
        h.stream = 0;
        return h;
    }
    if (first && (h.mask & CONTENT_OFF))
        throw Fail{3};
    if (h.mask & CONTENT_OFF)
        if (want_off != c.getvar())
            throw Fail{3};
    if ((h.mask & CONTENT_CHECK) == 7)
        if (c.get1() != 0) /* only SHA-256 is defined: broad.c:186-188 */
            throw Fail{3};
    if (h.mask & CONTENT_EXTRA) {
        unsigned extra = c.get1();
        if (parity(extra) || (extra & EXTRA_RESERVED))
            throw Fail{3};
        if (extra & EXTRA_MOD)
            (void)c.getvar();
        if (extra & EXTRA_NAME)
            c.skip(c.getvar());
        if (extra & EXTRA_EXTRA)
            c.skip(c.getvar());
        if (extra & EXTRA_METHOD) {
            unsigned method = c.get1();
            if (parity(method) || (method & METHOD_MUST_BE_ZERO))
                throw Fail{3};
        }
        if (extra & EXTRA_CHECK) { /* XXH32 of the header so far, low two bytes: broad.c:231-234 */
            unsigned chk = XXH32(c.buf + start, c.next - start, 0) & 0xffff;
            if (chk != c.getn(2))
                throw Fail{3};
        }
    }
    h.stream = c.next;
    return h;
}

/* The offsets of the headers, found from the end of the file without decoding anything: the walk of
   braid.c:146-202.  False when the file does not carry a complete chain of reverse offsets (or is
   damaged): then nothing is predicted and every stream is decoded when the forward walk reaches it. */
bool index_backward(const uint8_t *b, size_t len, std::vector<size_t> &headers, size_t &trailer)
{
    if (len < 5 || memcmp(b, "\xce\xb2\xcf\x81", 4))
        return false;
    size_t at = len;
    auto rget1 = [&](unsigned &ch) -> bool {
        if (at <= 4)
            return false;
        ch = b[--at];
        return true;
    };
    auto rbvar = [&](uintmax_t &v) -> bool { /* braid.c:44-54 */
        unsigned ch;
        if (!rget1(ch) || (ch & 0x80) == 0)
            return false;
        v = ch & 0x7f;
        unsigned n = 0;
        do {
            if (!rget1(ch) || ++n > 9)
                return false;
            v = (v << 7) | (ch & 0x7f);
        } while ((ch & 0x80) == 0);
        return true;
    };
    unsigned trail;
    do {
        if (!rget1(trail))
            return false;
    } while (trail == 0); /* trailing zeros */
    if (parity(trail) || !(trail & CONTENT_TRAIL) || (trail & CONTENT_EXTRA))
        return false;
    if ((trail & CONTENT_CHECK) != 7) {
        size_t n = (size_t)1 << (trail & 3);
        if (at < 4 + n)
            return false;
        at -= n;
    }
    uintmax_t v, dist = 0;
    if ((trail & CONTENT_LEN) && !rbvar(v))
        return false;
    if ((trail & CONTENT_OFF) && !rbvar(dist))
        return false;
    if (trail != (CONTENT_TRAIL | 7)) {
        unsigned lead;
        if (!rget1(lead) || lead != trail)
            return false;
    }
    trailer = at;
    headers.clear();
    if (at > 4 && !(trail & CONTENT_OFF))
        return false;
    std::vector<size_t> rev;
    while (dist) {
        if (dist > at - 4)
            return false;
        at -= (size_t)dist;
        rev.push_back(at);
        if (at == 4)
            break;
        Cursor c{b, at, len};
        try {
            unsigned mask = c.get1();
            if (parity(mask) || (mask & CONTENT_TRAIL) || !(mask & CONTENT_OFF))
                return false;
            dist = c.getvar();
        } catch (Fail &) {
            return false;
        }
    }
    if (rev.empty() || rev.back() != 4)
        return false;
    headers.assign(rev.rbegin(), rev.rend());
    return true;
}

struct Decoded {
    int rc;
    void *data;
    size_t got, used;
    uint64_t check; /* of the batch kind that belongs to the header's check (none for SHA-256) */
    bool have_sha;
    uint8_t sha[32];
};

} // namespace

extern "C" int yeast_broad(const void *br, size_t len, void **out, size_t *out_len, int write, size_t *segments)
{
    const uint8_t *b = (const uint8_t *)br;
    if (out)
        *out = nullptr;
    if (out_len)
        *out_len = 0;
    if (segments)
        *segments = 0;

    /* ---- decode ahead: every stream the reverse offsets lead to, in one batch per check kind ---- */
    std::map<size_t, Decoded> ready; /* by offset of the brotli stream */
    auto drop_ready = [&]() {
        for (auto &kv : ready)
            free(kv.second.data);
        ready.clear();
    };
    {
        std::vector<size_t> headers;
        size_t trailer = 0;
        struct Plan {
            size_t stream, bound;
            unsigned mask;
        };
        std::vector<Plan> plan;
        if (index_backward(b, len, headers, trailer)) {
            try {
                for (size_t k = 0; k < headers.size(); k++) {
                    Cursor c{b, headers[k], len};
                    Header h = read_header(c, k == 0, k ? headers[k] - headers[k - 1] : 0);
                    size_t bound = k + 1 < headers.size() ? headers[k + 1] : trailer;
                    if ((h.mask & CONTENT_TRAIL) || h.stream > bound) {
                        plan.clear();
                        break;
                    }
                    plan.push_back(Plan{h.stream, bound, h.mask});
                }
            } catch (Fail &) {
                plan.clear(); /* the forward walk will find out what is wrong, in its own order */
            }
        }
        for (unsigned kind = 0; kind < 4 && !plan.empty(); kind++) {
            std::vector<size_t> pick;
            for (size_t k = 0; k < plan.size(); k++)
                if (batch_kind(plan[k].mask) == kind)
                    pick.push_back(k);
            if (pick.empty())
                continue;
            size_t n = pick.size();
            std::vector<const void *> src(n);
            std::vector<size_t> ln(n), got(n);
            std::vector<void *> dest(n, nullptr);
            std::vector<int> rc(n);
            std::vector<uint64_t> chk(n);
            for (size_t i = 0; i < n; i++) {
                src[i] = b + plan[pick[i]].stream;
                ln[i] = plan[pick[i]].bound - plan[pick[i]].stream;
            }
            int e = yeast_batch(n, src.data(), ln.data(), dest.data(), got.data(), rc.data(), kind, chk.data());
            if (e) {
                for (void *p : dest)
                    free(p);
                drop_ready();
                return e;
            }
            std::vector<const void *> sbuf;
            std::vector<size_t> slen, sidx;
            for (size_t i = 0; i < n; i++) {
                if (rc[i] == 2) { /* ran into the bound the index gave it: decided by the forward walk */
                    free(dest[i]);
                    continue;
                }
                Decoded d;
                d.rc = rc[i];
                d.data = dest[i];
                d.got = got[i];
                d.used = ln[i];
                d.check = chk[i];
                d.have_sha = false;
                ready[plan[pick[i]].stream] = d;
                if (kind == YEAST_CHECK_NONE && rc[i] == 0) {
                    sbuf.push_back(dest[i]);
                    slen.push_back(got[i]);
                    sidx.push_back(plan[pick[i]].stream);
                }
            }
            if (!sbuf.empty()) { /* SHA-256 of all such segments in one launch */
                std::vector<uint8_t> dig(32 * sbuf.size());
                if (yeast_sha256_batch(sbuf.size(), sbuf.data(), slen.data(), dig.data())) {
                    drop_ready();
                    return -1;
                }
                for (size_t i = 0; i < sidx.size(); i++) {
                    Decoded &d = ready[sidx[i]];
                    d.have_sha = true;
                    memcpy(d.sha, dig.data() + 32 * i, 32);
                }
            }
        }
    }

    /* ---- the reference's walk: broad.c:154-341 ---- */
    uint8_t *written = nullptr;
    size_t nwritten = 0, cap = 0, nseg = 0;
    std::vector<uint8_t> checks; /* the stored check values in order: broad.c:294 */
    void *un = nullptr;
    int code = 0;
    Cursor c{b, 0, len};
    try {
        if (c.getn(4) != 0x81cfb2ce)
            throw Fail{3};
        uintmax_t total = 0;
        size_t last, curr = 0;
        Header h;
        for (;;) {
            last = curr;
            curr = c.next;
            h = read_header(c, last == 0, curr - last);
            if (h.mask & CONTENT_TRAIL)
                break;
            /* the stream: decoded ahead, or now */
            Decoded d;
            auto it = ready.find(c.next);
            if (it != ready.end()) {
                d = it->second;
                ready.erase(it);
            } else {
                d.data = nullptr;
                d.got = 0;
                d.used = len - c.next;
                d.have_sha = false;
                d.check = 0;
                d.rc = yeast(&d.data, &d.got, b + c.next, &d.used, 0);
                if (d.rc < 0) {
                    free(d.data);
                    throw Fail{-1};
                }
                if (d.rc == 0 && (h.mask & CONTENT_CHECK) != 7)
                    d.check = check_of(h.mask, d.data, d.got); /* uncut: cut below */
            }
            un = d.data;
            c.next += d.used;
            total += d.got;
            nseg++;
            if (d.rc)
                throw Fail{4}; /* broad.c:248-249: whatever yeast() returned */
            if (h.mask & CONTENT_LEN)
                if (d.got != c.getvar()) /* :256-259 */
                    throw Fail{5};
            unsigned n;
            if ((h.mask & CONTENT_CHECK) == 7) { /* SHA-256: :263-269 */
                n = 32;
                if (!d.have_sha && yeast_sha256(un, d.got, d.sha))
                    throw Fail{-1};
                c.skip(n);
                if (memcmp(d.sha, b + c.next - n, n))
                    throw Fail{5};
            } else {
                n = 1u << (h.mask & 3);
                if (cut_check(h.mask & 7, d.check) != c.getn(n)) /* :278-286 */
                    throw Fail{5};
            }
            checks.insert(checks.end(), b + c.next - n, b + c.next);
            if (write && d.got) { /* :297-301 */
                if (nwritten + d.got > cap) {
                    size_t want = (nwritten + d.got) * 2 + 64;
                    uint8_t *w2 = (uint8_t *)realloc(written, want);
                    if (!w2)
                        throw Fail{1};
                    written = w2;
                    cap = want;
                }
                memcpy(written + nwritten, un, d.got);
                nwritten += d.got;
            }
            free(un);
            un = nullptr;
        }
        /* trailer: :310-341 */
        unsigned mask = h.mask;
        if (mask & CONTENT_EXTRA)
            throw Fail{3};
        if (mask & CONTENT_OFF)
            if (curr - last != c.getbvar())
                throw Fail{3};
        if (mask & CONTENT_LEN)
            if (total != c.getbvar())
                throw Fail{5};
        if ((mask & CONTENT_CHECK) != 7) {
            uintmax_t chk = c.getn(1u << (mask & 3));
            if (check_of(mask, checks.data(), checks.size()) != chk)
                throw Fail{5};
        }
        if ((mask & CONTENT_CHECK) != 7 || (mask & (CONTENT_LEN | CONTENT_OFF)))
            if (c.get1() != mask)
                throw Fail{3};
    } catch (Fail &f) {
        code = f.code;
    } catch (std::bad_alloc &) {
        code = 1;
    }
    free(un);
    drop_ready();
    if (out && write && nwritten)
        *out = written;
    else
        free(written);
    if (out_len)
        *out_len = write ? nwritten : 0;
    if (segments)
        *segments = nseg;
    return code;
}
