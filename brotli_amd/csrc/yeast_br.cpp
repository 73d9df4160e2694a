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
#include <thread>
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
        if (next >= len)
            throw Fail{2};
        return buf[next++];
    }
    void skip(uintmax_t n) /* broad.c:46-55 */
    {
        size_t pass = next < len ? len - next : 0;
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
/* The check functions keep the reference's signatures and so have no error value of their own: a call that could
   not run on the GPU returns 0 and is counted (yeast_check_error()).  Here every use asks, and a value that was
   not computed ends the call with -1 instead of being compared or written. */
uintmax_t check_value(unsigned kind, const void *p, size_t n) /* kind: low three bits of a content mask */
{
    (void)yeast_check_error();
    uintmax_t v = kind < 3 ? XXH32(p, n, 0) : kind == 3 ? XXH64(p, n, 0) : crc32c(0, p, n);
    if (yeast_check_error())
        throw Fail{-1};
    return v;
}
unsigned xxh32_of(const void *p, size_t n) { return (unsigned)check_value(2, p, n); }
uintmax_t check_of(unsigned kind, const void *p, size_t n)
{
    kind &= 7;
    return cut_check(kind, check_value(kind, p, n));
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
            unsigned chk = xxh32_of(c.buf + start, c.next - start) & 0xffff;
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
    unsigned kind; /* the batch kind it was decoded ahead with (the header the backward index parsed) */
    bool fetched;  /* data holds the decoded bytes (else they stayed on the device) */
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
            /* verifying only (broad -t): the decoded bytes stay on the device unless SHA-256 needs them */
            const bool fetch = write || kind == YEAST_CHECK_NONE;
            int e = yeast_batch(n, src.data(), ln.data(), fetch ? dest.data() : nullptr, got.data(), rc.data(), kind,
                                chk.data());
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
                d.kind = kind;
                d.fetched = fetch;
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
    size_t nwritten = 0, nseg = 0;
    struct Piece {
        void *data;
        size_t got;
    };
    std::vector<Piece> pieces; /* the verified segments in order: put together when the walk is over */
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
            bool ahead = false;
            if (it != ready.end()) {
                /* only for the header it was decoded for: the walk may reach the same stream offset through
                   another header (one that begins inside the indexed header's name or extra field), whose
                   check is of another kind; then the stream is decoded here like any that was not indexed */
                d = it->second;
                ready.erase(it);
                ahead = d.kind == batch_kind(h.mask) && (d.fetched || !write);
                if (!ahead)
                    free(d.data);
            }
            if (!ahead) {
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
                pieces.push_back(Piece{un, d.got});
                nwritten += d.got;
                un = nullptr;
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
    /* what broad() has written by now, as one buffer: allocated once at its final size and filled by a few
       host threads (most of the cost is the first touch of its pages) */
    if (out && write && nwritten) {
        written = (uint8_t *)malloc(nwritten);
        if (written) {
            std::vector<size_t> at(pieces.size() + 1, 0);
            for (size_t k = 0; k < pieces.size(); k++)
                at[k + 1] = at[k] + pieces[k].got;
            auto fill = [&](size_t k0, size_t k1) {
                for (size_t k = k0; k < k1; k++)
                    memcpy(written + at[k], pieces[k].data, pieces[k].got);
            };
            unsigned t = std::thread::hardware_concurrency();
            t = t > 8 ? 8 : t;
            if (nwritten < ((size_t)8 << 20) || pieces.size() < 64 || t < 2)
                fill(0, pieces.size());
            else {
                std::vector<std::thread> th;
                unsigned started = 0;
                try {
                    for (; started < t; started++)
                        th.emplace_back(fill, pieces.size() * started / t, pieces.size() * (started + 1) / t);
                } catch (...) { /* no more threads to be had: this one does the rest */
                }
                for (unsigned k = started; k < t; k++)
                    fill(pieces.size() * k / t, pieces.size() * (k + 1) / t);
                for (std::thread &x : th)
                    x.join();
            }
            *out = written;
        } else {
            code = 1;
            nwritten = 0;
        }
    }
    for (Piece &q : pieces)
        free(q.data);
    if (out_len)
        *out_len = write ? nwritten : 0;
    if (segments)
        *segments = nseg;
    return code;
}

/* ======================================================================== writer and merger
 * yeast_brand(): brand.c:93-263 (wrap() and main()); yeast_braid(): braid.c:146-422.  Host work is
 * byte shuffling only: the stream is decoded and its check value computed on the GPU (yeast_batch /
 * yeast_sha256), header checks and the check of checks through XXH32() of this library.
 */
namespace {

void put_var(std::vector<uint8_t> &o, uintmax_t n) /* brand.c:67-76, braid.c:218-225 */
{
    while (n > 0x7f) {
        o.push_back((uint8_t)(n & 0x7f));
        n >>= 7;
    }
    o.push_back((uint8_t)(0x80 | n));
}
void put_bvar(std::vector<uint8_t> &o, uintmax_t n) /* brand.c:54-63, braid.c:94-99 */
{
    o.push_back((uint8_t)(0x80 | (n & 0x7f)));
    while ((n >>= 7) > 0x7f)
        o.push_back((uint8_t)(n & 0x7f));
    o.push_back((uint8_t)(0x80 | n));
}
void put_little(std::vector<uint8_t> &o, uintmax_t v, size_t k) /* brand.c:44-49 */
{
    do {
        o.push_back((uint8_t)v);
        v >>= 8;
    } while (--k);
}
int hand_over(const std::vector<uint8_t> &o, void **out, size_t *out_len)
{
    *out = nullptr;
    *out_len = o.size();
    if (o.empty())
        return 0;
    *out = malloc(o.size());
    if (!*out)
        return 1;
    memcpy(*out, o.data(), o.size());
    return 0;
}

} // namespace

#include <time.h>

extern "C" int yeast_brand(const void *brotli, size_t len, const char *opt, const char *name, void **out,
                           size_t *out_len)
{
    *out = nullptr;
    *out_len = 0;
    if (!opt)
        opt = "";
    if (!name)
        name = "filename"; /* brand.c:308 */
    /* options: brand.c:96-182 */
    enum { xxh32, xxh64, crc32, sha256 } type = xxh64;
    size_t check_len = 8;
    int tail = CONTENT_LEN | CONTENT_OFF, mod = 0, file = 0;
    for (; *opt; opt++)
        switch (*opt) {
        case '1': case '2': case '4': case '8':
            check_len = (size_t)(*opt - '0');
            if (type == xxh32 && check_len == 8)
                type = xxh64;
            else if (type == xxh64 && check_len < 8)
                type = xxh32;
            else if (type == crc32 && check_len == 8)
                check_len = 4;
            else if (type == sha256)
                check_len = 32;
            break;
        case 'c':
            if (check_len > 4)
                check_len = 4;
            type = crc32;
            break;
        case 's':
            type = sha256;
            check_len = 32;
            break;
        case 'x':
            if (check_len > 8)
                check_len = 8;
            type = check_len < 8 ? xxh32 : xxh64;
            break;
        case 'n': tail = 0; break;
        case 'u': tail = CONTENT_LEN; break;
        case 'r': tail = CONTENT_OFF; break;
        case 'b': tail = CONTENT_LEN | CONTENT_OFF; break;
        case 'm': mod = 1; break;
        case 'f': file = 1; break;
        }
    /* decode, with the check of the chosen type (brand.c:296-304, 238-250) */
    const void *src[1] = {brotli};
    size_t ln[1] = {len}, got[1] = {0};
    void *dest[1] = {nullptr};
    int rc[1] = {0};
    uint64_t chk[1] = {0};
    unsigned kind = type == xxh64 ? YEAST_CHECK_XXH64 : type == xxh32 ? YEAST_CHECK_XXH32
                    : type == crc32 ? YEAST_CHECK_CRC32C : YEAST_CHECK_NONE;
    int e = yeast_batch(1, src, ln, type == sha256 ? dest : nullptr, got, rc, kind, chk);
    if (e < 0)
        return e;
    if (e || rc[0] || ln[0] != len) {
        free(dest[0]);
        return 1; /* "error decompressing stream -- aborting" */
    }
    uint8_t sha[32];
    if (type == sha256) {
        int se = yeast_sha256(dest[0], got[0], sha);
        free(dest[0]);
        if (se)
            return -1;
    }
    std::vector<uint8_t> o;
    o.insert(o.end(), {0xce, 0xb2, 0xcf, 0x81});
    size_t start = o.size();
    unsigned mask = type == xxh64 ? 3u : type == xxh32 ? (check_len == 4 ? 2u : check_len == 2 ? 1u : 0u)
                    : type == crc32 ? (check_len == 4 ? 6u : check_len == 2 ? 5u : 4u) : 7u;
    mask |= CONTENT_EXTRA; /* brand.c:207: the name is never null there */
    o.push_back((uint8_t)(mask ^ parity(mask)));
    if ((mask & 7) == 7)
        o.push_back(0);
    {
        unsigned extra = (mod ? EXTRA_MOD : 0) | (file ? EXTRA_NAME : 0);
        o.push_back((uint8_t)(extra ^ parity(extra)));
        if (mod)
            put_var(o, ((uintmax_t)time(nullptr) + 35) << 1);
        if (file) {
            size_t nl = strlen(name);
            put_var(o, nl);
            o.insert(o.end(), (const uint8_t *)name, (const uint8_t *)name + nl);
        }
    }
    o.insert(o.end(), (const uint8_t *)brotli, (const uint8_t *)brotli + len);
    if (type == sha256)
        o.insert(o.end(), sha, sha + 32);
    else
        put_little(o, chk[0], check_len);
    size_t writ = o.size() - start;
    unsigned t = (unsigned)tail | 7 | CONTENT_TRAIL;
    t ^= parity(t);
    o.push_back((uint8_t)t);
    if (t & CONTENT_OFF)
        put_bvar(o, writ);
    if (t & CONTENT_LEN)
        put_bvar(o, got[0]);
    if (t & (CONTENT_OFF | CONTENT_LEN))
        o.push_back((uint8_t)t);
    return hand_over(o, out, out_len);
}

namespace {

/* scan(): braid.c:146-202 over a buffer.  The offsets of the headers and of the trailer, front to
   back; throws the reference's codes. */
std::vector<size_t> braid_scan(const uint8_t *b, size_t len)
{
    if (len < 4 || memcmp(b, "\xce\xb2\xcf\x81", 4))
        throw Fail{2};
    int64_t at = (int64_t)len;
    auto rget1 = [&]() -> unsigned { /* braid.c:33-41 */
        if (at == 0)
            throw Fail{1};
        return b[--at];
    };
    auto rbvar = [&]() -> uintmax_t { /* braid.c:44-54 */
        unsigned ch = rget1();
        if ((ch & 0x80) == 0)
            throw Fail{3};
        uintmax_t v = ch & 0x7f;
        do {
            ch = rget1();
            v = (v << 7) | (ch & 0x7f);
        } while ((ch & 0x80) == 0);
        return v;
    };
    unsigned trail;
    while ((trail = rget1()) == 0)
        ;
    if (parity(trail) || !(trail & CONTENT_TRAIL) || (trail & CONTENT_EXTRA))
        throw Fail{3};
    if ((trail & CONTENT_CHECK) != 7) {
        int64_t n = (int64_t)1 << (trail & 3);
        if (at < n)
            throw Fail{1};
        at -= n;
    }
    if (trail & CONTENT_LEN)
        (void)rbvar();
    uintmax_t dist = 0;
    if (trail & CONTENT_OFF)
        dist = rbvar();
    if (trail != (CONTENT_TRAIL | 7))
        if (rget1() != trail)
            throw Fail{3};
    if (at > 4 && !(trail & CONTENT_OFF))
        throw Fail{4};
    /* The reference keeps offsets in an off_t and subtracts whatever the file says (braid.c:182, :195):
       a distance of 2^63 and more moves the offset forward, one beyond the offset makes it negative.
       The same arithmetic here, on unsigned values so that it is defined; an offset is only ever used
       as a position in the buffer when it lies inside it. */
    auto back_by = [&](uintmax_t d) { at = (int64_t)((uint64_t)at - (uint64_t)d); };
    std::vector<size_t> rev;
    rev.push_back((size_t)at);
    if (dist) {
        back_by(dist);
        if (at >= 0 && (uint64_t)at <= len)
            rev.push_back((size_t)at);
    }
    while (at > 4) {
        if ((uint64_t)at >= len)
            throw Fail{1}; /* the reference seeks there and reads end of file: braid.c:189-190, :60 */
        Cursor c{b, (size_t)at, len};
        unsigned mask;
        uintmax_t back;
        try {
            mask = c.get1();
            if (parity(mask) || (mask & CONTENT_TRAIL))
                throw Fail{3};
            if (!(mask & CONTENT_OFF))
                throw Fail{4};
            back = c.getvar();
        } catch (Fail &f) {
            throw Fail{f.code == 2 ? 1 : f.code}; /* "premature end of file" is 1 in braid.c:60 */
        }
        back_by(back);
        if (at >= 0 && (uint64_t)at <= len)
            rev.push_back((size_t)at);
    }
    if (at != 4)
        throw Fail{3};
    return std::vector<size_t>(rev.rbegin(), rev.rend());
}

} // namespace

extern "C" int yeast_braid(size_t n, void const *const *br, const size_t *len, void **out, size_t *out_len)
{
    *out = nullptr;
    *out_len = 0;
    int ret = 0;
    unsigned count = 0;       /* brotli streams, up to 2: braid.c:343 */
    intmax_t total = 0;       /* total uncompressed length or -1 */
    std::vector<uint8_t> o, checks; /* checks: every check value copied, in order (braid.c:337) */
    o.insert(o.end(), {0xce, 0xb2, 0xcf, 0x81});
    size_t last = 0;
    for (size_t f = 0; f < n; f++) {
        const uint8_t *b = (const uint8_t *)br[f];
        const size_t flen = len[f];
        size_t keep_o = o.size(), keep_c = checks.size(), keep_last = last;
        unsigned keep_count = count;
        try {
            std::vector<size_t> pos = braid_scan(b, flen);
            bool any = pos.size() > 1;
            for (size_t k = 0; k + 1 < pos.size(); k++) { /* copy(): braid.c:252-339 */
                Cursor c{b, pos[k], flen};
                auto getc_ = [&]() -> unsigned { return c.next < c.len ? c.buf[c.next++] : 0xffu; }; /* EOF reads as -1 */
                unsigned mask = getc_();
                if (mask & CONTENT_OFF)
                    (void)c.getvar();
                unsigned id = 0;
                if ((mask & CONTENT_CHECK) == 7)
                    id = getc_();
                unsigned extra = 0, strip = 0;
                bool head = true;
                if (mask & CONTENT_EXTRA) {
                    strip = extra = getc_();
                    if (last)
                        strip &= ~(unsigned)(EXTRA_MOD | EXTRA_NAME);
                    if (!(extra & EXTRA_CHECK))
                        head = false;
                }
                size_t here = o.size();
                if (last)
                    mask |= CONTENT_OFF;
                unsigned newmask = mask;
                if ((strip & 0x67) == 0)
                    newmask &= ~(unsigned)CONTENT_EXTRA;
                o.push_back((uint8_t)(newmask ^ parity(newmask)));
                if (last)
                    put_var(o, here - last);
                last = here;
                if ((mask & CONTENT_CHECK) == 7)
                    o.push_back((uint8_t)id);
                if (mask & CONTENT_EXTRA) {
                    if (strip & 0x67)
                        o.push_back((uint8_t)(strip ^ parity(strip)));
                    if (extra & EXTRA_MOD) {
                        uintmax_t m = c.getvar();
                        if (strip & EXTRA_MOD)
                            put_var(o, m);
                    }
                    if (extra & EXTRA_NAME) {
                        uintmax_t nl = c.getvar();
                        size_t from = c.next;
                        c.skip(nl);
                        if (strip & EXTRA_NAME) {
                            put_var(o, nl);
                            o.insert(o.end(), b + from, b + c.next);
                        }
                    }
                    if (extra & EXTRA_EXTRA) {
                        uintmax_t el = c.getvar();
                        size_t from = c.next;
                        c.skip(el);
                        put_var(o, el);
                        o.insert(o.end(), b + from, b + c.next);
                    }
                    if (extra & EXTRA_METHOD)
                        o.push_back((uint8_t)getc_());
                    if (extra & EXTRA_CHECK) {
                        (void)getc_();
                        (void)getc_();
                        unsigned x = head ? xxh32_of(o.data() + here, o.size() - here) & 0xffff : 0;
                        o.push_back((uint8_t)(x & 0xff));
                        o.push_back((uint8_t)(x >> 8));
                    }
                }
                size_t cn = (mask & CONTENT_CHECK) == 7 ? 32 : (size_t)1 << (mask & 3);
                if (pos[k + 1] < c.next + cn || pos[k + 1] > flen)
                    throw Fail{1};
                o.insert(o.end(), b + c.next, b + pos[k + 1]);
                checks.insert(checks.end(), b + pos[k + 1] - cn, b + pos[k + 1]);
                if (count < 2)
                    count++;
            }
            if (total != -1) { /* braid.c:372-381 */
                Cursor c{b, pos.back(), flen};
                unsigned trail = c.get1();
                if (trail & CONTENT_LEN) {
                    if (trail & CONTENT_OFF)
                        (void)c.getbvar();
                    total += (intmax_t)c.getbvar();
                } else if (any)
                    total = -1;
            }
        } catch (Fail &f) { /* "-- skipping": braid.c:392-396 */
            if (f.code < 0) /* a header check could not be computed: no output rather than a wrong one */
                return -1;
            o.resize(keep_o);
            checks.resize(keep_c);
            last = keep_last;
            count = keep_count;
            ret = 1;
        }
    }
    /* trailer: braid.c:399-416 */
    unsigned trail = CONTENT_TRAIL | (count > 1 ? 2u : 7u);
    if (count == 0)
        total = -1;
    if (total != -1)
        trail |= CONTENT_LEN;
    if (last)
        trail |= CONTENT_OFF;
    trail ^= parity(trail);
    size_t off = o.size();
    o.push_back((uint8_t)trail);
    if (last)
        put_bvar(o, off - last);
    if (total != -1)
        put_bvar(o, (uintmax_t)total);
    if (count > 1) {
        try {
            put_little(o, xxh32_of(checks.data(), checks.size()), 4);
        } catch (Fail &) {
            return -1;
        }
    }
    if (trail != (CONTENT_TRAIL | 7))
        o.push_back((uint8_t)trail);
    if (hand_over(o, out, out_len))
        return 1;
    return ret;
}
