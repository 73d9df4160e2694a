/*
 * oracle/br_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the reference's .br container reader, broad()
 * (broad.c:142-353; format: br-format-v3.txt), on top of this directory's
 * restatement of the decoder and the check functions (yeast_oracle.c), plus
 * SHA-256 (FIPS 180-4; the reference takes it from OpenSSL: broad.c:14,266).
 * Only tests/ may load it.  Nothing under brotli_amd/ may include, link or call it.
 *
 * Pinning: tests/test_oracle_golden.py checks it against tests/golden/br_streams.*:
 * .br files written by the reference's own brand and braid and judged by the
 * reference's own broad() (tools/make_br_fixtures.py; oracle/Makefile target brref),
 * the two streams of br-format-v3.txt:416-429 among them.
 *
 * The structure is this repo's own: a cursor over a memory buffer with status
 * codes returned through one longjmp frame, one-shot checks over byte ranges
 * instead of running check states.
 */
#include <setjmp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int oracle_yeast(void **dest, size_t *got, const void *source, size_t *len, int cmp);
uint32_t oracle_crc32c(uint32_t crc, const void *buf, size_t len);
uint32_t oracle_xxh32(const void *input, size_t len, uint32_t seed);
uint64_t oracle_xxh64(const void *input, size_t len, uint64_t seed);

/* ---- SHA-256 (FIPS 180-4) ------------------------------------------------ */

static uint32_t ror(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }

void oracle_sha256(const void *data, size_t len, uint8_t out[32])
{
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98,
        0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786,
        0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8,
        0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
        0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819,
        0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a,
        0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7,
        0xc67178f2};
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    const uint8_t *p = (const uint8_t *)data;
    uint64_t bits = (uint64_t)len * 8;
    size_t blocks = (len + 9 + 63) / 64;
    for (size_t b = 0; b < blocks; b++) {
        uint8_t m[64];
        for (int i = 0; i < 64; i++) {
            size_t at = b * 64 + i;
            m[i] = at < len ? p[at] : at == len ? 0x80 : 0;
        }
        if (b == blocks - 1)
            for (int i = 0; i < 8; i++)
                m[56 + i] = (uint8_t)(bits >> (56 - 8 * i));
        uint32_t w[64];
        for (int i = 0; i < 16; i++)
            w[i] = (uint32_t)m[4 * i] << 24 | (uint32_t)m[4 * i + 1] << 16 | (uint32_t)m[4 * i + 2] << 8 | m[4 * i + 3];
        for (int i = 16; i < 64; i++)
            w[i] = w[i - 16] + (ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] +
                   (ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10));
        uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            uint32_t t1 = hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
        h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 4; j++)
            out[4 * i + j] = (uint8_t)(h[i] >> (24 - 8 * j));
}

/* ---- broad() --------------------------------------------------------------- */

typedef struct {
    const uint8_t *buf;
    size_t next, len;
    jmp_buf env;
} cur_t;

#define FAIL(s, code) longjmp((s)->env, (code))

/* get1(): broad.c:59-64 */
static unsigned get1(cur_t *s)
{
    if (s->next == s->len)
        FAIL(s, 2);
    return s->buf[s->next++];
}

/* skip(): broad.c:46-55 (the pointer goes to the end before the throw) */
static void skip(cur_t *s, uintmax_t n)
{
    size_t pass = s->len - s->next;
    if (pass > n)
        pass = (size_t)n;
    s->next += pass;
    if (pass < n)
        FAIL(s, 2);
}

/* getn(): broad.c:67-76 */
static uintmax_t getn(cur_t *s, unsigned n)
{
    uintmax_t val = 0;
    for (unsigned shift = 0; n; n--, shift += 8)
        val |= (uintmax_t)get1(s) << shift;
    return val;
}

/* getvar(): broad.c:79-89.  A run of more than nine continuation bytes shifts by 64 or more,
   which C leaves undefined; the reference as built here (x86-64) shifts modulo 64. */
static uintmax_t getvar(cur_t *s)
{
    uintmax_t val = 0;
    unsigned ch, shift = 0;
    do {
        ch = get1(s);
        val |= (uintmax_t)(ch & 0x7f) << (shift & 63);
        shift += 7;
    } while ((ch & 0x80) == 0);
    return val;
}

/* getbvar(): broad.c:92-104 */
static uintmax_t getbvar(cur_t *s)
{
    unsigned ch = get1(s);
    if ((ch & 0x80) == 0)
        FAIL(s, 3);
    uintmax_t val = ch & 0x7f;
    unsigned shift = 0;
    do {
        ch = get1(s);
        shift += 7;
        val |= (uintmax_t)(ch & 0x7f) << (shift & 63);
    } while ((ch & 0x80) == 0);
    return val;
}

/* parity(): broad.c:137-139 */
static unsigned parity(unsigned n) { return (0x34cb00 >> ((n ^ (n >> 4)) & 0xf)) & 0x80; }

/* the check named by the low three bits of a content mask over a byte string, cut to its stored
   width (broad.c:278-284 for data, :128-133 for the check of checks) */
static uintmax_t check_of(unsigned kind, const uint8_t *p, size_t n)
{
    kind &= 7;
    uintmax_t v = kind < 3 ? oracle_xxh32(p, n, 0) : kind == 3 ? oracle_xxh64(p, n, 0) : oracle_crc32c(0, p, n);
    unsigned width = 1u << (kind & 3);
    if (width < 4)
        v &= width == 2 ? 0xffff : 0xff;
    return v;
}

/* Counts the streams that ran out of input at the end of the file.  The reference passes yeast() the
   length of the whole file instead of what is left of it (broad.c:242), so there it reads on past the
   end of its buffer and what it returns is undefined; the fixture generator uses this count to mark
   such cases, and the tests do not hold the reference's verdict against them. */
int oracle_broad_overreads = 0;

/* Returns broad()'s code (0 ok, 2 premature end, 3 invalid format, 4 invalid compressed data,
   5 length or check mismatch; 1 out of memory).  *out receives what broad() would have written
   (malloc'd, possibly a prefix when the code is not 0), *segments the streams decoded. */
int oracle_broad(const void *br, size_t len, void **out, size_t *out_len, size_t *segments)
{
    cur_t *s = (cur_t *)malloc(sizeof(cur_t));
    uint8_t *volatile written = NULL;
    volatile size_t nwritten = 0, cap = 0, nseg = 0;
    uint8_t *volatile checks = NULL; /* the stored check values, in order (broad.c:294) */
    volatile size_t nchecks = 0, ccap = 0;
    void *volatile un = NULL;
    s->buf = (const uint8_t *)br;
    s->next = 0;
    s->len = len;
    int code = setjmp(s->env);
    if (code == 0) {
        if (getn(s, 4) != 0x81cfb2ce) /* broad.c:155 */
            FAIL(s, 3);
        uintmax_t total = 0;
        unsigned mask;
        size_t last, curr = 0;
        for (;;) {
            last = curr; /* broad.c:164-165 */
            curr = s->next;
            mask = get1(s);
            if (parity(mask))
                FAIL(s, 3);
            if (mask & 0x20) /* trailer */
                break;
            if (last == 0 && (mask & 0x10)) /* :174 */
                FAIL(s, 3);
            if (mask & 0x10)
                if (curr - last != getvar(s)) /* :179 */
                    FAIL(s, 3);
            if ((mask & 7) == 7)
                if (get1(s) != 0) /* :186-188: only SHA-256 is defined */
                    FAIL(s, 3);
            if (mask & 0x40) { /* extra mask :192-238 */
                unsigned extra = get1(s);
                if (parity(extra) || (extra & 0x18))
                    FAIL(s, 3);
                if (extra & 1)
                    (void)getvar(s);
                if (extra & 2)
                    skip(s, getvar(s));
                if (extra & 4)
                    skip(s, getvar(s));
                if (extra & 0x40) {
                    unsigned method = get1(s);
                    if (parity(method) || (method & 0x47))
                        FAIL(s, 3);
                }
                if (extra & 0x20) { /* header check over the header so far :231-234 */
                    unsigned chk = oracle_xxh32(s->buf + curr, s->next - curr, 0) & 0xffff;
                    if (chk != getn(s, 2))
                        FAIL(s, 3);
                }
            }
            /* decompress :241-249 */
            size_t got = 0, used = s->len - s->next;
            void *u = NULL;
            int ret = oracle_yeast(&u, &got, s->buf + s->next, &used, 0);
            un = u;
            s->next += used;
            total += got;
            nseg = nseg + 1;
            if (ret == 2)
                oracle_broad_overreads++;
            if (ret)
                FAIL(s, 4);
            if (mask & 8)
                if (got != getvar(s)) /* :256-259 */
                    FAIL(s, 5);
            unsigned n;
            if ((mask & 7) == 7) { /* SHA-256 :263-269 */
                n = 32;
                uint8_t sha[32];
                oracle_sha256(un, got, sha);
                skip(s, n);
                if (memcmp(sha, s->buf + s->next - n, n))
                    FAIL(s, 5);
            } else {
                n = 1u << (mask & 3);
                if (check_of(mask, (const uint8_t *)un, got) != getn(s, n)) /* :278-286 */
                    FAIL(s, 5);
            }
            if (nchecks + n > ccap) {
                ccap = (nchecks + n) * 2 + 64;
                uint8_t *c2 = (uint8_t *)realloc(checks, ccap);
                if (c2 == NULL)
                    FAIL(s, 1);
                checks = c2;
            }
            memcpy(checks + nchecks, s->buf + s->next - n, n);
            nchecks = nchecks + n;
            if (nwritten + got > cap) { /* fwrite :298 */
                cap = (nwritten + got) * 2 + 64;
                uint8_t *w2 = (uint8_t *)realloc(written, cap);
                if (w2 == NULL)
                    FAIL(s, 1);
                written = w2;
            }
            if (got)
                memcpy(written + nwritten, un, got);
            nwritten = nwritten + got;
            free(un);
            un = NULL;
        }
        /* trailer :310-341 */
        if (mask & 0x40)
            FAIL(s, 3);
        if (mask & 0x10)
            if (curr - last != getbvar(s))
                FAIL(s, 3);
        if (mask & 8)
            if (total != getbvar(s))
                FAIL(s, 5);
        if ((mask & 7) != 7) {
            uintmax_t chk = getn(s, 1u << (mask & 3));
            if (check_of(mask, checks, nchecks) != chk)
                FAIL(s, 5);
        }
        if ((mask & 7) != 7 || (mask & 0x18))
            if (get1(s) != mask)
                FAIL(s, 3);
    }
    free(un);
    free(checks);
    free(s);
    *out = written;
    *out_len = nwritten;
    if (segments)
        *segments = nseg;
    return code;
}
