/*
 * include/yeast_br.h -- the .br container reader of libyeast_mi355x.so.
 *
 * Replaces the reference's broad() (broad.c:142-353: `int broad(FILE *in, FILE *out, int verbose,
 * int write)`, format br-format-v3.txt) for callers that hold the file in memory: the same checks in
 * the same order with the same codes, but the brotli streams of the file are decoded on the GPU --
 * all of them in one batch launch when the file carries the reverse offsets that let its segments be
 * found without decoding (br-format-v3.txt:36-48; the backward walk of braid.c:146-202), one after
 * the other through yeast() when it does not.  Check values (XXH32, XXH64, CRC-32C, SHA-256) are
 * computed on the GPU as well.
 */
#ifndef YEAST_BR_H
#define YEAST_BR_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Unwrap and verify the .br file br[0 .. len).
 *   *out, *out_len   what broad() writes to `out`: the decoded data of every segment that passed its
 *                    checks, in order (a malloc() buffer the caller frees, NULL when empty) -- also
 *                    when the return value is not 0: broad() has written the earlier segments by then.
 *                    With write == 0 (broad's -t) nothing is returned, only verified.
 *   *segments        brotli streams decoded (may be NULL)
 * Returns broad()'s codes (broad.c:152-341): 0 ok, 1 out of memory, 2 premature end of the file,
 * 3 invalid format (signature, parity, reserved bits, reverse offsets, header check, trailer),
 * 4 invalid compressed data, 5 length or check value mismatch; -1 when the GPU path is unavailable.
 * One difference, in the reference's favour of being defined: a brotli stream is bounded by the end of
 * the file here; broad.c:242 passes yeast() the length of the whole file and reads past its buffer when
 * the last stream is cut short.
 */
int yeast_broad(const void *br, size_t len, void **out, size_t *out_len, int write, size_t *segments);

/*
 * Wrap one brotli stream into a .br file: the reference's brand program (brand.c:93-314; wrap() and
 * main()).  opt is brand's option string (brand.c:79-92: 1 2 4 8 x c s n u r b f m; NULL = ""), name
 * the file name stored with 'f' (NULL = "filename", brand.c:308).  The stream is decoded on the GPU to
 * get its length and check value.  *out is a malloc() buffer.  Returns 0, 1 when the stream does not
 * decode to its end exactly (brand.c:300-304) or memory ran out, -1 when the GPU path is unavailable.
 */
int yeast_brand(const void *brotli, size_t len, const char *opt, const char *name, void **out, size_t *out_len);

/*
 * Merge n .br files into one without decoding them: the reference's braid program (braid.c:146-422).
 * Files without a complete chain of reverse offsets are skipped (braid.c:392-396) and make the
 * return value 1, as braid's exit status; otherwise 0.  *out is a malloc() buffer.
 */
int yeast_braid(size_t n, void const *const *br, const size_t *len, void **out, size_t *out_len);

/* SHA-256 over host buffers, computed on the GPU (the reference: OpenSSL's SHA256(), broad.c:266,
   brand.c:241).  Return 0, or -1 when the GPU path is unavailable. */
int yeast_sha256(const void *buf, size_t len, unsigned char out[32]);
int yeast_sha256_batch(size_t n, void const *const *buf, const size_t *len, unsigned char *out /* 32 * n */);

#ifdef __cplusplus
}
#endif
#endif
