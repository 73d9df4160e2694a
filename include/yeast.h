/*
 * include/yeast.h -- the drop-in decode entry point of libyeast_mi355x.so.
 *
 * Replaces the reference interface /root/reference/yeast.h:25 (implemented by
 * yeast.c:1353-1400) symbol for symbol, so the reference's own callers
 * (deb.c:106,127  juxt.c:88  brand.c:299  broad.c:243) link against this
 * library unchanged.  The decode itself runs on the GPU (HIP kernels for
 * gfx950); there is no CPU decode path in this library.
 *
 * Contract kept from the reference (yeast.h:7-23):
 *   cmp == 0: decode source[0..*len-1].  On return -- success or failure --
 *     *len is the number of input bytes consumed, *dest a malloc()-family
 *     buffer with the output (NULL when *got is 0) that the caller free()s,
 *     *got the number of bytes produced so far.
 *   cmp != 0: *dest / *got hold the expected data; nothing is written to it,
 *     *dest is left alone, *got returns how far the output matched.
 *   return 0 ok, 1 out of memory, 2 premature end of input, 3 invalid
 *     compressed data, 4 compare mismatch (cmp only).
 * One addition: -1 is returned (and a message written to stderr) when no
 * usable HIP device / kernel image is available -- the library never falls
 * back to a CPU decoder.
 */
#ifndef YEAST_H
#define YEAST_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

int yeast(void **dest, size_t *got, void const *source, size_t *len, int cmp);

/* yeast.h:30-32 / yeast.c:36: the reference's harnesses built with -DDEBUG
   write this.  Level >= 1 prints one line per call to stderr. */
extern int yeast_verbosity;

#ifdef __cplusplus
}
#endif
#endif
