/*
 * yeast_common.h -- internal definitions shared by the HIP kernels and the
 * host side of libyeast_mi355x.so.  Not installed; the public C-ABI is
 * include/yeast.h and include/yeast_batch.h.
 */
#ifndef YEAST_COMMON_H
#define YEAST_COMMON_H

#include <stdint.h>
#include <stddef.h>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;
typedef int32_t i32;

/* Per-stream status.  0..4 are the reference's return codes (yeast.h:13-22);
   5 and 6 never leave the library through yeast(): the host retries. */
enum {
    YS_OK = 0,
    YS_NOMEM = 1,
    YS_EOF = 2,
    YS_INVALID = 3,
    YS_MISMATCH = 4,
    YS_CAPACITY = 5,  /* output region too small; record.check = bytes needed so far */
    YS_WORKSPACE = 6  /* per-stream table arena too small; retry with a larger one */
};

/* Result record, one per stream (32 bytes, what RCCL gathers). */
typedef struct {
    int32_t rc;
    uint32_t flags; /* bits 0..7: window bits; bits 8..31: meta-blocks seen (saturating) */
    uint64_t got;   /* bytes produced (or matched, in compare mode) */
    uint64_t used;  /* input bytes consumed, exactly as yeast.c:1388 reports them */
    uint64_t check; /* check value (kind chosen by the caller), or bytes needed for rc 5 */
} ys_record;

/* Decode launch parameters (device pointers). */
typedef struct {
    const u8 *src;       /* packed compressed streams; >= YS_SRC_PAD readable bytes after the end */
    const u64 *src_off;  /* n+1 byte offsets into src */
    u8 *dst;             /* output arena (decode mode) or expected data (compare mode) */
    const u64 *dst_off;  /* n+1 byte offsets: stream i owns dst[dst_off[i] .. dst_off[i+1]) */
    ys_record *rec;      /* n records */
    u8 *ws;              /* workspace: one slot of ws_stride bytes per resident group */
    u64 ws_stride;
    u32 *counter;        /* work-queue head, zeroed before launch */
    u32 *retry_count;    /* number of streams queued for the large-arena pass */
    u32 *retry_list;     /* their indices (n entries) */
    const u32 *order;    /* batch kernel: the streams in the order they are handed out (longest first), or null */
    u32 debug;           /* measurement aids of the batch kernel (YEAST_DEBUG), never set in normal use */
    u32 retry;           /* 0: first pass over all n streams; 1: pass over retry_list */
    u32 n;
    const u8 *dict;      /* 122,784-byte static dictionary */
    const u8 *xforms;    /* 121 x 20-byte transform records */
    const u8 *xdict;     /* every transform of every dictionary word (built on the device at start-up) */
    const u32 *xdict_base; /* [121][21]: byte offset in xdict of (transform, word length 4..24), entries back to back */
    const u32 *maxlen;   /* batch kernel: the longest stream's compressed bytes (written by the hand-out order's kernels) */
    u32 delay;           /* batch kernel, one round with every block resident: a wavefront whose streams are shorter than the
                            longest begins that much later, in ns per byte of difference (0 = at once): see yeast_lane.hip */
    u32 rootcaps;        /* root table widths: other | literal << 8 | insert-and-copy << 16 | distance << 24;
                            0 = 8 bits everywhere (the only setting of the kernels other than the lean one) */
} ys_params;

#define YS_SRC_PAD 32

/* workspace slot layout (bytes from the slot base) */
#define YS_WS_ITAB 0        /* u32[256]  byte offset (from the slot base) of each insert-and-copy table */
#define YS_WS_DTAB 1024     /* u32[1024] byte offset of the distance table per (block type, context) */
#define YS_WS_MODE 5120     /* u8[256]   literal context modes */
#define YS_WS_DMAP 5376     /* u8[1024]  distance context map */
#define YS_WS_SORT 6400     /* u16[704]  symbols in canonical order (table build scratch) */
#define YS_WS_MTF 7808      /* u8[256]   move-to-front list */
#define YS_WS_LDIR 8064     /* u32[256]  byte offset of each literal table (header only) */
#define YS_WS_DDIR 9088     /* u32[256]  byte offset of each distance table (header only) */
#define YS_WS_ARENA 10240   /* block-switch tables, literal context map, literal table map, code tables */
/* worst case: 16 KiB literal map + 64 KiB literal table map + 774 tables of at most 1,104 u16 entries */
/* output bytes up to which the batch kernel (one stream per lane) keeps a stream; longer ones are handed to the
   retry pass (one wavefront per stream, wide copies) */
#define YS_LANE_MAX_OUT (1ull << 20)
#define YS_WS_MAX (YS_WS_ARENA + 16384 + 65536 + 774 * 2208 + 4096)

#endif
