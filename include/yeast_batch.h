/*
 * include/yeast_batch.h -- additive batch / device-resident entry points of
 * libyeast_mi355x.so, and the check-value functions.
 *
 * The reference decodes one stream per call (yeast.h:25).  Its callers that
 * handle many streams loop over yeast() (juxt.c:77-93 over files, broad.c:142-353
 * over the segments of a .br file) and then compute a check over each output
 * (brand.c:245-248, broad.c:278-281).  These entry points are what an FFI
 * binding for that loop would call instead: the same per-stream results
 * (rc, *got, *len, output bytes, check value) for n independent streams in
 * one launch.  Plain pointers and sizes only.
 */
#ifndef YEAST_BATCH_H
#define YEAST_BATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* check kinds (the three non-cryptographic checks of br.h / brand.c:245-248) */
#define YEAST_CHECK_NONE 0
#define YEAST_CHECK_CRC32C 1
#define YEAST_CHECK_XXH32 2
#define YEAST_CHECK_XXH64 3

/* Per-stream result record (32 bytes).  rc uses yeast()'s codes 0..4; the
   device-resident call can also report 5 = the stream's output region was too
   small (check then holds the size needed so far). */
typedef struct {
    int32_t rc;
    uint32_t flags; /* bits 0..7 window bits, bits 8..31 meta-blocks decoded */
    uint64_t got;   /* == *got of yeast() */
    uint64_t used;  /* == *len of yeast() on return */
    uint64_t check; /* check value of the got output bytes (kind chosen by the caller) */
} yeast_record;

/* Extra readable bytes required after the last stream of a device-resident
   source buffer (the bit reader prefetches whole words). */
#define YEAST_SRC_PAD 32

/*
 * Host-pointer batch: yeast(cmp=0) for each of n streams (replaces a loop
 * over yeast.h:25 such as juxt.c:77-93).  len[i] is in: size / out: bytes
 * consumed; dest[i] receives a malloc() buffer (NULL when got[i] is 0);
 * rc[i] the per-stream return code.  If check_kind != 0, check[i] receives
 * the check value of stream i's output (crc32c(0,..), XXH32(..,0), XXH64(..,0)).
 * Returns 0, or -1 when the GPU path is unavailable (message on stderr) and 1
 * when host memory ran out.
 */
int yeast_batch(size_t n, void const *const *src, size_t *len, void **dest, size_t *got, int *rc,
                unsigned check_kind, uint64_t *check);

/*
 * Device-resident batch.  All pointers are device pointers on the current
 * HIP device; the call only enqueues work on hip_stream (a hipStream_t, or
 * NULL for the default stream) and returns.
 *   d_src      packed compressed streams, YEAST_SRC_PAD readable bytes after the end
 *   d_src_off  n+1 offsets: stream i is d_src[d_src_off[i] .. d_src_off[i+1])
 *   d_dst      output arena; stream i may use d_dst[d_dst_off[i] .. d_dst_off[i+1])
 *   d_rec      n records, written by the kernels
 *   lanes_per_stream  0 = library default (by batch size), else 1 (batch kernel: one stream per lane) or 8, 16, 64
 *              (cooperative kernel: that many lanes of a wavefront cooperate on one stream); 65 = one wavefront
 *              per stream with the stream's prefix-code tables in LDS (lowest latency; the default below
 *              6,656 streams, 1 from there on)
 * Calls on one device share the library's per-device counters and table workspaces: a call on another stream
 * than the last one first waits (on the device, through an event) for the last call's kernels, so batches never
 * overlap on one device whatever streams they are given to; calls from several host threads are serialised.
 * Returns 0 or a negative error (see yeast_last_error()).
 */
int yeast_batch_device(const void *d_src, const uint64_t *d_src_off, void *d_dst, const uint64_t *d_dst_off,
                       yeast_record *d_rec, size_t n, unsigned check_kind, int lanes_per_stream,
                       void *hip_stream);

/*
 * yeast_batch_device() for callers that cannot allocate: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  No counterpart in the reference.  Same arguments, same records, same output
 * bytes; the call makes no allocation of any kind (device, pinned or through a library): the decode kernels'
 * workspace -- counters and the table arenas of the resident wavefronts, which the other entry points allocate
 * -- is a static device array of 29 MB that each GPU the library runs on holds until the process ends.
 *   - the device must have been set up before, by yeast_init() (which allocates the dictionary tables once
 *     per device) or by any earlier call of another entry point; otherwise the call fails with -1;
 *   - at most 16,384 streams per call; more is refused with -1 and a message (yeast_last_error());
 *   - sixteen wavefronts decode (one stream each at a time, worst-case table arenas, so every valid stream
 *     decodes): meant for modest batches, not for throughput;
 *   - every call on a GPU uses that GPU's one array, so calls that overlap on one GPU race on it.  Ordinary
 *     (eager) calls the library orders itself, one after the other, across streams too (an event per device).
 *     On a stream that is being captured into a HIP graph it neither waits for nor records that event (an event
 *     from outside the capture would invalidate it): the captured call, and every replay of the graph, is the
 *     caller's to order against other calls of the library on the same GPU.  Other processes' kernels are not
 *     the library's business either.
 */
int yeast_init(void); /* set up the current device (allocates its constant tables once); 0 or -1 */
int yeast_batch_device_static(const void *d_src, const uint64_t *d_src_off, void *d_dst, const uint64_t *d_dst_off,
                              yeast_record *d_rec, size_t n, unsigned check_kind, void *hip_stream);

/* Check values only, over outputs already in device memory: fills
   d_rec[i].check from d_dst[d_dst_off[i] .. +d_rec[i].got). */
int yeast_check_device(const void *d_dst, const uint64_t *d_dst_off, yeast_record *d_rec, size_t n,
                       unsigned check_kind, void *hip_stream);

/*
 * The reference's check functions over host buffers, computed on the GPU:
 *   crc32c()  replaces crc32c.h:13 / crc32c.c:258 (chainable through crc)
 *   XXH32()   replaces xxhash.h XXH32  (xxhash.c:365)
 *   XXH64()   replaces xxhash.h XXH64  (xxhash.c:505)
 * There is no in-band error value in these signatures: on a GPU failure they write a message to stderr,
 * return 0 and count the failure; yeast_check_error() returns (and clears) the calling thread's count, so a
 * caller that must tell a check of 0 from a failure asks it after the call (the digest functions of the
 * streaming forms below do the same).  The host process is never aborted.
 */
uint32_t crc32c(uint32_t crc, void const *buf, size_t len);
unsigned int XXH32(const void *input, size_t len, unsigned int seed);
unsigned long long XXH64(const void *input, size_t len, unsigned long long seed);
/* Measurement aid: the name of the decode kernel the calling thread's last batch call launched ("k_decode_lean<32, 2>",
   "k_decode_wave", ...), as a profiler prints it; "" before any call. */
const char *yeast_last_kernel(void);
int yeast_check_error(void); /* check-value calls of this thread that failed since the last call of this; 0 = none */

/*
 * The streaming forms of the two hashes (xxhash.h:139-167, xxhash.c:588-960), which broad.c:31-62,121-122
 * and braid.c:205-242,345 use for header checks and the check of checks; crc32c() above is already
 * chainable through its first argument (crc32c.c:258).  Same names, argument meaning and return
 * values (0 = XXH_OK, 1 = XXH_ERROR); the state types have the reference's sizes.  The state holds
 * the accumulators and the bytes of an unfinished stripe; every update that completes a stripe and
 * every digest is one small launch.
 */
typedef struct { long long ll[6]; } XXH32_state_t;
typedef struct { long long ll[11]; } XXH64_state_t;
XXH32_state_t *XXH32_createState(void);
int XXH32_freeState(XXH32_state_t *state);
XXH64_state_t *XXH64_createState(void);
int XXH64_freeState(XXH64_state_t *state);
int XXH32_reset(XXH32_state_t *state, unsigned seed);
int XXH32_update(XXH32_state_t *state, const void *input, size_t length);
unsigned int XXH32_digest(const XXH32_state_t *state);
int XXH64_reset(XXH64_state_t *state, unsigned long long seed);
int XXH64_update(XXH64_state_t *state, const void *input, size_t length);
unsigned long long XXH64_digest(const XXH64_state_t *state);

/* Give back what the library holds on the current device between calls: its input, output and
   workspace buffers in HBM and the pinned host staging of the host-pointer calls (both sized by the
   largest call so far; the dictionary tables, 20 MB, stay).  The next call allocates again.  No
   counterpart in the reference, whose yeast() keeps nothing between calls; for long-running hosts after
   an unusually large batch.  Waits for the device's outstanding work; 0 on success. */
int yeast_trim(void);

/* The YEAST_* environment knobs (measurement and test aids: INTEGRATION.md) are read when the library is first
   used, never on the path of a call; this reads them again.  No counterpart in the reference. */
void yeast_reload_knobs(void);

/* Diagnostics. */
const char *yeast_last_error(void);   /* thread-local text of the last failure, "" if none */
int yeast_device_count(void);         /* HIP devices visible, <= 0 if none */
const unsigned char *yeast_debug_table(int which, size_t *size);
/* which: 0,1,2 = the three literal-context tables (yeast.c:700-740), 3 = dictionary,
   4 = transform records; for table known-answer tests (check.c:94-99) */

#ifdef __cplusplus
}
#endif
#endif
