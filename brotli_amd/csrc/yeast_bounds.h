/*
 * yeast_bounds.h -- the bounds-checked debug build of the kernels (-DYS_BOUNDS, tools/build_bounds.sh ->
 * brotli_amd/libyeast_bounds.so): the counterpart of a sanitizer for device code (GPU AddressSanitizer is not to be
 * had on every pool).  No counterpart in the reference, whose C code runs under the host's sanitizers.
 *
 * Every global-memory access of the decode and check kernels goes through a helper (the gld, gst, sld and lds_load families); in
 * this build each helper first looks the address up in a list of ranges -- the caller's buffers of the call and the
 * library's own buffers, which the host side writes into each code object before the launches -- and an access
 * outside all of them is not made (a load yields 0, a store is dropped) but counted, the first one with its
 * address and size.  The host side reads the count back after the call and fails it with a message naming the
 * nearest buffer.  Without -DYS_BOUNDS all of this compiles to nothing.
 */
#ifndef YEAST_BOUNDS_H
#define YEAST_BOUNDS_H

#define YSB_MAX 16
struct ys_bounds_t {
    unsigned long long lo[YSB_MAX], hi[YSB_MAX];
    unsigned n, hits;        /* n == 0: nothing is checked */
    unsigned long long addr; /* of the first access outside every range */
    unsigned size, pad;
    unsigned long long more[6]; /* the next ones (address; size in the top byte) */
};

#if defined(YS_BOUNDS) && defined(__HIPCC__)
__device__ ys_bounds_t g_ysb; /* one per code object */
__device__ __forceinline__ bool ysb_ok(const void *p, unsigned size)
{
    const unsigned n = g_ysb.n;
    if (n == 0)
        return true;
    const unsigned long long a = (unsigned long long)(size_t)p;
    for (unsigned i = 0; i < n; i++)
        if (a >= g_ysb.lo[i] && a + size <= g_ysb.hi[i])
            return true;
    const unsigned k = atomicAdd(&g_ysb.hits, 1u);
    if (k == 0) {
        g_ysb.addr = a;
        g_ysb.size = size;
    } else if (k <= 6)
        g_ysb.more[k - 1] = a | ((unsigned long long)size << 56);
    return false;
}
#define YSB(p, n) ysb_ok((const void *)(p), (n))
/* the host-side access to this code object's list: extern "C" int ysb_set_<tu>(const ys_bounds_t *), ysb_get_<tu>(ys_bounds_t *) */
#define YSB_HOST(tu)                                                                                                   \
    extern "C" int ysb_set_##tu(const ys_bounds_t *b) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ysb), b, sizeof(*b)); } \
    extern "C" int ysb_get_##tu(ys_bounds_t *b) { return (int)hipMemcpyFromSymbol(b, HIP_SYMBOL(g_ysb), sizeof(*b)); }
#else
#define YSB(p, n) true
#define YSB_HOST(tu)
#endif

#endif
