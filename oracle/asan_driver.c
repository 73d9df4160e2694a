/* oracle/asan_driver.c -- TEST INFRASTRUCTURE ONLY.  Runs this repo's CPU restatement (yeast_oracle.c,
 * br_oracle.c) over a file of cases in a process built with AddressSanitizer and UBSan (`make -C oracle asan`),
 * so that a restatement that reads or writes out of bounds on some fixture is found out, not trusted
 * (SURVEY section 5, row 2).  tests/test_oracle_golden.py writes the case file and compares the lines printed
 * here with the fixtures.
 *
 * case file: u32 count, then per case  u32 kind (0 decode, 1 compare, 2 .br file), u64 length, the bytes,
 *            and for kind 1  u64 length, the expected bytes.
 * output:    one line per case:  rc got used xxh64-of-output (hex)      (.br: code length segments xxh64)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int oracle_yeast(void **dest, size_t *got, const void *source, size_t *len, int cmp);
uint64_t oracle_xxh64(const void *input, size_t len, uint64_t seed);
int oracle_broad(const void *br, size_t len, void **out, size_t *out_len, size_t *segments);

static void *take(FILE *f, uint64_t *n)
{
    if (fread(n, 8, 1, f) != 1)
        exit(2);
    /* exactly n bytes, so that the sanitizer sees a read one past the input */
    void *p = malloc(*n ? *n : 1);
    if (!p || (*n && fread(p, 1, *n, f) != *n))
        exit(2);
    return p;
}

int main(int argc, char **argv)
{
    FILE *f = argc > 1 ? fopen(argv[1], "rb") : NULL;
    uint32_t count, kind;
    if (!f || fread(&count, 4, 1, f) != 1)
        return 2;
    for (uint32_t i = 0; i < count; i++) {
        uint64_t n, en = 0;
        if (fread(&kind, 4, 1, f) != 1)
            return 2;
        void *src = take(f, &n), *exp = kind == 1 ? take(f, &en) : NULL;
        if (kind == 2) {
            void *out = NULL;
            size_t on = 0, seg = 0;
            int rc = oracle_broad(src, n, &out, &on, &seg);
            printf("%d %zu %zu %016llx\n", rc, on, seg, (unsigned long long)oracle_xxh64(out ? out : "", on, 0));
            free(out);
        } else {
            void *dest = exp;
            size_t got = en, len = n;
            int rc = oracle_yeast(&dest, &got, src, &len, kind == 1);
            uint64_t h = kind == 0 && rc == 0 ? oracle_xxh64(dest ? dest : "", got, 0) : 0;
            printf("%d %zu %zu %016llx\n", rc, got, len, (unsigned long long)h);
            if (kind == 0)
                free(dest);
        }
        free(src);
        free(exp);
    }
    fclose(f);
    return 0;
}
