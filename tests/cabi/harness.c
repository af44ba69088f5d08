/* A plain-C caller of the C ABI (include/zkstark.h) -- what the Rust `zkstark-sys` shim of INTEGRATION.md does, with no
 * Python in the process: PolynomialBatch::from_values on seeded columns (the generator of tests/golden/commit_caps.json),
 * then the Memory-table generator, and the error path.  Prints one line per result; tests/test_gpu_cabi_harness.py
 * compares them with the golden fixture.  Build: gcc -std=c11 harness.c -I include -L zk_evm_amd -lzkstark_hip.
 * usage: harness n_cols log_n rate_bits cap_height hasher seed */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zkstark.h"

static uint64_t splitmix64(uint64_t seed, uint64_t i) { /* element i (1-based stream index) */
    uint64_t z = i * 0x9E3779B97F4A7C15ULL + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

#define CHECK(call)                                                                                    \
    do {                                                                                               \
        int rc_ = (call);                                                                              \
        if (rc_ != ZK_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, zk_last_error(ctx)); return 1; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc != 7) { fprintf(stderr, "usage: %s n_cols log_n rate_bits cap_height hasher seed\n", argv[0]); return 2; }
    const size_t n_cols = strtoull(argv[1], 0, 10);
    const unsigned log_n = (unsigned)strtoul(argv[2], 0, 10);
    zk_cfg cfg = {(uint32_t)strtoul(argv[3], 0, 10), (uint32_t)strtoul(argv[4], 0, 10), (uint32_t)strtoul(argv[5], 0, 10), 2, 16, 84, 4, 5};
    const uint64_t seed = strtoull(argv[6], 0, 10);
    const size_t n = (size_t)1 << log_n;
    zk_ctx *ctx = NULL;
    if (zk_ctx_create(0, &ctx) != ZK_OK) { fprintf(stderr, "zk_ctx_create failed (no GPU?)\n"); return 3; }
    printf("version %s\n", zk_version());

    uint64_t **cols = malloc(n_cols * sizeof *cols);
    for (size_t c = 0; c < n_cols; ++c) {
        cols[c] = malloc(n * sizeof(uint64_t));
        for (size_t i = 0; i < n; ++i) cols[c][i] = splitmix64(seed + c, i + 1);
    }
    zk_batch *b = NULL;
    CHECK(zk_commit_columns(ctx, &cfg, (const uint64_t *const *)cols, n_cols, log_n, &b));
    const size_t cap_words = (size_t)4 << cfg.cap_height;
    uint64_t *cap = malloc(cap_words * 8);
    CHECK(zk_batch_cap(b, cap));
    printf("cap");
    for (size_t i = 0; i < cap_words; ++i) printf(" %" PRIu64, cap[i]);
    printf("\n");
    zk_batch_free(b);

    /* Memory table from a three-operation log: write 7 at (0,1,0) ts 3, read it back at ts 9, write at (0,1,1) */
    uint64_t ops[3 * 9] = {2 /* write, filter */, 3, 0, 1, 0, 7, 0, 0, 0,
                           3 /* read,  filter */, 9, 0, 1, 0, 7, 0, 0, 0,
                           2, 4, 0, 1, 1, 5, 0, 0, 0};
    zk_memory_gen *gen = NULL;
    CHECK(zk_memory_trace_begin(ctx, ops, 3, NULL, 0, &gen));
    printf("memory log_n %u unpadded %zu\n", zk_memory_gen_log_n(gen), zk_memory_gen_unpadded_length(gen));
    zk_memory_gen_free(gen);

    /* error path: a null config is refused with a message, nothing crashes */
    if (zk_commit_columns(ctx, NULL, (const uint64_t *const *)cols, n_cols, log_n, &b) == ZK_OK) return 4;
    printf("error \"%s\"\n", zk_last_error(ctx));
    zk_ctx_destroy(ctx);
    for (size_t c = 0; c < n_cols; ++c) free(cols[c]);
    free(cols);
    free(cap);
    return 0;
}
