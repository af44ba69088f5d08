/* A whole segment proof from plain C: include/zkstark.h + the generated AllStark registry include/zk_all_stark.h + the
 * HIP runtime for device buffers -- no Python, no torch in the process.  This is the body a Rust `prove_with_traces`
 * would have after the switch.  Input: one binary file of little-endian u64 words written by
 * tests/test_gpu_cabi_harness.py:
 *   cdk_erigon, 8 zk_cfg fields, 4 kernel labels, n_public_values, public value elements,
 *   then per table: in_use, log_n, n_cols * 2^log_n trace words (column-major)
 * Output: per table an FNV-1a digest over every word of its proof, the CTL challenges and the two memory caps; the
 * test compares them with the same proof obtained through the Python mirror.
 * Build: gcc -std=c11 segment.c -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -lzkstark_hip -lamdhip64 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>

#include <hip/hip_runtime_api.h>

#include "zk_all_stark.h"
#include "zkstark.h"

static uint64_t fnv(uint64_t h, const uint64_t *w, size_t n) {
    for (size_t i = 0; i < n; ++i)
        for (int b = 0; b < 8; ++b) h = (h ^ ((w[i] >> (8 * b)) & 0xFF)) * 0x100000001B3ULL;
    return h;
}

static uint64_t rd(FILE *f) {
    uint64_t v = 0;
    if (fread(&v, 8, 1, f) != 1) { fprintf(stderr, "short input file\n"); exit(2); }
    return v;
}

int main(int argc, char **argv) {
    if (argc != 2) { fprintf(stderr, "usage: %s segment.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    const int erigon = (int)rd(f);
    zk_cfg cfg;
    uint32_t *cw = (uint32_t *)&cfg;
    for (int i = 0; i < 8; ++i) cw[i] = (uint32_t)rd(f);
    uint64_t labels[4];
    for (int i = 0; i < 4; ++i) labels[i] = rd(f);
    const size_t n_pv = (size_t)rd(f);
    uint64_t *pv = malloc((n_pv ? n_pv : 1) * 8);
    for (size_t i = 0; i < n_pv; ++i) pv[i] = rd(f);

    const size_t n_tables = erigon ? ZK_ALLSTARK_ERIGON_NUM_TABLES : ZK_ALLSTARK_NUM_TABLES;
    const uint32_t *columns = erigon ? ZK_ALLSTARK_ERIGON_TABLE_COLUMNS : ZK_ALLSTARK_TABLE_COLUMNS;
    const uint32_t *air = erigon ? ZK_ALLSTARK_ERIGON_TABLE_AIR : ZK_ALLSTARK_TABLE_AIR;
    const int *optional = erigon ? ZK_ALLSTARK_ERIGON_TABLE_OPTIONAL : ZK_ALLSTARK_TABLE_OPTIONAL;
    const uint64_t *const *lookups = erigon ? ZK_ALLSTARK_ERIGON_LOOKUP_PROGRAM : ZK_ALLSTARK_LOOKUP_PROGRAM;
    const size_t *lookup_words = erigon ? ZK_ALLSTARK_ERIGON_LOOKUP_WORDS : ZK_ALLSTARK_LOOKUP_WORDS;
    const char *const *names = erigon ? ZK_ALLSTARK_ERIGON_TABLE_NAMES : ZK_ALLSTARK_TABLE_NAMES;
    const size_t cpu = erigon ? ZK_ALLSTARK_ERIGON_CPU : ZK_ALLSTARK_CPU;

    zk_ctx *ctx = NULL;
    if (zk_ctx_create(0, &ctx) != ZK_OK) { fprintf(stderr, "zk_ctx_create failed (no GPU?)\n"); return 3; }
    zk_table_in tables[ZK_ALLSTARK_ERIGON_NUM_TABLES] = {{0}};
    void *dev[ZK_ALLSTARK_ERIGON_NUM_TABLES] = {0};
    for (size_t t = 0; t < n_tables; ++t) {
        const int in_use = (int)rd(f);
        const unsigned log_n = (unsigned)rd(f);
        const size_t words = (size_t)columns[t] << log_n;
        uint64_t *host = malloc(words * 8);
        if (fread(host, 8, words, f) != words) { fprintf(stderr, "short trace for table %zu\n", t); return 2; }
        if (hipMalloc(&dev[t], words * 8) != hipSuccess ||
            hipMemcpy(dev[t], host, words * 8, hipMemcpyHostToDevice) != hipSuccess) {
            fprintf(stderr, "hipMalloc / hipMemcpy failed\n");
            return 3;
        }
        free(host);
        tables[t].d_trace = dev[t];
        tables[t].col_stride = (size_t)1 << log_n;
        tables[t].n_cols = columns[t];
        tables[t].log_n = log_n;
        tables[t].air_id = air[t];
        tables[t].air_consts = t == cpu ? labels : NULL;
        tables[t].n_air_consts = t == cpu ? 4 : 0;
        tables[t].lookup_program = lookups[t];
        tables[t].lookup_words = lookup_words[t];
        tables[t].in_use = in_use;
        tables[t].optional = optional[t];
    }
    fclose(f);

    zk_segment_proof *proof = NULL;
    int rc = zk_prove_segment(ctx, &cfg, tables, n_tables, erigon ? ZK_ALLSTARK_ERIGON_CTL_WIRING : ZK_ALLSTARK_CTL_WIRING,
                              erigon ? ZK_ALLSTARK_ERIGON_CTL_WIRING_WORDS : ZK_ALLSTARK_CTL_WIRING_WORDS, pv, n_pv,
                              ZK_ALLSTARK_CONSTRAINT_DEGREE, ZK_ALLSTARK_MEM_BEFORE, ZK_ALLSTARK_MEM_AFTER, &proof);
    if (rc != ZK_OK) { fprintf(stderr, "zk_prove_segment -> %d: %s\n", rc, zk_last_error(ctx)); return 1; }

    uint64_t cc[16];
    const size_t ncc = zk_segment_proof_ctl_challenges(proof, cc, 16);
    printf("ctl_challenges");
    for (size_t i = 0; i < ncc; ++i) printf(" %" PRIu64, cc[i]);
    printf("\n");
    for (size_t t = 0; t < n_tables; ++t) {
        const zk_table_proof *tp = zk_segment_proof_table(proof, t);
        if (!tp) { printf("table %s absent\n", names[t]); continue; }
        zk_table_proof_view v;
        if (zk_table_proof_get(tp, &v) != ZK_OK) return 1;
        uint64_t h = 0xCBF29CE484222325ULL;
        h = fnv(h, v.init_challenger_state, 12);
        h = fnv(h, v.trace_cap, 4 * v.cap_digests);
        if (v.aux_cap) h = fnv(h, v.aux_cap, 4 * v.cap_digests);
        h = fnv(h, v.quotient_cap, 4 * v.cap_digests);
        h = fnv(h, v.openings, 2 * v.n_openings);
        h = fnv(h, v.opening_proof, v.proof_words);
        printf("table %s degree_bits %u words %zu fnv %016" PRIx64 "\n", names[t], v.degree_bits,
               12 + 4 * v.cap_digests * (v.aux_cap ? 3 : 2) + 2 * v.n_openings + v.proof_words, h);
    }
    const size_t cap_words = (size_t)4 << cfg.cap_height;
    uint64_t *mb = malloc(cap_words * 8), *ma = malloc(cap_words * 8);
    zk_segment_proof_mem_caps(proof, mb, ma, cap_words);
    printf("mem_caps fnv %016" PRIx64 " %016" PRIx64 "\n", fnv(0xCBF29CE484222325ULL, mb, cap_words),
           fnv(0xCBF29CE484222325ULL, ma, cap_words));
    zk_segment_proof_free(proof);
    for (size_t t = 0; t < n_tables; ++t) hipFree(dev[t]);
    zk_ctx_destroy(ctx);
    free(pv); free(mb); free(ma);
    return 0;
}
