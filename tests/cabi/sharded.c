/* The multi-GPU provers from plain C (r04 verdict, next-round item 2): W processes -- here forked from one binary and sharing
 * ONE GPU through the host-staged transport; on a node, one per GPU through zk_comm_create -- prove ONE segment together with
 * `zk_prove_segment_table_parallel` (SURVEY 8(e) level 2), the tables named on the command line ROW-SHARDED over all ranks
 * (level 3).  No Python, no torch, no torch.distributed in any of the processes: include/zkstark.h, the generated AllStark
 * registry, the HIP runtime for device buffers.  This is the body a Rust caller has after the switch.
 * Input: the segment.bin of tests/cabi/segment.c (eth_mainnet tables).  Every rank writes the digest lines of segment.c to
 * <out_prefix>.<rank>; the test compares each of them with the single-GPU proof of the Python mirror.
 * Build: gcc -std=c11 sharded.c -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -lzkstark_hip -lamdhip64
 * Usage: sharded segment.bin <world> <comm name> <out prefix> <fri_mode> [row-sharded table index ...] */
#define _POSIX_C_SOURCE 200809L
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

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

static int run_rank(const char *path, unsigned rank, unsigned world, const char *name, const char *out_prefix, unsigned fri_mode,
                    const uint8_t *wide) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); return 2; }
    if (rd(f) != 0) { fprintf(stderr, "eth_mainnet segments only\n"); return 2; }
    zk_cfg cfg;
    uint32_t *cw = (uint32_t *)&cfg;
    for (int i = 0; i < 8; ++i) cw[i] = (uint32_t)rd(f);
    uint64_t labels[4];
    for (int i = 0; i < 4; ++i) labels[i] = rd(f);
    const size_t n_pv = (size_t)rd(f);
    uint64_t *pv = malloc((n_pv ? n_pv : 1) * 8);
    for (size_t i = 0; i < n_pv; ++i) pv[i] = rd(f);
    enum { NT = ZK_ALLSTARK_NUM_TABLES };

    zk_ctx *ctx = NULL;
    if (zk_ctx_create(0, &ctx) != ZK_OK) { fprintf(stderr, "zk_ctx_create failed (no GPU?)\n"); return 3; }
    zk_comm *comm = NULL;
    if (zk_comm_create_host(ctx, name, rank, world, 0, &comm) != ZK_OK) { fprintf(stderr, "rank %u: zk_comm_create_host: %s\n", rank, zk_last_error(ctx)); return 3; }

    /* first pass over the file: the shapes (every rank needs every table's height for the assignment) */
    zk_table_in tables[NT];
    memset(tables, 0, sizeof tables);
    uint64_t *host[NT];
    size_t n_cols[NT];
    unsigned log_n[NT];
    for (size_t t = 0; t < NT; ++t) {
        const int in_use = (int)rd(f);
        log_n[t] = (unsigned)rd(f);
        n_cols[t] = ZK_ALLSTARK_TABLE_COLUMNS[t];
        const size_t words = n_cols[t] << log_n[t];
        host[t] = malloc(words * 8);
        if (fread(host[t], 8, words, f) != words) { fprintf(stderr, "short trace for table %zu\n", t); return 2; }
        tables[t].n_cols = n_cols[t];
        tables[t].log_n = log_n[t];
        tables[t].air_id = ZK_ALLSTARK_TABLE_AIR[t];
        tables[t].air_consts = t == ZK_ALLSTARK_CPU ? labels : NULL;
        tables[t].n_air_consts = t == ZK_ALLSTARK_CPU ? 4 : 0;
        tables[t].lookup_program = ZK_ALLSTARK_LOOKUP_PROGRAM[t];
        tables[t].lookup_words = ZK_ALLSTARK_LOOKUP_WORDS[t];
        tables[t].in_use = in_use;
        tables[t].optional = ZK_ALLSTARK_TABLE_OPTIONAL[t];
    }
    fclose(f);
    uint32_t owner[NT];
    if (zk_assign_tables(n_cols, log_n, NT, world, wide, owner) != ZK_OK) return 1;
    /* a rank uploads what it holds: the whole trace of a table it owns, its ROW BLOCK of a row-sharded one */
    void *dev[NT] = {0};
    for (size_t t = 0; t < NT; ++t) {
        const size_t n = (size_t)1 << log_n[t];
        if (wide[t]) {
            const size_t nb = n / world;
            if (hipMalloc(&dev[t], n_cols[t] * nb * 8) != hipSuccess) return 3;
            if (hipMemcpy2D(dev[t], nb * 8, host[t] + (size_t)rank * nb, n * 8, nb * 8, n_cols[t], hipMemcpyHostToDevice) != hipSuccess) return 3;
            tables[t].d_trace = dev[t];
            tables[t].col_stride = nb;
        } else if (owner[t] == rank) {
            if (hipMalloc(&dev[t], n_cols[t] * n * 8) != hipSuccess || hipMemcpy(dev[t], host[t], n_cols[t] * n * 8, hipMemcpyHostToDevice) != hipSuccess) return 3;
            tables[t].d_trace = dev[t];
            tables[t].col_stride = n;
        }
        free(host[t]);
    }

    zk_segment_proof *proof = NULL;
    int rc = zk_prove_segment_table_parallel(ctx, comm, &cfg, tables, NT, wide, ZK_ALLSTARK_CTL_WIRING, ZK_ALLSTARK_CTL_WIRING_WORDS, pv, n_pv,
                                             ZK_ALLSTARK_CONSTRAINT_DEGREE, ZK_ALLSTARK_MEM_BEFORE, ZK_ALLSTARK_MEM_AFTER, fri_mode, &proof);
    if (rc != ZK_OK) { fprintf(stderr, "rank %u: zk_prove_segment_table_parallel -> %d: %s\n", rank, rc, zk_last_error(ctx)); return 1; }

    char out_path[512];
    snprintf(out_path, sizeof out_path, "%s.%u", out_prefix, rank);
    FILE *o = fopen(out_path, "w");
    if (!o) { perror(out_path); return 2; }
    uint64_t cc[16];
    const size_t ncc = zk_segment_proof_ctl_challenges(proof, cc, 16);
    fprintf(o, "ctl_challenges");
    for (size_t i = 0; i < ncc; ++i) fprintf(o, " %" PRIu64, cc[i]);
    fprintf(o, "\n");
    for (size_t t = 0; t < NT; ++t) {
        const zk_table_proof *tp = zk_segment_proof_table(proof, t);
        if (!tp) { fprintf(o, "table %s absent\n", ZK_ALLSTARK_TABLE_NAMES[t]); continue; }
        zk_table_proof_view v;
        if (zk_table_proof_get(tp, &v) != ZK_OK) return 1;
        uint64_t h = 0xCBF29CE484222325ULL;
        h = fnv(h, v.init_challenger_state, 12);
        h = fnv(h, v.trace_cap, 4 * v.cap_digests);
        if (v.aux_cap) h = fnv(h, v.aux_cap, 4 * v.cap_digests);
        h = fnv(h, v.quotient_cap, 4 * v.cap_digests);
        h = fnv(h, v.openings, 2 * v.n_openings);
        h = fnv(h, v.opening_proof, v.proof_words);
        fprintf(o, "table %s degree_bits %u words %zu fnv %016" PRIx64 "\n", ZK_ALLSTARK_TABLE_NAMES[t], v.degree_bits,
                12 + 4 * v.cap_digests * (v.aux_cap ? 3 : 2) + 2 * v.n_openings + v.proof_words, h);
    }
    const size_t cap_words = (size_t)4 << cfg.cap_height;
    uint64_t *mb = malloc(cap_words * 8), *ma = malloc(cap_words * 8);
    zk_segment_proof_mem_caps(proof, mb, ma, cap_words);
    fprintf(o, "mem_caps fnv %016" PRIx64 " %016" PRIx64 "\n", fnv(0xCBF29CE484222325ULL, mb, cap_words), fnv(0xCBF29CE484222325ULL, ma, cap_words));
    uint64_t st[3];
    zk_comm_stats(comm, st);
    fprintf(o, "comm %s rank %u of %u sent %" PRIu64 " received %" PRIu64 " collectives %" PRIu64 "\n", zk_comm_transport(comm), zk_comm_rank(comm),
            zk_comm_world(comm), st[0], st[1], st[2]);
    fclose(o);
    zk_segment_proof_free(proof);
    for (size_t t = 0; t < NT; ++t) if (dev[t]) hipFree(dev[t]);
    zk_comm_free(comm);
    zk_ctx_destroy(ctx);
    free(pv); free(mb); free(ma);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s segment.bin world comm_name out_prefix fri_mode [row-sharded table ...]\n", argv[0]); return 2; }
    const unsigned world = (unsigned)atoi(argv[2]), fri_mode = (unsigned)atoi(argv[5]);
    uint8_t wide[ZK_ALLSTARK_NUM_TABLES] = {0};
    for (int i = 6; i < argc; ++i) { const int t = atoi(argv[i]); if (t >= 0 && t < ZK_ALLSTARK_NUM_TABLES) wide[t] = 1; }
    if (world == 0 || world > 16) return 2;
    /* fork BEFORE anything touches the HIP runtime: every rank is a process of its own, as it would be on a node */
    pid_t pids[16];
    for (unsigned r = 0; r < world; ++r) {
        pids[r] = fork();
        if (pids[r] < 0) { perror("fork"); return 3; }
        if (pids[r] == 0) _exit(run_rank(argv[1], r, world, argv[3], argv[4], fri_mode, wide));
    }
    int bad = 0;
    for (unsigned r = 0; r < world; ++r) {
        int st = 0;
        if (waitpid(pids[r], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) { fprintf(stderr, "rank %u failed (status %d)\n", r, st); bad = 1; }
    }
    return bad;
}
