/* zk_ntt_tune <device> -- the process in which libzkstark_hip.so tries its NTT plans (csrc/ntt_host.inc, ntt_swap_decide).
 *
 * The library spawns this program the first time a transform shape with two candidate plans is asked for: every such shape is
 * run through both plans here, on the device, compared word for word and timed, and the verdicts go back on stdout -- first
 * line "<v|d><L>f<free>=<1|2>;..." (the form ZK_NTT_SWAP_PLANS takes), then one human-readable line per shape.  A kernel that
 * faults takes this process down, not the caller's: the library reads that as "tile kernels everywhere".
 * Built next to the library by zk_evm_amd/build.py (gcc, -lzkstark_hip, rpath $ORIGIN). */
#include <stdio.h>
#include <stdlib.h>

#include "../../include/zkstark.h"

size_t zki_ntt_tune_report(char *out, size_t max);
size_t zki_ntt_tune_export(char *out, size_t max);
int zki_ntt_tune_all(zk_ctx *ctx);
int zki_tree_batch_trial(zk_ctx *ctx, const unsigned *log_ns);

int main(int argc, char **argv) {
    const int device = argc > 1 ? atoi(argv[1]) : 0;
    zk_ctx *ctx = NULL;
    int rc = zk_ctx_create(device, &ctx);
    if (rc != ZK_OK || !ctx) { fprintf(stderr, "zk_ntt_tune: zk_ctx_create(%d) = %d\n", device, rc); return 3; }
    rc = zki_ntt_tune_all(ctx);
    if (rc != ZK_OK) { fprintf(stderr, "zk_ntt_tune: %d %s\n", rc, zk_last_error(ctx)); zk_ctx_destroy(ctx); return 4; }
    if (!getenv("ZK_TUNE_SKIP_TREES")) {      /* with the NTT plans just decided: a segment proven with the tree tops per tree and batched */
        rc = zki_tree_batch_trial(ctx, NULL);
        if (rc != ZK_OK) { fprintf(stderr, "zk_ntt_tune: tree trial %d %s\n", rc, zk_last_error(ctx)); zk_ctx_destroy(ctx); return 6; }
    }
    static char verdicts[4096], report[1 << 16];
    zki_ntt_tune_export(verdicts, sizeof verdicts);
    zki_ntt_tune_report(report, sizeof report);
    zk_ctx_destroy(ctx);
    printf("%s\n%s", verdicts, report);
    return fflush(stdout) == 0 ? 0 : 5;
}
