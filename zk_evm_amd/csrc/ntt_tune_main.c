/* zk_ntt_tune [device] -- the OFFLINE tuner of libzkstark_hip.so's plan table (csrc/ntt_host.inc "the plan table").
 *
 * Three decisions of the library have two implementations that produce the same words: the NTT passes of a transform shape (LDS tile
 * kernels | lane-swap kernels), from_values over all columns at once | in column batches, the small Merkle levels of a segment's
 * trace trees per tree | batched.  This program runs both forms of every shape on the device, compares their outputs word for word,
 * times them, and prints
 *     line 1:  the plan string ("v20f0=2;d21f1=1;b20r1=96x1;T=1;...") -- what ZK_NTT_SWAP_PLANS / zk_ctx_set_plans take and what a
 *              maintainer pastes into kBuiltinPlans once a hardware parity run (tests/test_gpu_zz_plans.py) is green on it;
 *     the rest: one human-readable line per trial (both times, mismatching words).
 * Exit status: 0 = every second form produced identical words; 7 = at least one DIFFERED (a parity failure of shipped code: fix
 * or delete that kernel); 3 .. 6 = the run itself failed.  The library never starts this program and never measures anything
 * on its own: which kernels serve a shape is data.
 * Built next to the library by zk_evm_amd/build.py (gcc, -lzkstark_hip, rpath $ORIGIN). */
#include <stdio.h>
#include <stdlib.h>

#include "../../include/zkstark.h"

size_t zki_ntt_tune_report(const zk_ctx *ctx, char *out, size_t max);
int zki_ntt_tune_all(zk_ctx *ctx, int *n_differ);
int zki_tree_batch_trial(zk_ctx *ctx, const unsigned *log_ns, int *differ);

int main(int argc, char **argv) {
    const int device = argc > 1 ? atoi(argv[1]) : 0;
    zk_ctx *ctx = NULL;
    int rc = zk_ctx_create(device, &ctx), differ = 0, tree_differ = 0;
    if (rc != ZK_OK || !ctx) { fprintf(stderr, "zk_ntt_tune: zk_ctx_create(%d) = %d\n", device, rc); return 3; }
    zk_ctx_set_plans(ctx, "");                 /* start from the empty table, whatever the environment says */
    rc = zki_ntt_tune_all(ctx, &differ);
    if (rc != ZK_OK) { fprintf(stderr, "zk_ntt_tune: %d %s\n", rc, zk_last_error(ctx)); zk_ctx_destroy(ctx); return 4; }
    if (!getenv("ZK_TUNE_SKIP_TREES")) {      /* with the NTT plans just decided: a segment proven with the tree tops per tree and batched */
        rc = zki_tree_batch_trial(ctx, NULL, &tree_differ);
        if (rc != ZK_OK) { fprintf(stderr, "zk_ntt_tune: tree trial %d %s\n", rc, zk_last_error(ctx)); zk_ctx_destroy(ctx); return 6; }
    }
    static char plans[4100], report[1 << 16];
    zk_ctx_get_plans(ctx, plans, sizeof plans);
    zki_ntt_tune_report(ctx, report, sizeof report);
    zk_ctx_destroy(ctx);
    printf("%s\n%s", plans, report);
    if (fflush(stdout) != 0) return 5;
    return differ + tree_differ ? 7 : 0;
}
