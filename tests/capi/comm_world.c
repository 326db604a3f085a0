/* comm_world.c -- TEST PROGRAM (tests/test_capi_comm_world.py): the multi-GPU step of the C ABI at world size N, as
 * N PROCESSES of one machine -- fcd_comm_create, a beam search per rank on its shard, ONE fcd_gather_results_dev to
 * rank 0, which compares with its own decode of all the reads (include/fcd.h; SURVEY.md 8e: reads shard, no
 * collective inside the search).  Linked against the emulator build of the library, with tests/stubs' shared-memory
 * stand-in for RCCL (FCD_RCCL_LIBRARY): no GPU anywhere -- "device" pointers are plain malloc'd memory here; on
 * MI355X they come from hipMalloc and the very same calls run over RCCL / xGMI.
 *
 *   comm_world WORLD SCENARIO
 *     uneven      shards of different sizes, two of them EMPTY
 *     wide        out_stride above 65535 rows on every rank (4-byte time indices on the wire)
 *     mixed       rank 1's results padded to another out_stride than everybody else's, across the 65535 boundary
 *     badheader   rank 1 and rank 0 disagree about rank 1's read count: the destination must report the shard
 *     allocfail   rank WORLD-1 cannot allocate its gather buffers (FCD_DEBUG_FAIL_GATHER_ALLOC): EVERY rank must come
 *                 back with FCD_E_NOMEM, nobody may hang in a collective
 *     prepfail    rank WORLD-1 fails BEFORE the size agreement (its offsets workspace: FCD_DEBUG_FAIL_GATHER_PREP): it
 *                 must still join the all-reduce, and every rank comes back with an error (the failing one with
 *                 FCD_E_NOMEM, the others with FCD_E_HIP "another rank failed")
 * exit status 0 = the scenario behaved; a watchdog (alarm) turns a hang into a failure. */
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include "fcd.h"

enum { T = 48, N = 5, BEAM = 5 };

#define CHECK(expr)                                                                              \
    do {                                                                                         \
        int rc__ = (expr);                                                                       \
        if (rc__ != FCD_OK) {                                                                    \
            fprintf(stderr, "rank %d: %s -> %d (%s)\n", g_rank, #expr, rc__, g_h ? fcd_last_error(g_h) : ""); \
            return 10;                                                                           \
        }                                                                                        \
    } while (0)

static int g_rank = -1;
static fcd_handle *g_h = NULL;

/* read `id` of the job: T x N posteriors from a little generator (the same on every rank) */
static void make_read(int64_t id, float *x) {
    uint32_t s = (uint32_t)(id * 2654435761u + 12345u);
    int t, c;
    for (t = 0; t < T; ++t) {
        float sum = 0.0f;
        for (c = 0; c < N; ++c) {
            s = s * 1664525u + 1013904223u;
            x[t * N + c] = (float)((s >> 8) & 0xFFFF) / 65536.0f + 0.01f;
            sum += x[t * N + c];
        }
        for (c = 0; c < N; ++c) x[t * N + c] /= sum;
    }
}

struct result_mem {
    uint8_t *labels;
    uint32_t *path, *out_len;
    int32_t *status;
    fcd_result r;
};

static int alloc_result(struct result_mem *m, int64_t n, int64_t stride) {
    memset(m, 0, sizeof *m);
    m->labels = (uint8_t *)calloc((size_t)(n ? n : 1) * (size_t)stride, 1);
    m->path = (uint32_t *)calloc((size_t)(n ? n : 1) * (size_t)stride, 4);
    m->out_len = (uint32_t *)calloc((size_t)(n ? n : 1), 4);
    m->status = (int32_t *)calloc((size_t)(n ? n : 1), 4);
    if (!m->labels || !m->path || !m->out_len || !m->status) return 1;
    m->r.labels = m->labels;
    m->r.path = m->path;
    m->r.out_len = m->out_len;
    m->r.status = m->status;
    m->r.out_stride = stride;
    return 0;
}

static int decode(fcd_handle *h, int64_t first, int64_t n, struct result_mem *m) {
    fcd_batch b;
    float *x = (float *)malloc((size_t)(n ? n : 1) * T * N * sizeof(float));
    int64_t i;
    int rc;
    if (!x) return 1;
    for (i = 0; i < n; ++i) make_read(first + i, x + i * T * N);
    memset(&b, 0, sizeof b);
    b.post = x;
    b.n_reads = n;
    b.T = T;
    b.S = 1;
    b.N = N;
    b.stride_read = T * N;
    b.stride_t = N;
    b.stride_n = 1;
    rc = fcd_beam_search_dev(h, &b, BEAM, 0.0f, 1, FCD_KERNEL_AUTO, &m->r);
    if (rc == FCD_OK) rc = fcd_synchronize(h);
    free(x);
    return rc;
}

static int run_rank(int world, int rank, const uint8_t *id, const char *scenario) {
    int64_t counts[64], view[64], first = 0, total = 0, stride = T, dst_stride = T;
    struct result_mem mine, all, want;
    fcd_comm *c = NULL;
    int k, rc, expect_nomem = !strcmp(scenario, "allocfail"), badheader = !strcmp(scenario, "badheader");
    const int prepfail = !strcmp(scenario, "prepfail");
    g_rank = rank;
    for (k = 0; k < world; ++k) counts[k] = 1 + (k * 5 + 2) % 4;
    if (!strcmp(scenario, "uneven") && world >= 2) {
        counts[1] = 0;
        counts[world - 1] = 0;
        if (world > 2) counts[0] = 7;
    }
    if (!strcmp(scenario, "wide")) stride = dst_stride = 66000;
    if (!strcmp(scenario, "mixed")) {
        stride = rank == 1 ? 66000 : T;
        dst_stride = 66000;
    }
    memcpy(view, counts, sizeof counts);
    if (badheader && rank == 0) view[1] = counts[1] + 1 <= 4 ? counts[1] + 1 : counts[1] - 1; /* (the largest count stays the largest) */
    for (k = 0; k < rank; ++k) first += counts[k];
    for (k = 0; k < world; ++k) total += view[k];
    /* one process per GPU: with FCD_EMU_DEVICES=N (the emulator's device count) rank k takes device k */
    CHECK(fcd_create(fcd_device_count() > 1 ? rank % fcd_device_count() : 0, &g_h));
    CHECK(fcd_comm_create(g_h, world, rank, id, &c));
    if (alloc_result(&mine, counts[rank], stride)) return 11;
    CHECK(decode(g_h, first, counts[rank], &mine));
    if (alloc_result(&all, total, dst_stride)) return 11;
    rc = fcd_gather_results_dev(c, &mine.r, counts[rank], view, 0, rank == 0 ? &all.r : NULL);
    if (prepfail) {
        const int want_rc = rank == world - 1 ? FCD_E_NOMEM : (world > 1 ? FCD_E_HIP : FCD_E_NOMEM);
        if (rc != want_rc) {
            fprintf(stderr, "rank %d: a rank failed before the size agreement, this one got %d, not %d (%s)\n", rank, rc, want_rc,
                    fcd_last_error(g_h));
            return 19;
        }
        (void)fcd_comm_destroy(c);
        (void)fcd_destroy(g_h);
        return 0;
    }
    if (expect_nomem) {
        if (rc != FCD_E_NOMEM) {
            fprintf(stderr, "rank %d: an allocation failed on one rank, this one got %d (%s)\n", rank, rc, fcd_last_error(g_h));
            return 12;
        }
        (void)fcd_comm_destroy(c);
        (void)fcd_destroy(g_h);
        return 0;
    }
    if (rc != FCD_OK) {
        fprintf(stderr, "rank %d: fcd_gather_results_dev -> %d (%s)\n", rank, rc, fcd_last_error(g_h));
        return 13;
    }
    rc = fcd_comm_synchronize(c);
    if (badheader) {
        if (rank == 0 && rc != FCD_E_INVALID) {
            fprintf(stderr, "rank 0: a shard with the wrong read count went unreported (%d)\n", rc);
            return 14;
        }
        if (rank != 0 && rc != FCD_OK) return 15;
    } else {
        if (rc != FCD_OK) {
            fprintf(stderr, "rank %d: fcd_comm_synchronize -> %d (%s)\n", rank, rc, fcd_last_error(g_h));
            return 16;
        }
        if (rank == 0) { /* the single-process decode of every read */
            int64_t i, j;
            if (alloc_result(&want, total, T)) return 11;
            CHECK(decode(g_h, 0, total, &want));
            for (i = 0; i < total; ++i) {
                if (all.out_len[i] != want.out_len[i] || all.status[i] != want.status[i]) {
                    fprintf(stderr, "read %ld: gathered (len %u, status %d), decoded here (len %u, status %d)\n", (long)i,
                            all.out_len[i], all.status[i], want.out_len[i], want.status[i]);
                    return 17;
                }
                for (j = 0; j < (int64_t)want.out_len[i]; ++j)
                    if (all.labels[i * dst_stride + j] != want.labels[i * T + j] || all.path[i * dst_stride + j] != want.path[i * T + j]) {
                        fprintf(stderr, "read %ld differs at position %ld\n", (long)i, (long)j);
                        return 18;
                    }
            }
            printf("comm_world: %d ranks, %ld reads, scenario %s: gathered == decoded in one process\n", world, (long)total, scenario);
        }
    }
    CHECK(fcd_comm_destroy(c));
    CHECK(fcd_destroy(g_h));
    return 0;
}

int main(int argc, char **argv) {
    uint8_t id[FCD_COMM_ID_BYTES];
    pid_t pids[64];
    int world, k, bad = 0;
    if (argc < 3 || (world = atoi(argv[1])) < 1 || world > 64) {
        fprintf(stderr, "usage: comm_world WORLD uneven|wide|mixed|badheader|allocfail|prepfail\n");
        return 2;
    }
    if (fcd_comm_unique_id(id) != FCD_OK) {
        fprintf(stderr, "fcd_comm_unique_id failed (FCD_RCCL_LIBRARY?)\n");
        return 3;
    }
    if (!strcmp(argv[2], "allocfail")) {
        char v[16];
        snprintf(v, sizeof v, "%d", world - 1);
        setenv("FCD_DEBUG_FAIL_GATHER_ALLOC", v, 1);
    }
    if (!strcmp(argv[2], "prepfail")) {
        char v[16];
        snprintf(v, sizeof v, "%d", world - 1);
        setenv("FCD_DEBUG_FAIL_GATHER_PREP", v, 1);
    }
    fflush(NULL);
    for (k = 0; k < world; ++k) {
        pids[k] = fork();
        if (pids[k] < 0) return 4;
        if (pids[k] == 0) {
            alarm(120); /* a rank left waiting in a collective dies here instead of hanging the suite */
            {
                const int rc = run_rank(world, k, id, argv[2]);
                fflush(NULL);
                _exit(rc);
            }
        }
    }
    for (k = 0; k < world; ++k) {
        int st = 0;
        if (waitpid(pids[k], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) {
            fprintf(stderr, "rank %d: %s %d\n", k, WIFSIGNALED(st) ? "killed by signal" : "exit status",
                    WIFSIGNALED(st) ? WTERMSIG(st) : WEXITSTATUS(st));
            bad = 1;
        }
    }
    if (!bad) printf("comm_world: scenario %s at world size %d ok\n", argv[2], world);
    return bad;
}
