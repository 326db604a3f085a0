/*
 * consumer.c -- a C99 translation unit against include/fcd.h, linked with the C-ABI library (libfcd_hip.so; in the CPU
 * suite: the emulator build of the same sources) and nothing else.  No Python anywhere in the process: this is what a
 * host in another language -- the Rust of /root/reference/src/lib.rs:353, through `extern "C"` -- would do.
 *
 *   consumer DIR
 *
 * DIR holds raw little-endian files written by tests/test_capi_consumer.py from tests/golden/vectors.npz:
 *   cases.txt                       one line per beam-search case: name T N beam thr collapse status n_labels
 *   <name>.x.f32                    posteriors (T x N float32)
 *   <name>.labels.u8 / .path.u32    the expected result
 *   viterbi.x.f32 (+ .labels.u8 .path.u32), listed in cases.txt with beam 0
 * Every case is decoded three ways and compared byte for byte with the expected result:
 *   fcd_beam_search_host                     one blocking call (src/search.rs:159-165 through the batch entry point)
 *   fcd_beam_search_host_begin / fcd_job_*   the same reads as a stream of result chunks (all cases of one shape at once)
 *   fcd_viterbi_search_host                  (src/search.rs:320-327)
 * Exit code 0 and "consumer: N cases ok" when everything matches.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fcd.h"

#define MAX_CASES 32

typedef struct {
    char name[64];
    int T, N, beam, collapse, status, n_labels;
    float thr;
    float *x;
    uint8_t *labels;
    uint32_t *path;
} test_case;

static void *slurp(const char *dir, const char *name, const char *ext, size_t bytes) {
    char path[1024];
    FILE *f;
    void *buf = malloc(bytes ? bytes : 1);
    snprintf(path, sizeof path, "%s/%s.%s", dir, name, ext);
    f = fopen(path, "rb");
    if (!f || !buf || fread(buf, 1, bytes, f) != bytes) {
        fprintf(stderr, "consumer: cannot read %zu bytes of %s\n", bytes, path);
        exit(2);
    }
    fclose(f);
    return buf;
}

static int check(const char *what, const test_case *c, int status, uint32_t n, const uint8_t *labels, const uint32_t *path) {
    if (status != c->status) {
        fprintf(stderr, "consumer: %s %s: status %d, expected %d\n", what, c->name, status, c->status);
        return 1;
    }
    if (status != FCD_ST_OK) return 0;
    if ((int)n != c->n_labels || memcmp(labels, c->labels, n) != 0 || memcmp(path, c->path, 4 * (size_t)n) != 0) {
        fprintf(stderr, "consumer: %s %s: result differs from the golden vector (%u labels, expected %d)\n", what, c->name,
                (unsigned)n, c->n_labels);
        return 1;
    }
    return 0;
}

int main(int argc, char **argv) {
    test_case cases[MAX_CASES];
    int n_cases = 0, bad = 0, i;
    char line[512], path[1024];
    FILE *f;
    fcd_handle *h = NULL;
    if (argc < 2) {
        fprintf(stderr, "usage: consumer DIR\n");
        return 2;
    }
    snprintf(path, sizeof path, "%s/cases.txt", argv[1]);
    f = fopen(path, "r");
    if (!f) {
        fprintf(stderr, "consumer: no %s\n", path);
        return 2;
    }
    while (n_cases < MAX_CASES && fgets(line, sizeof line, f)) {
        test_case *c = &cases[n_cases];
        if (sscanf(line, "%63s %d %d %d %f %d %d %d", c->name, &c->T, &c->N, &c->beam, &c->thr, &c->collapse, &c->status,
                   &c->n_labels) != 8)
            continue;
        c->x = (float *)slurp(argv[1], c->name, "x.f32", sizeof(float) * (size_t)c->T * (size_t)c->N);
        c->labels = (uint8_t *)slurp(argv[1], c->name, "labels.u8", (size_t)c->n_labels);
        c->path = (uint32_t *)slurp(argv[1], c->name, "path.u32", 4 * (size_t)c->n_labels);
        ++n_cases;
    }
    fclose(f);
    if (fcd_version() != FCD_VERSION_MAJOR * 1000 + FCD_VERSION_MINOR) {
        fprintf(stderr, "consumer: header and library disagree about the version\n");
        return 1;
    }
    if (fcd_create(0, &h) != FCD_OK) {
        fprintf(stderr, "consumer: fcd_create failed (no usable device: the library has no CPU path)\n");
        return 3;
    }
    if (fcd_get_tie_order(h) != FCD_TIE_PDQ178) {  /* the golden vectors follow the default order */
        fprintf(stderr, "consumer: unexpected default tie order\n");
        return 1;
    }
    for (i = 0; i < n_cases; ++i) {
        const test_case *c = &cases[i];
        fcd_batch in;
        fcd_result out;
        uint8_t *labels = (uint8_t *)malloc((size_t)c->T + 1);
        uint32_t *pth = (uint32_t *)malloc(4 * ((size_t)c->T + 1));
        uint32_t out_len = 0;
        int32_t status = -1;
        int rc;
        memset(&in, 0, sizeof in);
        memset(&out, 0, sizeof out);
        in.post = c->x;
        in.n_reads = 1;
        in.T = c->T;
        in.S = 1;
        in.N = c->N;
        in.stride_read = (int64_t)c->T * c->N;
        in.stride_t = c->N;
        in.stride_s = 0;
        in.stride_n = 1;
        in.dtype = FCD_DTYPE_F32;
        out.labels = labels;
        out.path = pth;
        out.out_len = &out_len;
        out.status = &status;
        out.out_stride = c->T > 0 ? c->T : 1;
        if (c->beam == 0) {
            rc = fcd_viterbi_search_host(h, &in, c->collapse, &out);
            if (rc != FCD_OK) {
                fprintf(stderr, "consumer: fcd_viterbi_search_host: %d %s\n", rc, fcd_last_error(h));
                return 1;
            }
            bad += check("fcd_viterbi_search_host", c, 0, out_len, labels, pth);
        } else {
            fcd_job *job = NULL;
            fcd_chunk ch;
            int64_t n_chunks, chunk_reads = 0;
            int n_lanes = 0, got = 0;
            rc = fcd_beam_search_host(h, &in, c->beam, c->thr, c->collapse, FCD_KERNEL_AUTO, &out);
            if (rc != FCD_OK) {
                fprintf(stderr, "consumer: fcd_beam_search_host: %d %s\n", rc, fcd_last_error(h));
                return 1;
            }
            bad += check("fcd_beam_search_host", c, status, out_len, labels, pth);
            /* the same read as a job: result chunks with the used prefixes only */
            rc = fcd_beam_search_host_begin(h, &in, c->beam, c->thr, c->collapse, FCD_KERNEL_AUTO, FCD_JOB_PATH, &job);
            if (rc != FCD_OK) {
                fprintf(stderr, "consumer: fcd_beam_search_host_begin: %d %s\n", rc, fcd_last_error(h));
                return 1;
            }
            n_chunks = fcd_job_chunks(job, &chunk_reads, &n_lanes);
            if (n_chunks != 1 || fcd_destroy(h) != FCD_E_INVALID) { /* (a handle with an open job cannot be destroyed) */
                fprintf(stderr, "consumer: job of one read: %lld chunks\n", (long long)n_chunks);
                return 1;
            }
            while ((rc = fcd_job_next(job, &ch)) == FCD_OK) {
                uint32_t tmp[8192];
                uint32_t k, n = ch.out_len[0];
                const uint64_t off = ch.offsets[0];
                if (ch.n_reads != 1 || ch.read_begin != 0 || n > 8192 || (ch.path_bytes != 2 && ch.path_bytes != 4)) {
                    fprintf(stderr, "consumer: unexpected chunk shape\n");
                    return 1;
                }
                for (k = 0; k < n; ++k)
                    tmp[k] = ch.path_bytes == 2 ? ((const uint16_t *)ch.path)[off + k] : ((const uint32_t *)ch.path)[off + k];
                bad += check("fcd_job_next", c, ch.status[0], n, ch.labels + off, tmp);
                ++got;
            }
            if (rc != FCD_JOB_DONE || got != 1 || fcd_job_end(job) != FCD_OK) {
                fprintf(stderr, "consumer: job stream ended with %d after %d chunks\n", rc, got);
                return 1;
            }
        }
        free(labels);
        free(pth);
    }
    /* the other order is selectable and, on these vectors (no tie among more than 20 candidates), changes nothing */
    if (fcd_set_tie_order(h, FCD_TIE_STABLE) != FCD_OK || fcd_get_tie_order(h) != FCD_TIE_STABLE ||
        fcd_set_tie_order(h, 7) != FCD_E_INVALID || fcd_set_tie_order(h, FCD_TIE_DEFAULT) != FCD_OK) {
        fprintf(stderr, "consumer: fcd_set_tie_order misbehaves\n");
        return 1;
    }
    if (strcmp(fcd_status_string(FCD_ST_RAN_OUT_OF_BEAM), "Ran out of search space (beam_cut_threshold too high)") != 0) {
        fprintf(stderr, "consumer: status text differs from the reference's (src/lib.rs:46-53)\n");
        return 1;
    }
    if (fcd_destroy(h) != FCD_OK) return 1;
    if (bad) return 1;
    printf("consumer: %d cases ok\n", n_cases);
    return 0;
}
