/* librccl_stub.c -- TEST INFRASTRUCTURE: a stand-in for the five RCCL entry points csrc/comm.hip uses
 * (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllReduce, ncclGather; + ncclGetErrorString), over POSIX
 * shared memory between PROCESSES of one machine.  With the emulator build of the library (tests/hipemu: "device"
 * memory is host memory, streams run to completion) it lets the C-ABI multi-GPU step -- fcd_comm_create,
 * fcd_gather_results_dev, its size-agreement all-reduces, its late errors -- run at world sizes 2 and 8 without a GPU
 * (tests/capi/comm_world.c, tests/test_capi_comm_world.py).  The product never loads it: comm.hip finds RCCL with
 * dlopen, and the test points it here with FCD_RCCL_LIBRARY.
 *
 * Semantics kept: a collective returns when every rank of the communicator has entered it (two barriers around a
 * shared staging area), counts are per rank, the gather's receive buffer is only touched on the root.  Only the
 * datatype / operator combinations comm.hip uses are implemented (uint64 MAX all-reduce, uint8 gather). */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define STUB_ID_BYTES 128
#define STUB_MAX_WORLD 64
#define STUB_SLOT ((size_t)16 << 20) /* staging bytes per rank (sparse until touched) */

typedef struct { char internal[STUB_ID_BYTES]; } ncclUniqueId;

struct shared {
    volatile int ready;
    int world;
    pthread_barrier_t bar;
    size_t count[STUB_MAX_WORLD];
};

struct stub_comm {
    struct shared *sh;
    char *slots;
    size_t map_bytes;
    int world, rank;
    char name[STUB_ID_BYTES];
};
typedef struct stub_comm *ncclComm_t;

static size_t header_bytes(void) { return (sizeof(struct shared) + 4095) & ~(size_t)4095; }

int ncclGetUniqueId(ncclUniqueId *id) {
    static int counter = 0;
    struct timespec ts;
    if (!id) return 4;
    clock_gettime(CLOCK_REALTIME, &ts);
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/fcd_rccl_stub_%ld_%ld_%d", (long)getpid(), (long)ts.tv_nsec, counter++);
    return 0;
}

int ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    struct stub_comm *c;
    int fd, creator = 1;
    if (!comm || nranks < 1 || nranks > STUB_MAX_WORLD || rank < 0 || rank >= nranks) return 4;
    c = (struct stub_comm *)calloc(1, sizeof *c);
    if (!c) return 2;
    memcpy(c->name, id.internal, STUB_ID_BYTES);
    c->name[STUB_ID_BYTES - 1] = 0;
    c->world = nranks;
    c->rank = rank;
    c->map_bytes = header_bytes() + (size_t)nranks * STUB_SLOT;
    fd = shm_open(c->name, O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0 && errno == EEXIST) {
        creator = 0;
        fd = shm_open(c->name, O_RDWR, 0600);
    }
    if (fd < 0) { free(c); return 2; }
    if (creator && ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); free(c); return 2; }
    if (!creator) { /* the creator may not have sized the segment yet */
        struct stat st;
        int tries = 0;
        while (fstat(fd, &st) == 0 && (size_t)st.st_size < c->map_bytes && tries++ < 20000) usleep(100);
    }
    c->sh = (struct shared *)mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->sh == MAP_FAILED) { free(c); return 2; }
    c->slots = (char *)c->sh + header_bytes();
    if (creator) {
        pthread_barrierattr_t a;
        pthread_barrierattr_init(&a);
        pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&c->sh->bar, &a, (unsigned)nranks);
        pthread_barrierattr_destroy(&a);
        c->sh->world = nranks;
        __sync_synchronize();
        c->sh->ready = 1;
    } else {
        int tries = 0;
        while (!c->sh->ready && tries++ < 200000) usleep(50);
        if (!c->sh->ready || c->sh->world != nranks) { munmap(c->sh, c->map_bytes); free(c); return 3; }
    }
    pthread_barrier_wait(&c->sh->bar); /* like RCCL: returns once every rank has joined */
    *comm = c;
    return 0;
}

int ncclCommDestroy(ncclComm_t c) {
    if (!c) return 0;
    if (c->rank == 0) shm_unlink(c->name); /* (the mappings outlive the name) */
    munmap(c->sh, c->map_bytes);
    free(c);
    return 0;
}

/* datatype 5 = ncclUint64, op 2 = ncclMax */
int ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, int datatype, int op, ncclComm_t c, void *stream) {
    size_t i;
    int k;
    (void)stream;
    if (!c || datatype != 5 || op != 2 || count * 8 > STUB_SLOT) return 4;
    memcpy(c->slots + (size_t)c->rank * STUB_SLOT, sendbuff, count * 8);
    pthread_barrier_wait(&c->sh->bar);
    for (i = 0; i < count; ++i) {
        uint64_t m = 0;
        for (k = 0; k < c->world; ++k) {
            uint64_t v;
            memcpy(&v, c->slots + (size_t)k * STUB_SLOT + i * 8, 8);
            if (v > m) m = v;
        }
        memcpy((char *)recvbuff + i * 8, &m, 8);
    }
    pthread_barrier_wait(&c->sh->bar);
    return 0;
}

/* datatype 1 = ncclUint8; sendcount elements from every rank land at recvbuff + rank * sendcount on the root */
int ncclGather(const void *sendbuff, void *recvbuff, size_t sendcount, int datatype, int root, ncclComm_t c, void *stream) {
    int k, bad = 0;
    (void)stream;
    if (!c || datatype != 1 || sendcount > STUB_SLOT || root < 0 || root >= c->world) return 4;
    memcpy(c->slots + (size_t)c->rank * STUB_SLOT, sendbuff, sendcount);
    c->sh->count[c->rank] = sendcount;
    pthread_barrier_wait(&c->sh->bar);
    for (k = 0; k < c->world; ++k) bad |= c->sh->count[k] != sendcount; /* (a real gather would hang or corrupt) */
    if (c->rank == root && !bad)
        for (k = 0; k < c->world; ++k) memcpy((char *)recvbuff + (size_t)k * sendcount, c->slots + (size_t)k * STUB_SLOT, sendcount);
    pthread_barrier_wait(&c->sh->bar);
    return bad ? 5 : 0;
}

const char *ncclGetErrorString(int rc) {
    switch (rc) {
        case 0: return "no error";
        case 2: return "stub: system error";
        case 3: return "stub: rendezvous failed";
        case 4: return "stub: invalid argument / unsupported datatype";
        case 5: return "stub: the ranks passed different counts to a gather";
        default: return "stub: error";
    }
}
