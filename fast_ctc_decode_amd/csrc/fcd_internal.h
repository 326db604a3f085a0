// fcd_internal.h -- shared declarations of libfcd_hip.so (host side + kernel launchers).
// gfx950 / CDNA4 only.  No CUDA compatibility paths by design.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fcd.h"
#include "../../include/fcd_debug.h"

namespace fcd {

// Everything a search kernel needs to know about the input batch (device pointers).
struct BatchDesc {
    const float *post;
    const int64_t *lengths;  // nullable
    int64_t n_reads;
    int64_t T;
    int64_t stride_read, stride_t, stride_s, stride_n;
    int S;
    int N;
    int dtype;  // element type of `post`: 0 f32, 1 f16, 2 bf16 (strides are in elements of that type)
};

// Device-side output arrays (see fcd_result in include/fcd.h).
struct ResultDesc {
    uint8_t *labels;
    uint32_t *path;
    float *qual;
    uint32_t *out_len;
    int32_t *status;
    int64_t out_stride;
    uint32_t *ambiguous;  // nullable: per-read count of unpinned tie steps (beam searches only)
};

// Parameters of the 1D beam searches (search::beam_search / search::crf_beam_search).
struct BeamArgs {
    int beam_size;
    float thr;
    int collapse;
    int crf;
    const float *init;  // CRF only: [n_reads * init_stride]
    int64_t n_init;
    int64_t init_stride;
    int force_one_read_per_wave;  // wave kernel: skip the two-reads-per-wavefront variant
    // developer instrument (fcd_beam_search_profile_dev): per wavefront, shader cycles spent in each block
    // of the step, summed over the read -- [n_wavefronts][8] u32, nullable
    uint32_t *prof;
    int tie_order;  // FCD_TIE_PDQ178 / FCD_TIE_STABLE (include/fcd.h): order of equal probabilities above 20 candidates
};

// Per-chunk tree arena of the LDS-resident ("generic") beam kernel: one slab per read.
//   rec  : int4 {parent, time, label, depth} per node            (tree.rs LabelNode)
//   rows : NL int32 per node, child index or -1                  (tree.rs children: Vec2D<i32>)
struct GenericArena {
    int4 *rec;
    int32_t *rows;
    int64_t cap_nodes;  // nodes per read
};

// Tree arena of the register-resident ("wave") beam kernel.
//   rec  : i32 (parent + 1) << 3 | label per node; the creation time -- what `path` reports -- is the upper part
//          of the node id (wave kernel) or looked up in `first` (lane kernel, dense ids)
//   jmp  : i32 per node, written for nodes at depth % 64 == 0: next such ancestor (traceback)
//   rows : int4 per node (NL <= 4) or 8 x int32 (NL <= 6..7); entry = child | EVER bit, or -1
struct WaveArena {
    int32_t *rec;
    int32_t *jmp;
    int32_t *rows;
    int32_t *first;        // lane kernel: first[slab][t] = the read's node count when step t began
    int64_t first_stride;  // words per slab of `first`
    int64_t cap_nodes;
    int row_words;  // 4 or 8
    // retry pass of the lane kernel (one read per wavefront): only reads whose first-pass slab overflowed
    // (status FCD_ST_INTERNAL) run, in worst-case slabs; the counter says how many did
    int32_t *retry_counter;  // nullable
    // lane kernel, large jobs: the slabs (pairs of slabs at two reads per wavefront) come from a device-side pool,
    // taken when a wavefront starts and handed back after its traceback (slab_pool.h); null = slab i belongs to read i
    unsigned long long *pool;
};

size_t beam_generic_lds_bytes(int beam_size, int N, int tie_order);  // (the quicksort's list and scratch only under FCD_TIE_PDQ178 above 20 candidates)
hipError_t launch_beam_generic(const BatchDesc &in, int64_t read_begin, int64_t n_reads,
                               const BeamArgs &a, const GenericArena &arena, const ResultDesc &out,
                               hipStream_t stream);

int beam_lane_reads_per_wave(int beam_size);  // of a first pass (a retry pass: one)
// wavefronts of the instantiation the current device holds at once: the size of a slab pool nobody ever waits for
int beam_lane_resident_waves(int beam_size, int N, int crf, bool first_pass, bool ambiguous, int tie_order);
hipError_t slab_pool_init(unsigned long long *pool, int slabs, hipStream_t stream);  // `pool`: slab_pool::bytes(slabs) of device memory
bool beam_wave_supported(int beam_size, int N, int crf, int S);
// node ids of the wave kernel are (time step << shift) | index among the step's new nodes: slots per step = 1 << shift
int beam_wave_id_shift(int beam_size, int N, int force_one_read_per_wave);
hipError_t launch_beam_wave(const BatchDesc &in, int64_t read_begin, int64_t n_reads,
                            const BeamArgs &a, const WaveArena &arena, const ResultDesc &out,
                            hipStream_t stream);

// one beam entry per lane: beam_size <= 64, N <= 8 (CRF: N = 5, S a power of two >= 4); uses the wave arena layout
bool beam_lane_supported(int beam_size, int N, int crf, int S);
hipError_t launch_beam_lane(const BatchDesc &in, int64_t read_begin, int64_t n_reads,
                            const BeamArgs &a, const WaveArena &arena, const ResultDesc &out,
                            hipStream_t stream);

hipError_t launch_viterbi(const BatchDesc &in, int collapse, const ResultDesc &out,
                          hipStream_t stream);

hipError_t launch_crf_greedy(const BatchDesc &in, const float *init, int64_t n_init,
                             int64_t init_stride, const ResultDesc &out, hipStream_t stream);

// duplex::beam_search (src/duplex.rs:443-650) launch arguments; all pointers are device memory.
struct DuplexArgs {
    const float *ln1, *ln2;  // log-space posteriors [pair][Tcap][N]
    int64_t T1cap, T2cap;
    const int64_t *len1, *len2;
    const uint64_t *env;
    int64_t env_stride;
    int N, beam_size;
    float thr_ln;
    int collapse, mode;
    int S, crf;                  // CRF: S transition states, init scores per pair
    const float *init1, *init2;
    int64_t n_init1, n_init2, init1_stride, init2_stride;
    int4 *meta;
    float *nmax;
    int32_t *rlo;
    int32_t *rows;
    float *vec;
    float *rootgap;
    int64_t cap_nodes;
    int Wcap;
    int staged;
    ResultDesc out;
    uint32_t *prof;  // developer instrument: [pair][8] cycle account, nullable
    int tie_order;   // FCD_TIE_PDQ178 / FCD_TIE_STABLE
    // slot-resident kernel (duplex_slots.hip): `meta` is the base of the arena, ONE slab of `pair_stride` bytes per pair
    // (meta | aux | rings of Wcap floats | NLp child ids per node | the root's T2cap + 1 floats), below 4 GiB each
    int4 *aux;
    int NLp;
    int64_t pair_stride;
};

size_t duplex_lds_bytes(int beam_size, int N, int Wmax, int S, int tie_order);
// duplex_slots.hip: beam_size * N <= 64, N <= 8, and the rings of every live node + the read-2 tile fit 64 KiB of LDS
bool duplex_slots_supported(int beam_size, int N, int S, int width, int tie_order);
size_t duplex_slots_pair_bytes(int64_t cap_nodes, int N, int ring_rows, int64_t T2cap);  // one pair's slab
int duplex_slots_ring_rows(int width);  // ring capacity for a widest envelope row of `width`
size_t duplex_slots_lds_bytes(int beam_size, int N, int S, int ring_rows, int tie_order);
hipError_t launch_duplex_slots(const DuplexArgs &a, int64_t pair_begin, int64_t n_pairs, hipStream_t stream);
hipError_t launch_ln_convert(const float *x, int dtype, int64_t n_reads, int64_t T, int S, int N, int64_t s_read,
                             int64_t s_t, int64_t s_s, int64_t s_n, float *out, int glibc235, hipStream_t stream);
hipError_t launch_env_width(const uint64_t *env, int64_t n_pairs, int64_t env_stride, int64_t T1cap,
                            int64_t T2cap, const int64_t *len1, const int64_t *len2, int *out,
                            hipStream_t stream);
hipError_t launch_duplex(const DuplexArgs &a, int64_t pair_begin, int64_t n_pairs,
                         hipStream_t stream);
// alignment-band estimator (envelope.hip); all pointers are device memory
struct EnvelopeArgs {
    const uint8_t *labels1, *labels2;  // [pair][stride] label indices of the two reads
    const uint32_t *path1, *path2;     // [pair][stride] emission times of those labels
    const uint32_t *len1, *len2;       // labels per read
    int64_t stride1, stride2;
    const int64_t *T1, *T2;            // nullable per-pair row counts
    int64_t T1cap, T2cap;
    int64_t L2cap;                     // most labels any read 2 of the batch holds (sizes the LDS rows)
    int64_t band;
    uint64_t *env;                     // [pair][env_stride][2]
    int64_t env_stride;
    uint64_t *dirs;                    // workspace: [pair][dirs_stride] decision ballots
    int64_t dirs_stride;
    int nchunk;                        // 64-column chunks per DP row
    int32_t *anchor;                   // workspace: [pair][T1cap + 1]
};
size_t envelope_lds_bytes(int64_t L2cap);
hipError_t launch_max_u32(const uint32_t *a, const uint32_t *b, int64_t n, uint32_t *out2, hipStream_t stream);
hipError_t launch_envelope(const EnvelopeArgs &a, int64_t pair_begin, int64_t n_pairs, hipStream_t stream);

// compact wire format of decoded results (pack.hip)
hipError_t launch_result_offsets(const uint32_t *len, int64_t n, int64_t stride, uint64_t *offsets, hipStream_t stream);
// (res.qual non-null: an f32 region follows the path region, 4-byte aligned -- host result chunks only)
hipError_t launch_pack(const ResultDesc &res, int64_t n, int path_bytes, const uint64_t *offsets, uint8_t *buf,
                       hipStream_t stream);
hipError_t launch_unpack(const uint8_t *buf, int64_t n, const uint64_t *offsets, const ResultDesc &out,
                         hipStream_t stream);
// every shard of a gathered buffer (world x stride bytes) at once; first = [world + 1] prefix sums of the
// per-rank read counts (device), offsets = workspace of n_total + world u64, *bad = 1 + shard on a bad header
hipError_t launch_unpack_gathered(const uint8_t *gathered, int64_t stride, int world, const int64_t *first,
                                  int64_t n_total, uint64_t *offsets, const ResultDesc &out, int32_t *bad,
                                  hipStream_t stream);

hipError_t launch_logspace_probe(const float *a, const float *b, float *out_add, float *out_ln,
                                 int64_t n, int mode, hipStream_t stream);
hipError_t launch_glibc235_apply(int which, const float *x, float *y, int64_t n, hipStream_t stream);
hipError_t launch_logadd_chain(int n_chain, int mode, uint64_t *cycles, float *sink, hipStream_t stream);
hipError_t launch_logadd_sweep(int which, uint32_t first, uint32_t last, unsigned long long *counts, hipStream_t stream);
hipError_t launch_pdq178_probe(uint64_t *lists, int64_t n_lists, int64_t stride, const int32_t *lens, hipStream_t stream);
hipError_t launch_pdq178_coop_probe(uint64_t *lists, int64_t n_lists, int64_t stride, const int32_t *lens, int planes,
                                    int keep, hipStream_t stream);
hipError_t coop_prof_read(unsigned long long *out16, bool reset);  // developer instrument of the probe kernels
// the replay's std-form word (pdq178.h g_std_form) of each translation unit that holds a replay, on the current device
hipError_t beam_wave_set_pdq178_std_form(int bits);
hipError_t beam_lane_set_pdq178_std_form(int bits);
hipError_t beam_generic_set_pdq178_std_form(int bits);
hipError_t tieorder_set_pdq178_std_form(int bits);
hipError_t lane_tie_prof_read(unsigned long long *out16, bool reset);  // ... of a -DFCD_LANE_TIE_PROF build of beam_lane.hip
// the tie order searches on this handle use (capi.hip)
int effective_tie_order(const fcd_handle *h);

// ---- host side of the four 1D searches (capi.hip, hostjob.hip) ----
enum class HostOp { Viterbi, Beam, CrfBeam, CrfGreedy };

struct HostCall {
    HostOp op;
    int collapse = 1;
    int64_t beam_size = 5;
    float thr = 0.0f;
    int kernel = FCD_KERNEL_AUTO;
    const float *init = nullptr;  // host pointer
    int64_t n_init = 0, init_stride = 0;
};

// where one staged host call lives inside fcd_handle::stage (byte offsets)
struct HostStage {
    size_t o_in, o_len, o_init, o_lab, o_path, o_qual, o_olen, o_stat, o_amb, used;
    size_t n_in, n_out, n_init;
    int64_t B;
    bool want_amb, mirror;
};

int host_check(fcd_handle *h, const fcd_batch *in, const fcd_result *out, const HostCall &c);
int host_upload(fcd_handle *h, const fcd_batch *in, const fcd_result *shape, const HostCall &c, bool allow_mirror,
                HostStage *st, fcd_batch *din, fcd_result *dout);
int host_search(fcd_handle *h, const HostStage &st, const fcd_batch *din, const HostCall &c, const fcd_result *dout);
int host_download(fcd_handle *h, const HostStage &st, const fcd_result &dout, const fcd_result *out);
// large host batches: chunks on internal lanes, upload || search || packed download (hostjob.hip)
bool host_job_wanted(fcd_handle *h, const fcd_batch *in, const HostCall &c);
int host_job_run_fixed(fcd_handle *h, const fcd_batch *in, const fcd_result *out, const HostCall &c);
void host_job_release_lanes(fcd_handle *h, bool destroy);

}  // namespace fcd

struct fcd_host_lane;

struct fcd_handle {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // ring of (start, stop) event pairs, one per search call, recorded on the launch stream
    static constexpr int kTimingRing = 256;
    std::vector<hipEvent_t> ev0, ev1;
    int64_t n_timed = 0;  // calls recorded since the last fcd_timing_reset
    int64_t ws_limit = 0;  // 0 = auto (half of the free device memory)
    int first_pass_div = 2;  // lane kernel, two-pass sizing: first-pass slabs hold 1/div of the worst case
    bool first_pass_div_pinned = false;  // set by fcd_debug_set_first_pass_divisor: no adaptation
    // grow-only device workspace (tree arenas, staging for *_host calls)
    void *arena = nullptr;
    size_t arena_bytes = 0;
    size_t arena_region = 0;  // fcd_set_overlap: internal stream k's calls use the arena from k * arena_region on
    void *stage = nullptr;
    size_t stage_bytes = 0;
    void *pin = nullptr;  // page-locked host mirror of `stage` for small *_host calls (one DMA each way)
    size_t pin_bytes = 0;
    void *lnbuf = nullptr;  // duplex: log-space copies of both reads + scalars
    size_t lnbuf_bytes = 0;
    size_t lnbuf_region = 0;  // (fcd_set_overlap: as arena_region)
    // wide-beam kernel, large jobs: slabs handed out on the device (slab_pool.h).  An allocation of its own: calls in
    // flight on the overlap streams keep using it while other entry points size `arena` for themselves
    void *pool_arena = nullptr;
    size_t pool_arena_bytes = 0;
    void *pool_ctl = nullptr;  // the two rings: first-pass (pairs of) slabs, worst-case slabs of the retry pass
    size_t pool_ctl_bytes = 0;
    struct PoolGeom {
        int64_t cap_nodes = 0, cap_worst = 0, first_stride = 0;
        int rpw = 0, row_words = 0, p1 = 0, p2 = 0;
    } pool_geom;
    // fcd_set_overlap: wide-beam calls go round-robin to internal streams, ordered after the handle's stream as it stood
    // at the call and not after one another; what each stream still runs writes the output ranges listed here
    static constexpr int kMaxOverlap = 8;
    int overlap_n = 0;
    hipStream_t ov_stream[kMaxOverlap] = {};
    hipEvent_t ov_last[kMaxOverlap] = {};  // recorded behind the last call of the stream
    bool ov_used[kMaxOverlap] = {};
    hipEvent_t ov_fork = nullptr;
    uint64_t ov_seq = 0;
    int ov_last_slot = -1;  // the internal stream the latest overlapping call went to
    struct Range { uintptr_t lo, hi; };
    // the calls each internal stream has not finished yet, oldest first: what they write, and an event behind each (so that
    // a stream that never idles does not accumulate ranges: finished calls are dropped from the front)
    struct Flight {
        hipEvent_t done = nullptr;
        Range r[6];
        int n = 0;
    };
    std::deque<Flight> ov_flights[kMaxOverlap];
    std::vector<hipEvent_t> ov_event_pool;
    void *retry_counter = nullptr;  // lane kernel, two-pass sizing: overflow counter of the retry rounds
    size_t retry_counter_bytes = 0;
    void *retry_host = nullptr;     // page-locked word the first retry round's overflow count is copied to, read one call late
    hipEvent_t retry_ev = nullptr;  // ... recorded behind that copy
    bool retry_pending = false;
    int64_t retry_n = 0;            // reads of the chunk the pending count belongs to
    // chunk lanes of the pipelined host path (hostjob.hip): sub-handles with their own stream and workspace
    std::vector<fcd_host_lane *> lanes;
    bool job_active = false;  // a host job owns the lanes from begin to end
    bool is_lane = false;
    uint32_t *duplex_prof = nullptr;  // fcd_debug_set_duplex_profile
    int duplex_kernel = 0;            // fcd_debug_set_duplex_kernel: 0 automatic, 1 the any-shape kernel (duplex.hip), 2 the slot-resident one
    int tie_order = FCD_TIE_DEFAULT;  // fcd_set_tie_order; FCD_TIE_DEFAULT = follow the process default
    int pipe_lanes = 0;          // fcd_set_host_pipeline: 0 = default (FCD_HOST_LANES or 4)
    int64_t pipe_chunk = 0;      // reads per chunk, 0 = automatic
    int64_t pipe_min_bytes = -1; // fcd_*_host batches of at least this many input bytes take the pipeline; -1 = default
    std::string err;
    std::recursive_mutex mu;  // a *_host call holds it from staging to copy-back, the *_dev call inside re-enters
};
