// viterbi.hip -- greedy decoders: search::viterbi_search (/root/reference/src/search.rs:303-383)
// and search::crf_greedy_search (:385-423), one read per wavefront.
//
// viterbi: the wave walks the read in tiles of 64 rows, one row per lane.  Per row: first-maximum
// argmax with the reference's strict '>' fold (:303-318); emission test against the previous
// row's label (lane-1 via a wave shuffle, lane 0 from the carry); wave-level stream compaction of
// the emissions with ballot + prefix popcount, so labels / path are written coalesced.  The
// optional quality output reproduces the reference's per-run f32 running sum IN ROW ORDER
// (:348-376) -- a tree reduction would round differently -- by letting every run head absorb one
// following row per iteration.
#include "device_utils.h"
#include "fcd_internal.h"

namespace fcd {

namespace {

constexpr int kWavesPerBlock = 4;

// Running state of one read's scan, carried across 64-row sub-tiles.
struct ScanState {
    int n_out = 0;          // emissions so far (wave-uniform)
    int carry_label = -1;   // label of the last row of the previous sub-tile (None)
    float run_total = 0.0f; // open quality run carried across sub-tiles (:337-339)
    int run_count = 0;
};

// One sub-tile: lane holds (label, prob) of row base+lane; emits, compacts, accumulates quality.
__device__ __forceinline__ void scan_subtile(ScanState &st, int label, float prob, bool act,
                                             int64_t base, int64_t T, int collapse, uint8_t *lab,
                                             uint32_t *pth, float *qual) {
    const int lane = threadIdx.x & 63;
    const int64_t row = base + lane;
    int prev = __shfl_up(label, 1);
    if (lane == 0) prev = st.carry_label;
    const bool emit = act && label != 0 && (!collapse || prev != label);  // :347
    const uint64_t m_emit = ballot(emit);
    const int my_out = st.n_out + popc64(m_emit & lanemask_lt());
    if (emit) {
        lab[my_out] = (uint8_t)label;
        if (pth) pth[my_out] = (uint32_t)row;
    }
    if (qual) {
        // Each emission opens a run that owns every following non-blank row up to the next
        // emission; its mean probability is what the reference feeds to phred().
        const bool nonblank = act && label != 0;
        const float contrib = nonblank ? prob : 0.0f;  // adding +0.0 is exact
        const int cnt1 = nonblank ? 1 : 0;
        // (a) the run carried in from the previous sub-tile absorbs rows [0, first emission)
        const int first_emit = m_emit ? __builtin_ctzll(m_emit) : kWave;
        const int tile_rows = (int)((T - base) < kWave ? (T - base) : kWave);
        const int lim = first_emit < tile_rows ? first_emit : tile_rows;
        for (int j = 0; j < lim; ++j) {
            st.run_total += __shfl(contrib, j);
            st.run_count += __shfl(cnt1, j);
        }
        if (m_emit && st.run_count > 0) {
            if (lane == 0) qual[st.n_out - 1] = st.run_total / (float)st.run_count;
            st.run_total = 0.0f;
            st.run_count = 0;
        }
        // (b) runs that start in this sub-tile: every head absorbs one more row per iteration
        float acc = contrib;  // the emission row itself (:362-365)
        int cnt = cnt1;
        bool alive = emit;
        float sh_c = contrib;
        int sh_n = cnt1;
        bool sh_e = emit;
        bool sh_a = act;
        for (int d = 1; d < kWave; ++d) {
            sh_c = __shfl_down(sh_c, 1);
            sh_n = __shfl_down(sh_n, 1);
            sh_e = __shfl_down((int)sh_e, 1) != 0;
            sh_a = __shfl_down((int)sh_a, 1) != 0;
            const bool in_tile = lane + d < kWave && sh_a;
            if (alive && (!in_tile || sh_e)) alive = false;
            if (alive) {
                acc += sh_c;
                cnt += sh_n;
            }
            if (ballot(alive) == 0ull) break;
        }
        // a head whose run is closed by a later emission in this sub-tile writes its mean now;
        // the last head's run stays open and becomes the carry
        const uint64_t later = m_emit & ~(lanemask_lt() | (1ull << lane));
        if (emit && later) qual[my_out] = acc / (float)cnt;
        if (m_emit) {
            const int last = 63 - __builtin_clzll(m_emit);
            st.run_total = __shfl(acc, last);
            st.run_count = __shfl(cnt, last);
        }
    }
    st.n_out += popc64(m_emit);
    const int last_lane = (int)((T - base) < kWave ? (T - base - 1) : (kWave - 1));
    st.carry_label = __shfl(label, last_lane);
}

// The same for a sub-tile whose 64 rows all exist, without quality values: the neighbour's label through a DPP
// wave shift (no LDS round trip), 32-bit output offsets against wave-uniform bases -- a third of the instructions.
__device__ __forceinline__ void scan_subtile_full(ScanState &st, int label, uint32_t row, int collapse, uint8_t *lab,
                                                  uint32_t *pth) {
    // wave_shr:1 -- lane 0 has no source lane and keeps `old`, the label carried over from the previous sub-tile
    const int prev = __builtin_amdgcn_update_dpp(st.carry_label, label, 0x138, 0xf, 0xf, false);
    const bool emit = label != 0 && (!collapse || prev != label);  // :347
    const uint64_t m_emit = ballot(emit);
    const uint32_t my_out = (uint32_t)st.n_out + (uint32_t)popc64(m_emit & lanemask_lt());
    if (emit) {
        lab[my_out] = (uint8_t)label;
        if (pth) pth[my_out] = row;
    }
    st.n_out += popc64(m_emit);
    st.carry_label = __builtin_amdgcn_readlane(label, 63);
}

__device__ __forceinline__ void scan_finish(ScanState &st, int64_t r, float *qual,
                                            const ResultDesc &out) {
    const int lane = threadIdx.x & 63;
    if (qual && st.run_count > 0 && lane == 0)
        qual[st.n_out - 1] = st.run_total / (float)st.run_count;  // :370-376
    if (lane == 0) {
        out.out_len[r] = (uint32_t)st.n_out;
        if (out.status) out.status[r] = FCD_ST_OK;
    }
}

// Generic path: any N, any strides; one row per lane, strided 4-byte loads.
__global__ __launch_bounds__(64 * kWavesPerBlock) void viterbi_kernel(BatchDesc in, int collapse,
                                                                    ResultDesc out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (r >= in.n_reads) return;
    int64_t T = in.T;
    if (in.lengths) {
        int64_t t = in.lengths[r];
        T = t < 0 ? 0 : (t < T ? t : T);
    }
    const int N = in.N;
    const int dt = in.dtype;
    const float *post = post_at(in.post, r * in.stride_read, dt);
    uint8_t *lab = out.labels + r * out.out_stride;
    uint32_t *pth = out.path ? out.path + r * out.out_stride : nullptr;
    float *qual = out.qual ? out.qual + r * out.out_stride : nullptr;
    ScanState st;
    for (int64_t base = 0; base < T; base += kWave) {
        const int64_t row = base + lane;
        const bool act = row < T;
        int label = 0;
        float prob = 0.0f;
        if (act) {
            const float *pr = post_at(post, row * in.stride_t, dt);
            prob = load_post(pr, 0, dt);
            for (int j = 1; j < N; ++j) {  // find_max: strict '>' keeps the first maximum
                const float v = load_post(pr, j * in.stride_n, dt);
                if (v > prob) {
                    prob = v;
                    label = j;
                }
            }
        }
        scan_subtile(st, label, prob, act, base, T, collapse, lab, pth, qual);
    }
    scan_finish(st, r, qual, out);
}

// Streaming path for C-contiguous reads (stride_n == 1, stride_t == N, 16-byte aligned reads):
// the wave pulls a 256-row tile with N fully coalesced 16-byte loads per lane (1 KiB per wave
// instruction), parks it in LDS and reads it back one row per lane (row stride N dwords: conflict
// free for odd N), so HBM sees only wide contiguous requests.  The next tile's loads are issued
// before the current tile is scanned.
constexpr int kTileRows = 256;

// DT: the posteriors' element type.  Half-precision reads (what basecaller networks emit; the reference forces a
// host float32 copy, src/lib.rs:182) stream at HALF the bytes: a 16-byte load brings eight elements, which are
// converted in registers -- exactly -- and parked in the same f32 tile, so everything after the tile is shared.
template <int N, int DT>
__global__ __launch_bounds__(64 * kWavesPerBlock) void viterbi_stream_kernel(BatchDesc in,
                                                                           int collapse,
                                                                           ResultDesc out) {
    constexpr int EPL = DT == kF32 ? 4 : 8;                             // elements per 16-byte load
    constexpr int NLOAD = (kTileRows * N + 64 * EPL - 1) / (64 * EPL);  // 16-byte loads per lane and tile
    // (f32: NLOAD = N and the tile is covered exactly; 16-bit, odd N: the last load is half used)
    __shared__ __attribute__((aligned(16))) float s_tile[kWavesPerBlock][NLOAD * 64 * EPL];
    const int lane = threadIdx.x & 63;
    // (wave-uniform, and said so: the read's index, its length and every pointer derived from them live in scalar registers)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t r = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    if (r >= in.n_reads) return;
    int64_t T = in.T;
    if (in.lengths) {
        int64_t t = in.lengths[r];
        T = t < 0 ? 0 : (t < T ? t : T);
    }
    const float *post = post_at(in.post, r * in.stride_read, DT);
    uint8_t *lab = out.labels + r * out.out_stride;
    uint32_t *pth = out.path ? out.path + r * out.out_stride : nullptr;
    float *qual = out.qual ? out.qual + r * out.out_stride : nullptr;
    float *tile = s_tile[wave];
    const int64_t total = T * N;  // elements in this read

    auto fetch = [&](int64_t tile_row0, uint4 (&v)[NLOAD]) {
        const int64_t f0 = tile_row0 * N;
#pragma unroll
        for (int m = 0; m < NLOAD; ++m) {
            const int64_t f = f0 + (int64_t)(lane + 64 * m) * EPL;
            const bool in_tile = (lane + 64 * m) * EPL < kTileRows * N;
            if (in_tile && f + EPL - 1 < total) {
                v[m] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(post) + f * (DT == kF32 ? 4 : 2));
            } else {  // the ragged end of the read: element by element, zeros beyond it
                uint32_t w[4] = {0u, 0u, 0u, 0u};
                if (in_tile) {
                    if (DT == kF32) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (f + e < total) w[e] = __float_as_uint(post[f + e]);
                    } else {
                        const uint16_t *ph = reinterpret_cast<const uint16_t *>(post);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (f + e < total) w[e >> 1] |= (uint32_t)ph[f + e] << (16 * (e & 1));
                    }
                }
                v[m] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    };
    auto park = [&](const uint4 (&v)[NLOAD]) {
#pragma unroll
        for (int m = 0; m < NLOAD; ++m) {
            float *dst = tile + (lane + 64 * m) * EPL;
            if (DT == kF32) {
                *reinterpret_cast<uint4 *>(dst) = v[m];
            } else {
                const uint32_t w[4] = {v[m].x, v[m].y, v[m].z, v[m].w};
                float4 lo4, hi4;
                if (DT == kF16) {
                    lo4 = make_float4(f16_bits_to_f32((uint16_t)w[0]), f16_bits_to_f32((uint16_t)(w[0] >> 16)),
                                      f16_bits_to_f32((uint16_t)w[1]), f16_bits_to_f32((uint16_t)(w[1] >> 16)));
                    hi4 = make_float4(f16_bits_to_f32((uint16_t)w[2]), f16_bits_to_f32((uint16_t)(w[2] >> 16)),
                                      f16_bits_to_f32((uint16_t)w[3]), f16_bits_to_f32((uint16_t)(w[3] >> 16)));
                } else {
                    lo4 = make_float4(__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xFFFF0000u),
                                      __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xFFFF0000u));
                    hi4 = make_float4(__uint_as_float(w[2] << 16), __uint_as_float(w[2] & 0xFFFF0000u),
                                      __uint_as_float(w[3] << 16), __uint_as_float(w[3] & 0xFFFF0000u));
                }
                *reinterpret_cast<float4 *>(dst) = lo4;
                *reinterpret_cast<float4 *>(dst + 4) = hi4;
            }
        }
    };

    // a tile that lies inside the read entirely: no bounds to check (256 * N elements are a whole number of loads'
    // in-tile lanes for every element type)
    auto fetch_full = [&](int64_t tile_row0, uint4 (&v)[NLOAD]) {
        const char *src = reinterpret_cast<const char *>(post) + tile_row0 * N * (DT == kF32 ? 4 : 2) + lane * 16;
#pragma unroll
        for (int m = 0; m < NLOAD; ++m) {
            if ((lane + 64 * m) * EPL < kTileRows * N) v[m] = *reinterpret_cast<const uint4 *>(src + m * 1024);
            else v[m] = make_uint4(0u, 0u, 0u, 0u);
        }
    };

    ScanState st;
    uint4 cur[NLOAD], nxt[NLOAD];
    if (T >= kTileRows) fetch_full(0, cur);
    else if (T > 0) fetch(0, cur);
    int64_t base = 0;
    // ---- whole tiles: nothing to test per row ----
    for (; base + kTileRows <= T; base += kTileRows) {
        const int64_t nb = base + kTileRows;
        if (nb + kTileRows <= T) fetch_full(nb, nxt);
        else if (nb < T) fetch(nb, nxt);
        park(cur);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int u = 0; u < kTileRows / 64; ++u) {
            const float *pr = tile + (64 * u + lane) * N;
            float prob = pr[0];
            int label = 0;
#pragma unroll
            for (int j = 1; j < N; ++j) {  // find_max: strict '>' keeps the first maximum
                const float v = pr[j];
                if (v > prob) {
                    prob = v;
                    label = j;
                }
            }
            if (qual) scan_subtile(st, label, prob, true, base + 64 * u, T, collapse, lab, pth, qual);
            else scan_subtile_full(st, label, (uint32_t)(base + 64 * u) + (uint32_t)lane, collapse, lab, pth);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < NLOAD; ++m) cur[m] = nxt[m];
    }
    // ---- the ragged last tile ----
    for (; base < T; base += kTileRows) {
        if (base + kTileRows < T) fetch(base + kTileRows, nxt);
        park(cur);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int u = 0; u < kTileRows / 64; ++u) {
            const int64_t sub = base + 64 * u;
            if (sub >= T) break;
            const bool act = sub + lane < T;
            const float *pr = tile + (64 * u + lane) * N;
            float prob = pr[0];
            int label = 0;
#pragma unroll
            for (int j = 1; j < N; ++j) {  // find_max: strict '>' keeps the first maximum
                const float v = pr[j];
                if (v > prob) {
                    prob = v;
                    label = j;
                }
            }
            if (!act) {
                label = 0;
                prob = 0.0f;
            }
            scan_subtile(st, label, prob, act, sub, T, collapse, lab, pth, qual);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < NLOAD; ++m) cur[m] = nxt[m];
    }
    scan_finish(st, r, qual, out);
}

// Time-major storage: element (read r, row t, column c) at post[t * stride_t + r * N + c] -- the (T, B, N) tensor a
// basecaller network emits, handed over as a (B, T, N) batch by its strides, no transposition.  A row of one read is
// N elements, but a row of EIGHT neighbouring reads is 8 * N contiguous elements: a workgroup of four wavefronts takes
// 8 reads, pulls 64-row tiles of them with 16-byte loads (each time step of the group is one contiguous run), parks the
// tile in LDS as [step][read][column] with a pitch of 8 * N + 1 words -- so that the row-per-lane read back, lanes one
// time step apart, hits all banks -- and every wavefront scans two of the reads exactly as the read-major kernel does.
constexpr int kTmReads = 8;
constexpr int kTmSteps = 64;

template <int N, int DT>
__global__ __launch_bounds__(256) void viterbi_tm_kernel(BatchDesc in, int collapse, ResultDesc out) {
    constexpr int EPL = DT == kF32 ? 4 : 8;      // elements per 16-byte load
    constexpr int ESZ = DT == kF32 ? 4 : 2;
    constexpr int G = kTmReads * N;               // elements of one time step of the group (a whole number of 16-byte units)
    constexpr int U = G / EPL;                    // 16-byte units per time step
    constexpr int UNITS = kTmSteps * U;           // per tile
    constexpr int NLD = (UNITS + 255) / 256;      // loads per thread and tile
    constexpr int PITCH = G + 1;
    __shared__ float s_tile[kTmSteps * PITCH];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Neighbouring groups share cache lines (a group's run of 8 * N elements is not a whole number of 128-byte lines),
    // and consecutive workgroup ids go round the eight XCDs, each with its own L2: hand every XCD a CONTIGUOUS range of
    // groups, so that a shared line is fetched into one L2, not two.
    const unsigned n_groups = gridDim.x, per_xcd = (n_groups + 7) / 8;
    unsigned gidx = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (n_groups < 64 || gidx >= n_groups || (n_groups & 7)) gidx = blockIdx.x;  // (small or ragged launches: as they come)
    const int64_t r0 = (int64_t)gidx * kTmReads;  // (the launch covers whole groups only)
    const int64_t T = in.T;
    const char *base = reinterpret_cast<const char *>(in.post) + r0 * in.stride_read * ESZ;
    const int64_t pitch_b = in.stride_t * ESZ;

    auto fetch = [&](int64_t t0, uint4 (&v)[NLD]) {
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int u = tid + 256 * m;
            const int sidx = u / U, o = u - sidx * U;
            v[m] = make_uint4(0u, 0u, 0u, 0u);
            if (u < UNITS && t0 + sidx < T)
                v[m] = *reinterpret_cast<const uint4 *>(base + (t0 + sidx) * pitch_b + o * 16);
        }
    };
    auto park = [&](const uint4 (&v)[NLD]) {
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int u = tid + 256 * m;
            if (u >= UNITS) continue;
            const int sidx = u / U, o = u - sidx * U;
            float *dst = s_tile + sidx * PITCH + o * EPL;  // (the odd pitch leaves only 4-byte alignment)
            const uint32_t w[4] = {v[m].x, v[m].y, v[m].z, v[m].w};
            if (DT == kF32) {
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[e] = __uint_as_float(w[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint16_t h = (uint16_t)(w[e >> 1] >> (16 * (e & 1)));
                    dst[e] = DT == kF16 ? f16_bits_to_f32(h) : bf16_bits_to_f32(h);
                }
            }
        }
    };

    constexpr int RPW = kTmReads / 4;  // reads per wavefront
    ScanState st[RPW];
    int64_t Tr[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int64_t r = r0 + wave * RPW + i;
        int64_t t = T;
        if (in.lengths) {
            const int64_t tl = in.lengths[r];
            t = tl < 0 ? 0 : (tl < T ? tl : T);
        }
        Tr[i] = t;
    }
    uint4 cur[NLD], nxt[NLD];
    if (T > 0) fetch(0, cur);
    for (int64_t t0 = 0; t0 < T; t0 += kTmSteps) {
        if (t0 + kTmSteps < T) fetch(t0 + kTmSteps, nxt);
        park(cur);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            if (t0 >= Tr[i]) continue;  // (wave-uniform)
            const int rr = wave * RPW + i;
            const int64_t r = r0 + rr;
            const float *pr = s_tile + lane * PITCH + rr * N;
            float prob = pr[0];
            int label = 0;
#pragma unroll
            for (int j = 1; j < N; ++j) {  // find_max: strict '>' keeps the first maximum
                const float v = pr[j];
                if (v > prob) {
                    prob = v;
                    label = j;
                }
            }
            uint8_t *lab = out.labels + r * out.out_stride;
            uint32_t *pth = out.path ? out.path + r * out.out_stride : nullptr;
            float *qual = out.qual ? out.qual + r * out.out_stride : nullptr;
            if (!qual && t0 + kTmSteps <= Tr[i]) {
                scan_subtile_full(st[i], label, (uint32_t)t0 + (uint32_t)lane, collapse, lab, pth);
            } else {
                const bool act = t0 + lane < Tr[i];
                if (!act) {
                    label = 0;
                    prob = 0.0f;
                }
                scan_subtile(st[i], label, prob, act, t0, Tr[i], collapse, lab, pth, qual);
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < NLD; ++m) cur[m] = nxt[m];
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int64_t r = r0 + wave * RPW + i;
        scan_finish(st[i], r, out.qual ? out.qual + r * out.out_stride : nullptr, out);
    }
}

// crf_greedy_search (:385-423): the state walk is a serial dependency; one wave per read,
// lanes 0..N-1 hold the current state's row and reduce it with a first-maximum argmax.
__global__ __launch_bounds__(64) void crf_greedy_kernel(BatchDesc in, const float *init_all,
                                                      int64_t n_init, int64_t init_stride,
                                                      ResultDesc out) {
    const int lane = threadIdx.x;
    const int64_t r = blockIdx.x;
    int64_t T = in.T;
    if (in.lengths) {
        int64_t t = in.lengths[r];
        T = t < 0 ? 0 : (t < T ? t : T);
    }
    const int N = in.N, S = in.S, n_base = N - 1;
    const int dt = in.dtype;
    const float *post = post_at(in.post, r * in.stride_read, dt);
    uint8_t *lab = out.labels + r * out.out_stride;
    uint32_t *pth = out.path ? out.path + r * out.out_stride : nullptr;
    float *qual = out.qual ? out.qual + r * out.out_stride : nullptr;

    // init_state.argmax() :399 (first maximum; NaN -> the reference panics)
    const float *init = init_all + r * init_stride;
    int state = 0;
    bool bad = n_init <= 0;
    if (!bad) {
        float m = init[0];
        bad = m != m;
        for (int64_t j = 1; j < n_init && !bad; ++j) {
            const float e = init[j];
            if (e != e) bad = true;
            if (e > m) {
                m = e;
                state = (int)j;
            }
        }
    }
    int n_out = 0;
    for (int64_t t = 0; t < T && !bad; ++t) {
        if (state < 0 || state >= S) {
            bad = true;
            break;
        }
        const float *pr = post_at(post, t * in.stride_t + (int64_t)state * in.stride_s, dt);
        // argmax over N columns, N may exceed 64: strided per-lane first-max, then wave reduce
        float best = 0.0f;
        int arg = 1 << 30;
        bool nan = false;
        for (int j = lane; j < N; j += kWave) {
            const float v = load_post(pr, j * in.stride_n, dt);
            nan = nan || (v != v);
            if (arg == (1 << 30) || v > best) {
                best = v;
                arg = j;
            }
        }
        if (__ballot(nan) != 0ull) {
            bad = true;
            break;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off);
            const int oa = __shfl_xor(arg, off);
            if (oa != (1 << 30) && (arg == (1 << 30) || ob > best || (ob == best && oa < arg))) {
                best = ob;
                arg = oa;
            }
        }
        if (arg > 0) {  // :409-416
            if (lane == 0) {
                lab[n_out] = (uint8_t)arg;
                if (pth) pth[n_out] = (uint32_t)t;
                if (qual) qual[n_out] = best;
            }
            n_out++;
            state = (int)(((int64_t)state * n_base) % S) + (arg - 1);
        }
    }
    if (lane == 0) {
        out.out_len[r] = bad ? 0u : (uint32_t)n_out;
        if (out.status) out.status[r] = bad ? FCD_ST_BAD_STATE : FCD_ST_OK;
    }
}


// crf_greedy_search for small state counts (S <= 8), streamed at HBM rate.
//
// The state walk looks serial -- row t is read at the state row t-1 left behind -- but over S states
// row t is just a function f_t : state -> state (with one extra absorbing value for "out of range",
// where the reference aborts).  So: every lane takes one row of a 64-row tile, evaluates the
// first-maximum argmax for ALL S states of its row, packs f_t into 4-bit entries of a 64-bit word,
// and an inclusive wave scan of function COMPOSITION (six shuffle steps) yields the state every row is
// entered with.  The visited state then selects each row's label / probability / NaN flag, and the
// emissions are compacted with ballot + prefix popcount like viterbi_search's.  Tiles arrive through
// LDS with 16-byte coalesced loads; the next tile is in flight while the current one is scanned.
// A state -> state function over S + 1 <= 16 values, as 4-bit entries of a 64-bit word; for S <= 7
// the entries are whole BYTES instead (8 of them), because then composing two functions is exactly
// what v_perm_b32 does -- pick bytes of one operand pair by the selector bytes of another: two
// instructions per composition instead of ~30 shift/mask operations.
template <int S>
struct FnTable {
    static constexpr bool kBytes = S <= 7;
    static constexpr int kBits = kBytes ? 8 : 4;
    __device__ static __forceinline__ uint64_t compose(uint64_t first, uint64_t then) {
        // (then o first)[s] = then[first[s]]
        if (kBytes) {
            const uint32_t tl = (uint32_t)then, th = (uint32_t)(then >> 32);
            const uint32_t lo = __builtin_amdgcn_perm(th, tl, (uint32_t)first);
            const uint32_t hi = __builtin_amdgcn_perm(th, tl, (uint32_t)(first >> 32));
            return ((uint64_t)hi << 32) | lo;
        }
        uint64_t out = 0;
#pragma unroll
        for (int s = 0; s <= S; ++s) {
            const int mid = (int)((first >> (4 * s)) & 15ull);
            out |= ((then >> (4 * mid)) & 15ull) << (4 * s);
        }
        return out;
    }
    __device__ static __forceinline__ uint64_t identity() {
        if (kBytes) return 0x0706050403020100ull;
        uint64_t id = 0;
#pragma unroll
        for (int s = 0; s <= S; ++s) id |= (uint64_t)s << (4 * s);
        return id;
    }
    __device__ static __forceinline__ uint64_t entry(int s, int value) { return (uint64_t)value << (kBits * s); }
    __device__ static __forceinline__ int at(uint64_t f, int s) {
        return (int)((f >> (kBits * s)) & (kBytes ? 255ull : 15ull));
    }
};

__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int o) {
    const int lo = __shfl_up((int)(uint32_t)v, o), hi = __shfl_up((int)(uint32_t)(v >> 32), o);
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}

constexpr int kCrfTileRows = 64;

// TM: time-major storage, (T, B, S, N) seen as a batch (stride_read = S * N): the four wavefronts of a workgroup take four
// NEIGHBOURING reads, whose rows of one time step are one contiguous run of 4 * S * N floats; the workgroup pulls 64-step
// tiles of the four together (coalesced 16-byte loads), parks them as [step][read][S * N] with an odd pitch, and every
// wavefront scans its read exactly as below.  (viterbi_tm_kernel is the same idea for the plain search.)
template <int S, bool TM>
__global__ __launch_bounds__(64 * kWavesPerBlock) void crf_greedy_stream_kernel(
    BatchDesc in, const float *init_all, int64_t n_init, int64_t init_stride, ResultDesc out) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const int lane = threadIdx.x & 63;
    // (wave-uniform, and said so: the read's index, its length and every pointer derived from them live in scalar registers)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int64_t r = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    if (TM) {  // every XCD takes a contiguous range of read groups: neighbouring groups share cache lines
        const unsigned n_groups = gridDim.x, per_xcd = (n_groups + 7) / 8;
        unsigned gidx = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if (n_groups < 64 || gidx >= n_groups || (n_groups & 7)) gidx = blockIdx.x;
        r = (int64_t)gidx * kWavesPerBlock + wave;  // (the launch covers whole groups only: every read exists)
    } else if (r >= in.n_reads) {
        return;
    }
    int64_t T = in.T;
    if (in.lengths) {
        int64_t t = in.lengths[r];
        T = t < 0 ? 0 : (t < T ? t : T);
    }
    const int N = in.N, n_base = N - 1;
    const int E = S * N;                      // floats per row
    const int tile_f = kCrfTileRows * E;      // floats per tile
    const int nload = TM ? (E + 3) / 4 : (tile_f / 4 + 63) / 64;  // float4 loads per lane and tile
    const int pitch = TM ? kWavesPerBlock * E + 1 : E;            // floats between consecutive rows of a read in the tile
    float *tile = TM ? s_dyn + wave * E : s_dyn + (size_t)wave * tile_f;
    const float *post = in.post + r * in.stride_read;
    const int64_t total = T * E;
    const int64_t T_loop = TM ? in.T : T;     // (TM: the workgroup's tiles cover the longest read of the batch)
    const int tm_step = (int)(threadIdx.x >> 2), tm_sub = (int)(threadIdx.x & 3);  // TM: four threads per time step
    const float *tm_base = in.post + (r - wave) * in.stride_read;                  // TM: the group's first read
    uint8_t *lab = out.labels + r * out.out_stride;
    uint32_t *pth = out.path ? out.path + r * out.out_stride : nullptr;
    float *qual = out.qual ? out.qual + r * out.out_stride : nullptr;

    // init_state.argmax() :399 (first maximum; NaN -> the reference panics)
    const float *init = init_all + r * init_stride;
    int state = 0;
    bool bad = n_init <= 0;
    if (!bad) {
        float m = init[0];
        bad = m != m;
        for (int64_t j = 1; j < n_init && !bad; ++j) {
            const float e = init[j];
            if (e != e) bad = true;
            if (e > m) {
                m = e;
                state = (int)j;
            }
        }
    }
    if (state >= S) state = S;  // out of range: the first row would abort

    constexpr int kMaxLoads = 8;  // S * N <= 32 floats per row (the launcher checks)
    float4 cur[kMaxLoads], nxt[kMaxLoads];
    auto fetch = [&](int64_t row0, float4 (&v)[kMaxLoads]) {
        if (TM) {
            // thread (step, sub) takes the 16-byte units sub, sub + 4, ... of its time step's run of 4 * E floats
            const float *src = tm_base + (row0 + tm_step) * in.stride_t;
#pragma unroll
            for (int m = 0; m < kMaxLoads; ++m) {
                if (m >= nload) break;
                const int o = tm_sub + 4 * m;
                v[m] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (o < E && row0 + tm_step < in.T) v[m] = *reinterpret_cast<const float4 *>(src + o * 4);
            }
            return;
        }
        const int64_t f0 = row0 * E;
#pragma unroll
        for (int m = 0; m < kMaxLoads; ++m) {
            if (m >= nload) break;
            const int64_t fl = (int64_t)(lane + 64 * m) * 4;  // float offset inside the tile
            const int64_t f = f0 + fl;
            if (fl + 3 < tile_f && f + 3 < total) {
                v[m] = *reinterpret_cast<const float4 *>(post + f);
            } else {
                v[m].x = (fl < tile_f && f < total) ? post[f] : 0.0f;
                v[m].y = (fl + 1 < tile_f && f + 1 < total) ? post[f + 1] : 0.0f;
                v[m].z = (fl + 2 < tile_f && f + 2 < total) ? post[f + 2] : 0.0f;
                v[m].w = (fl + 3 < tile_f && f + 3 < total) ? post[f + 3] : 0.0f;
            }
        }
    };

    int n_out = 0;
    if (T_loop > 0) fetch(0, cur);
    for (int64_t base = 0; base < T_loop; base += kCrfTileRows) {
        if (base + kCrfTileRows < T_loop) fetch(base + kCrfTileRows, nxt);
        if (TM) {
#pragma unroll
            for (int m = 0; m < kMaxLoads; ++m) {
                if (m >= nload) break;
                const int o = tm_sub + 4 * m;
                if (o < E) {
                    float *dst = s_dyn + tm_step * pitch + o * 4;  // (odd pitch: 4-byte alignment only)
                    dst[0] = cur[m].x;
                    dst[1] = cur[m].y;
                    dst[2] = cur[m].z;
                    dst[3] = cur[m].w;
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int m = 0; m < kMaxLoads; ++m) {
            if (TM || m >= nload) break;
            const int fl = (lane + 64 * m) * 4;
            if (fl + 3 < tile_f) {
                *reinterpret_cast<float4 *>(tile + fl) = cur[m];
            } else {
                if (fl < tile_f) tile[fl] = cur[m].x;
                if (fl + 1 < tile_f) tile[fl + 1] = cur[m].y;
                if (fl + 2 < tile_f) tile[fl + 2] = cur[m].z;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        const int64_t row = base + lane;
        const bool act = row < T;
        if (!TM || base < T) {  // (TM: a wavefront whose read has ended still takes part in the tile loads)
        // this row, for every state: first-maximum argmax (:405), its probability, NaN anywhere in the row
        const float *pr = tile + lane * pitch;
        float prob_s[S];
        uint32_t labels_packed = 0, nan_mask = 0;
        uint64_t f = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float best = pr[s * N];
            int arg = 0;
            bool nan = best != best;
            for (int j = 1; j < N; ++j) {
                const float v = pr[s * N + j];
                nan = nan || (v != v);
                if (v > best) {
                    best = v;
                    arg = j;
                }
            }
            prob_s[s] = best;
            labels_packed |= (uint32_t)arg << (4 * s);
            nan_mask |= nan ? (1u << s) : 0u;
            int next = arg > 0 ? (s * n_base) % S + (arg - 1) : s;  // :415
            if (next >= S) next = S;
            f |= FnTable<S>::entry(s, next);
        }
        // out of range stays out of range (byte tables: the unused upper entries map to themselves)
        if (FnTable<S>::kBytes) {
#pragma unroll
            for (int s2 = S; s2 < 8; ++s2) f |= FnTable<S>::entry(s2, s2 == S ? S : s2);
        } else {
            f |= FnTable<S>::entry(S, S);
        }
        if (!act) f = FnTable<S>::identity();  // rows past the end change nothing

        // inclusive scan of composition: F_k = f_k o ... o f_0
        uint64_t F = f;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const uint64_t g = shfl_up_u64(F, o);
            if (lane >= o) F = FnTable<S>::compose(g, F);
        }
        // the state this row is entered with
        uint64_t Fprev = shfl_up_u64(F, 1);
        if (lane == 0) Fprev = FnTable<S>::identity();
        const int st = FnTable<S>::at(Fprev, state);
        // the state the tile leaves behind (wave-uniform)
        const uint64_t Flast = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(F >> 32), 63) << 32) |
                               (uint32_t)__shfl((int)(uint32_t)F, 63);
        const int state_out = FnTable<S>::at(Flast, state);

        // a row entered out of range, or a NaN in the visited state's row: the reference aborts
        const bool row_bad = act && (st >= S || ((nan_mask >> (st < S ? st : 0)) & 1u));
        if (ballot(row_bad) != 0ull) bad = true;
        int label = 0;
        float prob = 0.0f;
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (st == s) {
                label = (int)((labels_packed >> (4 * s)) & 15u);
                prob = prob_s[s];
            }
        const bool emit = act && !row_bad && label > 0;
        const uint64_t m_emit = ballot(emit);
        const int my_out = n_out + popc64(m_emit & lanemask_lt());
        if (emit) {
            lab[my_out] = (uint8_t)label;
            if (pth) pth[my_out] = (uint32_t)row;
            if (qual) qual[my_out] = prob;
        }
        n_out += popc64(m_emit);
        state = state_out;
        }
        if (TM) __syncthreads();
        else __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < kMaxLoads; ++m) cur[m] = nxt[m];
    }
    if (lane == 0) {
        out.out_len[r] = bad ? 0u : (uint32_t)n_out;
        if (out.status) out.status[r] = bad ? FCD_ST_BAD_STATE : FCD_ST_OK;
    }
}

}  // namespace

hipError_t launch_viterbi(const BatchDesc &in, int collapse, const ResultDesc &out,
                          hipStream_t stream) {
    if (in.n_reads <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((in.n_reads + kWavesPerBlock - 1) / kWavesPerBlock);
    const dim3 grid(blocks), block(64 * kWavesPerBlock);
    // 16-byte loads: every read must start on a 16-byte boundary (4 f32 / 8 half elements)
    const bool stream_ok = in.stride_n == 1 && in.stride_t == in.N && (in.stride_read % (in.dtype == kF32 ? 4 : 8)) == 0 &&
                           (reinterpret_cast<uintptr_t>(in.post) % 16) == 0;
    // time-major storage ((T, B, N) seen as a batch): neighbouring reads are N elements apart
    const int64_t esz = in.dtype == kF32 ? 4 : 2;
    const bool tm_ok = in.stride_n == 1 && in.stride_read == in.N && in.N >= 2 && in.N <= 8 && in.n_reads >= kTmReads &&
                       in.stride_t >= in.n_reads * in.N && (in.stride_t * esz) % 16 == 0 &&
                       (reinterpret_cast<uintptr_t>(in.post) % 16) == 0;
    if (tm_ok) {
        const int64_t groups = in.n_reads / kTmReads, done = groups * kTmReads;
        switch (in.N) {
#define FCD_VTM(NN)                                                                                                 \
    case NN:                                                                                                        \
        if (in.dtype == kF32)                                                                                       \
            hipLaunchKernelGGL((viterbi_tm_kernel<NN, kF32>), dim3((unsigned)groups), dim3(256), 0, stream, in, collapse, out);  \
        else if (in.dtype == kF16)                                                                                  \
            hipLaunchKernelGGL((viterbi_tm_kernel<NN, kF16>), dim3((unsigned)groups), dim3(256), 0, stream, in, collapse, out);  \
        else                                                                                                        \
            hipLaunchKernelGGL((viterbi_tm_kernel<NN, kBF16>), dim3((unsigned)groups), dim3(256), 0, stream, in, collapse, out); \
        break;
            FCD_VTM(2) FCD_VTM(3) FCD_VTM(4) FCD_VTM(5) FCD_VTM(6) FCD_VTM(7) FCD_VTM(8)
#undef FCD_VTM
            default: break;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess || done == in.n_reads) return e;
        // the last n_reads mod 8 reads: the strided kernel on the tail of the batch
        BatchDesc in2 = in;
        in2.post = reinterpret_cast<const float *>(reinterpret_cast<const char *>(in.post) + done * in.stride_read * esz);
        in2.lengths = in.lengths ? in.lengths + done : nullptr;
        in2.n_reads = in.n_reads - done;
        ResultDesc out2 = out;
        out2.labels = out.labels + done * out.out_stride;
        out2.path = out.path ? out.path + done * out.out_stride : nullptr;
        out2.qual = out.qual ? out.qual + done * out.out_stride : nullptr;
        out2.out_len = out.out_len + done;
        out2.status = out.status ? out.status + done : nullptr;
        const unsigned blocks2 = (unsigned)((in2.n_reads + kWavesPerBlock - 1) / kWavesPerBlock);
        hipLaunchKernelGGL(viterbi_kernel, dim3(blocks2), block, 0, stream, in2, collapse, out2);
        return hipGetLastError();
    }
    if (stream_ok) {
        switch (in.N) {
#define FCD_VSTREAM(NN)                                                                                         \
    case NN:                                                                                                    \
        if (in.dtype == kF32)                                                                                   \
            hipLaunchKernelGGL((viterbi_stream_kernel<NN, kF32>), grid, block, 0, stream, in, collapse, out);   \
        else if (in.dtype == kF16)                                                                              \
            hipLaunchKernelGGL((viterbi_stream_kernel<NN, kF16>), grid, block, 0, stream, in, collapse, out);   \
        else                                                                                                    \
            hipLaunchKernelGGL((viterbi_stream_kernel<NN, kBF16>), grid, block, 0, stream, in, collapse, out);  \
        return hipGetLastError();
            FCD_VSTREAM(2) FCD_VSTREAM(3) FCD_VSTREAM(4) FCD_VSTREAM(5) FCD_VSTREAM(6) FCD_VSTREAM(7)
            FCD_VSTREAM(8)
#undef FCD_VSTREAM
            default: break;
        }
    }
    hipLaunchKernelGGL(viterbi_kernel, grid, block, 0, stream, in, collapse, out);
    return hipGetLastError();
}

hipError_t launch_crf_greedy(const BatchDesc &in, const float *init, int64_t n_init,
                             int64_t init_stride, const ResultDesc &out, hipStream_t stream) {
    if (in.n_reads <= 0) return hipSuccess;
    // small state counts, C-contiguous rows: the streaming kernel (function-composition scan)
    const int64_t E = (int64_t)in.S * in.N;
    // (half-precision input takes the serial kernel below: the function-composition scan reads f32 tiles)
    const bool stream_ok = in.dtype == kF32 && in.S >= 1 && in.S <= 8 && in.N >= 1 && in.N <= 15 && E <= 32 && in.stride_n == 1 &&
                           in.stride_s == in.N && in.stride_t == E && (in.stride_read % 4) == 0 &&
                           (reinterpret_cast<uintptr_t>(in.post) % 16) == 0;
    if (stream_ok) {
        const unsigned blocks = (unsigned)((in.n_reads + kWavesPerBlock - 1) / kWavesPerBlock);
        const size_t lds = (size_t)kWavesPerBlock * kCrfTileRows * E * sizeof(float);
        switch (in.S) {
#define FCD_GSTREAM(SS)                                                                                   \
    case SS:                                                                                              \
        hipLaunchKernelGGL((crf_greedy_stream_kernel<SS, false>), dim3(blocks), dim3(64 * kWavesPerBlock), lds, stream, \
                           in, init, n_init, init_stride, out);                                           \
        return hipGetLastError();
            FCD_GSTREAM(1) FCD_GSTREAM(2) FCD_GSTREAM(3) FCD_GSTREAM(4) FCD_GSTREAM(5) FCD_GSTREAM(6)
            FCD_GSTREAM(7) FCD_GSTREAM(8)
#undef FCD_GSTREAM
            default: break;
        }
    }
    // time-major storage ((T, B, S, N) seen as a batch): neighbouring reads are S * N floats apart
    const bool tm_ok = in.dtype == kF32 && in.S >= 1 && in.S <= 8 && in.N >= 1 && in.N <= 15 && E <= 32 && in.stride_n == 1 &&
                       in.stride_s == in.N && in.stride_read == E && in.n_reads >= kWavesPerBlock &&
                       in.stride_t >= in.n_reads * E && (in.stride_t % 4) == 0 &&
                       (reinterpret_cast<uintptr_t>(in.post) % 16) == 0;
    int64_t done = 0;
    if (tm_ok) {
        const int64_t groups = in.n_reads / kWavesPerBlock;
        const size_t lds = (size_t)kCrfTileRows * (kWavesPerBlock * E + 1) * sizeof(float);
        switch (in.S) {
#define FCD_GTM(SS)                                                                                              \
    case SS:                                                                                                     \
        hipLaunchKernelGGL((crf_greedy_stream_kernel<SS, true>), dim3((unsigned)groups), dim3(64 * kWavesPerBlock), lds, \
                           stream, in, init, n_init, init_stride, out);                                          \
        done = groups * kWavesPerBlock;                                                                          \
        break;
            FCD_GTM(1) FCD_GTM(2) FCD_GTM(3) FCD_GTM(4) FCD_GTM(5) FCD_GTM(6) FCD_GTM(7) FCD_GTM(8)
#undef FCD_GTM
            default: break;
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess || done == in.n_reads) return e;
    }
    // (everything else, and the last n_reads mod 4 reads of a time-major batch: the serial walk)
    BatchDesc in2 = in;
    in2.post = in.post + done * in.stride_read;  // (done > 0 only for float32 input: plain element arithmetic)
    in2.lengths = in.lengths ? in.lengths + done : nullptr;
    in2.n_reads = in.n_reads - done;
    ResultDesc out2 = out;
    out2.labels = out.labels + done * out.out_stride;
    out2.path = out.path ? out.path + done * out.out_stride : nullptr;
    out2.qual = out.qual ? out.qual + done * out.out_stride : nullptr;
    out2.out_len = out.out_len + done;
    out2.status = out.status ? out.status + done : nullptr;
    hipLaunchKernelGGL(crf_greedy_kernel, dim3((unsigned)in2.n_reads), dim3(64), 0, stream, in2, init + done * init_stride,
                       n_init, init_stride, out2);
    return hipGetLastError();
}

}  // namespace fcd
