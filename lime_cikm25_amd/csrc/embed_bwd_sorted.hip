// lime_embed_bwd_sorted_f32: dTable[id] = sum over the rows r with ids[r] == id of dX[r], WITHOUT atomics.
//
// lime_embed_bwd_f32 adds with float atomics (a word row receives contributions from many tokens, the padding word from half
// of them), which made the word-table gradient the one part of a training step that is not bitwise reproducible.  Here the
// caller sorts the token positions by id once (a stable sort: torch.sort) and the sum becomes a segmented sum over the sorted
// order with a FIXED association:
//   pass A  one wave per chunk of 256 consecutive sorted positions walks its rows in order (eight row loads in flight), keeping
//           a running sum per run of equal ids.  A run that lies inside the chunk is owned by this wave: plain store to
//           dTable[id].  A run that crosses the chunk's first / last position leaves a partial ("head" / "tail") in the workspace;
//   pass B  one wave per run that crosses chunk borders adds its tail partial and the following chunks' head partials in chunk
//           order (the padding word: ~250 partials) and stores dTable[id].
// Every row of dTable that receives a contribution is written exactly once; rows without one keep the caller's value (zero
// them first).  Word ids are >= 0.
// The FIRST run of the sorted order (the smallest id: the padding word, 70 % of all positions of a MIND-shaped batch) is taken out
// of that scheme when it is long: 256 workgroups sum equal slices of it (pass H, partial rows in the workspace) and one more wave adds
// the 256 partials in order -- as one run it was ~340 partials summed by ONE wave in pass B (83 us) behind a pass A whose chunk count
// (one wave per 256 positions) left three quarters of the chip idle (108 us).  Passes A / B then start behind it, in chunks of 64.
#include "common.h"

namespace {

constexpr int CH = 64;        // sorted positions per chunk (one wave)
constexpr int HOT_G = 256;    // workgroups (= partial rows) of the first run's sum
constexpr int HOT_MIN = 8192; // shorter first runs stay in passes A / B

struct HotInfo { int n0, id0, pad0, pad1; };       // n0: positions of the first run handled by pass H (0: none)
constexpr int CPL = 5;        // columns per lane: dim <= 320

struct ChunkFlags { int head_valid, head_cont, tail_valid, tail_id; };

__device__ __forceinline__ void store_row(float* dst, const float (&acc)[CPL], int lane, int dim) {
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int c = lane + 64 * j;
        if (c < dim) dst[c] = acc[j];
    }
}

__global__ __launch_bounds__(256) void embed_bwd_chunks_kernel(const int* __restrict__ order, const int* __restrict__ sorted_ids,
                                                               const float* __restrict__ dx, long lddx, float* __restrict__ dtable,
                                                               long ldt, long rows, int dim, float* __restrict__ ws_head,
                                                               float* __restrict__ ws_tail, ChunkFlags* __restrict__ flags, long wsld,
                                                               const HotInfo* __restrict__ hot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long c = (long)blockIdx.x * 4 + wave;
    const long start = (long)hot->n0 + c * CH;             // behind the run pass H owns
    if (start >= rows) return;
    const long end = min(rows, start + CH);
    const int prev_id = start > hot->n0 ? sorted_ids[start - 1] : -1;
    const int next_id = end < rows ? sorted_ids[end] : -2;
    ChunkFlags f = {0, 0, 0, -1};
    float acc[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[j] = 0.f;
    int cur = sorted_ids[start];
    bool at_start = true;                                  // the current run began at the chunk's first position

    auto flush = [&](bool at_end) {
        const bool cont_prev = at_start && prev_id == cur;
        const bool cont_next = at_end && next_id == cur;
        if (!cont_prev && !cont_next) {
            store_row(dtable + (long)cur * ldt, acc, lane, dim);           // the run lies inside this chunk: this wave owns the row
        } else if (cont_prev) {
            store_row(ws_head + c * wsld, acc, lane, dim);
            f.head_valid = 1;
            f.head_cont = cont_next ? 1 : 0;                               // the run covers the whole chunk and goes on
        } else {
            store_row(ws_tail + c * wsld, acc, lane, dim);
            f.tail_valid = 1;
            f.tail_id = cur;
        }
    };

    for (long g0 = start; g0 < end; g0 += 64) {                            // 64 positions: ids / row numbers in lanes
        const long gi = g0 + lane;
        const int id_v = gi < end ? sorted_ids[gi] : -3;
        const int row_v = gi < end ? order[gi] : 0;
        const int n = (int)min((long)64, end - g0);
        for (int k0 = 0; k0 < n; k0 += 8) {                               // eight rows in flight
            float v[8][CPL];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = __shfl(row_v, (k0 + u) & 63);
                const bool ok = k0 + u < n;
                const float* p = dx + (long)r * lddx;
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    const int col = lane + 64 * j;
                    v[u][j] = (ok && col < dim) ? p[col] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (k0 + u < n) {                                          // wave-uniform
                    const int id = __shfl(id_v, (k0 + u) & 63);
                    if (id != cur) {
                        flush(false);
                        cur = id;
                        at_start = false;
#pragma unroll
                        for (int j = 0; j < CPL; ++j) acc[j] = 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < CPL; ++j) acc[j] += v[u][j];
                }
            }
        }
    }
    flush(true);
    if (lane == 0) flags[c] = f;
}

__global__ __launch_bounds__(256) void embed_bwd_combine_kernel(const float* __restrict__ ws_head, const float* __restrict__ ws_tail,
                                                                const ChunkFlags* __restrict__ flags, long rows,
                                                                float* __restrict__ dtable, long ldt, int dim, long wsld,
                                                                const HotInfo* __restrict__ hot, const float* __restrict__ ws_hot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == gridDim.x - 1) {                     // the extra workgroup: pass H's 256 partial rows, in order, by one wave
        if (wave == 0 && hot->n0 > 0) {
            float acc[CPL];
#pragma unroll
            for (int j = 0; j < CPL; ++j) acc[j] = 0.f;
            for (int g0 = 0; g0 < HOT_G; g0 += 8) {
                float v[8][CPL];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const int col = lane + 64 * j;
                        v[u][j] = col < dim ? ws_hot[(long)(g0 + u) * wsld + col] : 0.f;
                    }
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int j = 0; j < CPL; ++j) acc[j] += v[u][j];
            }
            store_row(dtable + (long)hot->id0 * ldt, acc, lane, dim);
        }
        return;
    }
    const long n_chunks = (rows - hot->n0 + CH - 1) / CH;
    const long c = (long)blockIdx.x * 4 + wave;
    if (c >= n_chunks) return;
    const ChunkFlags f = flags[c];
    if (!f.tail_valid) return;                             // no run starts here and leaves the chunk
    float acc[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int col = lane + 64 * j;
        acc[j] = col < dim ? ws_tail[c * wsld + col] : 0.f;
    }
    for (long j0 = c + 1; j0 < n_chunks; j0 += 8) {        // the following chunks' head partials, in chunk order
        float v[8][CPL];
        int cont[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long jj = j0 + u;
            const bool ok = jj < n_chunks;
            cont[u] = ok ? flags[jj].head_cont : 0;
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const int col = lane + 64 * j;
                v[u][j] = (ok && col < dim) ? ws_head[jj * wsld + col] : 0.f;
            }
        }
        bool done = false;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (!done) {
#pragma unroll
                for (int j = 0; j < CPL; ++j) acc[j] += v[u][j];
                if (!cont[u]) done = true;                 // this head partial closed the run
            }
        }
        if (done) break;
    }
    store_row(dtable + (long)f.tail_id * ldt, acc, lane, dim);
}

// pass H: the first run of the sorted order (positions 0 .. n0 - 1, all of id sorted_ids[0]) summed by HOT_G workgroups over equal
// slices: a wave takes its slice's positions w, w + 4, ... (eight row loads in flight), the four waves' sums are added in wave order
__global__ __launch_bounds__(256) void embed_bwd_hot_kernel(const int* __restrict__ order, const int* __restrict__ sorted_ids,
                                                            const float* __restrict__ dx, long lddx, long rows, int dim,
                                                            float* __restrict__ ws_hot, long wsld, HotInfo* __restrict__ hot) {
    __shared__ float red[4][64 * CPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int id0 = sorted_ids[0];
    long lo = 0, hi = rows;                                 // upper bound of id0 (uniform: every thread walks the same 20 steps)
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (sorted_ids[mid] <= id0) lo = mid + 1; else hi = mid;
    }
    const long n0 = lo >= HOT_MIN ? lo : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) *hot = HotInfo{(int)n0, id0, 0, 0};
    if (n0 == 0) return;
    const long L = (n0 + HOT_G - 1) / HOT_G;
    const long p0 = (long)blockIdx.x * L, p1 = min(n0, p0 + L);
    float acc[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[j] = 0.f;
    for (long q = p0 + wave; q < p1; q += 32) {             // this wave: q, q + 4, ..., eight at a time
        float v[8][CPL];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long pos = q + 4 * u;
            const bool ok = pos < p1;
            const float* pr = dx + (long)order[ok ? pos : p0] * lddx;
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const int col = lane + 64 * j;
                v[u][j] = (ok && col < dim) ? pr[col] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < CPL; ++j) acc[j] += v[u][j];
    }
#pragma unroll
    for (int j = 0; j < CPL; ++j) red[wave][lane + 64 * j] = acc[j];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int col = lane + 64 * j;
            if (col < dim) ws_hot[(long)blockIdx.x * wsld + col] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        }
    }
}

}  // namespace

extern "C" int64_t lime_embed_bwd_sorted_workspace(int64_t rows, int32_t dim) {
    if (rows <= 0 || dim <= 0) return 0;
    const int64_t n_chunks = (rows + CH - 1) / CH;
    const int64_t wsld = (dim + 3) / 4 * 4;
    return n_chunks * (2 * wsld + 4) + HOT_G * wsld + 4;  // head + tail partials + the four flag words + pass H's partial rows + its info, in floats
}

extern "C" int lime_embed_bwd_sorted_f32(const int32_t* order, const int32_t* sorted_ids, const float* dx, int64_t lddx, float* dtable,
                                         int64_t ld_table, int64_t rows, int32_t dim, float* workspace, int64_t workspace_floats,
                                         void* stream) {
    LIME_REQUIRE(order && sorted_ids && dx && dtable && workspace, LIME_ERR_BAD_ARG, "lime_embed_bwd_sorted_f32: null pointer");
    LIME_REQUIRE(rows >= 0 && dim > 0 && lddx >= dim && ld_table >= dim, LIME_ERR_BAD_ARG, "lime_embed_bwd_sorted_f32: bad dimensions");
    LIME_REQUIRE(dim <= 64 * CPL, LIME_ERR_UNSUPPORTED, "lime_embed_bwd_sorted_f32: dim = %d > %d", dim, 64 * CPL);
    LIME_REQUIRE(workspace_floats >= lime_embed_bwd_sorted_workspace(rows, dim), LIME_ERR_BAD_ARG, "lime_embed_bwd_sorted_f32: workspace too small");
    if (rows == 0) return LIME_OK;
    const long n_chunks = (rows + CH - 1) / CH;
    const long wsld = (dim + 3) / 4 * 4;
    float* ws_head = workspace;
    float* ws_tail = workspace + n_chunks * wsld;
    ChunkFlags* flags = reinterpret_cast<ChunkFlags*>(workspace + 2 * n_chunks * wsld);
    float* ws_hot = workspace + n_chunks * (2 * wsld + 4);
    HotInfo* hot = reinterpret_cast<HotInfo*>(ws_hot + (long)HOT_G * wsld);
    hipStream_t s = (hipStream_t)stream;
    const int grid = (int)((n_chunks + 3) / 4);
    hipLaunchKernelGGL(embed_bwd_hot_kernel, dim3(HOT_G), dim3(256), 0, s, order, sorted_ids, dx, (long)lddx, (long)rows, dim, ws_hot, wsld, hot);
    hipLaunchKernelGGL(embed_bwd_chunks_kernel, dim3(grid), dim3(256), 0, s, order, sorted_ids, dx, (long)lddx, dtable, (long)ld_table,
                       (long)rows, dim, ws_head, ws_tail, flags, wsld, (const HotInfo*)hot);
    hipLaunchKernelGGL(embed_bwd_combine_kernel, dim3(grid + 1), dim3(256), 0, s, (const float*)ws_head, (const float*)ws_tail,
                       (const ChunkFlags*)flags, (long)rows, dtable, (long)ld_table, dim, wsld, (const HotInfo*)hot, (const float*)ws_hot);
    return lime_check_launch("lime_embed_bwd_sorted_f32");
}
