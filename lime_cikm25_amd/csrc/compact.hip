// lime_compact_sequences: which token sequences of a batch are worth encoding, and which of their rows.
//
// The reference pushes every slot of every impression through the token encoders (newsEncoders.py:311-321): H history slots
// that are padded with the all-zero <PAD> news (corpus.py:476-477, dataset.py:105-141) and, inside every real news, the
// padding tokens behind its text.  Both are exact repetitions:
//   * an all-padding sequence has the same pooled output wherever it occurs (the encoder layer has no mask and no
//     cross-sequence term), so ONE representative is encoded and every such slot reads its result;
//   * a padding token's in_proj row depends on its position only ((E[0] + PE[t]) W^T + b), so in_proj runs over the live
//     tokens and attention reads the S table rows for the rest (lime_token_attention_rows_f32).
// Nothing is approximated and nothing is cached between forwards.  This file turns the [n_seq, S] id matrix into the index
// lists those kernels consume -- on the device, in fixed-size buffers, with the counts in device memory, so that the
// whole forward stays one HIP graph:
//   seq_inv  [n_seq]           original sequence -> compact sequence (all-padding sequences -> n_live, the representative)
//   ids_c    [(n_seq + 1) S]   token ids in compact order (representative: zeros)
//   row_map  [(n_seq + 1) S]   compact token row -> its q/k/v row: itself when live, pad_base + t when padding
//   tok_ids  [(n_seq + 1) S]   the live tokens' ids, compacted in (compact sequence, position) order
//   tok_rows [(n_seq + 1) S]   the compact token row of each of them
//   counts   [5]               n_c = n_live + 1, n_c * S, number of live tokens, n_live, live tokens + S
// Behind the live tokens tok_ids / tok_rows carry S more entries (id 0 -> row pad_base + t): the padding rows themselves, so that
// ONE in_proj launch over counts[4] rows produces the live rows and the table the row map points the padding tokens at.
// Everything is ordered and deterministic (no atomics): a single-workgroup scan over the per-sequence counts between two
// wide passes.
#include "common.h"

namespace {

// pass 1: one wave per sequence -> number of live (non-zero) tokens
__global__ __launch_bounds__(256) void seq_count_kernel(const int* __restrict__ ids, int n_seq, int S, int* __restrict__ live_cnt) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_seq) return;
    int c = 0;
    for (int t = lane; t < S; t += 64) c += ids[(long)s * S + t] != 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0) live_cnt[s] = c;
}

// pass 2: ONE workgroup: ordered exclusive scans over the sequences (live flag -> compact index, live tokens -> row offset)
__global__ __launch_bounds__(1024) void seq_scan_kernel(const int* __restrict__ live_cnt, int n_seq, int S, int* __restrict__ seq_src,
                                                        int* __restrict__ seq_inv, int* __restrict__ tok_off, int* __restrict__ counts) {
    __shared__ int part_seq[1024], part_tok[1024];
    const int tid = threadIdx.x;
    const int per = (n_seq + 1023) / 1024;
    const int lo = tid * per, hi = min(n_seq, lo + per);
    int ns = 0, nt = 0;
    for (int s = lo; s < hi; ++s) {
        const int c = live_cnt[s];
        ns += c > 0;
        nt += c;
    }
    part_seq[tid] = ns;
    part_tok[tid] = nt;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 1024 partials (two arrays)
    for (int o = 1; o < 1024; o <<= 1) {
        int a = 0, b = 0;
        if (tid >= o) { a = part_seq[tid - o]; b = part_tok[tid - o]; }
        __syncthreads();
        part_seq[tid] += a;
        part_tok[tid] += b;
        __syncthreads();
    }
    const int n_live = part_seq[1023], n_tok = part_tok[1023];
    int ps = part_seq[tid] - ns, pt = part_tok[tid] - nt;        // exclusive prefixes of this thread's range
    for (int s = lo; s < hi; ++s) {
        const int c = live_cnt[s];
        if (c > 0) {
            seq_src[ps] = s;
            seq_inv[s] = ps;
            tok_off[ps] = pt;
            ++ps;
            pt += c;
        } else {
            seq_inv[s] = n_live;                                   // the all-padding representative
        }
    }
    for (int i = n_live + 1 + tid; i <= n_seq; i += 1024) seq_src[i] = -1;      // unused compact slots: no source sequence
    if (tid == 0) {
        seq_src[n_live] = -1;
        tok_off[n_live] = n_tok;
        counts[0] = n_live + 1;
        counts[1] = (n_live + 1) * S;
        counts[2] = n_tok;
        counts[3] = n_live;
        counts[4] = n_tok + S;
    }
}

// pass 3: one wave per compact sequence -> ids_c, row_map, and the live tokens' (id, row) appended at the sequence's offset
__global__ __launch_bounds__(256) void seq_emit_kernel(const int* __restrict__ ids, int n_seq, int S, const int* __restrict__ seq_src,
                                                       const int* __restrict__ tok_off, const int* __restrict__ counts, int pad_base,
                                                       int* __restrict__ ids_c, int* __restrict__ row_map, int* __restrict__ tok_ids,
                                                       int* __restrict__ tok_rows) {
    const int lane = threadIdx.x & 63;
    const int cs = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_c = counts[0];
    if (cs >= n_c) return;
    const int src = seq_src[cs];
    int base = tok_off[cs];
    for (int t0 = 0; t0 < S; t0 += 64) {
        const int t = t0 + lane;
        int id = 0;
        if (t < S && src >= 0) id = ids[(long)src * S + t];
        const bool live = id != 0;
        const unsigned long long m = __ballot(live);
        if (t < S) {
            const int row = cs * S + t;
            ids_c[row] = id;
            row_map[row] = live ? row : pad_base + t;
            if (cs == n_c - 1) {                       // the representative: its offset is the end of the live list
                tok_ids[base + t] = 0;
                tok_rows[base + t] = pad_base + t;
            }
            if (live) {
                const int k = base + __popcll(m & ((1ull << lane) - 1ull));
                tok_ids[k] = id;
                tok_rows[k] = row;
            }
        }
        base += __popcll(m);
    }
}

}  // namespace

extern "C" int lime_compact_sequences(const int32_t* ids, int32_t n_seq, int32_t S, int32_t pad_base, int32_t* seq_inv, int32_t* ids_c,
                                      int32_t* row_map, int32_t* tok_ids, int32_t* tok_rows, int32_t* counts, int32_t* work, void* stream) {
    LIME_REQUIRE(ids && seq_inv && ids_c && row_map && tok_ids && tok_rows && counts && work, LIME_ERR_BAD_ARG, "lime_compact_sequences: NULL pointer");
    LIME_REQUIRE(n_seq > 0 && S > 0 && pad_base >= 0, LIME_ERR_BAD_ARG, "lime_compact_sequences: bad dims n_seq=%d S=%d pad_base=%d", n_seq, S, pad_base);
    LIME_REQUIRE((long)(n_seq + 1) * S + S < 0x7FFFFFFFL && (long)pad_base + S < 0x7FFFFFFFL, LIME_ERR_UNSUPPORTED, "lime_compact_sequences: too many rows");
    hipStream_t s = (hipStream_t)stream;
    int* live_cnt = work;                        // [n_seq]
    int* seq_src = work + n_seq;                 // [n_seq + 1]
    int* tok_off = work + 2 * n_seq + 1;         // [n_seq + 1]
    hipLaunchKernelGGL(seq_count_kernel, dim3((n_seq + 3) / 4), dim3(256), 0, s, ids, n_seq, S, live_cnt);
    hipLaunchKernelGGL(seq_scan_kernel, dim3(1), dim3(1024), 0, s, live_cnt, n_seq, S, seq_src, seq_inv, tok_off, counts);
    hipLaunchKernelGGL(seq_emit_kernel, dim3((n_seq + 1 + 3) / 4), dim3(256), 0, s, ids, n_seq, S, seq_src, tok_off, counts, pad_base, ids_c,
                       row_map, tok_ids, tok_rows);
    return lime_check_launch("lime_compact_sequences");
}

// int32 words `work` must hold
extern "C" int64_t lime_compact_sequences_workspace(int32_t n_seq) { return 3L * n_seq + 2; }
