// lime_inproj_bf16: the q / k / v projection of an encoder layer (nn.MultiheadAttention's in_proj inside the
// TransformerEncoderLayers of newsEncoders.py:244-247, 316-321) on the bf16 matrix cores, activation-stationary.
//
// The same structure as csrc/ffn_bf16.hip (see there for the reasons): one four-wave workgroup per CU keeps a 128-token tile of
// the layer input in LDS -- here the word-table rows of the batch's live tokens, gathered by id straight into the swizzled
// [chunk][row][64 bytes] image by LDS-DMA -- and streams the weight through a ring of four 20 KB slots filled three steps
// ahead (counted s_waitcnt vmcnt, raw s_barrier).  N = 3 x 320 columns (ten heads padded to 32 columns each, as the attention
// kernels read them) are produced in passes of 320: the tile is read from HBM once instead of once per column block, and the
// weight slots are contiguous L2-resident blocks (lime_inproj_pack_bf16).  Against gemm_pp_kernel<.., BF> on the same launch
// the CU pulls 0.68 MB per 128 tokens through its L2 -> LDS path instead of 0.86 MB, with three times the bytes in flight.
//   * a wave owns 32 tokens x all 320 columns of a pass (2 x 20 accumulator tiles); a pass is ten 32-deep k steps of 40
//     v_mfma_f32_16x16x32_bf16 each, the wave's share of the refill DMAs between the MFMA groups, the next step's first
//     fragments read across the barrier.
//   * pass epilogue: + the fp32 periodic rows (positional term x weight + bias, prepared by the caller: row % period of the
//     OUTPUT row), rounded to bf16, stored to row c_ids[r] (the compacted batch scatters its live tokens) -- the fp32 rows are
//     fetched two steps ahead.
//   * during the last pass every chunk of the image is refilled with the NEXT tile's rows as soon as this wave is done with it
//     (a wave only ever touches its own 32 rows of the image), so no gather latency is exposed between tiles; the row ids
//     travel two tiles ahead as plain loads that are only consumed at a tile's first step, where everything is waited for anyway.
#include <type_traits>

#include "lds_dma.h"

using namespace lime_dev;

namespace {

constexpr int BM = 128;
constexpr int NT = 20, PWD = 16 * NT;      // columns per pass: 320
constexpr int NCH = 10;                    // 32-deep k chunks (K <= 320; columns beyond K are zero-filled / zero weights)
constexpr int SLAB = BM * 64;
constexpr int XS_BYTES = NCH * SLAB;       // 81,920
constexpr int SLOT = PWD * 64;             // 20,480
constexpr int LDS_BYTES = XS_BYTES + 4 * SLOT;      // 163,840: all of the CU's LDS

struct InP {
    const uint16_t* a; long lda; const int* a_ids;       // rows (or a table gathered by a_ids), K columns valid
    const uint16_t* wp;                                  // [n_pass][NCH][320][32]
    const float* add; long ld_add; int period;           // fp32 [period, >= N]
    const int* c_ids;
    uint16_t* out; long ldo;
    int M, N, K;
    const int* m_dev;
};

__device__ __forceinline__ int lane_here() {            // see ffn_bf16.hip
    int z = 0;
    asm volatile("" : "+v"(z));
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
}

__global__ __launch_bounds__(256, 1) void inproj_bf16_kernel(const InP p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, kg = lane >> 4;
    int M = p.M;
    if (p.m_dev) {
        const int m = __builtin_amdgcn_readfirstlane(*p.m_dev);
        M = m < M ? (m > 0 ? m : 0) : M;
    }
    const int ntiles = (M + BM - 1) / BM;
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    const int NP = p.N / PWD;
    const int stride = (int)gridDim.x;

    const int srow = lane >> 2;
    const int lseg = (lane & 3) ^ swz4((lane >> 4) & 3);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.wp);
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(p.a);
    const unsigned ldab = (unsigned)p.lda * 2u;
    unsigned char* const ring = lds + XS_BYTES;
    const unsigned w_lane = (unsigned)srow * 64u + (unsigned)lseg * 16u;

    // instruction i (0..4) of this wave's share of the slot of (pass, chunk c): 20 instructions of 16 rows, 5 per wave; the
    // wave-uniform part of the source offset rides in the scalar offset
    auto issue_w = [&](int slot, int pass, int c, int i) {
        const int idx = 5 * wave + i;
        dma16(rs_w, ring + slot * SLOT + idx * 1024, w_lane, (pass * NCH + c) * SLOT + idx * 1024);
    };
    // rows: a_rows[j] = source row of this lane's staging row j (16 (2 wave + j) + srow) of the current tile, n_rows: of the next
    unsigned a_voff[2] = {OOB, OOB}, n_voff[2] = {OOB, OOB};
    int ids2[2] = {0, 0};                              // a_ids of the tile after next (plain loads, consumed at the next tile's first step)
    int cid_cur[2] = {0, 0}, cid_next[2] = {0, 0};     // c_ids of this lane's two OUTPUT rows (32 wave + 16 tt + fi), this tile / the next
    // (id arrays through buffer loads with 32-bit offsets: out of range -> 0, and no 64-bit address pairs to carry)
    const __amdgpu_buffer_rsrc_t rs_cid = make_rsrc(p.c_ids ? (const void*)p.c_ids : (const void*)p.wp);
    const __amdgpu_buffer_rsrc_t rs_aid = make_rsrc(p.a_ids ? (const void*)p.a_ids : (const void*)p.wp);
    auto load_cids = [&](long t, int* dst) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const long row = t * BM + 32 * wave + 16 * tt + fi;
            dst[tt] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_cid, (p.c_ids && row < M) ? (unsigned)row * 4u : OOB, 0, 0);
        }
    };
    auto row_of = [&](long t, int j) { return t * BM + 16 * (2 * wave + j) + srow; };
    auto voff_of = [&](long row, int id) {
        return row < M ? (p.a_ids ? (unsigned)id : (unsigned)row) * ldab + (unsigned)lseg * 16u : OOB;
    };
    auto load_ids = [&](long t, int* dst) {            // ids of tile t (any t: out of range reads row 0)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long row = row_of(t, j);
            dst[j] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_aid, (p.a_ids && row < M) ? (unsigned)row * 4u : OOB, 0, 0);
        }
    };
    auto issue_x = [&](const unsigned* vo, int c) {    // chunk c of this wave's 32 rows -> the image
#pragma unroll
        for (int j = 0; j < 2; ++j) dma16(rs_a, lds + c * SLAB + (2 * wave + j) * 1024, (c * 32 + lseg * 8 < p.K) ? vo[j] : OOB, c * 64);
    };

    const int pseg = (kg ^ swz4((fi >> 2) & 3)) * 16;
    const int x_off = (32 * wave + fi) * 64 + pseg;
    const int w_off = fi * 64 + pseg;
    f32x4 acc[2][NT];
    bf16x8 nw[4], nx[2];
    auto prefetch = [&](int slot, int xc) {
        const unsigned char* const sb = ring + slot * SLOT + w_off;
#pragma unroll
        for (int t = 0; t < 4; ++t) nw[t] = *reinterpret_cast<const bf16x8*>(sb + t * 1024);
        if (xc >= 0) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) nx[tt] = *reinterpret_cast<const bf16x8*>(lds + xc * SLAB + x_off + tt * 1024);
        }
    };
    // one step: 20 tiles x 2 token halves in five groups of four tiles; part(g) behind group g
    auto compute = [&](int slot, const bf16x8& b0, const bf16x8& b1, auto&& part, auto&& tail) {
        const unsigned char* const sb = ring + slot * SLOT + w_off;
        constexpr int GT = 4, NG = NT / GT;
        bf16x8 wf[2][GT];
#pragma unroll
        for (int t = 0; t < GT; ++t) wf[0][t] = nw[t];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) {
#pragma unroll
                for (int t = 0; t < GT; ++t) wf[(g + 1) & 1][t] = *reinterpret_cast<const bf16x8*>(sb + ((g + 1) * GT + t) * 1024);
            } else {
                tail();
            }
#pragma unroll
            for (int t = 0; t < GT; ++t)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    acc[tt][g * GT + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[g & 1][t], tt ? b1 : b0, acc[tt][g * GT + t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            part(g);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    int gs = 0;
    bool last = false;
    // Step (pass, chunk C): reads slot gs & 3; in front of it this wave's part of the next step's slot has landed (counted wait: the
    // slot after that -- and, in a tile's last pass, the image refill of the previous step -- stay in flight), then the barrier.
    auto step = [&](auto c_c, auto lp_c, int pass) {
        constexpr int C = decltype(c_c)::value;
        constexpr bool LASTP = decltype(lp_c)::value;    // the tile's last pass: the image is refilled behind the steps
        const bool tile_start = C == 0 && pass == 0;
        if (tile_start) {
            wait_vm<0>();
        } else if (LASTP && last && C >= NCH - 2) {
            wait_vm<0>();                              // the ring runs dry behind the last tile
        } else if (LASTP && C >= 2 && !last) {
            wait_vm<7>();                              // + the two image-refill DMAs of the previous step
        } else {
            wait_vm<5>();
        }
        ring_barrier();
        if constexpr (LASTP) {
            if (C >= 1 && !last) issue_x(n_voff, C - 1);   // chunk C - 1 is done with (this wave's rows): the next tile's rows move in
        }
        int fp = pass, fc = C + 3;
        bool go = true;
        if (fc >= NCH) {
            fc -= NCH;
            ++fp;
            if (fp == NP) { fp = 0; go = !last; }
        }
        const int fslot = (gs + 3) & 3;
        auto part = [&](int g) {
            if (go) issue_w(fslot, fp, fc, g);
        };
        const int nxc = C + 1 < NCH ? C + 1 : ((LASTP) ? -1 : 0);      // the next step's chunk while the image it reads is in place
        auto tail = [&]() { prefetch((gs + 1) & 3, nxc); };
        bf16x8 a0, a1;
        if (tile_start) {
            a0 = *reinterpret_cast<const bf16x8*>(lds + x_off);
            a1 = *reinterpret_cast<const bf16x8*>(lds + x_off + 1024);
        } else {
            a0 = nx[0];
            a1 = nx[1];
        }
        compute(gs & 3, a0, a1, part, tail);
        __builtin_amdgcn_sched_barrier(0);
        ++gs;
    };

    // prologue: the first three slots, this tile's rows, the next tile's ids
    int ids1[2];
    load_ids(tile, ids1);
#pragma unroll
    for (int j = 0; j < 2; ++j) a_voff[j] = voff_of(row_of(tile, j), ids1[j]);
    load_ids((long)tile + stride, ids1);
    load_ids((long)tile + 2 * stride, ids2);
    load_cids(tile, cid_next);
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 5; ++i) issue_w(q, 0, q, i);
#pragma unroll
    for (int c = 0; c < NCH; ++c) issue_x(a_voff, c);
    wait_vm<10 + 2 * NCH>();                           // slot 0
    ring_barrier();
    prefetch(0, -1);

    const __amdgpu_buffer_rsrc_t rs_add = make_rsrc(p.add);
    for (; tile < ntiles; tile += stride) {
        last = tile + stride >= ntiles;
        // (behind the first step's wait everything issued so far has landed: the id loads cost no drain of their own)
        const long row0 = (long)tile * BM;
        constexpr int EB = 4, NEB = NT / EB;               // the pass epilogue runs in five batches of four column tiles
        f32x4 pe[2][2][EB];                                // fp32 rows of the batch in hand / the next one
        unsigned pof[2], cof[2];
        for (int pass = 0; pass < NP; ++pass) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[tt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            const bool lastp = pass == NP - 1;
            // The weight rows of a 32-column block are packed so that tiles 2 u and 2 u + 1 hold, in a lane, the EIGHT consecutive
            // output columns 32 u + 8 kg .. + 7 (lime_inproj_pack_bf16): a row's results leave as 16-byte stores, half as many as
            // with the natural order (the scattered bf16 stores are issue bound).
            auto load_add = [&](int b) {               // fp32 rows of column tiles 4 b .. 4 b + 3 -> pe[b & 1]
#pragma unroll
                for (int v = 0; v < EB; ++v)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int t = b * EB + v;
                        pe[b & 1][tt][v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                            rs_add, pof[tt] == OOB ? OOB : pof[tt] + (unsigned)(t >> 1) * 128u + (unsigned)(t & 1) * 16u, pass * PWD * 4, 0));
                    }
            };
            auto steps = [&](auto lp_c) {
                step(std::integral_constant<int, 0>{}, lp_c, pass);
                if (pass == 0) {
                    // this tile's output rows; the rows the NEXT tile reads (its ids came in a tile ago), the ids of the tile after it
                    const int le = lane_here();
                    const int fi_ = le & 15, kg_ = le >> 4;
                    cid_cur[0] = cid_next[0];
                    cid_cur[1] = cid_next[1];
                    load_cids((long)tile + stride, cid_next);
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const long row = row0 + 32 * wave + 16 * tt + fi_;
                        if (row < M) {
                            const unsigned orow = p.c_ids ? (unsigned)cid_cur[tt] : (unsigned)row;       // 32-bit: a 64-bit % is a long routine
                            pof[tt] = (orow % (unsigned)p.period) * (unsigned)(p.ld_add * 4) + (unsigned)kg_ * 32u;
                            // scattered rows: offsets from the base of out; rows in place: from the tile's first row (any M)
                            cof[tt] = (p.c_ids ? orow : (unsigned)(row - row0)) * (unsigned)(p.ldo * 2) + (unsigned)kg_ * 16u;
                        } else {
                            pof[tt] = OOB;
                            cof[tt] = OOB;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) n_voff[j] = voff_of(row_of((long)tile + stride, j), ids1[j]);
                    ids1[0] = ids2[0];
                    ids1[1] = ids2[1];
                    load_ids((long)tile + 3L * stride, ids2);
                }
                step(std::integral_constant<int, 1>{}, lp_c, pass);
                step(std::integral_constant<int, 2>{}, lp_c, pass);
                step(std::integral_constant<int, 3>{}, lp_c, pass);
                step(std::integral_constant<int, 4>{}, lp_c, pass);
                step(std::integral_constant<int, 5>{}, lp_c, pass);
                step(std::integral_constant<int, 6>{}, lp_c, pass);
                step(std::integral_constant<int, 7>{}, lp_c, pass);
                load_add(0);                           // two steps ahead of the epilogue; the other batches while the one before is stored
                step(std::integral_constant<int, 8>{}, lp_c, pass);
                step(std::integral_constant<int, 9>{}, lp_c, pass);
            };
            if (lastp) steps(std::true_type{});
            else steps(std::false_type{});
            if (lastp && !last) issue_x(n_voff, NCH - 1);
            mfma_settle();
            // pass epilogue: + the fp32 rows, bf16, to row c_ids[r]
            const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(p.c_ids ? p.out : p.out + row0 * p.ldo);
#pragma unroll
            for (int b = 0; b < NEB; ++b) {
                if (b + 1 < NEB) load_add(b + 1);
#pragma unroll
                for (int u = 0; u < EB / 2; ++u)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const f32x4 v0 = acc_read4(acc[tt][b * EB + 2 * u]) + pe[b & 1][tt][2 * u];
                        const f32x4 v1 = acc_read4(acc[tt][b * EB + 2 * u + 1]) + pe[b & 1][tt][2 * u + 1];
                        u32x4 o;
                        o[0] = pack_bf16(v0[0], v0[1]);
                        o[1] = pack_bf16(v0[2], v0[3]);
                        o[2] = pack_bf16(v1[0], v1[1]);
                        o[3] = pack_bf16(v1[2], v1[3]);
                        __builtin_amdgcn_raw_buffer_store_b128(o, rs_c, cof[tt] == OOB ? OOB : cof[tt] + (unsigned)(b * (EB / 2) + u) * 64u,
                                                               pass * PWD * 2, 0);
                    }
            }
        }
    }
}

// w fp32 [N, K] (ld ldw; N a multiple of 320: the heads already padded to 32 columns) -> bf16 [N / 320][10][320 rows][32 k], zero beyond
// K, the rows of every 32-row block in the order the pass epilogue stores them
__global__ void inproj_pack_kernel(const float* __restrict__ w, long ldw, int N, int K, uint16_t* __restrict__ wp) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * (NCH * 32)) return;
    const int kk = (int)(i & 31), r = (int)((i >> 5) % PWD), c = (int)((i / (32 * PWD)) % NCH), pass = (int)(i / (32L * PWD * NCH));
    // MFMA row r = 16 t + 4 kg + q of the pass (tile t) computes output column 32 (t >> 1) + 8 kg + 4 (t & 1) + q
    const int t = r >> 4, kgq = r & 15;
    const int n = PWD * pass + 32 * (t >> 1) + 8 * (kgq >> 2) + 4 * (t & 1) + (kgq & 3), k = 32 * c + kk;
    wp[i] = k < K ? (uint16_t)(pack_bf16(w[n * ldw + k], 0.f) & 0xFFFFu) : (uint16_t)0;
}

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

}  // namespace

extern "C" int64_t lime_inproj_pack_bf16_size(int32_t N) { return (int64_t)N * (NCH * 32); }

extern "C" int lime_inproj_pack_bf16(const float* w, int64_t ldw, int32_t N, int32_t K, uint16_t* wp, void* stream) {
    LIME_REQUIRE(w && wp, LIME_ERR_BAD_ARG, "lime_inproj_pack_bf16: NULL pointer");
    LIME_REQUIRE(N > 0 && N % PWD == 0 && K > 0 && K <= NCH * 32 && ldw >= K, LIME_ERR_UNSUPPORTED,
                 "lime_inproj_pack_bf16: built for N a multiple of %d (N = %d) and K <= %d (K = %d)", PWD, N, NCH * 32, K);
    const long n = (long)N * (NCH * 32);
    hipLaunchKernelGGL(inproj_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (long)ldw, N, K, wp);
    return lime_check_launch("lime_inproj_pack_bf16");
}

extern "C" int lime_inproj_bf16(const lime_inproj_bf16_args* a, void* stream) {
    LIME_REQUIRE(a != nullptr, LIME_ERR_BAD_ARG, "lime_inproj_bf16: args is NULL");
    LIME_REQUIRE(a->a && a->wp && a->add_rows && a->out, LIME_ERR_BAD_ARG, "lime_inproj_bf16: NULL pointer");
    LIME_REQUIRE(a->M >= 0 && a->N > 0 && a->N % PWD == 0 && a->K > 0 && a->K <= NCH * 32 && a->K % 8 == 0, LIME_ERR_UNSUPPORTED,
                 "lime_inproj_bf16: built for N a multiple of %d (N = %d) and K <= %d, K %% 8 == 0 (K = %d)", PWD, a->N, NCH * 32, a->K);
    LIME_REQUIRE(a->lda >= a->K && a->lda % 8 == 0 && (uintptr_t)a->a % 16 == 0 && (uintptr_t)a->wp % 16 == 0, LIME_ERR_BAD_ARG,
                 "lime_inproj_bf16: a rows / the packed weight must be 16-byte aligned (lda %% 8 == 0)");
    LIME_REQUIRE(a->add_period > 0 && a->ld_add >= a->N && a->ld_add % 4 == 0 && (uintptr_t)a->add_rows % 16 == 0, LIME_ERR_BAD_ARG,
                 "lime_inproj_bf16: add_rows must be fp32 [add_period >= 1, >= N], 16-byte aligned rows");
    LIME_REQUIRE(a->ldo >= a->N && a->ldo % 4 == 0 && (uintptr_t)a->out % 8 == 0, LIME_ERR_BAD_ARG, "lime_inproj_bf16: out rows must be 8-byte aligned");
    const long lim = 0x7FFFFFF0L;
    LIME_REQUIRE((long)a->a_rows * a->lda * 2 < lim && (long)(a->c_ids ? a->out_rows : 128) * a->ldo * 2 < lim &&
                 (long)a->add_period * a->ld_add * 4 < lim && (long)a->N * 320 * 2 < lim && (long)a->M * 4 < lim, LIME_ERR_UNSUPPORTED,
                 "lime_inproj_bf16: operand too large for 32-bit offsets (a rows x lda, scattered out rows x ldo < 2 GB)");
    LIME_REQUIRE(a->a_ids || a->a_rows >= a->M, LIME_ERR_BAD_ARG, "lime_inproj_bf16: a has fewer rows than M");
    LIME_REQUIRE(a->c_ids || a->out_rows >= a->M, LIME_ERR_BAD_ARG, "lime_inproj_bf16: out has fewer rows than M");
    if (a->M == 0) return LIME_OK;
    InP p{};
    p.a = a->a; p.lda = a->lda; p.a_ids = a->a_ids; p.wp = a->wp;
    p.add = a->add_rows; p.ld_add = a->ld_add; p.period = a->add_period;
    p.c_ids = a->c_ids; p.out = a->out; p.ldo = a->ldo;
    p.M = a->M; p.N = a->N; p.K = a->K; p.m_dev = a->m_dev;
    const long ntiles = ((long)a->M + BM - 1) / BM;
    long nwg = num_cus();
    if (nwg > ntiles) nwg = ntiles;
    hipLaunchKernelGGL(inproj_bf16_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, p);
    return lime_check_launch("lime_inproj_bf16");
}
