// Backward of the fused tail kernels of small_ops.hip (training step): each kernel recomputes
// the few forward intermediates it needs (they are cheaper to redo than to keep) and writes every gradient of its stage in
// one launch.  Reductions over the rows (bias / LayerNorm / affine2 gradients) go through per-workgroup partials in a caller
// workspace and are summed in a fixed order by reduce_rows_kernel -- no atomics.
//
//   intent_fuse_bwd_kernel      layers.Attention over the k intents + cosine similarity + concat   (newsEncoders.py:355-371)
//   gate_ln_bwd_kernel          gated residual + LayerNorm of CandidateAware_ClickedNewsAttention  (layers.py:84-89)
//   interest_match_bwd_kernel   history-vs-candidate attention + dot product + lifetime weight     (userEncoders.py:158-169, util.py:23-49)
//   cand_attn_train_kernel      candidate-aware attention weights, forward WITH the p = 0.2 dropout of layers.py:74 (training
//                               mode) and backward                                                 (layers.py:66-81)
#include "common.h"
#include "dropout.h"

namespace {

constexpr int MAX_INTENT = 8;

// sum over the 256 threads; `red` >= 4 floats of LDS; every thread gets the total
__device__ __forceinline__ float bsum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// out[c] (+)= sum_s part[s * stride + c].  A workgroup owns 64 columns; its sixteen waves take the splits s = wave, wave + 16, ...
// (coalesced 256-byte reads; with four waves a 768-split sum was 192 loads per lane: 18 us, five of them per training step) and the
// sixteen sums are added in a fixed order.
__global__ __launch_bounds__(1024) void reduce_rows_kernel(const float* __restrict__ part, long stride, int splits, float* __restrict__ out,
                                                            int cols, int accumulate) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < cols) {
        int i = g;
        for (; i + 112 < splits; i += 128) {               // eight loads in flight, added in split order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(long)(i + 16 * u) * stride + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; i < splits; i += 16) s += part[(long)i * stride + c];
    }
    red[g][lane] = s;
    __syncthreads();
    if (g == 0 && c < cols) {
        float t = red[0][lane];
#pragma unroll
        for (int u = 1; u < 16; ++u) t += red[u][lane];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// ---------------------------------------------------------------------------------------------------
// intent fuse backward.  Forward (per news m, halves tb = title / body): a_k = hidden_k . aff2, alpha = softmax_k(a),
// pooled_tb = sum_k alpha_k intents_k;  s = (cos(t, b) + 1) / 2;  content = [t, s * b].
// Persistent grid: a workgroup walks the rows m = blockIdx.x, + gridDim.x, ... and keeps its share of d aff2 in registers.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void intent_fuse_bwd_kernel(const float* __restrict__ intents, const float* __restrict__ hidden,
                                                               const float* __restrict__ aff2_t, const float* __restrict__ aff2_b,
                                                               const float* __restrict__ dcontent, long ldc, float* __restrict__ d_intents,
                                                               float* __restrict__ d_hidden, float* __restrict__ ws, long M, int k,
                                                               int D, int A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // pooled [2][D], dpool [2][D], red [4]
    float* pooled = sm;
    float* dpool = sm + 2 * D;
    float* red = sm + 4 * D;
    constexpr int APT = 4;                                       // affine2 columns per thread (A <= 1024)
    float da2[2][APT];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int u = 0; u < APT; ++u) da2[tb][u] = 0.f;
    for (long m = blockIdx.x; m < M; m += gridDim.x) {
        float alpha[2][MAX_INTENT];
        for (int tb = 0; tb < 2; ++tb) {
            const float* aff2 = tb == 0 ? aff2_t : aff2_b;
            const float* hid = hidden + ((long)tb * M + m) * k * A;
            const float* itn = intents + ((long)tb * M + m) * k * D;
            float mx = -INFINITY;
            for (int kk = 0; kk < k; ++kk) {
                float part = 0.f;
                for (int j = threadIdx.x; j < A; j += 256) part += hid[kk * A + j] * aff2[j];
                alpha[tb][kk] = bsum(part, red);
                mx = fmaxf(mx, alpha[tb][kk]);
            }
            float den = 0.f;
            for (int kk = 0; kk < k; ++kk) {
                alpha[tb][kk] = expf(alpha[tb][kk] - mx);
                den += alpha[tb][kk];
            }
            const float inv = 1.0f / den;
            for (int kk = 0; kk < k; ++kk) alpha[tb][kk] *= inv;
            for (int d = threadIdx.x; d < D; d += 256) {
                float x = 0.f;
                for (int kk = 0; kk < k; ++kk) x += alpha[tb][kk] * itn[kk * D + d];
                pooled[tb * D + d] = x;
            }
        }
        __syncthreads();
        float dot = 0.f, n1 = 0.f, n2 = 0.f, dsim = 0.f;
        for (int d = threadIdx.x; d < D; d += 256) {
            const float t = pooled[d], b = pooled[D + d];
            dot += t * b;
            n1 += t * t;
            n2 += b * b;
            dsim += dcontent[m * ldc + D + d] * b;
        }
        dot = bsum(dot, red);
        n1 = bsum(n1, red);
        n2 = bsum(n2, red);
        dsim = bsum(dsim, red);
        const float nt_raw = sqrtf(n1), nb_raw = sqrtf(n2);
        const float nt = fmaxf(nt_raw, 1e-8f), nb = fmaxf(nb_raw, 1e-8f);
        const float cosv = dot / (nt * nb);
        const float s = (cosv + 1.0f) * 0.5f;
        const float dcos = 0.5f * dsim;
        // a clamped norm is a constant: its derivative term drops out
        const float kt = nt_raw > 1e-8f ? cosv / (nt * nt) : 0.f, kb = nb_raw > 1e-8f ? cosv / (nb * nb) : 0.f;
        const float inv_nn = 1.0f / (nt * nb);
        for (int d = threadIdx.x; d < D; d += 256) {
            const float t = pooled[d], b = pooled[D + d];
            dpool[d] = dcontent[m * ldc + d] + dcos * (b * inv_nn - kt * t);
            dpool[D + d] = s * dcontent[m * ldc + D + d] + dcos * (t * inv_nn - kb * b);
        }
        __syncthreads();
        for (int tb = 0; tb < 2; ++tb) {
            const float* aff2 = tb == 0 ? aff2_t : aff2_b;
            const float* hid = hidden + ((long)tb * M + m) * k * A;
            const float* itn = intents + ((long)tb * M + m) * k * D;
            float* dit = d_intents + ((long)tb * M + m) * k * D;
            float* dhid = d_hidden + ((long)tb * M + m) * k * A;
            float dal[MAX_INTENT];
            float mix = 0.f;
            for (int kk = 0; kk < k; ++kk) {
                float part = 0.f;
                for (int d = threadIdx.x; d < D; d += 256) {
                    const float dp = dpool[tb * D + d];
                    part += dp * itn[kk * D + d];
                    dit[kk * D + d] = alpha[tb][kk] * dp;
                }
                dal[kk] = bsum(part, red);
                mix += alpha[tb][kk] * dal[kk];
            }
            for (int kk = 0; kk < k; ++kk) {
                const float ds = alpha[tb][kk] * (dal[kk] - mix);
#pragma unroll
                for (int u = 0; u < APT; ++u) {
                    const int j = threadIdx.x + 256 * u;
                    if (j < A) {
                        dhid[kk * A + j] = ds * aff2[j];
                        da2[tb][u] += ds * hid[kk * A + j];
                    }
                }
            }
        }
        __syncthreads();                                          // pooled / dpool are rewritten by the next row
    }
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int u = 0; u < APT; ++u) {
            const int j = threadIdx.x + 256 * u;
            if (j < A) ws[((long)blockIdx.x * 2 + tb) * A + j] = da2[tb][u];
        }
}

// ---------------------------------------------------------------------------------------------------
// gated residual + LayerNorm backward.  Forward (row r, s = scale[r]): g = sigmoid(s y + bias), v = x (1 + g (s - 1)),
// out = LayerNorm(v).  Persistent grid; column sums (d bias, d gamma, d beta) in registers, partials [grid][3][D].
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_ln_bwd_kernel(const float* __restrict__ y, const float* __restrict__ x,
                                                           const float* __restrict__ scale, const float* __restrict__ bias,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                           const float* __restrict__ dout, float* __restrict__ dy, float* __restrict__ dx,
                                                           float* __restrict__ dscale, float* __restrict__ ws, long rows, int D) {
    __shared__ float red[4];
    constexpr int DPT = 4;                                       // columns per thread (D <= 1024)
    float cb[DPT], cg[DPT], ce[DPT];
#pragma unroll
    for (int u = 0; u < DPT; ++u) cb[u] = cg[u] = ce[u] = 0.f;
    const float inv_d = 1.0f / (float)D;
    for (long r = blockIdx.x; r < rows; r += gridDim.x) {
        const float s = scale[r];
        float xv[DPT], yv[DPT], g[DPT], v[DPT], go[DPT];
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < DPT; ++u) {
            const int d = threadIdx.x + 256 * u;
            const bool in = d < D;
            xv[u] = in ? x[r * D + d] : 0.f;
            yv[u] = in ? y[r * D + d] : 0.f;
            g[u] = in ? lime_sigmoid(s * yv[u] + bias[d]) : 0.f;
            v[u] = xv[u] * (1.0f + g[u] * (s - 1.0f));
            go[u] = in ? dout[r * D + d] : 0.f;
            part += v[u];
        }
        const float mean = bsum(part, red) * inv_d;
        part = 0.f;
#pragma unroll
        for (int u = 0; u < DPT; ++u) {
            const int d = threadIdx.x + 256 * u;
            v[u] = d < D ? v[u] - mean : 0.f;
            part += v[u] * v[u];
        }
        const float rstd = 1.0f / sqrtf(bsum(part, red) * inv_d + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int u = 0; u < DPT; ++u) {
            const int d = threadIdx.x + 256 * u;
            const float xh = v[u] * rstd;
            const float gy = d < D ? go[u] * gamma[d] : 0.f;
            s1 += gy;
            s2 += gy * xh;
            cg[u] += go[u] * xh;
            ce[u] += go[u];
            v[u] = xh;                                            // keep xhat
            go[u] = gy;                                           // keep dout * gamma
        }
        s1 = bsum(s1, red) * inv_d;
        s2 = bsum(s2, red) * inv_d;
        float ds = 0.f;
#pragma unroll
        for (int u = 0; u < DPT; ++u) {
            const int d = threadIdx.x + 256 * u;
            if (d < D) {
                const float dv = rstd * (go[u] - s1 - v[u] * s2);
                const float dg = dv * xv[u] * (s - 1.0f);
                const float dpre = dg * g[u] * (1.0f - g[u]);
                dx[r * D + d] = dv * (1.0f + g[u] * (s - 1.0f));
                dy[r * D + d] = dpre * s;
                ds += dv * xv[u] * g[u] + dpre * yv[u];
                cb[u] += dpre;
            }
        }
        ds = bsum(ds, red);
        if (threadIdx.x == 0) dscale[r] = ds;
    }
#pragma unroll
    for (int u = 0; u < DPT; ++u) {
        const int d = threadIdx.x + 256 * u;
        if (d < D) {
            ws[((long)blockIdx.x * 3 + 0) * D + d] = cb[u];
            ws[((long)blockIdx.x * 3 + 1) * D + d] = cg[u];
            ws[((long)blockIdx.x * 3 + 2) * D + d] = ce[u];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// interest match backward: one workgroup per (impression row b, candidate n), as the forward.  d g and d kp of row b
// collect contributions from its N candidates: each workgroup writes its share to part_g / part_kp [b][n][H][.], which the
// host sums over n (fixed order).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void interest_match_bwd_kernel(const float* __restrict__ kp, const float* __restrict__ qp,
                                                                  const float* __restrict__ g, const float* __restrict__ cand,
                                                                  const float* __restrict__ remaining, const float* __restrict__ dlogits,
                                                                  float* __restrict__ dqp, float* __restrict__ dcand,
                                                                  float* __restrict__ part_kp, float* __restrict__ part_g, int N, int H,
                                                                  int A, int D, float scale, float alpha_s, float beta_s, int use_weight,
                                                                  int use_penalty) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Qs = sm;                     // [A]
    float* al = Qs + A;                 // [H] attention weights
    float* dal = al + H;                // [H] d alpha, then d a
    float* du = dal + H;                // [D]
    float* red = du + D;                // [4]
    const long bn = blockIdx.x;
    const long b = bn / N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < A; e += 256) Qs[e] = qp[bn * A + e];
    __syncthreads();
    for (int h = wave; h < H; h += 4) {
        const float* krow = kp + (b * H + h) * A;
        float part = 0.f;
        for (int j = lane; j < A; j += 64) part += krow[j] * Qs[j];
        part = wave_sum(part);
        if (lane == 0) al[h] = part * scale;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int h = threadIdx.x; h < H; h += 256) mx = fmaxf(mx, al[h]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float den = 0.f;
    for (int h = threadIdx.x; h < H; h += 256) {
        const float e = expf(al[h] - mx);
        al[h] = e;
        den += e;
    }
    const float inv = 1.0f / bsum(den, red);
    for (int h = threadIdx.x; h < H; h += 256) al[h] *= inv;
    float w = 1.0f;
    if (use_weight) {
        const float r = remaining[bn];
        if (use_penalty) {
            w = lime_sigmoid(alpha_s * r);
            w = (r >= 0.f ? 1.f : 0.f) * w + (r < 0.f ? 1.f : 0.f) * beta_s * w;
        } else {
            w = lime_sigmoid(alpha_s * fabsf(r));
        }
    }
    const float dbase = dlogits[bn] * w;
    __syncthreads();
    // u, d cand = dbase * u, d u = dbase * cand
    for (int d = threadIdx.x; d < D; d += 256) {
        float u = 0.f;
        for (int h = 0; h < H; ++h) u += al[h] * g[(b * H + h) * D + d];
        dcand[bn * D + d] = dbase * u;
        du[d] = dbase * cand[bn * D + d];
    }
    __syncthreads();
    // d alpha_h = du . g[b, h, :]   and   this candidate's share of d g[b, h, :] = alpha_h du
    for (int h = wave; h < H; h += 4) {
        const float* grow = g + (b * H + h) * D;
        float* pg = part_g + (bn * H + h) * D;
        const float a = al[h];
        float part = 0.f;
        for (int d = lane; d < D; d += 64) {
            part += du[d] * grow[d];
            pg[d] = a * du[d];
        }
        part = wave_sum(part);
        if (lane == 0) dal[h] = part;
    }
    __syncthreads();
    float mix = 0.f;
    for (int h = threadIdx.x; h < H; h += 256) mix += al[h] * dal[h];
    mix = bsum(mix, red);
    for (int h = threadIdx.x; h < H; h += 256) dal[h] = al[h] * (dal[h] - mix) * scale;        // d (kp . q), scale folded in
    __syncthreads();
    // d kp share and d qp
    for (int h = wave; h < H; h += 4) {
        float* pk = part_kp + (bn * H + h) * A;
        const float da = dal[h];
        for (int j = lane; j < A; j += 64) pk[j] = da * Qs[j];
    }
    for (int j = threadIdx.x; j < A; j += 256) {
        float s = 0.f;
        for (int h = 0; h < H; ++h) s += dal[h] * kp[(b * H + h) * A + j];
        dqp[bn * A + j] = s;
    }
}

// ---------------------------------------------------------------------------------------------------
// candidate-aware attention weights in training mode.  One workgroup per impression row b.
//   s[h, n, j] = Q[n, h] . K[j, h] / sqrt(D), -1e9 where mask[j] == 0;  a = softmax_j(s);  ad = dropout(a)   (layers.py:70-74)
//   qw = softmax_n(|Q_n|_2);  pre[j] = sum_n qw[n] sum_h ad[h, n, j];  agg = softmax_j(pre)                  (:79-81)
// mode 0 writes agg; mode 1 recomputes the above and writes dQ [N, D], dK [H, D] from d agg.
// LDS: Q [N D], K [H D], a and ad [n_head N H] each, a few vectors.  H <= 256 (one to four keys per lane), N <= 16.
// ---------------------------------------------------------------------------------------------------
constexpr int CA_MAXN = 16;

__global__ __launch_bounds__(256) void cand_attn_train_kernel(const float* __restrict__ qp, const float* __restrict__ kp,
                                                               const unsigned char* __restrict__ mask, const float* __restrict__ dagg,
                                                               float* __restrict__ agg_out, float* __restrict__ dqp, float* __restrict__ dkp,
                                                               int N, int H, int D, int n_head, float inv_scale, LimeDropout drop,
                                                               int mode) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hd = D / n_head;
    float* Qs = sm;                              // [N][D]
    float* Ks = Qs + N * D;                      // [H][D]
    float* As = Ks + H * D;                      // [n_head][N][H]  a, later ds
    float* Ad = As + n_head * N * H;             // [n_head][N][H]  dropped a, later the dropout factor
    float* pre = Ad + n_head * N * H;            // [H]  pre, then agg, then dpre
    float* qn = pre + H;                         // [CA_MAXN] norms
    float* qw = qn + CA_MAXN;                    // [CA_MAXN]
    float* tn = qw + CA_MAXN;                    // [CA_MAXN] scratch per candidate
    float* red = tn + CA_MAXN;                   // [4]
    const long b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < N * D; e += 256) Qs[e] = qp[b * N * D + e];
    for (int e = tid; e < H * D; e += 256) Ks[e] = kp[b * H * D + e];
    __syncthreads();
    // rows (h, n) of the score tensor: a wave per row, lanes over the keys
    for (int row = wave; row < n_head * N; row += 4) {
        const int h = row / N, n = row - h * N;
        const float* qv = Qs + n * D + h * hd;
        float sv[4];
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = lane + 64 * u;
            float d = -INFINITY;
            if (j < H) {
                d = 0.f;
                const float* kv = Ks + j * D + h * hd;
                for (int e = 0; e < hd; ++e) d += qv[e] * kv[e];
                d = mask[b * H + j] ? d * inv_scale : -1e9f;
            }
            sv[u] = d;
            mx = fmaxf(mx, d);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sv[u] = (lane + 64 * u < H) ? expf(sv[u] - mx) : 0.f;
            sum += sv[u];
        }
        const float inv = 1.0f / wave_sum(sum);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = lane + 64 * u;
            if (j < H) {
                const float a = sv[u] * inv;
                const float f = (drop.thresh == 0 || lime_keep(drop, (uint64_t)(((b * n_head + h) * N + n) * (long)H + j))) ? drop.scale : 0.f;
                As[row * H + j] = a;
                Ad[row * H + j] = a * f;
            }
        }
    }
    // candidate norms and their softmax
    for (int n = wave; n < N; n += 4) {
        float part = 0.f;
        for (int e = lane; e < D; e += 64) part += Qs[n * D + e] * Qs[n * D + e];
        part = wave_sum(part);
        if (lane == 0) qn[n] = sqrtf(part);
    }
    __syncthreads();
    if (tid == 0) {
        float mx = -INFINITY, den = 0.f;
        for (int n = 0; n < N; ++n) mx = fmaxf(mx, qn[n]);
        for (int n = 0; n < N; ++n) { qw[n] = expf(qn[n] - mx); den += qw[n]; }
        for (int n = 0; n < N; ++n) qw[n] /= den;
    }
    __syncthreads();
    for (int j = tid; j < H; j += 256) {
        float p = 0.f;
        for (int n = 0; n < N; ++n) {
            float hsum = 0.f;
            for (int h = 0; h < n_head; ++h) hsum += Ad[(h * N + n) * H + j];
            p += qw[n] * hsum;
        }
        pre[j] = p;
    }
    __syncthreads();
    {   // agg = softmax_j(pre)
        float mx = -INFINITY;
        for (int j = tid; j < H; j += 256) mx = fmaxf(mx, pre[j]);
        mx = wave_max(mx);
        __syncthreads();
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float den = 0.f;
        for (int j = tid; j < H; j += 256) den += expf(pre[j] - mx);
        den = bsum(den, red);
        for (int j = tid; j < H; j += 256) pre[j] = expf(pre[j] - mx) / den;
    }
    __syncthreads();
    if (mode == 0) {
        for (int j = tid; j < H; j += 256) agg_out[b * H + j] = pre[j];
        return;
    }
    // ---- backward ------------------------------------------------------------------------------------------------
    float mix = 0.f;
    for (int j = tid; j < H; j += 256) mix += pre[j] * dagg[b * H + j];
    mix = bsum(mix, red);
    for (int j = tid; j < H; j += 256) pre[j] = pre[j] * (dagg[b * H + j] - mix);             // d pre
    __syncthreads();
    // d qw[n] = sum_j dpre[j] sum_h ad[h, n, j]
    for (int n = wave; n < N; n += 4) {
        float part = 0.f;
        for (int j = lane; j < H; j += 64) {
            float hsum = 0.f;
            for (int h = 0; h < n_head; ++h) hsum += Ad[(h * N + n) * H + j];
            part += pre[j] * hsum;
        }
        part = wave_sum(part);
        if (lane == 0) tn[n] = part;
    }
    __syncthreads();
    if (tid == 0) {                                                  // softmax over the candidates, then the norm
        float m2 = 0.f;
        for (int n = 0; n < N; ++n) m2 += qw[n] * tn[n];
        for (int n = 0; n < N; ++n) {
            const float dqn = qw[n] * (tn[n] - m2);
            tn[n] = qn[n] > 0.f ? dqn / qn[n] : 0.f;                   // factor on Q_n
        }
    }
    __syncthreads();
    // d s[h, n, :] = a (d a - sum_j a d a), d a = qw[n] dpre[j] * dropout factor; masked keys are constants
    for (int row = wave; row < n_head * N; row += 4) {
        const int n = row % N;
        float da[4];
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = lane + 64 * u;
            da[u] = 0.f;
            if (j < H) {
                const float a = As[row * H + j];
                const float f = a > 0.f ? Ad[row * H + j] / a : 0.f;  // the dropout factor (0 or 1 / (1 - p)); a = 0 carries no gradient
                da[u] = qw[n] * pre[j] * f;
                part += a * da[u];
            }
        }
        part = wave_sum(part);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = lane + 64 * u;
            if (j < H) As[row * H + j] = mask[b * H + j] ? As[row * H + j] * (da[u] - part) * inv_scale : 0.f;
        }
    }
    __syncthreads();
    // d Q[n, h hd + e] = sum_j ds[h, n, j] K[j, .] + dnorm factor * Q;   d K[j, h hd + e] = sum_n ds[h, n, j] Q[n, .]
    for (int e = tid; e < N * D; e += 256) {
        const int n = e / D, c = e - n * D, h = c / hd;
        float s2 = 0.f;
        for (int j = 0; j < H; ++j) s2 += As[(h * N + n) * H + j] * Ks[j * D + c];
        dqp[b * N * D + e] = s2 + tn[n] * Qs[e];
    }
    for (int e = tid; e < H * D; e += 256) {
        const int j = e / D, c = e - j * D, h = c / hd;
        float s2 = 0.f;
        for (int n = 0; n < N; ++n) s2 += As[(h * N + n) * H + j] * Qs[n * D + c];
        dkp[b * H * D + e] = s2;
    }
}

// out[b][c] = sum_n part[(b * N + n) * cols + c]
__global__ __launch_bounds__(256) void sum_candidates_kernel(const float* __restrict__ part, float* __restrict__ out, long B, int N,
                                                              long cols) {
    const long total = B * cols;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long b = e / cols, c = e - b * cols;
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += part[(b * N + n) * cols + c];
        out[e] = s;
    }
}

// workgroups of the row-walking backward kernels: three per CU (with one per CU the 1760 news rows of a config-2b step were seven rows
// per workgroup, one latency chain of ~17 us each, back to back: 121 us; one row per workgroup makes the partial-row sums behind them --
// reduce_rows_kernel over as many rows -- cost more than it saves)
int persistent_grid(long rows) { return (int)(rows < 768 ? (rows < 1 ? 1 : rows) : 768); }

// Backward of additive_pool_kernel (small_ops.hip; layers.Attention over the tokens of a title, layers.py:285-300 as called at
// newsEncoders.py:591-592): one workgroup per sequence recomputes alpha = softmax_t(mask(hidden_t . a2)), then
//   dalpha_t = dout . x_t,   ds_t = alpha_t (dalpha_t - sum_u alpha_u dalpha_u)   (0 on masked tokens: their score is a constant),
//   dx_t = alpha_t dout,     dhidden_t = ds_t a2,     da2[seq] = sum_t ds_t hidden_t   (per-sequence partial rows; the caller sums them).
__global__ __launch_bounds__(256) void additive_pool_bwd_kernel(const float* __restrict__ hidden, long ldh, const float* __restrict__ aff2,
                                                                 int A, const float* __restrict__ x, long ldx, int D,
                                                                 const unsigned char* __restrict__ mask, const float* __restrict__ dout,
                                                                 long ldo, float* __restrict__ dhidden, long lddh, float* __restrict__ dx,
                                                                 long lddx, float* __restrict__ da2_part, int S) {
    __shared__ float alpha[512];
    __shared__ float ds[512];
    __shared__ float red[4];
    const long s = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* drow = dout + s * ldo;
    // scores and dalpha: a wave per token
    for (int t = wave; t < S; t += 4) {
        const float* h = hidden + (s * S + t) * ldh;
        const float* xr = x + (s * S + t) * ldx;
        float part = 0.f, dal = 0.f;
        for (int j = lane; j < A; j += 64) part += h[j] * aff2[j];
        for (int d = lane; d < D; d += 64) dal += drow[d] * xr[d];
        part = wave_sum(part);
        dal = wave_sum(dal);
        if (lane == 0) {
            alpha[t] = (mask && mask[s * S + t] == 0) ? -1e9f : part;
            ds[t] = dal;
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int t = threadIdx.x; t < S; t += 256) mx = fmaxf(mx, alpha[t]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float part = 0.f;
    for (int t = threadIdx.x; t < S; t += 256) {
        const float e = expf(alpha[t] - mx);
        alpha[t] = e;
        part += e;
    }
    const float inv = 1.0f / bsum(part, red);
    float dotp = 0.f;
    for (int t = threadIdx.x; t < S; t += 256) {
        alpha[t] *= inv;
        dotp += alpha[t] * ds[t];
    }
    const float dot = bsum(dotp, red);
    for (int t = threadIdx.x; t < S; t += 256) ds[t] = (mask && mask[s * S + t] == 0) ? 0.f : alpha[t] * (ds[t] - dot);
    __syncthreads();
    for (int t = wave; t < S; t += 4) {
        const float a = alpha[t], g = ds[t];
        float* dxr = dx + (s * S + t) * lddx;
        float* dhr = dhidden + (s * S + t) * lddh;
        for (int d = lane; d < D; d += 64) dxr[d] = a * drow[d];
        for (int j = lane; j < A; j += 64) dhr[j] = g * aff2[j];
    }
    for (int j = threadIdx.x; j < A; j += 256) {
        float acc = 0.f;
        for (int t = 0; t < S; ++t) acc += ds[t] * hidden[(s * S + t) * ldh + j];
        da2_part[s * A + j] = acc;
    }
}

}  // namespace

extern "C" int64_t lime_intent_fuse_bwd_workspace(int64_t M, int32_t A) { return (int64_t)persistent_grid(M) * 2 * A; }

extern "C" int lime_intent_fuse_bwd_f32(const float* intents, const float* hidden, const float* aff2_title, const float* aff2_body,
                                        const float* dcontent, int64_t ldc, float* d_intents, float* d_hidden, float* d_aff2_title,
                                        float* d_aff2_body, int64_t M, int32_t k, int32_t D, int32_t A, float* workspace,
                                        int64_t workspace_floats, void* stream) {
    LIME_REQUIRE(intents && hidden && aff2_title && aff2_body && dcontent && d_intents && d_hidden && d_aff2_title && d_aff2_body &&
                 workspace, LIME_ERR_BAD_ARG, "lime_intent_fuse_bwd_f32: null pointer");
    LIME_REQUIRE(M >= 0 && k >= 1 && k <= MAX_INTENT && D > 0 && A > 0 && A <= 1024 && ldc >= 2 * D, LIME_ERR_BAD_ARG,
                 "lime_intent_fuse_bwd_f32: bad dimensions (k <= %d, A <= 1024, ldc >= 2 D)", MAX_INTENT);
    const int grid = persistent_grid(M);
    LIME_REQUIRE(workspace_floats >= (int64_t)grid * 2 * A, LIME_ERR_BAD_ARG, "lime_intent_fuse_bwd_f32: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) {
        (void)hipMemsetAsync(d_aff2_title, 0, (size_t)A * 4, s);
        (void)hipMemsetAsync(d_aff2_body, 0, (size_t)A * 4, s);
        return LIME_OK;
    }
    intent_fuse_bwd_kernel<<<grid, 256, (4 * D + 4) * sizeof(float), s>>>(intents, hidden, aff2_title, aff2_body, dcontent, ldc, d_intents,
                                                                         d_hidden, workspace, M, k, D, A);
    int st = lime_check_launch("intent_fuse_bwd_kernel");
    if (st != LIME_OK) return st;
    reduce_rows_kernel<<<(A + 63) / 64, 1024, 0, s>>>(workspace, 2L * A, grid, d_aff2_title, A, 0);
    reduce_rows_kernel<<<(A + 63) / 64, 1024, 0, s>>>(workspace + A, 2L * A, grid, d_aff2_body, A, 0);
    return lime_check_launch("reduce_rows_kernel");
}

extern "C" int64_t lime_gate_ln_bwd_workspace(int64_t rows, int32_t D) { return (int64_t)persistent_grid(rows) * 3 * D; }

extern "C" int lime_gate_ln_bwd_f32(const float* y, const float* x, const float* scale, const float* bias, const float* gamma,
                                    const float* beta, float eps, const float* dout, float* dy, float* dx, float* dscale, float* dbias,
                                    float* dgamma, float* dbeta, int64_t rows, int32_t D, float* workspace, int64_t workspace_floats,
                                    void* stream) {
    LIME_REQUIRE(y && x && scale && bias && gamma && beta && dout && dy && dx && dscale && dbias && dgamma && dbeta && workspace,
                 LIME_ERR_BAD_ARG, "lime_gate_ln_bwd_f32: null pointer");
    LIME_REQUIRE(rows >= 0 && D > 0 && D <= 1024, LIME_ERR_BAD_ARG, "lime_gate_ln_bwd_f32: bad dimensions (D <= 1024)");
    const int grid = persistent_grid(rows);
    LIME_REQUIRE(workspace_floats >= (int64_t)grid * 3 * D, LIME_ERR_BAD_ARG, "lime_gate_ln_bwd_f32: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    if (rows == 0) {
        (void)hipMemsetAsync(dbias, 0, (size_t)D * 4, s);
        (void)hipMemsetAsync(dgamma, 0, (size_t)D * 4, s);
        (void)hipMemsetAsync(dbeta, 0, (size_t)D * 4, s);
        return LIME_OK;
    }
    gate_ln_bwd_kernel<<<grid, 256, 0, s>>>(y, x, scale, bias, gamma, beta, eps, dout, dy, dx, dscale, workspace, rows, D);
    int st = lime_check_launch("gate_ln_bwd_kernel");
    if (st != LIME_OK) return st;
    float* outs[3] = {dbias, dgamma, dbeta};
    for (int i = 0; i < 3; ++i) reduce_rows_kernel<<<(D + 63) / 64, 1024, 0, s>>>(workspace + (long)i * D, 3L * D, grid, outs[i], D, 0);
    return lime_check_launch("reduce_rows_kernel");
}

extern "C" int64_t lime_interest_match_bwd_workspace(int32_t B, int32_t N, int32_t H, int32_t A, int32_t D) {
    return (int64_t)B * N * H * ((int64_t)A + D);
}

extern "C" int lime_interest_match_bwd_f32(const float* kp, const float* qp, const float* g, const float* cand, const float* remaining,
                                           const float* dlogits, float* dkp, float* dqp, float* dg, float* dcand, int32_t B, int32_t N,
                                           int32_t H, int32_t A, int32_t D, float scale, float alpha, float beta, int32_t use_weight,
                                           int32_t use_penalty, float* workspace, int64_t workspace_floats, void* stream) {
    LIME_REQUIRE(kp && qp && g && cand && dlogits && dkp && dqp && dg && dcand && workspace, LIME_ERR_BAD_ARG,
                 "lime_interest_match_bwd_f32: null pointer");
    LIME_REQUIRE(!use_weight || remaining, LIME_ERR_BAD_ARG, "lime_interest_match_bwd_f32: the lifetime weight needs `remaining`");
    LIME_REQUIRE(B >= 0 && N > 0 && H > 0 && A > 0 && D > 0, LIME_ERR_BAD_ARG, "lime_interest_match_bwd_f32: bad dimensions");
    LIME_REQUIRE((size_t)(A + 2 * H + D + 4) * sizeof(float) <= 64 * 1024, LIME_ERR_UNSUPPORTED, "lime_interest_match_bwd_f32: A + 2 H + D too large");
    LIME_REQUIRE(workspace_floats >= lime_interest_match_bwd_workspace(B, N, H, A, D), LIME_ERR_BAD_ARG,
                 "lime_interest_match_bwd_f32: workspace too small");
    if (B == 0) return LIME_OK;
    hipStream_t s = (hipStream_t)stream;
    float* part_kp = workspace;
    float* part_g = workspace + (int64_t)B * N * H * A;
    interest_match_bwd_kernel<<<B * N, 256, (A + 2 * H + D + 4) * sizeof(float), s>>>(kp, qp, g, cand, remaining, dlogits, dqp, dcand, part_kp,
                                                                                   part_g, N, H, A, D, scale, alpha, beta, use_weight,
                                                                                   use_penalty);
    int st = lime_check_launch("interest_match_bwd_kernel");
    if (st != LIME_OK) return st;
    const long ck = (long)H * A, cg = (long)H * D;
    sum_candidates_kernel<<<(int)((B * ck + 255) / 256 > 4096 ? 4096 : (B * ck + 255) / 256), 256, 0, s>>>(part_kp, dkp, B, N, ck);
    sum_candidates_kernel<<<(int)((B * cg + 255) / 256 > 4096 ? 4096 : (B * cg + 255) / 256), 256, 0, s>>>(part_g, dg, B, N, cg);
    return lime_check_launch("sum_candidates_kernel");
}

namespace {
int cand_attn_train(const float* qp, const float* kp, const uint8_t* mask, const float* dagg, float* agg, float* dqp, float* dkp, int B,
                    int N, int H, int D, int n_head, float p, uint64_t seed, uint32_t site, int mode, void* stream, const char* who) {
    LIME_REQUIRE(qp && kp && mask, LIME_ERR_BAD_ARG, "%s: null pointer", who);
    LIME_REQUIRE(B >= 0 && N > 0 && N <= CA_MAXN && H > 0 && H <= 256 && D > 0 && n_head > 0 && D % n_head == 0, LIME_ERR_UNSUPPORTED,
                 "%s: needs N <= %d, H <= 256, D a multiple of n_head (N=%d H=%d D=%d n_head=%d)", who, CA_MAXN, N, H, D, n_head);
    LIME_REQUIRE(p >= 0.f && p < 1.f, LIME_ERR_BAD_ARG, "%s: dropout p outside [0, 1)", who);
    const size_t bytes = ((size_t)N * D + (size_t)H * D + 2 * (size_t)n_head * N * H + H + 3 * CA_MAXN + 4) * sizeof(float);
    LIME_REQUIRE(bytes <= 160 * 1024 - 512, LIME_ERR_UNSUPPORTED, "%s: %zu bytes of LDS needed", who, bytes);
    if (B == 0) return LIME_OK;
    static size_t configured = 0;
    if (bytes > configured) {
        const hipError_t e = hipFuncSetAttribute((const void*)cand_attn_train_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        LIME_REQUIRE(e == hipSuccess, LIME_ERR_LAUNCH, "%s: cannot reserve %zu bytes of LDS: %s", who, bytes, hipGetErrorString(e));
        configured = bytes;
    }
    cand_attn_train_kernel<<<B, 256, bytes, (hipStream_t)stream>>>(qp, kp, mask, dagg, agg, dqp, dkp, N, H, D, n_head,
                                                                  1.0f / sqrtf((float)D), lime_make_dropout(p, seed, site), mode);
    return lime_check_launch("cand_attn_train_kernel");
}
}  // namespace

extern "C" int lime_cand_attn_weights_train_f32(const float* qp, const float* kp, const uint8_t* mask, float* agg, int32_t B, int32_t N,
                                                int32_t H, int32_t D, int32_t n_head, float dropout_p, uint64_t seed, uint32_t site,
                                                void* stream) {
    LIME_REQUIRE(agg, LIME_ERR_BAD_ARG, "lime_cand_attn_weights_train_f32: null pointer");
    return cand_attn_train(qp, kp, mask, nullptr, agg, nullptr, nullptr, B, N, H, D, n_head, dropout_p, seed, site, 0, stream,
                           "lime_cand_attn_weights_train_f32");
}

extern "C" int lime_cand_attn_weights_bwd_f32(const float* qp, const float* kp, const uint8_t* mask, const float* dagg, float* dqp,
                                              float* dkp, int32_t B, int32_t N, int32_t H, int32_t D, int32_t n_head, float dropout_p,
                                              uint64_t seed, uint32_t site, void* stream) {
    LIME_REQUIRE(dagg && dqp && dkp, LIME_ERR_BAD_ARG, "lime_cand_attn_weights_bwd_f32: null pointer");
    return cand_attn_train(qp, kp, mask, dagg, nullptr, dqp, dkp, B, N, H, D, n_head, dropout_p, seed, site, 1, stream,
                           "lime_cand_attn_weights_bwd_f32");
}

/* Backward of lime_additive_pool_f32 (see additive_pool_bwd_kernel): dhidden [n_seq * S, A], dx [n_seq * S, D] and the per-sequence
 * partial rows of the affine2 gradient da2_part [n_seq, A] (summed by the caller: lime_colsum_f32). */
extern "C" int lime_additive_pool_bwd_f32(const float* hidden, int64_t ldh, const float* affine2, int32_t A, const float* x, int64_t ldx,
                                          int32_t D, const uint8_t* mask, const float* dout, int64_t ldo, float* dhidden, int64_t lddh,
                                          float* dx, int64_t lddx, float* da2_part, int32_t n_seq, int32_t S, void* stream) {
    LIME_REQUIRE(hidden && affine2 && x && dout && dhidden && dx && da2_part, LIME_ERR_BAD_ARG, "lime_additive_pool_bwd_f32: NULL pointer");
    LIME_REQUIRE(n_seq >= 0 && S > 0 && A > 0 && D > 0 && ldh >= A && ldx >= D && ldo >= D && lddh >= A && lddx >= D, LIME_ERR_BAD_ARG,
                 "lime_additive_pool_bwd_f32: bad dims");
    LIME_REQUIRE(S <= 512, LIME_ERR_UNSUPPORTED, "lime_additive_pool_bwd_f32: S %d > 512", S);
    if (n_seq == 0) return LIME_OK;
    hipLaunchKernelGGL(additive_pool_bwd_kernel, dim3((unsigned)n_seq), dim3(256), 0, (hipStream_t)stream, hidden, (long)ldh, affine2, A, x,
                       (long)ldx, D, mask, dout, (long)ldo, dhidden, (long)lddh, dx, (long)lddx, da2_part, S);
    return lime_check_launch("lime_additive_pool_bwd_f32");
}
