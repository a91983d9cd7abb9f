// savad_generic.h -- SelfAttentiveVAD.forward for ANY d_model (vad/models/self_attention.py:7-21 and
// vad/models/model_factory.py:42-48 accept every value; the reference ships 128, which has the tuned MFMA kernels of
// savad_kernels.h).  A handle created with d_model != 128 runs this path: plain fp32 kernels that follow the reference
// operation by operation on row-major [rows][features] buffers -- one LDS-tiled GEMM on the fp32 matrix cores with a fused
// epilogue (bias, positional encoding, ReLU, residual), LayerNorm with its affine part, softmax over the keys of a materialised score tile, classifier +
// log-softmax.  Correctness and the full boundary first: it is several times slower per FLOP than the d_model = 128 path
// (runtime shapes with bounds checks, no operand reuse beyond one 64 x 64 tile, the [T,T] scores make a round trip through HBM) and fp32 only.
#pragma once
#include <hip/hip_runtime.h>

namespace savad {
namespace gen {

// C[b][m][n] = alpha * sum_k A[b][m][k] * Bm[b][k * ldk + n * ldn]  (+ bias[n]) (+ add[(m % add_rows)][n]) -> relu -> (+ res[b][m][n])
// nn.Linear (weight [N][K]): ldk = 1, ldn = K.   P . V (V [K][N]): ldk = N, ldn = 1.
struct GemmArgs {
    const float* A;
    long lda, sA;
    const float* Bm;
    long ldk, ldn, sB;
    float* C;
    long ldc, sC;
    int M, N, K;
    float alpha;
    const float* bias;  // [N] or null
    const float* add;   // [add_rows][N] or null (positional encoding, already divided by sqrt(d_model))
    int add_rows;
    const float* res;   // same layout as C or null (may alias C: every element is read and written by one thread)
    int relu;
};

constexpr int GT = 64;   // output tile edge
constexpr int GK = 16;   // K step

__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    __shared__ float As[GK][GT + 1];
    __shared__ float Bs[GK][GT + 1];
    // 64 x 64 output tile per workgroup, one 32 x 32 quadrant per wave on the fp32 matrix cores (v_mfma_f32_32x32x2f32):
    // A operand = weight values (lane & 31 = output feature, lane >> 5 = k), B operand = data values (lane & 31 = data row),
    // so a lane ends up with 16 output features of ONE data row: register r = feature 8 (r / 4) + 4 (lane >> 5) + (r & 3).
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
    const long b = blockIdx.z;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    const float* A = g.A + b * g.sA;
    const float* Bm = g.Bm + b * g.sB;
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // tile loads: 64 x 16 elements each, 4 per thread.  A: row = tid / 4, k = 4 (tid % 4) + e (contiguous in k).
    // B: whichever of k / n is contiguous in memory runs fastest across the threads.
    const int ar = tid >> 2, ak = (tid & 3) * 4;
    const bool b_k_contig = g.ldk == 1;
    const int bn = b_k_contig ? (tid >> 2) : (tid & 63), bk = b_k_contig ? (tid & 3) * 4 : (tid >> 6) * 4;
    const int kh = lane >> 5, l31 = lane & 31;
    for (int k0 = 0; k0 < g.K; k0 += GK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = m0 + ar, k = k0 + ak + e;
            As[ak + e][ar] = (m < g.M && k < g.K) ? A[(long)m * g.lda + k] : 0.0f;
            const int n = n0 + bn, kb = k0 + bk + e;
            Bs[bk + e][bn] = (n < g.N && kb < g.K) ? Bm[(long)kb * g.ldk + (long)n * g.ldn] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < GK; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Bs[k + kh][wn * 32 + l31], As[k + kh][wm * 32 + l31], acc, 0, 0, 0);
        __syncthreads();
    }
    const int m = m0 + wm * 32 + l31;
    if (m >= g.M) return;
    float* C = g.C + b * g.sC + (long)m * g.ldc;
    const float* R = g.res ? g.res + b * g.sC + (long)m * g.ldc : nullptr;
    const float* addr = g.add ? g.add + (long)(m % g.add_rows) * g.N : nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
        if (n >= g.N) continue;
        float v = acc[r] * g.alpha;
        if (g.bias) v += g.bias[n];
        if (addr) v += addr[n];
        if (g.relu) v = v > 0.0f ? v : 0.0f;
        if (R) v += R[n];
        C[n] = v;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// nn.LayerNorm(D): biased variance, eps = 1e-5, affine (vad/modeling/transformer.py:22,231); one wave per row, two passes
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y, long rows, int Dm) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + row * Dm;
    float s = 0.0f;
    for (int d = lane; d < Dm; d += 64) s += xr[d];
    const float mean = wave_sum(s) / (float)Dm;
    float ss = 0.0f;
    for (int d = lane; d < Dm; d += 64) {
        const float c = xr[d] - mean;
        ss = __builtin_fmaf(c, c, ss);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)Dm + 1e-5f);
    float* yr = y + row * Dm;
    for (int d = lane; d < Dm; d += 64) yr[d] = (xr[d] - mean) * rstd * gamma[d] + beta[d];
}

// softmax over the keys (vad/modeling/transformer.py:333), in place; one wave per score row
__global__ __launch_bounds__(256) void softmax_kernel(float* __restrict__ s, long rows, int T) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* r = s + row * T;
    float mx = -INFINITY;
    for (int j = lane; j < T; j += 64) mx = fmaxf(mx, r[j]);
    mx = wave_max(mx);
    float den = 0.0f;
    for (int j = lane; j < T; j += 64) {
        const float e = expf(r[j] - mx);
        r[j] = e;
        den += e;
    }
    den = wave_sum(den);
    for (int j = lane; j < T; j += 64) r[j] = r[j] / den;
}

// classifier Linear(D, 2) + LogSoftmax(dim=2) (vad/models/self_attention.py:20-21,26-27); one wave per row
__global__ __launch_bounds__(256) void classifier_kernel(const float* __restrict__ x, const float* __restrict__ wc,
                                                         const float* __restrict__ bc, float* __restrict__ out, long rows, int Dm) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + row * Dm;
    float z0 = 0.0f, z1 = 0.0f;
    for (int d = lane; d < Dm; d += 64) {
        z0 = __builtin_fmaf(xr[d], wc[d], z0);
        z1 = __builtin_fmaf(xr[d], wc[Dm + d], z1);
    }
    z0 = wave_sum(z0) + bc[0];
    z1 = wave_sum(z1) + bc[1];
    const float mx = fmaxf(z0, z1);
    const float lse = mx + logf(expf(z0 - mx) + expf(z1 - mx));
    if (lane == 0) {
        out[row * 2] = z0 - lse;
        out[row * 2 + 1] = z1 - lse;
    }
}

// positional-encoding table for any even d_model, already divided by sqrt(d_model) (vad/modeling/transformer.py:389-414)
inline void build_pe_host(float* pe, int T, int Dm) {
    const float cexp = (float)(-(log(10000.0) / (double)Dm));
    const float scale = (float)sqrt((double)Dm);
    for (int i = 0; i < Dm / 2; ++i) {
        const float arg = (float)(2 * i) * cexp;
        const float wv = (float)exp((double)arg);
        for (int t = 0; t < T; ++t) {
            const float a = (float)t * wv;
            pe[(size_t)t * Dm + 2 * i] = (float)sin((double)a) / scale;
            pe[(size_t)t * Dm + 2 * i + 1] = (float)cos((double)a) / scale;
        }
    }
}

// workspace (floats): h | n | q | k | v | ctx (rows x d_model each), ff (rows x 4 d_model), one score tile
constexpr size_t SCORE_CAP = (size_t)32 << 20;   // floats: 128 MiB of scores per attention pass
struct Plan {
    size_t h, n, q, k, v, ctx, ff, scores, total;  // float offsets
    int cb, tq;                                    // sequences and query rows per score tile
};
inline Plan plan(int B, int T, int Dm, int force_query_tiles) {
    Plan p;
    const size_t md = (size_t)B * T * Dm;
    size_t off = 0;
    auto take = [&](size_t n) {
        const size_t o = off;
        off += (n + 63) & ~size_t(63);
        return o;
    };
    p.h = take(md);
    p.n = take(md);
    p.q = take(md);
    p.k = take(md);
    p.v = take(md);
    p.ctx = take(md);
    p.ff = take(4 * md);
    const size_t tt = (size_t)T * T;
    if (force_query_tiles > 1 || tt > SCORE_CAP) {
        p.cb = 1;
        long tq = force_query_tiles > 1 ? (T + force_query_tiles - 1) / force_query_tiles : (long)(SCORE_CAP / T);
        p.tq = (int)(tq < 1 ? 1 : (tq > T ? T : tq));
    } else {
        size_t cb = SCORE_CAP / tt;
        cb = cb > 65535 ? 65535 : cb;  // cb is a grid.z of the score GEMMs (HIP: at most 65535); the b0 loop takes the rest
        p.cb = (int)(cb > (size_t)B ? (size_t)B : cb);
        p.tq = T;
    }
    p.scores = take((size_t)p.cb * p.tq * T);
    p.total = off;
    return p;
}

}  // namespace gen
}  // namespace savad
