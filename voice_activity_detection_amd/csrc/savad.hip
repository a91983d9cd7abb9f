// savad.hip -- host side of libsavad.so: the C ABI declared in include/savad.h.
// Owns the packed weights, the positional-encoding cache, and the 7-launch forward schedule:
//   input_qkv -> [attention(l) -> row(l)] x L      (row(L-1) ends in classifier + log-softmax)
#include "savad_kernels.h"
#include "savad_kernels_bf16.h"
#include "savad_attn_pw_bf16.h"
#include "savad_packed_bf16.h"
#include "savad_kernels_f32s.h"
#include "savad_generic.h"
#include <type_traits>
#include "savad_logmel.h"
#include "savad_post.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/savad.h"

#define SAVAD_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(SAVAD_E_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                          __FILE__, __LINE__);                                        \
    } while (0)

struct Param {
    std::string key;
    size_t numel;
    size_t off;  // float offset into d_raw
    bool set;
};

constexpr int MAX_EVENTS = 64;

}  // namespace

struct savad_model {
    savad_config cfg;
    bool generic = false;  // d_model != 128: the plain fp32 kernels of savad_generic.h on the raw parameters
    std::vector<Param> params;
    float* d_raw = nullptr;     // parameters exactly as handed over (state_dict layout)
    float* d_packed = nullptr;  // LayerNorm-folded weights
    size_t raw_floats = 0, packed_floats = 0;
    bool dirty = true;
    // positional-encoding cache (mirrors SinusoidalPositionalEncoding: rebuilt when T grows,
    // vad/modeling/transformer.py:392-397; initial length 10: vad/models/self_attention.py:14)
    float* d_pe = nullptr;
    int pe_len = 0;
    std::vector<float> h_pe;
    int splits = 0;
    int row_mode = 0;  // 0 auto, 1 N-split (32-row tiles), 2 M-split (128-row tiles)
    bool batch_invariant = false;   // bf16: the persistent attention kernel without key-split tail items (savad_set_batch_invariant)
    int precision = 0;  // 0 = fp32 MFMA, 1 = bf16 MFMA operands (fp32 accumulate / statistics / residual stream),
                        // 2 = "fp32s": fp32 parity on the bf16 pipe, every operand as three bf16 pieces (savad_kernels_f32s.h)
    char* d_frag3 = nullptr;  // fp32s weight triples (savad_kernels_f32s.h), filled when precision == 2
    size_t frag3_bytes = 0;
    bool frag3_dirty = true;
    bool lds_attrs3_set = false;
    size_t f3_win = 0;
    unsigned* d_sat = nullptr;  // bf16 path: elements of the fp16-stored residual stream that saturated since the last query
    char* d_frag = nullptr;  // bf16 weight fragments (savad_kernels_bf16.h), filled when precision == 1
    size_t frag_bytes = 0;
    bool frag_dirty = true;
    bool lds_attrs_set = false;  // hipFuncSetAttribute(MaxDynamicSharedMemorySize) done for the bf16 kernels
    int n_cu = 256;              // compute units of the handle's device (launch-shape decisions)
    size_t f_win = 0;
    struct LayerFrag {
        size_t wqkv, wo, w1, w2;
    };
    std::vector<LayerFrag> lf;
    std::vector<LayerFrag> lf3;  // byte offsets into d_frag3
    // profiling
    int prof_capacity = 0, prof_used = 0, prof_nk = 0, prof_skip = 0;
    std::vector<hipEvent_t> events;  // prof_capacity * MAX_EVENTS
    std::vector<const char*> knames;

    // raw offsets
    size_t r_win, r_bin, r_lnf_w, r_lnf_b, r_wc, r_bc;
    struct LayerRaw {
        size_t wq, bq, wk, bk, wv, bv, wo, bo, ln1w, ln1b, w1, b1, w2, b2, ln2w, ln2b;
    };
    std::vector<LayerRaw> lr;
    // packed offsets
    struct LayerPacked {
        size_t wqkv, bqkv, w1, b1;
        size_t frag;  // the layer's matrices in fragment order (packed_forward_kernel)
    };
    std::vector<LayerPacked> lp;
    size_t p_wc, p_bc;
    size_t p_bias = 0;  // [L][LBIAS] b1' | b2 | bqkv' | bo (packed_forward_kernel stages them in one sweep)
    int FP = 0;           // feature size rounded up to a multiple of 16 (kernels' K granularity)
    size_t p_win_pad = 0;  // [D][FP] zero-padded copy of input_layer.0.weight (only when FP != feature_size)
};

namespace {

using namespace savad;

size_t add_param(savad_model* m, const std::string& key, size_t numel) {
    const size_t off = m->raw_floats;
    m->params.push_back(Param{key, numel, off, false});
    m->raw_floats += (numel + 3) & ~size_t(3);  // keep every tensor 16-byte aligned
    return off;
}

int choose_splits(const savad_model* m, int B, int T) {
    if (T <= 32) return 1;
    const int NT = (T + 31) / 32, QB = NT;
    if (m->splits > 0) return m->splits < NT ? m->splits : NT;
    // Work quantisation model (MFMA-bound): a workgroup puts one wave on each SIMD of a CU, so a
    // CU that receives n workgroups needs n * ceil(NT/S) tile-times whether or not they are
    // co-resident; prologue + epilogue + partial write/re-read cost about 1.5 tile-times per
    // workgroup.  Measured at B=32, T=800: S=1 120 us, S=2 119 us (+4 us in the row kernel), S=5
    // 121 us (+25 us): splitting only pays when it fills idle CUs (small batches).
    auto cost = [&](int S) {
        const long wgs = (long)B * ((QB + 3) / 4) * S;
        return (double)((wgs + 255) / 256) * ((NT + S - 1) / S + 1.5);
    };
    const double cost1 = cost(1);
    double best = cost1;
    int bestS = 1;
    for (int S = 2; S <= 8 && S <= NT; ++S) {
        const double cs = cost(S);
        if (cs < 0.93 * cost1 && cs < best) {
            best = cs;
            bestS = S;
        }
    }
    return bestS;
}

struct Workspace {
    size_t rows, rows_pad;
    int S;
    bool msplit;  // row-wise stages on 128-row tiles with the weight stream shared through LDS
    bool fused;   // attention + row chain in one launch per layer (q/k/v double-buffered: q2, k2, v2)
    size_t h, q, k, v, q2, k2, v2, opart, ml, xpad, total;  // float offsets
};

Workspace plan(const savad_model* m, int B, int T) {
    Workspace w;
    w.rows = (size_t)B * T;
    w.rows_pad = (w.rows + 127) / 128 * 128;  // whole 128-row tiles (M-split kernels); also a multiple of TILE
    w.S = choose_splits(m, B, T);
    // Row-wise stages: 128-row tiles with the weight stream shared through LDS (M split) when that
    // fills the chip; 32-row tiles with the output features split over the 4 waves (N split) when
    // the batch is small and the critical path per workgroup matters more than weight traffic.
    // N-split works through ceil(tiles / 256) rounds of ~45 us, M-split through one round of ~118 us per 256 workgroups
    // of 128 rows: M wins from the third N-split round on (more than 512 tiles of 32 rows).  Measured at T=800: B=20
    // (500 tiles) N 0.504 / M 0.648 ms; B=24 (600 tiles) N 0.611 / M 0.589 ms.
    const int row_mode = m->row_mode == 4 ? 0 : m->row_mode;  // 4 only differs from automatic for T <= 32 (savad_forward)
    w.msplit = row_mode == 2 || row_mode == 3 || (row_mode == 0 && w.rows_pad / 32 > 512);
    // In the M-split regime without key splits the attention stage and the row chain of a query-block group
    // run back to back in one workgroup (attention_row_kernel).  row_mode 2 keeps them as separate launches.
    // Automatic: only when a query-block group keeps at least 80 % of its 4 wave slots busy -- waves without a
    // query block sit out the whole row chain (measured: T=50, two blocks per group, B=512: 0.63 ms fused against
    // 0.42 ms separate; T=200 (7 blocks in 2 groups) 0.434 / 0.453; T=400 (13 in 4) 0.508 / 0.524; T=800 (25 in
    // 7) 0.640 / 0.668).
    const int QBp = (T + 31) / 32, NGp = (QBp + 3) / 4;
    const bool ragged = QBp * 5 < NGp * 4 * 4;  // QB / (4 NG) < 0.8
    w.fused = w.msplit && T > 32 && w.S == 1 && (row_mode == 3 || (row_mode != 2 && !ragged));
    size_t off = 0;
    w.h = off;
    off += w.rows_pad * D;
    w.q = off;
    off += (w.rows_pad + TILE) * D;  // +32 rows of slack: key/value tiles may over-read the last block
    w.k = off;
    off += (w.rows_pad + TILE) * D;
    w.v = off;
    off += (w.rows_pad + TILE) * D;
    w.q2 = w.k2 = w.v2 = off;
    if (w.fused) {
        w.q2 = off;
        off += (w.rows_pad + TILE) * D;
        w.k2 = off;
        off += (w.rows_pad + TILE) * D;
        w.v2 = off;
        off += (w.rows_pad + TILE) * D;
    }
    w.opart = off;
    off += (size_t)w.S * w.rows_pad * D;
    w.ml = off;
    off += (size_t)w.S * w.rows_pad * 2;
    w.xpad = off;
    if (m->FP != m->cfg.feature_size) off += w.rows * (size_t)m->FP;  // zero-padded features
    w.total = off;
    return w;
}

// a3: vad/modeling/transformer.py:403-414 (fp32 semantics), pre-divided by sqrt(D) (:389,401)
void build_pe(std::vector<float>& pe, int T) {
    pe.resize((size_t)T * D);
    const float cexp = (float)(-(log(10000.0) / (double)D));
    const float scale = (float)sqrt((double)D);
    for (int i = 0; i < D / 2; ++i) {
        const float arg = (float)(2 * i) * cexp;
        const float wv = (float)exp((double)arg);
        for (int t = 0; t < T; ++t) {
            const float a = (float)t * wv;
            pe[(size_t)t * D + 2 * i] = (float)sin((double)a) / scale;
            pe[(size_t)t * D + 2 * i + 1] = (float)cos((double)a) / scale;
        }
    }
}

int ensure_pe(savad_model* m, int T, hipStream_t st) {
    if (T <= m->pe_len) return SAVAD_OK;
    int cap = m->pe_len > 0 ? m->pe_len : 10;
    while (cap < T) cap *= 2;
    if (m->d_pe) {
        HIP_TRY(hipStreamSynchronize(st));  // kernels of earlier forwards may still read the old table
        HIP_TRY(hipFree(m->d_pe));
        m->d_pe = nullptr;
        m->pe_len = 0;
    }
    const size_t Dm = m->cfg.d_model;
    HIP_TRY(hipMalloc(&m->d_pe, sizeof(float) * (size_t)cap * Dm));
    if (m->generic) {
        m->h_pe.resize((size_t)cap * Dm);
        gen::build_pe_host(m->h_pe.data(), cap, (int)Dm);
    } else {
        build_pe(m->h_pe, cap);
    }
    HIP_TRY(hipMemcpyAsync(m->d_pe, m->h_pe.data(), sizeof(float) * (size_t)cap * Dm, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // h_pe is pageable and reused
    m->pe_len = cap;
    return SAVAD_OK;
}

int fold(savad_model* m, hipStream_t st, size_t w, size_t b, size_t g, size_t be, size_t wout, size_t bout, int N,
         int K) {
    hipLaunchKernelGGL(fold_ln_kernel, dim3(N), dim3(128), 0, st, m->d_raw + w, m->d_raw + b, m->d_raw + g,
                       m->d_raw + be, m->d_packed + wout, m->d_packed + bout, K);
    HIP_TRY(hipGetLastError());
    return SAVAD_OK;
}

int prepare_weights(savad_model* m, hipStream_t st) {
    if (!m->dirty) return SAVAD_OK;
    for (const Param& p : m->params)
        if (!p.set) return fail(SAVAD_E_STATE, "parameter '%s' was never set", p.key.c_str());
    const int L = m->cfg.num_layers;
    for (int l = 0; l < L; ++l) {
        const auto& r = m->lr[l];
        const auto& p = m->lp[l];
        int rc;
        if ((rc = fold(m, st, r.wq, r.bq, r.ln1w, r.ln1b, p.wqkv, p.bqkv, D, D))) return rc;
        if ((rc = fold(m, st, r.wk, r.bk, r.ln1w, r.ln1b, p.wqkv + (size_t)D * D, p.bqkv + D, D, D))) return rc;
        if ((rc = fold(m, st, r.wv, r.bv, r.ln1w, r.ln1b, p.wqkv + (size_t)2 * D * D, p.bqkv + 2 * D, D, D))) return rc;
        if ((rc = fold(m, st, r.w1, r.b1, r.ln2w, r.ln2b, p.w1, p.b1, DFF, D))) return rc;
        float* frag = m->d_packed + p.frag;
        hipLaunchKernelGGL(pack_frag32_kernel, dim3(192), dim3(256), 0, st, m->d_packed + p.wqkv, 0, 12, frag);
        hipLaunchKernelGGL(pack_frag32_kernel, dim3(64), dim3(256), 0, st, m->d_raw + r.wo, 0, 4, frag + 12 * FRAG_BLOCK);
        hipLaunchKernelGGL(pack_frag32_kernel, dim3(256), dim3(256), 0, st, m->d_packed + p.w1, 0, 16, frag + 16 * FRAG_BLOCK);
        hipLaunchKernelGGL(pack_frag32_kernel, dim3(256), dim3(256), 0, st, m->d_raw + r.w2, 1, 16, frag + 32 * FRAG_BLOCK);
        HIP_TRY(hipGetLastError());
        float* lb = m->d_packed + m->p_bias + (size_t)l * LBIAS;
        const size_t f = sizeof(float);
        HIP_TRY(hipMemcpyAsync(lb, m->d_packed + p.b1, DFF * f, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(lb + DFF, m->d_raw + r.b2, D * f, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(lb + DFF + D, m->d_packed + p.bqkv, 3 * D * f, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(lb + DFF + 4 * D, m->d_raw + r.bo, D * f, hipMemcpyDeviceToDevice, st));
    }
    int rc = fold(m, st, m->r_wc, m->r_bc, m->r_lnf_w, m->r_lnf_b, m->p_wc, m->p_bc, 2, D);
    if (rc) return rc;
    if (m->FP != m->cfg.feature_size) {
        hipLaunchKernelGGL(pad_rows_kernel<float>, dim3(64), dim3(256), 0, st, m->d_raw + m->r_win, (size_t)D,
                           m->cfg.feature_size, m->FP, m->d_packed + m->p_win_pad);
        HIP_TRY(hipGetLastError());
    }
    m->dirty = false;
    return SAVAD_OK;
}

// input weight as the kernels read it: [D][FP], the raw tensor itself when no padding is needed
const float* win_fp32(const savad_model* m) {
    return m->FP == m->cfg.feature_size ? m->d_raw + m->r_win : m->d_packed + m->p_win_pad;
}

int pack_frags(savad_model* m, hipStream_t st, const float* W, int N, int K, size_t off) {
    const size_t total = (size_t)N * K;
    const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(bf::pack_weight_frags_kernel, dim3(grid), dim3(256), 0, st, W, N, K,
                       reinterpret_cast<__bf16*>(m->d_frag + off));
    HIP_TRY(hipGetLastError());
    return SAVAD_OK;
}

int prepare_frags(savad_model* m, hipStream_t st) {
    if (!m->frag_dirty) return SAVAD_OK;
    const int F = m->cfg.feature_size, L = m->cfg.num_layers;
    int rc;
    (void)F;
    if ((rc = pack_frags(m, st, win_fp32(m), D, m->FP, m->f_win))) return rc;
    for (int l = 0; l < L; ++l) {
        if ((rc = pack_frags(m, st, m->d_packed + m->lp[l].wqkv, 3 * D, D, m->lf[l].wqkv))) return rc;
        if ((rc = pack_frags(m, st, m->d_raw + m->lr[l].wo, D, D, m->lf[l].wo))) return rc;
        if ((rc = pack_frags(m, st, m->d_packed + m->lp[l].w1, DFF, D, m->lf[l].w1))) return rc;
        if ((rc = pack_frags(m, st, m->d_raw + m->lr[l].w2, D, DFF, m->lf[l].w2))) return rc;
    }
    m->frag_dirty = false;
    return SAVAD_OK;
}

// block space of the bf16 path (savad_kernels_bf16.h)
struct BlockPlan {
    int nblk, nblk_pad;
    bool fused;  // attention + row chain in one launch per layer (q/k/v^T double-buffered: q2, k2, vt2)
    size_t h, q, k, vt, q2, k2, vt2, ctx, xpad, total;  // byte offsets
};
BlockPlan plan_blocks(const savad_model* m, int B, int T) {
    BlockPlan p;
    if (T > 32)
        p.nblk = B * ((T + 31) / 32);
    else
        p.nblk = (B + (32 / T) - 1) / (32 / T);
    p.nblk_pad = (p.nblk + 7) / 8 * 8;  // whole workgroups for both the 4- and the 8-wave kernels
    size_t off = 0;
    p.h = off;
    off += (size_t)p.nblk_pad * bf::HBLK_FLOATS * sizeof(bf::hres_t);
    const size_t fb = (size_t)(p.nblk_pad + 1) * bf::BLK_BYTES;  // +1 block: a 2-block key stage may over-read
    p.q = off;
    off += fb;
    p.k = off;
    off += fb;
    p.vt = off;
    off += fb;
    p.ctx = off;
    off += fb;
    // row_mode 1 / 2 keep attention and row chain as separate launches (4- / 8-wave workgroups), 3 fuses them.
    // Automatic: fused up to ~4 workgroups per CU.  Measured on MI355X at T=800 (fused vs separate, ms per
    // forward): B=32 0.128 / 0.142, B=64 0.197 / 0.204, B=128 0.355 / 0.370, B=192 0.509 / 0.501, B=256 0.642 /
    // 0.641 -- with more work per CU the wave slots a ragged query-block group leaves idle (3 of 28 at T=800) cost
    // the row chain as much as the context round trip and the extra launches cost the separate form.
    const int QBp = (T + 31) / 32, NGp = (QBp + 3) / 4;
    const long groups = T > 32 ? (long)B * NGp : 0;
    const bool ragged = QBp * 5 < NGp * 4 * 4;  // fewer than 80 % of a group's wave slots hold a query block
    // (below one workgroup per CU the forward is launch / latency bound and fusing wins even with idle slots:
    // B=64, T=50: 0.070 / 0.074 ms; B=32, T=160: 0.072 / 0.080 ms)
    const bool automatic = m->row_mode == 0 || m->row_mode == 4;
    p.fused = T > 32 && (m->row_mode == 3 || (automatic && groups <= 1024 && (!ragged || groups <= 256)));
    p.q2 = p.k2 = p.vt2 = off;
    if (p.fused) {
        p.q2 = off;
        off += fb;
        p.k2 = off;
        off += fb;
        p.vt2 = off;
        off += fb;
    }
    p.xpad = off;
    if (m->FP != m->cfg.feature_size) off += (size_t)B * T * m->FP * sizeof(float);
    p.total = off;
    return p;
}

// block space of the fp32s path (savad_kernels_f32s.h): fp32 residual blocks, Q / K / V^T as triples, double-buffered between
// layers (the fused launch of layer l writes layer l + 1's Q / K / V^T while other workgroups still read layer l's)
struct BlockPlan3 {
    int nblk, nblk_pad;
    size_t h, q, k, vt, q2, k2, vt2, xpad, total;  // byte offsets
};
BlockPlan3 plan_blocks3(const savad_model* m, int B, int T) {
    BlockPlan3 p;
    if (T > 32)
        p.nblk = B * ((T + 31) / 32);
    else
        p.nblk = (B + (32 / T) - 1) / (32 / T);
    p.nblk_pad = (p.nblk + 3) / 4 * 4;
    size_t off = 0;
    p.h = off;
    off += (size_t)p.nblk_pad * fs::HBLK_BYTES;
    const size_t fb = (size_t)p.nblk_pad * fs::BLK3_BYTES;
    size_t* slots[6] = {&p.q, &p.k, &p.vt, &p.q2, &p.k2, &p.vt2};
    for (size_t* s : slots) {
        *s = off;
        off += fb;
    }
    p.xpad = off;
    if (m->FP != m->cfg.feature_size) off += (size_t)B * T * m->FP * sizeof(float);
    p.total = off;
    return p;
}

// T <= 32 in precision 2: ONE launch for the whole forward -- the latency variant (one packed block per workgroup, its four waves
// splitting every GEMM's output features; round 6) while the blocks fill the CUs at most SAVAD_F32S_NS_MAX_ROUNDS times, the
// wave-per-block kernel (four blocks per workgroup share the weight stream through the LDS ring; a block's chain is 7 320 bf16 MFMAs)
// beyond.  (Until the latency variant existed, short clips ran the exact-fp32 kernels of precision 0: SAVAD_F32S_PACKED_MIN_BLOCKS.)
#ifndef SAVAD_F32S_PACKED_MIN_BLOCKS
#define SAVAD_F32S_PACKED_MIN_BLOCKS 0
#endif
bool packed_f32s_applies(const savad_model* m, int B, int T) {
    // row_mode 0 (automatic) and 4: the single launch in the variant the number of blocks suggests (launch_packed_forward_f32s);
    // 5 - 7: the wave-per-block variant, 8: the latency variant (one block per workgroup); 1 - 3 keep the per-layer launches
    // (the cross-check of the tests).  SAVAD_F32S_PACKED_MIN_BLOCKS > 0 (experiment builds): exact-fp32 kernels below that many blocks
    if (T > 32 || m->cfg.num_layers > fs::PACKED_F32S_MAX_LAYERS) return false;
    const long nblk = ((long)B + 32 / T - 1) / (32 / T);
    return m->row_mode >= 4 || (m->row_mode == 0 && nblk >= SAVAD_F32S_PACKED_MIN_BLOCKS);
}
// precision 2 shapes that run the exact-fp32 kernels under the automatic schedule ("fp32s" promises the fp32 result at the best speed
// the library has, not a particular instruction): sequences longer than 32 frames in batches of at most SAVAD_F32S_MIN_BLOCKS_PER_CU
// 32-row blocks per CU.  There a forward's time is the latency of ONE block's chain, and the exact-fp32 kernels split a block's
// GEMMs over the four waves of a workgroup where the fp32s fused launch gives a block to one wave: same-box sweep
// (scripts/ubench/f32s_vs_f32_sweep.py, us, exact fp32 / fp32s): [1,800] 208 / 295, [8,800] 253 / 300, [12,800] 371 / 301,
// [2,3200] 403 / 668, [4,3200] 747 / 673, [64,100] 187 / 203, [24,400] 340 / 246 -- the crossing sits at one block per CU for every T.
// Any non-zero row_mode keeps the fp32s kernels (3: its fused launches at every size -- the tests' way to reach them, and the way
// to results that do not depend on the batch a sequence arrives in: the two kernel families agree to fp32 rounding, not bit for bit).
#ifndef SAVAD_F32S_MIN_BLOCKS_PER_CU
#define SAVAD_F32S_MIN_BLOCKS_PER_CU 1
#endif
bool f32s_uses_exact_fp32(const savad_model* m, int B, int T) {
    if (m->row_mode != 0) return false;
    if (T <= 32) return !packed_f32s_applies(m, B, T);
    return (long)B * ((T + 31) / 32) <= (long)SAVAD_F32S_MIN_BLOCKS_PER_CU * m->n_cu;
}
int pack_frags3(savad_model* m, hipStream_t st, const float* W, int N, int K, size_t off) {
    const size_t total = (size_t)N * K;
    const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(fs::pack_weight_frags3_kernel, dim3(grid), dim3(256), 0, st, W, N, K, reinterpret_cast<__bf16*>(m->d_frag3 + off));
    HIP_TRY(hipGetLastError());
    return SAVAD_OK;
}

int prepare_frags3(savad_model* m, hipStream_t st) {
    if (!m->frag3_dirty) return SAVAD_OK;
    const int L = m->cfg.num_layers;
    int rc;
    if ((rc = pack_frags3(m, st, win_fp32(m), D, m->FP, m->f3_win))) return rc;
    for (int l = 0; l < L; ++l) {
        if ((rc = pack_frags3(m, st, m->d_packed + m->lp[l].wqkv, 3 * D, D, m->lf3[l].wqkv))) return rc;
        if ((rc = pack_frags3(m, st, m->d_raw + m->lr[l].wo, D, D, m->lf3[l].wo))) return rc;
        if ((rc = pack_frags3(m, st, m->d_packed + m->lp[l].w1, DFF, D, m->lf3[l].w1))) return rc;
        if ((rc = pack_frags3(m, st, m->d_raw + m->lr[l].w2, D, DFF, m->lf3[l].w2))) return rc;
    }
    m->frag3_dirty = false;
    return SAVAD_OK;
}

template <typename KernelT>
int allow_lds(KernelT kernel, int bytes) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return SAVAD_OK;
}

struct Prof {
    savad_model* m;
    hipStream_t st;
    hipEvent_t* ev;
    int n;
    Prof(savad_model* mm, hipStream_t s) : m(mm), st(s), ev(nullptr), n(0) {
        if (m->prof_capacity > 0 && m->prof_skip > 0) {
            --m->prof_skip;  // settle-in forwards after savad_set_profiling: launched as usual, not recorded
        } else if (m->prof_capacity > 0 && m->prof_used < m->prof_capacity) {
            ev = m->events.data() + (size_t)m->prof_used * MAX_EVENTS;
            m->knames.clear();
            hipEventRecord(ev[0], st);
        }
    }
    void mark(const char* name) {
        if (!ev || n + 1 >= MAX_EVENTS) return;
        ++n;
        hipEventRecord(ev[n], st);
        m->knames.push_back(name);
    }
    void done() {
        if (!ev) return;
        m->prof_nk = n;
        m->prof_used++;
    }
};


// ---- d_model != 128: same parameter inventory with the runtime width, no folded / fragment copies
int create_generic(const savad_config* cfg, savad_handle* out) {
    if (cfg->feature_size <= 0 || cfg->feature_size > 4096) return fail(SAVAD_E_INVALID, "feature_size=%d", cfg->feature_size);
    if (cfg->num_layers < 1 || cfg->num_layers > 64) return fail(SAVAD_E_INVALID, "num_layers=%d", cfg->num_layers);
    savad_model* m = new savad_model();
    m->cfg = *cfg;
    m->generic = true;
    const size_t Dm = cfg->d_model, Fm = cfg->feature_size, Hm = 4 * Dm;  // d_ff = 4 d_model: vad/models/self_attention.py:10
    const int L = cfg->num_layers;
    m->r_win = add_param(m, "input_layer.0.weight", Dm * Fm);
    m->r_bin = add_param(m, "input_layer.0.bias", Dm);
    m->lr.resize(L);
    for (int l = 0; l < L; ++l) {
        const std::string p = "encoder.layers." + std::to_string(l) + ".";
        auto& r = m->lr[l];
        r.wq = add_param(m, p + "self_attention.query_projection.weight", Dm * Dm);
        r.bq = add_param(m, p + "self_attention.query_projection.bias", Dm);
        r.wk = add_param(m, p + "self_attention.key_projection.weight", Dm * Dm);
        r.bk = add_param(m, p + "self_attention.key_projection.bias", Dm);
        r.wv = add_param(m, p + "self_attention.value_projection.weight", Dm * Dm);
        r.bv = add_param(m, p + "self_attention.value_projection.bias", Dm);
        r.wo = add_param(m, p + "self_attention.final_projection.weight", Dm * Dm);
        r.bo = add_param(m, p + "self_attention.final_projection.bias", Dm);
        r.ln1w = add_param(m, p + "self_attention_sublayer.layer_norm.weight", Dm);
        r.ln1b = add_param(m, p + "self_attention_sublayer.layer_norm.bias", Dm);
        r.w1 = add_param(m, p + "feed_forward.feed_forward.0.weight", Hm * Dm);
        r.b1 = add_param(m, p + "feed_forward.feed_forward.0.bias", Hm);
        r.w2 = add_param(m, p + "feed_forward.feed_forward.3.weight", Dm * Hm);
        r.b2 = add_param(m, p + "feed_forward.feed_forward.3.bias", Dm);
        r.ln2w = add_param(m, p + "feed_forward_sublayer.layer_norm.weight", Dm);
        r.ln2b = add_param(m, p + "feed_forward_sublayer.layer_norm.bias", Dm);
    }
    m->r_lnf_w = add_param(m, "encoder.layer_norm.weight", Dm);
    m->r_lnf_b = add_param(m, "encoder.layer_norm.bias", Dm);
    m->r_wc = add_param(m, "classifier.weight", 2 * Dm);
    m->r_bc = add_param(m, "classifier.bias", 2);
    m->FP = cfg->feature_size;
    hipError_t e = hipMalloc(&m->d_raw, sizeof(float) * m->raw_floats);
    if (e != hipSuccess) {
        delete m;
        return fail(SAVAD_E_HIP, "hipMalloc(weights): %s", hipGetErrorString(e));
    }
    *out = m;
    return SAVAD_OK;
}

void launch_gemm(hipStream_t st, const gen::GemmArgs& g, int batch) {
    hipLaunchKernelGGL(gen::gemm_kernel, dim3((g.N + gen::GT - 1) / gen::GT, (g.M + gen::GT - 1) / gen::GT, batch), dim3(256), 0, st, g);
}

// nn.Linear on `rows` rows: y = act(x W^T + b [+ pe]) [+ res]
void launch_linear(hipStream_t st, const float* x, long rows, int K, const float* W, const float* b, int N, float* y, const float* res,
                   bool relu, const float* pe, int T) {
    // rows per launch: a multiple of T (the positional-encoding row of output row m is m % T) that keeps grid.y inside its limit
    const long step = T >= (1L << 21) ? T : (1L << 21) / T * T;
    for (long r0 = 0; r0 < rows; r0 += step) {
        gen::GemmArgs g{};
        g.M = (int)(rows - r0 < step ? rows - r0 : step);
        g.N = N;
        g.K = K;
        g.A = x + r0 * K;
        g.lda = K;
        g.Bm = W;
        g.ldk = 1;
        g.ldn = K;
        g.C = y + r0 * N;
        g.ldc = N;
        g.alpha = 1.0f;
        g.bias = b;
        g.add = pe;
        g.add_rows = pe ? T : 1;
        g.res = res ? res + r0 * N : nullptr;
        g.relu = relu ? 1 : 0;
        launch_gemm(st, g, 1);
    }
}

// SelfAttentiveVAD.forward for any d_model, the reference's operation sequence (vad/models/self_attention.py:23-28) kernel by kernel
int forward_generic(savad_model* m, const float* x, int B, int T, float* out, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const int Dm = m->cfg.d_model, F = m->cfg.feature_size, L = m->cfg.num_layers;
    const gen::Plan p = gen::plan(B, T, Dm, m->splits);
    if (workspace_bytes < p.total * sizeof(float))
        return fail(SAVAD_E_INVALID, "workspace too small: %zu < %zu bytes", workspace_bytes, p.total * sizeof(float));
    for (const Param& q : m->params)
        if (!q.set) return fail(SAVAD_E_NOKEY, "missing key '%s' in state_dict", q.key.c_str());
    int rc;
    if ((rc = ensure_pe(m, T, st))) return rc;
    float* W = (float*)workspace;
    float *h = W + p.h, *n = W + p.n, *q = W + p.q, *k = W + p.k, *v = W + p.v, *ctx = W + p.ctx, *ff = W + p.ff, *sc = W + p.scores;
    const float* R = m->d_raw;
    const long rows = (long)B * T;
    const int ln_grid = (int)((rows + 3) / 4);
    Prof prof(m, st);
    // input Linear + positional encoding / sqrt(d_model) (self_attention.py:13-15, transformer.py:401); dropout = identity
    launch_linear(st, x, rows, F, R + m->r_win, R + m->r_bin, Dm, h, nullptr, false, m->d_pe, T);
    prof.mark("input_generic");
    const float alpha = (float)(1.0 / sqrt((double)Dm));  // / sqrt(d_head), one head (transformer.py:362)
    for (int l = 0; l < L; ++l) {
        const auto& r = m->lr[l];
        hipLaunchKernelGGL(gen::layernorm_kernel, dim3(ln_grid), dim3(256), 0, st, h, R + r.ln1w, R + r.ln1b, n, rows, Dm);
        launch_linear(st, n, rows, Dm, R + r.wq, R + r.bq, Dm, q, nullptr, false, nullptr, T);
        launch_linear(st, n, rows, Dm, R + r.wk, R + r.bk, Dm, k, nullptr, false, nullptr, T);
        launch_linear(st, n, rows, Dm, R + r.wv, R + r.bv, Dm, v, nullptr, false, nullptr, T);
        prof.mark("qkv_generic");
        for (int b0 = 0; b0 < B; b0 += p.cb) {
            const int nb = B - b0 < p.cb ? B - b0 : p.cb;
            for (int t0 = 0; t0 < T; t0 += p.tq) {
                const int nq = T - t0 < p.tq ? T - t0 : p.tq;
                gen::GemmArgs g{};
                g.A = q + ((size_t)b0 * T + t0) * Dm;  // scores = q k^T / sqrt(d) (transformer.py:351-363)
                g.lda = Dm;
                g.sA = (long)T * Dm;
                g.Bm = k + (size_t)b0 * T * Dm;
                g.ldk = 1;
                g.ldn = Dm;
                g.sB = (long)T * Dm;
                g.C = sc;
                g.ldc = T;
                g.sC = (long)nq * T;
                g.M = nq;
                g.N = T;
                g.K = Dm;
                g.alpha = alpha;
                g.add_rows = 1;
                launch_gemm(st, g, nb);
                const long srows = (long)nb * nq;
                hipLaunchKernelGGL(gen::softmax_kernel, dim3((unsigned)((srows + 3) / 4)), dim3(256), 0, st, sc, srows, T);
                gen::GemmArgs c{};
                c.A = sc;  // context = A V (transformer.py:338-346)
                c.lda = T;
                c.sA = (long)nq * T;
                c.Bm = v + (size_t)b0 * T * Dm;
                c.ldk = Dm;
                c.ldn = 1;
                c.sB = (long)T * Dm;
                c.C = ctx + ((size_t)b0 * T + t0) * Dm;
                c.ldc = Dm;
                c.sC = (long)T * Dm;
                c.M = nq;
                c.N = Dm;
                c.K = T;
                c.alpha = 1.0f;
                c.add_rows = 1;
                launch_gemm(st, c, nb);
            }
        }
        prof.mark("attention_generic");
        // final_projection + residual onto the un-normalised x (transformer.py:347,237); FFN sublayer (:366-382)
        launch_linear(st, ctx, rows, Dm, R + r.wo, R + r.bo, Dm, h, h, false, nullptr, T);
        hipLaunchKernelGGL(gen::layernorm_kernel, dim3(ln_grid), dim3(256), 0, st, h, R + r.ln2w, R + r.ln2b, n, rows, Dm);
        launch_linear(st, n, rows, Dm, R + r.w1, R + r.b1, 4 * Dm, ff, nullptr, true, nullptr, T);
        launch_linear(st, ff, rows, 4 * Dm, R + r.w2, R + r.b2, Dm, h, h, false, nullptr, T);
        prof.mark("row_generic");
    }
    hipLaunchKernelGGL(gen::layernorm_kernel, dim3(ln_grid), dim3(256), 0, st, h, R + m->r_lnf_w, R + m->r_lnf_b, n, rows, Dm);
    hipLaunchKernelGGL(gen::classifier_kernel, dim3(ln_grid), dim3(256), 0, st, n, R + m->r_wc, R + m->r_bc, out, rows, Dm);
    prof.mark("classifier_generic");
    prof.done();
    HIP_TRY(hipGetLastError());
    return SAVAD_OK;
}

}  // namespace

SAVAD_EXPORT const char* savad_last_error(void) { return g_err; }
SAVAD_EXPORT const char* savad_version(void) { return "savad 0.1 (gfx950, fp32 MFMA)"; }

SAVAD_EXPORT int savad_create(const savad_config* cfg, savad_handle* out) {
    if (!cfg || !out) return fail(SAVAD_E_INVALID, "null argument");
    if (cfg->d_model < 2 || cfg->d_model > 4096 || cfg->d_model % 2)  // the reference's positional encoding pairs sin / cos columns
        return fail(SAVAD_E_INVALID, "d_model=%d (an even value in [2, 4096])", cfg->d_model);
    if (cfg->d_model != D) return create_generic(cfg, out);
    if (cfg->feature_size <= 0 || cfg->feature_size > 4096)
        return fail(SAVAD_E_INVALID, "feature_size=%d", cfg->feature_size);
    if (cfg->num_layers < 1 || cfg->num_layers > 64) return fail(SAVAD_E_INVALID, "num_layers=%d", cfg->num_layers);
    savad_model* m = new savad_model();
    m->cfg = *cfg;
    {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) m->n_cu = n;
    }
    const int F = cfg->feature_size, L = cfg->num_layers;
    // state_dict inventory: SURVEY.md section 8a / vad/models/self_attention.py:7-21
    m->r_win = add_param(m, "input_layer.0.weight", (size_t)D * F);
    m->r_bin = add_param(m, "input_layer.0.bias", D);
    m->lr.resize(L);
    m->lp.resize(L);
    for (int l = 0; l < L; ++l) {
        const std::string p = "encoder.layers." + std::to_string(l) + ".";
        auto& r = m->lr[l];
        r.wq = add_param(m, p + "self_attention.query_projection.weight", (size_t)D * D);
        r.bq = add_param(m, p + "self_attention.query_projection.bias", D);
        r.wk = add_param(m, p + "self_attention.key_projection.weight", (size_t)D * D);
        r.bk = add_param(m, p + "self_attention.key_projection.bias", D);
        r.wv = add_param(m, p + "self_attention.value_projection.weight", (size_t)D * D);
        r.bv = add_param(m, p + "self_attention.value_projection.bias", D);
        r.wo = add_param(m, p + "self_attention.final_projection.weight", (size_t)D * D);
        r.bo = add_param(m, p + "self_attention.final_projection.bias", D);
        r.ln1w = add_param(m, p + "self_attention_sublayer.layer_norm.weight", D);
        r.ln1b = add_param(m, p + "self_attention_sublayer.layer_norm.bias", D);
        r.w1 = add_param(m, p + "feed_forward.feed_forward.0.weight", (size_t)DFF * D);
        r.b1 = add_param(m, p + "feed_forward.feed_forward.0.bias", DFF);
        r.w2 = add_param(m, p + "feed_forward.feed_forward.3.weight", (size_t)D * DFF);
        r.b2 = add_param(m, p + "feed_forward.feed_forward.3.bias", D);
        r.ln2w = add_param(m, p + "feed_forward_sublayer.layer_norm.weight", D);
        r.ln2b = add_param(m, p + "feed_forward_sublayer.layer_norm.bias", D);
        auto& q = m->lp[l];
        q.wqkv = m->packed_floats;
        m->packed_floats += (size_t)3 * D * D;
        q.bqkv = m->packed_floats;
        m->packed_floats += 3 * D;
        q.w1 = m->packed_floats;
        m->packed_floats += (size_t)DFF * D;
        q.b1 = m->packed_floats;
        m->packed_floats += DFF;
        q.frag = m->packed_floats;
        m->packed_floats += FRAG_LAYER;
    }
    m->r_lnf_w = add_param(m, "encoder.layer_norm.weight", D);
    m->r_lnf_b = add_param(m, "encoder.layer_norm.bias", D);
    m->r_wc = add_param(m, "classifier.weight", 2 * D);
    m->r_bc = add_param(m, "classifier.bias", 2);
    m->p_bias = m->packed_floats;
    m->packed_floats += (size_t)L * LBIAS;
    m->p_wc = m->packed_floats;
    m->packed_floats += 2 * D;
    m->p_bc = m->packed_floats;
    m->packed_floats += 4;
    m->FP = (F + 15) / 16 * 16;
    m->p_win_pad = m->packed_floats;
    m->packed_floats += (size_t)D * m->FP;
    m->lf.resize(L);
    m->f_win = 0;
    m->frag_bytes = (size_t)D * ((F + 15) / 16 * 16) * 2;
    for (int l = 0; l < L; ++l) {
        auto& fl = m->lf[l];
        fl.wqkv = m->frag_bytes;
        m->frag_bytes += (size_t)3 * D * D * 2;
        fl.wo = m->frag_bytes;
        m->frag_bytes += (size_t)D * D * 2;
        fl.w1 = m->frag_bytes;
        m->frag_bytes += (size_t)DFF * D * 2;
        fl.w2 = m->frag_bytes;
        m->frag_bytes += (size_t)D * DFF * 2;
    }
    m->lf3.resize(L);
    m->f3_win = 0;
    m->frag3_bytes = (size_t)D * ((F + 15) / 16 * 16) * 6;
    for (int l = 0; l < L; ++l) {
        auto& fl = m->lf3[l];
        fl.wqkv = m->frag3_bytes;
        m->frag3_bytes += (size_t)3 * D * D * 6;
        fl.wo = m->frag3_bytes;
        m->frag3_bytes += (size_t)D * D * 6;
        fl.w1 = m->frag3_bytes;
        m->frag3_bytes += (size_t)DFF * D * 6;
        fl.w2 = m->frag3_bytes;
        m->frag3_bytes += (size_t)D * DFF * 6;
    }
    hipError_t e = hipMalloc(&m->d_raw, sizeof(float) * m->raw_floats);
    if (e == hipSuccess) e = hipMalloc(&m->d_frag, m->frag_bytes);
    if (e == hipSuccess) e = hipMalloc(&m->d_frag3, m->frag3_bytes);
    if (e == hipSuccess) e = hipMalloc(&m->d_sat, sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(m->d_sat, 0, sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc(&m->d_packed, sizeof(float) * m->packed_floats);
    if (e != hipSuccess) {
        if (m->d_raw) hipFree(m->d_raw);
        delete m;
        return fail(SAVAD_E_HIP, "hipMalloc(weights): %s", hipGetErrorString(e));
    }
    *out = m;
    return SAVAD_OK;
}

SAVAD_EXPORT void savad_destroy(savad_handle m) {
    if (!m) return;
    for (hipEvent_t e : m->events) hipEventDestroy(e);
    if (m->d_raw) hipFree(m->d_raw);
    if (m->d_packed) hipFree(m->d_packed);
    if (m->d_frag) hipFree(m->d_frag);
    if (m->d_frag3) hipFree(m->d_frag3);
    if (m->d_sat) hipFree(m->d_sat);
    if (m->d_pe) hipFree(m->d_pe);
    delete m;
}

SAVAD_EXPORT int savad_num_params(savad_handle m) { return m ? (int)m->params.size() : 0; }
SAVAD_EXPORT const char* savad_param_key(savad_handle m, int i) {
    return (m && i >= 0 && i < (int)m->params.size()) ? m->params[i].key.c_str() : nullptr;
}
SAVAD_EXPORT size_t savad_param_numel(savad_handle m, int i) {
    return (m && i >= 0 && i < (int)m->params.size()) ? m->params[i].numel : 0;
}

SAVAD_EXPORT int savad_set_param(savad_handle m, const char* key, const float* data, size_t numel, void* stream) {
    if (!m || !key || !data) return fail(SAVAD_E_INVALID, "null argument");
    for (Param& p : m->params) {
        if (p.key != key) continue;
        if (p.numel != numel)
            return fail(SAVAD_E_INVALID, "size mismatch for '%s': got %zu elements, expected %zu", key, numel, p.numel);
        HIP_TRY(hipMemcpyAsync(m->d_raw + p.off, data, sizeof(float) * numel, hipMemcpyDefault, (hipStream_t)stream));
        p.set = true;
        m->dirty = true;
        m->frag_dirty = true;
        m->frag3_dirty = true;
        return SAVAD_OK;
    }
    return fail(SAVAD_E_NOKEY, "unexpected key '%s' in state_dict", key);
}

SAVAD_EXPORT int savad_set_attention_splits(savad_handle m, int splits) {
    if (!m || splits < 0 || splits > 64) return fail(SAVAD_E_INVALID, "splits=%d", splits);
    m->splits = splits;
    return SAVAD_OK;
}

SAVAD_EXPORT int savad_set_row_mode(savad_handle m, int mode) {
    if (!m || mode < 0 || mode > 8) return fail(SAVAD_E_INVALID, "row mode %d", mode);
    m->row_mode = mode;
    return SAVAD_OK;
}

SAVAD_EXPORT int savad_set_batch_invariant(savad_handle m, int on) {
    if (!m || on < 0 || on > 1) return fail(SAVAD_E_INVALID, "batch_invariant %d (0 or 1)", on);
    m->batch_invariant = on != 0;
    return SAVAD_OK;
}

SAVAD_EXPORT int savad_workspace_bytes(savad_handle m, int B, int T, size_t* bytes) {
    if (!m || !bytes || B < 0 || T < 0) return fail(SAVAD_E_INVALID, "bad argument");
    if ((double)B * T * D >= 2.0e9) return fail(SAVAD_E_UNSUPPORTED, "B*T=%ld rows exceed the 32-bit tile index range", (long)B * T);
    if (B == 0 || T == 0)
        *bytes = 0;
    else if (m->generic)
        *bytes = gen::plan(B, T, m->cfg.d_model, m->splits).total * sizeof(float);
    else if (m->precision == 1)
        *bytes = plan_blocks(m, B, T).total;
    else if (m->precision == 2)
        *bytes = f32s_uses_exact_fp32(m, B, T) ? plan(m, B, T).total * sizeof(float) : plan_blocks3(m, B, T).total;
    else
        *bytes = plan(m, B, T).total * sizeof(float);
    return SAVAD_OK;
}

// Sizes everything savad_forward may otherwise have to (re)allocate for sequences of up to T_max frames -- today the
// positional-encoding table -- so that later forwards with T <= T_max neither allocate nor synchronise.
// Once every parameter has been set it also folds / packs the weights for the selected precision (and raises the bf16
// kernels' LDS limits), so that even the FIRST forward after it launches nothing but its own kernels and can be captured
// into a HIP graph.  With parameters still missing only the table is sized (the forward reports the missing key).
namespace {
int prepare_bf16_launch(savad_model* m);
int prepare_f32s_launch(savad_model* m);
}
SAVAD_EXPORT int savad_reserve(savad_handle m, int T_max, void* stream) {
    if (!m || T_max < 0) return fail(SAVAD_E_INVALID, "bad argument");
    if ((double)T_max * D >= 2.0e9) return fail(SAVAD_E_UNSUPPORTED, "T_max=%d too large", T_max);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = ensure_pe(m, T_max, st))) return rc;
    if (m->generic) return SAVAD_OK;  // nothing to fold or pack: the generic kernels read the raw parameters
    bool all_set = true;
    for (const Param& p : m->params) all_set = all_set && p.set;
    if (!all_set) return SAVAD_OK;
    if ((rc = prepare_weights(m, st))) return rc;
    if (m->precision == 1) {
        if ((rc = prepare_frags(m, st))) return rc;
        if ((rc = prepare_bf16_launch(m))) return rc;
    }
    if (m->precision == 2) {
        if ((rc = prepare_frags3(m, st))) return rc;
        if ((rc = prepare_f32s_launch(m))) return rc;
    }
    return SAVAD_OK;
}

// bf16 precision stores the residual stream between kernels as fp16 (+-65504); every element that had to be clamped is
// counted.  Reads the count accumulated since the last call and clears it; synchronises `stream`.
SAVAD_EXPORT int savad_residual_saturations(savad_handle m, unsigned long long* count, void* stream) {
    if (!m || !count) return fail(SAVAD_E_INVALID, "null argument");
    if (!m->d_sat) {  // fp32-only handle (d_model != 128): no fp16-stored residual stream, nothing can saturate
        *count = 0;
        return SAVAD_OK;
    }
    unsigned c = 0;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(&c, m->d_sat, sizeof(c), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemsetAsync(m->d_sat, 0, sizeof(unsigned), st));
    HIP_TRY(hipStreamSynchronize(st));
    *count = c;
    return SAVAD_OK;
}

SAVAD_EXPORT int savad_set_precision(savad_handle m, int precision) {
    if (!m || precision < 0 || precision > 2) return fail(SAVAD_E_INVALID, "precision %d (0 = fp32, 1 = bf16, 2 = fp32s)", precision);
    if (m->generic && precision != 0)
        return fail(SAVAD_E_UNSUPPORTED, "bf16 / split-bf16 operands are implemented for d_model=128 only (this handle: d_model=%d, fp32)", m->cfg.d_model);
    m->precision = precision;
    return SAVAD_OK;
}

#ifndef SAVAD_INPUT_PERSISTENT
#define SAVAD_INPUT_PERSISTENT 1   // 0: experiment builds that keep the ring form of the bf16 input stage everywhere (scripts/ubench/input_p_ab.py)
#endif
namespace {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) for every bf16 kernel, once per handle
int prepare_bf16_launch(savad_model* m) {
    int rc;
    if (m->lds_attrs_set) return SAVAD_OK;
    constexpr int r4 = bf::Ring<4>::NRING * bf::RING_BYTES, r8 = bf::Ring<8>::NRING * bf::RING_BYTES;
    if ((rc = allow_lds(bf::input_qkv_kernel_bf16<float, 4>, r4 + 3 * D * 4))) return rc;
    if ((rc = allow_lds(bf::input_qkv_kernel_bf16<__bf16, 4>, r4 + 3 * D * 4))) return rc;
    if ((rc = allow_lds(bf::input_qkv_kernel_bf16_p<float, 8, 5>, bf::input_p_lds_bytes(5)))) return rc;
    if ((rc = allow_lds(bf::input_qkv_kernel_bf16_p<__bf16, 8, 5>, bf::input_p_lds_bytes(5)))) return rc;
    if ((rc = allow_lds(bf::input_qkv_kernel_bf16_p<float, 8, 0>, bf::input_p_lds_bytes(15)))) return rc;
    if ((rc = allow_lds(bf::input_qkv_kernel_bf16_p<__bf16, 8, 0>, bf::input_p_lds_bytes(15)))) return rc;
    if ((rc = allow_lds(bf::attention_kernel_bf16<4>, r4))) return rc;
    if ((rc = allow_lds(bf::row_kernel_bf16<false, 4>, r4 + 9 * D * 4))) return rc;
    if ((rc = allow_lds(bf::row_kernel_bf16<true, 4>, r4 + 9 * D * 4))) return rc;
    if ((rc = allow_lds(bf::attention_row_kernel_bf16<false, 4>, r4 + 9 * D * 4))) return rc;
    if ((rc = allow_lds(bf::attention_row_kernel_bf16<true, 4>, r4 + 9 * D * 4))) return rc;
    if ((rc = allow_lds(bf::input_qkv_kernel_bf16<float, 8>, r8 + 3 * D * 4))) return rc;
    if ((rc = allow_lds(bf::input_qkv_kernel_bf16<__bf16, 8>, r8 + 3 * D * 4))) return rc;
    if ((rc = allow_lds(bf::attention_kernel_bf16<8>, r8))) return rc;
    if ((rc = allow_lds(bf::row_kernel_bf16<false, 8>, r8 + 9 * D * 4))) return rc;
    if ((rc = allow_lds(bf::row_kernel_bf16<true, 8>, r8 + 9 * D * 4))) return rc;
    if ((rc = allow_lds(bf::attention_pw_kernel_bf16, bf::PW_LDS_BYTES))) return rc;
    if ((rc = allow_lds(bf::attention_pw_kernel_bf16_nosplit, bf::PW_LDS_BYTES))) return rc;
    if ((rc = allow_lds(bf::packed_forward_kernel_bf16<4, 2, 0>, r4 + (bf::PACKED_BF16_MAX_LAYERS * LBIAS + 2 * D + 4) * 4))) return rc;
    if ((rc = allow_lds(bf::packed_forward_kernel_bf16<4, 4, 4>, r8 + (bf::PACKED_BF16_MAX_LAYERS * LBIAS + 2 * D + 4) * 4))) return rc;
    if ((rc = allow_lds(bf::packed_forward_kernel_bf16<8, 4, 0>, r8 + (bf::PACKED_BF16_MAX_LAYERS * LBIAS + 2 * D + 4) * 4))) return rc;
    if ((rc = allow_lds(bf::packed_forward_kernel_bf16_ns, bf::ns_lds_bytes(bf::PACKED_BF16_MAX_LAYERS)))) return rc;
    m->lds_attrs_set = true;
    return SAVAD_OK;
}

// T <= 32 with bf16 operands: the whole forward in one launch (savad_packed_bf16.h); a wave per packed block, NW blocks per
// workgroup.  Weights, fragments, the PE table and the kernels' LDS attributes must be ready.
#ifndef SAVAD_NS_MAX_ROUNDS
#define SAVAD_NS_MAX_ROUNDS 1   // blocks per CU up to which one block per workgroup beats four (scripts/ubench/packed_bf16_bench.py)
#endif
bool packed_bf16_applies(const savad_model* m, int T) {
    // row_mode 0 (automatic) and 4: picked by the number of blocks; 5 - 7: a fixed variant (launch_packed_forward_bf16; tuning
    // knobs at T <= 32, where the persistent attention kernel that 5 selects for long sequences does not exist); 1 - 3 keep
    // the per-layer launches (the cross-check of the tests)
    return T <= 32 && m->cfg.num_layers <= bf::PACKED_BF16_MAX_LAYERS && (m->row_mode == 0 || m->row_mode >= 4);
}
void launch_packed_forward_bf16(savad_model* m, hipStream_t st, const float* x, int B, int T, int F, float* out, const WindowOffsets& wo,
                                int win_base) {
    const int L = m->cfg.num_layers;
    const char* Fr = m->d_frag;
    const int G = 32 / T, nblk = (B + G - 1) / G;
    bf::PackedBf16Model pm;
    for (int l = 0; l < bf::PACKED_BF16_MAX_LAYERS; ++l) {
        const auto& f = m->lf[l < L ? l : 0];
        pm.layer[l] = bf::PackedBf16Layer{Fr + f.wqkv, Fr + f.wo, Fr + f.w1, Fr + f.w2};
    }
    pm.win = Fr + m->f_win;
    pm.bin = m->d_raw + m->r_bin;
    pm.pe = m->d_pe;
    pm.bias = m->d_packed + m->p_bias;
    pm.wc = m->d_packed + m->p_wc;
    pm.bc = m->d_packed + m->p_bc;
    pm.L = L;
    const float c = (float)(1.4426950408889634 / sqrt((double)D));
    const size_t bias_bytes = ((size_t)L * LBIAS + 2 * D + 4) * 4;   // every layer's biases + the classifier
    // variant: row_mode 5 = 8-wave workgroups, 6 = 4 waves + 4 that move the weight stream through a 4-slot ring, 7 = 4 waves +
    // 2 slots; automatic: 6 while the 4-block workgroups fill at most half of the CUs ([1000,7,80], 63 workgroups: 0.044 against 0.049 ms;
    // [4000,7,80], 250 workgroups: 0.059 against 0.053; scripts/ubench/packed_bf16_bench.py)
    // 8 = the latency variant: ONE block per workgroup, its four waves split the output features (savad_packed_bf16.h)
    const int variant = m->row_mode >= 5 ? m->row_mode : (nblk <= SAVAD_NS_MAX_ROUNDS * m->n_cu ? 8 : ((nblk + 3) / 4 <= m->n_cu / 2 ? 6 : 7));
    const size_t ring2 = (size_t)2 * bf::RING_BYTES, ring4 = (size_t)4 * bf::RING_BYTES;
    if (variant == 8)
        hipLaunchKernelGGL(bf::packed_forward_kernel_bf16_ns, dim3(nblk), dim3(256), bf::ns_lds_bytes(L), st, x, B, T, F, nblk, pm, c, out, wo, win_base,
                           m->d_sat);
    else if (variant == 5)
        hipLaunchKernelGGL((bf::packed_forward_kernel_bf16<8, 4, 0>), dim3((nblk + 7) / 8), dim3(512), ring4 + bias_bytes, st, x, B, T, F, nblk, pm, c, out,
                           wo, win_base, m->d_sat);
    else if (variant == 6)
        hipLaunchKernelGGL((bf::packed_forward_kernel_bf16<4, 4, 4>), dim3((nblk + 3) / 4), dim3(512), ring4 + bias_bytes, st, x, B, T, F, nblk, pm, c, out,
                           wo, win_base, m->d_sat);
    else
        hipLaunchKernelGGL((bf::packed_forward_kernel_bf16<4, 2, 0>), dim3((nblk + 3) / 4), dim3(256), ring2 + bias_bytes, st, x, B, T, F, nblk, pm, c, out,
                           wo, win_base, m->d_sat);
}

// bf16-operand forward: input_qkv -> [attention -> row] x L on fragment-major buffers
int forward_bf16(savad_model* m, const void* x, int x_is_bf16, int B, int T, float* out, void* workspace,
                 size_t workspace_bytes, hipStream_t st, long xbs_in = 0 /* elements between consecutive sequences of x; 0: T * F (savad_forward_strided) */) {
    const BlockPlan bp = plan_blocks(m, B, T);
    if (workspace_bytes < bp.total) return fail(SAVAD_E_INVALID, "workspace too small: %zu < %zu bytes", workspace_bytes, bp.total);
    int rc;
    if ((rc = prepare_weights(m, st))) return rc;
    if ((rc = prepare_frags(m, st))) return rc;
    if ((rc = ensure_pe(m, T, st))) return rc;
    char* W = (char*)workspace;
    bf::hres_t* hb = (bf::hres_t*)(W + bp.h);
    char *qf = W + bp.q, *kf = W + bp.k, *vtf = W + bp.vt, *ctxf = W + bp.ctx;
    const int L = m->cfg.num_layers;
    int F = m->cfg.feature_size;
    if (m->FP != F) {  // zero-pad the features to the kernels' K granularity (fp32 copy)
        float* xp = (float*)(W + bp.xpad);
        const size_t rows = (size_t)B * T;
        const int grid = (int)((rows * m->FP + 255) / 256 < 4096 ? (rows * m->FP + 255) / 256 : 4096);
        if (x_is_bf16)
            hipLaunchKernelGGL(pad_rows_kernel<__bf16>, dim3(grid), dim3(256), 0, st, (const __bf16*)x, rows, F, m->FP, xp);
        else
            hipLaunchKernelGGL(pad_rows_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, rows, F, m->FP, xp);
        x = xp;
        x_is_bf16 = 0;
        F = m->FP;
    }
    if (xbs_in > 0 && (m->FP != m->cfg.feature_size || packed_bf16_applies(m, T))) return fail(SAVAD_E_UNSUPPORTED, "strided input: no kernel takes the stride for this shape");
    const long xbs = xbs_in > 0 ? xbs_in : (long)T * F;
    const float c = (float)(1.4426950408889634 / sqrt((double)D));
    const float* R = m->d_raw;
    const float* P = m->d_packed;
    const char* Fr = m->d_frag;
    // 4-wave workgroups (two per CU, 2-slot ring) by default.  row_mode 2 selects the 8-wave variant with a
    // 4-deep ring (half the DMA stream per data row, one workgroup per CU): measured SLOWER on MI355X at
    // every size tried (B=256, T=800: 0.86 vs 0.75 ms), kept as a tuning knob and covered by the tests.
    // row_mode 0 / 3 fuse attention and row chain per layer when T > 32 (plan_blocks); 1 / 2 keep them apart.
    const bool wide = m->row_mode == 2;
    if ((rc = prepare_bf16_launch(m))) return rc;
    Prof prof(m, st);
    if (!x_is_bf16 && packed_bf16_applies(m, T)) {
        WindowOffsets none;
        none.w = 0;
        launch_packed_forward_bf16(m, st, (const float*)x, B, T, F, out, none, 0);
        prof.mark("packed_forward_bf16");
        prof.done();
        HIP_TRY(hipGetLastError());
        return SAVAD_OK;
    }
    const bool automatic_bf16 = m->row_mode == 0 || m->row_mode == 4;
    // The persistent attention kernel (one 4 x 64-row workgroup per CU walking (sequence, 8 query blocks) items) against the
    // first-generation one: a cost model of both, from the sweep scripts/ubench/pw_sweep.py (round 4, us per launch, first-generation /
    // persistent): [96,800] 43.4 / 50.1, [128,800] 52.6 / 52.0, [160,800] 67.3 / 58.3, [192,800] 77.9 / 74.1, [224,800] 90.8 / 78.1,
    // [256,800] 97.7 / 82.4, [512,800] 193.3 / 163.7, [256,1000] 143.7 / 117.7, [128,1600] 176.1 / 144.3, [64,3200] 333.9 / 284.5,
    // [512,400] 65.3 / 63.8.  Persistent: the busiest workgroup's items (the cursor of scripts/gen_attn_pw.py restated: full groups
    // with a stride of 32 per XCD, a sequence's tail group attached to one of them; a key-split tail costs 0.55 of a full item) times
    // 0.62 us per key block + 7 us per item, + 5 us per launch.  First generation: 0.54 ns per (query block x key block) + 2.5 ns per
    // query block, per sequence.  The persistent kernel is picked unless the model has it more than 5 % behind.
    const bool ks_tail = !m->batch_invariant;   // key-split tail items (0.55 of a full item) or ordinary ones (a full item's time)
    auto pw_pays = [ks_tail](int Bq, int Tq) {
        const int QBq = (Tq + 31) / 32, NGFq = QBq >> 3, TQq = QBq & 7;
        if (NGFq == 0) return false;
        const int wg = bf::PW_GRID / 8;
        auto ff1 = [](int x) { return __builtin_ctz((unsigned)x); };
        const int t0 = ff1(NGFq) < ff1(wg) ? ff1(NGFq) : ff1(wg), sh = ff1(wg) - t0, mask = (1 << t0) - 1;
        const int S = (Bq + 7) / 8;  // sequences of the fullest XCD
        const double ctail = TQq == 0 ? 0.0 : (TQq <= 2 && ks_tail ? 0.55 : 1.0);
        double busiest = 0.0;
        for (int j = 0; j < wg; ++j) {
            double n = 0.0;
            for (long i = j; i / NGFq < S; i += wg) {
                const int bi = (int)(i / NGFq), g = (int)(i % NGFq);
                n += 1.0 + ((TQq && g == ((bi >> sh) & mask)) ? ctail : 0.0);
            }
            busiest = n > busiest ? n : busiest;
        }
        const double t_pw = busiest * (0.62 * QBq + 7.0) + 5.0;
        const double t_first = (double)Bq * (5.4e-4 * QBq * QBq + 2.5e-3 * QBq);
        return t_pw < 1.05 * t_first;   // (the busiest-workgroup figure errs on the high side when the last round is thin: [320,800] 106.6 measured, 119.7 priced)
    };
    auto run = [&](auto nw_tag) {
        constexpr int NW = decltype(nw_tag)::value;
        constexpr int ring = bf::Ring<NW>::NRING * bf::RING_BYTES;
        const int grid_rows = bp.nblk_pad / NW;
        const dim3 wg(64 * NW);
        // Persistent weights-resident form of the stage (input_qkv_kernel_bf16_p) in the automatic schedules and in 5, from one block per
        // CU up (scripts/ubench/input_p_ab.py, us per launch ring / persistent, fp32 features at T = 800: B=16 19.4 / 17.1, 32 20.5 / 19.1,
        // 64 22.4 / 22.7, 128 39.9 / 33.2, 256 79.6 / 64.9, 512 151.4 / 114.6; the same bits); row_mode 1 - 3 keep the ring kernel.
        const int KSx = F / 16;
        if (SAVAD_INPUT_PERSISTENT && (automatic_bf16 || m->row_mode == 5) && KSx >= 1 && KSx <= 15 && bp.nblk_pad >= m->n_cu) {
            auto go = [&](auto xt, auto ks_tag) {
                using XT = decltype(xt);
                constexpr int KSC = decltype(ks_tag)::value;
                hipLaunchKernelGGL((bf::input_qkv_kernel_bf16_p<XT, 8, KSC>), dim3(m->n_cu), dim3(512), bf::input_p_lds_bytes(KSx), st,
                                   (const XT*)x, xbs, B, T, F, bp.nblk, bp.nblk_pad, Fr + m->f_win, R + m->r_bin, m->d_pe,
                                   Fr + m->lf[0].wqkv, P + m->lp[0].bqkv, hb, qf, kf, vtf, c, m->d_sat);
            };
            if (KSx == 5) {
                if (x_is_bf16) go(__bf16{}, std::integral_constant<int, 5>{}); else go(float{}, std::integral_constant<int, 5>{});
            } else {
                if (x_is_bf16) go(__bf16{}, std::integral_constant<int, 0>{}); else go(float{}, std::integral_constant<int, 0>{});
            }
        } else if (x_is_bf16)
            hipLaunchKernelGGL((bf::input_qkv_kernel_bf16<__bf16, NW>), dim3(grid_rows), wg, ring + 3 * D * 4, st, (const __bf16*)x, xbs,
                               B, T, F, bp.nblk, Fr + m->f_win, R + m->r_bin, m->d_pe, Fr + m->lf[0].wqkv, P + m->lp[0].bqkv, hb,
                               qf, kf, vtf, c, m->d_sat);
        else
            hipLaunchKernelGGL((bf::input_qkv_kernel_bf16<float, NW>), dim3(grid_rows), wg, ring + 3 * D * 4, st, (const float*)x, xbs, B,
                               T, F, bp.nblk, Fr + m->f_win, R + m->r_bin, m->d_pe, Fr + m->lf[0].wqkv, P + m->lp[0].bqkv, hb, qf,
                               kf, vtf, c, m->d_sat);
        prof.mark("input_qkv_bf16");
        char* sets[2][3] = {{qf, kf, vtf}, {W + bp.q2, W + bp.k2, W + bp.vt2}};
        for (int l = 0; l < L; ++l) {
            const auto& r = m->lr[l];
            const auto& p = m->lp[l];
            const auto& f = m->lf[l];
            const bool last = l + 1 == L;
            char** cur = bp.fused ? sets[l & 1] : sets[0];
            char** nxt = bp.fused ? sets[(l + 1) & 1] : sets[0];
            bf::RowArgsBf16 A;
            A.B = B;
            A.T = T;
            A.nblk = bp.nblk;
            A.hbuf = hb;
            A.wo_frag = Fr + f.wo;
            A.bo = R + r.bo;
            A.w1_frag = Fr + f.w1;
            A.b1 = P + p.b1;
            A.w2_frag = Fr + f.w2;
            A.b2 = R + r.b2;
            A.wn_frag = last ? nullptr : Fr + m->lf[l + 1].wqkv;
            A.wc = last ? P + m->p_wc : nullptr;
            A.bn = last ? P + m->p_bc : P + m->lp[l + 1].bqkv;
            A.qf = nxt[0];
            A.kf = nxt[1];
            A.vtf = nxt[2];
            A.out = out;
            A.qscale = c;
            A.satcnt = m->d_sat;
            if (bp.fused) {
                const int QB = (T + 31) / 32, NG = (QB + NW - 1) / NW;
                const dim3 grid(8 * (((long)B * NG + 7) / 8));
                if (last)
                    hipLaunchKernelGGL((bf::attention_row_kernel_bf16<true, NW>), grid, wg, ring + 9 * D * 4, st, cur[0], cur[1], cur[2], NG, A);
                else
                    hipLaunchKernelGGL((bf::attention_row_kernel_bf16<false, NW>), grid, wg, ring + 9 * D * 4, st, cur[0], cur[1], cur[2], NG, A);
                prof.mark(last ? "attention_row_last_bf16" : "attention_row_bf16");
                continue;
            }
            if (T <= 32) {
                hipLaunchKernelGGL(bf::attention_packed_kernel_bf16, dim3((bp.nblk + 3) / 4), dim3(256), 0, st, qf, kf, vtf, ctxf,
                                   B, T, bp.nblk);
            } else if (m->row_mode == 5 || (automatic_bf16 && pw_pays(B, T))) {  // persistent 4 x 64-row attention (savad_attn_pw_bf16.h)
                if (m->batch_invariant)
                    hipLaunchKernelGGL(bf::attention_pw_kernel_bf16_nosplit, dim3(bf::PW_GRID), dim3(256), bf::PW_LDS_BYTES, st, qf, kf, vtf, ctxf, B, T);
                else
                    hipLaunchKernelGGL(bf::attention_pw_kernel_bf16, dim3(bf::PW_GRID), dim3(256), bf::PW_LDS_BYTES, st, qf, kf, vtf, ctxf, B, T);
            } else {
                const int QB = (T + 31) / 32, NG = (QB + NW - 1) / NW;
                hipLaunchKernelGGL((bf::attention_kernel_bf16<NW>), dim3(8 * (((long)B * NG + 7) / 8)), wg, ring, st, qf, kf, vtf, ctxf, B,
                                   T, NG);
            }
            prof.mark("attention_bf16");
            if (last)
                hipLaunchKernelGGL((bf::row_kernel_bf16<true, NW>), dim3(grid_rows), wg, ring + 9 * D * 4, st, ctxf, A);
            else
                hipLaunchKernelGGL((bf::row_kernel_bf16<false, NW>), dim3(grid_rows), wg, ring + 9 * D * 4, st, ctxf, A);
            prof.mark(last ? "row_last_bf16" : "row_bf16");
        }
    };
    if (wide)
        run(std::integral_constant<int, 8>{});
    else
        run(std::integral_constant<int, 4>{});
    prof.done();
    HIP_TRY(hipGetLastError());
    return SAVAD_OK;
}


int prepare_f32s_launch(savad_model* m) {
    int rc;
    if (m->lds_attrs3_set) return SAVAD_OK;
    if ((rc = allow_lds(fs::input_qkv_kernel_f32s, fs::NRING3 * fs::SLOT_BYTES + 3 * D * 4))) return rc;
    if ((rc = allow_lds(fs::attention_row_kernel_f32s<false, false>, fs::ROW_LDS_BYTES))) return rc;
    if ((rc = allow_lds(fs::attention_row_kernel_f32s<true, false>, fs::ROW_LDS_BYTES))) return rc;
    if ((rc = allow_lds(fs::attention_row_kernel_f32s<false, true>, fs::ROW_LDS_BYTES))) return rc;
    if ((rc = allow_lds(fs::attention_row_kernel_f32s<true, true>, fs::ROW_LDS_BYTES))) return rc;
    if ((rc = allow_lds(fs::packed_forward_kernel_f32s, fs::packed_f32s_lds_bytes(fs::PACKED_F32S_MAX_LAYERS)))) return rc;
    if ((rc = allow_lds(fs::packed_forward_kernel_f32s_ns, fs::nsf_lds_bytes(fs::PACKED_F32S_MAX_LAYERS)))) return rc;
    m->lds_attrs3_set = true;
    return SAVAD_OK;
}

#ifndef SAVAD_F32S_NS_MAX_ROUNDS
#define SAVAD_F32S_NS_MAX_ROUNDS 2   // blocks per CU up to which one block per workgroup beats a wave per block (scripts/ubench/f32s_check_t7.py)
#endif
void launch_packed_forward_f32s(savad_model* m, hipStream_t st, const float* x, int B, int T, int F, float* out, const WindowOffsets& wo,
                                int win_base) {
    const int L = m->cfg.num_layers;
    const char* Fr = m->d_frag3;
    const int G = 32 / T, nblk = (B + G - 1) / G;
    fs::PackedF32sModel pm;
    for (int l = 0; l < fs::PACKED_F32S_MAX_LAYERS; ++l) {
        const auto& f = m->lf3[l < L ? l : 0];
        pm.layer[l] = fs::PackedF32sLayer{Fr + f.wqkv, Fr + f.wo, Fr + f.w1, Fr + f.w2};
    }
    pm.win = Fr + m->f3_win;
    pm.bin = m->d_raw + m->r_bin;
    pm.pe = m->d_pe;
    pm.bias = m->d_packed + m->p_bias;
    pm.wc = m->d_packed + m->p_wc;
    pm.bc = m->d_packed + m->p_bc;
    pm.L = L;
    const float c = (float)(1.4426950408889634 / sqrt((double)D));
    // the latency variant (one block per workgroup, its four waves splitting the output features) up to SAVAD_F32S_NS_MAX_ROUNDS blocks
    // per CU; beyond, a wave per block with the weight stream shared through the LDS ring
    const bool ns = m->row_mode >= 5 ? m->row_mode == 8 : nblk <= SAVAD_F32S_NS_MAX_ROUNDS * m->n_cu;
    if (ns)
        hipLaunchKernelGGL(fs::packed_forward_kernel_f32s_ns, dim3(nblk), dim3(256), fs::nsf_lds_bytes(L), st, x, B, T, F, nblk, pm, c, out, wo, win_base);
    else
        hipLaunchKernelGGL(fs::packed_forward_kernel_f32s, dim3((nblk + 3) / 4), dim3(256), fs::packed_f32s_lds_bytes(L), st, x, B, T, F, nblk, pm, c,
                           out, wo, win_base);
}

// fp32s forward (precision 2): input_qkv -> [attention + row chain] x L, every GEMM as six bf16 MFMA products of three-piece operands
int forward_f32s(savad_model* m, const float* x, int B, int T, float* out, void* workspace, size_t workspace_bytes, hipStream_t st,
                 long xbs_in = 0 /* elements between consecutive sequences of x; 0: T * F (savad_forward_strided) */) {
    const BlockPlan3 bp = plan_blocks3(m, B, T);
    if (workspace_bytes < bp.total) return fail(SAVAD_E_INVALID, "workspace too small: %zu < %zu bytes", workspace_bytes, bp.total);
    int rc;
    if ((rc = prepare_weights(m, st))) return rc;
    if ((rc = prepare_frags3(m, st))) return rc;
    if ((rc = ensure_pe(m, T, st))) return rc;
    if ((rc = prepare_f32s_launch(m))) return rc;
    char* W = (char*)workspace;
    float* hb = (float*)(W + bp.h);
    const int L = m->cfg.num_layers;
    int F = m->cfg.feature_size;
    if (m->FP != F) {  // zero-pad the features to the kernels' K granularity
        float* xp = (float*)(W + bp.xpad);
        const size_t rows = (size_t)B * T;
        const int grid = (int)((rows * m->FP + 255) / 256 < 4096 ? (rows * m->FP + 255) / 256 : 4096);
        hipLaunchKernelGGL(pad_rows_kernel<float>, dim3(grid), dim3(256), 0, st, x, rows, F, m->FP, xp);
        x = xp;
        F = m->FP;
    }
    if (xbs_in > 0 && (m->FP != m->cfg.feature_size || T <= 32)) return fail(SAVAD_E_UNSUPPORTED, "strided input: no kernel takes the stride for this shape");
    const long xbs = xbs_in > 0 ? xbs_in : (long)T * F;
    const float c = (float)(1.4426950408889634 / sqrt((double)D));
    const float* R = m->d_raw;
    const float* P = m->d_packed;
    const char* Fr = m->d_frag3;
    Prof prof(m, st);
    if (packed_f32s_applies(m, B, T)) {
        WindowOffsets none;
        none.w = 0;
        launch_packed_forward_f32s(m, st, x, B, T, F, out, none, 0);
        prof.mark("packed_forward_f32s");
        prof.done();
        HIP_TRY(hipGetLastError());
        return SAVAD_OK;
    }
    char* sets[2][3] = {{W + bp.q, W + bp.k, W + bp.vt}, {W + bp.q2, W + bp.k2, W + bp.vt2}};
    hipLaunchKernelGGL(fs::input_qkv_kernel_f32s, dim3(bp.nblk_pad / 4), dim3(256), fs::NRING3 * fs::SLOT_BYTES + 3 * D * 4, st, x, xbs, B, T, F,
                       bp.nblk, Fr + m->f3_win, R + m->r_bin, m->d_pe, Fr + m->lf3[0].wqkv, P + m->lp[0].bqkv, hb, sets[0][0], sets[0][1],
                       sets[0][2], c);
    prof.mark("input_qkv_f32s");
    const bool packed = T <= 32;
    const int QB = (T + 31) / 32, NG = (QB + 3) / 4;
    const dim3 grid(packed ? bp.nblk_pad / 4 : 8 * (((long)B * NG + 7) / 8));
    for (int l = 0; l < L; ++l) {
        const auto& r = m->lr[l];
        const auto& p = m->lp[l];
        const auto& f = m->lf3[l];
        const bool last = l + 1 == L;
        char** cur = sets[l & 1];
        char** nxt = sets[(l + 1) & 1];
        fs::RowArgs3 A;
        A.B = B;
        A.T = T;
        A.nblk = bp.nblk;
        A.hbuf = hb;
        A.wo_frag = Fr + f.wo;
        A.bo = R + r.bo;
        A.w1_frag = Fr + f.w1;
        A.b1 = P + p.b1;
        A.w2_frag = Fr + f.w2;
        A.b2 = R + r.b2;
        A.wn_frag = last ? nullptr : Fr + m->lf3[l + 1].wqkv;
        A.wc = last ? P + m->p_wc : nullptr;
        A.bn = last ? P + m->p_bc : P + m->lp[l + 1].bqkv;
        A.qf = nxt[0];
        A.kf = nxt[1];
        A.vtf = nxt[2];
        A.out = out;
        A.qscale = c;
        if (packed) {
            if (last) hipLaunchKernelGGL((fs::attention_row_kernel_f32s<true, true>), grid, dim3(256), fs::ROW_LDS_BYTES, st, cur[0], cur[1], cur[2], NG, A);
            else hipLaunchKernelGGL((fs::attention_row_kernel_f32s<false, true>), grid, dim3(256), fs::ROW_LDS_BYTES, st, cur[0], cur[1], cur[2], NG, A);
        } else {
            if (last) hipLaunchKernelGGL((fs::attention_row_kernel_f32s<true, false>), grid, dim3(256), fs::ROW_LDS_BYTES, st, cur[0], cur[1], cur[2], NG, A);
            else hipLaunchKernelGGL((fs::attention_row_kernel_f32s<false, false>), grid, dim3(256), fs::ROW_LDS_BYTES, st, cur[0], cur[1], cur[2], NG, A);
        }
        prof.mark(last ? "attention_row_last_f32s" : "attention_row_f32s");
    }
    prof.done();
    HIP_TRY(hipGetLastError());
    return SAVAD_OK;
}

}  // namespace

namespace {
// the single-launch T <= 32 forward (packed_forward_kernel); weights and the PE table must be ready
void launch_packed_forward(savad_model* m, hipStream_t st, const float* x, int B, int T, int F, float* out, const WindowOffsets& wo,
                           int win_base) {
    const int L = m->cfg.num_layers;
    const float* R = m->d_raw;
    const float* P = m->d_packed;
    const float c = (float)(1.4426950408889634 / sqrt((double)D));  // log2(e) / sqrt(d_head)
    const int G = 32 / T, nblk = (B + G - 1) / G;
    PackedModel pm;
    for (int l = 0; l < L; ++l) pm.layer[l] = PackedLayer{P + m->lp[l].frag};
    for (int l = L; l < PACKED_MAX_LAYERS; ++l) pm.layer[l] = pm.layer[0];
    pm.bias = P + m->p_bias;
    pm.win = win_fp32(m);
    pm.bin = R + m->r_bin;
    pm.pe = m->d_pe;
    pm.wc = P + m->p_wc;
    pm.bc = P + m->p_bc;
    pm.L = L;
    hipLaunchKernelGGL(packed_forward_kernel, dim3(nblk), dim3(256), 0, st, x, B * T, T, F, pm, c, out, G * T, wo, win_base);
}
}  // namespace

// x_dtype: 0 = fp32 features, 1 = bf16 features (bf16 precision only)
SAVAD_EXPORT int savad_forward_ex(savad_handle m, const void* x, int x_dtype, int B, int T, float* out, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    if (!m) return fail(SAVAD_E_INVALID, "null handle");
    if (x_dtype == 0 && m->precision != 1) return savad_forward(m, (const float*)x, B, T, out, workspace, workspace_bytes, stream);
    if (x_dtype < 0 || x_dtype > 1) return fail(SAVAD_E_INVALID, "x_dtype %d", x_dtype);
    if (m->generic) return fail(SAVAD_E_UNSUPPORTED, "bf16 features need the d_model=128 kernels (this handle: d_model=%d, fp32)", m->cfg.d_model);
    if (m->precision != 1) return fail(SAVAD_E_UNSUPPORTED, "bf16 features need savad_set_precision(h, 1)");
    if (B < 0 || T < 0) return fail(SAVAD_E_INVALID, "negative shape B=%d T=%d", B, T);
    if (B == 0 || T == 0) return SAVAD_OK;
    if (!x || !out || !workspace) return fail(SAVAD_E_INVALID, "null tensor pointer");
    if ((double)B * T * D >= 2.0e9) return fail(SAVAD_E_UNSUPPORTED, "B*T too large");
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)workspace) & 15)
        return fail(SAVAD_E_INVALID, "x, out and workspace must be 16-byte aligned");
    return forward_bf16(m, x, x_dtype, B, T, out, workspace, workspace_bytes, (hipStream_t)stream);
}

int forward_any(savad_handle m, const void* xv, int x_dtype, int B, int T, long xbs_in, float* out, void* workspace, size_t workspace_bytes,
                void* stream);
SAVAD_EXPORT int savad_forward_strided(savad_handle m, const void* x, int x_dtype, int B, int T, long x_batch_stride, float* out,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    if (!m) return fail(SAVAD_E_INVALID, "null handle");
    if (m->generic) return fail(SAVAD_E_UNSUPPORTED, "strided input needs the d_model=128 kernels");
    if (T <= 32) return fail(SAVAD_E_UNSUPPORTED, "strided input is for sequences longer than 32 frames (windows of T <= 32 are read in place by savad_predict_probabilities)");
    if (m->FP != m->cfg.feature_size) return fail(SAVAD_E_UNSUPPORTED, "strided input needs feature_size %% 16 == 0 (no padding copy)");
    const long F = m->cfg.feature_size;
    if (x_batch_stride <= 0 || x_batch_stride % 4 || x_batch_stride % F)
        return fail(SAVAD_E_INVALID, "x_batch_stride=%ld (a positive multiple of feature_size and of 4 elements)", x_batch_stride);
    if (x_dtype < 0 || x_dtype > 1 || (x_dtype == 1 && m->precision != 1)) return fail(SAVAD_E_INVALID, "x_dtype %d (bf16 features need savad_set_precision(h, 1))", x_dtype);
    return forward_any(m, x, x_dtype, B, T, x_batch_stride, out, workspace, workspace_bytes, stream);
}

SAVAD_EXPORT int savad_forward(savad_handle m, const float* x, int B, int T, float* out, void* workspace,
                               size_t workspace_bytes, void* stream) {
    return forward_any(m, x, 0, B, T, 0, out, workspace, workspace_bytes, stream);
}

// every forward entry point ends here.  xbs_in: elements between consecutive sequences of x (savad_forward_strided), 0 = T * F; it
// travels as an ARGUMENT to the one kernel per precision that reads the features, and every path that cannot honour it refuses
int forward_any(savad_handle m, const void* xv, int x_dtype, int B, int T, long xbs_in, float* out, void* workspace, size_t workspace_bytes,
                void* stream) {
    const float* x = (const float*)xv;
    if (!m) return fail(SAVAD_E_INVALID, "null handle");
    if (B < 0 || T < 0) return fail(SAVAD_E_INVALID, "negative shape B=%d T=%d", B, T);
    if (B == 0 || T == 0) return SAVAD_OK;
    if (!x || !out || !workspace) return fail(SAVAD_E_INVALID, "null tensor pointer");
    if ((double)B * T * D >= 2.0e9) return fail(SAVAD_E_UNSUPPORTED, "B*T too large");
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)workspace) & 15)
        return fail(SAVAD_E_INVALID, "x, out and workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (m->generic) return xbs_in > 0 ? fail(SAVAD_E_UNSUPPORTED, "strided input needs the d_model=128 kernels") : forward_generic(m, x, B, T, out, workspace, workspace_bytes, st);
    if (m->precision == 1) return forward_bf16(m, xv, x_dtype, B, T, out, workspace, workspace_bytes, st, xbs_in);
    if (m->precision == 2 && !f32s_uses_exact_fp32(m, B, T)) return forward_f32s(m, x, B, T, out, workspace, workspace_bytes, st, xbs_in);
    if (xbs_in > 0 && (m->FP != m->cfg.feature_size || T <= 32)) return fail(SAVAD_E_UNSUPPORTED, "strided input: no kernel takes the stride for this shape");
    const Workspace ws = plan(m, B, T);
    if (workspace_bytes < ws.total * sizeof(float))
        return fail(SAVAD_E_INVALID, "workspace too small: %zu < %zu bytes", workspace_bytes, ws.total * sizeof(float));
    int rc;
    if ((rc = prepare_weights(m, st))) return rc;
    if ((rc = ensure_pe(m, T, st))) return rc;

    float* W = (float*)workspace;
    float *hb = W + ws.h, *q = W + ws.q, *k = W + ws.k, *v = W + ws.v, *op = W + ws.opart, *ml = W + ws.ml;
    const int L = m->cfg.num_layers;
    int F = m->cfg.feature_size;
    if (m->FP != F) {  // zero-pad the features to the kernels' K granularity
        float* xp = W + ws.xpad;
        const int grid = (int)((ws.rows * m->FP + 255) / 256 < 4096 ? (ws.rows * m->FP + 255) / 256 : 4096);
        hipLaunchKernelGGL(pad_rows_kernel<float>, dim3(grid), dim3(256), 0, st, x, ws.rows, F, m->FP, xp);
        x = xp;
        F = m->FP;
    }
    const long xbs = xbs_in > 0 ? xbs_in : (long)T * F;
    const int tiles = (int)(ws.rows_pad / TILE);
    const float c = (float)(1.4426950408889634 / sqrt((double)D));  // log2(e) / sqrt(d_head)
    const float* R = m->d_raw;
    const float* P = m->d_packed;
    Prof prof(m, st);

    const int tiles_m = (int)(ws.rows_pad / 128);
    const bool msplit = ws.msplit;
    // T <= 32 in the small-batch regime (the reference pipeline's 7-frame windows): the whole forward is ONE launch,
    // a workgroup per packed tile of floor(32/T) sequences keeps every activation on its CU (packed_forward_kernel).
    // Automatic up to 1024 tiles (four rounds of the 256 CUs); beyond that the 128-row M-split tiles, which fetch the
    // weight stream once per 128 rows instead of once per tile, are ahead.  Measured at T=7, ms per forward, single
    // launch / per-layer N-split launches / M-split: 250 tiles 0.100 / 0.156 / 0.356; 512 tiles 0.186 / 0.272 / 0.364;
    // 1024 tiles 0.366 / 0.506 / 0.383; 2048 tiles 0.728 / 0.878 / 0.699; 4096 tiles (the predictor's 16384-window
    // batches) 1.454 / 1.736 / 1.346.  row_mode 4 forces the single launch for any T <= 32 batch.
    // A packed tile holds floor(32/T)*T of 32 rows (28 at T=7, 20 at T=20, 17 at T=17): while the DENSE 32-row tiles of the
    // per-layer N-split launches still fit fewer rounds of the CUs, those win (T=20, B=400: 400 packed / 250 dense tiles,
    // 0.184 against 0.160 ms).  Round model fitted to scripts/ubench/policy_sweep.py: 92 us per round of packed tiles, 45 +
    // 110 us per round of dense tiles.
    const long tiles_packed = T <= 32 ? ((long)B + 32 / T - 1) / (32 / T) : 0, tiles_dense = ((long)B * T + 31) / 32;
    const bool one_launch = tiles_packed <= 1024 &&
                            (tiles_dense > 512 || 92 * ((tiles_packed + 255) / 256) <= 45 + 110 * ((tiles_dense + 255) / 256));
    if (T <= 32 && L <= PACKED_MAX_LAYERS && (m->row_mode == 4 || (m->row_mode == 0 && one_launch))) {
        WindowOffsets none;
        none.w = 0;
        launch_packed_forward(m, st, x, B, T, F, out, none, 0);
        prof.mark("packed_forward");
        prof.done();
        HIP_TRY(hipGetLastError());
        return SAVAD_OK;
    }
    if (msplit)
        hipLaunchKernelGGL(input_qkv_kernel_m, dim3(tiles_m), dim3(256), 0, st, x, xbs, (int)ws.rows, T, F, win_fp32(m),
                           R + m->r_bin, m->d_pe, P + m->lp[0].wqkv, P + m->lp[0].bqkv, hb, q, k, v);
    else
        hipLaunchKernelGGL(input_qkv_kernel, dim3(tiles), dim3(256), 0, st, x, xbs, (int)ws.rows, T, F, win_fp32(m),
                           R + m->r_bin, m->d_pe, P + m->lp[0].frag, P + m->lp[0].bqkv, hb, q, k, v);
    prof.mark("input_qkv");
    if (ws.fused) {
        float* qkv[2][3] = {{q, k, v}, {W + ws.q2, W + ws.k2, W + ws.v2}};
        const int QB = (T + 31) / 32, NG = (QB + 3) / 4;
        const int grid = (int)(8 * (((long)B * NG + 7) / 8));
        for (int l = 0; l < L; ++l) {
            const auto& r = m->lr[l];
            const auto& p = m->lp[l];
            float** cur = qkv[l & 1];
            float** nxt = qkv[(l + 1) & 1];
            if (l + 1 < L) {
                hipLaunchKernelGGL(attention_row_kernel<false>, dim3(grid), dim3(256), 0, st, cur[0], cur[1], cur[2], B, T, NG, c, hb,
                                   R + r.wo, R + r.bo, P + p.w1, P + p.b1, R + r.w2, R + r.b2, P + m->lp[l + 1].wqkv,
                                   P + m->lp[l + 1].bqkv, nxt[0], nxt[1], nxt[2], out);
                prof.mark("attention_row");
            } else {
                hipLaunchKernelGGL(attention_row_kernel<true>, dim3(grid), dim3(256), 0, st, cur[0], cur[1], cur[2], B, T, NG, c, hb,
                                   R + r.wo, R + r.bo, P + p.w1, P + p.b1, R + r.w2, R + r.b2, P + m->p_wc, P + m->p_bc, nxt[0],
                                   nxt[1], nxt[2], out);
                prof.mark("attention_row_last");
            }
        }
        prof.done();
        HIP_TRY(hipGetLastError());
        return SAVAD_OK;
    }
    for (int l = 0; l < L; ++l) {
        if (T <= 32) {
            const int G = 32 / T, nblk = (B + G - 1) / G;
            hipLaunchKernelGGL(attention_packed_kernel, dim3(nblk), dim3(64), 0, st, q, k, v, op, ml, B, T, (int)ws.rows, c);
        } else {
            const int QB = (T + 31) / 32, NG = (QB + 3) / 4;
            const int grid = (int)(8 * (((long)B * NG * ws.S + 7) / 8));
            hipLaunchKernelGGL(attention_kernel, dim3(grid), dim3(256), 0, st, q, k, v, op, ml, B, T, (int)ws.rows_pad,
                               ws.S, NG, c);
        }
        prof.mark("attention");
        const auto& r = m->lr[l];
        const auto& p = m->lp[l];
#define SAVAD_ROW_ARGS(WN, BN) op, ml, ws.S, (int)ws.rows, (int)ws.rows_pad, c, hb, R + r.wo, R + r.bo, P + p.w1, P + p.b1, \
                               R + r.w2, R + r.b2, WN, BN, q, k, v, out
#define SAVAD_ROWN_ARGS(NFRAG, WN, BN) op, ml, ws.S, (int)ws.rows, (int)ws.rows_pad, c, hb, P + p.frag, R + r.bo, P + p.b1, R + r.b2, \
                                       NFRAG, WN, BN, q, k, v, out
        if (l + 1 < L) {
            if (msplit)
                hipLaunchKernelGGL(row_kernel_m<false>, dim3(tiles_m), dim3(256), 0, st,
                                   SAVAD_ROW_ARGS(P + m->lp[l + 1].wqkv, P + m->lp[l + 1].bqkv));
            else
                hipLaunchKernelGGL(row_kernel<false>, dim3(tiles), dim3(256), 0, st, SAVAD_ROWN_ARGS(P + m->lp[l + 1].frag, P, P + m->lp[l + 1].bqkv));
            prof.mark("row");
        } else {
            if (msplit)
                hipLaunchKernelGGL(row_kernel_m<true>, dim3(tiles_m), dim3(256), 0, st,
                                   SAVAD_ROW_ARGS(P + m->p_wc, P + m->p_bc));
            else
                hipLaunchKernelGGL(row_kernel<true>, dim3(tiles), dim3(256), 0, st, SAVAD_ROWN_ARGS(P + p.frag, P + m->p_wc, P + m->p_bc));
            prof.mark("row_last");
        }
#undef SAVAD_ROW_ARGS
#undef SAVAD_ROWN_ARGS
    }
    prof.done();
    HIP_TRY(hipGetLastError());
    return SAVAD_OK;
}

SAVAD_EXPORT int savad_set_profiling(savad_handle m, int capacity) {
    if (!m || capacity < 0) return fail(SAVAD_E_INVALID, "bad argument");
    for (hipEvent_t e : m->events) hipEventDestroy(e);
    m->events.clear();
    m->prof_capacity = capacity;
    m->prof_used = 0;
    m->prof_nk = 0;
    m->prof_skip = 0;
    m->events.resize((size_t)capacity * MAX_EVENTS);
    for (auto& e : m->events) HIP_TRY(hipEventCreate(&e));
    return SAVAD_OK;
}

SAVAD_EXPORT int savad_profiling_skip(savad_handle m, int forwards) {
    if (!m || forwards < 0) return fail(SAVAD_E_INVALID, "bad argument");
    m->prof_skip = forwards;
    return SAVAD_OK;
}

// Average duration (ms) of each launch position over the profiled forwards recorded since
// savad_set_profiling; the caller must have synchronised the stream.
SAVAD_EXPORT int savad_last_kernel_times(savad_handle m, const char** names, float* ms, int max) {
    if (!m || m->prof_used == 0) return 0;
    const int nk = m->prof_nk < max ? m->prof_nk : max;
    for (int i = 0; i < nk; ++i) {
        double acc = 0;
        for (int f = 0; f < m->prof_used; ++f) {
            hipEvent_t* ev = m->events.data() + (size_t)f * MAX_EVENTS;
            float t = 0;
            if (hipEventElapsedTime(&t, ev[i], ev[i + 1]) != hipSuccess) return fail(SAVAD_E_HIP, "hipEventElapsedTime failed");
            acc += t;
        }
        ms[i] = (float)(acc / m->prof_used);
        if (names) names[i] = m->knames[i];
    }
    m->prof_used = 0;
    return nk;
}

SAVAD_EXPORT int savad_window_offsets(int half, int jump, int32_t* offsets) {
    if (half < 0 || jump <= 0) return fail(SAVAD_E_INVALID, "half=%d jump=%d", half, jump);
    int w = 0;
    for (int o = -half; o < 0; o += jump, ++w)
        if (offsets) offsets[w] = o;
    if (offsets) offsets[w] = 0;
    ++w;
    for (int o = 1; o < half + 1; o += jump, ++w)
        if (offsets) offsets[w] = o;
    return w;
}

SAVAD_EXPORT int savad_gather_windows(const float* feature, int N, int F, int half, int jump, int first, int count,
                                      float* windows, int64_t* positions, void* stream) {
    if (count == 0) return SAVAD_OK;
    if (!feature || !windows || N <= 0 || F <= 0 || F % 4 || first < 0 || count < 0)
        return fail(SAVAD_E_INVALID, "bad argument (F must be a multiple of 4)");
    WindowOffsets wo;
    int32_t off[256];
    if (savad_window_offsets(half, jump, nullptr) > 64) return fail(SAVAD_E_UNSUPPORTED, "window longer than 64 frames");
    wo.w = savad_window_offsets(half, jump, off);
    for (int i = 0; i < wo.w; ++i) wo.off[i] = off[i];
    if ((long)half + first + count - 1 + off[wo.w - 1] >= N || half + first + off[0] < 0)
        return fail(SAVAD_E_INVALID, "window [%d,%d) reaches outside the %d feature frames", first, first + count, N);
    const size_t total = (size_t)count * wo.w * (F / 4);
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(gather_windows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, feature, F, half, first,
                       count, wo, windows, positions);
    HIP_TRY(hipGetLastError());
    return SAVAD_OK;
}

SAVAD_EXPORT int savad_boost(const float* logp, const int64_t* positions, int count, int N, int W, float* boosted_ws,
                             float* probs, float* mean, void* stream) {
    if (N <= 0) return SAVAD_OK;
    if (!boosted_ws || !probs || W <= 0 || count < 0 || (count > 0 && (!logp || !positions)))
        return fail(SAVAD_E_INVALID, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(boosted_ws, 0, sizeof(float) * (size_t)N * W * 2, st));
    if (count > 0) {
        const size_t cw = (size_t)count * W;
        const int grid = (int)((cw + 255) / 256 < 2048 ? (cw + 255) / 256 : 2048);
        hipLaunchKernelGGL(boost_scatter_kernel, dim3(grid), dim3(256), 0, st, logp, positions, cw, W, boosted_ws);
    }
    const int grid2 = (N + 255) / 256 < 2048 ? (N + 255) / 256 : 2048;
    hipLaunchKernelGGL(boost_softmax_kernel, dim3(grid2), dim3(256), 0, st, boosted_ws, N, W, probs, mean);
    HIP_TRY(hipGetLastError());
    return SAVAD_OK;
}

// ---- the whole of predict_probabilities (vad/predictor.py:159-262) in one call -------------------------------
namespace {
struct PredictPlan {
    int W, n_items, chunk;
    bool windowed;  // the single-launch forward reads its windows straight out of the feature matrix
    bool f32s;      // ... and it is the fp32s one (precision 2 from SAVAD_F32S_PACKED_MIN_BLOCKS blocks up; shorter clips: the exact-fp32 one)
    size_t logp, windows, fwd, total;  // byte offsets into the workspace
    size_t fwd_bytes;
};
int plan_predict(savad_model* m, int N, int half, int jump, int chunk, PredictPlan* p) {
    if (N < 0 || half < 0 || jump <= 0 || chunk <= 0) return fail(SAVAD_E_INVALID, "N=%d half=%d jump=%d chunk=%d", N, half, jump, chunk);
    p->W = savad_window_offsets(half, jump, nullptr);
    if (p->W > 64) return fail(SAVAD_E_UNSUPPORTED, "window longer than 64 frames");
    p->n_items = N - 2 * half > 0 ? N - 2 * half : 0;  // vad/predictor.py:169
    const int F = m->cfg.feature_size;
    // windowed: the whole clip in ONE single-launch forward (up to 1024 packed tiles = 4096 windows of 7 frames, ~41 s of
    // audio: savad_forward's own limit for that kernel); longer inputs go through `chunk`-sized M-split forwards, which
    // are ~9 % faster per window than 4096-window launches (5.27 vs 5.36 ms for 10 min of audio)
    // (bf16 operands: the single launch amortises the weight stream over the workgroup's blocks, so it takes any number of windows)
    p->f32s = m->precision == 2 && !m->generic && p->W <= 32 && m->FP == F && p->n_items > 0 && packed_f32s_applies(m, p->n_items, p->W);
    if (m->precision == 1)
        p->windowed = !m->generic && p->W <= 32 && m->FP == F && packed_bf16_applies(m, p->W);
    else if (p->f32s)   // (the fp32s single launch takes any number of windows)
        p->windowed = true;
    else
        p->windowed = !m->generic && p->W <= 32 && m->cfg.num_layers <= PACKED_MAX_LAYERS && m->FP == F &&
                      (m->row_mode == 4 || (m->row_mode == 0 && p->n_items <= 1024 * (32 / p->W)));
    // chunk-sized forwards write their log-probs at logp + first*W*2 floats and savad_forward wants 16-byte aligned
    // pointers: an even chunk keeps every offset a multiple of 16 bytes whatever W is (windows are independent, so the
    // chunking never changes a result beyond fp32 summation order)
    p->chunk = p->windowed ? (m->precision == 1 || p->f32s ? (1 << 22) : 1024 * (32 / p->W)) : chunk + (chunk & 1);
    if (p->chunk > p->n_items) p->chunk = p->n_items > 0 ? p->n_items : 1;
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    size_t off = 0;
    p->logp = off;
    off += up(sizeof(float) * (size_t)(p->n_items > 0 ? p->n_items : 1) * p->W * 2);
    p->windows = p->fwd = off;
    p->fwd_bytes = 0;
    if (!p->windowed) {
        off += up(sizeof(float) * (size_t)p->chunk * p->W * F);
        p->fwd = off;
        int rc = savad_workspace_bytes(m, p->chunk, p->W, &p->fwd_bytes);
        if (rc) return rc;
        off += up(p->fwd_bytes);
    }
    p->total = off;
    return SAVAD_OK;
}
}  // namespace

SAVAD_EXPORT int savad_predict_workspace_bytes(savad_handle m, int N, int half, int jump, int chunk, size_t* bytes) {
    if (!m || !bytes) return fail(SAVAD_E_INVALID, "null argument");
    PredictPlan p;
    int rc = plan_predict(m, N, half, jump, chunk, &p);
    if (rc) return rc;
    *bytes = p.total;
    return SAVAD_OK;
}

SAVAD_EXPORT int savad_predict_probabilities(savad_handle m, const float* feature, int N, int half, int jump, int chunk, float* probs,
                                             float* mean, void* workspace, size_t workspace_bytes, void* stream) {
    if (!m) return fail(SAVAD_E_INVALID, "null handle");
    if (N == 0) return SAVAD_OK;
    if (!feature || !probs || !workspace) return fail(SAVAD_E_INVALID, "null tensor pointer");
    if (((uintptr_t)feature | (uintptr_t)workspace) & 15) return fail(SAVAD_E_INVALID, "feature and workspace must be 16-byte aligned");
    PredictPlan p;
    int rc = plan_predict(m, N, half, jump, chunk, &p);
    if (rc) return rc;
    if (workspace_bytes < p.total) return fail(SAVAD_E_INVALID, "workspace too small: %zu < %zu bytes", workspace_bytes, p.total);
    hipStream_t st = (hipStream_t)stream;
    const int F = m->cfg.feature_size, W = p.W;
    if (F % 4) return fail(SAVAD_E_INVALID, "F must be a multiple of 4");
    WindowOffsets wo;
    int32_t off[64];
    wo.w = savad_window_offsets(half, jump, off);
    for (int i = 0; i < wo.w; ++i) wo.off[i] = off[i];
    if (p.n_items > 0 && (half + off[0] < 0 || (long)half + p.n_items - 1 + off[W - 1] >= N))
        return fail(SAVAD_E_INVALID, "windows reach outside the %d feature frames", N);
    char* ws = (char*)workspace;
    float* logp = (float*)(ws + p.logp);
    if (p.windowed && p.n_items > 0) {
        if ((rc = prepare_weights(m, st))) return rc;
        if ((rc = ensure_pe(m, W, st))) return rc;
        if (m->precision == 1) {
            if ((rc = prepare_frags(m, st))) return rc;
            if ((rc = prepare_bf16_launch(m))) return rc;
        }
        if (p.f32s) {
            if ((rc = prepare_frags3(m, st))) return rc;
            if ((rc = prepare_f32s_launch(m))) return rc;
        }
    }
    for (int first = 0; first < p.n_items; first += p.chunk) {
        const int count = p.n_items - first < p.chunk ? p.n_items - first : p.chunk;
        float* out = logp + (size_t)first * W * 2;
        if (p.windowed && m->precision == 1) {
            launch_packed_forward_bf16(m, st, feature, count, W, F, out, wo, half + first);
        } else if (p.windowed && p.f32s) {
            launch_packed_forward_f32s(m, st, feature, count, W, F, out, wo, half + first);
        } else if (p.windowed) {
            launch_packed_forward(m, st, feature, count, W, F, out, wo, half + first);
        } else {
            float* win = (float*)(ws + p.windows);
            if ((rc = savad_gather_windows(feature, N, F, half, jump, first, count, win, nullptr, stream))) return rc;
            if ((rc = savad_forward(m, win, count, W, out, ws + p.fwd, p.fwd_bytes, stream))) return rc;
        }
    }
    const int grid = (N + 255) / 256 < 2048 ? (N + 255) / 256 : 2048;
    hipLaunchKernelGGL(boost_gather_kernel, dim3(grid), dim3(256), 0, st, logp, p.n_items, N, half, wo, probs, mean);
    HIP_TRY(hipGetLastError());
    return SAVAD_OK;
}

SAVAD_EXPORT int savad_stream_window_count(int N, int T, int hop) {
    if (N <= 0 || T <= 0 || hop <= 0) return fail(SAVAD_E_INVALID, "N=%d T=%d hop=%d", N, T, hop);
    return N <= T ? 1 : (N - T + hop - 1) / hop + 1;
}

SAVAD_EXPORT int savad_gather_strided(const float* feature, int N, int F, int T, int hop, int first, int count,
                                      float* windows, void* stream) {
    if (count == 0) return SAVAD_OK;
    if (!feature || !windows || N <= 0 || F <= 0 || F % 4 || T <= 0 || hop <= 0 || first < 0 || count < 0)
        return fail(SAVAD_E_INVALID, "bad argument (F must be a multiple of 4)");
    const size_t total = (size_t)count * T * (F / 4);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(gather_strided_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, feature, N, F, T, hop, first,
                       count, windows);
    HIP_TRY(hipGetLastError());
    return SAVAD_OK;
}

SAVAD_EXPORT int savad_overlap_merge(const float* logp, int W, int N, int T, int hop, float* probs, void* stream) {
    if (N <= 0) return SAVAD_OK;
    if (!logp || !probs || W <= 0 || T <= 0 || hop <= 0) return fail(SAVAD_E_INVALID, "bad argument");
    const int grid = (N + 255) / 256 < 2048 ? (N + 255) / 256 : 2048;
    hipLaunchKernelGGL(overlap_merge_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logp, W, N, T, hop, probs);
    HIP_TRY(hipGetLastError());
    return SAVAD_OK;
}

// ---- log-mel front-end (savad_logmel.h) ---------------------------------------------------------
namespace {

struct MelTables {
    float* d_dft = nullptr;  // DFT-as-GEMM kernel (algorithm 1)
    float* d_mel = nullptr;
    float* d_t1 = nullptr;  // factored kernel (algorithm 0)
    float* d_t3 = nullptr;
    float* d_tm = nullptr;
    int n_cu = 0;
};
std::mutex g_mel_mutex;
std::map<int, MelTables> g_mel_by_device;  // built once per device, never freed (process lifetime, 1.1 MB)
int g_logmel_algorithm = 0;

double hz_to_mel(double f) {  // Slaney scale (librosa htk=False)
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double mm) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    return mm >= min_log_mel ? min_log_hz * exp(logstep * (mm - min_log_mel)) : f_sp * mm;
}

// Slaney mel filterbank (librosa.filters.mel, norm="slaney", float32): M[mel][bin], 80 x 257
std::vector<float> mel_filterbank() {
    using namespace mel;
    const int NB = N_FFT / 2 + 1;
    std::vector<double> mel_f(N_MELS + 2);
    const double m_lo = hz_to_mel(0.0), m_hi = hz_to_mel(8000.0);
    for (int i = 0; i < N_MELS + 2; ++i) mel_f[i] = mel_to_hz(m_lo + (m_hi - m_lo) * i / (N_MELS + 1));
    std::vector<float> M((size_t)N_MELS * NB, 0.0f);
    for (int i = 0; i < N_MELS; ++i) {
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int b = 0; b < NB; ++b) {
            const double fr = 8000.0 * b / (NB - 1);
            const double lower = (fr - mel_f[i]) / (mel_f[i + 1] - mel_f[i]);
            const double upper = (mel_f[i + 2] - fr) / (mel_f[i + 2] - mel_f[i + 1]);
            const double wgt = fmax(0.0, fmin(lower, upper));
            M[(size_t)i * NB + b] = (float)wgt * (float)enorm;
        }
    }
    return M;
}

// A operands of logmel_fft_kernel (layouts: savad_logmel.h).  Pure host code; false when a filter weight falls outside
// the kernel's fixed (register pair -> mel block) pattern, which would be a bug in the bin ordering.
bool build_fft_tables(std::vector<float>& t1, std::vector<float>& t3, std::vector<float>& tm) {
    using namespace mel;
    const double PI = 3.14159265358979323846;
    const int NB = N_FFT / 2 + 1;
    t1.assign(FFT_T1_FLOATS, 0.0f);
    t3.assign(FFT_T3_FLOATS, 0.0f);
    tm.assign(FFT_TM_FLOATS, 0.0f);
    // step 1: row i = lane & 31 of the output tile is (k1 = 4 (i >> 3) + (i & 3), re | im = (i >> 2) & 1); k-step s of
    // lane half h is n1 = 3 + 2 s + h; the slot of im(k1 = 0) carries re(k1 = 16)
    for (int w = 0; w < 4; ++w)
        for (int s = 0; s < 13; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int i = lane & 31, h = lane >> 5, n1 = 3 + 2 * s + h, n2 = 4 * w + e, n = 16 * n1 + n2;
                    int k1 = 4 * (i >> 3) + (i & 3);
                    bool im = (i >> 2) & 1;
                    if (k1 == 0 && im) {
                        k1 = 16;
                        im = false;
                    }
                    float win = 0.0f;
                    if (n >= LPAD && n < LPAD + WIN) win = (float)(0.5 - 0.5 * cos(2.0 * PI * (n - LPAD) / WIN));  // periodic Hann(400), float32 as librosa's
                    const double ph = 2.0 * PI * (double)(n1 * k1 % 32) / 32.0;
                    t1[(((size_t)w * 13 + s) * 64 + lane) * 4 + e] = (float)(win * (im ? -sin(ph) : cos(ph)));
                }
    // bins of a group, lowest first.  kind 0: bin K = 32 k2 from Y[0] (real, low lane half); kind 1: K = 16 + 32 k2 from
    // Y[16] (real, high lane half); kind 2: K = k1 + 32 k2 from the complex Y[k1].  label = the bin below 257 with the same power.
    struct Bin {
        int kind, K, label;
    };
    std::vector<std::vector<Bin>> groups(16);
    for (int grp = 0; grp < 16; ++grp) {
        std::vector<Bin>& g = groups[grp];
        if (grp == 0) {
            for (int k2 = 1; k2 < 8; ++k2) g.push_back({0, 32 * k2, 32 * k2});
            for (int k2 = 0; k2 < 8; ++k2) g.push_back({1, 16 + 32 * k2, 16 + 32 * k2});
        } else {
            for (int k2 = 0; k2 < 16; ++k2) {
                const int K = grp + 32 * k2;
                g.push_back({2, K, K <= 256 ? K : 512 - K});
            }
        }
        for (size_t a = 1; a < g.size(); ++a)  // insertion sort by label
            for (size_t b = a; b > 0 && g[b].label < g[b - 1].label; --b) std::swap(g[b], g[b - 1]);
    }
    // step 3: row i of the output tile is (register pair p = 2 (i >> 3) + ((i & 3) >> 1), lane half hD = (i >> 2) & 1,
    // re | im = i & 1) = bin number 2 p + hD of the group; k-step n2 of lane half h multiplies re (h = 0) / im (h = 1) of Y
    for (int grp = 0; grp < 16; ++grp)
        for (int n2 = 0; n2 < 16; ++n2)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, h = lane >> 5;
                const int p = 2 * (i >> 3) + ((i & 3) >> 1), hD = (i >> 2) & 1, ro = i & 1, idx = 2 * p + hD;
                float v = 0.0f;
                if (idx < (int)groups[grp].size()) {
                    const Bin& b = groups[grp][idx];
                    const double th = 2.0 * PI * (double)((long)n2 * b.K % N_FFT) / N_FFT, c = cos(th), sn = sin(th);
                    // X = sum (Yre + i Yim)(c - i sn):  re = Yre c + Yim sn,  im = Yim c - Yre sn
                    if (b.kind == 2)
                        v = (float)(h == 0 ? (ro == 0 ? c : -sn) : (ro == 0 ? sn : c));
                    else if (b.kind == h)
                        v = (float)(ro == 0 ? c : -sn);
                }
                t3[(((size_t)grp * 4 + (n2 >> 2)) * 64 + lane) * 4 + (n2 & 3)] = v;
            }
    // mel: entry t of a group = (pair, mel block) in the kernel's fixed order
    static const int PAIR[10] = {0, 1, 1, 2, 3, 4, 4, 5, 6, 7}, BLOCK[10] = {0, 0, 1, 1, 1, 1, 2, 2, 2, 2};
    const std::vector<float> M = mel_filterbank();
    std::vector<char> covered((size_t)N_MELS * NB, 0);
    for (int grp = 0; grp < 16; ++grp)
        for (int t = 0; t < 10; ++t)
            for (int lane = 0; lane < 64; ++lane) {
                const int j = lane & 31, h = lane >> 5, idx = 2 * PAIR[t] + h, ml = 32 * BLOCK[t] + j;
                if (idx < (int)groups[grp].size() && ml < N_MELS) {
                    const int label = groups[grp][idx].label;
                    tm[(((size_t)grp * 3 + (t >> 2)) * 64 + lane) * 4 + (t & 3)] = M[(size_t)ml * NB + label];
                    covered[(size_t)ml * NB + label] = 1;
                }
            }
    for (size_t q = 0; q < M.size(); ++q)
        if (M[q] != 0.0f && !covered[q]) return false;
    return true;
}

int ensure_mel_tables(hipStream_t st, MelTables* out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mel_mutex);
    auto it = g_mel_by_device.find(dev);
    if (it != g_mel_by_device.end()) {
        *out = it->second;
        return SAVAD_OK;
    }
    MelTables g_mel;
    using namespace mel;
    const double PI = 3.14159265358979323846;
    // window-folded DFT rows in fragment order [row block 16][G 50][lane 64][4]:
    // row 0 = re(bin 0), row 1 = re(bin 256) (both imaginary parts are identically 0), row 2b / 2b+1 = re / im of bin b
    std::vector<float> dft((size_t)DFT_FRAG_FLOATS);
    for (int rb = 0; rb < 16; ++rb)
        for (int G = 0; G < KG; ++G)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int n = lane & 31, hh = lane >> 5, row = 32 * rb + n, kk = 8 * G + 4 * hh + e;
                    const double win = 0.5 - 0.5 * cos(2.0 * PI * kk / WIN);  // periodic Hann(400)
                    const int kp = kk + LPAD;                                  // position inside the 512-sample frame
                    int bin = row >> 1;
                    bool im = row & 1;
                    if (row == 1) {
                        bin = 256;
                        im = false;
                    }
                    const double ph = 2.0 * PI * (double)((long)bin * kp % N_FFT) / N_FFT;
                    dft[(((size_t)rb * KG + G) * 64 + lane) * 4 + e] = (float)((float)win * (im ? -sin(ph) : cos(ph)));
                }
    const int NB = N_FFT / 2 + 1;
    const std::vector<float> M = mel_filterbank();
    // mel fragments [pass 4][mel block 3][row block 4][g pair 2][lane 64][4]; element e -> g = 2gp + (e>>1), bin
    // 64 pass + 16 rbl + 4 g + 2 h + (e&1).  Bins 0 and 256 have zero weight in every filter (fmin 0, fmax 8 kHz).
    std::vector<float> melf((size_t)MEL_FRAG_FLOATS, 0.0f);
    for (int pass = 0; pass < 4; ++pass)
        for (int mb = 0; mb < 3; ++mb)
            for (int rbl = 0; rbl < 4; ++rbl)
                for (int gp = 0; gp < 2; ++gp)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int n = lane & 31, hh = lane >> 5, g = 2 * gp + (e >> 1);
                            const int bin = 64 * pass + 16 * rbl + 4 * g + 2 * hh + (e & 1);
                            const int ml_ = 32 * mb + n;
                            float v = 0.0f;
                            if (ml_ < N_MELS && bin >= 1 && bin < 256) v = M[(size_t)ml_ * NB + bin];
                            melf[(((((size_t)pass * 3 + mb) * 4 + rbl) * 2 + gp) * 64 + lane) * 4 + e] = v;
                        }
    std::vector<float> t1, t3, tm;
    if (!build_fft_tables(t1, t3, tm)) return fail(SAVAD_E_STATE, "log-mel tables: a filter weight falls outside the kernel's pair -> mel block pattern");
    HIP_TRY(hipDeviceGetAttribute(&g_mel.n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mel::logmel_fft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, mel::FFT_LDS_BYTES));
    float* base = nullptr;  // one allocation: every table starts 256-byte aligned
    auto up = [](size_t n) { return (n + 63) / 64 * 64; };
    const size_t total = up(dft.size()) + up(melf.size()) + up(t1.size()) + up(t3.size()) + up(tm.size());
    HIP_TRY(hipMalloc(&base, total * sizeof(float)));
    size_t off = 0;
    auto put = [&](const std::vector<float>& v, float** d) -> hipError_t {
        *d = base + off;
        off += up(v.size());
        return hipMemcpyAsync(*d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice, st);
    };
    HIP_TRY(put(dft, &g_mel.d_dft));
    HIP_TRY(put(melf, &g_mel.d_mel));
    HIP_TRY(put(t1, &g_mel.d_t1));
    HIP_TRY(put(t3, &g_mel.d_t3));
    HIP_TRY(put(tm, &g_mel.d_tm));
    HIP_TRY(hipStreamSynchronize(st));
    g_mel_by_device[dev] = g_mel;
    *out = g_mel;
    return SAVAD_OK;
}

// samples [first, first + count) of the signal that frames [frame_first, frame_first + frame_count) read, reflections
// at the signal's ends included; first is rounded down to a multiple of 4 (16-byte alignment of the direct reads)
void span_samples(long n, int frame_first, int frame_count, long* first, long* count) {
    const long lo = (long)mel::HOP * frame_first - 208, hi = (long)mel::HOP * (frame_first + frame_count - 1) + 208;  // [lo, hi)
    long a = lo < 0 ? 0 : lo, b = hi > n ? n : hi;
    if (lo < 0 && -lo + 1 > b) b = -lo + 1;
    if (hi > n && 2 * (n - 1) - (hi - 1) < a) a = 2 * (n - 1) - (hi - 1);
    if (a < 0) a = 0;
    if (b > n) b = n;
    a &= ~3L;
    *first = a;
    *count = b - a;
}

}  // namespace

SAVAD_EXPORT int savad_logmel_frames(int n_samples) { return n_samples < 0 ? fail(SAVAD_E_INVALID, "n_samples") : 1 + n_samples / mel::HOP; }
SAVAD_EXPORT size_t savad_logmel_workspace_bytes(int n_samples) { return ((size_t)n_samples + mel::N_FFT + 64) * sizeof(float); }
SAVAD_EXPORT size_t savad_logmel_span_workspace_bytes(int frame_count) {
    return ((size_t)mel::HOP * (frame_count > 0 ? frame_count : 0) + mel::N_FFT + 64 + 2048) * sizeof(float);
}
SAVAD_EXPORT int savad_logmel_set_algorithm(int algorithm) {
    if (algorithm < 0 || algorithm > 1) return fail(SAVAD_E_INVALID, "log-mel algorithm %d (0 = factored DFT, 1 = DFT as one GEMM)", algorithm);
    g_logmel_algorithm = algorithm;
    return SAVAD_OK;
}
SAVAD_EXPORT int savad_logmel_table_floats(int which) {
    return which == 0 ? mel::FFT_T1_FLOATS : which == 1 ? mel::FFT_T3_FLOATS : which == 2 ? mel::FFT_TM_FLOATS : fail(SAVAD_E_INVALID, "table %d", which);
}
SAVAD_EXPORT int savad_logmel_tables_host(float* t1, float* t3, float* tm) {
    if (!t1 || !t3 || !tm) return fail(SAVAD_E_INVALID, "null argument");
    std::vector<float> a, b, c;
    if (!build_fft_tables(a, b, c)) return fail(SAVAD_E_STATE, "log-mel tables: a filter weight falls outside the kernel's pair -> mel block pattern");
    memcpy(t1, a.data(), a.size() * sizeof(float));
    memcpy(t3, b.data(), b.size() * sizeof(float));
    memcpy(tm, c.data(), c.size() * sizeof(float));
    return SAVAD_OK;
}
// 16-bit PCM -> float32 in [-1, 1): sample / 32768 (exact), what soundfile hands the reference for a PCM16 file
// (vad/data_models/audio_data.py:21-24,32).  The source format of an upload from the host: half the bytes of the float signal.
__global__ void pcm16_to_f32_kernel(const short* __restrict__ pcm, long n, float* __restrict__ out) {
    const long n4 = ((uintptr_t)pcm & 7) == 0 ? n / 4 : 0;   // 8-byte pieces when the slice starts on one
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const short4 v = reinterpret_cast<const short4*>(pcm)[i];
        st4(out + 4 * i, f32x4{v.x * (1.0f / 32768.0f), v.y * (1.0f / 32768.0f), v.z * (1.0f / 32768.0f), v.w * (1.0f / 32768.0f)});
    }
    for (long i = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = pcm[i] * (1.0f / 32768.0f);
}
SAVAD_EXPORT int savad_pcm16_to_f32(const short* pcm, long n_samples, float* audio, void* stream) {
    if (n_samples == 0) return SAVAD_OK;
    if (!pcm || !audio || n_samples < 0) return fail(SAVAD_E_INVALID, "bad argument");
    if (((uintptr_t)pcm & 1) || ((uintptr_t)audio & 15)) return fail(SAVAD_E_INVALID, "pcm must be 2-byte, audio 16-byte aligned");
    const long work = (n_samples + 3) / 4;
    const int grid = (int)((work + 255) / 256 < 4096 ? (work + 255) / 256 : 4096);
    hipLaunchKernelGGL(pcm16_to_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pcm, n_samples, audio);
    HIP_TRY(hipGetLastError());
    return SAVAD_OK;
}

SAVAD_EXPORT int savad_logmel_span_samples(long n_samples, int frame_first, int frame_count, long* first, long* count) {
    if (!first || !count || n_samples < 1 || frame_first < 0 || frame_count < 1 || frame_first + (long)frame_count > 1 + n_samples / mel::HOP)
        return fail(SAVAD_E_INVALID, "bad frame span [%d, +%d) of %ld samples", frame_first, frame_count, n_samples);
    span_samples(n_samples, frame_first, frame_count, first, count);
    return SAVAD_OK;
}

SAVAD_EXPORT int savad_logmel_span(const float* audio, long audio_first, long audio_count, long n_samples, int frame_first,
                                   int frame_count, float* workspace, float* features, void* stream) {
    if (!audio || !workspace || !features || n_samples < 1 || n_samples > 2000000000L || audio_first < 0 || audio_count < 1 ||
        audio_first + audio_count > n_samples)
        return fail(SAVAD_E_INVALID, "bad argument");
    if (frame_first < 0 || frame_count < 1 || frame_first + (long)frame_count > 1 + n_samples / mel::HOP)
        return fail(SAVAD_E_INVALID, "frames [%d, +%d) outside the %ld frames of %ld samples", frame_first, frame_count, 1 + n_samples / mel::HOP, n_samples);
    if (((uintptr_t)workspace | (uintptr_t)features) & 15) return fail(SAVAD_E_INVALID, "workspace and features must be 16-byte aligned");
    long need_first, need_count;
    span_samples(n_samples, frame_first, frame_count, &need_first, &need_count);
    if (audio_first > need_first + 3 || audio_first + audio_count < need_first + need_count)  // (+3: need_first was rounded down)
        return fail(SAVAD_E_INVALID, "frames [%d, +%d) read samples [%ld, +%ld); the audio slice holds [%ld, +%ld)", frame_first, frame_count,
                    need_first, need_count, audio_first, audio_count);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    MelTables g_mel;
    if ((rc = ensure_mel_tables(st, &g_mel))) return rc;
    const float* y0 = audio - audio_first;  // y0[i] = sample i (only indices inside the slice are ever read)
    // the frames read padded indices [j_first, j_last + 4), in 16-byte chunks (savad_logmel.h: sample stage)
    const long j_first = (long)mel::HOP * frame_first + 48, j_last = (long)mel::HOP * (frame_first + frame_count - 1) + 460;
    constexpr long NEVER = 1L << 40;
    mel::FftSrc src{};
    mel::PadSeg A{workspace, 0, 0}, B{workspace + 256, 0, 0};
    src.y0 = y0;
    src.jA_end = -NEVER;
    src.jB0 = NEVER;
    const bool direct = (((uintptr_t)audio - (uintptr_t)audio_first * 4u) & 15) == 0;
    if (direct) {
        if (j_first < mel::N_FFT / 2) {  // the stretch that mirrors the head of the signal
            A.j0 = j_first;
            A.count = (int)((j_last + 4 < mel::N_FFT / 2 ? j_last + 4 : mel::N_FFT / 2) - j_first);
            src.jA_end = mel::N_FFT / 2;
        }
        const long jb = (n_samples + 253 + 3) & ~3L;  // first chunk that runs past the last sample
        if (j_last >= jb) {
            B.j0 = jb;
            B.count = (int)(j_last + 4 - jb);  // <= 212
            src.jB0 = jb;
        }
    } else {  // unaligned audio: the whole span from a padded (and thereby aligned) copy
        A.j0 = j_first;
        A.count = (int)(j_last + 4 - j_first);
        src.jA_end = NEVER;
    }
    src.padA = A.dst;
    src.jA0 = A.j0;
    src.padB = B.dst;
    if (A.count + B.count > 0) {
        const long total = (long)A.count + B.count;
        const int g1 = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(mel::reflect_pad_segments_kernel, dim3(g1), dim3(256), 0, st, y0, n_samples, A, B);
    }
    const int tiles = (frame_count + 31) / 32;
    const int grid = tiles < g_mel.n_cu ? tiles : g_mel.n_cu;
    hipLaunchKernelGGL(mel::logmel_fft_kernel, dim3(grid), dim3(256), mel::FFT_LDS_BYTES, st, src, frame_first, frame_count, tiles, g_mel.d_t1,
                       g_mel.d_t3, g_mel.d_tm, features);
    HIP_TRY(hipGetLastError());
    return SAVAD_OK;
}

SAVAD_EXPORT int savad_logmel(const float* audio, int n_samples, float* workspace, float* features, void* stream) {
    if (!audio || !workspace || !features || n_samples < 1) return fail(SAVAD_E_INVALID, "bad argument");
    if (g_logmel_algorithm == 0)
        return savad_logmel_span(audio, 0, n_samples, n_samples, 0, 1 + n_samples / mel::HOP, workspace, features, stream);
    if (((uintptr_t)workspace | (uintptr_t)features) & 15) return fail(SAVAD_E_INVALID, "workspace and features must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    MelTables g_mel;
    if ((rc = ensure_mel_tables(st, &g_mel))) return rc;
    const int n_frames = 1 + n_samples / mel::HOP;
    const long total = (long)n_samples + mel::N_FFT;
    const int g1 = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(mel::reflect_pad_kernel, dim3(g1), dim3(256), 0, st, audio, n_samples, workspace);
    const int tiles = (n_frames + 31) / 32;
    hipLaunchKernelGGL(mel::logmel_kernel<4>, dim3(tiles), dim3(256), 0, st, workspace, n_frames, g_mel.d_dft, g_mel.d_mel,
                       features);
    HIP_TRY(hipGetLastError());
    return SAVAD_OK;
}

// ---- post-processing (host arrays; savad_post.h) ------------------------------------------------
SAVAD_EXPORT int savad_trim_voice_activity(const uint8_t* pred, int n, int min_vally, int min_hill, int hang_before,
                                           int hang_over, uint8_t* out) {
    if (n < 0 || (n > 0 && (!pred || !out))) return fail(SAVAD_E_INVALID, "bad argument");
    savad::post::trim_voice_activity(pred, n, min_vally, min_hill, hang_before, hang_over, out);
    return SAVAD_OK;
}
SAVAD_EXPORT long savad_frames_to_samples(const double* frames, int n, int sample_rate, double hop_ms, double window_ms,
                                          double* out) {
    if (n < 0 || (n > 0 && !frames)) return fail(SAVAD_E_INVALID, "bad argument");
    return savad::post::frames_to_samples(frames, n, sample_rate, hop_ms, window_ms, out);
}
SAVAD_EXPORT int savad_samples_to_segments(const double* samples, long n, long* starts, long* ends, int cap) {
    if (n < 0 || (n > 0 && !samples) || cap < 0) return fail(SAVAD_E_INVALID, "bad argument");
    return savad::post::samples_to_segments(samples, n, starts, ends, cap);
}
SAVAD_EXPORT int savad_optimal_split(const double* pred, const double* probs, long n, long max_samples, double* out) {
    if (n < 0 || (n > 0 && (!pred || !probs || !out)) || max_samples <= 1) return fail(SAVAD_E_INVALID, "bad argument");
    savad::post::optimal_split(pred, probs, n, max_samples, out);
    return SAVAD_OK;
}

#ifdef SAVAD_TIMING
// experiments only: read the phase stamps of the last row_kernel_m launch
SAVAD_EXPORT int savad_debug_wg_stamps(long long* out, int n) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(savad::g_savad_wg), sizeof(long long) * n));
    return SAVAD_OK;
}
SAVAD_EXPORT int savad_debug_stamps(long long* out, int n) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(savad::g_savad_dbg), sizeof(long long) * n));
    return SAVAD_OK;
}
#endif
