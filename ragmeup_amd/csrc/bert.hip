// bert.hip -- BERT-6x384 encoder forward (bi-encoder pooling / cross-encoder head) for gfx950 (MI355X only).
//
// Replaces the transformer forwards the reference reaches through
//   HuggingFaceEmbeddings.embed_documents / embed_query   (server/RAGHelper_local.py:107-117, called from the
//       indexing loop server/RAGHelper.py:423-434 and every retriever query, :497-499)          SURVEY 8(a2,a3)
//   HuggingFaceCrossEncoder.score                          (server/RAGHelper.py:483-486 ->
//       server/ScoredCrossEncoderReranker.py:42)                                                 SURVEY 8(a7)
// i.e. transformers' BertModel / BertForSequenceClassification (6 layers, hidden 384, 12 heads x 32, FFN 1536,
// GELU(erf), LayerNorm eps 1e-12) + sentence-transformers pooling (masked mean, L2 normalise) or the
// pooler-tanh + Linear(384,1) head.
//
// Layout: tokens are PACKED (no padding rows): sequence b owns rows [cu[b], cu[b+1]) of every [T, *] activation.
// Activations are bf16 in HBM, every accumulation / LayerNorm / softmax is fp32.  Per layer, above 16 384 tokens:
//   k_gemm3   QKV projection: persistent 256 x 192 tiles, v_mfma_f32_32x32x16_bf16, 5-slot LDS-DMA ring, head-major stores
//   k_attn3   attention per (sequence, head): two-pass softmax off the MFMA accumulator; writes ctx as 1-KiB operand blocks
//   k_gemm    out-proj + residual (v_mfma_f32_16x16x32_bf16, both operands tiled)
//   k_ffn3    LayerNorm 1 + FFN1 + GELU + FFN2 + residual + LayerNorm 2, two waves per SIMD; writes h tiled between layers
// and k_gemm_small / the GEMM pair below that; k_embed_ln in front, k_pool / k_cls_head / k_tokens_out behind.  LDS staging is
// LDS-DMA (global_load_lds_dwordx4) with the XOR swizzle applied to the per-lane SOURCE address (the DMA writes lane-linear) or
// baked into a tiled copy of the operand.  In the GEMMs the operands are swapped (D^T = W . A^T) so that a lane ends up with
// consecutive output features of one token.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "rmu_common.h"
#include "../../include/rmu.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __bf16 bf16;

namespace {

constexpr int H = 384, NH = 12, DH = 32, FF = 1536;

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }

// Swizzle of the SHARED 64-byte-row operand images (k_tile_w's weight copies, the tiled ctx / h activations, the ring slots of k_gemm
// at 32-k stages and of k_gemm3): logical 16-byte unit u of row r sits at physical unit u ^ tswz(r), tswz = perm[(r >> 2) & 3] with
// perm = {0, 2, 3, 1}.  The hardware services a ds_read_b128 in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (not 16
// consecutive lanes): with the identity on those two row bits the 16x16x32 fragment mapping (row = lane & 15, k group = lane >> 4) put
// rows 0-3 and rows 4-7 of such a group on the same four slots -- HALF of the out-proj GEMM's LDS cycles were bank conflicts
// (profiles/r03_encoder_lds_pmc.md).  This permutation is conflict-free for that mapping and for the 32x32x16 one (row = lane & 31).
__host__ __device__ __forceinline__ int tswz(int row) { return (0x78 >> ((row >> 1) & 6)) & 3; }

// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, two orders below the bf16 resolution of the output):
// one rcp + one exp + 6 fma instead of libm's ~40-instruction erff
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(r, x);
}

// GELU(erf) for the fused FFN kernel, where one wave per SIMD has to fit the activation into the MFMA shadow: no
// reciprocal, no exponential.  erf(t) ~= clamp(tc * P(tc^2), -1, 1), tc = clamp(t, -3, 3), P of degree 7 (minimax fit on
// [0, 3]): |error| <= 1e-4 in erf, i.e. <= 5e-5 |v| in GELU -- an order of magnitude below the bf16 rounding of the result.
__device__ __forceinline__ float gelu_poly(float v) {
    const float t = v * 0.70710678118654752f;
    const float tc = __builtin_amdgcn_fmed3f(t, -3.0f, 3.0f);
    const float u = tc * tc;
    float p = -4.055360137e-07f;
    p = fmaf(p, u, 1.715983126e-05f);
    p = fmaf(p, u, -3.145953815e-04f);
    p = fmaf(p, u, 3.318710718e-03f);
    p = fmaf(p, u, -2.268579789e-02f);
    p = fmaf(p, u, 1.077178270e-01f);
    p = fmaf(p, u, -3.732314110e-01f);
    p = fmaf(p, u, 1.127895713e+00f);
    const float e = __builtin_amdgcn_fmed3f(tc * p, -1.0f, 1.0f);
    const float hv = 0.5f * v;
    return fmaf(hv, e, hv);
}

// the same on four values at once, written with vector operations so that every polynomial step is emitted for all four
// elements before the next step (four independent dependency chains in flight instead of one).  hipcc turns this into
// v_pk_fma_f32 / v_pk_mul_f32; the same polynomial as 28 scalar inline-asm v_fma_f32 measured 5 % slower in k_ffn_fused.
__device__ __forceinline__ f32x4 gelu_poly4(f32x4 v) {
    auto sp = [](float c) { return f32x4{c, c, c, c}; };
    const f32x4 t = v * 0.70710678118654752f;
    f32x4 tc;
#pragma unroll
    for (int e = 0; e < 4; ++e) tc[e] = __builtin_amdgcn_fmed3f(t[e], -3.0f, 3.0f);
    const f32x4 u = tc * tc;
    f32x4 p = __builtin_elementwise_fma(sp(-4.055360137e-07f), u, sp(1.715983126e-05f));
    p = __builtin_elementwise_fma(p, u, sp(-3.145953815e-04f));
    p = __builtin_elementwise_fma(p, u, sp(3.318710718e-03f));
    p = __builtin_elementwise_fma(p, u, sp(-2.268579789e-02f));
    p = __builtin_elementwise_fma(p, u, sp(1.077178270e-01f));
    p = __builtin_elementwise_fma(p, u, sp(-3.732314110e-01f));
    p = __builtin_elementwise_fma(p, u, sp(1.127895713e+00f));
    const f32x4 ep = tc * p;
    f32x4 e4;
#pragma unroll
    for (int e = 0; e < 4; ++e) e4[e] = __builtin_amdgcn_fmed3f(ep[e], -1.0f, 1.0f);
    const f32x4 hv = v * 0.5f;
    return __builtin_elementwise_fma(hv, e4, hv);
}

// ------------------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------------------
__global__ void k_f32_to_bf16(const float* __restrict__ src, bf16* __restrict__ dst, int64_t n, float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (bf16)(src[i] * scale);
}
__global__ void k_scale_copy(const float* __restrict__ src, float* __restrict__ dst, int64_t n, float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] * scale;
}

// W2 copy for k_ffn_fused: inside every block of 16 columns, slot 8 h + e holds column (e < 4 ? 4 h + e : 8 + 4 h + e - 4)
// -- the order in which a lane half h of GEMM1's 32x32 accumulator layout holds 8 of the block's 16 features (see k_ffn_fused)
__global__ void k_permute_w2(const bf16* __restrict__ src, bf16* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t blk = i & ~(int64_t)15;
    const int s16 = (int)(i & 15), h = s16 >> 3, e = s16 & 7;
    dst[i] = src[blk + (e < 4 ? 4 * h + e : 8 + 4 * h + (e - 4))];
}

// Weight streams for k_ffn3: the very slab images its LDS slots hold, in stream order, so that a DMA wave instruction reads 1 KiB of
// contiguous bytes and the 2-3 instructions a wave issues per slab differ only by the instruction's immediate offset.
//   W1 stream: [chunk c][slab i] 16 KiB = [64 rows][16 units]; physical unit pu of row r holds logical lu = pu ^ (r & 15);
//              lu < 8: k [64 i + 8 lu, +8), lu >= 8: k [192 + 64 i + 8 (lu - 8), +8) of W1 row 64 c + r
//   W2 stream: [chunk c][tile t] 24 KiB = [384 rows][4 units]; pu of row r holds lu = pu ^ ((r >> 2) & 3): k [64 c + 32 t + 8 lu, +8) of
//              the k_permute_w2 copy's row r
#ifdef RMU_DEBUG_KERNELS
__global__ void k_pack_ffn3(const bf16* __restrict__ w1, const bf16* __restrict__ w2p, bf16* __restrict__ w1s, bf16* __restrict__ w2s) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n1 = (int64_t)FF * H / 8;
    if (d < n1) {
        const int slab = (int)(d >> 10), within = (int)(d & 1023), c = slab / 3, i = slab % 3;
        const int r = within >> 4, pu = within & 15, lu = pu ^ (r & 15);
        const int k0 = lu < 8 ? 64 * i + 8 * lu : 192 + 64 * i + 8 * (lu - 8);
        *(bf16x8*)(w1s + d * 8) = *(const bf16x8*)(w1 + (int64_t)(64 * c + r) * H + k0);
    } else if (d < 2 * n1) {
        const int64_t e = d - n1;
        const int slab = (int)(e / 1536), within = (int)(e % 1536), c = slab >> 1, t = slab & 1;
        const int r = within >> 2, pu = within & 3, lu = pu ^ ((r >> 2) & 3);
        *(bf16x8*)(w2s + e * 8) = *(const bf16x8*)(w2p + (int64_t)r * FF + 64 * c + 32 * t + 8 * lu);
    }
}

#endif

// Weight copy for k_gemm3's LDS-DMA: block (nb, kb) = rows [16 nb, +16) x k [32 kb, +32) stored as the very 1 KiB its ring
// slot holds (row r of the block at byte 64 r, logical 16-byte unit u at physical unit u ^ tswz(r)), blocks in [nb][kb]
// order: one DMA wave instruction then reads 1 KiB of CONTIGUOUS bytes (8 cache lines) instead of 16 rows x 64 B (16 lines) --
// the address path of a CU is paid per line touched.  One thread per 16-byte unit of the destination.
__global__ void k_tile_w(const bf16* __restrict__ src, bf16* __restrict__ dst, int N, int K) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= (int64_t)N * K / 8) return;
    const int64_t blk = d >> 6;
    const int within = (int)(d & 63), r = within >> 2, pu = within & 3;
    const int u = pu ^ tswz(r);
    const int kbn = K / 32;
    const int64_t nb = blk / kbn;
    const int kb = (int)(blk % kbn);
    *(bf16x8*)(dst + d * 8) = *(const bf16x8*)(src + (nb * 16 + r) * K + kb * 32 + u * 8);
}

// cu[0] = 0, cu[b+1] = cu[b] + clamp(lens[b], 0, max_len); one block
__global__ void k_cu_seqlens(const int* __restrict__ lens, int batch, int max_len, int* __restrict__ cu) {
    __shared__ int part[1024];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int per = (batch + nt - 1) / nt;
    const int lo = tid * per, hi = min(batch, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += min(max(lens[i], 0), max_len);
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < nt; ++i) { const int v = part[i]; part[i] = run; run += v; }
        cu[0] = 0;
    }
    __syncthreads();
    int run = part[tid];
    for (int i = lo; i < hi; ++i) { run += min(max(lens[i], 0), max_len); cu[i + 1] = run; }
}

// wave-wide LayerNorm of 384 values held 8 per lane by lanes 0..47 (16-byte row pieces: a CU's load/store path is paid per
// wave instruction, ~70 cycles per store whatever its width -- with 4-byte pieces k_layernorm was bound by store issue,
// 0.39 ms per call for 1.6 GB of traffic); lanes 48..63 hold zeros and are masked out of the variance
__device__ __forceinline__ void ln_row(float (&v)[8], bool act, const float* __restrict__ g, const float* __restrict__ b, int c0,
                                       float eps) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mu = s * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float d = act ? v[i] - mu : 0.f; q = fmaf(d, d, q); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rs = rsqrtf(q * (1.0f / H) + eps);
    if (act) {
        const f32x4 g0 = *(const f32x4*)(g + c0), g1 = *(const f32x4*)(g + c0 + 4);
        const f32x4 b0 = *(const f32x4*)(b + c0), b1 = *(const f32x4*)(b + c0 + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = (v[i] - mu) * rs * g0[i] + b0[i];
            v[4 + i] = (v[4 + i] - mu) * rs * g1[i] + b1[i];
        }
    }
}

// embeddings + LayerNorm: one wave per (sequence, position); packed output row cu[b] + pos.  (Four slots per wave, as in
// k_layernorm below, measured slower here: half the slots are padding and exit at once.)
// CU_HERE (round 5, the interactive path: batch <= 256): `cu` is an OUTPUT -- every workgroup rebuilds the sequence offsets from `lens` in
// LDS (a few adds) and workgroup 0 writes them for the launches behind it; the k_cu_seqlens launch (one of ~34 dependent ~4.4-us launches
// of a query forward) is gone.
template <bool CU_HERE>
__global__ __launch_bounds__(256) void k_embed_ln(const int* __restrict__ ids, const int* __restrict__ type_ids,
                                                  const int* __restrict__ cu, int batch, int max_len,
                                                  const float* __restrict__ wemb, const float* __restrict__ pemb,
                                                  const float* __restrict__ temb, const float* __restrict__ g,
                                                  const float* __restrict__ bta, float eps, int vocab, int type_vocab,
                                                  bf16* __restrict__ out, const int* __restrict__ lens = nullptr, int* __restrict__ cu_out = nullptr) {
    __shared__ int s_len[CU_HERE ? 256 : 1], s_cu[CU_HERE ? 257 : 1];
    if constexpr (CU_HERE) {
        const int tid = threadIdx.x;
        if (tid < batch) s_len[tid] = min(max(lens[tid], 0), max_len);
        __syncthreads();
        for (int j = tid; j <= batch; j += 256) {      // (batch <= 256: entry `batch` is the second turn of thread 0)
            int run = 0;
            for (int i = 0; i < j; ++i) run += s_len[i];
            s_cu[j] = run;
            if (blockIdx.x == 0) cu_out[j] = run;
        }
        __syncthreads();
        cu = s_cu;
    }
    const int lane = threadIdx.x & 63;
    const int64_t slot = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= (int64_t)batch * max_len) return;
    const int b = (int)(slot / max_len), pos = (int)(slot % max_len);
    const int len = cu[b + 1] - cu[b];
    if (pos >= len) return;
    int id = ids[slot];
    id = min(max(id, 0), vocab - 1);
    int tt = type_ids ? type_ids[slot] : 0;
    tt = min(max(tt, 0), type_vocab - 1);
    const bool act = lane < 48;
    const int c0 = lane * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (act) {
        const float* we = wemb + (int64_t)id * H + c0;
        const float* pe = pemb + (int64_t)pos * H + c0;
        const float* te = temb + (int64_t)tt * H + c0;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const f32x4 a = *(const f32x4*)(we + 4 * hf), p = *(const f32x4*)(pe + 4 * hf), t = *(const f32x4*)(te + 4 * hf);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * hf + i] = a[i] + p[i] + t[i];
        }
    }
    ln_row(v, act, g, bta, c0, eps);
    if (act) {
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (bf16)v[i];
        *(bf16x8*)(out + ((int64_t)cu[b] + pos) * H + c0) = o;
    }
}

// The same for BULK batches (round 5): k_embed_ln gives every (sequence, position) SLOT a wave -- half of them padding that exits after one
// round trip -- and a live wave walks a chain of three dependent round trips (offsets -> ids -> embedding rows) for ONE token: latency x
// occupancy, 0.65 ms per 1.04 M tokens.  Here a workgroup takes a SEQUENCE: its waves walk the positions four at a time (ids by scalar loads,
// the twelve row pieces of four tokens in flight together, the four LayerNorms' butterflies interleaved as in k_layernorm); no wave is ever
// started for padding.  Same arithmetic per row as k_embed_ln (a + p + t, two-pass statistics over the same butterfly): identical output.
__global__ __launch_bounds__(256) void k_embed_ln_seq(const int* __restrict__ ids, const int* __restrict__ type_ids,
                                                      const int* __restrict__ cu, int batch, int max_len,
                                                      const float* __restrict__ wemb, const float* __restrict__ pemb,
                                                      const float* __restrict__ temb, const float* __restrict__ g,
                                                      const float* __restrict__ bta, float eps, int vocab, int type_vocab,
                                                      bf16* __restrict__ out) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t0 = cu[b], L = cu[b + 1] - t0;
    if (L <= 0) return;
    const bool act = lane < 48;
    const int c0 = (act ? lane : 0) * 8;
    f32x4 g0 = {}, g1 = {}, b0 = {}, b1 = {};
    if (act) { g0 = *(const f32x4*)(g + c0); g1 = *(const f32x4*)(g + c0 + 4); b0 = *(const f32x4*)(bta + c0); b1 = *(const f32x4*)(bta + c0 + 4); }
    const int* idr = ids + (int64_t)b * max_len;
    const int* ttr = type_ids ? type_ids + (int64_t)b * max_len : nullptr;
    for (int p0 = w * 4; p0 < L; p0 += 16) {
        int id[4], tt[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pos = min(p0 + u, L - 1);            // (rows past the sequence are computed on the last token and not stored)
            id[u] = min(max(idr[pos], 0), vocab - 1);
            tt[u] = ttr ? min(max(ttr[pos], 0), type_vocab - 1) : 0;
        }
        float v[4][8], sm[4], q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pos = min(p0 + u, L - 1);
            const float* we = wemb + (int64_t)id[u] * H + c0;
            const float* pe = pemb + (int64_t)pos * H + c0;
            const float* te = temb + (int64_t)tt[u] * H + c0;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f32x4 a = {}, p = {}, t = {};
                if (act) { a = *(const f32x4*)(we + 4 * hf); p = *(const f32x4*)(pe + 4 * hf); t = *(const f32x4*)(te + 4 * hf); }
#pragma unroll
                for (int i = 0; i < 4; ++i) v[u][4 * hf + i] = a[i] + p[i] + t[i];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sm[u] = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) sm[u] += v[u][i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < 4; ++u) sm[u] += __shfl_xor(sm[u], o);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sm[u] *= (1.0f / H);
            q[u] = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = act ? v[u][i] - sm[u] : 0.f; q[u] = fmaf(d, d, q[u]); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] += __shfl_xor(q[u], o);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float rs = rsqrtf(q[u] * (1.0f / H) + eps);
            if (act && p0 + u < L) {
                bf16x8 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o[i] = (bf16)((v[u][i] - sm[u]) * rs * g0[i] + b0[i]);
                    o[4 + i] = (bf16)((v[u][4 + i] - sm[u]) * rs * g1[i] + b1[i]);
                }
                __builtin_nontemporal_store(o, (bf16x8*)(out + ((int64_t)t0 + p0 + u) * H + c0));
            }
        }
    }
}

// out = LayerNorm(y) (y already holds GEMM + bias + residual); a wave takes LN_ROWS consecutive tokens with all their loads in
// flight before the first reduction (one row per wave left the kernel latency-bound at ~4 TB/s of its 1.6 GB)
constexpr int LN_ROWS = 4;
__global__ __launch_bounds__(256) void k_layernorm(const bf16* __restrict__ y, const int* __restrict__ cu, int batch,
                                                   const float* __restrict__ g, const float* __restrict__ bta, float eps,
                                                   bf16* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t t0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_ROWS;
    const int64_t M = cu[batch];
    if (t0 >= M) return;
    const bool act = lane < 48;
    const int c0 = lane * 8;
    bf16x8 r[LN_ROWS];
#pragma unroll
    for (int u = 0; u < LN_ROWS; ++u) {
        r[u] = bf16x8{};
        if (act && t0 + u < M) r[u] = *(const bf16x8*)(y + (t0 + u) * H + c0);
    }
    f32x4 g0 = {}, g1 = {}, b0 = {}, b1 = {};
    if (act) { g0 = *(const f32x4*)(g + c0); g1 = *(const f32x4*)(g + c0 + 4); b0 = *(const f32x4*)(bta + c0); b1 = *(const f32x4*)(bta + c0 + 4); }
    float v[LN_ROWS][8], s[LN_ROWS], q[LN_ROWS];
#pragma unroll
    for (int u = 0; u < LN_ROWS; ++u) {
        s[u] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[u][i] = bf2f(r[u][i]); s[u] += v[u][i]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int u = 0; u < LN_ROWS; ++u) s[u] += __shfl_xor(s[u], o);
#pragma unroll
    for (int u = 0; u < LN_ROWS; ++u) {
        s[u] *= (1.0f / H);
        q[u] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = act ? v[u][i] - s[u] : 0.f; q[u] = fmaf(d, d, q[u]); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int u = 0; u < LN_ROWS; ++u) q[u] += __shfl_xor(q[u], o);
#pragma unroll
    for (int u = 0; u < LN_ROWS; ++u) {
        const float rs = rsqrtf(q[u] * (1.0f / H) + eps);
        if (act && t0 + u < M) {
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i] = (bf16)((v[u][i] - s[u]) * rs * g0[i] + b0[i]);
                o[4 + i] = (bf16)((v[u][4 + i] - s[u]) * rs * g1[i] + b1[i]);
            }
            *(bf16x8*)(out + (t0 + u) * H + c0) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// GEMM: out[m, n] = epi( sum_k A[m, k] * W[n, k] + bias[n] )     A [M, K] bf16, W [N, K] bf16 (HF Linear layout)
// ------------------------------------------------------------------------------------------------------------
constexpr int BN = 128;
enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RESID = 2 };

extern __shared__ __attribute__((aligned(16))) char gsm[];

// Tile = (64*WM tokens) x 128 features, 2*WM waves (each 64 x 64), BK = K-slab per stage, ST = ring stages.
// The kernel is bound by the L2->LDS operand fill rate (measured ~6-7 TB/s chip-wide), not by MFMA issue:
// flop per staged byte is what matters (128x128: 64, 256x128: 87), and more resident workgroups beat deeper rings.
// TA: A and W are TILED copies -- 1-KiB blocks of 16 rows x 32 k in exactly the swizzled image a ring slot holds, blocks in
// [row block][k block] order (W: k_tile_w; A: the attention kernel writes ctx that way) -- so a DMA wave instruction reads one
// contiguous KiB (8 cache lines) instead of 16 rows x 64 B (16 lines): the CU's in-order memory queue is paid per line touched.
template <int EPI, int WM, int BK, int ST, bool TA = false>
__global__ __launch_bounds__(128 * WM) void k_gemm(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                   const float* __restrict__ bias, const bf16* __restrict__ resid,
                                                   bf16* __restrict__ out, const int* __restrict__ cu, int batch, int N, int K, int dbg) {
    constexpr int BM = 64 * WM, NWV = 2 * WM, NTH = 64 * NWV;
    constexpr int UPR = BK / 8;                    // 16-B units per row of a slab
    constexpr int NITA = BM * UPR / NTH;           // DMA wave-instructions per wave per slab (tokens)
    constexpr int NITW = BN * UPR / NTH;           // ... (features)
    constexpr int SLABA = BM * BK * 2, SLABW = BN * BK * 2;
    constexpr int KS = BK / 32;                    // MFMA k-substeps per slab
    static_assert(NITA >= 1 && NITW >= 1, "tile too small for the thread count");
    const int M = cu[batch];                       // real token count (device side: no host sync)
    // block -> (token tile, feature tile): the N/BN feature tiles that re-read one token tile run on ONE XCD
    // (dispatch places block b on XCD b % 8) so the tile comes through one L2 instead of eight
    const int ntn = N / BN;
    const int xcd = blockIdx.x & 7, mloc = blockIdx.x >> 3;
    const int m0 = ((mloc / ntn) * 8 + xcd) * BM, n0 = (mloc % ntn) * BN;
    if (m0 >= M) return;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;             // wave -> 64 tokens x 64 features
    char* As = gsm;                                // [ST][BM rows][UPR units]   (tokens x k)
    char* Ws = gsm + ST * SLABA;                   // [ST][128 rows][UPR units]  (features x k)
    auto swz = [](int row) { return UPR == 8 ? (row & 7) : tswz(row); };

    // DMA source map: LDS unit f = (it*NWV + w)*64 + lane  ->  row f/UPR, physical unit f%UPR, logical unit p ^ swz(row)
    int arow[NITA], acol[NITA], wrow[NITW], wcol[NITW];
#pragma unroll
    for (int it = 0; it < NITA; ++it) {
        const int f = (it * NWV + w) * 64 + lane;
        const int row = f / UPR, p = f % UPR;
        acol[it] = (p ^ swz(row)) * 8;             // element offset inside the slab
        arow[it] = min(m0 + row, M - 1);           // clamp: rows >= M are never stored
    }
#pragma unroll
    for (int it = 0; it < NITW; ++it) {
        const int f = (it * NWV + w) * 64 + lane;
        const int row = f / UPR, p = f % UPR;
        wcol[it] = (p ^ swz(row)) * 8;
        wrow[it] = n0 + row;
    }
    const int nk = K / BK;
    static_assert(!TA || BK == 32, "tiled operands are 32-k blocks");
    auto stage = [&](int kt) {                     // kt may run past nk: harmless reloads keep vmcnt uniform
        const int buf = kt % ST;
        const int k0 = (kt < nk ? kt : nk - 1) * BK;
        if constexpr (TA) {
            const int kb = k0 / 32, nkb = K / 32;
#pragma unroll
            for (int it = 0; it < NITA; ++it)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)A + ((int64_t)((m0 >> 4) + it * NWV + w) * nkb + kb) * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(As + buf * SLABA + (it * NWV + w) * 1024), 16, 0, 0);
#pragma unroll
            for (int it = 0; it < NITW; ++it)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)W + ((int64_t)((n0 >> 4) + it * NWV + w) * nkb + kb) * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(Ws + buf * SLABW + (it * NWV + w) * 1024), 16, 0, 0);
            return;
        }
#pragma unroll
        for (int it = 0; it < NITA; ++it)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (int64_t)arow[it] * K + k0 + acol[it]),
                                             (__attribute__((address_space(3))) void*)(As + buf * SLABA + (it * NWV + w) * 1024), 16, 0, 0);
#pragma unroll
        for (int it = 0; it < NITW; ++it)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + (int64_t)wrow[it] * K + k0 + wcol[it]),
                                             (__attribute__((address_space(3))) void*)(Ws + buf * SLABW + (it * NWV + w) * 1024), 16, 0, 0);
    };

    f32x4 acc[4][4];                               // [feature tile][token tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int p = 0; p < ST - 1; ++p) stage(p);
    for (int kt = 0; kt < ((dbg & 2) ? 0 : nk); ++kt) {
        // slab kt landed (own DMA) once at most ST-2 younger groups are outstanding; own LDS reads returned
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NITA + NITW) * (ST - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        stage(kt + ST - 1);                         // refills the slot everyone finished reading last round
        const char* as = As + (kt % ST) * SLABA;
        const char* ws = Ws + (kt % ST) * SLABW;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 wf[4], af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wn * 64 + i * 16 + fr;
                wf[i] = *(const bf16x8*)(ws + (row * UPR + ((ks * 4 + kg) ^ swz(row))) * 16);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wm * 64 + j * 16 + fr;
                af[j] = *(const bf16x8*)(as + (row * UPR + ((ks * 4 + kg) ^ swz(row))) * 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
    }
    // (round 5) The epilogue's global reads -- the bias and, for EPI_RESID, the residual rows this lane will add -- are requested HERE, in
    // front of the drain of the ring's tail reloads: they used to be two more dependent round trips behind it (bias -> staging -> residual
    // -> store), exposed whenever the co-resident workgroup was in its own epilogue.
    f32x4 bvp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bvp[i] = *(const f32x4*)(bias + n0 + wn * 64 + i * 16 + (lane >> 4) * 4);
    bf16x8 rvp[8];
    if (EPI == EPI_RESID) {
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
            const int row = r8 * 8 + (lane >> 3), u = lane & 7;
            const int m = min(m0 + wm * 64 + row, M - 1);          // (rows >= M are never stored)
            int64_t roff = (int64_t)m * N + n0 + wn * 64 + u * 8;
            if (dbg & 256) {                       // the residual is a TILED activation (1-KiB blocks of 16 tokens x 32 features, see k_ffn3's store)
                const int col = n0 + wn * 64 + u * 8, r = m & 15;
                roff = ((int64_t)(m >> 4) * (N / 32) + (col >> 5)) * 512 + r * 32 + ((((col >> 3) & 3) ^ tswz(r)) * 8);
            }
            rvp[r8] = __builtin_nontemporal_load((const bf16x8*)(resid + roff));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the tail reloads (and the reads above): the ring is reused below
    __syncthreads();
    // epilogue.  A lane holds features n..n+3 (rows of D^T) of token m (column of D^T): written straight to HBM
    // that is 16 rows x 32 B per store instruction.  Instead each wave parks its 64 x 64 bf16 tile in LDS
    // (rows of 128 B + 16 B pad) and writes it back out as FULL 128-B lines, 8 lanes x 16 B per token row.
    // (Measured on FFN1: the 3.2 GB intermediate written as 32-B pieces cost more than the whole MFMA loop.)
    constexpr int TSTR = 144;                       // bytes per token row of the staging tile
    char* tile = gsm + w * (64 * TSTR);             // 9 KiB per wave, inside the (now idle) ring
    // (the launcher sizes dynamic LDS as max(ring, 2*WM*64*TSTR))
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int nl = i * 16 + kg * 4;         // feature inside the wave tile
            const f32x4 bv = bvp[i];
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + bv[e];
            if (EPI == EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + fast_erf(v[e] * 0.70710678118654752f));
            }
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
            *(bf16x4*)(tile + (j * 16 + fr) * TSTR + nl * 2) = o;
        }
    }
    // wave-private tile: no barrier needed, only the wave's own LDS writes must have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) {
        const int row = r8 * 8 + (lane >> 3), u = lane & 7;      // 8 token rows x 8 units of 16 B per instruction
        const int m = m0 + wm * 64 + row;
        if (m < M && !(dbg & 1)) {
            bf16x8 o = *(const bf16x8*)(tile + row * TSTR + u * 16);
            const int64_t off = (int64_t)m * N + n0 + wn * 64 + u * 8;
            if (EPI == EPI_RESID) {
                const bf16x8 rv = rvp[r8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16)(bf2f(o[e]) + bf2f(rv[e]));
            }
            __builtin_nontemporal_store(o, (bf16x8*)(out + off));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_gemm_small -- the same GEMM for a HANDFUL of tokens (one query through embed_query, a few short passages): with M <= 256
// the tiled kernels above run one or two workgroups for 12-48 pipeline stages each (11-17 us per launch, 0.47 ms per forward
// of a 16-token query), while the whole weight matrix is only 0.3-1.2 MB.  Here the FEATURES are spread over the chip: grid =
// N / 32 workgroups of 4 waves; wave w owns the K quarter [w K/4, (w+1) K/4) of the workgroup's 32 features -- its weight rows
// sit in registers as MFMA A fragments (K/64 of them), token fragments come straight from global memory (no LDS staging), the
// four partial sums meet in LDS and all 256 threads run the epilogue (bias | + GELU | + residual).
// ------------------------------------------------------------------------------------------------------------
constexpr int SMALL_M = 256;                           // tokens (upper bound batch * max_len) below which launch_gemm's callers take k_gemm_small
// ... and up to which QKV projection + attention are one launch (k_qkv_attn_small).  tools/small_ab.sh, one box, every output bit-identical:
// a 16-token query forward 0.182 ms fused / 0.183 apart (no gain: inside a replayed graph a kernel boundary is not the 4.4 us it costs under
// the profiler, a kernel's own load -> MFMA -> store chain is), the 14-pair rerank forward (1.5k tokens) 0.417 fused / 0.437 apart: the
// whole folded path takes it.
constexpr int QKV_ATTN_TOKENS = 2560, QKV_ATTN_MIN_TOKENS = 128;
constexpr int FOLD_TOKENS = 2560;                      // ... and up to which a whole forward runs on it with the LayerNorms folded (enqueue_forward)
// Round 4: the token dimension is a grid dimension too -- workgroup (x, y) takes features [32 x, +32) of tokens [SMALL_TB y, +SMALL_TB)
// -- so the same kernel serves a few THOUSAND tokens (the reference's rerank call: <= 14 (query, passage) pairs, ~1.5k tokens,
// twice per /chat request): the tiled kernels run 6-36 workgroups there (10-17 us per launch), this one N/32 x M/128 short ones.
// LayerNorm folded into its consumers (round 4; one query per call is ~45 launches of ~4.4 us each: the two k_layernorm launches of
// a layer are 12 of them).  A pre-LN sum y [M, 384] stays what it is in memory and
//   * LNA: a GEMM whose A operand is LN(y) normalises its token fragments on the way in -- the two-pass statistics of k_layernorm
//     (fp32 mean, then variance about it) over the four waves' K quarters through LDS, the same rounding point (bf16 of the
//     normalised value) -- and workgroup x = 0 leaves (mean, rstd) per token in `stats_out`;
//   * LNR: a GEMM whose RESIDUAL is LN(y) rebuilds it in the epilogue from y and those statistics (the launch that wrote them is
//     always an earlier one on the stream: FFN1 -> FFN2, next layer's QKV -> its out-proj).
// Token tiles are walked with the NEXT tile's fragments (and this tile's residual rows / statistics) already requested: a tile is
// 6-12 MFMAs per wave, a global round trip is a microsecond -- unpipelined, a workgroup that walks 8 tiles spends 8 round trips
// (round 4, first version: 8-22 us per launch at 1.5k tokens, no better than the tiled kernels).  K = 1536 is split over EIGHT waves
// (NWV): 12 weight + 2 x 12 token fragments per wave instead of 24 + 2 x 24.
template <int EPI, int K, bool LNA = false, bool LNR = false, int NWV = (K > 512 ? 8 : 4)>
__global__ __launch_bounds__(64 * NWV) void k_gemm_small(const bf16* __restrict__ A, const bf16* __restrict__ W, const float* __restrict__ bias,
                                                    const bf16* __restrict__ resid, bf16* __restrict__ out, const int* __restrict__ cu,
                                                    int batch, int N, int SMALL_TB /* tokens per workgroup along grid.y, a multiple of 32 */,
                                                    const float* __restrict__ lng = nullptr, const float* __restrict__ lnb = nullptr, float eps = 0.f,
                                                    float2* __restrict__ stats_out = nullptr, const float* __restrict__ rg = nullptr,
                                                    const float* __restrict__ rb = nullptr, const float2* __restrict__ rstats = nullptr) {
    static_assert(!LNA || (K == H && NWV == 4), "the normalised operand is a hidden-state row, its statistics meet over four K quarters");
    constexpr int KW = K / NWV, NF = KW / 16;
    __shared__ float red[NWV][32][36];                 // [K slice][token][feature (+4 pad)]
    __shared__ float lnp[2][4][32];                    // LNA: [sum | squared deviations][K quarter][token]
    const int M = cu[batch];
    const int tb0 = blockIdx.y * SMALL_TB;
    if (tb0 >= M) return;                              // (the grid is sized by the shape's upper bound batch * max_len)
    const int tb1 = min(M, tb0 + SMALL_TB);
    const int n0 = blockIdx.x * 32;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r31 = lane & 31, hh = lane >> 5;
    bf16x8 wf[NF];
    {
        const bf16* wr = W + (int64_t)(n0 + r31) * K + w * KW + hh * 8;
#pragma unroll
        for (int f = 0; f < NF; ++f) wf[f] = *(const bf16x8*)(wr + f * 16);
    }
    float lg[LNA ? NF : 1][8], lb[LNA ? NF : 1][8];    // LNA: gamma / beta of this lane's k positions (k = w KW + 16 f + 8 hh + e)
    if constexpr (LNA) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int k0 = w * KW + f * 16 + hh * 8;
            const f32x4 g0 = *(const f32x4*)(lng + k0), g1 = *(const f32x4*)(lng + k0 + 4);
            const f32x4 b0 = *(const f32x4*)(lnb + k0), b1 = *(const f32x4*)(lnb + k0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { lg[f][e] = g0[e]; lg[f][4 + e] = g1[e]; lb[f][e] = b0[e]; lb[f][4 + e] = b1[e]; }
        }
    }
    const bool epi_thread = threadIdx.x < 256;         // epilogue: token et, features ef .. ef + 3 (the first four waves)
    const int et = (threadIdx.x & 255) >> 3, ef = (threadIdx.x & 7) * 4;
    const f32x4 bv = *(const f32x4*)(bias + n0 + ef);
    f32x4 rgv = {}, rbv = {};
    if constexpr (LNR) { rgv = *(const f32x4*)(rg + n0 + ef); rbv = *(const f32x4*)(rb + n0 + ef); }
    auto load_x = [&](int t0, bf16x8 (&x)[NF]) {
        const int tok = min(t0 + r31, M - 1);
        const bf16* xr = A + (int64_t)tok * K + w * KW + hh * 8;
#pragma unroll
        for (int f = 0; f < NF; ++f) x[f] = *(const bf16x8*)(xr + f * 16);
    };
    bf16x8 xn[NF];
    load_x(tb0, xn);
    for (int t0 = tb0; t0 < tb1; t0 += 32) {
        bf16x8 xf[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) xf[f] = xn[f];
        if (t0 + 32 < tb1) load_x(t0 + 32, xn);        // the next tile's fragments: in flight across this tile's MFMAs and epilogue
        // this tile's residual rows and statistics: requested now, used behind the MFMAs
        const int m = t0 + et;
        const int64_t off = (int64_t)m * N + n0 + ef;
        bf16x4 rv = {};
        float2 st = {0.f, 1.f};
        if (EPI == EPI_RESID && epi_thread && m < M) {
            rv = *(const bf16x4*)(resid + off);
            if constexpr (LNR) st = rstats[m];
        }
        if constexpr (LNA) {
            float v[NF][8], sm = 0.f;
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[f][e] = bf2f(xf[f][e]); sm += v[f][e]; }
            sm += __shfl_xor(sm, 32);
            if (hh == 0) lnp[0][w][r31] = sm;
            __syncthreads();
            const float mu = ((lnp[0][0][r31] + lnp[0][1][r31]) + (lnp[0][2][r31] + lnp[0][3][r31])) * (1.0f / H);
            float qv = 0.f;
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[f][e] - mu; qv = fmaf(d, d, qv); }
            qv += __shfl_xor(qv, 32);
            if (hh == 0) lnp[1][w][r31] = qv;
            __syncthreads();
            const float rs = rsqrtf(((lnp[1][0][r31] + lnp[1][1][r31]) + (lnp[1][2][r31] + lnp[1][3][r31])) * (1.0f / H) + eps);
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int e = 0; e < 8; ++e) xf[f][e] = (bf16)((v[f][e] - mu) * rs * lg[f][e] + lb[f][e]);
            if (stats_out && blockIdx.x == 0 && w == 0 && hh == 0 && t0 + r31 < M) stats_out[t0 + r31] = float2{mu, rs};
        }
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int f = 0; f < NF; ++f) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[f], xf[f], acc, 0, 0, 0);
        // acc[4 q + e] = feature 8 q + 4 hh + e of token r31, summed over this wave's K slice
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *(f32x4*)&red[w][r31][q * 8 + hh * 4] = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        __syncthreads();
        if (epi_thread && m < M) {
            f32x4 v = bv;
#pragma unroll
            for (int ww = 0; ww < NWV; ++ww) v += *(const f32x4*)&red[ww][et][ef];
            if (EPI == EPI_GELU) v = gelu_poly4(v);
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
            if (EPI == EPI_RESID) {
                if constexpr (LNR) {                   // the residual is LN(resid row): (mean, rstd) from the launch that normalised it as ITS operand
#pragma unroll
                    for (int e = 0; e < 4; ++e) rv[e] = (bf16)((bf2f(rv[e]) - st.x) * st.y * rgv[e] + rbv[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)(bf2f(o[e]) + bf2f(rv[e]));
            }
            *(bf16x4*)(out + off) = o;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_gemm_mid -- the same GEMM (and the same LayerNorm folds) for a few HUNDRED to a few THOUSAND tokens: the reference's rerank call
// (<= 14 (query, passage) pairs, ~1.5k tokens, twice per /chat request).  k_gemm_small's K split costs four barriers and an LDS round
// trip per 32-token tile -- ~2 us a tile whatever is prefetched (8-20 us per launch at 1.5k tokens), and the tiled kernels run 6-36
// workgroups of 6-24 latency-bound stages there (11-15 us).  Here NOTHING is shared: a wave owns one 32-feature x 32-token output tile
// for the whole K -- both operands straight from global memory / L2 as MFMA fragments, in k chunks of CK with the next chunk's
// fragments requested before the current chunk's MFMAs (the weight rows are re-read per token tile: 56 MB of L2 reads for FFN2 at
// 1.5k tokens, microseconds) -- no LDS, no barrier, the LayerNorm statistics of a token are this lane's and lane ^ 32's.  grid =
// (N / 32) x ceil(tokens / 128); the four waves of a workgroup take four consecutive token tiles of one feature block.
// ------------------------------------------------------------------------------------------------------------
template <int EPI, int K, bool LNA = false, bool LNR = false>
__global__ __launch_bounds__(256) void k_gemm_mid(const bf16* __restrict__ A, const bf16* __restrict__ W, const float* __restrict__ bias,
                                                  const bf16* __restrict__ resid, bf16* __restrict__ out, const int* __restrict__ cu,
                                                  int batch, int N, const float* __restrict__ lng, const float* __restrict__ lnb, float eps,
                                                  float2* __restrict__ stats_out, const float* __restrict__ rg, const float* __restrict__ rb,
                                                  const float2* __restrict__ rstats) {
    static_assert(!LNA || K == H, "the normalised operand is a hidden-state row");
    constexpr int CK = LNA ? K : 192;                  // k per chunk (LNA: the whole row at once -- its statistics come first)
    constexpr int NC = K / CK, NF = CK / 16;
    const int M = cu[batch];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r31 = lane & 31, hh = lane >> 5;
    __shared__ float sgb[LNA ? 2 * H : 2];             // LNA: gamma | beta (read per fragment from LDS: hoisted global loads of all 192 pairs
    if constexpr (LNA) {                                // per lane cost 384 registers and spilled)
        for (int i = threadIdx.x; i < H; i += 256) { sgb[i] = lng[i]; sgb[H + i] = lnb[i]; }
        __syncthreads();
    }
    const int t0 = (blockIdx.y * 4 + w) * 32;
    if (t0 >= M) return;
    const int n0 = blockIdx.x * 32;
    const int tok = min(t0 + r31, M - 1);
    const bf16* wr = W + (int64_t)(n0 + r31) * K + hh * 8;       // A operand: feature n0 + r31, k = 16 f + 8 hh ..
    const bf16* xr = A + (int64_t)tok * K + hh * 8;              // B operand: token, same k
    bf16x8 wf[2][NF], xf[2][NF];
    auto request = [&](int c, int buf) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            wf[buf][f] = *(const bf16x8*)(wr + c * CK + f * 16);
            xf[buf][f] = *(const bf16x8*)(xr + c * CK + f * 16);
        }
    };
    request(0, 0);
    // epilogue operands, requested now: acc[4 q + e] = feature n0 + 8 q + 4 hh + e of token t0 + r31
    const bool live = t0 + r31 < M;
    f32x4 bq[4];
    bf16x4 rv[4];
    float2 st = {0.f, 1.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bq[q] = *(const f32x4*)(bias + n0 + 8 * q + 4 * hh);
        rv[q] = bf16x4{};
        if (EPI == EPI_RESID && live) rv[q] = *(const bf16x4*)(resid + (int64_t)(t0 + r31) * N + n0 + 8 * q + 4 * hh);
    }
    if constexpr (LNR) { if (live) st = rstats[t0 + r31]; }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c + 1 < NC) request(c + 1, (c + 1) & 1);
        if constexpr (LNA) {
            // LN(y) of this token on the way in: two-pass fp32 statistics over the 384 values = this lane's 192 and lane ^ 32's.  The
            // fragments are handled as packed 32-bit PAIRS (a shift / a mask widens a bf16, one v_cvt_pk_bf16_f32 packs two results):
            // element-wise bf16 arithmetic made hipcc keep every element in a register of its own (512 registers + 752 bytes of scratch)
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            auto lo_f = [](u32 u) { return __builtin_bit_cast(float, u << 16); };
            auto hi_f = [](u32 u) { return __builtin_bit_cast(float, u & 0xffff0000u); };
            float sm = 0.f;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const u32x4 u = __builtin_bit_cast(u32x4, xf[0][f]);
#pragma unroll
                for (int j = 0; j < 4; ++j) sm += lo_f(u[j]) + hi_f(u[j]);
            }
            sm += __shfl_xor(sm, 32);
            const float mu = sm * (1.0f / H);
            float qv = 0.f;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const u32x4 u = __builtin_bit_cast(u32x4, xf[0][f]);
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d0 = lo_f(u[j]) - mu, d1 = hi_f(u[j]) - mu; qv = fmaf(d0, d0, qv); qv = fmaf(d1, d1, qv); }
            }
            qv += __shfl_xor(qv, 32);
            const float rs = rsqrtf(qv * (1.0f / H) + eps);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int k0 = f * 16 + hh * 8;
                const u32x4 u = __builtin_bit_cast(u32x4, xf[0][f]);
                u32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = (lo_f(u[j]) - mu) * rs * sgb[k0 + 2 * j] + sgb[H + k0 + 2 * j];
                    const float b = (hi_f(u[j]) - mu) * rs * sgb[k0 + 2 * j + 1] + sgb[H + k0 + 2 * j + 1];
                    u32 pk;
                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b));
                    o[j] = pk;
                }
                xf[0][f] = __builtin_bit_cast(bf16x8, o);
                if (f % 4 == 3) asm volatile("" ::: "memory");     // at most four fragments' parameters in registers at a time
            }
            if (stats_out && blockIdx.x == 0 && hh == 0 && live) stats_out[t0 + r31] = float2{mu, rs};
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c & 1][f], xf[c & 1][f], acc, 0, 0, 0);
    }
    if (!live) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc[4 * q] + bq[q][0], acc[4 * q + 1] + bq[q][1], acc[4 * q + 2] + bq[q][2], acc[4 * q + 3] + bq[q][3]};
        if (EPI == EPI_GELU) v = gelu_poly4(v);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
        if (EPI == EPI_RESID) {
            bf16x4 r4 = rv[q];
            if constexpr (LNR) {
                const f32x4 gq = *(const f32x4*)(rg + n0 + 8 * q + 4 * hh), bb = *(const f32x4*)(rb + n0 + 8 * q + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) r4[e] = (bf16)((bf2f(r4[e]) - st.x) * st.y * gq[e] + bb[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16)(bf2f(o[e]) + bf2f(r4[e]));
        }
        *(bf16x4*)(out + (int64_t)(t0 + r31) * N + n0 + 8 * q + 4 * hh) = o;
    }
}

template <int EPI, int K, bool LNA = false, bool LNR = false>
static void launch_mid(hipStream_t s, int64_t m_cap, const bf16* A, const bf16* W, const float* bias, const bf16* resid, bf16* out, const int* cu,
                       int batch, int N, const float* lng = nullptr, const float* lnb = nullptr, float eps = 0.f, float2* stats_out = nullptr,
                       const float* rg = nullptr, const float* rb = nullptr, const float2* rstats = nullptr) {
    const dim3 grid((unsigned)(N / 32), (unsigned)((m_cap + 127) / 128));
    hipLaunchKernelGGL((k_gemm_mid<EPI, K, LNA, LNR>), grid, dim3(256), 0, s, A, W, bias, resid, out, cu, batch, N, lng, lnb, eps, stats_out, rg, rb, rstats);
}

// one k_gemm_small launch: grid = N / 32 feature blocks x token blocks of `tb`
template <int EPI, int K, bool LNA = false, bool LNR = false>
static void launch_small(hipStream_t s, int64_t m_cap, int tb, const bf16* A, const bf16* W, const float* bias, const bf16* resid, bf16* out, const int* cu,
                         int batch, int N, const float* lng = nullptr, const float* lnb = nullptr, float eps = 0.f, float2* stats_out = nullptr,
                         const float* rg = nullptr, const float* rb = nullptr, const float2* rstats = nullptr) {
    constexpr int NWV = K > 512 ? 8 : 4;
    const dim3 grid((unsigned)(N / 32), (unsigned)((m_cap + tb - 1) / tb));
    hipLaunchKernelGGL((k_gemm_small<EPI, K, LNA, LNR, NWV>), grid, dim3(64 * NWV), 0, s, A, W, bias, resid, out, cu, batch, N, tb, lng, lnb, eps, stats_out, rg, rb, rstats);
}

// ------------------------------------------------------------------------------------------------------------
// Fused FFN block: out = LayerNorm( GELU(h1 . W1^T + b1) . W2^T + b2 + h1 ) for a tile of 128 tokens.
//
// Unfused, the 1536-wide intermediate makes two HBM round trips per layer (3.2 GB written by FFN1, read again by FFN2),
// the pre-LN sum a third and the LayerNorm a fourth; here the intermediate never leaves the registers:
//   * 4 waves (one per SIMD, 512-register budget), wave w owns tokens [32w, 32w+32) of the tile for the WHOLE kernel:
//     its h1 rows are v_mfma_f32_32x32x16_bf16 B-fragments held in registers (96), the FFN2 accumulators for all 384
//     outputs too (192);
//   * the 1536 intermediate features are walked in 24 chunks of 64 (two MFMA tiles; with 128 the accumulators would fill
//     all 256 AGPRs and hipcc then spills the token fragments to scratch, whose reloads wait vmcnt(0) and drain the
//     LDS-DMA ring at every k-step): GEMM1 (K = 384) into 32 accumulator registers
//     (initialised with the bias), GELU(erf) + bf16 rounding (the same rounding point as the unfused `mid` tensor), and the
//     result is used DIRECTLY as the B operand of GEMM2: with swapped operands (A = weights) the D layout -- lane = token,
//     registers 8s .. 8s+7 = features {4h+e, 8+4h+e} of the 16-block s -- is a k-permuted B fragment of k-step s, and the
//     W2 copy this kernel reads has the columns of every 16-block permuted the same way (k_permute_w2);
//   * only the weights stream: W1 in [64 x 128] slabs, W2 in [384 x 32] slabs through a 4-slot LDS-DMA ring with
//     counted vmcnt (XOR swizzles on the source address as in k_gemm); 5 slabs = 5 barriers per chunk;
//   * a lone wave per SIMD hides only ~5 single-issue instructions behind a 32-cycle MFMA (MI355X_MICROARCH.md), so the
//     inner loops are ONE fragment read (address = per-k-step base register + literal offset), ONE counted wait and ONE
//     MFMA per step, four reads in flight; the activation of the next 32 features is slotted between GEMM2's MFMAs;
//   * epilogue as k_gemm_ln: bf16(acc + b2) parked in LDS, residual added, LayerNorm per token row.
// Weight traffic L2->LDS: 2.36 MB per 128 tokens (the layer's FFN weights stay L2-resident).
// ------------------------------------------------------------------------------------------------------------
namespace ffn {
constexpr int TOK = 128, CH = 64, NCH = FF / CH;      // tokens per tile, intermediate features per chunk, chunks
constexpr int SLOT = 24 * 1024, NSLOT = 4;            // ring slot (holds the larger slab type), slots (a power of two)
constexpr int W1_SLABS = 3, W2_SLABS = 2, PERIOD = W1_SLABS + W2_SLABS;
constexpr int W1_NI = 4, W2_NI = 6;                    // LDS-DMA wave-instructions per wave per slab
constexpr int TSTR = H * 2 + 16;                       // pre-LN tile row stride (epilogue, reuses the ring)
constexpr int B1_OFF = NSLOT * SLOT > TOK * TSTR ? NSLOT * SLOT : ((TOK * TSTR + 255) / 256) * 256;   // b1 (1536 fp32) behind ring / tile
constexpr int LDS_BYTES = B1_OFF + FF * 4;
constexpr int PRE = 4;                                 // fragment reads in flight (8 measured the same: 3.35 ms)
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
__host__ __device__ constexpr int slab_ni(int t) { return (t % PERIOD) < W1_SLABS ? W1_NI : W2_NI; }
// DMA instructions that may still be in flight when slab t must have landed: those of slabs t+1 .. t+NSLOT-2
__host__ __device__ constexpr int wait_n(int t) {
    int n = 0;
    for (int u = 1; u <= NSLOT - 2; ++u) n += slab_ni(t + u);
    return n;
}
// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{}) -- the index is a
// constant expression inside f, which inline-asm immediates (ds_read offset:, s_waitcnt) require
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
template <int OFF>
__device__ __forceinline__ void ds_read16(bf16x8& d, u32 addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// wait until at most N fragment reads are outstanding (the oldest has landed); tied to the register the MFMA reads
template <int N>
__device__ __forceinline__ void frag_wait(bf16x8& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); }
// a wave-uniform pointer pinned into an SGPR pair: `sgpr_ptr(base) + lane_offset` then selects the saddr + 32-bit voffset
// form of global_load_lds (left alone hipcc re-associates base + stride + offset into 64-bit VGPR adds per instruction)
__device__ __forceinline__ const char* sgpr_ptr(const char* p) {
    const unsigned long long v = (unsigned long long)p;
    const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)v), hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}
}  // namespace ffn

template <bool DBG>      // DBG: cycle counters into `dbg` (RMU_FFN_DBG diagnostics); the product instantiation carries none of it
__global__ __launch_bounds__(256) void k_ffn_fused(const bf16* __restrict__ h1, const bf16* __restrict__ W1,
                                                   const float* __restrict__ b1, const bf16* __restrict__ W2,
                                                   const float* __restrict__ b2, const float* __restrict__ g,
                                                   const float* __restrict__ bta, float eps, bf16* __restrict__ out,
                                                   const int* __restrict__ cu, int batch,
                                                   unsigned long long* __restrict__ dbg /* RMU_FFN_DBG: cycle counters, else null */) {
    using namespace ffn;
    const unsigned long long t_begin = DBG ? clock64() : 0;
    const int dflags = DBG ? (int)dbg[7] : 0;   // timing ablations (wrong results): 1 no DMA in the loop, 2 no fragment reads, 4 no GELU
    unsigned long long t_wait = 0, t_g1 = 0, t_g2 = 0;
    const int M = cu[batch];
    const int m0 = blockIdx.x * TOK;
    if (m0 >= M) return;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r31 = lane & 31, hh = lane >> 5;       // MFMA 32x32x16: row / column of the operand tile, k-half
    char* ring = gsm;
    float* b1s = (float*)(gsm + B1_OFF);

    // ---- h1 rows of this wave as B fragments (k-step ks covers k [16 ks, +16); lane half hh owns 8 of them), and b1 into
    // LDS: all ordinary loads are issued and waited for BEFORE the first LDS-DMA (an ordinary load beside an in-flight DMA
    // makes hipcc drain the ring) ------------------------------------------------------------------------------------------
    bf16x8 hf[24];
    {
        const int tok = min(m0 + 32 * w + r31, M - 1);
        const bf16* row = h1 + (int64_t)tok * H + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) hf[ks] = *(const bf16x8*)(row + ks * 16);
    }
    for (int i = threadIdx.x; i < FF; i += 256) b1s[i] = b1[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- slab stream: slab (c, i), i < 3: W1 rows [64 c, +64) x k [128 i, +128);  i >= 3: W2 rows [0, 384) x
    // k [64 c + 32 (i - 3), +32).  Slabs past the last chunk re-load the last chunk (keeps the vmcnt bookkeeping uniform).
    // Per-lane byte offsets are fixed for the whole kernel; a DMA issue is scalar base + 32-bit lane offset (no VALU).
    // (LDS unit f = (it * 4 + w) * 64 + lane of a slab: W1 row f >> 4, W2 row f >> 2; the swizzle term of a lane does not
    // depend on `it`, so instruction `it` is the lane's offset for it = 0 plus a SCALAR stride: 16 W1 rows / 64 W2 rows)
    u32 w1off0, w2off0;
    {
        const int f = w * 64 + lane;
        const int row1 = f >> 4, p1 = f & 15;
        w1off0 = (u32)((row1 * H + ((p1 ^ (row1 & 15)) * 8)) * 2);
        const int row2 = f >> 2, p2 = f & 3;
        w2off0 = (u32)((row2 * FF + ((p2 ^ ((row2 >> 2) & 3)) * 8)) * 2);
    }
    // DMA instruction `it` of slab number i of the period (compile time) for chunk cc (run time, clamped), into ring slot
    // `slotn`.  One at a time: an LDS-DMA issue costs the wave 60-180 cycles (MI355X_MICROARCH.md), so the instructions of a
    // slab are handed out BETWEEN the MFMAs of the slab being computed, not as a burst behind the barrier.
    auto issue_part = [&](int cc, int i, int slotn, int it) {
        if (cc >= NCH) cc = NCH - 1;
        char* slot = ring + slotn * SLOT;
        if (i < W1_SLABS) {
            const char* sbase = (const char*)W1 + ((size_t)cc * CH * H + (size_t)i * 128) * 2;
            u32 o = w1off0;
            asm volatile("" : "+v"(o));      // opaque: keeps hipcc from hoisting base + offset into loop-invariant 64-bit VGPR pairs
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sgpr_ptr(sbase + (size_t)it * (16 * H * 2)) + o),
                                             (__attribute__((address_space(3))) void*)(slot + (it * 4 + w) * 1024), 16, 0, 0);
        } else {
            const char* sbase = (const char*)W2 + ((size_t)cc * CH + (size_t)(i - W1_SLABS) * 32) * 2;
            u32 o = w2off0;
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sgpr_ptr(sbase + (size_t)it * (64 * FF * 2)) + o),
                                             (__attribute__((address_space(3))) void*)(slot + (it * 4 + w) * 1024), 16, 0, 0);
        }
    };
    auto issue = [&](int cc, int i, int slotn) {
        const int ni = i < W1_SLABS ? W1_NI : W2_NI;
#pragma unroll
        for (int it = 0; it < W2_NI; ++it)
            if (it < ni) issue_part(cc, i, slotn, it);
    };

    f32x16 acc2[12];                                // [output feature tile of 32]: rows = output features, columns = tokens
#pragma unroll
    for (int o = 0; o < 12; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[o][r] = 0.f;

    // Fragment addressing.  W1 slab image: [128 rows][16 units of 16 B], unit u of row r at physical unit u ^ (r & 15);
    // fragment (feature tile ft, k-step ks) of lane (row r31, half hh) = unit 2 ks + hh of row 32 ft + r31: one base register
    // per k-step, the feature tile is the literal offset ft * 8192.  W2 slab image: [384 rows][4 units], physical unit
    // u ^ ((r >> 2) & 3); fragment (output tile ot, k-step s) = unit 2 s + hh of row 32 ot + r31: base per s, offset ot * 2048.
    const u32 ring_addr = (u32)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
    u32 a1rel[8], a2rel[2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) a1rel[ks] = (u32)(r31 * 256 + (((2 * ks + hh) ^ (r31 & 15)) * 16));
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) a2rel[s2] = (u32)(r31 * 64 + (((2 * s2 + hh) ^ ((r31 >> 2) & 3)) * 16));
    bf16x8 fq[PRE];
#pragma unroll
    for (int m = 0; m < PRE; ++m) fq[m] = bf16x8{};

    static_assert(NSLOT - 1 <= PERIOD, "prologue stays inside chunk 0");
#pragma unroll
    for (int t = 0; t < NSLOT - 1; ++t) issue(0, t, t);
    int slot_cur = 0;                               // ring slot of the slab being consumed (run time: 84 slabs over 4 slots)
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    u32x4 pfu[2][2];                                // [tile parity][k-step]: B fragments of GEMM2 (current / next tile), as dwords
    for (int c = 0; c < NCH; ++c) {
        // GEMM1 accumulators start from the bias: register 4q + e of lane half hh <-> feature 32 ft + 8 q + 4 hh + e
        f32x16 acc1[2];                             // [intermediate feature tile of 32]
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *(const f32x4*)(b1s + c * CH + ft * 32 + q * 8 + hh * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1[ft][4 * q + e] = bv[e];
            }
        // activation of register group q (4 values) of feature tile t -> two dwords of the B fragment of k-step q / 2
        auto gelu_group = [&](int t, int q) {
            f32x4 v = {acc1[t][4 * q], acc1[t][4 * q + 1], acc1[t][4 * q + 2], acc1[t][4 * q + 3]};
            asm volatile("" : "+v"(v));              // program-order anchor: the work below cannot be hoisted above this point
            const f32x4 r = gelu_poly4(v);
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            bf16x2 lo, hi;
            lo[0] = (bf16)r[0]; lo[1] = (bf16)r[1]; hi[0] = (bf16)r[2]; hi[1] = (bf16)r[3];
            u32 d0 = __builtin_bit_cast(u32, lo), d1 = __builtin_bit_cast(u32, hi);
            asm volatile("" : "+v"(d0), "+v"(d1));   // ... nor sunk below this one: the group stays in the MFMA shadow it was given
            pfu[t & 1][q >> 1][(q & 1) * 2] = d0;
            pfu[t & 1][q >> 1][(q & 1) * 2 + 1] = d1;
        };
        static_for<PERIOD>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            // slab i of the chunk has landed (own DMA) once only the younger slabs' instructions are outstanding; own LDS reads
            // returned; after the barrier every wave is done with the slot that the new issue refills
            const unsigned long long tw0 = DBG ? clock64() : 0;
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(wait_n(i)) : "memory");
            __builtin_amdgcn_s_barrier();
            if (DBG) t_wait += clock64() - tw0;
            // the slab prefetched now (NSLOT - 1 ahead) goes into the slot everyone finished reading before this barrier
            constexpr int pi = (i + NSLOT - 1) % PERIOD, pni = pi < W1_SLABS ? W1_NI : W2_NI;
            const int pc = c + (i + NSLOT - 1) / PERIOD, pslot = (slot_cur + NSLOT - 1) & (NSLOT - 1);
            const u32 sa = ring_addr + (u32)slot_cur * SLOT;
            slot_cur = (slot_cur + 1) & (NSLOT - 1);
            const unsigned long long tc0 = DBG ? clock64() : 0;
            if constexpr (i < W1_SLABS) {
                // GEMM1: 8 k-steps x 2 feature tiles = 16 fragments, n = ks * 2 + ft; A = W1 rows (32 intermediate features
                // x 16 k), B = the token fragments in registers
                // The LAST k-slab runs feature tile 0 first (n = ft * 8 + ks): tile 0 is complete after 8 steps and the first half
                // of its activation (k-step 0 of GEMM2) is computed in the shadow of tile 1's MFMAs.
                constexpr bool last = i == W1_SLABS - 1;
                static_for<PRE>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    constexpr int ks = last ? (n & 7) : (n >> 1), ft = last ? (n >> 3) : (n & 1);
                    ds_read16<ft * 8192>(fq[n], sa + a1rel[ks]);
                });
                static_for<16>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    constexpr int ks = last ? (n & 7) : (n >> 1), ft = last ? (n >> 3) : (n & 1);
                    // reads n+1 .. min(n+PRE-1, 15) may still be in flight.  NO read is issued past the slab: a fragment register
                    // with a read in flight looks dead to hipcc, which would hand it out (as an address register!) under the data
                    frag_wait<(16 - 1 - n < PRE - 1 ? 16 - 1 - n : PRE - 1)>(fq[n % PRE]);
                    acc1[ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[n % PRE], hf[i * 8 + ks], acc1[ft], 0, 0, 0);
                    if constexpr (n + PRE < 16) {
                        constexpr int nn = n + PRE;
                        constexpr int ks2 = last ? (nn & 7) : (nn >> 1), ft2 = last ? (nn >> 3) : (nn & 1);
                        if (!DBG || !(dflags & 2)) ds_read16<ft2 * 8192>(fq[n % PRE], sa + a1rel[ks2]);
                    }
                    if constexpr (n % 2 == 1 && n / 2 < pni) { if (!DBG || !(dflags & 1)) issue_part(pc, pi, pslot, n / 2); }     // steps 1, 3, 5, ...
                    if (!DBG || !(dflags & 4)) {
                        if constexpr (last && n == 10) gelu_group(0, 0);
                        if constexpr (last && n == 13) gelu_group(0, 1);
                    }
                });
            } else {
                // GEMM2 over feature tile t = i - 3 of the chunk: 2 k-steps x 12 output tiles = 24 fragments, n = s * 12 + ot.
                // The activations are slotted between MFMAs (schedule below), so their VALU work sits in the MFMA shadow.
                constexpr int t = i - W1_SLABS;
                u32 ab[2];
                ab[0] = sa + a2rel[0];
                ab[1] = sa + a2rel[1];
                static_for<PRE>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    ds_read16<(n % 12) * 2048>(fq[n], ab[n / 12]);
                });
                static_for<24>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    constexpr int s2 = n / 12, ot = n % 12;
                    frag_wait<(24 - 1 - n < PRE - 1 ? 24 - 1 - n : PRE - 1)>(fq[n % PRE]);
                    acc2[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[n % PRE], __builtin_bit_cast(bf16x8, pfu[t & 1][s2]), acc2[ot], 0, 0, 0);
                    if constexpr (n + PRE < 24) {
                        constexpr int nn = n + PRE;
                        if (!DBG || !(dflags & 2)) ds_read16<(nn % 12) * 2048>(fq[n % PRE], ab[nn / 12]);
                    }
                    // activation schedule: every 12-MFMA half-slab carries two register groups, each the half-slab before the
                    // k-step that consumes it: tile 0 k-step 1 | tile 1 k-step 0 | tile 1 k-step 1 | (none)
                    if (!DBG || !(dflags & 4)) {
                        if constexpr (t == 0 && n == 2) gelu_group(0, 2);
                        if constexpr (t == 0 && n == 7) gelu_group(0, 3);
                        if constexpr (t == 0 && n == 14) gelu_group(1, 0);
                        if constexpr (t == 0 && n == 19) gelu_group(1, 1);
                        if constexpr (t == 1 && n == 2) gelu_group(1, 2);
                        if constexpr (t == 1 && n == 7) gelu_group(1, 3);
                    }
                    if constexpr (n % 4 == 0 && n / 4 < pni) { if (!DBG || !(dflags & 1)) issue_part(pc, pi, pslot, n / 4); }     // steps 0, 4, 8, ...
                });
            }
            if (DBG) { if (i < W1_SLABS) t_g1 += clock64() - tc0; else t_g2 += clock64() - tc0; }
        });
    }
    if (DBG && lane == 0) {     // [0] cycles waiting for slabs + barrier, [1] main-loop cycles, [2] wave count
        atomicAdd(dbg + 0, t_wait);
        atomicAdd(dbg + 1, (unsigned long long)(clock64() - t_begin));
        atomicAdd(dbg + 2, 1ull);
        atomicAdd(dbg + 3, t_g1);
        atomicAdd(dbg + 4, t_g2);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // drain the tail reloads: the ring becomes the pre-LN tile
    __syncthreads();
    char* tile = gsm;
    {
        // acc2[ot][4 q + e] = output feature n = 32 ot + 8 q + 4 hh + e of token 32 w + r31.  The residual h1[token][n] is
        // already on chip: hf[ks], ks = 2 ot + (q >> 1), holds features 16 ks + 8 h' .. + 7 in lane half h' -- the lane needs
        // element 8 (q & 1) + 4 hh + e of that block, which is its own register when (q & 1) == hh and the other half-lane's
        // (lane ^ 32) otherwise: one 2-dword exchange per k-step instead of 48 scattered 8-byte global loads.
        // y = bf16(bf16(acc + b2) + resid): the rounding points of the unfused path.
        const int tr = 32 * w + r31;
        typedef u32 u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int ot = 0; ot < 12; ++ot)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const u32x4 hv = __builtin_bit_cast(u32x4, hf[2 * ot + qq]);
                const u32x2 own = hh ? u32x2{hv[2], hv[3]} : u32x2{hv[0], hv[1]};      // elements 4 hh .. 4 hh + 3 of this half's 8
                const u32x2 oth = hh ? u32x2{hv[0], hv[1]} : u32x2{hv[2], hv[3]};      // what the partner half-lane needs
                u32x2 rcv;
                rcv[0] = (u32)__shfl_xor((int)oth[0], 32);
                rcv[1] = (u32)__shfl_xor((int)oth[1], 32);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const int q = 2 * qq + qb;
                    const u32x2 rs2 = (qb == hh) ? own : rcv;
                    const bf16x4 res = __builtin_bit_cast(bf16x4, rs2);
                    const int n = ot * 32 + q * 8 + hh * 4;
                    const f32x4 bv = *(const f32x4*)(b2 + n);
                    bf16x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (bf16)(bf2f((bf16)(acc2[ot][4 * q + e] + bv[e])) + bf2f(res[e]));
                    *(bf16x4*)(tile + tr * TSTR + n * 2) = v;
                }
            }
    }
    __syncthreads();
    // LayerNorm: each wave normalises its own 32 token rows; lanes 0..47 own 8 consecutive columns (16-byte accesses)
    {
        const bool act = lane < 48;
        const int c0 = (act ? lane : 0) * 8;
        float gg[8], bb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { gg[i] = g[c0 + i]; bb[i] = bta[c0 + i]; }
        for (int r = 0; r < 32; ++r) {
            const int tr = w * 32 + r;
            const int m = m0 + tr;
            if (m >= M) break;                         // uniform per wave
            const bf16x8 yv = *(const bf16x8*)(tile + tr * TSTR + c0 * 2);
            float v[8];
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[i] = act ? bf2f(yv[i]) : 0.f; sm += v[i]; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
            const float mu = sm * (1.0f / H);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = act ? v[i] - mu : 0.f; q = fmaf(d, d, q); }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
            const float rs = rsqrtf(q * (1.0f / H) + eps);
            bf16x8 ov;
#pragma unroll
            for (int i = 0; i < 8; ++i) ov[i] = (bf16)((v[i] - mu) * rs * gg[i] + bb[i]);
            if (act) *(bf16x8*)(out + (int64_t)m * H + c0) = ov;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_ffn2 -- the fused FFN block, second form (round 3): out = LN2( GELU(h1 . W1^T + b1) . W2^T + b2 + h1 ), optionally with
// h1 = LN1(y) computed in the prologue (the out-proj GEMM's pre-LN sum comes in, no k_layernorm launch and no h1 round trip).
//
// Same ownership as k_ffn_fused (4 waves = 4 SIMDs, wave w keeps tokens [32w, +32) as B fragments and all 384 FFN2 accumulators
// for the whole kernel, only weights stream through LDS), rebuilt around what bounded that kernel: a lone wave per SIMD issues at
// most ~one instruction per 4 cycles, i.e. <= ~5 besides the MFMA in each 32-cycle MFMA slot (MI355X_MICROARCH.md), and the old
// loop carried 7.9 (ISA count per 96-MFMA chunk: 369 VALU + 32 accvgpr reads, 104 ds_read, 102 s_waitcnt, ~100 SALU, 26 s_nop,
// 24 DMA) with the activation lumped in 40-instruction blocks between two MFMAs.  Here:
//   * GEMM2 runs ONE CHUNK BEHIND GEMM1: iteration c does GEMM1 of chunk c (48 MFMAs) and GEMM2 of chunk c-1 (48 MFMAs); the
//     activation of chunk c-1 (8 groups of 4 values per lane) is cut into 7 stages of 2-4 instructions, one stage behind each
//     MFMA of GEMM1(c) (and of the first GEMM2 slab for the last two groups): no MFMA ever waits for an activation, and no
//     gap carries more than stage + fragment read + DMA;
//   * GELU with the constants folded: gelu(v) = v (0.5 + vc Q(vc^2)), vc = clamp(v, +-3 sqrt 2), Q = the degree-7 erf fit
//     rescaled (same 4.7e-5 |v| error as gelu_poly): 1 clamp + 10 packable ops per value instead of 2 + 13;
//   * ring of 5 slots = the 5 slabs of a chunk, so every LDS address in the loop is base register + literal offset (no VALU),
//     and the slab base pointers advance on the scalar unit once per slab;
//   * fragment reads four ahead with ONE counted wait per two MFMAs.
// Stream per iteration c: slots 0-2 = W1 rows [64 c, +64) x k [128 i, +128); slots 3-4 = W2p rows [0, 384) x k [64 (c-1) + 32 t,
// +32).  Iteration 0 has no GEMM2, iteration NCH no GEMM1 (their slabs are loaded and ignored: 5 of 125 barriers).
// ------------------------------------------------------------------------------------------------------------
namespace ffn2 {
constexpr int TOK = 128, CH = 64, NCH = FF / CH;
constexpr int SLOT = 24 * 1024, NSLOT = 5;
constexpr int TSTR = H * 2 + 16;
constexpr int RING = NSLOT * SLOT;
constexpr int B1_OFF = RING;
constexpr int LDS_BYTES = B1_OFF + FF * 4;
static_assert(RING >= TOK * TSTR && LDS_BYTES <= 160 * 1024, "LDS");
// DMA instructions per wave that may still be in flight when slab i of an iteration must have landed (issue order per
// iteration: S4 during slab 0; S0', S1', S2' of the next iteration during slab 3; S3' during slab 4; W1 slab = 4, W2 slab = 6)
__host__ __device__ constexpr int wait_n(int i) { return i == 0 ? 14 : i == 1 ? 16 : i == 2 ? 12 : i == 3 ? 6 : 12; }
// folded GELU polynomial Q(u), u = vc^2 (highest degree first)
#define RMU_GQ0 -1.120145568e-09f
#define RMU_GQ1 9.479557069e-08f
#define RMU_GQ2 -3.475820744e-06f
#define RMU_GQ3 7.333383917e-05f
#define RMU_GQ4 -1.002580095e-03f
#define RMU_GQ5 9.521000741e-03f
#define RMU_GQ6 -6.597861542e-02f
#define RMU_GQ7 3.987713536e-01f
#define RMU_GCLAMP 4.2426406871f
}  // namespace ffn2

template <bool LN_IN, int GV, int PF>   // LN_IN: x is the pre-LayerNorm sum y, h1 = LN1(y) is formed here; GV: 0 packed / 1 scalar polynomial;
                                        // PF: weight fragments read ahead of the MFMA that eats them (4 or 8)
__global__ __launch_bounds__(256) void k_ffn2(const bf16* __restrict__ x, const bf16* __restrict__ W1, const float* __restrict__ b1,
                                              const bf16* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ g,
                                              const float* __restrict__ bta, float eps, bf16* __restrict__ out,
                                              const int* __restrict__ cu, int batch, const float* __restrict__ g1,
                                              const float* __restrict__ bta1) {
    using namespace ffn2;
    using ffn::static_for; using ffn::ds_read16; using ffn::sgpr_ptr;
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const int M = cu[batch];
    const int m0 = blockIdx.x * TOK;
    if (m0 >= M) return;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r31 = lane & 31, hh = lane >> 5;
    char* ring = gsm;
    float* b1s = (float*)(gsm + B1_OFF);

    // ---- token rows as B fragments (k-step ks = k [16 ks, +16), lane half hh owns 8 of them); b1 into LDS.  All ordinary
    // loads are issued and waited for BEFORE the first LDS-DMA. -------------------------------------------------------------
    bf16x8 hf[24];
    {
        const int tok = min(m0 + 32 * w + r31, M - 1);
        const bf16* row = x + (int64_t)tok * H + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) hf[ks] = *(const bf16x8*)(row + ks * 16);
    }
    for (int i = threadIdx.x; i < FF; i += 256) b1s[i] = b1[i];
    if (LN_IN) {
        // h1 = bf16(LN1(y)): the lane holds 192 of its token's 384 values, lane ^ 32 the others (same rounding point as the
        // k_layernorm launch this replaces: statistics in fp32 over the bf16 y, two passes)
        float sm = 0.f;
#pragma unroll
        for (int ks = 0; ks < 24; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) sm += bf2f(hf[ks][e]);
        sm += __shfl_xor(sm, 32);
        const float mu = sm * (1.0f / H);
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < 24; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = bf2f(hf[ks][e]) - mu; q = fmaf(d, d, q); }
        q += __shfl_xor(q, 32);
        const float rs = rsqrtf(q * (1.0f / H) + eps);
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) {
            const f32x4 ga = *(const f32x4*)(g1 + ks * 16 + hh * 8), gb = *(const f32x4*)(g1 + ks * 16 + hh * 8 + 4);
            const f32x4 ba = *(const f32x4*)(bta1 + ks * 16 + hh * 8), bb = *(const f32x4*)(bta1 + ks * 16 + hh * 8 + 4);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (bf16)((bf2f(hf[ks][e]) - mu) * rs * ga[e] + ba[e]);
                o[4 + e] = (bf16)((bf2f(hf[ks][4 + e]) - mu) * rs * gb[e] + bb[e]);
            }
            hf[ks] = o;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- slab stream -------------------------------------------------------------------------------------------------------
    // per-lane byte offsets inside a slab's source (LDS unit f = (it * 4 + w) * 64 + lane: W1 row f >> 4, W2 row f >> 2; the
    // swizzle term does not depend on `it`, so instruction `it` = offset of it 0 + a scalar stride: 16 W1 rows / 64 W2 rows)
    u32 w1off0, w2off0;
    {
        const int f = w * 64 + lane;
        const int row1 = f >> 4, p1 = f & 15;
        w1off0 = (u32)((row1 * H + ((p1 ^ (row1 & 15)) * 8)) * 2);
        const int row2 = f >> 2, p2 = f & 3;
        w2off0 = (u32)((row2 * FF + ((p2 ^ ((row2 >> 2) & 3)) * 8)) * 2);
    }
    // DMA instruction `it` of slab i (compile time) of iteration ci (run time; chunk clamped), into slot i.  The weight
    // matrices stay the scalar base of every load (the kernel-argument SGPR pair); a chunk / slab / instruction is a 32-bit
    // offset: chunk part on the scalar unit, one v_add_u32 into the lane offset per instruction (a 64-bit scalar base per
    // instruction cost s_add_u32 + s_addc_u32 + a VGPR copy + wait states behind the SGPR write).
    auto issue_part = [&](int ci, auto ic, int it) {
        constexpr int i = decltype(ic)::value;
        char* slot = ring + i * SLOT;
        if constexpr (i < 3) {
            const u32 cc = (u32)(ci < NCH ? ci : NCH - 1);
            const u32 o = w1off0 + (cc * (u32)(CH * H * 2) + (u32)(i * 256 + it * (16 * H * 2)));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)W1 + o),
                                             (__attribute__((address_space(3))) void*)(slot + (it * 4 + w) * 1024), 16, 0, 0);
        } else {
            const u32 cc = (u32)(ci < 1 ? 0 : (ci > NCH ? NCH - 1 : ci - 1));
            const u32 o = w2off0 + (cc * (u32)(CH * 2) + (u32)((i - 3) * 64 + it * (64 * FF * 2)));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)W2 + o),
                                             (__attribute__((address_space(3))) void*)(slot + (it * 4 + w) * 1024), 16, 0, 0);
        }
    };

    f32x16 acc2[12];
#pragma unroll
    for (int o = 0; o < 12; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[o][r] = 0.f;

    // fragment addressing: W1 slab image [64 rows][16 units], unit u of row r at u ^ (r & 15); W2 slab image [384 rows][4 units],
    // unit u ^ ((r >> 2) & 3).  Every address = one of these registers + a literal (slot, tile).
    const u32 ring_addr = (u32)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
    u32 a1[8], a2[2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) a1[ks] = ring_addr + (u32)(r31 * 256 + (((2 * ks + hh) ^ (r31 & 15)) * 16));
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) a2[s2] = ring_addr + (u32)(3 * SLOT) + (u32)(r31 * 64 + (((2 * s2 + hh) ^ ((r31 >> 2) & 3)) * 16));
    bf16x8 fq[PF];
#pragma unroll
    for (int m = 0; m < PF; ++m) fq[m] = bf16x8{};

    f32x16 acc1[2];                                 // GEMM1 accumulators of the chunk in flight [feature tile of 32]
    f32x16 prv[2];                                  // ... of the chunk before it (pre-activation, bias included): GELU source
    u32x4 pfu[2][2];                                // [feature tile][k-step]: B fragments of GEMM2 (chunk c - 1)
    f32x4 gc = {}, gu = {}, gp = {};                // the activation group in flight: clamped values, squares, polynomial
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) prv[t][r] = 0.f;   // iteration 0 activates zeros: its GEMM2 adds W2 . 0

    // GEMM1 accumulators start from the bias: register 4 q + e of lane half hh <-> feature 32 ft + 8 q + 4 hh + e
    auto load_bias = [&](int cc) {
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *(const f32x4*)(b1s + cc * CH + ft * 32 + q * 8 + hh * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1[ft][4 * q + e] = bv[e];
            }
    };

    // one stage (of six) of the activation of group gi = 4 t + q of the finished chunk: values prv[t][4 q ..], result -> pfu
    auto gelu_stage = [&](int gi, int st) {
        const int t = gi >> 2, q = gi & 3;
        auto sp = [](float c) { return f32x4{c, c, c, c}; };
        auto fma4 = [&](f32x4 a, f32x4 b, f32x4 c) {
            if (GV == 0) return __builtin_elementwise_fma(a, b, c);
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[e]) : "v"(a[e]), "v"(b[e]), "v"(c[e]));
            return r;
        };
        switch (st) {
            case 0:
                asm volatile("" : "+v"(gc));        // (program-order anchors: a stage stays in the MFMA gap it was given)
#pragma unroll
                for (int e = 0; e < 4; ++e) gc[e] = __builtin_amdgcn_fmed3f(prv[t][4 * q + e], -RMU_GCLAMP, RMU_GCLAMP);
                asm volatile("" : "+v"(gc));
                break;
            case 1:
                asm volatile("" : "+v"(gc));
                gu = gc * gc;
                gp = fma4(sp(RMU_GQ0), gu, sp(RMU_GQ1));
                asm volatile("" : "+v"(gu), "+v"(gp));
                break;
            case 2:
                asm volatile("" : "+v"(gp));
                gp = fma4(gp, gu, sp(RMU_GQ2));
                gp = fma4(gp, gu, sp(RMU_GQ3));
                asm volatile("" : "+v"(gp));
                break;
            case 3:
                asm volatile("" : "+v"(gp));
                gp = fma4(gp, gu, sp(RMU_GQ4));
                gp = fma4(gp, gu, sp(RMU_GQ5));
                asm volatile("" : "+v"(gp));
                break;
            case 4:
                asm volatile("" : "+v"(gp));
                gp = fma4(gp, gu, sp(RMU_GQ6));
                gp = fma4(gp, gu, sp(RMU_GQ7));
                asm volatile("" : "+v"(gp));
                break;
            default: {
                asm volatile("" : "+v"(gp));
                gp = fma4(gc, gp, sp(0.5f));
                const f32x4 pv = {prv[t][4 * q], prv[t][4 * q + 1], prv[t][4 * q + 2], prv[t][4 * q + 3]};
                gp = pv * gp;
                bf16x2 lo, hi;
                lo[0] = (bf16)gp[0]; lo[1] = (bf16)gp[1]; hi[0] = (bf16)gp[2]; hi[1] = (bf16)gp[3];
                u32 d0 = __builtin_bit_cast(u32, lo), d1 = __builtin_bit_cast(u32, hi);
                asm volatile("" : "+v"(d0), "+v"(d1));
                pfu[t][q >> 1][(q & 1) * 2] = d0;
                pfu[t][q >> 1][(q & 1) * 2 + 1] = d1;
            }
        }
    };

    // prologue of the stream: slabs 0 .. 3 of iteration 0
    static_for<4>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int ni = i < 3 ? 4 : 6;
#pragma unroll
        for (int it = 0; it < ni; ++it) issue_part(0, ic, it);
    });
    load_bias(0);

    // Iteration c: GEMM1 of chunk min(c, NCH - 1) (the last one recomputes a chunk whose result nobody reads), activation of
    // chunk c - 1, GEMM2 of chunk c - 1 -- ONE body for all NCH + 1 iterations (two uniform variants of this loop made hipcc
    // shuffle the 192 FFN2 accumulators between register ranges and spill them; 96 idle MFMAs of 2400 are the cheaper price).
    for (int c = 0; c <= NCH; ++c) {
        static_for<5>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(wait_n(i)) : "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (i == 0) asm volatile("" : "+a"(acc1[0]), "+a"(acc1[1]));   // the bias loads have landed (the wait above): no compiler wait later
            if constexpr (i < 3) {
                // ---- GEMM1 slab i: 8 k-steps x 2 feature tiles, n = 2 ks + ft; behind MFMA n: activation stage 16 i + n of chunk c - 1
                static_for<PF>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    ds_read16<i * SLOT + (n & 1) * 8192>(fq[n], a1[n >> 1]);
                });
                static_for<16>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    // fragments n, n + 1 have landed once at most the PF - 2 younger reads (fewer near the slab's end) are in flight
                    if constexpr (n % 2 == 0)
                        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fq[n % PF]), "+v"(fq[(n + 1) % PF]) : "n"(n + PF <= 16 ? PF - 2 : 16 - 2 - n));
                    acc1[n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[n % PF], hf[i * 8 + (n >> 1)], acc1[n & 1], 0, 0, 0);
                    if constexpr (n + PF < 16) ds_read16<i * SLOT + ((n + PF) & 1) * 8192>(fq[n % PF], a1[(n + PF) >> 1]);
                    else asm volatile("" : "+v"(fq[n % PF]));
                    if constexpr (i == 0 && n % 2 == 1 && n / 2 < 6) issue_part(c, std::integral_constant<int, 4>{}, n / 2);
                    gelu_stage((16 * i + n) / 6, (16 * i + n) % 6);
                });
            } else {
                // ---- GEMM2 over feature tile t of chunk c - 1: 2 k-steps x 12 output tiles, n = 12 s + ot.  Slab 3 also moves the
                // finished GEMM1 accumulators to `prv` (two values per step) and issues the next iteration's W1 slabs; slab 4 the
                // next iteration's first W2 slab and, behind its last MFMA, the next chunk's bias. --------------------------------
                constexpr int t = i - 3;
                static_for<PF>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    ds_read16<t * SLOT + n * 2048>(fq[n], a2[0]);
                });
                static_for<24>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    if constexpr (n % 2 == 0)
                        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fq[n % PF]), "+v"(fq[(n + 1) % PF]) : "n"(n + PF <= 24 ? PF - 2 : 24 - 2 - n));
                    acc2[n % 12] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[n % PF], __builtin_bit_cast(bf16x8, pfu[t][n / 12]), acc2[n % 12], 0, 0, 0);
                    if constexpr (n + PF < 24) ds_read16<t * SLOT + ((n + PF) % 12) * 2048>(fq[n % PF], a2[(n + PF) / 12]);
                    else asm volatile("" : "+v"(fq[n % PF]));
                    if constexpr (t == 0) {
                        if constexpr (n % 2 == 0) {          // S0', S1', S2' of iteration c + 1: 12 instructions over 24 steps
                            constexpr int k = n / 2;
                            if constexpr (k < 4) issue_part(c + 1, std::integral_constant<int, 0>{}, k);
                            else if constexpr (k < 8) issue_part(c + 1, std::integral_constant<int, 1>{}, k - 4);
                            else issue_part(c + 1, std::integral_constant<int, 2>{}, k - 8);
                        }
                        if constexpr (n >= 4 && n < 20) {    // (the last GEMM1 MFMAs have retired by step 4)
                            constexpr int e0 = 2 * (n - 4);
                            float m0v = acc1[e0 >> 4][e0 & 15], m1v = acc1[(e0 + 1) >> 4][(e0 + 1) & 15];
                            asm volatile("" : "+v"(m0v), "+v"(m1v));
                            prv[e0 >> 4][e0 & 15] = m0v;
                            prv[(e0 + 1) >> 4][(e0 + 1) & 15] = m1v;
                        }
                    } else {
                        if constexpr (n % 4 == 0) issue_part(c + 1, std::integral_constant<int, 3>{}, n / 4);
                    }
                });
                if constexpr (t == 1) load_bias(c + 1 < NCH ? c + 1 : NCH - 1);
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // drain the tail reloads: the ring becomes the pre-LN tile
    __syncthreads();
    char* tile = gsm;
    {
        // acc2[ot][4 q + e] = output feature n = 32 ot + 8 q + 4 hh + e of token 32 w + r31; the residual h1[token][n] sits in
        // hf[2 ot + (q >> 1)] of this lane ((q & 1) == hh) or of lane ^ 32 (see k_ffn_fused).  y = bf16(bf16(acc + b2) + resid).
        const int tr = 32 * w + r31;
        typedef u32 u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int ot = 0; ot < 12; ++ot)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const u32x4 hv = __builtin_bit_cast(u32x4, hf[2 * ot + qq]);
                const u32x2 own = hh ? u32x2{hv[2], hv[3]} : u32x2{hv[0], hv[1]};
                const u32x2 oth = hh ? u32x2{hv[0], hv[1]} : u32x2{hv[2], hv[3]};
                u32x2 rcv;
                rcv[0] = (u32)__shfl_xor((int)oth[0], 32);
                rcv[1] = (u32)__shfl_xor((int)oth[1], 32);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const int q = 2 * qq + qb;
                    const u32x2 rs2 = (qb == hh) ? own : rcv;
                    const bf16x4 res = __builtin_bit_cast(bf16x4, rs2);
                    const int n = ot * 32 + q * 8 + hh * 4;
                    const f32x4 bv = *(const f32x4*)(b2 + n);
                    bf16x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (bf16)(bf2f((bf16)(acc2[ot][4 * q + e] + bv[e])) + bf2f(res[e]));
                    *(bf16x4*)(tile + tr * TSTR + n * 2) = v;
                }
            }
    }
    __syncthreads();
    {
        const bool act = lane < 48;
        const int c0 = (act ? lane : 0) * 8;
        float gg[8], bb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { gg[i] = g[c0 + i]; bb[i] = bta[c0 + i]; }
        for (int r = 0; r < 32; ++r) {
            const int tr = w * 32 + r;
            const int m = m0 + tr;
            if (m >= M) break;
            const bf16x8 yv = *(const bf16x8*)(tile + tr * TSTR + c0 * 2);
            float v[8];
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[i] = act ? bf2f(yv[i]) : 0.f; sm += v[i]; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
            const float mu = sm * (1.0f / H);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = act ? v[i] - mu : 0.f; q = fmaf(d, d, q); }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
            const float rs = rsqrtf(q * (1.0f / H) + eps);
            bf16x8 ov;
#pragma unroll
            for (int i = 0; i < 8; ++i) ov[i] = (bf16)((v[i] - mu) * rs * gg[i] + bb[i]);
            if (act) *(bf16x8*)(out + (int64_t)m * H + c0) = ov;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_ffn3 -- the fused FFN block with TWO waves per SIMD (round 3).  tools/ubench/mfma_issue.hip: the loop skeleton of k_ffn2
// (wait + MFMA + fragment read, LDS-DMA, vmcnt + barrier) plus five VALU instructions per MFMA runs at 66.9 cycles per MFMA
// with one wave per SIMD and at 44.3 per SIMD with two -- the second wave issues while the first sits in a wait or a dependent
// VALU chain.  k_ffn2's wave needs ~370 registers, so the second wave has to come from splitting its work, not from occupancy:
//   * a workgroup is still 128 tokens, now 8 waves: waves p and p + 4 (p = 0..3) share tokens [32 p, +32);
//   * FFN1 is split over K: wave half s keeps the token fragments of k in [192 s, +192) only (48 registers instead of 96) and
//     produces, for both 32-feature tiles of a chunk, the partial sum over its half.  It owns tile s: the partner's partial for
//     that tile arrives through LDS (fp32, 4 KiB per wave and chunk) and is added before the activation;
//   * the activation is split over the feature tiles: each wave activates its own tile (16 values per lane) and publishes the
//     bf16 B fragments (2 KiB) for both waves of the pair;
//   * FFN2 is split over the OUTPUT features: wave half s accumulates outputs [192 s, +192) (96 registers instead of 192) over
//     all 64 features of the chunk; the residual of exactly those outputs sits in its own token fragments.
// Per wave and chunk: 24 + 24 MFMAs (as many per SIMD as k_ffn2), 48 fragment reads, 12 LDS-DMA instructions, 16 activations,
// 6 + 10 exchange LDS instructions; both exchanges ride on barriers the slab stream has anyway.  The W1 slab i now holds, for
// the chunk's 64 rows, k in [64 i, +64) and [192 + 64 i, +64) side by side (both halves work on every slab).  W1 slots are
// 16 KiB and W2 slots 24 KiB (slot = slab index, as in k_ffn2), which leaves room for the exchange buffers: 150 KiB.
// ------------------------------------------------------------------------------------------------------------
namespace ffn3 {
constexpr int TOK = 128, CH = 64, NCH = FF / CH;
constexpr int S1 = 16 * 1024, S2 = 24 * 1024;
constexpr int W2_OFF = 3 * S1, RING = 3 * S1 + 2 * S2;
constexpr int B1_OFF = RING;
constexpr int XP_OFF = B1_OFF + FF * 4;             // 8 waves x 4 KiB: the fp32 partial of the partner's tile
constexpr int PX_OFF = XP_OFF + 8 * 4096;           // 4 pairs x 2 tiles x 2 KiB: activated B fragments
constexpr int LN_OFF = PX_OFF + 8 * 2048;           // (round 5) gamma1 | beta1 | gamma2 | beta2 | b2, fp32: the two LayerNorms / the epilogue read them behind a barrier --
                                                    // from LDS that is a ~100-cycle read, from global an exposed L2 round trip with nothing else on the CU
constexpr int LDS_BYTES = LN_OFF + 5 * H * 4;      // (+ b2)
constexpr int TSTR = H * 2 + 16;
static_assert(LDS_BYTES <= 160 * 1024 && TOK * TSTR <= LDS_BYTES, "LDS");
// DMA instructions per wave that may still be in flight when slab i must have landed (W1 slab = 2 per wave, W2 slab = 3; issue
// order per iteration as in k_ffn2: S4 during slab 0; S0', S1', S2' during slab 3; S3' during slab 4)
__host__ __device__ constexpr int wait_n(int i) { return i == 0 ? 7 : i == 1 ? 8 : i == 2 ? 6 : i == 3 ? 3 : 6; }
// VAR bit 2, LOOK-AHEAD (round 4): at the barrier that opens slab i, slab i + 1 has landed as well, so the first PF fragments of slab
// i + 1 are read during the last PF steps of slab i and the MFMAs behind a barrier start at once -- without it all eight waves of a
// workgroup (both waves of every SIMD) sit out an LDS round trip behind each of the five barriers of a chunk at the same time.  The
// ring stays one chunk deep (5 slots = 5 slabs); the price is paid in DMA lead instead: the refill of slot j is issued during slab
// j + 1 (everybody has passed barrier j + 1, i.e. is done reading slot j) and must have landed by barrier j - 1 of the next round:
// three slab times for every slab (issue order R4, R0', R1', R2', R3', one refill per slab).  In flight at barrier i: the refills
// issued during slabs i - 1 and i - 2, i.e. of slots i - 2 and i - 3 (2 instructions per wave for a W1 slot, 3 for a W2 slot).
__host__ __device__ constexpr int slot_ni(int j) { return ((j % 5) + 5) % 5 < 3 ? 2 : 3; }
__host__ __device__ constexpr int wait_la(int i) { return slot_ni(i - 2) + slot_ni(i - 3); }
static_assert(wait_la(0) == 5 && wait_la(1) == 6 && wait_la(2) == 5 && wait_la(3) == 4 && wait_la(4) == 4, "look-ahead wait counts");
}  // namespace ffn3

template <bool LN_IN, int PF, bool DBG, int VAR, bool OP = false>   // OP: x is the attention output ctx -- the out-proj GEMM + bias + residual runs in the prologue (see below); VAR bit 0: activation as packed-f32 instructions (else hipcc's vector code), bit 1: k_pack_ffn3 weight streams (else W1 / W2p rows); DBG: ablation flags (RMU_FFN3_DBG, debug build): 1 no activation, 2 no fragment reads, 4 no DMA, 8 no exchange, 16 no barriers
__global__ __launch_bounds__(512) void k_ffn3(const bf16* __restrict__ x, const bf16* __restrict__ W1, const float* __restrict__ b1,
                                              const bf16* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ g,
                                              const float* __restrict__ bta, float eps, bf16* __restrict__ out,
                                              const int* __restrict__ cu, int batch, const float* __restrict__ g1,
                                              const float* __restrict__ bta1, int dflags, const bf16* __restrict__ wo = nullptr,
                                              const float* __restrict__ bo = nullptr, const bf16* __restrict__ resid = nullptr) {
    using namespace ffn3;
    using ffn::static_for; using ffn::ds_read16;
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    // (round 5, measured and not kept: issuing the token / parameter loads BEFORE M = cu[batch] is known, to share one round trip -- the half
    // of the grid that lies past M (the grid is sized by batch * max_len) then pays for loads it abandons: 3 186 vs 3 057 us per launch)
    const int M = cu[batch];
    const int m0 = blockIdx.x * TOK;
    if (m0 >= M) return;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = w & 3, s = w >> 2;
    const int r31 = lane & 31, hh = lane >> 5;
    char* ring = gsm;
    float* b1s = (float*)(gsm + B1_OFF);

    // ---- slab stream: W1 / W2 are the k_pack_ffn3 streams (slab images, contiguous); wave w moves bytes [2048 w, +2048) of a W1 slab and
    // [3072 w, +3072) of a W2 slab, 1 KiB per instruction: one LDS base (M0) and one address per slab, the pieces are immediate offsets
    // (applied to the global and the LDS address alike) -------------------------------------------------------------------------
    const u32 v1 = (u32)lane * 16 + (u32)w * 2048, v2 = (u32)lane * 16 + (u32)w * 3072;
    // (VAR bit 1 clear: W1 / W2 are the row-major matrices; DMA instruction q = it * 8 + w of a slab covers LDS units [64 q, +64), the
    // swizzle lives in the per-lane source offset -- 4 rows x 256 B resp. 16 rows x 64 B per instruction, one address + M0 per instruction)
    u32 w1off0 = 0, w2off0 = 0;
    if constexpr (!(VAR & 2)) {
        const int row1 = w * 4 + (lane >> 4), lu = (lane & 15) ^ (row1 & 15);
        w1off0 = (u32)((row1 * H + (lu & 7) * 8 + (lu >> 3) * 192) * 2);
        const int row2 = w * 16 + (lane >> 2), p2 = lane & 3;
        w2off0 = (u32)((row2 * FF + ((p2 ^ ((row2 >> 2) & 3)) * 8)) * 2);
    }
    auto issue_part = [&](int ci, auto ic, auto itc) {
        constexpr int i = decltype(ic)::value;
        constexpr int it = decltype(itc)::value;
        if (DBG && (dflags & 4)) return;
        if constexpr (i < 3) {
            const u32 cc = (u32)(ci < NCH ? ci : NCH - 1);
            if constexpr (VAR & 2) {
                const u32 o = v1 + (cc * (u32)(3 * S1) + (u32)(i * S1));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)W1 + o),
                                                 (__attribute__((address_space(3))) void*)(ring + i * S1 + w * 2048), 16, it * 1024, 0);
            } else {
                const u32 o = w1off0 + (cc * (u32)(CH * H * 2) + (u32)(i * 128 + it * (32 * H * 2)));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)W1 + o),
                                                 (__attribute__((address_space(3))) void*)(ring + i * S1 + (it * 8 + w) * 1024), 16, 0, 0);
            }
        } else {
            const u32 cc = (u32)(ci < 1 ? 0 : (ci > NCH ? NCH - 1 : ci - 1));
            if constexpr (VAR & 2) {
                const u32 o = v2 + (cc * (u32)(2 * S2) + (u32)((i - 3) * S2));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)W2 + o),
                                                 (__attribute__((address_space(3))) void*)(ring + W2_OFF + (i - 3) * S2 + w * 3072), 16, it * 1024, 0);
            } else {
                const u32 o = w2off0 + (cc * (u32)(CH * 2) + (u32)((i - 3) * 64 + it * (128 * FF * 2)));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)W2 + o),
                                                 (__attribute__((address_space(3))) void*)(ring + W2_OFF + (i - 3) * S2 + (it * 8 + w) * 1024), 16, 0, 0);
            }
        }
    };

    // (round 5) the stream's first four slabs are requested BEFORE the token rows: the ring is not touched by the LayerNorm prologue, and a
    // workgroup is alone on its CU (150 KiB of LDS) -- nobody else covers the L2 round trip of slab 0 behind the prologue
    auto issue_stream_prologue = [&]() {
        static_for<4>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            issue_part(0, ic, std::integral_constant<int, 0>{});
            issue_part(0, ic, std::integral_constant<int, 1>{});
            if constexpr (i == 3) issue_part(0, ic, std::integral_constant<int, 2>{});
        });
    };

    // ---- this half's k range of the token rows as B fragments; b1 into LDS ---------------------------------------------------
    bf16x8 hf[12];
    if constexpr (!OP) {
        const int tok = min(m0 + 32 * p + r31, M - 1);
        const bf16* row = x + (int64_t)tok * H + s * 192 + hh * 8;
#pragma unroll
        for (int j = 0; j < 12; ++j) hf[j] = *(const bf16x8*)(row + j * 16);   // (non-temporal here: +3.5 % -- the out-proj launch has just written these rows)
        issue_stream_prologue();       // behind the token loads in the CU's in-order memory queue: LayerNorm 1 starts on the rows while the slabs land
    }
    for (int i = threadIdx.x; i < FF; i += 512) b1s[i] = b1[i];
    float* lnp = (float*)(gsm + LN_OFF);
    if (threadIdx.x < H) {
        const int i = threadIdx.x;
        lnp[i] = g1 ? g1[i] : 1.f; lnp[H + i] = bta1 ? bta1[i] : 0.f; lnp[2 * H + i] = g[i]; lnp[3 * H + i] = bta[i]; lnp[4 * H + i] = b2[i];
    }
    if constexpr (OP) {
        // ---- the attention block's output projection, here instead of in a launch of its own: y = ctx . Wo^T + bo + h for this wave's
        // 32 tokens x ITS 192 features (the half s it needs as token fragments), then LayerNorm 1 -- the pre-LN sum never exists in memory,
        // the out-proj launch (0.56 ms per layer, 2.4 GB of traffic) and the statistics re-read are gone; the price is 144 MFMAs per wave in
        // front of the 1200 of the FFN and 288 KiB of Wo through the ring.  Same operand roles as FFN2: A = Wo rows out of W2-shaped slabs
        // ([384 rows][32 k], 12 of them through FOUR 24-KiB slots = the whole ring), B = the ctx rows of the tokens (all 24 k-steps in registers
        // for the duration, 96; the FFN's accumulators are not live yet), D = features x tokens.  D's register layout (register 4 q + e of
        // lane half hh = feature 8 q + 4 hh + e of the tile) IS a k-permuted B fragment -- as for FFN2's activations -- so LayerNorm's output goes
        // straight into the token fragments and W1 is read from its k-permuted copy (k_permute_w2 on W1: L.w1p); the residual of FFN2's
        // outputs is then the same register of the same lane (no lane exchange in the epilogue).
        const int tok = min(m0 + 32 * p + r31, M - 1);
        bf16x8 cf[24];
        if (dflags & 512) {                      // tiled ctx (k_attn3's store): block (token / 16, head = k / 32), unit ^ ((r >> 2) & 3)
            const int r = tok & 15, sw = tswz(r);
            const bf16* blk = x + (int64_t)(tok >> 4) * (NH * 512) + r * 32;
#pragma unroll
            for (int ks = 0; ks < 24; ++ks) cf[ks] = *(const bf16x8*)(blk + (ks >> 1) * 512 + ((((ks & 1) * 2 + hh) ^ sw) * 8));
        } else {
            const bf16* row = x + (int64_t)tok * H + hh * 8;
#pragma unroll
            for (int ks = 0; ks < 24; ++ks) cf[ks] = *(const bf16x8*)(row + ks * 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // ordinary loads done before the first LDS-DMA (hipcc drains the DMA queue for them otherwise)
        f32x16 co[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) co[j][r] = 0.f;
        u32 wooff0;
        {
            const int row2 = w * 16 + (lane >> 2), p2 = lane & 3;
            wooff0 = (u32)((row2 * H + ((p2 ^ ((row2 >> 2) & 3)) * 8)) * 2);
        }
        auto issue_wo = [&](int kb) {            // slab kb (clamped: harmless re-loads keep the counted wait uniform) into slot kb & 3
            const u32 kc = (u32)(kb < 12 ? kb : 11);
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const u32 o = wooff0 + (kc * 64u + (u32)(it * (128 * H * 2)));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)wo + o),
                                                 (__attribute__((address_space(3))) void*)(ring + (kb & 3) * S2 + (it * 8 + w) * 1024), 16, 0, 0);
            }
        };
        const u32 lds0p = (u32)(uintptr_t)(__attribute__((address_space(3))) char*)gsm;
        u32 ao[2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) ao[k2] = lds0p + (u32)((192 * s + r31) * 64 + (((2 * k2 + hh) ^ ((r31 >> 2) & 3)) * 16));
        issue_wo(0); issue_wo(1); issue_wo(2);
        bf16x8 fo[4];
#pragma unroll
        for (int kb = 0; kb < 12; ++kb) {                     // (unrolled: the ctx fragments are indexed by kb)
            asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");     // slab kb landed (own pieces); the two younger slabs may be in flight
            __builtin_amdgcn_s_barrier();                                     // ... everybody's pieces; and slot (kb + 3) & 3 = slab kb - 1's is free
            issue_wo(kb + 3);
            const u32 base0 = ao[0] + (u32)((kb & 3) * S2), base1 = ao[1] + (u32)((kb & 3) * S2);
            // 2 k-steps x 6 tiles, fragments four ahead, one counted wait per two MFMAs (as the FFN2 slabs below)
#pragma unroll
            for (int n = 0; n < 4; ++n) asm volatile("ds_read_b128 %0, %1" : "=v"(fo[n]) : "v"(base0 + (u32)(n * 2048)));
            const bf16x8 cf0 = cf[2 * kb], cf1 = cf[2 * kb + 1];
            static_for<12>([&, base0, base1, cf0, cf1](auto nc) {
                constexpr int n = decltype(nc)::value;
                if constexpr (n % 2 == 0)
                    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fo[n % 4]), "+v"(fo[(n + 1) % 4]) : "n"(n + 4 <= 12 ? 2 : 12 - 2 - n));
                co[n % 6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo[n % 4], n / 6 ? cf1 : cf0, co[n % 6], 0, 0, 0);
                if constexpr (n + 4 < 12) asm volatile("ds_read_b128 %0, %1" : "=v"(fo[n % 4]) : "v"(((n + 4) / 6 ? base1 : base0) + (u32)(((n + 4) % 6) * 2048)));
                else asm volatile("" : "+v"(fo[n % 4]));
            });
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");           // tail re-loads drained before the residual's ordinary loads
        // y = bf16(bf16(acc + bo) + h) (the out-proj kernel's rounding points), kept as fp32 of those bf16 values
        float sm = 0.f;
        {
            const int r = tok & 15, sw = tswz(r);
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f0 = s * 192 + j * 32 + q * 8 + hh * 4;
                    const bf16* rp = (dflags & 1024) ? resid + ((int64_t)(tok >> 4) * (H / 32) + (f0 >> 5)) * 512 + r * 32 + ((((f0 >> 3) & 3) ^ sw) * 8) + (f0 & 7)
                                                     : resid + (int64_t)tok * H + f0;
                    const bf16x4 rv = *(const bf16x4*)rp;
                    const f32x4 bv = *(const f32x4*)(bo + f0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float yv = bf2f((bf16)(bf2f((bf16)(co[j][4 * q + e] + bv[e])) + bf2f(rv[e])));
                        co[j][4 * q + e] = yv;
                        sm += yv;
                    }
                }
        }
        float* sc = (float*)(gsm + XP_OFF);
        sm += __shfl_xor(sm, 32);
        if (hh == 0) sc[w * 32 + r31] = sm;
        __syncthreads();
        sm += sc[(w ^ 4) * 32 + r31];
        const float mu = sm * (1.0f / H);
        float qv = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = co[j][r] - mu; qv = fmaf(d, d, qv); }
        qv += __shfl_xor(qv, 32);
        if (hh == 0) sc[256 + w * 32 + r31] = qv;
        __syncthreads();
        qv += sc[256 + (w ^ 4) * 32 + r31];
        const float rs = rsqrtf(qv * (1.0f / H) + eps);
        // token fragment of k-step 2 j + qh = registers of tile j, q in {2 qh, 2 qh + 1}: elements [0, 4) <- q even, [4, 8) <- q odd
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int qh = 0; qh < 2; ++qh) {
                bf16x8 o;
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const int q = 2 * qh + qb, f0 = s * 192 + j * 32 + q * 8 + hh * 4;
                    const f32x4 ga = *(const f32x4*)(g1 + f0), ba = *(const f32x4*)(bta1 + f0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[4 * qb + e] = (bf16)((co[j][4 * q + e] - mu) * rs * ga[e] + ba[e]);
                }
                hf[2 * j + qh] = o;
            }
    } else if (LN_IN) {
        // h1 = bf16(LN1(y)): statistics over all 384 values of the token = this lane's 96, lane ^ 32's, and the partner wave's
        // two lanes' (through LDS).  Same arithmetic as k_ffn2 (fp32, two passes) up to the order of the partial sums.
        float* sc = (float*)(gsm + XP_OFF);
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < 12; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) sm += bf2f(hf[j][e]);
        sm += __shfl_xor(sm, 32);
        if (hh == 0) sc[w * 32 + r31] = sm;
        __syncthreads();
        sm += sc[(w ^ 4) * 32 + r31];
        const float mu = sm * (1.0f / H);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 12; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = bf2f(hf[j][e]) - mu; q = fmaf(d, d, q); }
        q += __shfl_xor(q, 32);
        if (hh == 0) sc[256 + w * 32 + r31] = q;
        __syncthreads();
        q += sc[256 + (w ^ 4) * 32 + r31];
        const float rs = rsqrtf(q * (1.0f / H) + eps);
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int f0 = s * 192 + j * 16 + hh * 8;
            const f32x4 ga = *(const f32x4*)(lnp + f0), gb = *(const f32x4*)(lnp + f0 + 4);
            const f32x4 ba = *(const f32x4*)(lnp + H + f0), bb = *(const f32x4*)(lnp + H + f0 + 4);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (bf16)((bf2f(hf[j][e]) - mu) * rs * ga[e] + ba[e]);
                o[4 + e] = (bf16)((bf2f(hf[j][4 + e]) - mu) * rs * gb[e] + bb[e]);
            }
            hf[j] = o;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // (round 6, measured and removed: static s_setprio 1 for either half of the workgroup through the slab loop -- 3 301 / 3 306 vs 3 304 us, nothing;
    // and the two uniform branches that selected it cost the allocator 196 bytes of scratch in this kernel: NOTES_r06.md 6)

    f32x16 acc2[6];                                 // outputs [192 s + 32 j, +32)
#pragma unroll
    for (int o = 0; o < 6; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[o][r] = 0.f;

    const u32 lds0 = (u32)(uintptr_t)(__attribute__((address_space(3))) char*)gsm;
    u32 a1A[4], a1B[4], a2[2];                      // fragment addresses: own tile (rows 32 s + r31), partner's tile, W2 rows 192 s + r31
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const u32 unit = (u32)(((8 * s + 2 * ks + hh) ^ (r31 & 15)) * 16);
        a1A[ks] = lds0 + (u32)((32 * s + r31) * 256) + unit;
        a1B[ks] = lds0 + (u32)((32 * (1 - s) + r31) * 256) + unit;
    }
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) a2[k2] = lds0 + (u32)W2_OFF + (u32)((192 * s + r31) * 64 + (((2 * k2 + hh) ^ ((r31 >> 2) & 3)) * 16));
    const u32 xp_w = lds0 + (u32)XP_OFF + (u32)(w * 4096) + (u32)lane * 16;             // + j * 1024: registers 4 j .. 4 j + 3
    const u32 xp_r = lds0 + (u32)XP_OFF + (u32)((w ^ 4) * 4096) + (u32)lane * 16;
    const u32 px_w = lds0 + (u32)PX_OFF + (u32)((p * 2 + s) * 2048) + (u32)lane * 16;   // + k-step * 1024
    const u32 px_r = lds0 + (u32)PX_OFF + (u32)(p * 2 * 2048) + (u32)lane * 16;         // + tile * 2048 + k-step * 1024
    bf16x8 fq[PF];
#pragma unroll
    for (int m = 0; m < PF; ++m) fq[m] = bf16x8{};

    f32x16 accA, accB;                              // FFN1 partial sums over this half's k: own tile (bias included), partner's tile
    f32x16 prv;                                     // own tile of the chunk before: full pre-activation (GELU source)
    u32x4 pf[2];                                    // slabs 0-2: the own tile's activated fragments being built; slabs 3-4: the B fragments of the tile in use
#pragma unroll
    for (int r = 0; r < 16; ++r) { prv[r] = 0.f; accB[r] = 0.f; }
    pf[0] = u32x4{}; pf[1] = u32x4{};

    auto load_bias = [&](int cc) {                  // register 4 q + e of lane half hh <-> feature 32 s + 8 q + 4 hh + e of the chunk
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *(const f32x4*)(b1s + cc * CH + s * 32 + q * 8 + hh * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) accA[4 * q + e] = bv[e];
        }
    };
    // The activation in PACKED fp32 (v_pk_*_f32, two values per instruction), written as instructions because hipcc scalarises most of
    // the vector form once registers are tight (88 v_fma_f32 + 20 v_pk_fma_f32 per chunk).  gelu(v) = v (0.5 + Q0 vc P(u)), vc =
    // clamp(v), u = vc^2, P = Q / Q0 MONIC (its first Horner step is an add, and every step takes ONE constant: an SGPR pair --
    // gfx950 packed-f32 operands are 64-bit, one scalar source per instruction).  Same polynomial as k_ffn2's up to fp32 rounding.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto c2 = [](float c) { return f32x2{c, c}; };
    auto pk_fma_s = [](f32x2& d, f32x2 a, f32x2 k) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d) : "v"(a), "s"(k)); };   // d = d a + k
    f32x2 gcl = {}, gch = {}, gul = {}, guh = {}, gpl = {}, gph = {};
    // one stage (of six) of the activation of group q of the own tile: values prv[4 q ..] -> pf
    f32x4 gc = {}, gu = {}, gp = {};
    auto sp = [](float c) { return f32x4{c, c, c, c}; };
    auto gelu_stage_vec = [&](int q, int st) {        // VAR bit 0 clear: k_ffn2's stages, left to hipcc
        switch (st) {
            case 0:
                asm volatile("" : "+v"(gc));
#pragma unroll
                for (int e = 0; e < 4; ++e) gc[e] = __builtin_amdgcn_fmed3f(prv[4 * q + e], -RMU_GCLAMP, RMU_GCLAMP);
                asm volatile("" : "+v"(gc));
                break;
            case 1:
                asm volatile("" : "+v"(gc));
                gu = gc * gc;
                gp = __builtin_elementwise_fma(sp(RMU_GQ0), gu, sp(RMU_GQ1));
                asm volatile("" : "+v"(gu), "+v"(gp));
                break;
            case 2: case 3: case 4: {
                const float ca = st == 2 ? RMU_GQ2 : st == 3 ? RMU_GQ4 : RMU_GQ6, cb = st == 2 ? RMU_GQ3 : st == 3 ? RMU_GQ5 : RMU_GQ7;
                asm volatile("" : "+v"(gp));
                gp = __builtin_elementwise_fma(gp, gu, sp(ca));
                gp = __builtin_elementwise_fma(gp, gu, sp(cb));
                asm volatile("" : "+v"(gp));
                break;
            }
            default: {
                asm volatile("" : "+v"(gp));
                gp = __builtin_elementwise_fma(gc, gp, sp(0.5f));
                const f32x4 pv = {prv[4 * q], prv[4 * q + 1], prv[4 * q + 2], prv[4 * q + 3]};
                gp = pv * gp;
                bf16x2 lo, hi;
                lo[0] = (bf16)gp[0]; lo[1] = (bf16)gp[1]; hi[0] = (bf16)gp[2]; hi[1] = (bf16)gp[3];
                u32 d0 = __builtin_bit_cast(u32, lo), d1 = __builtin_bit_cast(u32, hi);
                asm volatile("" : "+v"(d0), "+v"(d1));
                pf[q >> 1][(q & 1) * 2] = d0;
                pf[q >> 1][(q & 1) * 2 + 1] = d1;
            }
        }
    };
    auto gelu_stage = [&](int q, int st) {
        if (DBG && (dflags & 1)) return;
        if constexpr (!(VAR & 1)) { gelu_stage_vec(q, st); return; }
        switch (st) {
            case 0:
                asm volatile("v_med3_f32 %0, %1, -%2, %2" : "=v"(gcl[0]) : "v"(prv[4 * q]), "s"(RMU_GCLAMP));   // one scalar source, negated in place
                asm volatile("v_med3_f32 %0, %1, -%2, %2" : "=v"(gcl[1]) : "v"(prv[4 * q + 1]), "s"(RMU_GCLAMP));   // one scalar source, negated in place
                asm volatile("v_med3_f32 %0, %1, -%2, %2" : "=v"(gch[0]) : "v"(prv[4 * q + 2]), "s"(RMU_GCLAMP));   // one scalar source, negated in place
                asm volatile("v_med3_f32 %0, %1, -%2, %2" : "=v"(gch[1]) : "v"(prv[4 * q + 3]), "s"(RMU_GCLAMP));   // one scalar source, negated in place
                break;
            case 1:
                asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(gul) : "v"(gcl));
                asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(guh) : "v"(gch));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(gpl) : "v"(gul), "s"(c2(RMU_GQ1 / RMU_GQ0)));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(gph) : "v"(guh), "s"(c2(RMU_GQ1 / RMU_GQ0)));
                break;
            case 2:
                pk_fma_s(gpl, gul, c2(RMU_GQ2 / RMU_GQ0)); pk_fma_s(gph, guh, c2(RMU_GQ2 / RMU_GQ0));
                pk_fma_s(gpl, gul, c2(RMU_GQ3 / RMU_GQ0)); pk_fma_s(gph, guh, c2(RMU_GQ3 / RMU_GQ0));
                break;
            case 3:
                pk_fma_s(gpl, gul, c2(RMU_GQ4 / RMU_GQ0)); pk_fma_s(gph, guh, c2(RMU_GQ4 / RMU_GQ0));
                pk_fma_s(gpl, gul, c2(RMU_GQ5 / RMU_GQ0)); pk_fma_s(gph, guh, c2(RMU_GQ5 / RMU_GQ0));
                break;
            case 4:
                pk_fma_s(gpl, gul, c2(RMU_GQ6 / RMU_GQ0)); pk_fma_s(gph, guh, c2(RMU_GQ6 / RMU_GQ0));
                pk_fma_s(gpl, gul, c2(RMU_GQ7 / RMU_GQ0)); pk_fma_s(gph, guh, c2(RMU_GQ7 / RMU_GQ0));
                break;
            default: {
                // gp = vc P;  gp = Q0 gp + 0.5;  gp = v gp;  -> bf16 pairs
                f32x2 pvl = {prv[4 * q], prv[4 * q + 1]}, pvh = {prv[4 * q + 2], prv[4 * q + 3]};
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(gpl) : "v"(gcl));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(gph) : "v"(gch));
                asm volatile("v_pk_fma_f32 %0, %0, %1, 0.5 op_sel_hi:[1,1,0]" : "+v"(gpl) : "s"(c2(RMU_GQ0)));
                asm volatile("v_pk_fma_f32 %0, %0, %1, 0.5 op_sel_hi:[1,1,0]" : "+v"(gph) : "s"(c2(RMU_GQ0)));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(gpl) : "v"(pvl));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(gph) : "v"(pvh));
                u32 d0, d1;
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d0) : "v"(gpl[0]), "v"(gpl[1]));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d1) : "v"(gph[0]), "v"(gph[1]));
                pf[q >> 1][(q & 1) * 2] = d0;
                pf[q >> 1][(q & 1) * 2 + 1] = d1;
            }
        }
    };

    // prologue of the stream: slabs 0 .. 3 of iteration 0 (OP: the ring carried Wo until here; else they were requested at the top)
    if constexpr (OP) issue_stream_prologue();
    load_bias(0);
    constexpr bool LA = (VAR & 4) != 0;
    if constexpr (LA) {
        // slab 0 of iteration 0 has no barrier in front of it that proves everybody's pieces landed: one extra wait + barrier, then
        // its first fragments (in flight behind it: slabs 1, 2, 3 = 2 + 2 + 3 instructions)
        asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        static_for<PF>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            ds_read16<0>(fq[n], (n & 1) ? a1B[n >> 1] : a1A[n >> 1]);
        });
    }

    // (measured and not kept, round 5: s_setprio 1 for the second-dispatched half of the workgroup -- waves 4-7 lose every contested issue
    // slot to their older partners -- 3424.8 vs 3429.4 us per launch, same box: nothing)
    // Iteration c (ONE body, as k_ffn2): FFN1 partials of chunk min(c, NCH - 1); activation of the own tile of chunk c - 1;
    // FFN2 of chunk c - 1 over this half's outputs.
    for (int c = 0; c <= NCH; ++c) {
        static_for<5>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LA ? wait_la(i) : wait_n(i)) : "memory");
            if (!DBG || !(dflags & 16)) __builtin_amdgcn_s_barrier();
            if constexpr (i == 0) asm volatile("" : "+v"(accA));      // the bias reads have landed (the wait above)
            if constexpr (i < 3) {
                // ---- FFN1 slab i: 4 k-steps x 2 tiles, n = 2 ks + (0: own tile, 1: partner's); behind MFMA n: activation stage 8 i + n
                if constexpr (!LA) static_for<PF>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    if (!DBG || !(dflags & 2)) ds_read16<i * S1>(fq[n], (n & 1) ? a1B[n >> 1] : a1A[n >> 1]);
                });
                static_for<8>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    if constexpr (n % 2 == 0)
                        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fq[n % PF]), "+v"(fq[(n + 1) % PF]) : "n"((LA || n + PF <= 8) ? PF - 2 : 8 - 2 - n));
                    if constexpr (n & 1) accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[n % PF], hf[i * 4 + (n >> 1)], accB, 0, 0, 0);
                    else accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[n % PF], hf[i * 4 + (n >> 1)], accA, 0, 0, 0);
                    if constexpr (n + PF < 8) { if (!DBG || !(dflags & 2)) ds_read16<i * S1>(fq[n % PF], ((n + PF) & 1) ? a1B[(n + PF) >> 1] : a1A[(n + PF) >> 1]); }
                    else if constexpr (LA) {                   // fragment m of the NEXT slab (landed: look-ahead wait at this slab's barrier)
                        constexpr int m = n + PF - 8;
                        if constexpr (i < 2) ds_read16<(i + 1) * S1>(fq[n % PF], (m & 1) ? a1B[m >> 1] : a1A[m >> 1]);
                        else ds_read16<(m % 6) * 2048>(fq[n % PF], a2[m / 6]);
                    } else asm volatile("" : "+v"(fq[n % PF]));
                    if constexpr (LA) {                        // one refill per slab: slot 4 during slab 0, W1 slot i - 1 of the next chunk during slabs 1, 2
                        if constexpr (i == 0 && n % 2 == 1 && n / 2 < 3) issue_part(c, std::integral_constant<int, 4>{}, std::integral_constant<int, n / 2>{});
                        if constexpr (i > 0 && n % 2 == 1 && n / 2 < 2) issue_part(c + 1, std::integral_constant<int, i - 1>{}, std::integral_constant<int, n / 2>{});
                    } else {
                        if constexpr (i == 0 && n % 2 == 1 && n / 2 < 3) issue_part(c, std::integral_constant<int, 4>{}, std::integral_constant<int, n / 2>{});
                    }
                    gelu_stage((8 * i + n) / 6, (8 * i + n) % 6);
                });
                if constexpr (i == 2) if (!DBG || !(dflags & 8)) {
                    // both exchanges (completed by the wait + barrier that open slab 3): the partner's tile partial, the own activated tile
                    asm volatile("ds_write_b128 %0, %1" ::"v"(px_w), "v"(pf[0]) : "memory");
                    asm volatile("ds_write_b128 %0, %1 offset:1024" ::"v"(px_w), "v"(pf[1]) : "memory");
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 v = {accB[4 * j], accB[4 * j + 1], accB[4 * j + 2], accB[4 * j + 3]};
                        if (j == 0) asm volatile("ds_write_b128 %0, %1" ::"v"(xp_w), "v"(v) : "memory");
                        if (j == 1) asm volatile("ds_write_b128 %0, %1 offset:1024" ::"v"(xp_w), "v"(v) : "memory");
                        if (j == 2) asm volatile("ds_write_b128 %0, %1 offset:2048" ::"v"(xp_w), "v"(v) : "memory");
                        if (j == 3) asm volatile("ds_write_b128 %0, %1 offset:3072" ::"v"(xp_w), "v"(v) : "memory");
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) accB[r] = 0.f;
                }
            } else {
                // ---- FFN2 over feature tile t of chunk c - 1: 2 k-steps x 6 output tiles, n = 6 k2 + j -------------------------------
                constexpr int t = i - 3;
                // exchange reads first (older than every fragment read: the first counted wait below covers them)
                f32x4 pp[4];
                if (DBG && (dflags & 8)) { pp[0] = pp[1] = pp[2] = pp[3] = f32x4{}; asm volatile("" : "+v"(pp[0]), "+v"(pp[1]), "+v"(pp[2]), "+v"(pp[3])); }
                if (!DBG || !(dflags & 8)) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pf[0]) : "v"(px_r), "n"(t * 2048));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pf[1]) : "v"(px_r), "n"(t * 2048 + 1024));
                }
                if constexpr (t == 0) if (!DBG || !(dflags & 8)) {
                    asm volatile("ds_read_b128 %0, %1" : "=v"(pp[0]) : "v"(xp_r));
                    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(pp[1]) : "v"(xp_r));
                    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(pp[2]) : "v"(xp_r));
                    asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(pp[3]) : "v"(xp_r));
                }
                if constexpr (!LA) static_for<PF>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    if (!DBG || !(dflags & 2)) ds_read16<t * S2 + (n % 6) * 2048>(fq[n], a2[n / 6]);
                });
                static_for<12>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    // look-ahead: the slab's first PF fragments are OLDER than the exchange reads above (they were issued during the previous
                    // slab); the first wait must reach past them to pf[0] (2, resp. 2 + 4 exchange reads, all but the first may stay in flight)
                    if constexpr (LA && n == 0)
                        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fq[0]), "+v"(fq[1]) : "n"(t == 0 ? 5 : 1));
                    else if constexpr (n % 2 == 0)
                        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fq[n % PF]), "+v"(fq[(n + 1) % PF]) : "n"((LA || n + PF <= 12) ? PF - 2 : 12 - 2 - n));
                    if constexpr (n == 0) asm volatile("" : "+v"(pf[0]), "+v"(pf[1]));
                    acc2[n % 6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[n % PF], __builtin_bit_cast(bf16x8, pf[n / 6]), acc2[n % 6], 0, 0, 0);
                    if constexpr (n + PF < 12) { if (!DBG || !(dflags & 2)) ds_read16<t * S2 + ((n + PF) % 6) * 2048>(fq[n % PF], a2[(n + PF) / 6]); }
                    else if constexpr (LA) {                   // fragment m of the next slab: W2 tile 1, resp. W1 slab 0 of the next iteration
                        constexpr int m = n + PF - 12;
                        if constexpr (t == 0) ds_read16<S2 + (m % 6) * 2048>(fq[n % PF], a2[m / 6]);
                        else ds_read16<0>(fq[n % PF], (m & 1) ? a1B[m >> 1] : a1A[m >> 1]);
                    } else asm volatile("" : "+v"(fq[n % PF]));
                    if constexpr (t == 0) {
                        if constexpr (LA) {                  // refill of W1 slot 2 (chunk c + 1): 2 instructions
                            if constexpr (n == 0 || n == 2) issue_part(c + 1, std::integral_constant<int, 2>{}, std::integral_constant<int, n / 2>{});
                        } else if constexpr (n % 2 == 0) {          // S0', S1', S2' of iteration c + 1: 6 instructions over 12 steps
                            constexpr int k = n / 2;
                            issue_part(c + 1, std::integral_constant<int, k / 2>{}, std::integral_constant<int, k % 2>{});
                        }
                        if constexpr (n >= 2 && n < 10) {    // own tile: full pre-activation = own partial + the partner's (two values per step)
                            constexpr int e0 = 2 * (n - 2);
                            if constexpr (e0 % 4 == 0) asm volatile("" : "+v"(pp[e0 / 4]));
                            float m0v = accA[e0] + pp[e0 / 4][e0 % 4], m1v = accA[e0 + 1] + pp[(e0 + 1) / 4][(e0 + 1) % 4];
                            asm volatile("" : "+v"(m0v), "+v"(m1v));
                            prv[e0] = m0v;
                            prv[e0 + 1] = m1v;
                        }
                    } else {
                        if constexpr (n % 4 == 0) issue_part(c + 1, std::integral_constant<int, 3>{}, std::integral_constant<int, n / 4>{});
                    }
                });
                if constexpr (t == 1) load_bias(c + 1 < NCH ? c + 1 : NCH - 1);
            }
        });
    }
    // (measured, round 5: moving this drain behind the epilogue's statistics -- it only has to precede the tile writes -- changes nothing:
    // 3 077 / 3 083 vs 3 065 / 3 092 us per launch)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // drain the tail reloads: the LDS becomes the output tile
    __syncthreads();
    char* tile = gsm;
    // ---- epilogue (round 5): y = bf16(bf16(acc + b2) + resid), LayerNorm 2, store.  The round-3/4 form wrote y into an LDS tile and gave
    // each wave 16 rows to normalise ONE AFTER THE OTHER, each with two 6-step ds_bpermute butterflies: 13 dependent LDS round trips per
    // row, 208 per wave, with nothing else resident on the CU (one workgroup per CU) -- ~25 k cycles of a ~190 k-cycle workgroup.  Here the
    // statistics are taken where the values are: a lane holds 96 of its token's 384 values, lane ^ 32 another 96, the partner wave
    // (w ^ 4) the other 192 -- the same two exchanges as LayerNorm 1 in the prologue (two-pass, fp32); the tile receives the FINISHED
    // bf16 rows and is copied out with 16-byte accesses, all reads in flight.  Measured (profiles/r05_ab_ffn3.txt, same box): 3193.4 -> 3056.2 us
    // per launch.  The old form stays in debug builds (dflags bit 2048, RMU_FFN3_EPI=0); with both in one kernel hipcc spills two registers.
#ifdef RMU_DEBUG_KERNELS
    if (!(dflags & 2048)) {
#else
    {
#endif
        // acc2[j][4 q + e] = output feature n = 192 s + 32 j + 8 q + 4 hh + e of token 32 p + r31; the residual h1[token][n] sits in
        // hf[2 j + (q >> 1)] of this lane ((q & 1) == hh) or of lane ^ 32 (as k_ffn2); OP: in this lane's fragment in accumulator order
        const int tr = 32 * p + r31;
        typedef u32 u32x2 __attribute__((ext_vector_type(2)));
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const u32x4 hv = __builtin_bit_cast(u32x4, hf[2 * j + qq]);
                const u32x2 own = hh ? u32x2{hv[2], hv[3]} : u32x2{hv[0], hv[1]};
                const u32x2 oth = hh ? u32x2{hv[0], hv[1]} : u32x2{hv[2], hv[3]};
                u32x2 rcv = {0u, 0u};
                if constexpr (!OP) {
                    rcv[0] = (u32)__shfl_xor((int)oth[0], 32);
                    rcv[1] = (u32)__shfl_xor((int)oth[1], 32);
                }
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const int q = 2 * qq + qb;
                    const u32x2 rs2 = OP ? (qb ? u32x2{hv[2], hv[3]} : u32x2{hv[0], hv[1]}) : ((qb == hh) ? own : rcv);
                    const bf16x4 res = __builtin_bit_cast(bf16x4, rs2);
                    const int n = s * 192 + j * 32 + q * 8 + hh * 4;
                    const f32x4 bv = *(const f32x4*)(lnp + 4 * H + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float yv = bf2f((bf16)(bf2f((bf16)(acc2[j][4 * q + e] + bv[e])) + bf2f(res[e])));
                        acc2[j][4 * q + e] = yv;
                        sm += yv;
                    }
                }
            }
        float* sc = (float*)(gsm + XP_OFF);          // beyond the tile (TOK * TSTR <= RING); the loop's exchanges are over (barrier 4 of the last iteration)
        sm += __shfl_xor(sm, 32);
        if (hh == 0) sc[w * 32 + r31] = sm;
        __syncthreads();
        sm += sc[(w ^ 4) * 32 + r31];
        const float mu = sm * (1.0f / H);
        float qv = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = acc2[j][r] - mu; qv = fmaf(d, d, qv); }
        qv += __shfl_xor(qv, 32);
        if (hh == 0) sc[256 + w * 32 + r31] = qv;
        __syncthreads();
        qv += sc[256 + (w ^ 4) * 32 + r31];
        const float rs = rsqrtf(qv * (1.0f / H) + eps);
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = s * 192 + j * 32 + q * 8 + hh * 4;
                const f32x4 ga = *(const f32x4*)(lnp + 2 * H + n), ba = *(const f32x4*)(lnp + 3 * H + n);
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (bf16)((acc2[j][4 * q + e] - mu) * rs * ga[e] + ba[e]);
                *(bf16x4*)(tile + tr * TSTR + n * 2) = v;
            }
        __syncthreads();
        // copy-out: wave w owns rows [16 w, +16), lanes 0..47 a 16-byte piece each; every read issued before the first store
        const bool act = lane < 48;
        const int c0 = (act ? lane : 0) * 8;
        bf16x8 ov[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ov[r] = *(const bf16x8*)(tile + (w * 16 + r) * TSTR + c0 * 2);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + w * 16 + r;
            if (act && m < M) {
                // dflags bit 8: the TILED activation form -- 1-KiB blocks of 16 tokens x 32 features, [token block][feature block] order, unit u
                // (8 features) of row r at u ^ tswz(r): exactly what a ring slot of the next layer's QKV GEMM / the out-proj's
                // residual piece holds, so their DMA instructions read whole contiguous KiB
                int64_t off = (int64_t)m * H + c0;
                if (dflags & 256) { const int rr = m & 15; off = ((int64_t)(m >> 4) * (H / 32) + (c0 >> 5)) * 512 + rr * 32 + ((((c0 >> 3) & 3) ^ tswz(rr)) * 8); }
                *(bf16x8*)(out + off) = ov[r];                  // (non-temporal here measured +0.6 %: the next layer's GEMMs read h right away)
            }
        }
        return;
    }
#ifdef RMU_DEBUG_KERNELS
    {
        // acc2[j][4 q + e] = output feature n = 192 s + 32 j + 8 q + 4 hh + e of token 32 p + r31; the residual h1[token][n] sits in
        // hf[2 j + (q >> 1)] of this lane ((q & 1) == hh) or of lane ^ 32 (as k_ffn2).  y = bf16(bf16(acc + b2) + resid).
        const int tr = 32 * p + r31;
        typedef u32 u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const u32x4 hv = __builtin_bit_cast(u32x4, hf[2 * j + qq]);
                const u32x2 own = hh ? u32x2{hv[2], hv[3]} : u32x2{hv[0], hv[1]};
                const u32x2 oth = hh ? u32x2{hv[0], hv[1]} : u32x2{hv[2], hv[3]};
                u32x2 rcv = {0u, 0u};
                if constexpr (!OP) {
                    rcv[0] = (u32)__shfl_xor((int)oth[0], 32);
                    rcv[1] = (u32)__shfl_xor((int)oth[1], 32);
                }
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const int q = 2 * qq + qb;
                    // OP: the fragments are in accumulator order -- elements [4 qb, +4) of this lane's k-step 2 j + qq ARE features 8 q + 4 hh + e
                    const u32x2 rs2 = OP ? (qb ? u32x2{hv[2], hv[3]} : u32x2{hv[0], hv[1]}) : ((qb == hh) ? own : rcv);
                    const bf16x4 res = __builtin_bit_cast(bf16x4, rs2);
                    const int n = s * 192 + j * 32 + q * 8 + hh * 4;
                    const f32x4 bv = *(const f32x4*)(b2 + n);
                    bf16x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (bf16)(bf2f((bf16)(acc2[j][4 * q + e] + bv[e])) + bf2f(res[e]));
                    *(bf16x4*)(tile + tr * TSTR + n * 2) = v;
                }
            }
    }
    __syncthreads();
    {
        const bool act = lane < 48;
        const int c0 = (act ? lane : 0) * 8;
        float gg[8], bb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { gg[i] = g[c0 + i]; bb[i] = bta[c0 + i]; }
        for (int r = 0; r < 16; ++r) {
            const int tr = w * 16 + r;
            const int m = m0 + tr;
            if (m >= M) break;
            const bf16x8 yv = *(const bf16x8*)(tile + tr * TSTR + c0 * 2);
            float v[8];
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[i] = act ? bf2f(yv[i]) : 0.f; sm += v[i]; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
            const float mu = sm * (1.0f / H);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = act ? v[i] - mu : 0.f; q = fmaf(d, d, q); }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
            const float rs = rsqrtf(q * (1.0f / H) + eps);
            bf16x8 ov;
#pragma unroll
            for (int i = 0; i < 8; ++i) ov[i] = (bf16)((v[i] - mu) * rs * gg[i] + bb[i]);
            if (act) {
                // dflags bit 8: the TILED activation form -- 1-KiB blocks of 16 tokens x 32 features, [token block][feature block] order, unit u
                // (8 features) of row r at u ^ tswz(r): exactly what a ring slot of the next layer's QKV GEMM / the out-proj's
                // residual piece holds, so their DMA instructions read whole contiguous KiB
                int64_t off = (int64_t)m * H + c0;
                if (dflags & 256) { const int r = m & 15; off = ((int64_t)(m >> 4) * (H / 32) + (c0 >> 5)) * 512 + r * 32 + ((((c0 >> 3) & 3) ^ tswz(r)) * 8); }
                *(bf16x8*)(out + off) = ov;
            }
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------
// k_gemm3 -- persistent big-tile GEMM for the encoder's shapes: out[m, n] = epi( sum_k A[m, k] W[n, k] + bias[n] ).
//
// What the cycle counters of k_gemm / k_ffn_fused and tools/ubench/fill.hip say about a CU's memory pipeline (all 256 CUs busy,
// L2-resident source): an LDS fill (global_load_lds, or global_load + ds_write alike) runs at ~146 GB/s per CU when a wave
// instruction covers 1 KiB of contiguous bytes but ~75 GB/s as 16 rows x 64 B -- the address path is paid per cache line touched
// (~0.55 lines per clock), whatever the number of waves (4..16) or loads in flight; stores leave at ~14 B/clk per CU (a 1-KiB
// global_store_dwordx4 per ~70 cycles, whatever the pattern); loads and stores of all waves share one in-order queue.  With
// 32-k stages (64-B row pieces) k_gemm's 128 x 128 tile needs 16 such loads (~480 cycles) per 256 cycles of MFMA work: it is
// bound by that queue, not by the matrix cores, and so is every LDS-staged GEMM here whose tile is small.
// Hence: a 256-token x 192-feature tile per workgroup (448 rows per 768 MFMA cycles), 8 waves per CU (two per SIMD, <= 256
// registers: while one sits in a load/store issue the other feeds the MFMA pipe), a 64 x 96 tile per wave (5 fragment reads
// per 6 v_mfma_f32_32x32x16_bf16), and a PERSISTENT loop over tiles whose 5-slot LDS-DMA ring never drains: the stages of the
// next tile are already in flight while the current tile's accumulators are stored (K is only 12 stages deep for three of
// the four GEMMs of a layer, so a drained prologue per tile would cost as much as the tile's MFMAs).
// Measured (8192 chunks, 1.04 M tokens): QKV 1.22 ms vs k_gemm's 1.38; the loop runs at ~2300 cycles per stage against 768
// of MFMA work -- ablations: without the epilogue's stores 1770, without DMA issue 1700, MFMAs + barrier alone 1200.  The
// 64-B row pieces (840 cycles of queue time per stage) and the stores (~580) are what is left to remove: 128-B rows need
// 64-k stages, i.e. 3 x 56 KiB of LDS, which does not fit beside anything; a tiled activation layout would (next round).
//   * stage = 32 k (64-B rows: X 16 KiB + W 12 KiB); DMA instructions (SGPR base + lane offset) are handed out between MFMAs,
//     2 X pieces per wave in the second half-step, 1-2 W pieces in the first; counted vmcnt; ONE barrier per stage;
//   * fragments are read one 16-k step ahead of the MFMAs that eat them (inline asm, literal offsets, counted lgkmcnt 2/4/4):
//     the token fragments of the next step go to the other register set, each weight fragment is re-read in place behind
//     its two MFMAs;
//   * epilogue through a 2-KiB strip per wave BESIDE the ring (which stays in use): bias (scalar loads) | + GELU | + residual;
//     weight rows are read permuted so that a lane owns 16 consecutive features; stores are 16 rows x 64 B per instruction
//     (the CU's store path moves ~14 B/clk whatever the pattern; spreading the stores over the next tile's stages was slower);
//   * tiles: the column blocks of one token tile run on neighbouring CUs of one XCD (its L2 serves the re-reads of X).
// ------------------------------------------------------------------------------------------------------------
namespace g3 {
constexpr int BM = 256, BNF = 192;
constexpr int XBYTES = BM * 64, SLOT = XBYTES + BNF * 64, NSLOT = 5;   // 28 KiB per stage
constexpr int RING = NSLOT * SLOT;
constexpr int STRIP = 32 * 64;                                         // epilogue: one 32-token x 32-feature tile per wave (swizzled)
constexpr int LDS_BYTES = RING + 8 * STRIP;
}  // namespace g3

template <int EPI, bool DBG>
__global__ __launch_bounds__(512) void k_gemm3(const bf16* __restrict__ A, const bf16* __restrict__ W, const float* __restrict__ bias,
                                               const bf16* __restrict__ resid, bf16* __restrict__ out, const int* __restrict__ cu,
                                               int batch, int N, int K, unsigned long long* dbg, int dflags, long hm_stride) {
    // hm_stride > 0: the output is written HEAD-MAJOR -- [feature / 32][hm_stride token rows][32] -- the layout k_attn3 reads its (sequence,
    // head) blocks from as contiguous bytes (a store instruction's 16 rows x 64 B are then one contiguous KiB too)
    using namespace g3;
    using ffn::static_for; using ffn::ds_read16; using ffn::frag_wait; using ffn::sgpr_ptr;
    const int M = cu[batch];
    const int ncb = N / BNF, nk = K / 32;
    const int mtiles = (M + BM - 1) / BM;
    const int xcd = blockIdx.x & 7, cslot = blockIdx.x >> 3, ncu = gridDim.x >> 3;
    // tiles of this XCD: local index li -> token tile (li / ncb) * 8 + xcd, column block li % ncb; this workgroup takes li = cslot + j * ncu
    const int mt_x = mtiles > xcd ? (mtiles - xcd + 7) / 8 : 0;
    const int L = mt_x * ncb;
    const int ntiles = L > cslot ? (L - cslot + ncu - 1) / ncu : 0;
    if (ntiles == 0) return;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int r31 = lane & 31, hh = lane >> 5;
    char* ring = gsm;
    char* strip = gsm + RING + w * STRIP;
    unsigned long long tc0 = 0, tc1 = 0, twait = 0, tepi = 0;
    if (DBG) tc0 = __builtin_readcyclecounter();

    auto tile_of = [&](int j, int& m0, int& n0) {
        const int li = cslot + j * ncu;
        const int q = li / ncb;
        m0 = (q * 8 + xcd) * BM;
        n0 = (li - q * ncb) * BNF;
    };

    // ---- issue cursor: the stage the next DMA instructions belong to (runs NSLOT - 1 stages ahead of the MFMAs, across tiles) --
    // X piece xi = w + 8 i (i = 0, 1): rows 16 xi + (lane >> 2), physical unit lane & 3 holding logical unit (lane & 3) ^ tswz(row)
    // (row & 15 = lane >> 2); rows past the last token are clamped (never stored).  W piece wi = w (+ 8 for waves 0..3).
    int ij = 0, it = 0, islot = 0;
    const char *ia, *iw;
    u32 xoff[2];
    const bool a_tiled = (dflags & 256) != 0;          // A is a tiled activation (k_ffn3's store form)
    const size_t astep = a_tiled ? 1024 : 64;          // bytes from one 32-k stage to the next
    const u32 woff = (u32)lane * 16;                   // W is the k_tile_w copy: a DMA instruction reads one contiguous 1-KiB block
    auto set_issue_tile = [&](int j) {
        int m0, n0;
        tile_of(j, m0, n0);
        const int rows_here = M - m0;
        ia = a_tiled ? (const char*)A + (size_t)(m0 >> 4) * (K / 32) * 1024 : (const char*)(A + (size_t)m0 * K);
        iw = (const char*)W + (size_t)(n0 / 16) * (K / 32) * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (w + 8 * i) * 16 + (lane >> 2);
            const int rc = row < rows_here ? row : rows_here - 1;
            // tiled A (1-KiB blocks of 16 tokens x 32 k, already in slot order): piece xi of stage `it` = block (m0 / 16 + xi, it), read whole
            xoff[i] = a_tiled ? (u32)((w + 8 * i) * (K / 32) * 1024 + lane * 16) : (u32)(((size_t)rc * K + (((lane & 3) ^ tswz(lane >> 2)) * 8)) * 2);
        }
    };
    auto issue_x = [&](int i) {
        u32 o = xoff[i];
        asm volatile("" : "+v"(o));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sgpr_ptr(ia + (size_t)it * astep) + o),
                                         (__attribute__((address_space(3))) void*)(ring + islot * SLOT + (w + 8 * i) * 1024), 16, 0, 0);
    };
    auto issue_w = [&](int i) {
        if (i == 1 && w >= 4) return;
        u32 o = woff;
        asm volatile("" : "+v"(o));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sgpr_ptr(iw + ((size_t)(w + 8 * i) * (K / 32) + it) * 1024) + o),
                                         (__attribute__((address_space(3))) void*)(ring + islot * SLOT + XBYTES + (w + 8 * i) * 1024), 16, 0, 0);
    };
    auto advance = [&]() {
        islot = islot == NSLOT - 1 ? 0 : islot + 1;
        if (++it == nk) {
            if (ij + 1 < ntiles) { ++ij; set_issue_tile(ij); it = 0; }
            else it = nk - 1;                          // past the end: harmless re-loads of the last stage into free slots
        }
    };
    set_issue_tile(0);
#pragma unroll
    for (int s0 = 0; s0 < NSLOT - 1; ++s0) { issue_x(0); issue_x(1); issue_w(0); issue_w(1); advance(); }

    f32x16 acc[3][2];                                  // [feature tile of 32][token tile of 32]
#pragma unroll
    for (int ft = 0; ft < 3; ++ft)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ft][tt][e] = 0.f;

    // fragment addressing: lane (row r31, half hh) reads unit 2 s + hh of row (tile base + r31): base register per s, tile = literal
    const u32 ring_addr = (u32)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
    // Weight rows are read PERMUTED: MFMA row i of a 32-feature tile is LDS row f(i) = 16 ((i >> 2) & 1) + 4 (i >> 3) + (i & 3), so
    // that the accumulator registers 0..15 of a lane (D rows 8 q + 4 hh + e) hold the 16 CONSECUTIVE features 16 hh + 4 q + e of
    // its token: the epilogue moves 16-byte pieces.  (Bank-conflict-free like the identity: the 16-lane groups of a b128 read
    // still meet four different (row >> 2) & 3 swizzles.)
    const int fr = 16 * ((r31 >> 2) & 1) + 4 * (r31 >> 3) + (r31 & 3);
    u32 xrel[2], wrel[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        xrel[s2] = (u32)(r31 * 64 + (((2 * s2 + hh) ^ tswz(r31)) * 16)) + (u32)(wm * 64 * 64);
        wrel[s2] = (u32)(fr * 64 + (((2 * s2 + hh) ^ tswz(fr)) * 16)) + (u32)(XBYTES + wn * 96 * 64);
    }
    bf16x8 wf[3], xf[2][2];

    if (w < 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");   // stage 0 landed
    __syncthreads();
    if (DBG) tc1 = __builtin_readcyclecounter();
    ds_read16<0>(xf[0][0], ring_addr + xrel[0]);
    ds_read16<2048>(xf[0][1], ring_addr + xrel[0]);
    static_for<3>([&](auto c) { constexpr int ft = decltype(c)::value; ds_read16<ft * 2048>(wf[ft], ring_addr + wrel[0]); });

    // one 16-k step: 3 groups of 2 MFMAs on (wf[ft], xf[cur][0..1]); the reads of the NEXT step (bases nxb / nwb) go out underneath:
    // xf[1 - cur][0..1] during group 0, wf[ft] in place behind group ft.  DMA: 0 = none, 1 = the X pieces, 2 = the W pieces.
    auto step = [&](auto curc, u32 nxb, u32 nwb, auto dmac) {
        constexpr int cur = decltype(curc)::value;
        constexpr int dma = decltype(dmac)::value;
        // group 0 needs wf[0] and xf[cur]: everything but the last two reads (wf[1], wf[2]) has to have landed
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(wf[0]), "+v"(xf[cur][0]), "+v"(xf[cur][1]));
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], xf[cur][0], acc[0][0], 0, 0, 0);
        if (!DBG || !(dflags & 2)) ds_read16<0>(xf[1 - cur][0], nxb);
        if (!DBG || !(dflags & 1)) {
            if constexpr (dma == 1) issue_x(0);
            if constexpr (dma == 2) issue_w(0);
        }
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], xf[cur][1], acc[0][1], 0, 0, 0);
        if (!DBG || !(dflags & 2)) { ds_read16<2048>(xf[1 - cur][1], nxb); ds_read16<0>(wf[0], nwb); }
        frag_wait<4>(wf[1]);                           // behind wf[1]'s read: wf[2], 2 x, wf[0]
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1], xf[cur][0], acc[1][0], 0, 0, 0);
        if (!DBG || !(dflags & 1)) {
            if constexpr (dma == 1) issue_x(1);
            if constexpr (dma == 2) issue_w(1);
        }
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1], xf[cur][1], acc[1][1], 0, 0, 0);
        if (!DBG || !(dflags & 2)) ds_read16<2048>(wf[1], nwb);
        frag_wait<4>(wf[2]);                           // behind wf[2]'s read: 2 x, wf[0], wf[1]
        acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[2], xf[cur][0], acc[2][0], 0, 0, 0);
        acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[2], xf[cur][1], acc[2][1], 0, 0, 0);
        if (!DBG || !(dflags & 2)) ds_read16<4096>(wf[2], nwb);
    };

    int cj = 0, ct = 0, cs = 0, m0c, n0c;
    tile_of(0, m0c, n0c);

    // ---- the finished tile as 12 store-ready pieces per lane (16 rows x 64 B per instruction).  The store path of a CU moves
    // ~14 B/clk (one 1-KiB global_store_dwordx4 per ~70 cycles, whichever wave issues it): a tile's 96 KiB cost ~7k cycles.
    // Piece k = 4 ft + 2 tt + i: row 32 tt + 16 i + (lane >> 2) of the wave's 64, features 32 ft + 8 (lane & 3)...
    bf16x8 so[12];
    int spend = 12;                                    // next piece to store (12 = nothing parked)
    int srows = 0;                                     // rows of the parked tile below this lane's first row that exist (m < M)
    const bf16* sbase = out;
    const int64_t srow = hm_stride > 0 ? 32 : N, sft = hm_stride > 0 ? (int64_t)hm_stride * 32 : 32;   // element strides of a piece's rows / of the 32-feature tiles
    auto store_piece = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int ft = k >> 2, tt = (k >> 1) & 1, i = k & 1;
        if (tt * 32 + i * 16 < srows) {
            if (DBG && (dflags & 8)) asm volatile("" ::"v"(so[k]));
            // non-temporal: the 2.4 GB of Q | K | V per layer are read once, by the next launch, long after they left the L2 -- 1 109 -> 1 068 us per launch
            else __builtin_nontemporal_store(so[k], (bf16x8*)(sbase + (int64_t)(tt * 32 + i * 16) * srow + ft * sft));
        }
    };
    auto store_next = [&]() {                          // wave-uniform dispatch: the pieces live in fixed registers
        switch (spend) {
            case 0: store_piece(std::integral_constant<int, 0>{}); break;
            case 1: store_piece(std::integral_constant<int, 1>{}); break;
            case 2: store_piece(std::integral_constant<int, 2>{}); break;
            case 3: store_piece(std::integral_constant<int, 3>{}); break;
            case 4: store_piece(std::integral_constant<int, 4>{}); break;
            case 5: store_piece(std::integral_constant<int, 5>{}); break;
            case 6: store_piece(std::integral_constant<int, 6>{}); break;
            case 7: store_piece(std::integral_constant<int, 7>{}); break;
            case 8: store_piece(std::integral_constant<int, 8>{}); break;
            case 9: store_piece(std::integral_constant<int, 9>{}); break;
            case 10: store_piece(std::integral_constant<int, 10>{}); break;
            default: store_piece(std::integral_constant<int, 11>{}); break;
        }
        ++spend;
    };
    int s1 = 0, s2 = 0;                                // a store went out in the previous / the one-before-previous iteration
    // MEASURED (8192 chunks, QKV): deferring is SLOWER -- 1.69 ms vs 1.32 ms with the twelve stores issued at once behind the
    // epilogue: a store queued among the DMA instructions of a stage delays them (one in-order memory-instruction queue per CU),
    // and the counted waits then also wait for store acknowledgements.  Kept switchable (dflags bit 5, debug build) for the record.
    const bool defer = DBG && (dflags & 32);

    step(std::integral_constant<int, 0>{}, ring_addr + xrel[1], ring_addr + wrel[1], std::integral_constant<int, 0>{});
    for (;;) {
        // stage (current + 1) has to have landed for every wave before its fragments are read ahead; every wave is done with the
        // stage before the current one (its last reads completed before the current stage's first step began): its slot is refilled.
        // Younger than that stage's last DMA instruction: the DMA instructions of two stages and up to two deferred stores.
        unsigned long long tw = 0;
        if (DBG) tw = __builtin_readcyclecounter();
        switch ((w < 4 ? 8 : 6) + s1 + s2) {
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        }
        __builtin_amdgcn_s_barrier();
        if (DBG) twait += __builtin_readcyclecounter() - tw;
        const int ns = cs == NSLOT - 1 ? 0 : cs + 1;
        const u32 sn = ring_addr + (u32)ns * SLOT;
        step(std::integral_constant<int, 1>{}, sn + xrel[0], sn + wrel[0], std::integral_constant<int, 1>{});
        if (++ct == nk) {
            // ---- tile done: acc[ft][tt][i] = feature n0c + 96 wn + 32 ft + 16 hh + i of token m0c + 64 wm + 32 tt + r31.  Each
            // 32 x 32 tile goes through the wave's LDS strip (lane writes its 2 x 16 B, rows of 64 B, unit ^ ((row >> 2) & 3)) and
            // comes back as pieces of 16 rows x 64 contiguous bytes (nk >= 12: the previous tile's pieces have all left by now).
            unsigned long long te = 0;
            if (DBG) te = __builtin_readcyclecounter();
            const int row0 = m0c + wm * 64 + (lane >> 2);
            srows = M - row0;
            const int64_t lane_off = (int64_t)row0 * N + n0c + wn * 96 + (lane & 3) * 8;
            sbase = hm_stride > 0 ? out + ((int64_t)((n0c + wn * 96) / 32) * hm_stride + row0) * 32 + (lane & 3) * 8 : out + lane_off;
            if (EPI == EPI_RESID) {                    // the residual arrives in store layout, all twelve loads in flight at once
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    const int ft = k >> 2, tt = (k >> 1) & 1, i = k & 1;
                    so[k] = bf16x8{};
                    if (tt * 32 + i * 16 < srows) so[k] = *(const bf16x8*)(resid + lane_off + (int64_t)(tt * 32 + i * 16) * N + ft * 32);
                }
            }
#pragma unroll
            for (int ft = 0; ft < 3; ++ft) {
                const float* bp = bias + n0c + wn * 96 + ft * 32;      // uniform address: scalar loads
                float bv[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) { const float lo = bp[i], hi = bp[16 + i]; bv[i] = hh ? hi : lo; }
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x4 v0 = {acc[ft][tt][8 * j] + bv[8 * j], acc[ft][tt][8 * j + 1] + bv[8 * j + 1], acc[ft][tt][8 * j + 2] + bv[8 * j + 2], acc[ft][tt][8 * j + 3] + bv[8 * j + 3]};
                        f32x4 v1 = {acc[ft][tt][8 * j + 4] + bv[8 * j + 4], acc[ft][tt][8 * j + 5] + bv[8 * j + 5], acc[ft][tt][8 * j + 6] + bv[8 * j + 6], acc[ft][tt][8 * j + 7] + bv[8 * j + 7]};
                        if (EPI == EPI_GELU) { v0 = gelu_poly4(v0); v1 = gelu_poly4(v1); }
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { o[e] = (bf16)v0[e]; o[4 + e] = (bf16)v1[e]; }
                        *(bf16x8*)(strip + r31 * 64 + (((2 * hh + j) ^ ((r31 >> 2) & 3)) * 16)) = o;
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int row = (lane >> 2) + 16 * i, u = lane & 3;
                        bf16x8 o = *(const bf16x8*)(strip + row * 64 + ((u ^ ((row >> 2) & 3)) * 16));
                        const int k = ft * 4 + tt * 2 + i;
                        if (EPI == EPI_RESID) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = (bf16)(bf2f(o[e]) + bf2f(so[k][e]));
                        }
                        so[k] = o;
                    }
                }
            }
            spend = 0;
            if (!defer) {
#pragma unroll
                for (int k = 0; k < 12; ++k) store_next();
            }
#pragma unroll
            for (int ft = 0; ft < 3; ++ft)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[ft][tt][e] = 0.f;
            if (DBG) tepi += __builtin_readcyclecounter() - te;
            ct = 0;
            if (++cj == ntiles) break;
            tile_of(cj, m0c, n0c);
        }
        if (defer) {
            s2 = s1;
            s1 = 0;
            if (spend < 12) { store_next(); s1 = 1; }
        }
        step(std::integral_constant<int, 0>{}, sn + xrel[1], sn + wrel[1], std::integral_constant<int, 2>{});
        advance();
        cs = ns;
    }
    while (spend < 12) store_next();                   // the last tile's pieces
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // the fragments read ahead for a step that does not exist have landed now; their registers stayed allocated until here
    asm volatile("" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(xf[0][0]), "+v"(xf[0][1]), "+v"(xf[1][0]), "+v"(xf[1][1]));
    if (DBG) {
        const unsigned long long tc3 = __builtin_readcyclecounter();
        if (lane == 0) {
            atomicAdd(dbg + 0, 1ull); atomicAdd(dbg + 1, tc1 - tc0); atomicAdd(dbg + 2, tc3 - tc1); atomicAdd(dbg + 3, tepi); atomicAdd(dbg + 4, twait);
            atomicAdd(dbg + 5, (unsigned long long)ntiles);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// attention: one workgroup = (sequence, head); K rows and V^T of the whole sequence staged in LDS once; each wave
// takes 16 queries per round (rounds of 64) and keeps its full score strip S^T[keys, 16] in registers
// (<= 32 key tiles): exact softmax, no online rescaling.
// ------------------------------------------------------------------------------------------------------------
template <int MAXT>   // max key tiles of 16 (sequence length <= 16*MAXT)
__global__ __launch_bounds__(256) void k_attention(const bf16* __restrict__ qkv, const int* __restrict__ cu,
                                                   bf16* __restrict__ ctx) {
    constexpr int KSTR = 80;                       // bytes per K row in LDS (64 + 16 pad: conflict-free b128 reads)
    constexpr int LP = MAXT * 16;
    constexpr int VSTR = LP * 2 + 8;               // bytes per V^T row
    char* ks = gsm;                                // [LP][KSTR]
    char* vt = gsm + LP * KSTR;                    // [DH][VSTR]
    const int b = blockIdx.y, head = blockIdx.x;
    const int t0 = cu[b], L = cu[b + 1] - t0;
    if (L <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, kg = lane >> 4;
    const int nt = (L + 15) >> 4;                  // key tiles in use
    const int npair = (nt + 1) >> 1;               // 32-key blocks for P.V
    const int rows_fill = npair * 32;              // rows touched by the MFMAs (<= LP)
    const int64_t rs = 3 * H;                      // qkv row stride (elements)

    // Q fragment of the first query block: issued BEFORE the K/V staging so its latency overlaps it
    // (B operand: query q0+fr, dims 8*kg..; 1/sqrt(32) is folded into Wq at load time)
    bf16x8 qf = {};
    if (w * 16 + fr < L) qf = *(const bf16x8*)(qkv + (int64_t)(t0 + w * 16 + fr) * rs + head * DH + kg * 8);

    // ---- K rows (16-B pieces) and V transposed into LDS, once per (sequence, head); rows >= L are zero -------
    for (int p0 = 0; p0 < rows_fill * 4; p0 += 512) {
        uint4 kv[2], vv[2];
#pragma unroll
        for (int u2 = 0; u2 < 2; ++u2) {           // both loads in flight before either is used
            const int p = p0 + u2 * 256 + tid;
            const int r = p >> 2, u = p & 3;
            kv[u2] = uint4{0u, 0u, 0u, 0u};
            vv[u2] = uint4{0u, 0u, 0u, 0u};
            if (p < rows_fill * 4 && r < L) {
                const bf16* base = qkv + (int64_t)(t0 + r) * rs + head * DH + u * 8;
                kv[u2] = *(const uint4*)(base + H);
                vv[u2] = *(const uint4*)(base + 2 * H);
            }
        }
#pragma unroll
        for (int u2 = 0; u2 < 2; ++u2) {
            const int p = p0 + u2 * 256 + tid;
            if (p < rows_fill * 4) {
                const int r = p >> 2, u = p & 3;
                *(uint4*)(ks + r * KSTR + u * 16) = kv[u2];
                const unsigned short* ve = (const unsigned short*)&vv[u2];
#pragma unroll
                for (int e = 0; e < 8; ++e) *(unsigned short*)(vt + (u * 8 + e) * VSTR + r * 2) = ve[e];
            }
        }
    }
    __syncthreads();

    for (int q0 = w * 16; q0 < L; q0 += 64) {      // each wave: 16 queries per round; no barriers below
        // prefetch the next round's Q fragment while this round computes
        bf16x8 qn = {};
        if (q0 + 64 + fr < L) qn = *(const bf16x8*)(qkv + (int64_t)(t0 + q0 + 64 + fr) * rs + head * DH + kg * 8);

        // ---- S^T tiles: lane holds query fr, keys 16*kt + 4*kg + i --------------------------------------------
        f32x4 st[MAXT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < MAXT; ++kt) {
            st[kt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (kt < nt) {
                const bf16x8 kf = *(const bf16x8*)(ks + (kt * 16 + fr) * KSTR + kg * 16);
                f32x4 sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {       // (a uniform branch that masks only the last key tile measured 6 % slower)
                    if (kt * 16 + kg * 4 + i >= L) sc[i] = -INFINITY;
                    mx = fmaxf(mx, sc[i]);
                }
                st[kt] = sc;
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < MAXT; ++kt) {
            if (kt < nt) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = __builtin_amdgcn_exp2f(st[kt][i] - mx);   // scores carry log2(e) (folded into Wq); 2^-inf = 0 for masked keys
                    st[kt][i] = p;
                    sum += p;
                }
            }
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);

        // ---- O = P . V : k-slot (kg, i, half) of block pb <-> key 32*pb + 16*half + 4*kg + i on BOTH operands --
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int pb = 0; pb < MAXT / 2; ++pb) {
            if (pb < npair) {
                bf16x8 pf;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pf[i] = (bf16)st[2 * pb][i];
                    pf[4 + i] = (2 * pb + 1 < nt) ? (bf16)st[2 * pb + 1][i] : (bf16)0.f;
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const char* vrow = vt + (dt * 16 + fr) * VSTR + (32 * pb + 4 * kg) * 2;
                    const bf16x4 v0 = *(const bf16x4*)(vrow);
                    const bf16x4 v1 = *(const bf16x4*)(vrow + 32);
                    bf16x8 vf;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { vf[i] = v0[i]; vf[4 + i] = v1[i]; }
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);   // O^T: V^T is the A operand
                }
            }
        }
        // D layout of O^T: row = dim 4*kg + i (+16*dt), col = query fr -- the lane's own query, whose row sum it already holds.
        // Lanes kg and kg ^ 1 (lane ^ 16) trade one 8-byte half so that each stores 8 consecutive dims: ONE 16-byte store per
        // lane and round (2-byte stores of the O layout made this kernel store-issue-bound).
        {
            const float rden = 1.0f / sum;
            union { bf16x4 v; int i[2]; } a0, a1, snd, rcv;
#pragma unroll
            for (int i = 0; i < 4; ++i) { a0.v[i] = (bf16)(o[0][i] * rden); a1.v[i] = (bf16)(o[1][i] * rden); }
            const bool odd = kg & 1;
            snd.v = odd ? a0.v : a1.v;                 // even kg keeps dims 4kg.. and gets 4(kg+1)..; odd keeps 16+4kg.. and gets 16+4(kg-1)..
            rcv.i[0] = __shfl_xor(snd.i[0], 16);
            rcv.i[1] = __shfl_xor(snd.i[1], 16);
            if (q0 + fr < L) {
                bf16x8 ov;
#pragma unroll
                for (int i = 0; i < 4; ++i) { ov[i] = odd ? rcv.v[i] : a0.v[i]; ov[4 + i] = odd ? a1.v[i] : rcv.v[i]; }
                *(bf16x8*)(ctx + (int64_t)(t0 + q0 + fr) * H + head * DH + (odd ? 16 + 4 * (kg - 1) : 4 * kg)) = ov;
            }
        }
        qf = qn;
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_attn3 -- attention, third form (round 3; default).  k_attention above is VALU-bound: 38 VALU per MFMA = ~19 wave
// instructions per score (per-element masks, a v_sub and a v_add per score, 4-lane shuffles, a branch skeleton per key tile),
// 0.073 of the bf16 roof.  Here the matrix pipe does everything that is linear and the VALU keeps only max, exp2 and the bf16
// pack (~3 instructions per score):
//   * S^T = K . Q^T with v_mfma_f32_32x32x16_bf16 (A = K rows from LDS, B = Q straight from global memory): lane (query =
//     lane & 31, half h) holds ITS query's scores against 16 keys of every 32-key tile -- row max / sum are chains over the
//     lane's own registers plus ONE exchange with lane ^ 32;
//   * TWO passes over the key tiles instead of a score strip in registers: pass 1 only takes the row maximum; pass 2 recomputes
//     S^T with the accumulator INITIALISED to -max (the C operand is a separate 16-register tuple), so the MFMA output is
//     already s - max and goes straight into v_exp_f32 -- no subtraction, and padding keys are masked by the same operand (the
//     last tile's C tuple carries -inf in the padded key slots) -- no compare / select per score;
//   * O^T = V^T . P^T: the accumulator layout of S^T (registers 8 s .. 8 s + 7 <-> keys {4h+e, 8+4h+e} of the 16-key block s) IS
//     a k-permuted B fragment, so P never leaves the registers; V^T is staged with the same key permutation;
//   * row sums with v_dot2_f32_bf16 on the PACKED bf16 pairs against (1, 1): half the adds, and the sum is taken over exactly
//     the rounded probabilities the P.V product uses;
//   * ~120 registers per wave: four waves per SIMD hide the LDS / exp latencies (k_attention holds the whole strip: one).
// Workgroup = (sequence, head), 4 waves; wave w takes the 32-query tiles w, w + 4, ...; the heads of a sequence share an XCD.
// KT = key tiles of 32 (L <= 32 KT).  What bounds it (measured, 8192 sequences of ~128 tokens, 0.90 ms per layer at four waves
// per SIMD, 1.13 at three): with the staging AND all but one key tile removed 0.45 ms remain -- 98k workgroups each paying a
// cold round trip for 64-byte row pieces, i.e. latency x occupancy, not arithmetic.  One workgroup per SEQUENCE walking its 12
// heads (every second 64-byte piece an L2 hit, one launch per sequence) measured SLOWER, 1.15 ms: the heads then wait for each
// other's loads in series.
// ------------------------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(256, 4) void k_attn3(const bf16* __restrict__ qkv, const int* __restrict__ cu, int batch, bf16* __restrict__ ctx, int ctx_tiled, long hm_stride) {
    constexpr int LP = KT * 32;
    constexpr int VSTR = LP * 2 + 16;              // bytes per V^T row (dim): +16 spreads the 32 dims over the banks
    constexpr int KBYTES = LP * 64;
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    char* ks = gsm;                                // [LP keys][4 units of 16 B], unit u of key r at u ^ ((r >> 2) & 3)
    char* vt = gsm + KBYTES;                       // [32 dims][VSTR]: V^T, keys in GEMM-slot order inside every 16-block
    // block -> (sequence, head): the 12 heads of a sequence run on ONE XCD (block x lands on XCD x % 8), so the 128-byte lines
    // that hold two neighbouring heads' 64-byte K / V / Q pieces are fetched from HBM once
    int b, head;
    {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        b = (idx / NH) * 8 + xcd;
        head = idx % NH;
    }
    if (b >= batch) return;
    const int t0 = cu[b], L = cu[b + 1] - t0;
    if (L <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c31 = lane & 31, hh = lane >> 5;
    const int nkt = (L + 31) >> 5;                 // key tiles (= query tiles) in use
    // qkv row-major [token][q | k | v of 12 heads x 32]: row stride 1152, a (token, head) piece is 64 B of a 2304-B row; or HEAD-MAJOR
    // (hm_stride > 0, written so by k_gemm3): [3 x 12 (part, head)][hm_stride tokens][32] -- the sequence's Q / K / V of one head are contiguous
    const int64_t rs = hm_stride > 0 ? DH : 3 * H;
    // accumulator init of the LAST key tile: 0 for real keys, -inf for the padding (register r <-> key 32 (nkt-1) + 8 (r>>2) + 4 hh + (r&3))
    f32x16 maskc;
#pragma unroll
    for (int r = 0; r < 16; ++r) maskc[r] = ((nkt - 1) * 32 + 8 * (r >> 2) + 4 * hh + (r & 3) < L) ? 0.f : -INFINITY;
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    const bf16* base = hm_stride > 0 ? qkv + ((int64_t)head * hm_stride + t0) * DH : qkv + (int64_t)t0 * rs + head * DH;
    const int64_t koff = hm_stride > 0 ? (int64_t)NH * hm_stride * DH : H, voff = 2 * koff;      // from a Q piece to the same token's K / V piece

    // Q fragments of this wave's first tile (B operand: query c31, dims 16 s + 8 hh ..): in flight across the staging
    bf16x8 qf[2] = {bf16x8{}, bf16x8{}};
    if (w < nkt) {
        const int q = min(w * 32 + c31, L - 1);
        qf[0] = __builtin_nontemporal_load((const bf16x8*)(base + (int64_t)q * rs + hh * 8));
        qf[1] = __builtin_nontemporal_load((const bf16x8*)(base + (int64_t)q * rs + 16 + hh * 8));
    }
    // ---- stage K (row-major, swizzled) and V^T (transposed, key-permuted); rows [L, 32 nkt) are zero ------------------------
    for (int p0 = 0; p0 < nkt * 128; p0 += 512) {  // 2 x 256 pieces of 16 B per iteration: both loads in flight
        uint4 kv[2], vv[2];
#pragma unroll
        for (int u2 = 0; u2 < 2; ++u2) {
            const int p = p0 + u2 * 256 + tid, r = p >> 2, u = p & 3;
            kv[u2] = uint4{0u, 0u, 0u, 0u};
            vv[u2] = uint4{0u, 0u, 0u, 0u};
            if (r < L) {
                { typedef u32 u32x4n __attribute__((ext_vector_type(4))); const u32x4n t4 = __builtin_nontemporal_load((const u32x4n*)(base + (int64_t)r * rs + koff + u * 8)); kv[u2] = uint4{t4[0], t4[1], t4[2], t4[3]}; }
                { typedef u32 u32x4n __attribute__((ext_vector_type(4))); const u32x4n t4 = __builtin_nontemporal_load((const u32x4n*)(base + (int64_t)r * rs + voff + u * 8)); vv[u2] = uint4{t4[0], t4[1], t4[2], t4[3]}; }
            }
        }
#pragma unroll
        for (int u2 = 0; u2 < 2; ++u2) {
            const int p = p0 + u2 * 256 + tid, r = p >> 2, u = p & 3;
            if (p < nkt * 128) {
                *(uint4*)(ks + r * 64 + ((u ^ ((r >> 2) & 3)) * 16)) = kv[u2];
                // key r -> slot inside its 16-block: keys 0-3 -> 0-3, 4-7 -> 8-11, 8-11 -> 4-7, 12-15 -> 12-15
                const int r16 = r & 15, slot = (r & ~15) + ((r16 & 3) | ((r16 & 4) << 1) | ((r16 & 8) >> 1));
                const unsigned short* ve = (const unsigned short*)&vv[u2];
#pragma unroll
                for (int e = 0; e < 8; ++e) *(unsigned short*)(vt + (u * 8 + e) * VSTR + slot * 2) = ve[e];
            }
        }
    }
    __syncthreads();

    // K fragment (key tile kt, k-step s = dims 16 s ..): lane (key c31, half hh) reads unit 2 s + hh of key 32 kt + c31
    const u32 kaddr = (u32)(uintptr_t)(__attribute__((address_space(3))) char*)ks + (u32)(c31 * 64);
    const u32 ksw = (u32)((c31 >> 2) & 3);
    auto kfrag = [&](int kt, int s2) -> bf16x8 {
        return *(const bf16x8*)((const __attribute__((address_space(3))) char*)(uintptr_t)(kaddr + (u32)(kt * 2048) + (((u32)(2 * s2 + hh) ^ ksw) * 16)));
    };

    for (int qt = w; qt < nkt; qt += 4) {
        bf16x8 qn[2] = {qf[0], qf[1]};
        if (qt + 4 < nkt) {                         // next tile's Q in flight while this tile computes
            const int q = min((qt + 4) * 32 + c31, L - 1);
            qn[0] = __builtin_nontemporal_load((const bf16x8*)(base + (int64_t)q * rs + hh * 8));
            qn[1] = __builtin_nontemporal_load((const bf16x8*)(base + (int64_t)q * rs + 16 + hh * 8));
        }
        // ---- pass 1: row maximum (log2 domain: log2(e) / sqrt(d) is folded into W_q).  Run-time loops over the key tiles (the
        // last one, whose accumulator starts from the padding mask, peeled): unrolled over KT hipcc kept every K fragment of
        // pass 1 alive for pass 2 (208 registers at KT = 8: two waves per SIMD instead of four). ---------------------------------
        float mx = -INFINITY;
        auto pass1 = [&](int kt, const f32x16& c0) {
            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(kt, 0), qf[0], c0, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(kt, 1), qf[1], acc, 0, 0, 0);
            float m3 = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
            for (int r = 3; r + 1 < 16; r += 2) m3 = fmaxf(fmaxf(m3, acc[r]), acc[r + 1]);
            mx = fmaxf(mx, fmaxf(m3, acc[15]));
        };
#pragma unroll 1
        for (int kt = 0; kt < nkt - 1; ++kt) pass1(kt, zero16);
        pass1(nkt - 1, maskc);
        mx = fmaxf(mx, __shfl_xor(mx, 32));        // the other half of this query's keys (every query has >= 1 real key: finite)
        f32x16 negm, negmm;                        // C operands of pass 2: -max, and -max with the padding mask
#pragma unroll
        for (int r = 0; r < 16; ++r) { negm[r] = -mx; negmm[r] = maskc[r] - mx; }
        // ---- pass 2: P = exp2(S - max) straight off the MFMA, O^T += V^T . P^T ---------------------------------------------------
        f32x16 ot;
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[r] = 0.f;
        float sum = 0.f;
        const bf16x2 one2 = {(bf16)1.0f, (bf16)1.0f};
        auto pass2 = [&](int kt, const f32x16& c0) {
            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(kt, 0), qf[0], c0, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(kt, 1), qf[1], acc, 0, 0, 0);
            u32x4 pu[2];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                bf16x2 pb;
                pb[0] = (bf16)__builtin_amdgcn_exp2f(acc[r]);           // exp2(-inf) = 0 for the padding
                pb[1] = (bf16)__builtin_amdgcn_exp2f(acc[r + 1]);
                sum = __builtin_amdgcn_fdot2_f32_bf16(pb, one2, sum, false);
                pu[r >> 3][(r & 7) >> 1] = __builtin_bit_cast(u32, pb);
            }
            // registers 8 s .. 8 s + 7 of the tile = the B fragment of k-step s (keys 32 kt + 16 s + {4hh+e, 8+4hh+e})
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 vf = *(const bf16x8*)(vt + c31 * VSTR + (kt * 32 + s2 * 16 + hh * 8) * 2);
                ot = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, pu[s2]), ot, 0, 0, 0);
            }
        };
#pragma unroll 1
        for (int kt = 0; kt < nkt - 1; ++kt) pass2(kt, negm);
        pass2(nkt - 1, negmm);
        sum += __shfl_xor(sum, 32);
        const float inv = __builtin_amdgcn_rcpf(sum);
        // ot[4 g + e] = dim 8 g + 4 hh + e of query c31.  Lane halves trade two 8-byte pieces so that each stores 16 consecutive
        // dims: half 0 keeps g = 0, 1 (dims 0-3, 8-11) and receives dims 4-7, 12-15; half 1 keeps g = 2, 3 and receives 16-19, 24-27.
        {
            union { bf16x4 v; int i[2]; } pc[4], rcv[2];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int e = 0; e < 4; ++e) pc[g4].v[e] = (bf16)(ot[4 * g4 + e] * inv);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int snd0 = hh ? pc[j].i[0] : pc[2 + j].i[0], snd1 = hh ? pc[j].i[1] : pc[2 + j].i[1];
                rcv[j].i[0] = __shfl_xor(snd0, 32);
                rcv[j].i[1] = __shfl_xor(snd1, 32);
            }
            const int q = qt * 32 + c31;
            if (q < L) {
                bf16x8 o0, o1;                     // dims [16 hh, +8) and [16 hh + 8, +8)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o0[e] = hh ? rcv[0].v[e] : pc[0].v[e];
                    o0[4 + e] = hh ? pc[2].v[e] : rcv[0].v[e];
                    o1[e] = hh ? rcv[1].v[e] : pc[1].v[e];
                    o1[4 + e] = hh ? pc[3].v[e] : rcv[1].v[e];
                }
                if (ctx_tiled) {
                    // ctx as the out-proj GEMM's 1-KiB blocks: (16-token block, head) -> [16 rows][4 units of 8 dims], unit u of row r at u ^ ((r >> 2) & 3)
                    const int64_t m = t0 + q;
                    const int r = (int)(m & 15), sw = tswz(r);
                    bf16* blk = ctx + ((m >> 4) * NH + head) * 512 + r * 32;
                    __builtin_nontemporal_store(o0, (bf16x8*)(blk + ((2 * hh) ^ sw) * 8));
                    __builtin_nontemporal_store(o1, (bf16x8*)(blk + ((2 * hh + 1) ^ sw) * 8));
                } else {
                    bf16* dst = ctx + (int64_t)(t0 + q) * H + head * DH + hh * 16;
                    __builtin_nontemporal_store(o0, (bf16x8*)dst);
                    __builtin_nontemporal_store(o1, (bf16x8*)(dst + 8));
                }
            }
        }
        qf[0] = qn[0];
        qf[1] = qn[1];
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_qkv_attn_small -- the interactive path's QKV projection AND attention in ONE launch (round 5).  One query per call is a chain of
// ~34 dependent launches of ~4.4 us each; the QKV GEMM (k_gemm_small) and the attention (k_attn3) of a layer are two of the five.
// Workgroup = (head, sequence), 4 waves:
//   phase 1: Q_h | K_h | V_h = LN?(x) . W[part * 384 + 32 head .. +32]^T + b for the sequence's tokens, exactly as k_gemm_small computes
//            them -- wave w holds the K quarter [96 w, +96) of the 3 x 32 weight rows as MFMA A fragments (18), token fragments come
//            straight from global memory (the next tile's requested before this tile's MFMAs), LNA normalises them on the way in with the
//            same two-pass statistics (the workgroup of head 0 leaves (mean, rstd) for the out-proj's residual fold), the four partial
//            sums meet in LDS and are added in the same order, bias first, one bf16 rounding -- but the results go into the LDS images
//            k_attn3 stages from global memory: K rows (swizzled 16-byte units), V^T (key-permuted), and Q rows in K's layout;
//   phase 2: k_attn3's two-pass attention, unchanged arithmetic, Q fragments read from LDS.
// Bit-identical to k_gemm_small + k_attn3 (same summation orders and rounding points).  Sequences up to 32 KT tokens.
// ------------------------------------------------------------------------------------------------------------
template <int KT>
struct QkvAttnCfg {
    static constexpr int LP = KT * 32, VSTR = LP * 2 + 16;
    static constexpr int RED_OFF = 0, RED_BYTES = 4 * 32 * 100 * 4;          // [K quarter][token][96 features + 4 pad] fp32
    static constexpr int LNP_OFF = RED_OFF + RED_BYTES, LNP_BYTES = 2 * 4 * 32 * 4;
    static constexpr int QS_OFF = LNP_OFF + LNP_BYTES, KS_OFF = QS_OFF + LP * 64, VT_OFF = KS_OFF + LP * 64;
    static constexpr int LDS_BYTES = VT_OFF + 32 * VSTR;
};

template <int KT, bool LNA>
__global__ __launch_bounds__(256) void k_qkv_attn_small(const bf16* __restrict__ A, const bf16* __restrict__ W, const float* __restrict__ bias,
                                                        const int* __restrict__ cu, int batch, const float* __restrict__ lng,
                                                        const float* __restrict__ lnb, float eps, float2* __restrict__ stats_out,
                                                        bf16* __restrict__ ctx) {
    using C = QkvAttnCfg<KT>;
    constexpr int NF = 6, VSTR = C::VSTR;
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const int head = blockIdx.x % NH, b = blockIdx.x / NH;
    if (b >= batch) return;
    const int t0g = cu[b], L = cu[b + 1] - t0g;
    if (L <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r31 = lane & 31, hh = lane >> 5;
    const int nkt = (L + 31) >> 5;
    float (*red)[32][100] = (float (*)[32][100])(gsm + C::RED_OFF);
    float (*lnp)[4][32] = (float (*)[4][32])(gsm + C::LNP_OFF);
    char* qs = gsm + C::QS_OFF;
    char* ks = gsm + C::KS_OFF;
    char* vt = gsm + C::VT_OFF;

    // ---- phase 1 ------------------------------------------------------------------------------------------------------------
    bf16x8 wf[3][NF];
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        const bf16* wr = W + (int64_t)(part * H + head * DH + r31) * H + w * 96 + hh * 8;
#pragma unroll
        for (int f = 0; f < NF; ++f) wf[part][f] = *(const bf16x8*)(wr + f * 16);
    }
    float lg[LNA ? NF : 1][8], lb[LNA ? NF : 1][8];
    if constexpr (LNA) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int k0 = w * 96 + f * 16 + hh * 8;
            const f32x4 g0 = *(const f32x4*)(lng + k0), g1 = *(const f32x4*)(lng + k0 + 4);
            const f32x4 b0 = *(const f32x4*)(lnb + k0), b1 = *(const f32x4*)(lnb + k0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { lg[f][e] = g0[e]; lg[f][4 + e] = g1[e]; lb[f][e] = b0[e]; lb[f][4 + e] = b1[e]; }
        }
    }
    const int et = tid >> 3, ef = (tid & 7) * 4;           // epilogue: token et of the tile, features ef .. ef + 3 of each part
    f32x4 bv[3];
#pragma unroll
    for (int part = 0; part < 3; ++part) bv[part] = *(const f32x4*)(bias + part * H + head * DH + ef);
    auto load_x = [&](int t0, bf16x8 (&x)[NF]) {
        const int tok = t0g + min(t0 + r31, L - 1);
        const bf16* xr = A + (int64_t)tok * H + w * 96 + hh * 8;
#pragma unroll
        for (int f = 0; f < NF; ++f) x[f] = *(const bf16x8*)(xr + f * 16);
    };
    bf16x8 xn[NF];
    load_x(0, xn);
    for (int t0 = 0; t0 < nkt * 32; t0 += 32) {
        bf16x8 xf[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) xf[f] = xn[f];
        if (t0 + 32 < L) load_x(t0 + 32, xn);
        if constexpr (LNA) {
            float v[NF][8], sm = 0.f;
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[f][e] = bf2f(xf[f][e]); sm += v[f][e]; }
            sm += __shfl_xor(sm, 32);
            if (hh == 0) lnp[0][w][r31] = sm;
            __syncthreads();
            const float mu = ((lnp[0][0][r31] + lnp[0][1][r31]) + (lnp[0][2][r31] + lnp[0][3][r31])) * (1.0f / H);
            float qv = 0.f;
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[f][e] - mu; qv = fmaf(d, d, qv); }
            qv += __shfl_xor(qv, 32);
            if (hh == 0) lnp[1][w][r31] = qv;
            __syncthreads();
            const float rs = rsqrtf(((lnp[1][0][r31] + lnp[1][1][r31]) + (lnp[1][2][r31] + lnp[1][3][r31])) * (1.0f / H) + eps);
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int e = 0; e < 8; ++e) xf[f][e] = (bf16)((v[f][e] - mu) * rs * lg[f][e] + lb[f][e]);
            if (stats_out && head == 0 && w == 0 && hh == 0 && t0 + r31 < L) stats_out[t0g + t0 + r31] = float2{mu, rs};
        }
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int f = 0; f < NF; ++f) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[part][f], xf[f], acc, 0, 0, 0);
            // acc[4 q + e] = feature 8 q + 4 hh + e (of this part's 32) of token r31, summed over this wave's K quarter
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(f32x4*)&red[w][r31][part * 32 + q * 8 + hh * 4] = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        }
        __syncthreads();
        {
            const int r = t0 + et;                         // row (token of the sequence) this thread finishes
            const bool live = r < L;
            bf16x4 o[3];
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                f32x4 v = bv[part];
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) v += *(const f32x4*)&red[ww][et][part * 32 + ef];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[part][e] = live ? (bf16)v[e] : (bf16)0.0f;     // rows [L, 32 nkt) are zero (as k_attn3 stages them)
            }
            const int u = ef >> 3, sw = (r >> 2) & 3;
            *(bf16x4*)(qs + r * 64 + ((u ^ sw) * 16) + (ef & 7) * 2) = o[0];
            *(bf16x4*)(ks + r * 64 + ((u ^ sw) * 16) + (ef & 7) * 2) = o[1];
            // key r -> slot inside its 16-block: keys 0-3 -> 0-3, 4-7 -> 8-11, 8-11 -> 4-7, 12-15 -> 12-15 (k_attn3)
            const int r16 = r & 15, slot = (r & ~15) + ((r16 & 3) | ((r16 & 4) << 1) | ((r16 & 8) >> 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) *(bf16*)(vt + (ef + e) * VSTR + slot * 2) = o[2][e];
        }
        __syncthreads();
    }

    // ---- phase 2: k_attn3's attention over the LDS images ---------------------------------------------------------------------
    const int c31 = r31;
    f32x16 maskc;
#pragma unroll
    for (int r = 0; r < 16; ++r) maskc[r] = ((nkt - 1) * 32 + 8 * (r >> 2) + 4 * hh + (r & 3) < L) ? 0.f : -INFINITY;
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    const u32 ksw = (u32)((c31 >> 2) & 3);
    const u32 kaddr = (u32)(uintptr_t)(__attribute__((address_space(3))) char*)ks + (u32)(c31 * 64);
    const u32 qaddr = (u32)(uintptr_t)(__attribute__((address_space(3))) char*)qs + (u32)(c31 * 64);
    auto kfrag = [&](int kt, int s2) -> bf16x8 {
        return *(const bf16x8*)((const __attribute__((address_space(3))) char*)(uintptr_t)(kaddr + (u32)(kt * 2048) + (((u32)(2 * s2 + hh) ^ ksw) * 16)));
    };
    auto qfrag = [&](int qt, int s2) -> bf16x8 {
        return *(const bf16x8*)((const __attribute__((address_space(3))) char*)(uintptr_t)(qaddr + (u32)(qt * 2048) + (((u32)(2 * s2 + hh) ^ ksw) * 16)));
    };
    for (int qt = w; qt < nkt; qt += 4) {
        const bf16x8 qf0 = qfrag(qt, 0), qf1 = qfrag(qt, 1);
        float mx = -INFINITY;
        auto pass1 = [&](int kt, const f32x16& c0) {
            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(kt, 0), qf0, c0, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(kt, 1), qf1, acc, 0, 0, 0);
            float m3 = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
            for (int r = 3; r + 1 < 16; r += 2) m3 = fmaxf(fmaxf(m3, acc[r]), acc[r + 1]);
            mx = fmaxf(mx, fmaxf(m3, acc[15]));
        };
#pragma unroll 1
        for (int kt = 0; kt < nkt - 1; ++kt) pass1(kt, zero16);
        pass1(nkt - 1, maskc);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        f32x16 negm, negmm;
#pragma unroll
        for (int r = 0; r < 16; ++r) { negm[r] = -mx; negmm[r] = maskc[r] - mx; }
        f32x16 ot;
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[r] = 0.f;
        float sum = 0.f;
        const bf16x2 one2 = {(bf16)1.0f, (bf16)1.0f};
        auto pass2 = [&](int kt, const f32x16& c0) {
            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(kt, 0), qf0, c0, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(kt, 1), qf1, acc, 0, 0, 0);
            u32x4 pu[2];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                bf16x2 pb;
                pb[0] = (bf16)__builtin_amdgcn_exp2f(acc[r]);
                pb[1] = (bf16)__builtin_amdgcn_exp2f(acc[r + 1]);
                sum = __builtin_amdgcn_fdot2_f32_bf16(pb, one2, sum, false);
                pu[r >> 3][(r & 7) >> 1] = __builtin_bit_cast(u32, pb);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 vf = *(const bf16x8*)(vt + c31 * VSTR + (kt * 32 + s2 * 16 + hh * 8) * 2);
                ot = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, pu[s2]), ot, 0, 0, 0);
            }
        };
#pragma unroll 1
        for (int kt = 0; kt < nkt - 1; ++kt) pass2(kt, negm);
        pass2(nkt - 1, negmm);
        sum += __shfl_xor(sum, 32);
        const float inv = __builtin_amdgcn_rcpf(sum);
        {
            union { bf16x4 v; int i[2]; } pc[4], rcv[2];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int e = 0; e < 4; ++e) pc[g4].v[e] = (bf16)(ot[4 * g4 + e] * inv);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int snd0 = hh ? pc[j].i[0] : pc[2 + j].i[0], snd1 = hh ? pc[j].i[1] : pc[2 + j].i[1];
                rcv[j].i[0] = __shfl_xor(snd0, 32);
                rcv[j].i[1] = __shfl_xor(snd1, 32);
            }
            const int q = qt * 32 + c31;
            if (q < L) {
                bf16x8 o0, o1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o0[e] = hh ? rcv[0].v[e] : pc[0].v[e];
                    o0[4 + e] = hh ? pc[2].v[e] : rcv[0].v[e];
                    o1[e] = hh ? rcv[1].v[e] : pc[1].v[e];
                    o1[4 + e] = hh ? pc[3].v[e] : rcv[1].v[e];
                }
                bf16* dst = ctx + (int64_t)(t0g + q) * H + head * DH + hh * 16;
                *(bf16x8*)dst = o0;
                *(bf16x8*)(dst + 8) = o1;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_attn4 -- k_attn3 made PERSISTENT (round 4).  k_attn3 starts one workgroup per (sequence, head): 98k of them for 8192 chunks, each
// paying a cold global round trip for its K / V / Q pieces before the first MFMA -- by the round-3 ablation 0.45 of its 0.82 ms per
// layer is that start-up, not arithmetic.  Here a workgroup walks a list of (sequence, head) items (item it -> XCD it & 7, as the
// grid map of k_attn3: the heads of a sequence stay on one XCD) and the NEXT item's K / V pieces and Q fragments are requested
// before the current item's MFMAs begin, landing in registers while it computes; they go into the (single) LDS image after the
// barrier that ends the current item.  Same arithmetic, same LDS layouts, same output as k_attn3 bit for bit.
// ------------------------------------------------------------------------------------------------------------
template <int KT, int OCC>
__global__ __launch_bounds__(256, OCC) void k_attn4(const bf16* __restrict__ qkv, const int* __restrict__ cu, int batch, bf16* __restrict__ ctx, int ctx_tiled, long hm_stride) {
    constexpr int LP = KT * 32;
    constexpr int VSTR = LP * 2 + 16;
    constexpr int KBYTES = LP * 64;
    constexpr int NR = KT / 4;                     // staging rounds of 512 sixteen-byte pieces (two per thread for K, two for V)
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    char* ks = gsm;
    char* vt = gsm + KBYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c31 = lane & 31, hh = lane >> 5;
    const int n_items = (batch + 7) / 8 * 8 * NH;
    const int64_t rs = hm_stride > 0 ? DH : 3 * H;
    const int64_t koff = hm_stride > 0 ? (int64_t)NH * hm_stride * DH : H, voff = 2 * koff;
    struct Item { int t0, L, nkt, head; const bf16* base; bool ok; };
    auto item = [&](int it) -> Item {
        Item c{0, 0, 0, 0, qkv, false};
        if (it >= n_items) return c;
        const int xcd = it & 7, idx = it >> 3;
        const int b = (idx / NH) * 8 + xcd;
        c.head = idx % NH;
        if (b >= batch) return c;
        c.t0 = cu[b];
        c.L = cu[b + 1] - c.t0;
        if (c.L <= 0) return c;
        c.nkt = (c.L + 31) >> 5;
        c.base = hm_stride > 0 ? qkv + ((int64_t)c.head * hm_stride + c.t0) * DH : qkv + (int64_t)c.t0 * rs + c.head * DH;
        c.ok = true;
        return c;
    };
    uint4 kv[NR][2], vv[NR][2];
    auto request = [&](const Item& c, bf16x8 (&qf)[2]) {       // global loads of an item's K / V pieces and this wave's first Q fragments
        qf[0] = bf16x8{}; qf[1] = bf16x8{};
        if (w < c.nkt) {
            const int q = min(w * 32 + c31, c.L - 1);
            qf[0] = *(const bf16x8*)(c.base + (int64_t)q * rs + hh * 8);
            qf[1] = *(const bf16x8*)(c.base + (int64_t)q * rs + 16 + hh * 8);
        }
#pragma unroll
        for (int rd = 0; rd < NR; ++rd)
#pragma unroll
            for (int u2 = 0; u2 < 2; ++u2) {
                const int p = rd * 512 + u2 * 256 + tid, r = p >> 2, u = p & 3;
                kv[rd][u2] = uint4{0u, 0u, 0u, 0u};
                vv[rd][u2] = uint4{0u, 0u, 0u, 0u};
                if (r < c.L) {
                    kv[rd][u2] = *(const uint4*)(c.base + (int64_t)r * rs + koff + u * 8);
                    vv[rd][u2] = *(const uint4*)(c.base + (int64_t)r * rs + voff + u * 8);
                }
            }
    };
    auto stage = [&](const Item& c) {                           // registers -> the LDS images (K row-major swizzled, V^T key-permuted)
#pragma unroll
        for (int rd = 0; rd < NR; ++rd)
#pragma unroll
            for (int u2 = 0; u2 < 2; ++u2) {
                const int p = rd * 512 + u2 * 256 + tid, r = p >> 2, u = p & 3;
                if (p < c.nkt * 128) {
                    *(uint4*)(ks + r * 64 + ((u ^ ((r >> 2) & 3)) * 16)) = kv[rd][u2];
                    const int r16 = r & 15, slot = (r & ~15) + ((r16 & 3) | ((r16 & 4) << 1) | ((r16 & 8) >> 1));
                    const unsigned short* ve = (const unsigned short*)&vv[rd][u2];
#pragma unroll
                    for (int e = 0; e < 8; ++e) *(unsigned short*)(vt + (u * 8 + e) * VSTR + slot * 2) = ve[e];
                }
            }
    };
    const u32 kaddr = (u32)(uintptr_t)(__attribute__((address_space(3))) char*)ks + (u32)(c31 * 64);
    const u32 ksw = (u32)((c31 >> 2) & 3);
    auto kfrag = [&](int kt, int s2) -> bf16x8 {
        return *(const bf16x8*)((const __attribute__((address_space(3))) char*)(uintptr_t)(kaddr + (u32)(kt * 2048) + (((u32)(2 * s2 + hh) ^ ksw) * 16)));
    };
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

    // OCC == 4 (126 registers, k_attn3's occupancy): no room for the next item's pieces in registers -- they are pulled into this XCD's L2
    // instead (one 128-byte line per lane, value discarded), and requested for real when the item starts
    auto touch_l2 = [&](const Item& c) {
        const int n_lines = c.L * 64 / 128;                       // (head-major planes: an item's Q, K and V are three contiguous blocks)
        if (hm_stride > 0 && tid < 3 * n_lines) {
            const int part = tid / n_lines, ln = tid - part * n_lines;
            const char* p = (const char*)(c.base + (int64_t)part * koff) + ln * 128;
            u32 sink;
            asm volatile("global_load_dword %0, %1, off" : "=v"(sink) : "v"(p) : "memory");
        }
    };
    int it = blockIdx.x;
    Item cur = item(it);
    bf16x8 qf[2], qnx[2];
    request(cur, qf);
    for (;;) {
        if (cur.ok) stage(cur);
        __syncthreads();
        const int itn = it + (int)gridDim.x;
        const Item nxt = item(itn);
        qnx[0] = bf16x8{}; qnx[1] = bf16x8{};
        if (OCC == 4) { if (nxt.ok) touch_l2(nxt); }
        else if (nxt.ok) request(nxt, qnx);                      // in flight while this item computes
        if (cur.ok) {
            const int L = cur.L, nkt = cur.nkt, t0 = cur.t0, head = cur.head;
            const bf16* base = cur.base;
            f32x16 maskc;
#pragma unroll
            for (int r = 0; r < 16; ++r) maskc[r] = ((nkt - 1) * 32 + 8 * (r >> 2) + 4 * hh + (r & 3) < L) ? 0.f : -INFINITY;
            for (int qt = w; qt < nkt; qt += 4) {
                bf16x8 qn[2] = {qf[0], qf[1]};
                if (qt + 4 < nkt) {
                    const int q = min((qt + 4) * 32 + c31, L - 1);
                    qn[0] = *(const bf16x8*)(base + (int64_t)q * rs + hh * 8);
                    qn[1] = *(const bf16x8*)(base + (int64_t)q * rs + 16 + hh * 8);
                }
                float mx = -INFINITY;
                auto pass1 = [&](int kt, const f32x16& c0) {
                    f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(kt, 0), qf[0], c0, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(kt, 1), qf[1], acc, 0, 0, 0);
                    float m3 = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
                    for (int r = 3; r + 1 < 16; r += 2) m3 = fmaxf(fmaxf(m3, acc[r]), acc[r + 1]);
                    mx = fmaxf(mx, fmaxf(m3, acc[15]));
                };
#pragma unroll 1
                for (int kt = 0; kt < nkt - 1; ++kt) pass1(kt, zero16);
                pass1(nkt - 1, maskc);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                f32x16 negm, negmm;
#pragma unroll
                for (int r = 0; r < 16; ++r) { negm[r] = -mx; negmm[r] = maskc[r] - mx; }
                f32x16 ot;
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[r] = 0.f;
                float sum = 0.f;
                const bf16x2 one2 = {(bf16)1.0f, (bf16)1.0f};
                auto pass2 = [&](int kt, const f32x16& c0) {
                    f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(kt, 0), qf[0], c0, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(kt, 1), qf[1], acc, 0, 0, 0);
                    u32x4 pu[2];
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        bf16x2 pb;
                        pb[0] = (bf16)__builtin_amdgcn_exp2f(acc[r]);
                        pb[1] = (bf16)__builtin_amdgcn_exp2f(acc[r + 1]);
                        sum = __builtin_amdgcn_fdot2_f32_bf16(pb, one2, sum, false);
                        pu[r >> 3][(r & 7) >> 1] = __builtin_bit_cast(u32, pb);
                    }
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8 vf = *(const bf16x8*)(vt + c31 * VSTR + (kt * 32 + s2 * 16 + hh * 8) * 2);
                        ot = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, pu[s2]), ot, 0, 0, 0);
                    }
                };
#pragma unroll 1
                for (int kt = 0; kt < nkt - 1; ++kt) pass2(kt, negm);
                pass2(nkt - 1, negmm);
                sum += __shfl_xor(sum, 32);
                const float inv = __builtin_amdgcn_rcpf(sum);
                {
                    union { bf16x4 v; int i[2]; } pc[4], rcv[2];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                        for (int e = 0; e < 4; ++e) pc[g4].v[e] = (bf16)(ot[4 * g4 + e] * inv);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int snd0 = hh ? pc[j].i[0] : pc[2 + j].i[0], snd1 = hh ? pc[j].i[1] : pc[2 + j].i[1];
                        rcv[j].i[0] = __shfl_xor(snd0, 32);
                        rcv[j].i[1] = __shfl_xor(snd1, 32);
                    }
                    const int q = qt * 32 + c31;
                    if (q < L) {
                        bf16x8 o0, o1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o0[e] = hh ? rcv[0].v[e] : pc[0].v[e];
                            o0[4 + e] = hh ? pc[2].v[e] : rcv[0].v[e];
                            o1[e] = hh ? rcv[1].v[e] : pc[1].v[e];
                            o1[4 + e] = hh ? pc[3].v[e] : rcv[1].v[e];
                        }
                        if (ctx_tiled) {
                            const int64_t m = t0 + q;
                            const int r = (int)(m & 15), sw = tswz(r);
                            bf16* blk = ctx + ((m >> 4) * NH + head) * 512 + r * 32;
                            *(bf16x8*)(blk + ((2 * hh) ^ sw) * 8) = o0;
                            *(bf16x8*)(blk + ((2 * hh + 1) ^ sw) * 8) = o1;
                        } else {
                            bf16* dst = ctx + (int64_t)(t0 + q) * H + head * DH + hh * 16;
                            *(bf16x8*)dst = o0;
                            *(bf16x8*)(dst + 8) = o1;
                        }
                    }
                }
                qf[0] = qn[0];
                qf[1] = qn[1];
            }
        }
        if (itn >= n_items) break;
        __syncthreads();                                         // everybody is done reading this item's LDS images
        cur = nxt; it = itn;
        if (OCC == 4) request(cur, qf);
        else { qf[0] = qnx[0]; qf[1] = qnx[1]; }
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_qa -- the bulk path's QKV projection AND attention in ONE launch (round 6).  Per layer k_gemm3 wrote 2.4 GB of
// Q | K | V that k_attn3 read back 1 ms later (14.4 GB written + 14.6 GB fetched per forward, 98k cold workgroups each waiting for its
// 64-byte row pieces): here Q, K and V never leave the CU.
//   * Workgroup = one ITEM: up to 8 token tiles of 32 (wave w <-> tile slot w) from consecutive WHOLE sequences (k_qa_items packs them
//     greedily: a 128-token sequence fills 4 slots, two of them share a workgroup and its weight stream).  8 waves, two per SIMD.
//   * A wave keeps its 32 token rows as 24 MFMA B fragments in registers for the whole kernel (96, as k_ffn3 does).
//   * The layer's Wq | Wk | Wv arrive as the k_pack_qa stream -- per head 72 fragments of 1 KiB in consumption order (k-step major,
//     then Q / K / V), 3 slabs of 24 -- through a 3-slab LDS-DMA ring (72 KiB); wave w moves pieces w, w + 8, w + 16 of a slab; ONE
//     barrier per slab (slab g landed for everybody; everybody is done with slab g - 1, whose slot takes slab g + 2).
//   * Per head: 72 v_mfma_f32_32x32x16_bf16 (A = weight fragment from the ring, B = token fragment) into three accumulators
//     [32 features of Q / K / V][32 tokens]; bias + bf16 rounding exactly as k_gemm3's epilogue; Q goes back into the wave's registers
//     as the two B fragments of S^T = K Q^T (one exchange with lane ^ 32), K rows and V^T go into the LDS images k_attn3 stages from
//     global memory -- same layout, keys of slot s at rows [32 s, +32) -- one barrier, then k_attn3's two-pass attention of the wave's
//     query tile against the key tiles of ITS sequence, unchanged arithmetic.  The images are double-buffered over the heads (one
//     barrier per head; a wave may run ahead into the next head's projection).
// Bit-identical to k_gemm3 + k_attn3 (same summation orders and rounding points): tests compare the two paths exactly.
// Wait counting (vmcnt counts this wave's LDS-DMA instructions and its ctx stores, in order): at the barrier of slab g the instructions
// of slab g + 1 (3) may stay in flight; the two ctx stores of the previous head are younger still, so vmcnt(3) can only over-wait.
// ------------------------------------------------------------------------------------------------------------
namespace qa {
constexpr int SLOTS = 8;
constexpr int SLAB = 24 * 1024, NSLAB = 3, RING = NSLAB * SLAB;
constexpr int HEAD_SLABS = 3, LAYER_SLABS = NH * HEAD_SLABS;
constexpr int KS_BYTES = SLOTS * 32 * 64;
constexpr int VSTR = SLOTS * 32 * 2 + 16;
constexpr int VT_BYTES = 32 * VSTR;
constexpr int IMG = KS_BYTES + VT_BYTES;
constexpr int BIAS_OFF = RING + 2 * IMG;          // the layer's 1152 Q | K | V biases, fp32: the epilogue of every head reads 96 of them
constexpr int LDS_BYTES = BIAS_OFF + 3 * H * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
}  // namespace qa

// the layer's [1152][384] Q | K | V weight rows -> k_qa's stream: [head][fragment f = 3 * kstep + part][lane] 16-byte units, fragment
// (part, kstep) lane (r31, hh) = row part * 384 + 32 head + r31, k [16 kstep + 8 hh, +8).  One thread per unit.
__global__ void k_pack_qa(const bf16* __restrict__ wqkv, bf16* __restrict__ dst) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= 3 * H * H / 8) return;
    const int lane = d & 63, f = (d >> 6) % 72, head = (d >> 6) / 72;
    const int ks = f / 3, part = f % 3, r31 = lane & 31, hh = lane >> 5;
    *(bf16x8*)(dst + (int64_t)d * 8) = *(const bf16x8*)(wqkv + (int64_t)(part * H + head * DH + r31) * H + ks * 16 + hh * 8);
}

// items[i] = (first sequence, sequences) of workgroup i of k_qa; *n_items = their number.  A thread packs a run of >= 32 consecutive
// sequences greedily (a new item when the next sequence's tiles no longer fit the 8 slots); runs are independent, so the last item of
// a run may stay part empty (1 item in ~18).  One workgroup; batch <= 65536.
__global__ __launch_bounds__(1024) void k_qa_items(const int* __restrict__ cu, int batch, int2* __restrict__ items, int* __restrict__ n_items) {
    __shared__ int cnt[1024];
    const int t = threadIdx.x;
    const int per = max(32, (batch + 1023) / 1024);
    const int lo = min(batch, t * per), hi = min(batch, lo + per);
    auto walk = [&](int base, bool emit) {
        int n = 0, start = lo, tiles = 0;
        for (int b = lo; b < hi; ++b) {
            const int nt = (cu[b + 1] - cu[b] + 31) >> 5;
            if (tiles > 0 && tiles + nt > qa::SLOTS) {
                if (emit) items[base + n] = int2{start, b - start};
                ++n; start = b; tiles = 0;
            }
            tiles += nt;
        }
        if (tiles > 0) { if (emit) items[base + n] = int2{start, hi - start}; ++n; }
        return n;
    };
    const int mine = walk(0, false);
    cnt[t] = mine;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {          // inclusive scan
        const int v = t >= o ? cnt[t - o] : 0;
        __syncthreads();
        cnt[t] += v;
        __syncthreads();
    }
    walk(cnt[t] - mine, true);
    if (t == 1023) *n_items = cnt[1023];
}

template <bool DBG>      // DBG: cycle counters of the phases into `dbg` (RMU_QA_CLK=1, debug builds)
__global__ __launch_bounds__(512) void k_qa(const bf16* __restrict__ x, int x_tiled, const bf16* __restrict__ wstream, const float* __restrict__ bias,
                                            const int* __restrict__ cu, const int2* __restrict__ items, const int* __restrict__ n_items,
                                            bf16* __restrict__ ctx, int ctx_tiled, int dflags, unsigned long long* dbg) {
    // dflags (RMU_QA_DBG, timing ablations only -- results are wrong with any bit set): 1 no attention, 2 no projection MFMAs / fragment reads,
    // 4 no epilogue (bias, pack, image writes), 8 no weight DMA
    using namespace qa;
    using ffn::static_for; using ffn::sgpr_ptr;
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    if ((int)blockIdx.x >= *n_items) return;
    const int2 item = items[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c31 = lane & 31, hh = lane >> 5;
    // Waves w and w + 4 share a SIMD (round-robin placement): `half` 0 = waves 0-3, 1 = waves 4-7.  Tile slots alternate between the
    // halves (slot i -> half i & 1), so both halves have work whenever the item has two tiles or more.
    const int half = w >> 2, myslot = 2 * (w & 3) + half;
    // ---- this wave's tile: sequence b, query tile qt of it, the sequence's first slot s0 / length L / first token t0 ------------------
    int b = -1, qt = 0, s0 = 0, L = 0, t0 = 0;
    {
        int slot = 0;
        for (int i = 0; i < item.y; ++i) {
            const int a = cu[item.x + i], l = cu[item.x + i + 1] - a, nt = (l + 31) >> 5;
            if (myslot >= slot && myslot < slot + nt) { b = item.x + i; qt = myslot - slot; s0 = slot; L = l; t0 = a; }
            slot += nt;
        }
    }
    const bool active = b >= 0;                    // (wave-uniform) a wave without a tile still moves its ring pieces and meets the barriers
    const int nkt = (L + 31) >> 5;

    // ---- the weight stream: ring slab index g = 3 * segment + j; segment sg projects head sg >> 1 (every head passes TWICE: once per half)
    const u32 lane16 = (u32)lane * 16;
    auto issue = [&](int g, int slot) {            // stream slab g -> ring slot `slot` (= g % 3, a literal at every call site)
        const int src_slab = ((g / 3) >> 1) * HEAD_SLABS + g % 3;
        const char* src = (const char*)wstream + (size_t)src_slab * SLAB + (size_t)w * 1024;
        char* dst = gsm + slot * SLAB + w * 1024;
        if (dflags & 8) return;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            u32 o = lane16;
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sgpr_ptr(src + i * 8192) + o),
                                             (__attribute__((address_space(3))) void*)(dst + i * 8192), 16, 0, 0);
        }
    };
    issue(0, 0);
    issue(1, 1);
    float* bl = (float*)(gsm + BIAS_OFF);
    for (int i = tid; i < 3 * H; i += 512) bl[i] = bias[i];      // (six dependent scalar-load round trips per head and wave when read from global memory)

    // ---- token rows as B fragments: k-step ks = features [16 ks + 8 hh, +8) of token (tile row c31); rows past the sequence repeat its last
    bf16x8 xf[24];
    {
        const int tok = t0 + min(qt * 32 + c31, max(L - 1, 0));
        if (x_tiled) {                             // k_ffn3's tiled store: 1-KiB blocks (token / 16, feature / 32), unit ^ tswz(row)
            const int r = tok & 15, sw = tswz(r);
            const bf16* blk = x + (int64_t)(tok >> 4) * (NH * 512) + r * 32;
#pragma unroll
            for (int ks = 0; ks < 24; ++ks) xf[ks] = *(const bf16x8*)(blk + (ks >> 1) * 512 + ((((ks & 1) * 2 + hh) ^ sw) * 8));
        } else {
            const bf16* row = x + (int64_t)tok * H + hh * 8;
#pragma unroll
            for (int ks = 0; ks < 24; ++ks) xf[ks] = *(const bf16x8*)(row + ks * 16);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // rows, biases and the first two slabs (own pieces) landed: the counted waits below start from an empty queue

    const u32 ring_addr = (u32)(uintptr_t)(__attribute__((address_space(3))) char*)gsm;
    // attention constants of this wave's sequence (k_attn3).  The padding mask of the LAST key tile (0 for real keys, -inf for padding;
    // register r <-> key 32 (nkt - 1) + 8 (r >> 2) + 4 hh + (r & 3)) is rebuilt from `mlim` where it is needed: 16 registers less to carry
    const int mlim = L - (nkt - 1) * 32 - 4 * hh;
    [[maybe_unused]] auto mask_into = [&](f32x16& m, float base) {
        int lim = mlim;
        asm volatile("" : "+v"(lim));              // (opaque: hipcc otherwise hoists the 16 selects out of the segment loop and spills them)
#pragma unroll
        for (int r = 0; r < 16; ++r) m[r] = (8 * (r >> 2) + (r & 3) < lim) ? base : -INFINITY;
    };
    const u32 ksw = (u32)((c31 >> 2) & 3);
    // key r of the tile -> slot inside its 16-block: keys 0-3 -> 0-3, 4-7 -> 8-11, 8-11 -> 4-7, 12-15 -> 12-15 (k_attn3's V^T order)
    const int r16 = c31 & 15, vslot = myslot * 32 + (c31 & ~15) + ((r16 & 3) | ((r16 & 4) << 1) | ((r16 & 8) >> 1));
    const int n1 = (nkt + 1) >> 1;                 // key tiles of the first pass-2 chunk

    unsigned long long c_bar = 0, c_proj = 0, c_epi = 0, c_p1 = 0, c_p2a = 0, c_p2b = 0, c_all = 0, tk = 0;
    auto tick = [&](unsigned long long& acc_) { if (DBG) { const unsigned long long n_ = __builtin_readcyclecounter(); acc_ += n_ - tk; tk = n_; } };
    if (DBG) { tk = __builtin_readcyclecounter(); c_all = tk; }
    // one sub-step boundary: stream slab g landed for everybody, everybody is done with slab g - 1, whose slot takes slab g + 2
    constexpr int STREAM = 2 * LAYER_SLABS;        // 72 slabs: every head twice
    auto boundary = [&](int g, auto jc) {
        constexpr int j = decltype(jc)::value;
        if (g + 1 < STREAM) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (g + 2 < STREAM) issue(g + 2, (j + 2) % NSLAB);
        tick(c_bar);
    };

    // Q fragments of the heads this wave has projected and not yet attended to: even heads in qe, odd heads in qo (half 0 attends head h
    // three segments after projecting it, with head h + 1's projection in between)
    bf16x8 qe0 = bf16x8{}, qe1 = bf16x8{}, qo0 = bf16x8{}, qo1 = bf16x8{};

    // ---- 26 segments of three sub-steps.  Segment sg: half (sg & 1) PROJECTS head sg >> 1 (sg < 24) while the other half ATTENDS -- half 1
    // to head (sg - 2) / 2 (projected in segments sg - 2 and sg - 1), half 0 to head (sg - 3) / 2: on every SIMD a matrix-bound wave runs
    // beside a VALU-bound one (both waves of a SIMD in the same phase -- the lockstep form of this kernel -- measured 2.31 ms per layer
    // against 1.90 for k_gemm3 + k_attn3: nothing overlapped).  K rows / V^T of head h: written in segments 2h and 2h + 1 into image buffer
    // h & 1, read in segments 2h + 2 and 2h + 3; the buffer's next writer (head h + 2) starts in segment 2h + 4.
#pragma unroll 1
    for (int sg = 0; sg < 2 * NH + 2; ++sg) {
        if ((sg & 1) == half) {
            // ================= projection of head hp ======================================================================================
            const int hp = sg >> 1;
            const bool work = active && sg < 2 * NH;
            f32x16 acc[3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
            static_for<HEAD_SLABS>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                boundary(sg * HEAD_SLABS + j, jc);
                if (work && !(dflags & 2)) {
                    u32 base = ring_addr + (u32)(j * SLAB) + lane16;
                    asm volatile("" : "+v"(base));          // ONE address register per slab: the 24 fragments are immediate offsets (hipcc otherwise keeps 24 addresses -- spilled)
                    bf16x8 fo[4];
                    static_for<4>([&, base](auto nc) { (void)base; ffn::ds_read16<decltype(nc)::value * 1024>(fo[decltype(nc)::value], base); });
                    static_for<24>([&, base](auto nc) {
                        constexpr int n = decltype(nc)::value;
                        (void)base;
                        if constexpr (n % 2 == 0)
                            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fo[n % 4]), "+v"(fo[(n + 1) % 4]) : "n"(n + 4 <= 24 ? 2 : 24 - 2 - n));
                        acc[n % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo[n % 4], xf[8 * j + n / 3], acc[n % 3], 0, 0, 0);
                        if constexpr (n + 4 < 24) ffn::ds_read16<(n + 4) * 1024>(fo[n % 4], base);
                        else asm volatile("" : "+v"(fo[n % 4]));
                    });
                }
                tick(c_proj);
            });
            if (work && !(dflags & 4)) {
                char* ks = gsm + RING + (hp & 1) * IMG;
                char* vt = ks + KS_BYTES;
                // ---- acc[p][4 q + e] = feature 8 q + 4 hh + e of part p for token c31: + bias, one bf16 rounding (k_gemm3's epilogue) ----------
                union { bf16x4 v; int i[2]; } pq[4], pk4[4], pv[4];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const float* bp = bl + p * H + hp * DH + 4 * hh;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 bv = *(const f32x4*)(bp + 8 * q);
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[p][4 * q + e] + bv[e]);
                        if (p == 0) pq[q].v = o; else if (p == 1) pk4[q].v = o; else pv[q].v = o;
                    }
                }
                // Q: the B fragment of k-step s2 holds dims [16 s2 + 8 hh, +8) = pieces (q = 2 s2 + hh) of BOTH lane halves: elements 0-3 from
                // the half-0 lane of this token, 4-7 from its half-1 lane.  Half 0 sends pieces 1, 3 and keeps 0, 2; half 1 the other way round.
                {
                    union { bf16x4 v; int i[2]; } rc[2];
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const int snd0 = hh ? pq[2 * s2].i[0] : pq[2 * s2 + 1].i[0], snd1 = hh ? pq[2 * s2].i[1] : pq[2 * s2 + 1].i[1];
                        rc[s2].i[0] = __shfl_xor(snd0, 32);
                        rc[s2].i[1] = __shfl_xor(snd1, 32);
                    }
                    bf16x8 q0, q1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        q0[e] = hh ? rc[0].v[e] : pq[0].v[e];
                        q0[4 + e] = hh ? pq[1].v[e] : rc[0].v[e];
                        q1[e] = hh ? rc[1].v[e] : pq[2].v[e];
                        q1[4 + e] = hh ? pq[3].v[e] : rc[1].v[e];
                    }
                    if (hp & 1) { qo0 = q0; qo1 = q1; } else { qe0 = q0; qe1 = q1; }
                }
                // K row (slot myslot, key c31): 16-byte unit q holds dims [8 q, +8), this lane's piece at byte 8 hh of it; unit ^ ((row >> 2) & 3)
                {
                    char* krow = ks + (myslot * 32 + c31) * 64 + hh * 8;
#pragma unroll
                    for (int q = 0; q < 4; ++q) *(bf16x4*)(krow + (((u32)q ^ ksw) * 16)) = pk4[q].v;
                }
                // V^T: dim 8 q + 4 hh + e, key slot vslot
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) *(bf16*)(vt + (8 * q + 4 * hh + e) * VSTR + vslot * 2) = pv[q].v[e];
            }
            tick(c_epi);
        } else {
            // ================= attention to head ha: query tile qt against the key tiles [s0, s0 + nkt) of its sequence (k_attn3's two passes) ===
            const int ha = half ? (sg - 2) >> 1 : (sg - 3) >> 1;
            const bool work = active && !(dflags & 1) && (half ? sg >= 2 : sg >= 3) && ha < NH;
            char* ks = gsm + RING + (ha & 1) * IMG;
            char* vt = ks + KS_BYTES;
            const bf16x8 qf0 = (ha & 1) ? qo0 : qe0, qf1 = (ha & 1) ? qo1 : qe1;
            // K / V^T fragments are read with inline-asm ds_read_b128 and counted lgkmcnt waits, one key tile AHEAD of the MFMAs that eat them:
            // left to hipcc every S^T tile was read -> wait -> MFMA -> read -> wait -> MFMA, two exposed LDS round trips per tile and pass on a
            // wave that has no partner in the same phase to hide them (the first ping-pong form ran 2.87 ms per layer).
            const u32 kaddr = (u32)(uintptr_t)(__attribute__((address_space(3))) char*)ks + (u32)(s0 * 2048 + c31 * 64);
            const u32 koff0 = ((u32)hh ^ ksw) * 16, koff1 = ((u32)(2 + hh) ^ ksw) * 16;
            const u32 vaddr = (u32)(uintptr_t)(__attribute__((address_space(3))) char*)vt + (u32)(c31 * VSTR + (s0 * 32 + hh * 8) * 2);
            auto kread = [&](bf16x8& k0, bf16x8& k1, int kt) {      // key tile kt (clamped: a harmless re-read keeps the wait counts uniform)
                const u32 base = kaddr + (u32)(min(kt, nkt - 1) * 2048);
                asm volatile("ds_read_b128 %0, %1" : "=v"(k0) : "v"(base + koff0));
                asm volatile("ds_read_b128 %0, %1" : "=v"(k1) : "v"(base + koff1));
            };
            auto vread = [&](bf16x8& v0, bf16x8& v1, int kt) {      // V^T fragments of key tile kt: keys [32 kt + 16 s2 + 8 hh, +8) of dim c31
                const u32 base = vaddr + (u32)(min(kt, nkt - 1) * 64);
                asm volatile("ds_read_b128 %0, %1" : "=v"(v0) : "v"(base));
                asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(v1) : "v"(base));
            };
            // the padded key slots of the LAST key tile -> -inf (what k_attn3's masked C operand makes of them: -inf + finite = -inf)
            auto pad_fix = [&](f32x16& a) {
                int lim = mlim;
                asm volatile("" : "+v"(lim));
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] = (8 * (r >> 2) + (r & 3) < lim) ? a[r] : -INFINITY;
            };
            float mx = -INFINITY, sum = 0.f;
            f32x16 negm, ot;
            const bf16x2 one2 = {(bf16)1.0f, (bf16)1.0f};
            // pass 2 over the key tiles [lo, hi): P = exp2(S - max) straight off the MFMA, O^T += V^T . P^T.  Per tile t: first S^T MFMA of tile
            // t + 1 | the 16 exp2 of tile t | second MFMA of t + 1, K fragments of t + 2 requested | pack + row sum | P.V MFMAs of t, V^T of t + 1 requested
            auto pass2 = [&](int lo, int hi) {
                if (lo >= hi) return;
                bf16x8 kn0, kn1, v0, v1;
                kread(kn0, kn1, lo);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kn0), "+v"(kn1));
                f32x16 a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kn0, qf0, negm, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kn1, qf1, a, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                kread(kn0, kn1, lo + 1);
                vread(v0, v1, lo);
#pragma unroll 2
                for (int kt = lo; kt < hi; ++kt) {
                    const bool more = kt + 1 < hi;
                    f32x16 an = a;
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(kn0), "+v"(kn1));          // K(t + 1) landed; V^T(t) may still be on its way
                    if (more) an = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kn0, qf0, negm, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (kt == nkt - 1) pad_fix(a);
                    float ex[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) ex[r] = __builtin_amdgcn_exp2f(a[r]);       // exp2(-inf) = 0 for the padding
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) an = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kn1, qf1, an, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    kread(kn0, kn1, kt + 2);
                    u32x4 pu[2];
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        bf16x2 pb;
                        pb[0] = (bf16)ex[r];
                        pb[1] = (bf16)ex[r + 1];
                        sum = __builtin_amdgcn_fdot2_f32_bf16(pb, one2, sum, false);
                        pu[r >> 3][(r & 7) >> 1] = __builtin_bit_cast(u32, pb);
                    }
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(v0), "+v"(v1));            // V^T(t) landed; K(t + 2) may still be on its way
                    ot = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, __builtin_bit_cast(bf16x8, pu[0]), ot, 0, 0, 0);
                    ot = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, __builtin_bit_cast(bf16x8, pu[1]), ot, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    vread(v0, v1, kt + 1);
                    a = an;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kn0), "+v"(kn1), "+v"(v0), "+v"(v1));      // the reads issued past the last tile
            };
            // ---- sub-step 0: pass 1, the row maximum (log2 domain: log2(e) / sqrt(d) is folded into W_q) --------------------------------
            boundary(sg * HEAD_SLABS + 0, std::integral_constant<int, 0>{});
            if (work) {
                const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                bf16x8 kn0, kn1;
                kread(kn0, kn1, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kn0), "+v"(kn1));
                f32x16 a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kn0, qf0, zero16, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kn1, qf1, a, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                kread(kn0, kn1, 1);
#pragma unroll 2
                for (int kt = 0; kt < nkt; ++kt) {
                    const bool more = kt + 1 < nkt;
                    f32x16 an = a;
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kn0), "+v"(kn1));
                    if (more) an = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kn0, qf0, zero16, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (kt == nkt - 1) pad_fix(a);
                    float m3 = fmaxf(fmaxf(a[0], a[1]), a[2]);
#pragma unroll
                    for (int r = 3; r + 1 < 16; r += 2) m3 = fmaxf(fmaxf(m3, a[r]), a[r + 1]);
                    mx = fmaxf(mx, fmaxf(m3, a[15]));
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) an = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kn1, qf1, an, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    kread(kn0, kn1, kt + 2);
                    a = an;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kn0), "+v"(kn1));
                mx = fmaxf(mx, __shfl_xor(mx, 32));        // the other half of this query's keys (every query has >= 1 real key: finite)
            }
            tick(c_p1);
#pragma unroll
            for (int r = 0; r < 16; ++r) { negm[r] = -mx; ot[r] = 0.f; }
            // ---- sub-steps 1 and 2: pass 2 over the first / the second half of the key tiles ------------------------------------------
            boundary(sg * HEAD_SLABS + 1, std::integral_constant<int, 1>{});
            if (work) pass2(0, n1);
            tick(c_p2a);
            boundary(sg * HEAD_SLABS + 2, std::integral_constant<int, 2>{});
            if (work) {
                pass2(n1, nkt);
                sum += __shfl_xor(sum, 32);
                const float inv = __builtin_amdgcn_rcpf(sum);
                // ot[4 g + e] = dim 8 g + 4 hh + e of query c31.  Lane halves trade two 8-byte pieces so that each stores 16 consecutive dims
                union { bf16x4 v; int i[2]; } pc[4], rcv[2];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                    for (int e = 0; e < 4; ++e) pc[g4].v[e] = (bf16)(ot[4 * g4 + e] * inv);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int snd0 = hh ? pc[j].i[0] : pc[2 + j].i[0], snd1 = hh ? pc[j].i[1] : pc[2 + j].i[1];
                    rcv[j].i[0] = __shfl_xor(snd0, 32);
                    rcv[j].i[1] = __shfl_xor(snd1, 32);
                }
                int q = qt * 32 + c31;
                asm volatile("" : "+v"(q));            // (opaque: the store addresses are rebuilt per head instead of living -- spilled -- across the loop)
                if (q < L) {
                    bf16x8 o0, o1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o0[e] = hh ? rcv[0].v[e] : pc[0].v[e];
                        o0[4 + e] = hh ? pc[2].v[e] : rcv[0].v[e];
                        o1[e] = hh ? rcv[1].v[e] : pc[1].v[e];
                        o1[4 + e] = hh ? pc[3].v[e] : rcv[1].v[e];
                    }
                    if (ctx_tiled) {
                        const int64_t m = t0 + q;
                        const int r = (int)(m & 15), sw = tswz(r);
                        bf16* blk = ctx + ((m >> 4) * NH + ha) * 512 + r * 32;
                        __builtin_nontemporal_store(o0, (bf16x8*)(blk + ((2 * hh) ^ sw) * 8));
                        __builtin_nontemporal_store(o1, (bf16x8*)(blk + ((2 * hh + 1) ^ sw) * 8));
                    } else {
                        bf16* dst = ctx + (int64_t)(t0 + q) * H + ha * DH + hh * 16;
                        __builtin_nontemporal_store(o0, (bf16x8*)dst);
                        __builtin_nontemporal_store(o1, (bf16x8*)(dst + 8));
                    }
                }
            }
            tick(c_p2b);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (DBG && lane == 0 && active) {
        unsigned long long* d = dbg + half * 8;
        atomicAdd(d + 0, 1ull); atomicAdd(d + 1, c_bar); atomicAdd(d + 2, c_proj); atomicAdd(d + 3, c_epi); atomicAdd(d + 4, c_p1); atomicAdd(d + 5, c_p2a);
        atomicAdd(d + 6, c_p2b); atomicAdd(d + 7, __builtin_readcyclecounter() - c_all);
    }
}

// ------------------------------------------------------------------------------------------------------------
// pooling heads
// ------------------------------------------------------------------------------------------------------------
// sentence-transformers Pooling + Normalize: one wave per sequence.  pool_cls = 0: masked mean over the sequence's tokens
// (`pooling_mode_mean_tokens`: sum(mask h) / max(sum(mask), 1e-9)); 1: the first token's state (`pooling_mode_cls_token`,
// the bge / GIST family the reference's .env.template:3 names).  normalize: the checkpoint's Normalize module
// (x / max(|x|, 1e-12)); without it the pooled vector is written as it is.
__global__ __launch_bounds__(64) void k_pool(const bf16* __restrict__ h, const int* __restrict__ cu, float* __restrict__ out,
                                             int64_t out_stride, int pool_cls, int normalize) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int t0 = cu[b], Lr = cu[b + 1] - t0;
    const int L = pool_cls ? min(Lr, 1) : Lr;
    const bool act = lane < 48;                        // 16-byte row pieces, four rows in flight
    const int c0 = lane * 8;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (act) {
        const bf16* r = h + (int64_t)t0 * H + c0;
        int t = 0;
        for (; t + 4 <= L; t += 4) {
            bf16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const bf16x8*)(r + (int64_t)(t + u) * H);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) s[i] += bf2f(v[u][i]);
        }
        for (; t < L; ++t) {
            const bf16x8 v = *(const bf16x8*)(r + (int64_t)t * H);
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += bf2f(v[i]);
        }
    }
    const float inv = pool_cls ? 1.0f : 1.0f / fmaxf((float)L, 1e-9f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] *= inv; q = fmaf(s[i], s[i], q); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rn = normalize ? 1.0f / fmaxf(sqrtf(q), 1e-12f) : 1.0f;
    if (act) {
        float* dst = out + (int64_t)b * out_stride + c0;
        if ((((uintptr_t)out | (uintptr_t)(out_stride * 4)) & 15) == 0) {   // any caller stride is legal: wide stores when aligned
            *(f32x4*)dst = f32x4{s[0] * rn, s[1] * rn, s[2] * rn, s[3] * rn};
            *(f32x4*)(dst + 4) = f32x4{s[4] * rn, s[5] * rn, s[6] * rn, s[7] * rn};
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) dst[i] = s[i] * rn;
        }
    }
}

// final hidden states of every real token as fp32 rows (sentence-transformers' output_value = "token_embeddings"; the
// parity tests compare them with the fp64 oracle token by token): packed row cu[b] + pos -> out[(cu[b] + pos) * stride ..]
__global__ __launch_bounds__(256) void k_tokens_out(const bf16* __restrict__ h, const int* __restrict__ cu, int batch,
                                                    float* __restrict__ out, int64_t out_stride) {
    const int64_t M = cu[batch];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one thread per 8 features
    const int64_t row = i / (H / 8);
    const int c0 = (int)(i % (H / 8)) * 8;
    if (row >= M) return;
    const bf16x8 v = *(const bf16x8*)(h + row * H + c0);
    float* dst = out + row * out_stride + c0;
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[e] = bf2f(v[e]);
}

// BertForSequenceClassification(num_labels=1): logit = wc . tanh(Wp h_cls + bp) + bc ; one block per sequence
__global__ __launch_bounds__(512) void k_cls_head(const bf16* __restrict__ h, const int* __restrict__ cu,
                                                  const float* __restrict__ wp, const float* __restrict__ bp,
                                                  const float* __restrict__ wc, const float* __restrict__ bc,
                                                  float* __restrict__ out) {
    // logit = wc . tanh(Wp h_cls + bp) + bc.  One workgroup of 8 waves per sequence; wave w takes pooler rows [48 w, +48), sixteen at a
    // time: the 64 lanes read a row's 384 fp32 weights as six coalesced 256-byte pieces, 96 loads in flight per lane before the first
    // use, then shuffle reductions.  (Round 3 walked one row per THREAD -- 1536-byte strides between the lanes of a load, 384 dependent
    // steps: 26 us for 14 sequences, as long as two encoder layers of the rerank call it ends.  The kernel is a chain of L2 / fabric
    // round trips either way; this form has three of them instead of hundreds.)
    __shared__ float red[8];
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t0 = cu[b], L = cu[b + 1] - t0;
    float hc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) hc[j] = L > 0 ? bf2f(h[(int64_t)t0 * H + lane + 64 * j]) : 0.f;
    float part = 0.f;
#pragma unroll 1
    for (int o0 = w * 48; o0 < w * 48 + 48; o0 += 16) {
        float wv[16][6];
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < 6; ++j) wv[r][j] = wp[(int64_t)(o0 + r) * H + lane + 64 * j];
        float a[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            a[r] = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j) a[r] = fmaf(wv[r][j], hc[j], a[r]);
        }
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1)
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] += __shfl_xor(a[r], sft);
#pragma unroll
        for (int r = 0; r < 16; ++r) part = fmaf(wc[o0 + r], tanhf(a[r] + bp[o0 + r]), part);
    }
    if (lane == 0) red[w] = part;
    __syncthreads();
    if (threadIdx.x == 0) out[b] = (((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]))) + bc[0];
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// model object + C-ABI
// ------------------------------------------------------------------------------------------------------------
struct BertLayer {
    bf16 *wqkv, *wo, *w1, *w2;
    bf16* w2p;      // W2 with the columns of every 32-block permuted to the fused FFN kernel's k-slot order
    bf16* w1p;      // (debug builds) W1 with the same permutation of ITS columns: k_ffn3's fused out-proj prologue builds the token fragments in accumulator order
    bf16 *wqkv_t, *wo_t, *w1_t, *w2_t;   // k_tile_w copies for k_gemm3
    bf16* wqkv_s;                        // k_pack_qa stream for k_qa (per head: 72 fragments of 1 KiB in consumption order)
    bf16 *w1s, *w2s;                     // k_pack_ffn3 streams for k_ffn3's VAR bit 1 (debug builds only; nullptr otherwise)
    float *bqkv, *bo, *b1, *b2, *ln1g, *ln1b, *ln2g, *ln2b;
};
struct rmu_bert {
    rmu_bert_cfg cfg;
    float *wemb = nullptr, *pemb = nullptr, *temb = nullptr, *elng = nullptr, *elnb = nullptr;
    std::vector<BertLayer> layers;
    float *wp = nullptr, *bp = nullptr, *wc = nullptr, *bc = nullptr;
    std::vector<void*> owned;
    // workspace (guarded: one encode at a time per model)
    std::mutex mu;
    int64_t ws_tokens = 0;
    int ws_batch = 0;
    bf16 *h = nullptr, *h1 = nullptr, *y = nullptr, *qkv = nullptr, *ctx = nullptr, *mid = nullptr;
    float2 *st1 = nullptr, *st2 = nullptr;        // (mean, rstd) per token of the two LayerNorms folded into their consumers (small path)
    int* cu = nullptr;
    int2* qa_items = nullptr;                     // k_qa's work list (k_qa_items): [ws_batch] (first sequence, sequences); its length at qa_items[ws_batch]
    hipStream_t stream = nullptr;
    // small-batch host path (rmu_bert_encode_host): one captured graph per (batch, max_len, mode, token types) shape -- H2D of the
    // ids, the ~45 launches of a forward, D2H of the result -- replayed with ONE hipGraphLaunch.  All addresses inside are
    // the fixed staging buffers below, so a replay needs no node updates.
    struct SmallGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; bool warm = false; bool no_graph = false; uint64_t last_use = 0; };
    std::map<uint64_t, SmallGraph> graphs;
    uint64_t graph_clock = 0;                      // bumps per host call: the least recently used graph goes when the cache is full
    int32_t *h_in = nullptr, *d_in = nullptr;      // [ids | type ids | lens], pinned host / device
    float *h_out = nullptr, *d_out = nullptr;
    // (round 5) Host-path calls from SEVERAL threads (LCEL runs the reference's retriever branches in parallel: server/RAGHelper_local.py:254-258;
    // a bulk embed_documents beside an embed_query) used to queue on `mu` for the whole forward + synchronisation.  A call that finds this
    // model busy takes a CLONE: the same weight pointers, its own workspace, stream, staging buffers and graph cache (created on first
    // need, at most MAX_CLONES); the forwards of two threads then overlap on the device.  Clones own no weights.
    std::vector<rmu_bert*> clones;
    std::mutex clones_mu;
    bool is_clone = false;
    // A forward that rmu_bert_encode left IN FLIGHT on a caller's stream still owns this context's workspace.  Its end is marked with an
    // event; whatever this context enqueues next on ANOTHER stream (the library's own for stream-0 and host-path calls, or a second caller
    // stream) waits for that event on the device first.  (round 5: MI355XEmbeddings queues block i + 1's forward behind block i's.)
    hipEvent_t tail_ev = nullptr;
    hipStream_t tail_stream = nullptr;
    bool tail_set = false;
};
static constexpr size_t MAX_CLONES = 3;
// rmu_bert_encode_host / rmu_bert_search_mmr carry up to HOST_TOKENS tokens (batch * max_len): one query, or the <= 14 (query, passage)
// pairs of one rerank call (server/ScoredCrossEncoderReranker.py:42) -- staging: ids | type ids | lens; results: <= 256 rows of 384
// floats (pooled vectors / the token states of a 256-token call) or one logit per sequence
static constexpr int HOST_TOKENS = 4096;
static constexpr int SMALL_IN_INTS = 3 * HOST_TOKENS, SMALL_OUT_FLOATS = 256 * 384 > HOST_TOKENS ? 256 * 384 : HOST_TOKENS;
static constexpr size_t MAX_GRAPHS = 64;

extern "C" void rmu_set_error_(const char* msg);   // rmu_api.hip: thread-local message behind rmu_last_error()
static int bfail(int code, const std::string& m) { rmu_set_error_(m.c_str()); return code; }

#define B_TRY(expr)                                                                                      \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return bfail(e_ == hipErrorOutOfMemory ? RMU_E_OOM : RMU_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <class T>
static int dev_alloc(rmu_bert* m, T** p, size_t count) {
    void* q = nullptr;
    if (hipMalloc(&q, count * sizeof(T)) != hipSuccess) return RMU_E_OOM;
    m->owned.push_back(q);
    *p = (T*)q;
    return RMU_OK;
}

static int copy_f32(rmu_bert* m, float** dst, const void* src, size_t n, float scale, hipStream_t s) {
    if (dev_alloc(m, dst, n)) return RMU_E_OOM;
    hipLaunchKernelGGL(k_scale_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)src, *dst, (int64_t)n, scale);
    return RMU_OK;
}
static int conv_bf16(bf16* dst, const void* src, size_t n, float scale, hipStream_t s) {
    hipLaunchKernelGGL(k_f32_to_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)src, dst, (int64_t)n, scale);
    return RMU_OK;
}

static void quiesce(rmu_bert* m);
extern "C" int rmu_bert_free(rmu_bert_t* m) {
    RMU_ENTRY();
    if (!m) return RMU_OK;
    quiesce(m);                                  // this context's own work only -- never the whole device (rmu_common.h: captures)
    for (rmu_bert* c : m->clones) (void)rmu_bert_free(c);      // (their `owned` lists are empty: workspaces, staging, graphs, stream)
    m->clones.clear();
    for (void* p : m->owned) (void)hipFree(p);
    for (void* p : {(void*)m->h, (void*)m->h1, (void*)m->y, (void*)m->qkv, (void*)m->ctx, (void*)m->mid, (void*)m->cu, (void*)m->qa_items, (void*)m->st1, (void*)m->st2})
        if (p) (void)hipFree(p);
    for (auto& kv : m->graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    if (m->h_in) (void)hipHostFree(m->h_in);
    if (m->h_out) (void)hipHostFree(m->h_out);
    if (m->d_in) (void)hipFree(m->d_in);
    if (m->d_out) (void)hipFree(m->d_out);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    if (m->tail_ev) (void)hipEventDestroy(m->tail_ev);
    delete m;
    return RMU_OK;
}

// everything a context ever enqueued ran on its own stream or, left in flight by rmu_bert_encode, is marked by its tail event
static void quiesce(rmu_bert* m) {
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->tail_set && m->tail_ev) (void)hipEventSynchronize(m->tail_ev);
    m->tail_set = false;
    (void)hipGetLastError();
}

// (m->mu held) see rmu_bert::tail_ev
static int order_behind_tail(rmu_bert* m, hipStream_t s) {
    if (m->tail_set && m->tail_stream != s && hipStreamWaitEvent(s, m->tail_ev, 0) != hipSuccess) return RMU_E_HIP;
    return RMU_OK;
}
static int mark_tail(rmu_bert* m, hipStream_t s) {
    if (!m->tail_ev && hipEventCreateWithFlags(&m->tail_ev, hipEventDisableTiming) != hipSuccess) { m->tail_ev = nullptr; return RMU_E_HIP; }
    if (hipEventRecord(m->tail_ev, s) != hipSuccess) return RMU_E_HIP;
    m->tail_stream = s;
    m->tail_set = true;
    return RMU_OK;
}

extern "C" int rmu_bert_create(rmu_bert_t** out, const rmu_bert_cfg* cfg, const void* const* wptr, int n_weights) {
    RMU_ENTRY();
    if (!out || !cfg || !wptr) return bfail(RMU_E_INVALID, "rmu_bert_create: null argument");
    if (cfg->hidden != H || cfg->heads != NH || cfg->ffn != FF || cfg->layers < 1 || cfg->layers > 48)
        return bfail(RMU_E_INVALID, "rmu_bert_create: this build supports hidden 384, 12 heads, ffn 1536");
    if (cfg->max_pos < 1 || cfg->max_pos > 512 || cfg->vocab_size < 1 || cfg->type_vocab < 1)
        return bfail(RMU_E_INVALID, "rmu_bert_create: max_pos must be in [1, 512]");
    const int need = 5 + 16 * cfg->layers + (cfg->has_head ? 4 : 0);
    if (n_weights != need) return bfail(RMU_E_INVALID, "rmu_bert_create: wrong number of weight tensors");
    for (int i = 0; i < need; ++i)
        if (!wptr[i]) return bfail(RMU_E_INVALID, "rmu_bert_create: null weight pointer");
    auto* m = new (std::nothrow) rmu_bert();
    if (!m) return bfail(RMU_E_OOM, "rmu_bert_create: host alloc");
    m->cfg = *cfg;
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) { delete m; return bfail(RMU_E_HIP, "stream"); }
    hipStream_t s = m->stream;
    int rc = RMU_OK;
    int wi = 0;
    rc |= copy_f32(m, &m->wemb, wptr[wi++], (size_t)cfg->vocab_size * H, 1.f, s);
    rc |= copy_f32(m, &m->pemb, wptr[wi++], (size_t)cfg->max_pos * H, 1.f, s);
    rc |= copy_f32(m, &m->temb, wptr[wi++], (size_t)cfg->type_vocab * H, 1.f, s);
    rc |= copy_f32(m, &m->elng, wptr[wi++], H, 1.f, s);
    rc |= copy_f32(m, &m->elnb, wptr[wi++], H, 1.f, s);
    // softmax scale and log2(e) folded into the query projection: both attention kernels use exp2 on the raw MFMA output
    const float qs = 1.4426950408889634f / sqrtf((float)DH);
    m->layers.resize(cfg->layers);
    for (int l = 0; l < cfg->layers && !rc; ++l) {
        BertLayer& L = m->layers[l];
        const void *qw = wptr[wi++], *qb = wptr[wi++], *kw = wptr[wi++], *kb = wptr[wi++], *vw = wptr[wi++], *vb = wptr[wi++];
        rc |= dev_alloc(m, &L.wqkv, (size_t)3 * H * H);
        rc |= dev_alloc(m, &L.bqkv, (size_t)3 * H);
        if (rc) break;
        conv_bf16(L.wqkv, qw, (size_t)H * H, qs, s);
        conv_bf16(L.wqkv + H * H, kw, (size_t)H * H, 1.f, s);
        conv_bf16(L.wqkv + 2 * H * H, vw, (size_t)H * H, 1.f, s);
        hipLaunchKernelGGL(k_scale_copy, dim3(2), dim3(256), 0, s, (const float*)qb, L.bqkv, (int64_t)H, qs);
        hipLaunchKernelGGL(k_scale_copy, dim3(2), dim3(256), 0, s, (const float*)kb, L.bqkv + H, (int64_t)H, 1.f);
        hipLaunchKernelGGL(k_scale_copy, dim3(2), dim3(256), 0, s, (const float*)vb, L.bqkv + 2 * H, (int64_t)H, 1.f);
        rc |= dev_alloc(m, &L.wqkv_s, (size_t)3 * H * H);
        if (!rc) hipLaunchKernelGGL(k_pack_qa, dim3((unsigned)((3 * H * H / 8 + 255) / 256)), dim3(256), 0, s, (const bf16*)L.wqkv, L.wqkv_s);
        rc |= dev_alloc(m, &L.wo, (size_t)H * H);
        if (!rc) conv_bf16(L.wo, wptr[wi], (size_t)H * H, 1.f, s);
        wi++;
        rc |= copy_f32(m, &L.bo, wptr[wi++], H, 1.f, s);
        rc |= copy_f32(m, &L.ln1g, wptr[wi++], H, 1.f, s);
        rc |= copy_f32(m, &L.ln1b, wptr[wi++], H, 1.f, s);
        rc |= dev_alloc(m, &L.w1, (size_t)FF * H);
        if (!rc) conv_bf16(L.w1, wptr[wi], (size_t)FF * H, 1.f, s);
#ifdef RMU_DEBUG_KERNELS
        rc |= dev_alloc(m, &L.w1p, (size_t)FF * H);
        if (!rc) hipLaunchKernelGGL(k_permute_w2, dim3((unsigned)(((size_t)FF * H + 255) / 256)), dim3(256), 0, s, (const bf16*)L.w1, L.w1p, (int64_t)FF * H);
#endif
        wi++;
        rc |= copy_f32(m, &L.b1, wptr[wi++], FF, 1.f, s);
        rc |= dev_alloc(m, &L.w2, (size_t)H * FF);
        rc |= dev_alloc(m, &L.w2p, (size_t)H * FF);
        if (!rc) {
            conv_bf16(L.w2, wptr[wi], (size_t)H * FF, 1.f, s);
            hipLaunchKernelGGL(k_permute_w2, dim3((unsigned)(((size_t)H * FF + 255) / 256)), dim3(256), 0, s, (const bf16*)L.w2, L.w2p, (int64_t)H * FF);
        }
#ifdef RMU_DEBUG_KERNELS
        rc |= dev_alloc(m, &L.w1s, (size_t)FF * H);
        rc |= dev_alloc(m, &L.w2s, (size_t)H * FF);
        if (!rc) hipLaunchKernelGGL(k_pack_ffn3, dim3((unsigned)((2 * (size_t)FF * H / 8 + 255) / 256)), dim3(256), 0, s, (const bf16*)L.w1, (const bf16*)L.w2p, L.w1s, L.w2s);
#endif
        wi++;
        rc |= copy_f32(m, &L.b2, wptr[wi++], H, 1.f, s);
        rc |= copy_f32(m, &L.ln2g, wptr[wi++], H, 1.f, s);
        rc |= copy_f32(m, &L.ln2b, wptr[wi++], H, 1.f, s);
        {   // k_gemm3's tiled weight copies (all four: the RMU_GEMM3 mask can route any of the layer's GEMMs through it)
            struct { bf16* src; bf16** dst; int n, k; } tw[4] = {{L.wqkv, &L.wqkv_t, 3 * H, H}, {L.wo, &L.wo_t, H, H}, {L.w1, &L.w1_t, FF, H}, {L.w2, &L.w2_t, H, FF}};
            for (auto& t : tw) {
                rc |= dev_alloc(m, t.dst, (size_t)t.n * t.k);
                if (!rc) hipLaunchKernelGGL(k_tile_w, dim3((unsigned)(((size_t)t.n * t.k / 8 + 255) / 256)), dim3(256), 0, s, (const bf16*)t.src, *t.dst, t.n, t.k);
            }
        }
    }
    if (!rc && cfg->has_head) {
        rc |= copy_f32(m, &m->wp, wptr[wi++], (size_t)H * H, 1.f, s);
        rc |= copy_f32(m, &m->bp, wptr[wi++], H, 1.f, s);
        rc |= copy_f32(m, &m->wc, wptr[wi++], H, 1.f, s);
        rc |= copy_f32(m, &m->bc, wptr[wi++], 1, 1.f, s);
    }
    if (rc || hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) {
        rmu_bert_free(m);
        return bfail(rc ? rc : RMU_E_HIP, "rmu_bert_create: weight upload failed");
    }
    *out = m;
    return RMU_OK;
}

// the context a host-path call runs on: the model itself when it is free, else a free clone (created on first need), else -- everything
// busy -- the model itself, queued.  `lk` holds the chosen context's mutex on return.
static rmu_bert* acquire_ctx(rmu_bert* m, std::unique_lock<std::mutex>& lk) {
    lk = std::unique_lock<std::mutex>(m->mu, std::try_to_lock);
    if (lk.owns_lock()) {
        // free to lock is not free to run: a bulk forward left in flight on a caller's stream (rmu_bert_encode) still owns the workspace, and
        // a query would queue behind it on the device -- it takes a clone instead, as it does when the model is locked
        bool busy = false;
        if (m->tail_set) {
            busy = hipEventQuery(m->tail_ev) == hipErrorNotReady;
            (void)hipGetLastError();             // (not-ready is an answer, not an error to be found by a later check)
            if (!busy) m->tail_set = false;
        }
        if (!busy) return m;
        lk.unlock();
    }
    {
        std::lock_guard<std::mutex> g(m->clones_mu);
        for (rmu_bert* c : m->clones) {
            lk = std::unique_lock<std::mutex>(c->mu, std::try_to_lock);
            if (lk.owns_lock()) return c;
        }
        if (m->clones.size() < MAX_CLONES) {
            auto* c = new (std::nothrow) rmu_bert();
            if (c && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess) {
                c->cfg = m->cfg;
                c->wemb = m->wemb; c->pemb = m->pemb; c->temb = m->temb; c->elng = m->elng; c->elnb = m->elnb;
                c->layers = m->layers;                       // (weight POINTERS: the tensors stay the model's)
                c->wp = m->wp; c->bp = m->bp; c->wc = m->wc; c->bc = m->bc;
                c->is_clone = true;
                m->clones.push_back(c);
                lk = std::unique_lock<std::mutex>(c->mu);
                return c;
            }
            delete c;
        }
    }
    lk = std::unique_lock<std::mutex>(m->mu);
    return m;
}

static void drop_graphs(rmu_bert* m) {       // the captured launches hold workspace addresses
    for (auto& kv : m->graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    m->graphs.clear();
}

static int ensure_ws(rmu_bert* m, int64_t tokens, int batch) {
    if (tokens > m->ws_tokens || batch + 1 > m->ws_batch) drop_graphs(m);
    if (tokens > m->ws_tokens) {
        if (m->h) {                                   // (a fresh context -- a clone's first call -- has nothing in flight and nothing to free)
            quiesce(m);                               // the workspace's only users: this context's stream and the forward marked by its tail event
            for (void* p : {(void*)m->h, (void*)m->h1, (void*)m->y, (void*)m->qkv, (void*)m->ctx, (void*)m->mid, (void*)m->st1, (void*)m->st2})
                if (p) (void)hipFree(p);
        }
        m->h = m->h1 = m->y = m->qkv = m->ctx = m->mid = nullptr;
        m->st1 = m->st2 = nullptr;
        m->ws_tokens = 0;
        const int64_t t = tokens + tokens / 8 + 512;       // (+ a whole 256-token tile: the tiled out-proj operand is read in full blocks)
        if (hipMalloc((void**)&m->h, t * H * 2) != hipSuccess || hipMalloc((void**)&m->h1, t * H * 2) != hipSuccess ||
            hipMalloc((void**)&m->y, t * H * 2) != hipSuccess || hipMalloc((void**)&m->qkv, t * 3 * H * 2) != hipSuccess ||
            hipMalloc((void**)&m->ctx, t * H * 2) != hipSuccess || hipMalloc((void**)&m->mid, t * FF * 2) != hipSuccess ||
            hipMalloc((void**)&m->st1, (size_t)std::min<int64_t>(t, FOLD_TOKENS + 64) * sizeof(float2)) != hipSuccess ||
            hipMalloc((void**)&m->st2, (size_t)std::min<int64_t>(t, FOLD_TOKENS + 64) * sizeof(float2)) != hipSuccess)
            return RMU_E_OOM;
        m->ws_tokens = t;
    }
    if (batch + 1 > m->ws_batch) {
        if (m->cu) {
            quiesce(m);
            (void)hipFree(m->cu);
            if (m->qa_items) (void)hipFree(m->qa_items);
        }
        m->cu = nullptr; m->qa_items = nullptr; m->ws_batch = 0;
        const int nb = batch + batch / 8 + 64;
        if (hipMalloc((void**)&m->cu, (size_t)nb * sizeof(int)) != hipSuccess) return RMU_E_OOM;
        if (hipMalloc((void**)&m->qa_items, (size_t)(nb + 1) * sizeof(int2)) != hipSuccess) return RMU_E_OOM;
        m->ws_batch = nb;
    }
    return RMU_OK;
}

template <int EPI, int WM, int BK, int ST, bool TA = false>
static void launch_gemm_cfg(const bf16* A, const bf16* W, const float* bias, const bf16* resid, bf16* out, const int* cu,
                            int batch, int64_t m_cap, int N, int K, hipStream_t s, bool resid_tiled = false) {
    constexpr int BM = 64 * WM;
    constexpr int ring = ST * (BM * BK * 2 + BN * BK * 2), stagebuf = 2 * WM * 64 * 144;
    constexpr int lds = ring > stagebuf ? ring : stagebuf;
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)k_gemm<EPI, WM, BK, ST, TA>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)attr_rc;
    const int64_t mt = ((m_cap + BM - 1) / BM + 7) / 8 * 8;     // token tiles, padded to a multiple of 8 (XCD map)
    const dim3 grid((unsigned)(mt * (N / BN)));
#ifdef RMU_DEBUG_KERNELS
    static const int dbg = rmu_env("RMU_GEMM_DBG") ? atoi(rmu_env("RMU_GEMM_DBG")) : 0;   // timing ablations: 1 no stores, 2 no main loop
#else
    const int dbg = 0;
#endif
    hipLaunchKernelGGL((k_gemm<EPI, WM, BK, ST, TA>), grid, dim3(128 * WM), lds, s, A, W, bias, resid, out, cu, batch, N, K, dbg | (resid_tiled ? 256 : 0));
}
// Tokens (batch * max_len) up to which all four GEMMs of a layer take k_gemm_small, and from which the QKV projection takes the
// persistent k_gemm3.  Round 4 measured the rerank shape (14 pairs, ~1.5k tokens, tools/ce_probe.py) with k_gemm_small over token blocks
// of 32 / 64 / 128 up to 4k-64k tokens and with deeper / smaller tiled configurations (128x128 x 3-4 stages, 64x128 x 4-6): nothing beat
// the round-3 choice (0.48-0.50 ms per call either way; every launch there is 5-15 us of latency, not of work), so the thresholds stay.
// Debug builds keep the switches: RMU_MID_TOKENS, RMU_SMALL_TB, RMU_GEMM_CFG (+ RMU_GEMM_CFG_MAX), RMU_G3_MIN.
static int64_t mid_tokens() {
#ifdef RMU_DEBUG_KERNELS
    const int64_t v = rmu_env("RMU_MID_TOKENS") ? atoll(rmu_env("RMU_MID_TOKENS")) : SMALL_M;
    return v < SMALL_M ? SMALL_M : v;
#else
    return SMALL_M;
#endif
}
static int64_t g3_min_tokens() {
#ifdef RMU_DEBUG_KERNELS
    const int64_t v = rmu_env("RMU_G3_MIN") ? atoll(rmu_env("RMU_G3_MIN")) : 0;
    return v > mid_tokens() ? v : mid_tokens();
#else
    return SMALL_M;
#endif
}
template <int EPI>
static void launch_gemm(const bf16* A, const bf16* W, const float* bias, const bf16* resid, bf16* out, const int* cu,
                        int batch, int64_t m_cap, int N, int K, hipStream_t s) {
    static const bool small_ok = !(rmu_env("RMU_GEMM_SMALL") && atoi(rmu_env("RMU_GEMM_SMALL")) == 0);
    if (small_ok && m_cap <= mid_tokens() && (K == H || K == FF)) {
#ifdef RMU_DEBUG_KERNELS
        const int tb = rmu_env("RMU_SMALL_TB") ? atoi(rmu_env("RMU_SMALL_TB")) / 32 * 32 : 128;
#else
        const int tb = 128;
#endif
        if (K == H) launch_small<EPI, H>(s, m_cap, tb, A, W, bias, resid, out, cu, batch, N);
        else launch_small<EPI, FF>(s, m_cap, tb, A, W, bias, resid, out, cu, batch, N);
        return;
    }
    // few tiles (latency-bound: every workgroup walks K alone): 128 x 128 tiles in 64-k stages halve the stage count and double
    // the workgroups -- 4k tokens 0.70 -> 0.60 ms per forward, 16k tokens 1.12 -> 1.07; big batches keep 256 x 128 / 32-k stages
    // (best of the seven tile / stage / ring combinations measured in round 2)
#ifdef RMU_DEBUG_KERNELS
    {   // (A/B of the mid-size shapes, measured and not adopted: RMU_GEMM_CFG = 1: 128x128 / 3 stages, 2: 128x128 / 4, 3: 64x128 / 4, 4: 64x128 / 6)
        const int cfgv = rmu_env("RMU_GEMM_CFG") ? atoi(rmu_env("RMU_GEMM_CFG")) : 0;
        const int64_t lim = rmu_env("RMU_GEMM_CFG_MAX") ? atoll(rmu_env("RMU_GEMM_CFG_MAX")) : 8192;
        if (cfgv && m_cap <= lim) {
            if (cfgv == 1) return launch_gemm_cfg<EPI, 2, 64, 3>(A, W, bias, resid, out, cu, batch, m_cap, N, K, s);
            if (cfgv == 2) return launch_gemm_cfg<EPI, 2, 64, 4>(A, W, bias, resid, out, cu, batch, m_cap, N, K, s);
            if (cfgv == 3) return launch_gemm_cfg<EPI, 1, 64, 4>(A, W, bias, resid, out, cu, batch, m_cap, N, K, s);
            if (cfgv == 4) return launch_gemm_cfg<EPI, 1, 64, 6>(A, W, bias, resid, out, cu, batch, m_cap, N, K, s);
        }
    }
#endif
    if (m_cap <= 32768) return launch_gemm_cfg<EPI, 2, 64, 2>(A, W, bias, resid, out, cu, batch, m_cap, N, K, s);
    return launch_gemm_cfg<EPI, 4, 32, 2>(A, W, bias, resid, out, cu, batch, m_cap, N, K, s);
}

#ifdef RMU_DEBUG_KERNELS
// The cycle-counter / ablation instantiations (k_ffn_fused<true>, k_gemm3<*, true>: RMU_FFN_DBG, RMU_G3_DBG) exist only in a
// build with -DRMU_DEBUG_KERNELS (python -m ragmeup_amd.build --debug-kernels): the product library carries one instantiation
// of every kernel it can actually take.
static void launch_ffn_fused(const bf16* h1, const BertLayer& L, float eps, bf16* out, const int* cu, int batch, int64_t m_cap, hipStream_t s) {
    const dim3 grid((unsigned)((m_cap + ffn::TOK - 1) / ffn::TOK));
#ifdef RMU_DEBUG_KERNELS
    static const bool want_dbg = rmu_env("RMU_FFN_DBG") != nullptr;
    if (want_dbg) {
        static const hipError_t attr_dbg = hipFuncSetAttribute((const void*)k_ffn_fused<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ffn::LDS_BYTES);
        (void)attr_dbg;
        static unsigned long long* dbg = nullptr;
        if (!dbg) { (void)hipMalloc((void**)&dbg, 64); (void)hipMemset(dbg, 0, 64); }
        static const unsigned long long fl = rmu_env("RMU_FFN_FLAGS") ? strtoull(rmu_env("RMU_FFN_FLAGS"), nullptr, 10) : 0ull;
        (void)hipMemcpyAsync(dbg + 7, &fl, 8, hipMemcpyHostToDevice, s);
        hipLaunchKernelGGL(k_ffn_fused<true>, grid, dim3(256), ffn::LDS_BYTES, s, h1, L.w1, L.b1, L.w2p, L.b2, L.ln2g, L.ln2b, eps, out, cu, batch, dbg);
        unsigned long long h[5];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, dbg, 40, hipMemcpyDeviceToHost);
        (void)hipMemset(dbg, 0, 64);
        fprintf(stderr, "[ffn dbg] waves=%llu  cycles/wave: total=%.0f  wait(slab+barrier)=%.0f (%.1f%%)  gemm1=%.0f (per slab %.0f)  gemm2+gelu=%.0f (per slab %.0f)\n",
                h[2], (double)h[1] / (double)h[2], (double)h[0] / (double)h[2], 100.0 * (double)h[0] / (double)h[1], (double)h[3] / (double)h[2],
                (double)h[3] / (double)h[2] / 72.0, (double)h[4] / (double)h[2], (double)h[4] / (double)h[2] / 48.0);
        return;
    }
#endif
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)k_ffn_fused<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ffn::LDS_BYTES);
    (void)attr_rc;
    hipLaunchKernelGGL(k_ffn_fused<false>, grid, dim3(256), ffn::LDS_BYTES, s, h1, L.w1, L.b1, L.w2p, L.b2, L.ln2g, L.ln2b, eps, out, cu, batch,
                       (unsigned long long*)nullptr);
}

#endif

template <bool LN_IN, int GV, int PF>
static void launch_ffn2_t(const bf16* x, const BertLayer& L, float eps, bf16* out, const int* cu, int batch, int64_t m_cap, hipStream_t s) {
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)k_ffn2<LN_IN, GV, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, ffn2::LDS_BYTES);
    (void)attr_rc;
    const dim3 grid((unsigned)((m_cap + ffn2::TOK - 1) / ffn2::TOK));
    hipLaunchKernelGGL((k_ffn2<LN_IN, GV, PF>), grid, dim3(256), ffn2::LDS_BYTES, s, x, L.w1, L.b1, L.w2p, L.b2, L.ln2g, L.ln2b, eps, out, cu, batch,
                       L.ln1g, L.ln1b);
}
// x = h1 (ln_in false) or the pre-LN out-proj sum y (ln_in true: LN1 happens in the kernel's prologue)
static void launch_ffn2(const bf16* x, bool ln_in, const BertLayer& L, float eps, bf16* out, const int* cu, int batch, int64_t m_cap, hipStream_t s) {
    static const int gv = rmu_env("RMU_FFN_GELU") ? atoi(rmu_env("RMU_FFN_GELU")) : 0;   // 1: scalar v_fma_f32 polynomial (A/B measurement)
    static const int pf = rmu_env("RMU_FFN_PF") ? atoi(rmu_env("RMU_FFN_PF")) : 4;       // 8: eight weight fragments read ahead (A/B measurement)
    if (pf == 8 && ln_in && !gv) return launch_ffn2_t<true, 0, 8>(x, L, eps, out, cu, batch, m_cap, s);
    if (ln_in) { if (gv) launch_ffn2_t<true, 1, 4>(x, L, eps, out, cu, batch, m_cap, s); else launch_ffn2_t<true, 0, 4>(x, L, eps, out, cu, batch, m_cap, s); }
    else { if (gv) launch_ffn2_t<false, 1, 4>(x, L, eps, out, cu, batch, m_cap, s); else launch_ffn2_t<false, 0, 4>(x, L, eps, out, cu, batch, m_cap, s); }
}

template <bool LN_IN, int PF, int VAR>
static void launch_ffn3_t(const bf16* x, const BertLayer& L, float eps, bf16* out, const int* cu, int batch, int64_t m_cap, hipStream_t s, bool out_tiled) {
    const dim3 grid((unsigned)((m_cap + ffn3::TOK - 1) / ffn3::TOK));
    const bf16* w1 = (VAR & 2) ? L.w1s : L.w1;
    const bf16* w2 = (VAR & 2) ? L.w2s : L.w2p;
#ifdef RMU_DEBUG_KERNELS
    static const int dflags = rmu_env("RMU_FFN3_DBG") ? atoi(rmu_env("RMU_FFN3_DBG")) : 0;
    if (dflags) {
        static const hipError_t attr_d = hipFuncSetAttribute((const void*)k_ffn3<LN_IN, PF, true, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, ffn3::LDS_BYTES);
        (void)attr_d;
        hipLaunchKernelGGL((k_ffn3<LN_IN, PF, true, VAR>), grid, dim3(512), ffn3::LDS_BYTES, s, x, w1, L.b1, w2, L.b2, L.ln2g, L.ln2b, eps, out, cu, batch,
                           L.ln1g, L.ln1b, dflags | (out_tiled ? 256 : 0));
        return;
    }
#endif
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)k_ffn3<LN_IN, PF, false, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, ffn3::LDS_BYTES);
    (void)attr_rc;
    static const int epi_old = rmu_env("RMU_FFN3_EPI") && atoi(rmu_env("RMU_FFN3_EPI")) == 0 ? 2048 : 0;   // A/B: the round-3/4 epilogue (row-serial LayerNorm 2)
    hipLaunchKernelGGL((k_ffn3<LN_IN, PF, false, VAR>), grid, dim3(512), ffn3::LDS_BYTES, s, x, w1, L.b1, w2, L.b2, L.ln2g, L.ln2b, eps, out, cu, batch,
                       L.ln1g, L.ln1b, (out_tiled ? 256 : 0) | epi_old);
}
#ifdef RMU_DEBUG_KERNELS
// the attention output in, the layer output out: out-proj + residual + LayerNorm 1 + FFN + LayerNorm 2 in one launch.  MEASURED (8192
// chunks, same box): 3830 us per launch against 3314 (k_ffn3) + 569 (out-proj k_gemm) = 3883 -- the prologue's 12 barrier-separated slabs
// of 12 MFMAs per wave run at a third of the main loop's rate and eat what the saved launch and the 2.4 GB of traffic gave: -1.4 % per
// layer, not worth a second k_ffn3 instantiation, a third copy of W1 and 40 bytes of scratch.  Debug builds: RMU_FFN_OUTPROJ=1.
static void launch_ffn3_outproj(const bf16* ctx, bool ctx_tiled, const bf16* resid, bool resid_tiled, const BertLayer& L, float eps, bf16* out,
                                const int* cu, int batch, int64_t m_cap, hipStream_t s, bool out_tiled) {
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)k_ffn3<true, 4, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ffn3::LDS_BYTES);
    (void)attr_rc;
    const dim3 grid((unsigned)((m_cap + ffn3::TOK - 1) / ffn3::TOK));
    hipLaunchKernelGGL((k_ffn3<true, 4, false, 0, true>), grid, dim3(512), ffn3::LDS_BYTES, s, ctx, L.w1p, L.b1, L.w2p, L.b2, L.ln2g, L.ln2b, eps, out, cu, batch,
                       L.ln1g, L.ln1b, (out_tiled ? 256 : 0) | (ctx_tiled ? 512 : 0) | (resid_tiled ? 1024 : 0), L.wo, L.bo, resid);
}
#endif
static void launch_ffn3(const bf16* x, bool ln_in, const BertLayer& L, float eps, bf16* out, const int* cu, int batch, int64_t m_cap, hipStream_t s, bool out_tiled = false) {
#ifdef RMU_DEBUG_KERNELS
    // A/B forms (measured, 8192 chunks, per launch: VAR 0 3227 us, 1 3404, 2 3245, 3 3370 -- the packed-f32 activation is SLOWER than
    // hipcc's mostly scalar one although the loop shrinks from 431 to 357 instructions, and the stream form of the weights changes nothing)
    static const int var = rmu_env("RMU_FFN3_VAR") ? atoi(rmu_env("RMU_FFN3_VAR")) : 0;
#define RMU_FFN3_CASE(V) case V: if (ln_in) launch_ffn3_t<true, 4, V>(x, L, eps, out, cu, batch, m_cap, s, out_tiled); else launch_ffn3_t<false, 4, V>(x, L, eps, out, cu, batch, m_cap, s, out_tiled); return;
    switch (var & 3) { RMU_FFN3_CASE(1) RMU_FFN3_CASE(2) RMU_FFN3_CASE(3) default: break; }
#undef RMU_FFN3_CASE
#endif
#ifdef RMU_DEBUG_KERNELS
    // VAR bit 2, the LOOK-AHEAD form (round 4; ffn3::wait_la): bit-identical output, and MEASURED EQUAL -- bench.py's embed leg 34.49-34.76 ms
    // per 8192-chunk forward without it, 34.58-34.70 with it (three interleaved runs each, one box), k_ffn3 3346.7 vs 3351.7 us per launch
    // under rocprofv3, MFMA busy 0.393 vs 0.378: the LDS round trip behind each of a chunk's five barriers is NOT what the kernel waits
    // for (the second wave of the SIMD and the 3-slab DMA lead already cover it).  Debug builds: RMU_FFN3_LA=1.
    static const int la = rmu_env("RMU_FFN3_LA") ? atoi(rmu_env("RMU_FFN3_LA")) : 0;
    if (la) {
        if (ln_in) launch_ffn3_t<true, 4, 4>(x, L, eps, out, cu, batch, m_cap, s, out_tiled);
        else launch_ffn3_t<false, 4, 4>(x, L, eps, out, cu, batch, m_cap, s, out_tiled);
        return;
    }
#endif
    if (ln_in) launch_ffn3_t<true, 4, 0>(x, L, eps, out, cu, batch, m_cap, s, out_tiled);
    else launch_ffn3_t<false, 4, 0>(x, L, eps, out, cu, batch, m_cap, s, out_tiled);
}

template <int EPI>
static void launch_gemm3(const bf16* A, const bf16* W, const float* bias, const bf16* resid, bf16* out, const int* cu, int batch,
                         int N, int K, hipStream_t s, bool a_tiled = false, int64_t hm_stride = 0) {
    static const int n_wg = [] { int dev = 0, cus = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); return cus / 8 * 8; }();
#ifdef RMU_DEBUG_KERNELS
    static const bool want_dbg = rmu_env("RMU_G3_DBG") != nullptr;
    if (want_dbg) {
        static const hipError_t attr_dbg = hipFuncSetAttribute((const void*)k_gemm3<EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, g3::LDS_BYTES);
        (void)attr_dbg;
        static unsigned long long* dbg = nullptr;
        if (!dbg) { (void)hipMalloc((void**)&dbg, 64); (void)hipMemset(dbg, 0, 64); }
        static const int dflags = rmu_env("RMU_G3_FLAGS") ? atoi(rmu_env("RMU_G3_FLAGS")) : 0;   // 1: no DMA in the loop, 2: no fragment reads, 4: no epilogue (timing only)
        hipLaunchKernelGGL((k_gemm3<EPI, true>), dim3(n_wg), dim3(512), g3::LDS_BYTES, s, A, W, bias, resid, out, cu, batch, N, K, dbg, dflags | (a_tiled ? 256 : 0), (long)hm_stride);
        unsigned long long h[6];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, dbg, 48, hipMemcpyDeviceToHost);
        (void)hipMemset(dbg, 0, 64);
        const double nw = (double)h[0], stages = (double)h[5] / nw * (K / 32);
        fprintf(stderr, "[gemm3<%d> N=%d K=%d dbg] waves=%llu tiles/wg=%.1f ticks/wave: prologue=%.0f loop=%.0f (per stage %.0f; of which vmcnt+barrier %.0f, epilogue %.0f)\n",
                EPI, N, K, h[0], (double)h[5] / nw, h[1] / nw, h[2] / nw, h[2] / nw / stages, h[4] / nw / stages, h[3] / nw / stages);
        return;
    }
#endif
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)k_gemm3<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, g3::LDS_BYTES);
    (void)attr_rc;
    hipLaunchKernelGGL((k_gemm3<EPI, false>), dim3(n_wg), dim3(512), g3::LDS_BYTES, s, A, W, bias, resid, out, cu, batch, N, K, (unsigned long long*)nullptr, a_tiled ? 256 : 0, (long)hm_stride);
}

template <int KT>
static void launch_attn3(int batch, const bf16* qkv, const int* cu, bf16* ctx, bool ctx_tiled, int64_t hm_stride, hipStream_t s) {
    constexpr int lds = KT * 32 * 64 + 32 * (KT * 32 * 2 + 16);
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)k_attn3<KT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)attr_rc;
    const dim3 grid((unsigned)((batch + 7) / 8 * 8 * NH));
    hipLaunchKernelGGL(k_attn3<KT>, grid, dim3(256), lds, s, qkv, cu, batch, ctx, ctx_tiled ? 1 : 0, (long)hm_stride);
}

static void launch_qa(int batch, const bf16* x, bool x_tiled, const bf16* wstream, const float* bias, const int* cu, const int2* items, int items_cap,
                      bf16* ctx, bool ctx_tiled, hipStream_t s) {
    static const int dbg_flags = rmu_env("RMU_QA_DBG") ? atoi(rmu_env("RMU_QA_DBG")) : 0;
#ifdef RMU_DEBUG_KERNELS
    static const bool want_clk = rmu_env("RMU_QA_CLK") != nullptr;
    if (want_clk) {          // phase cycle counters (per wave, summed over the grid)
        static const hipError_t attr_dbg = hipFuncSetAttribute((const void*)k_qa<true>, hipFuncAttributeMaxDynamicSharedMemorySize, qa::LDS_BYTES);
        (void)attr_dbg;
        static unsigned long long* dbg = nullptr;
        if (!dbg) { (void)hipMalloc((void**)&dbg, 128); (void)hipMemsetAsync(dbg, 0, 128, s); }
        hipLaunchKernelGGL(k_qa<true>, dim3((unsigned)batch), dim3(512), qa::LDS_BYTES, s, x, x_tiled ? 1 : 0, wstream, bias, cu, items, (const int*)(items + items_cap),
                           ctx, ctx_tiled ? 1 : 0, dbg_flags, dbg);
        unsigned long long h[16];
        (void)hipMemcpyAsync(h, dbg, 128, hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        (void)hipMemsetAsync(dbg, 0, 128, s);
        for (int hf = 0; hf < 2; ++hf) {
            const double n = (double)h[hf * 8] > 0 ? (double)h[hf * 8] : 1.0;
            fprintf(stderr, "[k_qa clk half %d] waves %llu; cycles per wave: total %.0f = boundaries %.0f + projection %.0f + epilogue %.0f + pass1 %.0f + pass2a %.0f + pass2b/final %.0f\n", hf,
                    h[hf * 8], h[hf * 8 + 7] / n, h[hf * 8 + 1] / n, h[hf * 8 + 2] / n, h[hf * 8 + 3] / n, h[hf * 8 + 4] / n, h[hf * 8 + 5] / n, h[hf * 8 + 6] / n);
        }
        return;
    }
#endif
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)k_qa<false>, hipFuncAttributeMaxDynamicSharedMemorySize, qa::LDS_BYTES);
    (void)attr_rc;
    // one workgroup per item; their number is known on the device only: `batch` is its upper bound, the surplus workgroups return at once
    hipLaunchKernelGGL(k_qa<false>, dim3((unsigned)batch), dim3(512), qa::LDS_BYTES, s, x, x_tiled ? 1 : 0, wstream, bias, cu, items, (const int*)(items + items_cap),
                       ctx, ctx_tiled ? 1 : 0, dbg_flags, (unsigned long long*)nullptr);
}

template <int KT, bool LNA>
static void launch_qkv_attn_small(hipStream_t s, int batch, const bf16* A, const bf16* W, const float* bias, const int* cu, const float* lng,
                                  const float* lnb, float eps, float2* stats_out, bf16* ctx) {
    using C = QkvAttnCfg<KT>;
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)k_qkv_attn_small<KT, LNA>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    (void)attr_rc;
    hipLaunchKernelGGL((k_qkv_attn_small<KT, LNA>), dim3((unsigned)(batch * NH)), dim3(256), C::LDS_BYTES, s, A, W, bias, cu, batch, lng, lnb, eps, stats_out, ctx);
}

#ifdef RMU_DEBUG_KERNELS
template <int KT, int OCC>
static void launch_attn4(int batch, const bf16* qkv, const int* cu, bf16* ctx, bool ctx_tiled, int64_t hm_stride, hipStream_t s) {
    constexpr int lds = KT * 32 * 64 + 32 * (KT * 32 * 2 + 16);
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)k_attn4<KT, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)attr_rc;
    static const int n_cu = [] { int dev = 0, cus = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); return cus; }();
    const int items = (batch + 7) / 8 * 8 * NH;
    const int grid = std::min(items, std::max(8, n_cu * OCC / 8 * 8));
    hipLaunchKernelGGL((k_attn4<KT, OCC>), dim3((unsigned)grid), dim3(256), lds, s, qkv, cu, batch, ctx, ctx_tiled ? 1 : 0, (long)hm_stride);
}
template <int MAXT>
static void launch_attn(dim3 grid, const bf16* qkv, const int* cu, bf16* ctx, hipStream_t s) {
    constexpr int lds = MAXT * 16 * 80 + DH * (MAXT * 16 * 2 + 8);
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)k_attention<MAXT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)attr_rc;
    hipLaunchKernelGGL(k_attention<MAXT>, grid, dim3(256), lds, s, qkv, cu, ctx);
}
#endif

static int check_encode_args(rmu_bert_t* m, const void* ids, const void* lens, const void* out, int batch, int max_len, int mode, int64_t out_stride) {
    if (!m || !ids || !lens || !out) return bfail(RMU_E_INVALID, "rmu_bert_encode: null argument");
    if (batch < 1 || batch > 65535 || max_len < 1 || max_len > m->cfg.max_pos)
        return bfail(RMU_E_INVALID, "rmu_bert_encode: 1 <= batch <= 65535, 1 <= max_len <= max_pos");
    const int kind = mode & 0xff;
    if ((mode & ~(0xff | RMU_BERT_NO_NORMALIZE)) || kind > RMU_BERT_TOKENS)
        return bfail(RMU_E_INVALID, "rmu_bert_encode: mode must be RMU_BERT_POOL_MEAN / CE_LOGIT / POOL_CLS / TOKENS (| RMU_BERT_NO_NORMALIZE)");
    if (kind == RMU_BERT_CE_LOGIT && !m->cfg.has_head) return bfail(RMU_E_INVALID, "rmu_bert_encode: model has no classification head");
    if (kind != RMU_BERT_CE_LOGIT && out_stride < H) return bfail(RMU_E_INVALID, "rmu_bert_encode: out_stride < hidden");
    return RMU_OK;
}

// every launch of one forward on stream s (device pointers; no allocation, no synchronisation: capturable)
static void enqueue_forward(rmu_bert* m, const int32_t* ids, const int32_t* type_ids, const int32_t* lens, int batch, int max_len, int mode,
                            float* out_dev, int64_t out_stride, hipStream_t s) {
    const int kind = mode & 0xff;
    const bool normalize = !(mode & RMU_BERT_NO_NORMALIZE);
    const int64_t cap = (int64_t)batch * max_len;
    const float eps = m->cfg.ln_eps;

    // (thread 0 folds one partial per thread serially: a block no wider than the batch needs -- 1024 threads cost 12 us for 14 sequences)
    // (round 5) the interactive sizes: the sequence offsets are built inside k_embed_ln (RMU_SMALL_FUSE=0: the separate k_cu_seqlens launch)
    static const bool small_fuse = !(rmu_env("RMU_SMALL_FUSE") && atoi(rmu_env("RMU_SMALL_FUSE")) == 0);
    const bool cu_here = small_fuse && batch <= 256 && cap <= FOLD_TOKENS;
    if (cu_here) {
        hipLaunchKernelGGL(k_embed_ln<true>, dim3((unsigned)((cap + 3) / 4)), dim3(256), 0, s, (const int*)ids, (const int*)type_ids,
                           (const int*)nullptr, batch, max_len, m->wemb, m->pemb, m->temb, m->elng, m->elnb, eps, m->cfg.vocab_size,
                           m->cfg.type_vocab, m->h, (const int*)lens, m->cu);
    } else {
        hipLaunchKernelGGL(k_cu_seqlens, dim3(1), dim3(batch <= 64 ? 64 : batch <= 256 ? 256 : 1024), 0, s, (const int*)lens, batch, max_len, m->cu);
        // bulk batches: a workgroup per sequence (k_embed_ln_seq); RMU_EMBED_SEQ=0: a wave per (sequence, position) slot as before
        static const bool embed_seq = !(rmu_env("RMU_EMBED_SEQ") && atoi(rmu_env("RMU_EMBED_SEQ")) == 0);
        if (embed_seq && batch >= 512)
            hipLaunchKernelGGL(k_embed_ln_seq, dim3((unsigned)batch), dim3(256), 0, s, (const int*)ids, (const int*)type_ids, (const int*)m->cu, batch, max_len,
                               m->wemb, m->pemb, m->temb, m->elng, m->elnb, eps, m->cfg.vocab_size, m->cfg.type_vocab, m->h);
        else
        hipLaunchKernelGGL(k_embed_ln<false>, dim3((unsigned)((cap + 3) / 4)), dim3(256), 0, s, (const int*)ids, (const int*)type_ids,
                           (const int*)m->cu, batch, max_len, m->wemb, m->pemb, m->temb, m->elng, m->elnb, eps, m->cfg.vocab_size,
                           m->cfg.type_vocab, m->h);
    }
    const dim3 ln_grid((unsigned)((cap + 4 * LN_ROWS - 1) / (4 * LN_ROWS)));
    const dim3 at_grid(NH, (unsigned)batch);   // k_attention: one workgroup per (head, sequence); k_attn3: one per sequence
    size_t li = 0;
    bool h_in_tiled = false;                   // m->h as this layer reads it: row-major from k_embed_ln, tiled from a k_ffn3 that was told so
    // ---- the interactive sizes (one query; the <= 14 pairs of a rerank call, ~1.5k tokens; anything up to FOLD_TOKENS): five launches per
    // layer instead of seven -- both LayerNorms are folded into the k_gemm_small launches that consume them (see there): y1 = out-proj +
    // residual lives in m->y, y2 = FFN2 + residual in m->h1, neither is ever normalised in memory; one k_layernorm after the last layer
    // feeds the pooling heads.  Token blocks per workgroup: enough of them to fill the chip, few enough to amortise the weight rows a
    // workgroup keeps in registers (N / 32 feature blocks x cap / tb token blocks ~ 256-512 workgroups).
    static const bool small_ok_f = !(rmu_env("RMU_GEMM_SMALL") && atoi(rmu_env("RMU_GEMM_SMALL")) == 0);
    static const bool ln_fuse = !(rmu_env("RMU_LN_FUSE") && atoi(rmu_env("RMU_LN_FUSE")) == 0);
    // measured (tools/ce_probe.py, cross-encoder forward per call): 14 pairs / 1548 tokens 0.464-0.468 ms on the tiled kernels, 0.412-0.424 here;
    // 30 pairs / 3360 tokens (cap 4800: above the threshold) 0.50-0.51 tiled vs 0.65 here -- the fold pays up to ~2.5k tokens
    static const int64_t fold_tokens = rmu_env("RMU_FOLD_TOKENS") ? atoll(rmu_env("RMU_FOLD_TOKENS")) : FOLD_TOKENS;
    if (cap <= fold_tokens && max_len <= 256 && small_ok_f && ln_fuse) {
        auto tb_for = [&](int n_feature_blocks) {       // tokens per workgroup: ~384 workgroups in all, a multiple of 32, at least 32
            const int64_t blocks = std::max<int64_t>(1, 384 / n_feature_blocks);
            int64_t tb = (cap + blocks - 1) / blocks;
            tb = (tb + 31) / 32 * 32;
            return (int)std::min<int64_t>(std::max<int64_t>(tb, 32), 1024);
        };
        const int tq = tb_for(3 * H / 32), th = tb_for(H / 32), tf = tb_for(FF / 32);
        const int* cu = m->cu;
        bf16 *y1 = m->y, *y2 = m->h1;
        const BertLayer* prev = nullptr;
        // k_gemm_mid (a wave per output tile, both operands as MFMA fragments straight from global memory, nothing shared): measured and NOT
        // adopted -- 14 pairs / 1548 tokens: 0.58 ms per forward against 0.42 with k_gemm_small and 0.47 with the tiled kernels; per launch
        // 8-28 us: a fragment-shaped load touches 32 cache lines for 1 KiB and the CU's address path is paid per line (debug builds: RMU_GEMM_MID=1)
#ifdef RMU_DEBUG_KERNELS
        static const bool mid_env = rmu_env("RMU_GEMM_MID") && atoi(rmu_env("RMU_GEMM_MID")) != 0;
        const bool mid = cap > SMALL_M && mid_env;
#endif
        for (const BertLayer& L : m->layers) {
#ifdef RMU_DEBUG_KERNELS
            if (mid) {
                if (!prev) launch_mid<EPI_BIAS, H>(s, cap, m->h, L.wqkv, L.bqkv, nullptr, m->qkv, cu, batch, 3 * H);
                else launch_mid<EPI_BIAS, H, true>(s, cap, y2, L.wqkv, L.bqkv, nullptr, m->qkv, cu, batch, 3 * H, prev->ln2g, prev->ln2b, eps, m->st2);
                if (max_len <= 128) launch_attn3<4>(batch, m->qkv, m->cu, m->ctx, false, 0, s);
                else launch_attn3<8>(batch, m->qkv, m->cu, m->ctx, false, 0, s);
                if (!prev) launch_mid<EPI_RESID, H>(s, cap, m->ctx, L.wo, L.bo, m->h, y1, cu, batch, H);
                else launch_mid<EPI_RESID, H, false, true>(s, cap, m->ctx, L.wo, L.bo, y2, y1, cu, batch, H, nullptr, nullptr, eps, nullptr, prev->ln2g, prev->ln2b, m->st2);
                launch_mid<EPI_GELU, H, true>(s, cap, y1, L.w1, L.b1, nullptr, m->mid, cu, batch, FF, L.ln1g, L.ln1b, eps, m->st1);
                launch_mid<EPI_RESID, FF, false, true>(s, cap, m->mid, L.w2, L.b2, y1, y2, cu, batch, H, nullptr, nullptr, eps, nullptr, L.ln1g, L.ln1b, m->st1);
                prev = &L;
                continue;
            }
#endif
            // (round 5) up to QKV_ATTN_TOKENS tokens the QKV projection and the attention are ONE launch
            // (k_qkv_attn_small: a workgroup per (head, sequence); bit-identical to the pair below); RMU_QKV_ATTN_TOKENS=0 keeps them apart
            static const int64_t qa_tokens = rmu_env("RMU_QKV_ATTN_TOKENS") ? atoll(rmu_env("RMU_QKV_ATTN_TOKENS")) : QKV_ATTN_TOKENS;
            // (round 6) ... but not below QKV_ATTN_MIN_TOKENS: one short query (16-48 tokens) runs 2.6-4.5 % faster on the two launches (a
            // workgroup per (head, sequence) is 12 workgroups for one query), the 14-pair rerank call 3.7 % faster on the fused one
            // (profiles/r06_ab_interactive.txt).  Bit-identical either way.  RMU_QKV_ATTN_MIN=0: fused from the first token on.
            static const int64_t qa_min = rmu_env("RMU_QKV_ATTN_MIN") ? atoll(rmu_env("RMU_QKV_ATTN_MIN")) : QKV_ATTN_MIN_TOKENS;
            if (cap <= qa_tokens && cap > qa_min) {
                if (max_len <= 128) {
                    if (!prev) launch_qkv_attn_small<4, false>(s, batch, m->h, L.wqkv, L.bqkv, cu, nullptr, nullptr, eps, nullptr, m->ctx);
                    else launch_qkv_attn_small<4, true>(s, batch, y2, L.wqkv, L.bqkv, cu, prev->ln2g, prev->ln2b, eps, m->st2, m->ctx);
                } else {
                    if (!prev) launch_qkv_attn_small<8, false>(s, batch, m->h, L.wqkv, L.bqkv, cu, nullptr, nullptr, eps, nullptr, m->ctx);
                    else launch_qkv_attn_small<8, true>(s, batch, y2, L.wqkv, L.bqkv, cu, prev->ln2g, prev->ln2b, eps, m->st2, m->ctx);
                }
            } else {
            if (!prev) launch_small<EPI_BIAS, H>(s, cap, tq, m->h, L.wqkv, L.bqkv, nullptr, m->qkv, cu, batch, 3 * H);
            else launch_small<EPI_BIAS, H, true>(s, cap, tq, y2, L.wqkv, L.bqkv, nullptr, m->qkv, cu, batch, 3 * H, prev->ln2g, prev->ln2b, eps, m->st2);
            if (max_len <= 128) launch_attn3<4>(batch, m->qkv, m->cu, m->ctx, false, 0, s);
            else launch_attn3<8>(batch, m->qkv, m->cu, m->ctx, false, 0, s);
            }
            if (!prev) launch_small<EPI_RESID, H>(s, cap, th, m->ctx, L.wo, L.bo, m->h, y1, cu, batch, H);
            else launch_small<EPI_RESID, H, false, true>(s, cap, th, m->ctx, L.wo, L.bo, y2, y1, cu, batch, H, nullptr, nullptr, eps, nullptr, prev->ln2g, prev->ln2b, m->st2);
            launch_small<EPI_GELU, H, true>(s, cap, tf, y1, L.w1, L.b1, nullptr, m->mid, cu, batch, FF, L.ln1g, L.ln1b, eps, m->st1);
            launch_small<EPI_RESID, FF, false, true>(s, cap, th, m->mid, L.w2, L.b2, y1, y2, cu, batch, H, nullptr, nullptr, eps, nullptr, L.ln1g, L.ln1b, m->st1);
            prev = &L;
        }
        // (measured and dropped, round 5: the last LayerNorm inside the pooling launch -- one wave per sequence normalising its rows four at a
        // time -- made a 16-token query forward SLOWER, 0.182 vs 0.171-0.176 ms: k_layernorm spreads the rows over the chip, the fold serialises them)
        if (prev) hipLaunchKernelGGL(k_layernorm, ln_grid, dim3(256), 0, s, (const bf16*)y2, (const int*)m->cu, batch, prev->ln2g, prev->ln2b, eps, m->h);
    } else {
    // (round 6) RMU_QA=1: QKV projection + attention as ONE launch per layer (k_qa) for sequences up to 256 tokens.  Built, bit-identical to
    // k_gemm3 + k_attn3 (tests), and NOT the default: 2.31 ms per layer in its lockstep form, 2.99 as a ping-pong of the two wave halves,
    // against 1.90 for the pair -- at 8 waves per CU (the 96 token-fragment registers) one wave's attention chain takes ~6.6k cycles per
    // head whatever runs beside it (profiles/r06_qa_fusion.md)
    static const bool qa_env = rmu_env("RMU_QA") && atoi(rmu_env("RMU_QA")) != 0;
    static const int g3_mask0 = rmu_env("RMU_GEMM3") ? atoi(rmu_env("RMU_GEMM3")) : 1;
    const bool qa_on = qa_env && (g3_mask0 & 1) && max_len <= 256 && cap > g3_min_tokens() && batch <= 65536 && m->qa_items != nullptr;
    if (qa_on)
        hipLaunchKernelGGL(k_qa_items, dim3(1), dim3(1024), 0, s, (const int*)m->cu, batch, m->qa_items, (int*)(m->qa_items + m->ws_batch));
    for (const BertLayer& L : m->layers) {
        ++li;
        static const int g3_mask = rmu_env("RMU_GEMM3") ? atoi(rmu_env("RMU_GEMM3")) : 1;   // k_gemm3 for: bit 0 QKV (default: 1.22 vs 1.38 ms), bit 1 out-proj (0.67 vs 0.61), bit 2 FFN1 + FFN2 instead of k_ffn_fused (3.6 vs 3.35)
        // The round-1/2 kernels k_attention and k_ffn_fused are instantiated in debug builds only (RMU_ATTN_V=1, RMU_FFN_V=1: the A/B
        // numbers of DESIGN.md); the product library carries the kernels it takes by default plus k_ffn2 (RMU_FFN_V=2).
#ifdef RMU_DEBUG_KERNELS
        static const int attn_v = rmu_env("RMU_ATTN_V") ? atoi(rmu_env("RMU_ATTN_V")) : 3;
#else
        constexpr int attn_v = 3;
#endif
        // QKV head-major when k_gemm3 writes it and k_attn3 reads it (RMU_QKV_HM=0: row-major); the stride between (part, head) planes is the
        // workspace's token capacity
        static const bool hm_env = !(rmu_env("RMU_QKV_HM") && atoi(rmu_env("RMU_QKV_HM")) == 0);
        const int64_t hm_stride = (hm_env && (g3_mask & 1) && cap > g3_min_tokens() && attn_v == 3) ? m->ws_tokens : 0;
        if (qa_on && attn_v == 3) { /* below */ }
        else if ((g3_mask & 1) && cap > g3_min_tokens()) launch_gemm3<EPI_BIAS>(m->h, L.wqkv_t, L.bqkv, nullptr, m->qkv, m->cu, batch, 3 * H, H, s, h_in_tiled, hm_stride);
        else launch_gemm<EPI_BIAS>(m->h, L.wqkv, L.bqkv, nullptr, m->qkv, m->cu, batch, cap, 3 * H, H, s);
        // big batches: k_attn3 writes ctx as the 1-KiB operand blocks the out-proj GEMM's LDS-DMA reads whole (RMU_CTX_TILED=0: row-major)
        static const bool tiled_env = !(rmu_env("RMU_CTX_TILED") && atoi(rmu_env("RMU_CTX_TILED")) == 0);
        const bool ctx_tiled = tiled_env && attn_v == 3 && !(g3_mask & 2) && cap > 32768;
#ifdef RMU_DEBUG_KERNELS
        // k_attn4, the PERSISTENT form (round 4): measured SLOWER in every variant -- 8192 chunks, per layer under rocprofv3: k_attn3 837-851 us;
        // k_attn4 with the next item's pieces prefetched into registers 1372-1384 us (2 workgroups per CU), with an L2 touch of the next item
        // and k_attn3's occupancy 1244 us; bench.py's embed leg 33.8-34.0 ms vs 35.7-36.2.  The hardware dispatcher already overlaps one
        // workgroup's cold start with three others' arithmetic AND balances the 16..256-token items dynamically; a static item list
        // loses both.  Debug builds: RMU_ATTN4=1, RMU_ATTN4_OCC=2|3|4.
        static const int attn4 = rmu_env("RMU_ATTN4") ? atoi(rmu_env("RMU_ATTN4")) : 0;
        if (attn_v == 3 && attn4 && batch * NH > 4096 && max_len <= 256) {
            static const int occ = rmu_env("RMU_ATTN4_OCC") ? atoi(rmu_env("RMU_ATTN4_OCC")) : 3;
            if (max_len <= 128) {
                if (occ == 2) launch_attn4<4, 2>(batch, m->qkv, m->cu, m->ctx, ctx_tiled, hm_stride, s);
                else if (occ == 4) launch_attn4<4, 4>(batch, m->qkv, m->cu, m->ctx, ctx_tiled, hm_stride, s);
                else launch_attn4<4, 3>(batch, m->qkv, m->cu, m->ctx, ctx_tiled, hm_stride, s);
            } else if (occ == 4) launch_attn4<8, 4>(batch, m->qkv, m->cu, m->ctx, ctx_tiled, hm_stride, s);
            else launch_attn4<8, 2>(batch, m->qkv, m->cu, m->ctx, ctx_tiled, hm_stride, s);
        } else
#endif
        if (qa_on && attn_v == 3) {
            launch_qa(batch, m->h, h_in_tiled, L.wqkv_s, L.bqkv, m->cu, m->qa_items, m->ws_batch, m->ctx, ctx_tiled, s);
        } else if (attn_v == 3) {
            if (max_len <= 128) launch_attn3<4>(batch, m->qkv, m->cu, m->ctx, ctx_tiled, hm_stride, s);
            else if (max_len <= 256) launch_attn3<8>(batch, m->qkv, m->cu, m->ctx, ctx_tiled, hm_stride, s);
            else launch_attn3<16>(batch, m->qkv, m->cu, m->ctx, ctx_tiled, hm_stride, s);
        }
#ifdef RMU_DEBUG_KERNELS
        else if (max_len <= 128) launch_attn<8>(at_grid, m->qkv, m->cu, m->ctx, s);
        else if (max_len <= 256) launch_attn<16>(at_grid, m->qkv, m->cu, m->ctx, s);
        else launch_attn<32>(at_grid, m->qkv, m->cu, m->ctx, s);
#endif
#ifdef RMU_DEBUG_KERNELS
        // RMU_FFN_OUTPROJ=1: the out-proj GEMM runs inside k_ffn3's prologue (no launch, no pre-LN sum in memory)
        static const bool op_env = rmu_env("RMU_FFN_OUTPROJ") && atoi(rmu_env("RMU_FFN_OUTPROJ")) != 0;
        static const int fused_env0 = rmu_env("RMU_FUSED_FFN") ? atoi(rmu_env("RMU_FUSED_FFN")) : -1;
        static const bool lnin0 = !(rmu_env("RMU_FFN_LNIN") && atoi(rmu_env("RMU_FFN_LNIN")) == 0);
        static const bool v3 = !(rmu_env("RMU_FFN_V") && atoi(rmu_env("RMU_FFN_V")) != 3);
        if (op_env && v3 && lnin0 && !(g3_mask & 6) && attn_v == 3 && (fused_env0 < 0 ? cap > 16384 : fused_env0 != 0)) {
            static const bool h_env0 = !(rmu_env("RMU_H_TILED") && atoi(rmu_env("RMU_H_TILED")) == 0);
            const bool h_out_tiled = h_env0 && ctx_tiled && (g3_mask & 1) && li < m->layers.size();
            launch_ffn3_outproj(m->ctx, ctx_tiled, m->h, h_in_tiled, L, eps, m->h1, m->cu, batch, cap, s, h_out_tiled);
            std::swap(m->h, m->h1);                // the kernel reads the residual h while other workgroups write the layer output: two buffers
            h_in_tiled = h_out_tiled;
            continue;
        }
#endif
        if (g3_mask & 2) launch_gemm3<EPI_RESID>(m->ctx, L.wo_t, L.bo, m->h, m->y, m->cu, batch, H, H, s);
        else if (ctx_tiled) launch_gemm_cfg<EPI_RESID, 4, 32, 2, true>(m->ctx, L.wo_t, L.bo, m->h, m->y, m->cu, batch, cap, H, H, s, h_in_tiled);
        else launch_gemm<EPI_RESID>(m->ctx, L.wo, L.bo, m->h, m->y, m->cu, batch, cap, H, H, s);
        // The fused kernels give each 128-token tile to ONE workgroup, which then streams all 2.36 MB of FFN weights through one
        // CU (~100 us per layer whatever the batch): below ~128 tiles most CUs would idle and the GEMM pair, whose feature tiles
        // spread over the chip, is faster (measured: 8k tokens 0.82 vs 0.95 ms per forward, one 16-token query 0.44 vs 0.63 ms;
        // 32k tokens 1.70 vs 1.45).  RMU_FUSED_FFN=0 / 1 forces either path.
        static const int fused_env = rmu_env("RMU_FUSED_FFN") ? atoi(rmu_env("RMU_FUSED_FFN")) : -1;
        const bool fused_ffn = !(g3_mask & 4) && (fused_env < 0 ? cap > 16384 : fused_env != 0);
        // k_ffn3 (default; two waves per SIMD) and k_ffn2 (RMU_FFN_V=2; one) also take LayerNorm 1 into their prologue: the out-proj sum y
        // goes straight in (RMU_FFN_LNIN=0: separate k_layernorm launch; RMU_FFN_V=1: the round-2 kernel k_ffn_fused -- both kept
        // this round for the A/B numbers in DESIGN.md)
#ifdef RMU_DEBUG_KERNELS
        static const int ffn_v = rmu_env("RMU_FFN_V") ? atoi(rmu_env("RMU_FFN_V")) : 3;
#else
        static const int ffn_v = rmu_env("RMU_FFN_V") && atoi(rmu_env("RMU_FFN_V")) == 2 ? 2 : 3;
#endif
        static const bool ln_in = !(rmu_env("RMU_FFN_LNIN") && atoi(rmu_env("RMU_FFN_LNIN")) == 0);
        if (fused_ffn && ffn_v >= 2 && ln_in) {
            // Between layers h travels TILED (1-KiB blocks of 16 tokens x 32 features: the next QKV GEMM's A pieces and the next out-proj's
            // residual pieces become whole contiguous KiB); the last layer writes row-major for the pooling heads.  RMU_H_TILED=0: never.
            static const bool h_env = !(rmu_env("RMU_H_TILED") && atoi(rmu_env("RMU_H_TILED")) == 0);
            const bool h_out_tiled = h_env && ffn_v == 3 && ctx_tiled && (g3_mask & 1) && li < m->layers.size();
            if (ffn_v == 3) launch_ffn3(m->y, true, L, eps, m->h, m->cu, batch, cap, s, h_out_tiled);
            else launch_ffn2(m->y, true, L, eps, m->h, m->cu, batch, cap, s);
            h_in_tiled = h_out_tiled;
            continue;
        }
        h_in_tiled = false;
        hipLaunchKernelGGL(k_layernorm, ln_grid, dim3(256), 0, s, (const bf16*)m->y, (const int*)m->cu, batch, L.ln1g, L.ln1b, eps, m->h1);
        if (g3_mask & 4) {
            launch_gemm3<EPI_GELU>(m->h1, L.w1_t, L.b1, nullptr, m->mid, m->cu, batch, FF, H, s);
            launch_gemm3<EPI_RESID>(m->mid, L.w2_t, L.b2, m->h1, m->y, m->cu, batch, H, FF, s);
            hipLaunchKernelGGL(k_layernorm, ln_grid, dim3(256), 0, s, (const bf16*)m->y, (const int*)m->cu, batch, L.ln2g, L.ln2b, eps, m->h);
            continue;
        }
        if (fused_ffn) {          // FFN1 + GELU + FFN2 + residual + LayerNorm in one kernel: the 1536-wide intermediate stays on chip
            if (ffn_v == 3) launch_ffn3(m->h1, false, L, eps, m->h, m->cu, batch, cap, s);
            else if (ffn_v == 2) launch_ffn2(m->h1, false, L, eps, m->h, m->cu, batch, cap, s);
#ifdef RMU_DEBUG_KERNELS
            else launch_ffn_fused(m->h1, L, eps, m->h, m->cu, batch, cap, s);
#endif
            continue;
        }
        launch_gemm<EPI_GELU>(m->h1, L.w1, L.b1, nullptr, m->mid, m->cu, batch, cap, FF, H, s);
        launch_gemm<EPI_RESID>(m->mid, L.w2, L.b2, m->h1, m->y, m->cu, batch, cap, H, FF, s);
        hipLaunchKernelGGL(k_layernorm, ln_grid, dim3(256), 0, s, (const bf16*)m->y, (const int*)m->cu, batch, L.ln2g, L.ln2b, eps, m->h);
    }
    }
    if (kind == RMU_BERT_POOL_MEAN || kind == RMU_BERT_POOL_CLS)
        hipLaunchKernelGGL(k_pool, dim3((unsigned)batch), dim3(64), 0, s, (const bf16*)m->h, (const int*)m->cu, out_dev, out_stride,
                           kind == RMU_BERT_POOL_CLS ? 1 : 0, normalize ? 1 : 0);
    else if (kind == RMU_BERT_TOKENS)
        hipLaunchKernelGGL(k_tokens_out, dim3((unsigned)((cap * (H / 8) + 255) / 256)), dim3(256), 0, s, (const bf16*)m->h, (const int*)m->cu, batch,
                           out_dev, out_stride);
    else
        hipLaunchKernelGGL(k_cls_head, dim3((unsigned)batch), dim3(512), 0, s, (const bf16*)m->h, (const int*)m->cu, m->wp, m->bp, m->wc, m->bc, out_dev);
}

extern "C" int rmu_bert_encode(rmu_bert_t* m, const int32_t* ids, const int32_t* type_ids, const int32_t* lens, int batch,
                               int max_len, int mode, float* out_dev, int64_t out_stride, uint64_t hip_stream) {
    RMU_ENTRY();
    int rc = check_encode_args(m, ids, lens, out_dev, batch, max_len, mode, out_stride);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(m->mu);
    (void)hipGetLastError();                     // (an error some earlier call of this thread left behind -- the caller's framework polls events -- is not this call's)
    rc = ensure_ws(m, (int64_t)batch * max_len, batch);
    if (rc) return bfail(rc, "rmu_bert_encode: workspace");
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : m->stream;
    if ((rc = order_behind_tail(m, s))) return bfail(rc, "rmu_bert_encode: ordering behind the forward in flight");
    enqueue_forward(m, ids, type_ids, lens, batch, max_len, mode, out_dev, out_stride, s);
    B_TRY(hipGetLastError());
    if (!hip_stream) {
        B_TRY(hipStreamSynchronize(s));
        m->tail_set = false;                     // (everything this context ever enqueued has finished)
    } else if ((rc = mark_tail(m, s))) {
        (void)hipStreamSynchronize(s);           // no event: fall back to "complete on return"
        m->tail_set = false;
    }
    return RMU_OK;
}

// The interactive path (embed_query, a handful of passages to rerank) from HOST buffers.  A forward at that size is ~45 launches of
// 1-15 us of work each: launch-latency-bound.  The first call of a shape runs eagerly, the second captures H2D + launches + D2H into a
// hipGraph, every later one is ONE hipGraphLaunch.  Shapes are (batch, max_len, mode, token types): callers bucket them (the Python
// binding pads batch and max_len up to a few sizes -- padded sequences have length 0 and cost nothing in the packed-token kernels).
// The cache holds MAX_GRAPHS shapes, least recently used out.  m->mu is held by the caller; the forward is left IN FLIGHT on
// m->stream (result in m->d_out, and on its way to m->h_out): the caller synchronises.
static int host_forward_locked(rmu_bert* m, const int32_t* ids, const int32_t* type_ids, const int32_t* lens, int batch, int max_len, int mode,
                               const char* who) {
    const int64_t cap = (int64_t)batch * max_len;
    const int kind = mode & 0xff;
    if (cap > HOST_TOKENS) return bfail(RMU_E_INVALID, std::string(who) + ": batch * max_len must be <= 4096 (use rmu_bert_encode for bulk work)");
    if (kind != RMU_BERT_CE_LOGIT && (kind == RMU_BERT_TOKENS ? cap : (int64_t)batch) > 256)
        return bfail(RMU_E_INVALID, std::string(who) + ": at most 256 result rows (pooled vectors, or token states of a <= 256-token call)");
    int rc = ensure_ws(m, cap, batch);
    if (rc) return bfail(rc, std::string(who) + ": workspace");
    if (!m->h_in) {
        if (hipHostMalloc((void**)&m->h_in, SMALL_IN_INTS * sizeof(int32_t)) != hipSuccess || hipHostMalloc((void**)&m->h_out, SMALL_OUT_FLOATS * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&m->d_in, SMALL_IN_INTS * sizeof(int32_t)) != hipSuccess || hipMalloc((void**)&m->d_out, SMALL_OUT_FLOATS * sizeof(float)) != hipSuccess)
            return bfail(RMU_E_OOM, std::string(who) + ": staging buffers");
    }
    // staging layout: ids [cap] | type ids [cap] | lens [batch]
    memcpy(m->h_in, ids, (size_t)cap * 4);
    if (type_ids) memcpy(m->h_in + cap, type_ids, (size_t)cap * 4);
    memcpy(m->h_in + 2 * cap, lens, (size_t)batch * 4);
    const int64_t rows_out = kind == RMU_BERT_TOKENS ? cap : batch;                       // (the graph copies the shape's upper bound)
    const size_t out_floats = kind == RMU_BERT_CE_LOGIT ? (size_t)batch : (size_t)rows_out * H;
    const size_t in_bytes = (size_t)(2 * cap + batch) * 4;
    hipStream_t s = m->stream;
    if ((rc = order_behind_tail(m, s))) return bfail(rc, std::string(who) + ": ordering behind the forward in flight");
    auto enqueue_all = [&]() {
        (void)hipMemcpyAsync(m->d_in, m->h_in, in_bytes, hipMemcpyHostToDevice, s);
        enqueue_forward(m, m->d_in, type_ids ? m->d_in + cap : nullptr, m->d_in + 2 * cap, batch, max_len, mode, m->d_out, kind == RMU_BERT_CE_LOGIT ? 1 : H, s);
        (void)hipMemcpyAsync(m->h_out, m->d_out, out_floats * sizeof(float), hipMemcpyDeviceToHost, s);
    };
    static const bool use_graph = !(rmu_env_kill("RMU_GRAPH") && atoi(rmu_env_kill("RMU_GRAPH")) == 0);
    const uint64_t key = ((uint64_t)batch << 32) | ((uint64_t)max_len << 16) | ((uint64_t)(mode & 0xfff) << 1) | (type_ids ? 1u : 0u);
    if (use_graph && !m->graphs.count(key) && m->graphs.size() >= MAX_GRAPHS) {          // full: the least recently used shape goes
        auto victim = m->graphs.begin();
        for (auto it = m->graphs.begin(); it != m->graphs.end(); ++it)
            if (it->second.last_use < victim->second.last_use) victim = it;
        B_TRY(hipStreamSynchronize(s));                                                   // (a replay of it may still be running)
        if (victim->second.exec) (void)hipGraphExecDestroy(victim->second.exec);
        if (victim->second.graph) (void)hipGraphDestroy(victim->second.graph);
        m->graphs.erase(victim);
    }
    rmu_bert::SmallGraph& g = m->graphs[key];
    g.last_use = ++m->graph_clock;
    auto clear_errors = [] { for (int i = 0; i < 8 && hipGetLastError() != hipSuccess; ++i) {} };
    // Whatever state a failed capture / instantiation / replay left behind: end the capture if the stream is still in one, drop the pieces,
    // clear the thread's error, and never capture this shape on this context again -- the call falls back to the eager launches.  (Seen once
    // in the round's runs: four threads on clone contexts + a bulk encode, a capture reported "operation failed due to a previous error
    // during capture" and the call failed with it.  A graph is an optimisation: losing it must not cost the call its result.)
    auto abandon_graph = [&]() {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
            hipGraph_t junk = nullptr;
            (void)hipStreamEndCapture(s, &junk);
            if (junk) (void)hipGraphDestroy(junk);
        }
        if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
        if (g.graph) { (void)hipGraphDestroy(g.graph); g.graph = nullptr; }
        g.warm = false;
        g.no_graph = true;
        clear_errors();
        st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {      // still capturing: this stream is lost
            clear_errors();
            hipStream_t ns = nullptr;
            if (hipStreamCreateWithFlags(&ns, hipStreamNonBlocking) == hipSuccess) {
                (void)hipStreamDestroy(m->stream);
                m->stream = ns;
                s = ns;
            }
            clear_errors();
        }
    };
    const bool may_graph = use_graph && !g.no_graph;
    bool launched = false;
    clear_errors();                                    // (an error some earlier call of this thread left behind is not this call's)
    if (may_graph && g.exec) {
        if (hipGraphLaunch(g.exec, s) == hipSuccess) launched = true;
        else abandon_graph();
    } else if (may_graph && g.warm) {
        // second call of this shape: capture (every function attribute / first-use static of the launchers is set by now).  ONE capture at a
        // time in the process: two threads capturing on their clone contexts at the same moment failed together once ("invalid argument" in
        // one, "operation failed due to a previous error during capture" in the other) although both captures are thread-local
        bool ok;
        {
            std::lock_guard<std::mutex> cap(rmu_capture_mutex());      // (one capture of this library at a time, rmu_common.h)
            ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                enqueue_all();
                ok = hipStreamEndCapture(s, &g.graph) == hipSuccess && g.graph != nullptr;
                if (ok) ok = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0) == hipSuccess;
            }
        }
        if (ok) ok = hipGraphLaunch(g.exec, s) == hipSuccess;
        if (ok) launched = true;
        else abandon_graph();                          // no graph for this shape: stay eager (and do not try again)
    }
    if (!launched) {
        enqueue_all();
        if (may_graph && !g.no_graph) g.warm = true;
    }
    if (hipGetLastError() != hipSuccess) {
        // one more attempt, eagerly, on a stream known to be out of capture and drained (the launches are idempotent: the same inputs to the
        // same outputs); only a second failure is the call's
        abandon_graph();
        (void)hipStreamSynchronize(s);
        clear_errors();
        enqueue_all();
        B_TRY(hipGetLastError());
    }
    return RMU_OK;
}

extern "C" int rmu_bert_encode_host(rmu_bert_t* m, const int32_t* ids, const int32_t* type_ids, const int32_t* lens, int batch, int max_len,
                                    int mode, float* out_host, int64_t out_stride) {
    RMU_ENTRY();
    int rc = check_encode_args(m, ids, lens, out_host, batch, max_len, mode, out_stride);
    if (rc) return rc;
    const int kind = mode & 0xff;
    std::unique_lock<std::mutex> lk;
    m = acquire_ctx(m, lk);
    rc = host_forward_locked(m, ids, type_ids, lens, batch, max_len, mode, "rmu_bert_encode_host");
    if (rc) return rc;
    B_TRY(hipStreamSynchronize(m->stream));
    if (kind == RMU_BERT_CE_LOGIT) {
        memcpy(out_host, m->h_out, (size_t)batch * sizeof(float));
    } else {
        int64_t n_tok = 0;
        for (int b = 0; b < batch; ++b) n_tok += std::min(std::max(lens[b], 0), max_len);
        const int64_t rows = kind == RMU_BERT_TOKENS ? n_tok : batch;
        for (int64_t r = 0; r < rows; ++r) memcpy(out_host + r * out_stride, m->h_out + r * H, H * sizeof(float));
    }
    return RMU_OK;
}

// The reference's per-request retrieval in ONE call with ONE synchronisation (VectorStoreRetriever.invoke, server/RAGHelper.py:497-499:
// embed_query -> dense top-fetch_k -> maximal_marginal_relevance): the query's token ids in, row ids out.  The forward is the graph
// replay above; its pooled vector stays on the device and feeds the search + selection enqueued behind it on the same stream
// (rmu_api.hip), so the vector's round trip to the host, the second synchronisation and a second host call are gone.
extern "C" int rmu_index_search_mmr_dev_(rmu_index_t* idx, const float* q_dev, int64_t nq, int fetch_k, int k, double lambda_mult, int64_t row_base,
                                         int64_t* out_rows, float* out_scores, void* hip_stream);
extern "C" int rmu_bert_search_mmr(rmu_bert_t* m, rmu_index_t* idx, const int32_t* ids, const int32_t* type_ids, const int32_t* lens, int batch,
                                   int max_len, int mode, int fetch_k, int k, double lambda_mult, int64_t row_base, int64_t* out_rows,
                                   float* out_scores, float* out_vecs) {
    RMU_ENTRY();
    if (!idx || !out_rows) return bfail(RMU_E_INVALID, "rmu_bert_search_mmr: null argument");
    int rc = check_encode_args(m, ids, lens, out_rows, batch, max_len, mode, H);
    if (rc) return rc;
    const int kind = mode & 0xff;
    if (kind != RMU_BERT_POOL_MEAN && kind != RMU_BERT_POOL_CLS) return bfail(RMU_E_INVALID, "rmu_bert_search_mmr: mode must be a pooling mode");
    std::unique_lock<std::mutex> lk;
    m = acquire_ctx(m, lk);
    rc = host_forward_locked(m, ids, type_ids, lens, batch, max_len, mode, "rmu_bert_search_mmr");
    if (rc) return rc;
    // drains m->stream: forward, search, selection and the copies of the results
    rc = rmu_index_search_mmr_dev_(idx, m->d_out, batch, fetch_k, k, lambda_mult, row_base, out_rows, out_scores, (void*)m->stream);
    if (rc) { (void)hipStreamSynchronize(m->stream); return rc; }
    if (out_vecs) memcpy(out_vecs, m->h_out, (size_t)batch * H * sizeof(float));
    return RMU_OK;
}
