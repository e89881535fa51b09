// pmx_critic.hip -- the feed-forward half of the critic's post-LN encoder layer (nn.TransformerEncoderLayer with d_model 32,
// dim_feedforward 128, ReLU, dropout 0; pacman_mappo_resnet.py:126-141) as ONE forward and ONE backward kernel:
//     y = LayerNorm(x + W2 relu(W1 x + b1) + b2)
// on [tokens][32] bf16 rows.  Tokens are independent, so a wavefront owns tiles of 16 tokens and keeps everything in
// registers: every product is a v_mfma_f32_16x16x32_bf16 with the FEATURES on the rows and the TOKENS on the columns
// (D[feature][token] = A[feature][k] x B[k][token]); the B operand of the first product is one 16-byte load per lane from
// the token row, and the accumulators of one product, converted to bf16, ARE the B operand of the next one -- the
// contraction index may be enumerated in any order as long as the A operand (the weights, pre-packed) uses the same one:
//     hidden order   phi(s, g, j) = 32 s + 16 (j >> 2) + 4 g + (j & 3)     k-step s, lane group g = lane >> 4, element j
//     output order   psi(m, g, r) = 8 g + 4 m + r                         so that a lane ends up with features 8g .. 8g+7,
// exactly the ones it loaded -- the residual add, LayerNorm (8 values per lane + 2 shuffles over the 4 lane groups of a
// token) and the 16-byte store are lane-local.  No LDS in the forward kernel, the hidden activations never exist in memory.
//
// Backward recomputes the forward from x (nothing is saved), runs LayerNorm's and the two products' input gradients the same
// way (dH = W2^T dF, dX = W1^T dH + dz) and forms the weight gradients -- contractions over TOKENS -- from a 32-token
// staging area in LDS read back through ds_read_b64_tr_b16; their 32 output tiles (128 accumulator registers) stay in
// registers across all tiles of the wave and leave through one block-level LDS sum into the block's partial row (plain
// stores; a row-sum kernel adds the rows).
#pragma clang fp contract(fast)
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "../../include/pmx.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

// packed-parameter buffer (bytes)
constexpr int FRAG = 64 * 8;                                 // bf16 elements of one operand fragment
constexpr size_t P_A1 = 0;                                   // W1 forward     [8 hidden tiles]
constexpr size_t P_A2 = P_A1 + 8 * FRAG * 2;                 // W2 forward     [2 output halves][4 k-steps]
constexpr size_t P_A3 = P_A2 + 8 * FRAG * 2;                 // W2^T           [8 hidden tiles]
constexpr size_t P_A4 = P_A3 + 8 * FRAG * 2;                 // W1^T           [2 input halves][4 k-steps]
constexpr size_t P_B1 = P_A4 + 8 * FRAG * 2;                 // float b1[128], b2[32], gamma[32], beta[32]
constexpr size_t P_BYTES = P_B1 + (128 + 32 + 32 + 32) * 4;
static_assert(P_BYTES == PMX_FFN_PACK_BYTES, "include/pmx.h and pmx_critic.hip disagree on the pack size");
// gradient buffer (floats): dW2 [32][128], dW1 [128][32], db1 [128], db2 [32], dgamma [32], dbeta [32]
constexpr int G_W2 = 0, G_W1 = G_W2 + 32 * 128, G_B1 = G_W1 + 128 * 32, G_B2 = G_B1 + 128, G_GAMMA = G_B2 + 32, G_BETA = G_GAMMA + 32;
constexpr int G_FLOATS = G_BETA + 32;
static_assert(G_FLOATS == PMX_FFN_GRAD_FLOATS, "include/pmx.h and pmx_critic.hip disagree on the gradient size");

__device__ __forceinline__ uint32_t pack2(float a, float b)
{
    f32x2 f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ float lo_f(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_f(uint32_t u) { return __uint_as_float(u & 0xFFFF0000u); }

__device__ __forceinline__ bf16x8 frag_of(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    const uint4 u = {a, b, c, d};
    return __builtin_bit_cast(bf16x8, u);
}

// Cross-lane sums without the LDS pipeline (ds_bpermute): v_permlane16_swap / v_permlane32_swap (gfx950) exchange the odd rows
// / upper half of one register with the even rows / lower half of another, so `a = b = v; swap(a, b); a + b` is the xor-16 /
// xor-32 all-reduce step; within a 16-lane row DPP quad permutes and mirrors do the same as modifiers of the add.
#ifndef PMX_CRITIC_NO_DPP
__device__ __forceinline__ float xor16_sum(float v)
{
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float xor32_sum(float v)
{
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
template <int CTRL> __device__ __forceinline__ float dpp_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the four lane groups that hold one token's 32 features
__device__ __forceinline__ float token_sum(float v) { return xor32_sum(xor16_sum(v)); }
// sum over the 16 tokens of a tile (lanes with the same g)
__device__ __forceinline__ float tile_sum(float v)
{
    v += dpp_f<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);       // row_half_mirror
    v += dpp_f<0x140>(v);       // row_mirror
    return v;
}
#else
__device__ __forceinline__ float token_sum(float v)
{
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float tile_sum(float v)
{
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}
#endif

struct Weights {
    bf16x8 A1[8];        // W1 forward
    bf16x8 A2[2][4];     // W2 forward
};

__device__ __forceinline__ void load_fwd_weights(Weights &w, const char *pack, int lane)
{
    const bf16x8 *a1 = reinterpret_cast<const bf16x8 *>(pack + P_A1), *a2 = reinterpret_cast<const bf16x8 *>(pack + P_A2);
#pragma unroll
    for (int m = 0; m < 8; ++m) w.A1[m] = a1[m * 64 + lane];
#pragma unroll
    for (int mo = 0; mo < 2; ++mo)
#pragma unroll
        for (int s = 0; s < 4; ++s) w.A2[mo][s] = a2[(mo * 4 + s) * 64 + lane];
}

// The forward chain for one tile: xb = the lane's 8 input features (bf16 pairs).  Leaves relu(H) in hr (bf16 pairs, operand
// order), z = x + F + b2 in z[8] (true features 8g + j).
__device__ __forceinline__ void ffn_tile(const Weights &w, const uint4 xb, const float (&b1)[8][4], const float (&b2)[8], uint32_t (&hr)[8][2],
                                         float (&z)[8])
{
    const bf16x8 B0 = __builtin_bit_cast(bf16x8, xb);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x4 c = {b1[m][0], b1[m][1], b1[m][2], b1[m][3]};
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.A1[m], B0, c, 0, 0, 0);
        hr[m][0] = pack2(fmaxf(c[0], 0.f), fmaxf(c[1], 0.f));
        hr[m][1] = pack2(fmaxf(c[2], 0.f), fmaxf(c[3], 0.f));
    }
    f32x4 f0 = {b2[0], b2[1], b2[2], b2[3]}, f1 = {b2[4], b2[5], b2[6], b2[7]};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bf16x8 B1 = frag_of(hr[2 * s][0], hr[2 * s][1], hr[2 * s + 1][0], hr[2 * s + 1][1]);
        f0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.A2[0][s], B1, f0, 0, 0, 0);
        f1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.A2[1][s], B1, f1, 0, 0, 0);
    }
    const float x[8] = {lo_f(xb.x), hi_f(xb.x), lo_f(xb.y), hi_f(xb.y), lo_f(xb.z), hi_f(xb.z), lo_f(xb.w), hi_f(xb.w)};
#pragma unroll
    for (int r = 0; r < 4; ++r) z[r] = x[r] + f0[r], z[4 + r] = x[4 + r] + f1[r];
}

__device__ __forceinline__ void load_small_params(const char *pack, int g, float (&b1)[8][4], float (&b2)[8], float (&gamma)[8], float (&beta)[8])
{
    const float *pb = reinterpret_cast<const float *>(pack + P_B1);
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) b1[m][r] = pb[16 * m + 4 * g + r];
#pragma unroll
    for (int j = 0; j < 8; ++j) b2[j] = pb[128 + 8 * g + j], gamma[j] = pb[160 + 8 * g + j], beta[j] = pb[192 + 8 * g + j];
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void pmx_ffn_fwd_kernel(const uint4 *__restrict__ x, const char *__restrict__ pack, uint4 *__restrict__ y,
                                                            long T, float eps)
{
    const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4;
    const long n_tiles = (T + 15) >> 4;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long)gridDim.x * 4;
    Weights w;
    load_fwd_weights(w, pack, lane);
    float b1[8][4], b2[8], gamma[8], beta[8];
    load_small_params(pack, g, b1, b2, gamma, beta);
    // two tiles in flight ahead of the one being computed
    auto fetch = [&](long tile) -> uint4 {
        const long tok = tile * 16 + p;
        return (tile < n_tiles && tok < T) ? x[tok * 4 + g] : uint4{0, 0, 0, 0};
    };
    uint4 x0 = fetch(wave), x1 = fetch(wave + n_waves);
    for (long tile = wave; tile < n_tiles; tile += n_waves) {
        const uint4 x2 = fetch(tile + 2 * n_waves);
        uint32_t hr[8][2];
        float z[8];
        ffn_tile(w, x0, b1, b2, hr, z);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s1 += z[j];
        const float mean = token_sum(s1) * (1.0f / 32.0f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            z[j] -= mean;
            s2 = fmaf(z[j], z[j], s2);
        }
        const float rstd = __builtin_amdgcn_rsqf(token_sum(s2) * (1.0f / 32.0f) + eps);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaf(z[j] * rstd, gamma[j], beta[j]);
        const long tok = tile * 16 + p;
        if (tok < T) y[tok * 4 + g] = uint4{pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7])};
        x0 = x1, x1 = x2;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward.  One wave per SIMD (the weight-gradient accumulators and four operand sets leave no room for two).
// LDS per wave: a 32-token staging area [token][x 32 | dF 32 | H 128 | dH 128] bf16 = 640 bytes per token.
// ---------------------------------------------------------------------------------------------------------------
constexpr int STG_ROW = 320 * 2 + 16;          // bytes per staged token (+16: rows 4 banks apart instead of aligned)
constexpr int STG_X = 0, STG_DF = 64, STG_H = 128, STG_DH = 384;

__device__ __forceinline__ bf16x8 tr_frag(const char *stg, int col_byte, int lane)
{
    // operand fragment for a contraction over the 32 staged tokens: lane group g covers tokens 8g .. 8g+7, columns
    // col_byte/2 .. +15; lane 4*row + pc of the group addresses token row, columns 4*pc .. 4*pc+3
    const int g = lane >> 4, row = (lane & 15) >> 2, pc = lane & 3;
    const char *a0 = stg + (8 * g + row) * STG_ROW + col_byte + pc * 8;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3))) *)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3))) *)(a0 + 4 * STG_ROW));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ __launch_bounds__(256, 1) void pmx_ffn_bwd_kernel(const uint4 *__restrict__ x, const uint4 *__restrict__ dy, const char *__restrict__ pack,
                                                            uint4 *__restrict__ dx, float *__restrict__ grad, long T, float eps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    char *stg = smem + (size_t)wv * 32 * STG_ROW;
    const long n_pairs = (T + 31) >> 5;
    const long wave = (long)blockIdx.x * 4 + wv, n_waves = (long)gridDim.x * 4;
    Weights w;
    load_fwd_weights(w, pack, lane);
    bf16x8 A3[8], A4[2][4];
    {
        const bf16x8 *a3 = reinterpret_cast<const bf16x8 *>(pack + P_A3), *a4 = reinterpret_cast<const bf16x8 *>(pack + P_A4);
#pragma unroll
        for (int m = 0; m < 8; ++m) A3[m] = a3[m * 64 + lane];
#pragma unroll
        for (int mo = 0; mo < 2; ++mo)
#pragma unroll
            for (int s = 0; s < 4; ++s) A4[mo][s] = a4[(mo * 4 + s) * 64 + lane];
    }
    float b1[8][4], b2[8], gamma[8], beta[8];
    load_small_params(pack, g, b1, b2, gamma, beta);
    f32x4 aw2[2][8], aw1[8][2];                  // dW2 tiles [out half][hidden tile], dW1 tiles [hidden tile][in half]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) aw2[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}, aw1[b][a] = f32x4{0.f, 0.f, 0.f, 0.f};
    float db1[8][4], db2[8], dgam[8], dbet[8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) db1[m][r] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) db2[j] = dgam[j] = dbet[j] = 0.f;

    auto fetch = [&](const uint4 *src, long pair, int u) -> uint4 {
        const long tok = pair * 32 + 16 * u + p;
        return (pair < n_pairs && tok < T) ? src[tok * 4 + g] : uint4{0, 0, 0, 0};
    };
    uint4 xn[2] = {fetch(x, wave, 0), fetch(x, wave, 1)}, dn[2] = {fetch(dy, wave, 0), fetch(dy, wave, 1)};
    for (long pair = wave; pair < n_pairs; pair += n_waves) {
        const uint4 xc[2] = {xn[0], xn[1]}, dc[2] = {dn[0], dn[1]};
        xn[0] = fetch(x, pair + n_waves, 0), xn[1] = fetch(x, pair + n_waves, 1);
        dn[0] = fetch(dy, pair + n_waves, 0), dn[1] = fetch(dy, pair + n_waves, 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            uint32_t hr[8][2];
            float z[8];
            ffn_tile(w, xc[u], b1, b2, hr, z);
            // LayerNorm forward statistics, then its backward
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s1 += z[j];
            const float mean = token_sum(s1) * (1.0f / 32.0f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                z[j] -= mean;
                s2 = fmaf(z[j], z[j], s2);
            }
            const float rstd = __builtin_amdgcn_rsqf(token_sum(s2) * (1.0f / 32.0f) + eps);
            const float d[8] = {lo_f(dc[u].x), hi_f(dc[u].x), lo_f(dc[u].y), hi_f(dc[u].y), lo_f(dc[u].z), hi_f(dc[u].z), lo_f(dc[u].w), hi_f(dc[u].w)};
            float gd[8], t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = z[j] * rstd;
                z[j] = xh;
                dbet[j] += d[j];
                dgam[j] = fmaf(d[j], xh, dgam[j]);
                gd[j] = gamma[j] * d[j];
                t1 += gd[j];
                t2 = fmaf(gd[j], xh, t2);
            }
            t1 = token_sum(t1) * (1.0f / 32.0f), t2 = token_sum(t2) * (1.0f / 32.0f);
            float dz[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) dz[j] = rstd * (gd[j] - t1 - z[j] * t2);
            const uint4 dzb = {pack2(dz[0], dz[1]), pack2(dz[2], dz[3]), pack2(dz[4], dz[5]), pack2(dz[6], dz[7])};
            {   // the bias gradient of the second product sums what the matrix cores see
                const float q[8] = {lo_f(dzb.x), hi_f(dzb.x), lo_f(dzb.y), hi_f(dzb.y), lo_f(dzb.z), hi_f(dzb.z), lo_f(dzb.w), hi_f(dzb.w)};
#pragma unroll
                for (int j = 0; j < 8; ++j) db2[j] += q[j];
            }
            // dH = W2^T dF, masked by relu'(H); dX = W1^T dH + dz
            const bf16x8 Bd = __builtin_bit_cast(bf16x8, dzb);
            uint32_t dh[8][2];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A3[m], Bd, c, 0, 0, 0);
                const float h0 = lo_f(hr[m][0]), h1 = hi_f(hr[m][0]), h2 = lo_f(hr[m][1]), h3 = hi_f(hr[m][1]);
                const float c0 = h0 > 0.f ? c[0] : 0.f, c1 = h1 > 0.f ? c[1] : 0.f, c2 = h2 > 0.f ? c[2] : 0.f, c3 = h3 > 0.f ? c[3] : 0.f;
                dh[m][0] = pack2(c0, c1), dh[m][1] = pack2(c2, c3);
                db1[m][0] += lo_f(dh[m][0]), db1[m][1] += hi_f(dh[m][0]), db1[m][2] += lo_f(dh[m][1]), db1[m][3] += hi_f(dh[m][1]);
            }
            f32x4 e0 = {dz[0], dz[1], dz[2], dz[3]}, e1 = {dz[4], dz[5], dz[6], dz[7]};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 Bh = frag_of(dh[2 * s][0], dh[2 * s][1], dh[2 * s + 1][0], dh[2 * s + 1][1]);
                e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A4[0][s], Bh, e0, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A4[1][s], Bh, e1, 0, 0, 0);
            }
            const long tok = pair * 32 + 16 * u + p;
            if (tok < T) dx[tok * 4 + g] = uint4{pack2(e0[0], e0[1]), pack2(e0[2], e0[3]), pack2(e1[0], e1[1]), pack2(e1[2], e1[3])};
            // stage this tile's operands of the weight gradients: [token][x | dF | H | dH]
            char *row = stg + (16 * u + p) * STG_ROW;
            *reinterpret_cast<uint4 *>(row + STG_X + 16 * g) = xc[u];
            *reinterpret_cast<uint4 *>(row + STG_DF + 16 * g) = dzb;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                *reinterpret_cast<uint2 *>(row + STG_H + 32 * m + 8 * g) = uint2{hr[m][0], hr[m][1]};
                *reinterpret_cast<uint2 *>(row + STG_DH + 32 * m + 8 * g) = uint2{dh[m][0], dh[m][1]};
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // weight gradients over the pair's 32 tokens
        bf16x8 Fx[2], Fd[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) Fx[a] = tr_frag(stg, STG_X + 32 * a, lane), Fd[a] = tr_frag(stg, STG_DF + 32 * a, lane);
#pragma unroll
        for (int mh = 0; mh < 8; ++mh) {
            const bf16x8 Fh = tr_frag(stg, STG_H + 32 * mh, lane), Fdh = tr_frag(stg, STG_DH + 32 * mh, lane);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                aw2[a][mh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Fd[a], Fh, aw2[a][mh], 0, 0, 0);     // dW2[out][hidden] += dF^T H
                aw1[mh][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Fdh, Fx[a], aw1[mh][a], 0, 0, 0);    // dW1[hidden][in] += dH^T x
            }
        }
    }
    // ---- reductions: block-level sums through LDS; every block then writes ITS partial gradient row (plain stores) and a
    // second tiny kernel adds the rows up.  Float atomics from every block onto the same 8 416 addresses ran at the rate of
    // one contended row (MI355X_MICROARCH.md, global float atomics): ~90 us per call whatever the batch, which made the
    // 512-sample optimizer step slower than the unfused path. ----
    __syncthreads();
    float *mine = grad + (size_t)(1 + blockIdx.x) * G_FLOATS;
    float *red = reinterpret_cast<float *>(smem);           // 4 waves x 8 tiles x 1 KB = 32 KB per round (the staging areas are 80 KB)
#pragma unroll
    for (int round = 0; round < 4; ++round) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 v = round < 2 ? aw2[round][i] : aw1[i][round - 2];
            *reinterpret_cast<f32x4 *>(red + ((size_t)(wv * 8 + i) * 64 + lane) * 4) = v;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 8 * 64; i += 256) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(red + (size_t)i * 4);
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const f32x4 u = *reinterpret_cast<const f32x4 *>(red + ((size_t)q * 8 * 64 + i) * 4);
                v[0] += u[0], v[1] += u[1], v[2] += u[2], v[3] += u[3];
            }
            // tile i of this round, lane l, register r: D[row = 4 (l >> 4) + r][col = l & 15]
            const int t = i >> 6, l = i & 63, rg = l >> 4, col = l & 15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rowi = 4 * rg + r;
                float *dst = round < 2 ? mine + G_W2 + (16 * round + rowi) * 128 + 16 * t + col          // dW2[out][hidden]
                                       : mine + G_W1 + (16 * t + rowi) * 32 + 16 * (round - 2) + col;    // dW1[hidden][in]
                *dst = v[r];
            }
        }
        __syncthreads();
    }
    // per-channel sums: over the tile's tokens by shuffles, over the block's four waves through LDS
    float *chan = red + (size_t)wv * 224;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = tile_sum(db1[m][r]);
            if (p == 0) chan[16 * m + 4 * g + r] = v;
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float a = tile_sum(db2[j]), b = tile_sum(dgam[j]), c = tile_sum(dbet[j]);
        if (p == 0) chan[128 + 8 * g + j] = a, chan[160 + 8 * g + j] = b, chan[192 + 8 * g + j] = c;
    }
    __syncthreads();
    if (threadIdx.x < 224) mine[G_B1 + threadIdx.x] = red[threadIdx.x] + red[224 + threadIdx.x] + red[448 + threadIdx.x] + red[672 + threadIdx.x];
}

// out[i] = sum over the partial rows (rows 1 .. n of the same buffer): the second stage of the gradient reductions.  A block
// sums 32 columns with 8 slices of the rows side by side (128-byte segments per row, 8 loads in flight per thread) and adds the
// slices through LDS: one thread per column walking all the rows alone was a serial chain of n / 8 memory round trips
// (14.7 us per call at 512 rows, six calls per optimizer step).
__global__ __launch_bounds__(256) void pmx_sum_rows_kernel(float *__restrict__ buf, int n_rows, int floats)
{
    __shared__ float part[8][33];
    const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + c;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < floats) {
        int r = 1 + sl;
        for (; r + 24 <= n_rows; r += 32) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += buf[(size_t)(r + 8 * k) * floats + i];
        }
        for (; r <= n_rows; r += 8) acc[0] += buf[(size_t)r * floats + i];
    }
    part[sl][c] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (sl == 0 && i < floats) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][c];
        buf[i] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// float32 parameters -> operand fragments.  hidden order phi, output / input order psi (file header).
// ---------------------------------------------------------------------------------------------------------------
// (bx, nbx: the block's index and the number of blocks that share the job -- the stand-alone kernel passes blockIdx.x / gridDim.x, the
// one-launch encoder pack its own split)
__device__ __forceinline__ void ffn_pack_body(const float *__restrict__ w1, const float *__restrict__ b1, const float *__restrict__ w2,
                                              const float *__restrict__ b2, const float *__restrict__ gamma, const float *__restrict__ beta,
                                              char *__restrict__ pack, int bx, int nbx)
{
    // w1 [128][32] (linear1.weight), w2 [32][128] (linear2.weight)
    short *a1 = reinterpret_cast<short *>(pack + P_A1), *a2 = reinterpret_cast<short *>(pack + P_A2);
    short *a3 = reinterpret_cast<short *>(pack + P_A3), *a4 = reinterpret_cast<short *>(pack + P_A4);
    for (int i = bx * 256 + threadIdx.x; i < 8 * FRAG; i += nbx * 256) {
        const int j = i & 7, lane = (i >> 3) & 63, f = i >> 9;          // fragment f, lane, element j
        const int row = lane & 15, g = lane >> 4;
        const int psi_row = 8 * (row >> 2) + (row & 3);                 // + 4 * half
        auto phi = [&](int s) { return 32 * s + 16 * (j >> 2) + 4 * g + (j & 3); };
        // A1[m = f]: W1[hidden 16 m + row][in 8 g + j]
        a1[i] = (short)(pack2(w1[(16 * f + row) * 32 + 8 * g + j], 0.f) & 0xFFFF);
        // A2[mo][s], f = mo * 4 + s: W2[out psi(mo, row)][hidden phi(s, g, j)]
        a2[i] = (short)(pack2(w2[(psi_row + 4 * (f >> 2)) * 128 + phi(f & 3)], 0.f) & 0xFFFF);
        // A3[m = f]: W2[out 8 g + j][hidden 16 m + row]      (dH = W2^T dF)
        a3[i] = (short)(pack2(w2[(8 * g + j) * 128 + 16 * f + row], 0.f) & 0xFFFF);
        // A4[mo][s]: W1[hidden phi(s, g, j)][in psi(mo, row)]  (dX = W1^T dH)
        a4[i] = (short)(pack2(w1[phi(f & 3) * 32 + psi_row + 4 * (f >> 2)], 0.f) & 0xFFFF);
    }
    if (bx == 0) {
        float *pb = reinterpret_cast<float *>(pack + P_B1);
        for (int i = threadIdx.x; i < 128; i += 256) pb[i] = b1[i];
        if (threadIdx.x < 32) pb[128 + threadIdx.x] = b2[threadIdx.x], pb[160 + threadIdx.x] = gamma[threadIdx.x], pb[192 + threadIdx.x] = beta[threadIdx.x];
    }
}

__global__ __launch_bounds__(256) void pmx_ffn_pack_kernel(const float *__restrict__ w1, const float *__restrict__ b1, const float *__restrict__ w2,
                                                          const float *__restrict__ b2, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          char *__restrict__ pack)
{
    ffn_pack_body(w1, b1, w2, b2, gamma, beta, pack, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------
// Token-parallel linear layers with 32 input features, same scheme as above (features on the rows, tokens on the columns,
// output rows in the order psi within every 32-feature block so that a lane owns 8 consecutive features):
//     NP = 3, LN = false :  qkv = W a + b                      the attention in-projection (32 -> 96)
//     NP = 1, LN = true  :  y = LayerNorm(x + W a + b)         the attention out-projection + residual + norm1
// Backward (recomputing t = W a + b where LayerNorm needs it): da = W^T dt, dx = dz, dW += dt (x) a over tokens through the
// same 32-token LDS staging + transposing reads, bias / LayerNorm-affine gradients by per-lane sums.
// pack: A forward [2 NP] fragments, A backward [2][NP] fragments, then floats bias [32 NP], gamma [32], beta [32]
// grad: dW [32 NP][32], db [32 NP], dgamma [32], dbeta [32]
// ---------------------------------------------------------------------------------------------------------------
template <int NP> struct TokPack {
    static constexpr size_t A_FWD = 0, A_BWD = A_FWD + 2 * NP * FRAG * 2, FLT = A_BWD + 2 * NP * FRAG * 2;
    static constexpr size_t BYTES = FLT + (32 * NP + 64) * 4;
    static constexpr int G_W = 0, G_B = 32 * NP * 32, G_GAMMA = G_B + 32 * NP, G_BETA = G_GAMMA + 32, G_FLOATS = G_BETA + 32;
};
static_assert(TokPack<3>::BYTES == PMX_TOK96_PACK_BYTES && TokPack<1>::BYTES == PMX_TOK32_PACK_BYTES, "pmx.h pack sizes");
static_assert(TokPack<3>::G_FLOATS == PMX_TOK96_GRAD_FLOATS && TokPack<1>::G_FLOATS == PMX_TOK32_GRAD_FLOATS, "pmx.h gradient sizes");

template <int NP, bool LN>
__global__ __launch_bounds__(256, 2) void pmx_tok_fwd_kernel(const uint4 *__restrict__ a, const uint4 *__restrict__ x, const char *__restrict__ pack,
                                                            uint4 *__restrict__ y, long T, float eps)
{
    const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4;
    const long n_tiles = (T + 15) >> 4;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long)gridDim.x * 4;
    bf16x8 A[2 * NP];
    const bf16x8 *af = reinterpret_cast<const bf16x8 *>(pack + TokPack<NP>::A_FWD);
#pragma unroll
    for (int m = 0; m < 2 * NP; ++m) A[m] = af[m * 64 + lane];
    const float *pf = reinterpret_cast<const float *>(pack + TokPack<NP>::FLT);
    float bias[NP][8], gamma[8], beta[8];
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) bias[q][j] = pf[32 * q + 8 * g + j];
#pragma unroll
    for (int j = 0; j < 8; ++j) gamma[j] = pf[32 * NP + 8 * g + j], beta[j] = pf[32 * NP + 32 + 8 * g + j];
    auto fetch = [&](const uint4 *src, long tile) -> uint4 {
        const long tok = tile * 16 + p;
        return (tile < n_tiles && tok < T) ? src[tok * 4 + g] : uint4{0, 0, 0, 0};
    };
    uint4 a0 = fetch(a, wave), a1 = fetch(a, wave + n_waves);
    uint4 x0 = LN ? fetch(x, wave) : uint4{0, 0, 0, 0}, x1 = LN ? fetch(x, wave + n_waves) : uint4{0, 0, 0, 0};
    for (long tile = wave; tile < n_tiles; tile += n_waves) {
        const uint4 a2 = fetch(a, tile + 2 * n_waves);
        const uint4 x2 = LN ? fetch(x, tile + 2 * n_waves) : uint4{0, 0, 0, 0};
        const bf16x8 B0 = __builtin_bit_cast(bf16x8, a0);
        const long tok = tile * 16 + p;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            f32x4 c0 = {bias[q][0], bias[q][1], bias[q][2], bias[q][3]}, c1 = {bias[q][4], bias[q][5], bias[q][6], bias[q][7]};
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2 * q], B0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2 * q + 1], B0, c1, 0, 0, 0);
            float o[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
            if (LN) {
                const float xv[8] = {lo_f(x0.x), hi_f(x0.x), lo_f(x0.y), hi_f(x0.y), lo_f(x0.z), hi_f(x0.z), lo_f(x0.w), hi_f(x0.w)};
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += xv[j], s1 += o[j];
                const float mean = token_sum(s1) * (1.0f / 32.0f);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] -= mean, s2 = fmaf(o[j], o[j], s2);
                const float rstd = __builtin_amdgcn_rsqf(token_sum(s2) * (1.0f / 32.0f) + eps);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = fmaf(o[j] * rstd, gamma[j], beta[j]);
            }
            if (tok < T) y[tok * (4 * NP) + 4 * q + g] = uint4{pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7])};
        }
        a0 = a1, a1 = a2, x0 = x1, x1 = x2;
    }
}

template <int NP> constexpr int tok_stg_row() { return (32 + 32 * NP) * 2 + 16; }

template <int NP>
__device__ __forceinline__ bf16x8 tok_tr_frag(const char *stg, int col_byte, int lane)
{
    const int g = lane >> 4, row = (lane & 15) >> 2, pc = lane & 3;
    const char *a0 = stg + (8 * g + row) * tok_stg_row<NP>() + col_byte + pc * 8;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3))) *)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3))) *)(a0 + 4 * tok_stg_row<NP>()));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int NP, bool LN>
__global__ __launch_bounds__(256, 2) void pmx_tok_bwd_kernel(const uint4 *__restrict__ a, const uint4 *__restrict__ x, const uint4 *__restrict__ dy,
                                                            const char *__restrict__ pack, uint4 *__restrict__ da, uint4 *__restrict__ dx,
                                                            float *__restrict__ grad, long T, float eps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROW = tok_stg_row<NP>();
    const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    char *stg = smem + (size_t)wv * 32 * ROW;
    const long n_pairs = (T + 31) >> 5;
    const long wave = (long)blockIdx.x * 4 + wv, n_waves = (long)gridDim.x * 4;
    bf16x8 A[2 * NP], At[2][NP];
    {
        const bf16x8 *af = reinterpret_cast<const bf16x8 *>(pack + TokPack<NP>::A_FWD), *ab = reinterpret_cast<const bf16x8 *>(pack + TokPack<NP>::A_BWD);
#pragma unroll
        for (int m = 0; m < 2 * NP; ++m) A[m] = af[m * 64 + lane];
#pragma unroll
        for (int mo = 0; mo < 2; ++mo)
#pragma unroll
            for (int q = 0; q < NP; ++q) At[mo][q] = ab[(mo * NP + q) * 64 + lane];
    }
    const float *pf = reinterpret_cast<const float *>(pack + TokPack<NP>::FLT);
    float bias[NP][8], gamma[8];
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) bias[q][j] = pf[32 * q + 8 * g + j];
#pragma unroll
    for (int j = 0; j < 8; ++j) gamma[j] = pf[32 * NP + 8 * g + j];
    f32x4 aw[2 * NP][2];
#pragma unroll
    for (int m = 0; m < 2 * NP; ++m) aw[m][0] = aw[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    float db[NP][8], dgam[8], dbet[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        dgam[j] = dbet[j] = 0.f;
#pragma unroll
        for (int q = 0; q < NP; ++q) db[q][j] = 0.f;
    }
    auto fetch = [&](const uint4 *src, long pair, int u, int chunks, int chunk) -> uint4 {
        const long tok = pair * 32 + 16 * u + p;
        return (pair < n_pairs && tok < T) ? src[tok * chunks + chunk] : uint4{0, 0, 0, 0};
    };
    for (long pair = wave; pair < n_pairs; pair += n_waves) {
        uint4 av[2], xv4[2], dv[2][NP];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            av[u] = fetch(a, pair, u, 4, g);
            xv4[u] = (LN || x) ? fetch(x, pair, u, 4, g) : uint4{0, 0, 0, 0};    // !LN: x = a gradient to ADD to da (or null)
#pragma unroll
            for (int q = 0; q < NP; ++q) dv[u][q] = fetch(dy, pair, u, 4 * NP, 4 * q + g);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            uint4 dt[NP];
            if (LN) {
                // recompute z = x + W a + b and its statistics, then LayerNorm's backward
                const bf16x8 B0 = __builtin_bit_cast(bf16x8, av[u]);
                f32x4 c0 = {bias[0][0], bias[0][1], bias[0][2], bias[0][3]}, c1 = {bias[0][4], bias[0][5], bias[0][6], bias[0][7]};
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], B0, c1, 0, 0, 0);
                const uint4 xb = xv4[u];
                const float xf[8] = {lo_f(xb.x), hi_f(xb.x), lo_f(xb.y), hi_f(xb.y), lo_f(xb.z), hi_f(xb.z), lo_f(xb.w), hi_f(xb.w)};
                float z[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] += xf[j], s1 += z[j];
                const float mean = token_sum(s1) * (1.0f / 32.0f);
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] -= mean, s2 = fmaf(z[j], z[j], s2);
                const float rstd = __builtin_amdgcn_rsqf(token_sum(s2) * (1.0f / 32.0f) + eps);
                const uint4 db4 = dv[u][0];
                const float d[8] = {lo_f(db4.x), hi_f(db4.x), lo_f(db4.y), hi_f(db4.y), lo_f(db4.z), hi_f(db4.z), lo_f(db4.w), hi_f(db4.w)};
                float gd[8], t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = z[j] * rstd;
                    z[j] = xh;
                    dbet[j] += d[j];
                    dgam[j] = fmaf(d[j], xh, dgam[j]);
                    gd[j] = gamma[j] * d[j];
                    t1 += gd[j];
                    t2 = fmaf(gd[j], xh, t2);
                }
                t1 = token_sum(t1) * (1.0f / 32.0f), t2 = token_sum(t2) * (1.0f / 32.0f);
                float dz[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) dz[j] = rstd * (gd[j] - t1 - z[j] * t2);
                dt[0] = uint4{pack2(dz[0], dz[1]), pack2(dz[2], dz[3]), pack2(dz[4], dz[5]), pack2(dz[6], dz[7])};
                const long tok = pair * 32 + 16 * u + p;
                if (tok < T) dx[tok * 4 + g] = dt[0];
            } else {
#pragma unroll
                for (int q = 0; q < NP; ++q) dt[q] = dv[u][q];
            }
            // da = W^T dt
            f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const bf16x8 Bd = __builtin_bit_cast(bf16x8, dt[q]);
                e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(At[0][q], Bd, e0, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(At[1][q], Bd, e1, 0, 0, 0);
                const float v[8] = {lo_f(dt[q].x), hi_f(dt[q].x), lo_f(dt[q].y), hi_f(dt[q].y), lo_f(dt[q].z), hi_f(dt[q].z), lo_f(dt[q].w), hi_f(dt[q].w)};
#pragma unroll
                for (int j = 0; j < 8; ++j) db[q][j] += v[j];
            }
            if (!LN && x) {
                // the residual branch's gradient of the same tokens (the encoder layer's input feeds the in-projection AND the residual
                // add in front of norm1): summed here in float32 and rounded once, instead of by an add kernel of autograd's
                const uint4 r = xv4[u];
                e0[0] += lo_f(r.x), e0[1] += hi_f(r.x), e0[2] += lo_f(r.y), e0[3] += hi_f(r.y);
                e1[0] += lo_f(r.z), e1[1] += hi_f(r.z), e1[2] += lo_f(r.w), e1[3] += hi_f(r.w);
            }
            const long tok = pair * 32 + 16 * u + p;
            if (tok < T) da[tok * 4 + g] = uint4{pack2(e0[0], e0[1]), pack2(e0[2], e0[3]), pack2(e1[0], e1[1]), pack2(e1[2], e1[3])};
            char *row = stg + (16 * u + p) * ROW;
            *reinterpret_cast<uint4 *>(row + 16 * g) = av[u];
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<uint4 *>(row + 64 + 64 * q + 16 * g) = dt[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bf16x8 Fa0 = tok_tr_frag<NP>(stg, 0, lane), Fa1 = tok_tr_frag<NP>(stg, 32, lane);
#pragma unroll
        for (int m = 0; m < 2 * NP; ++m) {
            const bf16x8 Fd = tok_tr_frag<NP>(stg, 64 + 32 * m, lane);
            aw[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Fd, Fa0, aw[m][0], 0, 0, 0);      // dW[out 16m ..][in 0..15]
            aw[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Fd, Fa1, aw[m][1], 0, 0, 0);      // dW[out 16m ..][in 16..31]
        }
    }
    __syncthreads();
    float *mine = grad + (size_t)(1 + blockIdx.x) * TokPack<NP>::G_FLOATS;        // this block's partial row (see pmx_ffn_bwd_kernel)
    float *red = reinterpret_cast<float *>(smem);           // 4 waves x 4 NP tiles x 1 KB
#pragma unroll
    for (int i = 0; i < 4 * NP; ++i) *reinterpret_cast<f32x4 *>(red + ((size_t)(wv * 4 * NP + i) * 64 + lane) * 4) = aw[i >> 1][i & 1];
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * NP * 64; i += 256) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(red + (size_t)i * 4);
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const f32x4 u = *reinterpret_cast<const f32x4 *>(red + ((size_t)q * 4 * NP * 64 + i) * 4);
            v[0] += u[0], v[1] += u[1], v[2] += u[2], v[3] += u[3];
        }
        const int t = i >> 6, l = i & 63, m = t >> 1, half = t & 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[TokPack<NP>::G_W + (16 * m + 4 * (l >> 4) + r) * 32 + 16 * half + (l & 15)] = v[r];
    }
    __syncthreads();
    constexpr int NCH = 32 * NP + 64;
    float *chan = red + (size_t)wv * NCH;
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = tile_sum(db[q][j]);
            if (p == 0) chan[32 * q + 8 * g + j] = v;
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float b = tile_sum(dgam[j]), c = tile_sum(dbet[j]);
        if (p == 0) chan[32 * NP + 8 * g + j] = b, chan[32 * NP + 32 + 8 * g + j] = c;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NCH; i += 256) mine[TokPack<NP>::G_B + i] = red[i] + red[NCH + i] + red[2 * NCH + i] + red[3 * NCH + i];
}

template <int NP>
__device__ __forceinline__ void tok_pack_body(const float *__restrict__ w, const float *__restrict__ b, const float *__restrict__ gamma,
                                              const float *__restrict__ beta, char *__restrict__ pack, int bx, int nbx)
{
    // w [32 NP][32] (nn.Linear weight)
    short *af = reinterpret_cast<short *>(pack + TokPack<NP>::A_FWD), *ab = reinterpret_cast<short *>(pack + TokPack<NP>::A_BWD);
    for (int i = bx * 256 + threadIdx.x; i < 2 * NP * FRAG; i += nbx * 256) {
        const int j = i & 7, lane = (i >> 3) & 63, f = i >> 9;
        const int row = lane & 15, g = lane >> 4;
        const int psi_row = 8 * (row >> 2) + (row & 3);
        // forward fragment f = 2 q + m': W[out 32 q + psi(m', row)][in 8 g + j]
        af[i] = (short)(pack2(w[(32 * (f >> 1) + psi_row + 4 * (f & 1)) * 32 + 8 * g + j], 0.f) & 0xFFFF);
        // backward fragment f = mo * NP + q: W[out 32 q + 8 g + j][in psi(mo, row)]
        ab[i] = (short)(pack2(w[(32 * (f % NP) + 8 * g + j) * 32 + psi_row + 4 * (f / NP)], 0.f) & 0xFFFF);
    }
    if (bx == 0) {
        float *pf = reinterpret_cast<float *>(pack + TokPack<NP>::FLT);
        for (int i = threadIdx.x; i < 32 * NP; i += 256) pf[i] = b[i];
        if (threadIdx.x < 32) {
            pf[32 * NP + threadIdx.x] = gamma ? gamma[threadIdx.x] : 1.f;
            pf[32 * NP + 32 + threadIdx.x] = beta ? beta[threadIdx.x] : 0.f;
        }
    }
}

template <int NP>
__global__ __launch_bounds__(256) void pmx_tok_pack_kernel(const float *__restrict__ w, const float *__restrict__ b, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, char *__restrict__ pack)
{
    tok_pack_body<NP>(w, b, gamma, beta, pack, (int)blockIdx.x, (int)gridDim.x);
}

// every pack of up to four encoder layers in ONE launch: blockIdx.y = 3 * layer + (0 in-projection | 1 out-projection + norm1 |
// 2 feed-forward + norm2), 8 blocks per job (six small pack launches sat in front of the kernels of the launch-bound 512-sample step)
struct EncoderPackArgs {
    pmx_encoder_layer_params l[4];
};
__global__ __launch_bounds__(256) void pmx_encoder_pack_kernel(EncoderPackArgs a)
{
    const pmx_encoder_layer_params &L = a.l[blockIdx.y / 3];
    const int which = blockIdx.y % 3, bx = (int)blockIdx.x, nbx = (int)gridDim.x;
    if (which == 0) tok_pack_body<3>(L.in_proj_w, L.in_proj_b, nullptr, nullptr, reinterpret_cast<char *>(L.pack_in), bx, nbx);
    else if (which == 1) tok_pack_body<1>(L.out_proj_w, L.out_proj_b, L.norm1_w, L.norm1_b, reinterpret_cast<char *>(L.pack_out), bx, nbx);
    else ffn_pack_body(L.lin1_w, L.lin1_b, L.lin2_w, L.lin2_b, L.norm2_w, L.norm2_b, reinterpret_cast<char *>(L.pack_ffn), bx, nbx);
}

int cu_count()
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    return cus;
}

}   // namespace

extern "C" int pmx_encoder_pack(int32_t n_layers, const pmx_encoder_layer_params *layers, void *stream)
{
    if (n_layers < 1 || n_layers > 4 || !layers) return PMX_ERR_INVALID;
    EncoderPackArgs a;
    for (int i = 0; i < n_layers; ++i) {
        a.l[i] = layers[i];
        const pmx_encoder_layer_params &L = a.l[i];
        if (!L.in_proj_w || !L.in_proj_b || !L.out_proj_w || !L.out_proj_b || !L.norm1_w || !L.norm1_b || !L.lin1_w || !L.lin1_b || !L.lin2_w ||
            !L.lin2_b || !L.norm2_w || !L.norm2_b || !L.pack_in || !L.pack_out || !L.pack_ffn)
            return PMX_ERR_INVALID;
    }
    hipLaunchKernelGGL(pmx_encoder_pack_kernel, dim3(8, 3 * n_layers), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

extern "C" int pmx_ffn_pack(const float *w1, const float *b1, const float *w2, const float *b2, const float *gamma, const float *beta,
                            void *pack_dev, void *stream)
{
    if (!w1 || !b1 || !w2 || !b2 || !gamma || !beta || !pack_dev) return PMX_ERR_INVALID;
    hipLaunchKernelGGL(pmx_ffn_pack_kernel, dim3(16), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w1, b1, w2, b2, gamma, beta,
                       reinterpret_cast<char *>(pack_dev));
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

extern "C" int pmx_ffn_forward(const void *x_dev, const void *pack_dev, void *y_dev, int64_t tokens, float eps, void *stream)
{
    if (tokens == 0) return PMX_OK;
    if (!x_dev || !pack_dev || !y_dev || tokens < 0) return PMX_ERR_INVALID;
    const int64_t tiles = (tokens + 15) / 16;
    const int64_t want = (tiles + 3) / 4, cap = (int64_t)cu_count() * 2;
    hipLaunchKernelGGL(pmx_ffn_fwd_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       (const uint4 *)x_dev, (const char *)pack_dev, (uint4 *)y_dev, (long)tokens, eps);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

// Deferred row sums: with pmx_defer_row_sums(1) the backward entry points of this file and of pmx_heads.hip leave their partial rows
// unsummed and report how many there are (pmx_last_partial_rows); the caller adds them with pmx_sum_partial_rows on a stream of its
// choice -- a side stream, so that the second stage of one gradient reduction runs beside the next backward kernel instead of in
// front of it (six small launches on the critical path of the launch-bound 512-sample step).  Per host thread.
thread_local int pmx_defer_sums_flag = 0;
thread_local int pmx_last_rows_value = 0;
extern "C" int pmx_defer_row_sums(int32_t on) { pmx_defer_sums_flag = on ? 1 : 0; return PMX_OK; }
extern "C" int pmx_last_partial_rows(void) { return pmx_last_rows_value; }
extern "C" int pmx_sum_partial_rows(float *buf_dev, int32_t n_rows, int32_t floats, void *stream)
{
    if (!buf_dev || n_rows < 0 || floats < 1) return PMX_ERR_INVALID;
    if (n_rows == 0) return PMX_OK;
    hipLaunchKernelGGL(pmx_sum_rows_kernel, dim3((floats + 31) / 32), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), buf_dev, (int)n_rows, (int)floats);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

extern "C" int pmx_ffn_backward(const void *x_dev, const void *dy_dev, const void *pack_dev, void *dx_dev, float *grad_dev, int64_t tokens,
                                float eps, void *stream)
{
    if (!grad_dev) return PMX_ERR_INVALID;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    pmx_last_rows_value = 0;                                  // (row 0 is final on every path that launches no partial rows)
    if (tokens == 0) return hipMemsetAsync(grad_dev, 0, sizeof(float) * G_FLOATS, st) == hipSuccess ? PMX_OK : PMX_ERR_HIP;
    if (!x_dev || !dy_dev || !pack_dev || !dx_dev || tokens < 0) return PMX_ERR_INVALID;
    const size_t lds = (size_t)4 * 32 * STG_ROW;
    static bool attr_dev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PMX_ERR_HIP;
    if (lds > 65536 && !attr_dev[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(pmx_ffn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return PMX_ERR_HIP;
        attr_dev[dev] = true;
    }
    const int64_t pairs = (tokens + 31) / 32;
    int64_t want = (pairs + 3) / 4, cap = (int64_t)cu_count();
    if (cap > PMX_GRAD_PARTIAL_ROWS) cap = PMX_GRAD_PARTIAL_ROWS;
    const unsigned blocks = (unsigned)(want < cap ? want : cap);
    hipLaunchKernelGGL(pmx_ffn_bwd_kernel, dim3(blocks), dim3(256), lds, st, (const uint4 *)x_dev, (const uint4 *)dy_dev,
                       (const char *)pack_dev, (uint4 *)dx_dev, grad_dev, (long)tokens, eps);
    pmx_last_rows_value = (int)blocks;
    if (!pmx_defer_sums_flag)
        hipLaunchKernelGGL(pmx_sum_rows_kernel, dim3((G_FLOATS + 31) / 32), dim3(256), 0, st, grad_dev, (int)blocks, (int)G_FLOATS);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

namespace {
template <int NP, bool LN>
int tok_forward(const void *a, const void *x, const void *pack, void *y, int64_t tokens, float eps, hipStream_t st)
{
    if (tokens == 0) return PMX_OK;
    if (!a || !pack || !y || (LN && !x) || tokens < 0) return PMX_ERR_INVALID;
    const int64_t tiles = (tokens + 15) / 16, want = (tiles + 3) / 4, cap = (int64_t)cu_count() * 2;
    hipLaunchKernelGGL((pmx_tok_fwd_kernel<NP, LN>), dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, st, (const uint4 *)a, (const uint4 *)x,
                       (const char *)pack, (uint4 *)y, (long)tokens, eps);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}
template <int NP, bool LN>
int tok_backward(const void *a, const void *x, const void *dy, const void *pack, void *da, void *dx, float *grad, int64_t tokens, float eps,
                 hipStream_t st)
{
    if (!grad) return PMX_ERR_INVALID;
    pmx_last_rows_value = 0;
    if (tokens == 0) return hipMemsetAsync(grad, 0, sizeof(float) * TokPack<NP>::G_FLOATS, st) == hipSuccess ? PMX_OK : PMX_ERR_HIP;
    if (!a || !dy || !pack || !da || (LN && (!x || !dx)) || tokens < 0) return PMX_ERR_INVALID;
    const size_t stage = (size_t)4 * 32 * tok_stg_row<NP>(), reduce = (size_t)4 * 4 * NP * 1024 + 4096;   // staging areas, then the block-level sums
    const size_t lds = stage > reduce ? stage : reduce;
    int64_t pairs = (tokens + 31) / 32, want = (pairs + 3) / 4, cap = (int64_t)cu_count() * 2;
    if (cap > PMX_GRAD_PARTIAL_ROWS) cap = PMX_GRAD_PARTIAL_ROWS;
    const unsigned blocks = (unsigned)(want < cap ? want : cap);
    hipLaunchKernelGGL((pmx_tok_bwd_kernel<NP, LN>), dim3(blocks), dim3(256), lds, st, (const uint4 *)a, (const uint4 *)x,
                       (const uint4 *)dy, (const char *)pack, (uint4 *)da, (uint4 *)dx, grad, (long)tokens, eps);
    pmx_last_rows_value = (int)blocks;
    if (!pmx_defer_sums_flag)
        hipLaunchKernelGGL(pmx_sum_rows_kernel, dim3((TokPack<NP>::G_FLOATS + 31) / 32), dim3(256), 0, st, grad, (int)blocks, (int)TokPack<NP>::G_FLOATS);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}
}   // namespace

extern "C" int pmx_tok96_pack(const float *w, const float *b, void *pack_dev, void *stream)
{
    if (!w || !b || !pack_dev) return PMX_ERR_INVALID;
    hipLaunchKernelGGL(pmx_tok_pack_kernel<3>, dim3(12), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, b, (const float *)nullptr,
                       (const float *)nullptr, reinterpret_cast<char *>(pack_dev));
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}
extern "C" int pmx_tok96_forward(const void *a_dev, const void *pack_dev, void *y_dev, int64_t tokens, void *stream)
{
    return tok_forward<3, false>(a_dev, nullptr, pack_dev, y_dev, tokens, 0.f, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pmx_tok96_backward(const void *a_dev, const void *dy_dev, const void *pack_dev, void *da_dev, float *grad_dev, int64_t tokens, void *stream)
{
    return tok_backward<3, false>(a_dev, nullptr, dy_dev, pack_dev, da_dev, nullptr, grad_dev, tokens, 0.f, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pmx_tok96_backward_res(const void *a_dev, const void *dy_dev, const void *pack_dev, const void *res_dev, void *da_dev, float *grad_dev,
                                      int64_t tokens, void *stream)
{
    if (res_dev && res_dev == da_dev) return PMX_ERR_INVALID;          // the operands are read through restrict pointers
    return tok_backward<3, false>(a_dev, res_dev, dy_dev, pack_dev, da_dev, nullptr, grad_dev, tokens, 0.f, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pmx_tok32ln_pack(const float *w, const float *b, const float *gamma, const float *beta, void *pack_dev, void *stream)
{
    if (!w || !b || !gamma || !beta || !pack_dev) return PMX_ERR_INVALID;
    hipLaunchKernelGGL(pmx_tok_pack_kernel<1>, dim3(4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, b, gamma, beta,
                       reinterpret_cast<char *>(pack_dev));
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}
extern "C" int pmx_tok32ln_forward(const void *x_dev, const void *a_dev, const void *pack_dev, void *y_dev, int64_t tokens, float eps, void *stream)
{
    return tok_forward<1, true>(a_dev, x_dev, pack_dev, y_dev, tokens, eps, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pmx_tok32ln_backward(const void *x_dev, const void *a_dev, const void *dy_dev, const void *pack_dev, void *dx_dev, void *da_dev,
                                    float *grad_dev, int64_t tokens, float eps, void *stream)
{
    return tok_backward<1, true>(a_dev, x_dev, dy_dev, pack_dev, da_dev, dx_dev, grad_dev, tokens, eps, reinterpret_cast<hipStream_t>(stream));
}
