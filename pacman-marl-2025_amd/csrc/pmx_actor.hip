// pmx_actor.hip -- the actor's convolutional tower of MAPPOAgent (pacman_mappo_resnet.py:49-67, 104-113) as ONE forward
// and ONE backward kernel for gfx950:
//     conv3x3(8->16) GELU conv3x3(16->32) GELU  3 x [conv3x3 GN(4) GELU conv3x3 GN(4) (+x) GELU]
// bf16 operands on v_mfma_f32_16x16x32_bf16, fp32 accumulation / GroupNorm / GELU.
//
// Mapping.  One WAVEFRONT owns one sample from the first convolution to the last; nothing is shared between waves, so
// there is no barrier anywhere in the forward kernel.  A sample's activation lives in the wave's private LDS map as
// [padded position][32 channels] bf16 (64 bytes per position, plain addresses: see map_off).  Positions are the linear index
// q = (row+1)*(W+2) +
// (col+1) of the zero-padded board, so a 3x3 tap is a constant shift of q and a tile of 16 consecutive q is one MFMA
// column block; the two pad columns inside a tile cost 2/16 of the work and are masked to zero on every write, which keeps
// the padding zero for the next layer.  Every layer is run as 32 -> 32 channels (the stem's missing channels are zero
// weights), which makes all eight layers one loop body:
//     D[co][pos] += A[co][ci] (weights of one tap, 16 x 32)  x  B[ci][pos] (16 positions, one ds_read_b128 per lane)
// The accumulator of a tile then holds, per lane, FOUR CONSECUTIVE CHANNELS of ONE position -- the same ownership in
// every layer ("P layout": lane (p = lane & 15, g = lane >> 4), tile t, half m owns channels 16m + 4g .. +3 of position
// 16t + p).  Bias, GroupNorm statistics (two shuffles levels over the 32 lanes of a group), the residual add (the block
// input stays in registers, in P layout), exact GELU and the bf16 pack are therefore lane-local, and the result goes back
// to the LDS map with one ds_write_b64.  Weights never touch LDS: a layer's 18 operand fragments (72 VGPRs) are loaded
// from a pre-packed global buffer (L2-resident, 18 KB per layer) while the previous layer's epilogue runs.
//
// Training additionally dumps, per layer, the pre-activation (bf16-rounded convolution output) and the activation in P
// layout (512 contiguous bytes per wave store), which is all the backward kernel needs; it walks the layers in reverse
// with the same ownership: GELU' and the GroupNorm backward are lane-local again, the input gradient is the same
// convolution loop with flipped weights, and the weight gradient contracts over POSITIONS, whose operands come out of the
// very same [position][channel] maps through ds_read_b64_tr_b16 (the hardware transposing read).
#pragma clang fp contract(fast)
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/pmx.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

constexpr int NLAYER = 8;
constexpr int GUARD = 1;                       // map positions in front of padded index 0 (a tap reads q - WP - 1 >= -1)
constexpr int FRAG_PER_LAYER = 2 * 9 * 64 * 8; // bf16 elements of one layer's operand fragments
// byte offsets inside the packed-parameter buffer
constexpr size_t PACK_FWD = 0;
constexpr size_t PACK_BWD = PACK_FWD + (size_t)NLAYER * FRAG_PER_LAYER * 2;
constexpr size_t PACK_BIAS = PACK_BWD + (size_t)NLAYER * FRAG_PER_LAYER * 2;
constexpr size_t PACK_GNW = PACK_BIAS + NLAYER * 32 * 4;
constexpr size_t PACK_GNB = PACK_GNW + NLAYER * 32 * 4;
constexpr size_t PACK_BYTES = PACK_GNB + NLAYER * 32 * 4;
static_assert(PACK_BYTES == PMX_ACTOR_PACK_BYTES, "include/pmx.h and pmx_actor.hip disagree on the pack size");
// float offsets inside the gradient buffer
constexpr int GRAD_W = 0;                      // [8][36 tiles][64 lanes][4]
constexpr int GRAD_B = GRAD_W + NLAYER * 36 * 256;
constexpr int GRAD_GNW = GRAD_B + NLAYER * 32;
constexpr int GRAD_GNB = GRAD_GNW + NLAYER * 32;
constexpr int GRAD_FLOATS = GRAD_GNB + NLAYER * 32;
constexpr int W_PART_ROWS = 128;               // most sample chunks (partial rows) of the weight-gradient kernel
static_assert(GRAD_FLOATS == PMX_ACTOR_GRAD_FLOATS, "include/pmx.h and pmx_actor.hip disagree on the gradient size");

__host__ __device__ constexpr int map_positions(int nt, int wp) { return GUARD + wp + 32 * ((nt + 1) / 2) + wp + 3; }

// byte offset of 16-byte chunk `chunk` (channels 8*chunk .. +7) of map position `pos`
// (No bank swizzle: an XOR of the chunk with bit 2 of the position makes the ds_read_b128 of an operand conflict-free, but costs
// six vector-ALU instructions of address arithmetic per read, and these kernels are bound by their vector ALU, not by the
// LDS -- 8-16 % busy with the swizzle.  Plain addresses are 2-way conflicted on that read and leave the address one add.)
__device__ __forceinline__ int map_off(int pos, int chunk) { return pos * 64 + (chunk << 4); }

__device__ __forceinline__ uint32_t pack2(float a, float b)
{
    f32x2 f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ float lo_f(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_f(uint32_t u) { return __uint_as_float(u & 0xFFFF0000u); }

// Phi(z) = 0.5 (1 + erf(z / sqrt 2)) by Abramowitz-Stegun 7.1.26 (|error of erf| <= 1.5e-7) and e = exp(-z^2 / 2), for a PAIR of
// values.  GELU(z) = z Phi(z) (nn.GELU() exact form, pacman_mappo_resnet.py:53), GELU'(z) = Phi(z) + z e / sqrt(2 pi).
// Written so that every step but the reciprocal, the exponential, |z| and the final select is one packed instruction for the pair
// (v_pk_fma_f32 / v_pk_mul_f32 run at full rate on gfx950): t = 1 / (1 + 0.3275911 |z| / sqrt 2), 0.5 erfc(|z| / sqrt 2) =
// 0.5 t (a1 + t (a2 + ...)) e with the 1 / sqrt 2 and the 0.5 folded into the constants (a product of constants rounded once, and an
// exact power of two).
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 unpack2(uint32_t u) { return f32x2{lo_f(u), hi_f(u)}; }
__device__ __forceinline__ f32x2 phi_cdf2(f32x2 z, f32x2 &e)
{
    const f32x2 den = {fmaf(0.3275911f * 0.70710678118654752f, fabsf(z.x), 1.0f), fmaf(0.3275911f * 0.70710678118654752f, fabsf(z.y), 1.0f)};
    const f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    f32x2 poly = fma2(t, splat2(0.5f * 1.061405429f), splat2(0.5f * -1.453152027f));
    poly = fma2(poly, t, splat2(0.5f * 1.421413741f));
    poly = fma2(poly, t, splat2(0.5f * -0.284496736f));
    poly = fma2(poly, t, splat2(0.5f * 0.254829592f));
    poly *= t;
    const f32x2 arg = (z * z) * splat2(-0.5f * 1.4426950408889634f);
    e = f32x2{__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
    const f32x2 half_tail = poly * e, rest = splat2(1.0f) - half_tail;
    return f32x2{z.x >= 0.0f ? rest.x : half_tail.x, z.y >= 0.0f ? rest.y : half_tail.y};
}
// Pass 2 of the forward for the four channels a lane holds of one position: GroupNorm + affine + skip input, GELU, validity mask
__device__ __forceinline__ uint2 gelu_quad(uint2 h2, uint2 r2, float mean, float rstd, const float (&gw)[4], const float (&gb)[4], float vm)
{
    f32x2 v[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const f32x2 hv = unpack2(k ? h2.y : h2.x), rv = unpack2(k ? r2.y : r2.x);
        const f32x2 z = rv + fma2((hv - splat2(mean)) * splat2(rstd), f32x2{gw[2 * k], gw[2 * k + 1]}, f32x2{gb[2 * k], gb[2 * k + 1]});
        f32x2 e;
        v[k] = z * phi_cdf2(z, e) * splat2(vm);
    }
    return uint2{pack2(v[0].x, v[0].y), pack2(v[1].x, v[1].y)};
}
// Pass 1 of the data gradient for the four channels a lane holds of one position: z from the saved convolution output h (GroupNorm,
// affine, + the skip input x), dz = dY GELU'(z) rounded to bf16 as autograd would round it; the bf16-rounded dz joins the lane's
// sums for the affine gradients (dgb += dz, dgw += dz xhat).  68 vector instructions (the scalar form, as the compiler packed it,
// took 95).
__device__ __forceinline__ uint2 dgelu_quad(uint2 dy, uint2 h2, uint2 x2, float rstd, float nmr, const f32x2 (&gw2)[2], const f32x2 (&gb2)[2],
                                            f32x2 (&dgw2)[2], f32x2 (&dgb2)[2])
{
    f32x2 xh[2], dzf[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const f32x2 hv = unpack2(k ? h2.y : h2.x), xv = unpack2(k ? x2.y : x2.x), d = unpack2(k ? dy.y : dy.x);
        xh[k] = fma2(hv, splat2(rstd), splat2(nmr));
        const f32x2 z = xv + fma2(xh[k], gw2[k], gb2[k]);
        f32x2 e;
        const f32x2 phi = phi_cdf2(z, e);
        dzf[k] = d * fma2(z * splat2(0.3989422804014327f), e, phi);
    }
    uint2 dzq = {pack2(dzf[0].x, dzf[0].y), pack2(dzf[1].x, dzf[1].y)};
    asm volatile("" : "+v"(dzq.x), "+v"(dzq.y));
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const f32x2 dzr = unpack2(k ? dzq.y : dzq.x);
        dgb2[k] += dzr;
        dgw2[k] = fma2(dzr, xh[k], dgw2[k]);
    }
    return dzq;
}

template <typename T> __device__ __forceinline__ float in_to_f(T v);
template <> __device__ __forceinline__ float in_to_f<uint8_t>(uint8_t v) { return (float)v; }
template <> __device__ __forceinline__ float in_to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float in_to_f<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }

__device__ __forceinline__ void wave_lds_fence()
{
    // LDS operations of one wave execute in order; this only stops the compiler from moving them across the point
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Cross-lane sums with DPP row operations (v_add_f32 with a lane-permuting source: one vector instruction per step) instead of
// ds_bpermute round trips through the LDS pipeline: after xor 1, xor 2 (quad permutes), row_half_mirror and row_mirror every
// lane of a 16-lane row holds the row's sum.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v)
{
#ifdef PMX_ACTOR_NO_DPP         /* A/B: the ds_bpermute form */
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
#endif
    v += dpp_f<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);       // row_half_mirror
    v += dpp_f<0x140>(v);       // row_mirror
    return v;
}
// sum over the 32 lanes that share a GroupNorm group in P layout: all p (lane bits 0..3) and g & 1 (bit 4)
__device__ __forceinline__ float group_sum(float v)
{
    v = row16_sum(v);
#ifdef PMX_ACTOR_NO_DPP
    v += __shfl_xor(v, 16, 64);
    return v;
#else
    float a = v, b = v;          // xor-16 step: the odd rows of a swap with the even rows of b (v_permlane16_swap, gfx950)
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
#endif
}
// sum over the 16 lanes with the same g (same channels, different positions)
__device__ __forceinline__ float pos_sum(float v) { return row16_sum(v); }

struct Geom {
    int H, W, WP, HW, MP;
};

// Copies one sample's observation planes [8][H][W] into channels 0..7 of the map and clears channels 8..31 of the
// board's cells (they hold the previous sample's activations).
template <typename IN_T>
__device__ __forceinline__ void load_obs(const IN_T *__restrict__ obs, char *map, const Geom &G, int lane)
{
    for (int i = lane; i < G.HW; i += 64) {
        const int row = i / G.W, col = i - row * G.W;
        const int pos = (row + 1) * G.WP + col + 1 + GUARD;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = in_to_f<IN_T>(obs[c * G.HW + i]);
        uint4 w = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
        *reinterpret_cast<uint4 *>(map + map_off(pos, 0)) = w;
        const uint4 z = {0, 0, 0, 0};
#pragma unroll
        for (int ch = 1; ch < 4; ++ch) *reinterpret_cast<uint4 *>(map + map_off(pos, ch)) = z;
    }
}

__device__ __forceinline__ void load_frags(bf16x8 (&A)[2][9], const short *__restrict__ frag, int lane)
{
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int k = 0; k < 9; ++k) A[m][k] = *reinterpret_cast<const bf16x8 *>(frag + ((m * 9 + k) * 64 + lane) * 8);
}

// One tile of the 3x3 convolution: 9 taps x 2 output halves on the map.  lane_base = map + (p + GUARD) * 64 + g * 16 (the lane's
// part of every operand address), centre = byte offset of the tile's first position (wave-uniform), wp64 = (W + 2) * 64.
__device__ __forceinline__ void conv_tile(const char *lane_base, const bf16x8 (&A)[2][9], int centre, int wp64, f32x4 &a0, f32x4 &a1)
{
    a0 = f32x4{0.f, 0.f, 0.f, 0.f};
    a1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const char *row = lane_base + (centre + (ky - 1) * wp64 - 64);       // the three taps of a row are 64 bytes apart
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const bf16x8 b = *reinterpret_cast<const bf16x8 *>(row + kx * 64);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0][ky * 3 + kx], b, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1][ky * 3 + kx], b, a1, 0, 0, 0);
        }
    }
}

// P-layout dump: 8 bytes per lane per (tile, half); a sample is NT * 2 * 512 bytes
__device__ __forceinline__ size_t dump_index(size_t sample, int NT, int t, int m, int lane) { return ((sample * NT + t) * 2 + m) * 64 + lane; }

// ---------------------------------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------------------------------
// Forward-kernel tuning, measured on one box with tools/ab_build.sh / ab_run.sh (8 192 samples, training / inference variant):
// RD 2 + late affine loads 605 / 615 us, RD 4 + early 625 / 650 us, RD 6 636 / 654 us -- past two tiles the registers a
// deeper window costs (spills, each with an s_waitcnt vmcnt(0) that drains every store in flight) outweigh the latency it hides
#ifndef PMX_ACTOR_GW_EARLY
#define PMX_ACTOR_GW_EARLY 0                                  // 1: affine parameters loaded before the statistics instead of after
#endif
#ifndef PMX_ACTOR_RD
#define PMX_ACTOR_RD 2                                        // tiles of residual in flight ahead of pass 2
#endif
// Development aid (-DPMX_ACTOR_TIMING, never in the shipped build): one wave's s_memtime cycles per phase, summed over its
// layers and samples, read back with pmx_actor_ticks_read (tools/actor_ticks.py)
#ifdef PMX_ACTOR_TIMING
__device__ unsigned long long pmx_actor_ticks[16];
#define PMX_TICK(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define PMX_TICK_ADD(i, a, b) tick_sum[i] += (b) - (a)
#else
#define PMX_TICK(var)
#define PMX_TICK_ADD(i, a, b)
#endif
template <int NT, typename IN_T, bool SAVE>
__global__ __launch_bounds__(256, 2) void pmx_actor_fwd_kernel(const IN_T *__restrict__ obs, const char *__restrict__ pack,
                                                              uint2 *__restrict__ feat, uint2 *__restrict__ hsave,
                                                              uint2 *__restrict__ ysave, float *__restrict__ stats,
                                                              uint2 *__restrict__ rtmp, int B, int H, int W, float eps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 15, g = lane >> 4;
    Geom G;
    G.H = H, G.W = W, G.WP = W + 2, G.HW = H * W, G.MP = map_positions(NT, W + 2);
    char *map = smem + (size_t)wave * G.MP * 64;
    for (int i = lane; i < G.MP * 4; i += 64) reinterpret_cast<uint4 *>(map)[i] = uint4{0, 0, 0, 0};
    // position (tile t, column p) -> board cell (or -1): the index of the feature row the last layer writes.  A table, because the
    // row / column split is a division by the run-time padded width, which the compiler evaluated for every tile of EVERY layer
    // (~15 vector instructions each, a tenth of the kernel's vector work) although only the last layer stores features.
    __shared__ int cell_tab[NT * 16];
    for (int i = threadIdx.x; i < NT * 16; i += 256) {
        const int q = G.WP + i, row = q / G.WP, col = q - row * G.WP;
        cell_tab[i] = (col >= 1 && col <= W && row <= H) ? (row - 1) * W + (col - 1) : -1;
    }
    __syncthreads();
    // which of the lane's NT positions are board cells
    uint32_t vmask = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int q = G.WP + 16 * t + p, row = q / G.WP, col = q - row * G.WP;
        if (col >= 1 && col <= W && row <= H) vmask |= 1u << t;
    }
    const short *fw = reinterpret_cast<const short *>(pack + PACK_FWD);
    const float *biasp = reinterpret_cast<const float *>(pack + PACK_BIAS);
    const float *gnwp = reinterpret_cast<const float *>(pack + PACK_GNW);
    const float *gnbp = reinterpret_cast<const float *>(pack + PACK_GNB);
    const float inv_n = 1.0f / (float)(8 * G.HW);

#ifdef PMX_ACTOR_TIMING
    unsigned long long tick_sum[4] = {0, 0, 0, 0};
#endif
    for (int s = blockIdx.x * 4 + wave; s < B; s += gridDim.x * 4) {
        PMX_TICK(tk_s);
        wave_lds_fence();
        load_obs<IN_T>(obs + (size_t)s * 8 * G.HW, map, G, lane);
        wave_lds_fence();
        // values that do not change from sample to sample or layer to layer (weight fragments of layer 0, masks, LDS
        // addresses) must not be hoisted out of these loops -- there are hundreds and they would live in scratch; the empty
        // asm statements make their inputs opaque at the point of use
        const short *fws = fw;
        asm volatile("" : "+s"(fws));
        bf16x8 A[2][9];
        load_frags(A, fws, lane);
#pragma unroll 1
        for (int li = 0; li < NLAYER; ++li) {
            int l = li;
            asm volatile("" : "+s"(l));
            // everything below is branch-free: a layer without GroupNorm normalises with mean 0 / rstd 1 / weight 1 / bias 0
            // (the pack holds those), a layer without the skip connection selects 0 for the residual (a select, not a
            // product: the slot it would read may hold anything, NaNs included)
            const bool has_gn = l >= 2;
            const bool res_on = l >= 3 && (l & 1);
            int WPv = G.WP, pq = p + GUARD;
            uint32_t vmk = vmask;
            asm volatile("" : "+s"(WPv));
            asm volatile("" : "+v"(pq), "+v"(vmk));
            // the activation dump of this layer and of the block input two layers back (the residual).  Training keeps every
            // layer's dump for the backward pass; inference needs one slot per wave, for the residual only (written by
            // layers 1, 3, 5, read back by the same lanes two layers later: L2-resident)
            uint2 *ydst = SAVE ? ysave + dump_index((size_t)l * B + s, NT, 0, 0, lane)
                               : rtmp + dump_index((size_t)blockIdx.x * 4 + wave, NT, 0, 0, lane);
            const uint2 *rsrc = SAVE ? ysave + dump_index((size_t)(l >= 2 ? l - 2 : 0) * B + s, NT, 0, 0, lane) : ydst;
            const bool write_y = SAVE || (l & 1);
            float bias[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) bias[m][r] = biasp[l * 32 + 16 * m + 4 * g + r];
            PMX_TICK(tk0);
            if (li == 0) PMX_TICK_ADD(3, tk_s, tk0);
            uint2 hp[NT][2];                               // bf16-rounded convolution output (+ bias), packed
            float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
            const char *rbase = map + pq * 64 + g * 16;    // the lane's part of every operand read address
            char *wbase = map + pq * 64 + g * 8;           // ... and of every P-layout write (chunk 2m + (g >> 1), half g & 1)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 a[2];
                conv_tile(rbase, A, (WPv + 16 * t) * 64, WPv * 64, a[0], a[1]);
                const float vm = ((vmk >> t) & 1) ? 1.0f : 0.0f;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (a[m][r] + bias[m][r]) * vm;
                    uint2 h2 = {pack2(v[0], v[1]), pack2(v[2], v[3])};
                    asm volatile("" : "+v"(h2.x), "+v"(h2.y));   // pins the pack next to its tile (else it sinks to pass 2 and
                                                                 // all NT x 8 fp32 accumulators stay live through the layer)
                    hp[t][m] = h2;
                    const float h0 = lo_f(h2.x), h1 = hi_f(h2.x), h2f = lo_f(h2.y), h3 = hi_f(h2.y);
                    s1[m] += (h0 + h1) + (h2f + h3);
                    s2[m] += fmaf(h0, h0, h1 * h1) + fmaf(h2f, h2f, h3 * h3);
                    if (SAVE) hsave[dump_index((size_t)l * B + s, NT, t, m, lane)] = h2;
                }
                __builtin_amdgcn_sched_barrier(0);         // one tile at a time: bounds the registers the scheduler spends on overlap
            }
            PMX_TICK(tk1);
            PMX_TICK_ADD(0, tk0, tk1);
            // the map has been read for the last time in this layer: the next layer's weights can start to arrive
            load_frags(A, fws + (size_t)(l + 1 < NLAYER ? l + 1 : l) * FRAG_PER_LAYER, lane);
            // ... and so can everything pass 2 reads from global memory: the affine parameters and the whole residual (the block
            // input of two layers back), RD tiles ahead of their use.  Issued here, the latency hides behind the statistics;
            // loaded tile by tile inside pass 2 each of the 2 NT reads was a full round trip the wave sat out (half the
            // kernel's time at two waves per SIMD).
            constexpr int RD = PMX_ACTOR_RD;
            float gw[2][4], gb[2][4];
#if PMX_ACTOR_GW_EARLY
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) gw[m][r] = gnwp[l * 32 + 16 * m + 4 * g + r], gb[m][r] = gnbp[l * 32 + 16 * m + 4 * g + r];
#endif
            uint2 rres[NT][2];
#pragma unroll
            for (int t = 0; t < NT; ++t) rres[t][0] = rres[t][1] = uint2{0u, 0u};
            if (res_on) {
#pragma unroll
                for (int t = 0; t < RD; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m) rres[t][m] = rsrc[(t * 2 + m) * 64];
            }
            float mean[2], rstd[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const float a1 = group_sum(s1[m]) * inv_n, a2 = group_sum(s2[m]) * inv_n;
                mean[m] = has_gn ? a1 : 0.0f;
                rstd[m] = has_gn ? __builtin_amdgcn_rsqf(fmaxf(a2 - a1 * a1, 0.0f) + eps) : 1.0f;
            }
            if (SAVE && p == 0 && (g & 1) == 0) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    float *st = stats + (((size_t)l * B + s) * 4 + 2 * m + (g >> 1)) * 2;
                    st[0] = mean[m], st[1] = rstd[m];
                }
            }
#if !PMX_ACTOR_GW_EARLY
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) gw[m][r] = gnwp[l * 32 + 16 * m + 4 * g + r], gb[m][r] = gnbp[l * 32 + 16 * m + 4 * g + r];
#endif
            wave_lds_fence();
            PMX_TICK(tk2);
            PMX_TICK_ADD(1, tk1, tk2);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float vm = ((vmk >> t) & 1) ? 1.0f : 0.0f;
                if (t + RD < NT && res_on) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) rres[t + RD][m] = rsrc[((t + RD) * 2 + m) * 64];
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const uint2 y2 = gelu_quad(hp[t][m], rres[t][m], mean[m], rstd[m], gw[m], gb[m], vm);
                    *reinterpret_cast<uint2 *>(wbase + (WPv + 16 * t) * 64 + m * 32) = y2;
                    if (write_y) ydst[(t * 2 + m) * 64] = y2;
                    if (l == NLAYER - 1) {
                        const int ci = cell_tab[16 * t + p];
                        if (ci >= 0) feat[((size_t)s * G.HW + ci) * 8 + 4 * m + g] = y2;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            wave_lds_fence();
            PMX_TICK(tk3);
            PMX_TICK_ADD(2, tk2, tk3);
        }
    }
#ifdef PMX_ACTOR_TIMING
    if (blockIdx.x == 5 && wave == 1 && lane == 0)
        for (int i = 0; i < 4; ++i) pmx_actor_ticks[(SAVE ? 4 : 0) + i] += tick_sum[i];
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Forward for SMALL batches: one BLOCK (four waves) per sample, each wave owning a contiguous quarter of the position tiles.
// With one wave per sample a 1 024-sample minibatch is 1 024 waves on 1 024 SIMDs, each walking its sample's 8 layers alone for
// ~100 us -- a third of the reference-size (512-sample) optimizer step.  Split four ways a sample's layer is: every wave
// convolves ITS tiles from the block's shared LDS map and leaves its partial GroupNorm sums in LDS; barrier (all reads of the
// map done, all partial sums there); every wave normalises / activates its tiles and writes them back; barrier.  Same arithmetic
// per element as pmx_actor_fwd_kernel (the statistics are summed in another order), same dumps for the backward kernels.
// ---------------------------------------------------------------------------------------------------------------
template <typename IN_T>
__device__ __forceinline__ void load_obs_block(const IN_T *__restrict__ obs, char *map, const Geom &G, int tid, int nthreads)
{
    for (int i = tid; i < G.HW; i += nthreads) {
        const int row = i / G.W, col = i - row * G.W;
        const int pos = (row + 1) * G.WP + col + 1 + GUARD;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = in_to_f<IN_T>(obs[c * G.HW + i]);
        uint4 w = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
        *reinterpret_cast<uint4 *>(map + map_off(pos, 0)) = w;
        const uint4 z = {0, 0, 0, 0};
#pragma unroll
        for (int ch = 1; ch < 4; ++ch) *reinterpret_cast<uint4 *>(map + map_off(pos, ch)) = z;
    }
}

template <int NT, typename IN_T, bool SAVE, int WS>
__global__ __launch_bounds__(256, 2) void pmx_actor_fwd_split_kernel(const IN_T *__restrict__ obs, const char *__restrict__ pack,
                                                                    uint2 *__restrict__ feat, uint2 *__restrict__ hsave,
                                                                    uint2 *__restrict__ ysave, float *__restrict__ stats,
                                                                    uint2 *__restrict__ rtmp, int B, int H, int W, float eps)
{
    constexpr int NTW = (NT + WS - 1) / WS;                    // tiles per wave; WS waves per sample, 4 / WS samples per block
    constexpr int SPB = 4 / WS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 15, g = lane >> 4;
    Geom G;
    G.H = H, G.W = W, G.WP = W + 2, G.HW = H * W, G.MP = map_positions(NT, W + 2);
    const int sub = __builtin_amdgcn_readfirstlane(wave / WS), wq = __builtin_amdgcn_readfirstlane(wave % WS);
    char *map = smem + (size_t)sub * G.MP * 64;
    float *xs = reinterpret_cast<float *>(smem + (size_t)SPB * G.MP * 64) + sub * (WS * 8);   // [WS waves][4 groups][sum, sum of squares]
    for (int i = threadIdx.x; i < SPB * G.MP * 4; i += 256) reinterpret_cast<uint4 *>(smem)[i] = uint4{0, 0, 0, 0};
    __shared__ int cell_tab[NT * 16];                           // position -> board cell of the feature row (see pmx_actor_fwd_kernel)
    for (int i = threadIdx.x; i < NT * 16; i += 256) {
        const int q = G.WP + i, row = q / G.WP, col = q - row * G.WP;
        cell_tab[i] = (col >= 1 && col <= W && row <= H) ? (row - 1) * W + (col - 1) : -1;
    }
    const int t0 = __builtin_amdgcn_readfirstlane(wq * NTW);
    const int tid_s = threadIdx.x - sub * (WS * 64);           // thread index within the sample's waves
    uint32_t vmask = 0;                                         // which of the lane's positions in the wave's tiles are board cells
#pragma unroll
    for (int tl = 0; tl < NTW; ++tl) {
        const int q = G.WP + 16 * (t0 + tl) + p, row = q / G.WP, col = q - row * G.WP;
        if (t0 + tl < NT && col >= 1 && col <= W && row <= H) vmask |= 1u << tl;
    }
    const short *fw = reinterpret_cast<const short *>(pack + PACK_FWD);
    const float *biasp = reinterpret_cast<const float *>(pack + PACK_BIAS);
    const float *gnwp = reinterpret_cast<const float *>(pack + PACK_GNW);
    const float *gnbp = reinterpret_cast<const float *>(pack + PACK_GNB);
    const float inv_n = 1.0f / (float)(8 * G.HW);

    for (int s0 = blockIdx.x * SPB; s0 < B; s0 += gridDim.x * SPB) {
        // a block's samples walk the layers in lock-step (block-wide barriers); a slot past the end of the batch re-does the
        // last sample (same values to the same addresses: harmless)
        const int s = s0 + sub < B ? s0 + sub : B - 1;
        __syncthreads();                                        // the previous sample's last layer has been written and read
        load_obs_block<IN_T>(obs + (size_t)s * 8 * G.HW, map, G, tid_s, WS * 64);
        __syncthreads();
        const short *fws = fw;
        asm volatile("" : "+s"(fws));
        bf16x8 A[2][9];
        load_frags(A, fws, lane);
#pragma unroll 1
        for (int li = 0; li < NLAYER; ++li) {
            int l = li;
            asm volatile("" : "+s"(l));
            const bool has_gn = l >= 2;
            const bool res_on = l >= 3 && (l & 1);
            int WPv = G.WP, pq = p + GUARD;
            uint32_t vmk = vmask;
            asm volatile("" : "+s"(WPv));
            asm volatile("" : "+v"(pq), "+v"(vmk));
            uint2 *ydst = SAVE ? ysave + dump_index((size_t)l * B + s, NT, 0, 0, lane) : rtmp + dump_index((size_t)blockIdx.x * SPB + sub, NT, 0, 0, lane);
            const uint2 *rsrc = SAVE ? ysave + dump_index((size_t)(l >= 2 ? l - 2 : 0) * B + s, NT, 0, 0, lane) : ydst;
            const bool write_y = SAVE || (l & 1);
            float bias[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) bias[m][r] = biasp[l * 32 + 16 * m + 4 * g + r];
            uint2 hp[NTW][2];
            float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
            const char *rbase = map + pq * 64 + g * 16;
            char *wbase = map + pq * 64 + g * 8;
#pragma unroll
            for (int tl = 0; tl < NTW; ++tl) {
                const int t = t0 + tl;
                hp[tl][0] = hp[tl][1] = uint2{0u, 0u};
                if (t < NT) {                                   // wave-uniform
                    f32x4 a[2];
                    conv_tile(rbase, A, (WPv + 16 * t) * 64, WPv * 64, a[0], a[1]);
                    const float vm = ((vmk >> tl) & 1) ? 1.0f : 0.0f;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (a[m][r] + bias[m][r]) * vm;
                        uint2 h2 = {pack2(v[0], v[1]), pack2(v[2], v[3])};
                        asm volatile("" : "+v"(h2.x), "+v"(h2.y));
                        hp[tl][m] = h2;
                        const float h0 = lo_f(h2.x), h1 = hi_f(h2.x), h2f = lo_f(h2.y), h3 = hi_f(h2.y);
                        s1[m] += (h0 + h1) + (h2f + h3);
                        s2[m] += fmaf(h0, h0, h1 * h1) + fmaf(h2f, h2f, h3 * h3);
                        if (SAVE) hsave[dump_index((size_t)l * B + s, NT, t, m, lane)] = h2;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // the wave's partial sums of the four GroupNorm groups (group 2m + (g >> 1)) -> LDS
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const float a1 = group_sum(s1[m]), a2 = group_sum(s2[m]);
                if (p == 0 && (g & 1) == 0) {
                    xs[(wq * 4 + 2 * m + (g >> 1)) * 2] = a1;
                    xs[(wq * 4 + 2 * m + (g >> 1)) * 2 + 1] = a2;
                }
            }
            // global loads of pass 2 and of the next layer, in flight across the barrier
            load_frags(A, fws + (size_t)(l + 1 < NLAYER ? l + 1 : l) * FRAG_PER_LAYER, lane);
            float gw[2][4], gb[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) gw[m][r] = gnwp[l * 32 + 16 * m + 4 * g + r], gb[m][r] = gnbp[l * 32 + 16 * m + 4 * g + r];
            uint2 rres[NTW][2];
#pragma unroll
            for (int tl = 0; tl < NTW; ++tl) {
                rres[tl][0] = rres[tl][1] = uint2{0u, 0u};
                if (res_on && t0 + tl < NT) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) rres[tl][m] = rsrc[((t0 + tl) * 2 + m) * 64];
                }
            }
            __syncthreads();                                    // every wave has read the map and left its sums
            float mean[2], rstd[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int grp = 2 * m + (g >> 1);
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int w = 0; w < WS; ++w) { a1 += xs[(w * 4 + grp) * 2]; a2 += xs[(w * 4 + grp) * 2 + 1]; }
                a1 *= inv_n; a2 *= inv_n;
                mean[m] = has_gn ? a1 : 0.0f;
                rstd[m] = has_gn ? __builtin_amdgcn_rsqf(fmaxf(a2 - a1 * a1, 0.0f) + eps) : 1.0f;
            }
            if (SAVE && wq == 0 && p == 0 && (g & 1) == 0) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    float *st = stats + (((size_t)l * B + s) * 4 + 2 * m + (g >> 1)) * 2;
                    st[0] = mean[m], st[1] = rstd[m];
                }
            }
#pragma unroll
            for (int tl = 0; tl < NTW; ++tl) {
                const int t = t0 + tl;
                if (t < NT) {
                    const float vm = ((vmk >> tl) & 1) ? 1.0f : 0.0f;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const uint2 y2 = gelu_quad(hp[tl][m], rres[tl][m], mean[m], rstd[m], gw[m], gb[m], vm);
                        *reinterpret_cast<uint2 *>(wbase + (WPv + 16 * t) * 64 + m * 32) = y2;
                        if (write_y) ydst[(t * 2 + m) * 64] = y2;
                        if (l == NLAYER - 1) {
                            const int ci = cell_tab[16 * t + p];
                            if (ci >= 0) feat[((size_t)s * G.HW + ci) * 8 + 4 * m + g] = y2;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __syncthreads();                                    // the layer's output is complete in the map (and xs may be rewritten)
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward, data path: one wave owns a sample from the last layer back to the first, mirroring the forward kernel.  The
// gradient travels from layer to layer in REGISTERS (P layout, packed bf16 -- the rounding autograd applies to a bf16
// tensor), the skip-connection gradient of a block as well; per layer the kernel reads the saved pre-activation (and the
// skip input), writes dH -- the gradient of the convolution output -- to its LDS map for the input-gradient convolution
// (same loop as forward, flipped weights), and dumps dH to global memory in the OPERAND layout the weight-gradient kernel
// wants (transposed through ds_read_b64_tr_b16: A[co][k = position]), so that kernel needs no map for it.  The bias and
// GroupNorm-affine gradients are summed per wave in LDS over all its samples and added to global memory once at the end.
// ---------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256, 1) void pmx_actor_bwd_data_kernel(const char *__restrict__ pack, const uint2 *__restrict__ dfeat,
                                                                   const uint2 *__restrict__ hsave, const uint2 *__restrict__ ysave,
                                                                   const float *__restrict__ stats, bf16x8 *__restrict__ dasave,
                                                                   uint2 *__restrict__ sktmp, float *__restrict__ accpart, int B, int H,
                                                                   int W)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 15, g = lane >> 4;
    Geom G;
    G.H = H, G.W = W, G.WP = W + 2, G.HW = H * W, G.MP = map_positions(NT, W + 2);
    constexpr int KS = (NT + 1) / 2;
    constexpr int RD = 4;                                     // tiles of skip-connection data in flight ahead of their use
    char *map = smem + (size_t)wave * (G.MP * 64 + NLAYER * 96 * 4);
    float *acc = reinterpret_cast<float *>(map + (size_t)G.MP * 64);       // [8 layers][bias 32 | gn weight 32 | gn bias 32]
    for (int i = lane; i < G.MP * 4; i += 64) reinterpret_cast<uint4 *>(map)[i] = uint4{0, 0, 0, 0};
    for (int i = lane; i < NLAYER * 96; i += 64) acc[i] = 0.0f;
    uint32_t vmask = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int q = G.WP + 16 * t + p, row = q / G.WP, col = q - row * G.WP;
        if (col >= 1 && col <= W && row <= H) vmask |= 1u << t;
    }
    const short *bw = reinterpret_cast<const short *>(pack + PACK_BWD);
    const float *gnwp = reinterpret_cast<const float *>(pack + PACK_GNW);
    const float *gnbp = reinterpret_cast<const float *>(pack + PACK_GNB);
    const float inv_n = 1.0f / (float)(8 * G.HW);
    const int tr_row = (lane & 15) >> 2, tr_pc = lane & 3;     // transposing reads: lane 4*row + pc of a 16-lane group
    bool any = false;

    for (int s = blockIdx.x * 4 + wave; s < B; s += gridDim.x * 4) {
        any = true;
        uint2 dv[NT][2];                                       // dY of the current layer, then its dz, then the next dY
        // the gradient arriving over a block's skip connection waits in a per-wave global slot (L2-resident, same lanes
        // write and read) from the block's second convolution to its first: 44 registers the passes below need
        uint2 *skp = sktmp + dump_index((size_t)blockIdx.x * 4 + wave, NT, 0, 0, lane);
        {
            int WPo = G.WP;
            asm volatile("" : "+s"(WPo));
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int q = WPo + 16 * t + p, row = q / WPo, col = q - row * WPo;
                const bool valid = (vmask >> t) & 1;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    dv[t][m] = uint2{0, 0};
                    if (valid) dv[t][m] = dfeat[((size_t)s * G.HW + (row - 1) * W + (col - 1)) * 8 + 4 * m + g];
                }
            }
        }
        // What a layer reads from global memory -- its saved pre-activation hp, the block input xg behind its skip connection,
        // its affine parameters and statistics -- is loaded ONE LAYER AHEAD, into registers the loop carries (at one wave per
        // SIMD the file has room and nothing else hides a round trip to L2 / HBM).
        uint2 hp[NT][2], xg[NT][2];
        float gw[2][4], gb[2][4], mean[2], rstd[2], nmr[2];
        auto load_layer_inputs = [&](int ln) {
            const bool res_n = ln >= 3 && (ln & 1);
            const int lres = ln >= 2 ? ln - 2 : 0;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gw[m][r] = gnwp[ln * 32 + 16 * m + 4 * g + r], gb[m][r] = gnbp[ln * 32 + 16 * m + 4 * g + r];
                const float *st = stats + (((size_t)ln * B + s) * 4 + 2 * m + (g >> 1)) * 2;
                mean[m] = st[0], rstd[m] = st[1];                  // (0, 1) for the layers without GroupNorm
                nmr[m] = -mean[m] * rstd[m];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    hp[t][m] = hsave[dump_index((size_t)ln * B + s, NT, t, m, lane)];
                    if (t < RD) xg[t][m] = uint2{0u, 0u};
                }
            if (res_n) {                                           // the first RD tiles; pass 1 asks for the rest as it goes
#pragma unroll
                for (int t = 0; t < RD; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m) xg[t][m] = ysave[dump_index((size_t)lres * B + s, NT, t, m, lane)];
            }
        };
        load_layer_inputs(NLAYER - 1);
#pragma unroll 1
        for (int li = NLAYER - 1; li >= 0; --li) {
            int l = li;
            asm volatile("" : "+s"(l));
            const bool has_gn = l >= 2, has_res = l >= 3 && (l & 1), adds_skip = l >= 2 && !(l & 1);
            const float gn_on = has_gn ? 1.0f : 0.0f;
            int WPv = G.WP, pq = p + GUARD;
            uint32_t vmk = vmask;
            asm volatile("" : "+s"(WPv));
            asm volatile("" : "+v"(pq), "+v"(vmk));
            // ---- pass 1: dz = dY GELU'(z), sums for the GroupNorm backward and for the affine gradients -----------------
            float S1[2], S2[2];
            float dgw[2][4], dgb[2][4];
            f32x2 dgw2[2][2], dgb2[2][2], gw2[2][2], gb2[2][2];       // pairs of channels, for the packed arithmetic of pass 1
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    dgw2[m][k] = dgb2[m][k] = f32x2{0.0f, 0.0f};
                    gw2[m][k] = f32x2{gw[m][2 * k], gw[m][2 * k + 1]}, gb2[m][k] = f32x2{gb[m][2 * k], gb[m][2 * k + 1]};
                }
            const uint2 *xsrc = ysave + dump_index((size_t)(l >= 2 ? l - 2 : 0) * B + s, NT, 0, 0, lane);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t + RD < NT) {
                    xg[t + RD][0] = xg[t + RD][1] = uint2{0u, 0u};
                    if (has_res) {
#pragma unroll
                        for (int m = 0; m < 2; ++m) xg[t + RD][m] = xsrc[((t + RD) * 2 + m) * 64];
                    }
                }
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        dv[t][m] = dgelu_quad(dv[t][m], hp[t][m], xg[t][m], rstd[m], nmr[m], gw2[m], gb2[m], dgw2[m], dgb2[m]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    dgw[m][2 * k] = dgw2[m][k].x, dgw[m][2 * k + 1] = dgw2[m][k].y, dgb[m][2 * k] = dgb2[m][k].x, dgb[m][2 * k + 1] = dgb2[m][k].y;
            if (has_res) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m) skp[(t * 2 + m) * 64] = dv[t][m];
            }
            // The two sums of the GroupNorm backward are weighted sums of the per-channel sums the affine gradients need anyway:
            // sum(gw dz) = sum_c gw_c dgb_c, sum(gw dz xhat) = sum_c gw_c dgw_c (the sample's, before they join the wave's totals).
            // dh = rstd (gw dz - S1 - xhat S2) then is two fused multiply-adds per element: A dz + (K2 h + K3).
            float Ad[2][4], K2[2], K3[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) t1 = fmaf(gw[m][r], dgb[m][r], t1), t2 = fmaf(gw[m][r], dgw[m][r], t2);
                S1[m] = group_sum(t1) * inv_n * gn_on, S2[m] = group_sum(t2) * inv_n * gn_on;
                K2[m] = -rstd[m] * rstd[m] * S2[m];
                K3[m] = -mean[m] * K2[m] - rstd[m] * S1[m];
#pragma unroll
                for (int r = 0; r < 4; ++r) Ad[m][r] = rstd[m] * gw[m][r];
            }
            wave_lds_fence();
            // the flipped weights of the input-gradient convolution: needed after pass 2, asked for now
            // ---- pass 2: dh, into the map (for the input gradient) ---------------------------------------------------------
            float dbias[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) dbias[m][r] = 0.0f;
            char *wbase = map + pq * 64 + g * 8;            // the lane's part of every P-layout write address
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float vm = ((vmk >> t) & 1) ? 1.0f : 0.0f;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const uint2 h2 = hp[t][m];
                    const float hv[4] = {lo_f(h2.x), hi_f(h2.x), lo_f(h2.y), hi_f(h2.y)};
                    const float dzr[4] = {lo_f(dv[t][m].x), hi_f(dv[t][m].x), lo_f(dv[t][m].y), hi_f(dv[t][m].y)};
                    float dh[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) dh[r] = vm * fmaf(Ad[m][r], dzr[r], fmaf(hv[r], K2[m], K3[m]));
                    const uint2 d2 = {pack2(dh[0], dh[1]), pack2(dh[2], dh[3])};
                    // the bias gradient sums what the matrix cores see (the bf16-rounded dh), like autograd on a bf16 tensor
                    dbias[m][0] += lo_f(d2.x), dbias[m][1] += hi_f(d2.x), dbias[m][2] += lo_f(d2.y), dbias[m][3] += hi_f(d2.y);
                    *reinterpret_cast<uint2 *>(wbase + (WPv + 16 * t) * 64 + m * 32) = d2;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // hp, the affine parameters and the statistics have been used for the last time: the next layer's can start to
            // arrive, and so can the gradient parked on the skip connection
            uint2 skip[NT][2];
#pragma unroll
            for (int t = 0; t < RD; ++t) skip[t][0] = skip[t][1] = uint2{0u, 0u};
            if (adds_skip) {
#pragma unroll
                for (int t = 0; t < RD; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m) skip[t][m] = skp[(t * 2 + m) * 64];
            }
            load_layer_inputs(l > 0 ? l - 1 : 0);
            const short *bws = bw + (size_t)l * FRAG_PER_LAYER;
            asm volatile("" : "+s"(bws));
            bf16x8 A[2][9];
            load_frags(A, bws, lane);
            // per-channel sums over the sample's positions -> the wave's LDS accumulators (lanes p == 0 own 4 channels each)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float b = pos_sum(dbias[m][r]), w = pos_sum(dgw[m][r]), c = pos_sum(dgb[m][r]);
                    if (p == 0) {
                        float *a3 = acc + l * 96 + 16 * m + 4 * g + r;
                        atomicAdd(a3, b); atomicAdd(a3 + 32, w); atomicAdd(a3 + 64, c);      // ds_add_f32, no read-back to wait for
                    }
                }
            wave_lds_fence();
            // ---- dH in the weight-gradient kernel's operand layout: A[co][k = position], transposed out of the map --------
            {
                bf16x8 *dst = dasave + (((size_t)l * B + s) * KS * 2) * 64 + lane;
                const char *tbase = map + (8 * g + tr_row + GUARD) * 64 + tr_pc * 8;      // the lane's part of a transposing read
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int mo = 0; mo < 2; ++mo) {
                        const char *o0 = tbase + (WPv + 32 * ks) * 64 + mo * 32;
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3))) *)(o0));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3))) *)(o0 + 256));
                        dst[(ks * 2 + mo) * 64] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
                }
            }
            // ---- input gradient: the same convolution with flipped, transposed weights; it is the next layer's dY ----------
            if (l > 0) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f32x4 a[2];
                    if (t + RD < NT) {
                        skip[t + RD][0] = skip[t + RD][1] = uint2{0u, 0u};
                        if (adds_skip) {
#pragma unroll
                            for (int m = 0; m < 2; ++m) skip[t + RD][m] = skp[((t + RD) * 2 + m) * 64];
                        }
                    }
                    conv_tile(map + pq * 64 + g * 16, A, (WPv + 16 * t) * 64, WPv * 64, a[0], a[1]);
                    const float vm = ((vmk >> t) & 1) ? 1.0f : 0.0f;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const uint32_t sx = skip[t][m].x, sy = skip[t][m].y;
                        const float v0 = (lo_f(sx) + a[m][0]) * vm, v1 = (hi_f(sx) + a[m][1]) * vm;
                        const float v2 = (lo_f(sy) + a[m][2]) * vm, v3 = (hi_f(sy) + a[m][3]) * vm;
                        uint2 nx = {pack2(v0, v1), pack2(v2, v3)};
                        asm volatile("" : "+v"(nx.x), "+v"(nx.y));
                        dv[t][m] = nx;
                    }
                    // (no scheduling barrier here: at one wave per SIMD the compiler may overlap a tile's LDS reads with the previous
                    // tile's matrix instructions -- 883 -> 857 us -- and has the accumulator file to pay for it)
                }
            }
            wave_lds_fence();
        }
    }
    // the block's four waves' sums -> one partial row per block (plain stores; a second tiny kernel adds the rows: float
    // atomics from 2 048 waves onto these 768 addresses would run at the rate of one contended row)
    (void)any;
    __syncthreads();
    for (int i = threadIdx.x; i < NLAYER * 96; i += 256) {
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += reinterpret_cast<const float *>(smem + (size_t)w * (G.MP * 64 + NLAYER * 96 * 4) + (size_t)G.MP * 64)[i];
        accpart[(size_t)blockIdx.x * (NLAYER * 96) + i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The data path for SMALL batches, WS = 2 or 4 waves per sample (the counterpart of pmx_actor_fwd_split_kernel): a wave owns a
// contiguous share of the position tiles -- its slice of dY / dz, of the saved pre-activation and of the skip data -- and the
// sample's waves share one LDS map.  Per layer: pass 1 on the own tiles, partial GroupNorm-backward sums to LDS, BARRIER (sums
// complete; nobody still reads the map for the previous layer's input gradient), pass 2 writes dH for the own tiles, BARRIER
// (map complete), then the transposed dumps (key-pair blocks dealt round-robin) and the input-gradient convolution of the own
// tiles.  At 512 samples the one-wave kernel is 512 waves walking 8 layers alone (118 us); registers are no concern here (one
// wave per SIMD either way), the point is the length of a sample's serial chain.
// ---------------------------------------------------------------------------------------------------------------
template <int NT, int WS>
__global__ __launch_bounds__(256, 2) void pmx_actor_bwd_data_split_kernel(const char *__restrict__ pack, const uint2 *__restrict__ dfeat,
                                                                         const uint2 *__restrict__ hsave, const uint2 *__restrict__ ysave,
                                                                         const float *__restrict__ stats, bf16x8 *__restrict__ dasave,
                                                                         uint2 *__restrict__ sktmp, float *__restrict__ accpart, int B,
                                                                         int H, int W)
{
    constexpr int NTW = (NT + WS - 1) / WS, SPB = 4 / WS, KS = (NT + 1) / 2;
    // large boards (7 tiles per wave): the skip input of a layer is asked for at the top of that layer instead of one layer ahead
    // -- 28 registers that would be live across the input-gradient convolution next to the parked skip gradient (the kernel
    // spilled 27 registers at its 256-register budget with them)
    constexpr bool XG_LATE = NT > 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 15, g = lane >> 4;
    Geom G;
    G.H = H, G.W = W, G.WP = W + 2, G.HW = H * W, G.MP = map_positions(NT, W + 2);
    const int sub = __builtin_amdgcn_readfirstlane(wave / WS), wq = __builtin_amdgcn_readfirstlane(wave % WS);
    // LDS: [SPB maps][4 per-wave accumulators of NLAYER * 96 floats][SPB x WS x 8 floats of partial sums]
    char *map = smem + (size_t)sub * G.MP * 64;
    float *acc = reinterpret_cast<float *>(smem + (size_t)SPB * G.MP * 64) + wave * (NLAYER * 96);
    float *xs = reinterpret_cast<float *>(smem + (size_t)SPB * G.MP * 64) + 4 * (NLAYER * 96) + sub * (WS * 8);
    for (int i = threadIdx.x; i < SPB * G.MP * 4; i += 256) reinterpret_cast<uint4 *>(smem)[i] = uint4{0, 0, 0, 0};
    for (int i = lane; i < NLAYER * 96; i += 64) acc[i] = 0.0f;
    const int t0 = __builtin_amdgcn_readfirstlane(wq * NTW);
    uint32_t vmask = 0;
#pragma unroll
    for (int tl = 0; tl < NTW; ++tl) {
        const int q = G.WP + 16 * (t0 + tl) + p, row = q / G.WP, col = q - row * G.WP;
        if (t0 + tl < NT && col >= 1 && col <= W && row <= H) vmask |= 1u << tl;
    }
    const short *bw = reinterpret_cast<const short *>(pack + PACK_BWD);
    const float *gnwp = reinterpret_cast<const float *>(pack + PACK_GNW);
    const float *gnbp = reinterpret_cast<const float *>(pack + PACK_GNB);
    const float inv_n = 1.0f / (float)(8 * G.HW);
    const int tr_row = (lane & 15) >> 2, tr_pc = lane & 3;

    for (int s0 = blockIdx.x * SPB; s0 < B; s0 += gridDim.x * SPB) {
        // a block's samples walk the layers in lock-step; a slot past the end of the batch re-does the last sample (its dumps
        // are the same values to the same addresses) but adds nothing to the parameter gradients
        const bool live = s0 + sub < B;
        const int s = live ? s0 + sub : B - 1;
        uint2 dv[NTW][2];
        uint2 *skp = sktmp + dump_index((size_t)blockIdx.x * SPB + sub, NT, 0, 0, lane);
        {
            int WPo = G.WP;
            asm volatile("" : "+s"(WPo));
#pragma unroll
            for (int tl = 0; tl < NTW; ++tl) {
                const int q = WPo + 16 * (t0 + tl) + p, row = q / WPo, col = q - row * WPo;
                const bool valid = (vmask >> tl) & 1;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    dv[tl][m] = uint2{0, 0};
                    if (valid) dv[tl][m] = dfeat[((size_t)s * G.HW + (row - 1) * W + (col - 1)) * 8 + 4 * m + g];
                }
            }
        }
        uint2 hp[NTW][2], xg[NTW][2];
        float gw[2][4], gb[2][4], mean[2], rstd[2], nmr[2];
        auto load_layer_inputs = [&](int ln) {
            const bool res_n = ln >= 3 && (ln & 1);
            const int lres = ln >= 2 ? ln - 2 : 0;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gw[m][r] = gnwp[ln * 32 + 16 * m + 4 * g + r], gb[m][r] = gnbp[ln * 32 + 16 * m + 4 * g + r];
                const float *st = stats + (((size_t)ln * B + s) * 4 + 2 * m + (g >> 1)) * 2;
                mean[m] = st[0], rstd[m] = st[1];
                nmr[m] = -mean[m] * rstd[m];
            }
#pragma unroll
            for (int tl = 0; tl < NTW; ++tl)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    hp[tl][m] = uint2{0u, 0u};
                    if (!XG_LATE) xg[tl][m] = uint2{0u, 0u};
                    if (t0 + tl < NT) {
                        hp[tl][m] = hsave[dump_index((size_t)ln * B + s, NT, t0 + tl, m, lane)];
                        if (!XG_LATE && res_n) xg[tl][m] = ysave[dump_index((size_t)lres * B + s, NT, t0 + tl, m, lane)];
                    }
                }
        };
        load_layer_inputs(NLAYER - 1);
#pragma unroll 1
        for (int li = NLAYER - 1; li >= 0; --li) {
            int l = __builtin_amdgcn_readfirstlane(li);
            asm volatile("" : "+s"(l));
            const bool has_gn = l >= 2, has_res = l >= 3 && (l & 1), adds_skip = l >= 2 && !(l & 1);
            const float gn_on = has_gn ? 1.0f : 0.0f;
            int WPv = G.WP, pq = p + GUARD;
            uint32_t vmk = vmask;
            asm volatile("" : "+s"(WPv));
            asm volatile("" : "+v"(pq), "+v"(vmk));
            if (XG_LATE) {
#pragma unroll
                for (int tl = 0; tl < NTW; ++tl)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        xg[tl][m] = uint2{0u, 0u};
                        if (has_res && t0 + tl < NT) xg[tl][m] = ysave[dump_index((size_t)(l - 2) * B + s, NT, t0 + tl, m, lane)];
                    }
            }
            // ---- pass 1 on the own tiles ---------------------------------------------------------------------------------
            float S1[2], S2[2];
            float dgw[2][4], dgb[2][4];
            f32x2 dgw2[2][2], dgb2[2][2], gw2[2][2], gb2[2][2];       // pairs of channels, for the packed arithmetic of pass 1
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    dgw2[m][k] = dgb2[m][k] = f32x2{0.0f, 0.0f};
                    gw2[m][k] = f32x2{gw[m][2 * k], gw[m][2 * k + 1]}, gb2[m][k] = f32x2{gb[m][2 * k], gb[m][2 * k + 1]};
                }
#pragma unroll
            for (int tl = 0; tl < NTW; ++tl) {
                if (t0 + tl < NT) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        dv[tl][m] = dgelu_quad(dv[tl][m], hp[tl][m], xg[tl][m], rstd[m], nmr[m], gw2[m], gb2[m], dgw2[m], dgb2[m]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    dgw[m][2 * k] = dgw2[m][k].x, dgw[m][2 * k + 1] = dgw2[m][k].y, dgb[m][2 * k] = dgb2[m][k].x, dgb[m][2 * k + 1] = dgb2[m][k].y;
            if (has_res) {
#pragma unroll
                for (int tl = 0; tl < NTW; ++tl)
                    if (t0 + tl < NT) {
#pragma unroll
                        for (int m = 0; m < 2; ++m) skp[((t0 + tl) * 2 + m) * 64] = dv[tl][m];
                    }
            }
            // the wave's partial sums of the four groups -> LDS; after the barrier every wave adds all of them
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                // (sum(gw dz) and sum(gw dz xhat) as weighted sums of the per-channel sums: see pmx_actor_bwd_data_kernel)
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) t1 = fmaf(gw[m][r], dgb[m][r], t1), t2 = fmaf(gw[m][r], dgw[m][r], t2);
                const float a1 = group_sum(t1), a2 = group_sum(t2);
                if (p == 0 && (g & 1) == 0) {
                    xs[(wq * 4 + 2 * m + (g >> 1)) * 2] = a1;
                    xs[(wq * 4 + 2 * m + (g >> 1)) * 2 + 1] = a2;
                }
            }
            const short *bws = bw + (size_t)l * FRAG_PER_LAYER;
            asm volatile("" : "+s"(bws));
            bf16x8 A[2][9];
            load_frags(A, bws, lane);                           // in flight across the barriers
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int grp = 2 * m + (g >> 1);
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int w = 0; w < WS; ++w) { a1 += xs[(w * 4 + grp) * 2]; a2 += xs[(w * 4 + grp) * 2 + 1]; }
                S1[m] = a1 * inv_n * gn_on, S2[m] = a2 * inv_n * gn_on;
            }
            float Ad[2][4], K2[2], K3[2];                       // dh = A dz + (K2 h + K3)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                K2[m] = -rstd[m] * rstd[m] * S2[m];
                K3[m] = -mean[m] * K2[m] - rstd[m] * S1[m];
#pragma unroll
                for (int r = 0; r < 4; ++r) Ad[m][r] = rstd[m] * gw[m][r];
            }
            // ---- pass 2: dh of the own tiles into the shared map ----------------------------------------------------------
            float dbias[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) dbias[m][r] = 0.0f;
            char *wbase = map + pq * 64 + g * 8;
#pragma unroll
            for (int tl = 0; tl < NTW; ++tl) {
                const int t = t0 + tl;
                if (t < NT) {
                    const float vm = ((vmk >> tl) & 1) ? 1.0f : 0.0f;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const uint2 h2 = hp[tl][m];
                        const float hv[4] = {lo_f(h2.x), hi_f(h2.x), lo_f(h2.y), hi_f(h2.y)};
                        const float dzr[4] = {lo_f(dv[tl][m].x), hi_f(dv[tl][m].x), lo_f(dv[tl][m].y), hi_f(dv[tl][m].y)};
                        float dh[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) dh[r] = vm * fmaf(Ad[m][r], dzr[r], fmaf(hv[r], K2[m], K3[m]));
                        const uint2 d2 = {pack2(dh[0], dh[1]), pack2(dh[2], dh[3])};
                        dbias[m][0] += lo_f(d2.x), dbias[m][1] += hi_f(d2.x), dbias[m][2] += lo_f(d2.y), dbias[m][3] += hi_f(d2.y);
                        *reinterpret_cast<uint2 *>(wbase + (WPv + 16 * t) * 64 + m * 32) = d2;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            uint2 skip[NTW][2];
#pragma unroll
            for (int tl = 0; tl < NTW; ++tl) {
                skip[tl][0] = skip[tl][1] = uint2{0u, 0u};
                if (adds_skip && t0 + tl < NT) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) skip[tl][m] = skp[((t0 + tl) * 2 + m) * 64];
                }
            }
            load_layer_inputs(l > 0 ? l - 1 : 0);
            if (live) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float b = pos_sum(dbias[m][r]), w = pos_sum(dgw[m][r]), c = pos_sum(dgb[m][r]);
                        if (p == 0) {
                            float *a3 = acc + l * 96 + 16 * m + 4 * g + r;
                            a3[0] += b, a3[32] += w, a3[64] += c;
                        }
                    }
            }
            __syncthreads();                                    // the layer's dH is complete in the map
            // ---- dH in the weight-gradient kernel's operand layout: the key-pair blocks dealt round-robin to the waves -----
            {
                bf16x8 *dst = dasave + (((size_t)l * B + s) * KS * 2) * 64 + lane;
                const char *tbase = map + (8 * g + tr_row + GUARD) * 64 + tr_pc * 8;
#pragma unroll 1
                for (int ks = wq; ks < KS; ks += WS) {
#pragma unroll
                    for (int mo = 0; mo < 2; ++mo) {
                        const char *o0 = tbase + (WPv + 32 * ks) * 64 + mo * 32;
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3))) *)(o0));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3))) *)(o0 + 256));
                        dst[(ks * 2 + mo) * 64] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
                }
            }
            // ---- input gradient of the own tiles ---------------------------------------------------------------------------
            if (l > 0) {
#pragma unroll
                for (int tl = 0; tl < NTW; ++tl) {
                    const int t = t0 + tl;
                    if (t < NT) {
                        f32x4 a[2];
                        conv_tile(map + pq * 64 + g * 16, A, (WPv + 16 * t) * 64, WPv * 64, a[0], a[1]);
                        const float vm = ((vmk >> tl) & 1) ? 1.0f : 0.0f;
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const uint32_t sx = skip[tl][m].x, sy = skip[tl][m].y;
                            const float v0 = (lo_f(sx) + a[m][0]) * vm, v1 = (hi_f(sx) + a[m][1]) * vm;
                            const float v2 = (lo_f(sy) + a[m][2]) * vm, v3 = (hi_f(sy) + a[m][3]) * vm;
                            uint2 nx = {pack2(v0, v1), pack2(v2, v3)};
                            asm volatile("" : "+v"(nx.x), "+v"(nx.y));
                            dv[tl][m] = nx;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();                                        // the last layer's map reads are done before the next sample's writes
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NLAYER * 96; i += 256) {
        const float *a0 = reinterpret_cast<const float *>(smem + (size_t)SPB * G.MP * 64);
        accpart[(size_t)blockIdx.x * (NLAYER * 96) + i] = (a0[i] + a0[NLAYER * 96 + i]) + (a0[2 * NLAYER * 96 + i] + a0[3 * NLAYER * 96 + i]);
    }
}

// bias / GroupNorm-affine gradients: sum of the data kernel's per-block rows into the gradient buffer
__global__ __launch_bounds__(256) void pmx_actor_sum_acc_kernel(const float *__restrict__ accpart, int n_rows, float *__restrict__ grad)
{
    // 32 columns x 8 row slices per block, slices added through LDS (a lone thread per column was a serial chain of n_rows / 4
    // memory round trips: 53 us at 2 048 rows)
    __shared__ float part[8][33];
    const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + c;                               // NLAYER * 96 = 768 columns: 24 blocks
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int r = sl;
    for (; r + 24 < n_rows; r += 32) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += accpart[(size_t)(r + 8 * k) * (NLAYER * 96) + i];
    }
    for (; r < n_rows; r += 8) acc[0] += accpart[(size_t)r * (NLAYER * 96) + i];
    part[sl][c] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (sl == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][c];
        const int l = i / 96, j = i - l * 96, kind = j >> 5, ch = j & 31;
        grad[(kind == 0 ? GRAD_B : kind == 1 ? GRAD_GNW : GRAD_GNB) + l * 32 + ch] = t;
    }
}

// weight gradients: sum of the weight-gradient kernel's per-chunk rows ([n_rows][NLAYER * 36 * 256]) into the gradient buffer
__global__ __launch_bounds__(256) void pmx_actor_sum_w_kernel(const float *__restrict__ wpart, int n_rows, float *__restrict__ grad)
{
    __shared__ float part[8][33];
    const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + c;                               // NLAYER * 36 * 256 = 73 728 columns: 2 304 blocks
    constexpr size_t ROW = (size_t)NLAYER * 36 * 256;
    float acc = 0.f;
    for (int r = sl; r < n_rows; r += 8) acc += wpart[(size_t)r * ROW + i];
    part[sl][c] = acc;
    __syncthreads();
    if (sl == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][c];
        grad[GRAD_W + i] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward, weight gradient: dW[co][ci][tap] = sum over samples and positions of dH[pos][co] X[pos + shift(tap)][ci], a
// contraction over POSITIONS.  blockIdx.y = layer, blockIdx.x = chunk of samples; a wave keeps the layer's 36 output tiles
// (2 co-halves x 2 ci-halves x 9 taps, 144 accumulator registers) across its samples.  Per sample: the A operands
// (dH, already in operand layout) come straight from global memory, the layer's input activation goes into the wave's LDS
// map once and is read back tap by tap through the transposing read.  One block-level sum through LDS and one float
// plain store per value at the end (the chunk's partial row; pmx_actor_sum_w_kernel adds the rows).
// ---------------------------------------------------------------------------------------------------------------
template <int NT, typename IN_T>
__global__ __launch_bounds__(256, 2) void pmx_actor_bwd_weight_kernel(const IN_T *__restrict__ obs, const uint2 *__restrict__ ysave,
                                                                     const bf16x8 *__restrict__ dasave, float *__restrict__ grad,
                                                                     int B, int H, int W, int per_wave)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 15, g = lane >> 4;
    const int l = blockIdx.y;
    Geom G;
    G.H = H, G.W = W, G.WP = W + 2, G.HW = H * W, G.MP = map_positions(NT, W + 2);
    constexpr int KS = (NT + 1) / 2;
    char *mapB = smem + (size_t)wave * G.MP * 64;
    for (int i = lane; i < G.MP * 4; i += 64) reinterpret_cast<uint4 *>(mapB)[i] = uint4{0, 0, 0, 0};
    const int tr_row = (lane & 15) >> 2, tr_pc = lane & 3;
    f32x4 accw[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) accw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int s_begin = (blockIdx.x * 4 + wave) * per_wave;
    const int s_end = s_begin + per_wave < B ? s_begin + per_wave : B;
    for (int s = s_begin; s < s_end; ++s) {
        int WPv = G.WP, pq = p + GUARD;
        asm volatile("" : "+s"(WPv));
        asm volatile("" : "+v"(pq));
        wave_lds_fence();
        if (l == 0) {
            load_obs<IN_T>(obs + (size_t)s * 8 * G.HW, mapB, G, lane);
        } else {
            uint2 xin[NT][2];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m) xin[t][m] = ysave[dump_index((size_t)(l - 1) * B + s, NT, t, m, lane)];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    *reinterpret_cast<uint2 *>(mapB + pq * 64 + g * 8 + (WPv + 16 * t) * 64 + m * 32) = xin[t][m];
        }
        wave_lds_fence();
        const bf16x8 *asrc = dasave + (((size_t)l * B + s) * KS * 2) * 64 + lane;
#pragma unroll 1
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 Ad0 = asrc[(ks * 2 + 0) * 64], Ad1 = asrc[(ks * 2 + 1) * 64];
            const char *tbase = mapB + (8 * g + tr_row + GUARD) * 64 + tr_pc * 8 + (WPv + 32 * ks) * 64;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const char *tx = tbase + ((ky - 1) * WPv + (kx - 1)) * 64;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3))) *)(tx + ni * 32));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3))) *)(tx + ni * 32 + 256));
                        const bf16x8 Bx = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        accw[(0 * 2 + ni) * 9 + ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ad0, Bx, accw[(0 * 2 + ni) * 9 + ky * 3 + kx], 0, 0, 0);
                        accw[(1 * 2 + ni) * 9 + ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ad1, Bx, accw[(1 * 2 + ni) * 9 + ky * 3 + kx], 0, 0, 0);
                    }
                }
        }
    }
    // block-level sum through LDS (the maps are free now), then the block's partial row of the weight gradient
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem);
    constexpr int PER_ROUND = 9;                              // 4 waves x 9 tiles x 1 KB = 36 KB of scratch per round
#pragma unroll
    for (int round = 0; round < 4; ++round) {
#pragma unroll
        for (int i = 0; i < PER_ROUND; ++i)
            *reinterpret_cast<f32x4 *>(red + ((size_t)(wave * PER_ROUND + i) * 64 + lane) * 4) = accw[round * PER_ROUND + i];
        __syncthreads();
        for (int i = threadIdx.x; i < PER_ROUND * 64; i += 256) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(red + (size_t)i * 4);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const f32x4 u = *reinterpret_cast<const f32x4 *>(red + ((size_t)w * PER_ROUND * 64 + i) * 4);
                v[0] += u[0], v[1] += u[1], v[2] += u[2], v[3] += u[3];
            }
            // the block's partial row (plain stores; pmx_actor_sum_w_kernel adds the rows): 64 blocks adding 9 216 values each
            // with float atomics onto the same addresses cost more than the row sum, at small batches most of the kernel
            float *dst = grad + ((size_t)blockIdx.x * NLAYER * 36 + (size_t)l * 36 + round * PER_ROUND) * 256 + (size_t)i * 4;
            *reinterpret_cast<f32x4 *>(dst) = v;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The weight gradient for LARGE boards (20 x 20: 28 position tiles, a 31 KB map per sample): TWO waves per sample share the
// sample's LDS map, two samples per block.  Wave `ni` of a pair keeps the 18 gradient tiles of ITS input-channel half (2
// output halves x 9 taps, 72 accumulator registers) across the pair's samples: per key-pair block it reads both dH operand
// fragments from global memory (requested one block ahead) and nine transposed B operands of its channel half from the map
// -- the same one-LDS-read-per-MFMA ratio as the four-maps-per-block kernel above, at a quarter of its LDS per wave.  The
// pair's next input activation is fetched into registers (each wave half of the tiles) while the current one is multiplied.
// Every wave writes its 18 tiles to the pair's partial row; pmx_actor_sum_w_kernel adds the rows.
// ---------------------------------------------------------------------------------------------------------------
template <int NT, typename IN_T>
__global__ __launch_bounds__(256, 2) void pmx_actor_bwd_weight_split_kernel(const IN_T *__restrict__ obs, const uint2 *__restrict__ ysave,
                                                                           const bf16x8 *__restrict__ dasave, float *__restrict__ wpart,
                                                                           int B, int H, int W, int per_pair)
{
    constexpr int KS = (NT + 1) / 2, NTH = (NT + 1) / 2;       // key-pair blocks; tiles each wave of a pair carries into the map
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 15, g = lane >> 4;
    const int l = blockIdx.y;
    const int sub = __builtin_amdgcn_readfirstlane(wave >> 1), ni = __builtin_amdgcn_readfirstlane(wave & 1);
    Geom G;
    G.H = H, G.W = W, G.WP = W + 2, G.HW = H * W, G.MP = map_positions(NT, W + 2);
    char *mapB = smem + (size_t)sub * G.MP * 64;
    for (int i = threadIdx.x; i < 2 * G.MP * 4; i += 256) reinterpret_cast<uint4 *>(smem)[i] = uint4{0, 0, 0, 0};
    const int tr_row = (lane & 15) >> 2, tr_pc = lane & 3;
    f32x4 accw[18];                                             // [output half mo][tap]
#pragma unroll
    for (int i = 0; i < 18; ++i) accw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int pair = blockIdx.x * 2 + sub;
    const int s_begin = pair * per_pair;
    uint2 xin[NTH][2];
    auto fetch = [&](int s) {
#pragma unroll
        for (int tl = 0; tl < NTH; ++tl)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int t = ni * NTH + tl;
                xin[tl][m] = uint2{0u, 0u};
                if (t < NT) xin[tl][m] = ysave[dump_index((size_t)(l - 1) * B + s, NT, t, m, lane)];
            }
    };
    if (l > 0 && s_begin < B) fetch(s_begin);
    for (int i = 0; i < per_pair; ++i) {
        const int s = s_begin + i;
        const bool live = s < B;                                // wave-uniform; the two pairs of a block may differ
        int WPv = G.WP, pq = p + GUARD;
        asm volatile("" : "+s"(WPv));
        asm volatile("" : "+v"(pq));
        __syncthreads();                                        // the previous sample's map reads are done (and the clear, first time)
        if (live) {
            if (l == 0) {
                load_obs_block<IN_T>(obs + (size_t)s * 8 * G.HW, mapB, G, ni * 64 + lane, 128);
            } else {
#pragma unroll
                for (int tl = 0; tl < NTH; ++tl) {
                    const int t = ni * NTH + tl;
                    if (t < NT) {
#pragma unroll
                        for (int m = 0; m < 2; ++m)
                            *reinterpret_cast<uint2 *>(mapB + pq * 64 + g * 8 + (WPv + 16 * t) * 64 + m * 32) = xin[tl][m];
                    }
                }
            }
        }
        __syncthreads();
        if (l > 0 && i + 1 < per_pair && s + 1 < B) fetch(s + 1);
        if (live) {
            const bf16x8 *asrc = dasave + (((size_t)l * B + s) * KS * 2) * 64 + lane;
            bf16x8 An0 = asrc[0], An1 = asrc[64];
#pragma unroll 2
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 Ad0 = An0, Ad1 = An1;
                if (ks + 1 < KS) { An0 = asrc[((ks + 1) * 2 + 0) * 64]; An1 = asrc[((ks + 1) * 2 + 1) * 64]; }
                const char *tbase = mapB + (8 * g + tr_row + GUARD) * 64 + tr_pc * 8 + (WPv + 32 * ks) * 64 + ni * 32;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const char *tx = tbase + ((ky - 1) * WPv + (kx - 1)) * 64;
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3))) *)(tx));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3))) *)(tx + 256));
                        const bf16x8 Bx = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        accw[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ad0, Bx, accw[ky * 3 + kx], 0, 0, 0);
                        accw[9 + ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ad1, Bx, accw[9 + ky * 3 + kx], 0, 0, 0);
                    }
            }
        }
    }
    // the pair's partial row: tile (mo * 2 + ni) * 9 + tap, 64 lanes x 4 floats each (plain stores; every pair writes its row in
    // full, zeros included, so the row sum needs no memset)
    float *dst = wpart + ((size_t)pair * NLAYER * 36 + (size_t)l * 36) * 256 + (size_t)lane * 4;
#pragma unroll
    for (int mo = 0; mo < 2; ++mo)
#pragma unroll
        for (int k = 0; k < 9; ++k) *reinterpret_cast<f32x4 *>(dst + (size_t)((mo * 2 + ni) * 9 + k) * 256) = accw[mo * 9 + k];
}

// ---------------------------------------------------------------------------------------------------------------
// Parameter packing: fp32 [cout][cin][3][3] weights -> bf16 MFMA A fragments (forward and input-gradient order)
// ---------------------------------------------------------------------------------------------------------------
struct PackArgs {
    const float *w[NLAYER], *b[NLAYER], *gw[NLAYER], *gb[NLAYER];
    int cin[NLAYER], cout[NLAYER];
};

__global__ __launch_bounds__(256) void pmx_actor_pack_kernel(PackArgs a, char *__restrict__ pack)
{
    const int l = blockIdx.y;
    short *fw = reinterpret_cast<short *>(pack + PACK_FWD) + (size_t)l * FRAG_PER_LAYER;
    short *bw = reinterpret_cast<short *>(pack + PACK_BWD) + (size_t)l * FRAG_PER_LAYER;
    const int cin = a.cin[l], cout = a.cout[l];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < FRAG_PER_LAYER; i += gridDim.x * 256) {
        const int j = i & 7, lane = (i >> 3) & 63, k = (i >> 9) % 9, m = i / (512 * 9);
        const int row = 16 * m + (lane & 15), kk = 8 * (lane >> 4) + j;       // A[row][kk] of tap k
        // forward: row = output channel, kk = input channel
        float vf = 0.f, vb = 0.f;
        if (row < cout && kk < cin) vf = a.w[l][((size_t)row * cin + kk) * 9 + k];
        // input gradient: row = input channel, kk = output channel, tap flipped
        if (row < cin && kk < cout) vb = a.w[l][((size_t)kk * cin + row) * 9 + (8 - k)];
        fw[i] = (short)(pack2(vf, 0.f) & 0xFFFF);
        bw[i] = (short)(pack2(vb, 0.f) & 0xFFFF);
    }
    if (blockIdx.x == 0 && threadIdx.x < 32) {
        const int c = threadIdx.x;
        reinterpret_cast<float *>(pack + PACK_BIAS)[l * 32 + c] = c < cout ? a.b[l][c] : 0.f;
        reinterpret_cast<float *>(pack + PACK_GNW)[l * 32 + c] = a.gw[l] ? a.gw[l][c] : 1.f;
        reinterpret_cast<float *>(pack + PACK_GNB)[l * 32 + c] = a.gb[l] ? a.gb[l][c] : 0.f;
    }
}

struct UnpackArgs {
    float *w[NLAYER], *b[NLAYER], *gw[NLAYER], *gb[NLAYER];
    int cin[NLAYER], cout[NLAYER];
};

// gradient tiles (D layout: row = co = 4g + r, col = ci = p) -> the parameters' own shapes
__global__ __launch_bounds__(256) void pmx_actor_unpack_kernel(UnpackArgs a, const float *__restrict__ grad)
{
    const int l = blockIdx.y;
    const int cin = a.cin[l], cout = a.cout[l];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 36 * 256; i += gridDim.x * 256) {
        const int r = i & 3, lane = (i >> 2) & 63, tile = i >> 8;
        const int k = tile % 9, ni = (tile / 9) & 1, mo = tile / 18;
        const int co = 16 * mo + 4 * (lane >> 4) + r, ci = 16 * ni + (lane & 15);
        if (co < cout && ci < cin) a.w[l][((size_t)co * cin + ci) * 9 + k] = grad[GRAD_W + (size_t)l * 36 * 256 + i];
    }
    if (blockIdx.x == 0 && threadIdx.x < 32) {
        const int c = threadIdx.x;
        if (c < cout) a.b[l][c] = grad[GRAD_B + l * 32 + c];
        if (a.gw[l]) a.gw[l][c] = grad[GRAD_GNW + l * 32 + c];
        if (a.gb[l]) a.gb[l][c] = grad[GRAD_GNB + l * 32 + c];
    }
}

int tiles_for(int H, int W) { return (H * (W + 2) + 15) / 16; }

}   // namespace

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
static const int kCin[NLAYER] = {8, 16, 32, 32, 32, 32, 32, 32};
static const int kCout[NLAYER] = {16, 32, 32, 32, 32, 32, 32, 32};

// position tiles with a kernel instantiation: 10 / 11 (tinyCapture 7 x 20, smallCapture 11 x 14: one wave per sample, or 2 / 4
// waves per sample for small batches) and 28 (the 20 x 20 boards -- bloxCapture, the generated mazes: always four waves per sample)
static bool tiles_supported(int nt) { return nt == 10 || nt == 11 || nt == 28; }
constexpr bool large_board(int nt) { return nt > 16; }

extern "C" int pmx_actor_supported(int32_t H, int32_t W)
{
    if (H < 1 || W < 1 || H > PMX_MAX_DIM || W > PMX_MAX_DIM) return 0;
    return tiles_supported(tiles_for(H, W)) ? 1 : 0;
}

extern "C" int pmx_actor_sizes(int32_t H, int32_t W, int64_t B, int64_t *save_bytes, int64_t *scratch_bytes, int64_t *infer_scratch_bytes)
{
    if (!pmx_actor_supported(H, W) || B < 0) return PMX_ERR_UNSUPPORTED;
    const int64_t dump = (int64_t)tiles_for(H, W) * 1024;            // one P-layout dump of one sample
    if (save_bytes) *save_bytes = B * (8 * dump + 8 * dump + 8 * 4 * 2 * 4);
    const int64_t infer = 2048 * dump;                               // inference: one skip-input slot per resident sample
    if (infer_scratch_bytes) *infer_scratch_bytes = infer;
    // backward: dH operand fragments of the 8 layers + skip slots + the two kernels' partial rows
    if (scratch_bytes)
        *scratch_bytes = B * 8 * (int64_t)((tiles_for(H, W) + 1) / 2) * 2048 + infer + 1024 * 768 * 4 + (int64_t)W_PART_ROWS * NLAYER * 36 * 256 * 4;
    return PMX_OK;
}

extern "C" int pmx_actor_pack(const pmx_actor_params *p, void *pack_dev, void *stream)
{
    if (!p || !pack_dev) return PMX_ERR_INVALID;
    PackArgs a;
    for (int l = 0; l < NLAYER; ++l) {
        a.w[l] = p->conv_w[l], a.b[l] = p->conv_b[l];
        a.gw[l] = l >= 2 ? p->gn_w[l - 2] : nullptr, a.gb[l] = l >= 2 ? p->gn_b[l - 2] : nullptr;
        a.cin[l] = kCin[l], a.cout[l] = kCout[l];
        if (!a.w[l] || !a.b[l] || (l >= 2 && (!a.gw[l] || !a.gb[l]))) return PMX_ERR_INVALID;
    }
    hipLaunchKernelGGL(pmx_actor_pack_kernel, dim3(9, NLAYER), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a,
                       reinterpret_cast<char *>(pack_dev));
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

extern "C" int pmx_actor_unpack_grads(const float *grad_dev, const pmx_actor_params *out, void *stream)
{
    if (!grad_dev || !out) return PMX_ERR_INVALID;
    UnpackArgs a;
    for (int l = 0; l < NLAYER; ++l) {
        a.w[l] = const_cast<float *>(out->conv_w[l]), a.b[l] = const_cast<float *>(out->conv_b[l]);
        a.gw[l] = l >= 2 ? const_cast<float *>(out->gn_w[l - 2]) : nullptr;
        a.gb[l] = l >= 2 ? const_cast<float *>(out->gn_b[l - 2]) : nullptr;
        a.cin[l] = kCin[l], a.cout[l] = kCout[l];
        if (!a.w[l] || !a.b[l] || (l >= 2 && (!a.gw[l] || !a.gb[l]))) return PMX_ERR_INVALID;
    }
    hipLaunchKernelGGL(pmx_actor_unpack_kernel, dim3(9, NLAYER), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, grad_dev);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

namespace {

template <typename K> int allow_lds(K kernel, size_t lds)
{
    // the attribute belongs to ONE kernel on the CURRENT device: latched per (kernel address, device).  (A static flag inside this
    // template is shared by every instantiation with the same function-pointer TYPE -- e.g. the <10> and <11> data kernels.)
    if (lds <= 65536) return PMX_OK;
    struct Key { const void *fn; int dev; };
    static Key done[256];
    static int n_done = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return PMX_ERR_HIP;
    const void *fn = reinterpret_cast<const void *>(kernel);
    for (int i = 0; i < n_done; ++i)
        if (done[i].fn == fn && done[i].dev == dev) return PMX_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return PMX_ERR_HIP;
    if (n_done < 256) done[n_done++] = Key{fn, dev};               // (past 256 entries the attribute is simply set again)
    return PMX_OK;
}

int grid_for(int64_t B, int blocks_per_cu)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const int64_t want = (B + 3) / 4;
    const int64_t cap = (int64_t)cus * blocks_per_cu;
    return (int)(want < cap ? want : cap);
}

// batches up to this size take the four-waves-per-sample kernels (PMX_ACTOR_SPLIT_MAX overrides, 0 = never; read once)
int64_t split_max_batch()
{
    static const int64_t v = [] { const char *e = getenv("PMX_ACTOR_SPLIT_MAX"); return e ? (int64_t)atoll(e) : (int64_t)1536; }();
    return v;
}

int64_t split_bwd_max_batch()      // the data-gradient kernel splits samples over waves up to this batch (PMX_ACTOR_SPLIT_BWD_MAX, read once)
{
    static const int64_t v = [] { const char *e = getenv("PMX_ACTOR_SPLIT_BWD_MAX"); return e ? (int64_t)atoll(e) : (int64_t)512; }();
    return v;
}

// waves per sample of the small-batch forward kernel: four while every sample's block is resident at once (two blocks of ~190
// registers per CU: 512 samples), two beyond (measured at 1 024 samples: one wave per sample 112 us, two 85 us, four 103 us -- at
// four the blocks no longer fit in one round).  PMX_ACTOR_SPLIT_WAVES = 2 / 4 forces one (read once).
int split_waves(int64_t B)
{
    static const int forced = [] { const char *e = getenv("PMX_ACTOR_SPLIT_WAVES"); return e ? atoi(e) : 0; }();
    if (forced == 2 || forced == 4) return forced;
    return B <= 512 ? 4 : 2;
}

template <int NT, typename IN_T>
int launch_fwd(const void *obs, const void *pack, void *feat, void *save, void *scratch, int64_t B, int H, int W, hipStream_t st)
{
    const int64_t dump = (int64_t)NT * 128;                                 // uint2 elements of one dump
    uint2 *hs = reinterpret_cast<uint2 *>(save), *ys = hs ? hs + 8 * B * dump : nullptr;
    float *stt = hs ? reinterpret_cast<float *>(ys + 8 * B * dump) : nullptr;
    uint2 *rtmp = reinterpret_cast<uint2 *>(scratch);
    if (!save && !rtmp) return PMX_ERR_INVALID;
    if (large_board(NT) || B <= split_max_batch()) {
        // several waves per sample (pmx_actor_fwd_split_kernel): small batches, and every batch of a large board (its
        // pre-activations do not fit one wave's registers); 2 048 skip slots exist in the scratch area
        const int ws = large_board(NT) ? 4 : split_waves(B);
        const int spb = 4 / ws;
        const size_t lds_s = (size_t)spb * map_positions(NT, W + 2) * 64 + (size_t)spb * ws * 8 * sizeof(float);
        int64_t g64 = (B + spb - 1) / spb;
        if (g64 * spb > 2048) g64 = 2048 / spb;
        const unsigned grid = (unsigned)g64;
#define PMX_FWD_SPLIT(SAVEV, WSV)                                                                                           \
    do {                                                                                                                    \
        int rc = allow_lds(pmx_actor_fwd_split_kernel<NT, IN_T, SAVEV, WSV>, lds_s);                                        \
        if (rc) return rc;                                                                                                  \
        hipLaunchKernelGGL((pmx_actor_fwd_split_kernel<NT, IN_T, SAVEV, WSV>), dim3(grid), dim3(256), lds_s, st, (const IN_T *)obs, \
                           (const char *)pack, (uint2 *)feat, hs, ys, stt, rtmp, (int)B, H, W, 1e-5f);                       \
    } while (0)
        if constexpr (large_board(NT)) {
            if (save) PMX_FWD_SPLIT(true, 4); else PMX_FWD_SPLIT(false, 4);
        } else {
            if (save) { if (ws == 2) PMX_FWD_SPLIT(true, 2); else PMX_FWD_SPLIT(true, 4); }
            else { if (ws == 2) PMX_FWD_SPLIT(false, 2); else PMX_FWD_SPLIT(false, 4); }
        }
#undef PMX_FWD_SPLIT
        return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
    }
    if constexpr (!large_board(NT)) {
        const size_t lds = (size_t)4 * map_positions(NT, W + 2) * 64;
        if (save) {
            int rc = allow_lds(pmx_actor_fwd_kernel<NT, IN_T, true>, lds);
            if (rc) return rc;
            hipLaunchKernelGGL((pmx_actor_fwd_kernel<NT, IN_T, true>), dim3(grid_for(B, 2)), dim3(256), lds, st, (const IN_T *)obs,
                               (const char *)pack, (uint2 *)feat, hs, ys, stt, rtmp, (int)B, H, W, 1e-5f);
        } else {
            int rc = allow_lds(pmx_actor_fwd_kernel<NT, IN_T, false>, lds);
            if (rc) return rc;
            hipLaunchKernelGGL((pmx_actor_fwd_kernel<NT, IN_T, false>), dim3(grid_for(B, 2)), dim3(256), lds, st, (const IN_T *)obs,
                               (const char *)pack, (uint2 *)feat, hs, ys, stt, rtmp, (int)B, H, W, 1e-5f);
        }
    }
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

template <int NT, typename IN_T>
int launch_bwd(const void *obs, const void *pack, const void *save, const void *dfeat, void *scratch, float *grad, int64_t B,
               int H, int W, hipStream_t st)
{
    const int mp = map_positions(NT, W + 2);
    constexpr int KS = (NT + 1) / 2;
    const int64_t dump = (int64_t)NT * 128;
    const uint2 *hs = reinterpret_cast<const uint2 *>(save), *ys = hs + 8 * B * dump;
    const float *stt = reinterpret_cast<const float *>(ys + 8 * B * dump);
    bf16x8 *da = reinterpret_cast<bf16x8 *>(scratch);                       // [8][B][KS * 2][64] operand fragments of dH
    uint2 *sk = reinterpret_cast<uint2 *>(da + (size_t)8 * B * KS * 2 * 64);  // then one skip-gradient slot per resident sample
    float *accpart = reinterpret_cast<float *>(sk + (size_t)2048 * dump);       // then the data kernel's per-block partial sums
    int rc;
    int grid_d = grid_for(B, 2);
    // from this batch on, two waves per sample at two waves per SIMD (PMX_ACTOR_BWD_TWO_WAVE_MIN, read once; 0 = never)
    static const int64_t two_wave_min = [] { const char *e = getenv("PMX_ACTOR_BWD_TWO_WAVE_MIN"); return e ? (int64_t)atoll(e) : (int64_t)4096; }();
    const bool two_wave = two_wave_min > 0 && B >= two_wave_min;
    if (large_board(NT) || two_wave || (B <= split_max_batch() && B <= split_bwd_max_batch())) {
        // several waves per sample (pmx_actor_bwd_data_split_kernel): every batch of a large board; small batches otherwise
        // (four waves: 256 samples 56 us, 512 samples 69 us against 118 us for one wave per sample) and LARGE ones (two waves per
        // sample: the one-wave kernel holds a sample's 44 + 44 registers of gradient and pre-activation and runs one wave per
        // SIMD; halves of a sample fit two waves per SIMD, and the second wave fills the first one's stalls: backward of 16 384
        // samples 2.29 -> 2.19 ms, of 8 192 1.21 -> 1.14 ms; between 1 024 and 2 048 samples the one-wave kernel is still ahead)
        const int ws = large_board(NT) ? 4 : (two_wave ? 2 : split_waves(B)), spb = 4 / ws;
        const size_t lds_s = (size_t)spb * mp * 64 + (size_t)4 * NLAYER * 96 * 4 + (size_t)spb * ws * 8 * 4;
        int64_t g64 = (B + spb - 1) / spb;
        if (g64 > 1024) g64 = 1024;                          // accpart has 1 024 rows; the skip slots (2 048) cover grid * spb
        grid_d = (int)g64;
        if (!large_board(NT) && ws == 2) {
            if constexpr (!large_board(NT)) {
                rc = allow_lds(pmx_actor_bwd_data_split_kernel<NT, 2>, lds_s);
                if (rc) return rc;
                hipLaunchKernelGGL((pmx_actor_bwd_data_split_kernel<NT, 2>), dim3(grid_d), dim3(256), lds_s, st, (const char *)pack,
                                   (const uint2 *)dfeat, hs, ys, stt, da, sk, accpart, (int)B, H, W);
            }
        } else {
            rc = allow_lds(pmx_actor_bwd_data_split_kernel<NT, 4>, lds_s);
            if (rc) return rc;
            hipLaunchKernelGGL((pmx_actor_bwd_data_split_kernel<NT, 4>), dim3(grid_d), dim3(256), lds_s, st, (const char *)pack,
                               (const uint2 *)dfeat, hs, ys, stt, da, sk, accpart, (int)B, H, W);
        }
    } else {
        if constexpr (!large_board(NT)) {
            const size_t lds_d = (size_t)4 * (mp * 64 + NLAYER * 96 * 4);
            rc = allow_lds(pmx_actor_bwd_data_kernel<NT>, lds_d);
            if (rc) return rc;
            hipLaunchKernelGGL((pmx_actor_bwd_data_kernel<NT>), dim3(grid_d), dim3(256), lds_d, st, (const char *)pack, (const uint2 *)dfeat,
                               hs, ys, stt, da, sk, accpart, (int)B, H, W);
        }
    }
    hipLaunchKernelGGL(pmx_actor_sum_acc_kernel, dim3(NLAYER * 96 / 32), dim3(256), 0, st, (const float *)accpart, grid_d, grad);
    if (hipGetLastError() != hipSuccess) return PMX_ERR_HIP;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    float *wpart = accpart + (size_t)1024 * 768;                             // then the weight kernel's partial rows
    if constexpr (large_board(NT)) {
        // weight gradient, large boards: layers x blocks of two sample PAIRS-of-waves; a partial row per pair (<= W_PART_ROWS rows),
        // about two blocks per CU in total
        int64_t pairs = (2 * (int64_t)cus / NLAYER) * 2;
        if (pairs > W_PART_ROWS) pairs = W_PART_ROWS;
        int64_t per_pair = (B + pairs - 1) / pairs;
        if (per_pair < 2) per_pair = 2;
        pairs = (B + per_pair - 1) / per_pair;
        const int64_t blocks = (pairs + 1) / 2;                              // an odd pair count leaves one idle pair: it writes a row of zeros
        const size_t lds_w = (size_t)2 * mp * 64;
        rc = allow_lds(pmx_actor_bwd_weight_split_kernel<NT, IN_T>, lds_w);
        if (rc) return rc;
        hipLaunchKernelGGL((pmx_actor_bwd_weight_split_kernel<NT, IN_T>), dim3((unsigned)blocks, NLAYER), dim3(256), lds_w, st, (const IN_T *)obs,
                           ys, (const bf16x8 *)da, wpart, (int)B, H, W, (int)per_pair);
        hipLaunchKernelGGL(pmx_actor_sum_w_kernel, dim3(NLAYER * 36 * 256 / 32), dim3(256), 0, st, (const float *)wpart, (int)(blocks * 2), grad);
    } else {
        // weight gradient: layers x sample chunks; about two blocks per CU in total, each wave at least a few samples
        const size_t lds_w = (size_t)4 * mp * 64;
        if (lds_w < (size_t)4 * 9 * 64 * 16) return PMX_ERR_UNSUPPORTED;           // the reduction scratch must fit in the maps
        int64_t chunks = (2 * cus) / NLAYER;
        if (chunks > W_PART_ROWS) chunks = W_PART_ROWS;
        int64_t per_wave = (B + chunks * 4 - 1) / (chunks * 4);
        if (per_wave < 2) per_wave = 2;
        chunks = (B + per_wave * 4 - 1) / (per_wave * 4);
        rc = allow_lds(pmx_actor_bwd_weight_kernel<NT, IN_T>, lds_w);
        if (rc) return rc;
        hipLaunchKernelGGL((pmx_actor_bwd_weight_kernel<NT, IN_T>), dim3((unsigned)chunks, NLAYER), dim3(256), lds_w, st, (const IN_T *)obs, ys,
                           (const bf16x8 *)da, wpart, (int)B, H, W, (int)per_wave);
        hipLaunchKernelGGL(pmx_actor_sum_w_kernel, dim3(NLAYER * 36 * 256 / 32), dim3(256), 0, st, (const float *)wpart, (int)chunks, grad);
    }
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

}   // namespace

extern "C" int pmx_actor_forward(const void *obs_dev, int32_t obs_dtype, const void *pack_dev, void *feat_dev, void *save_dev,
                                 void *scratch_dev, int64_t B, int32_t H, int32_t W, void *stream)
{
    if (B == 0) return pmx_actor_supported(H, W) ? PMX_OK : PMX_ERR_UNSUPPORTED;      // nothing to do (pointers may be null)
    if (!obs_dev || !pack_dev || !feat_dev || B < 0) return PMX_ERR_INVALID;
    if (!pmx_actor_supported(H, W)) return PMX_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nt = tiles_for(H, W);
#ifdef PMX_ACTOR_EXP      /* compile-time experiments: one instantiation only */
#define PMX_FWD(NT) return launch_fwd<11, __hip_bfloat16>(obs_dev, pack_dev, feat_dev, save_dev, scratch_dev, B, H, W, st);
#else
#define PMX_FWD(NT)                                                                                                      \
    switch (obs_dtype) {                                                                                                 \
    case PMX_OBS_F32: return launch_fwd<NT, float>(obs_dev, pack_dev, feat_dev, save_dev, scratch_dev, B, H, W, st);                  \
    case PMX_OBS_BF16: return launch_fwd<NT, __hip_bfloat16>(obs_dev, pack_dev, feat_dev, save_dev, scratch_dev, B, H, W, st);        \
    case PMX_OBS_U8: return launch_fwd<NT, uint8_t>(obs_dev, pack_dev, feat_dev, save_dev, scratch_dev, B, H, W, st);                 \
    default: return PMX_ERR_INVALID;                                                                                     \
    }
#endif
    if (nt == 10) { PMX_FWD(10) }
    if (nt == 28) { PMX_FWD(28) }
    PMX_FWD(11)
#undef PMX_FWD
}

extern "C" int pmx_actor_backward(const void *obs_dev, int32_t obs_dtype, const void *pack_dev, const void *save_dev,
                                  const void *dfeat_dev, void *scratch_dev, float *grad_dev, int64_t B, int32_t H, int32_t W,
                                  void *stream)
{
    if (!obs_dev || !pack_dev || !save_dev || !dfeat_dev || !scratch_dev || !grad_dev || B < 0) return PMX_ERR_INVALID;
    if (!pmx_actor_supported(H, W)) return PMX_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (B == 0) return hipMemsetAsync(grad_dev, 0, sizeof(float) * GRAD_FLOATS, st) == hipSuccess ? PMX_OK : PMX_ERR_HIP;
    // (no memset otherwise: the two row-sum kernels write every word of the gradient)
    const int nt = tiles_for(H, W);
#ifdef PMX_ACTOR_EXP
#define PMX_BWD(NT) return launch_bwd<11, __hip_bfloat16>(obs_dev, pack_dev, save_dev, dfeat_dev, scratch_dev, grad_dev, B, H, W, st);
#else
#define PMX_BWD(NT)                                                                                                                  \
    switch (obs_dtype) {                                                                                                             \
    case PMX_OBS_F32: return launch_bwd<NT, float>(obs_dev, pack_dev, save_dev, dfeat_dev, scratch_dev, grad_dev, B, H, W, st);      \
    case PMX_OBS_BF16: return launch_bwd<NT, __hip_bfloat16>(obs_dev, pack_dev, save_dev, dfeat_dev, scratch_dev, grad_dev, B, H, W, st); \
    case PMX_OBS_U8: return launch_bwd<NT, uint8_t>(obs_dev, pack_dev, save_dev, dfeat_dev, scratch_dev, grad_dev, B, H, W, st);     \
    default: return PMX_ERR_INVALID;                                                                                                 \
    }
#endif
    if (nt == 10) { PMX_BWD(10) }
    if (nt == 28) { PMX_BWD(28) }
    PMX_BWD(11)
#undef PMX_BWD
}

// ---------------------------------------------------------------------------------------------------------------
// The critic's projector (pacman_mappo_resnet.py:126-127 critic_projector + :69-95 / :164 the positional encoding): conv3x3(8 -> 32)
// + bias + position table -> the token tensor [B][H*W][32] bf16 the batch-major encoder reads -- straight from the observation
// planes (bytes as they are), with the map machinery of the tower above.  Only 8 input channels exist, so the K = 32 of the matrix
// instruction is filled with the THREE taps of a kernel row: lane group g < 3 reads the 8 channels of the position shifted by
// g - 1 (one ds_read_b128 per lane, the address is the lane's own) and the A operand holds w[co][ci][ky][kx = g] in k-slots
// 8 g + ci -- 6 matrix instructions per 16 positions instead of 18.  The weight gradient contracts over positions: A = dTok^T
// and B = the shifted input through transposing reads, TWO taps per 16-column tile (columns 8 s + ci, each half of the reading
// lanes pointing at its own shift): 5 tap pairs x 2 output halves = 10 instructions per 32 positions.  The input needs no gradient.
// ---------------------------------------------------------------------------------------------------------------
namespace {

constexpr int PROJ_FRAG = 2 * 3 * 64 * 8;                           // bf16 elements of the forward fragments: [m][ky][lane][8]
constexpr size_t PROJ_PACK_BIAS = (size_t)PROJ_FRAG * 2;
constexpr size_t PROJ_PACK_BYTES_C = PROJ_PACK_BIAS + 32 * 4;
static_assert(PROJ_PACK_BYTES_C == PMX_PROJ_PACK_BYTES, "include/pmx.h and pmx_actor.hip disagree on the projector pack size");
constexpr int PROJ_GRAD_TILES = 10, PROJ_ROW = PROJ_GRAD_TILES * 256 + 32;      // [pair 5][mo 2] tiles of 64 lanes x 4, then db[32]
static_assert(PROJ_ROW == PMX_PROJ_GRAD_ROW_FLOATS, "include/pmx.h and pmx_actor.hip disagree on the projector gradient row");

__global__ __launch_bounds__(256) void pmx_proj_pack_kernel(const float *__restrict__ w, const float *__restrict__ b, char *__restrict__ pack)
{
    short *fw = reinterpret_cast<short *>(pack);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < PROJ_FRAG; i += gridDim.x * 256) {
        const int j = i & 7, lane = (i >> 3) & 63, ky = (i >> 9) % 3, m = i / (512 * 3);
        const int co = 16 * m + (lane & 15), kx = lane >> 4;                    // k-slot 8 kx + j = (tap kx of the row, input channel j)
        const float v = kx < 3 ? w[((size_t)co * 8 + j) * 9 + ky * 3 + kx] : 0.f;
        fw[i] = (short)(pack2(v, 0.f) & 0xFFFF);
    }
    if (blockIdx.x == 0 && threadIdx.x < 32) reinterpret_cast<float *>(pack + PROJ_PACK_BIAS)[threadIdx.x] = b[threadIdx.x];
}

template <int NT, typename IN_T, int WPB>
__global__ __launch_bounds__(64 * WPB) void pmx_proj_fwd_kernel(const IN_T *__restrict__ obs, const char *__restrict__ pack,
                                                                const float *__restrict__ pe, uint2 *__restrict__ tok, int B, int H, int W)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 15, g = lane >> 4;
    Geom G;
    G.H = H, G.W = W, G.WP = W + 2, G.HW = H * W, G.MP = map_positions(NT, W + 2);
    char *map = smem + (size_t)wave * G.MP * 64;
    for (int i = lane; i < G.MP * 4; i += 64) reinterpret_cast<uint4 *>(map)[i] = uint4{0, 0, 0, 0};
    __shared__ int cell_tab[NT * 16];
    for (int i = threadIdx.x; i < NT * 16; i += 64 * WPB) {
        const int q = G.WP + i, row = q / G.WP, col = q - row * G.WP;
        cell_tab[i] = (col >= 1 && col <= W && row <= H) ? (row - 1) * W + (col - 1) : -1;
    }
    __syncthreads();
    bf16x8 A[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) A[m][ky] = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const short *>(pack) + ((m * 3 + ky) * 64 + lane) * 8);
    float bias[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[m][r] = reinterpret_cast<const float *>(pack + PROJ_PACK_BIAS)[16 * m + 4 * g + r];
    // lane group g reads the position shifted by g - 1 (group 3: any finite operand, its k-slots meet zero weights)
    const char *rbase = map + (p + GUARD + (g < 3 ? g - 1 : 0)) * 64;
    for (int s = blockIdx.x * WPB + wave; s < B; s += gridDim.x * WPB) {
        wave_lds_fence();
        load_obs<IN_T>(obs + (size_t)s * 8 * G.HW, map, G, lane);
        wave_lds_fence();
#pragma unroll 2
        for (int t = 0; t < NT; ++t) {
            f32x4 a[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const bf16x8 b = *reinterpret_cast<const bf16x8 *>(rbase + (G.WP + 16 * t + (ky - 1) * G.WP) * 64);
                a[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0][ky], b, a[0], 0, 0, 0);
                a[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1][ky], b, a[1], 0, 0, 0);
            }
            const int ci = cell_tab[16 * t + p];
            if (ci >= 0) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const float4 e = *reinterpret_cast<const float4 *>(pe + (size_t)ci * 32 + 16 * m + 4 * g);
                    // what bf16 autocast computes: the convolution (+ bias) rounded to bf16, then the bf16 sum with the bf16 table
                    const uint2 c2 = {pack2(a[m][0] + bias[m][0], a[m][1] + bias[m][1]), pack2(a[m][2] + bias[m][2], a[m][3] + bias[m][3])};
                    const uint2 e2 = {pack2(e.x, e.y), pack2(e.z, e.w)};
                    const uint2 y2 = {pack2(lo_f(c2.x) + lo_f(e2.x), hi_f(c2.x) + hi_f(e2.x)), pack2(lo_f(c2.y) + lo_f(e2.y), hi_f(c2.y) + hi_f(e2.y))};
                    tok[((size_t)s * G.HW + ci) * 8 + 4 * m + g] = y2;
                }
            }
        }
    }
}

// weight / bias gradient: a block walks its samples one at a time -- all waves copy the sample's dTok rows and planes into the two
// maps, wave w then multiplies the key-pair blocks ks = w, w + 4, ..; 10 accumulator tiles per wave over the block's samples
template <int NT, typename IN_T>
__global__ __launch_bounds__(256, 2) void pmx_proj_wgrad_kernel(const IN_T *__restrict__ obs, const uint4 *__restrict__ dtok, float *__restrict__ part,
                                                                int B, int H, int W, int per_block)
{
    constexpr int KS = (NT + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned short pos_tab[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4;
    Geom G;
    G.H = H, G.W = W, G.WP = W + 2, G.HW = H * W, G.MP = map_positions(NT, W + 2);
    char *mapX = smem, *mapD = smem + (size_t)G.MP * 64;
    for (int i = threadIdx.x; i < 2 * G.MP * 4; i += 256) reinterpret_cast<uint4 *>(smem)[i] = uint4{0, 0, 0, 0};
    for (int i = threadIdx.x; i < G.HW; i += 256) {
        const int row = i / W, col = i - row * W;
        pos_tab[i] = (unsigned short)((row + 1) * G.WP + col + 1 + GUARD);
    }
    __syncthreads();
    f32x4 acc[PROJ_GRAD_TILES];
#pragma unroll
    for (int i = 0; i < PROJ_GRAD_TILES; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float db[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};        // channels 8 (tid & 3) .. + 7, summed over the rows this thread copies
    const int tr_row = (lane & 15) >> 2, tr_pc = lane & 3;
    const int s0 = blockIdx.x * per_block, s1 = s0 + per_block < B ? s0 + per_block : B;
    for (int s = s0; s < s1; ++s) {
        __syncthreads();                                            // the previous sample's reads are done
        load_obs_block<IN_T>(obs + (size_t)s * 8 * G.HW, mapX, G, threadIdx.x, 256);
        for (int i = threadIdx.x; i < G.HW * 4; i += 256) {
            const uint4 u = dtok[(size_t)s * G.HW * 4 + i];
            *reinterpret_cast<uint4 *>(mapD + (size_t)pos_tab[i >> 2] * 64 + (i & 3) * 16) = u;
            db[0] += lo_f(u.x), db[1] += hi_f(u.x), db[2] += lo_f(u.y), db[3] += hi_f(u.y);
            db[4] += lo_f(u.z), db[5] += hi_f(u.z), db[6] += lo_f(u.w), db[7] += hi_f(u.w);
        }
        __syncthreads();
        for (int ks = wave; ks < KS; ks += 4) {
            const int q0 = G.WP + 32 * ks + 8 * g + tr_row + GUARD;             // the lane's row of a transposing read
            bf16x8 Ad[2];
#pragma unroll
            for (int mo = 0; mo < 2; ++mo) {
                const char *a0 = mapD + q0 * 64 + mo * 32 + tr_pc * 8;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3))) *)(a0));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3))) *)(a0 + 256));
                Ad[mo] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int pr = 0; pr < 5; ++pr) {
                // columns 0..7 = tap 2 pr (read by the lanes with tr_pc < 2), 8..15 = tap 2 pr + 1 (tr_pc >= 2; the tenth tap does
                // not exist: those lanes re-read tap 8, its columns are dropped when the gradient is unpacked)
                const int tap = 2 * pr + (tr_pc >> 1) < 9 ? 2 * pr + (tr_pc >> 1) : 8;
                const int sh = (tap / 3 - 1) * G.WP + (tap % 3 - 1);
                const char *b0 = mapX + (q0 + sh) * 64 + (tr_pc & 1) * 8;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3))) *)(b0));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3))) *)(b0 + 256));
                const bf16x8 Bx = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                acc[pr * 2 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ad[0], Bx, acc[pr * 2 + 0], 0, 0, 0);
                acc[pr * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ad[1], Bx, acc[pr * 2 + 1], 0, 0, 0);
            }
        }
    }
    // block sum of the four waves' tiles and of the 256 threads' bias sums -> the block's partial row
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem);                  // 4 x 10 x 256 floats = 40 KB (the launch reserves at least that)
#pragma unroll
    for (int i = 0; i < PROJ_GRAD_TILES; ++i) *reinterpret_cast<f32x4 *>(red + ((size_t)(wave * PROJ_GRAD_TILES + i) * 64 + lane) * 4) = acc[i];
    __syncthreads();
    float *row = part + (size_t)blockIdx.x * PROJ_ROW;
    for (int i = threadIdx.x; i < PROJ_GRAD_TILES * 256; i += 256)
        row[i] = (red[i] + red[PROJ_GRAD_TILES * 256 + i]) + (red[2 * PROJ_GRAD_TILES * 256 + i] + red[3 * PROJ_GRAD_TILES * 256 + i]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) red[threadIdx.x * 8 + i] = db[i];
    __syncthreads();
    if (threadIdx.x < 32) {
        const int chunk = threadIdx.x >> 3, i = threadIdx.x & 7;
        float t = 0.f;
        for (int k = chunk; k < 256; k += 4) t += red[k * 8 + i];
        row[PROJ_GRAD_TILES * 256 + threadIdx.x] = t;
    }
}

// rows -> dW [32][8][3][3] and db [32] in the parameters' own layout (the row sum and the unpacking in one launch): 32 outputs x 8
// row slices per block, slices added through LDS (one thread per output walking all the rows alone is a chain of load round trips)
__global__ __launch_bounds__(256) void pmx_proj_sum_kernel(const float *__restrict__ part, int n_rows, float *__restrict__ dw, float *__restrict__ dbias)
{
    __shared__ float red[8][33];
    const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + c;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < 32 * 72 + 32) {
        size_t src;
        if (i < 32 * 72) {
            const int co = i / 72, rem = i - co * 72, ci = rem / 9, tap = rem - ci * 9;
            const int pr = tap >> 1, n = (tap & 1) * 8 + ci, mo = co >> 4, gg = (co & 15) >> 2, r = co & 3;
            src = (size_t)(pr * 2 + mo) * 256 + (size_t)(gg * 16 + n) * 4 + r;
        } else {
            src = (size_t)PROJ_GRAD_TILES * 256 + (i - 32 * 72);
        }
        int k = sl;
        for (; k + 24 < n_rows; k += 32) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += part[(size_t)(k + 8 * u) * PROJ_ROW + src];
        }
        for (; k < n_rows; k += 8) acc[0] += part[(size_t)k * PROJ_ROW + src];
    }
    red[sl][c] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (sl == 0 && i < 32 * 72 + 32) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][c];
        if (i < 32 * 72) dw[i] = t; else dbias[i - 32 * 72] = t;
    }
}

template <int NT, typename IN_T>
int launch_proj_fwd(const void *obs, const void *pack, const float *pe, void *tok, int64_t B, int H, int W, hipStream_t st)
{
    constexpr int WPB = large_board(NT) ? 2 : 4;
    const size_t lds = (size_t)WPB * map_positions(NT, W + 2) * 64;
    int rc = allow_lds(pmx_proj_fwd_kernel<NT, IN_T, WPB>, lds);
    if (rc) return rc;
    int64_t want = (B + WPB - 1) / WPB, cap = 256 * 8;
    hipLaunchKernelGGL((pmx_proj_fwd_kernel<NT, IN_T, WPB>), dim3((unsigned)(want < cap ? want : cap)), dim3(64 * WPB), lds, st, (const IN_T *)obs,
                       (const char *)pack, pe, (uint2 *)tok, (int)B, H, W);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

template <int NT, typename IN_T>
int launch_proj_bwd(const void *obs, const void *dtok, float *part, float *dw, float *db, int64_t B, int H, int W, hipStream_t st)
{
    const size_t maps = (size_t)2 * map_positions(NT, W + 2) * 64, scratch = (size_t)4 * PROJ_GRAD_TILES * 256 * 4;
    const size_t lds = maps > scratch ? maps : scratch;            // the block-sum scratch reuses the maps' space (40 KB at least)
    int rc = allow_lds(pmx_proj_wgrad_kernel<NT, IN_T>, lds);
    if (rc) return rc;
    int64_t blocks = B < PMX_PROJ_PARTIAL_ROWS ? B : PMX_PROJ_PARTIAL_ROWS;
    const int64_t per_block = (B + blocks - 1) / blocks;
    blocks = (B + per_block - 1) / per_block;
    hipLaunchKernelGGL((pmx_proj_wgrad_kernel<NT, IN_T>), dim3((unsigned)blocks), dim3(256), lds, st, (const IN_T *)obs, (const uint4 *)dtok, part,
                       (int)B, H, W, (int)per_block);
    hipLaunchKernelGGL(pmx_proj_sum_kernel, dim3((32 * 72 + 32 + 31) / 32), dim3(256), 0, st, (const float *)part, (int)blocks, dw, db);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

}   // namespace

extern "C" int pmx_proj_pack(const float *w, const float *b, void *pack_dev, void *stream)
{
    if (!w || !b || !pack_dev) return PMX_ERR_INVALID;
    hipLaunchKernelGGL(pmx_proj_pack_kernel, dim3(6), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, b, reinterpret_cast<char *>(pack_dev));
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

extern "C" int pmx_proj_forward(const void *obs_dev, int32_t obs_dtype, const void *pack_dev, const float *posenc_dev, void *tokens_dev, int64_t B,
                                int32_t H, int32_t W, void *stream)
{
    if (B == 0) return pmx_actor_supported(H, W) ? PMX_OK : PMX_ERR_UNSUPPORTED;
    if (!obs_dev || !pack_dev || !posenc_dev || !tokens_dev || B < 0) return PMX_ERR_INVALID;
    if (!pmx_actor_supported(H, W)) return PMX_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nt = tiles_for(H, W);
#define PMX_PF(NT)                                                                                                        \
    switch (obs_dtype) {                                                                                                  \
    case PMX_OBS_F32: return launch_proj_fwd<NT, float>(obs_dev, pack_dev, posenc_dev, tokens_dev, B, H, W, st);              \
    case PMX_OBS_BF16: return launch_proj_fwd<NT, __hip_bfloat16>(obs_dev, pack_dev, posenc_dev, tokens_dev, B, H, W, st);    \
    case PMX_OBS_U8: return launch_proj_fwd<NT, uint8_t>(obs_dev, pack_dev, posenc_dev, tokens_dev, B, H, W, st);             \
    default: return PMX_ERR_INVALID;                                                                                      \
    }
    if (nt == 10) { PMX_PF(10) }
    if (nt == 28) { PMX_PF(28) }
    PMX_PF(11)
#undef PMX_PF
}

extern "C" int pmx_proj_backward(const void *obs_dev, int32_t obs_dtype, const void *dtokens_dev, float *partial_dev, float *dw_dev, float *db_dev,
                                 int64_t B, int32_t H, int32_t W, void *stream)
{
    if (!dw_dev || !db_dev || B < 0) return PMX_ERR_INVALID;
    if (!pmx_actor_supported(H, W)) return PMX_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (B == 0) {
        if (hipMemsetAsync(dw_dev, 0, sizeof(float) * 32 * 72, st) != hipSuccess || hipMemsetAsync(db_dev, 0, sizeof(float) * 32, st) != hipSuccess) return PMX_ERR_HIP;
        return PMX_OK;
    }
    if (!obs_dev || !dtokens_dev || !partial_dev) return PMX_ERR_INVALID;
    const int nt = tiles_for(H, W);
#define PMX_PB(NT)                                                                                                                  \
    switch (obs_dtype) {                                                                                                            \
    case PMX_OBS_F32: return launch_proj_bwd<NT, float>(obs_dev, dtokens_dev, partial_dev, dw_dev, db_dev, B, H, W, st);                \
    case PMX_OBS_BF16: return launch_proj_bwd<NT, __hip_bfloat16>(obs_dev, dtokens_dev, partial_dev, dw_dev, db_dev, B, H, W, st);      \
    case PMX_OBS_U8: return launch_proj_bwd<NT, uint8_t>(obs_dev, dtokens_dev, partial_dev, dw_dev, db_dev, B, H, W, st);               \
    default: return PMX_ERR_INVALID;                                                                                                \
    }
    if (nt == 10) { PMX_PB(10) }
    if (nt == 28) { PMX_PB(28) }
    PMX_PB(11)
#undef PMX_PB
}

#ifdef PMX_ACTOR_TIMING
extern "C" int pmx_actor_ticks_read(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pmx_actor_ticks), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(pmx_actor_ticks), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
