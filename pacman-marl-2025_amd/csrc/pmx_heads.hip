// pmx_heads.hip -- the small ends of MAPPOAgent's two heads as one kernel each way (pacman_mappo_resnet.py:115-123 actor_head,
// :143-147 critic_head with the mean pool of :169), replacing a chain of ~25 library launches per optimizer step that moved a few
// hundred kilobytes each (LayerNorm, GELU, two skinny GEMMs, a GEMV, the mean over tokens, their backward kernels, bias reductions
// and dtype casts) -- a third of the launches of the launch-bound 512-sample step.
//
//   actor tail   logits[b] = W2 . gelu(LayerNorm_512(h[b])) + b2          h = the 512-wide output of the first head layer (hipBLASLt)
//   critic tail  value[b]  = w2 . gelu(W1 . mean_s tokens[b][s] + b1) + b2 tokens = the encoder output [B][S][32]
//
// One wavefront per sample, a lane owns 8 of the 512 hidden features; sums over the features are wave reductions, sums over the
// batch (the parameter gradients) are per-wave register accumulators -> one partial row per block -> a row-sum kernel.  Arithmetic
// is float32 on the values bf16 autocast would feed the library kernels (activations and weights of the two linears rounded to
// bf16, LayerNorm and GELU in float32); GELU is the exact erf form (nn.GELU() default, erff).
#pragma clang fp contract(fast)
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "../../include/pmx.h"

extern thread_local int pmx_defer_sums_flag;       // pmx_critic.hip: deferred row sums (pmx_defer_row_sums)
extern thread_local int pmx_last_rows_value;

namespace {

constexpr int HID = 512, PER_LANE = 8, NACT = 5, DM = 32;
static_assert(HID == 64 * PER_LANE, "a lane owns 8 hidden features");

__device__ __forceinline__ float bf_round(float x)
{
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    const f2 f = {x, 0.f};
    const uint32_t u = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, b2));
    return __uint_as_float(u << 16);
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xFFFF0000u); }
__device__ __forceinline__ uint32_t bf_pack(float a, float b)
{
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    const f2 f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, b2));
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float gelu_exact(float z) { return 0.5f * z * (1.0f + erff(z * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float z)
{
    return 0.5f * (1.0f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * __expf(-0.5f * z * z);
}

// the 8 features of a lane from a row of 512 bf16 (16 bytes per lane) or float32 (two 16-byte loads)
template <typename T> __device__ __forceinline__ void load8(const T *row, int lane, float (&v)[8]);
template <> __device__ __forceinline__ void load8<__hip_bfloat16>(const __hip_bfloat16 *row, int lane, float (&v)[8])
{
    const uint4 u = reinterpret_cast<const uint4 *>(row)[lane];
    v[0] = bf_lo(u.x), v[1] = bf_hi(u.x), v[2] = bf_lo(u.y), v[3] = bf_hi(u.y), v[4] = bf_lo(u.z), v[5] = bf_hi(u.z), v[6] = bf_lo(u.w), v[7] = bf_hi(u.w);
}
template <> __device__ __forceinline__ void load8<float>(const float *row, int lane, float (&v)[8])
{
    const float4 a = reinterpret_cast<const float4 *>(row)[2 * lane], b = reinterpret_cast<const float4 *>(row)[2 * lane + 1];
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
}
template <typename T> __device__ __forceinline__ void store8(T *row, int lane, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<__hip_bfloat16>(__hip_bfloat16 *row, int lane, const float (&v)[8])
{
    reinterpret_cast<uint4 *>(row)[lane] = uint4{bf_pack(v[0], v[1]), bf_pack(v[2], v[3]), bf_pack(v[4], v[5]), bf_pack(v[6], v[7])};
}
template <> __device__ __forceinline__ void store8<float>(float *row, int lane, const float (&v)[8])
{
    reinterpret_cast<float4 *>(row)[2 * lane] = float4{v[0], v[1], v[2], v[3]};
    reinterpret_cast<float4 *>(row)[2 * lane + 1] = float4{v[4], v[5], v[6], v[7]};
}

// ---------------------------------------------------------------------------------------------------------------
// Actor tail
// ---------------------------------------------------------------------------------------------------------------
constexpr int AT_DW2 = 0, AT_DB2 = NACT * HID, AT_DLNW = AT_DB2 + 8, AT_DLNB = AT_DLNW + HID, AT_FLOATS = AT_DLNB + HID;
static_assert(AT_FLOATS == PMX_ACTOR_TAIL_GRAD_FLOATS, "include/pmx.h and pmx_heads.hip disagree on the actor-tail gradient size");

template <typename T>
__global__ __launch_bounds__(256) void pmx_actor_tail_fwd_kernel(const T *__restrict__ h, const float *__restrict__ lnw, const float *__restrict__ lnb,
                                                                const float *__restrict__ w2, const float *__restrict__ b2,
                                                                float *__restrict__ logits, float *__restrict__ stats, int B, float eps)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float gw[8], gb[8], w[NACT][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        gw[i] = lnw[lane * 8 + i], gb[i] = lnb[lane * 8 + i];
#pragma unroll
        for (int o = 0; o < NACT; ++o) w[o][i] = bf_round(w2[o * HID + lane * 8 + i]);
    }
    for (int b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
        float x[8];
        load8<T>(h + (size_t)b * HID, lane, x);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += x[i];
        const float mean = wave_sum(s) * (1.0f / HID);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) q += (x[i] - mean) * (x[i] - mean);
        const float rstd = rsqrtf(wave_sum(q) * (1.0f / HID) + eps);
        float acc[NACT] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float g = bf_round(gelu_exact((x[i] - mean) * rstd * gw[i] + gb[i]));
#pragma unroll
            for (int o = 0; o < NACT; ++o) acc[o] = fmaf(g, w[o][i], acc[o]);
        }
#pragma unroll
        for (int o = 0; o < NACT; ++o) acc[o] = wave_sum(acc[o]);
        if (lane < NACT) {
            float v = acc[0];
#pragma unroll
            for (int o = 1; o < NACT; ++o) v = lane == o ? acc[o] : v;
            logits[(size_t)b * NACT + lane] = v + b2[lane];
        }
        if (lane == 0 && stats) stats[2 * (size_t)b] = mean, stats[2 * (size_t)b + 1] = rstd;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pmx_actor_tail_bwd_kernel(const T *__restrict__ h, const float *__restrict__ stats, const float *__restrict__ dlogits,
                                                                const float *__restrict__ lnw, const float *__restrict__ lnb,
                                                                const float *__restrict__ w2, T *__restrict__ dh, float *__restrict__ grad, int B)
{
    __shared__ float red[4][AT_FLOATS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float gw[8], gb[8], w[NACT][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        gw[i] = lnw[lane * 8 + i], gb[i] = lnb[lane * 8 + i];
#pragma unroll
        for (int o = 0; o < NACT; ++o) w[o][i] = bf_round(w2[o * HID + lane * 8 + i]);
    }
    float dw2[NACT][8], dlw[8], dlb[8], db2[NACT];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        dlw[i] = dlb[i] = 0.f;
#pragma unroll
        for (int o = 0; o < NACT; ++o) dw2[o][i] = 0.f;
    }
#pragma unroll
    for (int o = 0; o < NACT; ++o) db2[o] = 0.f;
    for (int b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
        float x[8], dl[NACT];
        load8<T>(h + (size_t)b * HID, lane, x);
        const float mean = stats[2 * (size_t)b], rstd = stats[2 * (size_t)b + 1];
#pragma unroll
        for (int o = 0; o < NACT; ++o) dl[o] = dlogits[(size_t)b * NACT + o], db2[o] += dl[o];
        float dz[8], xh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            xh[i] = (x[i] - mean) * rstd;
            const float z = xh[i] * gw[i] + gb[i];
            const float g = bf_round(gelu_exact(z));
            float dg = 0.f;
#pragma unroll
            for (int o = 0; o < NACT; ++o) {
                dg = fmaf(dl[o], w[o][i], dg);
                dw2[o][i] = fmaf(dl[o], g, dw2[o][i]);
            }
            dz[i] = dg * gelu_grad(z);                                  // gradient at the LayerNorm output
            dlw[i] = fmaf(dz[i], xh[i], dlw[i]);
            dlb[i] += dz[i];
            const float gd = dz[i] * gw[i];
            s1 += gd, s2 = fmaf(gd, xh[i], s2);
        }
        s1 = wave_sum(s1) * (1.0f / HID), s2 = wave_sum(s2) * (1.0f / HID);
        float out[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) out[i] = rstd * (dz[i] * gw[i] - s1 - xh[i] * s2);
        store8<T>(dh + (size_t)b * HID, lane, out);
    }
    // the block's four waves -> one partial row (row 1 + blockIdx.x; row 0 receives the sum of the rows)
    float *mine = red[wave];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int o = 0; o < NACT; ++o) mine[AT_DW2 + o * HID + lane * 8 + i] = dw2[o][i];
        mine[AT_DLNW + lane * 8 + i] = dlw[i];
        mine[AT_DLNB + lane * 8 + i] = dlb[i];
    }
    if (lane < 8) mine[AT_DB2 + lane] = lane < NACT ? (lane == 0 ? db2[0] : lane == 1 ? db2[1] : lane == 2 ? db2[2] : lane == 3 ? db2[3] : db2[4]) : 0.f;
    __syncthreads();
    float *row = grad + (size_t)(1 + blockIdx.x) * AT_FLOATS;
    for (int i = threadIdx.x; i < AT_FLOATS; i += 256) row[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// out[i] = sum of rows 1 .. n_rows of buf (row 0 receives it): 32 columns x 8 row slices per block
__global__ __launch_bounds__(256) void pmx_heads_sum_rows_kernel(float *__restrict__ buf, int n_rows, int floats)
{
    __shared__ float part[8][33];
    const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + c;
    float acc = 0.f;
    if (i < floats)
        for (int r = 1 + sl; r <= n_rows; r += 8) acc += buf[(size_t)r * floats + i];
    part[sl][c] = acc;
    __syncthreads();
    if (sl == 0 && i < floats) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][c];
        buf[i] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Critic tail
// ---------------------------------------------------------------------------------------------------------------
constexpr int CT_DW1 = 0, CT_DB1 = HID * DM, CT_DW2 = CT_DB1 + HID, CT_DB2 = CT_DW2 + HID, CT_FLOATS = CT_DB2 + 8;
static_assert(CT_FLOATS == PMX_CRITIC_TAIL_GRAD_FLOATS, "include/pmx.h and pmx_heads.hip disagree on the critic-tail gradient size");

// mean over the S tokens of one sample: tokens row-major [S][32] bf16; lane (t = lane >> 2, chunk = lane & 3) walks tokens t, t + 16, ..
// and owns features 8 chunk .. + 7; returns the lane's 8 partial sums reduced over the 16 lanes with the same chunk
__device__ __forceinline__ void pool8(const __hip_bfloat16 *tok, int S, int lane, float (&p)[8])
{
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = 0.f;
    const uint4 *rows = reinterpret_cast<const uint4 *>(tok);
    for (int t = lane >> 2; t < S; t += 16) {
        const uint4 u = rows[(size_t)t * 4 + (lane & 3)];
        p[0] += bf_lo(u.x), p[1] += bf_hi(u.x), p[2] += bf_lo(u.y), p[3] += bf_hi(u.y);
        p[4] += bf_lo(u.z), p[5] += bf_hi(u.z), p[6] += bf_lo(u.w), p[7] += bf_hi(u.w);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float v = p[i];
        v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        p[i] = v;
    }
}

// w1t: the first layer's weight transposed and rounded to bf16, as float32 [32][512] in LDS (a lane reads 8 consecutive hidden units
// of one input feature: two conflict-free 16-byte reads).  A thread carries whole rows of w1 (128 contiguous bytes, 8 loads of 16) and
// writes them down a column: for a fixed k consecutive threads write consecutive words (the first version walked w1 linearly and
// wrote with a stride of 512 words -- every store of a wave on ONE bank: 27 us for a kernel with 10 us of work).
__device__ __forceinline__ void stage_w1t(const float *__restrict__ w1, float *w1t)
{
    for (int j = threadIdx.x; j < HID; j += 256) {
        float4 r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = reinterpret_cast<const float4 *>(w1 + (size_t)j * DM)[q];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            w1t[(4 * q + 0) * HID + j] = bf_round(r[q].x);
            w1t[(4 * q + 1) * HID + j] = bf_round(r[q].y);
            w1t[(4 * q + 2) * HID + j] = bf_round(r[q].z);
            w1t[(4 * q + 3) * HID + j] = bf_round(r[q].w);
        }
    }
}

__global__ __launch_bounds__(256) void pmx_critic_tail_fwd_kernel(const __hip_bfloat16 *__restrict__ tok, const float *__restrict__ w1,
                                                                 const float *__restrict__ b1, const float *__restrict__ w2,
                                                                 const float *__restrict__ b2, float *__restrict__ value,
                                                                 float *__restrict__ pooled, int B, int S)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float *w1t = smem_f, *pl = smem_f + HID * DM;                      // [32][512], then [4 waves][32]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    stage_w1t(w1, w1t);
    float bb[8], ww[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bb[i] = bf_round(b1[lane * 8 + i]), ww[i] = bf_round(w2[lane * 8 + i]);
    __syncthreads();
    const float inv_s = 1.0f / (float)S;
    for (int b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
        float p[8];
        pool8(tok + (size_t)b * S * DM, S, lane, p);
        if (lane < 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float m = bf_round(p[i] * inv_s);                // the pooled token, as the bf16 linear sees it
                pl[wave * DM + lane * 8 + i] = m;
                pooled[(size_t)b * DM + lane * 8 + i] = m;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float pre[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) pre[i] = bb[i];
#pragma unroll 4
        for (int k = 0; k < DM; ++k) {
            const float pk = pl[wave * DM + k];
            const float4 a = *reinterpret_cast<const float4 *>(w1t + k * HID + lane * 8), c = *reinterpret_cast<const float4 *>(w1t + k * HID + lane * 8 + 4);
            pre[0] = fmaf(a.x, pk, pre[0]), pre[1] = fmaf(a.y, pk, pre[1]), pre[2] = fmaf(a.z, pk, pre[2]), pre[3] = fmaf(a.w, pk, pre[3]);
            pre[4] = fmaf(c.x, pk, pre[4]), pre[5] = fmaf(c.y, pk, pre[5]), pre[6] = fmaf(c.z, pk, pre[6]), pre[7] = fmaf(c.w, pk, pre[7]);
        }
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) v = fmaf(bf_round(gelu_exact(bf_round(pre[i]))), ww[i], v);
        v = wave_sum(v);
        if (lane == 0) value[b] = v + b2[0];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// per sample: dh = dv w2 gelu'(pre) (bf16, for the weight-gradient kernel), g = gelu(pre) (bf16), dtokens = W1^T dh / S on every token
__global__ __launch_bounds__(256) void pmx_critic_tail_bwd_kernel(const float *__restrict__ pooled, const float *__restrict__ dvalue,
                                                                 const float *__restrict__ w1, const float *__restrict__ b1,
                                                                 const float *__restrict__ w2, __hip_bfloat16 *__restrict__ dtok,
                                                                 __hip_bfloat16 *__restrict__ dh_out, __hip_bfloat16 *__restrict__ g_out, int B, int S)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float *w1t = smem_f, *pl = smem_f + HID * DM, *dp = pl + 4 * DM;   // [32][512], [4][32] pooled, [4][64][33] partial input gradients
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    stage_w1t(w1, w1t);
    float bb[8], ww[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bb[i] = bf_round(b1[lane * 8 + i]), ww[i] = bf_round(w2[lane * 8 + i]);
    __syncthreads();
    const float inv_s = 1.0f / (float)S;
    float *dpw = dp + (size_t)wave * 64 * 33;
    for (int b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
        if (lane < DM) pl[wave * DM + lane] = pooled[(size_t)b * DM + lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float dv = dvalue[b];
        float pre[8], dpl[DM];
#pragma unroll
        for (int i = 0; i < 8; ++i) pre[i] = bb[i];
#pragma unroll 4
        for (int k = 0; k < DM; ++k) {
            const float pk = pl[wave * DM + k];
            const float4 a = *reinterpret_cast<const float4 *>(w1t + k * HID + lane * 8), c = *reinterpret_cast<const float4 *>(w1t + k * HID + lane * 8 + 4);
            pre[0] = fmaf(a.x, pk, pre[0]), pre[1] = fmaf(a.y, pk, pre[1]), pre[2] = fmaf(a.z, pk, pre[2]), pre[3] = fmaf(a.w, pk, pre[3]);
            pre[4] = fmaf(c.x, pk, pre[4]), pre[5] = fmaf(c.y, pk, pre[5]), pre[6] = fmaf(c.z, pk, pre[6]), pre[7] = fmaf(c.w, pk, pre[7]);
        }
        float dh[8], gg[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float z = bf_round(pre[i]);
            gg[i] = gelu_exact(z);
            dh[i] = bf_round(dv * ww[i] * gelu_grad(z));
        }
        store8<__hip_bfloat16>(dh_out + (size_t)b * HID, lane, dh);
        store8<__hip_bfloat16>(g_out + (size_t)b * HID, lane, gg);
        // the lane's share of dpooled[k] = sum_j dh[j] W1[j][k], then the sum over the 64 lanes through LDS
#pragma unroll 4
        for (int k = 0; k < DM; ++k) {
            const float4 a = *reinterpret_cast<const float4 *>(w1t + k * HID + lane * 8), c = *reinterpret_cast<const float4 *>(w1t + k * HID + lane * 8 + 4);
            dpl[k] = (dh[0] * a.x + dh[1] * a.y + dh[2] * a.z + dh[3] * a.w) + (dh[4] * c.x + dh[5] * c.y + dh[6] * c.z + dh[7] * c.w);
        }
#pragma unroll
        for (int k = 0; k < DM; ++k) dpw[lane * 33 + k] = dpl[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float tot = 0.f;                                                // lane k (and k + 32) sums column k & 31
#pragma unroll 8
        for (int r = 0; r < 64; ++r) tot += dpw[r * 33 + (lane & 31)];
        tot *= inv_s;
        // every token of the sample receives dpooled / S: lane (t = lane >> 2, chunk = lane & 3) writes 8 features of tokens t, t + 16, ..
        float o8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o8[i] = __shfl(tot, (lane & 3) * 8 + i, 64);
        const uint4 w = uint4{bf_pack(o8[0], o8[1]), bf_pack(o8[2], o8[3]), bf_pack(o8[4], o8[5]), bf_pack(o8[6], o8[7])};
        uint4 *rows = reinterpret_cast<uint4 *>(dtok + (size_t)b * S * DM);
        for (int t = lane >> 2; t < S; t += 16) rows[(size_t)t * 4 + (lane & 3)] = w;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// parameter gradients of the critic tail from dh, g [B][512] bf16, pooled [B][32], dvalue [B]: block (x, y) owns 8 hidden units and a
// chunk of the batch, thread (unit, k) walks the chunk: dW1[j][k] = sum_b dh[b][j] pooled[b][k]; the k = 0 / 1 threads also carry
// db1[j] = sum_b dh[b][j] and dw2[j] = sum_b dv[b] g[b][j]; blocks x = 0 add db2 = sum_b dv[b].  One chunk: the sums go straight to
// row 0; several: to partial rows 1 + y, and the row-sum kernel follows.  (A thread's loop is a chain of load round trips, so the
// chunks are short -- 16 samples up to 2 048, B / 128 beyond -- and the blocks many: one 256-sample chunk took 80 us.)
__global__ __launch_bounds__(256) void pmx_critic_tail_wgrad_kernel(const __hip_bfloat16 *__restrict__ dh, const __hip_bfloat16 *__restrict__ g,
                                                                   const float *__restrict__ pooled, const float *__restrict__ dvalue,
                                                                   float *__restrict__ grad, int B, int chunk, int single)
{
    const int k = threadIdx.x & 31, j = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int b0 = blockIdx.y * chunk, b1 = b0 + chunk < B ? b0 + chunk : B;
    float *row = grad + (single ? 0 : (size_t)(1 + blockIdx.y) * CT_FLOATS);
    const unsigned short *dhs = reinterpret_cast<const unsigned short *>(dh), *gs = reinterpret_cast<const unsigned short *>(g);
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, sb = 0.f, sw = 0.f;
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float d = __uint_as_float((uint32_t)dhs[(size_t)(b + u) * HID + j] << 16);
            acc[u] = fmaf(d, pooled[(size_t)(b + u) * DM + k], acc[u]);
            if (k == 0) sb += d;
            if (k == 1) sw = fmaf(dvalue[b + u], __uint_as_float((uint32_t)gs[(size_t)(b + u) * HID + j] << 16), sw);
        }
    }
    for (; b < b1; ++b) {
        const float d = __uint_as_float((uint32_t)dhs[(size_t)b * HID + j] << 16);
        acc[0] = fmaf(d, pooled[(size_t)b * DM + k], acc[0]);
        if (k == 0) sb += d;
        if (k == 1) sw = fmaf(dvalue[b], __uint_as_float((uint32_t)gs[(size_t)b * HID + j] << 16), sw);
    }
    row[CT_DW1 + j * DM + k] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    if (k == 0) row[CT_DB1 + j] = sb;
    if (k == 1) row[CT_DW2 + j] = sw;
    if (blockIdx.x == 0 && threadIdx.x >= 192) {                      // one wave: db2 of the chunk (+ the 7 unused floats, zero)
        const int lane = threadIdx.x - 192;
        float s = 0.f;
        for (int i = b0 + lane; i < b1; i += 64) s += dvalue[i];
        s = wave_sum(s);
        if (lane < 8) row[CT_DB2 + lane] = lane == 0 ? s : 0.f;
    }
}

int heads_blocks(int64_t B, int cap)
{
    const int64_t want = (B + 3) / 4;
    return (int)(want < cap ? (want < 1 ? 1 : want) : cap);
}

}   // namespace

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
extern "C" int pmx_actor_tail_forward(const void *h_dev, int32_t h_bf16, const float *ln_w, const float *ln_b, const float *w2, const float *b2,
                                      float *logits_dev, float *stats_dev, int64_t B, float eps, void *stream)
{
    if (B == 0) return PMX_OK;
    if (!h_dev || !ln_w || !ln_b || !w2 || !b2 || !logits_dev || B < 0) return PMX_ERR_INVALID;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int blocks = heads_blocks(B, 1024);
    if (h_bf16) hipLaunchKernelGGL(pmx_actor_tail_fwd_kernel<__hip_bfloat16>, dim3(blocks), dim3(256), 0, st, (const __hip_bfloat16 *)h_dev, ln_w, ln_b, w2, b2, logits_dev, stats_dev, (int)B, eps);
    else hipLaunchKernelGGL(pmx_actor_tail_fwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float *)h_dev, ln_w, ln_b, w2, b2, logits_dev, stats_dev, (int)B, eps);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

extern "C" int pmx_actor_tail_backward(const void *h_dev, int32_t h_bf16, const float *stats_dev, const float *dlogits_dev, const float *ln_w,
                                       const float *ln_b, const float *w2, void *dh_dev, float *grad_dev, int64_t B, void *stream)
{
    if (!grad_dev || B < 0) return PMX_ERR_INVALID;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    pmx_last_rows_value = 0;
    if (B == 0) return hipMemsetAsync(grad_dev, 0, sizeof(float) * AT_FLOATS, st) == hipSuccess ? PMX_OK : PMX_ERR_HIP;
    if (!h_dev || !stats_dev || !dlogits_dev || !ln_w || !ln_b || !w2 || !dh_dev) return PMX_ERR_INVALID;
    const int blocks = heads_blocks(B, PMX_HEADS_PARTIAL_ROWS);
    if (h_bf16) hipLaunchKernelGGL(pmx_actor_tail_bwd_kernel<__hip_bfloat16>, dim3(blocks), dim3(256), 0, st, (const __hip_bfloat16 *)h_dev, stats_dev, dlogits_dev, ln_w, ln_b, w2, (__hip_bfloat16 *)dh_dev, grad_dev, (int)B);
    else hipLaunchKernelGGL(pmx_actor_tail_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float *)h_dev, stats_dev, dlogits_dev, ln_w, ln_b, w2, (float *)dh_dev, grad_dev, (int)B);
    pmx_last_rows_value = blocks;
    if (!pmx_defer_sums_flag)
        hipLaunchKernelGGL(pmx_heads_sum_rows_kernel, dim3((AT_FLOATS + 31) / 32), dim3(256), 0, st, grad_dev, blocks, (int)AT_FLOATS);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

namespace {
int allow_big_lds(const void *fn, size_t lds)
{
    if (lds <= 65536) return PMX_OK;
    struct Key { const void *fn; int dev; };
    static Key done[16];
    static int n_done = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return PMX_ERR_HIP;
    for (int i = 0; i < n_done; ++i)
        if (done[i].fn == fn && done[i].dev == dev) return PMX_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return PMX_ERR_HIP;
    if (n_done < 16) done[n_done++] = Key{fn, dev};
    return PMX_OK;
}
}   // namespace

extern "C" int pmx_critic_tail_forward(const void *tokens_dev, const float *w1, const float *b1, const float *w2, const float *b2,
                                       float *value_dev, float *pooled_dev, int64_t B, int32_t S, void *stream)
{
    if (B == 0) return PMX_OK;
    if (!tokens_dev || !w1 || !b1 || !w2 || !b2 || !value_dev || !pooled_dev || B < 0 || S < 1) return PMX_ERR_INVALID;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t lds = sizeof(float) * (HID * DM + 4 * DM);
    int rc = allow_big_lds(reinterpret_cast<const void *>(pmx_critic_tail_fwd_kernel), lds);
    if (rc) return rc;
    hipLaunchKernelGGL(pmx_critic_tail_fwd_kernel, dim3(heads_blocks(B, 512)), dim3(256), lds, st, (const __hip_bfloat16 *)tokens_dev, w1, b1, w2, b2,
                       value_dev, pooled_dev, (int)B, (int)S);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

extern "C" int pmx_critic_tail_backward(const float *pooled_dev, const float *dvalue_dev, const float *w1, const float *b1, const float *w2,
                                        void *dtokens_dev, void *scratch_dev, float *grad_dev, int64_t B, int32_t S, void *stream)
{
    if (!grad_dev || B < 0) return PMX_ERR_INVALID;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    pmx_last_rows_value = 0;
    if (B == 0) return hipMemsetAsync(grad_dev, 0, sizeof(float) * CT_FLOATS, st) == hipSuccess ? PMX_OK : PMX_ERR_HIP;
    if (!pooled_dev || !dvalue_dev || !w1 || !b1 || !w2 || !dtokens_dev || !scratch_dev || S < 1) return PMX_ERR_INVALID;
    __hip_bfloat16 *dh = reinterpret_cast<__hip_bfloat16 *>(scratch_dev), *g = dh + (size_t)B * HID;       // scratch: 2 x B x 512 bf16
    const size_t lds = sizeof(float) * (HID * DM + 4 * DM + 4 * 64 * 33);
    int rc = allow_big_lds(reinterpret_cast<const void *>(pmx_critic_tail_bwd_kernel), lds);
    if (rc) return rc;
    hipLaunchKernelGGL(pmx_critic_tail_bwd_kernel, dim3(heads_blocks(B, 512)), dim3(256), lds, st, pooled_dev, dvalue_dev, w1, b1, w2,
                       (__hip_bfloat16 *)dtokens_dev, dh, g, (int)B, (int)S);
    // a thread's walk over its chunk is a chain of load round trips (~0.3 us per 4 samples at these sizes): short chunks, many blocks
    int64_t chunk = 16;
    if ((B + chunk - 1) / chunk > PMX_HEADS_PARTIAL_ROWS) chunk = (B + PMX_HEADS_PARTIAL_ROWS - 1) / PMX_HEADS_PARTIAL_ROWS;
    const int n_chunks = (int)((B + chunk - 1) / chunk);
    hipLaunchKernelGGL(pmx_critic_tail_wgrad_kernel, dim3(HID / 8, n_chunks), dim3(256), 0, st, (const __hip_bfloat16 *)dh, (const __hip_bfloat16 *)g,
                       pooled_dev, dvalue_dev, grad_dev, (int)B, (int)chunk, n_chunks == 1 ? 1 : 0);
    pmx_last_rows_value = n_chunks > 1 ? n_chunks : 0;                 // (one chunk: the sums went straight to row 0)
    if (n_chunks > 1 && !pmx_defer_sums_flag)
        hipLaunchKernelGGL(pmx_heads_sum_rows_kernel, dim3((CT_FLOATS + 31) / 32), dim3(256), 0, st, grad_dev, n_chunks, (int)CT_FLOATS);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}
