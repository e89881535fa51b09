// pmx_train.hip -- the non-tick kernels of the path (gfx950): GAE reverse scan, maze distances, observation
// post-processing.  C-ABI entry points for pmx_gae / pmx_canonicalize_obs / pmx_merge_obs live here too.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include <hip/hip_bf16.h>

#include "../../include/pmx.h"
#include "pmx_device.h"

// ---------------------------------------------------------------------------------------------------------------
// GAE (pacman_mappo_resnet.py:277-290), series laid out [T][n].
//   lane-per-series: the vectorised trainer has n = envs x learners >= thousands of independent series and a short
//     horizon, so one lane owns one series and walks it backwards; every load/store is a coalesced row of the
//     [T][n] arrays and the arithmetic is the reference's float32 sequence exactly (bit-exact; contraction is off).
//   wave-per-series: the reference's own regime (n = 2, T = 2048): A_t = delta_t + c_t * A_{t+1} is a composition of
//     affine maps, so a wavefront scans 64 time steps at once with a Kogge-Stone prefix over __shfl_up.  The
//     re-association changes float rounding (documented tolerance 1e-5 relative), so it is used only when there are
//     too few series to fill the chip.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pmx_gae_lane_kernel(const float *__restrict__ rew, const float *__restrict__ val,
                                                           const float *__restrict__ done, const float *__restrict__ last,
                                                           int T, int n, double gamma, double lam, float *__restrict__ adv,
                                                           float *__restrict__ ret)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float g32 = (float)gamma;
    const float gl32 = (float)(gamma * lam);              // Python double product, rounded when it meets the tensor
    float lastgaelam = 0.0f;
    // gamma * last_value is a double product in the reference (both are Python floats), then rounded to float32
    float gv = (float)(gamma * (double)last[i]);
#pragma unroll 4
    for (int t = T - 1; t >= 0; --t) {
        const size_t o = (size_t)t * n + i;
        const float v = val[o];
        const float nonterm = 1.0f - done[o];
        const float delta = rew[o] + gv * nonterm - v;
        lastgaelam = delta + gl32 * nonterm * lastgaelam;
        adv[o] = lastgaelam;
        ret[o] = lastgaelam + v;
        gv = g32 * v;                                      // gamma * values[t] for step t-1
    }
}

__global__ __launch_bounds__(64) void pmx_gae_wave_kernel(const float *__restrict__ rew, const float *__restrict__ val,
                                                          const float *__restrict__ done, const float *__restrict__ last,
                                                          int T, int n, double gamma, double lam, float *__restrict__ adv,
                                                          float *__restrict__ ret)
{
    const int i = blockIdx.x;                              // one wavefront per series
    const int lane = threadIdx.x;
    const float g32 = (float)gamma, gl32 = (float)(gamma * lam);
    float carry = 0.0f;                                    // A_{t+1} entering the chunk
    for (int hi = T - 1; hi >= 0; hi -= 64) {
        const int t = hi - lane;                           // lane 0 = latest time of the chunk
        float a = 0.0f, b = 0.0f, v = 0.0f;
        if (t >= 0) {
            const size_t o = (size_t)t * n + i;
            v = val[o];
            const float nonterm = 1.0f - done[o];
            const float gv = (t == T - 1) ? (float)(gamma * (double)last[i]) : g32 * val[o + n];
            b = rew[o] + gv * nonterm - v;                 // delta_t
            a = gl32 * nonterm;
        }
        // inclusive scan of x -> a*x + b towards higher lanes (earlier times): f_l o f_{l-1} o ... o f_0
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float ao = __shfl_up(a, d), bo = __shfl_up(b, d);
            if (lane >= d) { b = a * bo + b; a = a * ao; }
        }
        const float A = a * carry + b;
        if (t >= 0) {
            const size_t o = (size_t)t * n + i;
            adv[o] = A;
            ret[o] = A + v;
        }
        const int last_lane = hi >= 63 ? 63 : hi;
        carry = __shfl(A, last_lane);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Maze distances (distanceCalculator.py:111-150): unit-cost UCS from every open cell == BFS.  One lane per SOURCE
// cell; the frontier / visited sets are 32 row masks held in registers (loops fully unrolled so every index is a
// compile-time constant), one BFS level is four shifted ORs per row.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void pmx_maze_kernel(const PmxLayoutDev *__restrict__ L, const int16_t *__restrict__ cell_index,
                                                      int n, const int8_t *__restrict__ cells, uint8_t *__restrict__ dist)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int H = L->H;
    uint32_t open[32], vis[32], fr[32];
#pragma unroll
    for (int y = 0; y < 32; ++y) { open[y] = y < H ? ~L->walls[y] & (L->lo_mask | L->hi_mask) : 0u; vis[y] = 0u; fr[y] = 0u; }
    const int sx = cells[2 * s], sy = cells[2 * s + 1];
    for (int t = 0; t < n; ++t) dist[(size_t)t * n + s] = 255;      // sys.maxsize in the reference
#pragma unroll
    for (int y = 0; y < 32; ++y) if (y == sy) { fr[y] = 1u << sx; vis[y] = fr[y]; }
    dist[(size_t)s * n + s] = 0;
    for (int level = 1; level < 255; ++level) {
        uint32_t nx[32];
        uint32_t any = 0;
#pragma unroll
        for (int y = 0; y < 32; ++y) {
            uint32_t m = (fr[y] << 1) | (fr[y] >> 1);
            if (y > 0) m |= fr[y - 1];
            if (y < 31) m |= fr[y + 1];
            m &= open[y] & ~vis[y];
            nx[y] = m;
            any |= m;
        }
        if (!any) break;
#pragma unroll
        for (int y = 0; y < 32; ++y) {
            fr[y] = nx[y];
            vis[y] |= nx[y];
            uint32_t m = nx[y];
            while (m) {
                const int x = __ffs(m) - 1;
                m &= m - 1;
                const int t = cell_index[y * 32 + x];
                dist[(size_t)t * n + s] = (uint8_t)level;
            }
        }
    }
}

extern "C" hipError_t pmx_launch_maze(const PmxLayoutDev *lay_dev, const int16_t *cell_index_dev, int n_cells,
                                      const int8_t *cells_dev, uint8_t *dist_dev, hipStream_t st)
{
    hipLaunchKernelGGL(pmx_maze_kernel, dim3((n_cells + 63) / 64), dim3(64), 0, st, lay_dev, cell_index_dev, n_cells, cells_dev,
                       dist_dev);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Observation post-processing (pacman_mappo_resnet.py:215-229, :267-274) on [n][8][H][W] blocks.  Observation values
// are non-negative (0, 1, 1 + carry), so for all three element types the unsigned bit pattern orders like the value.
// ---------------------------------------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(256) void pmx_canon_kernel(const E *__restrict__ in, E *__restrict__ out, long total, int H, int W)
{
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    const int x = (int)(k % W);
    const long r = k / W;
    const int y = (int)(r % H);
    const long pc = r / H;
    const int c = (int)(pc & 7);
    const long b = pc >> 3;
    const int sc = (c == 2) ? 3 : (c == 3) ? 2 : (c == 6) ? 7 : (c == 7) ? 6 : c;
    out[k] = in[((b * 8 + sc) * H + y) * W + (W - 1 - x)];
}

template <typename E>
__global__ __launch_bounds__(256) void pmx_merge_kernel(const E *__restrict__ a, const E *__restrict__ b, E *__restrict__ out,
                                                        long total, int plane)
{
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    const int c = (int)((k / plane) & 7);
    E v = a[k];
    if (c == 1) { const E w = b[k]; v = w > v ? w : v; }
    if (c == 4) v = 0;
    out[k] = v;
}

extern "C" int pmx_gae(const float *rewards_dev, const float *values_dev, const float *dones_dev, const float *last_value_dev,
                       int32_t T, int32_t n, double gamma, double lam, float *adv_dev, float *ret_dev, void *stream)
{
    if (!rewards_dev || !values_dev || !dones_dev || !last_value_dev || !adv_dev || !ret_dev || T < 1 || n < 1) return PMX_ERR_INVALID;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (n >= 2048 || T < 128)
        hipLaunchKernelGGL(pmx_gae_lane_kernel, dim3((n + 255) / 256), dim3(256), 0, st, rewards_dev, values_dev, dones_dev,
                           last_value_dev, T, n, gamma, lam, adv_dev, ret_dev);
    else
        hipLaunchKernelGGL(pmx_gae_wave_kernel, dim3(n), dim3(64), 0, st, rewards_dev, values_dev, dones_dev, last_value_dev, T, n,
                           gamma, lam, adv_dev, ret_dev);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

// forced variants for tests / benchmarks: mode 0 = lane-per-series, 1 = wave-per-series
extern "C" int pmx_gae_mode(const float *rewards_dev, const float *values_dev, const float *dones_dev, const float *last_value_dev,
                            int32_t T, int32_t n, double gamma, double lam, float *adv_dev, float *ret_dev, int mode, void *stream)
{
    if (T < 1 || n < 1) return PMX_ERR_INVALID;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (mode == 0)
        hipLaunchKernelGGL(pmx_gae_lane_kernel, dim3((n + 255) / 256), dim3(256), 0, st, rewards_dev, values_dev, dones_dev,
                           last_value_dev, T, n, gamma, lam, adv_dev, ret_dev);
    else
        hipLaunchKernelGGL(pmx_gae_wave_kernel, dim3(n), dim3(64), 0, st, rewards_dev, values_dev, dones_dev, last_value_dev, T, n,
                           gamma, lam, adv_dev, ret_dev);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

extern "C" int pmx_canonicalize_obs(const void *in_dev, void *out_dev, int32_t n, int32_t H, int32_t W, int32_t obs_dtype,
                                    void *stream)
{
    if (!in_dev || !out_dev || n < 0 || H < 1 || W < 1) return PMX_ERR_INVALID;
    const long total = (long)n * 8 * H * W;
    if (total == 0) return PMX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    switch (obs_dtype) {
    case PMX_OBS_F32: hipLaunchKernelGGL(pmx_canon_kernel<uint32_t>, grid, block, 0, st, (const uint32_t *)in_dev, (uint32_t *)out_dev, total, H, W); break;
    case PMX_OBS_BF16: hipLaunchKernelGGL(pmx_canon_kernel<uint16_t>, grid, block, 0, st, (const uint16_t *)in_dev, (uint16_t *)out_dev, total, H, W); break;
    case PMX_OBS_U8: hipLaunchKernelGGL(pmx_canon_kernel<uint8_t>, grid, block, 0, st, (const uint8_t *)in_dev, (uint8_t *)out_dev, total, H, W); break;
    default: return PMX_ERR_INVALID;
    }
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

extern "C" int pmx_merge_obs(const void *a_dev, const void *b_dev, void *out_dev, int32_t n, int32_t H, int32_t W, int32_t obs_dtype,
                             void *stream)
{
    if (!a_dev || !b_dev || !out_dev || n < 0 || H < 1 || W < 1) return PMX_ERR_INVALID;
    const long total = (long)n * 8 * H * W;
    if (total == 0) return PMX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    switch (obs_dtype) {
    case PMX_OBS_F32: hipLaunchKernelGGL(pmx_merge_kernel<uint32_t>, grid, block, 0, st, (const uint32_t *)a_dev, (const uint32_t *)b_dev, (uint32_t *)out_dev, total, H * W); break;
    case PMX_OBS_BF16: hipLaunchKernelGGL(pmx_merge_kernel<uint16_t>, grid, block, 0, st, (const uint16_t *)a_dev, (const uint16_t *)b_dev, (uint16_t *)out_dev, total, H * W); break;
    case PMX_OBS_U8: hipLaunchKernelGGL(pmx_merge_kernel<uint8_t>, grid, block, 0, st, (const uint8_t *)a_dev, (const uint8_t *)b_dev, (uint8_t *)out_dev, total, H * W); break;
    default: return PMX_ERR_INVALID;
    }
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}


// ---------------------------------------------------------------------------------------------------------------
// Column sums of a tall bfloat16 matrix [rows][C] (the bias gradients of the critic's token linears: rows = S * B =
// 630 k, C = 32 .. 128).  torch's generic reduction needs ~100 us for the 161 MB case; here consecutive lanes read
// consecutive 16 bytes, every thread keeps 8 float accumulators for its column octet, rows of a block are folded
// through LDS and every block writes one partial row (no atomics; the caller adds the PMX_COLSUM_BLOCKS rows up).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pmx_colsum_bf16_kernel(const __hip_bfloat16 *__restrict__ x, long rows, int C,
                                                              float *__restrict__ partial /*[gridDim.x][C]*/)
{
    extern __shared__ float red[];                       // [RP][C]
    const int vcn = C >> 3;                              // 16-byte columns per row
    const int RP = 256 / vcn;                            // rows per block pass
    const int vc = threadIdx.x % vcn, rsub = threadIdx.x / vcn;
    float acc[8] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    if (rsub < RP) {
        for (long r = (long)blockIdx.x * RP + rsub; r < rows; r += (long)gridDim.x * RP) {
            const uint4 t = *reinterpret_cast<const uint4 *>(x + r * C + vc * 8);
            const uint32_t w[4] = { t.x, t.y, t.z, t.w };
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[2 * j] += __uint_as_float(w[j] << 16); acc[2 * j + 1] += __uint_as_float(w[j] & 0xFFFF0000u); }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) red[rsub * C + vc * 8 + j] = acc[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int r = 0; r < RP; ++r) s += red[r * C + c];
        partial[(size_t)blockIdx.x * C + c] = s;
    }
}

// x_dev [rows][C] bfloat16, C a multiple of 8 and <= 256; partial_dev [PMX_COLSUM_BLOCKS][C] float32 is written in full.
extern "C" int pmx_colsum_bf16(const void *x_dev, int64_t rows, int32_t C, float *partial_dev, void *stream)
{
    if (!x_dev || !partial_dev || rows < 0 || C < 8 || C > 256 || (C & 7)) return PMX_ERR_INVALID;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int RP = 256 / (C >> 3);
    hipLaunchKernelGGL(pmx_colsum_bf16_kernel, dim3(PMX_COLSUM_BLOCKS), dim3(256), (size_t)RP * C * sizeof(float), st,
                       (const __hip_bfloat16 *)x_dev, (long)rows, C, partial_dev);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}


// ---------------------------------------------------------------------------------------------------------------
// Fused residual-add + LayerNorm over a SMALL feature dimension (the critic's d_model = 32,
// pacman_mappo_resnet.py:138-141 post-LN encoder layers): y = LayerNorm(x + a) * w + b.  One LANE per token row, the
// 32 features of the row in registers, 16-byte loads/stores; float32 statistics whatever the IO type.  torch's native
// LayerNorm spends a workgroup per row (1.7 ms for the [154*4096, 32] token matrix); the reduce + elementwise
// formulation needs ~6 kernels forward and ~12 backward.  The backward kernel recomputes x + a, produces the (shared)
// input gradient and accumulates the weight / bias gradients per lane over a grid-stride loop, reduces them across
// the wavefront with shuffles and adds 64 floats per wave to the global accumulators.
// ---------------------------------------------------------------------------------------------------------------
template <typename T> struct LnIO;
template <> struct LnIO<float> {
    static __device__ __forceinline__ void load(const float *p, float *v) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { float4 t = reinterpret_cast<const float4 *>(p)[k]; v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w; }
    }
    static __device__ __forceinline__ void store(float *p, const float *v) {
#pragma unroll
        for (int k = 0; k < 8; ++k) reinterpret_cast<float4 *>(p)[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    }
};
template <> struct LnIO<__hip_bfloat16> {
    static __device__ __forceinline__ void load(const __hip_bfloat16 *p, float *v) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint4 t = reinterpret_cast<const uint4 *>(p)[k];
            const uint32_t w[4] = { t.x, t.y, t.z, t.w };
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[8 * k + 2 * j] = __uint_as_float(w[j] << 16); v[8 * k + 2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u); }
        }
    }
    static __device__ __forceinline__ void store(__hip_bfloat16 *p, const float *v) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const __hip_bfloat16 lo = __float2bfloat16(v[8 * k + 2 * j]), hi = __float2bfloat16(v[8 * k + 2 * j + 1]);
                w[j] = (uint32_t)(*reinterpret_cast<const uint16_t *>(&lo)) | ((uint32_t)(*reinterpret_cast<const uint16_t *>(&hi)) << 16);
            }
            reinterpret_cast<uint4 *>(p)[k] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
};

template <typename T>
__global__ __launch_bounds__(256) void pmx_ln32_fwd_kernel(const T *__restrict__ x, const T *__restrict__ a, const float *__restrict__ w,
                                                           const float *__restrict__ b, T *__restrict__ y, float *__restrict__ mean_out,
                                                           float *__restrict__ rstd_out, long rows, float eps)
{
    constexpr int D = 32;
    float wv[D], bv[D];
#pragma unroll
    for (int j = 0; j < D; ++j) { wv[j] = w[j]; bv[j] = b[j]; }
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
        float z[D], t[D];
        LnIO<T>::load(x + r * D, z);
        LnIO<T>::load(a + r * D, t);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < D; ++j) { z[j] += t[j]; s += z[j]; }
        const float mean = s * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < D; ++j) { const float c = z[j] - mean; q += c * c; }
        const float rstd = rsqrtf(q * (1.0f / D) + eps);
#pragma unroll
        for (int j = 0; j < D; ++j) t[j] = (z[j] - mean) * rstd * wv[j] + bv[j];
        LnIO<T>::store(y + r * D, t);
        mean_out[r] = mean; rstd_out[r] = rstd;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pmx_ln32_bwd_kernel(const T *__restrict__ x, const T *__restrict__ a, const T *__restrict__ dy,
                                                           const float *__restrict__ w, const float *__restrict__ mean_in,
                                                           const float *__restrict__ rstd_in, T *__restrict__ dz,
                                                           float *__restrict__ partial /*[n_waves][64]*/, long rows)
{
    constexpr int D = 32;
    float wv[D], gw[D], gb[D];
#pragma unroll
    for (int j = 0; j < D; ++j) { wv[j] = w[j]; gw[j] = 0.f; gb[j] = 0.f; }
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
        float z[D], t[D], g[D];
        LnIO<T>::load(x + r * D, z);
        LnIO<T>::load(a + r * D, t);
        LnIO<T>::load(dy + r * D, g);
        const float mean = mean_in[r], rstd = rstd_in[r];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const float xh = (z[j] + t[j] - mean) * rstd;
            gw[j] += g[j] * xh; gb[j] += g[j];
            const float gg = g[j] * wv[j];
            c1 += gg; c2 += gg * xh;
            z[j] = xh; g[j] = gg;
        }
        c1 *= (1.0f / D); c2 *= (1.0f / D);
#pragma unroll
        for (int j = 0; j < D; ++j) t[j] = rstd * (g[j] - c1 - z[j] * c2);
        LnIO<T>::store(dz + r * D, t);
    }
    // wavefront reduction of the 64 accumulators; every wave stores its partial sums (no atomics: 4 k waves adding into
    // the same 64 floats serialise at the memory side -- 1.6 ms measured -- and would not be bitwise reproducible)
#pragma unroll
    for (int j = 0; j < D; ++j) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { gw[j] += __shfl_xor(gw[j], o); gb[j] += __shfl_xor(gb[j], o); }
    }
    if ((threadIdx.x & 63) == 0) {
        float *dst = partial + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64;
#pragma unroll
        for (int j = 0; j < D; ++j) { dst[j] = gw[j]; dst[32 + j] = gb[j]; }
    }
}

// dtype: 0 float32, 1 bfloat16.  Feature dimension fixed at 32.
extern "C" int pmx_ln32_forward(const void *x, const void *a, const float *w, const float *b, void *y, float *mean, float *rstd,
                                int64_t rows, float eps, int32_t dtype, void *stream)
{
    if (!x || !a || !w || !b || !y || !mean || !rstd || rows < 0) return PMX_ERR_INVALID;
    if (rows == 0) return PMX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)std::min<int64_t>((rows + 255) / 256, 4096);
    if (dtype == 0)
        hipLaunchKernelGGL(pmx_ln32_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)x, (const float *)a, w, b, (float *)y, mean, rstd, (long)rows, eps);
    else
        hipLaunchKernelGGL(pmx_ln32_fwd_kernel<__hip_bfloat16>, dim3(grid), dim3(256), 0, st, (const __hip_bfloat16 *)x, (const __hip_bfloat16 *)a, w, b,
                           (__hip_bfloat16 *)y, mean, rstd, (long)rows, eps);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

// partial: [PMX_LN32_PARTIAL_ROWS][64] float32, fully overwritten: row k holds one wavefront's sums of dw (first 32) and db
// (last 32); the caller adds the rows up (a tiny reduction).
extern "C" int pmx_ln32_backward(const void *x, const void *a, const void *dy, const float *w, const float *mean, const float *rstd,
                                 void *dz, float *partial, int64_t rows, int32_t dtype, void *stream)
{
    if (!x || !a || !dy || !w || !mean || !rstd || !dz || !partial || rows < 0) return PMX_ERR_INVALID;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned grid = PMX_LN32_PARTIAL_ROWS / 4;   // 4 wavefronts per block, every wave writes its row (zeros if idle)
    if (dtype == 0)
        hipLaunchKernelGGL(pmx_ln32_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)x, (const float *)a, (const float *)dy, w, mean, rstd,
                           (float *)dz, partial, (long)rows);
    else
        hipLaunchKernelGGL(pmx_ln32_bwd_kernel<__hip_bfloat16>, dim3(grid), dim3(256), 0, st, (const __hip_bfloat16 *)x, (const __hip_bfloat16 *)a,
                           (const __hip_bfloat16 *)dy, w, mean, rstd, (__hip_bfloat16 *)dz, partial, (long)rows);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}


// ---------------------------------------------------------------------------------------------------------------
// Fused GroupNorm + (residual add) + GELU for the actor's residual blocks (pacman_mappo_resnet.py:49-67:
// act(gn1(conv1(x))) and act(gn2(conv2(.)) + x)), NCHW, 8 channels per group (GroupNorm(4, 32)).
// One WAVEFRONT per (sample, group) row = 8 * H*W contiguous elements, read once into registers
// (v[channel][k], element lane + 64 k of the channel); float32 statistics via wavefront shuffles, exact (erf) GELU.
// The backward kernel recomputes the normalised values, applies GELU', returns d/d(input), d/d(residual) and
// per-sample partial sums of the weight / bias gradients ([B][C][2], summed over B by the caller: no atomics).
// ---------------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float gn_ld(const T *p);
template <> __device__ __forceinline__ float gn_ld<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float gn_ld<__hip_bfloat16>(const __hip_bfloat16 *p) { return __bfloat162float(*p); }
template <typename T> __device__ __forceinline__ void gn_st(T *p, float v);
template <> __device__ __forceinline__ void gn_st<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void gn_st<__hip_bfloat16>(__hip_bfloat16 *p, float v) { *p = __float2bfloat16(v); }

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float gelu_f(float z) { return 0.5f * z * (1.0f + erff(z * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float z)
{
    return 0.5f * (1.0f + erff(z * 0.70710678118654752f)) + z * 0.39894228040143268f * __expf(-0.5f * z * z);
}

template <typename T, int KMAX>
__global__ __launch_bounds__(256) void pmx_gn8_gelu_fwd_kernel(const T *__restrict__ h, const T *__restrict__ res, const float *__restrict__ w,
                                                               const float *__restrict__ b, T *__restrict__ y, float *__restrict__ mean_out,
                                                               float *__restrict__ rstd_out, long rows, int groups, int HW, float eps)
{
    constexpr int CPG = 8;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int g = (int)(row % groups);
    const size_t base = (size_t)row * CPG * HW;            // (n * C + g * 8) * HW with C = groups * 8
    float v[CPG][KMAX];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CPG; ++c)
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int e = lane + 64 * k;
            v[c][k] = e < HW ? gn_ld<T>(h + base + (size_t)c * HW + e) : 0.f;
            s += v[c][k];
        }
    const float inv = 1.0f / (float)(CPG * HW);
    const float mean = wave_sum(s) * inv;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CPG; ++c)
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int e = lane + 64 * k;
            const float d = e < HW ? v[c][k] - mean : 0.f;
            q += d * d;
        }
    const float rstd = rsqrtf(wave_sum(q) * inv + eps);
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
        const float wc = w[g * CPG + c] * rstd, bc = b[g * CPG + c];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int e = lane + 64 * k;
            if (e < HW) {
                const size_t idx = base + (size_t)c * HW + e;
                float z = (v[c][k] - mean) * wc + bc;
                if (res) z += gn_ld<T>(res + idx);
                gn_st<T>(y + idx, gelu_f(z));
            }
        }
    }
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

template <typename T, int KMAX>
__global__ __launch_bounds__(256) void pmx_gn8_gelu_bwd_kernel(const T *__restrict__ h, const T *__restrict__ res, const T *__restrict__ dy,
                                                               const float *__restrict__ w, const float *__restrict__ b,
                                                               const float *__restrict__ mean_in, const float *__restrict__ rstd_in,
                                                               T *__restrict__ dh, T *__restrict__ dres, float *__restrict__ partial /*[rows][8][2]*/,
                                                               long rows, int groups, int HW)
{
    constexpr int CPG = 8;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int g = (int)(row % groups);
    const size_t base = (size_t)row * CPG * HW;
    const float mean = mean_in[row], rstd = rstd_in[row];
    float xh[CPG][KMAX], gz[CPG][KMAX];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
        const float wc = w[g * CPG + c], bc = b[g * CPG + c];
        float sw = 0.f, sb = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int e = lane + 64 * k;
            xh[c][k] = 0.f; gz[c][k] = 0.f;
            if (e < HW) {
                const size_t idx = base + (size_t)c * HW + e;
                const float x = (gn_ld<T>(h + idx) - mean) * rstd;
                float z = x * wc + bc;
                if (res) z += gn_ld<T>(res + idx);
                const float dz = gn_ld<T>(dy + idx) * gelu_grad_f(z);
                if (dres) gn_st<T>(dres + idx, dz);
                sw += dz * x; sb += dz;
                const float gg = dz * wc;
                xh[c][k] = x; gz[c][k] = gg;
                c1 += gg; c2 += gg * x;
            }
        }
        sw = wave_sum(sw); sb = wave_sum(sb);
        if (lane == 0) { partial[((size_t)row * CPG + c) * 2] = sw; partial[((size_t)row * CPG + c) * 2 + 1] = sb; }
    }
    const float inv = 1.0f / (float)(CPG * HW);
    c1 = wave_sum(c1) * inv; c2 = wave_sum(c2) * inv;
#pragma unroll
    for (int c = 0; c < CPG; ++c)
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int e = lane + 64 * k;
            if (e < HW) gn_st<T>(dh + base + (size_t)c * HW + e, rstd * (gz[c][k] - c1 - xh[c][k] * c2));
        }
}

// [B][C][H*W] tensors with C = groups * 8; dtype 0 float32, 1 bfloat16; res_dev may be NULL (no residual).
extern "C" int pmx_gn8_gelu_forward(const void *h, const void *res, const float *w, const float *b, void *y, float *mean, float *rstd,
                                    int64_t B, int32_t groups, int32_t HW, float eps, int32_t dtype, void *stream)
{
    if (!h || !w || !b || !y || !mean || !rstd || B < 0 || groups < 1 || HW < 1 || HW > 1024) return PMX_ERR_INVALID;
    const long rows = (long)B * groups;
    if (rows == 0) return PMX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)((rows + 3) / 4);
#define PMX_GN_FWD(T, K) hipLaunchKernelGGL((pmx_gn8_gelu_fwd_kernel<T, K>), dim3(grid), dim3(256), 0, st, (const T *)h, (const T *)res, w, b, (T *)y, mean, rstd, rows, groups, HW, eps)
    if (dtype == 0) { if (HW <= 192) PMX_GN_FWD(float, 3); else if (HW <= 448) PMX_GN_FWD(float, 7); else PMX_GN_FWD(float, 16); }
    else { if (HW <= 192) PMX_GN_FWD(__hip_bfloat16, 3); else if (HW <= 448) PMX_GN_FWD(__hip_bfloat16, 7); else PMX_GN_FWD(__hip_bfloat16, 16); }
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

// dh (and dres when res_dev is given) are written in full; partial [B * groups][8][2] float32 receives, per (sample, group)
// row and channel of the group, the sums of dz * xhat (weight gradient) and dz (bias gradient).
extern "C" int pmx_gn8_gelu_backward(const void *h, const void *res, const void *dy, const float *w, const float *b, const float *mean,
                                     const float *rstd, void *dh, void *dres, float *partial, int64_t B, int32_t groups, int32_t HW,
                                     int32_t dtype, void *stream)
{
    if (!h || !dy || !w || !b || !mean || !rstd || !dh || !partial || B < 0 || groups < 1 || HW < 1 || HW > 1024) return PMX_ERR_INVALID;
    if ((res == nullptr) != (dres == nullptr)) return PMX_ERR_INVALID;
    const long rows = (long)B * groups;
    if (rows == 0) return PMX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)((rows + 3) / 4);
#define PMX_GN_BWD(T, K) hipLaunchKernelGGL((pmx_gn8_gelu_bwd_kernel<T, K>), dim3(grid), dim3(256), 0, st, (const T *)h, (const T *)res, (const T *)dy, w, b, mean, rstd, (T *)dh, (T *)dres, partial, rows, groups, HW)
    if (dtype == 0) { if (HW <= 192) PMX_GN_BWD(float, 3); else if (HW <= 448) PMX_GN_BWD(float, 7); else PMX_GN_BWD(float, 16); }
    else { if (HW <= 192) PMX_GN_BWD(__hip_bfloat16, 3); else if (HW <= 448) PMX_GN_BWD(__hip_bfloat16, 7); else PMX_GN_BWD(__hip_bfloat16, 16); }
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}


// ---------------------------------------------------------------------------------------------------------------
// Small-sequence self-attention forward on the matrix cores (the critic: S = H*W <= 1024 tokens, 4 heads x 8,
// pacman_mappo_resnet.py:138-141), bf16 in, f32 accumulate.  One WAVEFRONT per (sample, head), the four heads of a
// sample in one block; q, k, v are read straight from the packed in-projection output [S][B][3E] and the result is
// written as [S][B][E], so no head split / merge copies exist.
//   S^T tile = K_tile . Q_tile^T      v_mfma_f32_16x16x32_bf16, head_dim 8 zero-padded to the K = 32 of the instruction;
//                                     the accumulator holds S^T[key = 4*(lane>>4)+reg][query = lane&15]
//   online softmax per query column   (max / sum over the 8 keys a lane holds, then over the four lane groups: 2 shuffles)
//   O^T += V^T . P^T                  the exp'd accumulators of two key tiles ARE the B fragment of the next MFMA: the
//                                     contraction index (keys) may be enumerated in any order as long as the A operand
//                                     (V^T, staged transposed in LDS) uses the same one -- no LDS round trip for P.
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) short pmx_bf16x8;
typedef __attribute__((ext_vector_type(4))) float pmx_f32x4;
typedef __attribute__((ext_vector_type(2))) float pmx_f32x2;
typedef __attribute__((ext_vector_type(4))) short pmx_bf16x4;

__device__ __forceinline__ short pmx_f2bf(float f)
{
    const __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<const short *>(&h);
}

__global__ __launch_bounds__(512) void pmx_attn8_fwd_kernel(const __hip_bfloat16 *__restrict__ qkv, __hip_bfloat16 *__restrict__ out,
                                                            float *__restrict__ lse, int S, int B, float scale, int bm)
{
    constexpr int D = 8, HEADS = 4, E = 32;
    extern __shared__ __align__(16) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x, h = wave & 3, role = wave >> 2;        // two wavefronts per head: they split the query tiles
    const int S_pad = (S + 31) & ~31;
    // per head: K [S_pad][8] bf16, then V^T [8][S_pad] bf16
    short *Ks = reinterpret_cast<short *>(smem) + (size_t)h * 2 * S_pad * D;
    short *Vt = Ks + (size_t)S_pad * D;
    const short *base = reinterpret_cast<const short *>(qkv);
    // elements between consecutive sequence positions: sequence-major [S][B][.] (nn.MultiheadAttention's batch_first=False) or
    // batch-major [B][S][.] (bm: a sample's rows are one contiguous 192 S bytes)
    const size_t row_stride = bm ? (size_t)3 * E : (size_t)B * 3 * E;
    const size_t head_off = (size_t)b * 3 * E * (bm ? S : 1) + (size_t)h * D;
    const size_t out_row = bm ? (size_t)E : (size_t)B * E, out_off = (size_t)b * E * (bm ? S : 1) + (size_t)h * D;
    for (int s = lane + 64 * role; s < S_pad; s += 128) {
        uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (s < S) {
            kv = *reinterpret_cast<const uint4 *>(base + (size_t)s * row_stride + head_off + E);
            vv = *reinterpret_cast<const uint4 *>(base + (size_t)s * row_stride + head_off + 2 * E);
        }
        *reinterpret_cast<uint4 *>(Ks + (size_t)s * D) = kv;
        const short *vs = reinterpret_cast<const short *>(&vv);
#pragma unroll
        for (int d = 0; d < D; ++d) Vt[(size_t)d * S_pad + s] = vs[d];
    }
    __syncthreads();

    const int g = lane >> 4, c = lane & 15;
    const pmx_bf16x8 zero8 = { 0, 0, 0, 0, 0, 0, 0, 0 };
    const int n_qt = (S + 15) >> 4, n_kp = S_pad >> 5;
    for (int qt = role; qt < n_qt; qt += 2) {
        const int q_row = qt * 16 + c;
        pmx_bf16x8 qf = zero8;
        if (g == 0 && q_row < S) qf = *reinterpret_cast<const pmx_bf16x8 *>(base + (size_t)q_row * row_stride + head_off);
        // Scores stay in the raw (unscaled) domain; c2 = scale * log2(e) turns them into base-2 exponents with ONE fma per
        // element (exp2(s * c2 - m)), the running maximum m and the stored log-sum-exp are kept in that base-2 domain.
        float m = -1e30f, l = 0.f;
        pmx_f32x4 o = { 0.f, 0.f, 0.f, 0.f };
        const float c2 = scale * 1.44269504088896341f;
        auto tile = [&](int kp, bool last) {
            pmx_bf16x8 k0 = zero8, k1 = zero8;
            if (g == 0) {
                k0 = *reinterpret_cast<const pmx_bf16x8 *>(Ks + (size_t)(kp * 32 + c) * D);
                k1 = *reinterpret_cast<const pmx_bf16x8 *>(Ks + (size_t)(kp * 32 + 16 + c) * D);
            }
            const pmx_f32x4 z4 = { 0.f, 0.f, 0.f, 0.f };
            pmx_f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf, z4, 0, 0, 0);
            pmx_f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf, z4, 0, 0, 0);
            float p[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { p[r] = s0[r]; p[4 + r] = s1[r]; }
            if (last) {   // only the last key pair can hold padded keys: they must not enter the row sum
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key0 = kp * 32 + g * 4 + r, key1 = key0 + 16;
                    if (key0 >= S) p[r] = -1e30f;
                    if (key1 >= S) p[4 + r] = -1e30f;
                }
            }
            float mloc = fmaxf(fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])), fmaxf(fmaxf(p[4], p[5]), fmaxf(p[6], p[7])));
            mloc = fmaxf(mloc, __shfl_xor(mloc, 16));
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
            const float mnew = fmaxf(m, mloc * c2);
            const float alpha = __builtin_amdgcn_exp2f(m - mnew);
            float lsum = 0.f;
            pmx_bf16x8 pf;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(p[r], c2, -mnew));
                lsum += e;
                pf[r] = pmx_f2bf(e);
            }
            lsum += __shfl_xor(lsum, 16);
            lsum += __shfl_xor(lsum, 32);
            l = l * alpha + lsum;
            m = mnew;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] *= alpha;
            // A = V^T: row d = c (zero for d >= 8), k-slot j -> key kp*32 + 4g + j (j < 4), kp*32 + 16 + 4g + (j-4)
            pmx_bf16x8 vf = zero8;
            if (c < D) {
                const short *vrow = Vt + (size_t)c * S_pad + kp * 32 + g * 4;
                const uint2 lo = *reinterpret_cast<const uint2 *>(vrow);
                const uint2 hi = *reinterpret_cast<const uint2 *>(vrow + 16);
                const uint4 both = make_uint4(lo.x, lo.y, hi.x, hi.y);
                vf = *reinterpret_cast<const pmx_bf16x8 *>(&both);
            }
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o, 0, 0, 0);
        };
        for (int kp = 0; kp < n_kp - 1; ++kp) tile(kp, false);
        tile(n_kp - 1, true);
        // O^T[d = 4g + r][query c]: lanes of groups 0 and 1 hold d = 0..3 and 4..7
        if (q_row < S) {
            const float inv = 1.0f / l;
            if (g < 2) {
                short w4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) w4[r] = pmx_f2bf(o[r] * inv);
                *reinterpret_cast<uint2 *>(reinterpret_cast<short *>(out) + (size_t)q_row * out_row + out_off + g * 4) =
                    *reinterpret_cast<const uint2 *>(w4);
            }
            if (g == 0 && lse) lse[((size_t)b * HEADS + h) * S + q_row] = (m + __log2f(l)) * 0.69314718055994531f;   // back to the natural log
        }
    }
}

// squared norm of eight bf16 values (one head's row of Q or K)
__device__ __forceinline__ float sq_norm8(uint4 w)
{
    const uint32_t u[4] = {w.x, w.y, w.z, w.w};
    float n = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = __uint_as_float(u[i] << 16), hi = __uint_as_float(u[i] & 0xFFFF0000u);
        n = fmaf(lo, lo, n), n = fmaf(hi, hi, n);
    }
    return n;
}

// ---------------------------------------------------------------------------------------------------------------
// The forward kernel the product launches (pmx_attn8_fwd_kernel above is kept as the PMX_ATTN_FWD_V1 A/B reference).  The
// kernel is bound by the vector ALU work of the softmax, so this version removes everything from the inner loop that is not
// the exponential itself:
//   * The exponent's reference is FIXED before the loop over the keys: no running maximum, no rescaling of the output tile, no
//     cross-lane traffic inside the loop.  It enters as the initial accumulator of the score product (s - m comes out of the
//     matrix instruction), leaving exp2((s - m) c) = one multiply and one v_exp_f32 per score.  The reference is the
//     Cauchy-Schwarz bound max_i |q_i| max_j |k_j| of the head (both maxima are found while K and V are staged) whenever that is
//     harmless: every score lies within [-bound, bound], so with 2 c bound <= 88 the largest probability of a row is at least
//     2^-88 -- a normal float, and a normal bf16 -- and the common factor cancels in O = P V / l and in the log-sum-exp.  A head
//     past that limit takes the exact path: a first pass over the keys that only finds each query's largest raw score (two
//     matrix instructions and four v_maximum3_f32 per 32 keys), ~20 % of the kernel's issue slots when every head needed it.
//   * The row sum is a matrix product too: V^T is staged with a ninth row of ones, so row 8 of O^T accumulates the sum of the
//     (bf16-rounded) probabilities that multiply V -- the normaliser is consistent with the numerator and costs no vector add.
//     The log-sum-exp the backward kernels exponentiate against is taken from the UNROUNDED probabilities instead (four packed
//     adds per 32 keys): against a reference above the row's largest score no probability is exactly 1, and a row one key
//     dominates would carry that key's bf16 rounding (up to 2^-9) into every probability the backward recomputes.
//   * Operands need no masks: lane groups 1..3 of an A operand fill k-slots 8..31, which meet the zeros of the B operand (the
//     query row lives in group 0 only), so every lane simply reads its row (the four groups read the same 256 bytes: a broadcast).
// Per 32 keys and lane: 8 multiplies, 8 exponentials, 4 packed conversions (the first kernel: ~75 vector instructions).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void pmx_attn8_fwd2_kernel(const __hip_bfloat16 *__restrict__ qkv, __hip_bfloat16 *__restrict__ out,
                                                             float *__restrict__ lse, int S, int B, float scale, int bm)
{
    constexpr int D = 8, HEADS = 4, E = 32;
    extern __shared__ __align__(16) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x, h = wave & 3, role = wave >> 2;        // two wavefronts per head: they split the query tiles
    const int S_pad = (S + 31) & ~31;
    // per head: K [S_pad][8] bf16, then V^T [9][S_pad] bf16 (row 8 = ones)
    short *Ks = reinterpret_cast<short *>(smem) + (size_t)h * (D + 9) * S_pad;
    short *Vt = Ks + (size_t)S_pad * D;
    const short *base = reinterpret_cast<const short *>(qkv);
    const size_t row_stride = bm ? (size_t)3 * E : (size_t)B * 3 * E;
    const size_t head_off = (size_t)b * 3 * E * (bm ? S : 1) + (size_t)h * D;
    const size_t out_row = bm ? (size_t)E : (size_t)B * E, out_off = (size_t)b * E * (bm ? S : 1) + (size_t)h * D;
    float k2max = 0.0f, q2max = 0.0f;                                 // largest squared norms of the head's keys and queries
    for (int s = lane + 64 * role; s < S_pad; s += 128) {
        uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (s < S) {
            kv = *reinterpret_cast<const uint4 *>(base + (size_t)s * row_stride + head_off + E);
            vv = *reinterpret_cast<const uint4 *>(base + (size_t)s * row_stride + head_off + 2 * E);
            q2max = fmaxf(q2max, sq_norm8(*reinterpret_cast<const uint4 *>(base + (size_t)s * row_stride + head_off)));
        }
        *reinterpret_cast<uint4 *>(Ks + (size_t)s * D) = kv;
        const short *vs = reinterpret_cast<const short *>(&vv);
#pragma unroll
        for (int d = 0; d < D; ++d) Vt[(size_t)d * S_pad + s] = vs[d];
        Vt[(size_t)D * S_pad + s] = s < S ? (short)0x3F80 : (short)0;          // 1.0 for the real keys
        k2max = fmaxf(k2max, sq_norm8(kv));
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) k2max = fmaxf(k2max, __shfl_xor(k2max, off)), q2max = fmaxf(q2max, __shfl_xor(q2max, off));
    float *knorm = reinterpret_cast<float *>(reinterpret_cast<short *>(smem) + (size_t)HEADS * (D + 9) * S_pad);     // [8 waves][2]
    if (lane == 0) knorm[2 * wave] = k2max, knorm[2 * wave + 1] = q2max;
    __syncthreads();
    // the head's two waves staged alternate blocks of 64 rows.  |q| |k| <= bound for every pair of the head (a hair above: the
    // root and the products round)
    const float bound = __builtin_sqrtf(fmaxf(knorm[2 * h], knorm[2 * h + 8]) * fmaxf(knorm[2 * h + 1], knorm[2 * h + 9])) * 1.0001f;

    const int g = lane >> 4, c = lane & 15;
    const pmx_bf16x8 zero8 = { 0, 0, 0, 0, 0, 0, 0, 0 };
    const pmx_f32x4 z4 = { 0.f, 0.f, 0.f, 0.f };
    const int n_qt = (S + 15) >> 4, n_kp = S_pad >> 5;
    const float c2 = scale * 1.44269504088896341f;                  // raw score -> base-2 exponent
    const short *krow = Ks + (size_t)c * D;                                              // + kp * 32 * D (+ 16 * D)
    const short *vrow = Vt + (size_t)(c < D ? c : D) * S_pad + g * 4;                    // + kp * 32 (+ 16); rows >= 9 of the result are unused
    const int last_lo = S - (n_kp - 1) * 32 - g * 4;               // key index r (resp. 16 + r) of the last pair is real iff r < last_lo (- 16)
    const bool exact = bound * c2 > 44.0f;                          // (uniform over the head's two waves)
    pmx_f32x2 real[4];                                              // 1 for the lane's real keys of the last pair, 0 for its padded ones
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r0 = (i & 1) * 2 + (i >> 1) * 16;                 // e[2 i], e[2 i + 1] are keys r0, r0 + 1 (+ 4 g) of the pair
        real[i] = pmx_f32x2{ r0 < last_lo ? 1.f : 0.f, r0 + 1 < last_lo ? 1.f : 0.f };
    }
    for (int qt = role; qt < n_qt; qt += 2) {
        const int q_row = qt * 16 + c;
        pmx_bf16x8 qf = zero8;
        if (g == 0 && q_row < S) qf = *reinterpret_cast<const pmx_bf16x8 *>(base + (size_t)q_row * row_stride + head_off);
        float mx = bound;
        if (exact) {
        // ---- exact path, pass 1: the largest raw score of each query ------------------------------------------------------
        mx = -3.0e38f;
#pragma unroll 2
        for (int kp = 0; kp < n_kp - 1; ++kp) {
            const pmx_bf16x8 k0 = *reinterpret_cast<const pmx_bf16x8 *>(krow + (size_t)kp * 32 * D);
            const pmx_bf16x8 k1 = *reinterpret_cast<const pmx_bf16x8 *>(krow + (size_t)(kp * 32 + 16) * D);
            const pmx_f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf, z4, 0, 0, 0);
            const pmx_f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf, z4, 0, 0, 0);
            mx = __builtin_elementwise_maximum(__builtin_elementwise_maximum(s0[0], s0[1]), mx);
            mx = __builtin_elementwise_maximum(__builtin_elementwise_maximum(s0[2], s0[3]), mx);
            mx = __builtin_elementwise_maximum(__builtin_elementwise_maximum(s1[0], s1[1]), mx);
            mx = __builtin_elementwise_maximum(__builtin_elementwise_maximum(s1[2], s1[3]), mx);
        }
        {   // the last pair may hold padded keys (zero K rows: score 0): they must not raise the reference
            const int kp = n_kp - 1;
            const pmx_bf16x8 k0 = *reinterpret_cast<const pmx_bf16x8 *>(krow + (size_t)kp * 32 * D);
            const pmx_bf16x8 k1 = *reinterpret_cast<const pmx_bf16x8 *>(krow + (size_t)(kp * 32 + 16) * D);
            const pmx_f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf, z4, 0, 0, 0);
            const pmx_f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf, z4, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mx = __builtin_elementwise_maximum(r < last_lo ? s0[r] : -3.0e38f, mx);
                mx = __builtin_elementwise_maximum(r + 16 < last_lo ? s1[r] : -3.0e38f, mx);
            }
        }
        mx = __builtin_elementwise_maximum(mx, __shfl_xor(mx, 16));
        mx = __builtin_elementwise_maximum(mx, __shfl_xor(mx, 32));
        }
        // ---- pass 2: O^T (+ the row sums in row 8) against the fixed reference ---------------------------------------------
        const pmx_f32x4 negm = { -mx, -mx, -mx, -mx };
        pmx_f32x4 o = z4;
        pmx_f32x2 ls[2] = { { 0.f, 0.f }, { 0.f, 0.f } };          // the lane's sum of UNROUNDED probabilities, for the log-sum-exp
        // mode 0: a pair of key tiles without padding.  The last pair may hold padded keys (zero K rows, zero V^T columns INCLUDING the
        // row of ones: whatever their probability, they add nothing to O or to its normaliser).  Mode 2 (reference = the bound, so
        // the padded keys' exponent -bound c is <= 0): only the exact row sum must leave them out -- a multiply-add with the lane's 0 / 1
        // mask instead of the add.  Mode 1 (exact path: the largest score may be far below zero and exp2(-m c) overflow): the padded
        // probabilities are set to zero by compares, as everywhere before.
        auto tile = [&](int kp, auto mode_c) {
            constexpr int mode = decltype(mode_c)::value;
            const pmx_bf16x8 k0 = *reinterpret_cast<const pmx_bf16x8 *>(krow + (size_t)kp * 32 * D);
            const pmx_bf16x8 k1 = *reinterpret_cast<const pmx_bf16x8 *>(krow + (size_t)(kp * 32 + 16) * D);
            const pmx_f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf, negm, 0, 0, 0);     // s - m
            const pmx_f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf, negm, 0, 0, 0);
            const pmx_f32x2 cc = { c2, c2 };
            const pmx_f32x2 x[4] = { pmx_f32x2{ s0[0], s0[1] } * cc, pmx_f32x2{ s0[2], s0[3] } * cc, pmx_f32x2{ s1[0], s1[1] } * cc,
                                     pmx_f32x2{ s1[2], s1[3] } * cc };                            // (packed multiplies)
            float e[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) e[r] = __builtin_amdgcn_exp2f(x[r >> 1][r & 1]);
            if constexpr (mode == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r >= last_lo) e[r] = 0.f;
                    if (r + 16 >= last_lo) e[4 + r] = 0.f;
                }
            }
            if constexpr (mode == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ls[i & 1] = __builtin_elementwise_fma(pmx_f32x2{ e[2 * i], e[2 * i + 1] }, real[i], ls[i & 1]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) ls[i & 1] += pmx_f32x2{ e[2 * i], e[2 * i + 1] };
            }
            pmx_bf16x8 pf;
#pragma unroll
            for (int r = 0; r < 8; ++r) pf[r] = pmx_f2bf(e[r]);
            // A = V^T (+ ones): row c, k-slot j -> key kp*32 + 4g + j (j < 4), kp*32 + 16 + 4g + (j-4): the order of pf's slots
            const uint2 lo = *reinterpret_cast<const uint2 *>(vrow + kp * 32);
            const uint2 hi = *reinterpret_cast<const uint2 *>(vrow + kp * 32 + 16);
            const uint4 both = make_uint4(lo.x, lo.y, hi.x, hi.y);
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const pmx_bf16x8 *>(&both), pf, o, 0, 0, 0);
        };
#pragma unroll 2
        for (int kp = 0; kp < n_kp - 1; ++kp) tile(kp, std::integral_constant<int, 0>{});
        if (exact) tile(n_kp - 1, std::integral_constant<int, 1>{});
        else tile(n_kp - 1, std::integral_constant<int, 2>{});
        float lx = (ls[0][0] + ls[0][1]) + (ls[1][0] + ls[1][1]);   // the four lane groups hold different keys of query c
        lx += __shfl_xor(lx, 16);
        lx += __shfl_xor(lx, 32);
        // O^T[row 4g + r][query c]: groups 0 and 1 hold d = 0..3 and 4..7, group 2 holds the row sums in r = 0
        const float l = __shfl(o[0], 32 + c);
        if (q_row < S) {
            const float inv = 1.0f / l;
            if (g < 2) {
                short w4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) w4[r] = pmx_f2bf(o[r] * inv);
                *reinterpret_cast<uint2 *>(reinterpret_cast<short *>(out) + (size_t)q_row * out_row + out_off + g * 4) =
                    *reinterpret_cast<const uint2 *>(w4);
            }
            if (g == 0 && lse) lse[((size_t)b * HEADS + h) * S + q_row] = (mx * c2 + __log2f(lx)) * 0.69314718055994531f;  // natural log
        }
    }
}

// qkv_dev [S][B][96] bf16 (the packed in-projection of nn.MultiheadAttention with embed 32, 4 heads), out_dev [S][B][32]
// bf16, lse_dev [B][4][S] float32 (log-sum-exp of the scaled scores per query; may be NULL).  batch_major: [B][S][.] instead.
extern "C" int pmx_attn8_forward(const void *qkv_dev, void *out_dev, float *lse_dev, int32_t S, int32_t B, void *stream)
{
    return pmx_attn8_forward_layout(qkv_dev, out_dev, lse_dev, S, B, 0, stream);
}
extern "C" int pmx_attn8_forward_layout(const void *qkv_dev, void *out_dev, float *lse_dev, int32_t S, int32_t B, int32_t batch_major, void *stream)
{
    if (!qkv_dev || !out_dev || S < 1 || S > 1024 || B < 0) return PMX_ERR_INVALID;
    if (B == 0) return PMX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int S_pad = (S + 31) & ~31;
    int cur_dev = 0;
    if (hipGetDevice(&cur_dev) != hipSuccess || cur_dev < 0 || cur_dev >= 64) return PMX_ERR_HIP;
    static const bool v1 = getenv("PMX_ATTN_FWD_V1") != nullptr;                    // A/B switch, read once
    if (!v1) {
        const size_t lds2 = (size_t)4 * (8 + 9) * S_pad * sizeof(short) + 16 * sizeof(float);
        static bool attr2_dev[64] = {};
        if (lds2 > 65536 && !attr2_dev[cur_dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(pmx_attn8_fwd2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return PMX_ERR_HIP;
            attr2_dev[cur_dev] = true;
        }
        hipLaunchKernelGGL(pmx_attn8_fwd2_kernel, dim3(B), dim3(512), lds2, st, (const __hip_bfloat16 *)qkv_dev, (__hip_bfloat16 *)out_dev, lse_dev, S, B,
                           0.35355339059327379f /* 1/sqrt(8) */, batch_major ? 1 : 0);
        return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
    }
    const size_t lds = (size_t)4 * 2 * S_pad * 8 * sizeof(short);
    static bool attr_set_dev[64] = {};          // the attribute belongs to the function ON THE CURRENT DEVICE
    bool &attr_set = attr_set_dev[cur_dev];
    if (lds > 65536 && !attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(pmx_attn8_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return PMX_ERR_HIP;
        attr_set = true;
    }
    hipLaunchKernelGGL(pmx_attn8_fwd_kernel, dim3(B), dim3(512), lds, st, (const __hip_bfloat16 *)qkv_dev, (__hip_bfloat16 *)out_dev, lse_dev, S, B,
                       0.35355339059327379f /* 1/sqrt(8) */, batch_major ? 1 : 0);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}


// ---------------------------------------------------------------------------------------------------------------
// Backward of the small-sequence attention (same shapes as pmx_attn8_fwd_kernel), recomputing the probabilities from
// the saved log-sum-exp.  One wavefront per (sample, head), two passes, every product on v_mfma_f32_16x16x32_bf16:
//   pass A (per 16-query tile, lanes = queries):  S^T, dP^T tiles over key pairs -> dS^T = P^T o (dP^T - Delta_q)
//                                                 dQ^T += K^T . dS^T          (the accumulators are the B fragment)
//   pass B (per 16-key tile, lanes = keys):       S, dP tiles over query pairs -> P, dS = P o (dP - Delta_q)
//                                                 dV^T += dO^T . P,  dK^T += Q^T . dS
// K^T, Q^T, dO^T are staged transposed in LDS ([8][S_pad] bf16), Delta_q = sum_d dO.O and the LSE as float32 [S_pad];
// the row-major operands of the score products are read straight from global memory (16 bytes per lane, L1-resident).
// Gradients are written in the packed [S][B][96] layout of the in-projection output.
// ---------------------------------------------------------------------------------------------------------------
// NPF > 0: the sequence has exactly NPF pairs of 16-row tiles and each wave keeps ITS row operands (K and V rows for the dQ
// pass, Q and dO rows for the dK/dV pass) in registers for all of its tiles instead of re-reading them from global memory
// once per tile (10x for S = 154): 16 NPF registers, loaded once.
template <int NPF>
__global__ __launch_bounds__(512) void pmx_attn8_bwd_kernel(const __hip_bfloat16 *__restrict__ qkv, const __hip_bfloat16 *__restrict__ outp,
                                                            const __hip_bfloat16 *__restrict__ dout, const float *__restrict__ lse,
                                                            __hip_bfloat16 *__restrict__ dqkv, int S, int B, float scale, int bm)
{
    constexpr int D = 8, HEADS = 4, E = 32;
    extern __shared__ __align__(16) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x, h = wave & 3, role = wave >> 2;        // two wavefronts per head: role 0 = dQ pass, role 1 = dK/dV pass
    const int S_pad = (S + 31) & ~31;
    const size_t per_wave = (size_t)3 * D * S_pad * sizeof(short) + (size_t)2 * S_pad * sizeof(float);
    unsigned char *mine = smem + (size_t)h * per_wave;
    short *Kt = reinterpret_cast<short *>(mine);                    // [8][S_pad]
    short *Qt = Kt + (size_t)D * S_pad;
    short *dOt = Qt + (size_t)D * S_pad;
    float *lse_s = reinterpret_cast<float *>(dOt + (size_t)D * S_pad);
    float *delta_s = lse_s + S_pad;
    const short *base = reinterpret_cast<const short *>(qkv);
    const short *obase = reinterpret_cast<const short *>(outp);
    const short *dobase = reinterpret_cast<const short *>(dout);
    const size_t row_stride = bm ? (size_t)3 * E : (size_t)B * 3 * E, orow = bm ? (size_t)E : (size_t)B * E;      // as in the forward kernel
    const size_t head_off = (size_t)b * 3 * E * (bm ? S : 1) + (size_t)h * D, ohead = (size_t)b * E * (bm ? S : 1) + (size_t)h * D;

    for (int s = lane + 64 * role; s < S_pad; s += 128) {
        uint4 qv = make_uint4(0, 0, 0, 0), kv = qv, dov = qv, ov = qv;
        float ls = 1e30f;                                            // padded queries: exp(score - 1e30) = 0
        if (s < S) {
            qv = *reinterpret_cast<const uint4 *>(base + (size_t)s * row_stride + head_off);
            kv = *reinterpret_cast<const uint4 *>(base + (size_t)s * row_stride + head_off + E);
            dov = *reinterpret_cast<const uint4 *>(dobase + (size_t)s * orow + ohead);
            ov = *reinterpret_cast<const uint4 *>(obase + (size_t)s * orow + ohead);
            ls = lse[((size_t)b * HEADS + h) * S + s] * 1.44269504088896341f;      // natural log -> the base-2 domain of exp2 below
        }
        const short *q8 = reinterpret_cast<const short *>(&qv), *k8 = reinterpret_cast<const short *>(&kv);
        const short *d8 = reinterpret_cast<const short *>(&dov), *o8 = reinterpret_cast<const short *>(&ov);
        float delta = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            Kt[(size_t)d * S_pad + s] = k8[d];
            Qt[(size_t)d * S_pad + s] = q8[d];
            dOt[(size_t)d * S_pad + s] = d8[d];
            delta += __uint_as_float((uint32_t)(uint16_t)d8[d] << 16) * __uint_as_float((uint32_t)(uint16_t)o8[d] << 16);
        }
        lse_s[s] = ls; delta_s[s] = delta;
    }
    __syncthreads();

    const int g = lane >> 4, c = lane & 15;
    const pmx_bf16x8 zero8 = { 0, 0, 0, 0, 0, 0, 0, 0 };
    const pmx_f32x4 z4 = { 0.f, 0.f, 0.f, 0.f };
    const int n_t = (S + 15) >> 4, n_p = S_pad >> 5;
    const float c2 = scale * 1.44269504088896341f;     // scores -> base-2 exponents with one fma: exp2(s * c2 - lse2)
    short *dbase = reinterpret_cast<short *>(dqkv);

    auto row8 = [&](const short *src, size_t stride, size_t off, int r) -> pmx_bf16x8 {   // 8 bf16 of row r for lanes of group 0
        pmx_bf16x8 v = zero8;
        if (g == 0 && r < S) v = *reinterpret_cast<const pmx_bf16x8 *>(src + (size_t)r * stride + off);
        return v;
    };
    auto tfrag = [&](const short *T, int pair) -> pmx_bf16x8 {       // A fragment [row d = c][k-slot j]: index 32*pair + 4g + j / + 16
        pmx_bf16x8 v = zero8;
        if (c < D) {
            const short *p = T + (size_t)c * S_pad + pair * 32 + g * 4;
            const uint2 lo = *reinterpret_cast<const uint2 *>(p), hi = *reinterpret_cast<const uint2 *>(p + 16);
            const uint4 both = make_uint4(lo.x, lo.y, hi.x, hi.y);
            v = *reinterpret_cast<const pmx_bf16x8 *>(&both);
        }
        return v;
    };

    // ---- pass A: dQ
    if (NPF > 0 && role == 0) {
        pmx_bf16x8 kr[NPF > 0 ? 2 * NPF : 1], vr[NPF > 0 ? 2 * NPF : 1];
#pragma unroll
        for (int t = 0; t < 2 * NPF; ++t) {
            kr[t] = row8(base, row_stride, head_off + E, t * 16 + c);
            vr[t] = row8(base, row_stride, head_off + 2 * E, t * 16 + c);
        }
        for (int qt = 0; qt < n_t; ++qt) {
            const int q_row = qt * 16 + c;
            const pmx_bf16x8 qf = row8(base, row_stride, head_off, q_row);
            const pmx_bf16x8 dof = row8(dobase, orow, ohead, q_row);
            const float ls = fmaxf(lse_s[q_row < S_pad ? q_row : 0], -120.f), dl = delta_s[q_row < S_pad ? q_row : 0];
            pmx_f32x4 dq = z4;
#pragma unroll
            for (int kp = 0; kp < NPF; ++kp) {
                const pmx_f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kr[2 * kp], qf, z4, 0, 0, 0);
                const pmx_f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kr[2 * kp + 1], qf, z4, 0, 0, 0);
                const pmx_f32x4 p0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vr[2 * kp], dof, z4, 0, 0, 0);
                const pmx_f32x4 p1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vr[2 * kp + 1], dof, z4, 0, 0, 0);
                pmx_bf16x8 dsf;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c2, -ls));
                    const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c2, -ls));
                    dsf[r] = pmx_f2bf(e0 * (p0[r] - dl));
                    dsf[4 + r] = pmx_f2bf(e1 * (p1[r] - dl));
                }
                dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tfrag(Kt, kp), dsf, dq, 0, 0, 0);
            }
            if (q_row < S && g < 2) {
                short w4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) w4[r] = pmx_f2bf(dq[r] * scale);
                *reinterpret_cast<uint2 *>(dbase + (size_t)q_row * row_stride + head_off + g * 4) = *reinterpret_cast<const uint2 *>(w4);
            }
        }
    }
    if (NPF == 0 && role == 0)
    for (int qt = 0; qt < n_t; ++qt) {
        const int q_row = qt * 16 + c;
        const pmx_bf16x8 qf = row8(base, row_stride, head_off, q_row);
        const pmx_bf16x8 dof = row8(dobase, orow, ohead, q_row);
        // (the clamp only matters for the zero K rows of padded keys: exp2(-ls) must stay finite so that 0 * dS is 0 in the MFMA)
        const float ls = fmaxf(lse_s[q_row < S_pad ? q_row : 0], -120.f), dl = delta_s[q_row < S_pad ? q_row : 0];
        pmx_f32x4 dq = z4;
        // software pipeline: the row fragments of key pair kp + 1 are requested before pair kp is consumed (the loads are
        // L2 hits several hundred cycles away and the compiler does not hoist them across the loop by itself)
        pmx_bf16x8 k0n = row8(base, row_stride, head_off + E, c), k1n = row8(base, row_stride, head_off + E, 16 + c);
        pmx_bf16x8 v0n = row8(base, row_stride, head_off + 2 * E, c), v1n = row8(base, row_stride, head_off + 2 * E, 16 + c);
        for (int kp = 0; kp < n_p; ++kp) {
            const pmx_bf16x8 k0 = k0n, k1 = k1n, v0 = v0n, v1 = v1n;
            if (kp + 1 < n_p) {
                k0n = row8(base, row_stride, head_off + E, kp * 32 + 32 + c); k1n = row8(base, row_stride, head_off + E, kp * 32 + 48 + c);
                v0n = row8(base, row_stride, head_off + 2 * E, kp * 32 + 32 + c); v1n = row8(base, row_stride, head_off + 2 * E, kp * 32 + 48 + c);
            }
            const pmx_f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf, z4, 0, 0, 0);
            const pmx_f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf, z4, 0, 0, 0);
            const pmx_f32x4 p0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, dof, z4, 0, 0, 0);
            const pmx_f32x4 p1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, dof, z4, 0, 0, 0);
            // no key mask: the K^T columns of padded keys are zero in LDS, so whatever dS holds there contributes nothing
            pmx_bf16x8 dsf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c2, -ls));
                const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c2, -ls));
                dsf[r] = pmx_f2bf(e0 * (p0[r] - dl));
                dsf[4 + r] = pmx_f2bf(e1 * (p1[r] - dl));
            }
            dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tfrag(Kt, kp), dsf, dq, 0, 0, 0);
        }
        if (q_row < S && g < 2) {
            short w4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) w4[r] = pmx_f2bf(dq[r] * scale);
            *reinterpret_cast<uint2 *>(dbase + (size_t)q_row * row_stride + head_off + g * 4) = *reinterpret_cast<const uint2 *>(w4);
        }
    }
    // ---- pass B: dK, dV
    if (NPF > 0 && role == 1) {
        pmx_bf16x8 qr[NPF > 0 ? 2 * NPF : 1], dr[NPF > 0 ? 2 * NPF : 1];
#pragma unroll
        for (int t = 0; t < 2 * NPF; ++t) {
            qr[t] = row8(base, row_stride, head_off, t * 16 + c);
            dr[t] = row8(dobase, orow, ohead, t * 16 + c);
        }
        for (int kt = 0; kt < n_t; ++kt) {
            const int k_row = kt * 16 + c;
            const pmx_bf16x8 kf = row8(base, row_stride, head_off + E, k_row);
            const pmx_bf16x8 vf = row8(base, row_stride, head_off + 2 * E, k_row);
            pmx_f32x4 dk = z4, dv = z4;
#pragma unroll
            for (int qp = 0; qp < NPF; ++qp) {
                const pmx_f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qr[2 * qp], kf, z4, 0, 0, 0);
                const pmx_f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qr[2 * qp + 1], kf, z4, 0, 0, 0);
                const pmx_f32x4 p0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dr[2 * qp], vf, z4, 0, 0, 0);
                const pmx_f32x4 p1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dr[2 * qp + 1], vf, z4, 0, 0, 0);
                pmx_bf16x8 pf, dsf;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qa = qp * 32 + g * 4 + r, qb = qa + 16;
                    const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c2, -lse_s[qa]));
                    const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c2, -lse_s[qb]));
                    pf[r] = pmx_f2bf(e0); pf[4 + r] = pmx_f2bf(e1);
                    dsf[r] = pmx_f2bf(e0 * (p0[r] - delta_s[qa]));
                    dsf[4 + r] = pmx_f2bf(e1 * (p1[r] - delta_s[qb]));
                }
                dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tfrag(dOt, qp), pf, dv, 0, 0, 0);
                dk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tfrag(Qt, qp), dsf, dk, 0, 0, 0);
            }
            if (k_row < S && g < 2) {
                short wk[4], wv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { wk[r] = pmx_f2bf(dk[r] * scale); wv[r] = pmx_f2bf(dv[r]); }
                *reinterpret_cast<uint2 *>(dbase + (size_t)k_row * row_stride + head_off + E + g * 4) = *reinterpret_cast<const uint2 *>(wk);
                *reinterpret_cast<uint2 *>(dbase + (size_t)k_row * row_stride + head_off + 2 * E + g * 4) = *reinterpret_cast<const uint2 *>(wv);
            }
        }
    }
    if (NPF == 0 && role == 1)
    for (int kt = 0; kt < n_t; ++kt) {
        const int k_row = kt * 16 + c;
        const pmx_bf16x8 kf = row8(base, row_stride, head_off + E, k_row);
        const pmx_bf16x8 vf = row8(base, row_stride, head_off + 2 * E, k_row);
        pmx_f32x4 dk = z4, dv = z4;
        pmx_bf16x8 q0n = row8(base, row_stride, head_off, c), q1n = row8(base, row_stride, head_off, 16 + c);
        pmx_bf16x8 d0n = row8(dobase, orow, ohead, c), d1n = row8(dobase, orow, ohead, 16 + c);
        for (int qp = 0; qp < n_p; ++qp) {
            const pmx_bf16x8 q0 = q0n, q1 = q1n, d0 = d0n, d1 = d1n;
            if (qp + 1 < n_p) {
                q0n = row8(base, row_stride, head_off, qp * 32 + 32 + c); q1n = row8(base, row_stride, head_off, qp * 32 + 48 + c);
                d0n = row8(dobase, orow, ohead, qp * 32 + 32 + c); d1n = row8(dobase, orow, ohead, qp * 32 + 48 + c);
            }
            // S = Q . K^T: rows = queries (4g + r), column = key c
            const pmx_f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0, kf, z4, 0, 0, 0);
            const pmx_f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1, kf, z4, 0, 0, 0);
            const pmx_f32x4 p0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d0, vf, z4, 0, 0, 0);
            const pmx_f32x4 p1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d1, vf, z4, 0, 0, 0);
            pmx_bf16x8 pf, dsf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qa = qp * 32 + g * 4 + r, qb = qa + 16;
                // padded queries carry lse = 1e30 (probability 0); a padded KEY is this lane's own column, which is never stored
                const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c2, -lse_s[qa]));
                const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c2, -lse_s[qb]));
                pf[r] = pmx_f2bf(e0); pf[4 + r] = pmx_f2bf(e1);
                dsf[r] = pmx_f2bf(e0 * (p0[r] - delta_s[qa]));
                dsf[4 + r] = pmx_f2bf(e1 * (p1[r] - delta_s[qb]));
            }
            dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tfrag(dOt, qp), pf, dv, 0, 0, 0);
            dk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tfrag(Qt, qp), dsf, dk, 0, 0, 0);
        }
        if (k_row < S && g < 2) {
            short wk[4], wv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { wk[r] = pmx_f2bf(dk[r] * scale); wv[r] = pmx_f2bf(dv[r]); }
            *reinterpret_cast<uint2 *>(dbase + (size_t)k_row * row_stride + head_off + E + g * 4) = *reinterpret_cast<const uint2 *>(wk);
            *reinterpret_cast<uint2 *>(dbase + (size_t)k_row * row_stride + head_off + 2 * E + g * 4) = *reinterpret_cast<const uint2 *>(wv);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same backward with ONE pass and one wavefront per (sample, head), for sequences of exactly NPF pairs of 16-row tiles
// (160 padded tokens: tinyCapture / smallCapture).  The two-pass kernel computes every probability and every dS twice -- once
// with keys on the rows for dQ, once with queries on the rows for dK / dV -- and the softmax arithmetic is what bounds it.
// Here the dK / dV orientation is computed once per (key tile, query pair); its dS tile (queries on the accumulator rows, keys on
// the lanes) is written to a 32-row LDS staging block as [key][query] and read back through ds_read_b64_tr_b16 as the B operand
// dS^T[key][query] of dQ^T += K^T . dS^T, whose A operand is 8 consecutive keys of K^T per lane group straight from LDS.  dQ^T of
// all ten query tiles stays in registers (40) across the key tiles.
// ---------------------------------------------------------------------------------------------------------------
// QS = 2 (sequences too long for one wave's registers: 416 padded tokens = the 20 x 20 boards): TWO wavefronts per (sample,
// head) split the QUERY pairs -- each keeps the row operands and the dQ^T tiles of its own half (7 / 6 pairs: 168 registers) and
// walks all key tiles; its dK / dV tiles then only cover its half of the queries, so after every key tile the second wave hands
// its two tiles to the first through a double-buffered LDS slot (one block barrier per key tile), which adds them and stores.
template <int NPF, int QS>
__global__ __launch_bounds__(256 * QS) void pmx_attn8_bwd_fused_kernel(const __hip_bfloat16 *__restrict__ qkv, const __hip_bfloat16 *__restrict__ outp,
                                                                       const __hip_bfloat16 *__restrict__ dout, const float *__restrict__ lse,
                                                                       __hip_bfloat16 *__restrict__ dqkv, int S, int B, float scale, int bm)
{
    constexpr int D = 8, HEADS = 4, E = 32, S_pad = 32 * NPF, TROW = 80;       // TROW: bytes per staging row (32 queries + pad)
    constexpr int NQP = (NPF + QS - 1) / QS;                                   // query pairs per wave
    constexpr int XCH = 2 * 2 * 16 * 16;                                       // bytes of one exchange slot: (dK, dV) x 2 lane groups x 16 keys x 4 floats
    extern __shared__ __align__(16) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = wave & 3, role = __builtin_amdgcn_readfirstlane(wave >> 2);
    const int b = blockIdx.x;
    // QS > 1: the dO rows (A operands of the dP products) also sit in LDS, row-major, instead of in 8 NQP registers per wave
    constexpr size_t per_head = (size_t)3 * D * S_pad * sizeof(short) + (size_t)2 * S_pad * sizeof(float) + (size_t)QS * 32 * TROW +
                                (QS > 1 ? 2 * XCH + (size_t)S_pad * D * sizeof(short) : 0);
    unsigned char *mine = smem + (size_t)h * per_head;
    short *Kt = reinterpret_cast<short *>(mine);                    // [8][S_pad]
    short *Qt = Kt + (size_t)D * S_pad;
    short *dOt = Qt + (size_t)D * S_pad;
    float *lse_s = reinterpret_cast<float *>(dOt + (size_t)D * S_pad);
    float *delta_s = lse_s + S_pad;
    char *stg = reinterpret_cast<char *>(delta_s + S_pad) + (size_t)role * 32 * TROW;   // [32 rows = keys of the tile (16..31 stay zero)][32 queries] bf16, one per wave
    unsigned char *xch = reinterpret_cast<unsigned char *>(delta_s + S_pad) + (size_t)QS * 32 * TROW;
    short *dOr = reinterpret_cast<short *>(xch + 2 * XCH);          // [S_pad][8] (QS > 1 only)
    const short *base = reinterpret_cast<const short *>(qkv);
    const short *obase = reinterpret_cast<const short *>(outp);
    const short *dobase = reinterpret_cast<const short *>(dout);
    const size_t row_stride = bm ? (size_t)3 * E : (size_t)B * 3 * E, orow = bm ? (size_t)E : (size_t)B * E;
    const size_t head_off = (size_t)b * 3 * E * (bm ? S : 1) + (size_t)h * D, ohead = (size_t)b * E * (bm ? S : 1) + (size_t)h * D;

    for (int s = lane + 64 * role; s < S_pad; s += 64 * QS) {
        uint4 qv = make_uint4(0, 0, 0, 0), kv = qv, dov = qv, ov = qv;
        float ls = 1e30f;                                           // padded queries: exp2(c (score - 1e30 / c)) = 0
        if (s < S) {
            qv = *reinterpret_cast<const uint4 *>(base + (size_t)s * row_stride + head_off);
            kv = *reinterpret_cast<const uint4 *>(base + (size_t)s * row_stride + head_off + E);
            dov = *reinterpret_cast<const uint4 *>(dobase + (size_t)s * orow + ohead);
            ov = *reinterpret_cast<const uint4 *>(obase + (size_t)s * orow + ohead);
            ls = lse[((size_t)b * HEADS + h) * S + s] * 1.44269504088896341f;
        }
        const short *q8 = reinterpret_cast<const short *>(&qv), *k8 = reinterpret_cast<const short *>(&kv);
        const short *d8 = reinterpret_cast<const short *>(&dov), *o8 = reinterpret_cast<const short *>(&ov);
        float delta = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            Kt[(size_t)d * S_pad + s] = k8[d];
            Qt[(size_t)d * S_pad + s] = q8[d];
            dOt[(size_t)d * S_pad + s] = d8[d];
            delta += __uint_as_float((uint32_t)(uint16_t)d8[d] << 16) * __uint_as_float((uint32_t)(uint16_t)o8[d] << 16);
        }
        if (QS > 1) *reinterpret_cast<uint4 *>(dOr + (size_t)s * D) = dov;
        // Row constants as INITIAL ACCUMULATORS of the two score products: S' = Q K^T - lse / c comes out of the matrix instruction
        // ready for p = exp2(c S') (a multiply and the exponential), dP' = dO V^T - delta ready for dS = p dP' (one multiply).
        lse_s[s] = -ls / (scale * 1.44269504088896341f); delta_s[s] = -delta;
    }
    for (int i = lane; i < 32 * TROW / 4; i += 64) reinterpret_cast<uint32_t *>(stg)[i] = 0u;
    if (QS == 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // each head's LDS is private to its wavefront: no block barrier
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();                                            // the head's two waves staged half of the rows each
    }

    const int g = lane >> 4, c = lane & 15;
    const pmx_bf16x8 zero8 = { 0, 0, 0, 0, 0, 0, 0, 0 };
    const pmx_f32x4 z4 = { 0.f, 0.f, 0.f, 0.f };
    const int n_t = (S + 15) >> 4;
    const float c2 = scale * 1.44269504088896341f;
    short *dbase = reinterpret_cast<short *>(dqkv);
    auto row8 = [&](const short *src, size_t stride, size_t off, int r) -> pmx_bf16x8 {
        pmx_bf16x8 v = zero8;
        if (g == 0 && r < S) v = *reinterpret_cast<const pmx_bf16x8 *>(src + (size_t)r * stride + off);
        return v;
    };
    // A fragment [row d][k-slot j] of a transposed array: rows 8..15 of the operand only feed rows 8..15 of the product, which
    // nobody reads, so those lanes simply repeat rows 0..7 (no exec mask, no zero fill in the loop)
    auto tfrag = [&](const short *T, int pair) -> pmx_bf16x8 {
        const short *p = T + (size_t)(c & (D - 1)) * S_pad + pair * 32 + g * 4;
        const uint2 lo = *reinterpret_cast<const uint2 *>(p), hi = *reinterpret_cast<const uint2 *>(p + 16);
        const uint4 both = make_uint4(lo.x, lo.y, hi.x, hi.y);
        return *reinterpret_cast<const pmx_bf16x8 *>(&both);
    };
    auto pack2bf = [](float x, float y) -> uint32_t {             // one v_cvt_pk_bf16_f32
        typedef __attribute__((ext_vector_type(2))) float f2;
        typedef __attribute__((ext_vector_type(2))) __bf16 b2;
        const f2 f = {x, y};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, b2));
    };
    const int qp0 = role * NQP;                                     // this wave's query pairs: qp0 .. qp0 + NQP - 1 (those < NPF)
    pmx_bf16x8 qr[2 * NQP], dr[QS > 1 ? 1 : 2 * NQP];
    pmx_f32x4 dq[2 * NQP];
#pragma unroll
    for (int t = 0; t < 2 * NQP; ++t) {
        qr[t] = row8(base, row_stride, head_off, (2 * qp0 + t) * 16 + c);       // (rows past the sequence read as zero)
        if (QS == 1) dr[t] = row8(dobase, orow, ohead, (2 * qp0 + t) * 16 + c);
        dq[t] = z4;
    }
    // QS > 1: a dO row operand is one ds_read_b128 by EVERY lane (the four lane groups read the same 16 rows): the k-slots 8..31 it
    // fills in groups 1..3 meet the zeros of the B operand (the V rows live in group 0 only), so they need no masking
    const short *dOr_lane = dOr + (size_t)c * D;
    const int tr_row = (lane & 15) >> 2, tr_pc = lane & 3;
    // The K / V rows of a key tile come from global memory and its dK / dV rows go back there; loads and stores share one in-order
    // counter, so rows requested at the top of a tile would wait behind the previous tile's stores (a full write round trip per
    // tile).  They are requested one tile AHEAD, in front of those stores.
    pmx_bf16x8 kf_n = row8(base, row_stride, head_off + E, c), vf_n = row8(base, row_stride, head_off + 2 * E, c);
    for (int kt = 0; kt < n_t; ++kt) {
        const int k_row = kt * 16 + c;
        const pmx_bf16x8 kf = kf_n, vf = vf_n;
        if (kt + 1 < n_t) {
            kf_n = row8(base, row_stride, head_off + E, k_row + 16);
            vf_n = row8(base, row_stride, head_off + 2 * E, k_row + 16);
        }
        // A operand of dQ^T += K^T . dS^T for this key tile: row d = c, k-slot (g, j) = key 16 kt + 8 g + j (groups 2, 3: none)
        pmx_bf16x8 ka = zero8;
        if (c < D && g < 2) ka = *reinterpret_cast<const pmx_bf16x8 *>(Kt + (size_t)c * S_pad + kt * 16 + 8 * g);
        pmx_f32x4 dk = z4, dv = z4;
        // The pairs of a key tile as a SOFTWARE PIPELINE.  One pair is a chain  operand reads -> score products -> exponentials ->
        // dV / dK products, dS to the staging rows -> transposing reads -> dQ products, and with two waves per SIMD nothing else
        // covers its three LDS round trips (the kernel sat at ~50 % vector-ALU issue).  So the steps of consecutive pairs are
        // interleaved by hand, every consumer at least one other step behind its producer, and scheduling barriers keep the
        // compiler from re-serialising them:
        //   [reads j+1 | fragment reads j] [exponentials j] [dV, dK products j; dS j -> staging; transposing reads j]
        //   [score products j+1] [dQ products j]
        struct Ops { pmx_f32x4 nl0, nl1, nd0, nd1; pmx_bf16x8 d0, d1; };
        struct Sc { pmx_f32x4 s0, s1, p0, p1; };
        auto pair_live = [&](int j) { return !(QS > 1 && NQP * QS > NPF && j == NQP - 1 && qp0 + j >= NPF); };   // wave-uniform: the last
                                                                          // wave owns one pair less (run-time only for j = NQP - 1)
        auto read_ops = [&](int j) -> Ops {
            const int qp = qp0 + j;
            Ops o;
            // (accumulator row r of lane group g is query 32 qp + 4 g + r, resp. + 16: four consecutive floats each)
            o.nl0 = *reinterpret_cast<const pmx_f32x4 *>(lse_s + qp * 32 + g * 4), o.nl1 = *reinterpret_cast<const pmx_f32x4 *>(lse_s + qp * 32 + 16 + g * 4);
            o.nd0 = *reinterpret_cast<const pmx_f32x4 *>(delta_s + qp * 32 + g * 4), o.nd1 = *reinterpret_cast<const pmx_f32x4 *>(delta_s + qp * 32 + 16 + g * 4);
            if (QS > 1) {
                o.d0 = *reinterpret_cast<const pmx_bf16x8 *>(dOr_lane + (size_t)(qp * 32) * D);
                o.d1 = *reinterpret_cast<const pmx_bf16x8 *>(dOr_lane + (size_t)(qp * 32 + 16) * D);
            } else {
                o.d0 = dr[QS > 1 ? 0 : 2 * j], o.d1 = dr[QS > 1 ? 0 : 2 * j + 1];
            }
            return o;
        };
        auto score = [&](int j, const Ops &o) -> Sc {
            Sc r;
            r.s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qr[2 * j], kf, o.nl0, 0, 0, 0);
            r.s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qr[2 * j + 1], kf, o.nl1, 0, 0, 0);
            r.p0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o.d0, vf, o.nd0, 0, 0, 0);
            r.p1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o.d1, vf, o.nd1, 0, 0, 0);
            return r;
        };
        Sc cur = score(0, read_ops(0));
#pragma unroll
        for (int j = 0; j < NQP; ++j) {
            const int qp = qp0 + j;
            if (!pair_live(j)) continue;
            const bool more = j + 1 < NQP && pair_live(j + 1);
            // ---- reads for the next pair's score products and this pair's transposed fragments, then this pair's exponentials
            Ops nxt_ops;
            if (more) nxt_ops = read_ops(j + 1);
            const pmx_bf16x8 fo = tfrag(dOt, qp), fq = tfrag(Qt, qp);
            float e0[4], e1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e0[r] = __builtin_amdgcn_exp2f(cur.s0[r] * c2);
                e1[r] = __builtin_amdgcn_exp2f(cur.s1[r] * c2);
            }
            // probabilities and dS as bf16 pairs: slots 0..3 = queries 4g + r of the first tile of the pair, 4..7 = of the second
            const uint4 pw = make_uint4(pack2bf(e0[0], e0[1]), pack2bf(e0[2], e0[3]), pack2bf(e1[0], e1[1]), pack2bf(e1[2], e1[3]));
            const uint4 dw = make_uint4(pack2bf(e0[0] * cur.p0[0], e0[1] * cur.p0[1]), pack2bf(e0[2] * cur.p0[2], e0[3] * cur.p0[3]),
                                        pack2bf(e1[0] * cur.p1[0], e1[1] * cur.p1[1]), pack2bf(e1[2] * cur.p1[2], e1[3] * cur.p1[3]));
            const pmx_bf16x8 pf = *reinterpret_cast<const pmx_bf16x8 *>(&pw), dsf = *reinterpret_cast<const pmx_bf16x8 *>(&dw);
            __builtin_amdgcn_sched_barrier(0);
            // ---- dV / dK products; dS of the tile as [key c][queries 4g .. 4g+3 | 16 + 4g ..] -> staging rows 0..15 (a padded key's
            // column holds whatever the zero K row produced; its K^T entries in `ka` are zero, so it adds nothing); the transposing
            // reads are issued right behind the write.  Write, reads and the next pair's write are LDS operations of ONE wave on one
            // array: the hardware runs them in program order and the compiler keeps may-aliasing accesses in order.
            dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fo, pf, dv, 0, 0, 0);
            dk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq, dsf, dk, 0, 0, 0);
            *reinterpret_cast<uint2 *>(stg + c * TROW + (4 * g) * 2) = make_uint2(dw.x, dw.y);
            *reinterpret_cast<uint2 *>(stg + c * TROW + (16 + 4 * g) * 2) = make_uint2(dw.z, dw.w);
            pmx_bf16x8 bT[2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const char *a0 = stg + (8 * g + tr_row) * TROW + half * 32 + tr_pc * 8;
                const pmx_bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pmx_bf16x4 __attribute__((address_space(3))) *)(a0));
                const pmx_bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pmx_bf16x4 __attribute__((address_space(3))) *)(a0 + 4 * TROW));
                bT[half] = pmx_bf16x8{ lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3] };
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- the next pair's score products run while the transposing reads come back
            Sc nxt = cur;
            if (more) nxt = score(j + 1, nxt_ops);
            __builtin_amdgcn_sched_barrier(0);
            dq[2 * j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, bT[0], dq[2 * j], 0, 0, 0);
            dq[2 * j + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, bT[1], dq[2 * j + 1], 0, 0, 0);
            cur = nxt;
        }
        if (QS > 1) {
            // the second wave's partial dK^T / dV^T tiles (its half of the queries) -> slot kt & 1; the first wave adds them after the
            // barrier.  Double-buffered: the slot written for tile kt + 1 is not the one being read for tile kt, and the slot of
            // tile kt + 2 is written only after barrier kt + 1, which the reader reaches after its reads of tile kt.
            float *slot = reinterpret_cast<float *>(xch + (size_t)(kt & 1) * XCH);
            if (role == 1 && g < 2) {
                *reinterpret_cast<pmx_f32x4 *>(slot + ((0 * 2 + g) * 16 + c) * 4) = dk;
                *reinterpret_cast<pmx_f32x4 *>(slot + ((1 * 2 + g) * 16 + c) * 4) = dv;
            }
            __syncthreads();
            if (role == 0 && g < 2) {
                const pmx_f32x4 a = *reinterpret_cast<const pmx_f32x4 *>(slot + ((0 * 2 + g) * 16 + c) * 4);
                const pmx_f32x4 e = *reinterpret_cast<const pmx_f32x4 *>(slot + ((1 * 2 + g) * 16 + c) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) { dk[r] += a[r]; dv[r] += e[r]; }
            }
        }
        if (role == 0 && k_row < S && g < 2) {
            short wk[4], wv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { wk[r] = pmx_f2bf(dk[r] * scale); wv[r] = pmx_f2bf(dv[r]); }
            *reinterpret_cast<uint2 *>(dbase + (size_t)k_row * row_stride + head_off + E + g * 4) = *reinterpret_cast<const uint2 *>(wk);
            *reinterpret_cast<uint2 *>(dbase + (size_t)k_row * row_stride + head_off + 2 * E + g * 4) = *reinterpret_cast<const uint2 *>(wv);
        }
    }
#pragma unroll
    for (int t = 0; t < 2 * NQP; ++t) {
        const int q_row = (2 * qp0 + t) * 16 + c;
        if (q_row < S && g < 2) {
            short w4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) w4[r] = pmx_f2bf(dq[t][r] * scale);
            *reinterpret_cast<uint2 *>(dbase + (size_t)q_row * row_stride + head_off + g * 4) = *reinterpret_cast<const uint2 *>(w4);
        }
    }
}

// dqkv_dev [S][B][96] bf16 is written in full.  S <= 640 (LDS: 56 * S_pad bytes per wavefront).
extern "C" int pmx_attn8_backward(const void *qkv_dev, const void *out_dev, const void *dout_dev, const float *lse_dev, void *dqkv_dev,
                                  int32_t S, int32_t B, void *stream)
{
    return pmx_attn8_backward_layout(qkv_dev, out_dev, dout_dev, lse_dev, dqkv_dev, S, B, 0, stream);
}
extern "C" int pmx_attn8_backward_layout(const void *qkv_dev, const void *out_dev, const void *dout_dev, const float *lse_dev, void *dqkv_dev,
                                         int32_t S, int32_t B, int32_t batch_major, void *stream)
{
    if (!qkv_dev || !out_dev || !dout_dev || !lse_dev || !dqkv_dev || S < 1 || S > 640 || B < 0) return PMX_ERR_INVALID;
    if (B == 0) return PMX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int S_pad = (S + 31) & ~31;
    const size_t lds = (size_t)4 * ((size_t)3 * 8 * S_pad * sizeof(short) + (size_t)2 * S_pad * sizeof(float));
    static bool attr_set_dev[64] = {};          // the attribute belongs to the function ON THE CURRENT DEVICE
    int cur_dev = 0;
    if (hipGetDevice(&cur_dev) != hipSuccess || cur_dev < 0 || cur_dev >= 64) return PMX_ERR_HIP;
    bool &attr_set = attr_set_dev[cur_dev];
    if (lds > 65536 && !attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(pmx_attn8_bwd_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return PMX_ERR_HIP;
        attr_set = true;
    }
    static const bool generic_only = getenv("PMX_ATTN_BWD_GENERIC") != nullptr;      // A/B switch, read once
    static const bool two_pass = getenv("PMX_ATTN_BWD_TWO_PASS") != nullptr;         // A/B switch, read once
    if (S_pad == 160 && !generic_only && !two_pass) {
        // one pass, one wavefront per (sample, head): 4 x (3 x 8 x 160 x 2 + 2 x 160 x 4 + 32 x 80) = 45 KB of LDS
        constexpr size_t lds_f = (size_t)4 * ((size_t)3 * 8 * 160 * sizeof(short) + (size_t)2 * 160 * sizeof(float) + 32 * 80);
        hipLaunchKernelGGL((pmx_attn8_bwd_fused_kernel<5, 1>), dim3(B), dim3(256), lds_f, st, (const __hip_bfloat16 *)qkv_dev, (const __hip_bfloat16 *)out_dev,
                           (const __hip_bfloat16 *)dout_dev, lse_dev, (__hip_bfloat16 *)dqkv_dev, S, B, 0.35355339059327379f, batch_major ? 1 : 0);
        return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
    }
    if (S_pad == 416 && !generic_only && !two_pass) {
        // one pass, two wavefronts per (sample, head) splitting the query pairs (the 20 x 20 boards: 400 tokens):
        // 4 x (4 x 8 x 416 x 2 + 2 x 416 x 4 + 2 x 32 x 80 + 2 x 1 024) = 145 KB of LDS, one block of eight waves per CU
        constexpr size_t lds_f = (size_t)4 * ((size_t)4 * 8 * 416 * sizeof(short) + (size_t)2 * 416 * sizeof(float) + 2 * 32 * 80 + 2 * 1024);
        static bool attr2_dev[64] = {};
        if (!attr2_dev[cur_dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(pmx_attn8_bwd_fused_kernel<13, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return PMX_ERR_HIP;
            attr2_dev[cur_dev] = true;
        }
        hipLaunchKernelGGL((pmx_attn8_bwd_fused_kernel<13, 2>), dim3(B), dim3(512), lds_f, st, (const __hip_bfloat16 *)qkv_dev, (const __hip_bfloat16 *)out_dev,
                           (const __hip_bfloat16 *)dout_dev, lse_dev, (__hip_bfloat16 *)dqkv_dev, S, B, 0.35355339059327379f, batch_major ? 1 : 0);
        return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
    }
    if (S_pad == 160 && !generic_only)       // tinyCapture and smallCapture (154 cells): the row operands of a wave fit in 80 registers
        hipLaunchKernelGGL(pmx_attn8_bwd_kernel<5>, dim3(B), dim3(512), lds, st, (const __hip_bfloat16 *)qkv_dev, (const __hip_bfloat16 *)out_dev,
                           (const __hip_bfloat16 *)dout_dev, lse_dev, (__hip_bfloat16 *)dqkv_dev, S, B, 0.35355339059327379f, batch_major ? 1 : 0);
    else
        hipLaunchKernelGGL(pmx_attn8_bwd_kernel<0>, dim3(B), dim3(512), lds, st, (const __hip_bfloat16 *)qkv_dev, (const __hip_bfloat16 *)out_dev,
                           (const __hip_bfloat16 *)dout_dev, lse_dev, (__hip_bfloat16 *)dqkv_dev, S, B, 0.35355339059327379f, batch_major ? 1 : 0);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}


// ---------------------------------------------------------------------------------------------------------------
// The same fused GroupNorm(8 ch/group) + residual + GELU for channels-last tensors ([B][H*W][C] in memory, what
// MIOpen's NHWC implicit-GEMM convolutions produce and consume): the 8 channels of a group at one position are 16
// contiguous bytes of bf16, so a lane owns whole positions (lane + 64 k) and every access is one 16-byte load / store.
// Running the actor channels-last end to end removes the NCHW<->NHWC transposes around every convolution.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cl_load8(const __hip_bfloat16 *p, float *v)
{
    const uint4 t = *reinterpret_cast<const uint4 *>(p);
    const uint32_t w[4] = { t.x, t.y, t.z, t.w };
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w[j] << 16); v[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u); }
}
__device__ __forceinline__ void cl_store8(__hip_bfloat16 *p, const float *v)
{
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = (uint32_t)(uint16_t)pmx_f2bf(v[2 * j]) | ((uint32_t)(uint16_t)pmx_f2bf(v[2 * j + 1]) << 16);
    *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}

template <int KMAX>
__global__ __launch_bounds__(256) void pmx_gn8cl_gelu_fwd_kernel(const __hip_bfloat16 *__restrict__ h, const __hip_bfloat16 *__restrict__ res,
                                                                 const float *__restrict__ w, const float *__restrict__ b,
                                                                 __hip_bfloat16 *__restrict__ y, float *__restrict__ mean_out,
                                                                 float *__restrict__ rstd_out, long rows, int groups, int HW, float eps)
{
    constexpr int CPG = 8;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int g = (int)(row % groups);
    const long n = row / groups;
    const int C = groups * CPG;
    const size_t base = (size_t)n * HW * C + (size_t)g * CPG;       // + pos * C
    float v[KMAX][CPG];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int pos = lane + 64 * k;
        if (pos < HW) cl_load8(h + base + (size_t)pos * C, v[k]);
        else {
#pragma unroll
            for (int c = 0; c < CPG; ++c) v[k][c] = 0.f;
        }
#pragma unroll
        for (int c = 0; c < CPG; ++c) s += v[k][c];
    }
    const float inv = 1.0f / (float)(CPG * HW);
    const float mean = wave_sum(s) * inv;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (lane + 64 * k < HW) {
#pragma unroll
            for (int c = 0; c < CPG; ++c) { const float d = v[k][c] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) * inv + eps);
    float wc[CPG], bc[CPG];
#pragma unroll
    for (int c = 0; c < CPG; ++c) { wc[c] = w[g * CPG + c] * rstd; bc[c] = b[g * CPG + c]; }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int pos = lane + 64 * k;
        if (pos < HW) {
            float r8[CPG], o8[CPG];
            if (res) cl_load8(res + base + (size_t)pos * C, r8);
#pragma unroll
            for (int c = 0; c < CPG; ++c) {
                float z = (v[k][c] - mean) * wc[c] + bc[c];
                if (res) z += r8[c];
                o8[c] = gelu_f(z);
            }
            cl_store8(y + base + (size_t)pos * C, o8);
        }
    }
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

template <int KMAX>
__global__ __launch_bounds__(256) void pmx_gn8cl_gelu_bwd_kernel(const __hip_bfloat16 *__restrict__ h, const __hip_bfloat16 *__restrict__ res,
                                                                 const __hip_bfloat16 *__restrict__ dy, const float *__restrict__ w,
                                                                 const float *__restrict__ b, const float *__restrict__ mean_in,
                                                                 const float *__restrict__ rstd_in, __hip_bfloat16 *__restrict__ dh,
                                                                 __hip_bfloat16 *__restrict__ dres, float *__restrict__ partial, long rows,
                                                                 int groups, int HW)
{
    constexpr int CPG = 8;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int g = (int)(row % groups);
    const long n = row / groups;
    const int C = groups * CPG;
    const size_t base = (size_t)n * HW * C + (size_t)g * CPG;
    const float mean = mean_in[row], rstd = rstd_in[row];
    float wc[CPG], bc[CPG], sw[CPG], sb[CPG];
#pragma unroll
    for (int c = 0; c < CPG; ++c) { wc[c] = w[g * CPG + c]; bc[c] = b[g * CPG + c]; sw[c] = 0.f; sb[c] = 0.f; }
    float xh[KMAX][CPG], gz[KMAX][CPG];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int pos = lane + 64 * k;
#pragma unroll
        for (int c = 0; c < CPG; ++c) { xh[k][c] = 0.f; gz[k][c] = 0.f; }
        if (pos < HW) {
            float h8[CPG], r8[CPG], d8[CPG], dz8[CPG];
            cl_load8(h + base + (size_t)pos * C, h8);
            cl_load8(dy + base + (size_t)pos * C, d8);
            if (res) cl_load8(res + base + (size_t)pos * C, r8);
#pragma unroll
            for (int c = 0; c < CPG; ++c) {
                const float x = (h8[c] - mean) * rstd;
                float z = x * wc[c] + bc[c];
                if (res) z += r8[c];
                const float dz = d8[c] * gelu_grad_f(z);
                dz8[c] = dz;
                sw[c] += dz * x; sb[c] += dz;
                const float gg = dz * wc[c];
                xh[k][c] = x; gz[k][c] = gg;
                c1 += gg; c2 += gg * x;
            }
            if (dres) cl_store8(dres + base + (size_t)pos * C, dz8);
        }
    }
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
        const float a = wave_sum(sw[c]), bb = wave_sum(sb[c]);
        if (lane == 0) { partial[((size_t)row * CPG + c) * 2] = a; partial[((size_t)row * CPG + c) * 2 + 1] = bb; }
    }
    const float inv = 1.0f / (float)(CPG * HW);
    c1 = wave_sum(c1) * inv; c2 = wave_sum(c2) * inv;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int pos = lane + 64 * k;
        if (pos < HW) {
            float o8[CPG];
#pragma unroll
            for (int c = 0; c < CPG; ++c) o8[c] = rstd * (gz[k][c] - c1 - xh[k][c] * c2);
            cl_store8(dh + base + (size_t)pos * C, o8);
        }
    }
}

// channels-last (bfloat16 only): tensors are [B][H*W][groups*8] in memory
extern "C" int pmx_gn8cl_gelu_forward(const void *h, const void *res, const float *w, const float *b, void *y, float *mean, float *rstd,
                                      int64_t B, int32_t groups, int32_t HW, float eps, void *stream)
{
    if (!h || !w || !b || !y || !mean || !rstd || B < 0 || groups < 1 || HW < 1 || HW > 1024) return PMX_ERR_INVALID;
    const long rows = (long)B * groups;
    if (rows == 0) return PMX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)((rows + 3) / 4);
#define PMX_GNCL_FWD(K) hipLaunchKernelGGL((pmx_gn8cl_gelu_fwd_kernel<K>), dim3(grid), dim3(256), 0, st, (const __hip_bfloat16 *)h, (const __hip_bfloat16 *)res, w, b, (__hip_bfloat16 *)y, mean, rstd, rows, groups, HW, eps)
    if (HW <= 192) PMX_GNCL_FWD(3); else if (HW <= 448) PMX_GNCL_FWD(7); else PMX_GNCL_FWD(16);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

extern "C" int pmx_gn8cl_gelu_backward(const void *h, const void *res, const void *dy, const float *w, const float *b, const float *mean,
                                       const float *rstd, void *dh, void *dres, float *partial, int64_t B, int32_t groups, int32_t HW,
                                       void *stream)
{
    if (!h || !dy || !w || !b || !mean || !rstd || !dh || !partial || B < 0 || groups < 1 || HW < 1 || HW > 1024) return PMX_ERR_INVALID;
    if ((res == nullptr) != (dres == nullptr)) return PMX_ERR_INVALID;
    const long rows = (long)B * groups;
    if (rows == 0) return PMX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)((rows + 3) / 4);
#define PMX_GNCL_BWD(K) hipLaunchKernelGGL((pmx_gn8cl_gelu_bwd_kernel<K>), dim3(grid), dim3(256), 0, st, (const __hip_bfloat16 *)h, (const __hip_bfloat16 *)res, (const __hip_bfloat16 *)dy, w, b, mean, rstd, (__hip_bfloat16 *)dh, (__hip_bfloat16 *)dres, partial, rows, groups, HW)
    if (HW <= 192) PMX_GNCL_BWD(3); else if (HW <= 448) PMX_GNCL_BWD(7); else PMX_GNCL_BWD(16);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------------
// The minibatch objective of pacman_mappo_resnet.py:571-585 and its gradient with respect to the network outputs, as ONE
// workgroup: log-softmax / entropy of the 5 logits, log-probability of the taken action, advantage normalisation with the
// unbiased standard deviation of THIS minibatch (:577), clipped surrogate, value loss, entropy bonus, clip fraction.  In torch
// this is ~60 small kernels forward and backward -- a fifth of the launches of the 512-sample optimizer step, which is a
// chain of ~5 us kernels.  Sums are accumulated in float64 and reduced through LDS; the per-sample arithmetic is float32 in
// the reference's order.  Gradients follow torch's rules: torch.min splits the gradient evenly on a tie (which is the whole
// unclipped region, where clamp is the identity with gradient 1), clamp passes it on the closed interval.
//   logits [B][5] (float32 or bfloat16), values [BV] float32 with BV == B or, for paired minibatches, BV == B / 2 (rows 2k and
//   2k + 1 share value k); act int64; old_logp, adv, ret float32 [B].
//   stats[0..4] = pg, vl, entropy, clip_frac, loss;  dlogits [B][5] in the logits' type, dvalues [BV] float32 (of loss).
// ---------------------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ double block_sum_1024(double v, double *red)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();                       // red is reused from call to call
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    return t;
}
template <typename LT> __device__ __forceinline__ float logit_load(const LT *p);
template <> __device__ __forceinline__ float logit_load<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float logit_load<__hip_bfloat16>(const __hip_bfloat16 *p) { return __bfloat162float(*p); }
template <typename LT> __device__ __forceinline__ void logit_store(LT *p, float v);
template <> __device__ __forceinline__ void logit_store<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void logit_store<__hip_bfloat16>(__hip_bfloat16 *p, float v) { *p = __float2bfloat16(v); }
}  // namespace

template <typename LT>
__global__ __launch_bounds__(1024) void pmx_ppo_loss_kernel(const LT *__restrict__ logits, const float *__restrict__ values,
                                                           const int64_t *__restrict__ act, const float *__restrict__ old_logp,
                                                           const float *__restrict__ adv, const float *__restrict__ ret, int B, int BV,
                                                           const float *clip_dev, const float *ent_dev, float clip_host, float ent_host,
                                                           float vf_coef, float *__restrict__ stats, LT *__restrict__ dlogits,
                                                           float *__restrict__ dvalues)
{
    __shared__ double red[16];
    const float clip_eps = clip_dev ? *clip_dev : clip_host, ent_coef = ent_dev ? *ent_dev : ent_host;
    const int tid = threadIdx.x;
    // advantage statistics: mean, then the unbiased variance around it (two passes, as torch.std)
    double s = 0.0;
    for (int i = tid; i < B; i += 1024) s += (double)adv[i];
    const double mean = block_sum_1024(s, red) / (double)B;
    double ss = 0.0;
    for (int i = tid; i < B; i += 1024) { const double d = (double)adv[i] - mean; ss += d * d; }
    const double var = block_sum_1024(ss, red) / (double)(B > 1 ? B - 1 : 1);
    const float a_mean = (float)mean, a_den = (float)sqrt(var) + 1e-8f;
    const float invB = 1.0f / (float)B;
    const bool paired = BV * 2 == B;
    double pg_s = 0.0, ent_s = 0.0, clip_s = 0.0, vl_s = 0.0;
    for (int i = tid; i < B; i += 1024) {
        float z[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) z[k] = logit_load<LT>(logits + (size_t)i * 5 + k);
        const float zmax = fmaxf(fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], z[3])), z[4]);
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) se += expf(z[k] - zmax);
        const float lse = zmax + logf(se);
        float nrm[5], p[5], H = 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            nrm[k] = z[k] - lse;
            p[k] = expf(nrm[k]);
            H -= fmaxf(nrm[k], -3.4028234663852886e38f) * p[k];
        }
        const int a = (int)act[i];
        const float logp = a == 0 ? nrm[0] : (a == 1 ? nrm[1] : (a == 2 ? nrm[2] : (a == 3 ? nrm[3] : nrm[4])));
        const float na = (adv[i] - a_mean) / a_den;
        const float ratio = expf(logp - old_logp[i]);
        const float rc = fminf(fmaxf(ratio, 1.0f - clip_eps), 1.0f + clip_eps);
        const float t1 = na * ratio, t2 = na * rc;
        pg_s += (double)fminf(t1, t2);
        ent_s += (double)H;
        clip_s += fabsf(ratio - 1.0f) > clip_eps ? 1.0 : 0.0;
        // d min(t1, t2) / d ratio
        const float w1 = t1 < t2 ? 1.0f : (t1 == t2 ? 0.5f : 0.0f), w2 = t2 < t1 ? 1.0f : (t1 == t2 ? 0.5f : 0.0f);
        const float in_range = (ratio >= 1.0f - clip_eps && ratio <= 1.0f + clip_eps) ? 1.0f : 0.0f;
        const float dmin_dr = na * (w1 + w2 * in_range);
        const float dlogp = -invB * dmin_dr * ratio;                      // d loss / d logp_i
        const float dH = -ent_coef * invB;                                // d loss / d H_i
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float g = dlogp * ((k == a ? 1.0f : 0.0f) - p[k]) + dH * (-p[k] * (nrm[k] + H));
            logit_store<LT>(dlogits + (size_t)i * 5 + k, g);
        }
        const float v = values[paired ? (i >> 1) : i];
        const float dv = v - ret[i];
        vl_s += (double)(dv * dv);
        if (!paired) dvalues[i] = vf_coef * invB * dv;
    }
    if (paired) {
        for (int j = tid; j < BV; j += 1024) {
            const float v = values[j];
            dvalues[j] = vf_coef * invB * ((v - ret[2 * j]) + (v - ret[2 * j + 1]));
        }
    }
    const double pg = -block_sum_1024(pg_s, red) / (double)B;
    const double ent = block_sum_1024(ent_s, red) / (double)B;
    const double cf = block_sum_1024(clip_s, red) / (double)B;
    const double vl = 0.5 * block_sum_1024(vl_s, red) / (double)B;
    if (tid == 0) {
        stats[0] = (float)pg; stats[1] = (float)vl; stats[2] = (float)ent; stats[3] = (float)cf;
        stats[4] = (float)pg + vf_coef * (float)vl - ent_coef * (float)ent;
    }
}

extern "C" int pmx_ppo_loss(const void *logits_dev, int32_t logits_bf16, const float *values_dev, const int64_t *act_dev,
                            const float *old_logp_dev, const float *adv_dev, const float *ret_dev, int32_t B, int32_t BV,
                            const float *clip_eps_dev, const float *ent_coef_dev, float clip_eps, float ent_coef, float vf_coef,
                            float *stats_dev, void *dlogits_dev, float *dvalues_dev, void *stream)
{
    if (!logits_dev || !values_dev || !act_dev || !old_logp_dev || !adv_dev || !ret_dev || !stats_dev || !dlogits_dev || !dvalues_dev)
        return PMX_ERR_INVALID;
    if (B < 1 || (BV != B && BV * 2 != B)) return PMX_ERR_INVALID;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (logits_bf16)
        hipLaunchKernelGGL(pmx_ppo_loss_kernel<__hip_bfloat16>, dim3(1), dim3(1024), 0, st, (const __hip_bfloat16 *)logits_dev, values_dev, act_dev,
                           old_logp_dev, adv_dev, ret_dev, (int)B, (int)BV, clip_eps_dev, ent_coef_dev, clip_eps, ent_coef, vf_coef, stats_dev,
                           (__hip_bfloat16 *)dlogits_dev, dvalues_dev);
    else
        hipLaunchKernelGGL(pmx_ppo_loss_kernel<float>, dim3(1), dim3(1024), 0, st, (const float *)logits_dev, values_dev, act_dev, old_logp_dev,
                           adv_dev, ret_dev, (int)B, (int)BV, clip_eps_dev, ent_coef_dev, clip_eps, ent_coef, vf_coef, stats_dev,
                           (float *)dlogits_dev, dvalues_dev);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------------
// Minibatch assembly for the replayed optimizer step: up to 8 row gathers dst[r] = src[idx[r / m] * m + r % m] in one launch
// (m = rows per index: 2 for the per-learner tensors of a paired minibatch, whose index names an env-tick pair), straight into
// the graph's static input tensors.  In torch this was six index kernels, the index arithmetic and six copies per step.
// ---------------------------------------------------------------------------------------------------------------
struct PmxGatherArgs {
    const char *src[8];
    char *dst[8];
    const int64_t *idx[8];
    int32_t row_bytes[8], rows_per_index[8];
    int64_t n_rows[8];
    float *floats_dst;                  // optional: up to 8 scalars written by the same launch (pmx_gather_rows_set_floats)
    float floats[8];
    int32_t n_floats;
};
__global__ __launch_bounds__(256) void pmx_gather_rows_kernel(PmxGatherArgs a)
{
    const int t = blockIdx.y;
    if (blockIdx.x == 0 && t == 0 && (int)threadIdx.x < a.n_floats) a.floats_dst[threadIdx.x] = a.floats[threadIdx.x];
    const int rb = a.row_bytes[t], m = a.rows_per_index[t];
    const char *src = a.src[t];
    char *dst = a.dst[t];
    const int64_t *idx = a.idx[t];
    if ((rb & 15) == 0) {
        const int per_row = rb >> 4;
        const int64_t total = a.n_rows[t] * per_row;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
            const int64_t r = i / per_row;
            const int c = (int)(i - r * per_row);
            const int64_t s = idx[r / m] * m + r % m;
            reinterpret_cast<uint4 *>(dst + r * rb)[c] = reinterpret_cast<const uint4 *>(src + s * rb)[c];
        }
    } else {
        const int per_row = rb >> 2;                                      // rows of 4-byte words (checked by the host side)
        const int64_t total = a.n_rows[t] * per_row;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
            const int64_t r = i / per_row;
            const int c = (int)(i - r * per_row);
            const int64_t s = idx[r / m] * m + r % m;
            reinterpret_cast<uint32_t *>(dst + r * rb)[c] = reinterpret_cast<const uint32_t *>(src + s * rb)[c];
        }
    }
}

extern "C" int pmx_gather_rows(int32_t n, const void *const *src_dev, void *const *dst_dev, const int64_t *const *idx_dev,
                               const int32_t *row_bytes, const int32_t *rows_per_index, const int64_t *n_rows, void *stream)
{
    return pmx_gather_rows_set_floats(n, src_dev, dst_dev, idx_dev, row_bytes, rows_per_index, n_rows, nullptr, nullptr, 0, stream);
}
extern "C" int pmx_gather_rows_set_floats(int32_t n, const void *const *src_dev, void *const *dst_dev, const int64_t *const *idx_dev,
                                          const int32_t *row_bytes, const int32_t *rows_per_index, const int64_t *n_rows,
                                          float *floats_dst_dev, const float *values, int32_t n_values, void *stream)
{
    if (n < 1 || n > 8 || !src_dev || !dst_dev || !idx_dev || !row_bytes || !rows_per_index || !n_rows) return PMX_ERR_INVALID;
    if (n_values < 0 || n_values > 8 || (n_values > 0 && (!floats_dst_dev || !values))) return PMX_ERR_INVALID;
    PmxGatherArgs a;
    memset(&a, 0, sizeof(a));
    a.floats_dst = floats_dst_dev;
    a.n_floats = n_values;
    for (int i = 0; i < n_values; ++i) a.floats[i] = values[i];
    int64_t most = 0;
    for (int t = 0; t < n; ++t) {
        if (!src_dev[t] || !dst_dev[t] || !idx_dev[t] || row_bytes[t] < 4 || (row_bytes[t] & 3) || rows_per_index[t] < 1 || n_rows[t] < 0)
            return PMX_ERR_INVALID;
        if ((row_bytes[t] & 15) == 0 && (((uintptr_t)src_dev[t] | (uintptr_t)dst_dev[t]) & 15)) return PMX_ERR_INVALID;
        a.src[t] = (const char *)src_dev[t]; a.dst[t] = (char *)dst_dev[t]; a.idx[t] = idx_dev[t];
        a.row_bytes[t] = row_bytes[t]; a.rows_per_index[t] = rows_per_index[t]; a.n_rows[t] = n_rows[t];
        const int64_t units = n_rows[t] * (row_bytes[t] >> ((row_bytes[t] & 15) == 0 ? 4 : 2));
        most = units > most ? units : most;
    }
    if (most == 0 && n_values == 0) return PMX_OK;
    int64_t blocks = (most + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pmx_gather_rows_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

// up to 8 float32 values written to consecutive device words in one launch (the scalars a replayed graph reads: they travel
// as kernel arguments, so no host staging buffer is involved)
struct PmxFloats8 { float v[8]; };
__global__ void pmx_set_floats_kernel(float *dst, PmxFloats8 f, int n)
{
    if ((int)threadIdx.x < n) dst[threadIdx.x] = f.v[threadIdx.x];
}
extern "C" int pmx_set_floats(float *dst_dev, const float *values, int32_t n, void *stream)
{
    if (!dst_dev || !values || n < 1 || n > 8) return PMX_ERR_INVALID;
    PmxFloats8 f;
    for (int i = 0; i < 8; ++i) f.v[i] = i < n ? values[i] : 0.f;
    hipLaunchKernelGGL(pmx_set_floats_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), dst_dev, f, (int)n);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------------
// The tail of the optimizer step on the flat buffers (pacman_mappo_resnet.py:587-595): clip_grad_norm_(0.5) -> Adam(eps 1e-5,
// no weight decay, no amsgrad) -> EMA 0.995, as two launches instead of ~18 elementwise / reduction kernels over the 2.6 M
// parameters.  Launch 1: per-block sums of g^2 in float64.  Launch 2: every block adds those partial sums itself (<= 1024 of
// them), so the global norm and the clip factor need no third launch; then, per element, torch's formulas in float32:
//   g *= min(1, max_norm / (norm + 1e-6));  m = m + (1 - b1)(g - m);  v = b2 v + (1 - b2) g g;
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps);  ema = decay * ema + (1 - decay) * p.
// lr / bc1 and 1 / sqrt(bc2) come from two device floats when `sc_dev` is given (graph replay), else from the host values.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pmx_sqsum_partial_kernel(const float *__restrict__ g, long n, double *__restrict__ partial)
{
    __shared__ double red[4];
    double s = 0.0;
    const long n4 = n >> 2;
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = g4[i];
        s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s += (double)v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void pmx_adam_ema_kernel(float *__restrict__ g, float *__restrict__ p, float *__restrict__ m,
                                                           float *__restrict__ v, float *__restrict__ ema, long n,
                                                           const double *__restrict__ partial, int n_partial, const float *sc_dev,
                                                           float lr_bc1_host, float rsqrt_bc2_host, float b1, float b2, float eps,
                                                           float max_norm, float decay, float *__restrict__ norm_out,
                                                           uint16_t *__restrict__ p_bf16, const float *__restrict__ reports5,
                                                           float *__restrict__ report_sums6)
{
    __shared__ double red[4];
    __shared__ float s_scale;
    double s = 0.0;
    for (int i = threadIdx.x; i < n_partial; i += 256) s += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
        s_scale = fminf(max_norm / (norm + 1e-6f), 1.0f);
        if (blockIdx.x == 0 && norm_out) *norm_out = norm;
        if (blockIdx.x == 0 && report_sums6) {             // running sums of the step's reports, for the caller's averages
            if (reports5) {
#pragma unroll
                for (int k = 0; k < 5; ++k) report_sums6[k] += reports5[k];
            }
            report_sums6[5] += norm;
        }
    }
    __syncthreads();
    const float scale = s_scale;
    const float lr_bc1 = sc_dev ? sc_dev[0] : lr_bc1_host, rsqrt_bc2 = sc_dev ? sc_dev[1] : rsqrt_bc2_host;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * scale;
        const float mi = m[i] + (1.0f - b1) * (gi - m[i]);
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
        const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
        const float pi = p[i] - lr_bc1 * (mi / denom);
        g[i] = gi; m[i] = mi; v[i] = vi; p[i] = pi;
        ema[i] = ema[i] * decay + pi * (1.0f - decay);
        if (p_bf16) p_bf16[i] = (uint16_t)pmx_f2bf(pi);     // the bfloat16 copy the library GEMMs read (round to nearest even)
    }
}

extern "C" int pmx_clip_adam_ema(float *grad_dev, float *param_dev, float *exp_avg_dev, float *exp_avg_sq_dev, float *ema_dev, int64_t n,
                                 double *scratch_dev, const float *scalars_dev, float lr_over_bc1, float rsqrt_bc2, float beta1, float beta2,
                                 float eps, float max_norm, float ema_decay, float *norm_out_dev, void *stream)
{
    return pmx_clip_adam_ema_tail(grad_dev, param_dev, exp_avg_dev, exp_avg_sq_dev, ema_dev, n, scratch_dev, scalars_dev, lr_over_bc1, rsqrt_bc2,
                                  beta1, beta2, eps, max_norm, ema_decay, norm_out_dev, nullptr, nullptr, nullptr, stream);
}
extern "C" int pmx_clip_adam_ema_tail(float *grad_dev, float *param_dev, float *exp_avg_dev, float *exp_avg_sq_dev, float *ema_dev, int64_t n,
                                      double *scratch_dev, const float *scalars_dev, float lr_over_bc1, float rsqrt_bc2, float beta1,
                                      float beta2, float eps, float max_norm, float ema_decay, float *norm_out_dev, void *param_bf16_dev,
                                      const float *reports5_dev, float *report_sums6_dev, void *stream)
{
    if (!grad_dev || !param_dev || !exp_avg_dev || !exp_avg_sq_dev || !ema_dev || !scratch_dev || n < 1) return PMX_ERR_INVALID;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > PMX_OPT_PARTIALS) blocks = PMX_OPT_PARTIALS;
    hipLaunchKernelGGL(pmx_sqsum_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float *)grad_dev, (long)n, scratch_dev);
    int64_t b2k = (n + 255) / 256;
    if (b2k > 2048) b2k = 2048;
    hipLaunchKernelGGL(pmx_adam_ema_kernel, dim3((unsigned)b2k), dim3(256), 0, st, grad_dev, param_dev, exp_avg_dev, exp_avg_sq_dev, ema_dev, (long)n,
                       (const double *)scratch_dev, (int)blocks, scalars_dev, lr_over_bc1, rsqrt_bc2, beta1, beta2, eps, max_norm, ema_decay,
                       norm_out_dev, (uint16_t *)param_bf16_dev, reports5_dev, report_sums6_dev);
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------------
// The parameter gradients of one optimizer step -- float32 or bfloat16 tensors, wherever autograd left them -- into the flat
// float32 bucket in ONE launch (torch.cat needs one dtype, i.e. a cast kernel per bfloat16 gradient first, and three launches
// for ~55 inputs).  blockIdx.y = tensor.
// ---------------------------------------------------------------------------------------------------------------
#define PMX_FLATTEN_MAX 64
struct PmxFlattenArgs {
    const void *src[PMX_FLATTEN_MAX];
    int64_t off[PMX_FLATTEN_MAX];
    int32_t count[PMX_FLATTEN_MAX];
    int32_t rows[PMX_FLATTEN_MAX];                 // > 0: the source is rows 1 .. rows of a partial-row buffer (stride floats apart) still to be added
    int32_t stride[PMX_FLATTEN_MAX];
    uint8_t bf16[PMX_FLATTEN_MAX];
};
__global__ __launch_bounds__(256) void pmx_flatten_f32_kernel(PmxFlattenArgs a, float *__restrict__ dst)
{
    const int t = blockIdx.y;
    const int n = a.count[t];
    float *d = dst + a.off[t];
    if (a.rows[t] > 0) {
        // the second stage of a gradient reduction, done here instead of by a launch of its own behind the backward kernel: s points
        // into row 0 of the buffer, the partial rows follow `stride` floats apart.  As pmx_sum_rows_kernel does it: a block takes 32
        // columns with 8 slices of the rows side by side (four loads in flight per thread) and adds the slices through LDS.
        __shared__ float part[8][33];
        const float *s = reinterpret_cast<const float *>(a.src[t]);
        const int rows = a.rows[t];
        const size_t st = (size_t)a.stride[t];
        const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
        for (int c0 = blockIdx.x * 32; c0 < n; c0 += gridDim.x * 32) {             // (uniform over the block)
            const int i = c0 + c;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            if (i < n) {
                int r = 1 + sl;
                for (; r + 24 <= rows; r += 32) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] += s[(size_t)(r + 8 * k) * st + i];
                }
                for (; r <= rows; r += 8) acc[0] += s[(size_t)r * st + i];
            }
            part[sl][c] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            __syncthreads();
            if (sl == 0 && i < n) {
                float tt = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) tt += part[k][c];
                d[i] = tt;
            }
            __syncthreads();
        }
    } else if (a.bf16[t]) {
        const uint16_t *s = reinterpret_cast<const uint16_t *>(a.src[t]);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d[i] = __uint_as_float((uint32_t)s[i] << 16);
    } else {
        const float *s = reinterpret_cast<const float *>(a.src[t]);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d[i] = s[i];
    }
}
extern "C" int pmx_flatten_to_f32(int32_t n, const void *const *src_dev, const uint8_t *src_is_bf16, const int64_t *dst_offset,
                                  const int32_t *count, float *dst_dev, void *stream)
{
    return pmx_flatten_sum_to_f32(n, src_dev, src_is_bf16, nullptr, nullptr, dst_offset, count, dst_dev, stream);
}
extern "C" int pmx_flatten_sum_to_f32(int32_t n, const void *const *src_dev, const uint8_t *src_is_bf16, const int32_t *partial_rows,
                                      const int32_t *row_stride, const int64_t *dst_offset, const int32_t *count, float *dst_dev,
                                      void *stream)
{
    if (n < 1 || !src_dev || !src_is_bf16 || !dst_offset || !count || !dst_dev || ((partial_rows != nullptr) != (row_stride != nullptr)))
        return PMX_ERR_INVALID;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int base = 0; base < n; base += PMX_FLATTEN_MAX) {
        const int m = n - base < PMX_FLATTEN_MAX ? n - base : PMX_FLATTEN_MAX;
        PmxFlattenArgs a;
        memset(&a, 0, sizeof(a));
        int most = 0;
        for (int t = 0; t < m; ++t) {
            if (!src_dev[base + t] || count[base + t] < 0 || dst_offset[base + t] < 0) return PMX_ERR_INVALID;
            a.src[t] = src_dev[base + t]; a.off[t] = dst_offset[base + t]; a.count[t] = count[base + t]; a.bf16[t] = src_is_bf16[base + t] ? 1 : 0;
            if (partial_rows && partial_rows[base + t] > 0) {
                if (a.bf16[t] || row_stride[base + t] < count[base + t]) return PMX_ERR_INVALID;
                a.rows[t] = partial_rows[base + t]; a.stride[t] = row_stride[base + t];
            }
            most = count[base + t] > most ? count[base + t] : most;
        }
        int blocks = (most + 2047) / 2048;                     // ~8 elements per thread for the largest tensor
        if (blocks < 1) blocks = 1;
        if (blocks > 512) blocks = 512;
        hipLaunchKernelGGL(pmx_flatten_f32_kernel, dim3((unsigned)blocks, (unsigned)m), dim3(256), 0, st, a, dst_dev);
    }
    return hipGetLastError() == hipSuccess ? PMX_OK : PMX_ERR_HIP;
}
