// Convolutional representation network h (gfx950): the pieces of RepresentationNetwork.forward
// (src/neural_networks/neural_network.py:83-102) and of its building blocks ConvLayer, PoolLayer
// and ResBlock (src/neural_networks/network_builder.py:32-151) for G games at once.
//
// Activations are channels-last ([G][H][W][C], "NHWC"): a pixel's channels are contiguous, so a
// convolution is ONE matrix product per layer on the fp32 matrix cores -- rows = output pixels of
// all games, columns = output channels, reduction = (tap, input channel) -- whose A operand is
// gathered pixel by pixel through an index table (mz_conv_index + mz_wide_forward_gather in
// csrc/mz_tower.hip; nothing like an im2col matrix is ever stored).  Convolutions whose input
// channel count is no multiple of 64, or with a handful of output channels, run on the direct
// kernel here.
//
// BatchNorm: the reference never leaves train mode and calls h with ONE game per call
// (src/training_data_generator.py:117), so every BatchNorm2d normalises with the mean and biased
// variance over the H x W positions of that game alone.  mz_bn_stats forms them per (game, channel)
// in float64 and folds them with the affine parameters into a scale and a shift; mz_bn_apply applies
// them together with the residual sum and the activation of a ResBlock.
#include <algorithm>

#include "mz_common.h"

namespace {

constexpr int kCT = 256;

__device__ __forceinline__ float act_f(float v, int act) {
    switch (act) {
        case MZ_ACT_RELU: return v > 0.0f ? v : 0.0f;
        case MZ_ACT_TANH: return tanhf(v);
        case MZ_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        default: return v;
    }
}

// idx[tap][row]: input pixel (g * H + iy) * W + ix of output pixel `row` = (g, oy, ox) under tap
// (ky, kx), or -1 where the tap falls into the zero padding.
__global__ void __launch_bounds__(kCT)
k_conv_index(int32_t *__restrict__ idx, int G, int H, int W, int KS, int stride, int pad, int Ho, int Wo) {
    const long long rows = (long long)G * Ho * Wo;
    const long long total = rows * KS * KS;
    for (long long i = (long long)blockIdx.x * kCT + threadIdx.x; i < total; i += (long long)gridDim.x * kCT) {
        const int tap = (int)(i / rows);
        const long long r = i - (long long)tap * rows;
        const int ky = tap / KS, kx = tap - ky * KS;
        const int g = (int)(r / (Ho * Wo));
        const int rem = (int)(r - (long long)g * Ho * Wo);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
        idx[i] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? (g * H + iy) * W + ix : -1;
    }
}

// [G][C][P] -> [G][P][C] (to_nhwc) or back, through a 32 x 32 LDS tile.
__global__ void __launch_bounds__(kCT)
k_transpose(const float *__restrict__ in, float *__restrict__ out, int R, int Cc) {
    // per image: in [R][Cc] -> out [Cc][R]
    __shared__ float tile[32][33];
    const long long img = blockIdx.z;
    const float *src = in + img * (long long)R * Cc;
    float *dst = out + img * (long long)R * Cc;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < R && c < Cc) ? src[(long long)r * Cc + c] : 0.0f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < Cc && r < R) dst[(long long)c * R + r] = tile[tx][j];
    }
}

// The same through a 64 x 64 tile: 256-byte runs on both sides (the network input of h, [G][128][H*W]: 32-float
// runs leave a third of the memory rate unused).
__global__ void __launch_bounds__(kCT)
k_transpose64(const float *__restrict__ in, float *__restrict__ out, int R, int Cc) {
    __shared__ float tile[64][65];
    const long long img = blockIdx.z;
    const float *src = in + img * (long long)R * Cc;
    float *dst = out + img * (long long)R * Cc;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    for (int j = ty; j < 64; j += 4) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < R && c < Cc) ? src[(long long)r * Cc + c] : 0.0f;
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
        const int c = c0 + j, r = r0 + tx;
        if (c < Cc && r < R) dst[(long long)c * R + r] = tile[tx][j];
    }
}

// Per (game, channel): mean and biased variance over the P positions (float64 sums), folded with the
// BatchNorm affine parameters: scale = w / sqrt(var + eps), shift = b - mean * scale.
// x [G][P][C].  One workgroup per (game, 64 channels); 4 position groups x 64 channels.
__global__ void __launch_bounds__(kCT)
k_bn_stats(const float *__restrict__ x, int P, int C, const float *__restrict__ weight, const float *__restrict__ bias,
           float eps, float *__restrict__ scale, float *__restrict__ shift, float *__restrict__ mean_out,
           float *__restrict__ var_out) {
    __shared__ double s_sum[4][64], s_sq[4][64];
    const int g = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), pg = threadIdx.x >> 6;
    double sum = 0.0, sq = 0.0;
    if (c < C) {
        const float *px = x + (long long)g * P * C + c;
        for (int p = pg; p < P; p += 4) {
            const double v = (double)px[(long long)p * C];
            sum += v;
            sq += v * v;
        }
    }
    s_sum[pg][threadIdx.x & 63] = sum;
    s_sq[pg][threadIdx.x & 63] = sq;
    __syncthreads();
    if (pg == 0 && c < C) {
        const int l = threadIdx.x;
        const double S = ((s_sum[0][l] + s_sum[1][l]) + s_sum[2][l]) + s_sum[3][l];
        const double Q = ((s_sq[0][l] + s_sq[1][l]) + s_sq[2][l]) + s_sq[3][l];
        const double mean = S / P;
        double var = Q / P - mean * mean;
        if (var < 0.0) var = 0.0;
        const double sc = (double)weight[c] / sqrt(var + (double)eps);
        const long long o = (long long)g * C + c;
        scale[o] = (float)sc;
        shift[o] = (float)((double)bias[c] - mean * sc);
        if (mean_out) mean_out[o] = (float)mean;
        if (var_out) var_out[o] = (float)var;
    }
}

// y = act(x * scale[g][c] + shift[g][c] + residual) where residual = 0, r, or r * rscale[g][c] + rshift[g][c].
__global__ void __launch_bounds__(kCT)
k_bn_apply(const float *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
           const float *__restrict__ res, const float *__restrict__ rscale, const float *__restrict__ rshift, int act,
           float *__restrict__ y, long long total, int PC, int C) {
    for (long long i = (long long)blockIdx.x * kCT + threadIdx.x; i < total; i += (long long)gridDim.x * kCT) {
        const int g = (int)(i / PC);
        const int c = (int)(i % C);
        const long long o = (long long)g * C + c;
        float v = x[i] * scale[o] + shift[o];
        if (res) v += rscale ? (res[i] * rscale[o] + rshift[o]) : res[i];
        y[i] = act_f(v, act);
    }
}

// Max / average pooling with a k x k window, stride s, no padding (torch's floor mode), NHWC.
__global__ void __launch_bounds__(kCT)
k_pool(const float *__restrict__ x, int H, int W, int C, int k, int s, int Ho, int Wo, int is_max, float *__restrict__ y,
       long long total) {
    for (long long i = (long long)blockIdx.x * kCT + threadIdx.x; i < total; i += (long long)gridDim.x * kCT) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int g = (int)(r / Ho);
        float acc = is_max ? -__builtin_inff() : 0.0f;
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) {
                const float v = x[(((long long)g * H + oy * s + dy) * W + ox * s + dx) * C + c];
                acc = is_max ? fmaxf(acc, v) : acc + v;
            }
        y[i] = is_max ? acc : acc / (float)(k * k);
    }
}

// adaptive_avg_pool2d (torch bins: [floor(i H / h), ceil((i + 1) H / h))) from NHWC to the NCHW
// latent the search reads ([G][C][h][w]).
__global__ void __launch_bounds__(kCT)
k_adaptive_avg_to_nchw(const float *__restrict__ x, int H, int W, int C, int h, int w, float *__restrict__ y,
                       long long total) {
    for (long long i = (long long)blockIdx.x * kCT + threadIdx.x; i < total; i += (long long)gridDim.x * kCT) {
        // i enumerates (g, oy, ox, c) so that reads are coalesced along c
        const int c = (int)(i % C);
        long long r = i / C;
        const int ox = (int)(r % w);
        r /= w;
        const int oy = (int)(r % h);
        const int g = (int)(r / h);
        const int y0 = (oy * H) / h, y1 = ((oy + 1) * H + h - 1) / h;
        const int x0 = (ox * W) / w, x1 = ((ox + 1) * W + w - 1) / w;
        float acc = 0.0f;
        for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx) acc += x[(((long long)g * H + yy) * W + xx) * C + c];
        y[(((long long)g * C + c) * h + oy) * w + ox] = acc / (float)((y1 - y0) * (x1 - x0));
    }
}

// Direct convolution, NHWC, weights [Cout][KS][KS][Cin]: one lane per output value.  For layers the
// gathered matrix product does not cover (input channels no multiple of 64, a handful of outputs).
__global__ void __launch_bounds__(kCT)
k_conv_direct(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias, int H, int W,
              int Cin, int Cout, int KS, int stride, int pad, int Ho, int Wo, int act, float *__restrict__ y,
              long long total) {
    for (long long i = (long long)blockIdx.x * kCT + threadIdx.x; i < total; i += (long long)gridDim.x * kCT) {
        const int co = (int)(i % Cout);
        long long r = i / Cout;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int g = (int)(r / Ho);
        float acc = 0.0f;
        for (int ky = 0; ky < KS; ++ky) {
            const int iy = oy * stride + ky - pad;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < KS; ++kx) {
                const int ix = ox * stride + kx - pad;
                if (ix < 0 || ix >= W) continue;
                const float *px = x + (((long long)g * H + iy) * W + ix) * Cin;
                const float *pw = w + (((long long)co * KS + ky) * KS + kx) * Cin;
                for (int ci = 0; ci < Cin; ++ci) acc = fmaf(px[ci], pw[ci], acc);
            }
        }
        if (bias) acc += bias[co];
        y[i] = act_f(acc, act);
    }
}

// torch weight [Cout][Cin][KS][KS] -> [Cout][KS][KS][Cin] (the reduction order of both conv paths)
__global__ void __launch_bounds__(kCT)
k_conv_weight_kmajor(const float *__restrict__ w, int Cout, int Cin, int KK, float *__restrict__ out) {
    const long long total = (long long)Cout * Cin * KK;
    for (long long i = (long long)blockIdx.x * kCT + threadIdx.x; i < total; i += (long long)gridDim.x * kCT) {
        const int ci = (int)(i % Cin);
        long long r = i / Cin;
        const int t = (int)(r % KK);
        const int co = (int)(r / KK);
        out[i] = w[((long long)co * Cin + ci) * KK + t];
    }
}

// BatchNorm2d's running statistics after G train-mode calls of batch size 1 in game order
// (src/neural_networks/network_builder.py:94,104,118; torch: running <- (1 - m) running + m stat,
// the variance unbiased, n = P values per call): one lane per channel walks the games.
__global__ void __launch_bounds__(kCT)
k_bn_running(const float *__restrict__ mean, const float *__restrict__ var, int G, int C, int P, float momentum,
             float *__restrict__ running_mean, float *__restrict__ running_var) {
    const int c = blockIdx.x * kCT + threadIdx.x;
    if (c >= C) return;
    float rm = running_mean[c], rv = running_var[c];
    const float unbias = (float)P / (float)(P - 1);
    for (int g = 0; g < G; ++g) {
        rm = (1.0f - momentum) * rm + momentum * mean[(long long)g * C + c];
        rv = (1.0f - momentum) * rv + momentum * (var[(long long)g * C + c] * unbias);
    }
    running_mean[c] = rm;
    running_var[c] = rv;
}

inline int blocks_for(long long total) { return (int)std::min<long long>((total + kCT - 1) / kCT, 1 << 20); }

}  // namespace

extern "C" {

int mz_conv_index(int32_t *idx, int32_t G, int32_t H, int32_t W, int32_t KS, int32_t stride, int32_t pad, int32_t Ho,
                  int32_t Wo, void *stream) {
    if (!idx || G <= 0 || H <= 0 || W <= 0 || KS <= 0 || stride <= 0 || pad < 0 || Ho <= 0 || Wo <= 0)
        return mz::fail(MZ_ERR_ARG, "mz_conv_index: NULL table or a non-positive dimension");
    if ((Ho - 1) * stride + KS - pad > H + pad || (Wo - 1) * stride + KS - pad > W + pad)
        return mz::fail(MZ_ERR_ARG, "mz_conv_index: output %dx%d does not fit input %dx%d (k=%d, stride=%d, pad=%d)", Ho, Wo,
                        H, W, KS, stride, pad);
    if ((long long)G * H * W > 0x7fffffffLL) return mz::fail(MZ_ERR_ARG, "mz_conv_index: more than 2^31-1 input pixels");
    const long long total = (long long)G * Ho * Wo * KS * KS;
    hipLaunchKernelGGL(k_conv_index, dim3(blocks_for(total)), dim3(kCT), 0, (hipStream_t)stream, idx, G, H, W, KS, stride,
                       pad, Ho, Wo);
    return mz::check_launch("k_conv_index");
}

int mz_nchw_to_nhwc(const float *x, int32_t G, int32_t C, int32_t P, float *y, void *stream) {
    if (!x || !y || G <= 0 || C <= 0 || P <= 0) return mz::fail(MZ_ERR_ARG, "mz_nchw_to_nhwc: NULL or non-positive argument");
    if (C >= 64 && P >= 64 && (C + 63) / 64 <= 65535 && G <= 65535) {
        hipLaunchKernelGGL(k_transpose64, dim3((P + 63) / 64, (C + 63) / 64, G), dim3(kCT), 0, (hipStream_t)stream, x, y, C, P);
        return mz::check_launch("k_transpose64");
    }
    hipLaunchKernelGGL(k_transpose, dim3((P + 31) / 32, (C + 31) / 32, G), dim3(kCT), 0, (hipStream_t)stream, x, y, C, P);
    return mz::check_launch("k_transpose");
}

int mz_nhwc_to_nchw(const float *x, int32_t G, int32_t C, int32_t P, float *y, void *stream) {
    if (!x || !y || G <= 0 || C <= 0 || P <= 0) return mz::fail(MZ_ERR_ARG, "mz_nhwc_to_nchw: NULL or non-positive argument");
    hipLaunchKernelGGL(k_transpose, dim3((C + 31) / 32, (P + 31) / 32, G), dim3(kCT), 0, (hipStream_t)stream, x, y, P, C);
    return mz::check_launch("k_transpose");
}

int mz_bn_stats(const float *x, int32_t G, int32_t P, int32_t C, const float *weight, const float *bias, float eps,
                float *scale, float *shift, float *mean, float *var, void *stream) {
    if (!x || !weight || !bias || !scale || !shift || G <= 0 || P <= 0 || C <= 0)
        return mz::fail(MZ_ERR_ARG, "mz_bn_stats: NULL or non-positive argument");
    if (P < 2) return mz::fail(MZ_ERR_ARG, "mz_bn_stats: BatchNorm in train mode needs more than 1 value per channel");
    hipLaunchKernelGGL(k_bn_stats, dim3((C + 63) / 64, G), dim3(kCT), 0, (hipStream_t)stream, x, P, C, weight, bias, eps,
                       scale, shift, mean, var);
    return mz::check_launch("k_bn_stats");
}

int mz_bn_apply(const float *x, const float *scale, const float *shift, const float *residual, const float *res_scale,
                const float *res_shift, int32_t activation, float *y, int32_t G, int32_t P, int32_t C, void *stream) {
    if (!x || !scale || !shift || !y || G <= 0 || P <= 0 || C <= 0 || (res_scale && (!residual || !res_shift)))
        return mz::fail(MZ_ERR_ARG, "mz_bn_apply: NULL or non-positive argument");
    if (activation < MZ_ACT_NONE || activation > MZ_ACT_SIGMOID)
        return mz::fail(MZ_ERR_ARG, "mz_bn_apply: activation %d is not element-wise", activation);
    const long long total = (long long)G * P * C;
    hipLaunchKernelGGL(k_bn_apply, dim3(blocks_for(total)), dim3(kCT), 0, (hipStream_t)stream, x, scale, shift, residual,
                       res_scale, res_shift, activation, y, total, P * C, C);
    return mz::check_launch("k_bn_apply");
}

int mz_bn_running_update(const float *mean, const float *var, int32_t G, int32_t C, int32_t P, float momentum,
                         float *running_mean, float *running_var, void *stream) {
    if (!mean || !var || !running_mean || !running_var || G <= 0 || C <= 0 || P < 2)
        return mz::fail(MZ_ERR_ARG, "mz_bn_running_update: NULL argument, non-positive size or fewer than 2 values per call");
    hipLaunchKernelGGL(k_bn_running, dim3((C + kCT - 1) / kCT), dim3(kCT), 0, (hipStream_t)stream, mean, var, G, C, P,
                       momentum, running_mean, running_var);
    return mz::check_launch("k_bn_running");
}

int mz_pool2d_nhwc(const float *x, int32_t G, int32_t H, int32_t W, int32_t C, int32_t kernel, int32_t stride,
                   int32_t is_max, float *y, void *stream) {
    if (!x || !y || G <= 0 || H <= 0 || W <= 0 || C <= 0 || kernel <= 0 || stride <= 0 || kernel > H || kernel > W)
        return mz::fail(MZ_ERR_ARG, "mz_pool2d_nhwc: NULL argument or a window that does not fit");
    const int Ho = (H - kernel) / stride + 1, Wo = (W - kernel) / stride + 1;
    const long long total = (long long)G * Ho * Wo * C;
    hipLaunchKernelGGL(k_pool, dim3(blocks_for(total)), dim3(kCT), 0, (hipStream_t)stream, x, H, W, C, kernel, stride, Ho,
                       Wo, is_max, y, total);
    return mz::check_launch("k_pool");
}

int mz_adaptive_avg_to_nchw(const float *x, int32_t G, int32_t H, int32_t W, int32_t C, int32_t h, int32_t w, float *y,
                            void *stream) {
    if (!x || !y || G <= 0 || H <= 0 || W <= 0 || C <= 0 || h <= 0 || w <= 0)
        return mz::fail(MZ_ERR_ARG, "mz_adaptive_avg_to_nchw: NULL or non-positive argument");
    const long long total = (long long)G * h * w * C;
    hipLaunchKernelGGL(k_adaptive_avg_to_nchw, dim3(blocks_for(total)), dim3(kCT), 0, (hipStream_t)stream, x, H, W, C, h, w,
                       y, total);
    return mz::check_launch("k_adaptive_avg_to_nchw");
}

int mz_conv_direct_nhwc(const float *x, const float *weight_kmajor, const float *bias, int32_t G, int32_t H, int32_t W,
                        int32_t Cin, int32_t Cout, int32_t KS, int32_t stride, int32_t pad, int32_t activation, float *y,
                        void *stream) {
    if (!x || !weight_kmajor || !y || G <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KS <= 0 || stride <= 0 || pad < 0)
        return mz::fail(MZ_ERR_ARG, "mz_conv_direct_nhwc: NULL or non-positive argument");
    if (activation < MZ_ACT_NONE || activation > MZ_ACT_SIGMOID)
        return mz::fail(MZ_ERR_ARG, "mz_conv_direct_nhwc: activation %d is not element-wise", activation);
    const int Ho = (H + 2 * pad - KS) / stride + 1, Wo = (W + 2 * pad - KS) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return mz::fail(MZ_ERR_ARG, "mz_conv_direct_nhwc: the kernel does not fit the padded input");
    const long long total = (long long)G * Ho * Wo * Cout;
    hipLaunchKernelGGL(k_conv_direct, dim3(blocks_for(total)), dim3(kCT), 0, (hipStream_t)stream, x, weight_kmajor, bias, H, W,
                       Cin, Cout, KS, stride, pad, Ho, Wo, activation, y, total);
    return mz::check_launch("k_conv_direct");
}

int mz_conv_weight_kmajor(const float *weight, int32_t Cout, int32_t Cin, int32_t KS, float *out, void *stream) {
    if (!weight || !out || Cout <= 0 || Cin <= 0 || KS <= 0) return mz::fail(MZ_ERR_ARG, "mz_conv_weight_kmajor: NULL or non-positive argument");
    const long long total = (long long)Cout * Cin * KS * KS;
    hipLaunchKernelGGL(k_conv_weight_kmajor, dim3(blocks_for(total)), dim3(kCT), 0, (hipStream_t)stream, weight, Cout, Cin,
                       KS * KS, out);
    return mz::check_launch("k_conv_weight_kmajor");
}

}  // extern "C"
