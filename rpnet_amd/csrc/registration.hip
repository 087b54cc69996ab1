// Per-slice affine registration pre-step (dataset/few_shot_reader.py:109-198 with do_deformable=False,
// net/registration.py:316-357): theta = argmin MSE(grid_sample(moving, affine_grid(theta)), fixed) by 50 Adam steps.
// The reference runs the slices one after the other, ~20 tiny torch operators per step (on the CPU in this
// configuration).  Here ONE launch does everything: a block owns a slice, its 1024 threads sweep the pixels
// (bilinear sample with zero padding, analytic d loss / d theta), reduce the six gradient sums in fp64, thread 0
// applies torch.optim.Adam's single-tensor update to the six parameters in LDS, and the block goes round again —
// no host round trip, no intermediate tensors; slices run in parallel across the CUs.
// Conventions (PyTorch defaults, align_corners=False): base grid x_j = (2j + 1) / W - 1, sample position
// ix = ((gx + 1) W - 1) / 2, out-of-image corners read as zero in value AND in the gradient.
// The base grid xs[W], ys[H] is an INPUT (the host takes it from the very function the reference calls:
// F.affine_grid's torch.linspace(-1, 1, W) * (W - 1) / W) and the position follows the CPU grid sampler,
// ix = (gx + 1) * (W / 2) - 0.5.  This matters once: at theta = identity — the starting point — every sample sits on a
// pixel centre, a kink of the bilinear interpolant where d out / d ix is the right-hand difference if ix == j and the
// left-hand one if rounding put it a hair below.  torch.linspace rounds one of 128 coordinates below the centre; that
// single column changes the SIGN of a near-zero gradient component, Adam's first step is lr * sign(g), and theta
// ends 1e-2 away (so does the reference's own fp64 run).  With the library's grid the trajectory is the reference's.
#include <math.h>

#include "common.h"

namespace rpnet {

struct Bilinear {
    float out, dox, doy;      // value, d out / d ix, d out / d iy
};

__device__ __forceinline__ Bilinear sample_zeros(const float* __restrict__ img, int H, int W, float ix, float iy) {
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy;
    const bool xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W;
    const bool yin0 = y0 >= 0 && y0 < H, yin1 = y0 + 1 >= 0 && y0 + 1 < H;
    const float nw = (xin0 && yin0) ? img[y0 * W + x0] : 0.f;
    const float ne = (xin1 && yin0) ? img[y0 * W + x0 + 1] : 0.f;
    const float sw = (xin0 && yin1) ? img[(y0 + 1) * W + x0] : 0.f;
    const float se = (xin1 && yin1) ? img[(y0 + 1) * W + x0 + 1] : 0.f;
    Bilinear b;
    b.out = nw * (1.f - tx) * (1.f - ty) + ne * tx * (1.f - ty) + sw * (1.f - tx) * ty + se * tx * ty;
    b.dox = (ne - nw) * (1.f - ty) + (se - sw) * ty;
    b.doy = (sw - nw) * (1.f - tx) + (se - ne) * tx;
    return b;
}

__global__ __launch_bounds__(1024) void affine_register_kernel(const float* __restrict__ moving, const float* __restrict__ fixed,
                                                                const float* __restrict__ xs, const float* __restrict__ ys,
                                                                float* __restrict__ theta_out, float* __restrict__ loss_out,
                                                                const int H, const int W, const int iters, const double lr,
                                                                const double beta1, const double beta2, const double eps_d) {
    __shared__ float th[6], am[6], av[6];
    __shared__ double red[16][7];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int HW = H * W;
    const float* mov = moving + (size_t)blockIdx.x * HW;
    const float* fix = fixed + (size_t)blockIdx.x * HW;
    if (t < 6) {
        th[t] = (t == 0 || t == 4) ? 1.f : 0.f;      // identity (net/registration.py:320-322)
        am[t] = 0.f;
        av[t] = 0.f;
    }
    __syncthreads();
    const float invn2 = 2.f / (float)HW, hw2 = 0.5f * (float)W, hh2 = 0.5f * (float)H;
    // the optimiser's scalars are python doubles rounded to fp32 where they meet the fp32 tensors (1 - beta as a double first)
    const float omb1 = (float)(1.0 - beta1), b2f = (float)beta2, omb2 = (float)(1.0 - beta2), eps = (float)eps_d;
    for (int it = 1; it <= iters; ++it) {
        const float t0 = th[0], t1 = th[1], t2 = th[2], t3 = th[3], t4 = th[4], t5 = th[5];
        double s[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int p = t; p < HW; p += 1024) {
            const int i = p / W, j = p - i * W;
            const float xb = xs[j], yb = ys[i];
            const float gx = fmaf(1.f, t2, fmaf(yb, t1, xb * t0)), gy = fmaf(1.f, t5, fmaf(yb, t4, xb * t3));
            const Bilinear b = sample_zeros(mov, H, W, (gx + 1.f) * hw2 - 0.5f, (gy + 1.f) * hh2 - 0.5f);
            const float diff = b.out - fix[p];
            const float gl = diff * invn2;                   // d mean((fixed - warped)^2) / d warped
            const float gxs = gl * b.dox * hw2, gys = gl * b.doy * hh2;
            s[0] += (double)(gxs * xb); s[1] += (double)(gxs * yb); s[2] += (double)gxs;
            s[3] += (double)(gys * xb); s[4] += (double)(gys * yb); s[5] += (double)gys;
            s[6] += (double)(diff * diff);
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const double v = wave_sum(s[k]);
            if (lane == 0) red[wv][k] = v;
        }
        __syncthreads();
        if (t < 6) {
            double gsum = 0.0;
            for (int w16 = 0; w16 < 16; ++w16) gsum += red[w16][t];
            const float g = (float)gsum;
            // torch.optim.Adam, single-tensor path: exp_avg.lerp_(g, 1 - b1); exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2);
            // denom = exp_avg_sq.sqrt() / sqrt(1 - b2^t) + eps; param.addcdiv_(exp_avg, denom, value = -lr / (1 - b1^t))
            const float m = am[t] + omb1 * (g - am[t]);
            const float v = av[t] * b2f + omb2 * g * g;
            am[t] = m;
            av[t] = v;
            const float step = (float)(lr / (1.0 - pow(beta1, (double)it)));
            const float bc2s = (float)sqrt(1.0 - pow(beta2, (double)it));
            th[t] = th[t] - step * (m / (sqrtf(v) / bc2s + eps));
        }
        if (t == 6 && loss_out && it == iters) {
            double l = 0.0;
            for (int w16 = 0; w16 < 16; ++w16) l += red[w16][6];
            loss_out[blockIdx.x] = (float)(l / (double)HW);     // loss of the last evaluated theta (before its update)
        }
        __syncthreads();
    }
    if (t < 6) theta_out[blockIdx.x * 6 + t] = th[t];
}

// out = [threshold](scale * warp(x) + shift); MODE 0: affine grid of theta, MODE 1: compute_grid()'s identity grid
// (net/registration.py:171-187: gx = 2 (j / (W - 1) - 0.5)) sampled with align_corners=False (:258)
template <int MODE>
__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ x, const float* __restrict__ theta,
                                                    const float* __restrict__ xs, const float* __restrict__ ys,
                                                    float* __restrict__ out, const int H, const int W, const float threshold,
                                                    const float scale, const float shift) {
    const int HW = H * W;
    const float* img = x + (size_t)blockIdx.y * HW;
    float t0 = 1.f, t1 = 0.f, t2 = 0.f, t3 = 0.f, t4 = 1.f, t5 = 0.f;
    if (MODE == 0) {
        const float* th = theta + blockIdx.y * 6;
        t0 = th[0]; t1 = th[1]; t2 = th[2]; t3 = th[3]; t4 = th[4]; t5 = th[5];
    }
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        const int i = p / W, j = p - i * W;
        float gx, gy;
        if (MODE == 0) {
            const float xb = xs[j], yb = ys[i];
            gx = fmaf(1.f, t2, fmaf(yb, t1, xb * t0));
            gy = fmaf(1.f, t5, fmaf(yb, t4, xb * t3));
        } else {
            gx = 2.f * ((float)j / (float)(W - 1) - 0.5f);
            gy = 2.f * ((float)i / (float)(H - 1) - 0.5f);
        }
        const Bilinear b = sample_zeros(img, H, W, (gx + 1.f) * (0.5f * (float)W) - 0.5f, (gy + 1.f) * (0.5f * (float)H) - 0.5f);
        float v = b.out;
        if (threshold >= 0.f) v = v > threshold ? 1.f : 0.f;
        out[(size_t)blockIdx.y * HW + p] = v * scale + shift;
    }
}

}  // namespace rpnet

extern "C" int rpnet_affine_register(const float* moving, const float* fixed, const float* xs, const float* ys, float* theta,
                                     float* loss, int B, int H, int W, int iters, double lr, double beta1, double beta2,
                                     double eps, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(moving && fixed && xs && ys && theta, RPNET_ERR_ARG, "affine_register: null pointer");
    RPNET_REQUIRE(B >= 0 && H >= 2 && W >= 2 && iters >= 0 && (long)H * W < (1L << 30), RPNET_ERR_SHAPE,
                  "affine_register: B=%d H=%d W=%d iters=%d", B, H, W, iters);
    if (B == 0) return RPNET_OK;
    hipLaunchKernelGGL(affine_register_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, moving, fixed, xs, ys, theta, loss,
                       H, W, iters, lr, beta1, beta2, eps);
    return check_launch("affine_register");
}

extern "C" int rpnet_affine_warp(const float* x, const float* theta, const float* xs, const float* ys, float* out, int B, int H,
                                 int W, float threshold, float scale, float shift, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(x && theta && xs && ys && out, RPNET_ERR_ARG, "affine_warp: null pointer");
    RPNET_REQUIRE(B >= 0 && H >= 2 && W >= 2, RPNET_ERR_SHAPE, "affine_warp: B=%d H=%d W=%d", B, H, W);
    if (B == 0) return RPNET_OK;
    hipLaunchKernelGGL(warp_kernel<0>, dim3(cdiv((long)H * W, 256 * 4), B), dim3(256), 0, (hipStream_t)stream, x, theta, xs, ys, out,
                       H, W, threshold, scale, shift);
    return check_launch("affine_warp");
}

extern "C" int rpnet_identity_grid_warp(const float* x, float* out, int B, int H, int W, float threshold, float scale,
                                        float shift, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(x && out, RPNET_ERR_ARG, "identity_grid_warp: null pointer");
    RPNET_REQUIRE(B >= 0 && H >= 2 && W >= 2, RPNET_ERR_SHAPE, "identity_grid_warp: B=%d H=%d W=%d", B, H, W);
    if (B == 0) return RPNET_OK;
    hipLaunchKernelGGL(warp_kernel<1>, dim3(cdiv((long)H * W, 256 * 4), B), dim3(256), 0, (hipStream_t)stream, x,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, out, H, W, threshold, scale, shift);
    return check_launch("identity_grid_warp");
}
