// Deformable ("demons") stage of the registration pre-step, do_deformable: True (dataset/few_shot_reader.py:133-180,
// net/registration.py:195-212, 225-313): per slice a dense flow field [2][H][W] from zero by `iters` steps of
//   d = diffeomorphic(flow)          d_0 = flow / 2^10;  d_{i+1} = d_i + grid_sample(d_i, grid + d_i), ten times (:203-209)
//   loss = NCC(grid_sample(moving, grid + d), fixed)                                                (:157-160,246-261)
//   Adam(lr) on the flow, then flow <- Gaussian(sigma) * flow (conv2d, zero padding, per channel)   (:303-310,106-135)
// with `grid` = compute_grid() (align_corners=True-style coordinates 2 (j / (n - 1) - 0.5)) and the grid_sample defaults
// (bilinear, zero padding, align_corners=False).  The reference runs this slice after slice on cuda:0, ~150 small
// operators per step (autograd through ten chained grid_samples).  Here every step is 24 launches that each cover ALL
// slices: hand-derived backward of the scaling-and-squaring chain (the forward fields d_0 .. d_10 are kept; a step's
// backward = direct term + the sampling-position gradient + the bilinear scatter of the incoming gradient, the scatter
// by fp32 atomics into a pre-zeroed buffer out of a rotation of three), NCC from one pass of fp64 moments, the
// analytic NCC gradient, Adam and the 9x9 smoothing — no autograd graph, no per-slice loop, no host round trip.
// The launches are enqueued from C (rpnet_demons_register), asynchronously on the caller's stream.
#include <math.h>

#include "common.h"

namespace rpnet {

namespace {

struct Corners {
    int idx[4];          // nw, ne, sw, se linear pixel index, -1 when outside the image
    float wt[4];         // bilinear weights (s e, s w, n e, n w in torch's naming)
    float e, w, n, s;    // 1 - tx, tx, ty, 1 - ty
};

// sampling position of normalised coordinate (lx, ly): torch's CPU grid sampler, align_corners=False
__device__ __forceinline__ Corners corners(float lx, float ly, int H, int W) {
    const float ix = (lx + 1.f) * (0.5f * (float)W) - 0.5f, iy = (ly + 1.f) * (0.5f * (float)H) - 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    Corners c;
    c.w = ix - fx; c.e = 1.f - c.w; c.n = iy - fy; c.s = 1.f - c.n;
    // positions far outside (|coordinate| beyond int range) are outside either way: clamp before the conversion
    const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)W + 1.f), y0 = (int)fminf(fmaxf(fy, -2.f), (float)H + 1.f);
    const bool xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W;
    const bool yin0 = y0 >= 0 && y0 < H, yin1 = y0 + 1 >= 0 && y0 + 1 < H;
    c.idx[0] = (xin0 && yin0) ? y0 * W + x0 : -1;
    c.idx[1] = (xin1 && yin0) ? y0 * W + x0 + 1 : -1;
    c.idx[2] = (xin0 && yin1) ? (y0 + 1) * W + x0 : -1;
    c.idx[3] = (xin1 && yin1) ? (y0 + 1) * W + x0 + 1 : -1;
    c.wt[0] = c.s * c.e; c.wt[1] = c.s * c.w; c.wt[2] = c.n * c.e; c.wt[3] = c.n * c.w;
    return c;
}

__device__ __forceinline__ void grid_xy(int p, int H, int W, float& gx, float& gy) {
    const int i = p / W, j = p - i * W;
    gx = 2.f * ((float)j / (float)(W - 1) - 0.5f);
    gy = 2.f * ((float)i / (float)(H - 1) - 0.5f);
}

// one squaring step: d_out = d_in + grid_sample(d_in, grid + d_in)
__global__ __launch_bounds__(256) void compose_fwd_kernel(const float* __restrict__ din, float* __restrict__ dout, const int H,
                                                           const int W) {
    const int HW = H * W, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float* d0 = din + (size_t)blockIdx.y * 2 * HW;
    const float* d1 = d0 + HW;
    float gx, gy;
    grid_xy(p, H, W, gx, gy);
    const float dx = d0[p], dy = d1[p];
    const Corners c = corners(gx + dx, gy + dy, H, W);
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (c.idx[k] >= 0) {
            sx += d0[c.idx[k]] * c.wt[k];
            sy += d1[c.idx[k]] * c.wt[k];
        }
    float* o = dout + (size_t)blockIdx.y * 2 * HW;
    o[p] = dx + sx;
    o[HW + p] = dy + sy;
}

// warped = grid_sample(moving, grid + d); per-block fp64 partial moments (sum m, sum m^2, sum f m) for the NCC
__global__ __launch_bounds__(256) void warp_moments_kernel(const float* __restrict__ moving, const float* __restrict__ fixed,
                                                            const float* __restrict__ d, float* __restrict__ warped,
                                                            double* __restrict__ partial, const int H, const int W) {
    __shared__ double red[4];
    const int HW = H * W, p = blockIdx.x * 256 + threadIdx.x;
    const float* img = moving + (size_t)blockIdx.y * HW;
    double m1 = 0.0, m2 = 0.0, fm = 0.0;
    if (p < HW) {
        const float* d0 = d + (size_t)blockIdx.y * 2 * HW;
        float gx, gy;
        grid_xy(p, H, W, gx, gy);
        const Corners c = corners(gx + d0[p], gy + d0[HW + p], H, W);
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c.idx[k] >= 0) v += img[c.idx[k]] * c.wt[k];
        warped[(size_t)blockIdx.y * HW + p] = v;
        m1 = v; m2 = (double)v * v; fm = (double)v * fixed[(size_t)blockIdx.y * HW + p];
    }
    m1 = block_sum256(m1, red);
    m2 = block_sum256(m2, red);
    fm = block_sum256(fm, red);
    if (threadIdx.x == 0) {
        double* o = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3;
        o[0] = m1; o[1] = m2; o[2] = fm;
    }
}

// fixed-image statistics, once: fstat[slice] = {mean f, sum (f - mean)^2}
__global__ __launch_bounds__(256) void fixed_stats_kernel(const float* __restrict__ fixed, double* __restrict__ fstat, const int HW) {
    __shared__ double red[4];
    const float* f = fixed + (size_t)blockIdx.x * HW;
    double s = 0.0;
    for (int p = threadIdx.x; p < HW; p += 256) s += f[p];
    const double mean = block_sum256(s, red) / HW;
    double q = 0.0;
    for (int p = threadIdx.x; p < HW; p += 256) q += (f[p] - mean) * (f[p] - mean);
    q = block_sum256(q, red);
    if (threadIdx.x == 0) {
        fstat[blockIdx.x * 2] = mean;
        fstat[blockIdx.x * 2 + 1] = q;
    }
}

// gradient of the NCC with respect to d_10: gL(p) (d warped / d position); full overwrite of gout, zero of gzero
//   L = -A / D,  A = sum a b,  B = sum a^2,  C = sum b^2,  D = sqrt(B C + 1e-10),  a = f - mean f,  b = m - mean m
//   dL/dm_p = -a_p / D + A B b_p / D^3        (the mean terms of autograd vanish: sum a = sum b = 0)
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ moving, const float* __restrict__ fixed,
                                                        const float* __restrict__ d, const float* __restrict__ warped,
                                                        const double* __restrict__ partial, const double* __restrict__ fstat,
                                                        float* __restrict__ gout, float* __restrict__ gzero,
                                                        float* __restrict__ loss, const int H, const int W) {
    __shared__ double red[4];
    const int HW = H * W, p = blockIdx.x * 256 + threadIdx.x;
    double m1 = 0.0, m2 = 0.0, fm = 0.0;
    for (int k = threadIdx.x; k < (int)gridDim.x; k += 256) {
        const double* q = partial + ((size_t)blockIdx.y * gridDim.x + k) * 3;
        m1 += q[0]; m2 += q[1]; fm += q[2];
    }
    m1 = block_sum256(m1, red);
    m2 = block_sum256(m2, red);
    fm = block_sum256(fm, red);
    const double fmean = fstat[blockIdx.y * 2], Bq = fstat[blockIdx.y * 2 + 1];
    const double mmean = m1 / HW;
    const double A = fm - fmean * m1, C = fmax(m2 - mmean * m1, 0.0), D = sqrt(Bq * C + 1e-10);
    if (loss && blockIdx.x == 0 && threadIdx.x == 0) loss[blockIdx.y] = (float)(-A / D);
    if (p >= HW) return;
    const float* img = moving + (size_t)blockIdx.y * HW;
    const float* d0 = d + (size_t)blockIdx.y * 2 * HW;
    const float a = (float)(fixed[(size_t)blockIdx.y * HW + p] - fmean), b = (float)(warped[(size_t)blockIdx.y * HW + p] - mmean);
    const float gl = (float)(-1.0 / D) * a + (float)(A * Bq / (D * D * D)) * b;
    float gx, gy;
    grid_xy(p, H, W, gx, gy);
    const Corners c = corners(gx + d0[p], gy + d0[HW + p], H, W);
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = c.idx[k] >= 0 ? img[c.idx[k]] : 0.f;
    const float dox = (v[1] - v[0]) * c.s + (v[3] - v[2]) * c.n, doy = (v[2] - v[0]) * c.e + (v[3] - v[1]) * c.w;
    float* go = gout + (size_t)blockIdx.y * 2 * HW;
    go[p] = gl * dox * (0.5f * (float)W);
    go[HW + p] = gl * doy * (0.5f * (float)H);
    float* gz = gzero + (size_t)blockIdx.y * 2 * HW;
    gz[p] = 0.f;
    gz[HW + p] = 0.f;
}

// backward of d_{i+1} = d_i + S(d_i, grid + d_i): gacc (pre-zeroed) += direct + position gradient + bilinear scatter
__global__ __launch_bounds__(256) void compose_bwd_kernel(const float* __restrict__ gin, const float* __restrict__ din,
                                                           float* __restrict__ gacc, float* __restrict__ gzero, const int H,
                                                           const int W) {
    const int HW = H * W, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const size_t off = (size_t)blockIdx.y * 2 * HW;
    float* gz = gzero + off;
    gz[p] = 0.f;
    gz[HW + p] = 0.f;
    const float g0 = gin[off + p], g1 = gin[off + HW + p];
    if (g0 == 0.f && g1 == 0.f) return;
    const float* d0 = din + off;
    const float* d1 = d0 + HW;
    float gx, gy;
    grid_xy(p, H, W, gx, gy);
    const Corners c = corners(gx + d0[p], gy + d1[p], H, W);
    float vx[4], vy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        vx[k] = c.idx[k] >= 0 ? d0[c.idx[k]] : 0.f;
        vy[k] = c.idx[k] >= 0 ? d1[c.idx[k]] : 0.f;
    }
    const float dsx_dx = (vx[1] - vx[0]) * c.s + (vx[3] - vx[2]) * c.n, dsx_dy = (vx[2] - vx[0]) * c.e + (vx[3] - vx[1]) * c.w;
    const float dsy_dx = (vy[1] - vy[0]) * c.s + (vy[3] - vy[2]) * c.n, dsy_dy = (vy[2] - vy[0]) * c.e + (vy[3] - vy[1]) * c.w;
    float* a0 = gacc + off;
    float* a1 = a0 + HW;
    unsafeAtomicAdd(a0 + p, g0 + (g0 * dsx_dx + g1 * dsy_dx) * (0.5f * (float)W));
    unsafeAtomicAdd(a1 + p, g1 + (g0 * dsx_dy + g1 * dsy_dy) * (0.5f * (float)H));
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (c.idx[k] >= 0) {
            unsafeAtomicAdd(a0 + c.idx[k], g0 * c.wt[k]);
            unsafeAtomicAdd(a1 + c.idx[k], g1 * c.wt[k]);
        }
}

// torch.optim.Adam on the flow (gradient = g0 / 2^10), into `out`
__global__ __launch_bounds__(256) void flow_adam_kernel(const float* __restrict__ flow, const float* __restrict__ g0,
                                                         float* __restrict__ am, float* __restrict__ av, float* __restrict__ out,
                                                         const size_t n, const float inv_scale, const float omb1,
                                                         const float beta2, const float omb2, const float step,
                                                         const float bc2s, const float eps) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const float g = g0[e] * inv_scale;
    const float m = am[e] + omb1 * (g - am[e]);               // exp_avg.lerp_(g, 1 - beta1)
    const float v = av[e] * beta2 + omb2 * g * g;             // exp_avg_sq.mul_(beta2).addcmul_(g, g, value = 1 - beta2)
    am[e] = m;
    av[e] = v;
    out[e] = flow[e] - step * (m / (sqrtf(v) / bc2s + eps));
}

// flow = conv2d(tmp, K x K kernel, zero padding) per channel; d0 = flow / 2^10.  A block owns a 32 x 8 pixel tile and
// stages the tile + halo and the kernel in LDS (K <= 17); the taps are accumulated in the row-major order of the
// direct sum.
constexpr int kSmoothTW = 32, kSmoothTH = 8, kSmoothMaxK = 17;
__global__ __launch_bounds__(256) void smooth_kernel(const float* __restrict__ tmp, const float* __restrict__ kern, const int K,
                                                      float* __restrict__ flow, float* __restrict__ d0, const int H, const int W,
                                                      const float inv_scale) {
    __shared__ float tile[(kSmoothTH + kSmoothMaxK - 1) * (kSmoothTW + kSmoothMaxK - 1)];
    __shared__ float kw[kSmoothMaxK * kSmoothMaxK];
    const int r = K / 2, span_w = kSmoothTW + 2 * r, span_h = kSmoothTH + 2 * r, t = threadIdx.x;
    const int tiles_x = (W + kSmoothTW - 1) / kSmoothTW;
    const int ty0 = (blockIdx.x / tiles_x) * kSmoothTH, tx0 = (blockIdx.x % tiles_x) * kSmoothTW;
    const size_t HW = (size_t)H * W;
    const float* src = tmp + (size_t)blockIdx.y * HW;         // blockIdx.y = slice * 2 + channel
    for (int e = t; e < K * K; e += 256) kw[e] = kern[e];
    for (int e = t; e < span_w * span_h; e += 256) {
        const int y = ty0 - r + e / span_w, x = tx0 - r + e % span_w;
        tile[e] = (y >= 0 && y < H && x >= 0 && x < W) ? src[(size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    const int ly = t / kSmoothTW, lx = t % kSmoothTW, y = ty0 + ly, x = tx0 + lx;
    if (y >= H || x >= W) return;
    float acc = 0.f;
    for (int u = 0; u < K; ++u)
        for (int v = 0; v < K; ++v) acc += kw[u * K + v] * tile[(ly + u) * span_w + lx + v];
    flow[(size_t)blockIdx.y * HW + (size_t)y * W + x] = acc;
    d0[(size_t)blockIdx.y * HW + (size_t)y * W + x] = acc * inv_scale;
}

// out = post(grid_sample(x, grid + d))
__global__ __launch_bounds__(256) void displacement_warp_kernel(const float* __restrict__ x, const float* __restrict__ d,
                                                                 float* __restrict__ out, const int H, const int W,
                                                                 const float threshold, const float scale, const float shift) {
    const int HW = H * W, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float* img = x + (size_t)blockIdx.y * HW;
    const float* d0 = d + (size_t)blockIdx.y * 2 * HW;
    float gx, gy;
    grid_xy(p, H, W, gx, gy);
    const Corners c = corners(gx + d0[p], gy + d0[HW + p], H, W);
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (c.idx[k] >= 0) v += img[c.idx[k]] * c.wt[k];
    if (threshold >= 0.f) v = v > threshold ? 1.f : 0.f;
    out[(size_t)blockIdx.y * HW + p] = v * scale + shift;
}

constexpr int kSteps = 10;                                     // Diffeomorphic(10), net/registration.py:240

struct Workspace {
    float* d[kSteps];      // d_0 .. d_9 ([S][2][HW] each); d_10 lives in the caller's `disp`
    float* g[3];           // gradient rotation
    float* tmp;            // flow after Adam, before the smoothing
    float* am;
    float* av;
    float* warped;         // [S][HW]
    double* partial;       // [S][nblk][3]
    double* fstat;         // [S][2]
};

size_t carve(Workspace& w, void* base, int S, int H, int W) {
    const size_t HW = (size_t)H * W, field = (size_t)S * 2 * HW * sizeof(float);
    const int nblk = cdiv((long)HW, 256);
    size_t off = 0;
    unsigned char* b = (unsigned char*)base;
    auto take = [&](size_t bytes) { void* p = b ? b + off : nullptr; off += (bytes + 255) & ~(size_t)255; return p; };
    for (int i = 0; i < kSteps; ++i) w.d[i] = (float*)take(field);
    for (int i = 0; i < 3; ++i) w.g[i] = (float*)take(field);
    w.tmp = (float*)take(field);
    w.am = (float*)take(field);
    w.av = (float*)take(field);
    w.warped = (float*)take(field / 2);
    w.partial = (double*)take((size_t)S * nblk * 3 * sizeof(double));
    w.fstat = (double*)take((size_t)S * 2 * sizeof(double));
    return off;
}

}  // namespace

}  // namespace rpnet

extern "C" size_t rpnet_demons_workspace_bytes(int S, int H, int W) {
    rpnet::Workspace w;
    return S > 0 && H > 0 && W > 0 ? rpnet::carve(w, nullptr, S, H, W) : 0;
}

extern "C" int rpnet_demons_register(const float* moving, const float* fixed, const float* kernel, int ksize, float* flow,
                                     float* disp, float* loss, int S, int H, int W, int iters, double lr, double beta1,
                                     double beta2, double eps, void* workspace, size_t workspace_bytes, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(moving && fixed && kernel && flow && disp && workspace, RPNET_ERR_ARG, "demons_register: null pointer");
    RPNET_REQUIRE(S >= 0 && H >= 2 && W >= 2 && iters >= 0 && ksize >= 1 && (ksize & 1) && ksize <= kSmoothMaxK &&
                      (long)H * W < (1L << 28),
                  RPNET_ERR_SHAPE, "demons_register: S=%d H=%d W=%d iters=%d ksize=%d (odd, <= 17)", S, H, W, iters, ksize);
    if (S == 0) return RPNET_OK;
    RPNET_REQUIRE(workspace_bytes >= rpnet_demons_workspace_bytes(S, H, W), RPNET_ERR_WORKSPACE, "demons_register: workspace too small");
    Workspace w;
    carve(w, workspace, S, H, W);
    hipStream_t s = (hipStream_t)stream;
    const size_t HW = (size_t)H * W, n = (size_t)S * 2 * HW;
    const dim3 grid(cdiv((long)HW, 256), S), blk(256);
    const float inv_scale = 1.0f / (float)(1 << kSteps);
    // flow = 0 (net/registration.py:228,236), Adam state 0, d_0 = 0
    for (float* z : {flow, w.am, w.av, w.d[0]})
        if (hipMemsetAsync(z, 0, n * sizeof(float), s) != hipSuccess) return check_launch("demons_register (memset)");
    hipLaunchKernelGGL(fixed_stats_kernel, dim3(S), blk, 0, s, fixed, w.fstat, (int)HW);
    auto forward = [&]() {
        for (int i = 0; i < kSteps; ++i)
            hipLaunchKernelGGL(compose_fwd_kernel, grid, blk, 0, s, (const float*)w.d[i], i + 1 < kSteps ? w.d[i + 1] : disp, H, W);
    };
    for (int it = 1; it <= iters; ++it) {
        forward();
        hipLaunchKernelGGL(warp_moments_kernel, grid, blk, 0, s, moving, fixed, (const float*)disp, w.warped, w.partial, H, W);
        int x = 0, y = 1, z = 2;                               // gin, accumulate target (zeroed), next target (being zeroed)
        hipLaunchKernelGGL(warp_bwd_kernel, grid, blk, 0, s, moving, fixed, (const float*)disp, (const float*)w.warped,
                           (const double*)w.partial, (const double*)w.fstat, w.g[x], w.g[y], it == iters ? loss : (float*)nullptr, H, W);
        for (int i = kSteps - 1; i >= 0; --i) {
            hipLaunchKernelGGL(compose_bwd_kernel, grid, blk, 0, s, (const float*)w.g[x], (const float*)w.d[i], w.g[y], w.g[z], H, W);
            const int t = x; x = y; y = z; z = t;
        }
        const float step = (float)(lr / (1.0 - pow(beta1, (double)it))), bc2s = (float)sqrt(1.0 - pow(beta2, (double)it));
        hipLaunchKernelGGL(flow_adam_kernel, dim3(cdiv((long)n, 256)), blk, 0, s, (const float*)flow, (const float*)w.g[x], w.am, w.av,
                           w.tmp, n, inv_scale, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step, bc2s, (float)eps);
        hipLaunchKernelGGL(smooth_kernel, dim3(cdiv(H, kSmoothTH) * cdiv(W, kSmoothTW), S * 2), blk, 0, s, (const float*)w.tmp, kernel, ksize, flow, w.d[0], H, W, inv_scale);
    }
    forward();                                                 // the displacement of the final flow
    return check_launch("demons_register");
}

extern "C" int rpnet_displacement_warp(const float* x, const float* disp, float* out, int S, int H, int W, float threshold,
                                       float scale, float shift, rpnet_stream_t stream) {
    using namespace rpnet;
    RPNET_REQUIRE(x && disp && out, RPNET_ERR_ARG, "displacement_warp: null pointer");
    RPNET_REQUIRE(S >= 0 && H >= 2 && W >= 2, RPNET_ERR_SHAPE, "displacement_warp: S=%d H=%d W=%d", S, H, W);
    if (S == 0) return RPNET_OK;
    hipLaunchKernelGGL(displacement_warp_kernel, dim3(cdiv((long)H * W, 256), S), dim3(256), 0, (hipStream_t)stream, x, disp, out, H,
                       W, threshold, scale, shift);
    return check_launch("displacement_warp");
}
