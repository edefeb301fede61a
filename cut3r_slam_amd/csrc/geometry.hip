// HBM-bound geometry kernels of the tracking loop on gfx950 (compiled with -ffp-contract=off: every fp32 operation
// order below is explicit and identical to oracle/oracle_geom.c, so the integer decisions are bit-exact).
//   * reprojection-overlap counts   (hislam2/factor_graph.py:255-315 cal_overlap_batch / cal_overlap_bi)
//   * window alignment of one view  (hislam2/track_frontend.py:193-243)
//   * log-depth scale reduction     (hislam2/track_frontend.py:216-217)
//   * patch-overlap keyframe test   (hislam2/util/utils.py:726-736) on exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)
#include "common.h"
#include "../../include/cut3r_hip.h"

namespace {

struct Cam { float fx, fy, cx, cy; int W, H; float ax, bx, ay, by, mf; };

static Cam make_cam(float fx, float fy, float cx, float cy, int W, int H) {
    Cam c{fx, fy, cx, cy, W, H, 0.f, 0.f, 0.f, 0.f, 0.f};
    c.ax = cx + 0.5f; c.bx = (float)W - 0.5f - cx;
    c.ay = cy + 0.5f; c.by = (float)H - 0.5f - cy;
    // decision margin in pixels: 16x the worst rounding of the reference's  rint(fx*xc/zd + cx)  chain (one division and one
    // addition, each 2^-24 relative on magnitudes <= W + |cx|) and of the constants above
    c.mf = ((float)W + (float)H + fabsf(cx) + fabsf(cy) + 2.0f) * (1.0f / 1048576.0f);
    return c;
}

// the reference's test, literally (factor_graph.py:255-315): two IEEE divisions, round-half-even, bounds
DEVINL int proj_exact(float xc, float yc, float zc, float zd, const Cam& c) {
    const float u = rintf(c.fx * xc / zd + c.cx);
    const float v = rintf(c.fy * yc / zd + c.cy);
    return (u >= 0.0f) && (u < (float)c.W) && (v >= 0.0f) && (v < (float)c.H) && (zc > 0.0f);
}

// Same decision, same bits, without the divisions: with zd > 0,  -0.5 <= fx*xc/zd + cx < W - 0.5  is
// fx*xc + (cx + 0.5) zd >= 0  and  fx*xc - (W - 0.5 - cx) zd < 0  (one FMA each).  The two forms can only disagree within the
// rounding of the reference's chain of the bound, so a lane whose point lies within c.mf pixels of any bound (a few in a
// million) takes the literal test; the branch is wave-uniform and almost never taken.  Camera-space coordinates are the
// reference's own FMA chain, bit for bit.
DEVINL int proj_valid(const float* __restrict__ m, float x, float y, float z, const Cam& c, bool clamp_z) {
    const float xc = fmaf(m[2], z, fmaf(m[1], y, fmaf(m[0], x, m[3])));
    const float yc = fmaf(m[6], z, fmaf(m[5], y, fmaf(m[4], x, m[7])));
    const float zc = fmaf(m[10], z, fmaf(m[9], y, fmaf(m[8], x, m[11])));
    const float zd = clamp_z ? (zc < 1e-5f ? 1e-5f : zc) : zc;
    const float tx = c.fx * xc, ty = c.fy * yc;
    const float a = fmaf(c.ax, zd, tx), b = fmaf(-c.bx, zd, tx);
    const float e = fmaf(c.ay, zd, ty), d = fmaf(-c.by, zd, ty);
    int ins = (a >= 0.0f) && (b < 0.0f) && (e >= 0.0f) && (d < 0.0f) && (zc > 0.0f);
    const float mn = fminf(fminf(fabsf(a), fabsf(b)), fminf(fabsf(e), fabsf(d)));
    const bool near = (zc > 0.0f) && !(mn >= c.mf * zd);
    if (__any(near)) {
        if (near) ins = proj_exact(xc, yc, zc, zd, c);
    }
    return ins;
}

// forward test: ONE pointmap, B cameras.  Each thread keeps one point in registers and sweeps all cameras (the
// 3x4 matrices are wave-uniform -> scalar loads), so the pointmap is read from HBM exactly once.  Per-camera counts
// are accumulated in LDS (wave popcount of the ballot) and flushed with one global atomic per (block, camera).
constexpr int OVL_MAXB = 2048;
struct AlignArgs { float P[12]; float s; };

#ifndef CUT3R_FWD_PPT
#define CUT3R_FWD_PPT 2
#endif
constexpr int FWD_PPT = CUT3R_FWD_PPT;                  // points per thread: the camera rows (scalar loads) and the LDS atomic are shared by 2x64 tests (measured: 1, 2, 4 points -> 819, 869, 772 G tests/s at 1000 cameras)

struct Pt4 { float x[FWD_PPT], y[FWD_PPT], z[FWD_PPT]; bool ok[FWD_PPT]; };

// load FWD_PPT points (p, p + 256, ...) of a block's 1024-point span and, when asked, align them exactly as align_ds_kernel does
DEVINL Pt4 load_points(const float* __restrict__ pm, int N, int base, bool has_align, const float* __restrict__ P, float s) {
    Pt4 q;
#pragma unroll
    for (int k = 0; k < FWD_PPT; k++) {
        const int p = base + k * 256;
        q.ok[k] = p < N;
        float x = 0.f, y = 0.f, z = 0.f;
        if (q.ok[k]) { x = pm[3 * (size_t)p]; y = pm[3 * (size_t)p + 1]; z = pm[3 * (size_t)p + 2]; }
        if (has_align) {   // same operation order as align_ds_kernel / oracle_align_view: the full-res pointmap is never stored
            const float px = s * x, py = s * y, pz = s * z;
            x = fmaf(P[2], pz, fmaf(P[1], py, fmaf(P[0], px, P[3])));
            y = fmaf(P[6], pz, fmaf(P[5], py, fmaf(P[4], px, P[7])));
            z = fmaf(P[10], pz, fmaf(P[9], py, fmaf(P[8], px, P[11])));
        }
        q.x[k] = x; q.y[k] = y; q.z[k] = z;
    }
    return q;
}

// sweep cameras [c0, c1) (at most CAM_CHUNK of them: grid.y / grid.z splits the cameras so that a launch has enough workgroups
// whatever N is) over the block's points: per-camera counts in LDS (wave popcounts), one global atomic per (block, camera)
constexpr int CAM_CHUNK = 64;

DEVINL void sweep_cameras(const Pt4& q, const float* __restrict__ w2c, int c0, int c1, const Cam& cam, bool clamp_z, int32_t* cnt,
                          int32_t* __restrict__ out) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int nb = c1 - c0;
    if (tid < nb) cnt[tid] = 0;
    __syncthreads();
    for (int b = 0; b < nb; b++) {
        const float* m = w2c + 12 * (size_t)(c0 + b);
        int n = 0;
#pragma unroll
        for (int k = 0; k < FWD_PPT; k++) {
            const int v = q.ok[k] ? proj_valid(m, q.x[k], q.y[k], q.z[k], cam, clamp_z) : 0;
            n += (int)__popcll(__ballot(v));
        }
        if (lane == 0 && n) atomicAdd(&cnt[b], n);
    }
    __syncthreads();
    if (tid < nb && cnt[tid]) atomicAdd(&out[c0 + tid], cnt[tid]);
}

__global__ __launch_bounds__(256) void overlap_fwd_kernel(const float* __restrict__ pm, int N, const float* __restrict__ w2c, int B,
                                                          Cam cam, int32_t* __restrict__ counts, int has_align, AlignArgs al, int clamp_z) {
    __shared__ int32_t cnt[CAM_CHUNK];
    const int c0 = blockIdx.y * CAM_CHUNK, c1 = min(B, c0 + CAM_CHUNK);
    const Pt4 q = load_points(pm, N, blockIdx.x * (256 * FWD_PPT) + threadIdx.x, has_align != 0, al.P, al.s);
    if (clamp_z) sweep_cameras(q, w2c, c0, c1, cam, true, cnt, counts);
    else sweep_cameras(q, w2c, c0, c1, cam, false, cnt, counts);
}

// backward test: B pointmaps, ONE camera: a pure stream over B*N*12 bytes.  Each thread handles 4 consecutive points
// (three 16-byte loads); grid.y = pointmap, block-level reduction, one atomic per block.
__global__ __launch_bounds__(256) void overlap_bwd_kernel(const float* __restrict__ pms, int N, int grp, int grp_stride,
                                                          const float* __restrict__ w2c, Cam cam, int32_t* __restrict__ counts) {
    __shared__ int32_t wsum[4];
    __shared__ float m[12];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 12) m[tid] = w2c[tid];
    __syncthreads();
    const int b = blockIdx.y;
    const int slot = grp > 0 ? (b / grp) * grp_stride + (b % grp) : b;
    const float* pm = pms + (size_t)slot * N * 3;
    int c = 0;
    const int nquad = N >> 2;
    for (int qd = blockIdx.x * 256 + tid; qd < nquad; qd += gridDim.x * 256) {
        const f32x4* p4 = reinterpret_cast<const f32x4*>(pm + (size_t)qd * 12);
        const f32x4 a = p4[0], bb = p4[1], cc = p4[2];
        c += proj_valid(m, a[0], a[1], a[2], cam, false);
        c += proj_valid(m, a[3], bb[0], bb[1], cam, false);
        c += proj_valid(m, bb[2], bb[3], cc[0], cam, false);
        c += proj_valid(m, cc[1], cc[2], cc[3], cam, false);
    }
    if (blockIdx.x == 0)
        for (int p = (nquad << 2) + tid; p < N; p += 256) c += proj_valid(m, pm[3 * p], pm[3 * p + 1], pm[3 * p + 2], cam, false);
    c = wave_sum_i(c);
    if (lane == 0) wsum[wave] = c;
    __syncthreads();
    if (tid == 0) {
        const int t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (t) atomicAdd(&counts[b], t);
    }
}

__global__ __launch_bounds__(256) void align_depth_kernel(const float* __restrict__ pts, int n, float s, float* __restrict__ depth) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * blockDim.x)
        depth[i] = s * pts[3 * i + 2];
}

__global__ __launch_bounds__(256) void align_ds_kernel(const float* __restrict__ pts, const float* __restrict__ conf, int H, int W,
                                                       AlignArgs a, int ds, float* __restrict__ pm_ds, float* __restrict__ conf_ds) {
    const int Hd = H / ds, Wd = W / ds;
    const size_t total = (size_t)Hd * Wd;
    for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(o / Wd), x = (int)(o - (size_t)y * Wd);
        const size_t i = (size_t)(y * ds) * W + (size_t)(x * ds);
        const float px = a.s * pts[3 * i], py = a.s * pts[3 * i + 1], pz = a.s * pts[3 * i + 2];
        pm_ds[3 * o + 0] = fmaf(a.P[2], pz, fmaf(a.P[1], py, fmaf(a.P[0], px, a.P[3])));
        pm_ds[3 * o + 1] = fmaf(a.P[6], pz, fmaf(a.P[5], py, fmaf(a.P[4], px, a.P[7])));
        pm_ds[3 * o + 2] = fmaf(a.P[10], pz, fmaf(a.P[9], py, fmaf(a.P[8], px, a.P[11])));
        conf_ds[o] = 1.0f - 1.0f / conf[i];
    }
}

// sum(log prev - log z): per-thread double accumulation, wave shuffle + one double atomic per block.
__global__ __launch_bounds__(256) void logdepth_kernel(const float* __restrict__ prev, const float* __restrict__ pts, int n,
                                                       double* __restrict__ out) {
    __shared__ double wsum[4];
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * blockDim.x)
        acc += (double)logf(prev[i]) - (double)logf(pts[3 * i + 2]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}

// ------------------------------------------------------------------------------------------------ whole-window update
// One tracking window = V consecutive keyframes t0..t0+V-1 whose network outputs sit in one [V,H,W,*] block.  The
// per-view launches above (2 align + memset + fwd + memset + bwd per keyframe, ~35 launches per window) become FOUR
// launches; per-element arithmetic is the same code (align chain, proj_valid), so every count is unchanged.
struct WinArgs { float P[6][12]; float w2c_new[6][12]; float s; int V, t0, first, has_w2c; };

__global__ __launch_bounds__(256) void win_align_kernel(const float* __restrict__ pts, const float* __restrict__ conf, int H, int W,
                                                        WinArgs wa, int ds, float* __restrict__ pm_ds, float* __restrict__ conf_ds,
                                                        float* __restrict__ depth, float* __restrict__ w2c, int32_t* __restrict__ counts,
                                                        int ldc, double* __restrict__ lsum_reset) {
    const int v = blockIdx.y;
    // housekeeping that used to be separate stream operations: this view's world->camera row (host math, by value),
    // its zeroed count rows, and the reset of the log-depth accumulator for the NEXT window
    if (blockIdx.x == 0) {
        if (wa.has_w2c && threadIdx.x < 12) w2c[12 * (size_t)(wa.t0 + v) + threadIdx.x] = wa.w2c_new[v][threadIdx.x];
        if (v == 0 && threadIdx.x == 0 && lsum_reset) *lsum_reset = 0.0;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * ldc; i += gridDim.x * blockDim.x) counts[(size_t)v * 2 * ldc + i] = 0;
    const size_t n = (size_t)H * W;
    const float* p = pts + (size_t)v * n * 3;
    const float* c = conf + (size_t)v * n;
    float* d = depth + (size_t)v * n;
    const float s = wa.s;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s * p[3 * i + 2];
    const int Hd = H / ds, Wd = W / ds;
    const size_t total = (size_t)Hd * Wd;
    const float* P = wa.P[v];
    float* po = pm_ds + (size_t)v * total * 3;
    float* co = conf_ds + (size_t)v * total;
    for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(o / Wd), x = (int)(o - (size_t)y * Wd);
        const size_t i = (size_t)(y * ds) * W + (size_t)(x * ds);
        const float px = s * p[3 * i], py = s * p[3 * i + 1], pz = s * p[3 * i + 2];
        po[3 * o + 0] = fmaf(P[2], pz, fmaf(P[1], py, fmaf(P[0], px, P[3])));
        po[3 * o + 1] = fmaf(P[6], pz, fmaf(P[5], py, fmaf(P[4], px, P[7])));
        po[3 * o + 2] = fmaf(P[10], pz, fmaf(P[9], py, fmaf(P[8], px, P[11])));
        co[o] = 1.0f - 1.0f / c[i];
    }
}

// forward counts of every keyframe of the window: blockIdx.y = view v, keyframe i = t0 + v sees cameras 0..i-1
__global__ __launch_bounds__(256) void win_fwd_kernel(const float* __restrict__ pts, int N, const float* __restrict__ w2c, Cam cam,
                                                      int32_t* __restrict__ counts, int ldc, WinArgs wa) {
    __shared__ int32_t cnt[CAM_CHUNK];
    const int v = blockIdx.y, kfi = wa.t0 + v;
    const int c0 = blockIdx.z * CAM_CHUNK, c1 = min(kfi, c0 + CAM_CHUNK);
    if (kfi < wa.first || c0 >= kfi) return;
    const Pt4 q = load_points(pts + (size_t)v * N * 3, N, blockIdx.x * (256 * FWD_PPT) + threadIdx.x, true, wa.P[v], wa.s);
    sweep_cameras(q, w2c, c0, c1, cam, true, cnt, counts + (size_t)v * 2 * ldc);
}

// backward counts: blockIdx.y = stored pointmap b; the points are read ONCE and tested against the cameras of all the
// window's keyframes i = t0 + v > b (the per-keyframe entry point streams every stored pointmap once per keyframe)
__global__ __launch_bounds__(256) void win_bwd_kernel(const float* __restrict__ pms, int N, int grp, int grp_stride,
                                                      const float* __restrict__ w2c, Cam cam, int32_t* __restrict__ counts, int ldc,
                                                      WinArgs wa) {
    __shared__ int32_t wsum[6][4];
    __shared__ float m[6][12];
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 12 * wa.V) m[tid / 12][tid % 12] = w2c[12 * (size_t)(wa.t0 + tid / 12) + tid % 12];
    __syncthreads();
    const int slot = grp > 0 ? (b / grp) * grp_stride + (b % grp) : b;
    const float* pm = pms + (size_t)slot * N * 3;
    int c[6] = {0, 0, 0, 0, 0, 0};
    const int nquad = N >> 2;
    for (int qd = blockIdx.x * 256 + tid; qd < nquad; qd += gridDim.x * 256) {
        const f32x4* p4 = reinterpret_cast<const f32x4*>(pm + (size_t)qd * 12);
        const f32x4 a = p4[0], bb = p4[1], cc = p4[2];
#pragma unroll
        for (int v = 0; v < 6; v++) {
            if (v < wa.V) {
                c[v] += proj_valid(m[v], a[0], a[1], a[2], cam, false);
                c[v] += proj_valid(m[v], a[3], bb[0], bb[1], cam, false);
                c[v] += proj_valid(m[v], bb[2], bb[3], cc[0], cam, false);
                c[v] += proj_valid(m[v], cc[1], cc[2], cc[3], cam, false);
            }
        }
    }
    if (blockIdx.x == 0)
        for (int p = (nquad << 2) + tid; p < N; p += 256)
#pragma unroll
            for (int v = 0; v < 6; v++)
                if (v < wa.V) c[v] += proj_valid(m[v], pm[3 * p], pm[3 * p + 1], pm[3 * p + 2], cam, false);
#pragma unroll
    for (int v = 0; v < 6; v++) {
        c[v] = wave_sum_i(c[v]);
        if (lane == 0) wsum[v][wave] = c[v];
    }
    __syncthreads();
    if (tid < wa.V) {
        const int kfi = wa.t0 + tid;
        const int t = wsum[tid][0] + wsum[tid][1] + wsum[tid][2] + wsum[tid][3];
        if (kfi >= wa.first && b < kfi && t) atomicAdd(&counts[(size_t)tid * 2 * ldc + ldc + b], t);
    }
}

// ------------------------------------------------------------------------------------------------ patch overlap
// F.normalize(x, dim=1): x / max(||x||, 1e-12); one wave per row; rows 1.. only (row 0 dropped by the reference).
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ f, int Nv, int C, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Nv) return;
    const float* x = f + (size_t)(row + 1) * C;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + c);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    const float nrm = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    for (int c = lane * 4; c < C; c += 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + c);
        v[0] /= nrm; v[1] /= nrm; v[2] /= nrm; v[3] /= nrm;
        *reinterpret_cast<f32x4*>(out + (size_t)row * C + c) = v;
    }
}

// one wave per (32-row tile of f0, 32-row tile of f1): exact-fp32 MFMA 32x32x2; the k order is permuted identically
// for both operands (lane half h takes k = 8j + 4h + s), which lets every lane stream 16-byte pieces of its own row.
__global__ __launch_bounds__(64) void simmax_kernel(const float* __restrict__ f0, const float* __restrict__ f1, int Nv, int C,
                                                    unsigned int* __restrict__ rowmax_bits) {
    const int lane = threadIdx.x, r = lane & 31, hh = lane >> 5;
    const int rt = blockIdx.y, ct = blockIdx.x;
    const int ra = min(rt * 32 + r, Nv - 1), rb = min(ct * 32 + r, Nv - 1);
    const float* pa = f0 + (size_t)ra * C + 4 * hh;
    const float* pb = f1 + (size_t)rb * C + 4 * hh;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    for (int j = 0; j < C; j += 8) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(pa + j);
        const f32x4 b = *reinterpret_cast<const f32x4*>(pb + j);
#pragma unroll
        for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    }
    const bool colok = (ct * 32 + r) < Nv;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        float v = colok ? acc[i] : 0.f;
        v = fmaxf(v, 0.f);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));   // max over the 32 lanes of this half
        const int row = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (r == 0 && row < Nv) atomicMax(&rowmax_bits[row], __float_as_uint(v));
    }
}

__global__ __launch_bounds__(256) void count_gt_kernel(const unsigned int* __restrict__ rowmax_bits, int Nv, float thr,
                                                       int32_t* __restrict__ count) {
    int c = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Nv; i += gridDim.x * 256) c += __uint_as_float(rowmax_bits[i]) > thr;
    c = wave_sum_i(c);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

// ---- chained keyframe decisions (a whole look-ahead batch of candidates, no host round trip per candidate)
// simmax with the row operand chosen ON THE DEVICE: set `*state` + 1 of the normalised feature sets (set 0 = the last keyframe
// before the batch, set 1 + i = candidate i).
__global__ __launch_bounds__(64) void simmax_chain_kernel(const float* __restrict__ sets, size_t set_stride, const int32_t* __restrict__ state,
                                                          int cand, int Nv, int C, unsigned int* __restrict__ rowmax_bits) {
    const int lane = threadIdx.x, r = lane & 31, hh = lane >> 5;
    const int rt = blockIdx.y, ct = blockIdx.x;
    const int ra = min(rt * 32 + r, Nv - 1), rb = min(ct * 32 + r, Nv - 1);
    const float* f0 = sets + (size_t)(state[0] + 1) * set_stride;
    const float* f1 = sets + (size_t)(cand + 1) * set_stride;
    const float* pa = f0 + (size_t)ra * C + 4 * hh;
    const float* pb = f1 + (size_t)rb * C + 4 * hh;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    for (int j = 0; j < C; j += 8) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(pa + j);
        const f32x4 b = *reinterpret_cast<const f32x4*>(pb + j);
#pragma unroll
        for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    }
    const bool colok = (ct * 32 + r) < Nv;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        float v = colok ? acc[i] : 0.f;
        v = fmaxf(v, 0.f);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
        const int row = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (r == 0 && row < Nv) atomicMax(&rowmax_bits[row], __float_as_uint(v));
    }
}

// one workgroup: count the rows above the similarity threshold, take the keyframe decision of candidate `cand` exactly as the
// host does (ratio = fp32 count / fp32 rows, widened to double, compared with the double threshold: motion_filter.py:124 on
// `matched.mean().item()`), advance the device-side "last keyframe" and clear the row maxima for the next candidate.
__global__ __launch_bounds__(256) void chain_decide_kernel(unsigned int* __restrict__ rowmax_bits, int Nv, float thr_sim, double thr_ratio,
                                                           int cand, int forced, int32_t* __restrict__ state, int32_t* __restrict__ counts,
                                                           int32_t* __restrict__ decisions) {
    __shared__ int part[4];
    int c = 0;
    for (int i = threadIdx.x; i < Nv; i += 256) {
        c += __uint_as_float(rowmax_bits[i]) > thr_sim;
        rowmax_bits[i] = 0u;
    }
    c = wave_sum_i(c);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = part[0] + part[1] + part[2] + part[3];
        const float ratio = (float)total / (float)Nv;
        const int take = forced || ((double)ratio < thr_ratio);
        counts[cand] = total;
        decisions[cand] = take;
        if (take) state[0] = cand;
    }
}

inline int grid_for(size_t total, int block = 256) {
    size_t g = (total + block - 1) / block;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int cut3r_overlap_fwd(const float* pm, int N, const float* P_host, float s_align, const float* w2c, int B, float fx,
                                 float fy, float cx, float cy, int W, int H, int clamp_z, int32_t* counts, void* stream) {
    if (!pm || !w2c || !counts || N <= 0 || B <= 0) return CUT3R_ERR_ARG;
    AlignArgs al;
    for (int i = 0; i < 12; i++) al.P[i] = P_host ? P_host[i] : 0.f;
    al.s = s_align;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, sizeof(int32_t) * B, s) != hipSuccess) return CUT3R_ERR_LAUNCH;
    const Cam cam = make_cam(fx, fy, cx, cy, W, H);
    hipLaunchKernelGGL(overlap_fwd_kernel, dim3((N + 256 * FWD_PPT - 1) / (256 * FWD_PPT), (B + CAM_CHUNK - 1) / CAM_CHUNK), dim3(256), 0, s, pm, N, w2c, B, cam, counts, P_host ? 1 : 0, al, clamp_z);
    return cut3r_check_launch();
}

extern "C" int cut3r_overlap_bwd(const float* pms, int B, int N, int grp, int grp_stride, const float* w2c, float fx, float fy,
                                 float cx, float cy, int W, int H, int32_t* counts, void* stream) {
    if (!pms || !w2c || !counts || N <= 0 || B <= 0 || B > 65535 || grp < 0 || (grp > 0 && grp_stride < grp)) return CUT3R_ERR_ARG;
    if ((uintptr_t)pms & 15 || ((size_t)N * 12) & 15) return CUT3R_ERR_ARG;   // 16-B loads per pointmap
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, sizeof(int32_t) * B, s) != hipSuccess) return CUT3R_ERR_LAUNCH;
    const Cam cam = make_cam(fx, fy, cx, cy, W, H);
    int gx = ((N >> 2) + 255) / 256;
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(overlap_bwd_kernel, dim3(gx, B), dim3(256), 0, s, pms, N, grp, grp_stride, w2c, cam, counts);
    return cut3r_check_launch();
}

extern "C" int cut3r_align_view(const float* pts, const float* conf, int H, int W, const float* P_host, float s, int ds,
                                float* pm_ds, float* conf_ds, float* depth, void* stream) {
    if (!pts || !conf || !P_host || !pm_ds || !conf_ds || !depth || H <= 0 || W <= 0 || ds <= 0) return CUT3R_ERR_ARG;
    AlignArgs a;
    for (int i = 0; i < 12; i++) a.P[i] = P_host[i];
    a.s = s;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(align_depth_kernel, dim3(grid_for((size_t)H * W)), dim3(256), 0, st, pts, H * W, s, depth);
    hipLaunchKernelGGL(align_ds_kernel, dim3(grid_for((size_t)(H / ds) * (W / ds))), dim3(256), 0, st, pts, conf, H, W, a, ds, pm_ds,
                       conf_ds);
    return cut3r_check_launch();
}

extern "C" int cut3r_window_update(const float* pts, const float* conf, int V, int H, int W, const float* P_host, float s, int ds,
                                   float* pm_ds, float* conf_ds, float* depth, const float* store, int grp, int grp_stride,
                                   float* w2c, const float* w2c_new_host, int t0, int first, float fx, float fy, float cx, float cy,
                                   int32_t* counts, int ldc, double* lsum_reset, void* stream) {
    if (!pts || !conf || !P_host || !pm_ds || !conf_ds || !depth || !store || !w2c || !counts) return CUT3R_ERR_ARG;
    if (V < 1 || V > 6 || H <= 0 || W <= 0 || ds <= 0 || t0 < 0 || ldc < t0 + V || grp < 0 || (grp > 0 && grp_stride < grp)) return CUT3R_ERR_ARG;
    const int Nd = (H / ds) * (W / ds);
    if (((uintptr_t)store & 15) || (((size_t)Nd * 12) & 15)) return CUT3R_ERR_ARG;
    WinArgs wa;
    for (int v = 0; v < V; v++)
        for (int i = 0; i < 12; i++) wa.P[v][i] = P_host[v * 12 + i];
    wa.s = s; wa.V = V; wa.t0 = t0; wa.first = first; wa.has_w2c = w2c_new_host ? 1 : 0;
    for (int v = 0; v < V; v++)
        for (int i = 0; i < 12; i++) wa.w2c_new[v][i] = w2c_new_host ? w2c_new_host[v * 12 + i] : 0.f;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(win_align_kernel, dim3(grid_for((size_t)H * W), V), dim3(256), 0, st, pts, conf, H, W, wa, ds, pm_ds, conf_ds, depth,
                       w2c, counts, ldc, lsum_reset);
    const int last = t0 + V - 1;                  // newest keyframe: it sees cameras / pointmaps 0..last-1
    if (last >= first && last >= 1) {
        const Cam camf = make_cam(fx, fy, cx, cy, W, H);
        hipLaunchKernelGGL(win_fwd_kernel, dim3((H * W + 256 * FWD_PPT - 1) / (256 * FWD_PPT), V, (last + CAM_CHUNK - 1) / CAM_CHUNK), dim3(256), 0, st, pts, H * W, w2c, camf, counts, ldc, wa);
        // the reference tests the stored (stride-ds) pointmaps against the bounds of the DOWNSAMPLED map with the
        // full-resolution intrinsics (factor_graph.py:284-315 takes H, W from pointmap_i.shape: quirk kept)
        const Cam camb = make_cam(fx, fy, cx, cy, W / ds, H / ds);
        int gx = ((Nd >> 2) + 255) / 256;
        if (gx < 1) gx = 1;
        if (gx > 64) gx = 64;
        if (last > 65535) return CUT3R_ERR_ARG;
        hipLaunchKernelGGL(win_bwd_kernel, dim3(gx, last), dim3(256), 0, st, store, Nd, grp, grp_stride, w2c, camb, counts, ldc, wa);
    }
    return cut3r_check_launch();
}

extern "C" int cut3r_logdepth_sum(const float* prev_depth, const float* pts, int n, double* out, void* stream) {
    if (!prev_depth || !pts || !out || n <= 0) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(double), s) != hipSuccess) return CUT3R_ERR_LAUNCH;
    int g = grid_for((size_t)n);
    if (g > 512) g = 512;
    hipLaunchKernelGGL(logdepth_kernel, dim3(g), dim3(256), 0, s, prev_depth, pts, n, out);
    return cut3r_check_launch();
}

extern "C" int cut3r_logdepth_accum(const float* prev_depth, const float* pts, int n, double* out, void* stream) {
    if (!prev_depth || !pts || !out || n <= 0) return CUT3R_ERR_ARG;
    int g = grid_for((size_t)n);
    if (g > 512) g = 512;
    hipLaunchKernelGGL(logdepth_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, prev_depth, pts, n, out);
    return cut3r_check_launch();
}

extern "C" int cut3r_patch_overlap(const float* feat0, const float* feat1, int N, int C, float thr, void* ws, int32_t* count,
                                   void* stream) {
    if (!feat0 || !feat1 || !ws || !count || N < 2 || C <= 0 || (C & 7)) return CUT3R_ERR_ARG;
    if (((uintptr_t)feat0 | (uintptr_t)feat1 | (uintptr_t)ws) & 15) return CUT3R_ERR_ARG;
    const int Nv = N - 1;
    hipStream_t s = (hipStream_t)stream;
    float* n0 = (float*)ws;
    float* n1 = n0 + (size_t)Nv * C;
    unsigned int* rmax = (unsigned int*)(n1 + (size_t)Nv * C);
    if (hipMemsetAsync(rmax, 0, sizeof(unsigned int) * Nv, s) != hipSuccess) return CUT3R_ERR_LAUNCH;
    if (hipMemsetAsync(count, 0, sizeof(int32_t), s) != hipSuccess) return CUT3R_ERR_LAUNCH;
    hipLaunchKernelGGL(rownorm_kernel, dim3((Nv + 3) / 4), dim3(256), 0, s, feat0, Nv, C, n0);
    hipLaunchKernelGGL(rownorm_kernel, dim3((Nv + 3) / 4), dim3(256), 0, s, feat1, Nv, C, n1);
    const int T = (Nv + 31) / 32;
    hipLaunchKernelGGL(simmax_kernel, dim3(T, T), dim3(64), 0, s, n0, n1, Nv, C, rmax);
    hipLaunchKernelGGL(count_gt_kernel, dim3(grid_for((size_t)Nv)), dim3(256), 0, s, rmax, Nv, thr, count);
    return cut3r_check_launch();
}

extern "C" int cut3r_patch_overlap_chain(const float* feat_last, const float* feats, int B, int N, int C, float thr_sim, double thr_ratio,
                                         const int32_t* forced_host, void* ws, int32_t* state, int32_t* counts, int32_t* decisions,
                                         void* stream) {
    if (!feat_last || !feats || !ws || !state || !counts || !decisions || B < 1 || N < 2 || C <= 0 || (C & 7)) return CUT3R_ERR_ARG;
    if (((uintptr_t)feat_last | (uintptr_t)feats | (uintptr_t)ws) & 15) return CUT3R_ERR_ARG;
    const int Nv = N - 1;
    hipStream_t s = (hipStream_t)stream;
    float* sets = (float*)ws;                                  // [B+1][Nv][C] normalised rows 1..
    const size_t set_stride = (size_t)Nv * C;
    unsigned int* rmax = (unsigned int*)(sets + (size_t)(B + 1) * set_stride);
    if (hipMemsetAsync(rmax, 0, sizeof(unsigned int) * Nv, s) != hipSuccess) return CUT3R_ERR_LAUNCH;
    if (hipMemsetAsync(state, 0xff, sizeof(int32_t), s) != hipSuccess) return CUT3R_ERR_LAUNCH;           // -1: the keyframe before the batch
    hipLaunchKernelGGL(rownorm_kernel, dim3((Nv + 3) / 4), dim3(256), 0, s, feat_last, Nv, C, sets);
    for (int i = 0; i < B; i++)
        hipLaunchKernelGGL(rownorm_kernel, dim3((Nv + 3) / 4), dim3(256), 0, s, feats + (size_t)i * N * C, Nv, C, sets + (size_t)(i + 1) * set_stride);
    const int T = (Nv + 31) / 32;
    for (int i = 0; i < B; i++) {
        const int forced = forced_host ? (forced_host[i] != 0) : 0;
        if (!forced) hipLaunchKernelGGL(simmax_chain_kernel, dim3(T, T), dim3(64), 0, s, sets, set_stride, state, i, Nv, C, rmax);
        hipLaunchKernelGGL(chain_decide_kernel, dim3(1), dim3(256), 0, s, rmax, Nv, thr_sim, thr_ratio, i, forced, state, counts, decisions);
    }
    return cut3r_check_launch();
}
