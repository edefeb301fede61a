// Gaussian-splatting rasteriser for gfx950 (the operator behind the GS mapper's `render`,
// /root/reference/hislam2/gaussian/renderer/__init__.py:89-152 -> diff_gaussian_rasterization._C.rasterize_gaussians, the RaDe-GS
// flavour vendored under thirdparty/diff-gaussian-rasterization: colour, alpha, ray-distance depth (expected + median), camera-space
// coordinate (expected + median) and normal per pixel).
//
// Design for this chip (not the reference's kernel list):
//   * ONE 128-byte record per Gaussian (GS_REC floats, one cache line): the render kernel gathers whole records by sorted id into
//     LDS, 256 per batch, and every lane of the 16x16 tile reads the same record at the same time (LDS broadcast, no conflicts).
//   * the 3D covariance of a scale/rotation Gaussian is R S^2 R^T, so its inverse and its smallest eigenvector (the ray-space
//     plane of forward.cu:130-153) are written down directly from R and S -- no iterative eigen-solver per Gaussian.
//   * the per-Gaussian projection is ONE templated function: instantiated on float for the forward pass and on 10-wide forward-mode
//     dual numbers (mean, scale, quaternion) for the backward pass, which contracts the Jacobian with the gradients the render
//     backward accumulated per record -- the hand-derived chain of backward.cu:145-628 is not restated.
//   * binning: hipCUB inclusive scan + 64-bit radix sort of (tile << 32 | depth bits) keys, as the reference's CUB calls.
#include <hipcub/hipcub.hpp>
#include "common.h"
#include "lie_math.h"
#include "../../include/cut3r_hip.h"

namespace {
using namespace liemath;

constexpr int GS_REC = 32;                 // floats per Gaussian record
constexpr int GS_TILE = 16;                // tile edge (config.h:16-17)
constexpr int GS_BATCH = 256;              // Gaussians staged per round = threads per tile
constexpr int GS_STAGE = 25;               // leading floats of a record the render kernels read
enum { G_XY = 0, G_DEPTH = 2, G_TS = 3, G_CONIC = 4, G_OP = 7, G_RGB = 8, G_VP = 11, G_CP = 14, G_RP = 20, G_NRM = 22, G_RADIUS = 25,
       G_TILES = 26, G_RECT = 27, G_CLAMP = 31 };

struct GsCam {
    float view[16], proj[16], campos[3];
    int W, H;
    float tanx, tany, fx, fy, ks, mod;
    int deg, K;
};

template <typename T> struct GsProj {
    T xy[2], conic[3], coef, ts, vp[3], cp[6], rp[2], nrm[3], cov[3];
    bool in_front, det_ok;
};

DEVINL float Clampv(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }
template <int N> DEVINL Dual<N> Clampv(const Dual<N>& x, float lo, float hi) {
    if (x.v < lo) return Dual<N>(lo);
    if (x.v > hi) return Dual<N>(hi);
    return x;
}
DEVINL float Maxc(float x, float c) { return fmaxf(x, c); }
template <int N> DEVINL Dual<N> Maxc(const Dual<N>& x, float c) { return x.v < c ? Dual<N>(c) : x; }

// forward.cu:308-421 + :77-265 + :270-305 for one Gaussian, over T = float or dual numbers
template <typename T>
DEVINL GsProj<T> gs_project(const T* mean, const T* scale, const T* rot, const GsCam& cam) {
    GsProj<T> o;
    const float* V = cam.view;
    T pv[3], ph[4];
#pragma unroll
    for (int i = 0; i < 3; i++) pv[i] = mean[0] * T(V[i]) + mean[1] * T(V[4 + i]) + mean[2] * T(V[8 + i]) + T(V[12 + i]);
#pragma unroll
    for (int i = 0; i < 4; i++) ph[i] = mean[0] * T(cam.proj[i]) + mean[1] * T(cam.proj[4 + i]) + mean[2] * T(cam.proj[8 + i]) + T(cam.proj[12 + i]);
    o.in_front = val(pv[2]) > 0.2f;                                           // auxiliary.h:170
    const T p_w = T(1.f) / (ph[3] + T(1e-7f));
    o.xy[0] = ((ph[0] * p_w + T(1.f)) * T((float)cam.W) - T(1.f)) * T(0.5f);
    o.xy[1] = ((ph[1] * p_w + T(1.f)) * T((float)cam.H) - T(1.f)) * T(0.5f);
#pragma unroll
    for (int i = 0; i < 3; i++) o.vp[i] = pv[i];
    o.ts = Sqrt(pv[0] * pv[0] + pv[1] * pv[1] + pv[2] * pv[2]);
    // rotation of the (r, x, y, z) quaternion as given (the rasteriser does not normalise), squared scales
    const T r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    T R[3][3];
    R[0][0] = T(1.f) - T(2.f) * (y * y + z * z); R[0][1] = T(2.f) * (x * y - r * z); R[0][2] = T(2.f) * (x * z + r * y);
    R[1][0] = T(2.f) * (x * y + r * z); R[1][1] = T(1.f) - T(2.f) * (x * x + z * z); R[1][2] = T(2.f) * (y * z - r * x);
    R[2][0] = T(2.f) * (x * z - r * y); R[2][1] = T(2.f) * (y * z + r * x); R[2][2] = T(1.f) - T(2.f) * (x * x + y * y);
    T s2[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { const T s = T(cam.mod) * scale[k]; s2[k] = s * s; }
    // EWA: cov = (J Rw2c) Sigma (J Rw2c)^T with B = (J Rw2c) R  ->  cov = B diag(s2) B^T
    const T tz = o.in_front ? pv[2] : T(1.f);
    const T txtz = Clampv(pv[0] / tz, -1.3f * cam.tanx, 1.3f * cam.tanx), tytz = Clampv(pv[1] / tz, -1.3f * cam.tany, 1.3f * cam.tany);
    const T tx = txtz * tz, ty = tytz * tz;
    T A[2][3], B[2][3];
    {
        const T j00 = T(cam.fx) / tz, j02 = -(T(cam.fx) * tx) / (tz * tz), j11 = T(cam.fy) / tz, j12 = -(T(cam.fy) * ty) / (tz * tz);
#pragma unroll
        for (int j = 0; j < 3; j++) {                                          // Rw2c[i][j] = V[4 j + i]
            A[0][j] = j00 * T(V[4 * j + 0]) + j02 * T(V[4 * j + 2]);
            A[1][j] = j11 * T(V[4 * j + 1]) + j12 * T(V[4 * j + 2]);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) B[i][k] = A[i][0] * R[0][k] + A[i][1] * R[1][k] + A[i][2] * R[2][k];
    const T c00 = B[0][0] * B[0][0] * s2[0] + B[0][1] * B[0][1] * s2[1] + B[0][2] * B[0][2] * s2[2];
    const T c01 = B[0][0] * B[1][0] * s2[0] + B[0][1] * B[1][1] * s2[1] + B[0][2] * B[1][2] * s2[2];
    const T c11 = B[1][0] * B[1][0] * s2[0] + B[1][1] * B[1][1] * s2[1] + B[1][2] * B[1][2] * s2[2];
    const T a = c00 + T(cam.ks), b = c01, c = c11 + T(cam.ks);
    o.cov[0] = a; o.cov[1] = b; o.cov[2] = c;
    {
        const T d0 = Maxc(c00 * c11 - c01 * c01, 1e-6f), d1 = Maxc(a * c - c01 * c01, 1e-6f);
        o.coef = Sqrt(d0 / (d1 + T(1e-6f)) + T(1e-6f));
        if (val(d0) <= 1e-6f || val(d1) <= 1e-6f) o.coef = T(0.f);
    }
    const T det = a * c - b * b;
    o.det_ok = val(det) != 0.f;
    const T di = T(1.f) / (o.det_ok ? det : T(1.f));
    o.conic[0] = c * di; o.conic[1] = -b * di; o.conic[2] = a * di;
    // ray-space plane and normal (forward.cu:130-262): inverse covariance along the viewing ray, in the camera frame
    int kmin = val(s2[0]) > val(s2[1]) ? (val(s2[1]) > val(s2[2]) ? 2 : 1) : (val(s2[0]) > val(s2[2]) ? 2 : 0);
    const bool well = val(s2[kmin]) > 1e-8f;
    const T uvh[3] = {txtz, tytz, T(1.f)};
    T w[3], g[3], m[3];                                                         // w = Rw2c^T uvh, g = R^T w, m = Rw2c Sig_inv w
#pragma unroll
    for (int j = 0; j < 3; j++) w[j] = T(V[4 * j + 0]) * uvh[0] + T(V[4 * j + 1]) * uvh[1] + T(V[4 * j + 2]) * uvh[2];
#pragma unroll
    for (int k = 0; k < 3; k++) g[k] = R[0][k] * w[0] + R[1][k] * w[1] + R[2][k] * w[2];
    if (well) {
#pragma unroll
        for (int k = 0; k < 3; k++) g[k] = g[k] / s2[k];
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) if (k != kmin) g[k] = T(0.f);
    }
    T q[3];
#pragma unroll
    for (int i = 0; i < 3; i++) q[i] = R[i][0] * g[0] + R[i][1] * g[1] + R[i][2] * g[2];
#pragma unroll
    for (int i = 0; i < 3; i++) m[i] = T(V[0 + i]) * q[0] + T(V[4 + i]) * q[1] + T(V[8 + i]) * q[2];
    const T mlen = Sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
    const float mv = val(mlen);
    if (!(mv > 0.f) || mv != mv || isinf(mv)) {                                // isnan(normalize(.)) of forward.cu:155
#pragma unroll
        for (int k = 0; k < 6; k++) o.cp[k] = T(0.f);
        o.rp[0] = o.rp[1] = T(0.f);
        o.nrm[0] = o.nrm[1] = o.nrm[2] = T(0.f);
        return o;
    }
    const T n0 = m[0] / mlen, n1 = m[1] / mlen, n2 = m[2] / mlen;
    const T u2 = txtz * txtz, v2 = tytz * tytz, uv = txtz * tytz;
    const T l = Sqrt(tx * tx + ty * ty + tz * tz);
    const T vbn = Maxc(n0 * uvh[0] + n1 * uvh[1] + n2, 1e-7f);
    const T a0 = n0 / vbn, a1 = n1 / vbn, a2 = n2 / vbn;
    const T p0 = (v2 + T(1.f)) * a0 - uv * a1 - txtz * a2, p1 = (u2 + T(1.f)) * a1 - uv * a0 - tytz * a2;
    const T nl = u2 + v2 + T(1.f);
    const T ifx = T(1.f / cam.fx) / nl, ify = T(1.f / cam.fy) / nl;
    o.cp[0] = (p0 * tx - (v2 + T(1.f)) * tz) * ifx; o.cp[1] = (uv * tz + p1 * tx) * ify;
    o.cp[2] = (uv * tz + p0 * ty) * ifx;            o.cp[3] = (p1 * ty - (u2 + T(1.f)) * tz) * ify;
    o.cp[4] = (tx + p0 * tz) * ifx;                 o.cp[5] = (ty + p1 * tz) * ify;
    o.rp[0] = p0 * l * ifx; o.rp[1] = p1 * l * ify;
    const T fn = l / nl;
    const T r0 = -p0 * fn, r1 = -p1 * fn;                                      // ray normal (r0, r1, -1) through nJ
    const T c0 = r0 / tz - tx / l, c1 = r1 / tz - ty / l, c2 = -(r0 * tx + r1 * ty) / (tz * tz) - tz / l;
    const T cl = Sqrt(c0 * c0 + c1 * c1 + c2 * c2);
    o.nrm[0] = c0 / cl; o.nrm[1] = c1 / cl; o.nrm[2] = c2 / cl;
    return o;
}

__constant__ float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
__constant__ float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f, -0.4570457994644658f,
                               1.445305721320277f, -0.5900435899266435f};

// SH basis of forward.cu:23-74 at the unit direction (mean - campos): b[0..15]
template <typename T> DEVINL void sh_basis(const T* mean, const float* campos, int deg, T* b) {
    T d[3] = {mean[0] - T(campos[0]), mean[1] - T(campos[1]), mean[2] - T(campos[2])};
    const T len = Sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const T x = d[0] / len, y = d[1] / len, z = d[2] / len;
    b[0] = T(0.28209479177387814f);
    if (deg > 0) { b[1] = -T(0.4886025119029199f) * y; b[2] = T(0.4886025119029199f) * z; b[3] = -T(0.4886025119029199f) * x; }
    if (deg > 1) {
        const T xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        b[4] = T(SH_C2[0]) * xy; b[5] = T(SH_C2[1]) * yz; b[6] = T(SH_C2[2]) * (T(2.f) * zz - xx - yy); b[7] = T(SH_C2[3]) * xz;
        b[8] = T(SH_C2[4]) * (xx - yy);
        if (deg > 2) {
            b[9] = T(SH_C3[0]) * y * (T(3.f) * xx - yy); b[10] = T(SH_C3[1]) * xy * z; b[11] = T(SH_C3[2]) * y * (T(4.f) * zz - xx - yy);
            b[12] = T(SH_C3[3]) * z * (T(2.f) * zz - T(3.f) * xx - T(3.f) * yy); b[13] = T(SH_C3[4]) * x * (T(4.f) * zz - xx - yy);
            b[14] = T(SH_C3[5]) * z * (xx - yy); b[15] = T(SH_C3[6]) * x * (xx - T(3.f) * yy);
        }
    }
}

DEVINL int gs_tile_clamp(float v, int g) {                                     // auxiliary.h:62-72: (int) truncation, then [0, g]
    const int t = (int)(v / (float)GS_TILE);
    return t < 0 ? 0 : (t > g ? g : t);
}

__global__ __launch_bounds__(256) void gs_preprocess_kernel(int P, const float* __restrict__ means, const float* __restrict__ scales,
                                                            const float* __restrict__ rots, const float* __restrict__ opac,
                                                            const float* __restrict__ shs, const float* __restrict__ colors, GsCam cam,
                                                            float* __restrict__ geom, int* __restrict__ radii, unsigned* __restrict__ tiles) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float* rec = geom + (size_t)i * GS_REC;
#pragma unroll
    for (int k = 0; k < GS_REC; k++) rec[k] = 0.f;
    radii[i] = 0;
    tiles[i] = 0;
    const float mean[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
    const float sc[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
    const float rt[4] = {rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3]};
    const GsProj<float> o = gs_project<float>(mean, sc, rt, cam);
    if (!o.in_front || !o.det_ok) return;
    const float mid = 0.5f * (o.cov[0] + o.cov[2]);
    const float root = sqrtf(fmaxf(0.1f, mid * mid - (o.cov[0] * o.cov[2] - o.cov[1] * o.cov[1])));
    const float radius = ceilf(3.f * sqrtf(fmaxf(mid + root, mid - root)));
    const int gx = (cam.W + GS_TILE - 1) / GS_TILE, gy = (cam.H + GS_TILE - 1) / GS_TILE;
    const int rad = (int)radius;
    const int x0 = gs_tile_clamp(o.xy[0] - (float)rad, gx), y0 = gs_tile_clamp(o.xy[1] - (float)rad, gy);
    const int x1 = gs_tile_clamp(o.xy[0] + (float)rad + (float)(GS_TILE - 1), gx), y1 = gs_tile_clamp(o.xy[1] + (float)rad + (float)(GS_TILE - 1), gy);
    if ((x1 - x0) * (y1 - y0) == 0) return;
    float rgb[3];
    unsigned clampbits = 0;
    if (colors) {
        rgb[0] = colors[3 * i]; rgb[1] = colors[3 * i + 1]; rgb[2] = colors[3 * i + 2];
    } else {
        float b[16];
        sh_basis<float>(mean, cam.campos, cam.deg, b);
        const int nb = (cam.deg + 1) * (cam.deg + 1);
        const float* sh = shs + (size_t)i * cam.K * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            float v = 0.f;
            for (int k = 0; k < nb; k++) v += b[k] * sh[3 * k + ch];
            v += 0.5f;
            if (v < 0.f) { clampbits |= 1u << ch; v = 0.f; }
            rgb[ch] = v;
        }
    }
    rec[G_XY] = o.xy[0]; rec[G_XY + 1] = o.xy[1];
    rec[G_DEPTH] = o.vp[2];
    rec[G_TS] = o.ts;
    rec[G_CONIC] = o.conic[0]; rec[G_CONIC + 1] = o.conic[1]; rec[G_CONIC + 2] = o.conic[2];
    rec[G_OP] = opac[i] * o.coef;
#pragma unroll
    for (int k = 0; k < 3; k++) { rec[G_RGB + k] = rgb[k]; rec[G_VP + k] = o.vp[k]; rec[G_NRM + k] = o.nrm[k]; }
#pragma unroll
    for (int k = 0; k < 6; k++) rec[G_CP + k] = o.cp[k];
    rec[G_RP] = o.rp[0]; rec[G_RP + 1] = o.rp[1];
    rec[G_RADIUS] = radius;
    rec[G_TILES] = __int_as_float((x1 - x0) * (y1 - y0));
    rec[G_RECT] = __int_as_float(x0); rec[G_RECT + 1] = __int_as_float(y0); rec[G_RECT + 2] = __int_as_float(x1); rec[G_RECT + 3] = __int_as_float(y1);
    rec[G_CLAMP] = __uint_as_float(clampbits);
    radii[i] = rad;
    tiles[i] = (unsigned)((x1 - x0) * (y1 - y0));
}

// rasterizer_impl.cu:70-112: one (tile << 32 | depth bits, id) pair per covered tile
__global__ __launch_bounds__(256) void gs_duplicate_kernel(int P, const float* __restrict__ geom, const unsigned* __restrict__ offsets, int gx,
                                                           unsigned long long n_inst, unsigned long long* __restrict__ keys,
                                                           unsigned* __restrict__ vals, int* __restrict__ overflow) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float* rec = geom + (size_t)i * GS_REC;
    if (__float_as_int(rec[G_TILES]) <= 0) return;
    unsigned long long off = i == 0 ? 0 : offsets[i - 1];
    const int x0 = __float_as_int(rec[G_RECT]), y0 = __float_as_int(rec[G_RECT + 1]), x1 = __float_as_int(rec[G_RECT + 2]),
              y1 = __float_as_int(rec[G_RECT + 3]);
    const unsigned long long dbits = __float_as_uint(rec[G_DEPTH]);
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            if (off >= n_inst) {                                               // only in capacity mode: the buffers were sized by a guess
                if (overflow) *overflow = 1;
                return;
            }
            keys[off] = ((unsigned long long)(y * gx + x) << 32) | dbits;
            vals[off] = (unsigned)i;
            off++;
        }
}

// rasterizer_impl.cu:151-176: [start, end) of every tile in the sorted list
__global__ __launch_bounds__(256) void gs_ranges_kernel(unsigned long long n, const unsigned long long* __restrict__ keys, int n_tiles,
                                                        unsigned* __restrict__ ranges) {
    const unsigned long long i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    const unsigned t = (unsigned)(keys[i] >> 32);
    const bool ok = t < (unsigned)n_tiles;                                     // (padding keys of the capacity mode carry tile id n_tiles)
    if (i == 0) {
        if (ok) ranges[2 * t] = 0;
    } else {
        const unsigned tp = (unsigned)(keys[i - 1] >> 32);
        if (tp != t) {
            if (tp < (unsigned)n_tiles) ranges[2 * tp + 1] = (unsigned)i;
            if (ok) ranges[2 * t] = (unsigned)i;
        }
    }
    if (i == n - 1 && ok) ranges[2 * t + 1] = (unsigned)n;
}

__global__ __launch_bounds__(256) void gs_pad_keys_kernel(unsigned long long n, unsigned long long pad, unsigned long long* __restrict__ keys,
                                                          unsigned* __restrict__ vals) {
    const unsigned long long i = blockIdx.x * 256ull + threadIdx.x;
    if (i < n) { keys[i] = pad; vals[i] = 0; }
}

// forward.cu:429-692: one 16x16 tile per workgroup, one pixel per thread, front-to-back
__global__ __launch_bounds__(GS_BATCH) void gs_render_fwd_kernel(const unsigned* __restrict__ ranges, const unsigned* __restrict__ point_list,
                                                                 const float* __restrict__ geom, int W, int H, float fx, float fy, float bg0,
                                                                 float bg1, float bg2, float* __restrict__ out_color,
                                                                 float* __restrict__ out_coord, float* __restrict__ out_mcoord,
                                                                 float* __restrict__ out_depth, float* __restrict__ out_mdepth,
                                                                 float* __restrict__ out_alpha, float* __restrict__ out_normal,
                                                                 unsigned* __restrict__ n_contrib, float* __restrict__ aux) {
    __shared__ float stage[GS_BATCH * GS_STAGE];
    __shared__ unsigned stage_id[GS_BATCH];
    const int gx = (W + GS_TILE - 1) / GS_TILE;
    const int tid = threadIdx.x;
    const int px = blockIdx.x * GS_TILE + (tid & 15), py = blockIdx.y * GS_TILE + (tid >> 4);
    const bool inside = px < W && py < H;
    const size_t HW = (size_t)W * H, pix = (size_t)py * W + px;
    const float pxf = (float)px, pyf = (float)py;
    const unsigned r0 = ranges[2 * (blockIdx.y * gx + blockIdx.x)], r1 = ranges[2 * (blockIdx.y * gx + blockIdx.x) + 1];
    bool done = !inside;
    float T = 1.f, weight = 0.f, C[3] = {0.f, 0.f, 0.f}, Co[3] = {0.f, 0.f, 0.f}, mC[3] = {0.f, 0.f, 0.f}, Nr[3] = {0.f, 0.f, 0.f}, D = 0.f, mD = 0.f;
    unsigned contributor = 0, last = 0, maxc = 0xffffffffu;
    for (unsigned base = r0; base < r1; base += GS_BATCH) {
        if (__syncthreads_count(done) == GS_BATCH) break;
        const unsigned cnt = (r1 - base) < (unsigned)GS_BATCH ? (r1 - base) : (unsigned)GS_BATCH;
        if ((unsigned)tid < cnt) stage_id[tid] = point_list[base + tid];
        __syncthreads();
        for (unsigned e = tid; e < cnt * GS_STAGE; e += GS_BATCH) {            // coalesced over the 25 leading floats of each record
            const unsigned j = e / GS_STAGE, f = e - j * GS_STAGE;
            stage[e] = geom[(size_t)stage_id[j] * GS_REC + f];
        }
        __syncthreads();
        for (unsigned j = 0; !done && j < cnt; j++) {
            contributor++;
            const float* g = stage + j * GS_STAGE;
            const float dx = g[G_XY] - pxf, dy = g[G_XY + 1] - pyf;
            const float power = -0.5f * (g[G_CONIC] * dx * dx + g[G_CONIC + 2] * dy * dy) - g[G_CONIC + 1] * dx * dy;
            if (power > 0.f) continue;
            const float alpha = fminf(0.99f, g[G_OP] * expf(power));
            if (alpha < 1.f / 255.f) continue;
            const float test_T = T * (1.f - alpha);
            if (test_T < 0.0001f) { done = true; continue; }
            const float aT = alpha * T;
            const bool before = T > 0.5f;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                C[ch] += g[G_RGB + ch] * aT;
                Nr[ch] += g[G_NRM + ch] * aT;
                const float co = g[G_VP + ch] + g[G_CP + 2 * ch] * dx + g[G_CP + 2 * ch + 1] * dy;
                Co[ch] += co * aT;
                if (before) mC[ch] = co;
            }
            const float t = g[G_TS] + g[G_RP] * dx + g[G_RP + 1] * dy;
            D += t * aT;
            if (before) { mD = t; maxc = contributor; }
            weight += aT;
            T = test_T;
            last = contributor;
        }
    }
    if (!inside) return;
    const float nx = (pxf - 0.5f * (float)W) / fx, ny = (pyf - 0.5f * (float)H) / fy;
    const float ln = sqrtf(nx * nx + ny * ny + 1.f);
    n_contrib[pix] = last;
    n_contrib[HW + pix] = maxc;
    out_color[pix] = C[0] + T * bg0; out_color[HW + pix] = C[1] + T * bg1; out_color[2 * HW + pix] = C[2] + T * bg2;
    out_alpha[pix] = weight;
    const float iw = last ? 1.f / weight : 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) { out_coord[ch * HW + pix] = Co[ch] * iw; out_mcoord[ch * HW + pix] = mC[ch]; }
    out_depth[pix] = D / ln * iw;
    out_mdepth[pix] = mD / ln;
    float nlen = 1.f;
    if (last) {
        nlen = sqrtf(Nr[0] * Nr[0] + Nr[1] * Nr[1] + Nr[2] * Nr[2]);
        const float il = 1.f / fmaxf(nlen, 1e-12f);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) out_normal[ch * HW + pix] = Nr[ch] * il;
    } else {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) out_normal[ch * HW + pix] = 0.f;
    }
    aux[pix] = T;                                                              // final transmittance and normal length: backward pass
    aux[HW + pix] = nlen;
}

// ------------------------------------------------------------------------------------------------ backward pass
// Render backward (the job of backward.cu:631-1015, re-derived): per pixel the outputs are sums  Q_k = sum_i f_ki a_i T_i  over its
// contributors (colour, coordinate, ray distance, normal, weight) plus T_final * bg, then normalised.  With G_k = dL/dQ_k the
// scalar v_i = sum_k G_k f_ki gives  dL/da_i = v_i T_i - (sum_{j>i} v_j a_j T_j + G_T T_final) / (1 - a_i): ONE suffix sum S per pixel,
// walked back to front with T_i = T_{i+1} / (1 - a_i).  The gradients of a Gaussian's record fields are summed over the wave
// (transposing butterfly, below) before one atomic per wave and field.  dgeom uses the record layout; slot G_DEPTH carries the |d/dxy| sum of the alpha
// path (backward.cu:1005, the densification statistic).
constexpr int GS_NGRAD = 25;               // record slots 0..24 receive gradients

__global__ __launch_bounds__(GS_BATCH) void gs_render_bwd_kernel(const unsigned* __restrict__ ranges, const unsigned* __restrict__ point_list,
                                                                 const float* __restrict__ geom, int W, int H, float fx, float fy, float bg0,
                                                                 float bg1, float bg2, const unsigned* __restrict__ n_contrib,
                                                                 const float* __restrict__ aux, const float* __restrict__ out_alpha,
                                                                 const float* __restrict__ out_coord, const float* __restrict__ out_depth,
                                                                 const float* __restrict__ out_normal, const float* __restrict__ g_color,
                                                                 const float* __restrict__ g_coord, const float* __restrict__ g_mcoord,
                                                                 const float* __restrict__ g_depth, const float* __restrict__ g_mdepth,
                                                                 const float* __restrict__ g_alpha, const float* __restrict__ g_normal,
                                                                 float* __restrict__ dgeom) {
    __shared__ float stage[GS_BATCH * GS_STAGE];
    __shared__ unsigned stage_id[GS_BATCH];
    const int gx = (W + GS_TILE - 1) / GS_TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int px = blockIdx.x * GS_TILE + (tid & 15), py = blockIdx.y * GS_TILE + (tid >> 4);
    const bool inside = px < W && py < H;
    const size_t HW = (size_t)W * H, pix = inside ? (size_t)py * W + px : 0;
    const float pxf = (float)px, pyf = (float)py;
    const unsigned r0 = ranges[2 * (blockIdx.y * gx + blockIdx.x)], r1 = ranges[2 * (blockIdx.y * gx + blockIdx.x) + 1];
    const unsigned n_list = r1 - r0;
    unsigned last = 0, maxc = 0xffffffffu;
    float T = 1.f, S = 0.f, GC[3] = {0.f, 0.f, 0.f}, GCo[3] = {0.f, 0.f, 0.f}, GN[3] = {0.f, 0.f, 0.f}, GmC[3] = {0.f, 0.f, 0.f}, GD = 0.f, GW = 0.f,
          GmD = 0.f;
    if (inside) {
        last = n_contrib[pix];
        maxc = n_contrib[HW + pix];
        T = aux[pix];
        const float nlen = aux[HW + pix];
        const float nx = (pxf - 0.5f * (float)W) / fx, ny = (pyf - 0.5f * (float)H) / fy;
        const float ln = sqrtf(nx * nx + ny * ny + 1.f);
        GC[0] = g_color[pix]; GC[1] = g_color[HW + pix]; GC[2] = g_color[2 * HW + pix];
        S = (GC[0] * bg0 + GC[1] * bg1 + GC[2] * bg2) * T;
        GW = g_alpha[pix];
        GmD = g_mdepth[pix] / ln;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) GmC[ch] = g_mcoord[ch * HW + pix];
        if (last) {
            const float iw = 1.f / out_alpha[pix];
            float dotn = 0.f;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const float gc = g_coord[ch * HW + pix];
                GCo[ch] = gc * iw;
                GW -= gc * out_coord[ch * HW + pix] * iw;
                dotn += g_normal[ch * HW + pix] * out_normal[ch * HW + pix];
            }
            GD = g_depth[pix] * iw / ln;
            GW -= g_depth[pix] * out_depth[pix] * iw;
            const bool unit = nlen > 1e-12f;
            const float il = 1.f / fmaxf(nlen, 1e-12f);
#pragma unroll
            for (int ch = 0; ch < 3; ch++) GN[ch] = (g_normal[ch * HW + pix] - (unit ? out_normal[ch * HW + pix] * dotn : 0.f)) * il;
        }
    }
    const unsigned rounds = (n_list + GS_BATCH - 1) / GS_BATCH;
    for (unsigned rd = 0; rd < rounds; rd++) {
        // batch rd holds list positions hi-1 ... lo (back to front); slot j of the stage = position hi - 1 - j
        const unsigned hi = n_list - rd * GS_BATCH, cnt = hi < (unsigned)GS_BATCH ? hi : (unsigned)GS_BATCH;
        __syncthreads();
        if ((unsigned)tid < cnt) stage_id[tid] = point_list[r0 + hi - 1 - tid];
        __syncthreads();
        for (unsigned e = tid; e < cnt * GS_STAGE; e += GS_BATCH) {
            const unsigned j = e / GS_STAGE, f = e - j * GS_STAGE;
            stage[e] = geom[(size_t)stage_id[j] * GS_REC + f];
        }
        __syncthreads();
        if (!__any(inside && last > 0 && hi - cnt < last)) continue;          // wave-uniform: nothing of this batch reaches this wave
        for (unsigned j = 0; j < cnt; j++) {
            const unsigned pos1 = hi - j;                                     // 1-based contributor index of this list position
            const float* g = stage + j * GS_STAGE;
            float d[GS_NGRAD];
#pragma unroll
            for (int k = 0; k < GS_NGRAD; k++) d[k] = 0.f;
            bool act = inside && pos1 <= last;
            if (act) {
                const float dx = g[G_XY] - pxf, dy = g[G_XY + 1] - pyf;
                const float power = -0.5f * (g[G_CONIC] * dx * dx + g[G_CONIC + 2] * dy * dy) - g[G_CONIC + 1] * dx * dy;
                const float Gs = expf(power);
                const float araw = g[G_OP] * Gs;
                const float alpha = fminf(0.99f, araw);
                act = !(power > 0.f) && !(alpha < 1.f / 255.f);
                if (act) {
                    T = T / (1.f - alpha);
                    const float aT = alpha * T;
                    const float t = g[G_TS] + g[G_RP] * dx + g[G_RP + 1] * dy;
                    float v = GW + GD * t, ddx = GD * g[G_RP] * aT, ddy = GD * g[G_RP + 1] * aT;
                    d[G_TS] = GD * aT; d[G_RP] = GD * aT * dx; d[G_RP + 1] = GD * aT * dy;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        const float co = g[G_VP + ch] + g[G_CP + 2 * ch] * dx + g[G_CP + 2 * ch + 1] * dy;
                        v += GC[ch] * g[G_RGB + ch] + GCo[ch] * co + GN[ch] * g[G_NRM + ch];
                        d[G_RGB + ch] = GC[ch] * aT;
                        d[G_NRM + ch] = GN[ch] * aT;
                        d[G_VP + ch] = GCo[ch] * aT;
                        d[G_CP + 2 * ch] = GCo[ch] * aT * dx;
                        d[G_CP + 2 * ch + 1] = GCo[ch] * aT * dy;
                        ddx += GCo[ch] * aT * g[G_CP + 2 * ch];
                        ddy += GCo[ch] * aT * g[G_CP + 2 * ch + 1];
                    }
                    if (pos1 == maxc) {                                        // the median contributor carries mdepth / mcoord
                        d[G_TS] += GmD; d[G_RP] += GmD * dx; d[G_RP + 1] += GmD * dy;
                        ddx += GmD * g[G_RP]; ddy += GmD * g[G_RP + 1];
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            d[G_VP + ch] += GmC[ch]; d[G_CP + 2 * ch] += GmC[ch] * dx; d[G_CP + 2 * ch + 1] += GmC[ch] * dy;
                            ddx += GmC[ch] * g[G_CP + 2 * ch]; ddy += GmC[ch] * g[G_CP + 2 * ch + 1];
                        }
                    }
                    const float dalpha = v * T - S / (1.f - alpha);
                    S += v * aT;
                    if (araw <= 0.99f) {                                       // d min(0.99, .) = 0 beyond the cap
                        d[G_OP] = Gs * dalpha;
                        const float dpow = araw * dalpha;
                        d[G_CONIC] = -0.5f * dx * dx * dpow; d[G_CONIC + 1] = -dx * dy * dpow; d[G_CONIC + 2] = -0.5f * dy * dy * dpow;
                        const float ax = (-g[G_CONIC] * dx - g[G_CONIC + 1] * dy) * dpow, ay = (-g[G_CONIC + 2] * dy - g[G_CONIC + 1] * dx) * dpow;
                        ddx += ax; ddy += ay;
                        d[G_DEPTH] = fabsf(ax) * 0.5f * (float)W + fabsf(ay) * 0.5f * (float)H;
                    }
                    d[G_XY] = ddx; d[G_XY + 1] = ddy;
                }
            }
            if (!__any(act)) continue;
            // 25 sums over the 64 lanes as a transposing butterfly: at each of five steps a lane keeps one half of its slots and hands the
            // other half to its partner, so the slots per lane halve while the lanes summed double (16 + 8 + 4 + 2 + 1 exchanges instead of
            // 25 x 6); one more exchange folds the two 32-lane halves, and lanes 0..24 issue their slot's atomic together
            float r[32];
#pragma unroll
            for (int k = 0; k < 32; k++) r[k] = k < GS_NGRAD ? d[k] : 0.f;
#define GS_BFLY(HALF, BIT)                                                                          \
            {                                                                                       \
                const bool up = (lane & BIT) != 0;                                                  \
                _Pragma("unroll") for (int k = 0; k < HALF; k++) {                                  \
                    const float send = up ? r[k] : r[k + HALF], keep = up ? r[k + HALF] : r[k];     \
                    r[k] = keep + __shfl_xor(send, BIT);                                            \
                }                                                                                   \
            }
            GS_BFLY(16, 1) GS_BFLY(8, 2) GS_BFLY(4, 4) GS_BFLY(2, 8) GS_BFLY(1, 16)
#undef GS_BFLY
            const float total = r[0] + __shfl_xor(r[0], 32);
            const int slot = ((lane & 1) << 4) | ((lane & 2) << 2) | (lane & 4) | ((lane & 8) >> 2) | ((lane & 16) >> 4);
            if (lane < 32 && slot < GS_NGRAD && total != 0.f) atomicAdd(dgeom + (size_t)stage_id[j] * GS_REC + slot, total);
        }
    }
}

// Per-Gaussian backward: the projection again, on dual numbers seeded with (mean, scale, quaternion), contracted with the record
// gradients; SH colour by its (linear) basis.  Replaces backward.cu:21-143,145-628.
__global__ __launch_bounds__(64) void gs_preprocess_bwd_kernel(int P, const float* __restrict__ means, const float* __restrict__ scales,
                                                               const float* __restrict__ rots, const float* __restrict__ opac,
                                                               const float* __restrict__ shs, int use_sh, GsCam cam,
                                                               const float* __restrict__ geom, const float* __restrict__ dgeom,
                                                               float* __restrict__ d_means, float* __restrict__ d_scales,
                                                               float* __restrict__ d_rots, float* __restrict__ d_opac, float* __restrict__ d_shs,
                                                               float* __restrict__ d_colors, float* __restrict__ d_means2D) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= P) return;
    const float* rec = geom + (size_t)i * GS_REC;
    if (!(rec[G_RADIUS] > 0.f)) return;                                        // culled in the forward pass: every gradient stays 0
    const float* dg = dgeom + (size_t)i * GS_REC;
    typedef Dual<10> D;
    D mean[3], sc[3], rt[4];
#pragma unroll
    for (int k = 0; k < 3; k++) { mean[k] = D(means[3 * i + k]); mean[k].d[k] = 1.f; sc[k] = D(scales[3 * i + k]); sc[k].d[3 + k] = 1.f; }
#pragma unroll
    for (int k = 0; k < 4; k++) { rt[k] = D(rots[4 * i + k]); rt[k].d[6 + k] = 1.f; }
    const GsProj<D> o = gs_project<D>(mean, sc, rt, cam);
    float gin[10];
#pragma unroll
    for (int k = 0; k < 10; k++) gin[k] = 0.f;
    auto acc = [&](const D& f, float gf) {
#pragma unroll
        for (int k = 0; k < 10; k++) gin[k] += gf * f.d[k];
    };
    acc(o.xy[0], dg[G_XY]); acc(o.xy[1], dg[G_XY + 1]);
    acc(o.ts, dg[G_TS]);
    acc(o.conic[0], dg[G_CONIC]); acc(o.conic[1], dg[G_CONIC + 1]); acc(o.conic[2], dg[G_CONIC + 2]);
    acc(o.coef, dg[G_OP] * opac[i]);
#pragma unroll
    for (int k = 0; k < 3; k++) { acc(o.vp[k], dg[G_VP + k]); acc(o.nrm[k], dg[G_NRM + k]); }
#pragma unroll
    for (int k = 0; k < 6; k++) acc(o.cp[k], dg[G_CP + k]);
    acc(o.rp[0], dg[G_RP]); acc(o.rp[1], dg[G_RP + 1]);
    d_opac[i] = dg[G_OP] * o.coef.v;
    if (use_sh) {
        typedef Dual<3> D3;
        D3 m3[3], b[16];
#pragma unroll
        for (int k = 0; k < 3; k++) { m3[k] = D3(means[3 * i + k]); m3[k].d[k] = 1.f; }
#pragma unroll
        for (int k = 0; k < 16; k++) b[k] = D3(0.f);
        sh_basis<D3>(m3, cam.campos, cam.deg, b);
        const int nb = (cam.deg + 1) * (cam.deg + 1);
        const unsigned clampbits = __float_as_uint(rec[G_CLAMP]);
        const float* sh = shs + (size_t)i * cam.K * 3;
        float* dsh = d_shs + (size_t)i * cam.K * 3;
        for (int ch = 0; ch < 3; ch++) {
            const float gc = (clampbits >> ch) & 1u ? 0.f : dg[G_RGB + ch];    // clamped at 0 in the forward pass: no gradient
            for (int k = 0; k < nb; k++) {
                dsh[3 * k + ch] = gc * b[k].v;
                const float w = gc * sh[3 * k + ch];
                gin[0] += w * b[k].d[0]; gin[1] += w * b[k].d[1]; gin[2] += w * b[k].d[2];
            }
        }
    } else {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) d_colors[3 * i + ch] = dg[G_RGB + ch];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { d_means[3 * i + k] = gin[k]; d_scales[3 * i + k] = gin[3 + k]; }
#pragma unroll
    for (int k = 0; k < 4; k++) d_rots[4 * i + k] = gin[6 + k];
    d_means2D[3 * i] = dg[G_XY] * 0.5f * (float)cam.W;                         // backward.cu:1002-1006: in NDC units, |.| sum in z
    d_means2D[3 * i + 1] = dg[G_XY + 1] * 0.5f * (float)cam.H;
    d_means2D[3 * i + 2] = dg[G_DEPTH];
}

// ------------------------------------------------------------------------------------------------ simple_knn.distCUDA2
// Mean squared distance of every point to its 3 nearest OTHER points (call sites hislam2/gaussian/scene/gaussian_model.py:191,313:
// the initial scale of a new Gaussian).  The extension is not vendored in the reference tree (graphdeco-inria/simple-knn): the
// published operator is restated.  Exhaustive search instead of upstream's Morton-box pruning: candidates stream through LDS in tiles
// of 256 points and every lane of the wave reads the same candidate (broadcast), 3 compare-exchanges per pair -- exact, no tree
// to build, and a keyframe's worth of points (<= ~2e5) is a few milliseconds on this chip.
// Candidates are split into `gridDim.y` chunks so that a keyframe's worth of points (5e4) still fills the chip (196 workgroups of 256
// queries would leave 60 CUs idle and one wave per SIMD): every (query block, chunk) keeps its own best three, knn3_merge_kernel folds
// the chunks.
__global__ __launch_bounds__(256) void knn3_kernel(const float* __restrict__ pts, int P, int chunk, float* __restrict__ part) {
    __shared__ float sx[256], sy[256], sz[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < P;
    const float x = live ? pts[3 * (size_t)i] : 0.f, y = live ? pts[3 * (size_t)i + 1] : 0.f, z = live ? pts[3 * (size_t)i + 2] : 0.f;
    float b0 = 3.402823466e38f, b1 = b0, b2 = b0;
    const int c0 = blockIdx.y * chunk, c1 = min(P, c0 + chunk);
    for (int base = c0; base < c1; base += 256) {
        const int j = base + threadIdx.x;
        __syncthreads();
        if (j < c1) { sx[threadIdx.x] = pts[3 * (size_t)j]; sy[threadIdx.x] = pts[3 * (size_t)j + 1]; sz[threadIdx.x] = pts[3 * (size_t)j + 2]; }
        __syncthreads();
        const int cnt = c1 - base < 256 ? c1 - base : 256;
        for (int k = 0; k < cnt; k++) {
            const float dx = sx[k] - x, dy = sy[k] - y, dz = sz[k] - z;
            float d = dx * dx + dy * dy + dz * dz;
            if (base + k == i) d = 3.402823466e38f;                            // the point itself is not its own neighbour
            const float t0 = fminf(b0, d); d = fmaxf(b0, d); b0 = t0;
            const float t1 = fminf(b1, d); d = fmaxf(b1, d); b1 = t1;
            b2 = fminf(b2, d);
        }
    }
    if (live) {
        float* o = part + ((size_t)blockIdx.y * P + i) * 3;
        o[0] = b0; o[1] = b1; o[2] = b2;
    }
}

__global__ __launch_bounds__(256) void knn3_merge_kernel(const float* __restrict__ part, int P, int chunks, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float b0 = 3.402823466e38f, b1 = b0, b2 = b0;
    for (int c = 0; c < chunks; c++) {
        const float* o = part + ((size_t)c * P + i) * 3;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float d = o[k];
            const float t0 = fminf(b0, d); d = fmaxf(b0, d); b0 = t0;
            const float t1 = fminf(b1, d); d = fmaxf(b1, d); b1 = t1;
            b2 = fminf(b2, d);
        }
    }
    out[i] = (b0 + b1 + b2) / 3.f;
}

// ------------------------------------------------------------------------------------------------ 3-NN on a uniform grid (large maps)
// The exhaustive search above costs P^2 pairs (200 ms at 800 k points).  For large P the points are binned into a uniform grid
// (cell edge = longest bounding-box extent / G, G ~ sqrt(P) / 2 capped at 160: pointmaps are surfaces, so an occupied cell then holds a
// few tens of points), sorted by cell with a counting sort, and every point searches the shells of cells around its own in growing
// Chebyshev radius R until its third-best squared distance is <= (R * edge)^2 -- every unseen point lies in a cell at least R + 1 away,
// i.e. at least R * edge from the query -- so the result is EXACT: the same three distances as the exhaustive search.
struct KnnHdr { float minx, miny, minz, cs, inv_cs; int gx, gy, gz; };

DEVINL unsigned knn_enc(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }     // order-preserving
DEVINL float knn_dec(unsigned e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e); }

// bb[0..5]: order-preserving encodings of the per-axis minima / maxima; mom[0..5] (floats behind them): per-axis sum and sum of squares
// RELATIVE TO POINT 0 (so that the squares stay small for a map far from the origin)
__global__ __launch_bounds__(256) void knn_bbox_kernel(const float* __restrict__ pts, int P, unsigned* __restrict__ bb) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f}, d[3] = {0.f, 0.f, 0.f};
    if (i < P) {
#pragma unroll
        for (int a = 0; a < 3; a++) { lo[a] = hi[a] = pts[3 * (size_t)i + a]; d[a] = lo[a] - pts[a]; }
    }
    float* mom = reinterpret_cast<float*>(bb + 6);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float l = lo[a], h = hi[a], s1 = d[a], s2 = d[a] * d[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            l = fminf(l, __shfl_xor(l, o, 64)); h = fmaxf(h, __shfl_xor(h, o, 64));
            s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64);
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(bb + a, knn_enc(l)); atomicMax(bb + 3 + a, knn_enc(h)); atomicAdd(mom + a, s1); atomicAdd(mom + 3 + a, s2); }
    }
}

// second moment pass: only the points inside the first pass's box count (bb[12..17]: its lo / hi as floats; mom2 = bb[18..23], count bb[24]).
// Ten outliers at 100 x the extent among 2e5 surface points triple sigma by themselves; trimmed to the first box, sigma is the surface's.
__global__ __launch_bounds__(256) void knn_trim_kernel(const float* __restrict__ pts, int P, unsigned* __restrict__ bb) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float* box = reinterpret_cast<const float*>(bb + 12);
    float d[3] = {0.f, 0.f, 0.f};
    bool in = i < P;
    if (i < P) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float x = pts[3 * (size_t)i + a];
            in = in && x >= box[a] && x <= box[3 + a];
            d[a] = x - pts[a];
        }
    }
    float* mom = reinterpret_cast<float*>(bb + 18);
    const float c = in ? 1.f : 0.f;
    float cnt = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float s1 = c * d[a], s2 = c * d[a] * d[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(mom + a, s1); atomicAdd(mom + 3 + a, s2); }
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(mom + 6, cnt);
}

// The grid covers the ROBUST box: per axis [mean - 3 sigma, mean + 3 sigma] intersected with the true bounding box.  A few far outliers
// (sky / far-depth pixels with conf > 0) would otherwise stretch the box, the surface would fall into a handful of cells and every thread
// of the query kernel would scan them serially (O(P^2) from global memory).  Points outside the grid are clamped into its border cells
// (knn_cell); the shell bound of the query stays valid under clamping (a clamped point is never nearer than its cell says), so the
// result is still EXACT -- the statistics only steer the speed, which is why their atomic summation order does not matter.
// pass 0: box from the moments of ALL points -> bb[12..17] (read by knn_trim_kernel); pass 1: box from the trimmed moments -> the header
__global__ void knn_header_kernel(unsigned* __restrict__ bb, const float* __restrict__ pts, int P, int G, KnnHdr* __restrict__ hdr, int pass) {
    if (threadIdx.x != 0) return;
    const float* mom = reinterpret_cast<const float*>(bb + (pass == 0 ? 6 : 18));
    const float n = pass == 0 ? (float)P : fmaxf(mom[6], 1.f);
    float lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float mean_d = mom[a] / n;
        const float var = fmaxf(mom[3 + a] / n - mean_d * mean_d, 0.f);
        const float mu = pts[a] + mean_d, sg = sqrtf(var);
        lo[a] = fmaxf(knn_dec(bb[a]), mu - 3.f * sg);
        hi[a] = fminf(knn_dec(bb[3 + a]), mu + 3.f * sg);
        if (!(hi[a] >= lo[a])) { lo[a] = knn_dec(bb[a]); hi[a] = knn_dec(bb[3 + a]); }
    }
    if (pass == 0) {
        float* box = reinterpret_cast<float*>(bb + 12);
#pragma unroll
        for (int a = 0; a < 3; a++) { box[a] = lo[a]; box[3 + a] = hi[a]; }
        return;
    }
    const float lx = lo[0], ly = lo[1], lz = lo[2];
    const float ex = hi[0] - lx, ey = hi[1] - ly, ez = hi[2] - lz;
    float cs = fmaxf(ex, fmaxf(ey, ez)) / (float)G;
    if (!(cs > 0.f)) cs = 1.f;                                   // all points identical
    KnnHdr h;
    h.minx = lx; h.miny = ly; h.minz = lz; h.cs = cs; h.inv_cs = 1.0f / cs;
    h.gx = min(G + 1, (int)(ex * h.inv_cs) + 1); h.gy = min(G + 1, (int)(ey * h.inv_cs) + 1); h.gz = min(G + 1, (int)(ez * h.inv_cs) + 1);
    *hdr = h;
}

DEVINL void knn_cell(const KnnHdr& h, float x, float y, float z, int& cx, int& cy, int& cz) {
    cx = min(h.gx - 1, max(0, (int)((x - h.minx) * h.inv_cs)));
    cy = min(h.gy - 1, max(0, (int)((y - h.miny) * h.inv_cs)));
    cz = min(h.gz - 1, max(0, (int)((z - h.minz) * h.inv_cs)));
}

__global__ __launch_bounds__(256) void knn_count_kernel(const float* __restrict__ pts, int P, const KnnHdr* __restrict__ hdr, int* __restrict__ cell_of,
                                                        unsigned* __restrict__ counts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const KnnHdr h = *hdr;
    int cx, cy, cz;
    knn_cell(h, pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], cx, cy, cz);
    const int c = (cz * h.gy + cy) * h.gx + cx;
    cell_of[i] = c;
    atomicAdd(counts + c, 1u);
}

__global__ __launch_bounds__(256) void knn_scatter_kernel(const float* __restrict__ pts, int P, const int* __restrict__ cell_of,
                                                          const unsigned* __restrict__ starts, unsigned* __restrict__ cursor,
                                                          float* __restrict__ spts, int* __restrict__ sidx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int c = cell_of[i];
    const unsigned pos = starts[c] + atomicAdd(cursor + c, 1u);
    spts[3 * (size_t)pos] = pts[3 * (size_t)i]; spts[3 * (size_t)pos + 1] = pts[3 * (size_t)i + 1]; spts[3 * (size_t)pos + 2] = pts[3 * (size_t)i + 2];
    sidx[pos] = i;
}

// one thread per point in SORTED order (neighbouring threads share cells)
__global__ __launch_bounds__(256) void knn_query_kernel(int P, const KnnHdr* __restrict__ hdr, const unsigned* __restrict__ starts,
                                                        const float* __restrict__ spts, const int* __restrict__ sidx, float* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= P) return;
    const KnnHdr h = *hdr;
    const float x = spts[3 * (size_t)t], y = spts[3 * (size_t)t + 1], z = spts[3 * (size_t)t + 2];
    int cx, cy, cz;
    knn_cell(h, x, y, z, cx, cy, cz);
    float b0 = 3.402823466e38f, b1 = b0, b2 = b0;
    // a query OUTSIDE the grid (an outlier clamped into a border cell: the grid covers the robust box) is far from everything: its shell
    // search would walk every shell of the grid, one cell at a time, before the bound lets it stop (measured: 0.3 s for ten outliers).
    // It scans the point list instead -- exact by construction, ~1 ms, and outliers are few.
    const bool outside = x < h.minx || y < h.miny || z < h.minz || x > h.minx + h.gx * h.cs || y > h.miny + h.gy * h.cs || z > h.minz + h.gz * h.cs;
    if (outside) {
        for (int j = 0; j < P; j++) {
            if (j == t) continue;
            const float dx = spts[3 * (size_t)j] - x, dy = spts[3 * (size_t)j + 1] - y, dz = spts[3 * (size_t)j + 2] - z;
            float d = dx * dx + dy * dy + dz * dz;
            const float t0 = fminf(b0, d); d = fmaxf(b0, d); b0 = t0;
            const float t1 = fminf(b1, d); d = fmaxf(b1, d); b1 = t1;
            b2 = fminf(b2, d);
        }
        out[sidx[t]] = (b0 + b1 + b2) / 3.f;
        return;
    }
    const int rmax = max(h.gx, max(h.gy, h.gz));
    for (int R = 0; R <= rmax; R++) {
        const int z0 = max(0, cz - R), z1 = min(h.gz - 1, cz + R), y0 = max(0, cy - R), y1 = min(h.gy - 1, cy + R);
        const int x0 = max(0, cx - R), x1 = min(h.gx - 1, cx + R);
        for (int zz = z0; zz <= z1; zz++)
            for (int yy = y0; yy <= y1; yy++) {
                const bool face = (zz == cz - R) || (zz == cz + R) || (yy == cy - R) || (yy == cy + R);      // whole row on the shell
                for (int xx = x0; xx <= x1; xx += (face || R == 0) ? 1 : max(1, x1 - x0)) {
                    if (!face && R > 0 && xx != cx - R && xx != cx + R) continue;                            // interior rows: the two end cells only
                    const int c = (zz * h.gy + yy) * h.gx + xx;
                    const unsigned a = starts[c], e = starts[c + 1];
                    for (unsigned j = a; j < e; j++) {
                        if ((int)j == t) continue;                                                           // the point itself
                        const float dx = spts[3 * (size_t)j] - x, dy = spts[3 * (size_t)j + 1] - y, dz = spts[3 * (size_t)j + 2] - z;
                        float d = dx * dx + dy * dy + dz * dz;
                        const float t0 = fminf(b0, d); d = fmaxf(b0, d); b0 = t0;
                        const float t1 = fminf(b1, d); d = fmaxf(b1, d); b1 = t1;
                        b2 = fminf(b2, d);
                    }
                }
            }
        const float reach = (float)R * h.cs;
        if (b2 <= reach * reach) break;
    }
    out[sidx[t]] = (b0 + b1 + b2) / 3.f;
}

// ------------------------------------------------------------------------------------------------ SSIM (the mapper's colour loss)
// hislam2/gaussian/utils/loss_utils.py:129-170 (`ssim`, 11x11 Gaussian window sigma 1.5, zero padding, per channel) as two
// separable passes through LDS: a 16x16 output tile reads its 26x26 neighbourhood of both images once, filters the five moments
// (a, b, a^2, b^2, ab) horizontally into LDS and vertically into registers.  The forward pass also stores the three partials the
// backward pass needs (dS/dmu1, dS/dE[a^2], dS/dE[ab]); the backward pass filters those maps with the same (symmetric) window:
// dL/da(p) = g * [ (w * dS/dmu1)(p) + 2 a(p) (w * dS/dE[a^2])(p) + b(p) (w * dS/dE[ab])(p) ].
constexpr int SS_T = 16, SS_R = 5, SS_IN = SS_T + 2 * SS_R;       // tile, window radius, input tile edge
__constant__ float SS_G[11] = {0.00102838f, 0.00759876f, 0.03600077f, 0.10936069f, 0.21300553f, 0.26601172f,
                               0.21300553f, 0.10936069f, 0.03600077f, 0.00759876f, 0.00102838f};

template <int NMAP>
DEVINL void ss_filter(const float (&in)[NMAP][SS_IN][SS_IN + 1], float (&tmp)[NMAP][SS_IN][SS_T + 1], float* out, int tx, int ty) {
    // horizontal: SS_IN rows x SS_T columns per map, spread over the 256 threads
    for (int e = ty * SS_T + tx; e < SS_IN * SS_T; e += SS_T * SS_T) {
        const int r = e / SS_T, c = e - r * SS_T;
#pragma unroll
        for (int m = 0; m < NMAP; m++) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) acc = fmaf(SS_G[k], in[m][r][c + k], acc);
            tmp[m][r][c] = acc;
        }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NMAP; m++) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) acc = fmaf(SS_G[k], tmp[m][ty + k][tx], acc);
        out[m] = acc;
    }
}

__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int H, int W,
                                                       float* __restrict__ smap, float* __restrict__ dmu1, float* __restrict__ dx11,
                                                       float* __restrict__ dx12) {
    __shared__ float in[5][SS_IN][SS_IN + 1];
    __shared__ float tmp[5][SS_IN][SS_T + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, ch = blockIdx.z;
    const int x0 = blockIdx.x * SS_T - SS_R, y0 = blockIdx.y * SS_T - SS_R;
    const float* pa = a + (size_t)ch * H * W;
    const float* pb = b + (size_t)ch * H * W;
    for (int e = threadIdx.x; e < SS_IN * SS_IN; e += 256) {
        const int r = e / SS_IN, c = e - r * SS_IN, y = y0 + r, x = x0 + c;
        const bool ok = y >= 0 && y < H && x >= 0 && x < W;
        const float va = ok ? pa[(size_t)y * W + x] : 0.f, vb = ok ? pb[(size_t)y * W + x] : 0.f;
        in[0][r][c] = va; in[1][r][c] = vb; in[2][r][c] = va * va; in[3][r][c] = vb * vb; in[4][r][c] = va * vb;
    }
    __syncthreads();
    float f[5];
    ss_filter<5>(in, tmp, f, tx, ty);
    const int x = blockIdx.x * SS_T + tx, y = blockIdx.y * SS_T + ty;
    if (x >= W || y >= H) return;
    const float mu1 = f[0], mu2 = f[1], s11 = f[2] - mu1 * mu1, s22 = f[3] - mu2 * mu2, s12 = f[4] - mu1 * mu2;
    const float C1 = 0.0001f, C2 = 0.0009f;
    const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s11 + s22 + C2;
    const float iB = 1.f / (B1 * B2), S = A1 * A2 * iB;
    const size_t o = (size_t)ch * H * W + (size_t)y * W + x;
    smap[o] = S;
    dmu1[o] = 2.f * mu2 * (A2 - A1) * iB - S * (2.f * mu1 / B1 - 2.f * mu1 / B2);
    dx11[o] = -S / B2;
    dx12[o] = 2.f * A1 * iB;
}

__global__ __launch_bounds__(256) void ssim_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ dmu1, const float* __restrict__ dx11,
                                                       const float* __restrict__ dx12, int H, int W, const float* __restrict__ gscale,
                                                       float* __restrict__ grad_a) {
    __shared__ float in[3][SS_IN][SS_IN + 1];
    __shared__ float tmp[3][SS_IN][SS_T + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, ch = blockIdx.z;
    const int x0 = blockIdx.x * SS_T - SS_R, y0 = blockIdx.y * SS_T - SS_R;
    const size_t plane = (size_t)ch * H * W;
    for (int e = threadIdx.x; e < SS_IN * SS_IN; e += 256) {
        const int r = e / SS_IN, c = e - r * SS_IN, y = y0 + r, x = x0 + c;
        const bool ok = y >= 0 && y < H && x >= 0 && x < W;
        const size_t o = plane + (size_t)y * W + x;
        in[0][r][c] = ok ? dmu1[o] : 0.f; in[1][r][c] = ok ? dx11[o] : 0.f; in[2][r][c] = ok ? dx12[o] : 0.f;
    }
    __syncthreads();
    float f[3];
    ss_filter<3>(in, tmp, f, tx, ty);
    const int x = blockIdx.x * SS_T + tx, y = blockIdx.y * SS_T + ty;
    if (x >= W || y >= H) return;
    const size_t o = plane + (size_t)y * W + x;
    grad_a[o] = gscale[0] * (f[0] + 2.f * a[o] * f[1] + b[o] * f[2]);
}

// ------------------------------------------------------------------------------------------------ pixel losses of the mapper
// The colour L1, inverse-depth L1 and depth-normal terms of gs_backend_per_frame.py:516-531 in one forward and one backward kernel.
//   sums[0] = sum |gt - img| (3 channels), sums[1] = sum_mask |1/d - 1/gt_d|, sums[2] = sum_mask (1 - n(d) . g), sums[3] = |mask|
//   mask = gt_d > 0.001 and d > 0.001;  n(d) = normalised cross product of the central differences of the back-projected depth
//   (zero on the image border, as the padded tensor of the tensor formulation), g = the keyframe depth's own normal (constant).
// Backward: grad_img = c_rgb sign(img - gt);  grad_d(p) = c_d(p) + sum over the four neighbours q whose normal uses d(p) of
//   h_q . d c_q / d d(p),  h_q = -(g_q - n_q (n_q . g_q)) / |c_q|,  c_q = dx_q x dy_q,  dx_q = P(q + ex) - P(q - ex),  P(r) = ray(r) d(r).
struct PixCam { float fx, fy, cx, cy; };

DEVINL void pl_ray(const PixCam& k, int x, int y, float* r) { r[0] = ((float)x - k.cx) / k.fx; r[1] = ((float)y - k.cy) / k.fy; r[2] = 1.f; }
DEVINL void pl_cross(const float* a, const float* b, float* c) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
// central differences of the back-projected points around interior pixel (x, y)
DEVINL void pl_diffs(const float* __restrict__ d, const PixCam& k, int W, int x, int y, float* dx, float* dy) {
    float ra[3], rb[3];
    pl_ray(k, x + 1, y, ra); pl_ray(k, x - 1, y, rb);
    const float da = d[(size_t)y * W + x + 1], db = d[(size_t)y * W + x - 1];
#pragma unroll
    for (int i = 0; i < 3; i++) dx[i] = ra[i] * da - rb[i] * db;
    pl_ray(k, x, y + 1, ra); pl_ray(k, x, y - 1, rb);
    const float dc = d[(size_t)(y + 1) * W + x], dd = d[(size_t)(y - 1) * W + x];
#pragma unroll
    for (int i = 0; i < 3; i++) dy[i] = ra[i] * dc - rb[i] * dd;
}

__global__ __launch_bounds__(256) void pixel_loss_fwd_kernel(const float* __restrict__ img, const float* __restrict__ gt, const float* __restrict__ d,
                                                             const float* __restrict__ gd, const float* __restrict__ gn, int H, int W, PixCam k,
                                                             float* __restrict__ sums) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int HW = H * W;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (p < HW) {
        const int y = p / W, x = p - y * W;
#pragma unroll
        for (int c = 0; c < 3; c++) v[0] += fabsf(gt[(size_t)c * HW + p] - img[(size_t)c * HW + p]);
        const float dp = d[p], gp = gd[p];
        if (gp > 0.001f && dp > 0.001f) {
            v[1] = fabsf(1.f / dp - 1.f / gp);
            float dot = 0.f;
            if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {
                float dx[3], dy[3], c[3];
                pl_diffs(d, k, W, x, y, dx, dy);
                pl_cross(dx, dy, c);
                const float il = 1.f / fmaxf(sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]), 1e-12f);
                dot = (c[0] * gn[p] + c[1] * gn[HW + p] + c[2] * gn[2 * HW + p]) * il;
            }
            v[2] = 1.f - dot;
            v[3] = 1.f;
        }
    }
    __shared__ float red[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float t = wave_sum(v[i]);
        if (lane == 0) red[wave][i] = t;
    }
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(sums + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// coef (device): [c_rgb, c_depth, c_normal] = upstream gradient times weight over the normaliser of each term
__global__ __launch_bounds__(256) void pixel_loss_bwd_kernel(const float* __restrict__ img, const float* __restrict__ gt, const float* __restrict__ d,
                                                             const float* __restrict__ gd, const float* __restrict__ gn, int H, int W, PixCam k,
                                                             const float* __restrict__ coef, float* __restrict__ g_img, float* __restrict__ g_d) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int HW = H * W;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const float c_rgb = coef[0], c_dep = coef[1], c_nrm = coef[2];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float e = img[(size_t)c * HW + p] - gt[(size_t)c * HW + p];
        g_img[(size_t)c * HW + p] = e > 0.f ? c_rgb : (e < 0.f ? -c_rgb : 0.f);
    }
    float g = 0.f;
    const float dp = d[p], gp = gd[p];
    if (gp > 0.001f && dp > 0.001f) {
        const float e = 1.f / dp - 1.f / gp;
        g = (e > 0.f ? c_dep : (e < 0.f ? -c_dep : 0.f)) * (-1.f / (dp * dp));
    }
    float rp[3];
    pl_ray(k, x, y, rp);
    // the four pixels q whose normal reads d(p):  (qx, qy, sign, which): p = q + ex, q - ex, q + ey, q - ey
    const int qx[4] = {x - 1, x + 1, x, x}, qy[4] = {y, y, y - 1, y + 1};
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const int xx = qx[n], yy = qy[n];
        if (xx < 1 || xx > W - 2 || yy < 1 || yy > H - 2) continue;
        const int q = yy * W + xx;
        if (!(gd[q] > 0.001f && d[q] > 0.001f)) continue;
        float dx[3], dy[3], c[3], dc[3];
        pl_diffs(d, k, W, xx, yy, dx, dy);
        pl_cross(dx, dy, c);
        const float len = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        if (!(len > 1e-12f)) continue;                                         // (the clamped normalisation has no gradient through the norm there)
        const float il = 1.f / len;
        const float nq[3] = {c[0] * il, c[1] * il, c[2] * il}, gq[3] = {gn[q], gn[HW + q], gn[2 * HW + q]};
        const float ng = nq[0] * gq[0] + nq[1] * gq[1] + nq[2] * gq[2];
        const float h[3] = {-(gq[0] - nq[0] * ng) * il, -(gq[1] - nq[1] * ng) * il, -(gq[2] - nq[2] * ng) * il};
        if (n < 2) pl_cross(rp, dy, dc); else pl_cross(dx, rp, dc);            // d c_q / d d(p) up to the sign of the difference
        const float sgn = (n == 0 || n == 2) ? 1.f : -1.f;
        g += c_nrm * sgn * (h[0] * dc[0] + h[1] * dc[1] + h[2] * dc[2]);
    }
    g_d[p] = g;
}

// agreement of the RENDERED normal image with the normal of the rendered depth (global_BA, gs_backend_per_frame.py:996-1001:
// `normal_loss = (1 - (render_normal * depth_to_normal(depth)).sum(0)).mean()` over ALL pixels; the depth normal is zero on the image
// border, utils.depth_to_normal / F.pad).  forward: sum[0] = sum_p (1 - N(p) . n(d)(p)).  backward with coef = upstream * weight / HW:
// g_N = -coef n(d) and g_d += coef * d(-N . n)/d d gathered from the four neighbours whose normal reads d(p) (the stencil of
// pixel_loss_bwd_kernel without its depth mask).
__global__ __launch_bounds__(256) void normal_agree_fwd_kernel(const float* __restrict__ nrm, const float* __restrict__ d, int H, int W, PixCam k,
                                                               float* __restrict__ sum) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int HW = H * W;
    float v = 0.f;
    if (p < HW) {
        const int y = p / W, x = p - y * W;
        float dot = 0.f;
        if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {
            float dx[3], dy[3], c[3];
            pl_diffs(d, k, W, x, y, dx, dy);
            pl_cross(dx, dy, c);
            const float il = 1.f / fmaxf(sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]), 1e-12f);
            dot = (c[0] * nrm[p] + c[1] * nrm[HW + p] + c[2] * nrm[2 * HW + p]) * il;
        }
        v = 1.f - dot;
    }
    __shared__ float red[4];
    const float t = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void normal_agree_bwd_kernel(const float* __restrict__ nrm, const float* __restrict__ d, int H, int W, PixCam k,
                                                               float coef, float* __restrict__ g_nrm, float* __restrict__ g_d) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int HW = H * W;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    float gn[3] = {0.f, 0.f, 0.f};
    if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {
        float dx[3], dy[3], c[3];
        pl_diffs(d, k, W, x, y, dx, dy);
        pl_cross(dx, dy, c);
        const float il = 1.f / fmaxf(sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]), 1e-12f);
#pragma unroll
        for (int i = 0; i < 3; i++) gn[i] = -coef * c[i] * il;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) g_nrm[(size_t)i * HW + p] = gn[i];
    float rp[3];
    pl_ray(k, x, y, rp);
    float g = 0.f;
    const int qx[4] = {x - 1, x + 1, x, x}, qy[4] = {y, y, y - 1, y + 1};
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const int xx = qx[n], yy = qy[n];
        if (xx < 1 || xx > W - 2 || yy < 1 || yy > H - 2) continue;
        const int q = yy * W + xx;
        float dx[3], dy[3], c[3], dc[3];
        pl_diffs(d, k, W, xx, yy, dx, dy);
        pl_cross(dx, dy, c);
        const float len = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        if (!(len > 1e-12f)) continue;
        const float il = 1.f / len;
        const float nq[3] = {c[0] * il, c[1] * il, c[2] * il}, gq[3] = {nrm[q], nrm[HW + q], nrm[2 * HW + q]};
        const float ng = nq[0] * gq[0] + nq[1] * gq[1] + nq[2] * gq[2];
        const float h[3] = {-(gq[0] - nq[0] * ng) * il, -(gq[1] - nq[1] * ng) * il, -(gq[2] - nq[2] * ng) * il};
        if (n < 2) pl_cross(rp, dy, dc); else pl_cross(dx, rp, dc);
        const float sgn = (n == 0 || n == 2) ? 1.f : -1.f;
        g += coef * sgn * (h[0] * dc[0] + h[1] * dc[1] + h[2] * dc[2]);
    }
    g_d[p] += g;
}

// densification statistics of one rendered view (gaussian_model.py:779-790 add_densification_stats + the max_radii2D update of
// gs_backend_per_frame.py:1021-1027): visible = radii > 0
__global__ __launch_bounds__(256) void densify_stats_kernel(int P, const int* __restrict__ radii, const float* __restrict__ d_means2D,
                                                            float* __restrict__ max_radii, float* __restrict__ grad_accum,
                                                            float* __restrict__ grad_accum_abs, float* __restrict__ denom) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    if (r > 0) {
        max_radii[i] = fmaxf(max_radii[i], (float)r);
        const float gx = d_means2D[3 * i], gy = d_means2D[3 * i + 1];
        grad_accum[i] += sqrtf(gx * gx + gy * gy);
        grad_accum_abs[i] += fabsf(d_means2D[3 * i + 2]);            // (:781: the norm of the remaining channel = the absolute-gradient statistic)
        denom[i] += 1.f;
    }
}

// pose refinement (gs_backend_per_frame.py:240-262): colour L1 over the covered pixels (alpha > th) and the variance of
// log d - log gt_d over the covered pixels with both depths valid.  sums[5] = {sum_a |gt - img|, |a|, sum_m diff, sum_m diff^2, |m|}.
// backward: coef[3] (device) = {c_rgb, c_var, mean diff}:  grad_img = c_rgb sign(img - gt) on a;  grad_d = c_var 2 (diff - mean) / d on m.
__global__ __launch_bounds__(256) void refine_loss_fwd_kernel(const float* __restrict__ img, const float* __restrict__ gt, const float* __restrict__ d,
                                                              const float* __restrict__ gd, const float* __restrict__ alpha, float alpha_th, int HW,
                                                              float* __restrict__ sums) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (p < HW && alpha[p] > alpha_th) {
        v[1] = 1.f;
#pragma unroll
        for (int c = 0; c < 3; c++) v[0] += fabsf(gt[(size_t)c * HW + p] - img[(size_t)c * HW + p]);
        const float dp = d[p], gp = gd[p];
        if (gp > 0.001f && dp > 0.001f) {
            const float diff = logf(dp) - logf(gp);
            v[2] = diff; v[3] = diff * diff; v[4] = 1.f;
        }
    }
    __shared__ float red[4][5];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const float t = wave_sum(v[i]);
        if (lane == 0) red[wave][i] = t;
    }
    __syncthreads();
    if (threadIdx.x < 5) atomicAdd(sums + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void refine_loss_bwd_kernel(const float* __restrict__ img, const float* __restrict__ gt, const float* __restrict__ d,
                                                              const float* __restrict__ gd, const float* __restrict__ alpha, float alpha_th, int HW,
                                                              const float* __restrict__ coef, float* __restrict__ g_img, float* __restrict__ g_d) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const bool a = alpha[p] > alpha_th;
    const float c_rgb = coef[0], c_var = coef[1], mean = coef[2];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float e = img[(size_t)c * HW + p] - gt[(size_t)c * HW + p];
        g_img[(size_t)c * HW + p] = !a ? 0.f : (e > 0.f ? c_rgb : (e < 0.f ? -c_rgb : 0.f));
    }
    const float dp = d[p], gp = gd[p];
    g_d[p] = (a && gp > 0.001f && dp > 0.001f) ? c_var * 2.f * (logf(dp) - logf(gp) - mean) / dp : 0.f;
}

}  // namespace

static int gs_fill_cam(GsCam& cam, const float* view, const float* proj, const float* campos, int W, int H, float tanx, float tany, float ks,
                       float mod, int deg, int K) {
    if (!view || !proj || W <= 0 || H <= 0 || !(tanx > 0.f) || !(tany > 0.f) || deg < 0 || deg > 3) return CUT3R_ERR_ARG;
    for (int i = 0; i < 16; i++) { cam.view[i] = view[i]; cam.proj[i] = proj[i]; }
    for (int i = 0; i < 3; i++) cam.campos[i] = campos ? campos[i] : 0.f;
    cam.W = W; cam.H = H; cam.tanx = tanx; cam.tany = tany; cam.fx = (float)W / (2.f * tanx); cam.fy = (float)H / (2.f * tany);
    cam.ks = ks; cam.mod = mod; cam.deg = deg; cam.K = K;
    return CUT3R_OK;
}

extern "C" int cut3r_gs_preprocess(int P, const float* means, const float* scales, const float* rots, const float* opacities, const float* shs,
                                   int sh_degree, int sh_coeffs, const float* colors_precomp, const float* viewmatrix_host,
                                   const float* projmatrix_host, const float* campos_host, int W, int H, float tanfovx, float tanfovy,
                                   float kernel_size, float scale_modifier, float* geom, int* radii, unsigned* tiles_touched,
                                   unsigned* offsets, void* scan_ws, long long scan_ws_bytes, void* stream) {
    if (P <= 0 || !means || !scales || !rots || !opacities || !geom || !radii || !tiles_touched || !offsets) return CUT3R_ERR_ARG;
    if (!shs && !colors_precomp) return CUT3R_ERR_ARG;
    if (shs && !colors_precomp && sh_coeffs < (sh_degree + 1) * (sh_degree + 1)) return CUT3R_ERR_ARG;
    GsCam cam;
    const int rc = gs_fill_cam(cam, viewmatrix_host, projmatrix_host, campos_host, W, H, tanfovx, tanfovy, kernel_size, scale_modifier, sh_degree,
                               sh_coeffs);
    if (rc != CUT3R_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gs_preprocess_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means, scales, rots, opacities, shs, colors_precomp, cam, geom,
                       radii, tiles_touched);
    // (no size query here: the caller sized scan_ws with cut3r_gs_workspace_bytes; the library call refuses a buffer that is too small)
    if (!scan_ws || scan_ws_bytes <= 0) return CUT3R_ERR_ARG;
    size_t need = (size_t)scan_ws_bytes;
    if (hipcub::DeviceScan::InclusiveSum(scan_ws, need, tiles_touched, offsets, P, s) != hipSuccess) return CUT3R_ERR_ARG;
    return cut3r_check_launch();
}

extern "C" long long cut3r_gs_workspace_bytes(int P, long long n_instances) {
    size_t a = 0, b = 0;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, a, (unsigned*)nullptr, (unsigned*)nullptr, P > 0 ? P : 1, (hipStream_t)0);
    if (n_instances > 0)
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned*)nullptr,
                                           (unsigned*)nullptr, (int)n_instances, 0, 64, (hipStream_t)0);
    return (long long)((a > b ? a : b) + 256);
}

extern "C" int cut3r_gs_bin(int P, const float* geom, const unsigned* offsets, long long n_instances, int W, int H, unsigned long long* keys_tmp,
                            unsigned* vals_tmp, unsigned long long* keys_sorted, unsigned* point_list, unsigned* ranges, void* sort_ws,
                            long long sort_ws_bytes, int* overflow, void* stream) {
    if (P <= 0 || !geom || !offsets || W <= 0 || H <= 0 || !ranges || n_instances < 0 || n_instances > 0x7fffffffLL) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int gx = (W + GS_TILE - 1) / GS_TILE, gy = (H + GS_TILE - 1) / GS_TILE;
    if (hipMemsetAsync(ranges, 0, sizeof(unsigned) * 2 * (size_t)gx * gy, s) != hipSuccess) return CUT3R_ERR_LAUNCH;
    if (n_instances == 0) return CUT3R_OK;
    if (!keys_tmp || !vals_tmp || !keys_sorted || !point_list || !sort_ws) return CUT3R_ERR_ARG;
    if (overflow)      // capacity mode: n_instances is a guess made without reading the count back; unused entries sort behind every tile
        hipLaunchKernelGGL(gs_pad_keys_kernel, dim3((unsigned)((n_instances + 255) / 256)), dim3(256), 0, s, (unsigned long long)n_instances,
                           (unsigned long long)(gx * gy) << 32, keys_tmp, vals_tmp);
    hipLaunchKernelGGL(gs_duplicate_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, geom, offsets, gx, (unsigned long long)n_instances, keys_tmp,
                       vals_tmp, overflow);
    int bits = 32;                                                             // tile id bits above the 32 depth bits
    for (int t = gx * gy; t > 0; t >>= 1) bits++;
    if (sort_ws_bytes <= 0) return CUT3R_ERR_ARG;
    size_t need = (size_t)sort_ws_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(sort_ws, need, keys_tmp, keys_sorted, vals_tmp, point_list, (int)n_instances, 0, bits, s) != hipSuccess)
        return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(gs_ranges_kernel, dim3((unsigned)((n_instances + 255) / 256)), dim3(256), 0, s, (unsigned long long)n_instances, keys_sorted,
                       gx * gy, ranges);
    return cut3r_check_launch();
}

extern "C" int cut3r_gs_render_forward(const unsigned* ranges, const unsigned* point_list, const float* geom, int W, int H, float tanfovx,
                                       float tanfovy, const float* bg_host, float* out_color, float* out_coord, float* out_mcoord,
                                       float* out_depth, float* out_mdepth, float* out_alpha, float* out_normal, unsigned* n_contrib, float* aux,
                                       void* stream) {
    if (!ranges || !geom || W <= 0 || H <= 0 || !bg_host || !out_color || !out_coord || !out_mcoord || !out_depth || !out_mdepth || !out_alpha ||
        !out_normal || !n_contrib || !aux || !(tanfovx > 0.f) || !(tanfovy > 0.f))
        return CUT3R_ERR_ARG;
    const int gx = (W + GS_TILE - 1) / GS_TILE, gy = (H + GS_TILE - 1) / GS_TILE;
    hipLaunchKernelGGL(gs_render_fwd_kernel, dim3(gx, gy), dim3(GS_BATCH), 0, (hipStream_t)stream, ranges, point_list, geom, W, H,
                       (float)W / (2.f * tanfovx), (float)H / (2.f * tanfovy), bg_host[0], bg_host[1], bg_host[2], out_color, out_coord, out_mcoord,
                       out_depth, out_mdepth, out_alpha, out_normal, n_contrib, aux);
    return cut3r_check_launch();
}

extern "C" int cut3r_gs_render_backward(const unsigned* ranges, const unsigned* point_list, const float* geom, int P, int W, int H, float tanfovx,
                                        float tanfovy, const float* bg_host, const unsigned* n_contrib, const float* aux, const float* out_alpha,
                                        const float* out_coord, const float* out_depth, const float* out_normal, const float* g_color,
                                        const float* g_coord, const float* g_mcoord, const float* g_depth, const float* g_mdepth,
                                        const float* g_alpha, const float* g_normal, float* dgeom, void* stream) {
    if (!ranges || !geom || P <= 0 || W <= 0 || H <= 0 || !bg_host || !n_contrib || !aux || !out_alpha || !out_coord || !out_depth || !out_normal ||
        !g_color || !g_coord || !g_mcoord || !g_depth || !g_mdepth || !g_alpha || !g_normal || !dgeom || !(tanfovx > 0.f) || !(tanfovy > 0.f))
        return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(dgeom, 0, sizeof(float) * (size_t)P * GS_REC, s) != hipSuccess) return CUT3R_ERR_LAUNCH;
    const int gx = (W + GS_TILE - 1) / GS_TILE, gy = (H + GS_TILE - 1) / GS_TILE;
    hipLaunchKernelGGL(gs_render_bwd_kernel, dim3(gx, gy), dim3(GS_BATCH), 0, s, ranges, point_list, geom, W, H, (float)W / (2.f * tanfovx),
                       (float)H / (2.f * tanfovy), bg_host[0], bg_host[1], bg_host[2], n_contrib, aux, out_alpha, out_coord, out_depth, out_normal,
                       g_color, g_coord, g_mcoord, g_depth, g_mdepth, g_alpha, g_normal, dgeom);
    return cut3r_check_launch();
}

extern "C" int cut3r_gs_preprocess_backward(int P, const float* means, const float* scales, const float* rots, const float* opacities,
                                            const float* shs, int sh_degree, int sh_coeffs, const float* viewmatrix_host,
                                            const float* projmatrix_host, const float* campos_host, int W, int H, float tanfovx, float tanfovy,
                                            float kernel_size, float scale_modifier, const float* geom, const float* dgeom, float* d_means,
                                            float* d_scales, float* d_rots, float* d_opacities, float* d_shs, float* d_colors, float* d_means2D,
                                            void* stream) {
    if (P <= 0 || !means || !scales || !rots || !opacities || !geom || !dgeom || !d_means || !d_scales || !d_rots || !d_opacities || !d_means2D)
        return CUT3R_ERR_ARG;
    if (shs ? !d_shs : !d_colors) return CUT3R_ERR_ARG;
    if (shs && sh_coeffs < (sh_degree + 1) * (sh_degree + 1)) return CUT3R_ERR_ARG;
    GsCam cam;
    const int rc = gs_fill_cam(cam, viewmatrix_host, projmatrix_host, campos_host, W, H, tanfovx, tanfovy, kernel_size, scale_modifier, sh_degree,
                               sh_coeffs);
    if (rc != CUT3R_OK) return rc;
    hipLaunchKernelGGL(gs_preprocess_bwd_kernel, dim3((P + 63) / 64), dim3(64), 0, (hipStream_t)stream, P, means, scales, rots, opacities, shs,
                       shs ? 1 : 0, cam, geom, dgeom, d_means, d_scales, d_rots, d_opacities, d_shs, d_colors, d_means2D);
    return cut3r_check_launch();
}

extern "C" int cut3r_knn3_chunks(int P) {
    if (P < 4) return 0;
    const long long blocks = (P + 255) / 256;                                // aim at >= 2048 workgroups, chunks of >= 2048 candidates
    long long c = (2048 + blocks - 1) / blocks;
    const long long cmax = (P + 2047) / 2048;
    if (c > cmax) c = cmax;
    if (c < 1) c = 1;
    return (int)c;
}

extern "C" int cut3r_knn3_mean_dist2(const float* points, int P, float* out, float* workspace, void* stream) {
    if (!points || !out || !workspace || P < 4) return CUT3R_ERR_ARG;        // three OTHER points must exist; workspace: chunks*P*3 floats
    const int chunks = cut3r_knn3_chunks(P);
    const int chunk = ((P + chunks - 1) / chunks + 255) / 256 * 256;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(knn3_kernel, dim3((P + 255) / 256, chunks), dim3(256), 0, s, points, P, chunk, workspace);
    hipLaunchKernelGGL(knn3_merge_kernel, dim3((P + 255) / 256), dim3(256), 0, s, workspace, P, chunks, out);
    return cut3r_check_launch();
}

static int knn3_grid_G(int P) {
    int G = (int)ceil(sqrt((double)P) / 2.0);
    return G < 8 ? 8 : (G > 160 ? 160 : G);
}

extern "C" long long cut3r_knn3_grid_workspace_bytes(int P) {
    if (P <= 0) return 0;
    const long long G = knn3_grid_G(P), nc = (G + 1) * (G + 1) * (G + 1);
    size_t scan = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan, (unsigned*)nullptr, (unsigned*)nullptr, (int)(nc + 1), (hipStream_t)0);
    // header + bbox | cell_of [P] | counts [nc+1] | starts [nc+1] | cursor [nc] | sorted points [P,3] | sorted index [P] | scan scratch
    return 256 + 4LL * P + 4 * (nc + 1) + 4 * (nc + 1) + 4 * nc + 12LL * P + 4LL * P + (long long)scan + 1024;
}

extern "C" int cut3r_knn3_grid_mean_dist2(const float* points, int P, float* out, void* workspace, long long workspace_bytes, void* stream) {
    if (!points || !out || !workspace || P < 4 || workspace_bytes < cut3r_knn3_grid_workspace_bytes(P)) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int G = knn3_grid_G(P);
    const long long nc = (long long)(G + 1) * (G + 1) * (G + 1);
    char* w = (char*)workspace;
    KnnHdr* hdr = (KnnHdr*)w;
    unsigned* bb = (unsigned*)(w + 64);
    int* cell_of = (int*)(w + 256);
    unsigned* counts = (unsigned*)(cell_of + P);
    unsigned* starts = counts + (nc + 1);
    unsigned* cursor = starts + (nc + 1);
    float* spts = (float*)(cursor + nc);
    int* sidx = (int*)(spts + 3 * (size_t)P);
    void* scan_ws = (void*)(((uintptr_t)(sidx + P) + 255) & ~(uintptr_t)255);
    size_t scan_bytes = (size_t)((char*)workspace + workspace_bytes - (char*)scan_ws);
    if (hipMemsetAsync(bb, 0xFF, 3 * sizeof(unsigned), s) != hipSuccess) return CUT3R_ERR_LAUNCH;          // encoded minima start at the top
    if (hipMemsetAsync(bb + 3, 0, 22 * sizeof(unsigned), s) != hipSuccess) return CUT3R_ERR_LAUNCH;         // maxima, moments, first box, trimmed moments + count
    if (hipMemsetAsync(counts, 0, sizeof(unsigned) * (size_t)(nc + 1), s) != hipSuccess) return CUT3R_ERR_LAUNCH;
    if (hipMemsetAsync(cursor, 0, sizeof(unsigned) * (size_t)nc, s) != hipSuccess) return CUT3R_ERR_LAUNCH;
    const unsigned nb = (unsigned)((P + 255) / 256);
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(nb), dim3(256), 0, s, points, P, bb);
    hipLaunchKernelGGL(knn_header_kernel, dim3(1), dim3(64), 0, s, bb, points, P, G, hdr, 0);
    hipLaunchKernelGGL(knn_trim_kernel, dim3(nb), dim3(256), 0, s, points, P, bb);
    hipLaunchKernelGGL(knn_header_kernel, dim3(1), dim3(64), 0, s, bb, points, P, G, hdr, 1);
    hipLaunchKernelGGL(knn_count_kernel, dim3(nb), dim3(256), 0, s, points, P, hdr, cell_of, counts);
    if (hipcub::DeviceScan::ExclusiveSum(scan_ws, scan_bytes, counts, starts, (int)(nc + 1), s) != hipSuccess) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(knn_scatter_kernel, dim3(nb), dim3(256), 0, s, points, P, cell_of, starts, cursor, spts, sidx);
    hipLaunchKernelGGL(knn_query_kernel, dim3(nb), dim3(256), 0, s, P, hdr, starts, spts, sidx, out);
    return cut3r_check_launch();
}

extern "C" int cut3r_ssim_forward(const float* a, const float* b, int C, int H, int W, float* ssim_map, float* d_mu1, float* d_x11, float* d_x12,
                                  void* stream) {
    if (!a || !b || !ssim_map || !d_mu1 || !d_x11 || !d_x12 || C <= 0 || H <= 0 || W <= 0) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, C), dim3(256), 0, (hipStream_t)stream, a, b, H, W, ssim_map,
                       d_mu1, d_x11, d_x12);
    return cut3r_check_launch();
}

extern "C" int cut3r_ssim_backward(const float* a, const float* b, const float* d_mu1, const float* d_x11, const float* d_x12, int C, int H,
                                   int W, const float* grad_scale, float* grad_a, void* stream) {
    if (!a || !b || !d_mu1 || !d_x11 || !d_x12 || !grad_scale || !grad_a || C <= 0 || H <= 0 || W <= 0) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(ssim_bwd_kernel, dim3((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, C), dim3(256), 0, (hipStream_t)stream, a, b, d_mu1, d_x11,
                       d_x12, H, W, grad_scale, grad_a);
    return cut3r_check_launch();
}

extern "C" int cut3r_pixel_loss_forward(const float* img, const float* gt_img, const float* depth, const float* gt_depth, const float* gt_normal,
                                        int H, int W, float fx, float fy, float cx, float cy, float* sums, void* stream) {
    if (!img || !gt_img || !depth || !gt_depth || !gt_normal || !sums || H < 3 || W < 3) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(sums, 0, 4 * sizeof(float), s) != hipSuccess) return CUT3R_ERR_LAUNCH;
    hipLaunchKernelGGL(pixel_loss_fwd_kernel, dim3((H * W + 255) / 256), dim3(256), 0, s, img, gt_img, depth, gt_depth, gt_normal, H, W,
                       PixCam{fx, fy, cx, cy}, sums);
    return cut3r_check_launch();
}

extern "C" int cut3r_pixel_loss_backward(const float* img, const float* gt_img, const float* depth, const float* gt_depth, const float* gt_normal,
                                         int H, int W, float fx, float fy, float cx, float cy, const float* coef, float* grad_img,
                                         float* grad_depth, void* stream) {
    if (!img || !gt_img || !depth || !gt_depth || !gt_normal || !coef || !grad_img || !grad_depth || H < 3 || W < 3) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(pixel_loss_bwd_kernel, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, img, gt_img, depth, gt_depth, gt_normal, H,
                       W, PixCam{fx, fy, cx, cy}, coef, grad_img, grad_depth);
    return cut3r_check_launch();
}

extern "C" int cut3r_normal_agree_forward(const float* normal, const float* depth, int H, int W, float fx, float fy, float cx, float cy,
                                          float* sum, void* stream) {
    if (!normal || !depth || !sum || H < 3 || W < 3) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(sum, 0, sizeof(float), s) != hipSuccess) return CUT3R_ERR_LAUNCH;
    hipLaunchKernelGGL(normal_agree_fwd_kernel, dim3((H * W + 255) / 256), dim3(256), 0, s, normal, depth, H, W, PixCam{fx, fy, cx, cy}, sum);
    return cut3r_check_launch();
}

extern "C" int cut3r_normal_agree_backward(const float* normal, const float* depth, int H, int W, float fx, float fy, float cx, float cy,
                                           float coef, float* grad_normal, float* grad_depth, void* stream) {
    if (!normal || !depth || !grad_normal || !grad_depth || H < 3 || W < 3) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(normal_agree_bwd_kernel, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, normal, depth, H, W,
                       PixCam{fx, fy, cx, cy}, coef, grad_normal, grad_depth);
    return cut3r_check_launch();
}

extern "C" int cut3r_gs_densify_stats(int P, const int* radii, const float* d_means2D, float* max_radii2D, float* grad_accum,
                                      float* grad_accum_abs, float* denom, void* stream) {
    if (P <= 0 || !radii || !d_means2D || !max_radii2D || !grad_accum || !grad_accum_abs || !denom) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, radii, d_means2D, max_radii2D, grad_accum,
                       grad_accum_abs, denom);
    return cut3r_check_launch();
}

extern "C" int cut3r_refine_loss_forward(const float* img, const float* gt_img, const float* depth, const float* gt_depth, const float* alpha,
                                         float alpha_th, int H, int W, float* sums, void* stream) {
    if (!img || !gt_img || !depth || !gt_depth || !alpha || !sums || H <= 0 || W <= 0) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(sums, 0, 5 * sizeof(float), s) != hipSuccess) return CUT3R_ERR_LAUNCH;
    hipLaunchKernelGGL(refine_loss_fwd_kernel, dim3((H * W + 255) / 256), dim3(256), 0, s, img, gt_img, depth, gt_depth, alpha, alpha_th, H * W, sums);
    return cut3r_check_launch();
}

extern "C" int cut3r_refine_loss_backward(const float* img, const float* gt_img, const float* depth, const float* gt_depth, const float* alpha,
                                          float alpha_th, int H, int W, const float* coef, float* grad_img, float* grad_depth, void* stream) {
    if (!img || !gt_img || !depth || !gt_depth || !alpha || !coef || !grad_img || !grad_depth || H <= 0 || W <= 0) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(refine_loss_bwd_kernel, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, img, gt_img, depth, gt_depth, alpha,
                       alpha_th, H * W, coef, grad_img, grad_depth);
    return cut3r_check_launch();
}
