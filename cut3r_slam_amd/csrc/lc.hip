// Fused loop-closure optimiser on gfx950: Adam over per-submap se(3) corrections minimising the L1 disagreement of
// overlapping pointmaps (reference: /root/reference/hislam2/track_backend.py:256-299 `loop_closure_init`, :400-461).
//
// The reference runs ~60 tiny autograd kernels per iteration for 1000-2000 iterations (launch-bound).  Here one
// iteration is TWO launches:
//   lc_accum : one streaming pass over the first/last pointmaps of every submap (2*B*N*12 bytes, HBM/L2 bound); each
//              thread folds sign(residual) x [p;1] into the 3x4 gradient of both matrices of its pair in registers,
//              wave-shuffle + LDS reduce, one partial row per block (deterministic: no float atomics).
//   lc_adam  : one wave per submap: 12 lanes sum its partial rows in fixed order, the 3x4 gradient is pulled back through
//              matrix(exp(xi)) with forward-mode duals (lie_math.h), Adam step, lane 0 writes the new 3x4 matrix.
// Loss (identical to the reference):  mean_{masked (b,n), xyz} |T_b last_b - T_{b+1} first_{b+1}|
//                                    + mean_{n, xyz} |T_{B-1} cur - cur_lc|,  T_0 = I fixed.
#include "common.h"
#include "lie_math.h"
#include "../../include/cut3r_hip.h"

namespace {
using namespace liemath;

constexpr int LC_PPT = 8;               // points per thread
constexpr int LC_BLOCK = 256;
constexpr int LC_ROW = 28;              // 12 (grad of the 'a' matrix) + 12 (grad of the 'c' matrix) + loss + pad

DEVINL void apply34(const float* __restrict__ T, float x, float y, float z, float& ox, float& oy, float& oz) {
    ox = fmaf(T[2], z, fmaf(T[1], y, fmaf(T[0], x, T[3])));
    oy = fmaf(T[6], z, fmaf(T[5], y, fmaf(T[4], x, T[7])));
    oz = fmaf(T[10], z, fmaf(T[9], y, fmaf(T[8], x, T[11])));
}
DEVINL float sgn(float r) { return r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f); }

// 25 sums over the 64 lanes as a transposing butterfly (each step a lane keeps half of its slots and hands the other half to its
// partner: 16 + 8 + 4 + 2 + 1 + 1 exchanges instead of 25 x 6); fixed order, so the result is deterministic.  acc has LC_ROW >= 25 slots.
DEVINL void lc_wave_reduce25(float* acc, int lane, float* out) {
    float r[32];
#pragma unroll
    for (int k = 0; k < 32; k++) r[k] = k < 25 ? acc[k] : 0.f;
#define LC_BFLY(HALF, BIT)                                                                  \
    {                                                                                       \
        const bool up = (lane & BIT) != 0;                                                  \
        _Pragma("unroll") for (int k = 0; k < HALF; k++) {                                  \
            const float send = up ? r[k] : r[k + HALF], keep = up ? r[k + HALF] : r[k];     \
            r[k] = keep + __shfl_xor(send, BIT);                                            \
        }                                                                                   \
    }
    LC_BFLY(16, 1) LC_BFLY(8, 2) LC_BFLY(4, 4) LC_BFLY(2, 8) LC_BFLY(1, 16)
#undef LC_BFLY
    const float total = r[0] + __shfl_xor(r[0], 32);
    const int slot = ((lane & 1) << 4) | ((lane & 2) << 2) | (lane & 4) | ((lane & 8) >> 2) | ((lane & 16) >> 4);
    if (lane < 32 && slot < 25) out[slot] = total;
}

// pair p in [0, B-2]: a = T_p * last_p, c = T_{p+1} * first_{p+1} (masked);  pair B-1: a = T_{B-1} * cur, c = cur_lc (fixed)
__global__ __launch_bounds__(LC_BLOCK) void lc_accum_kernel(const float* __restrict__ first, const float* __restrict__ last,
                                                            long long sub_stride, const unsigned char* __restrict__ mask,
                                                            const float* __restrict__ cur, const float* __restrict__ cur_lc,
                                                            const float* __restrict__ T, int B, int N, float w_fl, float w_cur,
                                                            float* __restrict__ partial, int nblk) {
    __shared__ float red[4][LC_ROW];
    const int p = blockIdx.y;
    const bool is_cur = (p == B - 1);
    const float* pa = is_cur ? cur : last + (size_t)p * sub_stride;
    const float* pc = is_cur ? cur_lc : first + (size_t)(p + 1) * sub_stride;
    const float* Ta = T + 12 * (is_cur ? (B - 1) : p);
    const float* Tc = T + 12 * (is_cur ? 0 : (p + 1));
    const float w = is_cur ? w_cur : w_fl;
    float acc[LC_ROW];
#pragma unroll
    for (int k = 0; k < LC_ROW; k++) acc[k] = 0.f;
    const int base = blockIdx.x * (LC_BLOCK * LC_PPT) + threadIdx.x;
#pragma unroll
    for (int it = 0; it < LC_PPT; it++) {
        const int n = base + it * LC_BLOCK;
        if (n >= N) break;
        if (!is_cur && mask && !mask[(size_t)p * N + n]) continue;
        const float ax = pa[3 * (size_t)n], ay = pa[3 * (size_t)n + 1], az = pa[3 * (size_t)n + 2];
        const float cx = pc[3 * (size_t)n], cy = pc[3 * (size_t)n + 1], cz = pc[3 * (size_t)n + 2];
        float a0, a1, a2, c0, c1, c2;
        apply34(Ta, ax, ay, az, a0, a1, a2);
        if (is_cur) { c0 = cx; c1 = cy; c2 = cz; }
        else apply34(Tc, cx, cy, cz, c0, c1, c2);
        const float r0 = a0 - c0, r1 = a1 - c1, r2 = a2 - c2;
        const float s0 = sgn(r0), s1 = sgn(r1), s2 = sgn(r2);
        acc[24] += fabsf(r0) + fabsf(r1) + fabsf(r2);
        acc[0] += s0 * ax; acc[1] += s0 * ay; acc[2] += s0 * az; acc[3] += s0;
        acc[4] += s1 * ax; acc[5] += s1 * ay; acc[6] += s1 * az; acc[7] += s1;
        acc[8] += s2 * ax; acc[9] += s2 * ay; acc[10] += s2 * az; acc[11] += s2;
        if (!is_cur) {
            acc[12] -= s0 * cx; acc[13] -= s0 * cy; acc[14] -= s0 * cz; acc[15] -= s0;
            acc[16] -= s1 * cx; acc[17] -= s1 * cy; acc[18] -= s1 * cz; acc[19] -= s1;
            acc[20] -= s2 * cx; acc[21] -= s2 * cy; acc[22] -= s2 * cz; acc[23] -= s2;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lc_wave_reduce25(acc, lane, red[wave]);
    __syncthreads();
    if (threadIdx.x < 25) {
        const int k = threadIdx.x;
        partial[((size_t)p * nblk + blockIdx.x) * LC_ROW + k] = (red[0][k] + red[1][k] + red[2][k] + red[3][k]) * w;
    }
}

struct AdamHyper { float lr, b1, b2, eps; };

// total loss of one iteration from the partial rows (column 24), fixed order: lane l sums rows l, l+64, ... and the 64 lane
// sums are folded by wave_sum -- for logging / convergence tests only (it never feeds the optimiser)
DEVINL void lc_loss_row(const float* __restrict__ partial, int rows, float* __restrict__ loss_out, int step) {
    float L = 0.f;
    for (int r = threadIdx.x; r < rows; r += 64) L += partial[(size_t)r * LC_ROW + 24];
    L = wave_sum(L);
    if (threadIdx.x == 0) loss_out[step] = L;
}

// one wave per optimised submap b = 1..B-1  (b = 0 is the fixed identity; its wave reports the loss when asked to).
// Lanes 0..11 each sum one gradient element over the partial rows in block order (the same order a single thread would
// use), the 12 sums are broadcast and every lane carries the (tiny) dual-number pull-back; lane 0 writes.
__global__ __launch_bounds__(64) void lc_adam_kernel(const float* __restrict__ partial, int nblk, int B, float* __restrict__ xi,
                                                     float* __restrict__ m, float* __restrict__ v, float* __restrict__ T,
                                                     float* __restrict__ loss_out, int step, AdamHyper h) {
    const int b = blockIdx.x;
    if (b == 0) {
        if (loss_out) lc_loss_row(partial, B * nblk, loss_out, step);
        return;
    }
    float G[12];
    {
        // 'a' role: pair b (b <= B-2: last_b; b == B-1: the current term); 'c' role: pair b-1
        const int e = threadIdx.x < 12 ? threadIdx.x : 0;
        const float* ra = partial + (size_t)b * nblk * LC_ROW + e;
        const float* rc = partial + (size_t)(b - 1) * nblk * LC_ROW + 12 + e;
        float g = 0.f;
        for (int k0 = 0; k0 < nblk; k0 += 8) {       // (same sums, eight rows' loads in flight: see lc_terms_adam_kernel)
            float va[8], vc[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int k = k0 + j < nblk ? k0 + j : nblk - 1;
                va[j] = ra[(size_t)k * LC_ROW];
                vc[j] = rc[(size_t)k * LC_ROW];
            }
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (k0 + j < nblk) g += va[j] + vc[j];
        }
#pragma unroll
        for (int k = 0; k < 12; k++) G[k] = __shfl(g, k);
    }
    typedef Dual<6> D;
    D a[6], X[7], M[16];
#pragma unroll
    for (int k = 0; k < 6; k++) { a[k] = D(xi[(size_t)b * 6 + k]); a[k].d[k] = 1.f; }
    f_exp<1, D>(a, X);
    f_matrix<1, D>(X, M);
    const float bc1 = 1.f - powf(h.b1, (float)(step + 1)), bc2 = 1.f - powf(h.b2, (float)(step + 1));
    float nx[6], nm[6], nv[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        float g = 0.f;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) g += G[r * 4 + c] * M[r * 4 + c].d[j];
        const float mj = h.b1 * m[(size_t)b * 6 + j] + (1.f - h.b1) * g;
        const float vj = h.b2 * v[(size_t)b * 6 + j] + (1.f - h.b2) * g * g;
        nm[j] = mj;
        nv[j] = vj;
        // torch.optim.Adam: step_size = lr / bc1 ; denom = sqrt(v)/sqrt(bc2) + eps
        nx[j] = xi[(size_t)b * 6 + j] - (h.lr / bc1) * mj / (sqrtf(vj) / sqrtf(bc2) + h.eps);
    }
    float Xf[7], Mf[16];
    f_exp<1, float>(nx, Xf);
    f_matrix<1, float>(Xf, Mf);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < 6; j++) { m[(size_t)b * 6 + j] = nm[j]; v[(size_t)b * 6 + j] = nv[j]; xi[(size_t)b * 6 + j] = nx[j]; }
#pragma unroll
        for (int k = 0; k < 12; k++) T[(size_t)b * 12 + k] = Mf[k];
    }
}

// ------------------------------------------------------------------------------------------------ general term list
// Later loop closures (track_backend.py:400-461) add two families of residuals to the chain term: every re-tracked "lc"
// submap k has its own se(3) (matched_lie) and is tied (i) by its first map to the first map of the submap it matched and (ii)
// by its last map to the current keyframe's map as seen from that keyframe's submap.  All of them are the same shape:
//     w * sum_n | T[ia] a_n - T[ic] c_n |_1          (T[0] = identity, fixed; every other transform is a parameter)
// so one kernel evaluates a LIST of such terms (device array of LcTerm) and the Adam kernel gathers, per parameter, the
// partial rows of the terms that reference it, in term order (deterministic).
struct LcTerm {
    const float* a; const float* c; const unsigned char* mask;
    int ia, ic; float w; int pad;
};

__global__ __launch_bounds__(LC_BLOCK) void lc_terms_accum_kernel(const LcTerm* __restrict__ terms, const float* __restrict__ T, int N,
                                                                  float* __restrict__ partial, int nblk) {
    __shared__ float red[4][LC_ROW];
    const LcTerm tm = terms[blockIdx.y];
    const float* Ta = T + 12 * tm.ia;
    const float* Tc = T + 12 * tm.ic;
    float acc[LC_ROW];
#pragma unroll
    for (int k = 0; k < LC_ROW; k++) acc[k] = 0.f;
    const int base = blockIdx.x * (LC_BLOCK * LC_PPT) + threadIdx.x;
#pragma unroll
    for (int it = 0; it < LC_PPT; it++) {
        const int n = base + it * LC_BLOCK;
        if (n >= N) break;
        if (tm.mask && !tm.mask[n]) continue;
        const float ax = tm.a[3 * (size_t)n], ay = tm.a[3 * (size_t)n + 1], az = tm.a[3 * (size_t)n + 2];
        const float cx = tm.c[3 * (size_t)n], cy = tm.c[3 * (size_t)n + 1], cz = tm.c[3 * (size_t)n + 2];
        float a0, a1, a2, c0, c1, c2;
        apply34(Ta, ax, ay, az, a0, a1, a2);
        apply34(Tc, cx, cy, cz, c0, c1, c2);
        const float r0 = a0 - c0, r1 = a1 - c1, r2 = a2 - c2;
        const float s0 = sgn(r0), s1 = sgn(r1), s2 = sgn(r2);
        acc[24] += fabsf(r0) + fabsf(r1) + fabsf(r2);
        acc[0] += s0 * ax; acc[1] += s0 * ay; acc[2] += s0 * az; acc[3] += s0;
        acc[4] += s1 * ax; acc[5] += s1 * ay; acc[6] += s1 * az; acc[7] += s1;
        acc[8] += s2 * ax; acc[9] += s2 * ay; acc[10] += s2 * az; acc[11] += s2;
        acc[12] -= s0 * cx; acc[13] -= s0 * cy; acc[14] -= s0 * cz; acc[15] -= s0;
        acc[16] -= s1 * cx; acc[17] -= s1 * cy; acc[18] -= s1 * cz; acc[19] -= s1;
        acc[20] -= s2 * cx; acc[21] -= s2 * cy; acc[22] -= s2 * cz; acc[23] -= s2;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lc_wave_reduce25(acc, lane, red[wave]);
    __syncthreads();
    if (threadIdx.x < 25) {
        const int k = threadIdx.x;
        partial[((size_t)blockIdx.y * nblk + blockIdx.x) * LC_ROW + k] = (red[0][k] + red[1][k] + red[2][k] + red[3][k]) * tm.w;
    }
}

// one wave per transform p = 1..P-1 (p = 0 is the fixed identity and reports the loss): lanes 0..11 gather, in term order,
// the partial rows of the terms that reference p
__global__ __launch_bounds__(64) void lc_terms_adam_kernel(const LcTerm* __restrict__ terms, int n_terms, const float* __restrict__ partial,
                                                           int nblk, int P, float* __restrict__ xi, float* __restrict__ m,
                                                           float* __restrict__ v, float* __restrict__ T, float* __restrict__ loss_out,
                                                           int step, AdamHyper h) {
    const int b = blockIdx.x;
    if (b == 0) {
        if (loss_out) lc_loss_row(partial, n_terms * nblk, loss_out, step);
        return;
    }
    float G[12];
    {
        const int e = threadIdx.x < 12 ? threadIdx.x : 0;
        float g = 0.f;
        for (int t = 0; t < n_terms; t++) {
            const int ia = terms[t].ia, ic = terms[t].ic;
            if (ia != b && ic != b) continue;
            const float* row = partial + (size_t)t * nblk * LC_ROW + e;
            // the same additions in the same order (row k: the 'a' entry, then the 'c' entry), with the loads of eight rows requested
            // before the first of them is added: the gather was a chain of ~3 x nblk dependent L2 round trips per iteration
            const bool ua = ia == b, uc = ic == b;
            for (int k0 = 0; k0 < nblk; k0 += 8) {
                float va[8], vc[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int k = k0 + j < nblk ? k0 + j : nblk - 1;
                    va[j] = ua ? row[(size_t)k * LC_ROW] : 0.f;
                    vc[j] = uc ? row[(size_t)k * LC_ROW + 12] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (k0 + j < nblk) {
                        if (ua) g += va[j];
                        if (uc) g += vc[j];
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 12; k++) G[k] = __shfl(g, k);
    }
    typedef Dual<6> D;
    D a[6], X[7], M[16];
#pragma unroll
    for (int k = 0; k < 6; k++) { a[k] = D(xi[(size_t)b * 6 + k]); a[k].d[k] = 1.f; }
    f_exp<1, D>(a, X);
    f_matrix<1, D>(X, M);
    const float bc1 = 1.f - powf(h.b1, (float)(step + 1)), bc2 = 1.f - powf(h.b2, (float)(step + 1));
    float nx[6], nm[6], nv[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        float g = 0.f;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) g += G[r * 4 + c] * M[r * 4 + c].d[j];
        const float mj = h.b1 * m[(size_t)b * 6 + j] + (1.f - h.b1) * g;
        const float vj = h.b2 * v[(size_t)b * 6 + j] + (1.f - h.b2) * g * g;
        nm[j] = mj;
        nv[j] = vj;
        nx[j] = xi[(size_t)b * 6 + j] - (h.lr / bc1) * mj / (sqrtf(vj) / sqrtf(bc2) + h.eps);
    }
    float Xf[7], Mf[16];
    f_exp<1, float>(nx, Xf);
    f_matrix<1, float>(Xf, Mf);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < 6; j++) { m[(size_t)b * 6 + j] = nm[j]; v[(size_t)b * 6 + j] = nv[j]; xi[(size_t)b * 6 + j] = nx[j]; }
#pragma unroll
        for (int k = 0; k < 12; k++) T[(size_t)b * 12 + k] = Mf[k];
    }
}

// in-place p <- T_b p over every pointmap of submap b (the rewrite at track_backend.py:306-310)
__global__ __launch_bounds__(256) void transform_submaps_kernel(float* __restrict__ pts, const float* __restrict__ T, long long per_sub) {
    const int b = blockIdx.y;
    const float* Tb = T + 12 * b;
    float* p = pts + (size_t)b * per_sub * 3;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)per_sub; i += (size_t)gridDim.x * blockDim.x) {
        float ox, oy, oz;
        apply34(Tb, p[3 * i], p[3 * i + 1], p[3 * i + 2], ox, oy, oz);
        p[3 * i] = ox; p[3 * i + 1] = oy; p[3 * i + 2] = oz;
    }
}

}  // namespace

extern "C" int cut3r_lc_workspace_floats(int B, int N) {
    const int nblk = (N + LC_BLOCK * LC_PPT - 1) / (LC_BLOCK * LC_PPT);
    return B * nblk * LC_ROW;
}

extern "C" int cut3r_lc_optimize(const float* first, const float* last, long long sub_stride, const unsigned char* mask,
                                 const float* cur, const float* cur_lc, int B, int N, long long n_masked, int iters, float lr,
                                 float* xi, float* adam_m, float* adam_v, float* T, float* workspace, float* loss_out, void* stream) {
    if (!first || !last || !cur || !cur_lc || !xi || !adam_m || !adam_v || !T || !workspace) return CUT3R_ERR_ARG;
    if (B < 2 || N <= 0 || iters < 0 || n_masked < 0) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (N + LC_BLOCK * LC_PPT - 1) / (LC_BLOCK * LC_PPT);
    const float w_fl = n_masked > 0 ? 1.0f / (3.0f * (float)n_masked) : 0.f;
    const float w_cur = 1.0f / (3.0f * (float)N);
    AdamHyper h{lr, 0.9f, 0.999f, 1e-8f};
    for (int it = 0; it < iters; it++) {
        hipLaunchKernelGGL(lc_accum_kernel, dim3(nblk, B), dim3(LC_BLOCK), 0, s, first, last, sub_stride, mask, cur, cur_lc, T, B, N,
                           w_fl, w_cur, workspace, nblk);
        hipLaunchKernelGGL(lc_adam_kernel, dim3(B), dim3(64), 0, s, workspace, nblk, B, xi, adam_m, adam_v, T, loss_out, it, h);
    }
    return cut3r_check_launch();
}

extern "C" int cut3r_lc_optimize_terms(const void* terms_dev, int n_terms, int P, int N, int iters, float lr, float* xi, float* adam_m,
                                       float* adam_v, float* T, float* workspace, float* loss_out, void* stream) {
    static_assert(sizeof(LcTerm) == sizeof(cut3r_lc_term), "cut3r_lc_term layout");
    if (!terms_dev || !xi || !adam_m || !adam_v || !T || !workspace) return CUT3R_ERR_ARG;
    if (n_terms < 1 || P < 2 || N <= 0 || iters < 0) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (N + LC_BLOCK * LC_PPT - 1) / (LC_BLOCK * LC_PPT);
    AdamHyper h{lr, 0.9f, 0.999f, 1e-8f};
    const LcTerm* terms = (const LcTerm*)terms_dev;
    for (int it = 0; it < iters; it++) {
        hipLaunchKernelGGL(lc_terms_accum_kernel, dim3(nblk, n_terms), dim3(LC_BLOCK), 0, s, terms, T, N, workspace, nblk);
        hipLaunchKernelGGL(lc_terms_adam_kernel, dim3(P), dim3(64), 0, s, terms, n_terms, workspace, nblk, P, xi, adam_m, adam_v, T,
                           loss_out, it, h);
    }
    return cut3r_check_launch();
}

extern "C" int cut3r_transform_submaps(float* pts, const float* T, int B, long long points_per_submap, void* stream) {
    if (!pts || !T || B <= 0 || points_per_submap <= 0) return CUT3R_ERR_ARG;
    int gx = (int)((points_per_submap + 255) / 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(transform_submaps_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, pts, T, points_per_submap);
    return cut3r_check_launch();
}
