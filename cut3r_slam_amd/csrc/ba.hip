// Dense bundle adjustment operators (row A13 of SURVEY.md section 8) on gfx950 -- the DROID-style stack that the
// reference keeps only as dead Python (/root/reference/hislam2/geom/{projective_ops,pinhole,ba,chol}.py,
// /root/reference/hislam2/modules/corr.py) on top of the absent `droid_backends` CUDA sources.
//
//   corr_index fwd/bwd   bilinear lookup of a (2r+1)^2 window in an all-pairs correlation volume  (corr.py:6-21)
//   ba_edge              per source frame: reprojection residuals, 2x6 pose Jacobians (via the transposed adjoint of
//                        G_ij), depth Jacobian; emits the 6x6 Hessian blocks / gradient as per-block partials and the
//                        E rows, C, w per pixel  (projective_ops.py:44-74, ba.py:43-74)
//   ba_reduce_H          fixed-order sum of the partials into H [P,P,6,6], v [P,6]
//   ba_schur             S = H + damping - E Q E^T,  v' = v - E Q w          (chol.py:47-65)
//   ba_chol_solve        in-LDS Cholesky + two triangular solves (n = 6P <= 192), failure => dx = 0 (chol.py:9-18)
//   ba_dz                dz = Q (w - E^T dx)                                 (chol.py:71)
// All reductions are deterministic (no float atomics): every (source frame, pixel) is owned by one thread that walks
// that frame's edges in order, block partials are summed in fixed order.
#include "common.h"
#include "lie_math.h"
#include "../../include/cut3r_hip.h"

namespace {
using namespace liemath;

// ------------------------------------------------------------------------------------------------ correlation lookup
// volume [BN,h1,w1,h2,w2], coords [BN,2,h1,w1] (x,y), out [BN,rd,rd,h1,w1] with out[n,i,j,y,x] = bilinear(volume[n,y,x],
// x0 - r + i, y0 - r + j), zero outside (upstream DROID-SLAM correlation_kernels.cu semantics; i = x offset, j = y offset).
__global__ __launch_bounds__(256) void corr_index_fwd_kernel(const float* __restrict__ vol, const float* __restrict__ coords,
                                                             float* __restrict__ out, int BN, int h1, int w1, int h2, int w2, int r) {
    const int rd = 2 * r + 1;
    const size_t total = (size_t)BN * h1 * w1;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(p % w1);
        const int y = (int)((p / w1) % h1);
        const int n = (int)(p / ((size_t)w1 * h1));
        const float x0 = coords[(((size_t)n * 2 + 0) * h1 + y) * w1 + x];
        const float y0 = coords[(((size_t)n * 2 + 1) * h1 + y) * w1 + x];
        const float fx = floorf(x0), fy = floorf(y0);
        const float dx = x0 - fx, dy = y0 - fy;
        const float* v = vol + p * (size_t)h2 * w2;
        for (int i = 0; i < rd; i++)
            for (int j = 0; j < rd; j++) {
                const int xa = (int)fx - r + i, ya = (int)fy - r + j;
                float s = 0.f;
#pragma unroll
                for (int oy = 0; oy < 2; oy++)
#pragma unroll
                    for (int ox = 0; ox < 2; ox++) {
                        const int xx = xa + ox, yy = ya + oy;
                        if (xx >= 0 && xx < w2 && yy >= 0 && yy < h2)
                            s += v[(size_t)yy * w2 + xx] * (ox ? dx : 1.f - dx) * (oy ? dy : 1.f - dy);
                    }
                out[((((size_t)n * rd + i) * rd + j) * h1 + y) * w1 + x] = s;
            }
    }
}

// each (n,y,x) owns its own h2 x w2 slice of grad_volume -> no races, plain read-modify-write
__global__ __launch_bounds__(256) void corr_index_bwd_kernel(const float* __restrict__ coords, const float* __restrict__ gout,
                                                             float* __restrict__ gvol, int BN, int h1, int w1, int h2, int w2, int r) {
    const int rd = 2 * r + 1;
    const size_t total = (size_t)BN * h1 * w1;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(p % w1);
        const int y = (int)((p / w1) % h1);
        const int n = (int)(p / ((size_t)w1 * h1));
        const float x0 = coords[(((size_t)n * 2 + 0) * h1 + y) * w1 + x];
        const float y0 = coords[(((size_t)n * 2 + 1) * h1 + y) * w1 + x];
        const float fx = floorf(x0), fy = floorf(y0);
        const float dx = x0 - fx, dy = y0 - fy;
        float* v = gvol + p * (size_t)h2 * w2;
        for (int i = 0; i < rd; i++)
            for (int j = 0; j < rd; j++) {
                const float g = gout[((((size_t)n * rd + i) * rd + j) * h1 + y) * w1 + x];
                const int xa = (int)fx - r + i, ya = (int)fy - r + j;
#pragma unroll
                for (int oy = 0; oy < 2; oy++)
#pragma unroll
                    for (int ox = 0; ox < 2; ox++) {
                        const int xx = xa + ox, yy = ya + oy;
                        if (xx >= 0 && xx < w2 && yy >= 0 && yy < h2)
                            v[(size_t)yy * w2 + xx] += g * (ox ? dx : 1.f - dx) * (oy ? dy : 1.f - dy);
                    }
            }
    }
}

// ------------------------------------------------------------------------------------------------ BA: per-edge assembly
constexpr int BA_BLOCK = 256;
constexpr int BA_HROW = 120;        // Hii(36) Hij(36) Hjj(36) vi(6) vj(6)   (Hji = Hij^T)

// the same for the 42 sums of a Schur block (64 slots: 32 + 16 + 8 + 4 + 2 + 1 exchanges; one slot per lane at the end)
DEVINL void ba_wave_reduce42(float* acc, int lane, float* out) {
    float r[64];
#pragma unroll
    for (int k = 0; k < 64; k++) r[k] = k < 42 ? acc[k] : 0.f;
#define BA_BFLY(HALF, BIT)                                                                  \
    {                                                                                       \
        const bool up = (lane & BIT) != 0;                                                  \
        _Pragma("unroll") for (int k = 0; k < HALF; k++) {                                  \
            const float send = up ? r[k] : r[k + HALF], keep = up ? r[k + HALF] : r[k];     \
            r[k] = keep + __shfl_xor(send, BIT);                                            \
        }                                                                                   \
    }
    BA_BFLY(32, 1) BA_BFLY(16, 2) BA_BFLY(8, 4) BA_BFLY(4, 8) BA_BFLY(2, 16) BA_BFLY(1, 32)
#undef BA_BFLY
    const int slot = ((lane & 1) << 5) | ((lane & 2) << 3) | ((lane & 4) << 1) | ((lane & 8) >> 1) | ((lane & 16) >> 3) | ((lane & 32) >> 5);
    if (slot < 42) out[slot] = r[0];
}

// the 120 per-wave sums of an edge's Hessian blocks as ONE transposing butterfly over 128 slots (at each of six steps a lane keeps half
// of its slots and hands the other half to its partner: 64 + 32 + 16 + 8 + 4 + 2 = 126 exchanges instead of 120 x 6 shuffle-adds);
// every lane ends with the wave totals of two slots.  Fixed order: deterministic.
DEVINL void ba_wave_reduce120(float* loc, int lane, float* out) {
    float r[128];
#pragma unroll
    for (int k = 0; k < 128; k++) r[k] = k < BA_HROW ? loc[k] : 0.f;
#define BA_BFLY(HALF, BIT)                                                                  \
    {                                                                                       \
        const bool up = (lane & BIT) != 0;                                                  \
        _Pragma("unroll") for (int k = 0; k < HALF; k++) {                                  \
            const float send = up ? r[k] : r[k + HALF], keep = up ? r[k + HALF] : r[k];     \
            r[k] = keep + __shfl_xor(send, BIT);                                            \
        }                                                                                   \
    }
    BA_BFLY(64, 1) BA_BFLY(32, 2) BA_BFLY(16, 4) BA_BFLY(8, 8) BA_BFLY(4, 16) BA_BFLY(2, 32)
#undef BA_BFLY
    const int slot = ((lane & 1) << 6) | ((lane & 2) << 4) | ((lane & 4) << 2) | (lane & 8) | ((lane & 16) >> 2) | ((lane & 32) >> 4);
    if (slot < BA_HROW) out[slot] = r[0];
    if (slot + 1 < BA_HROW) out[slot + 1] = r[1];
}


struct BaGeom {
    int P, ht, wd, N, M, fixedp;
};

// poses [P,7] (world->camera, t q_xyzw), Gij [N,7] = G_j * G_i^-1, disps [P,HW], intr [P,4], target/weight [N,HW,2]
// CSR over source frames: src_ptr [M+1], src_edges [*] (edge ids), kx [M] (frame id of source m)
__global__ __launch_bounds__(BA_BLOCK) void ba_edge_kernel(const float* __restrict__ Gij, const float* __restrict__ disps,
                                                           const float* __restrict__ intr, const float* __restrict__ target,
                                                           const float* __restrict__ weight, const int* __restrict__ ii,
                                                           const int* __restrict__ jj, const int* __restrict__ src_ptr,
                                                           const int* __restrict__ src_edges, const int* __restrict__ kx, BaGeom g,
                                                           float* __restrict__ Hpart, float* __restrict__ E, float* __restrict__ Cm,
                                                           float* __restrict__ wm, const float* __restrict__ eta) {
    __shared__ float red[4][BA_HROW];
    const int m = blockIdx.y;
    const int HW = g.ht * g.wd;
    const int k = blockIdx.x * BA_BLOCK + threadIdx.x;
    const bool ok = k < HW;
    const int i = kx[m];
    const int Pf = g.P - g.fixedp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = ok ? k % g.wd : 0, py = ok ? k / g.wd : 0;
    const float fxi = intr[i * 4 + 0], fyi = intr[i * 4 + 1], cxi = intr[i * 4 + 2], cyi = intr[i * 4 + 3];
    const float d0 = ok ? disps[(size_t)i * HW + k] : 1.f;
    const float X0x = ((float)px - cxi) / fxi, X0y = ((float)py - cyi) / fyi;
    float Ei_sum[6] = {0, 0, 0, 0, 0, 0};
    float Csum = 0.f, wsum = 0.f;
    const int ip = i - g.fixedp;
    for (int s = src_ptr[m]; s < src_ptr[m + 1]; s++) {
        const int e = src_edges[s];
        const int j = jj[e];
        const int jp = j - g.fixedp;
        const float* G = Gij + (size_t)e * 7;
        const V3<float> t = {G[0], G[1], G[2]};
        const Q4<float> q = {G[3], G[4], G[5], G[6]};
        const Q4<float> qi = qconj(q);
        const float fxj = intr[j * 4 + 0], fyj = intr[j * 4 + 1], cxj = intr[j * 4 + 2], cyj = intr[j * 4 + 3];
        // X1 = G_ij * [X0, 1, d]
        const V3<float> RX = qrot(q, V3<float>{X0x, X0y, 1.f});
        const float X = RX.x + t.x * d0, Y = RX.y + t.y * d0, Z = RX.z + t.z * d0;
        const float Zc = (Z < 0.1f) ? 1.f : Z;
        const float dz = 1.0f / Zc;
        const float u = fxj * (X * dz) + cxj, v = fyj * (Y * dz) + cyj;
        const float valid = (ok && Z > 0.2f) ? 1.f : 0.f;          // X0.z == 1 > MIN_DEPTH always
        float r[2], w[2];
        if (ok) {
            r[0] = target[((size_t)e * HW + k) * 2 + 0] - u;
            r[1] = target[((size_t)e * HW + k) * 2 + 1] - v;
            w[0] = 0.001f * valid * weight[((size_t)e * HW + k) * 2 + 0];
            w[1] = 0.001f * valid * weight[((size_t)e * HW + k) * 2 + 1];
        } else { r[0] = r[1] = w[0] = w[1] = 0.f; }
        // Jp rows (2x3 part), Ja = [d I, -[X1]x]  (actp for SE3)
        const float jp0[3] = {fxj * dz, 0.f, -fxj * X * dz * dz};
        const float jp1[3] = {0.f, fyj * dz, -fyj * Y * dz * dz};
        float Jj[2][6], Ji[2][6], Jz[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const float* a = c ? jp1 : jp0;
            Jj[c][0] = a[0] * d0; Jj[c][1] = a[1] * d0; Jj[c][2] = a[2] * d0;
            // a^T * (-[X1]x) : columns 3..5 of Ja are (0,Z,-Y),(-Z,0,X),(Y,-X,0) as column vectors of rows
            Jj[c][3] = -a[1] * Z + a[2] * Y;
            Jj[c][4] = a[0] * Z - a[2] * X;
            Jj[c][5] = -a[0] * Y + a[1] * X;
            // Ji = -adjT_{Gij}(Jj):  [R^T tau ; R^T phi - R^T (t x tau)]
            const V3<float> tau = {Jj[c][0], Jj[c][1], Jj[c][2]}, phi = {Jj[c][3], Jj[c][4], Jj[c][5]};
            const V3<float> a_tau = qrot(qi, tau);
            const V3<float> a_phi = add(qrot(qi, phi), scale(-1.f, qrot(qi, cross(t, tau))));
            Ji[c][0] = -a_tau.x; Ji[c][1] = -a_tau.y; Ji[c][2] = -a_tau.z;
            Ji[c][3] = -a_phi.x; Ji[c][4] = -a_phi.y; Ji[c][5] = -a_phi.z;
            Jz[c] = a[0] * t.x + a[1] * t.y + a[2] * t.z;
        }
        // per-pixel depth terms
        float Ej[6];
#pragma unroll
        for (int a = 0; a < 6; a++) {
            Ei_sum[a] += w[0] * Ji[0][a] * Jz[0] + w[1] * Ji[1][a] * Jz[1];
            Ej[a] = w[0] * Jj[0][a] * Jz[0] + w[1] * Jj[1][a] * Jz[1];
        }
        Csum += w[0] * Jz[0] * Jz[0] + w[1] * Jz[1] * Jz[1];
        wsum += w[0] * r[0] * Jz[0] + w[1] * r[1] * Jz[1];
        if (ok && jp >= 0) {
            float* Eb = E + (((size_t)jp * g.M + m) * 6) * HW + k;
#pragma unroll
            for (int a = 0; a < 6; a++) Eb[(size_t)a * HW] += Ej[a];        // same thread owns (m,k): ordered, race-free
        }
        // 6x6 blocks + gradients, reduced over the block's pixels
        float loc[BA_HROW];
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
            for (int b = 0; b < 6; b++) {
                loc[a * 6 + b] = w[0] * Ji[0][a] * Ji[0][b] + w[1] * Ji[1][a] * Ji[1][b];
                loc[36 + a * 6 + b] = w[0] * Ji[0][a] * Jj[0][b] + w[1] * Ji[1][a] * Jj[1][b];
                loc[72 + a * 6 + b] = w[0] * Jj[0][a] * Jj[0][b] + w[1] * Jj[1][a] * Jj[1][b];
            }
            loc[108 + a] = w[0] * Ji[0][a] * r[0] + w[1] * Ji[1][a] * r[1];
            loc[114 + a] = w[0] * Jj[0][a] * r[0] + w[1] * Jj[1][a] * r[1];
        }
        ba_wave_reduce120(loc, lane, red[wave]);
        __syncthreads();
        if (threadIdx.x < BA_HROW)
            Hpart[((size_t)e * gridDim.x + blockIdx.x) * BA_HROW + threadIdx.x] =
                red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        __syncthreads();
    }
    if (ok) {
        if (ip >= 0) {
            float* Eb = E + (((size_t)ip * g.M + m) * 6) * HW + k;
#pragma unroll
            for (int a = 0; a < 6; a++) Eb[(size_t)a * HW] += Ei_sum[a];
        }
        Cm[(size_t)m * HW + k] = eta ? Csum + eta[(size_t)m * HW + k] + 1e-7f : Csum;      // eta == NULL: raw sums (proj_trans)
        wm[(size_t)m * HW + k] = wsum;
    }
    (void)Pf;
}

// the pixel-block partial rows of every edge folded first (block order, one thread per edge and entry): Hedge [N, BA_HROW], written over
// the first partial row of each edge
__global__ __launch_bounds__(128) void ba_fold_edges_kernel(float* __restrict__ Hpart, int nblk) {
    const int e = blockIdx.x, c = threadIdx.x;
    if (c >= BA_HROW) return;
    float* row = Hpart + (size_t)e * nblk * BA_HROW;
    float s = 0.f;
    for (int k = 0; k < nblk; k++) s += row[k * BA_HROW + c];
    row[c] = s;                                                            // (thread c only ever reads column c)
}

// H [Pf,Pf,6,6], v [Pf,6]: one thread per output element, edges in order (the same sums in the same order as folding inside this loop)
__global__ void ba_reduce_H_kernel(const float* __restrict__ Hpart, int nblk, const int* __restrict__ ii, const int* __restrict__ jj,
                                   int N, int Pf, int fixedp, float* __restrict__ H, float* __restrict__ v) {
    // block = one (p,q) pair [or the gradient row when q == Pf]; thread = one of the 36 (or 6) entries
    const int p = blockIdx.x, q = blockIdx.y, tix = threadIdx.x;
    const size_t estride = (size_t)nblk * BA_HROW;
    if (q < Pf) {
        if (tix >= 36) return;
        const int a = tix / 6, b = tix % 6;
        float s = 0.f;
        for (int e = 0; e < N; e++) {
            const int i = ii[e] - fixedp, j = jj[e] - fixedp;
            const float* row = Hpart + (size_t)e * estride;
            float c = 0.f;
            bool hit = false;
            if (i == p && i == q && i >= 0) { c += row[a * 6 + b]; hit = true; }
            if (i == p && j == q && i >= 0 && j >= 0) { c += row[36 + a * 6 + b]; hit = true; }
            if (j == p && i == q && i >= 0 && j >= 0) { c += row[36 + b * 6 + a]; hit = true; }
            if (j == p && j == q && j >= 0) { c += row[72 + a * 6 + b]; hit = true; }
            if (hit) s += c;
        }
        H[(((size_t)p * Pf + q) * 6 + a) * 6 + b] = s;
    } else {
        if (tix >= 6) return;
        float s = 0.f;
        for (int e = 0; e < N; e++) {
            const int i = ii[e] - fixedp, j = jj[e] - fixedp;
            const float* row = Hpart + (size_t)e * estride;
            if (i == p) s += row[108 + tix];
            if (j == p) s += row[114 + tix];
        }
        v[(size_t)p * 6 + tix] = s;
    }
}

// S[(p,a),(q,b)] = Hd[(p,a),(q,b)] - sum_m sum_k E[p,m,a,k] E[q,m,b,k] / C[m,k];  vS[(p,a)] = v - sum E w / C
// one block per (p,q); 36 (+6 when q == p) reductions over M*HW pixels.
__global__ __launch_bounds__(256) void ba_schur_kernel(const float* __restrict__ H, const float* __restrict__ v,
                                                       const float* __restrict__ E, const float* __restrict__ Cm,
                                                       const float* __restrict__ wm, const unsigned char* __restrict__ present,
                                                       int Pf, int M, int HW, float ep, float lm, float* __restrict__ S,
                                                       float* __restrict__ vS) {
    __shared__ float red[4][42];
    const int p = blockIdx.x, q = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[42];
#pragma unroll
    for (int c = 0; c < 42; c++) acc[c] = 0.f;
    for (int m = 0; m < M; m++) {
        if (!present[p * M + m] || !present[q * M + m]) continue;
        const float* Ep = E + (((size_t)p * M + m) * 6) * HW;
        const float* Eq = E + (((size_t)q * M + m) * 6) * HW;
        for (int k = threadIdx.x; k < HW; k += 256) {
            const float Q = 1.0f / Cm[(size_t)m * HW + k];
            float ea[6], eb[6];
#pragma unroll
            for (int a = 0; a < 6; a++) { ea[a] = Ep[(size_t)a * HW + k]; eb[a] = Eq[(size_t)a * HW + k] * Q; }
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int b = 0; b < 6; b++) acc[a * 6 + b] += ea[a] * eb[b];
            if (p == q) {
                const float wq = wm[(size_t)m * HW + k] * Q;
#pragma unroll
                for (int a = 0; a < 6; a++) acc[36 + a] += ea[a] * wq;
            }
        }
    }
    ba_wave_reduce42(acc, lane, red[wave]);
    __syncthreads();
    const int n = Pf * 6;
    if (threadIdx.x < 36) {
        const int a = threadIdx.x / 6, b = threadIdx.x % 6;
        float h = H[(((size_t)p * Pf + q) * 6 + a) * 6 + b];
        if (p == q && a == b) h = h + (ep + lm * h);                       // H + (ep + lm*H) * I   (chol.py:56-57)
        S[(size_t)(p * 6 + a) * n + (q * 6 + b)] = h - (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
    } else if (threadIdx.x < 42 && p == q) {
        const int a = threadIdx.x - 36;
        vS[p * 6 + a] = v[p * 6 + a] - (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
    }
}

// in-LDS Cholesky (lower) of S [n,n] (n <= 192) and solve S x = b.  flag[0] = 1 on a non-positive pivot (x = 0).
// Blocked, one thread per ROW, panels of CB = 8 columns, three barriers per PANEL:
//   * every thread factors the panel's 8x8 diagonal block itself, in registers (broadcast LDS reads, ~150 flops): nobody waits for
//     a designated thread and no barrier separates it from the next stage;
//   * a thread below the panel solves its own 8 panel entries against that block (registers), the panel rows store theirs;  -- barrier
//   * every thread updates its own row of the trailing matrix with the 8 panel columns at once (the partner row's 8 entries are a
//     broadcast read; rows padded to n + 1 floats so threads walking down a column hit different banks).                -- barrier
// The triangular solves are blocked the same way (the panel's 8 unknowns are solved redundantly by every thread).
// History (n = 132, tools/bench_ba.py): element-cyclic with idx / rem arithmetic and three barriers per column 587 us; one thread per row,
// column by column 242 us; this form: see profiles.
constexpr int CHOL_MAXN = 192;
constexpr int CB = 8;
__global__ __launch_bounds__(256) void ba_chol_solve_kernel(const float* __restrict__ S, const float* __restrict__ b, int n,
                                                            float* __restrict__ x, float* __restrict__ Lout, int* __restrict__ flag) {
    extern __shared__ float sm[];
    const int ld = n + 1;
    float* A = sm;                 // n rows of ld floats
    float* ys = sm + n * ld;       // n (+CB): right-hand side entries as the panels see them
    float* xs = ys + n + CB;       // n (+CB): solved entries
    __shared__ int bad;
    const int tid = threadIdx.x;
    if (tid == 0) bad = 0;
    for (int i = tid; i < n * n; i += 256) { const int r = i / n, c = i - r * n; A[r * ld + c] = S[i]; }
    __syncthreads();
    float* row = A + (tid < n ? tid : 0) * ld;
    int mybad = 0;
    for (int k0 = 0; k0 < n; k0 += CB) {
        // ---- the diagonal block, in registers (entries beyond n: identity)
        float D[CB][CB];
#pragma unroll
        for (int r = 0; r < CB; r++)
#pragma unroll
            for (int c = 0; c <= r; c++) D[r][c] = (k0 + r < n) ? A[(k0 + r) * ld + k0 + c] : (r == c ? 1.f : 0.f);
#pragma unroll
        for (int c = 0; c < CB; c++) {
            const float d = D[c][c];
            const bool okp = d > 0.f;
            if (!okp) mybad = 1;
            const float dc = sqrtf(okp ? d : 1.f);
            D[c][c] = dc;
#pragma unroll
            for (int r = c + 1; r < CB; r++) D[r][c] = D[r][c] / dc;
#pragma unroll
            for (int r = c + 1; r < CB; r++)
#pragma unroll
                for (int c2 = c + 1; c2 <= r; c2++) D[r][c2] = fmaf(-D[r][c], D[c2][c], D[r][c2]);
        }
        // ---- panel entries of this thread's row
        float a[CB];
#pragma unroll
        for (int c = 0; c < CB; c++) a[c] = 0.f;
        const bool below = tid >= k0 + CB && tid < n;
        if (below) {
#pragma unroll
            for (int c = 0; c < CB; c++) a[c] = row[k0 + c];
#pragma unroll
            for (int c = 0; c < CB; c++) {
                float v = a[c];
#pragma unroll
                for (int c2 = 0; c2 < c; c2++) v = fmaf(-a[c2], D[c][c2], v);
                a[c] = v / D[c][c];
            }
        }
        __syncthreads();                                                   // every thread has read the old diagonal block / its old panel entries
        if (below) {
#pragma unroll
            for (int c = 0; c < CB; c++) row[k0 + c] = a[c];
        }
#pragma unroll
        for (int r = 0; r < CB; r++)
            if (tid == k0 + r && tid < n) {
#pragma unroll
                for (int c = 0; c <= r; c++) row[k0 + c] = D[r][c];
            }
        __syncthreads();                                                   // the panel columns are final
        // ---- trailing update of this thread's row: columns k0 + CB .. tid
        if (below) {
            int j = k0 + CB;
            for (; j + 3 <= tid; j += 4) {                                 // four columns at a time: all 36 LDS reads before the first write
                float l[4][CB], acc[4];                                    // (row / A alias as far as the compiler knows)
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    acc[u] = row[j + u];
#pragma unroll
                    for (int c = 0; c < CB; c++) l[u][c] = A[(j + u) * ld + k0 + c];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
#pragma unroll
                    for (int c = 0; c < CB; c++) acc[u] = fmaf(-a[c], l[u][c], acc[u]);
                    row[j + u] = acc[u];
                }
            }
            for (; j <= tid; j++) {
                const float* lj = A + j * ld + k0;
                float acc = row[j];
#pragma unroll
                for (int c = 0; c < CB; c++) acc = fmaf(-a[c], lj[c], acc);
                row[j] = acc;
            }
        }
        __syncthreads();
    }
    if (mybad && tid == 0) bad = 1;                                        // (every thread saw the same pivots)
    // ---- forward: L y = b, panel by panel; y[i] lives in thread i's register until its panel comes up
    float yi = tid < n ? b[tid] : 0.f;
    for (int i = tid; i < n + CB; i += 256) { ys[i] = 0.f; xs[i] = 0.f; }
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += CB) {
        if (tid >= k0 && tid < k0 + CB && tid < n) ys[tid] = yi;
        __syncthreads();
        float z[CB];
#pragma unroll
        for (int c = 0; c < CB; c++) {                                     // the panel's unknowns, solved by every thread for itself
            float v = ys[k0 + c];
#pragma unroll
            for (int c2 = 0; c2 < c; c2++) v = fmaf(-((k0 + c < n) ? A[(k0 + c) * ld + k0 + c2] : 0.f), z[c2], v);
            z[c] = (k0 + c < n) ? v / A[(k0 + c) * ld + k0 + c] : 0.f;
        }
        if (tid >= k0 + CB && tid < n) {
#pragma unroll
            for (int c = 0; c < CB; c++) yi = fmaf(-row[k0 + c], z[c], yi);
        }
#pragma unroll
        for (int c = 0; c < CB; c++)
            if (tid == k0 + c && tid < n) { yi = z[c]; xs[tid] = z[c]; }
    }
    __syncthreads();
    // ---- backward: L^T x = y, panels from the end; thread i accumulates  y_i - sum_{j > i's panel} L_ji x_j
    if (tid < n) yi = xs[tid];
    __syncthreads();
    const int last0 = ((n - 1) / CB) * CB;
    for (int k0 = last0; k0 >= 0; k0 -= CB) {
        if (tid >= k0 && tid < k0 + CB && tid < n) ys[tid] = yi;
        __syncthreads();
        float z[CB];
#pragma unroll
        for (int c = CB - 1; c >= 0; c--) {
            float v = ys[k0 + c];
#pragma unroll
            for (int c2 = CB - 1; c2 > c; c2--) v = fmaf(-((k0 + c2 < n) ? A[(k0 + c2) * ld + k0 + c] : 0.f), z[c2], v);
            z[c] = (k0 + c < n) ? v / A[(k0 + c) * ld + k0 + c] : 0.f;
        }
        if (tid < k0) {
#pragma unroll
            for (int c = 0; c < CB; c++)
                if (k0 + c < n) yi = fmaf(-A[(k0 + c) * ld + tid], z[c], yi);
        }
#pragma unroll
        for (int c = 0; c < CB; c++)
            if (tid == k0 + c && tid < n) xs[tid] = z[c];
        __syncthreads();
    }
    if (tid < n) x[tid] = bad ? 0.f : xs[tid];
    if (Lout)
        for (int i = tid; i < n * n; i += 256) { const int r = i / n, c = i % n; Lout[i] = (c <= r) ? A[r * ld + c] : 0.f; }
    if (tid == 0) flag[0] = bad;
}

// dz[m,k] = (w - sum_p E[p,m,:,k] . dx[p]) / C
__global__ __launch_bounds__(256) void ba_dz_kernel(const float* __restrict__ E, const float* __restrict__ Cm, const float* __restrict__ wm,
                                                    const float* __restrict__ dx, const unsigned char* __restrict__ present, int Pf,
                                                    int M, int HW, float* __restrict__ dz) {
    const int m = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= HW) return;
    float s = wm[(size_t)m * HW + k];
    for (int p = 0; p < Pf; p++) {
        if (!present[p * M + m]) continue;
        const float* Ep = E + (((size_t)p * M + m) * 6) * HW + k;
#pragma unroll
        for (int a = 0; a < 6; a++) s -= Ep[(size_t)a * HW] * dx[p * 6 + a];
    }
    dz[(size_t)m * HW + k] = s / Cm[(size_t)m * HW + k];
}

// diagonal of the reduced pose Hessian (for the damping, which acts on the SUM over all ranks' edges)
__global__ void ba_hdiag_kernel(const float* __restrict__ H, int Pf, float* __restrict__ hd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Pf * 6) return;
    const int p = i / 6, a = i % 6;
    hd[i] = H[(((size_t)p * Pf + p) * 6 + a) * 6 + a];
}
// Sd = S + diag(ep + lm * hdiag)   (chol.py:56-57 on the reduced system: the Schur correction does not touch the damping)
__global__ void ba_damp_kernel(const float* __restrict__ S, const float* __restrict__ hd, int n, float ep, float lm, float* __restrict__ Sd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * n) return;
    const int r = i / n, c = i % n;
    Sd[i] = S[i] + ((r == c) ? (ep + lm * hd[r]) : 0.f);
}

// ------------------------------------------------------------------------------------------------ mono-prior Schur solve (geom/chol.py:80-107)
// S = H - E diag(1/C) E^T (undamped), vS = v - E (w / C), hdiag = diag(H) for H [n,n], E [n,cols], C, w [cols].  One workgroup per 16 x 16
// block of S: the two 16-row slabs of E go through LDS 64 columns at a time together with 1/C; thread (i, j) keeps S[i][j].  Blocks on the
// block diagonal also produce vS (threads j == 0) and hdiag.
__global__ __launch_bounds__(256) void mp_reduce_kernel(const float* __restrict__ H, const float* __restrict__ E, const float* __restrict__ Cd,
                                                        const float* __restrict__ wd, const float* __restrict__ v, int n, long long cols,
                                                        float* __restrict__ S, float* __restrict__ vS, float* __restrict__ hd) {
    __shared__ float Ea[16][65], Eb[16][65], Qs[64], Ws[64];
    const int bi = blockIdx.y * 16, bj = blockIdx.x * 16;
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
    float acc = 0.f, accv = 0.f;
    for (long long c0 = 0; c0 < cols; c0 += 64) {
        for (int e = threadIdx.x; e < 16 * 64; e += 256) {
            const int r = e >> 6, c = e & 63;
            const long long col = c0 + c;
            Ea[r][c] = (bi + r < n && col < cols) ? E[(size_t)(bi + r) * cols + col] : 0.f;
            Eb[r][c] = (bj + r < n && col < cols) ? E[(size_t)(bj + r) * cols + col] : 0.f;
        }
        if (threadIdx.x < 64) {
            const long long col = c0 + threadIdx.x;
            const float q = col < cols ? 1.0f / Cd[col] : 0.f;
            Qs[threadIdx.x] = q;
            Ws[threadIdx.x] = col < cols ? q * wd[col] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int c = 0; c < 64; c++) acc = fmaf(Ea[ti][c] * Qs[c], Eb[tj][c], acc);
        if (bi == bj && tj == 0) {
#pragma unroll 8
            for (int c = 0; c < 64; c++) accv = fmaf(Ea[ti][c], Ws[c], accv);
        }
        __syncthreads();
    }
    const int i = bi + ti, j = bj + tj;
    if (i < n && j < n) S[(size_t)i * n + j] = H[(size_t)i * n + j] - acc;
    if (bi == bj && tj == 0 && i < n) {
        vS[i] = v[i] - accv;
        hd[i] = H[(size_t)i * n + i];
    }
}

// per column c: dz = (w - E[:,c] . dso) / C; with cov != NULL also dzcov = sum_i x_i^2 + 1/C where L x = E[:,c] / C (forward substitution:
// L in LDS as packed rows, x_i kept in the column's own slot of `xs` [n, cols] -- coalesced across the threads of a workgroup)
__global__ __launch_bounds__(256) void mp_backsub_kernel(const float* __restrict__ E, const float* __restrict__ Cd, const float* __restrict__ wd,
                                                         const float* __restrict__ dso, const float* __restrict__ L, const int* __restrict__ flag,
                                                         int n, long long cols, float* __restrict__ dz, float* __restrict__ cov,
                                                         float* __restrict__ xs) {
    extern __shared__ float Ls[];                  // packed lower triangle: row i at i (i + 1) / 2
    float* ds = Ls + (cov ? (size_t)n * (n + 1) / 2 : 0);
    const int failed = flag[0];
    if (cov)
        for (int e = threadIdx.x; e < n * n; e += 256) {
            const int r = e / n, c = e - r * n;
            if (c <= r) Ls[(size_t)r * (r + 1) / 2 + c] = L[e];
        }
    for (int e = threadIdx.x; e < n; e += 256) ds[e] = failed ? 0.f : dso[e];
    __syncthreads();
    const long long col = blockIdx.x * (long long)256 + threadIdx.x;
    if (col >= cols) return;
    const float q = 1.0f / Cd[col];
    float s = wd[col];
    for (int i = 0; i < n; i++) s = fmaf(-E[(size_t)i * cols + col], ds[i], s);
    dz[col] = q * s;
    if (!cov) return;
    float sq = 0.f;
    if (!failed) {                                 // (CholeskySolver returns zeros on failure, chol.py:13-18: the covariance is then 1/C)
        for (int i = 0; i < n; i++) {
            const float* Li = Ls + (size_t)i * (i + 1) / 2;
            float x = E[(size_t)i * cols + col] * q;
            for (int j = 0; j < i; j++) x = fmaf(-Li[j], xs[(size_t)j * cols + col], x);
            x /= Li[i];
            xs[(size_t)i * cols + col] = x;
            sq = fmaf(x, x, sq);
        }
    }
    cov[col] = sq + q;
}

// JDSA blocks of source frame m (geom/ba.py:213-228): Jso[p,a] = -mask_p prior_p Jbi[m,p,a];  H_m = alpha Jso^T Jso [D,D] on the block
// diagonal of H [n,n], E_m = alpha Jso^T [D,HW] on the block diagonal of E [n, M*HW], v_m = -alpha Jso^T rd.  One workgroup per frame;
// H, E, v are written completely (zeros off the diagonal blocks).
__global__ __launch_bounds__(256) void jdsa_blocks_kernel(const float* __restrict__ prior, const float* __restrict__ Jbi, const float* __restrict__ rd,
                                                          float alpha, int M, int HW, int D, float* __restrict__ H, float* __restrict__ E,
                                                          float* __restrict__ v) {
    extern __shared__ float red[];                 // [256] partial sums
    const int m = blockIdx.x, n = M * D;
    const size_t cols = (size_t)M * HW;
    // E rows of this frame: zero everywhere except its own column range
    for (int a = 0; a < D; a++) {
        float* Er = E + (size_t)(m * D + a) * cols;
        for (size_t c = threadIdx.x; c < cols; c += 256) {
            float val = 0.f;
            if (c >= (size_t)m * HW && c < (size_t)(m + 1) * HW) {
                const size_t p = c - (size_t)m * HW;
                const float pr = prior[(size_t)m * HW + p];
                val = alpha * (pr > 0.f ? -pr * Jbi[((size_t)m * HW + p) * D + a] : 0.f);
            }
            Er[c] = val;
        }
    }
    for (int a = 0; a < D; a++) {
        for (int b = 0; b <= D; b++) {             // b == D: the right-hand side
            float acc = 0.f;
            for (int p = threadIdx.x; p < HW; p += 256) {
                const float pr = prior[(size_t)m * HW + p];
                if (pr > 0.f) {
                    const float ja = -pr * Jbi[((size_t)m * HW + p) * D + a];
                    acc = fmaf(alpha * ja, b < D ? -pr * Jbi[((size_t)m * HW + p) * D + b] : -rd[(size_t)m * HW + p], acc);
                }
            }
            red[threadIdx.x] = acc;
            __syncthreads();
            for (int st = 128; st > 0; st >>= 1) {
                if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                if (b < D) H[(size_t)(m * D + a) * n + m * D + b] = red[0];
                else v[m * D + a] = red[0];
            }
            __syncthreads();
        }
        for (int j = threadIdx.x; j < n; j += 256)
            if (j < m * D || j >= (m + 1) * D) H[(size_t)(m * D + a) * n + j] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ bi_inter (geom/ba.py:160-170)
// scales [M,hs,ws], grid [M,ht,wd,2] (x, y in scale-grid units) -> vals [M,ht,wd] (bilinear), J [M,ht,wd,hs*ws] (d val / d node)
__global__ __launch_bounds__(256) void bi_inter_kernel(const float* __restrict__ scales, const float* __restrict__ grid, int M, int hs, int ws,
                                                       int HW, float* __restrict__ vals, float* __restrict__ J) {
    const size_t p = blockIdx.x * (size_t)256 + threadIdx.x;
    if (p >= (size_t)M * HW) return;
    const int m = (int)(p / HW);
    const float gx = grid[2 * p], gy = grid[2 * p + 1];
    const float fx = floorf(gx), fy = floorf(gy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float dx = gx - fx, dy = gy - fy;
    const float* sc = scales + (size_t)m * hs * ws;
    float* Jp = J + p * (size_t)(hs * ws);
    for (int i = 0; i < hs * ws; i++) Jp[i] = 0.f;
    float v = 0.f;
#pragma unroll
    for (int oy = 0; oy < 2; oy++)
#pragma unroll
        for (int ox = 0; ox < 2; ox++) {
            const int xx = x0 + ox, yy = y0 + oy;
            const float w = (ox ? dx : 1.f - dx) * (oy ? dy : 1.f - dy);
            if (xx >= 0 && xx < ws && yy >= 0 && yy < hs) {
                v += w * sc[yy * ws + xx];
                Jp[yy * ws + xx] += w;
            }
        }
    vals[p] = v;
}

// ------------------------------------------------------------------------------------------------ depth_filter
// droid_backends.depth_filter (call site hislam2/util/droid_visualization.py:100; droid_backends itself is absent from the
// reference tree: restated from the published DROID-SLAM kernel `depth_filter_kernel`, src/droid_kernels.cu).  For frame
// ix = inds[m] and each of its six neighbours jx in {ix-1, ix-2, ix-3, ix+3, ix+4, ix+5} (upstream's `neigh < 3 ? ix - neigh - 1 :
// ix + neigh`) inside [0, n): a pixel is carried into jx with the relative pose T_j T_i^-1 and its inverse depth; it counts
// once for that neighbour when its depth agrees within thresh[m] with one of the four stored depths around the landing point.
// count [M,ht,wd] float, whole numbers.  One thread per (m, pixel) walks the six neighbours: no atomics.
__global__ __launch_bounds__(256) void depth_filter_kernel(const float* __restrict__ poses, const float* __restrict__ disps,
                                                           const float* __restrict__ intr, const long long* __restrict__ inds,
                                                           const float* __restrict__ thresh, int n, int ht, int wd,
                                                           float* __restrict__ count) {
    const int m = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= ht * wd) return;
    const int i = p / wd, j = p - i * wd;
    const int ix = (int)inds[m];
    const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    const float t = thresh[m];
    const size_t HW = (size_t)ht * wd;
    const float di = disps[(size_t)ix * HW + p];
    const float Xi[4] = {((float)j - cx) / fx, ((float)i - cy) / fy, 1.f, di};
    float Ti_inv[7];
    f_inv<1, float>(poses + (size_t)ix * 7, Ti_inv);
    float c = 0.f;
    for (int neigh = 0; neigh < 6; neigh++) {
        const int jx = neigh < 3 ? ix - neigh - 1 : ix + neigh;
        if (jx < 0 || jx >= n) continue;
        float Tij[7], Xj[4];
        f_mul<1, float>(poses + (size_t)jx * 7, Ti_inv, Tij);
        f_act<1, float>(Tij, Xi, 4, Xj);
        const float uj = fx * (Xj[0] / Xj[2]) + cx, vj = fy * (Xj[1] / Xj[2]) + cy, dj = Xj[3] / Xj[2];
        const float fu = floorf(uj), fv = floorf(vj);
        if (!(fu >= 0.f && fv >= 0.f && fu < (float)(wd - 1) && fv < (float)(ht - 1))) continue;
        const int u0 = (int)fu, v0 = (int)fv;
        const float* dn = disps + (size_t)jx * HW + (size_t)v0 * wd + u0;
        const float zi = 1.f / dj;
        if (fabsf(zi - 1.f / dn[0]) < t || fabsf(zi - 1.f / dn[1]) < t || fabsf(zi - 1.f / dn[wd]) < t || fabsf(zi - 1.f / dn[wd + 1]) < t)
            c += 1.f;
    }
    count[(size_t)m * HW + p] = c;
}

// ------------------------------------------------------------------------------------------------ alt-corr (modules/corr.py:74-139)
// fmap1 [BN,H,W,C], fmap2 [BN,H2,W2,C], coords [BN,S,H,W,2] (x,y in fmap2 pixels) -> corr [BN,S,(2r+1)^2,H,W]:
// corr[n,s,(i,j),y,x] = sum_c fmap1[n,y,x,c] * bilinear(fmap2[n])(coords + (i - r, j - r))[c]  (x offset = i, y offset = j,
// zero outside) -- the on-the-fly form of CorrBlock: identical values to the lookup in the all-pairs volume.
__global__ __launch_bounds__(64) void altcorr_fwd_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                         const float* __restrict__ coords, int BN, int S, int H, int W, int H2, int W2,
                                                         int C, int r, float* __restrict__ corr) {
    const int rd = 2 * r + 1;
    const size_t pix = blockIdx.x;                     // (n, s, y, x)
    const int x = (int)(pix % W), y = (int)((pix / W) % H), s_ = (int)((pix / ((size_t)W * H)) % S), n = (int)(pix / ((size_t)W * H * S));
    const int lane = threadIdx.x;
    const float cx = coords[pix * 2], cy = coords[pix * 2 + 1];
    const float fx = floorf(cx), fy = floorf(cy);
    const float dx = cx - fx, dy = cy - fy;
    const float* a = f1 + (((size_t)n * H + y) * W + x) * C;
    for (int i = 0; i < rd; i++)
        for (int j = 0; j < rd; j++) {
            const int xa = (int)fx - r + i, ya = (int)fy - r + j;
            float acc = 0.f;
#pragma unroll
            for (int oy = 0; oy < 2; oy++)
#pragma unroll
                for (int ox = 0; ox < 2; ox++) {
                    const int xx = xa + ox, yy = ya + oy;
                    if (xx < 0 || xx >= W2 || yy < 0 || yy >= H2) continue;
                    const float w = (ox ? dx : 1.f - dx) * (oy ? dy : 1.f - dy);
                    const float* b = f2 + (((size_t)n * H2 + yy) * W2 + xx) * C;
                    float d = 0.f;
                    for (int c = lane; c < C; c += 64) d = fmaf(a[c], b[c], d);
                    acc += w * d;
                }
            acc = wave_sum(acc);
            if (lane == 0) corr[((((size_t)n * S + s_) * rd * rd + (size_t)i * rd + j) * H + y) * W + x] = acc;
        }
}
// gradients w.r.t. fmap1 and fmap2 (atomic adds into zeroed buffers; coords get no gradient, as in DROID-SLAM)
__global__ __launch_bounds__(64) void altcorr_bwd_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                         const float* __restrict__ coords, const float* __restrict__ gcorr, int BN, int S,
                                                         int H, int W, int H2, int W2, int C, int r, float* __restrict__ g1,
                                                         float* __restrict__ g2) {
    const int rd = 2 * r + 1;
    const size_t pix = blockIdx.x;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), s_ = (int)((pix / ((size_t)W * H)) % S), n = (int)(pix / ((size_t)W * H * S));
    const int lane = threadIdx.x;
    const float cx = coords[pix * 2], cy = coords[pix * 2 + 1];
    const float fx = floorf(cx), fy = floorf(cy);
    const float dx = cx - fx, dy = cy - fy;
    const size_t o1 = (((size_t)n * H + y) * W + x) * C;
    for (int i = 0; i < rd; i++)
        for (int j = 0; j < rd; j++) {
            const float g = gcorr[((((size_t)n * S + s_) * rd * rd + (size_t)i * rd + j) * H + y) * W + x];
            if (g == 0.f) continue;
            const int xa = (int)fx - r + i, ya = (int)fy - r + j;
#pragma unroll
            for (int oy = 0; oy < 2; oy++)
#pragma unroll
                for (int ox = 0; ox < 2; ox++) {
                    const int xx = xa + ox, yy = ya + oy;
                    if (xx < 0 || xx >= W2 || yy < 0 || yy >= H2) continue;
                    const float w = g * (ox ? dx : 1.f - dx) * (oy ? dy : 1.f - dy);
                    const size_t o2 = (((size_t)n * H2 + yy) * W2 + xx) * C;
                    for (int c = lane; c < C; c += 64) {
                        atomicAdd(&g1[o1 + c], w * f2[o2 + c]);
                        atomicAdd(&g2[o2 + c], w * f1[o1 + c]);
                    }
                }
        }
}

inline int grid_for(size_t total, int block = 256) {
    size_t gsz = (total + block - 1) / block;
    if (gsz > 8192) gsz = 8192;
    if (gsz < 1) gsz = 1;
    return (int)gsz;
}

}  // namespace

extern "C" int cut3r_corr_index_forward(const float* volume, const float* coords, float* out, int BN, int h1, int w1, int h2, int w2,
                                        int radius, void* stream) {
    if (!volume || !coords || !out || BN <= 0 || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0 || radius < 0) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(corr_index_fwd_kernel, dim3(grid_for((size_t)BN * h1 * w1)), dim3(256), 0, (hipStream_t)stream, volume, coords, out,
                       BN, h1, w1, h2, w2, radius);
    return cut3r_check_launch();
}

extern "C" int cut3r_corr_index_backward(const float* coords, const float* grad_out, float* grad_volume, int BN, int h1, int w1, int h2,
                                         int w2, int radius, void* stream) {
    if (!coords || !grad_out || !grad_volume || BN <= 0 || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0 || radius < 0) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(grad_volume, 0, sizeof(float) * (size_t)BN * h1 * w1 * h2 * w2, s) != hipSuccess) return CUT3R_ERR_LAUNCH;
    hipLaunchKernelGGL(corr_index_bwd_kernel, dim3(grid_for((size_t)BN * h1 * w1)), dim3(256), 0, s, coords, grad_out, grad_volume, BN,
                       h1, w1, h2, w2, radius);
    return cut3r_check_launch();
}

extern "C" long long cut3r_ba_workspace_floats(int P, int ht, int wd, int N, int M, int fixedp) {
    const long long HW = (long long)ht * wd, Pf = P - fixedp, nblk = (HW + BA_BLOCK - 1) / BA_BLOCK, n = Pf * 6;
    //      Hpart                      E                 C, w, dz     H            v    S      vS  L
    return (long long)N * nblk * BA_HROW + Pf * M * 6 * HW + 3 * M * HW + Pf * Pf * 36 + Pf * 6 + n * n + n + n * n + 64;
}

// workspace layout shared by the staged entry points
struct BaWs { float *Hpart, *E, *Cm, *wm, *H, *v, *S, *vS, *L; int nblk, n; };
static BaWs ba_ws(float* workspace, int P, int ht, int wd, int N, int M, int fixedp) {
    const long long HW = (long long)ht * wd;
    const int Pf = P - fixedp;
    BaWs w;
    w.nblk = (int)((HW + BA_BLOCK - 1) / BA_BLOCK);
    w.n = Pf * 6;
    w.Hpart = workspace;
    w.E = w.Hpart + (size_t)N * w.nblk * BA_HROW;
    w.Cm = w.E + (size_t)Pf * M * 6 * HW;
    w.wm = w.Cm + (size_t)M * HW;
    w.H = w.wm + (size_t)M * HW + (size_t)M * HW;     // (third M*HW slot reserved)
    w.v = w.H + (size_t)Pf * Pf * 36;
    w.S = w.v + (size_t)Pf * 6;
    w.vS = w.S + (size_t)w.n * w.n;
    w.L = w.vS + w.n;
    return w;
}

extern "C" int cut3r_ba_assemble(const float* Gij, const float* disps, const float* intr, const float* target, const float* weight,
                                 const float* eta, const int* ii, const int* jj, const int* src_ptr, const int* src_edges, const int* kx,
                                 const unsigned char* present, int P, int ht, int wd, int N, int M, int fixedp, int motion_only,
                                 float* workspace, float* S_out, float* vS_out, float* hdiag_out, void* stream) {
    if (!Gij || !disps || !intr || !target || !weight || !ii || !jj || !src_ptr || !src_edges || !kx || !present || !workspace || !S_out ||
        !vS_out || !hdiag_out)
        return CUT3R_ERR_ARG;
    const int Pf = P - fixedp;
    if (P <= 0 || Pf <= 0 || ht <= 0 || wd <= 0 || N <= 0 || M <= 0 || fixedp < 0 || Pf * 6 > CHOL_MAXN) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long long HW = (long long)ht * wd;
    const BaWs w = ba_ws(workspace, P, ht, wd, N, M, fixedp);
    if (hipMemsetAsync(w.E, 0, sizeof(float) * (size_t)Pf * M * 6 * HW, s) != hipSuccess) return CUT3R_ERR_LAUNCH;
    BaGeom g{P, ht, wd, N, M, fixedp};
    hipLaunchKernelGGL(ba_edge_kernel, dim3(w.nblk, M), dim3(BA_BLOCK), 0, s, Gij, disps, intr, target, weight, ii, jj, src_ptr, src_edges,
                       kx, g, w.Hpart, w.E, w.Cm, w.wm, eta);
    hipLaunchKernelGGL(ba_fold_edges_kernel, dim3(N), dim3(128), 0, s, w.Hpart, w.nblk);
    hipLaunchKernelGGL(ba_reduce_H_kernel, dim3(Pf, Pf + 1), dim3(64), 0, s, w.Hpart, w.nblk, ii, jj, N, Pf, fixedp, w.H, w.v);
    // undamped reduced system of THESE edges (motion only: no Schur correction, i.e. M = 0 for the reduction)
    hipLaunchKernelGGL(ba_schur_kernel, dim3(Pf, Pf), dim3(256), 0, s, w.H, w.v, w.E, w.Cm, w.wm, present, Pf, motion_only ? 0 : M, (int)HW,
                       0.f, 0.f, S_out, vS_out);
    hipLaunchKernelGGL(ba_hdiag_kernel, dim3((w.n + 63) / 64), dim3(64), 0, s, w.H, Pf, hdiag_out);
    return cut3r_check_launch();
}

extern "C" int cut3r_ba_solve(const float* S, const float* vS, const float* hdiag, int n, float ep, float lm, float* scratch, float* dx,
                              int* flag, void* stream) {
    if (!S || !vS || !hdiag || !scratch || !dx || !flag || n <= 0 || n > CHOL_MAXN) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ba_damp_kernel, dim3((n * n + 255) / 256), dim3(256), 0, s, S, hdiag, n, ep, lm, scratch);
    const size_t chol_lds = sizeof(float) * ((size_t)n * (n + 1) + 2 * (n + 8));
    if (hipFuncSetAttribute((const void*)ba_chol_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chol_lds) != hipSuccess)
        return CUT3R_ERR_LAUNCH;
    hipLaunchKernelGGL(ba_chol_solve_kernel, dim3(1), dim3(256), chol_lds, s, scratch, vS, n, dx, (float*)nullptr, flag);
    return cut3r_check_launch();
}

extern "C" int cut3r_ba_backsub(float* workspace, const float* dx, const unsigned char* present, int P, int ht, int wd, int N, int M,
                                int fixedp, float* dz, void* stream) {
    if (!workspace || !dx || !present || !dz || P - fixedp <= 0 || M <= 0) return CUT3R_ERR_ARG;
    const BaWs w = ba_ws(workspace, P, ht, wd, N, M, fixedp);
    hipLaunchKernelGGL(ba_dz_kernel, dim3(w.nblk, M), dim3(256), 0, (hipStream_t)stream, w.E, w.Cm, w.wm, dx, present, P - fixedp, M, ht * wd, dz);
    return cut3r_check_launch();
}

extern "C" int cut3r_ba_step(const float* Gij, const float* disps, const float* intr, const float* target, const float* weight,
                             const float* eta, const int* ii, const int* jj, const int* src_ptr, const int* src_edges, const int* kx,
                             const unsigned char* present, int P, int ht, int wd, int N, int M, int fixedp, float ep, float lm,
                             float* workspace, float* dx, float* dz, int* flag, void* stream) {
    if (!workspace || !eta) return CUT3R_ERR_ARG;
    const BaWs w = ba_ws(workspace, P, ht, wd, N, M, fixedp);
    int rc = cut3r_ba_assemble(Gij, disps, intr, target, weight, eta, ii, jj, src_ptr, src_edges, kx, present, P, ht, wd, N, M, fixedp, 0,
                               workspace, w.S, w.vS, w.L, stream);                  // (hdiag in the first n floats of the L region)
    if (rc != CUT3R_OK) return rc;
    rc = cut3r_ba_solve(w.S, w.vS, w.L, w.n, ep, lm, w.L + w.n, dx, flag, stream);
    if (rc != CUT3R_OK) return rc;
    return cut3r_ba_backsub(workspace, dx, present, P, ht, wd, N, M, fixedp, dz, stream);
}

/* per-source depth normal equations with the poses held fixed (droid_backends.proj_trans, geom/ba.py:200): C = sum_e w Jz^2,
 * w = sum_e w r Jz over the edges of each source frame (no eta) */
extern "C" int cut3r_ba_proj_trans(const float* Gij, const float* disps, const float* intr, const float* target, const float* weight,
                                   const int* ii, const int* jj, const int* src_ptr, const int* src_edges, const int* kx, int P, int ht,
                                   int wd, int N, int M, float* workspace, float* C_out, float* w_out, void* stream) {
    if (!Gij || !disps || !intr || !target || !weight || !ii || !jj || !src_ptr || !src_edges || !kx || !workspace || !C_out || !w_out)
        return CUT3R_ERR_ARG;
    if (P <= 0 || ht <= 0 || wd <= 0 || N <= 0 || M <= 0) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long long HW = (long long)ht * wd;
    const int fixedp = P;                                   // every pose fixed: no E blocks are written (ip, jp < 0)
    const int nblk = (int)((HW + BA_BLOCK - 1) / BA_BLOCK);
    BaGeom g{P, ht, wd, N, M, fixedp};
    hipLaunchKernelGGL(ba_edge_kernel, dim3(nblk, M), dim3(BA_BLOCK), 0, s, Gij, disps, intr, target, weight, ii, jj, src_ptr, src_edges, kx,
                       g, workspace /* Hpart: N*nblk*120 floats */, (float*)nullptr, C_out, w_out, (const float*)nullptr);
    return cut3r_check_launch();
}

extern "C" int cut3r_bi_inter(const float* scales, const float* grid, int M, int hs, int ws, int ht, int wd, float* vals, float* J,
                              void* stream) {
    if (!scales || !grid || !vals || !J || M <= 0 || hs <= 0 || ws <= 0 || ht <= 0 || wd <= 0) return CUT3R_ERR_ARG;
    const size_t tot = (size_t)M * ht * wd;
    hipLaunchKernelGGL(bi_inter_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scales, grid, M, hs, ws, ht * wd,
                       vals, J);
    return cut3r_check_launch();
}

extern "C" long long cut3r_schur_mono_prior_workspace_floats(int n, long long cols) {
    return (long long)n * n * 3 + 3LL * n + (long long)n * cols;
}

extern "C" int cut3r_schur_mono_prior(const float* C, const float* w, const float* H, const float* E, const float* v, int n, long long cols,
                                      float ep, float lm, float* workspace, float* dso, float* dz, float* dzcov, int* flag, void* stream) {
    if (!C || !w || !H || !E || !v || !workspace || !dso || !dz || !flag || n <= 0 || n > CHOL_MAXN || cols <= 0) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    float* S = workspace;
    float* Sd = S + (size_t)n * n;
    float* L = Sd + (size_t)n * n;
    float* vS = L + (size_t)n * n;
    float* hd = vS + n;
    float* xs = hd + 2 * n;
    const int nb = (n + 15) / 16;
    hipLaunchKernelGGL(mp_reduce_kernel, dim3(nb, nb), dim3(256), 0, s, H, E, C, w, v, n, cols, S, vS, hd);
    hipLaunchKernelGGL(ba_damp_kernel, dim3((n * n + 255) / 256), dim3(256), 0, s, S, hd, n, ep, lm, Sd);
    const size_t chol_lds = sizeof(float) * ((size_t)n * (n + 1) + 2 * (n + 8));
    if (hipFuncSetAttribute((const void*)ba_chol_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chol_lds) != hipSuccess)
        return CUT3R_ERR_LAUNCH;
    hipLaunchKernelGGL(ba_chol_solve_kernel, dim3(1), dim3(256), chol_lds, s, Sd, vS, n, dso, L, flag);
    const size_t bs_lds = sizeof(float) * ((dzcov ? (size_t)n * (n + 1) / 2 : 0) + n);
    if (hipFuncSetAttribute((const void*)mp_backsub_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bs_lds) != hipSuccess)
        return CUT3R_ERR_LAUNCH;
    hipLaunchKernelGGL(mp_backsub_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), bs_lds, s, E, C, w, dso, L, flag, n, cols, dz, dzcov, xs);
    return cut3r_check_launch();
}

extern "C" int cut3r_jdsa_blocks(const float* prior, const float* Jbi, const float* rd, float alpha, int M, int HW, int D, float* H, float* E,
                                 float* v, void* stream) {
    if (!prior || !Jbi || !rd || !H || !E || !v || M <= 0 || HW <= 0 || D <= 0 || M * D > CHOL_MAXN) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(jdsa_blocks_kernel, dim3(M), dim3(256), 256 * sizeof(float), (hipStream_t)stream, prior, Jbi, rd, alpha, M, HW, D, H, E, v);
    return cut3r_check_launch();
}

extern "C" int cut3r_depth_filter(const float* poses, const float* disps, const float* intr, const long long* inds, const float* thresh, int n,
                                  int M, int ht, int wd, float* count, void* stream) {
    if (!poses || !disps || !intr || !inds || !thresh || !count || n <= 0 || M <= 0 || ht <= 1 || wd <= 1) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(depth_filter_kernel, dim3((unsigned)((ht * wd + 255) / 256), (unsigned)M), dim3(256), 0, (hipStream_t)stream, poses, disps,
                       intr, inds, thresh, n, ht, wd, count);
    return cut3r_check_launch();
}

extern "C" int cut3r_altcorr_forward(const float* fmap1, const float* fmap2, const float* coords, int BN, int S, int H, int W, int H2, int W2,
                                     int C, int radius, float* corr, void* stream) {
    if (!fmap1 || !fmap2 || !coords || !corr || BN <= 0 || S <= 0 || H <= 0 || W <= 0 || H2 <= 0 || W2 <= 0 || C <= 0 || radius < 0)
        return CUT3R_ERR_ARG;
    const size_t tot = (size_t)BN * S * H * W;
    if (tot > 0x7fffffffULL) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(altcorr_fwd_kernel, dim3((unsigned)tot), dim3(64), 0, (hipStream_t)stream, fmap1, fmap2, coords, BN, S, H, W, H2, W2, C,
                       radius, corr);
    return cut3r_check_launch();
}

extern "C" int cut3r_altcorr_backward(const float* fmap1, const float* fmap2, const float* coords, const float* grad_corr, int BN, int S,
                                      int H, int W, int H2, int W2, int C, int radius, float* grad1, float* grad2, void* stream) {
    if (!fmap1 || !fmap2 || !coords || !grad_corr || !grad1 || !grad2 || BN <= 0 || S <= 0 || H <= 0 || W <= 0 || C <= 0 || radius < 0)
        return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t tot = (size_t)BN * S * H * W;
    if (tot > 0x7fffffffULL) return CUT3R_ERR_ARG;
    if (hipMemsetAsync(grad1, 0, sizeof(float) * (size_t)BN * H * W * C, s) != hipSuccess) return CUT3R_ERR_LAUNCH;
    if (hipMemsetAsync(grad2, 0, sizeof(float) * (size_t)BN * H2 * W2 * C, s) != hipSuccess) return CUT3R_ERR_LAUNCH;
    hipLaunchKernelGGL(altcorr_bwd_kernel, dim3((unsigned)tot), dim3(64), 0, s, fmap1, fmap2, coords, grad_corr, BN, S, H, W, H2, W2, C, radius,
                       grad1, grad2);
    return cut3r_check_launch();
}
