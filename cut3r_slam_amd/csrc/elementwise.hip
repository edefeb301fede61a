// Memory-bound kernels of the ViT path on gfx950: RoPE-2D, LayerNorm(+adaLN), patch im2col, casts, column mean,
// bilinear x2 upsample, DPT output activations.  All are HBM-bound: coalesced 8/16-byte accesses, one pass.
#include "common.h"
#include "../../include/cut3r_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------- RoPE-2D
// /root/reference/src/croco/models/curope/kernels.cu:17-82.  One workgroup per (b, n) token, as in the reference kernel:
// cos/sin of the token's (y, x) position are evaluated once per (X, q) in fp32 with the reference's operation order
// (inv_freq = fwd / powf(base, q/Q); freq = pos * inv_freq) into LDS and shared by every head -- and by BOTH tensors
// when a second one (k next to q, same positions) is passed.  Each thread rotates 4 consecutive (u, v) pairs with
// 8/16-byte accesses.
template <typename T> struct alignas(4 * sizeof(T)) Vec4 { T v[4]; };

template <typename T>
__global__ __launch_bounds__(256) void rope2d_kernel(T* __restrict__ tok, T* __restrict__ tok2, const int64_t* __restrict__ pos,
                                                     int N, int H, int D, long long sB, long long sN, long long sH, long long sB2,
                                                     long long sN2, float base, float fwd) {
    __shared__ float cs[2][2][64];          // [X][cos|sin][q]   (Q <= 64, i.e. D <= 256)
    const int t = blockIdx.x;
    const int b = t / N, n = t - b * N;
    const int Q = D >> 2;
    if (threadIdx.x < 2 * Q) {
        const int X = threadIdx.x / Q, q = threadIdx.x - X * Q;
        const float inv = fwd / powf(base, (float)q / (float)Q);
        const float fr = (float)pos[(size_t)t * 2 + X] * inv;
        cs[X][0][q] = cosf(fr);
        cs[X][1][q] = sinf(fr);
    }
    __syncthreads();
    const int Q4 = Q >> 2;                  // 4-pair chunks per (head, X)
    const int per_tensor = H * 2 * Q4;
    const int total = tok2 ? 2 * per_tensor : per_tensor;
    for (int w = threadIdx.x; w < total; w += blockDim.x) {
        const int which = w / per_tensor;
        int r = w - which * per_tensor;
        const int h = r / (2 * Q4);
        r -= h * 2 * Q4;
        const int X = r / Q4, c = r - X * Q4;
        T* base_p = which ? (tok2 + (size_t)b * sB2 + (size_t)n * sN2) : (tok + (size_t)b * sB + (size_t)n * sN);
        T* pu = base_p + (size_t)h * sH + X * 2 * Q + 4 * c;
        T* pv = pu + Q;
        Vec4<T> u = *reinterpret_cast<Vec4<T>*>(pu);
        Vec4<T> v = *reinterpret_cast<Vec4<T>*>(pv);
        Vec4<T> ou, ov;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float co = cs[X][0][4 * c + e], si = cs[X][1][4 * c + e];
            const float uf = (float)u.v[e], vf = (float)v.v[e];
            float t1 = vf * si, t2 = uf * si;               // the GEMM-fused form (gemm.hip rope_store4) repeats this exactly
            asm volatile("" : "+v"(t1), "+v"(t2));
            ou.v[e] = (T)rope_rot(uf, co, -t1);
            ov.v[e] = (T)rope_rot(vf, co, t2);
        }
        *reinterpret_cast<Vec4<T>*>(pu) = ou;
        *reinterpret_cast<Vec4<T>*>(pv) = ov;
    }
}

// Table-driven form for the network's own launches: cos/sin come from the table of rope2d_table_kernel (same expressions, so the
// same bits), there is no per-token workgroup, no transcendental and no barrier: one thread rotates VE (u, v) pairs with
// 8/16-byte accesses, a launch covers TWO token ranges (q and k of a self-attention; q of one token stream and k of the other in
// a cross-attention).  Positions outside the table are evaluated in place by the expressions of rope2d_kernel.
struct RopeSeg {
    h16* p;
    const int64_t* pos;
    long long sTok;         // token stride in elements (head stride = D)
};

template <int VE> struct alignas(2 * VE) HVec { h16 v[VE]; };

template <int VE>
__global__ __launch_bounds__(256) void rope2d_tab_kernel(RopeSeg a, RopeSeg b, long long work_a, long long work_total, int H, int D,
                                                         const float* __restrict__ table, int pmin, int npos, float base, float fwd) {
    long long w = (long long)blockIdx.x * 256 + threadIdx.x;
    if (w >= work_total) return;
    RopeSeg s = a;
    if (w >= work_a) { w -= work_a; s = b; }
    const int Q = D >> 2, QV = Q / VE, per_tok = H * 2 * QV;
    const long long t = w / per_tok;
    int r = (int)(w - t * per_tok);
    const int h = r / (2 * QV);
    r -= h * 2 * QV;
    const int X = r / QV, c = r - X * QV;
    h16* pu = s.p + t * s.sTok + (long long)h * D + X * 2 * Q + VE * c;
    h16* pv = pu + Q;
    const HVec<VE> u = *reinterpret_cast<const HVec<VE>*>(pu);
    const HVec<VE> v = *reinterpret_cast<const HVec<VE>*>(pv);
    const long long pp = s.pos[t * 2 + X];
    const long long pi = pp - pmin;
    float co[VE], si[VE];
    if (pi >= 0 && pi < npos) {
        const float* tc = table + pi * Q + VE * c;
        const float* ts = tc + (long long)npos * Q;
#pragma unroll
        for (int e = 0; e < VE; e += 4) {
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(tc + e), s4 = *reinterpret_cast<const f32x4*>(ts + e);
#pragma unroll
            for (int j = 0; j < 4; j++) { co[e + j] = c4[j]; si[e + j] = s4[j]; }
        }
    } else {
#pragma unroll
        for (int e = 0; e < VE; e++) {
            const float inv = fwd / powf(base, (float)(VE * c + e) / (float)Q);
            const float fr = (float)pp * inv;
            co[e] = cosf(fr);
            si[e] = sinf(fr);
        }
    }
    HVec<VE> ou, ov;
#pragma unroll
    for (int e = 0; e < VE; e++) {
        const float uf = (float)u.v[e], vf = (float)v.v[e];
        float t1 = vf * si[e], t2 = uf * si[e];
        asm volatile("" : "+v"(t1), "+v"(t2));
        ou.v[e] = (h16)rope_rot(uf, co[e], -t1);
        ov.v[e] = (h16)rope_rot(vf, co[e], t2);
    }
    *reinterpret_cast<HVec<VE>*>(pu) = ou;
    *reinterpret_cast<HVec<VE>*>(pv) = ov;
}

// cos / sin of every (position, frequency) pair for a quarter head dimension Q, by the expressions of rope2d_kernel
__global__ void rope2d_table_kernel(float* __restrict__ table, int pmin, int npos, int Q, float base, float fwd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npos * Q) return;
    const int p = i / Q, q = i - p * Q;
    const float inv = fwd / powf(base, (float)q / (float)Q);
    const float fr = (float)(long long)(pmin + p) * inv;
    table[i] = cosf(fr);
    table[npos * Q + i] = sinf(fr);
}

// ------------------------------------------------------------------------------------------------- LayerNorm
// One wave per row; the row lives in registers (C <= 64*VPL*... handled by a strided loop), two-pass mean/variance
// exactly as torch (biased variance of (x-mean)).
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, int M, int C,
                                                        h16* __restrict__ y16, int ld16, float* __restrict__ y32, int ld32,
                                                        const float* __restrict__ mscale, const float* __restrict__ mshift) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx;
    constexpr int MAXV = 8;                 // supports C <= 64*4*8 = 2048
    f32x4 v[MAXV];
    float s = 0.f;
    const int nv = C >> 2;                  // float4 count
#pragma unroll
    for (int i = 0; i < MAXV; i++) {
        int c4 = lane + i * 64;
        if (c4 < nv) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + c4 * 4);
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; i++) {
        int c4 = lane + i * 64;
        if (c4 < nv) {
#pragma unroll
            for (int e = 0; e < 4; e++) { float d = v[i][e] - mean; q += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXV; i++) {
        int c4 = lane + i * 64;
        if (c4 < nv) {
            f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c4 * 4);
            f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c4 * 4);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = (v[i][e] - mean) * rstd * g[e] + bt[e];
            if (mscale) {
                f32x4 sc = *reinterpret_cast<const f32x4*>(mscale + c4 * 4);
                f32x4 sh = *reinterpret_cast<const f32x4*>(mshift + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = o[e] * (1.0f + sc[e]) + sh[e];
            }
            if (y32) *reinterpret_cast<f32x4*>(y32 + (size_t)row * ld32 + c4 * 4) = o;
            if (y16) {
                half4_t ho = {(h16)o[0], (h16)o[1], (h16)o[2], (h16)o[3]};
                *reinterpret_cast<half4_t*>(y16 + (size_t)row * ld16 + c4 * 4) = ho;
            }
        }
    }
}

// Specialised LayerNorm for C == 256*NV (768 / 1024 / 1536: every width of the production network): compile-time trip
// counts, the row AND gamma/beta are fetched up front (one memory round trip instead of two), 16-byte accesses.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_fast_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, int M, h16* __restrict__ y16,
                                                             int ld16, float* __restrict__ y32, int ld32,
                                                             const float* __restrict__ mscale, const float* __restrict__ mshift) {
    constexpr int C = 256 * NV;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx;
    f32x4 v[NV], g[NV], bt[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = *reinterpret_cast<const f32x4*>(xr + (lane + i * 64) * 4);
#pragma unroll
    for (int i = 0; i < NV; i++) {
        g[i] = *reinterpret_cast<const f32x4*>(gamma + (lane + i * 64) * 4);
        bt[i] = *reinterpret_cast<const f32x4*>(beta + (lane + i * 64) * 4);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++)
#pragma unroll
        for (int e = 0; e < 4; e++) { const float d = v[i][e] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int c = (lane + i * 64) * 4;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = (v[i][e] - mean) * rstd * g[i][e] + bt[i][e];
        if (mscale) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(mscale + c);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(mshift + c);
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = o[e] * (1.0f + sc[e]) + sh[e];
        }
        if (y32) *reinterpret_cast<f32x4*>(y32 + (size_t)row * ld32 + c) = o;
        if (y16) {
            const half4_t ho = {(h16)o[0], (h16)o[1], (h16)o[2], (h16)o[3]};
            *reinterpret_cast<half4_t*>(y16 + (size_t)row * ld16 + c) = ho;
        }
    }
}

// LayerNorm of ONE tensor with TWO affine parameter sets -> two fp16 outputs (the row statistics are shared).  In a
// decoder layer the image tokens are normalised by the image block's norm1 and by the state block's norm_y (and vice
// versa for the state tokens): one read of the fp32 row instead of two, one launch instead of two.  Same arithmetic per
// output as layernorm_fast_kernel.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_dual_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g1,
                                                             const float* __restrict__ b1, h16* __restrict__ y1, int ld1,
                                                             const float* __restrict__ g2, const float* __restrict__ b2,
                                                             h16* __restrict__ y2, int ld2, float eps, int M) {
    constexpr int C = 256 * NV;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx;
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = *reinterpret_cast<const f32x4*>(xr + (lane + i * 64) * 4);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++)
#pragma unroll
        for (int e = 0; e < 4; e++) { const float d = v[i][e] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int c = (lane + i * 64) * 4;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(g1 + c), ba = *reinterpret_cast<const f32x4*>(b1 + c);
        const f32x4 gb = *reinterpret_cast<const f32x4*>(g2 + c), bb = *reinterpret_cast<const f32x4*>(b2 + c);
        half4_t o1, o2;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float n = (v[i][e] - mean) * rstd;
            o1[e] = (h16)(n * ga[e] + ba[e]);
            o2[e] = (h16)(n * gb[e] + bb[e]);
        }
        *reinterpret_cast<half4_t*>(y1 + (size_t)row * ld1 + c) = o1;
        *reinterpret_cast<half4_t*>(y2 + (size_t)row * ld2 + c) = o2;
    }
}

// ------------------------------------------------------------------------------------------------- im2col (patch embed)
// out[(b, py, px), (c, iy, ix)] = img[b, c, py*P+iy, px*P+ix]; one thread per 8 contiguous ix.
template <bool U8>
__global__ __launch_bounds__(256) void im2col_kernel(const void* __restrict__ img, int B, int C, int H, int W, int P,
                                                     h16* __restrict__ out) {
    const int nh = H / P, nw = W / P;
    const int Kp = C * P * P;
    const size_t total = (size_t)B * nh * nw * (Kp / 8);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int kc = (int)(i % (Kp / 8));
        const size_t m = i / (Kp / 8);
        const int k = kc * 8;
        const int c = k / (P * P), rem = k - c * P * P;
        const int iy = rem / P, ix = rem - iy * P;
        const int px = (int)(m % nw);
        const int py = (int)((m / nw) % nh);
        const int b = (int)(m / ((size_t)nw * nh));
        const size_t src = (((size_t)b * C + c) * H + (py * P + iy)) * W + (px * P + ix);
        half8_t o;
        if (U8) {
            const unsigned char* p = (const unsigned char*)img + src;
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = (h16)(((float)p[e] / 255.0f - 0.5f) / 0.5f);
        } else {
            const float* p = (const float*)img + src;
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = (h16)p[e];
        }
        *reinterpret_cast<half8_t*>(out + m * Kp + k) = o;
    }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ x, int ldx, h16* __restrict__ y, int ldy, int M, int C4) {
    const size_t total = (size_t)M * C4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const size_t m = i / C4;
        f32x4 v = *reinterpret_cast<const f32x4*>(x + m * ldx + c * 4);
        half4_t o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
        *reinterpret_cast<half4_t*>(y + m * ldy + c * 4) = o;
    }
}

// column mean: each block owns 64 columns; its 4 waves split the rows (stride 4) with 4 independent accumulators per
// thread for memory-level parallelism; partials are combined through LDS in fixed order (deterministic).
__global__ __launch_bounds__(256) void colmean_kernel(const float* __restrict__ x, int ldx, int M, int C, float* __restrict__ y,
                                                      long long sx, long long sy) {
    __shared__ float part[16][64];
    x += (size_t)blockIdx.y * sx;          // blockIdx.y: independent matrices (one per tracking window)
    y += (size_t)blockIdx.y * sy;
    const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;      // 4 row groups per block
    const int c = blockIdx.x * 64 + col;
    // 4 independent accumulators per thread (ILP), rows strided by 16
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
        int m = rg;
        for (; m + 12 < M; m += 16) {
            s0 += x[(size_t)m * ldx + c];
            s1 += x[(size_t)(m + 4) * ldx + c];
            s2 += x[(size_t)(m + 8) * ldx + c];
            s3 += x[(size_t)(m + 12) * ldx + c];
        }
        for (; m < M; m += 4) s0 += x[(size_t)m * ldx + c];
    }
    part[rg][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && c < C) y[c] = (part[0][col] + part[1][col] + part[2][col] + part[3][col]) / (float)M;
}

// ------------------------------------------------------------------------------------------------- bilinear x2 (align_corners)
// /root/reference/src/croco/models/dpt_block.py:215-221: src = dst * (in-1)/(out-1).  NHWC fp16, 8 channels per thread.
__global__ __launch_bounds__(256) void upsample2x_kernel(const h16* __restrict__ in, h16* __restrict__ out, int B, int H, int W, int C8) {
    const int Ho = 2 * H, Wo = 2 * W;
    const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const size_t total = (size_t)B * Ho * Wo * C8;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8);
        size_t p = i / C8;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        const float sy = ry * (float)oy, sx = rx * (float)ox;
        int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const h16* base = in + (size_t)b * H * W * C8 * 8 + c * 8;
        const half8_t v00 = *reinterpret_cast<const half8_t*>(base + ((size_t)y0 * W + x0) * C8 * 8);
        const half8_t v01 = *reinterpret_cast<const half8_t*>(base + ((size_t)y0 * W + x1) * C8 * 8);
        const half8_t v10 = *reinterpret_cast<const half8_t*>(base + ((size_t)y1 * W + x0) * C8 * 8);
        const half8_t v11 = *reinterpret_cast<const half8_t*>(base + ((size_t)y1 * W + x1) * C8 * 8);
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; e++)
            o[e] = (h16)(hy * (hx * (float)v00[e] + lx * (float)v01[e]) + ly * (hx * (float)v10[e] + lx * (float)v11[e]));
        *reinterpret_cast<half8_t*>(out + i * 8) = o;
    }
}

// ------------------------------------------------------------------------------------------------- output activations
DEVINL void act_pts_conf(float x, float y, float z, float c, bool has_conf, bool pos_z, float* pts, float* conf, size_t p) {
    if (pos_z) {
        float sg = (z > 0.f) ? 1.f : ((z < 0.f) ? -1.f : 0.f);
        x *= sg; y *= sg; z *= sg;
    }
    const float d = sqrtf(x * x + y * y + z * z);
    const float dc = fmaxf(d, 1e-8f);
    const float e = expm1f(d);
    pts[3 * p + 0] = x / dc * e;
    pts[3 * p + 1] = y / dc * e;
    pts[3 * p + 2] = z / dc * e;
    if (has_conf) conf[p] = 1.0f + expf(c);
}

// final 1x1 conv (Cin -> nout<=4) + activations; one wave per pixel group: each lane handles one pixel, reading its
// Cin fp16 channels as 16-B vectors (rows are Cin*2 bytes apart -> each lane streams its own 256-B row).
__global__ __launch_bounds__(256) void dpt_final_kernel(const h16* __restrict__ in, int P, int Cin, const float* __restrict__ w,
                                                        const float* __restrict__ bsv, int mode, float* __restrict__ pts,
                                                        float* __restrict__ conf) {
    extern __shared__ float sw[];          // [nout][Cin]
    const int nout = mode == 0 ? 4 : 3;
    for (int i = threadIdx.x; i < nout * Cin; i += blockDim.x) sw[i] = w[i];
    __syncthreads();
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < (size_t)P; p += (size_t)gridDim.x * blockDim.x) {
        float a0 = bsv[0], a1 = bsv[1], a2 = bsv[2], a3 = nout == 4 ? bsv[3] : 0.f;
        const h16* row = in + p * Cin;
        for (int k = 0; k < Cin; k += 8) {
            half8_t v = *reinterpret_cast<const half8_t*>(row + k);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float xv = (float)v[e];
                a0 = fmaf(xv, sw[k + e], a0);
                a1 = fmaf(xv, sw[Cin + k + e], a1);
                a2 = fmaf(xv, sw[2 * Cin + k + e], a2);
                if (nout == 4) a3 = fmaf(xv, sw[3 * Cin + k + e], a3);
            }
        }
        if (mode == 0) {
            act_pts_conf(a0, a1, a2, a3, true, false, pts, conf, p);
        } else {
            const float eps = 1e-6f;
            float o[3] = {a0, a1, a2};
#pragma unroll
            for (int e = 0; e < 3; e++) {
                float sg = 1.0f / (1.0f + expf(-o[e]));
                pts[3 * p + e] = ((sg * (1 - 2 * eps) + eps) - 0.5f) * 2.0f;
            }
        }
    }
}

// final 1x1 conv (Cin -> nout<=4) + activations, coalesced form: a wave reads 1 KiB of consecutive channels per
// instruction (LPP = Cin/8 lanes per pixel, 64/LPP pixels per wave-instruction), every lane keeps the 4 x 8 weights of
// its channel chunk in registers, the LPP partial dot products are combined by xor-shuffles.  (The earlier form gave each
// lane its own 256-B pixel row: 64 cache lines per load instruction, 1.1 TB/s.)
__global__ __launch_bounds__(256) void dpt_final_coalesced_kernel(const h16* __restrict__ in, int P, int Cin, const float* __restrict__ w,
                                                                  const float* __restrict__ bsv, int mode, float* __restrict__ pts,
                                                                  float* __restrict__ conf) {
    const int nout = mode == 0 ? 4 : 3;
    const int lane = threadIdx.x & 63;
    const int lpp = Cin >> 3;                      // lanes per pixel (divides 64)
    const int ppw = 64 / lpp;                      // pixels per wave-instruction
    const int sub = lane / lpp, chunk = lane - sub * lpp;
    float wr[4][8];
#pragma unroll
    for (int o = 0; o < 4; o++)
#pragma unroll
        for (int e = 0; e < 8; e++) wr[o][e] = o < nout ? w[o * Cin + chunk * 8 + e] : 0.f;
    const float b0 = bsv[0], b1 = bsv[1], b2 = bsv[2], b3 = nout == 4 ? bsv[3] : 0.f;
    const size_t wave_global = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t p0 = wave_global * ppw; p0 < (size_t)P; p0 += nwaves * ppw) {
        const size_t p = p0 + sub;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (p < (size_t)P) {
            const half8_t v = *reinterpret_cast<const half8_t*>(in + p * Cin + chunk * 8);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float xv = (float)v[e];
                a0 = fmaf(xv, wr[0][e], a0);
                a1 = fmaf(xv, wr[1][e], a1);
                a2 = fmaf(xv, wr[2][e], a2);
                a3 = fmaf(xv, wr[3][e], a3);
            }
        }
        for (int o = lpp >> 1; o > 0; o >>= 1) {
            a0 += __shfl_xor(a0, o, 64);
            a1 += __shfl_xor(a1, o, 64);
            a2 += __shfl_xor(a2, o, 64);
            a3 += __shfl_xor(a3, o, 64);
        }
        if (chunk == 0 && p < (size_t)P) {
            a0 += b0; a1 += b1; a2 += b2; a3 += b3;
            if (mode == 0) {
                act_pts_conf(a0, a1, a2, a3, true, false, pts, conf, p);
            } else {
                const float eps = 1e-6f;
                const float o3[3] = {a0, a1, a2};
#pragma unroll
                for (int e = 0; e < 3; e++) {
                    const float sg = 1.0f / (1.0f + expf(-o3[e]));
                    pts[3 * p + e] = ((sg * (1 - 2 * eps) + eps) - 0.5f) * 2.0f;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void postprocess_pts_kernel(const float* __restrict__ raw, int P, int nch, int pos_z,
                                                              float* __restrict__ pts, float* __restrict__ conf) {
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < (size_t)P; p += (size_t)gridDim.x * blockDim.x) {
        const float* r = raw + p * nch;
        act_pts_conf(r[0], r[1], r[2], nch > 3 ? r[3] : 0.f, nch > 3, pos_z != 0, pts, conf, p);
    }
}

__global__ void postprocess_pose_kernel(const float* __restrict__ raw, int B, float* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* r = raw + b * 7;
    const float d = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    const float sc = expm1f(d) / fmaxf(d, 1e-8f);
    float qn = sqrtf(r[3] * r[3] + r[4] * r[4] + r[5] * r[5] + r[6] * r[6]);
    qn = fmaxf(qn, 1e-12f);                                 // F.normalize eps
    float q[4] = {r[3] / qn, r[4] / qn, r[5] / qn, r[6] / qn};
    const float sg = q[0] < 0.f ? -1.f : 1.f;
    out[b * 7 + 0] = r[0] * sc; out[b * 7 + 1] = r[1] * sc; out[b * 7 + 2] = r[2] * sc;
    for (int e = 0; e < 4; e++) out[b * 7 + 3 + e] = sg * q[e];
}

inline int grid_for(size_t total, int block = 256) {
    size_t g = (total + block - 1) / block;
    if (g > 2048 * 4) g = 2048 * 4;
    if (g < 1) g = 1;
    return (int)g;
}

// ---------------------------------------------------------------------------------------------------------------
// cv2.resize(img, (W1, H1)) with INTER_LINEAR on 8-bit interleaved images (demo_s.py:72,83): OpenCV's fixed-point
// bilinear (11-bit coefficients, horizontal pass in int, vertical pass with the >>4 / >>16 / +2 >>2 rounding) and its
// 2x-decimation special case (2x2 box with rounding).  One thread per output pixel, all channels; the output is written
// channel-planar [C,H1,W1] (what the tracker consumes: demo_s.py:73 permutes to CHW) or interleaved.
DEVINL short coef_q11(float v) { return (short)__float2int_rn(v * 2048.0f); }

__global__ __launch_bounds__(256) void resize_linear_u8_kernel(const unsigned char* __restrict__ src, int H0, int W0, int C,
                                                               unsigned char* __restrict__ dst, int H1, int W1, int chw) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W1) return;
    if (W0 == 2 * W1 && H0 == 2 * H1) {
        for (int c = 0; c < C; c++) {
            const unsigned char* p = src + ((size_t)(2 * y) * W0 + 2 * x) * C + c;
            const int v = (p[0] + p[C] + p[(size_t)W0 * C] + p[(size_t)W0 * C + C] + 2) >> 2;
            dst[chw ? ((size_t)c * H1 + y) * W1 + x : ((size_t)y * W1 + x) * C + c] = (unsigned char)v;
        }
        return;
    }
    const double sxs = (double)W0 / W1, sys = (double)H0 / H1;
    float fy = (float)((y + 0.5) * sys - 0.5);
    int sy = (int)floorf(fy);
    fy -= sy;
    const short b0 = coef_q11(1.0f - fy), b1 = coef_q11(fy);
    const int y0 = min(max(sy, 0), H0 - 1), y1 = min(max(sy + 1, 0), H0 - 1);
    float fx = (float)((x + 0.5) * sxs - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= W0 - 1) { fx = 0.f; sx = W0 - 1; }
    const short a0 = coef_q11(1.0f - fx), a1 = coef_q11(fx);
    const int x1 = min(sx + 1, W0 - 1);
    for (int c = 0; c < C; c++) {
        const int d0 = src[((size_t)y0 * W0 + sx) * C + c] * a0 + src[((size_t)y0 * W0 + x1) * C + c] * a1;
        const int d1 = src[((size_t)y1 * W0 + sx) * C + c] * a0 + src[((size_t)y1 * W0 + x1) * C + c] * a1;
        const int v = (((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2;
        dst[chw ? ((size_t)c * H1 + y) * W1 + x : ((size_t)y * W1 + x) * C + c] = (unsigned char)v;
    }
}

// cv2.remap(src, map, INTER_LINEAR, BORDER_CONSTANT 0) on uint8 HWC with OpenCV's fixed-point map: coordinates in 1/32 pixel
// (ix = round(32 u), integer part ix >> 5, fraction ix & 31), bilinear weights 32*a*b with a in {32 - fx, fx}, b in {32 - fy, fy}
// (= OpenCV's BilinearTab_i, whose entries are exact for 5-bit fractions), result (sum + 2^14) >> 15.  This is the second half of
// cv2.undistort (the map comes from initUndistortRectifyMap, built on the host once per sequence: cut3r_slam_amd/stream.py).
__global__ __launch_bounds__(256) void remap_linear_u8_kernel(const unsigned char* __restrict__ src, int H, int W, int C,
                                                              const int* __restrict__ map_ix, const int* __restrict__ map_iy,
                                                              unsigned char* __restrict__ dst, int Ho, int Wo) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= Wo) return;
    const size_t o = (size_t)y * Wo + x;
    const int ix = map_ix[o], iy = map_iy[o];
    const int sx = ix >> 5, sy = iy >> 5, fx = ix & 31, fy = iy & 31;
    const int w00 = 32 * (32 - fx) * (32 - fy), w01 = 32 * fx * (32 - fy), w10 = 32 * (32 - fx) * fy, w11 = 32 * fx * fy;
    const bool x0 = sx >= 0 && sx < W, x1 = sx + 1 >= 0 && sx + 1 < W, y0 = sy >= 0 && sy < H, y1 = sy + 1 >= 0 && sy + 1 < H;
    for (int c = 0; c < C; c++) {
        const int p00 = (x0 && y0) ? src[((size_t)sy * W + sx) * C + c] : 0;
        const int p01 = (x1 && y0) ? src[((size_t)sy * W + sx + 1) * C + c] : 0;
        const int p10 = (x0 && y1) ? src[((size_t)(sy + 1) * W + sx) * C + c] : 0;
        const int p11 = (x1 && y1) ? src[((size_t)(sy + 1) * W + sx + 1) * C + c] : 0;
        const int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
        dst[o * C + c] = (unsigned char)(v > 255 ? 255 : v);
    }
}

}  // namespace

extern "C" int cut3r_abi_version(void) { return 1; }

static int launch_rope(void* tokens, void* tokens2, int dtype, const int64_t* positions, int B, int N, int H, int D, long long sB,
                       long long sN, long long sH, long long sB2, long long sN2, float base, float fwd, void* stream) {
    if (!tokens || !positions || B <= 0 || N <= 0 || H <= 0 || D <= 0 || (D & 15) || D > 256) return CUT3R_ERR_ARG;
    if ((sB | sN | sH | sB2 | sN2) & 3) return CUT3R_ERR_ARG;        // 8/16-byte vector accesses
    const int work = (tokens2 ? 2 : 1) * H * 2 * (D / 16);
    int threads = ((work + 63) / 64) * 64;
    if (threads < 64) threads = 64;
    if (threads < D / 2) threads = ((D / 2 + 63) / 64) * 64;
    if (threads > 256) threads = 256;
    dim3 grid(B * N), block(threads);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL((rope2d_kernel<float>), grid, block, 0, s, (float*)tokens, (float*)tokens2, positions, N, H, D, sB, sN, sH, sB2,
                           sN2, base, fwd);
    else if (dtype == 1)
        hipLaunchKernelGGL((rope2d_kernel<h16>), grid, block, 0, s, (h16*)tokens, (h16*)tokens2, positions, N, H, D, sB, sN, sH, sB2, sN2,
                           base, fwd);
    else
        return CUT3R_ERR_ARG;
    return cut3r_check_launch();
}

extern "C" int cut3r_rope2d_table(float* table, int pmin, int npos, int Q, float base, float fwd, void* stream) {
    if (!table || npos < 1 || Q < 1 || Q > 64) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(rope2d_table_kernel, dim3((npos * Q + 255) / 256), dim3(256), 0, (hipStream_t)stream, table, pmin, npos, Q, base, fwd);
    return cut3r_check_launch();
}

extern "C" int cut3r_rope2d(void* tokens, int dtype, const int64_t* positions, int B, int N, int H, int D, long long sB,
                            long long sN, long long sH, float base, float fwd, void* stream) {
    return launch_rope(tokens, nullptr, dtype, positions, B, N, H, D, sB, sN, sH, 0, 0, base, fwd, stream);
}

extern "C" int cut3r_rope2d_tab(void* t0, const int64_t* pos0, long long ntok0, long long stride0, void* t1, const int64_t* pos1,
                                long long ntok1, long long stride1, int H, int D, const float* table, int pmin, int npos, float base,
                                float fwd, void* stream) {
    if (!t0 || !pos0 || !table || ntok0 <= 0 || H <= 0 || D <= 0 || (D & 15) || D > 256 || npos < 1) return CUT3R_ERR_ARG;
    if (t1 && (!pos1 || ntok1 <= 0)) return CUT3R_ERR_ARG;
    const int Q = D >> 2;
    const int VE = (Q % 8 == 0) ? 8 : 4;
    const long long am = VE - 1;                                    // 16-byte (VE = 8) or 8-byte vectors
    if ((stride0 & am) || (t1 && (stride1 & am)) || ((uintptr_t)t0 & (2 * VE - 1)) || (t1 && ((uintptr_t)t1 & (2 * VE - 1)))) return CUT3R_ERR_ARG;
    if (stride0 < (long long)H * D || (t1 && stride1 < (long long)H * D)) return CUT3R_ERR_ARG;
    const long long per_tok = (long long)H * 2 * (Q / VE);
    const long long wa = ntok0 * per_tok, wt = wa + (t1 ? ntok1 * per_tok : 0);
    RopeSeg a{(h16*)t0, pos0, stride0}, b{(h16*)(t1 ? t1 : t0), t1 ? pos1 : pos0, t1 ? stride1 : stride0};
    dim3 grid((unsigned)((wt + 255) / 256)), block(256);
    if (VE == 8)
        hipLaunchKernelGGL((rope2d_tab_kernel<8>), grid, block, 0, (hipStream_t)stream, a, b, wa, wt, H, D, table, pmin, npos, base, fwd);
    else
        hipLaunchKernelGGL((rope2d_tab_kernel<4>), grid, block, 0, (hipStream_t)stream, a, b, wa, wt, H, D, table, pmin, npos, base, fwd);
    return cut3r_check_launch();
}

extern "C" int cut3r_rope2d_qk(void* q, void* k, int dtype, const int64_t* positions, int B, int N, int H, int D, long long q_sB,
                               long long q_sN, long long k_sB, long long k_sN, float base, float fwd, void* stream) {
    if (!k) return CUT3R_ERR_ARG;
    return launch_rope(q, k, dtype, positions, B, N, H, D, q_sB, q_sN, D, k_sB, k_sN, base, fwd, stream);
}

extern "C" int cut3r_layernorm(const float* x, int ldx, const float* gamma, const float* beta, float eps, int M, int C, void* y16,
                               int ld16, float* y32, int ld32, const float* mod_scale, const float* mod_shift, void* stream) {
    if (!x || !gamma || !beta || M <= 0 || C <= 0 || (C & 3) || C > 2048 || (ldx & 3)) return CUT3R_ERR_ARG;
    if (!y16 && !y32) return CUT3R_ERR_ARG;
    if ((y16 && (ld16 & 3)) || (y32 && (ld32 & 3))) return CUT3R_ERR_ARG;
    if ((mod_scale == nullptr) != (mod_shift == nullptr)) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((M + 3) / 4), block(256);
    if (C == 768)
        hipLaunchKernelGGL((layernorm_fast_kernel<3>), grid, block, 0, s, x, ldx, gamma, beta, eps, M, (h16*)y16, ld16, y32, ld32, mod_scale, mod_shift);
    else if (C == 1024)
        hipLaunchKernelGGL((layernorm_fast_kernel<4>), grid, block, 0, s, x, ldx, gamma, beta, eps, M, (h16*)y16, ld16, y32, ld32, mod_scale, mod_shift);
    else if (C == 1536)
        hipLaunchKernelGGL((layernorm_fast_kernel<6>), grid, block, 0, s, x, ldx, gamma, beta, eps, M, (h16*)y16, ld16, y32, ld32, mod_scale, mod_shift);
    else
        hipLaunchKernelGGL(layernorm_kernel, grid, block, 0, s, x, ldx, gamma, beta, eps, M, C, (h16*)y16, ld16, y32, ld32, mod_scale, mod_shift);
    return cut3r_check_launch();
}

extern "C" int cut3r_layernorm_dual(const float* x, int ldx, const float* g1, const float* b1, void* y1, int ld1, const float* g2,
                                    const float* b2, void* y2, int ld2, float eps, int M, int C, void* stream) {
    if (!x || !g1 || !b1 || !y1 || !g2 || !b2 || !y2 || M <= 0 || (ldx & 3) || (ld1 & 3) || (ld2 & 3)) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((M + 3) / 4), block(256);
    if (C == 768) hipLaunchKernelGGL((layernorm_dual_kernel<3>), grid, block, 0, s, x, ldx, g1, b1, (h16*)y1, ld1, g2, b2, (h16*)y2, ld2, eps, M);
    else if (C == 1024) hipLaunchKernelGGL((layernorm_dual_kernel<4>), grid, block, 0, s, x, ldx, g1, b1, (h16*)y1, ld1, g2, b2, (h16*)y2, ld2, eps, M);
    else if (C == 1536) hipLaunchKernelGGL((layernorm_dual_kernel<6>), grid, block, 0, s, x, ldx, g1, b1, (h16*)y1, ld1, g2, b2, (h16*)y2, ld2, eps, M);
    else return CUT3R_ERR_ARG;      // widths without the fast path: call cut3r_layernorm twice
    return cut3r_check_launch();
}

extern "C" int cut3r_im2col_patch(const void* img, int u8, int B, int C, int H, int W, int P, void* out, void* stream) {
    if (!img || !out || B <= 0 || C <= 0 || P <= 0 || (P & 7) || H % P || W % P) return CUT3R_ERR_ARG;
    const size_t total = (size_t)B * (H / P) * (W / P) * (C * P * P / 8);
    if (u8)
        hipLaunchKernelGGL((im2col_kernel<true>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, img, B, C, H, W, P, (h16*)out);
    else
        hipLaunchKernelGGL((im2col_kernel<false>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, img, B, C, H, W, P, (h16*)out);
    return cut3r_check_launch();
}

extern "C" int cut3r_cast_f32_f16(const float* x, int ldx, void* y, int ldy, int M, int C, void* stream) {
    if (!x || !y || M <= 0 || C <= 0 || (C & 3) || (ldx & 3) || (ldy & 3)) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(cast_kernel, dim3(grid_for((size_t)M * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, (h16*)y, ldy, M, C / 4);
    return cut3r_check_launch();
}

extern "C" int cut3r_colmean(const float* x, int ldx, int M, int C, float* y, void* stream) {
    if (!x || !y || M <= 0 || C <= 0) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(colmean_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, ldx, M, C, y, 0LL, 0LL);
    return cut3r_check_launch();
}

extern "C" int cut3r_colmean_batched(const float* x, int B, long long stride_x, int ldx, int M, int C, float* y, long long stride_y,
                                     void* stream) {
    if (!x || !y || B <= 0 || B > 65535 || M <= 0 || C <= 0) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(colmean_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x, ldx, M, C, y, stride_x, stride_y);
    return cut3r_check_launch();
}

extern "C" int cut3r_upsample2x_nhwc(const void* in, void* out, int B, int H, int W, int C, void* stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return CUT3R_ERR_ARG;
    const size_t total = (size_t)B * 2 * H * 2 * W * (C / 8);
    hipLaunchKernelGGL(upsample2x_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const h16*)in, (h16*)out, B, H, W, C / 8);
    return cut3r_check_launch();
}

extern "C" int cut3r_dpt_final(const void* in, int P, int Cin, const float* w, const float* b, int mode, float* pts, float* conf,
                               void* stream) {
    if (!in || !w || !b || !pts || P <= 0 || Cin <= 0 || (Cin & 7) || Cin > 2048 || (mode != 0 && mode != 1)) return CUT3R_ERR_ARG;
    if (mode == 0 && !conf) return CUT3R_ERR_ARG;
    const int lpp = Cin / 8;
    if (lpp <= 64 && 64 % lpp == 0) {      // Cin in {8,16,...,512}: coalesced form
        const int ppw = 64 / lpp;
        size_t waves = ((size_t)P + ppw - 1) / ppw;
        size_t blocks = (waves + 3) / 4;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(dpt_final_coalesced_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const h16*)in, P, Cin, w, b,
                           mode, pts, conf);
    } else {
        hipLaunchKernelGGL(dpt_final_kernel, dim3(grid_for((size_t)P)), dim3(256), 4 * Cin * sizeof(float), (hipStream_t)stream,
                           (const h16*)in, P, Cin, w, b, mode, pts, conf);
    }
    return cut3r_check_launch();
}

extern "C" int cut3r_postprocess_pts(const float* raw, int P, int nch, int pos_z, float* pts, float* conf, void* stream) {
    if (!raw || !pts || P <= 0 || (nch != 3 && nch != 4) || (nch == 4 && !conf)) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(postprocess_pts_kernel, dim3(grid_for((size_t)P)), dim3(256), 0, (hipStream_t)stream, raw, P, nch, pos_z, pts, conf);
    return cut3r_check_launch();
}

extern "C" int cut3r_postprocess_pose(const float* raw, int B, float* out, void* stream) {
    if (!raw || !out || B <= 0) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(postprocess_pose_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, raw, B, out);
    return cut3r_check_launch();
}

extern "C" int cut3r_resize_linear_u8(const void* src, int H0, int W0, int C, void* dst, int H1, int W1, int chw_out, void* stream) {
    if (!src || !dst || H0 < 1 || W0 < 1 || H1 < 1 || W1 < 1 || C < 1 || C > 4 || H1 > 65535) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(resize_linear_u8_kernel, dim3((W1 + 255) / 256, H1), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)src, H0, W0, C, (unsigned char*)dst, H1, W1, chw_out);
    return cut3r_check_launch();
}

extern "C" int cut3r_remap_linear_u8(const void* src, int H, int W, int C, const int32_t* map_ix, const int32_t* map_iy, void* dst, int Ho,
                                     int Wo, void* stream) {
    if (!src || !dst || !map_ix || !map_iy || H < 1 || W < 1 || Ho < 1 || Wo < 1 || C < 1 || C > 4 || Ho > 65535) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(remap_linear_u8_kernel, dim3((Wo + 255) / 256, Ho), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src, H, W,
                       C, map_ix, map_iy, (unsigned char*)dst, Ho, Wo);
    return cut3r_check_launch();
}
