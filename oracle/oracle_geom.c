/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the byte/index-exact geometry on the hot path.
 * Compiled by oracle/Makefile into oracle/_build/liboracle_c.so; used by tests/, smoke() and bench.py's
 * cpu_baseline leg as the checker for the HIP kernels.  Never linked into the product.
 *
 * Every function fixes ONE fp32 evaluation order (explicit fmaf chains, IEEE divide, rintf = round-half-even
 * like torch.round) so the HIP kernels can be bit-exact against it.  The reference evaluates the same formulas
 * through torch.bmm/torch.inverse (hislam2/factor_graph.py:255-315), whose summation order is backend-defined;
 * tests/test_oracle_golden.py pins these functions against the reference's own outputs (tests/golden/graph.npz).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

/* rope_2d: /root/reference/src/croco/models/curope/kernels.cu:17-82 (the CUDA order: inv_freq first, then pos*inv_freq).
 * tokens [B,N,H,D] fp32 contiguous, pos [B,N,2] int64. */
void oracle_rope2d_f32(float *tok, const int64_t *pos, int B, int N, int H, int D, float base, float fwd) {
    int Q = D / 4;
    for (int b = 0; b < B; b++)
        for (int n = 0; n < N; n++)
            for (int X = 0; X < 2; X++) {
                float p = (float)pos[((size_t)b * N + n) * 2 + X];
                for (int q = 0; q < Q; q++) {
                    float inv = fwd / powf(base, (float)q / (float)Q);
                    float fr = p * inv;
                    float c = cosf(fr), s = sinf(fr);
                    for (int h = 0; h < H; h++) {
                        float *t = tok + (((size_t)b * N + n) * H + h) * D + X * 2 * Q;
                        float u = t[q], v = t[q + Q];
                        t[q] = u * c - v * s;
                        t[q + Q] = v * c + u * s;
                    }
                }
            }
}

/* One projected point: returns 1 if it lands inside [0,W)x[0,H) with z>0.
 * w2c is the 3x4 top of a world->camera matrix, row-major. clamp_z mirrors `.clamp(min=1e-5)` in
 * cal_overlap_batch (factor_graph.py:272) -- cal_overlap_bi (:304) divides by the raw z. */
static inline int proj_valid(const float *w2c, float x, float y, float z, float fx, float fy, float cx, float cy,
                             int W, int H, int clamp_z) {
    float xc = fmaf(w2c[2], z, fmaf(w2c[1], y, fmaf(w2c[0], x, w2c[3])));
    float yc = fmaf(w2c[6], z, fmaf(w2c[5], y, fmaf(w2c[4], x, w2c[7])));
    float zc = fmaf(w2c[10], z, fmaf(w2c[9], y, fmaf(w2c[8], x, w2c[11])));
    float zd = clamp_z ? (zc < 1e-5f ? 1e-5f : zc) : zc;
    float u = rintf(fx * xc / zd + cx);
    float v = rintf(fy * yc / zd + cy);
    return (u >= 0.0f) && (u < (float)W) && (v >= 0.0f) && (v < (float)H) && (zc > 0.0f);
}

void oracle_overlap_fwd_ex(const float *pm, int N, const float *w2c, int B, const float *K4, int W, int H, int clamp_z,
                           int32_t *counts);

/* cal_overlap_batch (factor_graph.py:255-282): project ONE pointmap [N,3] into B cameras; counts[b] = #valid. */
void oracle_overlap_fwd(const float *pm, int N, const float *w2c, int B, const float *K4, int W, int H,
                        int32_t *counts) {
    oracle_overlap_fwd_ex(pm, N, w2c, B, K4, W, H, 1, counts);
}

/* same, with the divide convention selectable: clamp_z = 0 is cal_overlap_bi called with ONE pointmap and B cameras
 * (factor_graph.py:567 in NMS) */
void oracle_overlap_fwd_ex(const float *pm, int N, const float *w2c, int B, const float *K4, int W, int H, int clamp_z,
                           int32_t *counts) {
    for (int b = 0; b < B; b++) {
        int c = 0;
        for (int n = 0; n < N; n++)
            c += proj_valid(w2c + 12 * b, pm[3 * n], pm[3 * n + 1], pm[3 * n + 2], K4[0], K4[1], K4[2], K4[3], W, H, clamp_z);
        counts[b] = c;
    }
}

/* cal_overlap_bi with B2 == 1 (factor_graph.py:284-315 as called at :186 and :566): project B pointmaps
 * [B,N,3] into ONE camera. */
void oracle_overlap_bwd(const float *pms, int B, int N, const float *w2c, const float *K4, int W, int H,
                        int32_t *counts) {
    for (int b = 0; b < B; b++) {
        int c = 0;
        const float *pm = pms + (size_t)b * N * 3;
        for (int n = 0; n < N; n++)
            c += proj_valid(w2c, pm[3 * n], pm[3 * n + 1], pm[3 * n + 2], K4[0], K4[1], K4[2], K4[3], W, H, 0);
        counts[b] = c;
    }
}

/* Window alignment of one view (hislam2/track_frontend.py:193-243): given pose P (3x4 row-major c2w, already
 * chained), scale s, full-res pts [H,W,3] and conf [H,W]:
 *   pointmap = P * (s*pts)  (geotrf, src/dust3r/utils/geometry.py:49-115), conf' = 1 - 1/conf,
 *   depth' = s * pts_z, outputs stride-`ds` downsampled pointmap/conf and full-res depth. */
void oracle_align_view(const float *pts, const float *conf, int H, int W, const float *P, float s, int ds,
                       float *pm_ds, float *conf_ds, float *depth) {
    int Hd = H / ds, Wd = W / ds;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            size_t i = (size_t)y * W + x;
            depth[i] = s * pts[3 * i + 2];
        }
    for (int y = 0; y < Hd; y++)
        for (int x = 0; x < Wd; x++) {
            size_t i = (size_t)(y * ds) * W + (x * ds), o = (size_t)y * Wd + x;
            float px = s * pts[3 * i], py = s * pts[3 * i + 1], pz = s * pts[3 * i + 2];
            pm_ds[3 * o + 0] = fmaf(P[2], pz, fmaf(P[1], py, fmaf(P[0], px, P[3])));
            pm_ds[3 * o + 1] = fmaf(P[6], pz, fmaf(P[5], py, fmaf(P[4], px, P[7])));
            pm_ds[3 * o + 2] = fmaf(P[10], pz, fmaf(P[9], py, fmaf(P[8], px, P[11])));
            conf_ds[o] = 1.0f - 1.0f / conf[i];
        }
}

/* sum over pixels of log(prev_depth) - log(pts_z) in double (track_frontend.py:216-217 takes the fp32 mean;
 * the double sum is the exact value both fp32 evaluations approximate). */
double oracle_logdepth_sum(const float *prev_depth, const float *pts, int n) {
    double acc = 0.0;
    for (int i = 0; i < n; i++) acc += (double)logf(prev_depth[i]) - (double)logf(pts[3 * i + 2]);
    return acc;
}

/* ---------------------------------------------------------------------------------------------------------------
 * cv2.resize(src, (W1, H1)) with the default INTER_LINEAR on 8-bit images -- what demo_s.py:72,83 applies to every frame.
 * cv2 is NOT in this container and OpenCV is not part of the reference tree: PARITY UNPINNED.  This restates the published
 * algorithm of OpenCV 4.x imgproc/resize.cpp (resizeGeneric_ with HResizeLinear<uchar,int,short,INTER_RESIZE_COEF_SCALE>
 * and VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,INTER_RESIZE_COEF_BITS*2>>):
 *   scale = src/dst (double); f = (float)((d + 0.5) * scale - 0.5); s = floor(f); f -= s;
 *   x only: s < 0 -> (s, f) = (0, 0);  s >= W0-1 -> (s, f) = (W0-1, 0);   y: the two rows s, s+1 are clipped to [0, H0-1]
 *   coefficients short(round((1-f) * 2048)), short(round(f * 2048)) (round half to even, as cvRound);
 *   horizontal: int D = S[s]*a0 + S[s+1]*a1;  vertical: u8 = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2.
 * Exact 2x decimation (W0 == 2*W1 and H0 == 2*H1) takes OpenCV's INTER_AREA fast path instead: (a + b + c + d + 2) >> 2.
 * src/dst interleaved HWC, C channels. */
#include <math.h>
static short coef_q11(float v) { return (short)lrintf(v * 2048.0f); }

void oracle_resize_linear_u8(const unsigned char *src, int H0, int W0, int C, unsigned char *dst, int H1, int W1) {
    if (W0 == 2 * W1 && H0 == 2 * H1) {
        for (int y = 0; y < H1; y++)
            for (int x = 0; x < W1; x++)
                for (int c = 0; c < C; c++) {
                    const unsigned char *p = src + ((size_t)(2 * y) * W0 + 2 * x) * C + c;
                    dst[((size_t)y * W1 + x) * C + c] = (unsigned char)((p[0] + p[C] + p[(size_t)W0 * C] + p[(size_t)W0 * C + C] + 2) >> 2);
                }
        return;
    }
    const double sx_ = (double)W0 / W1, sy_ = (double)H0 / H1;
    for (int y = 0; y < H1; y++) {
        float fy = (float)((y + 0.5) * sy_ - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        const short b0 = coef_q11(1.0f - fy), b1 = coef_q11(fy);
        int y0 = sy < 0 ? 0 : (sy > H0 - 1 ? H0 - 1 : sy);
        int y1 = sy + 1 < 0 ? 0 : (sy + 1 > H0 - 1 ? H0 - 1 : sy + 1);
        for (int x = 0; x < W1; x++) {
            float fx = (float)((x + 0.5) * sx_ - 0.5);
            int sx = (int)floorf(fx);
            fx -= sx;
            if (sx < 0) { fx = 0.f; sx = 0; }
            if (sx >= W0 - 1) { fx = 0.f; sx = W0 - 1; }
            const short a0 = coef_q11(1.0f - fx), a1 = coef_q11(fx);
            const int x1 = sx + 1 > W0 - 1 ? W0 - 1 : sx + 1;
            for (int c = 0; c < C; c++) {
                const int d0 = src[((size_t)y0 * W0 + sx) * C + c] * a0 + src[((size_t)y0 * W0 + x1) * C + c] * a1;
                const int d1 = src[((size_t)y1 * W0 + sx) * C + c] * a0 + src[((size_t)y1 * W0 + x1) * C + c] * a1;
                dst[((size_t)y * W1 + x) * C + c] = (unsigned char)((((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2);
            }
        }
    }
}

/* cv2.remap(src, map, INTER_LINEAR, BORDER_CONSTANT 0) for uint8 HWC with a fixed-point map (ix, iy = round(32 * source coordinate)):
 * OpenCV's remapBilinear with FixedPtCast<int, uchar, 15> -- weights BilinearTab_i = 32*a*b (exact for 5-bit fractions),
 * (sum + 2^14) >> 15, taps outside the image read the border value 0.  TEST INFRASTRUCTURE; parity unpinned vs cv2. */
void oracle_remap_linear_u8(const unsigned char *src, int H, int W, int C, const int *map_ix, const int *map_iy, unsigned char *dst,
                            int Ho, int Wo) {
    for (int y = 0; y < Ho; y++)
        for (int x = 0; x < Wo; x++) {
            const int ix = map_ix[(size_t)y * Wo + x], iy = map_iy[(size_t)y * Wo + x];
            const int sx = ix >> 5, sy = iy >> 5, fx = ix & 31, fy = iy & 31;
            const int w[4] = {32 * (32 - fx) * (32 - fy), 32 * fx * (32 - fy), 32 * (32 - fx) * fy, 32 * fx * fy};
            for (int c = 0; c < C; c++) {
                int acc = 1 << 14;
                for (int k = 0; k < 4; k++) {
                    const int xx = sx + (k & 1), yy = sy + (k >> 1);
                    const int p = (xx >= 0 && xx < W && yy >= 0 && yy < H) ? src[((size_t)yy * W + xx) * C + c] : 0;
                    acc += p * w[k];
                }
                acc >>= 15;
                dst[((size_t)y * Wo + x) * C + c] = (unsigned char)(acc > 255 ? 255 : acc);
            }
        }
}
