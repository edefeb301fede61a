// The Gaussian mapper's training step WITHOUT an autograd tape: everything around the rasteriser passes of one mapping / pose-refinement
// iteration (hislam2/gs_backend_per_frame.py:202-326 pose_refine, :451-587 optimization) as a handful of fused kernels.
//
// What the reference does per rendered view, in ~100 small torch launches (activations, transform_to_frame of renderer/__init__.py:89-152,
// loss algebra, Adam of every parameter group, slam_utils.py:77-102 update_pose):
//
//   theta [P,14] (xyz | colour | opacity logit | log scale | quaternion rxyz),  pose = exp([tau, phi]) * T_w2c
//     -> camera-frame means, world->camera rotated quaternions, exp / sigmoid activations          gs_activate_kernel        (1 launch)
//     -> rasteriser forward (gs.hip), loss kernels (gs.hip), rasteriser backward (gs.hip)
//     -> gradients of theta (activation chain rules, isotropy term), and 16 sums per view from which the
//        gradient of the 6 pose increments follows                                                 gs_activate_bwd_kernel    (1 launch)
//     -> pose gradient through exp() (forward-mode duals), Adam of the increments, optional fold   gs_pose_step_kernel       (1 launch)
//     -> Adam of all Gaussian parameters                                                           gs_adam_kernel            (1 launch)
//
// The pose gradient: p_cam = R_E y + t_E with y = T p and (t_E, q_E) = exp(tau, phi); q_cam = q_E * (q_T * q_g).  Both are linear in
// R_E, t_E, q_E, so   dL/d delta_k = <dR_E/d delta_k, M> + <dt_E/d delta_k, s> + <dq_E/d delta_k, r>   with
// M = sum_p g_p y_p^T (9), s = sum_p g_p (3), r = sum_p g^q_p * conj(q_T * q_g,p) (4): 16 sums over the Gaussians, then a one-thread finish.
#include "common.h"
#include "lie_math.h"
#include "../../include/cut3r_hip.h"

namespace {
using namespace liemath;

// pose_state: 32 floats per view -- [0:7] world->camera (t, q_xyzw) | [7:13] increment (tau, phi) | [13:19] Adam m | [19:25] Adam v | [25] steps

struct Pose { V3<float> t; Q4<float> q; V3<float> tT; Q4<float> qT; };

DEVINL Pose load_pose(const float* __restrict__ ps) {
    Pose o;
    o.tT = {ps[0], ps[1], ps[2]};
    o.qT = {ps[3], ps[4], ps[5], ps[6]};
    V3<float> tE; Q4<float> qE;
    se3_exp<float>({ps[7], ps[8], ps[9]}, {ps[10], ps[11], ps[12]}, tE, qE);
    o.q = qmul(qE, o.qT);
    o.t = add(qrot(qE, o.tT), tE);
    return o;
}

__global__ __launch_bounds__(256) void gs_activate_kernel(int P, const float* __restrict__ theta, const float* __restrict__ ps,
                                                          float* __restrict__ means, float* __restrict__ scales, float* __restrict__ rots,
                                                          float* __restrict__ opac, float* __restrict__ shs) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const Pose pz = load_pose(ps);
    const float* th = theta + (size_t)i * 14;
    const V3<float> pc = add(qrot(pz.q, V3<float>{th[0], th[1], th[2]}), pz.t);
    means[3 * i + 0] = pc.x; means[3 * i + 1] = pc.y; means[3 * i + 2] = pc.z;
    shs[3 * i + 0] = th[3]; shs[3 * i + 1] = th[4]; shs[3 * i + 2] = th[5];
    opac[i] = 1.0f / (1.0f + expf(-th[6]));
    scales[3 * i + 0] = expf(th[7]); scales[3 * i + 1] = expf(th[8]); scales[3 * i + 2] = expf(th[9]);
    const float qr = th[10], qx = th[11], qy = th[12], qz = th[13];
    const float inv = 1.0f / fmaxf(sqrtf(qr * qr + qx * qx + qy * qy + qz * qz), 1e-12f);      // F.normalize
    const Q4<float> c = qmul(pz.q, Q4<float>{qx * inv, qy * inv, qz * inv, qr * inv});
    rots[4 * i + 0] = c.w; rots[4 * i + 1] = c.x; rots[4 * i + 2] = c.y; rots[4 * i + 3] = c.z;
}

__global__ __launch_bounds__(256) void gs_count_vis_kernel(int P, const int* __restrict__ radii, float* __restrict__ nvis) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float c = (i < P && radii[i] > 0) ? 1.f : 0.f;
    c = wave_sum(c);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float s = part[0] + part[1] + part[2] + part[3];
        if (s != 0.f) atomicAdd(nvis, s);
    }
}

// gtheta (nullable) accumulates; sums (nullable) accumulates the 16 pose sums
__global__ __launch_bounds__(256) void gs_activate_bwd_kernel(int P, const float* __restrict__ theta, const float* __restrict__ ps,
                                                              const float* __restrict__ d_means, const float* __restrict__ d_scales,
                                                              const float* __restrict__ d_rots, const float* __restrict__ d_opac,
                                                              const float* __restrict__ d_shs, const int* __restrict__ radii,
                                                              const float* __restrict__ nvis, float iso_coef, float* __restrict__ gtheta,
                                                              float* __restrict__ sums) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; k++) acc[k] = 0.f;
    if (i < P) {
        const Pose pz = load_pose(ps);
        const float* th = theta + (size_t)i * 14;
        const V3<float> g = {d_means[3 * i], d_means[3 * i + 1], d_means[3 * i + 2]};
        const float qr = th[10], qx = th[11], qy = th[12], qz = th[13];
        const float nrm = fmaxf(sqrtf(qr * qr + qx * qx + qy * qy + qz * qz), 1e-12f), inv = 1.0f / nrm;
        const Q4<float> gq = {qx * inv, qy * inv, qz * inv, qr * inv};
        const Q4<float> gc = {d_rots[4 * i + 1], d_rots[4 * i + 2], d_rots[4 * i + 3], d_rots[4 * i + 0]};      // xyzw
        if (gtheta) {
            float* o = gtheta + (size_t)i * 14;
            const V3<float> gx = qrot(qconj(pz.q), g);                                         // R^T g
            o[0] += gx.x; o[1] += gx.y; o[2] += gx.z;
            o[3] += d_shs[3 * i]; o[4] += d_shs[3 * i + 1]; o[5] += d_shs[3 * i + 2];
            const float sg = 1.0f / (1.0f + expf(-th[6]));
            o[6] += d_opac[i] * sg * (1.0f - sg);
            const float s0 = expf(th[7]), s1 = expf(th[8]), s2 = expf(th[9]);
            float i0 = 0.f, i1 = 0.f, i2 = 0.f;
            if (iso_coef != 0.f && radii[i] > 0) {
                // d/ds of sum_k |s_k - mean(s)| / max(3 nvis, 1): sign_k - mean(sign)   (torch.abs: sign(0) = 0)
                const float mean = (s0 + s1 + s2) * (1.0f / 3.0f);
                const float a0 = (s0 > mean) - (s0 < mean), a1 = (s1 > mean) - (s1 < mean), a2 = (s2 > mean) - (s2 < mean);
                const float am = (a0 + a1 + a2) * (1.0f / 3.0f);
                const float c = iso_coef / fmaxf(3.0f * nvis[0], 1.0f);
                i0 = c * (a0 - am); i1 = c * (a1 - am); i2 = c * (a2 - am);
            }
            o[7] += (d_scales[3 * i] + i0) * s0; o[8] += (d_scales[3 * i + 1] + i1) * s1; o[9] += (d_scales[3 * i + 2] + i2) * s2;
            const Q4<float> gh = qmul(qconj(pz.q), gc);                                         // dL/d normalised quaternion (xyzw)
            const float dot = gh.x * gq.x + gh.y * gq.y + gh.z * gq.z + gh.w * gq.w;
            o[10] += (gh.w - gq.w * dot) * inv; o[11] += (gh.x - gq.x * dot) * inv;
            o[12] += (gh.y - gq.y * dot) * inv; o[13] += (gh.z - gq.z * dot) * inv;
        }
        if (sums) {
            const V3<float> y = add(qrot(pz.qT, V3<float>{th[0], th[1], th[2]}), pz.tT);
            acc[0] = g.x * y.x; acc[1] = g.x * y.y; acc[2] = g.x * y.z;
            acc[3] = g.y * y.x; acc[4] = g.y * y.y; acc[5] = g.y * y.z;
            acc[6] = g.z * y.x; acc[7] = g.z * y.y; acc[8] = g.z * y.z;
            acc[9] = g.x; acc[10] = g.y; acc[11] = g.z;
            const Q4<float> r = qmul(gc, qconj(qmul(pz.qT, gq)));
            acc[12] = r.x; acc[13] = r.y; acc[14] = r.z; acc[15] = r.w;
        }
    }
    if (!sums) return;
    __shared__ float part[4][16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const float v = wave_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const float v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (v != 0.f) atomicAdd(sums + threadIdx.x, v);
    }
}

// one thread: gradient of the increments from the 16 sums (+ the pull to the starting pose of pose_refine: prior * (2 - ratio) * |delta|^2),
// torch.optim.Adam step (lr_trans for tau, lr_rot for phi, betas 0.9 / 0.999, eps 1e-8), and with `fold` the update of slam_utils.py:77-91
// (T <- exp(delta) T, delta <- 0)
__global__ void gs_pose_step_kernel(float* __restrict__ ps, const float* __restrict__ sums, float prior, const float* __restrict__ ratio,
                                    float lr_rot, float lr_trans, int fold) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (fold != 2) {                               // (fold == 2: only the fold below, no gradient step)
    typedef Dual<6> D;
    V3<D> tau, phi;
    D* in[6] = {&tau.x, &tau.y, &tau.z, &phi.x, &phi.y, &phi.z};
    for (int k = 0; k < 6; k++) {
        *in[k] = D(ps[7 + k]);
        in[k]->d[k] = 1.f;
    }
    V3<D> tE; Q4<D> qE;
    se3_exp<D>(tau, phi, tE, qE);
    // rotation matrix of q_E with its derivatives
    const D two(2.0f), one(1.0f);
    D R[9];
    R[0] = one - two * (qE.y * qE.y + qE.z * qE.z); R[1] = two * (qE.x * qE.y - qE.z * qE.w); R[2] = two * (qE.x * qE.z + qE.y * qE.w);
    R[3] = two * (qE.x * qE.y + qE.z * qE.w); R[4] = one - two * (qE.x * qE.x + qE.z * qE.z); R[5] = two * (qE.y * qE.z - qE.x * qE.w);
    R[6] = two * (qE.x * qE.z - qE.y * qE.w); R[7] = two * (qE.y * qE.z + qE.x * qE.w); R[8] = one - two * (qE.x * qE.x + qE.y * qE.y);
    const float pc = prior * (ratio ? (2.0f - ratio[0]) : 1.0f);
    float g[6];
    for (int k = 0; k < 6; k++) {
        float a = 0.f;
        for (int e = 0; e < 9; e++) a = fmaf(R[e].d[k], sums[e], a);
        a = fmaf(tE.x.d[k], sums[9], a); a = fmaf(tE.y.d[k], sums[10], a); a = fmaf(tE.z.d[k], sums[11], a);
        a = fmaf(qE.x.d[k], sums[12], a); a = fmaf(qE.y.d[k], sums[13], a); a = fmaf(qE.z.d[k], sums[14], a); a = fmaf(qE.w.d[k], sums[15], a);
        g[k] = a + 2.0f * pc * ps[7 + k];
    }
    const float step = ps[25] + 1.0f;
    ps[25] = step;
    const float b1 = 0.9f, b2 = 0.999f;
    const float bc1 = 1.0f - powf(b1, step), bc2 = 1.0f - powf(b2, step);
    for (int k = 0; k < 6; k++) {
        const float m = b1 * ps[13 + k] + (1.0f - b1) * g[k];
        const float v = b2 * ps[19 + k] + (1.0f - b2) * g[k] * g[k];
        ps[13 + k] = m; ps[19 + k] = v;
        const float lr = k < 3 ? lr_trans : lr_rot;
        ps[7 + k] -= (lr / bc1) * m / (sqrtf(v) / sqrtf(bc2) + 1e-8f);
    }
    }
    if (fold) {
        V3<float> tE2; Q4<float> qE2;
        se3_exp<float>({ps[7], ps[8], ps[9]}, {ps[10], ps[11], ps[12]}, tE2, qE2);
        const Q4<float> qT = {ps[3], ps[4], ps[5], ps[6]};
        const V3<float> t = add(qrot(qE2, V3<float>{ps[0], ps[1], ps[2]}), tE2);
        const Q4<float> q = qmul(qE2, qT);
        ps[0] = t.x; ps[1] = t.y; ps[2] = t.z; ps[3] = q.x; ps[4] = q.y; ps[5] = q.z; ps[6] = q.w;
        for (int k = 0; k < 6; k++) ps[7 + k] = 0.f;
    }
}

// theta -= lr * (m / bc1) / (sqrt(v / bc2) + eps)   (GaussianMap.step: torch.optim.Adam with eps = 1e-15, gaussian_model.py:374-417)
__global__ __launch_bounds__(256) void gs_adam_kernel(long long n, float* __restrict__ theta, float* __restrict__ m, float* __restrict__ v,
                                                      const float* __restrict__ g, const float* __restrict__ lr14, float b1, float b2, float bc1,
                                                      float bc2, float eps) {
    const long long i = blockIdx.x * (long long)256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    theta[i] -= lr14[i % 14] * (mi / bc1) / (sqrtf(vi / bc2) + eps);
}

// coefficients of cut3r_pixel_loss_backward from the forward sums (gs_backend_per_frame.py:516-531): upstream gradient g
__global__ void gs_map_coef_kernel(const float* __restrict__ sums, float w_rgb, float w_depth, float w_normal, float g, float hw,
                                   float* __restrict__ coef, float* __restrict__ loss_acc) {
    if (threadIdx.x != 0) return;
    const float nd = fmaxf(sums[3], 1.0f);
    coef[0] = g * w_rgb / (3.0f * hw);
    coef[1] = g * w_depth / nd;
    coef[2] = g * w_normal / nd;
    if (loss_acc) loss_acc[0] += g * (w_rgb * sums[0] / (3.0f * hw) + (w_depth * sums[1] + w_normal * sums[2]) / nd);
}

// coefficients of cut3r_refine_loss_backward from the forward sums (gs_backend_per_frame.py:240-262); ratio_out[0] = covered share
__global__ void gs_refine_coef_kernel(const float* __restrict__ sums, float g_rgb, float g_var, float hw, float* __restrict__ coef,
                                      float* __restrict__ ratio_out, float* __restrict__ loss_acc) {
    if (threadIdx.x != 0) return;
    const float na = fmaxf(sums[1], 1.0f), nm = fmaxf(sums[4], 1.0f);
    const float ratio = sums[1] / hw, mean = sums[2] / nm;
    coef[0] = g_rgb * ratio / (3.0f * na);
    coef[1] = g_var * ratio / nm;
    coef[2] = mean;
    ratio_out[0] = ratio;
    if (loss_acc) loss_acc[0] += g_rgb * ratio * sums[0] / (3.0f * na) + g_var * ratio * (sums[3] / nm - mean * mean);
}

}  // namespace

extern "C" int cut3r_gs_activate(int P, const float* theta, const float* pose_state, float* means, float* scales, float* rots, float* opac,
                                 float* shs, void* stream) {
    if (P <= 0 || !theta || !pose_state || !means || !scales || !rots || !opac || !shs) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(gs_activate_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, theta, pose_state, means, scales, rots, opac,
                       shs);
    return cut3r_check_launch();
}

extern "C" int cut3r_gs_activate_backward(int P, const float* theta, const float* pose_state, const float* d_means, const float* d_scales,
                                          const float* d_rots, const float* d_opac, const float* d_shs, const int* radii, float iso_coef,
                                          float* nvis_ws, float* gtheta, float* pose_sums, void* stream) {
    if (P <= 0 || !theta || !pose_state || !d_means || !d_scales || !d_rots || !d_opac || !d_shs || (!gtheta && !pose_sums)) return CUT3R_ERR_ARG;
    if (iso_coef != 0.f && (!radii || !nvis_ws)) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (iso_coef != 0.f && gtheta) {
        if (hipMemsetAsync(nvis_ws, 0, sizeof(float), s) != hipSuccess) return CUT3R_ERR_LAUNCH;
        hipLaunchKernelGGL(gs_count_vis_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, radii, nvis_ws);
    }
    hipLaunchKernelGGL(gs_activate_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, theta, pose_state, d_means, d_scales, d_rots, d_opac, d_shs,
                       radii, nvis_ws, gtheta ? iso_coef : 0.f, gtheta, pose_sums);
    return cut3r_check_launch();
}

extern "C" int cut3r_gs_pose_step(float* pose_state, const float* pose_sums, float prior, const float* ratio, float lr_rot, float lr_trans,
                                  int fold, void* stream) {
    if (!pose_state || !pose_sums) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(gs_pose_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pose_state, pose_sums, prior, ratio, lr_rot, lr_trans, fold);
    return cut3r_check_launch();
}

extern "C" int cut3r_gs_adam(long long n, float* theta, float* m, float* v, const float* grad, const float* lr14, float b1, float b2, float bc1,
                             float bc2, float eps, void* stream) {
    if (n <= 0 || n % 14 || !theta || !m || !v || !grad || !lr14) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(gs_adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, theta, m, v, grad, lr14, b1, b2, bc1,
                       bc2, eps);
    return cut3r_check_launch();
}

extern "C" int cut3r_gs_map_coef(const float* sums, float w_rgb, float w_depth, float w_normal, float g, int H, int W, float* coef,
                                 float* loss_acc, void* stream) {
    if (!sums || !coef || H <= 0 || W <= 0) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(gs_map_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, w_rgb, w_depth, w_normal, g, (float)H * (float)W, coef,
                       loss_acc);
    return cut3r_check_launch();
}

extern "C" int cut3r_gs_refine_coef(const float* sums, float g_rgb, float g_var, int H, int W, float* coef, float* ratio_out, float* loss_acc,
                                    void* stream) {
    if (!sums || !coef || !ratio_out || H <= 0 || W <= 0) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(gs_refine_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, g_rgb, g_var, (float)H * (float)W, coef, ratio_out,
                       loss_acc);
    return cut3r_check_launch();
}
