// Lie-group kernels (SO3 / SE3 / Sim3) for the loop-closure optimiser and the dense-BA operators on gfx950.
//
// Fills the slot of the reference's un-vendored dependency princeton-vl/lietorch ("lietorch_backends", version 0.2 --
// /root/reference/thirdparty/lietorch is an empty directory): exp, log, group product, inverse, point action,
// adjoint / transposed adjoint, 4x4 matrix, each with its vector-Jacobian product for autograd.  Conventions follow
// the reference call sites: data = [t(3), q_xyzw(4) (, s)] (hislam2/keyframe.py:25, gs_backend_per_frame.py:721-725),
// tangent = [tau(3), phi(3) (, sigma)], retr(a) = exp(a) * X (hislam2/geom/ba.py:29,37).
//
// Every op is one thread per group element (these are O(#poses) = tens..hundreds of elements: pure launch latency,
// so the backward passes evaluate the SAME templated forward code on forward-mode dual numbers and contract the small
// Jacobian in registers instead of hand-deriving each adjoint).
#include "common.h"
#include "lie_math.h"
#include "../../include/cut3r_hip.h"

namespace {

using namespace liemath;

// ------------------------------------------------------------------------------------------------ kernels
enum { OP_EXP = 0, OP_LOG = 1, OP_INV = 2, OP_MATRIX = 3 };

template <int G, int OP> __global__ void lie_unary_kernel(const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int NI = (OP == OP_EXP) ? tdim(G) : ddim(G);
    constexpr int NO = (OP == OP_EXP) ? ddim(G) : (OP == OP_LOG ? tdim(G) : (OP == OP_INV ? ddim(G) : 16));
    float a[NI], o[NO];
#pragma unroll
    for (int k = 0; k < NI; k++) a[k] = in[(size_t)i * NI + k];
    if (OP == OP_EXP) f_exp<G, float>(a, o);
    else if (OP == OP_LOG) f_log<G, float>(a, o);
    else if (OP == OP_INV) f_inv<G, float>(a, o);
    else f_matrix<G, float>(a, o);
#pragma unroll
    for (int k = 0; k < NO; k++) out[(size_t)i * NO + k] = o[k];
}

// vector-Jacobian product of a unary op: gin[j] = sum_k gout[k] * d out[k] / d in[j]
template <int G, int OP> __global__ void lie_unary_bwd_kernel(const float* __restrict__ in, const float* __restrict__ gout,
                                                              float* __restrict__ gin, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int NI = (OP == OP_EXP) ? tdim(G) : ddim(G);
    constexpr int NO = (OP == OP_EXP) ? ddim(G) : (OP == OP_LOG ? tdim(G) : (OP == OP_INV ? ddim(G) : 16));
    typedef Dual<NI> D;
    D a[NI], o[NO];
#pragma unroll
    for (int k = 0; k < NI; k++) { a[k] = D(in[(size_t)i * NI + k]); a[k].d[k] = 1.f; }
    if (OP == OP_EXP) f_exp<G, D>(a, o);
    else if (OP == OP_LOG) f_log<G, D>(a, o);
    else if (OP == OP_INV) f_inv<G, D>(a, o);
    else f_matrix<G, D>(a, o);
#pragma unroll
    for (int j = 0; j < NI; j++) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NO; k++) s += gout[(size_t)i * NO + k] * o[k].d[j];
        gin[(size_t)i * NI + j] = s;
    }
}

template <int G> __global__ void lie_mul_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int ND = ddim(G);
    float a[ND], b[ND], o[ND];
#pragma unroll
    for (int k = 0; k < ND; k++) { a[k] = x[(size_t)i * ND + k]; b[k] = y[(size_t)i * ND + k]; }
    f_mul<G, float>(a, b, o);
#pragma unroll
    for (int k = 0; k < ND; k++) out[(size_t)i * ND + k] = o[k];
}

template <int G> __global__ void lie_mul_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                    const float* __restrict__ gout, float* __restrict__ gx, float* __restrict__ gy, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int ND = ddim(G);
    typedef Dual<2 * ND> D;
    D a[ND], b[ND], o[ND];
#pragma unroll
    for (int k = 0; k < ND; k++) {
        a[k] = D(x[(size_t)i * ND + k]); a[k].d[k] = 1.f;
        b[k] = D(y[(size_t)i * ND + k]); b[k].d[ND + k] = 1.f;
    }
    f_mul<G, D>(a, b, o);
#pragma unroll
    for (int j = 0; j < ND; j++) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int k = 0; k < ND; k++) { const float gk = gout[(size_t)i * ND + k]; s0 += gk * o[k].d[j]; s1 += gk * o[k].d[ND + j]; }
        gx[(size_t)i * ND + j] = s0;
        gy[(size_t)i * ND + j] = s1;
    }
}

// points: one element acts on P points (x broadcast over points when xs == 0)
template <int G> __global__ void lie_act_kernel(const float* __restrict__ x, const float* __restrict__ p, float* __restrict__ out,
                                                int n, int P, int pd) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= (size_t)n * P) return;
    constexpr int ND = ddim(G);
    const int e = (int)(i / P);
    float a[ND], q[4], o[4];
#pragma unroll
    for (int k = 0; k < ND; k++) a[k] = x[(size_t)e * ND + k];
    for (int k = 0; k < pd; k++) q[k] = p[i * pd + k];
    f_act<G, float>(a, q, pd, o);
    for (int k = 0; k < pd; k++) out[i * pd + k] = o[k];
}

template <int G> __global__ void lie_act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ p, const float* __restrict__ gout,
                                                    float* __restrict__ gx_partial, float* __restrict__ gp, int n, int P, int pd) {
    // one thread per (element, point): gp exact; gx accumulated with atomics into [n, ND] (zeroed by the launcher)
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= (size_t)n * P) return;
    constexpr int ND = ddim(G);
    typedef Dual<ND + 4> D;
    const int e = (int)(i / P);
    D a[ND], q[4], o[4];
#pragma unroll
    for (int k = 0; k < ND; k++) { a[k] = D(x[(size_t)e * ND + k]); a[k].d[k] = 1.f; }
    for (int k = 0; k < 4; k++) q[k] = D(0.f);
    for (int k = 0; k < pd; k++) { q[k] = D(p[i * pd + k]); q[k].d[ND + k] = 1.f; }
    f_act<G, D>(a, q, pd, o);
    for (int j = 0; j < pd; j++) {
        float s = 0.f;
        for (int k = 0; k < pd; k++) s += gout[i * pd + k] * o[k].d[ND + j];
        gp[i * pd + j] = s;
    }
#pragma unroll
    for (int j = 0; j < ND; j++) {
        float s = 0.f;
        for (int k = 0; k < pd; k++) s += gout[i * pd + k] * o[k].d[j];
        atomicAdd(&gx_partial[(size_t)e * ND + j], s);
    }
}

template <int G> __global__ void lie_adj_kernel(const float* __restrict__ x, const float* __restrict__ a, float* __restrict__ out, int n,
                                                int transpose) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int ND = ddim(G), NT = tdim(G);
    float e[ND], v[NT], o[NT];
#pragma unroll
    for (int k = 0; k < ND; k++) e[k] = x[(size_t)i * ND + k];
#pragma unroll
    for (int k = 0; k < NT; k++) v[k] = a[(size_t)i * NT + k];
    f_adj<G, float>(e, v, transpose, o);
#pragma unroll
    for (int k = 0; k < NT; k++) out[(size_t)i * NT + k] = o[k];
}

#define LAUNCH1(kern, ...) hipLaunchKernelGGL(kern, dim3((n + 127) / 128), dim3(128), 0, s, __VA_ARGS__)

template <int OP> int unary_dispatch(int group, const float* in, float* out, int n, hipStream_t s) {
    switch (group) {
        case 0: LAUNCH1((lie_unary_kernel<0, OP>), in, out, n); break;
        case 1: LAUNCH1((lie_unary_kernel<1, OP>), in, out, n); break;
        case 2: LAUNCH1((lie_unary_kernel<2, OP>), in, out, n); break;
        default: return CUT3R_ERR_ARG;
    }
    return cut3r_check_launch();
}
template <int OP> int unary_bwd_dispatch(int group, const float* in, const float* gout, float* gin, int n, hipStream_t s) {
    switch (group) {
        case 0: LAUNCH1((lie_unary_bwd_kernel<0, OP>), in, gout, gin, n); break;
        case 1: LAUNCH1((lie_unary_bwd_kernel<1, OP>), in, gout, gin, n); break;
        case 2: LAUNCH1((lie_unary_bwd_kernel<2, OP>), in, gout, gin, n); break;
        default: return CUT3R_ERR_ARG;
    }
    return cut3r_check_launch();
}

}  // namespace

extern "C" int cut3r_lie_unary(int group, int op, const float* in, float* out, int n, void* stream) {
    if (!in || !out || n <= 0) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (op) {
        case OP_EXP: return unary_dispatch<OP_EXP>(group, in, out, n, s);
        case OP_LOG: return unary_dispatch<OP_LOG>(group, in, out, n, s);
        case OP_INV: return unary_dispatch<OP_INV>(group, in, out, n, s);
        case OP_MATRIX: return unary_dispatch<OP_MATRIX>(group, in, out, n, s);
        default: return CUT3R_ERR_ARG;
    }
}

extern "C" int cut3r_lie_unary_bwd(int group, int op, const float* in, const float* grad_out, float* grad_in, int n, void* stream) {
    if (!in || !grad_out || !grad_in || n <= 0) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (op) {
        case OP_EXP: return unary_bwd_dispatch<OP_EXP>(group, in, grad_out, grad_in, n, s);
        case OP_LOG: return unary_bwd_dispatch<OP_LOG>(group, in, grad_out, grad_in, n, s);
        case OP_INV: return unary_bwd_dispatch<OP_INV>(group, in, grad_out, grad_in, n, s);
        case OP_MATRIX: return unary_bwd_dispatch<OP_MATRIX>(group, in, grad_out, grad_in, n, s);
        default: return CUT3R_ERR_ARG;
    }
}

extern "C" int cut3r_lie_mul(int group, const float* x, const float* y, float* out, int n, void* stream) {
    if (!x || !y || !out || n <= 0) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (group) {
        case 0: LAUNCH1((lie_mul_kernel<0>), x, y, out, n); break;
        case 1: LAUNCH1((lie_mul_kernel<1>), x, y, out, n); break;
        case 2: LAUNCH1((lie_mul_kernel<2>), x, y, out, n); break;
        default: return CUT3R_ERR_ARG;
    }
    return cut3r_check_launch();
}

extern "C" int cut3r_lie_mul_bwd(int group, const float* x, const float* y, const float* grad_out, float* grad_x, float* grad_y, int n,
                                 void* stream) {
    if (!x || !y || !grad_out || !grad_x || !grad_y || n <= 0) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (group) {
        case 0: LAUNCH1((lie_mul_bwd_kernel<0>), x, y, grad_out, grad_x, grad_y, n); break;
        case 1: LAUNCH1((lie_mul_bwd_kernel<1>), x, y, grad_out, grad_x, grad_y, n); break;
        case 2: LAUNCH1((lie_mul_bwd_kernel<2>), x, y, grad_out, grad_x, grad_y, n); break;
        default: return CUT3R_ERR_ARG;
    }
    return cut3r_check_launch();
}

extern "C" int cut3r_lie_act(int group, const float* x, const float* p, float* out, int n, int P, int pd, void* stream) {
    if (!x || !p || !out || n <= 0 || P <= 0 || (pd != 3 && pd != 4)) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t tot = (size_t)n * P;
    dim3 grid((unsigned)((tot + 255) / 256)), block(256);
    switch (group) {
        case 0: hipLaunchKernelGGL((lie_act_kernel<0>), grid, block, 0, s, x, p, out, n, P, pd); break;
        case 1: hipLaunchKernelGGL((lie_act_kernel<1>), grid, block, 0, s, x, p, out, n, P, pd); break;
        case 2: hipLaunchKernelGGL((lie_act_kernel<2>), grid, block, 0, s, x, p, out, n, P, pd); break;
        default: return CUT3R_ERR_ARG;
    }
    return cut3r_check_launch();
}

extern "C" int cut3r_lie_act_bwd(int group, const float* x, const float* p, const float* grad_out, float* grad_x, float* grad_p, int n,
                                 int P, int pd, void* stream) {
    if (!x || !p || !grad_out || !grad_x || !grad_p || n <= 0 || P <= 0 || (pd != 3 && pd != 4)) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nd = ddim(group);
    if (hipMemsetAsync(grad_x, 0, sizeof(float) * (size_t)n * nd, s) != hipSuccess) return CUT3R_ERR_LAUNCH;
    const size_t tot = (size_t)n * P;
    dim3 grid((unsigned)((tot + 127) / 128)), block(128);
    switch (group) {
        case 0: hipLaunchKernelGGL((lie_act_bwd_kernel<0>), grid, block, 0, s, x, p, grad_out, grad_x, grad_p, n, P, pd); break;
        case 1: hipLaunchKernelGGL((lie_act_bwd_kernel<1>), grid, block, 0, s, x, p, grad_out, grad_x, grad_p, n, P, pd); break;
        case 2: hipLaunchKernelGGL((lie_act_bwd_kernel<2>), grid, block, 0, s, x, p, grad_out, grad_x, grad_p, n, P, pd); break;
        default: return CUT3R_ERR_ARG;
    }
    return cut3r_check_launch();
}

extern "C" int cut3r_lie_adj(int group, const float* x, const float* a, float* out, int n, int transpose, void* stream) {
    if (!x || !a || !out || n <= 0) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (group) {
        case 0: LAUNCH1((lie_adj_kernel<0>), x, a, out, n, transpose); break;
        case 1: LAUNCH1((lie_adj_kernel<1>), x, a, out, n, transpose); break;
        case 2: LAUNCH1((lie_adj_kernel<2>), x, a, out, n, transpose); break;
        default: return CUT3R_ERR_ARG;
    }
    return cut3r_check_launch();
}
