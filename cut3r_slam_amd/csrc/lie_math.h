// Templated Lie-group math (SO3/SE3/Sim3) over float or forward-mode dual numbers; shared by lie.hip and lc.hip.
#pragma once
#include "common.h"

namespace liemath {

// ------------------------------------------------------------------------------------------------ dual numbers
template <int N>
struct Dual {
    float v;
    float d[N];
    DEVINL Dual() {}
    DEVINL Dual(float x) : v(x) {
#pragma unroll
        for (int i = 0; i < N; i++) d[i] = 0.f;
    }
};
template <int N> DEVINL Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] + b.d[i];
    return r;
}
template <int N> DEVINL Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] - b.d[i];
    return r;
}
template <int N> DEVINL Dual<N> operator-(const Dual<N>& a) {
    Dual<N> r; r.v = -a.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = -a.d[i];
    return r;
}
template <int N> DEVINL Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
template <int N> DEVINL Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; const float inv = 1.0f / b.v; r.v = a.v * inv;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
template <int N, typename F> DEVINL Dual<N> chain(const Dual<N>& a, float fv, float dfv) {
    Dual<N> r; r.v = fv;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = dfv * a.d[i];
    return r;
}
#define DUAL_UNARY(name, fexpr, dexpr)                                                   \
    DEVINL float name(float x) { return fexpr; }                                         \
    template <int N> DEVINL Dual<N> name(const Dual<N>& a) {                             \
        const float x = a.v; Dual<N> r; r.v = fexpr; const float df = dexpr;             \
        _Pragma("unroll") for (int i = 0; i < N; i++) r.d[i] = df * a.d[i];             \
        return r;                                                                        \
    }
DUAL_UNARY(Sin, sinf(x), cosf(x))
DUAL_UNARY(Cos, cosf(x), -sinf(x))
DUAL_UNARY(Exp, expf(x), expf(x))
DUAL_UNARY(Log, logf(x), 1.0f / x)
DUAL_UNARY(Sqrt, sqrtf(x), 0.5f / sqrtf(x))
DEVINL float Atan2(float y, float x) { return atan2f(y, x); }
template <int N> DEVINL Dual<N> Atan2(const Dual<N>& y, const Dual<N>& x) {
    Dual<N> r; r.v = atan2f(y.v, x.v);
    const float den = x.v * x.v + y.v * y.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) / den;
    return r;
}
DEVINL float val(float x) { return x; }
template <int N> DEVINL float val(const Dual<N>& a) { return a.v; }

// ------------------------------------------------------------------------------------------------ small algebra
template <typename T> struct V3 { T x, y, z; };
template <typename T> struct Q4 { T x, y, z, w; };

template <typename T> DEVINL V3<T> cross(const V3<T>& a, const V3<T>& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename T> DEVINL V3<T> add(const V3<T>& a, const V3<T>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> DEVINL V3<T> scale(const T& s, const V3<T>& a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename T> DEVINL Q4<T> qmul(const Q4<T>& a, const Q4<T>& b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
            a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
template <typename T> DEVINL Q4<T> qconj(const Q4<T>& a) { return {-a.x, -a.y, -a.z, a.w}; }
// rotate: v' = v + 2 w (u x v) + 2 u x (u x v)
template <typename T> DEVINL V3<T> qrot(const Q4<T>& q, const V3<T>& v) {
    const V3<T> u = {q.x, q.y, q.z};
    const V3<T> uv = cross(u, v);
    const V3<T> uuv = cross(u, uv);
    const T two = T(2.0f);
    return add(v, add(scale(two * q.w, uv), scale(two, uuv)));
}

// ------------------------------------------------------------------------------------------------ exp / log
constexpr float EPS = 1e-6f;

template <typename T> DEVINL Q4<T> so3_exp(const V3<T>& phi) {
    const T th2 = phi.x * phi.x + phi.y * phi.y + phi.z * phi.z;
    T imag, real;
    if (val(th2) < EPS * EPS) {
        const T th4 = th2 * th2;
        imag = T(0.5f) - th2 * T(1.0f / 48.0f) + th4 * T(1.0f / 3840.0f);
        real = T(1.0f) - th2 * T(1.0f / 8.0f) + th4 * T(1.0f / 384.0f);
    } else {
        const T th = Sqrt(th2);
        imag = Sin(th * T(0.5f)) / th;
        real = Cos(th * T(0.5f));
    }
    return {imag * phi.x, imag * phi.y, imag * phi.z, real};
}

template <typename T> DEVINL V3<T> so3_log(const Q4<T>& q) {
    const T n2 = q.x * q.x + q.y * q.y + q.z * q.z;
    T k;
    if (val(n2) < EPS * EPS) {
        k = T(2.0f) / q.w - T(2.0f / 3.0f) * n2 / (q.w * q.w * q.w);
    } else {
        const T n = Sqrt(n2);
        // 2*atan(n/w)/n with the sign convention of a unit quaternion double cover (w may be negative)
        k = (val(q.w) < 0.f) ? (T(-2.0f) * Atan2(n, -q.w) / n) : (T(2.0f) * Atan2(n, q.w) / n);
    }
    return {k * q.x, k * q.y, k * q.z};
}

// V(phi) tau (SE3) : t = tau + a (phi x tau) + b phi x (phi x tau)
template <typename T> DEVINL void se3_coeffs(const T& th2, T& a, T& b) {
    if (val(th2) < 1e-8f) {
        a = T(0.5f) - th2 * T(1.0f / 24.0f);
        b = T(1.0f / 6.0f) - th2 * T(1.0f / 120.0f);
    } else {
        const T th = Sqrt(th2);
        a = (T(1.0f) - Cos(th)) / th2;
        b = (th - Sin(th)) / (th2 * th);
    }
}

template <typename T> DEVINL void se3_exp(const V3<T>& tau, const V3<T>& phi, V3<T>& t, Q4<T>& q) {
    q = so3_exp(phi);
    const T th2 = phi.x * phi.x + phi.y * phi.y + phi.z * phi.z;
    T a, b;
    se3_coeffs(th2, a, b);
    const V3<T> pt = cross(phi, tau);
    t = add(tau, add(scale(a, pt), scale(b, cross(phi, pt))));
}

// V^-1 t = t - 1/2 phi x t + c phi x (phi x t),  c = 1/th^2 - (1+cos th)/(2 th sin th)
template <typename T> DEVINL void se3_log(const V3<T>& t, const Q4<T>& q, V3<T>& tau, V3<T>& phi) {
    phi = so3_log(q);
    const T th2 = phi.x * phi.x + phi.y * phi.y + phi.z * phi.z;
    T c;
    if (val(th2) < 1e-6f) {
        c = T(1.0f / 12.0f) + th2 * T(1.0f / 720.0f);
    } else {
        const T th = Sqrt(th2);
        c = T(1.0f) / th2 - (T(1.0f) + Cos(th)) / (T(2.0f) * th * Sin(th));
    }
    const V3<T> pt = cross(phi, t);
    tau = add(t, add(scale(T(-0.5f), pt), scale(c, cross(phi, pt))));
}

// Sim3: W = A [phi]x + B [phi]x^2 + C I   (Sophus Sim3::exp)
template <typename T> DEVINL void sim3_coeffs(const T& th2, const T& sigma, T& A, T& B, T& C) {
    const T s = Exp(sigma);
    const float sv = val(sigma), tv = val(th2);
    if (fabsf(sv) < 1e-4f) {
        C = T(1.0f) + sigma * T(0.5f);
        if (tv < 1e-8f) { A = T(0.5f); B = T(1.0f / 6.0f); }
        else { const T th = Sqrt(th2); A = (T(1.0f) - Cos(th)) / th2; B = (th - Sin(th)) / (th2 * th); }
    } else {
        C = (s - T(1.0f)) / sigma;
        const T sg2 = sigma * sigma;
        if (tv < 1e-8f) {
            A = ((sigma - T(1.0f)) * s + T(1.0f)) / sg2;
            B = (s * T(0.5f) * sg2 + s - T(1.0f) - sigma * s) / (sg2 * sigma);
        } else {
            const T th = Sqrt(th2);
            const T a = s * Sin(th), b = s * Cos(th), c = th2 + sg2;
            A = (a * sigma + (T(1.0f) - b) * th) / (th * c);
            B = (C - ((b - T(1.0f)) * sigma + a * th) / c) / th2;
        }
    }
}

template <typename T> DEVINL void sim3_exp(const V3<T>& tau, const V3<T>& phi, const T& sigma, V3<T>& t, Q4<T>& q, T& s) {
    q = so3_exp(phi);
    s = Exp(sigma);
    const T th2 = phi.x * phi.x + phi.y * phi.y + phi.z * phi.z;
    T A, B, C;
    sim3_coeffs(th2, sigma, A, B, C);
    const V3<T> pt = cross(phi, tau);
    t = add(scale(C, tau), add(scale(A, pt), scale(B, cross(phi, pt))));
}

// solve W tau = t with W = C I + A [phi]x + B [phi]x^2 by Cramer's rule (3x3)
template <typename T> DEVINL void sim3_log(const V3<T>& t, const Q4<T>& q, const T& s, V3<T>& tau, V3<T>& phi, T& sigma) {
    phi = so3_log(q);
    sigma = Log(s);
    const T th2 = phi.x * phi.x + phi.y * phi.y + phi.z * phi.z;
    T A, B, C;
    sim3_coeffs(th2, sigma, A, B, C);
    const T px = phi.x, py = phi.y, pz = phi.z;
    // [phi]x^2 = phi phi^T - th2 I
    const T m00 = C + B * (px * px - th2), m01 = -A * pz + B * px * py, m02 = A * py + B * px * pz;
    const T m10 = A * pz + B * px * py, m11 = C + B * (py * py - th2), m12 = -A * px + B * py * pz;
    const T m20 = -A * py + B * px * pz, m21 = A * px + B * py * pz, m22 = C + B * (pz * pz - th2);
    const T c00 = m11 * m22 - m12 * m21, c01 = m12 * m20 - m10 * m22, c02 = m10 * m21 - m11 * m20;
    const T det = m00 * c00 + m01 * c01 + m02 * c02;
    const T i00 = c00 / det, i01 = (m02 * m21 - m01 * m22) / det, i02 = (m01 * m12 - m02 * m11) / det;
    const T i10 = c01 / det, i11 = (m00 * m22 - m02 * m20) / det, i12 = (m02 * m10 - m00 * m12) / det;
    const T i20 = c02 / det, i21 = (m01 * m20 - m00 * m21) / det, i22 = (m00 * m11 - m01 * m10) / det;
    tau = {i00 * t.x + i01 * t.y + i02 * t.z, i10 * t.x + i11 * t.y + i12 * t.z, i20 * t.x + i21 * t.y + i22 * t.z};
}

// ------------------------------------------------------------------------------------------------ generic element ops
// group: 0 SO3 (tangent 3, data 4), 1 SE3 (6, 7), 2 Sim3 (7, 8)
__host__ __device__ constexpr int tdim(int g) { return g == 0 ? 3 : (g == 1 ? 6 : 7); }
__host__ __device__ constexpr int ddim(int g) { return g == 0 ? 4 : (g == 1 ? 7 : 8); }

template <int G, typename T> struct Elem { V3<T> t; Q4<T> q; T s; };

template <int G, typename T> DEVINL Elem<G, T> load_elem(const T* d) {
    Elem<G, T> e;
    if (G == 0) { e.t = {T(0.f), T(0.f), T(0.f)}; e.q = {d[0], d[1], d[2], d[3]}; e.s = T(1.f); }
    else { e.t = {d[0], d[1], d[2]}; e.q = {d[3], d[4], d[5], d[6]}; e.s = (G == 2) ? d[7] : T(1.f); }
    return e;
}
template <int G, typename T> DEVINL void store_elem(const Elem<G, T>& e, T* d) {
    if (G == 0) { d[0] = e.q.x; d[1] = e.q.y; d[2] = e.q.z; d[3] = e.q.w; }
    else { d[0] = e.t.x; d[1] = e.t.y; d[2] = e.t.z; d[3] = e.q.x; d[4] = e.q.y; d[5] = e.q.z; d[6] = e.q.w; if (G == 2) d[7] = e.s; }
}

template <int G, typename T> DEVINL void f_exp(const T* a, T* out) {
    Elem<G, T> e;
    if (G == 0) { e.q = so3_exp<T>({a[0], a[1], a[2]}); e.t = {T(0.f), T(0.f), T(0.f)}; e.s = T(1.f); }
    else if (G == 1) { se3_exp<T>({a[0], a[1], a[2]}, {a[3], a[4], a[5]}, e.t, e.q); e.s = T(1.f); }
    else { sim3_exp<T>({a[0], a[1], a[2]}, {a[3], a[4], a[5]}, a[6], e.t, e.q, e.s); }
    store_elem<G, T>(e, out);
}
template <int G, typename T> DEVINL void f_log(const T* d, T* out) {
    const Elem<G, T> e = load_elem<G, T>(d);
    V3<T> tau, phi; T sigma;
    if (G == 0) { phi = so3_log(e.q); out[0] = phi.x; out[1] = phi.y; out[2] = phi.z; }
    else if (G == 1) { se3_log(e.t, e.q, tau, phi); out[0] = tau.x; out[1] = tau.y; out[2] = tau.z; out[3] = phi.x; out[4] = phi.y; out[5] = phi.z; }
    else { sim3_log(e.t, e.q, e.s, tau, phi, sigma); out[0] = tau.x; out[1] = tau.y; out[2] = tau.z; out[3] = phi.x; out[4] = phi.y; out[5] = phi.z; out[6] = sigma; }
}
template <int G, typename T> DEVINL void f_mul(const T* x, const T* y, T* out) {
    const Elem<G, T> a = load_elem<G, T>(x), b = load_elem<G, T>(y);
    Elem<G, T> r;
    r.q = qmul(a.q, b.q);
    r.t = add(a.t, scale(a.s, qrot(a.q, b.t)));
    r.s = a.s * b.s;
    store_elem<G, T>(r, out);
}
template <int G, typename T> DEVINL void f_inv(const T* x, T* out) {
    const Elem<G, T> a = load_elem<G, T>(x);
    Elem<G, T> r;
    r.q = qconj(a.q);
    r.s = T(1.f) / a.s;
    const V3<T> rt = qrot(r.q, a.t);
    r.t = scale(-r.s, rt);
    store_elem<G, T>(r, out);
}
// act on a homogeneous point [X,Y,Z,W] (lietorch act4: R p + t W, W kept) or a 3-vector (W = 1)
template <int G, typename T> DEVINL void f_act(const T* x, const T* p, int pd, T* out) {
    const Elem<G, T> a = load_elem<G, T>(x);
    const V3<T> rp = scale(a.s, qrot(a.q, V3<T>{p[0], p[1], p[2]}));
    if (pd == 4) { out[0] = rp.x + a.t.x * p[3]; out[1] = rp.y + a.t.y * p[3]; out[2] = rp.z + a.t.z * p[3]; out[3] = p[3]; }
    else { out[0] = rp.x + a.t.x; out[1] = rp.y + a.t.y; out[2] = rp.z + a.t.z; }
}
template <int G, typename T> DEVINL void f_matrix(const T* x, T* m) {
    const Elem<G, T> a = load_elem<G, T>(x);
    const V3<T> c0 = scale(a.s, qrot(a.q, V3<T>{T(1.f), T(0.f), T(0.f)}));
    const V3<T> c1 = scale(a.s, qrot(a.q, V3<T>{T(0.f), T(1.f), T(0.f)}));
    const V3<T> c2 = scale(a.s, qrot(a.q, V3<T>{T(0.f), T(0.f), T(1.f)}));
    m[0] = c0.x; m[1] = c1.x; m[2] = c2.x; m[3] = a.t.x;
    m[4] = c0.y; m[5] = c1.y; m[6] = c2.y; m[7] = a.t.y;
    m[8] = c0.z; m[9] = c1.z; m[10] = c2.z; m[11] = a.t.z;
    m[12] = T(0.f); m[13] = T(0.f); m[14] = T(0.f); m[15] = T(1.f);
}
// adjoint action on a tangent vector a: Ad_X a ; transposed: Ad_X^T a   (SE3/Sim3: [tau, phi(, sigma)])
template <int G, typename T> DEVINL void f_adj(const T* x, const T* a, int transpose, T* out) {
    const Elem<G, T> e = load_elem<G, T>(x);
    if (G == 0) {
        const V3<T> r = transpose ? qrot(qconj(e.q), V3<T>{a[0], a[1], a[2]}) : qrot(e.q, V3<T>{a[0], a[1], a[2]});
        out[0] = r.x; out[1] = r.y; out[2] = r.z;
        return;
    }
    const V3<T> tau = {a[0], a[1], a[2]}, phi = {a[3], a[4], a[5]};
    const T sig = (G == 2) ? a[6] : T(0.f);
    if (!transpose) {
        // Ad = [[sR, [t]x R, -t],[0, R, 0],[0,0,1]]
        const V3<T> Rphi = qrot(e.q, phi);
        const V3<T> o_tau = add(scale(e.s, qrot(e.q, tau)), add(cross(e.t, Rphi), scale(-sig, e.t)));
        out[0] = o_tau.x; out[1] = o_tau.y; out[2] = o_tau.z; out[3] = Rphi.x; out[4] = Rphi.y; out[5] = Rphi.z;
        if (G == 2) out[6] = sig;
    } else {
        const Q4<T> qi = qconj(e.q);
        const V3<T> o_tau = scale(e.s, qrot(qi, tau));
        const V3<T> o_phi = add(qrot(qi, phi), scale(T(-1.f), qrot(qi, cross(e.t, tau))));
        out[0] = o_tau.x; out[1] = o_tau.y; out[2] = o_tau.z; out[3] = o_phi.x; out[4] = o_phi.y; out[5] = o_phi.z;
        if (G == 2) out[6] = sig - (e.t.x * tau.x + e.t.y * tau.y + e.t.z * tau.z);
    }
}


}  // namespace liemath
