// Fused (flash-style) attention for the CUT3R ViT on gfx950: softmax(q k^T * scale) v, no mask, fp16 I/O,
// fp32 online softmax.  Replaces F.scaled_dot_product_attention at
//   /root/reference/src/croco/models/blocks.py:139-145 (encoder, 16 heads x 64)
//   /root/reference/src/dust3r/blocks.py:123-129, 233-239 (decoder self/cross: 12x64, 16x48; memory blocks 12x128)
//
// Structure (per workgroup = NW waves, one (batch, head), 32*NW query rows; KV tile = 64 keys):
//   * "swapped" product S^T = K Q^T on v_mfma_f32_32x32x16_f16, so a lane owns ONE query column and its scores for
//     16 of the tile's 32 keys sit in its own accumulator registers: the row max / row sum are 15 in-lane ops and
//     one exchange with lane^32 -- no LDS, no 32-lane shuffle reductions.
//   * the S^T accumulator tile is converted in registers (fp32 -> packed fp16) and used directly as the B operand
//     of O^T += V^T P^T (accumulator rows = MFMA k index; the k permutation inside a 16-step is folded into the
//     V^T fragment addresses), so P never touches LDS.
//   * K tile row-major in LDS (padded rows, conflict-free ds_read_b128); V tile ALSO row-major (16-byte staging
//     writes) and consumed column-wise by the hardware transposing read ds_read_b64_tr_b16 (two per k-step), rows padded
//     so that the 4-row x 16-column blocks of a 32-lane half fall on disjoint banks.
//   * register prefetch of the next K/V tile overlaps the global loads with the MFMAs of the current tile.
#include <cstdlib>
#include "common.h"
#include "../../include/cut3r_hip.h"

namespace {

typedef __fp16 fp16x4_t __attribute__((ext_vector_type(4)));

// x of this lane and of lane ^ 32, without the LDS round trip of ds_bpermute: v_permlane32_swap_b32 (gfx950) exchanges lanes 32..63
// of one register with lanes 0..31 of another in ONE vector instruction; applied to two copies of x it leaves {x[lane & 31], x[lane | 32]}
// in every lane -- the two halves' values, which is all a commutative combine (max, sum) needs.
DEVINL void halves(float x, float& lo, float& hi) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}

struct AttnArgs {
    const h16* q; const h16* k; const h16* v; h16* o;
    int Nq, Nk;
    long long q_sb, q_sn, k_sb, k_sn, v_sb, v_sn, o_sb, o_sn;
    float scale_log2;
};

// Occupancy: the natural allocation of the D <= 64 instances is 190 registers = 2 waves per SIMD, which leaves the matrix pipe
// idle whenever both waves are in their softmax; bounding them to 3 waves per SIMD (136 registers, no scratch) lets a third
// workgroup's MFMAs run under it.
//
// A key count of the form 64 m + 1 (the decoder's image tokens: 768 patches + the pose token, src/dust3r/model.py:666-667)
// would need an almost empty 13th key tile.  Instead key 0 is folded in before the loop -- it initialises the online softmax
// (running max = its score, running sum = 1, O = its value row) -- and the tiles cover keys 1..64 m.  Waves whose 32 query
// rows lie beyond Nq (the 7th query block of 769 rows has one row) take part in the staging and barriers only.
template <int D, int NW>
__global__ __launch_bounds__(NW * 64, (D <= 64 && NW == 4) ? 3 : 1) void attn_kernel(const AttnArgs a) {
    constexpr int NTHR = NW * 64;
    constexpr int KT = 64;                       // keys per tile
    constexpr int DQ = D / 16;                   // k-steps of the QK^T product
    constexpr int DP = (D + 31) / 32;            // 32-row d-tiles of the PV product
    constexpr int KS_LD = D + 8;                 // halves per K row in LDS (pad 16 B)
    constexpr int VS_BYTES = (64 * DP) % 128 == 0 ? 64 * DP + 64 : 64 * DP;   // V row stride: == 64 (mod 128) bytes
    constexpr int V_LD = VS_BYTES / 2;           // halves per V row (DP*32 data/zero columns + pad)
    constexpr int CHUNKS = KT * (D / 8);         // 16-B chunks per K (or V) tile
    constexpr int NCH = (CHUNKS + NTHR - 1) / NTHR;
    __shared__ __attribute__((aligned(16))) h16 Ks[KT * KS_LD];
    __shared__ __attribute__((aligned(16))) h16 Vs[KT * V_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * (32 * NW) + wave * 32;
    const h16* qp = a.q + (size_t)b * a.q_sb + (size_t)h * D;
    const h16* kp = a.k + (size_t)b * a.k_sb + (size_t)h * D;
    const h16* vp = a.v + (size_t)b * a.v_sb + (size_t)h * D;

    // zero the V image once (covers the d >= D padding columns read by the last d-tile)
    for (int i = tid; i < KT * V_LD; i += NTHR) Vs[i] = (h16)0;

    // Q fragments (B operand: element j = Q[q0+r][16 s + 8 hh + j])
    half8_t qf[DQ];
    {
        int qr = q0 + r;
        if (qr > a.Nq - 1) qr = a.Nq - 1;
        const h16* qrow = qp + (size_t)qr * a.q_sn;
#pragma unroll
        for (int s = 0; s < DQ; s++) qf[s] = *reinterpret_cast<const half8_t*>(qrow + 16 * s + 8 * hh);
    }

    f32x16 ot[DP];
#pragma unroll
    for (int d = 0; d < DP; d++)
#pragma unroll
        for (int i = 0; i < 16; i++) ot[d][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const bool active = q0 < a.Nq;               // wave-uniform
    const int extra = (a.Nk > KT && (a.Nk % KT) == 1) ? 1 : 0;
    if (extra) {
        // key 0 first: score = q . k0 (this lane's 8*DQ elements + the partner half's), p = 1
        float dot = 0.f;
#pragma unroll
        for (int s = 0; s < DQ; s++) {
            const half8_t k0 = *reinterpret_cast<const half8_t*>(kp + 16 * s + 8 * hh);
#pragma unroll
            for (int j = 0; j < 8; j++) dot = fmaf((float)qf[s][j], (float)k0[j], dot);
        }
        { float d0, d1; halves(dot, d0, d1); dot = d0 + d1; }
        m_run = dot * a.scale_log2;
        l_run = hh == 0 ? 1.f : 0.f;             // the two halves' sums are added at the end
#pragma unroll
        for (int d = 0; d < DP; d++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const int dd = d * 32 + 8 * g4 + 4 * hh;
                if (dd < D) {
                    const half4_t v0 = *reinterpret_cast<const half4_t*>(vp + dd);
#pragma unroll
                    for (int e = 0; e < 4; e++) ot[d][4 * g4 + e] = (float)v0[e];
                }
            }
    }

    half8_t rk[NCH], rv[NCH];
    // this thread's chunks of a K / V tile: the pointers are computed once and advanced by one tile per call (s_memtime: the
    // per-tile 64-bit address arithmetic, bounds tests and zero fills of four loads cost 470 of a tile's 2750 cycles); only a
    // ragged LAST tile takes the bounds-tested form
    const h16* kptr[NCH];
    const h16* vptr[NCH];
    int kkey[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int id = tid + c * NTHR;
        const int key = id / (D / 8), ch = id - key * (D / 8);
        kkey[c] = key;
        kptr[c] = kp + (size_t)(extra + key) * a.k_sn + ch * 8;
        vptr[c] = vp + (size_t)(extra + key) * a.v_sn + ch * 8;
    }
    auto load_kv = [&](int t) {
        const int kbase = extra + t * KT;
        const bool full = kbase + KT <= a.Nk;          // wave-uniform
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            half8_t zk = {0, 0, 0, 0, 0, 0, 0, 0}, zv = zk;
            if (NCH * NTHR == CHUNKS || tid + c * NTHR < CHUNKS) {
                if (full || kbase + kkey[c] < a.Nk) {
                    zk = *reinterpret_cast<const half8_t*>(kptr[c]);
                    zv = *reinterpret_cast<const half8_t*>(vptr[c]);
                }
            }
            rk[c] = zk; rv[c] = zv;
            kptr[c] += (size_t)KT * a.k_sn;
            vptr[c] += (size_t)KT * a.v_sn;
        }
    };
    auto store_kv = [&]() {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            int id = tid + c * NTHR;
            if (NCH * NTHR == CHUNKS || id < CHUNKS) {
                int key = id / (D / 8), ch = id - key * (D / 8);
                *reinterpret_cast<half8_t*>(&Ks[key * KS_LD + ch * 8]) = rk[c];
                *reinterpret_cast<half8_t*>(&Vs[key * V_LD + ch * 8]) = rv[c];
            }
        }
    };

    const int ntiles = (a.Nk - extra + KT - 1) / KT;
    load_kv(0);
    for (int t = 0; t < ntiles; t++) {
        __syncthreads();
        store_kv();
        // every prefetched register has been consumed: say so on ALL paths (the guarded staging writes are branches whose
        // skipped side keeps the loads "pending" for the compiler's wait-count pass, which then makes the first MFMAs of the
        // tile wait for the NEXT tile's loads -- the prefetch would hide nothing)
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
        __syncthreads();
        if (t + 1 < ntiles) load_kv(t + 1);
        if (!active) continue;

        // ---- S^T = K Q^T for the two 32-key sub-tiles
        f32x16 st[2];
        half8_t kf[2][DQ];                        // every K fragment of the tile is requested before the first MFMA
#pragma unroll
        for (int kt2 = 0; kt2 < 2; kt2++)
#pragma unroll
            for (int s = 0; s < DQ; s++) kf[kt2][s] = *reinterpret_cast<const half8_t*>(&Ks[(kt2 * 32 + r) * KS_LD + 16 * s + 8 * hh]);
        __builtin_amdgcn_sched_barrier(0);        // (the scheduler would sink each read to its MFMA: read, wait, MFMA, eight times)
#pragma unroll
        for (int kt2 = 0; kt2 < 2; kt2++) {
#pragma unroll
            for (int i = 0; i < 16; i++) st[kt2][i] = 0.f;
#pragma unroll
            for (int s = 0; s < DQ; s++) st[kt2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kt2][s], qf[s], st[kt2], 0, 0, 0);
        }
        // ---- online softmax (per query column == per lane; partner lane^32 holds the other 32 keys)
        const int kbase = extra + t * KT;
        float mloc = -INFINITY;
        if (kbase + KT <= a.Nk) {                 // full tile: no masking
#pragma unroll
            for (int kt2 = 0; kt2 < 2; kt2++)
#pragma unroll
                for (int i = 0; i < 16; i++) mloc = fmaxf(mloc, st[kt2][i]);
        } else {
#pragma unroll
            for (int kt2 = 0; kt2 < 2; kt2++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int key = kbase + kt2 * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    const float sv = key < a.Nk ? st[kt2][i] : -INFINITY;
                    st[kt2][i] = sv;
                    mloc = fmaxf(mloc, sv);
                }
        }
        { float m0_, m1_; halves(mloc, m0_, m1_); mloc = fmaxf(m0_, m1_) * a.scale_log2; }      // scale > 0: max commutes with the scaling
        const float m_new = fmaxf(m_run, mloc);
        const bool grew = m_new > m_run;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float lsum = 0.f;
#pragma unroll
        for (int kt2 = 0; kt2 < 2; kt2++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float p = __builtin_amdgcn_exp2f(fmaf(st[kt2][i], a.scale_log2, -m_new));
                st[kt2][i] = p;
                lsum += p;
            }
        l_run = l_run * alpha + lsum;
        if (__any(grew)) {                        // wave-uniform: after the first tiles the running max rarely moves
#pragma unroll
            for (int d = 0; d < DP; d++)
#pragma unroll
                for (int i = 0; i < 16; i++) ot[d][i] *= alpha;
        }

        // ---- O^T += V^T P^T   (B fragment of k-step s2 = accumulator registers 8 s2 .. 8 s2 + 7; A fragment = V^T via
        //      two transposing reads: 16-lane group g = lane>>4 reads the 4-key x 16-d block (keys k0..k0+3, d-columns
        //      c0..c0+15) with lane j of the group addressing row j>>2, columns 4(j&3).., and receives column j)
        const int g16 = lane >> 4, j16 = lane & 15;
#pragma unroll
        for (int kt2 = 0; kt2 < 2; kt2++)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                half8_t pf;
#pragma unroll
                for (int j = 0; j < 8; j++) pf[j] = (h16)st[kt2][8 * s2 + j];
#pragma unroll
                for (int d = 0; d < DP; d++) {
                    const int k0 = kt2 * 32 + 16 * s2 + 4 * (g16 >> 1);
                    const int c0 = d * 32 + 16 * (g16 & 1);
                    const h16* vp = &Vs[(k0 + (j16 >> 2)) * V_LD + c0 + 4 * (j16 & 3)];
                    const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(vp));
                    const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(vp + 8 * V_LD));
                    half8_t vf;
                    __builtin_memcpy(&vf, &lo, 8);
                    __builtin_memcpy(reinterpret_cast<char*>(&vf) + 8, &hi, 8);
                    ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, ot[d], 0, 0, 0);
                }
            }
    }

    float l0_, l1_;
    halves(l_run, l0_, l1_);
    const float l_tot = l0_ + l1_;
    const float inv = 1.0f / l_tot;
    const int qr = q0 + r;
    if (qr < a.Nq) {
        h16* orow = a.o + (size_t)b * a.o_sb + (size_t)qr * a.o_sn + (size_t)h * D;
#pragma unroll
        for (int d = 0; d < DP; d++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                int dd = d * 32 + 8 * g4 + 4 * hh;
                if (dd < D) {
                    half4_t o = {(h16)(ot[d][4 * g4 + 0] * inv), (h16)(ot[d][4 * g4 + 1] * inv), (h16)(ot[d][4 * g4 + 2] * inv),
                                 (h16)(ot[d][4 * g4 + 3] * inv)};
                    *reinterpret_cast<half4_t*>(orow + dd) = o;
                }
            }
    }
}

template <int D>
int launch_attn(const AttnArgs& a, int B, int H, hipStream_t s) {
    const long long blocks128 = (long long)B * H * ((a.Nq + 127) / 128);
    static const long long nw4_min = [] { const char* e = getenv("CUT3R_ATTN_NW4_MIN"); return e ? atoll(e) : 384LL; }();
    if (blocks128 >= nw4_min || D >= 128) {
        dim3 grid((a.Nq + 127) / 128, H, B);
        hipLaunchKernelGGL((attn_kernel<D, 4>), grid, dim3(256), 0, s, a);
    } else {
        dim3 grid((a.Nq + 63) / 64, H, B);
        hipLaunchKernelGGL((attn_kernel<D, 2>), grid, dim3(128), 0, s, a);
    }
    return cut3r_check_launch();
}

}  // namespace

extern "C" int cut3r_attention_f16(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk, int D,
                                   long long q_sb, long long q_sn, long long k_sb, long long k_sn, long long v_sb, long long v_sn,
                                   long long o_sb, long long o_sn, float scale, void* stream) {
    if (!q || !k || !v || !out || B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return CUT3R_ERR_ARG;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) return CUT3R_ERR_ARG;
    if ((uintptr_t)out & 7) return CUT3R_ERR_ARG;
    if ((q_sn | k_sn | v_sn | q_sb | k_sb | v_sb) & 7) return CUT3R_ERR_ARG;   // 16-B vector loads
    if ((o_sn | o_sb) & 3) return CUT3R_ERR_ARG;
    AttnArgs a;
    a.q = (const h16*)q; a.k = (const h16*)k; a.v = (const h16*)v; a.o = (h16*)out;
    a.Nq = Nq; a.Nk = Nk;
    a.q_sb = q_sb; a.q_sn = q_sn; a.k_sb = k_sb; a.k_sn = k_sn; a.v_sb = v_sb; a.v_sn = v_sn; a.o_sb = o_sb; a.o_sn = o_sn;
    a.scale_log2 = scale * 1.44269504088896340736f;
    hipStream_t s = (hipStream_t)stream;
    switch (D) {
        case 16: return launch_attn<16>(a, B, H, s);
        case 32: return launch_attn<32>(a, B, H, s);
        case 48: return launch_attn<48>(a, B, H, s);
        case 64: return launch_attn<64>(a, B, H, s);
        case 128: return launch_attn<128>(a, B, H, s);
        default: return CUT3R_ERR_ARG;
    }
}
