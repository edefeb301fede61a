// MFMA GEMM family for the CUT3R ViT / DPT stack on gfx950.
//
//   C[M,N] = epilogue( A[M,K] (fp16, K-major) x B[N,K]^T (fp16, K-major, = nn.Linear weight layout) ), fp32 accumulate.
//
// One kernel template covers every dense contraction on the hot path:
//   * nn.Linear (+bias)(+exact GELU)(+residual) of the encoder/decoder blocks
//       /root/reference/src/croco/models/blocks.py:68-148, src/dust3r/blocks.py:87-243
//   * 1x1 / 3x3 (stride 1|2, pad 1) convolutions of the DPT head as implicit GEMM over NHWC fp16 activations
//       /root/reference/src/croco/models/dpt_block.py:84-232, 281-513 (optional ReLU-on-load = ResidualConvUnit pre-activation)
//   * ConvTranspose2d with kernel == stride as GEMM + pixel-shuffle scatter epilogue (dpt_block.py:416-446)
//
// Design (MI355X): v_mfma_f32_16x16x32_f16, 256-thread workgroups (4 waves, 2x2), BK = 64, LDS-DMA staging
// (global_load_lds_dwordx4) into an N-stage LDS ring with counted s_waitcnt vmcnt and one raw s_barrier per K-tile,
// XOR-swizzled 16-B LDS chunks (swizzle on the DMA source address + on the ds_read_b128 side), epilogue staged through
// LDS so bias / residual / output traffic is 16-B coalesced.  Tile 128x128 with 8 waves (4x2; two waves per SIMD hide
// each other's LDS-DMA issue: +25-30 % over 4 waves) for large grids, 64x64 when the grid would not fill 256 CUs
// (batch-1 decoder GEMMs; 8 waves when K >= 2048).  blockIdx.z batches independent problems with element strides.
#include <cstdlib>
#include "common.h"
#include "../../include/cut3r_hip.h"

namespace {

constexpr int BK = 64;
constexpr int KCH = BK / 8;  // 16-byte chunks per tile row

struct GemmArgs {
    const h16* A; const h16* B; void* C;
    const float* bias; const void* res1; const void* res2;
    int M, N, K, lda, ldb, ldc, ldr1, ldr2;
    int act, out_f16, res1_f16, res2_f16;
    long long sA, sB, sC, sBias, sR1, sR2;       // batch strides (elements), blockIdx.z
    // implicit-GEMM convolution (A = NHWC fp16 [Bimg,H,W,Cin]); conv_k = 0 (plain), 1 or 3
    int conv_k, H, W, Cin, cstride, Ho, Wo, relu_in;
    // pixel-shuffle scatter epilogue (ConvTranspose k == stride == shuf): n = (i*shuf + j)*Cout + co
    int shuf, shuf_cout, shuf_Hin, shuf_Win;
    int swz;          // 1: 1-D grid with the XCD-aware tile rasterisation
    int prio;         // 1: s_setprio(1) around the MFMA cluster
    // fused 2-D RoPE (head dimension rope_d = 64 or 48) on output columns < rope_cols
    const long long* rope_pos; const float* rope_table; int rope_cols, rope_pmin, rope_npos, rope_d;
    // LayerNorm folded into the GEMMs (see "LayerNorm fold" below): consumer side (ln_stats != null) and producer side (stats_out != null)
    const float* ln_stats; const float* ln_c; int ln_nslab; float ln_eps;
    float* stats_out; h16* out16; int ld16;
};

DEVINL half8_t relu8(half8_t v) {
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = v[i] > (h16)0 ? v[i] : (h16)0;
    return v;
}

// 16 bytes of zeros in HBM: the source of every out-of-range / padding chunk of the LDS-DMA loader
__device__ __attribute__((aligned(16))) unsigned char g_zero16[16];

template <int N>
DEVINL void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until all but the youngest `fl` K-tiles (NL LDS-DMA instructions each) have landed; fl <= MAXF (the ring depth minus two)
template <int NL, int MAXF>
DEVINL void wait_tiles_in_flight(int fl) {
    static_assert(MAXF * NL <= 63, "vmcnt is a 6-bit counter");
    if constexpr (MAXF >= 6) { if (fl >= 6) { wait_vmcnt<6 * NL>(); return; } }
    if constexpr (MAXF >= 5) { if (fl == 5) { wait_vmcnt<5 * NL>(); return; } }
    if constexpr (MAXF >= 4) { if (fl == 4) { wait_vmcnt<4 * NL>(); return; } }
    if constexpr (MAXF >= 3) { if (fl == 3) { wait_vmcnt<3 * NL>(); return; } }
    if constexpr (MAXF >= 2) { if (fl == 2) { wait_vmcnt<2 * NL>(); return; } }
    if constexpr (MAXF >= 1) { if (fl == 1) { wait_vmcnt<NL>(); return; } }
    wait_vmcnt<0>();
}

// Fused output of 4 consecutive columns (gn..gn+3) of row gm: bias, GELU|ReLU, up to two residuals, fp16|fp32 store,
// optional ConvTranspose pixel-shuffle scatter.  Shared by every GEMM kernel of this file.  The residual operands are
// LOADED by load_res4 (so an epilogue can issue the loads of several rows before it consumes any: a dependent global load
// per row was the largest single cost of the short-K launches) and applied by fused_finish4 in a fixed order: bias,
// activation, res1, res2.
DEVINL f32x4 load_res4(const void* res, int is_f16, size_t off) {
    if (is_f16) {
        const half4_t rv = *reinterpret_cast<const half4_t*>((const h16*)res + off);
        return f32x4{(float)rv[0], (float)rv[1], (float)rv[2], (float)rv[3]};
    }
    return *reinterpret_cast<const f32x4*>((const float*)res + off);
}
DEVINL f32x4 fused_finish4(const GemmArgs& g, f32x4 v, const f32x4& b4, bool has_bias, const f32x4& r1, const f32x4& r2) {
    if (has_bias) v += b4;
    if (g.act == 1) {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = g.out_f16 ? gelu_fast(v[e]) : gelu_erf(v[e]);
    } else if (g.act == 2) {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
    }
    if (g.res1) v += r1;
    if (g.res2) v += r2;
    return v;
}
DEVINL size_t out_offset4(const GemmArgs& g, int gm, int gn, int& bcol) {
    bcol = gn;
    if (g.shuf) {
        const int ij = gn / g.shuf_cout;
        bcol = gn - ij * g.shuf_cout;
        const int i_ = ij / g.shuf, j_ = ij - i_ * g.shuf;
        const int hw = g.shuf_Hin * g.shuf_Win;
        const int b = gm / hw, rem = gm - b * hw;
        const int y = rem / g.shuf_Win, x = rem - y * g.shuf_Win;
        return (((size_t)b * g.shuf_Hin * g.shuf + (y * g.shuf + i_)) * (g.shuf_Win * g.shuf) + (x * g.shuf + j_)) * (size_t)g.ldc + bcol;
    }
    return (size_t)gm * g.ldc + gn;
}
DEVINL void store_out4(const GemmArgs& g, int z, size_t orow_off, f32x4 v) {
    if (g.out_f16) {
        half4_t o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
        *reinterpret_cast<half4_t*>((h16*)g.C + (size_t)z * g.sC + orow_off) = o;
    } else {
        *reinterpret_cast<f32x4*>((float*)g.C + (size_t)z * g.sC + orow_off) = v;
    }
}
DEVINL void fused_store4(const GemmArgs& g, int z, const float* bias, int gm, int gn, f32x4 v) {
    int bcol;
    const size_t orow_off = out_offset4(g, gm, gn, bcol);
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f}, r1 = b4, r2 = b4;
    if (bias) b4 = *reinterpret_cast<const f32x4*>(bias + bcol);
    if (g.res1) r1 = load_res4(g.res1, g.res1_f16, (size_t)z * g.sR1 + (size_t)gm * g.ldr1 + gn);
    if (g.res2) r2 = load_res4(g.res2, g.res2_f16, (size_t)z * g.sR2 + (size_t)gm * g.ldr2 + gn);
    store_out4(g, z, orow_off, fused_finish4(g, v, b4, bias != nullptr, r1, r2));
}

// ---- LayerNorm fold (round 4).  y = LN(x) W^T + b  =  rstd (x (gamma . W)^T - mu c) + d   with  c_n = sum_k gamma_k W_nk,
// d = W beta + b  (croco/models/blocks.py:187-190: x = x + attn(norm1(x)); x = x + mlp(norm2(x)); dust3r/blocks.py:292-297).
// PRODUCER (the fp32 + residual GEMM that writes the residual stream x): besides x it stores an fp16 copy of x (the consumer's A
// operand) and, per row and 64-column slab, (sum, m2) with m2 = sum (x - sum/64)^2 -- in a FIXED order (in-lane pairs, then a
// 16-lane butterfly: quad xor 1, xor 2, half mirror, row mirror), the same in every tile kernel, so the statistics of a row are
// the same bits whatever tile or batch computed it.  CONSUMER (qkv / projq / projkv / fc1 with gamma folded into the fp16 weight
// panel, d passed as the bias): combines a row's slabs in slab order (Chan's parallel variance: no E[x^2] - mu^2 cancellation),
// rstd = 1 / sqrt(m2 / K + eps), and applies  acc * rstd - (rstd mu) c_n  to the raw accumulator before the usual epilogue.
template <int CTRL>
DEVINL float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
DEVINL float row16_sum(float v) {       // total over the 16 lanes of a DPP row, in every lane; all 64 lanes must be active
    v += dpp_mov<0xB1>(v);               // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);               // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);              // row_half_mirror
    v += dpp_mov<0x140>(v);              // row_mirror
    return v;
}
// (sum, m2) of one row's 64-column slab; `v` = this lane's 4 consecutive columns, lane & 15 = column group.  Every lane gets both.
DEVINL void slab_stats4(const f32x4& v, float& sum, float& m2) {
    sum = row16_sum((v[0] + v[1]) + (v[2] + v[3]));
    const float mean = sum * (1.0f / 64.0f);
    const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
    m2 = row16_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
}
// the same statistics from EIGHT consecutive columns per lane (lane & 7 = column group of 8: the 16-byte-store layout of the 256 x 256
// producer epilogue).  Bit-identical to slab_stats4: the in-lane pair sums are step 1 of the 16-group butterfly (a + b == b + a), the
// three DPP steps pair the same partial sums as its steps 2-4.
DEVINL float row8_sum(float v) {
    v += dpp_mov<0xB1>(v);               // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);               // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);              // row_half_mirror
    return v;
}
DEVINL void slab_stats8(const f32x4& a, const f32x4& b, float& sum, float& m2) {
    sum = row8_sum(((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3])));
    const float mean = sum * (1.0f / 64.0f);
    const float d0 = a[0] - mean, d1 = a[1] - mean, d2 = a[2] - mean, d3 = a[3] - mean;
    const float e0 = b[0] - mean, e1 = b[1] - mean, e2 = b[2] - mean, e3 = b[3] - mean;
    m2 = row8_sum(((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3)));
}
// row parameters (rstd, rstd * mu) from the row's slab statistics (8-B aligned), combined in slab order, in two steps so
// that a kernel can REQUEST the statistics before it issues its DMA prologue and USE them after it: vmcnt retires in order, so a load
// issued behind the prologue's LDS-DMA could only be waited for by draining the whole prologue.  All of a row's slabs are requested
// together (clamped index: every load is unconditional, so the compiler keeps them in flight; a loop over the run-time slab count had
// ONE 8-byte load outstanding at a time: 12-16 L2 round trips per tile, measured as -10 % end to end); slabs beyond nslab contribute an
// exact + 0.0f.
constexpr int LN_MAXS = 16;                        // K <= 1024 (the widths the network normalises: 768, 1024 and the test configs)
struct LnRow { float2 t[LN_MAXS]; };
// (statistics are stored SLAB-MAJOR, [nslab][M][2]: consecutive threads own consecutive rows, so each of these loads is one coalesced
//  512-byte access per wave; row-major, a wave's load touched 64 different lines)
DEVINL void ln_row_load(const float* __restrict__ st, int row, int M, int nslab, LnRow& r) {
#pragma unroll
    for (int i = 0; i < LN_MAXS; i++) r.t[i] = reinterpret_cast<const float2*>(st)[(size_t)min(i, nslab - 1) * M + row];
}
DEVINL void ln_row_finish(const LnRow& r, int nslab, int K, float eps, float& rs, float& rm) {
    float S = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXS; i++) S += (i < nslab) ? r.t[i].x : 0.f;
    const float mu = S / (float)K;
    float M2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXS; i++) {
        const float e = r.t[i].x * (1.0f / 64.0f) - mu;
        M2 += (i < nslab) ? r.t[i].y + 64.0f * (e * e) : 0.f;
    }
    rs = 1.0f / sqrtf(M2 / (float)K + eps);
    rm = rs * mu;
}
DEVINL float ln_pre(float acc, float rs, float rm, float c) { return fmaf(acc, rs, -(rm * c)); }
DEVINL f32x4 ln_pre4(const f32x4& a, float rs, float rm, const f32x4& c) {
    return f32x4{ln_pre(a[0], rs, rm, c[0]), ln_pre(a[1], rs, rm, c[1]), ln_pre(a[2], rs, rm, c[2]), ln_pre(a[3], rs, rm, c[3])};
}

// Fused 2-D RoPE of 4 consecutive output columns gn..gn+3 (< rope_cols) of row gm, head dimension D = 64 or 48 (quarter
// Q = D/4): a head is [y-block | x-block] of D/2 columns, inside a block column j < Q pairs with column j + Q.  `cs_row` is
// this row of the epilogue's LDS staging (raw accumulators), c4 the tile-local column of gn; the partner values come from
// c4 +- Q (tiles start on head boundaries).  Same arithmetic as rope2d_kernel on the fp16-rounded projection.
DEVINL void rope_store4(const GemmArgs& g, int z, const float* bias, int gm, int gn, const float* cs_row, int c4, float ln_rs = 1.f, float ln_rm = 0.f) {
    const int D = g.rope_d, Q = D >> 2, half = D >> 1;
    const int hl = gn % D, X = hl / half, within = hl - X * half;
    const bool lower = within < Q;
    const int j = lower ? within : within - Q;
    const int po = lower ? Q : -Q;
    f32x4 own = *reinterpret_cast<const f32x4*>(cs_row + c4);
    f32x4 partner = *reinterpret_cast<const f32x4*>(cs_row + c4 + po);
    if (g.ln_stats) {              // LayerNorm fold: the raw accumulators become the projection of the normalised row first
        own = ln_pre4(own, ln_rs, ln_rm, *reinterpret_cast<const f32x4*>(g.ln_c + gn));
        partner = ln_pre4(partner, ln_rs, ln_rm, *reinterpret_cast<const f32x4*>(g.ln_c + gn + po));
    }
    if (bias) {
        own += *reinterpret_cast<const f32x4*>(bias + gn);
        partner += *reinterpret_cast<const f32x4*>(bias + gn + po);
    }
    long long pv = g.rope_pos[(size_t)gm * 2 + X] - g.rope_pmin;
    pv = pv < 0 ? 0 : (pv >= g.rope_npos ? g.rope_npos - 1 : pv);
    const float* ct = g.rope_table + (size_t)pv * Q + j;
    const float* st = ct + (size_t)g.rope_npos * Q;
    half4_t o;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const float a = (float)(h16)own[e], b = (float)(h16)partner[e];
        float t = b * st[e];
        asm volatile("" : "+v"(t));                 // a separate multiply, as in rope2d_kernel (no re-contraction)
        o[e] = (h16)rope_rot(a, ct[e], lower ? -t : t);
    }
    *reinterpret_cast<half4_t*>((h16*)g.C + (size_t)z * g.sC + (size_t)gm * g.ldc + gn) = o;
}

// XCD-aware rasterisation of a 1-D grid: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
// CONTIGUOUS range of tiles (bijective remap), then walk that range in bands of 8 row-tiles so neighbouring
// workgroups of one XCD share A row-panels and W column-panels in that XCD's L2.
DEVINL void xcd_tile(int orig, int npm, int npn, int& pid_m, int& pid_n) {
    const int nwg = npm * npn;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    constexpr int GROUP = 8;
    const int in_group = GROUP * npn;
    const int group_id = wgid / in_group;
    const int first_m = group_id * GROUP;
    const int gsz = min(npm - first_m, GROUP);
    const int rem = wgid - group_id * in_group;
    pid_m = first_m + rem % gsz;
    pid_n = rem / gsz;
}

// Loader: global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write).  One wave-instruction
// writes 1 KiB = 8 tile rows x 128 B linearly; the XOR swizzle that makes the ds_read_b128 fragment reads
// conflict-free is applied to the per-lane SOURCE chunk (lane l sits at row l>>3, chunk l&7 of its 8-row group and
// fetches chunk (l&7)^(l>>3)), and again on the read side.  NSTAGE-deep ring, counted vmcnt, ONE raw s_barrier per
// K-tile: NSTAGE-2 tiles stay in flight across the barrier.
// ADDR: how a DMA piece gets its source address.  0: generic (any K, any convolution; per-lane tap decoding and bounds tests per
// piece).  1: plain operands with whole K-tiles -- a per-lane pointer computed once plus the scalar K offset.  2: 3x3 convolution
// with Cin a power of two >= 64 -- the tap of a K-tile is wave-uniform (shift), a per-lane pointer to the pixel's top-left
// neighbour plus a scalar offset, the nine padding tests folded into a per-lane bit mask.  (s_memtime on the 256x256 kernel: the
// load segments set the pace of a K-tile, and the vector address arithmetic was most of them.)
template <int BM, int BN, int NSTAGE, int WAVES_M, int WAVES_N, int ADDR = 0, int EPI = 0>
DEVINL void gemm_tile_body(const GemmArgs& g, const int bx, const int by, const int bz) {
    constexpr int NWAVE = WAVES_M * WAVES_N;
    constexpr int NTHR = 64 * NWAVE;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;   // wave tile
    constexpr int MT = WM / 16, NT = WN / 16;     // 16x16 MFMA tiles per wave
    constexpr int A_CH = BM / (8 * NWAVE);        // LDS-DMA instructions per wave per stage (A)
    constexpr int B_CH = BN / (8 * NWAVE);
    static_assert(A_CH >= 1 && B_CH >= 1 && A_CH * 8 * NWAVE == BM && B_CH * 8 * NWAVE == BN, "tile / wave mismatch");
    constexpr int NL = A_CH + B_CH;
    constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
    constexpr int CPAD = BN + 4;
    // the epilogue stages the tile through LDS in EPI_PASSES row bands so that it never needs more LDS than the ring
    constexpr int EPI_PASSES = (BM * CPAD * 4 > NSTAGE * STAGE_BYTES && (BM % 32) == 0) ? 2 : 1;
    constexpr int EPI_ROWS = BM / EPI_PASSES;
    constexpr int EPI_BYTES = EPI_ROWS * CPAD * 4;
    constexpr int LDS_BYTES = (NSTAGE * STAGE_BYTES > EPI_BYTES) ? NSTAGE * STAGE_BYTES : EPI_BYTES;
    // LayerNorm fold, consumer side: (rstd, rstd * mu) of the tile's BM rows live behind the epilogue's staging band (the ring is dead by then)
    constexpr int LNP_OFF = LDS_BYTES - BM * 8;
    static_assert(EPI_BYTES <= LNP_OFF, "no room for the LayerNorm row parameters behind the staging band");
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int z = bz;
    const h16* __restrict__ A = g.A + (size_t)z * g.sA;
    const h16* __restrict__ Bm = g.B + (size_t)z * g.sB;
    const int M = g.M, N = g.N, K = g.K;
    int pid_m, pid_n;
    if (g.swz) {
        xcd_tile(bx, (M + BM - 1) / BM, (N + BN - 1) / BN, pid_m, pid_n);
    } else {
        pid_m = by;
        pid_n = bx;
    }
    const int m0 = pid_m * BM, n0 = pid_n * BN;
    const h16* zero = reinterpret_cast<const h16*>(g_zero16);

    // ---- per-thread loader coordinates (tile-invariant part)
    const int lrow = lane >> 3;                    // row inside the 8-row group
    const int csrc = (lane & 7) ^ lrow;            // swizzled source chunk
    bool a_ok[A_CH];
    const h16* a_base[A_CH];
    int a_oy[A_CH], a_ox[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; i++) {
        const int r = (wave + NWAVE * i) * 8 + lrow;
        const int gm = m0 + r;
        a_ok[i] = gm < M;
        if (g.conv_k == 3) {
            const int gmc = a_ok[i] ? gm : 0;
            const int hw = g.Ho * g.Wo;
            const int b = gmc / hw, rem = gmc - b * hw;
            const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
            a_oy[i] = oy * g.cstride - 1;
            a_ox[i] = ox * g.cstride - 1;
            a_base[i] = A + (size_t)b * g.H * g.W * g.Cin;
        } else {
            a_oy[i] = a_ox[i] = 0;
            a_base[i] = A + (size_t)(a_ok[i] ? gm : 0) * g.lda;
        }
    }
    const h16* b_base[B_CH]; bool b_ok[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; i++) {
        const int gn = n0 + (wave + NWAVE * i) * 8 + lrow;
        b_ok[i] = gn < N;
        b_base[i] = Bm + (size_t)(b_ok[i] ? gn : 0) * g.ldb;
    }

    // ---- ADDR 1 / 2: per-lane source pointers (and the padding mask of the nine taps), computed once
    const h16* a_p[A_CH];
    const h16* b_p[B_CH];
    unsigned a_mask[A_CH];
    if (ADDR != 0) {
#pragma unroll
        for (int i = 0; i < A_CH; i++) {
            a_mask[i] = 0;
            if (ADDR == 2) {
                a_p[i] = a_base[i] + ((ptrdiff_t)a_oy[i] * g.W + a_ox[i]) * g.Cin + csrc * 8;      // (may lie outside the image: masked)
#pragma unroll
                for (int t = 0; t < 9; t++) {
                    const int iy = a_oy[i] + t / 3, ix = a_ox[i] + t % 3;
                    if (a_ok[i] && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) a_mask[i] |= 1u << t;
                }
            } else {
                a_p[i] = A + (size_t)min(m0 + (wave + NWAVE * i) * 8 + lrow, M - 1) * g.lda + csrc * 8;   // rows >= M: never stored
            }
        }
#pragma unroll
        for (int i = 0; i < B_CH; i++) b_p[i] = Bm + (size_t)min(n0 + (wave + NWAVE * i) * 8 + lrow, N - 1) * g.ldb + csrc * 8;
    }
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int cin_shift = 31 - __builtin_clz((unsigned)(g.Cin > 0 ? g.Cin : 1));

    auto issue_tile = [&](int kt, int stage) {
        if constexpr (ADDR != 0) {
            unsigned char* sa = smem + stage * STAGE_BYTES;
            unsigned char* sb = sa + BM * BK * 2;
            const int k0 = kt * BK;                                   // wave-uniform
            int tap = 0, delta = k0;
            if (ADDR == 2) {
                tap = k0 >> cin_shift;
                const int dy = (tap >= 3) + (tap >= 6), dx = tap - 3 * dy;
                delta = (dy * g.W + dx) * g.Cin + (k0 & (g.Cin - 1));
            }
#pragma unroll
            for (int i = 0; i < A_CH; i++) {
                const h16* src = a_p[i] + delta;
                if (ADDR == 2 && !((a_mask[i] >> tap) & 1)) src = zero;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(sa + (wave_s + NWAVE * i) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < B_CH; i++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_p[i] + k0),
                                                 (__attribute__((address_space(3))) void*)(sb + (wave_s + NWAVE * i) * 1024), 16, 0, 0);
            return;
        }
        const int k = kt * BK + csrc * 8;
        const bool kok = k < K;
        unsigned char* sa = smem + stage * STAGE_BYTES;
        unsigned char* sb = sa + BM * BK * 2;
        int tap = 0, ci = 0, dy = 0, dx = 0;
        if (g.conv_k == 3) {
            tap = k / g.Cin; ci = k - tap * g.Cin;
            dy = tap / 3; dx = tap - dy * 3;
        }
#pragma unroll
        for (int i = 0; i < A_CH; i++) {
            const h16* src = zero;
            if (g.conv_k == 3) {
                const int iy = a_oy[i] + dy, ix = a_ox[i] + dx;
                if (kok && a_ok[i] && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                    src = a_base[i] + ((size_t)iy * g.W + ix) * g.Cin + ci;
            } else if (kok && a_ok[i]) {
                src = a_base[i] + k;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sa + (wave + NWAVE * i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_CH; i++) {
            const h16* src = (kok && b_ok[i]) ? b_base[i] + k : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sb + (wave + NWAVE * i) * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = (K + BK - 1) / BK;
    // LayerNorm fold, consumer side: thread t < BM requests row t's slab statistics BEFORE the DMA prologue (vmcnt retires in order) and
    // turns them into (rstd, rstd * mu) while the first K-tiles are in flight; the pair waits in two registers until the ring is dead
    // and then moves behind the staging band
    float ln_rs = 1.f, ln_rm = 0.f;
    LnRow ln_raw;
    const bool ln_mine = g.ln_stats && tid < BM;
    if (ln_mine) ln_row_load(g.ln_stats, min(m0 + tid, M - 1), M, g.ln_nslab, ln_raw);
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; s++)
        if (s < nt) issue_tile(s, s);
    if (ln_mine) ln_row_finish(ln_raw, g.ln_nslab, K, g.ln_eps, ln_rs, ln_rm);
    asm volatile("" : "+v"(ln_rs), "+v"(ln_rm));          // computed HERE, in the prologue's shadow (not sunk behind the main loop)

    const int fr = lane & 15, fq = lane >> 4;   // fragment row, k-chunk
    int stage = 0;
    for (int t = 0; t < nt; t++) {
        // tiles that may stay in flight behind tile t: min(NSTAGE-2, nt-1-t) -- the whole ring depth is used (round 4: at one window the
        // 64 x 64 launches are bound by the HBM latency of their weight panels with only two K-tiles in flight)
        const int ahead = nt - 1 - t;
        constexpr int MAXF = NSTAGE - 2 > 6 ? 6 : NSTAGE - 2;
        wait_tiles_in_flight<NL, MAXF>(ahead < MAXF ? ahead : MAXF);
        __builtin_amdgcn_s_barrier();
        if (t + NSTAGE - 1 < nt) {
            int st2 = stage + NSTAGE - 1;
            if (st2 >= NSTAGE) st2 -= NSTAGE;
            issue_tile(t + NSTAGE - 1, st2);
        }
        const unsigned char* sa = smem + stage * STAGE_BYTES;
        const unsigned char* sb = sa + BM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < BK / 32; kk++) {
            half8_t fa[MT], fb[NT];
            const int ch = kk * 4 + fq;
#pragma unroll
            for (int i = 0; i < MT; i++) {
                int r = wm * WM + i * 16 + fr;
                fa[i] = *reinterpret_cast<const half8_t*>(sa + r * (BK * 2) + ((ch ^ (r & 7)) << 4));
                if (g.relu_in) fa[i] = relu8(fa[i]);
            }
#pragma unroll
            for (int j = 0; j < NT; j++) {
                int r = wn * WN + j * 16 + fr;
                fb[j] = *reinterpret_cast<const half8_t*>(sb + r * (BK * 2) + ((ch ^ (r & 7)) << 4));
            }
            if (g.prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int j = 0; j < NT; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            if (g.prio) __builtin_amdgcn_s_setprio(0);
        }
        if (++stage == NSTAGE) stage = 0;
    }
    __syncthreads();

    // ---- epilogue: accumulators -> LDS (fp32) -> coalesced fused store, one row band per pass.  The residual rows a thread
    // will add are loaded BEFORE the band is staged (their latency overlaps the staging and the barrier).
    float* cs = reinterpret_cast<float*>(smem);
    const float* lnp = reinterpret_cast<const float*>(smem + LNP_OFF);       // [BM][2], valid after the staging barrier
    if (g.ln_stats && tid < BM) {
        float* w = reinterpret_cast<float*>(smem + LNP_OFF) + 2 * tid;
        w[0] = ln_rs; w[1] = ln_rm;
    }
    const float* bias = g.bias ? g.bias + (size_t)z * g.sBias : nullptr;
    constexpr int TPR = BN / 4;               // threads per output row (4 columns each)
    constexpr int RPP = NTHR / TPR;           // rows per pass
    constexpr int ITER = (EPI_ROWS + RPP - 1) / RPP;
    // residual rows in flight per thread (registers: the 8-wave tiles must stay <= 128 VGPRs, see the launch bounds)
    constexpr int PF = (MT * NT >= 12) ? 2 : (ITER < 4 ? ITER : 4);
    const int c4 = (tid % TPR) * 4;
    const int gn = n0 + c4;
    const int r0 = tid / TPR;
    const bool col_ok = gn < N && tid < RPP * TPR;          // (TPR need not divide the block: 128x192 has 48 threads per row)
    const bool plain = !g.shuf && g.rope_cols == 0;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 b4 = zero4;
    if (bias && col_ok && plain) b4 = *reinterpret_cast<const f32x4*>(bias + gn);
    if constexpr (EPI != 0) {
        // compile-time epilogue (the EPI codes of gemm256_body): plain row-major output, bias present, vector-aligned rows
        constexpr bool F16OUT = EPI != 3, RES = (EPI == 3 || EPI == 5);
        const int gnc = col_ok ? gn : 0;
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + gnc);
#pragma unroll
        for (int p = 0; p < EPI_PASSES; p++) {
            f32x4 rv[ITER], sv[ITER];
            if (RES) {
#pragma unroll
                for (int i = 0; i < ITER; i++) {
                    const size_t gm = (size_t)min(m0 + p * EPI_ROWS + r0 + i * RPP, M - 1);
                    if (EPI == 3) {
                        rv[i] = *reinterpret_cast<const f32x4*>((const float*)g.res1 + (size_t)z * g.sR1 + gm * g.ldr1 + gnc);
                    } else {
                        const half4_t h1 = *reinterpret_cast<const half4_t*>((const h16*)g.res1 + (size_t)z * g.sR1 + gm * g.ldr1 + gnc);
                        rv[i] = f32x4{(float)h1[0], (float)h1[1], (float)h1[2], (float)h1[3]};
                        if (g.res2) {
                            const half4_t h2 = *reinterpret_cast<const half4_t*>((const h16*)g.res2 + (size_t)z * g.sR2 + gm * g.ldr2 + gnc);
                            sv[i] = f32x4{(float)h2[0], (float)h2[1], (float)h2[2], (float)h2[3]};
                        }
                    }
                }
            }
            if (p > 0) __syncthreads();
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int j = 0; j < NT; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = wm * WM + i * 16 + fq * 4 + e - p * EPI_ROWS;
                        const int c = wn * WN + j * 16 + fr;
                        if (EPI_PASSES == 1 || (r >= 0 && r < EPI_ROWS)) cs[r * CPAD + c] = acc[i][j][e];
                    }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < ITER; i++) {
                const int r = r0 + i * RPP;
                const int gm = m0 + p * EPI_ROWS + r;
                // (every lane runs the arithmetic -- the slab statistics are 16-lane butterflies --, only the stores are bounded)
                const bool ok = col_ok && r < EPI_ROWS && gm < M;
                {
                    const int rc = min(r, EPI_ROWS - 1);
                    f32x4 v = *reinterpret_cast<const f32x4*>(cs + rc * CPAD + c4);
                    if (g.ln_stats) {
                        const int lr = p * EPI_ROWS + rc;
                        v = ln_pre4(v, lnp[2 * lr], lnp[2 * lr + 1], *reinterpret_cast<const f32x4*>(g.ln_c + gnc));
                    }
                    v += bb;
                    if (EPI == 2) {
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = gelu_fast(v[e]);
                    }
                    if (EPI == 4) {
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
                    }
                    if (RES) {
                        v += rv[i];
                        if (EPI == 5 && g.res2) v += sv[i];
                    }
                    const size_t off = (size_t)z * g.sC + (size_t)gm * g.ldc + gn;
                    if (F16OUT) {
                        const half4_t o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                        if (ok) *reinterpret_cast<half4_t*>((h16*)g.C + off) = o;
                    } else {
                        if (ok) *reinterpret_cast<f32x4*>((float*)g.C + off) = v;
                        if (EPI == 3 && g.stats_out) {            // producer side of the LayerNorm fold: fp16 copy + slab statistics
                            float ssum, sm2;
                            slab_stats4(v, ssum, sm2);
                            if (ok) {
                                const half4_t o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                                *reinterpret_cast<half4_t*>(g.out16 + (size_t)gm * g.ld16 + gn) = o;
                                if ((tid & 15) == 0) *reinterpret_cast<float2*>(g.stats_out + ((size_t)(gn >> 6) * M + gm) * 2) = float2{ssum, sm2};
                            }
                        }
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < EPI_PASSES; p++) {
        f32x4 r1v[PF];
        auto load_r1 = [&](int i) {
            const int r = r0 + i * RPP;
            const int gm = m0 + p * EPI_ROWS + r;
            return (r < EPI_ROWS && gm < M) ? load_res4(g.res1, g.res1_f16, (size_t)z * g.sR1 + (size_t)gm * g.ldr1 + gn) : zero4;
        };
        if (plain && col_ok && g.res1) {
#pragma unroll
            for (int i = 0; i < PF; i++) r1v[i] = load_r1(i);
        }
        if (p > 0) __syncthreads();           // the previous band has been read
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int j = 0; j < NT; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int r = wm * WM + i * 16 + fq * 4 + e - p * EPI_ROWS;
                    const int c = wn * WN + j * 16 + fr;
                    if (EPI_PASSES == 1 || (r >= 0 && r < EPI_ROWS)) cs[r * CPAD + c] = acc[i][j][e];
                }
        __syncthreads();
        if (col_ok) {
            if (plain) {
#pragma unroll
                for (int i = 0; i < ITER; i++) {
                    const int r = r0 + i * RPP;
                    const int gm = m0 + p * EPI_ROWS + r;
                    f32x4 r1 = zero4;
                    if (g.res1) {
                        r1 = r1v[i % PF];
                        if (i + PF < ITER) r1v[i % PF] = load_r1(i + PF);
                    }
                    if (r < EPI_ROWS && gm < M) {
                        const f32x4 r2 = g.res2 ? load_res4(g.res2, g.res2_f16, (size_t)z * g.sR2 + (size_t)gm * g.ldr2 + gn) : zero4;
                        f32x4 v = *reinterpret_cast<const f32x4*>(cs + r * CPAD + c4);
                        if (g.ln_stats) v = ln_pre4(v, lnp[2 * (p * EPI_ROWS + r)], lnp[2 * (p * EPI_ROWS + r) + 1], *reinterpret_cast<const f32x4*>(g.ln_c + gn));
                        store_out4(g, z, (size_t)gm * g.ldc + gn, fused_finish4(g, v, b4, bias != nullptr, r1, r2));
                    }
                }
            } else {
                for (int r = r0; r < EPI_ROWS; r += RPP) {
                    const int gm = m0 + p * EPI_ROWS + r;
                    if (gm >= M) break;
                    f32x4 v = *reinterpret_cast<const f32x4*>(cs + r * CPAD + c4);
                    const float rs = g.ln_stats ? lnp[2 * (p * EPI_ROWS + r)] : 1.f, rm = g.ln_stats ? lnp[2 * (p * EPI_ROWS + r) + 1] : 0.f;
                    if (gn < g.rope_cols) rope_store4(g, z, bias, gm, gn, cs + r * CPAD, c4, rs, rm);
                    else {
                        if (g.ln_stats) v = ln_pre4(v, rs, rm, *reinterpret_cast<const f32x4*>(g.ln_c + gn));
                        fused_store4(g, z, bias, gm, gn, v);
                    }
                }
            }
        }
    }
}

// (8-wave tiles whose LDS lets two workgroups share a CU are held to 128 VGPRs: 4 waves per SIMD)
#define CUT3R_TILE_BOUNDS __launch_bounds__(64 * WAVES_M * WAVES_N, (WAVES_M * WAVES_N == 8 && (BM + BN) * BK * 2 * NSTAGE <= 81920) ? 4 : 1)
template <int BM, int BN, int NSTAGE, int WAVES_M = 2, int WAVES_N = 2, int ADDR = 0, int EPI = 0>
__global__ CUT3R_TILE_BOUNDS void gemm_kernel(const GemmArgs g) {
    gemm_tile_body<BM, BN, NSTAGE, WAVES_M, WAVES_N, ADDR, EPI>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}

// the addressing mode of gemm_tile_body a problem qualifies for (0: generic)
static int tile_addr_mode(const GemmArgs& g) {
    static const bool off = [] { const char* e = getenv("CUT3R_GEMM_FASTADDR"); return e && atoi(e) == 0; }();
    if (off || (g.K % BK) != 0) return 0;
    if (g.conv_k == 3) return (g.Cin >= 64 && (g.Cin & (g.Cin - 1)) == 0 && g.K == 9 * g.Cin) ? 2 : 0;
    return 1;
}

// TWO independent problems in one launch (1-D grid: the first nblk0 workgroups belong to problem 0): the state-side and the
// image-side GEMM of a decoder layer have the same N, K and epilogue but their own operands and row counts
// (src/dust3r/model.py:669-692: both blocks of a layer read the previous layer's pair, so they are independent).  One launch
// of 2 x 294 tiles fills the chip where two launches of 294 tiles each leave 43 % of the workgroup slots empty.
struct GemmPairArgs { GemmArgs p[2]; int nblk0; };
template <int BM, int BN, int NSTAGE, int WAVES_M = 2, int WAVES_N = 2, int ADDR = 0, int EPI = 0>
__global__ CUT3R_TILE_BOUNDS void gemm_pair_kernel(const GemmPairArgs a) {
    const int sel = (int)blockIdx.x >= a.nblk0 ? 1 : 0;
    gemm_tile_body<BM, BN, NSTAGE, WAVES_M, WAVES_N, ADDR, EPI>(a.p[sel], blockIdx.x - sel * a.nblk0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, 8 waves (2 x 4, 128 x 64 outputs per wave), ping-pong schedule: the kernel for the large GEMMs
// (ViT-L encoder linears at batch >= 8 keyframes, DPT 3x3 convolutions, decoder linears at window batch >= 8).
//
// Why a second tile: a 128^2 tile moves (128+128)*64*2 B from L2 into LDS per 2.1 MFLOP (64 FLOP/B); measured on
// MI355X the L2->LDS path delivers ~45-50 GB/s per CU, which caps that tile near 700-800 TFLOP/s whatever the MFMA
// schedule does.  256^2 halves the bytes per FLOP, and one workgroup per CU with 128 KiB of LDS is the natural
// occupancy for it; latency is then hidden inside the workgroup instead of by a co-resident one:
//   * waves 0-3 (wr = 0) and waves 4-7 (wr = 1) share the 4 SIMDs pairwise and run ONE s_barrier apart: while one
//     group is in a 16-MFMA section (a 64 x 32 quadrant of its outputs x K = 64) the other group is in its load
//     section (ds_read_b128 fragment reads + 2 LDS-DMA pieces), so each SIMD's matrix pipe always has a feeder;
//   * a K-tile is staged as four 16-KiB units ordered by first use: U0 = A rows of quadrant-row 0, U1 = B rows
//     (output columns) of quadrant-column 0, U2 = B of quadrant-column 1, U3 = A of quadrant-row 1; two K-tiles of
//     LDS (2 x 64 KiB).  Phase p of K-tile t reads {U0,U1 | U2 | U3 | -} and re-stages {U2(t+1) | U3(t+1) | U0(t+2) |
//     U1(t+2)}: every unit is overwritten >= 2 phases (4 barriers) after its last ds_read has been retired (WAR) and
//     is issued >= 4 phases before its first read; ONE counted wait per K-tile (phase 3: s_waitcnt vmcnt(4), two
//     units stay in flight across the barriers) retires it a full phase + barrier before that read (RAW).
//   * B fragments of both quadrant-columns stay in registers, so a K-tile costs 24 ds_read_b128 per wave for 64
//     MFMAs (LDS array ~45 % busy incl. the DMA writes).
// Epilogue: each wave stages its own 128 x 64 block through its private 16 KiB of LDS (no workgroup barrier) and
// stores full 128/256-B row segments through fused_store4.
#ifndef EPI_UNROLL
#define EPI_UNROLL 1
#endif
// EPI: compile-time epilogue.  0 = every case at run time (activation, one or two residuals of either type, either output type,
// pixel shuffle, RoPE): ~50 vector instructions, scalar branches and spill reloads between two stores (s_memtime: 400-560 of an
// iteration's 650-800 cycles).  1 = fp16 out + bias; 2 = fp16 out + bias + GELU; 3 = fp32 out + bias + fp32 residual: the three
// epilogues of the network's Linear layers; 4 = fp16 out + bias + ReLU, 5 = fp16 out + bias + one or two fp16 residuals: the DPT
// convolutions; 6 = fp16 out + bias + 2-D RoPE of the 64-wide heads in columns < rope_cols (a wave's 64-column slab is one head: the
// rotation partner of a lane's 8 columns is 16 columns away in the wave's own staging row).  Straight-line; same per-element
// arithmetic, same order (bias, activation, residuals; RoPE on the fp16-rounded projection, as the stand-alone kernel sees it).
// LN (round 4): consumer side of the LayerNorm fold for the fp16-output epilogues 1 / 2 / 6 -- thread t < 256 turns the slab statistics of
// the tile's row t into (rstd, rstd * mu) while the prologue's DMA is in flight and parks them in 2 KiB behind the ring (published by the
// prologue's barrier); the epilogue applies acc * rstd - (rstd mu) c_n before the bias.  The
// producer side (EPI 3 with g.stats_out: fp16 copy + slab statistics of the fp32 residual stream it writes) needs no LDS and is decided
// at run time.
template <bool CONV3, bool RELU_IN, bool FAST_DMA = false, int EPI = 0, bool LN = false>
DEVINL void gemm256_body(const GemmArgs& g, const int bx, const int bz) {
    constexpr int UNIT = 128 * BK * 2;      // 16 KiB: 128 rows x 64 halfs
    constexpr int BUF = 4 * UNIT;
    static_assert(!LN || EPI == 1 || EPI == 2 || EPI == 6, "the LayerNorm fold feeds the fp16-output epilogues");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF + (LN ? 2 * 1024 : 0)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int z = bz;
    const h16* __restrict__ A = g.A + (size_t)z * g.sA;
    const h16* __restrict__ Bm = g.B + (size_t)z * g.sB;
    const int M = g.M, N = g.N, K = g.K;
    int pid_m, pid_n;
    xcd_tile(bx, (M + 255) / 256, (N + 255) / 256, pid_m, pid_n);
    const int m0 = pid_m * 256, n0 = pid_n * 256;
    const h16* zero = reinterpret_cast<const h16*>(g_zero16);

    // ---- loader coordinates: the two LDS-DMA pieces a thread contributes to a unit cover unit rows u0 and u0 + 64
    const int lrow = lane >> 3;
    const int csrc = (lane & 7) ^ lrow;            // swizzled source chunk (unit rows are 8-aligned per piece)
    const h16* a_ptr[2][2];                        // [quadrant-row][piece]
    int a_oy[2][2], a_ox[2][2];
    bool a_ok[2][2];
    const h16* b_ptr[2][2];                        // [quadrant-column][piece]
    bool b_ok[2][2];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int u = (wave + 8 * i) * 8 + lrow;                       // unit row 0..127
            const int gm = m0 + (u >> 6) * 128 + q * 64 + (u & 63);        // A unit row -> wave-row block wr = u>>6
            a_ok[q][i] = gm < M;
            if (CONV3) {
                const int gmc = a_ok[q][i] ? gm : 0;
                const int hw = g.Ho * g.Wo;
                const int b = gmc / hw, rem = gmc - b * hw;
                const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
                a_oy[q][i] = oy * g.cstride - 1;
                a_ox[q][i] = ox * g.cstride - 1;
                a_ptr[q][i] = A + (size_t)b * g.H * g.W * g.Cin;
            } else {
                a_oy[q][i] = a_ox[q][i] = 0;
                a_ptr[q][i] = A + (size_t)(a_ok[q][i] ? gm : 0) * g.lda;
            }
            const int gn = n0 + (u >> 5) * 64 + q * 32 + (u & 31);         // B unit row -> wave-column block wc = u>>5
            b_ok[q][i] = gn < N;
            b_ptr[q][i] = Bm + (size_t)(b_ok[q][i] ? gn : 0) * g.ldb;
        }

    const int nt = (K + BK - 1) / BK;
    // Plain operands with whole K-tiles (every Linear of the network): a DMA piece is ONE instruction with a scalar base (the
    // K-tile's column of A or W, advanced on the scalar unit) and a per-lane 32-bit byte offset computed once -- no per-piece
    // vector address arithmetic in the loop.  (Timed with s_memtime: the load segments of a K-tile, not its MFMAs, set the pace of
    // the two wave groups; a piece with 64-bit per-lane addresses, two selects and a readfirstlane costs ~70 cycles of wave time.)
    // Rows beyond M / N read the last valid row instead of zeros: those accumulators are never stored.
    // (FAST_DMA is chosen at launch: gemm256_fast_ok)
    constexpr bool fast_dma = FAST_DMA;
    unsigned a_off[2][2], b_off[2][2];
    // 3x3 convolution with Cin a power of two >= 64 (ADDR 2 of gemm_tile_body): wave-uniform tap, per-lane pointer + padding mask
    const h16* a_pc[2][2];
    unsigned a_msk[2][2];
    const int cin_shift = 31 - __builtin_clz((unsigned)(g.Cin > 0 ? g.Cin : 1));
    if (FAST_DMA && CONV3) {
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                a_pc[q][i] = a_ptr[q][i] + ((ptrdiff_t)a_oy[q][i] * g.W + a_ox[q][i]) * g.Cin + csrc * 8;
                unsigned mk = 0;
#pragma unroll
                for (int t = 0; t < 9; t++) {
                    const int iy = a_oy[q][i] + t / 3, ix = a_ox[q][i] + t % 3;
                    if (a_ok[q][i] && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) mk |= 1u << t;
                }
                a_msk[q][i] = mk;
            }
    }
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int u = (wave + 8 * i) * 8 + lrow;
            const int gm = min(m0 + (u >> 6) * 128 + q * 64 + (u & 63), M - 1);
            const int gn = min(n0 + (u >> 5) * 64 + q * 32 + (u & 31), N - 1);
            a_off[q][i] = (unsigned)(((size_t)gm * g.lda + csrc * 8) * 2);
            b_off[q][i] = (unsigned)(((size_t)gn * g.ldb + csrc * 8) * 2);
        }
    auto issue_fast = [&](const h16* mat, const unsigned (&off)[2], int kt, int unit) {
        const char* base = reinterpret_cast<const char*>(mat) + (size_t)kt * (BK * 2);       // wave-uniform
        unsigned char* slot = smem + (kt & 1) * BUF + unit * UNIT;
        const int wave_s = __builtin_amdgcn_readfirstlane(wave);        // (scalar: the LDS destination goes to M0)
#pragma unroll
        for (int i = 0; i < 2; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off[i]),
                                             (__attribute__((address_space(3))) void*)(slot + (wave_s + 8 * i) * 1024), 16, 0, 0);
    };
    // unit U of K-tile kt -> LDS-DMA (skipped past the last tile; the vmcnt below accounts for that)
    auto issue_a = [&](int q, int kt, int unit) {
        if (kt >= nt) return;
        if constexpr (fast_dma && !CONV3) { issue_fast(A, a_off[q], kt, unit); return; }
        if constexpr (fast_dma && CONV3) {
            const int k0 = kt * BK;
            const int tap = k0 >> cin_shift;
            const int dy = (tap >= 3) + (tap >= 6), dx = tap - 3 * dy;
            const int delta = (dy * g.W + dx) * g.Cin + (k0 & (g.Cin - 1));
            unsigned char* slot = smem + (kt & 1) * BUF + unit * UNIT;
            const int wave_s = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const h16* src = ((a_msk[q][i] >> tap) & 1) ? a_pc[q][i] + delta : zero;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(slot + (wave_s + 8 * i) * 1024), 16, 0, 0);
            }
            return;
        }
        unsigned char* slot = smem + (kt & 1) * BUF + unit * UNIT;
        const int k = kt * BK + csrc * 8;
        const bool kok = k < K;
        int ci = 0, dy = 0, dx = 0;
        if (CONV3) {
            const int tap = k / g.Cin;
            ci = k - tap * g.Cin;
            dy = tap / 3; dx = tap - dy * 3;
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const h16* src = zero;
            if (CONV3) {
                const int iy = a_oy[q][i] + dy, ix = a_ox[q][i] + dx;
                if (kok && a_ok[q][i] && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                    src = a_ptr[q][i] + ((size_t)iy * g.W + ix) * g.Cin + ci;
            } else if (kok && a_ok[q][i]) {
                src = a_ptr[q][i] + k;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(slot + (wave + 8 * i) * 1024), 16, 0, 0);
        }
    };
    auto issue_b = [&](int q, int kt, int unit) {
        if (kt >= nt) return;
        if constexpr (fast_dma) { issue_fast(Bm, b_off[q], kt, unit); return; }
        unsigned char* slot = smem + (kt & 1) * BUF + unit * UNIT;
        const int k = kt * BK + csrc * 8;
        const bool kok = k < K;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const h16* src = (kok && b_ok[q][i]) ? b_ptr[q][i] + k : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(slot + (wave + 8 * i) * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes inside a unit): row = block*16*i + fr, 16-B chunk (kk*4 + fq) ^ (row & 7)
    const int fr = lane & 15, fq = lane >> 4;
    const int offk0 = fr * 128 + (((0 + fq) ^ (fr & 7)) << 4);
    const int offk1 = fr * 128 + (((4 + fq) ^ (fr & 7)) << 4);
    const int a_base = wr * 64 * 128;              // this wave's 64 rows inside an A unit
    const int b_base = wc * 32 * 128;              // this wave's 32 rows inside a B unit

    f32x4 acc[8][4];           // (zeroed after the prologue's DMA has been issued: 128 moves under the first memory latency)
    half8_t fa[4][2], fb0[2][2], fb1[2][2];

    auto read_a = [&](const unsigned char* unit) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            fa[i][0] = *reinterpret_cast<const half8_t*>(unit + a_base + i * 2048 + offk0);
            fa[i][1] = *reinterpret_cast<const half8_t*>(unit + a_base + i * 2048 + offk1);
            if (RELU_IN) { fa[i][0] = relu8(fa[i][0]); fa[i][1] = relu8(fa[i][1]); }
        }
    };
    auto read_b = [&](const unsigned char* unit, half8_t (&fb)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            fb[j][0] = *reinterpret_cast<const half8_t*>(unit + b_base + j * 2048 + offk0);
            fb[j][1] = *reinterpret_cast<const half8_t*>(unit + b_base + j * 2048 + offk1);
        }
    };
#define CUT3R_BARRIER() asm volatile("s_barrier" ::: "memory")
#define CUT3R_QUADRANT(I0, J0, FB)                                                                              \
    do {                                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                           \
        _Pragma("unroll") for (int kk = 0; kk < 2; kk++)                                                         \
            _Pragma("unroll") for (int i = 0; i < 4; i++)                                                        \
                _Pragma("unroll") for (int j = 0; j < 2; j++)                                                    \
                    acc[I0 + i][J0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][kk], FB[j][kk], acc[I0 + i][J0 + j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                           \
    } while (0)

#ifndef CUT3R_G256_PHASES
#define CUT3R_G256_PHASES 2
#endif
    static_assert(!LN || CUT3R_G256_PHASES == 2, "the LayerNorm fold is wired into the two-phase schedule");
#if CUT3R_G256_PHASES == 2
    // ---- TWO phases per K-tile (one per 64-row half of the wave's 128 rows): phase A reads A-half 0 (U0) and both B halves (U1, U2)
    // and runs 32 MFMAs, phase B reads A-half 1 (U3) and runs the other 32 -- four barriers per K-tile, 32 MFMAs per barrier pair.
    // A slot is refilled as soon as both wave groups have read it: U0..U2 of K-tile t+2 go into the CURRENT buffer during phase B
    // (they were last read in phase A, by the lagging group one barrier later), U3 of K-tile t+1 into the other buffer during
    // phase A.  In flight behind the unit a phase needs: 4 units = 8 DMA instructions per wave (one counted wait per phase).
    LnRow ln_raw;
    if constexpr (LN) {          // thread t < 256 owns row t of the tile: requested BEFORE the DMA prologue (vmcnt retires in order), used after it
        if (tid < 256) ln_row_load(g.ln_stats, min(m0 + tid, M - 1), M, g.ln_nslab, ln_raw);
    }
    issue_a(0, 0, 0); issue_b(0, 0, 1); issue_b(1, 0, 2); issue_a(1, 0, 3);
    issue_a(0, 1, 0); issue_b(0, 1, 1); issue_b(1, 1, 2);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (LN) {
        // (rstd, rstd * mu) of the tile's 256 rows go to 2 KiB behind the ring, published by the prologue's barrier.  The store pins the
        // arithmetic HERE (left alone, the compiler sinks it to its first use behind the main loop and carries the raw registers there
        // through scratch)
        if (tid < 256) {
            float rs_, rm_;
            ln_row_finish(ln_raw, g.ln_nslab, K, g.ln_eps, rs_, rm_);
            *reinterpret_cast<float2*>(smem + 2 * BUF + tid * 8) = float2{rs_, rm_};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (nt > 1) wait_vmcnt<6>(); else wait_vmcnt<0>();
    CUT3R_BARRIER();
    if (wr == 1) CUT3R_BARRIER();          // stagger: the second wave group runs one barrier behind the first

    for (int t = 0; t < nt; t++) {
        const unsigned char* buf = smem + (t & 1) * BUF;
        // phase A: rows 0..63 of the wave tile
        read_a(buf);
        read_b(buf + UNIT, fb0);
        read_b(buf + 2 * UNIT, fb1);
        issue_a(1, t + 1, 3);
        if (t + 1 < nt) wait_vmcnt<8>(); else wait_vmcnt<0>();      // U3 of K-tile t has landed
        // this phase's fragments are in registers before the barrier: the other group refills these slots right after it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        CUT3R_BARRIER();
        CUT3R_QUADRANT(0, 0, fb0);
        CUT3R_QUADRANT(0, 2, fb1);
        CUT3R_BARRIER();
        // phase B: rows 64..127
        read_a(buf + 3 * UNIT);
        issue_a(0, t + 2, 0); issue_b(0, t + 2, 1); issue_b(1, t + 2, 2);
        if (t + 2 < nt) wait_vmcnt<8>(); else if (t + 1 < nt) wait_vmcnt<2>(); else wait_vmcnt<0>();      // U0..U2 of K-tile t+1 have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        CUT3R_BARRIER();
        CUT3R_QUADRANT(4, 0, fb0);
        CUT3R_QUADRANT(4, 2, fb1);
        CUT3R_BARRIER();
    }
#else
    // ---- prologue: K-tile 0 complete, U0/U1 of K-tile 1 in flight
    issue_a(0, 0, 0); issue_b(0, 0, 1); issue_b(1, 0, 2); issue_a(1, 0, 3);
    issue_a(0, 1, 0); issue_b(0, 1, 1);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nt > 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
    CUT3R_BARRIER();
    if (wr == 1) CUT3R_BARRIER();          // stagger: the second wave group runs one barrier behind the first

    for (int t = 0; t < nt; t++) {
        const unsigned char* buf = smem + (t & 1) * BUF;
        // phase 0: quadrant (0,0)
        read_a(buf);
        read_b(buf + UNIT, fb0);
        issue_b(1, t + 1, 2);
        CUT3R_BARRIER();
        CUT3R_QUADRANT(0, 0, fb0);
        CUT3R_BARRIER();
        // phase 1: quadrant (0,1)
        read_b(buf + 2 * UNIT, fb1);
        issue_a(1, t + 1, 3);
        CUT3R_BARRIER();
        CUT3R_QUADRANT(0, 2, fb1);
        CUT3R_BARRIER();
        // phase 2: quadrant (1,1)
        read_a(buf + 3 * UNIT);
        issue_a(0, t + 2, 0);
        CUT3R_BARRIER();
        CUT3R_QUADRANT(4, 2, fb1);
        CUT3R_BARRIER();
        // phase 3: quadrant (1,0); the one counted wait of the K-tile retires every unit of K-tile t+1
        issue_b(0, t + 2, 1);
        if (t + 2 < nt) wait_vmcnt<4>(); else wait_vmcnt<0>();
        CUT3R_BARRIER();
        CUT3R_QUADRANT(4, 0, fb0);
        CUT3R_BARRIER();
    }
#endif
    if (wr == 0) CUT3R_BARRIER();          // re-join the two groups: every ds_read of the workgroup has been consumed
#undef CUT3R_QUADRANT

    // ---- epilogue: per-wave staging (32 rows x 64 columns per pass) in this wave's private 16 KiB
    // (measured alternative: operand-swapped MFMAs + direct 8-byte stores from the accumulators -- 32-B row segments,
    //  +8 us per tile).  Residual rows are loaded before the band is staged (a dependent global load per row made this
    //  epilogue 1.8x the K = 1024 main loop); fp16 outputs leave as 16-byte stores (8 columns per lane: stores are
    //  issue-bound, half the instructions).  Per-element arithmetic is the same in every variant.
    constexpr int CP = 68;
    float* cs = reinterpret_cast<float*>(smem + wave * 16384);
    const float* bias = g.bias ? g.bias + (size_t)z * g.sBias : nullptr;
    const bool plain = !g.shuf && g.rope_cols == 0;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float* lnp = reinterpret_cast<const float*>(smem + 2 * BUF) + wr * 256;      // LN: (rstd, rstd * mu) of this wave's 128 rows
    if constexpr (EPI == 1 || EPI == 2 || EPI == 4 || EPI == 5 || EPI == 6) {
        const int er = lane >> 3, ec = (lane & 7) * 8;
        const int gn = n0 + wc * 64 + ec;
        const bool col_ok = gn + 8 <= N;               // (every lane stages its accumulators; only the readers are column-bounded)
        const int gnc = col_ok ? gn : 0;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + gnc), b1 = *reinterpret_cast<const f32x4*>(bias + gnc + 4);
        f32x4 lc0 = zero4, lc1 = zero4, lpc0 = zero4, lpc1 = zero4;      // LN: c_n of this lane's 8 columns (and of its RoPE partner's)
        if constexpr (LN) { lc0 = *reinterpret_cast<const f32x4*>(g.ln_c + gnc); lc1 = *reinterpret_cast<const f32x4*>(g.ln_c + gnc + 4); }
        // EPI 6: the slab n0 + wc*64 .. +63 is one head; rotated when it lies below rope_cols (wave-uniform); partner's bias
        const bool rope_here = EPI == 6 && (n0 + wc * 64) < g.rope_cols && col_ok;
        f32x4 pb0 = zero4, pb1 = zero4;
        if (rope_here) {
            const int pc = gnc + ((ec & 16) ? -16 : 16);
            pb0 = *reinterpret_cast<const f32x4*>(bias + pc);
            pb1 = *reinterpret_cast<const f32x4*>(bias + pc + 4);
            if constexpr (LN) { lpc0 = *reinterpret_cast<const f32x4*>(g.ln_c + pc); lpc1 = *reinterpret_cast<const f32x4*>(g.ln_c + pc + 4); }
        }
        // EPI 6: the wave keeps the first ROPE_LDS rows of the cos|sin table (row = 16 cos | 16 sin) and the clamped table rows of its
        // 128 output rows behind its staging band, in its own 16 KiB: the rotation of a row then costs LDS reads instead of a dependent
        // chain position -> table in global memory per row (measured: that chain made this epilogue 2.4x the plain fp16 one).  Rows of
        // the table beyond ROPE_LDS (images wider than 46 patches) keep the global path, decided per wave and row group.
        constexpr int ROPE_LDS = 48;
        float* tl = cs + 32 * CP;
        int* psl = reinterpret_cast<int*>(tl + ROPE_LDS * 32);
        if (EPI == 6 && __builtin_amdgcn_readfirstlane((n0 + wc * 64) < g.rope_cols)) {
#pragma unroll
            for (int f0 = 0; f0 < ROPE_LDS * 8; f0 += 64) {
                const int f = f0 + lane, p = f >> 3, part = f & 7;
                f32x4 v = zero4;
                if (p < g.rope_npos) v = *reinterpret_cast<const f32x4*>(g.rope_table + (size_t)(part >> 2) * g.rope_npos * 16 + p * 16 + (part & 3) * 4);
                *reinterpret_cast<f32x4*>(tl + p * 32 + part * 4) = v;
            }
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int r = lane + 64 * i;
                const int gmr = min(m0 + wr * 128 + r, M - 1);
                const longlong2 pp = *reinterpret_cast<const longlong2*>(g.rope_pos + (size_t)gmr * 2);
                long long py = pp.x - g.rope_pmin, px = pp.y - g.rope_pmin;
                py = py < 0 ? 0 : (py >= g.rope_npos ? g.rope_npos - 1 : py);
                px = px < 0 ? 0 : (px >= g.rope_npos ? g.rope_npos - 1 : px);
                psl[r * 2] = (int)py;
                psl[r * 2 + 1] = (int)px;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        h16* crow = (h16*)g.C + (size_t)z * g.sC + (size_t)(m0 + wr * 128 + er) * g.ldc + gnc;
        const size_t step8 = (size_t)8 * g.ldc;
#pragma unroll
        for (int mp = 0; mp < 4; mp++) {
            f32x4 ra[4], rb[4], sa[4], sb[4];           // EPI 5: the residual rows of the band, loaded before it is staged
            if (EPI == 5) {
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const size_t gm = (size_t)min(m0 + wr * 128 + mp * 32 + it * 8 + er, M - 1);
                    const half8_t h1 = *reinterpret_cast<const half8_t*>((const h16*)g.res1 + (size_t)z * g.sR1 + gm * g.ldr1 + gnc);
                    ra[it] = f32x4{(float)h1[0], (float)h1[1], (float)h1[2], (float)h1[3]};
                    rb[it] = f32x4{(float)h1[4], (float)h1[5], (float)h1[6], (float)h1[7]};
                    if (g.res2) {
                        const half8_t h2 = *reinterpret_cast<const half8_t*>((const h16*)g.res2 + (size_t)z * g.sR2 + gm * g.ldr2 + gnc);
                        sa[it] = f32x4{(float)h2[0], (float)h2[1], (float)h2[2], (float)h2[3]};
                        sb[it] = f32x4{(float)h2[4], (float)h2[5], (float)h2[6], (float)h2[7]};
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++) cs[(i * 16 + fq * 4 + e) * CP + j * 16 + fr] = acc[mp * 2 + i][j][e];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int rr = it * 8 + er;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(cs + rr * CP + ec);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(cs + rr * CP + ec + 4);
                float rs = 1.f, rm = 0.f;
                if constexpr (LN) {
                    const float2 pr = *reinterpret_cast<const float2*>(lnp + 2 * (mp * 32 + rr));
                    rs = pr.x; rm = pr.y;
                    v0 = ln_pre4(v0, rs, rm, lc0);
                    v1 = ln_pre4(v1, rs, rm, lc1);
                }
                v0 += b0;
                v1 += b1;
                if (EPI == 2) {
#pragma unroll
                    for (int e = 0; e < 4; e++) { v0[e] = gelu_fast(v0[e]); v1[e] = gelu_fast(v1[e]); }
                }
                if (EPI == 4) {
#pragma unroll
                    for (int e = 0; e < 4; e++) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
                }
                if (EPI == 5) {
                    v0 += ra[it]; v1 += rb[it];
                    if (g.res2) { v0 += sa[it]; v1 += sb[it]; }
                }
                if (EPI == 6 && rope_here) {
                    // head-local column ec: half X = ec / 32 (y or x position), pair offset +-16 inside the half, frequency index ec % 16 ..
                    const int po = (ec & 16) ? -16 : 16;
                    f32x4 p0 = *reinterpret_cast<const f32x4*>(cs + rr * CP + ec + po);
                    f32x4 p1 = *reinterpret_cast<const f32x4*>(cs + rr * CP + ec + po + 4);
                    if constexpr (LN) { p0 = ln_pre4(p0, rs, rm, lpc0); p1 = ln_pre4(p1, rs, rm, lpc1); }
                    p0 += pb0;
                    p1 += pb1;
                    const int pv = psl[(mp * 32 + rr) * 2 + (ec >> 5)];
                    f32x4 c0, c1, s0, s1;
                    if (__builtin_amdgcn_ballot_w64(pv >= ROPE_LDS) == 0) {
                        const float* ct = tl + pv * 32 + (ec & 15);
                        c0 = *reinterpret_cast<const f32x4*>(ct); c1 = *reinterpret_cast<const f32x4*>(ct + 4);
                        s0 = *reinterpret_cast<const f32x4*>(ct + 16); s1 = *reinterpret_cast<const f32x4*>(ct + 20);
                    } else {
                        const float* ct = g.rope_table + (size_t)pv * 16 + (ec & 15);
                        const float* st = ct + (size_t)g.rope_npos * 16;
                        c0 = *reinterpret_cast<const f32x4*>(ct); c1 = *reinterpret_cast<const f32x4*>(ct + 4);
                        s0 = *reinterpret_cast<const f32x4*>(st); s1 = *reinterpret_cast<const f32x4*>(st + 4);
                    }
                    const bool lower = !(ec & 16);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float t0 = (float)(h16)p0[e] * s0[e], t1 = (float)(h16)p1[e] * s1[e];
                        asm volatile("" : "+v"(t0), "+v"(t1));          // separate multiplies, as in rope2d_kernel (no re-contraction)
                        v0[e] = rope_rot((float)(h16)v0[e], c0[e], lower ? -t0 : t0);
                        v1[e] = rope_rot((float)(h16)v1[e], c1[e], lower ? -t1 : t1);
                    }
                }
                const half8_t o = {(h16)v0[0], (h16)v0[1], (h16)v0[2], (h16)v0[3], (h16)v1[0], (h16)v1[1], (h16)v1[2], (h16)v1[3]};
                if (col_ok && m0 + wr * 128 + mp * 32 + rr < M) *reinterpret_cast<half8_t*>(crow + (size_t)(mp * 4 + it) * step8) = o;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if (EPI == 3 && g.stats_out) {
        // producer side of the LayerNorm fold: the same sums in the same order with EIGHT columns per lane, so that the fp16 copy leaves
        // as 16-byte stores (stores are issue-bound: one per lane and row instead of two)
        const int er = lane >> 3, ec = (lane & 7) * 8;
        const int gn = n0 + wc * 64 + ec;
        const bool col_ok = gn + 8 <= N;
        const int gnc = col_ok ? gn : 0;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + gnc), b1 = *reinterpret_cast<const f32x4*>(bias + gnc + 4);
        const float* rrow = (const float*)g.res1 + (size_t)z * g.sR1 + gnc;
        float* crow = (float*)g.C + (size_t)z * g.sC + gnc;
        f32x4 ra[2][4], rb[2][4];           // (the next band's residual rows are requested before this one is staged: see the plain path below)
        auto load_band = [&](int mp, f32x4 (&a_)[4], f32x4 (&b_)[4]) {
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int gm = min(m0 + wr * 128 + mp * 32 + it * 8 + er, M - 1);
                a_[it] = *reinterpret_cast<const f32x4*>(rrow + (size_t)gm * g.ldr1);
                b_[it] = *reinterpret_cast<const f32x4*>(rrow + (size_t)gm * g.ldr1 + 4);
            }
        };
        load_band(0, ra[0], rb[0]);
#pragma unroll
        for (int mp = 0; mp < 4; mp++) {
            if (mp + 1 < 4) load_band(mp + 1, ra[(mp + 1) & 1], rb[(mp + 1) & 1]);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++) cs[(i * 16 + fq * 4 + e) * CP + j * 16 + fr] = acc[mp * 2 + i][j][e];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int rr = it * 8 + er;
                const int gm = m0 + wr * 128 + mp * 32 + rr;
                const f32x4 v0 = (*reinterpret_cast<const f32x4*>(cs + rr * CP + ec) + b0) + ra[mp & 1][it];
                const f32x4 v1 = (*reinterpret_cast<const f32x4*>(cs + rr * CP + ec + 4) + b1) + rb[mp & 1][it];
                float ssum, sm2;
                slab_stats8(v0, v1, ssum, sm2);
                if (col_ok && gm < M) {
                    *reinterpret_cast<f32x4*>(crow + (size_t)gm * g.ldc) = v0;
                    *reinterpret_cast<f32x4*>(crow + (size_t)gm * g.ldc + 4) = v1;
                    const half8_t o = {(h16)v0[0], (h16)v0[1], (h16)v0[2], (h16)v0[3], (h16)v1[0], (h16)v1[1], (h16)v1[2], (h16)v1[3]};
                    *reinterpret_cast<half8_t*>(g.out16 + (size_t)gm * g.ld16 + gnc) = o;
                    if ((lane & 7) == 0) *reinterpret_cast<float2*>(g.stats_out + ((size_t)(gnc >> 6) * M + gm) * 2) = float2{ssum, sm2};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if constexpr (EPI == 3) {
        const int er = lane >> 4, ec = (lane & 15) * 4;
        const int gn = n0 + wc * 64 + ec;
        const bool col_ok = gn < N;
        const int gnc = col_ok ? gn : 0;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + gnc);
        const float* rrow = (const float*)g.res1 + (size_t)z * g.sR1 + gnc;
        float* crow = (float*)g.C + (size_t)z * g.sC + gnc;
        // The residual rows of band mp + 1 are requested BEFORE band mp is staged and stored (two register sets; the accumulators a band
        // has staged are dead, so the budget holds): a band no longer starts with an exposed HBM round trip -- this epilogue is 512 KB of
        // traffic per tile and was latency-, not bandwidth-bound (27 GB/s per CU with 40 % of the CUs in it).  (In-place residual: a band's
        // loads and the previous band's stores touch different rows.)
        f32x4 r1v[2][8];
        auto load_band = [&](int mp, f32x4 (&r)[8]) {
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const int gm = min(m0 + wr * 128 + mp * 32 + it * 4 + er, M - 1);
                r[it] = *reinterpret_cast<const f32x4*>(rrow + (size_t)gm * g.ldr1);
            }
        };
        load_band(0, r1v[0]);
#pragma unroll
        for (int mp = 0; mp < 4; mp++) {
            if (mp + 1 < 4) load_band(mp + 1, r1v[(mp + 1) & 1]);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++) cs[(i * 16 + fq * 4 + e) * CP + j * 16 + fr] = acc[mp * 2 + i][j][e];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const int rr = it * 4 + er;
                const int gm = m0 + wr * 128 + mp * 32 + rr;
                const f32x4 v = (*reinterpret_cast<const f32x4*>(cs + rr * CP + ec) + b4) + r1v[mp & 1][it];
                if (col_ok && gm < M) *reinterpret_cast<f32x4*>(crow + (size_t)gm * g.ldc) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if (plain && g.out_f16 && !g.res2 && (N & 7) == 0 && (g.ldc & 7) == 0) {
        const int er = lane >> 3, ec = (lane & 7) * 8;
        const int gn = n0 + wc * 64 + ec;
        const bool col_ok = gn + 8 <= N;
        f32x4 b0 = zero4, b1 = zero4;
        if (bias && col_ok) { b0 = *reinterpret_cast<const f32x4*>(bias + gn); b1 = *reinterpret_cast<const f32x4*>(bias + gn + 4); }
#pragma unroll
        for (int mp = 0; mp < 4; mp++) {
            f32x4 ra[4], rb[4];
            if (g.res1 && col_ok) {
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const int gm = m0 + wr * 128 + mp * 32 + it * 8 + er;
                    const size_t off = (size_t)z * g.sR1 + (size_t)min(gm, M - 1) * g.ldr1 + gn;
                    ra[it] = load_res4(g.res1, g.res1_f16, off);
                    rb[it] = load_res4(g.res1, g.res1_f16, off + 4);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++) cs[(i * 16 + fq * 4 + e) * CP + j * 16 + fr] = acc[mp * 2 + i][j][e];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (col_ok) {
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const int rr = it * 8 + er;
                    const int gm = m0 + wr * 128 + mp * 32 + rr;
                    if (gm < M) {
                        const f32x4 v0 = fused_finish4(g, *reinterpret_cast<const f32x4*>(cs + rr * CP + ec), b0, bias != nullptr, ra[it], zero4);
                        const f32x4 v1 = fused_finish4(g, *reinterpret_cast<const f32x4*>(cs + rr * CP + ec + 4), b1, bias != nullptr, rb[it], zero4);
                        half8_t o = {(h16)v0[0], (h16)v0[1], (h16)v0[2], (h16)v0[3], (h16)v1[0], (h16)v1[1], (h16)v1[2], (h16)v1[3]};
                        *reinterpret_cast<half8_t*>((h16*)g.C + (size_t)z * g.sC + (size_t)gm * g.ldc + gn) = o;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else {
        const int er = lane >> 4, ec = (lane & 15) * 4;
        const int gn = n0 + wc * 64 + ec;
        f32x4 b4 = zero4;
        if (bias && gn < N && plain) b4 = *reinterpret_cast<const f32x4*>(bias + gn);
#pragma unroll
        for (int mp = 0; mp < 4; mp++) {
            f32x4 r1v[8], r2v[8];
            if (plain && gn < N) {
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    const int gm = min(m0 + wr * 128 + mp * 32 + it * 4 + er, M - 1);
                    r1v[it] = g.res1 ? load_res4(g.res1, g.res1_f16, (size_t)z * g.sR1 + (size_t)gm * g.ldr1 + gn) : zero4;
                    r2v[it] = g.res2 ? load_res4(g.res2, g.res2_f16, (size_t)z * g.sR2 + (size_t)gm * g.ldr2 + gn) : zero4;
                }
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++) cs[(i * 16 + fq * 4 + e) * CP + j * 16 + fr] = acc[mp * 2 + i][j][e];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (gn < N) {
                if (plain) {
#pragma unroll
                    for (int it = 0; it < 8; it++) {
                        const int rr = it * 4 + er;
                        const int gm = m0 + wr * 128 + mp * 32 + rr;
                        if (gm < M)
                            store_out4(g, z, (size_t)gm * g.ldc + gn,
                                       fused_finish4(g, *reinterpret_cast<const f32x4*>(cs + rr * CP + ec), b4, bias != nullptr, r1v[it], r2v[it]));
                    }
                } else {
                    for (int it = 0; it < 8; it++) {
                        const int rr = it * 4 + er;
                        const int gm = m0 + wr * 128 + mp * 32 + rr;
                        if (gm < M) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(cs + rr * CP + ec);
                            if (gn < g.rope_cols) rope_store4(g, z, bias, gm, gn, cs + rr * CP, ec);
                            else fused_store4(g, z, bias, gm, gn, v);
                        }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
#undef CUT3R_BARRIER
}

template <bool CONV3, bool RELU_IN, bool FAST_DMA = false, int EPI = 0, bool LN = false>
__global__ __launch_bounds__(512) void gemm256_kernel(const GemmArgs g) { gemm256_body<CONV3, RELU_IN, FAST_DMA, EPI, LN>(g, blockIdx.x, blockIdx.z); }

// which compile-time epilogue of gemm256_body a plain Linear qualifies for (0: the run-time one)
static int gemm256_epi_mode(const GemmArgs& g) {
    static const bool off = [] { const char* e = getenv("CUT3R_GEMM_EPI"); return e && atoi(e) == 0; }();
    if (off || g.shuf || !g.bias || ((uintptr_t)g.bias & 15) || (g.sBias & 3)) return 0;
    if (g.rope_cols) {
        const bool ok = g.rope_d == 64 && (g.rope_cols & 63) == 0 && g.out_f16 && !g.res1 && !g.res2 && g.act == 0 && (g.N & 63) == 0 &&
                        (g.ldc & 7) == 0 && (g.sC & 7) == 0 && ((uintptr_t)g.C & 15) == 0 && g.rope_pos && g.rope_table &&
                        ((uintptr_t)g.rope_table & 15) == 0;
        return ok ? 6 : 0;
    }
    if (g.out_f16 && !g.res1 && (g.N & 7) == 0 && (g.ldc & 7) == 0 && (g.sC & 7) == 0 && ((uintptr_t)g.C & 15) == 0)
        return g.act == 1 ? 2 : (g.act == 2 ? 4 : 1);
    if (g.out_f16 && g.res1 && g.res1_f16 && g.act == 0 && (g.N & 7) == 0 && (g.ldc & 7) == 0 && (g.sC & 7) == 0 && ((uintptr_t)g.C & 15) == 0 &&
        (g.ldr1 & 7) == 0 && (g.sR1 & 7) == 0 && ((uintptr_t)g.res1 & 15) == 0 &&
        (!g.res2 || (g.res2_f16 && (g.ldr2 & 7) == 0 && (g.sR2 & 7) == 0 && ((uintptr_t)g.res2 & 15) == 0)))
        return 5;
    if (!g.out_f16 && g.res1 && !g.res2 && !g.res1_f16 && g.act == 0 && (g.N & 3) == 0 && (g.ldc & 3) == 0 && (g.ldr1 & 3) == 0 && (g.sC & 3) == 0 &&
        (g.sR1 & 3) == 0 && (((uintptr_t)g.C | (uintptr_t)g.res1) & 15) == 0)
        return 3;
    return 0;
}

// plain operands, whole K-tiles, 32-bit byte offsets: the one-instruction DMA pieces of gemm256_body
static bool gemm256_fast_ok(const GemmArgs& g) {
    static const bool off = [] { const char* e = getenv("CUT3R_GEMM_FASTADDR"); return e && atoi(e) == 0; }();
    if (off || (g.K % BK) != 0 || (size_t)g.N * g.ldb * 2 >= 0xFFFF0000ull) return false;
    if (g.conv_k == 3) return g.Cin >= 64 && (g.Cin & (g.Cin - 1)) == 0 && g.K == 9 * g.Cin;
    return (size_t)g.M * g.lda * 2 < 0xFFFF0000ull;
}

__global__ __launch_bounds__(512) void gemm256_pair_kernel(const GemmPairArgs a) {
    const int sel = (int)blockIdx.x >= a.nblk0 ? 1 : 0;
    gemm256_body<false, false>(a.p[sel], blockIdx.x - sel * a.nblk0, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// Skinny GEMM on MFMA (M <= 64 rows): the pose-memory read blocks, pose MLP and proj_q run with one row per tracking
// window (M = window batch).  Weight-read-bound: a 64-row tile puts only N/64 workgroups on the chip.  Here one workgroup
// owns 16 output columns, its NW waves split K, every lane streams its 16 bytes of W per k-step straight from global
// memory into the MFMA B operand (no LDS) and reuses it for the MB = ceil(M/16) row blocks; the per-wave partial sums
// are added in wave order through LDS.  The summation order of a row depends on K only, never on M: results are
// identical for every window batch.
template <int NW, int MB>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_kernel(const GemmArgs g) {
    __shared__ f32x4 red[NW][MB][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int z = blockIdx.z;
    const h16* __restrict__ A = g.A + (size_t)z * g.sA;
    const h16* __restrict__ Bm = g.B + (size_t)z * g.sB;
    const int M = g.M, N = g.N, K = g.K;
    const int n0 = blockIdx.x * 16;
    const int ksteps = (K + 31) / 32;
    const int per = (ksteps + NW - 1) / NW;
    const int ks0 = wave * per, ks1 = min(ksteps, ks0 + per);
    const bool b_ok = n0 + fr < N;
    const h16* bp = Bm + (size_t)(b_ok ? n0 + fr : 0) * g.ldb + fq * 8;
    bool a_ok[MB];
    const h16* ap[MB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++) {
        a_ok[mb] = mb * 16 + fr < M;
        ap[mb] = A + (size_t)(a_ok[mb] ? mb * 16 + fr : 0) * g.lda + fq * 8;
    }
    const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int U = MB == 1 ? 8 : (MB == 2 ? 4 : 2);
    for (int ks = ks0; ks < ks1; ks += U) {
        half8_t fa[U][MB], fb[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = (ks + u) * 32 + fq * 8;
            const bool ok = (ks + u) < ks1 && k < K;
            fb[u] = (ok && b_ok) ? *reinterpret_cast<const half8_t*>(bp + (size_t)(ks + u) * 32) : zero8;
#pragma unroll
            for (int mb = 0; mb < MB; mb++)
                fa[u][mb] = (ok && a_ok[mb]) ? *reinterpret_cast<const half8_t*>(ap[mb] + (size_t)(ks + u) * 32) : zero8;
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int mb = 0; mb < MB; mb++) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[u][mb], fb[u], acc[mb], 0, 0, 0);
    }
#pragma unroll
    for (int mb = 0; mb < MB; mb++) red[wave][mb][lane] = acc[mb];
    __syncthreads();
    if (wave != 0) return;
    const int gn = n0 + fr;
    if (gn >= N) return;
    const float bias = g.bias ? g.bias[(size_t)z * g.sBias + gn] : 0.f;
#pragma unroll
    for (int mb = 0; mb < MB; mb++) {
        f32x4 v = red[0][mb][lane];
#pragma unroll
        for (int w = 1; w < NW; w++) v += red[w][mb][lane];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int gm = mb * 16 + fq * 4 + e;
            if (gm >= M) continue;
            float o = v[e] + bias;
            if (g.act == 1) o = g.out_f16 ? gelu_fast(o) : gelu_erf(o);
            else if (g.act == 2) o = fmaxf(o, 0.f);
            if (g.res1) o += g.res1_f16 ? (float)((const h16*)g.res1)[(size_t)z * g.sR1 + (size_t)gm * g.ldr1 + gn]
                                        : ((const float*)g.res1)[(size_t)z * g.sR1 + (size_t)gm * g.ldr1 + gn];
            if (g.res2) o += g.res2_f16 ? (float)((const h16*)g.res2)[(size_t)z * g.sR2 + (size_t)gm * g.ldr2 + gn]
                                        : ((const float*)g.res2)[(size_t)z * g.sR2 + (size_t)gm * g.ldr2 + gn];
            if (g.out_f16) ((h16*)g.C)[(size_t)z * g.sC + (size_t)gm * g.ldc + gn] = (h16)o;
            else ((float*)g.C)[(size_t)z * g.sC + (size_t)gm * g.ldc + gn] = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Skinny GEMM (M <= 8 rows): weight-read-bound GEMV.  One wave per output column block, x rows kept in registers.
// Used for the 1-token pose-memory read, pose MLP and adaLN modulation (M = 1).
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ X, int ldx, const h16* __restrict__ W, int ldw,
                                                   const float* __restrict__ bias, float* __restrict__ Y, int ldy, int M,
                                                   int N, int K, int act, const float* __restrict__ res, int ldr, int silu_in) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const h16* w = W + (size_t)n * ldw;
    for (int m = 0; m < M; m++) {
        float s = 0.f;
        for (int k = lane * 8; k < K; k += 64 * 8) {
            half8_t wv = *reinterpret_cast<const half8_t*>(w + k);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float x = X[(size_t)m * ldx + k + e];
                if (silu_in) x = x / (1.0f + __expf(-x));
                // match the MFMA path: activations are rounded to fp16 before the product
                s = fmaf((float)(h16)x, (float)wv[e], s);
            }
        }
        s = wave_sum(s);
        if (lane == 0) {
            if (bias) s += bias[n];
            if (act == 1) s = gelu_erf(s);
            if (res) s += res[(size_t)m * ldr + n];
            Y[(size_t)m * ldy + n] = s;
        }
    }
}

}  // namespace

static inline int batch_of(const cut3r_gemm_desc* d) { return d->batch > 0 ? d->batch : 1; }

extern "C" int cut3r_gemm_tile_for(const cut3r_gemm_desc* d) {
    if (!d) return 0;
    if (d->tile != 0) return d->tile;
    if (d->rope_pos && d->rope_cols && d->rope_d == 48) return 128192;      // the only tile whose width is a multiple of 48
    const int batch = d->batch > 0 ? d->batch : 1;
    const long long big_blocks = (long long)((d->M + 127) / 128) * ((d->N + 127) / 128) * batch;
    // measured (tools/bench_gemm256.py, tools/bench_gemm.py): the 256^2 ping-pong kernel wins once its grid fills the
    // chip (>= 200 tiles) on plain linears with N a multiple of 256; 3x3 convolutions and short grids stay on 128^2
    // (two co-resident workgroups hide each other's prologue / epilogue); 64^2 below 128 tiles of 128^2
    static const long long t256_min = [] { const char* e = getenv("CUT3R_GEMM_T256_MIN"); return e ? atoll(e) : 128LL; }();
    const long long blocks256 = (long long)((d->M + 255) / 256) * ((d->N + 255) / 256) * batch;
    // one workgroup per CU: a grid of 300 tiles costs two full rounds, so beyond one round require >= 85 % of the last one
    // (round 2, tools/bench_gemm_r2.py: from 128 tiles the 256^2 kernel beats 128^2 by 20-25 % on K = 768 / 1536 projections;
    //  at 72 tiles it loses by 50 %)
    const long long rounds = (blocks256 + 255) / 256;
    static const long long fill_pct = [] { const char* e = getenv("CUT3R_GEMM_T256_FILL"); return e ? atoll(e) : 85LL; }();
    const bool fills = rounds == 1 || blocks256 * 100 >= rounds * 256 * fill_pct;
    static const int conv256 = [] { const char* e = getenv("CUT3R_GEMM_CONV256"); return e ? atoi(e) : 1; }();
    if ((d->conv_k != 3 || conv256) && !d->shuf && (d->N & 255) == 0 && blocks256 >= t256_min && fills) return 256;
    // 192 x 128 (48 x 64 per wave: fewer LDS reads and L2->LDS bytes per FLOP than 128^2, still two workgroups per CU): round 2
    // A/B (tools/bench_gemm_r2.py, same box): +8-11 % on the DPT 3x3 convolutions with 128 output channels (head.0, head.2) and
    // on the 96x128 fusion convolutions; level on the M ~ 6k decoder projections, so it is the default for convolutions with
    // enough tiles to fill the chip and an explicit choice (tile = 192128, CUT3R_GEMM_T192_MIN_M) elsewhere
    static const long long t192_min = [] { const char* e = getenv("CUT3R_GEMM_T192_MIN_M"); return e ? atoll(e) : (1LL << 60); }();
    static const int conv192 = [] { const char* e = getenv("CUT3R_GEMM_CONV192"); return e ? atoi(e) : 1; }();
    const long long blocks192 = (long long)((d->M + 191) / 192) * ((d->N + 127) / 128) * batch;
    if (d->conv_k == 3 && conv192 && !d->shuf && blocks192 >= 512) return 192128;
    if (!d->shuf && d->M >= t192_min && big_blocks >= 128) return 192128;
    return (big_blocks >= 128) ? 128 : 64;
}

// validation + translation of a descriptor into kernel arguments (shared by the single and the pair entry point)
static int fill_args(const cut3r_gemm_desc* d, GemmArgs& g) {
    if (!d || !d->A || !d->B || !d->C) return CUT3R_ERR_ARG;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0) return CUT3R_ERR_ARG;
    if ((d->K & 7) || (d->N & 3) || (d->ldb & 7) || (d->ldc & 3)) return CUT3R_ERR_ARG;
    if (d->conv_k != 3 && (d->lda & 7)) return CUT3R_ERR_ARG;
    if (d->conv_k == 3 && ((d->Cin & 7) || d->K != 9 * d->Cin)) return CUT3R_ERR_ARG;
    if (((uintptr_t)d->A | (uintptr_t)d->B | (uintptr_t)d->C) & 15) return CUT3R_ERR_ARG;
    if (d->res1 && (d->ldr1 & 3)) return CUT3R_ERR_ARG;
    if (d->res2 && (d->ldr2 & 3)) return CUT3R_ERR_ARG;
    if (d->shuf && ((d->shuf_cout & 3) || d->N != d->shuf * d->shuf * d->shuf_cout)) return CUT3R_ERR_ARG;
    g.A = (const h16*)d->A; g.B = (const h16*)d->B; g.C = d->C;
    g.bias = d->bias; g.res1 = d->res1; g.res2 = d->res2;
    g.M = d->M; g.N = d->N; g.K = d->K; g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc; g.ldr1 = d->ldr1; g.ldr2 = d->ldr2;
    g.act = d->act; g.out_f16 = d->out_f16; g.res1_f16 = d->res1_f16; g.res2_f16 = d->res2_f16;
    g.sA = d->strideA; g.sB = d->strideB; g.sC = d->strideC; g.sBias = d->strideBias; g.sR1 = d->strideR1; g.sR2 = d->strideR2;
    g.conv_k = d->conv_k; g.H = d->H; g.W = d->W; g.Cin = d->Cin; g.cstride = d->conv_stride; g.Ho = d->Ho; g.Wo = d->Wo;
    g.relu_in = d->relu_in;
    g.shuf = d->shuf; g.shuf_cout = d->shuf_cout; g.shuf_Hin = d->shuf_Hin; g.shuf_Win = d->shuf_Win;
    g.swz = 0;
    g.prio = (d->stages == 12) ? 1 : 0;
    g.rope_pos = (const long long*)d->rope_pos; g.rope_table = d->rope_table;
    g.rope_cols = d->rope_pos ? d->rope_cols : 0; g.rope_pmin = d->rope_pmin; g.rope_npos = d->rope_npos;
    g.rope_d = d->rope_d ? d->rope_d : 64;
    // LayerNorm fold: consumer (ln_stats + ln_colsum) and producer (stats_out + out16) sides
    g.ln_stats = d->ln_stats; g.ln_c = d->ln_colsum; g.ln_nslab = d->ln_nslab; g.ln_eps = d->ln_eps;
    g.stats_out = d->stats_out; g.out16 = (h16*)d->out16; g.ld16 = d->ld16;
    if ((d->ln_stats != nullptr) != (d->ln_colsum != nullptr) || (d->stats_out != nullptr) != (d->out16 != nullptr)) return CUT3R_ERR_ARG;
    if (d->ln_stats) {
        if (!d->bias || d->conv_k == 3 || d->shuf || d->relu_in || batch_of(d) != 1 || d->ln_nslab * 64 != d->K || !(d->ln_eps > 0.f) ||
            ((uintptr_t)d->ln_stats & 7) || ((uintptr_t)d->ln_colsum & 15))
            return CUT3R_ERR_ARG;
    }
    if (d->stats_out) {
        if (d->out_f16 || !d->res1 || d->res1_f16 || d->res2 || d->act || d->conv_k == 3 || d->shuf || batch_of(d) != 1 || (d->N & 63) ||
            (d->ld16 & 7) || ((uintptr_t)d->out16 & 15) || ((uintptr_t)d->stats_out & 7))
            return CUT3R_ERR_ARG;
    }
    if (g.rope_cols) {
        // whole heads inside the N range and inside every tile (tile widths 64/128/256 hold 64-wide heads, 192 holds both
        // 64- and 48-wide heads), fp16 output only
        if (!d->rope_table || d->rope_npos < 1 || (g.rope_d != 64 && g.rope_d != 48) || (g.rope_cols % g.rope_d) || g.rope_cols > d->N ||
            !d->out_f16 || d->act || d->res1 || d->res2 || d->shuf || d->conv_k == 3 || (d->N % g.rope_d) || batch_of(d) != 1)
            return CUT3R_ERR_ARG;
        if (g.rope_d == 48 && cut3r_gemm_tile_for(d) != 128192) return CUT3R_ERR_ARG;     // 48-wide heads need the 192-column tile
        if (g.rope_d == 64 && (d->N & 63)) return CUT3R_ERR_ARG;
    }
    return CUT3R_OK;
}

extern "C" int cut3r_gemm_f16(const cut3r_gemm_desc* d, void* stream) {
    GemmArgs g;
    const int rc = fill_args(d, g);
    if (rc != CUT3R_OK) return rc;
    const int batch = d->batch > 0 ? d->batch : 1;
    hipStream_t s = (hipStream_t)stream;
    const int tile = cut3r_gemm_tile_for(d);
    if (tile == 16) {
        if (d->M > 64 || d->conv_k == 3 || d->shuf || d->relu_in || g.rope_cols || g.ln_stats || g.stats_out) return CUT3R_ERR_ARG;
        dim3 grid((d->N + 15) / 16, 1, batch);
        const int mb = (d->M + 15) / 16;            // 1..4 row blocks; K >= 2048 splits over 8 waves, else 4
#define CUT3R_SKINNY(NWV, MBV) hipLaunchKernelGGL((gemm_skinny_kernel<NWV, MBV>), grid, dim3(64 * NWV), 0, s, g)
        if (d->K >= 2048) { if (mb == 1) CUT3R_SKINNY(8, 1); else if (mb == 2) CUT3R_SKINNY(8, 2); else CUT3R_SKINNY(8, 4); }
        else { if (mb == 1) CUT3R_SKINNY(4, 1); else if (mb == 2) CUT3R_SKINNY(4, 2); else CUT3R_SKINNY(4, 4); }
#undef CUT3R_SKINNY
    } else if (tile == 256) {
        dim3 grid(((d->N + 255) / 256) * ((d->M + 255) / 256), 1, batch);
        const bool fast = gemm256_fast_ok(g);
        const int epi = fast ? gemm256_epi_mode(g) : 0;
        // the LayerNorm fold lives in the compile-time epilogues: consumer 1 / 2 / 6, producer 3
        if (g.ln_stats && !(fast && d->conv_k != 3 && !d->relu_in && (epi == 1 || epi == 2 || epi == 6))) return CUT3R_ERR_ARG;
        if (g.stats_out && !(fast && d->conv_k != 3 && !d->relu_in && epi == 3)) return CUT3R_ERR_ARG;
        if (d->conv_k == 3 && d->relu_in && fast && epi == 4) hipLaunchKernelGGL((gemm256_kernel<true, true, true, 4>), grid, dim3(512), 0, s, g);
        else if (d->conv_k == 3 && !d->relu_in && fast && epi == 5) hipLaunchKernelGGL((gemm256_kernel<true, false, true, 5>), grid, dim3(512), 0, s, g);
        else if (d->conv_k == 3 && !d->relu_in && fast && epi == 1) hipLaunchKernelGGL((gemm256_kernel<true, false, true, 1>), grid, dim3(512), 0, s, g);
        else if (d->conv_k == 3 && !d->relu_in && fast && epi == 4) hipLaunchKernelGGL((gemm256_kernel<true, false, true, 4>), grid, dim3(512), 0, s, g);
        else if (d->conv_k == 3 && d->relu_in && fast) hipLaunchKernelGGL((gemm256_kernel<true, true, true>), grid, dim3(512), 0, s, g);
        else if (d->conv_k == 3 && fast) hipLaunchKernelGGL((gemm256_kernel<true, false, true>), grid, dim3(512), 0, s, g);
        else if (d->conv_k == 3 && d->relu_in) hipLaunchKernelGGL((gemm256_kernel<true, true>), grid, dim3(512), 0, s, g);
        else if (d->conv_k == 3) hipLaunchKernelGGL((gemm256_kernel<true, false>), grid, dim3(512), 0, s, g);
        else if (d->relu_in) hipLaunchKernelGGL((gemm256_kernel<false, true>), grid, dim3(512), 0, s, g);
        else if (fast && g.ln_stats) {
            switch (epi) {
                case 1: hipLaunchKernelGGL((gemm256_kernel<false, false, true, 1, true>), grid, dim3(512), 0, s, g); break;
                case 2: hipLaunchKernelGGL((gemm256_kernel<false, false, true, 2, true>), grid, dim3(512), 0, s, g); break;
                default: hipLaunchKernelGGL((gemm256_kernel<false, false, true, 6, true>), grid, dim3(512), 0, s, g);
            }
        }
        else if (fast) {
            switch (epi) {
                case 1: hipLaunchKernelGGL((gemm256_kernel<false, false, true, 1>), grid, dim3(512), 0, s, g); break;
                case 2: hipLaunchKernelGGL((gemm256_kernel<false, false, true, 2>), grid, dim3(512), 0, s, g); break;
                case 3: hipLaunchKernelGGL((gemm256_kernel<false, false, true, 3>), grid, dim3(512), 0, s, g); break;
                case 6: hipLaunchKernelGGL((gemm256_kernel<false, false, true, 6>), grid, dim3(512), 0, s, g); break;
                default: hipLaunchKernelGGL((gemm256_kernel<false, false, true>), grid, dim3(512), 0, s, g);
            }
        }
        else hipLaunchKernelGGL((gemm256_kernel<false, false>), grid, dim3(512), 0, s, g);
    } else if (tile == 128) {
        dim3 grid((d->N + 127) / 128, (d->M + 127) / 128, batch);
        static const long long swz_min = [] { const char* e = getenv("CUT3R_GEMM_SWZ_MIN"); return e ? atoll(e) : 200LL; }();
        if (d->stages == 13 || (d->stages == 0 && (long long)grid.x * grid.y >= swz_min)) {   // XCD-aware rasterisation
            g.swz = 1;
            grid = dim3(grid.x * grid.y, 1, batch);
        }
        if (g.stats_out && d->stages != 0) return CUT3R_ERR_ARG;
        if (d->stages == 3) hipLaunchKernelGGL((gemm_kernel<128, 128, 3>), grid, dim3(256), 0, s, g);
        else if (d->stages == 8) hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 2, 4>), grid, dim3(512), 0, s, g);   // 8 waves
        else if (d->stages == 9) hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 4, 2>), grid, dim3(512), 0, s, g);   // 8 waves
        else if (d->stages == 10) hipLaunchKernelGGL((gemm_kernel<128, 128, 3, 4, 2>), grid, dim3(512), 0, s, g);
        else if (d->stages == 14) hipLaunchKernelGGL((gemm_kernel<128, 128, 4, 4, 2>), grid, dim3(512), 0, s, g);      // 128 KiB ring, one workgroup per CU
        else if (d->stages == 4) hipLaunchKernelGGL((gemm_kernel<128, 128, 2>), grid, dim3(256), 0, s, g);
        else {
            const int am = tile_addr_mode(g), ep = gemm256_epi_mode(g);
            if (g.stats_out && !(am == 1 && ep == 3)) return CUT3R_ERR_ARG;      // the producer side lives in the compile-time epilogue 3
            if (am == 1 && ep == 1) hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 4, 2, 1, 1>), grid, dim3(512), 0, s, g);
            else if (am == 1 && ep == 2) hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 4, 2, 1, 2>), grid, dim3(512), 0, s, g);
            else if (am == 1 && ep == 3) hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 4, 2, 1, 3>), grid, dim3(512), 0, s, g);
            else if (am == 1) hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 4, 2, 1>), grid, dim3(512), 0, s, g);
            else if (am == 2) hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 4, 2, 2>), grid, dim3(512), 0, s, g);
            else hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 4, 2>), grid, dim3(512), 0, s, g);
        }
    } else if (g.stats_out && tile != 64) {
        return CUT3R_ERR_ARG;              // producer side of the LayerNorm fold: tiles 256, 128 and 64 only
    } else if (tile == 128192) {       // 128 x 192 (four 48-wide or three 64-wide heads per tile), 8 waves (32 x 96 per wave), 80 KB LDS
        dim3 grid(((d->N + 191) / 192) * ((d->M + 127) / 128), 1, batch);
        g.swz = 1;
        hipLaunchKernelGGL((gemm_kernel<128, 192, 2, 4, 2>), grid, dim3(512), 0, s, g);
    } else if (tile == 192128) {       // 192 x 128, 8 waves (48 x 64 per wave), 80 KB LDS: two workgroups per CU
        dim3 grid(((d->N + 127) / 128) * ((d->M + 191) / 192), 1, batch);
        g.swz = 1;
        if (d->stages == 3) hipLaunchKernelGGL((gemm_kernel<192, 128, 3, 4, 2>), grid, dim3(512), 0, s, g);                 // 120 KiB ring
        else {
            const int am = tile_addr_mode(g), ep = gemm256_epi_mode(g);
            if (am == 2 && ep == 1) hipLaunchKernelGGL((gemm_kernel<192, 128, 2, 4, 2, 2, 1>), grid, dim3(512), 0, s, g);
            else if (am == 2 && ep == 4) hipLaunchKernelGGL((gemm_kernel<192, 128, 2, 4, 2, 2, 4>), grid, dim3(512), 0, s, g);
            else if (am == 1) hipLaunchKernelGGL((gemm_kernel<192, 128, 2, 4, 2, 1>), grid, dim3(512), 0, s, g);
            else if (am == 2) hipLaunchKernelGGL((gemm_kernel<192, 128, 2, 4, 2, 2>), grid, dim3(512), 0, s, g);
            else hipLaunchKernelGGL((gemm_kernel<192, 128, 2, 4, 2>), grid, dim3(512), 0, s, g);
        }
    } else if (tile == 256128) {
        dim3 grid((d->N + 127) / 128, (d->M + 255) / 256, batch);
        if (d->stages == 3) hipLaunchKernelGGL((gemm_kernel<256, 128, 3, 4, 2>), grid, dim3(512), 0, s, g);                 // 144 KiB ring
        else hipLaunchKernelGGL((gemm_kernel<256, 128, 2, 4, 2>), grid, dim3(512), 0, s, g);
    } else if (tile == 12864) {
        dim3 grid((d->N + 63) / 64, (d->M + 127) / 128, batch);
        if (d->stages == 3) hipLaunchKernelGGL((gemm_kernel<128, 64, 3>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((gemm_kernel<128, 64, 2>), grid, dim3(256), 0, s, g);
    } else if (tile == 64) {
        dim3 grid((d->N + 63) / 64, (d->M + 63) / 64, batch);
        if (g.stats_out && d->stages != 0) return CUT3R_ERR_ARG;
        if (d->stages == 2) hipLaunchKernelGGL((gemm_kernel<64, 64, 2>), grid, dim3(256), 0, s, g);
        else if (d->stages == 8 || (d->stages == 0 && d->K >= 2048)) {   // long K: 8 waves
            const int am = d->stages == 0 ? tile_addr_mode(g) : 0;
            const int ep = am == 1 ? gemm256_epi_mode(g) : 0;
            if (g.stats_out && ep != 3) return CUT3R_ERR_ARG;
            // ring depth (CUT3R_GEMM64_STAGES): these launches put <= 2 workgroups on a CU and wait on HBM for their weight panels
            static const int ns8 = [] { const char* e = getenv("CUT3R_GEMM64_STAGES"); return e ? atoi(e) : 3; }();
            if (ep == 3 && ns8 == 6) hipLaunchKernelGGL((gemm_kernel<64, 64, 6, 2, 4, 1, 3>), grid, dim3(512), 0, s, g);
            else if (ep == 3 && ns8 == 4) hipLaunchKernelGGL((gemm_kernel<64, 64, 4, 2, 4, 1, 3>), grid, dim3(512), 0, s, g);
            else if (ep == 3) hipLaunchKernelGGL((gemm_kernel<64, 64, 3, 2, 4, 1, 3>), grid, dim3(512), 0, s, g);     // fc2 + residual at one window
            else if (am == 1) hipLaunchKernelGGL((gemm_kernel<64, 64, 3, 2, 4, 1>), grid, dim3(512), 0, s, g);
            else if (am == 2) hipLaunchKernelGGL((gemm_kernel<64, 64, 3, 2, 4, 2>), grid, dim3(512), 0, s, g);
            else hipLaunchKernelGGL((gemm_kernel<64, 64, 3, 2, 4>), grid, dim3(512), 0, s, g);
        }
        else if (d->stages == 4) hipLaunchKernelGGL((gemm_kernel<64, 64, 4>), grid, dim3(256), 0, s, g);
        else {
            const int am = d->stages == 0 ? tile_addr_mode(g) : 0;
            // compile-time epilogues for the one-window (M = 769) Linear layers: fp16 + bias, + GELU, fp32 + bias + fp32 residual
            const int ep = am == 1 ? gemm256_epi_mode(g) : 0;
            if (g.stats_out && ep != 3) return CUT3R_ERR_ARG;
            static const int ns4 = [] { const char* e = getenv("CUT3R_GEMM64_STAGES"); return e ? atoi(e) : 3; }();
#define CUT3R_T64(NS)                                                                                                   \
    do {                                                                                                                 \
        if (ep == 1) hipLaunchKernelGGL((gemm_kernel<64, 64, NS, 2, 2, 1, 1>), grid, dim3(256), 0, s, g);                \
        else if (ep == 2) hipLaunchKernelGGL((gemm_kernel<64, 64, NS, 2, 2, 1, 2>), grid, dim3(256), 0, s, g);           \
        else if (ep == 3) hipLaunchKernelGGL((gemm_kernel<64, 64, NS, 2, 2, 1, 3>), grid, dim3(256), 0, s, g);           \
        else hipLaunchKernelGGL((gemm_kernel<64, 64, NS, 2, 2, 1>), grid, dim3(256), 0, s, g);                           \
    } while (0)
            if (am == 1 && ns4 == 6) CUT3R_T64(6);
            else if (am == 1 && ns4 == 4) CUT3R_T64(4);
            else if (am == 1) CUT3R_T64(3);
#undef CUT3R_T64
            else if (am == 2) hipLaunchKernelGGL((gemm_kernel<64, 64, 3, 2, 2, 2>), grid, dim3(256), 0, s, g);
            else hipLaunchKernelGGL((gemm_kernel<64, 64, 3>), grid, dim3(256), 0, s, g);
        }
    } else {
        return CUT3R_ERR_ARG;
    }
    return cut3r_check_launch();
}

extern "C" int cut3r_gemm_f16_pair(const cut3r_gemm_desc* d0, const cut3r_gemm_desc* d1, void* stream) {
    if (!d0 || !d1) return CUT3R_ERR_ARG;
    GemmPairArgs a;
    int rc = fill_args(d0, a.p[0]);
    if (rc != CUT3R_OK) return rc;
    rc = fill_args(d1, a.p[1]);
    if (rc != CUT3R_OK) return rc;
    // linears only, one problem each (no batch), same N and K so both take the same tile kernel
    for (const cut3r_gemm_desc* d : {d0, d1})
        if (d->conv_k == 3 || d->shuf || d->relu_in || (d->batch > 1)) return CUT3R_ERR_ARG;
    if (d0->N != d1->N || d0->K != d1->K) return CUT3R_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const bool special = d0->ln_stats || d1->ln_stats || d0->stats_out || d1->stats_out || a.p[0].rope_cols || a.p[1].rope_cols;
    // tile choice on the COMBINED grid (the two problems fill the chip together)
    int tile = d0->tile;
    if (tile == 0) {
        const long long t256 = (long long)((d0->M + 255) / 256 + (d1->M + 255) / 256) * ((d0->N + 255) / 256);
        const long long rounds = (t256 + 255) / 256;
        const bool fills = rounds == 1 || t256 * 100 >= rounds * 256 * 85;
        const long long t128 = (long long)((d0->M + 127) / 128 + (d1->M + 127) / 128) * ((d0->N + 127) / 128);
        tile = ((d0->N & 255) == 0 && t256 >= 128 && fills && !special) ? 256 : ((t128 >= 128 && !special) ? 128 : 64);
    }
    if (special && tile != 64) return CUT3R_ERR_ARG;          // fused RoPE / LayerNorm fold ride the 64 x 64 pair kernels (the one-window schedule)
    if (tile == 64) {
        // the one-window (M = 768 / 769) decoder: state-side + image-side projection of a layer in ONE launch of 2 x 156 tiles.  Compile-time
        // epilogue when both problems qualify for the same one (fc1: 2, residual projections: 3 -- also the LayerNorm fold's producer), else
        // the run-time epilogue (fused RoPE of the image side next to the plain state side; LayerNorm fold consumer in either).
        const int n0 = ((d0->N + 63) / 64) * ((d0->M + 63) / 64), n1 = ((d1->N + 63) / 64) * ((d1->M + 63) / 64);
        a.nblk0 = n0;
        a.p[0].swz = a.p[1].swz = 1;             // 1-D grid: tile (pid_m, pid_n) from the XCD-aware rasterisation of each problem
        const bool am = tile_addr_mode(a.p[0]) == 1 && tile_addr_mode(a.p[1]) == 1;
        const int e0 = gemm256_epi_mode(a.p[0]), e1 = gemm256_epi_mode(a.p[1]);
        const int ep = (am && e0 == e1 && (e0 == 1 || e0 == 2 || e0 == 3)) ? e0 : 0;
        if ((d0->stats_out || d1->stats_out) && ep != 3) return CUT3R_ERR_ARG;
        const dim3 grid(n0 + n1);
        if (d0->K >= 2048) {
            if (ep == 3) hipLaunchKernelGGL((gemm_pair_kernel<64, 64, 3, 2, 4, 1, 3>), grid, dim3(512), 0, s, a);
            else if (am) hipLaunchKernelGGL((gemm_pair_kernel<64, 64, 3, 2, 4, 1, 0>), grid, dim3(512), 0, s, a);
            else hipLaunchKernelGGL((gemm_pair_kernel<64, 64, 3, 2, 4, 0, 0>), grid, dim3(512), 0, s, a);
        } else {
            if (ep == 1) hipLaunchKernelGGL((gemm_pair_kernel<64, 64, 3, 2, 2, 1, 1>), grid, dim3(256), 0, s, a);
            else if (ep == 2) hipLaunchKernelGGL((gemm_pair_kernel<64, 64, 3, 2, 2, 1, 2>), grid, dim3(256), 0, s, a);
            else if (ep == 3) hipLaunchKernelGGL((gemm_pair_kernel<64, 64, 3, 2, 2, 1, 3>), grid, dim3(256), 0, s, a);
            else if (am) hipLaunchKernelGGL((gemm_pair_kernel<64, 64, 3, 2, 2, 1, 0>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((gemm_pair_kernel<64, 64, 3, 2, 2, 0, 0>), grid, dim3(256), 0, s, a);
        }
    } else if (tile == 256) {
        const int n0 = ((d0->N + 255) / 256) * ((d0->M + 255) / 256), n1 = ((d1->N + 255) / 256) * ((d1->M + 255) / 256);
        a.nblk0 = n0;
        hipLaunchKernelGGL(gemm256_pair_kernel, dim3(n0 + n1), dim3(512), 0, s, a);
    } else if (tile == 128) {
        const int n0 = ((d0->N + 127) / 128) * ((d0->M + 127) / 128), n1 = ((d1->N + 127) / 128) * ((d1->M + 127) / 128);
        a.nblk0 = n0;
        a.p[0].swz = a.p[1].swz = 1;
        hipLaunchKernelGGL((gemm_pair_kernel<128, 128, 2, 4, 2>), dim3(n0 + n1), dim3(512), 0, s, a);
    } else if (tile == 192128) {
        const int n0 = ((d0->N + 127) / 128) * ((d0->M + 191) / 192), n1 = ((d1->N + 127) / 128) * ((d1->M + 191) / 192);
        a.nblk0 = n0;
        a.p[0].swz = a.p[1].swz = 1;
        hipLaunchKernelGGL((gemm_pair_kernel<192, 128, 2, 4, 2>), dim3(n0 + n1), dim3(512), 0, s, a);
    } else {
        return CUT3R_ERR_ARG;
    }
    return cut3r_check_launch();
}

extern "C" int cut3r_gemv_f16w(const float* X, int ldx, const void* W, int ldw, const float* bias, float* Y, int ldy, int M,
                               int N, int K, int act, const float* res, int ldr, int silu_in, void* stream) {
    if (!X || !W || !Y || M <= 0 || M > 64 || N <= 0 || K <= 0 || (K & 7) || (ldw & 7)) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(gemv_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, X, ldx, (const h16*)W, ldw, bias, Y, ldy,
                       M, N, K, act, res, ldr, silu_in);
    return cut3r_check_launch();
}
