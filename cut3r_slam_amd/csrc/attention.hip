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
// attn_kernel is this structure (all head widths; the only form for 16, 32 and 128); attn_pipe_kernel below is the software-pipelined
// LDS-DMA form that serves the 48- and 64-wide heads by default.
#include <atomic>
#include <cstdlib>
#include <type_traits>
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


// =====================================================================================================================
// attn_pipe_kernel -- the pipelined form (round 4; the default for 48- and 64-wide heads).
//
//   * workgroup = NW waves = one (batch, head) and 32 NW QB query rows; a wave owns QB 32-row query blocks.  Built and used:
//     NW = 4, QB = 1, NST = 4 -- 214 registers, two workgroups per CU (two waves per SIMD), 64 KiB of LDS each.  (QB = 2 with one
//     wave per SIMD, the 512-register form of the guide, was written and measured: hipcc parks half the state in AGPRs and pays
//     ~270 v_accvgpr copies per step, 66 TF/s; NW = 8 with an 8-deep ring: 632 TF/s against 663 -- profiles/r04/attn_pipe_bench.txt.)
//   * K/V tiles (64 keys) arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write) into a ring of NST stages;
//     ONE raw s_barrier per tile and counted vmcnt: NST-4 tiles stay in flight across it.  A key row is 128 B in LDS (eight
//     16-byte chunks; a 48-wide head fills six, the other two stay zero), DMA pieces are linear (1 KiB = 8 rows), and the
//     bank-conflict swizzles are applied to the SOURCE chunk of a lane: K chunk ^= (row >> 1) & 7 (16 rows of a ds_read_b128 phase
//     on 16 distinct 16-byte bank groups), V chunk ^= 4 ((row >> 1) & 1) (the 4-row x 64-byte block of a ds_read_b64_tr_b16 phase
//     on four distinct 64-byte segments);
//   * step j of the tile loop holds three INDEPENDENT instruction streams in one basic block:
//         matrix   O += V(j-1)^T P(j-1)^T   then   S(j+1)^T = K(j+1) Q^T
//         vector   online softmax of S(j) -> P(j)  (fp32: row maximum, exp2, row sum, fp16 conversion)
//     interleaved by sched_group_barrier (one MFMA, two fragment reads for the MFMA three further on, nine VALU), so a wave's
//     softmax runs in the shadow of its own MFMAs.  S and P are double buffered by tile parity; the rescale of O by alpha(j) sits
//     between PV(j-1) and PV(j), as in attn_kernel;
//   * the nblk query blocks of one (batch, head) run on ONE XCD, one dispatch round apart (1-D grid, see the id decoding);
//   * per row the arithmetic is attn_kernel's, operation for operation and in the same order: both kernels give the SAME BITS
//     (tests/test_kernels_gpu.py), which keeps a window's result independent of how many windows share the launch.
//   What bounds it (tools/memtime_attn_pipe.py, profiles/r04/memtime_attn_pipe.txt): a step is 148 VALU instructions = 852 issue
//   cycles of its SIMD (33 v_exp at 8, the rest at 4, 16 MFMAs holding the port for 8 each) against 512 cycles of matrix pipe; two
//   waves share the SIMD's vector port, and the measured 1480 cycles per step and SIMD are 58 % of that VALU floor -- the matrix pipe
//   cannot be more than 60 % busy at this head width whatever the schedule.
// One LDS-DMA piece (64 lanes x 16 B -> 1 KiB of LDS at `lds`, linear), issued as inline asm: the compiler must not know that this
// is a write to LDS -- it would put s_waitcnt vmcnt(0) in front of the next ds_read_b64_tr_b16 (it cannot tell that the read targets
// another ring slot), i.e. wait for the tile it has just requested.  The vector-memory counter of this kernel is kept by hand.
DEVINL void dma_piece(const char* gbase /* wave-uniform */, unsigned voff, unsigned lds /* wave-uniform */) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(gbase), "s"(lds) : "memory", "m0");
}
DEVINL unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

template <int N>
DEVINL void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// tiles_ahead tiles of P DMA instructions each may stay in flight
template <int P, int MAXT>
DEVINL void wait_tiles_ahead(int tiles_ahead) {
    if constexpr (MAXT == 0) wait_vm<0>();
    else {
        if (tiles_ahead >= MAXT) wait_vm<P * MAXT>();
        else wait_tiles_ahead<P, MAXT - 1>(tiles_ahead);
    }
}

template <int D, int QB, int NST, int NW>
__global__ __launch_bounds__(64 * NW, (QB == 1 ? 2 : 1)) void attn_pipe_kernel(const AttnArgs a, const int nblk, const int HB, const int H) {
    constexpr int KT = 64;
    constexpr int PPW = 8 / NW;                  // DMA pieces per wave, tile and image (a K or V image = 8 pieces of 1 KiB)
    constexpr int AHEAD = NST - 4;               // tiles that may still be in flight when a step starts
    constexpr int DQ = D / 16;                   // k-steps of the QK^T product
    constexpr int DP = (D + 31) / 32;            // 32-row d-tiles of the PV product
    constexpr int CH = D / 8;                    // 16-byte chunks of a key row that carry data
    constexpr int VOFF = KT * 128;               // V image behind the K image of a stage
    constexpr int STAGE = 2 * KT * 128;
    static_assert(D % 16 == 0 && D <= 64 && NST >= 4 && (NW == 4 || NW == 8), "head dims 16..64; the ring holds tiles j-1 .. j+2 and NST-4 more in flight");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    // workgroup L runs on XCD L % 8: the nblk query blocks of one (batch, head) get ids 8 (nblk g + x) + c -- the same XCD, one
    // dispatch round apart -- so that its K/V stream is fetched into ONE L2 and read nblk times from there
    const int L = blockIdx.x;
    const int xb = (L >> 3) % nblk;
    const int bh = ((L >> 3) / nblk) * 8 + (L & 7);
    if (bh >= HB) return;                        // (the grid is padded to whole groups of 8 (batch, head) pairs)
    const int h = bh % H, b = bh / H;
    const int q0 = xb * (32 * NW * QB) + wave * (32 * QB);
    const h16* qp = a.q + (size_t)b * a.q_sb + (size_t)h * D;
    const h16* kp = a.k + (size_t)b * a.k_sb + (size_t)h * D;
    const h16* vp = a.v + (size_t)b * a.v_sb + (size_t)h * D;
    const bool active = q0 < a.Nq;               // wave-uniform; an idle wave still carries its share of the DMA and the barriers
    const int extra = (a.Nk > KT && (a.Nk % KT) == 1) ? 1 : 0;
    const int ntiles = (a.Nk - extra + KT - 1) / KT;

    if (CH < 8) {                                // the chunks no DMA lane ever writes: zero, once, before the first piece lands
        for (int i = tid; i < NST * STAGE / 16; i += 64 * NW) *reinterpret_cast<f32x4*>(smem + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
    }

    // ---- this lane's part of a tile's DMA: pieces PPW wave .. PPW wave + PPW - 1 of the K image and of the V image.  Source = wave-uniform tile base
    // + a 32-bit per-lane byte offset (row of the piece, swizzled chunk): two registers per image for the whole loop
    auto lane_src = [&](int c, int row_limit, unsigned& ko, unsigned& vo, bool& kval, bool& vval) {
        const int row = (wave * PPW + c) * 8 + (lane >> 3), slot = lane & 7;
        const int kc = slot ^ ((row >> 1) & 7), vc = slot ^ (((row >> 1) & 1) << 2);
        const int rr = min(row, row_limit);      // ragged last tile: keys past the end read the last key (finite values; their scores are masked)
        ko = (unsigned)rr * (unsigned)(a.k_sn * 2) + kc * 16;
        vo = (unsigned)rr * (unsigned)(a.v_sn * 2) + vc * 16;
        kval = kc < CH;
        vval = vc < CH;
    };
    unsigned koffb[PPW], voffb[PPW];
#pragma unroll
    for (int c = 0; c < PPW; c++) {
        bool kv_, vv_;
        lane_src(c, KT - 1, koffb[c], voffb[c], kv_, vv_);
    }
    const char* ktile = reinterpret_cast<const char*>(kp + (size_t)extra * a.k_sn);     // wave-uniform: key 0 of the next tile to request
    const char* vtile = reinterpret_cast<const char*>(vp + (size_t)extra * a.v_sn);
    int st_issue = 0;                            // ring slot of the next tile to request (tiles are requested in order)
    auto issue = [&](int t) {
        if (t >= ntiles) return;                 // wave-uniform
        unsigned char* st = smem + st_issue * STAGE;
        st_issue = st_issue + 1 == NST ? 0 : st_issue + 1;
        const int left = a.Nk - (extra + t * KT);            // keys of this tile that exist
#pragma unroll
        for (int c = 0; c < PPW; c++) {
            unsigned ko = koffb[c], vo = voffb[c];
            bool kval = true, vval = true;
            if (CH < 8 || left < KT) lane_src(c, left < KT ? left - 1 : KT - 1, ko, vo, kval, vval);
            const unsigned dk = __builtin_amdgcn_readfirstlane(lds_addr(st) + (wave * PPW + c) * 1024);
            if (CH == 8 || kval) dma_piece(ktile, ko, dk);
            if (CH == 8 || vval) dma_piece(vtile, vo, dk + VOFF);
        }
        ktile += (size_t)KT * a.k_sn * 2;
        vtile += (size_t)KT * a.v_sn * 2;
    };

    if (!active) {
        // a wave without query rows (the tail of the last 128 QB-row block) carries its share of the DMA and meets every barrier: the
        // synchronisation skeleton of the loop below, nothing else
#pragma unroll
        for (int t = 0; t < NST - 3; t++) issue(t);
        for (int j = -1; j + 1 < ntiles; j++) {
            wait_tiles_ahead<2 * PPW, AHEAD>(ntiles - 2 - j);
            asm volatile("s_barrier" ::: "memory");
            issue(j + NST - 2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ---- Q fragments (B operand: element j = Q[row][16 s + 8 hh + j]) and the running state of the QB query blocks
    half8_t qf[QB][DQ];
    f32x16 ot[QB][DP];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
        int qr = q0 + qb * 32 + r;
        if (qr > a.Nq - 1) qr = a.Nq - 1;
        const h16* qrow = qp + (size_t)qr * a.q_sn;
#pragma unroll
        for (int s = 0; s < DQ; s++) qf[qb][s] = *reinterpret_cast<const half8_t*>(qrow + 16 * s + 8 * hh);
#pragma unroll
        for (int d = 0; d < DP; d++)
#pragma unroll
            for (int i = 0; i < 16; i++) ot[qb][d][i] = 0.f;
        m_run[qb] = -INFINITY;
        l_run[qb] = 0.f;
        if (extra) {                             // key 0 first (see attn_kernel)
            float dot = 0.f;
#pragma unroll
            for (int s = 0; s < DQ; s++) {
                const half8_t k0 = *reinterpret_cast<const half8_t*>(kp + 16 * s + 8 * hh);
#pragma unroll
                for (int j = 0; j < 8; j++) dot = fmaf((float)qf[qb][s][j], (float)k0[j], dot);
            }
            { float d0, d1; halves(dot, d0, d1); dot = d0 + d1; }
            m_run[qb] = dot * a.scale_log2;
            l_run[qb] = hh == 0 ? 1.f : 0.f;
#pragma unroll
            for (int d = 0; d < DP; d++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const int dd = d * 32 + 8 * g4 + 4 * hh;
                    if (dd < D) {
                        const half4_t v0 = *reinterpret_cast<const half4_t*>(vp + dd);
#pragma unroll
                        for (int e = 0; e < 4; e++) ot[qb][d][4 * g4 + e] = (float)v0[e];
                    }
                }
        }
    }
    // Q (and key 0) are in registers before the first DMA piece is requested: from here on vmcnt counts DMA pieces only.  (The empty
    // asm statements USE the fragments: the compiler puts its own wait for the loads here instead of in front of the first MFMA of
    // every loop block, where it would also wait for the DMA pieces it knows nothing about.)
#pragma unroll
    for (int qb = 0; qb < QB; qb++)
#pragma unroll
        for (int s = 0; s < DQ; s++) asm volatile("" ::"v"(qf[qb][s]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- lane constants of the fragment reads
    int koff[DQ];                                // K: row kt2*32 + r, chunk 2 s + hh
#pragma unroll
    for (int s = 0; s < DQ; s++) koff[s] = r * 128 + (((2 * s + hh) ^ ((r >> 1) & 7)) << 4);
    const int g16 = lane >> 4, j16 = lane & 15;
    int voff[DP];                                // V: rows 4 (g16 >> 1) + (j16 >> 2) of a 16-key step, columns d*32 + 16 (g16 & 1) + 4 (j16 & 3)
#pragma unroll
    for (int d = 0; d < DP; d++) {
        const int row = 4 * (g16 >> 1) + (j16 >> 2);
        const int chunk = d * 4 + 2 * (g16 & 1) + ((j16 & 3) >> 1);
        voff[d] = VOFF + row * 128 + ((chunk ^ (((row >> 1) & 1) << 2)) << 4) + (j16 & 1) * 8;
    }

    f32x16 S[2][QB][2];                          // [tile parity][query block][32-key half]
    half8_t P[2][QB][2][2];                      // [tile parity][query block][32-key half][16-key step]
    float alpha[QB];
    bool grew = false;

    // one step: PAR = j & 1; HASC: PV of tile j-1; HASA: QK^T of tile j+1; HASB: softmax of tile j (MASK: its keys past Nk)
    auto step = [&](auto PAR_, auto HASC_, auto HASA_, auto HASB_, auto MASK_, int j) {
        constexpr int PAR = decltype(PAR_)::value;
        constexpr bool HASC = decltype(HASC_)::value, HASA = decltype(HASA_)::value, HASB = decltype(HASB_)::value, MASK = decltype(MASK_)::value;
        // every fragment read of the previous step has returned; tile j+1 has landed (mine: vmcnt; everyone's: the barrier); after the
        // barrier nobody reads tile j-2 any more: its slot takes tile j-2+NST
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (HASA) {
            wait_tiles_ahead<2 * PPW, AHEAD>(ntiles - 2 - j);      // tile j+1 has landed; tiles j+2 .. min(j+NST-3, ntiles-1) may be in flight
            asm volatile("s_barrier" ::: "memory");
            issue(j + NST - 2);
        }
        half8_t vf[2][2][DP], kf[2][DQ];
        if (HASC) {
            const unsigned char* st = smem + ((j - 1) % NST) * STAGE;
#pragma unroll
            for (int kt2 = 0; kt2 < 2; kt2++)
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                    for (int d = 0; d < DP; d++) {
                        const unsigned char* vq = st + (kt2 * 32 + 16 * s2) * 128 + voff[d];
                        const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(vq));
                        const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(vq + 8 * 128));
                        __builtin_memcpy(&vf[kt2][s2][d], &lo, 8);
                        __builtin_memcpy(reinterpret_cast<char*>(&vf[kt2][s2][d]) + 8, &hi, 8);
                    }
        }
        if (HASA) {
            const unsigned char* st = smem + ((j + 1) % NST) * STAGE;
#pragma unroll
            for (int kt2 = 0; kt2 < 2; kt2++)
#pragma unroll
                for (int s = 0; s < DQ; s++) kf[kt2][s] = *reinterpret_cast<const half8_t*>(st + kt2 * 32 * 128 + koff[s]);
        }
        if (HASC) {
#pragma unroll
            for (int kt2 = 0; kt2 < 2; kt2++)
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                    for (int d = 0; d < DP; d++)
#pragma unroll
                        for (int qb = 0; qb < QB; qb++)
                            ot[qb][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[kt2][s2][d], P[PAR ^ 1][qb][kt2][s2], ot[qb][d], 0, 0, 0);
        }
        if (HASA) {
#pragma unroll
            for (int kt2 = 0; kt2 < 2; kt2++)
#pragma unroll
                for (int qb = 0; qb < QB; qb++) {
#pragma unroll
                    for (int i = 0; i < 16; i++) S[PAR ^ 1][qb][kt2][i] = 0.f;
#pragma unroll
                    for (int s = 0; s < DQ; s++)
                        S[PAR ^ 1][qb][kt2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kt2][s], qf[qb][s], S[PAR ^ 1][qb][kt2], 0, 0, 0);
                }
        }
        if (HASB) {
            const int kbase = extra + j * KT;
            grew = false;
#pragma unroll
            for (int qb = 0; qb < QB; qb++) {
                f32x16(&st)[2] = S[PAR][qb];
                float mloc = -INFINITY;
                if (!MASK) {
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
                { float m0_, m1_; halves(mloc, m0_, m1_); mloc = fmaxf(m0_, m1_) * a.scale_log2; }
                const float m_new = fmaxf(m_run[qb], mloc);
                grew = grew || (m_new > m_run[qb]);
                alpha[qb] = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
                m_run[qb] = m_new;
                float lsum = 0.f;
#pragma unroll
                for (int kt2 = 0; kt2 < 2; kt2++)
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(st[kt2][i], a.scale_log2, -m_new));
                        st[kt2][i] = p;
                        lsum += p;
                    }
                l_run[qb] = l_run[qb] * alpha[qb] + lsum;
#pragma unroll
                for (int kt2 = 0; kt2 < 2; kt2++)
#pragma unroll
                    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                        for (int jj = 0; jj < 8; jj++) P[PAR][qb][kt2][s2][jj] = (h16)st[kt2][8 * s2 + jj];
            }
            // the rescale of O is the LAST thing of the step: left alone, the compiler hoists this branch to the point where alpha is known
            // (right behind the row maximum) and the exponentials land in a block of their own, with no MFMA beside them
            if (HASC && HASA) {                  // steady state: one MFMA, its fragment reads, a slice of the softmax -- 8 * QB * (DP + DQ) times
                // the fragment reads run THREE MFMAs ahead of their consumer (an LDS round trip is two to three MFMA gaps)
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
#pragma unroll
                for (int m = 0; m < 4 * QB * (DP + DQ); m++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                    // DS reads (of the MFMA three further on)
                    __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);                    // VALU (the softmax)
                }
            }
            int gflag = grew ? 1 : 0;
#pragma unroll
            for (int qb = 0; qb < QB; qb++) {    // (the pin: every P fragment and the row sums are inputs of an empty asm the flag passes through)
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                u32x4 p0, p1, p2, p3;
                __builtin_memcpy(&p0, &P[PAR][qb][0][0], 16);
                __builtin_memcpy(&p1, &P[PAR][qb][0][1], 16);
                __builtin_memcpy(&p2, &P[PAR][qb][1][0], 16);
                __builtin_memcpy(&p3, &P[PAR][qb][1][1], 16);
                asm volatile("" : "+v"(gflag) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(l_run[qb]));
            }
            if (__any(gflag)) {                  // wave-uniform: after the first tiles the running maxima rarely move
#pragma unroll
                for (int qb = 0; qb < QB; qb++)
#pragma unroll
                    for (int d = 0; d < DP; d++)
#pragma unroll
                        for (int i = 0; i < 16; i++) ot[qb][d][i] *= alpha[qb];
            }
        }
    };
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using Y = std::true_type;
    using N = std::false_type;

    // ---- prologue: tiles 0 .. NST-3 requested; step -1 forms S(0)
#pragma unroll
    for (int t = 0; t < NST - 3; t++) issue(t);
    {
        // (step -1 by hand: its wait leaves NST-4 later tiles in flight)
        wait_tiles_ahead<2 * PPW, AHEAD>(ntiles - 1);
        asm volatile("s_barrier" ::: "memory");
        issue(NST - 3);
        {
            half8_t kf[2][DQ];
#pragma unroll
            for (int kt2 = 0; kt2 < 2; kt2++)
#pragma unroll
                for (int s = 0; s < DQ; s++) kf[kt2][s] = *reinterpret_cast<const half8_t*>(smem + kt2 * 32 * 128 + koff[s]);
#pragma unroll
            for (int kt2 = 0; kt2 < 2; kt2++)
#pragma unroll
                for (int qb = 0; qb < QB; qb++) {
#pragma unroll
                    for (int i = 0; i < 16; i++) S[0][qb][kt2][i] = 0.f;
#pragma unroll
                    for (int s = 0; s < DQ; s++) S[0][qb][kt2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kt2][s], qf[qb][s], S[0][qb][kt2], 0, 0, 0);
                }
        }
    }
    const bool ragged = extra + ntiles * KT > a.Nk;          // the last tile has keys past the end
    // step 0
    if (ntiles > 1) step(T0{}, N{}, Y{}, Y{}, N{}, 0);
    else if (ragged) step(T0{}, N{}, N{}, Y{}, Y{}, 0);
    else step(T0{}, N{}, N{}, Y{}, N{}, 0);
    // steady state
    int j = 1;
    for (; j + 2 < ntiles; j += 2) {             // two steps per trip: the parities are compile-time, no join between differently allocated halves
        step(T1{}, Y{}, Y{}, Y{}, N{}, j);
        step(T0{}, Y{}, Y{}, Y{}, N{}, j + 1);
    }
    if (j + 1 < ntiles) { step(T1{}, Y{}, Y{}, Y{}, N{}, j); j++; }
    // last tile's softmax beside PV of the one before it
    if (ntiles > 1) {
        if (ragged) {
            if (j & 1) step(T1{}, Y{}, N{}, Y{}, Y{}, j); else step(T0{}, Y{}, N{}, Y{}, Y{}, j);
        } else {
            if (j & 1) step(T1{}, Y{}, N{}, Y{}, N{}, j); else step(T0{}, Y{}, N{}, Y{}, N{}, j);
        }
        j++;
    }
    // PV of the last tile (j == ntiles)
    if (j & 1) step(T1{}, Y{}, N{}, N{}, N{}, j); else step(T0{}, Y{}, N{}, N{}, N{}, j);

#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
        float l0_, l1_;
        halves(l_run[qb], l0_, l1_);
        const float l_tot = l0_ + l1_;
        const float inv = 1.0f / l_tot;
        const int qr = q0 + qb * 32 + r;
        if (qr < a.Nq) {
            h16* orow = a.o + (size_t)b * a.o_sb + (size_t)qr * a.o_sn + (size_t)h * D;
#pragma unroll
            for (int d = 0; d < DP; d++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    int dd = d * 32 + 8 * g4 + 4 * hh;
                    if (dd < D) {
                        half4_t o = {(h16)(ot[qb][d][4 * g4 + 0] * inv), (h16)(ot[qb][d][4 * g4 + 1] * inv), (h16)(ot[qb][d][4 * g4 + 2] * inv),
                                     (h16)(ot[qb][d][4 * g4 + 3] * inv)};
                        *reinterpret_cast<half4_t*>(orow + dd) = o;
                    }
                }
        }
    }
}

std::atomic<int> g_attn_variant{[] { const char* e = getenv("CUT3R_ATTN_PIPE"); return e ? atoi(e) : 1; }()};

template <int D>
int launch_attn(const AttnArgs& a, int B, int H, hipStream_t s) {
    const long long blocks128 = (long long)B * H * ((a.Nq + 127) / 128);
    static const long long nw4_min = [] { const char* e = getenv("CUT3R_ATTN_NW4_MIN"); return e ? atoll(e) : 384LL; }();
    if constexpr (D == 48 || D == 64) {
        // the pipelined form (attn_pipe_kernel) is the default for the network's head widths at every size: measured round 4
        // (profiles/r04/attn_pipe_bench.txt) 665 vs 624 TF/s on the encoder shape, 521-533 vs 476-511 on the decoder's 64-wide heads,
        // 444-465 vs 444-447 on its 48-wide ones, and 13.4 vs 19.1 us for ONE window's launch.  Same bits as attn_kernel, so the choice
        // is free; cut3r_attention_variant(0) / CUT3R_ATTN_PIPE=0 selects attn_kernel (tests, A/B runs)
        if (g_attn_variant.load(std::memory_order_relaxed) != 0) {
            const int nblk = (a.Nq + 127) / 128, HB = H * B;
            dim3 grid((unsigned)(((HB + 7) / 8) * 8 * nblk));
            hipLaunchKernelGGL((attn_pipe_kernel<D, 1, 4, 4>), grid, dim3(256), 0, s, a, nblk, HB, H);
            return cut3r_check_launch();
        }
    }
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

extern "C" int cut3r_attention_variant(int v) {
    if (v < 0) return g_attn_variant.load(std::memory_order_relaxed);
    return g_attn_variant.exchange(v ? 1 : 0, std::memory_order_relaxed);
}

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
