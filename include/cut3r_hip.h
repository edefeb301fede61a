/* cut3r_hip.h -- C ABI of libcut3r_hip.so: the MI355X (gfx950) kernels behind the CUT3R-SLAM hot path.
 *
 * Plain pointers + sizes only (device pointers unless stated), `stream` is a hipStream_t passed as void*.
 * Every entry point returns 0 on success, 1 on an argument/shape/alignment violation (nothing launched),
 * 2 if the launch itself failed.  Nothing here allocates, synchronises or touches the default stream, so every
 * call can be captured into a hipGraph.
 *
 * Each function cites the reference interface it replaces (paths relative to /root/reference).
 */
#ifndef CUT3R_HIP_H
#define CUT3R_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- version / probe ------------------------------------------------------------------------------------- */
int cut3r_abi_version(void);                         /* bumps when a signature changes */

/* ---- RoPE-2D -------------------------------------------------------------------------------------------------
 * replaces curope.rope_2d(tokens, positions, base, fwd)  (src/croco/models/curope/curope.cpp:49-67,
 * kernels.cu:17-108).  In place on a (B,N,H,D) view: element (b,n,h,d) at tokens + b*sB + n*sN + h*sH + d
 * (strides in elements, multiples of 4; the reference requires sH == D, stride(3) == 1; D % 16 == 0).  positions: int64 [B,N,2] contiguous,
 * may be negative.  dtype: 0 = fp32, 1 = fp16 (fp32 math, rounded on store -- the CUDA kernel's contract). */
int cut3r_rope2d(void* tokens, int dtype, const int64_t* positions, int B, int N, int H, int D,
                 long long sB, long long sN, long long sH, float base, float fwd, void* stream);
/* the two rope_2d calls of a self-attention (croco/models/blocks.py:127-128, dust3r/blocks.py:118-119: q then k, same
 * positions) in ONE launch: head stride == D for both; cos/sin are evaluated once per token. */
int cut3r_rope2d_qk(void* q, void* k, int dtype, const int64_t* positions, int B, int N, int H, int D, long long q_sB,
                    long long q_sN, long long k_sB, long long k_sN, float base, float fwd, void* stream);

/* the same rotation driven by the cos|sin table of cut3r_rope2d_table (identical bits: the table is filled by the expressions of
 * cut3r_rope2d), for fp16 tokens with head stride D: ONE launch rotates two token ranges -- q and k of a self-attention, or q of
 * one token stream and k of the other in a cross-attention (dust3r/blocks.py:118-119,226-227).  t_i: ntok_i tokens, token stride
 * stride_i elements (multiple of 8 when D % 32 == 0, else of 4), pos_i int64 [ntok_i,2]; t1 may be NULL.  Positions outside
 * [pmin, pmin+npos) are evaluated in place. */
int cut3r_rope2d_tab(void* t0, const int64_t* pos0, long long ntok0, long long stride0, void* t1, const int64_t* pos1, long long ntok1,
                     long long stride1, int H, int D, const float* table, int pmin, int npos, float base, float fwd, void* stream);

/* ---- LayerNorm (+adaLN modulation) -------------------------------------------------------------------------
 * replaces nn.LayerNorm(eps=1e-6) calls in croco/models/blocks.py:187-190, dust3r/blocks.py:292-297 and
 * ModLN (dust3r/blocks.py:356-379: y = LN(x)*(1+scale)+shift).  x fp32 [M,C] (row stride ldx).  Writes any of:
 * y16 (fp16, ld16) and y32 (fp32, ld32).  mod_scale/mod_shift: optional fp32 [C] (NULL = plain LN). */
int cut3r_layernorm(const float* x, int ldx, const float* gamma, const float* beta, float eps, int M, int C,
                    void* y16, int ld16, float* y32, int ld32, const float* mod_scale, const float* mod_shift,
                    void* stream);
/* one tensor, two affine parameter sets, two fp16 outputs (shared row statistics): norm1 of one decoder block and norm_y of
 * the other act on the same tokens (dust3r/blocks.py:292-297).  C in {768, 1024, 1536}. */
int cut3r_layernorm_dual(const float* x, int ldx, const float* g1, const float* b1, void* y1, int ld1, const float* g2,
                         const float* b2, void* y2, int ld2, float eps, int M, int C, void* stream);

/* ---- GEMM family ---------------------------------------------------------------------------------------------
 * replaces nn.Linear / nn.Conv2d / nn.ConvTranspose2d(+bias)(+GELU|ReLU)(+residual) in the ViT and DPT stacks
 * (croco/models/blocks.py:68-148, dust3r/blocks.py:87-243, croco/models/dpt_block.py:84-232,281-513).
 * C[M,N] = epi(A[M,K] * B[N,K]^T); A,B fp16, fp32 accumulate (v_mfma_f32_16x16x32_f16). */
typedef struct cut3r_gemm_desc {
    const void* A;        /* fp16 [M,K] row-major (lda) -- or NHWC [Bimg,H,W,Cin] when conv_k == 3 */
    const void* B;        /* fp16 [N,K] row-major (ldb) == nn.Linear.weight; conv: [Cout][ky][kx][Cin] */
    void* C;              /* fp16 or fp32 [M,N] (ldc) */
    const float* bias;    /* fp32 [N] or NULL */
    const void* res1;     /* optional residual [M,N] (ldr1), fp32 or fp16 */
    const void* res2;     /* optional second residual */
    int M, N, K, lda, ldb, ldc, ldr1, ldr2;
    int act;              /* 0 none, 1 exact GELU (erf), 2 ReLU */
    int out_f16, res1_f16, res2_f16;
    int batch;            /* >= 1: independent problems on blockIdx.z with the element strides below */
    long long strideA, strideB, strideC, strideBias, strideR1, strideR2;
    int conv_k;           /* 0/1 = plain GEMM (1x1 conv is a plain GEMM over pixels), 3 = implicit 3x3, pad 1 */
    int H, W, Cin, conv_stride, Ho, Wo;   /* conv_k == 3: input/output spatial sizes, M = Bimg*Ho*Wo, K = 9*Cin */
    int relu_in;          /* apply ReLU to A on load (ResidualConvUnit pre-activation) */
    int shuf;             /* > 0: ConvTranspose(k == stride == shuf) scatter; N = shuf*shuf*shuf_cout, ldc = Cout */
    int shuf_cout, shuf_Hin, shuf_Win;
    int tile;             /* 0 = auto (cut3r_gemm_tile_for), 16 (M <= 64 skinny), 64, 128, 192128 (192 x 128), 128192 (128 x 192) or 256 */
    int stages;           /* 0 = default; LDS ring depth override (tuning): 2|3 for tile 128, 2|3|4 for tile 64 */
    /* fused 2-D RoPE on the first rope_cols output columns (head dimension 64 only; fp16 output, no activation/residual):
     * what curope.rope_2d does to q / k right after the projection (croco/models/blocks.py:126-127, dust3r/blocks.py
     * :113-114,215-223), applied to the fp16-rounded projection exactly as the separate kernel would.  rope_pos int64 [M,2]
     * (y, x) per output row; rope_table from cut3r_rope2d_table (cos | sin for positions rope_pmin..rope_pmin+npos-1). */
    const void* rope_pos;
    const float* rope_table;
    int rope_cols, rope_pmin, rope_npos;
    int rope_d;           /* head dimension of the fused RoPE: 64 (default when 0) or 48 (forces the 128 x 192 tile) */
    /* LayerNorm folded into the GEMMs (round 4; replaces the nn.LayerNorm in front of qkv / projq / projk|projv / fc1:
     * croco/models/blocks.py:187-190, dust3r/blocks.py:292-297).  y = LN(x) W^T + b = rstd (x (gamma.W)^T - mu c) + d.
     * CONSUMER: A = fp16 copy of the UN-normalised rows x, B = fp16(gamma . W), bias = d = W beta + b, ln_colsum = c
     * (c_n = sum_k B_nk, fp32 [N]), ln_stats = fp32 [K/64][M][2] (slab-major): (sum, m2 = sum (x - sum/64)^2) of every 64-column
     * slab of every row, ln_nslab = K/64, ln_eps = the LayerNorm's eps.  The row's slabs are combined in slab order (parallel
     * variance), the raw accumulator becomes acc*rstd - (rstd*mu)*c_n, then the usual epilogue runs (bias, GELU, RoPE, fp16).
     * PRODUCER: an fp32-output GEMM with an fp32 residual (the one that writes the residual stream) also stores out16 = fp16
     * copy of its output rows (ld16) and stats_out = their slab statistics, fp32 [N/64][M][2] (N % 64 == 0), computed in one
     * fixed order in every tile kernel.  Tiles 256 / 128 / 64 (producer), + 128 x 192 (consumer); not the skinny tile. */
    const float* ln_stats;
    const float* ln_colsum;
    int ln_nslab;
    float ln_eps;
    float* stats_out;
    void* out16;
    int ld16;
} cut3r_gemm_desc;
int cut3r_gemm_f16(const cut3r_gemm_desc* d, void* stream);
/* table[0][p][q] = cos((pmin+p) * fwd / base^(q/Q)), table[1][p][q] = sin(...), q < Q = head_dim/4, p < npos: the angles
 * of cut3r_rope2d, evaluated by the same device functions (fp32 [2][npos][Q]) */
int cut3r_rope2d_table(float* table, int pmin, int npos, int Q, float base, float fwd, void* stream);
/* the tile the launcher picks for this problem when desc->tile == 0: 256 (256x256x64 ping-pong kernel), 128 or 64.
 * tile 16 (M <= 64: skinny weight-streaming MFMA kernel) is never chosen automatically: its K-split changes the fp32
 * summation order, so callers request it for operands whose row count is the batch (one row per tracking window) and
 * keep every other GEMM on the tile kernels, whose rows are bit-identical across tile sizes and batch sizes. */
int cut3r_gemm_tile_for(const cut3r_gemm_desc* d);
/* TWO independent linear problems (same N, K; own operands, row counts and epilogues) in ONE launch: the state-side and the
 * image-side GEMM of a decoder layer -- both DecoderBlocks of a layer read the previous layer's pair
 * (src/dust3r/model.py:669-692), so they are independent; one grid over both fills the chip where each alone does not.
 * Linears only (no convolution / pixel shuffle / batch).  Rows are bit-identical to cut3r_gemm_f16.  Tile 256 / 128 on large combined
 * grids; below 128 tiles of 128 x 128 the 64 x 64 pair kernels (the one-window schedule: 2 x 156 tiles at M = 768 / 769), which also
 * carry the fused RoPE and both sides of the LayerNorm fold, each problem with its own flags. */
int cut3r_gemm_f16_pair(const cut3r_gemm_desc* d0, const cut3r_gemm_desc* d1, void* stream);

/* skinny M<=64 path: Y[M,N] = act(X[M,K] (fp32, optional SiLU on load) * W[N,K]^T (fp16) + bias) (+res) */
int cut3r_gemv_f16w(const float* X, int ldx, const void* W, int ldw, const float* bias, float* Y, int ldy,
                    int M, int N, int K, int act, const float* res, int ldr, int silu_in, void* stream);

/* ---- fused attention ---------------------------------------------------------------------------------------------
 * replaces F.scaled_dot_product_attention(q,k,v, scale) (no mask, p=0) at croco/models/blocks.py:139-145 and
 * dust3r/blocks.py:123-129,233-239.  fp16 q/k/v, fp32 online softmax, fp16 out.
 * Element (b,n,h,d) of q lives at q + b*q_sb + n*q_sn + h*D + d (same for k,v with their strides; out likewise).
 * D in {16,32,48,64,128}. */
int cut3r_attention_f16(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk, int D,
                        long long q_sb, long long q_sn, long long k_sb, long long k_sn, long long v_sb, long long v_sn,
                        long long o_sb, long long o_sn, float scale, void* stream);

/* which kernel serves the 48- and 64-wide heads: 1 (default) the software-pipelined LDS-DMA kernel, 0 the register-staged one.
 * Both give the same bits (tests/test_kernels_gpu.py); the switch exists for that test and for A/B timing.  v < 0 only queries.
 * Returns the previous setting.  (No reference counterpart: a tuning knob of this library.) */
int cut3r_attention_variant(int v);

/* ---- elementwise / layout helpers of the ViT path -------------------------------------------------------------- */
/* PatchEmbed conv 16x16/s16 as im2col (src/dust3r/patch_embed.py:18-32): img fp32 [B,C,H,W] -> fp16 [B*(H/P)*(W/P), C*P*P],
 * k ordered (c,py,px) == Conv2d weight.flatten(1).  If u8 != 0 the input is uint8 and (x/255-0.5)/0.5
 * (model.py:1111-1114 `normalize`) is fused into the load. */
int cut3r_im2col_patch(const void* img, int u8, int B, int C, int H, int W, int P, void* out, void* stream);
/* fp32 -> fp16 cast of a 2-D view (row strides in elements) */
int cut3r_cast_f32_f16(const float* x, int ldx, void* y, int ldy, int M, int C, void* stream);
/* column mean over rows: y[c] = mean_m x[m,c]  (model.py:732 `_get_img_level_feat`) ; x fp32 [M,C] */
int cut3r_colmean(const float* x, int ldx, int M, int C, float* y, void* stream);
/* B independent matrices (element strides stride_x / stride_y between them) in one launch; same summation order */
int cut3r_colmean_batched(const float* x, int B, long long stride_x, int ldx, int M, int C, float* y, long long stride_y, void* stream);

/* ---- DPT head helpers (NHWC fp16 activations) ------------------------------------------------------------------ */
/* bilinear x2 upsample, align_corners=True (dpt_block.py:215-221, :262-268): in [B,H,W,C] -> out [B,2H,2W,C] */
int cut3r_upsample2x_nhwc(const void* in, void* out, int B, int H, int W, int C, void* stream);
/* final 1x1 conv (Cin -> 4 or 3) + output activations (heads/postprocess.py:11-28,113-151):
 * mode 0: pts3d = xyz/|xyz| * expm1(|xyz|), conf = 1 + exp(c)   -> pts [P,3] fp32, conf [P] fp32
 * mode 1: rgb = (sigmoid(x)*(1-2e-6)+1e-6 - 0.5)*2              -> pts [P,3] fp32
 * in: fp16 [P,Cin]; w: fp32 [nout,Cin]; b: fp32 [nout] */
int cut3r_dpt_final(const void* in, int P, int Cin, const float* w, const float* b, int mode, float* pts, float* conf,
                    void* stream);
/* postprocess on raw fp32 [P,4|3] maps (linear head path, pos_z option: linear_head.py:316) */
int cut3r_postprocess_pts(const float* raw, int P, int nch, int pos_z, float* pts, float* conf, void* stream);
/* camera pose activation (heads/postprocess.py:30-63): in fp32 [B,7] -> out [B,7] (t*expm1|t|/|t|, unit quat w>=0) */
int cut3r_postprocess_pose(const float* raw, int B, float* out, void* stream);

/* ---- keyframe selection ---------------------------------------------------------------------------------------------
 * replaces compute_patch_overlap_ratio (hislam2/util/utils.py:726-736): rows 1.. of feat0/feat1 fp32 [N,C] are
 * L2-normalised, sim = f0 f1^T (exact fp32 MFMA), count = #rows with max_j sim > thr.  count: int32[1] (device),
 * zeroed by the call.  ws: 16-B aligned fp32 workspace of 2*(N-1)*C + (N-1) elements. */
int cut3r_patch_overlap(const float* feat0, const float* feat1, int N, int C, float thr, void* ws, int32_t* count,
                        void* stream);
/* The keyframe decisions of a whole look-ahead batch with no host round trip per candidate: replaces the decision loop of
 * MotionFilter.kfFilter (hislam2/motion_filter.py:98-124) over B consecutive tested frames whose encoder features
 * feats fp32 [B,N,C] are already resident.  Candidate i is compared with the LAST KEYFRAME SO FAR -- feat_last [N,C] until
 * a candidate is taken, then that candidate (kept in *state on the device: -1 | index) -- with the arithmetic of
 * cut3r_patch_overlap (bit-identical counts), and is taken when (double)(float(count)/float(N-1)) < thr_ratio, or
 * unconditionally when forced_host[i] != 0 (first / second-last / last frame, :87-96; forced_host may be NULL).
 * counts, decisions: int32 [B] (device); state: int32 [1] (device).
 * ws: 16-B aligned fp32 workspace of (B+1)*(N-1)*C + (N-1) elements. */
int cut3r_patch_overlap_chain(const float* feat_last, const float* feats, int B, int N, int C, float thr_sim, double thr_ratio,
                              const int32_t* forced_host, void* ws, int32_t* state, int32_t* counts, int32_t* decisions,
                              void* stream);

/* ---- covisibility graph geometry ------------------------------------------------------------------------------------
 * replaces FactorGraph.cal_overlap_batch / cal_overlap_bi (hislam2/factor_graph.py:255-315).
 * w2c: fp32 [B,12] = top 3x4 of inverse(c2w) row-major; K4 = (fx,fy,cx,cy) HOST floats; counts int32[B] (device).
 * fwd: ONE pointmap pm [N,3] projected into B cameras (z clamped at 1e-5 for the divide, :272).  If P_host != NULL the
 *      points are first mapped p <- P*(s_align*p) (12 HOST floats, 3x4 row-major): the tracker's chained full-resolution
 *      pointmap (track_frontend.py:234,259) is then never materialised.
 * bwd: B pointmaps projected into ONE camera (raw z divide, :304).  Pointmap b starts at
 *      pms + slot(b)*N*3 with slot(b) = b when grp == 0, else (b/grp)*grp_stride + b%grp -- the keyframe order of the
 *      [submap][6 slots] store (hislam2/keyframe.py:28, track_frontend.py:251-255) without gathering a copy. */
int cut3r_overlap_fwd(const float* pm, int N, const float* P_host, float s_align, const float* w2c, int B, float fx, float fy,
                      float cx, float cy, int W, int H, int clamp_z /* 1: cal_overlap_batch, 0: cal_overlap_bi with B1 == 1 */,
                      int32_t* counts, void* stream);
int cut3r_overlap_bwd(const float* pms, int B, int N, int grp, int grp_stride, const float* w2c, float fx, float fy, float cx,
                      float cy, int W, int H, int32_t* counts, void* stream);

/* ---- whole-window update -------------------------------------------------------------------------------------------
 * One call for everything TrackFrontend.track does per pixel for a window of V <= 6 consecutive keyframes t0..t0+V-1
 * (hislam2/track_frontend.py:193-262 + factor_graph.py:148-197, :255-315): the V cut3r_align_view results (pm_ds, conf_ds,
 * depth are the V consecutive slots / rows of the resident stores) and, for every keyframe i = t0+v >= first, the forward
 * counts of its full-resolution chained pointmap in cameras 0..i-1 and the backward counts of stored pointmaps 0..i-1
 * (slot addressing as cut3r_overlap_bwd, over `store`) in camera i.  pts [V,H,W,3], conf [V,H,W] contiguous; P_host
 * V*12 HOST floats; w2c device [>= t0+V, 12]; counts int32 device [V][2][ldc] (row 0 forward, row 1 backward; ldc >=
 * t0+V), zeroed here.  w2c_new_host (optional, V*12 HOST floats): rows t0..t0+V-1 of w2c are written from it first (the
 * keyframe store's device mirror, hislam2/keyframe.py:24 pose) -- no separate upload.  lsum_reset (optional): a
 * cut3r_logdepth_accum accumulator zeroed for the next window.  Three launches instead of ~6 per keyframe; every count
 * equals the per-keyframe entry points'. */
int cut3r_window_update(const float* pts, const float* conf, int V, int H, int W, const float* P_host, float s, int ds,
                        float* pm_ds, float* conf_ds, float* depth, const float* store, int grp, int grp_stride,
                        float* w2c, const float* w2c_new_host, int t0, int first, float fx, float fy, float cx, float cy,
                        int32_t* counts, int ldc, double* lsum_reset, void* stream);

/* ---- window alignment -----------------------------------------------------------------------------------------------
 * replaces the per-view tensor math of TrackFrontend.track (hislam2/track_frontend.py:193-243): pointmap = P*(s*pts),
 * conf <- 1-1/conf, depth = s*z, stride-`ds` downsample.  P_host: 12 HOST floats (chained c2w 3x4, row-major), s by value
 * (both are products of tiny 4x4 host math on the 7-float camera poses, exactly as in the reference). */
/* cut3r_logdepth_sum without the zeroing memset: *out += sum(log prev - log z) (the caller keeps *out zeroed, e.g. through
 * cut3r_window_update's lsum_reset) */
int cut3r_logdepth_accum(const float* prev_depth, const float* pts, int n, double* out, void* stream);

int cut3r_align_view(const float* pts, const float* conf, int H, int W, const float* P_host, float s, int ds,
                     float* pm_ds, float* conf_ds, float* depth, void* stream);
/* sum_i log(prev_depth[i]) - log(pts[i].z) -> out[0] (fp64 accumulate, device double[1], zeroed by the call) */
int cut3r_logdepth_sum(const float* prev_depth, const float* pts, int n, double* out, void* stream);

/* ---- Lie groups (slot of the un-vendored princeton-vl/lietorch 0.2, "lietorch_backends") -------------------------------
 * call sites: hislam2/track_backend.py:269-270,298-299,418-425,458-459 (SE3.exp / .matrix / .data),
 * hislam2/gs_backend_per_frame.py:721-731, hislam2/geom/projective_ops.py:17,51,67,69, hislam2/geom/ba.py:29,37.
 * group: 0 = SO3 (tangent 3, data 4 = q_xyzw), 1 = SE3 (6 = [tau,phi], 7 = [t,q_xyzw]), 2 = Sim3 (7, 8 = [t,q,s]).
 * All tensors fp32, contiguous, n elements.  op: 0 exp (tangent->data), 1 log (data->tangent), 2 inv (data->data),
 * 3 matrix (data -> 16 floats, row-major 4x4).  *_bwd are vector-Jacobian products w.r.t. the Euclidean components. */
int cut3r_lie_unary(int group, int op, const float* in, float* out, int n, void* stream);
int cut3r_lie_unary_bwd(int group, int op, const float* in, const float* grad_out, float* grad_in, int n, void* stream);
int cut3r_lie_mul(int group, const float* x, const float* y, float* out, int n, void* stream);
int cut3r_lie_mul_bwd(int group, const float* x, const float* y, const float* grad_out, float* grad_x, float* grad_y, int n,
                      void* stream);
/* element e acts on its P points p[e,:,pd] (pd = 3, or 4 = homogeneous [X,Y,Z,W] -> [R*XYZ*s + t*W, W]) */
int cut3r_lie_act(int group, const float* x, const float* p, float* out, int n, int P, int pd, void* stream);
int cut3r_lie_act_bwd(int group, const float* x, const float* p, const float* grad_out, float* grad_x, float* grad_p, int n, int P,
                      int pd, void* stream);
/* adjoint (transpose = 0) or transposed adjoint (1) of element e applied to tangent vector a[e] */
int cut3r_lie_adj(int group, const float* x, const float* a, float* out, int n, int transpose, void* stream);

/* ---- loop closure: fused submap-alignment optimiser -----------------------------------------------------------------
 * replaces the Adam loop over SE3.exp(xi).matrix() of TrackBackend.loop_closure_init
 * (hislam2/track_backend.py:256-299; lr 5e-4, `iteration` steps) and the pointmap rewrite (:301-310).
 * first/last: device pointers to slot 0 / slot 5 of the [B][6][N][3] submap store (sub_stride = floats between
 * consecutive submaps); mask: uint8 [B-1,N] (conf of `last` > 0) or NULL; cur/cur_lc: [N,3] current pointmap in the
 * global frame / in the matched submap's frame.  xi,adam_m,adam_v: [B,6] (row 0 unused, zero-initialised by the caller),
 * T: [B,12] row-major 3x4 (identity-initialised); workspace: cut3r_lc_workspace_floats(B,N) floats; loss_out: NULL or
 * float[iters].  Two launches per iteration, deterministic (no float atomics). */
int cut3r_lc_workspace_floats(int B, int N);
int cut3r_lc_optimize(const float* first, const float* last, long long sub_stride, const unsigned char* mask, const float* cur,
                      const float* cur_lc, int B, int N, long long n_masked, int iters, float lr, float* xi, float* adam_m,
                      float* adam_v, float* T, float* workspace, float* loss_out, void* stream);
/* General form for the second and later loop closures (TrackBackend.loop_closure, hislam2/track_backend.py:400-461: the
 * chain term + the matched-submap term + the current-vs-lc term, Adam over `_align_lie` AND `matched_lie`).  The objective
 * is a LIST of L1 terms  w * sum_n | T[ia] a_n - T[ic] c_n |_1  over [N,3] point arrays; T[0] is the fixed identity, every
 * other row of T [P,12] is exp of a row of xi [P,6] and is optimised.  terms_dev: DEVICE array of n_terms cut3r_lc_term.
 * workspace: cut3r_lc_workspace_floats(n_terms, N) floats.  xi/adam_m/adam_v zero- and T identity-initialised by the
 * caller.  Two launches per iteration, deterministic. */
typedef struct cut3r_lc_term {
    const float* a;              /* [N,3] points moved by T[ia] */
    const float* c;              /* [N,3] points moved by T[ic] */
    const unsigned char* mask;   /* NULL or uint8 [N]: points with mask 0 are skipped */
    int ia, ic;                  /* transform indices, 0 = fixed identity */
    float w;                     /* weight of the term (1 / (3 * #points of its mean)) */
    int pad;
} cut3r_lc_term;
int cut3r_lc_optimize_terms(const void* terms_dev, int n_terms, int P, int N, int iters, float lr, float* xi, float* adam_m,
                            float* adam_v, float* T, float* workspace, float* loss_out, void* stream);
/* in place p <- T_b p for the points_per_submap points of each of the B submaps (pts: [B, points_per_submap, 3]) */
int cut3r_transform_submaps(float* pts, const float* T, int B, long long points_per_submap, void* stream);

/* ---- legacy dense-BA operators (`droid_backends` sources are absent from the reference) --------------
 * corr_index: replaces droid_backends.corr_index_forward/backward (call sites hislam2/modules/corr.py:12,19).
 * volume [BN,h1,w1,h2,w2], coords [BN,2,h1,w1] (x,y), out [BN,2r+1,2r+1,h1,w1]: bilinear lookup with zero padding,
 * out[n,i,j,y,x] = volume[n,y,x] sampled at (x0 - r + i, y0 - r + j). */
int cut3r_corr_index_forward(const float* volume, const float* coords, float* out, int BN, int h1, int w1, int h2, int w2,
                             int radius, void* stream);
int cut3r_corr_index_backward(const float* coords, const float* grad_out, float* grad_volume, int BN, int h1, int w1, int h2, int w2,
                              int radius, void* stream);
/* one Gauss-Newton step of geom.ba.BA (hislam2/geom/ba.py:32-107, geom/chol.py:47-78, geom/projective_ops.py:44-74):
 * Gij [N,7] = G_j*G_i^-1 (SE3 data), disps [P,ht*wd], intr [P,4], target/weight [N,ht*wd,2], eta [M,ht*wd];
 * ii,jj int32 [N]; CSR of edges by source frame (src_ptr [M+1], src_edges [N], kx [M] = sorted unique(ii));
 * present uint8 [P-fixedp, M] marks the non-zero E blocks.  Outputs dx [P-fixedp,6], dz [M,ht*wd], flag[0] = 1 if
 * the Cholesky failed (dx = 0, as geom/chol.py:13-18).  Requires 6*(P-fixedp) <= 192. */
long long cut3r_ba_workspace_floats(int P, int ht, int wd, int N, int M, int fixedp);
int cut3r_ba_step(const float* Gij, const float* disps, const float* intr, const float* target, const float* weight, const float* eta,
                  const int* ii, const int* jj, const int* src_ptr, const int* src_edges, const int* kx, const unsigned char* present,
                  int P, int ht, int wd, int N, int M, int fixedp, float ep, float lm, float* workspace, float* dx, float* dz, int* flag,
                  void* stream);
/* The same step in three stages, so that EDGES CAN BE SHARDED by source frame over several GPUs (north_star: all-reduce of the
 * normal-equation blocks): every rank assembles the UNDAMPED reduced system S = H - E C^-1 E^T, vS = v - E C^-1 w of ITS edges
 * (each source frame m -- with its depth blocks C_m, w_m, E_.m -- lives on exactly one rank) plus diag(H); S, vS and the
 * diagonal (6(P-fixedp))^2 + 2*6(P-fixedp) floats are summed over ranks; every rank then damps (S + (ep + lm*diag H) I,
 * geom/chol.py:56-57), solves, and back-substitutes dz for its own source frames.  motion_only != 0 drops the Schur
 * correction: the reduced system of MoBA (geom/ba.py:110-158).  S_out [n,n], vS_out [n], hdiag_out [n], n = 6(P-fixedp);
 * scratch: n*n floats. */
int cut3r_ba_assemble(const float* Gij, const float* disps, const float* intr, const float* target, const float* weight, const float* eta,
                      const int* ii, const int* jj, const int* src_ptr, const int* src_edges, const int* kx, const unsigned char* present,
                      int P, int ht, int wd, int N, int M, int fixedp, int motion_only, float* workspace, float* S_out, float* vS_out,
                      float* hdiag_out, void* stream);
int cut3r_ba_solve(const float* S, const float* vS, const float* hdiag, int n, float ep, float lm, float* scratch, float* dx, int* flag,
                   void* stream);
int cut3r_ba_backsub(float* workspace, const float* dx, const unsigned char* present, int P, int ht, int wd, int N, int M, int fixedp,
                     float* dz, void* stream);
/* droid_backends.proj_trans (call site hislam2/geom/ba.py:200): per-source depth normal equations with the poses held fixed,
 * C [M,ht*wd] = sum_e w Jz^2, w [M,ht*wd] = sum_e w r Jz.  workspace: N * ceil(ht*wd/256) * 120 floats. */
int cut3r_ba_proj_trans(const float* Gij, const float* disps, const float* intr, const float* target, const float* weight, const int* ii,
                        const int* jj, const int* src_ptr, const int* src_edges, const int* kx, int P, int ht, int wd, int N, int M,
                        float* workspace, float* C_out, float* w_out, void* stream);
/* geom.chol.schur_solve_mono_prior (/root/reference/hislam2/geom/chol.py:80-107; call site geom/ba.py:235): the scale-grid block of
 * JDSA reduced over the per-pixel disparities.  H [n,n], E [n,cols], v [n] with n = M*D scale nodes (<= 192) and cols = M*ht*wd
 * disparities, in the layout the reference builds with permute + reshape; C, w [cols].  S = H + (ep + lm diag H) I - E C^-1 E^T,
 * in-LDS Cholesky, dso = S^-1 (v - E C^-1 w), dz = C^-1 (w - E^T dso), dzcov = |L^-1 E C^-1|^2 column-wise + C^-1 (NULL: skipped).
 * A failed factorisation gives dso = 0 (CholeskySolver, chol.py:13-18) and sets flag[0].  workspace:
 * cut3r_schur_mono_prior_workspace_floats(n, cols) floats. */
long long cut3r_schur_mono_prior_workspace_floats(int n, long long cols);
int cut3r_schur_mono_prior(const float* C, const float* w, const float* H, const float* E, const float* v, int n, long long cols, float ep,
                           float lm, float* workspace, float* dso, float* dz, float* dzcov, int* flag, void* stream);
/* the normal-equation blocks of geom.ba.JDSA's scale grids (/root/reference/hislam2/geom/ba.py:213-228): per source frame m,
 * Jso = -[prior > 0] prior Jbi[m] ([HW,D], Jbi from cut3r_bi_inter), H_m = alpha Jso^T Jso, E_m = alpha Jso^T, v_m = -alpha Jso^T rd,
 * written as the block-diagonal dense H [M*D, M*D], E [M*D, M*HW], v [M*D] that cut3r_schur_mono_prior takes. */
int cut3r_jdsa_blocks(const float* prior, const float* Jbi, const float* rd, float alpha, int M, int HW, int D, float* H, float* E, float* v,
                      void* stream);
/* droid_backends.bi_inter (call site hislam2/geom/ba.py:167): bilinear interpolation of per-frame scale grids scales [M,hs,ws] at
 * grid [M,ht,wd,2] (x, y) -> vals [M,ht,wd] and the dense Jacobian J [M,ht,wd,hs*ws] w.r.t. the grid nodes. */
int cut3r_bi_inter(const float* scales, const float* grid, int M, int hs, int ws, int ht, int wd, float* vals, float* J, void* stream);
/* droid_backends.depth_filter (call site hislam2/util/droid_visualization.py:100; the extension is absent from the reference tree,
 * semantics from the published DROID-SLAM kernel): poses [n,7] world->camera SE3 data (t, q_xyzw), disps [n,ht,wd] inverse depths,
 * intr [4] (fx, fy, cx, cy), inds [M] int64 frame indices (each must lie in [0, n): checked by the Python mirror), thresh [M] ->
 * count [M,ht,wd]: in how many of the six neighbour frames {ix-1, ix-2, ix-3, ix+3, ix+4, ix+5} the pixel's depth is confirmed. */
int cut3r_depth_filter(const float* poses, const float* disps, const float* intr, const long long* inds, const float* thresh, int n, int M,
                       int ht, int wd, float* count, void* stream);
/* droid_backends.altcorr_forward / altcorr_backward (call sites hislam2/modules/corr.py:79,87): on-the-fly correlation of fmap1
 * [BN,H,W,C] with fmap2 [BN,H2,W2,C] sampled bilinearly in a (2r+1)^2 window around coords [BN,S,H,W,2] ->
 * corr [BN,S,(2r+1)^2,H,W]; backward gives the gradients w.r.t. both feature maps (coords get none). */
int cut3r_altcorr_forward(const float* fmap1, const float* fmap2, const float* coords, int BN, int S, int H, int W, int H2, int W2, int C,
                          int radius, float* corr, void* stream);
int cut3r_altcorr_backward(const float* fmap1, const float* fmap2, const float* coords, const float* grad_corr, int BN, int S, int H, int W,
                           int H2, int W2, int C, int radius, float* grad1, float* grad2, void* stream);

/* ---- stream preprocessing -------------------------------------------------------------------------------------------
 * replaces cv2.resize(image, (w1, h1)) of demo_s.py:72,83 (default INTER_LINEAR, 8-bit): OpenCV's fixed-point bilinear
 * and its exact-2x box special case, restated from the published algorithm (OpenCV is not in the reference tree and cv2 is
 * not in the build image: parity unpinned, see oracle/oracle_geom.c).  src u8 [H0,W0,C] interleaved (device); dst u8
 * [C,H1,W1] when chw_out != 0 (the layout demo_s.py:73 permutes to) else [H1,W1,C]. */
int cut3r_resize_linear_u8(const void* src, int H0, int W0, int C, void* dst, int H1, int W1, int chw_out, void* stream);
/* cv2.undistort's resampling step (demo_s.py:63-64 `cv2.undistort(image, K, calib[4:])` = initUndistortRectifyMap + remap with
 * INTER_LINEAR, BORDER_CONSTANT 0): src/dst uint8 HWC; map_ix/map_iy int32 [Ho,Wo] = source coordinates in 1/32 pixel
 * (round(32 u)), built once per sequence on the host (cut3r_slam_amd/stream.py:undistort_map).  PARITY UNPINNED vs cv2
 * (absent from the build image): follows the published fixed-point algorithm (5 fraction bits, 15-bit weights). */
int cut3r_remap_linear_u8(const void* src, int H, int W, int C, const int32_t* map_ix, const int32_t* map_iy, void* dst, int Ho, int Wo,
                          void* stream);

/* ---- Gaussian-splatting rasteriser ---------------------------------------------------------------
 * replaces diff_gaussian_rasterization._C.rasterize_gaussians (thirdparty/diff-gaussian-rasterization/rasterize_points.cu:55-195,
 * bound at ext.cpp:16; called from diff_gaussian_rasterization/__init__.py:75-84, hislam2/gaussian/renderer/__init__.py:132-140),
 * split into its three stages so that the caller owns every buffer.  All pointers are device pointers except the *_host ones
 * (16 / 16 / 3 / 3 floats read on the host when the call is made).  Matrices as the reference passes them: the TRANSPOSED 4x4s
 * (p_view = [p,1] @ viewmatrix).  geom: P records of 32 floats (xy, view depth, ray distance, conic + opacity, rgb, view point,
 * camera plane, ray plane, normal, radius, tile rectangle): the render stages gather from it.
 *   preprocess: forward.cu:308-421 per Gaussian + inclusive scan of the covered-tile counts -> offsets[P] (offsets[P-1] = number of
 *               instances, read back by the caller to size the binning buffers, as rasterizer_impl.cu:346-354 does).
 *   bin:        rasterizer_impl.cu:70-112,151-176: instance keys (tile << 32 | depth bits), radix sort, per-tile [start, end).
 *               overflow == NULL: n_instances is offsets[P-1] read back by the caller.  overflow != NULL (capacity mode, for callers
 *               that must not stop the host, e.g. inside a captured graph): n_instances is a capacity, unused entries are padded
 *               behind the last tile, *overflow is set to 1 on the device if the scene needed more.
 *   render:     forward.cu:429-692.  color/coord/mcoord/normal [3,H,W]; depth/mdepth/alpha [1,H,W]; n_contrib uint32 [2,H,W];
 *               aux float [2,H,W] (final transmittance, normal length: kept for the backward pass). */
int cut3r_gs_preprocess(int P, const float* means, const float* scales, const float* rots, const float* opacities, const float* shs,
                        int sh_degree, int sh_coeffs, const float* colors_precomp, const float* viewmatrix_host, const float* projmatrix_host,
                        const float* campos_host, int W, int H, float tanfovx, float tanfovy, float kernel_size, float scale_modifier, float* geom,
                        int* radii, unsigned* tiles_touched, unsigned* offsets, void* scan_ws, long long scan_ws_bytes, void* stream);
long long cut3r_gs_workspace_bytes(int P, long long n_instances);
int cut3r_gs_bin(int P, const float* geom, const unsigned* offsets, long long n_instances, int W, int H, unsigned long long* keys_tmp,
                 unsigned* vals_tmp, unsigned long long* keys_sorted, unsigned* point_list, unsigned* ranges, void* sort_ws,
                 long long sort_ws_bytes, int* overflow, void* stream);
int cut3r_gs_render_forward(const unsigned* ranges, const unsigned* point_list, const float* geom, int W, int H, float tanfovx, float tanfovy,
                            const float* bg_host, float* out_color, float* out_coord, float* out_mcoord, float* out_depth, float* out_mdepth,
                            float* out_alpha, float* out_normal, unsigned* n_contrib, float* aux, void* stream);
/* backward of the rasteriser (replaces _C.rasterize_gaussians_backward, rasterize_points.cu:197-330 -> backward.cu:145-1160).
 *   render_backward:     per-pixel gradients of the seven images -> dgeom [P,32]: gradients of the record fields (record layout; it is
 *                        zeroed here; slot 2 carries the |d/dxy| sum of backward.cu:1005).  Needs the forward's geom, point_list,
 *                        ranges, n_contrib, aux and its alpha / coord / depth / normal images.
 *   preprocess_backward: dgeom -> d_means [P,3], d_scales [P,3], d_rots [P,4], d_opacities [P], d_shs [P,sh_coeffs,3] (rows beyond the
 *                        active degree are left untouched: pass zeros) or d_colors [P,3] when shs == NULL, d_means2D [P,3] (x, y in
 *                        NDC units as backward.cu:1002-1003, z the |.| sum).  Gaussians culled by the forward pass are not written:
 *                        pass zero-initialised outputs.  Exact derivatives of the forward function (forward-mode duals). */
int cut3r_gs_render_backward(const unsigned* ranges, const unsigned* point_list, const float* geom, int P, int W, int H, float tanfovx,
                             float tanfovy, const float* bg_host, const unsigned* n_contrib, const float* aux, const float* out_alpha,
                             const float* out_coord, const float* out_depth, const float* out_normal, const float* g_color, const float* g_coord,
                             const float* g_mcoord, const float* g_depth, const float* g_mdepth, const float* g_alpha, const float* g_normal,
                             float* dgeom, void* stream);
int cut3r_gs_preprocess_backward(int P, const float* means, const float* scales, const float* rots, const float* opacities, const float* shs,
                                 int sh_degree, int sh_coeffs, const float* viewmatrix_host, const float* projmatrix_host,
                                 const float* campos_host, int W, int H, float tanfovx, float tanfovy, float kernel_size, float scale_modifier,
                                 const float* geom, const float* dgeom, float* d_means, float* d_scales, float* d_rots, float* d_opacities,
                                 float* d_shs, float* d_colors, float* d_means2D, void* stream);
/* ---- the mapper's training step without an autograd tape (csrc/gs_train.hip) ------------------------------------------------------
 * What the reference spreads over torch autograd per rendered view: `render` moves the Gaussians into the camera frame
 * (hislam2/gaussian/renderer/__init__.py:89-152 with utils/slam_utils.py:93-102 get_pose: pose = exp([tau, phi]) * T_w2c), the
 * activations of scene/gaussian_model.py:77-101, torch.optim.Adam of every parameter group (:374-417) and of the pose increments
 * (hislam2/gs_backend_per_frame.py:451-475), update_pose (slam_utils.py:77-91).
 * pose_state: 32 floats per view -- [0:7] world->camera (t, q_xyzw), [7:13] increment (tau, phi), [13:19] / [19:25] Adam moments, [25] steps.
 *   activate:          theta [P,14] (xyz | DC colour | opacity logit | log scale | quaternion rxyz) -> camera-frame means [P,3], scales
 *                      [P,3], rotations [P,4] (rxyz, pose rotation applied), opacities [P], colours [P,3]: the rasteriser's inputs.
 *   activate_backward: their gradients (cut3r_gs_preprocess_backward's outputs) -> gtheta [P,14] += (NULL: poses only) and pose_sums [16]
 *                      += (NULL: Gaussians only); iso_coef != 0 adds the gradient of iso_coef * sum_visible |s - mean s| / max(3 n_visible, 1)
 *                      (gs_backend_per_frame.py:533-538; radii > 0 = visible, nvis_ws: one float of scratch).
 *   pose_step:         gradient of the increments from pose_sums through exp() + 2 * prior * (2 - *ratio) * delta (the pull of pose_refine,
 *                      :262; ratio NULL: factor 1), Adam (lr_trans for tau, lr_rot for phi), fold != 0: T <- exp(delta) T, delta <- 0.
 *   adam:              GaussianMap.step over n = 14 P elements with the per-column rates lr14 [14] and host-side bias corrections.
 *   map_coef / refine_coef: the scalar algebra between cut3r_pixel_loss_forward / cut3r_refine_loss_forward and their backward passes
 *                      (upstream gradient g, resp. g_rgb / g_var); loss_acc (nullable) += the weighted loss value. */
int cut3r_gs_activate(int P, const float* theta, const float* pose_state, float* means, float* scales, float* rots, float* opac, float* shs,
                      void* stream);
int cut3r_gs_activate_backward(int P, const float* theta, const float* pose_state, const float* d_means, const float* d_scales,
                               const float* d_rots, const float* d_opac, const float* d_shs, const int* radii, float iso_coef, float* nvis_ws,
                               float* gtheta, float* pose_sums, void* stream);
int cut3r_gs_pose_step(float* pose_state, const float* pose_sums, float prior, const float* ratio, float lr_rot, float lr_trans, int fold,
                       void* stream);
int cut3r_gs_adam(long long n, float* theta, float* m, float* v, const float* grad, const float* lr14, float b1, float b2, float bc1, float bc2,
                  float eps, void* stream);
int cut3r_gs_map_coef(const float* sums, float w_rgb, float w_depth, float w_normal, float g, int H, int W, float* coef, float* loss_acc,
                      void* stream);
int cut3r_gs_refine_coef(const float* sums, float g_rgb, float g_var, int H, int W, float* coef, float* ratio_out, float* loss_acc,
                         void* stream);
/* simple_knn._C.distCUDA2 (call sites hislam2/gaussian/scene/gaussian_model.py:191,313; the extension is not vendored in the
 * reference tree): points [P,3] -> out [P], the mean squared distance to the 3 nearest other points.  P >= 4.
 * workspace: cut3r_knn3_chunks(P) * P * 3 floats (the candidates are searched in that many chunks, merged by a second kernel). */
int cut3r_knn3_chunks(int P);
int cut3r_knn3_mean_dist2(const float* points, int P, float* out, float* workspace, void* stream);
/* the same quantity (exact: the three nearest squared distances of the exhaustive search) through a uniform grid, for large maps
 * (P >= ~2e5: `gaussian_reinit` over every keyframe's pointmap, hislam2/gs_backend_per_frame.py:865-944): counting sort by cell, shells of
 * growing Chebyshev radius until the third-best distance is covered.  workspace: cut3r_knn3_grid_workspace_bytes(P) bytes. */
long long cut3r_knn3_grid_workspace_bytes(int P);
int cut3r_knn3_grid_mean_dist2(const float* points, int P, float* out, void* workspace, long long workspace_bytes, void* stream);
/* SSIM of the mapper's colour loss (hislam2/gaussian/utils/loss_utils.py:129-170: 11x11 Gaussian window, sigma 1.5, zero padding,
 * per channel).  forward: a, b [C,H,W] -> ssim_map [C,H,W] and the three partials the backward pass filters (d S / d mu1,
 * d S / d E[a^2], d S / d E[ab]).  backward: grad_a = grad_scale[0] * d(sum ssim_map)/d a  (grad_scale: ONE device float, e.g. the
 * upstream gradient over C*H*W for a mean).  b is treated as constant (the keyframe image). */
int cut3r_ssim_forward(const float* a, const float* b, int C, int H, int W, float* ssim_map, float* d_mu1, float* d_x11, float* d_x12,
                       void* stream);
int cut3r_ssim_backward(const float* a, const float* b, const float* d_mu1, const float* d_x11, const float* d_x12, int C, int H, int W,
                        const float* grad_scale, float* grad_a, void* stream);
/* the mapper's per-pixel loss terms (hislam2/gs_backend_per_frame.py:516-531: colour L1, inverse-depth L1, agreement of the normals
 * of the rendered depth with those of the keyframe depth) in one pass.  img, gt_img [3,H,W]; depth, gt_depth [H,W]; gt_normal [3,H,W]
 * (camera-frame normals of gt_depth, borders zero).  forward: sums[4] = {sum |gt - img|, sum_mask |1/d - 1/gt_d|,
 * sum_mask (1 - n(d) . gt_normal), |mask|}, mask = gt_d > 0.001 and d > 0.001.  backward: coef[3] (device) = upstream gradient times
 * weight over the normaliser of each term -> grad_img [3,H,W], grad_depth [H,W]. */
int cut3r_pixel_loss_forward(const float* img, const float* gt_img, const float* depth, const float* gt_depth, const float* gt_normal, int H,
                             int W, float fx, float fy, float cx, float cy, float* sums, void* stream);
int cut3r_pixel_loss_backward(const float* img, const float* gt_img, const float* depth, const float* gt_depth, const float* gt_normal, int H,
                              int W, float fx, float fy, float cx, float cy, const float* coef, float* grad_img, float* grad_depth,
                              void* stream);
/* global_BA's rendered-normal term (hislam2/gs_backend_per_frame.py:996-1001): mean over ALL pixels of 1 - N . n(depth), N the rendered
 * normal image [3,H,W], n(depth) the camera-frame normal of the rendered depth [H,W] (central differences of the back-projected
 * neighbours, zero on the border).  forward: sum[0] (device) = the pixel sum.  backward: coef = upstream gradient * weight / (H W);
 * grad_normal [3,H,W] is written, grad_depth [H,W] is ACCUMULATED into (after cut3r_pixel_loss_backward wrote it). */
int cut3r_normal_agree_forward(const float* normal, const float* depth, int H, int W, float fx, float fy, float cx, float cy, float* sum,
                               void* stream);
int cut3r_normal_agree_backward(const float* normal, const float* depth, int H, int W, float fx, float fy, float cx, float cy, float coef,
                                float* grad_normal, float* grad_depth, void* stream);
/* densification statistics of one rendered view (hislam2/gaussian/scene/gaussian_model.py:779-790 add_densification_stats and the
 * max_radii2D update of gs_backend_per_frame.py:1021-1027): for visible Gaussians (radii > 0) max_radii2D = max(., radius),
 * grad_accum += |d_means2D.xy|, grad_accum_abs += |d_means2D.z| (the absolute-gradient statistic of the rasteriser's backward, :781), denom += 1. */
int cut3r_gs_densify_stats(int P, const int* radii, const float* d_means2D, float* max_radii2D, float* grad_accum, float* grad_accum_abs,
                           float* denom, void* stream);
/* the pose-refinement loss terms (hislam2/gs_backend_per_frame.py:240-262) in one pass: a = alpha > alpha_th, m = a and both depths >
 * 0.001.  forward: sums[5] = {sum_a |gt - img|, |a|, sum_m diff, sum_m diff^2, |m|}, diff = log d - log gt_d.  backward: coef[3] (device)
 * = {c_rgb, c_var, mean diff} -> grad_img = c_rgb sign(img - gt) on a, grad_depth = 2 c_var (diff - mean) / d on m. */
int cut3r_refine_loss_forward(const float* img, const float* gt_img, const float* depth, const float* gt_depth, const float* alpha,
                              float alpha_th, int H, int W, float* sums, void* stream);
int cut3r_refine_loss_backward(const float* img, const float* gt_img, const float* depth, const float* gt_depth, const float* alpha,
                               float alpha_th, int H, int W, const float* coef, float* grad_img, float* grad_depth, void* stream);

/* Measurement aid of bench.py's roofline (no reference counterpart; not on the product path): a bare MFMA loop (v_mfma_f32_16x16x32_f16,
 * 16 independent accumulator chains per wave, 8 waves per workgroup, `grid` workgroups, `iters` x 16 MFMAs per wave, operands = 16-byte
 * chunks of data[nhalf] fp16, nhalf a power of two >= 32768) with s_memtime / s_memrealtime stamps around the loop.  stamps [grid,2] u64 =
 * (shader cycles, 100-MHz ticks) per workgroup; sink [grid*512] receives the accumulator sums.  FLOP = grid * 8 * iters * 16 * 16384. */
int cut3r_mfma_probe(const void* data, int nhalf, int iters, int grid, float* sink, unsigned long long* stamps, void* stream);

#ifdef __cplusplus
}
#endif
#endif
