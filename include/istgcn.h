/* libistgcn_hip.so -- C ABI of the MI355X (gfx950) IST-GCN hot path.
 *
 * The reference (julycrow/IST-GCN) has no FFI of its own: its hot path is a stack of
 * nn.Module.forward calls dispatching ATen ops.  Each entry point below states the reference
 * call site (file:line under /root/reference) whose arithmetic it replaces; the Python
 * nn.Module mirror in ist-gcn_amd/net/ binds them with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - plain pointers + sizes; the caller (PyTorch) owns every buffer; nothing is allocated,
 *    freed or synchronised here; `stream` is a hipStream_t passed as void*.
 *  - activations are "NTVC": x[n][t][v][c], c innermost (== torch channels_last of the
 *    reference's (N*M, C, T, V) tensors); one n is one (clip, person) sequence.
 *  - dtype: 0 = float32, 1 = bfloat16, 2 = float16 storage of activations and activation gradients; accumulation is
 *    always fp32, parameters / parameter gradients fp32, BatchNorm sums fp64.  (float16 = BASELINE config 5,
 *    net/st_gcn_mstcn_1x1_deep.py; its 5-bit exponent needs a loss scale for the backward pass, see istgcn_sgd_step.)
 *  - return value: 0 ok, 1 invalid argument (nothing launched), 2 launch failure.
 *  - reentrant, no global mutable state: callable from any host thread (nn.DataParallel replicas).
 */
#ifndef ISTGCN_H
#define ISTGCN_H
#ifdef __cplusplus
extern "C" {
#endif

/* Tiling constants the host needs to lay out fragment-ordered weights for a (Cin, Cout, K, dtype)
 * graph-conv problem: input-channel chunk CCeff, number of chunks nch, padded contraction length per
 * chunk KKp, number of 32-row output tiles MTtot, elements per 16-byte lane fragment EPL.
 * Packed weight element ((((ch*MTtot + mt)*NKG + kg)*2 + h)*32 + r)*EPL + j, NKG = KKp/(2*EPL), holds
 *   Wr[c = 32*mt + r][k][i]  with  k*CCeff + (i - ch*CCeff) = kg*2*EPL + h*EPL + j   (zero outside). */
int istgcn_gcn_geometry(int Cin, int Cout, int K, int dtype, int* CCeff, int* nch, int* KKp, int* MTtot,
                        int* EPL);

/* Register-chained layout (round 3, csrc/gcn_rc.hip): for 16-bit storage, Cin in {3 (zero-padded to 16),64,128,256}, Cout % 64 == 0, K <= 4 (and
 * K*Cin*128 bytes <= 100 KB of LDS) the packed weights carry a SECOND section behind the one above, at element offset
 * istgcn_gcn_rc_offset (-1: none; istgcn_gcn_rc_layout: 1 / 0):
 *   element (((jt*K + k)*(Cin/16) + s)*64 + 32*h + c)*8 + e  holds  Wr[32*jt + c][k][16*s + 8*h + e]
 * = the B operand fragments of H_k = x W_k^T (MFMA 32x32x16, channels in natural k order).  float32 storage (Cin in
 * {64,128,256}, K <= 3; csrc/gcn_rc_f32.hip, MFMA 32x32x2 f32): element ((((jt*K + k)*(Cin/64) + q)*8 + s4)*64 + 32*h + c)*4 + e
 * holds Wr[32*jt + c][k][64*q + 32*h + 4*s4 + e].  istgcn_gcn_fwd picks the
 * register-chained kernel at launch when also V <= 32 and not (stats and addend) -- same arithmetic as
 * net/utils/tgcn.py:79-86, GEMM first (the 1x1 Conv2d :79), then the einsum :86 on the accumulator tile in registers. */
int istgcn_gcn_rc_layout(int Cin, int Cout, int K, int dtype);
long long istgcn_gcn_rc_offset(int Cin, int Cout, int K, int dtype);

/* Graph-convolution unit  y[n,t*os,w,c] (+)= sum_k sum_v sum_i Wr[c][k][i] * A[k][v][w] * x[n,t*is,v,i]
 *                                            + bterm[w][c]           for t in [0, Tlog)
 * = ConvTemporalGraphical.forward  net/utils/tgcn.py:76-89  (1x1 Conv2d :79 then einsum :86), and with the
 * folded adjacency also net/utils/tgcn_multi3_fix_3A.py:86-89 and net/utils/inceptionv2_gcn.py:69-80.
 * K=1, A=I, is=stride gives the residual Conv2d(1x1, stride) of net/st_gcnold.py:186-191.  Run on dy with
 * A^T and Wr^T it is the unit's input gradient (autograd of the same lines).
 *   x      [NM][Tin][V][Cin]       A  [K][V][V] fp32, A[k][v][w]       Wp  fragment-ordered (see above)
 *   bterm  [V][Cout] fp32 or NULL  (= sum_k bias[k*Cout+c] * sum_v A[k][v][w], the Conv2d bias pushed
 *                                    through the einsum)
 *   addend NULL or [NM][Tout][V][Cout] added to the result (may alias y: accumulate)
 *   y      [NM][Tout][V][Cout]
 *   stats  NULL or [stats_rep][2][Cout] fp64, += per-channel sum(y), sum(y*y) (train-mode BatchNorm2d that
 *          follows: st_gcnold.py:165); caller zeroes it; replicas are summed by istgcn_bn_finalize
 *   status NULL or one int set to 1 if A has more non-zeros than nnz_cap (result then invalid)
 *   nnz_cap capacity of the in-LDS sparse column lists, >= nnz(A), <= K*V*V.  With fewer slots than non-zeros the
 *          lists are TRUNCATED (wrong y, no fault): pass K*V*V unless the pattern is known, or check `status`. */
int istgcn_gcn_fwd(const void* x, const float* A, const void* Wp, const float* bterm, const void* addend,
                   void* y, double* stats, int stats_rep, int* status, int NM, int Tin, int Tout, int Tlog,
                   int V, int Cin, int Cout, int K, int in_t_stride, int out_t_stride, int nnz_cap,
                   int dtype, int grid_cap, void* stream);

/* Backward of the graph-convolution unit (autograd of net/utils/tgcn.py:79-86 and the folded variants), two launches.
 *
 * istgcn_gcn_bwd_data: dx[n,t,v,i] = sum_k sum_w A[k][v][w] * dxa_k[n,t,w,i] (+ addend),  dxa_k = sum_c W[k][c][i] dy[..,c]
 *                      dA[k][v][w] += sum_{n,t,i} x[n,t,v,i] * dxa_k[n,t,w,i]   only where pattern[k][v][w] != 0
 *                      (pattern [K][V][V] fp32: the constant adjacency B of A = B (.) importance -- the only entries an
 *                      importance gradient B (.) dA can see, and unlike A itself it does not lose an entry when an
 *                      importance value is exactly 0; NULL: the non-zeros of A; all-ones: a dense learnable A as autograd
 *                      of tgcn.py:86 gives).  nnz(pattern) <= nnz_cap.  dA may be NULL (then x may be NULL too).
 *   dy [NM][T][V][Cout], x / addend / dx [NM][T][V][Cin] (addend NULL or the identity-residual gradient, may alias dx),
 *   Wb: fragments of W for the dxa product, element [ich][cch][kg][mt][h][r][e] =
 *       W[k][cch*CCc + kg*2*EPL + h*EPL + e][ich*CCi + il]  with  k*CCi + il = 32*mt + r  (zero padded);
 *       CCi, nchi, CCc, nchc, KKp (rows, = 32*MTK), EPL from istgcn_gcn_bwd_geometry.
 * istgcn_gcn_wgrad:    dW[k][c][i] += sum_{n,t,w} dy[n,t,w,c] * sum_v A[k][v][w] x[n,t,v,i]   (= Conv2d weight grad)
 *                      S[w][c]     += sum_{n,t} dy[n,t,w,c]        (gradient of the bias term; S may be NULL)
 * A [K][V][V] fp32, K <= 4.  dW / dA / S are fp32 and ACCUMULATED (the caller zeroes them). */
int istgcn_gcn_bwd_geometry(int Cin, int Cout, int K, int dtype, int* CCi, int* nchi, int* CCc, int* nchc, int* KKp,
                            int* EPL);
/* Register-chained layout of Wb (round 3, csrc/gcn_rc_bwd.hip): for 16-bit storage, Cout in {64,128,256}, Cin % 64 == 0,
 * K <= 4 (K*Cout*128 bytes <= 100 KB) a second section follows at element offset istgcn_gcn_bwd_rc_offset (-1: none):
 *   element (((it*K + k)*(Cout/16) + s)*64 + 32*h + c)*8 + e  holds  W[k][16*s + 8*h + e][32*it + p(c)],
 *   p(c) = c with bits 2 and 3 swapped.  istgcn_gcn_bwd_data uses the register-chained kernel when also V <= 32.
 * Cin = 3, Cout = 64 (the models' first layer, whose input needs no gradient) also has the section (one zero-padded
 * 32-channel tile); for it dx may be NULL: only dA is computed. */
int istgcn_gcn_bwd_rc_layout(int Cin, int Cout, int K, int dtype);
long long istgcn_gcn_bwd_rc_offset(int Cin, int Cout, int K, int dtype);
/* addend_mask (round 5; NULL or, where istgcn_gcn_bwd_addend_mask_ok(...) == 1: the register-chained kernel): the ReLU byte mask
 * of istgcn_block_out_fwd over the addend's shape -- dx += addend * [bit], i.e. the identity-residual gradient
 * dout * [out > 0] of the st_gcn block (net/st_gcnold.py:181-182,201-203) read from dout itself, so that istgcn_block_out_bwd
 * need not write it (dres == NULL there). */
int istgcn_gcn_bwd_addend_mask_ok(int V, int Cin, int Cout, int K, int dtype);
int istgcn_gcn_bwd_data(const void* dy, const void* x, const float* A, const float* pattern, const void* Wb,
                        const void* addend, const unsigned char* addend_mask, void* dx, float* dA, int NM, int T, int V,
                        int Cin, int Cout, int K, int nnz_cap, int dtype, int grid_cap, void* stream);
int istgcn_gcn_wgrad(const void* dy, const void* x, const float* A, float* dW, float* S, int NM, int T, int V, int Cin,
                     int Cout, int K, int nnz_cap, int dtype, int grid_cap, float* ws, long long ws_floats,
                     void* stream);

/* Temporal (k,1) convolution over the frame axis as an implicit GEMM (and its data gradient):
 *   out[n, out_mul*m + out_off, v, o] = epi( sum_j sum_i Wf[j][o][i] * pre(in[n, in_mul*m + tap_off[j], v, i]) ),
 *   m in [0, Mlog); input frames outside [0, Tin) contribute zero; tap_off must be equally spaced.
 * = nn.Conv2d(C, C, (9,1), (stride,1), (4,0)) net/st_gcnold.py:167-173 with the preceding BatchNorm2d+ReLU
 *   (:165-166) applied on the fly (`pre` = [2][Cin] scale, shift; pre_relu) and the following BatchNorm2d's
 *   (:174) batch sums emitted (`stats`); the three-branch Inception-TCN of
 *   net/st_gcn_multi3_fix_3A_mstcn.py:160-180,212-215 / net/st_gcn_mstcn.py:189-209,242-245 is the same call with
 *   15 host-pre-summed taps.  Forward: in_mul = stride, tap_off[j] = j - pad, out_mul = 1, out_off = 0.
 *   Data gradient: per phase p of the output frames, taps with (p + pad - j) % stride == 0, tap_off = (p+pad-j)/stride,
 *   in_mul = 1, out_mul = stride, out_off = p, Wf[j][i][o] = W[o][i][j].
 * mode 0: epi = + bias[o];            stats += sum(out), sum(out^2)
 * mode 1: epi = * [aux*maux[0]+maux[1] > 0] (ReLU mask of the producer BatchNorm, aux = its input, same shape as out);
 *                                     stats += sum(out), sum(out * (aux - maux[2]) * maux[3])   (BatchNorm backward)
 * mode 2: epi = relu( . + bias[o] + res' ), res' = aux*maux[0] + maux[1] (aux NULL: 0; maux NULL: res' = aux); no stats.
 *         Inference tail of the block, net/st_gcnold.py:174-175,201-203 with tcn.3's (eval-mode, hence affine) BatchNorm
 *         folded into Wf / bias by the host and Dropout the identity: an eval-mode st_gcn block is istgcn_gcn_fwd +
 *         this launch (+ the strided 1x1 residual conv where the block has one).
 * Wp: fragment-ordered weights, element (((((ch*ntaps + j)*NKG + kg)*MTtot + mt)*2 + h)*32 + r)*EPL + e holds
 *   Wf[j][32*mt + r][ch*CC + kg*2*EPL + h*EPL + e]; CC, nch, MTtot, EPL from istgcn_tconv_geometry, NKG = CC/(2*EPL). */
int istgcn_tconv_geometry(int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int dtype, int* CC,
                          int* nch, int* MTtot, int* EPL);
int istgcn_tconv(const void* in, const void* Wp, const float* bias, const float* pre, int pre_relu, const void* aux,
                 const float* maux, void* out, double* stats, int stats_rep, int mode, int NM, int Tin, int Tout,
                 int Mlog, int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int out_mul,
                 int out_off, int dtype, int grid_cap, void* stream);

/* Weight / bias gradient of the temporal convolution (autograd of net/st_gcnold.py:167-173 and of the pre-summed
 * Inception-TCN): dW[j][o][i] += sum_{n,m,v} dz[n,m,v,o] * pre(g[n, in_mul*m + tap_off[j], v, i]),  dbias[o] += sum dz.
 *   dz [NM][Tz][V][Cout]   g [NM][Tin][V][Cin]   pre [2][Cin] scale, shift (+ReLU) or NULL
 *   dW [ntaps][Cout][Cin] fp32 and dbias [Cout] fp32 (or NULL) are ACCUMULATED into: the caller zeroes them. */
int istgcn_tconv_wgrad(const void* dz, const void* g, const float* pre, int pre_relu, float* dW, float* dbias,
                       int NM, int Tin, int Tz, int V, int Cin, int Cout, int ntaps, const int* tap_off,
                       int in_mul, int dtype, int grid_cap, float* ws, long long ws_floats, void* stream);

/* Weight packers: fp32 parameter views (element strides given, read in place) -> the fragment orders above, zero padded
 * and cast to `dtype`; one launch per weight.  The *_elems functions return the number of dst elements (or -1).
 *   istgcn_pack_gcn:     Wr[c][k][i] at src + c*s_o + k*s_k + i*s_i  (nn.Conv2d(Cin, K*Cout, 1).weight viewed [K][Cout][Cin]:
 *                        net/utils/tgcn.py:55-63) -> Wp of istgcn_gcn_fwd
 *   istgcn_pack_tconv:   Wf[tap_sel[j]][o][i] at src + t*s_t + o*s_o + i*s_i for the ntaps packed taps
 *                        (nn.Conv2d(C, C, (k,1)).weight [o][i][t][1]: s_o = Cin*k, s_i = k, s_t = 1; the data gradient
 *                        swaps s_o / s_i and selects the taps of its output phase) -> Wp of istgcn_tconv
 *   istgcn_pack_gcn_bwd: W3[k][c][i] at src + k*s_k + c*s_c + i*s_i -> Wb of istgcn_gcn_bwd_data
 * The gcn contraction length K*CCeff is padded (with zero columns) to whole rounds of the kernel's weight ring inside
 * the layout itself (istgcn_gcn_geometry's KKp). */
long long istgcn_pack_gcn_elems(int Cin, int Cout, int K, int dtype);
int istgcn_pack_gcn(const float* src, long long s_o, long long s_k, long long s_i, void* dst, int Cin, int Cout, int K,
                    int dtype, void* stream);
long long istgcn_pack_tconv_elems(int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int dtype);
int istgcn_pack_tconv(const float* src, long long s_t, long long s_o, long long s_i, const int* tap_sel, void* dst, int V,
                      int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int dtype, void* stream);
long long istgcn_pack_gcn_bwd_elems(int Cin, int Cout, int K, int dtype);
int istgcn_pack_gcn_bwd(const float* src, long long s_k, long long s_c, long long s_i, void* dst, int Cin, int Cout, int K,
                        int dtype, void* stream);

/* All weight packs of a model in ONE launch (a training step repacks every weight after the optimiser update: ~46 tiny
 * launches for a 10-block model).  The host fills one opaque record of istgcn_pack_job_bytes() bytes per weight with the
 * istgcn_pack_job_* functions (arguments as for the single-weight packers above; they return the number of workgroups
 * the job needs, or < 0), uploads the records and the exclusive prefix sums block_start[njobs] of those counts, and
 * launches istgcn_pack_batch whenever the parameters have changed.  Pointers inside the records must stay valid. */
int istgcn_pack_job_bytes(void);
int istgcn_pack_job_gcn(void* rec, const float* src, long long s_o, long long s_k, long long s_i, void* dst, int Cin, int Cout,
                        int K, int dtype);
int istgcn_pack_job_tconv(void* rec, const float* src, long long s_t, long long s_o, long long s_i, const int* tap_sel,
                          void* dst, int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int dtype);
int istgcn_pack_job_gcn_bwd(void* rec, const float* src, long long s_k, long long s_c, long long s_i, void* dst, int Cin,
                            int Cout, int K, int dtype);
int istgcn_pack_batch(const void* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, int dtype, void* stream);

/* Parameter folds of the graph-conv unit (one launch each way; everything fp32):
 *   A_eff = sum_j B_j (.) imp_j   (J = 1: A*importance, st_gcnold.py:86; J = 3: the Inception-GCN sum of
 *           st_gcn_msgcn.py:116-117 or the elementwise powers of tgcn_multi3_fix_3A.py:86-89; B = [J][K][V][V])
 *   bterm[w][c] = sum_k bias[k*C+c] * sum_v A_eff[k][v][w]   (Conv2d bias pushed through the einsum, tgcn.py:79-86)
 * and their gradients: dimp_j = B_j (.) (dA + dcol broadcast over v), dbias (see csrc/fold.hip). K*V <= 512 and K*V*V <= 12288. */
int istgcn_fold_fwd(const float* B, int J, const float* imp0, const float* imp1, const float* imp2, const float* bias,
                    float* A_eff, float* bterm, int K, int V, int C, void* stream);
int istgcn_fold_bwd(const float* B, int J, const float* imp0, const float* imp1, const float* imp2, const float* bias,
                    const float* dA, const float* S, float* dimp0, float* dimp1, float* dimp2, float* dbias, int K, int V,
                    int C, void* stream);

/* The same folds for all nb <= 16 blocks of a model in one launch each way (one workgroup per block).  imps / dimps are
 * [nb][3] pointer tables (entries >= J unused), biases / A_eff / bterm / dA / S / dbias [nb] pointer tables (biases, bterm,
 * dA, S, dbias may be NULL or hold NULL entries with the meaning above), Cs [nb] the blocks' output channel counts; B, J, K,
 * V are shared.  The tables are host arrays read at launch. */
int istgcn_fold_fwd_batch(int nb, const float* B, int J, const float* const* imps, const float* const* biases,
                          float* const* A_eff, float* const* bterm, const int* Cs, int K, int V, void* stream);
int istgcn_fold_bwd_batch(int nb, const float* B, int J, const float* const* imps, const float* const* biases,
                          const float* const* dA, const float* const* S, float* const* dimps, float* const* dbias,
                          const int* Cs, int K, int V, void* stream);

/* The three-branch Inception-TCN folded into ONE 15-tap convolution (linear in the weights), and its gradient:
 *   taps[j][o][i] = scale*(m0*W1[o][i][j-6] + m1*W2[o][i][j-3] + m2*W3[o][i][j]),  bias = scale*(m0*b1 + m1*b2 + m2*b3)
 *   (net/st_gcn_multi3_fix_3A_mstcn.py:160-180,212-215 with scale = 1; net/st_gcn_mstcn.py:189-209,242-245 with 1/3).
 * W_s are the nn.Conv2d(C, C, (k,1)) weights [Co][Ci][k] for k = 3, 9, 15 (contiguous), mst [3] the branch importances.
 * bwd: dW_s, db_s written; dmst [3] ACCUMULATED (caller zeroes). */
int istgcn_tcn_fold_fwd(const float* w1, const float* w2, const float* w3, const float* b1, const float* b2, const float* b3,
                        const float* mst, float scale, float* taps, float* bias, int Co, int Ci, void* stream);
int istgcn_tcn_fold_bwd(const float* dtaps, const float* dbias, const float* w1, const float* w2, const float* w3,
                        const float* b1, const float* b2, const float* b3, const float* mst, float scale, float* dw1,
                        float* dw2, float* dw3, float* db1, float* db2, float* db3, float* dmst, int Co, int Ci,
                        void* stream);

/* BatchNorm2d bookkeeping (train-mode statistics are batch sums the MFMA kernels emit in their epilogues).
 * istgcn_bn_finalize: stats [rep][2][C] fp64 (sum, sum of squares) over `count` elements per channel ->
 *   coef [4][C] fp32 = scale (gamma*rstd), shift (beta - mean*scale), mean, rstd; training != 0 also updates
 *   running_mean / running_var in place (momentum, unbiased variance) exactly as nn.BatchNorm2d does
 *   (net/st_gcnold.py:165,174,192); training == 0 derives coef from the running statistics (eval mode).
 * istgcn_bn_bwd_coef: stats [rep][2][C] = (sum d, sum d*xhat) -> abc [3][C] with dx = abc0*d + abc1*x + abc2,
 *   dgamma = sum d*xhat, dbeta = sum d (autograd of BatchNorm2d); training == 0: dx = gamma*rstd*d.
 * clear != 0: the sums are zeroed as they are read, so one scratch buffer serves every producer in turn without a
 *   memset launch between uses. */
int istgcn_bn_finalize(double* stats, int stats_rep, int clear, double count, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float momentum, float eps, int training,
                       float* coef, int C, void* stream);
int istgcn_bn_bwd_coef(double* stats, int stats_rep, int clear, double count, const float* gamma, const float* coef,
                       int training, float* abc, float* dgamma, float* dbeta, int C, void* stream);

/* Tail of the st_gcn block, net/st_gcnold.py:174-175 + 201-203 (tcn.3 BatchNorm, tcn.4 Dropout, + residual, ReLU):
 *   out = relu( dropout_p( z*coef2[0] + coef2[1] ) + res' ),  res' = res (identity) or res*coefr[0] + coefr[1]
 *   (the BatchNorm of the strided 1x1 residual branch, :186-193) or 0 (res == NULL).  rows = NM*T*V, C channels.
 * Dropout: counter-based Philox4x32-10 keyed by (seed, flat element index); the same (p, seed) in the backward
 * calls regenerates the mask.  p = 0 disables it.  seed_epoch (device pointer or NULL): *seed_epoch is added to `seed`
 * when the kernel RUNS, so launches recorded once in a hipGraph draw a fresh mask per replay (harness.GraphedStep).
 * istgcn_block_out_bwd: dres = dout * [out > 0] (the gradient of both the residual branch and, after dropout, of
 *   tcn.3); stats2 += (sum dres*mask, sum dres*mask*zhat); statsr += (sum dres, sum dres*rhat) when the residual
 *   branch has a BatchNorm (r = its input, coefr = its coef[4][C]); r == NULL otherwise.
 *   dres may be NULL when relu_mask is given (round 5): the sums only; every consumer of dres then takes dout and the
 *   byte mask itself (istgcn_affine2m, the addend_mask of istgcn_gcn_bwd_data) -- one tensor write less per block.
 * istgcn_affine2: out = abc[0]*d*mask + abc[1]*x + abc[2]  (elementwise part of BatchNorm backward; x may be NULL).
 * istgcn_affine2m: the same with d := d * [bit of relu_mask] first (relu_mask NULL: istgcn_affine2).
 * relu_mask (optional, both directions; only where istgcn_relu_mask_ok(C, dtype) == 1): [rows * C / vector width] bytes,
 *   bit j of byte i = (element j of the i-th 16-byte vector of `out`, as stored, is > 0).  The forward writes it; given
 *   to the backward it replaces the read of `out` (which may then be NULL): 1/16 of that tensor's bytes. */
int istgcn_relu_mask_ok(int C, int dtype);
int istgcn_block_out_fwd(const void* z, const float* coef2, const void* res, const float* coefr, void* out,
                         unsigned char* relu_mask, long long rows, int C, float p_drop, unsigned long long seed,
                         const unsigned long long* seed_epoch, int dtype, void* stream);
int istgcn_block_out_bwd(const void* dout, const void* out, const unsigned char* relu_mask, const void* z,
                         const float* coef2, const void* r, const float* coefr, void* dres, double* stats2,
                         double* statsr, int stats_rep, long long rows, int C, float p_drop, unsigned long long seed,
                         const unsigned long long* seed_epoch, int dtype, void* stream);
int istgcn_affine2(const void* d, const void* x, const float* abc, void* out, long long rows, int C, float p_drop,
                   unsigned long long seed, const unsigned long long* seed_epoch, int dtype, void* stream);
int istgcn_affine2m(const void* d, const unsigned char* relu_mask, const void* x, const float* abc, void* out, long long rows,
                    int C, float p_drop, unsigned long long seed, const unsigned long long* seed_epoch, int dtype, void* stream);

/* Global average pooling of the trunk output and its backward (net/st_gcnold.py:89-91: F.avg_pool2d over (T, V), then the
 * mean over the M persons of a clip).  y [NM][P = T*V][C] in `dtype`.
 * istgcn_pool_fwd: psum [NM][S][C] fp32 = sums of y over S row slices of every sequence (written, not accumulated); the
 *   clip feature is the sum of its M*S partial rows divided by M*P.
 * istgcn_pool_bwd: dy[nm][p][c] = scale * dfeat[nm / M][c] for all p (dfeat [NM/M][C] fp32; scale = 1/(M*P)). */
int istgcn_pool_fwd(const void* y, float* psum, int NM, int P, int C, int S, int dtype, void* stream);
int istgcn_pool_bwd(const float* dfeat, void* dy, int NM, int P, int C, int M, float scale, int dtype, void* stream);

/* Input stage: the feeder's augmentation (feeder/tools.py:31-101) and the data_bn prologue (net/st_gcnold.py:74-80) on
 * the GPU.  raw [N][C][Traw][V][M] fp32 is the clip batch as the reference's DataLoader delivers it;
 *   shift [N] int or NULL: source frame of output frame t is t + shift[n], frames outside [0, Traw) are zero
 *         (auto_pading tools.py:31-41: shift 0, T > Traw; random_choose :44-57: crop shift = +begin, pad shift = -begin);
 *   move  [N][T][6] fp64 or NULL: per (clip, frame) affine (m00, m01, tx, m10, m11, ty) applied to channels 0,1 AFTER
 *         the shift: x' = m00*x + m01*y + tx, y' = m10*x + m11*y + ty, in double, rounded once to fp32
 *         (random_move tools.py:60-101; the host draws the node values with the reference's generator calls and
 *         interpolates them per frame, see ist-gcn_amd/feeder_gpu.py).
 * istgcn_feeder_augment: out [N][C][T][V][M] fp32 = the augmented clips (what the reference's feeder would have yielded).
 * istgcn_input_stats:    stats [rep][2][V*C] fp64 += per BatchNorm1d channel (v*C + c) sum / sum of squares of the
 *                        augmented clips over (n, m, t)   (train-mode data_bn; finish with istgcn_bn_finalize, C' = V*C).
 * istgcn_input_apply:    out [N*M][T][V][C] (dtype) = x*coef[0][v*C+c] + coef[1][v*C+c]: BatchNorm affine + the two
 *                        permutes of st_gcnold.py:75-80 in one pass.
 * istgcn_input_bwd:      stats += (sum dy, sum dy*xhat) per channel from dout [N*M][T][V][C] and coef [4][V*C]
 *                        (scale, shift, mean, rstd); finish with istgcn_bn_bwd_coef -> d(data_bn.weight), d(data_bn.bias).
 * C*V*M <= 1024. */
int istgcn_feeder_augment(const float* raw, const int* shift, const double* move, float* out, int N, int C, int Traw,
                          int T, int V, int M, void* stream);
int istgcn_input_stats(const float* raw, const int* shift, const double* move, double* stats, int stats_rep, int N, int C,
                       int Traw, int T, int V, int M, void* stream);
int istgcn_input_apply(const float* raw, const int* shift, const double* move, const float* coef, void* out, int N, int C,
                       int Traw, int T, int V, int M, int dtype, void* stream);
int istgcn_input_bwd(const float* raw, const int* shift, const double* move, const void* dout, const float* coef,
                     double* stats, int stats_rep, int N, int C, int Traw, int T, int V, int M, int dtype, void* stream);

/* SGD with momentum / Nesterov / weight decay over ONE flat fp32 range = torch.optim.SGD as configured at
 * processor/recognition.py:154-159 and stepped at :289, for all live parameters of the model in one launch:
 *   g' = grad_scale*g + weight_decay*p;  m = momentum*m + g';  p -= lr*(nesterov ? g' + momentum*m : m)
 * params / grads / momentum_buf: n floats each, 16-byte aligned (views of the host's three flat buffers; grads is the
 * buffer the data-parallel all-reduce ran on).  grad_scale folds the 1/world of a SUM all-reduce and the 1/loss_scale
 * of float16 training into the update.  momentum_buf starts at zero (first step: m = g', as torch initialises it).
 * found_inf (device int or NULL): set to 1 if any gradient element is inf / NaN; such elements leave their parameter
 * and momentum untouched (torch.optim.SGD would write the NaN into both for good).
 * skip_if (device int or NULL): when *skip_if != 0 at launch the WHOLE step is a no-op (and *found_inf is raised) -- the
 * flag istgcn_grad_nonfinite computed over the same gradient buffer, i.e. torch.cuda.amp.GradScaler's "skip the step on
 * overflow" for the float16 configuration (static loss scale, BASELINE config 5). */
int istgcn_sgd_step(float* params, const float* grads, float* momentum_buf, long long n, float lr, float momentum,
                    float weight_decay, int nesterov, float grad_scale, int* found_inf, const int* skip_if, void* stream);
/* *flag |= 1 if any of the n gradients is inf / NaN (flag: device int the caller zeroed; grads 16-byte aligned). */
int istgcn_grad_nonfinite(const float* grads, long long n, int* flag, void* stream);

/* Bottleneck temporal unit of the "1x1" models (16-bit storage; csrc/bneck_rc.hip), replacing the reference's
 *   tcn_start -> conv_1x1_start -> tcn_1/2/3 (x mstcn_importance, summed) -> conv_1x1_end -> tcn_end sums
 *   (net/st_gcn_mstcn_1x1.py:174-213, 258-265; net/st_gcn_mstcn_1x1_deep.py likewise) and its autograd:
 * istgcn_bneck_in:   y[p][n] = sum_c W(n,c) * pre(x[p][c]) + bias[n]   for n < Wn, zeros for Wn <= n < Wp
 *                    x: [rows][C], y: [rows][Wp]; W(n,c) = W[n*w_rs + c*w_cs] fp32 (any strides: the transposed use needs
 *                    no copy); pre = optional [2][C] affine (+ ReLU): the BatchNorm2d + ReLU in front (tcn_start).
 *                    forward q = Ws u + bs;  backward dyb = We^T dz.
 * istgcn_bneck_out:  yb[n, out_mul*m + out_off, v, :] = sum_j Wt(j,:,:) q[n, in_mul*m + off0 + j, v, :] + bt    (narrow, saved)
 *                    z [n, out_mul*m + out_off, v, o] = epi( sum_n' We(o,n') yb[...][n'] + be[o] )
 *                    for m in [0, Mlog); frames outside [0, Tin) are zeros (the Conv2d padding).  Taps are consecutive
 *                    input frames: tap j of the launch is weight slice tap_sel[j], Wt(j,r,c) = Wt[tap_sel[j]*wt_ts + r*wt_rs
 *                    + c*wt_cs]; We(o,n') = We[o*we_rs + n'*we_cs].  ntaps <= 15.  in_mul = 2 only with mode 0.
 *                    mode 0: epi = identity, stats += sum(z), sum(z^2)            (forward: yb, z and tcn_end's batch sums)
 *                    mode 1: epi = * [aux*maux[0]+maux[1] > 0], stats += sum(z), sum(z * (aux - maux[2]) * maux[3])
 *                            (backward through tcn_start: q := dyb, yb := dq, z := d1, aux = the graph conv's output)
 * Shapes: istgcn_bneck_ok -- V <= 32, C in {64, 128, 256}, 1 <= Wn <= Wp, Wp in {8, 16}, dtype 1 or 2. */
int istgcn_bneck_ok(int V, int C, int Wn, int Wp, int dtype);
/* istgcn_bneck_wgrad: weight gradients of the two 1x1 convolutions of the chain, accumulated (+=) into fp32 buffers:
 *   dW(n, c) += sum_p nrw[p][n] * pre(wide[p][c]),  stored [C][Wp] (wide_is_out = 1: conv_1x1_end, wide = dz, nrw = yb)
 *   or [Wp][C] (wide_is_out = 0: conv_1x1_start, wide = the graph conv's output behind `pre`, nrw = dq);
 *   db += sum_p wide[p][:] ([C], db_wide = 1) or sum_p nrw[p][:] ([Wp], db_wide = 0); db may be NULL.
 *   ws: partial-sum workspace of ws_floats floats (one slice per workgroup + the reduce kernel of the other wgrad kernels). */
/* istgcn_bneck_bwd_in: the backward pass INTO the chain as ONE stream (C = 64 / 128: istgcn_bneck_bwd_in_ok) -- the
 * elementwise half of tcn_end's BatchNorm backward, dz = abc[0][c]*dropmask*dres + abc[1][c]*z + abc[2][c] (what
 * istgcn_affine2 writes as a tensor; same Philox stream: p_drop, seed, seed_epoch), formed in registers and rounded to the
 * storage type, then dyb = W dz (W(n,c) = W[n*w_rs + c*w_cs]: pass We^T), dW [C][Wp] += dz^T yb, db [C] += sum dz.
 * Replaces istgcn_affine2 + istgcn_bneck_in + istgcn_bneck_wgrad (5 passes over wide tensors) by 2. */
int istgcn_bneck_bwd_in_ok(int C, int Wn, int Wp, int dtype);
int istgcn_bneck_bwd_in(const void* dres, const void* z, const float* abc, float p_drop, unsigned long long seed,
                        const unsigned long long* seed_epoch, const void* yb, const float* W, long long w_rs, long long w_cs,
                        void* dyb, float* dW, float* db, long long rows, int C, int Wn, int Wp, int dtype, int grid_cap,
                        float* ws, long long ws_floats, void* stream);
/* istgcn_bneck_wgrad_taps: weight gradient of the narrow temporal conv (tcn_1/2/3 pre-summed):
 *   dW[j][n'][n] += sum_{seq, m, v} dy[seq, m, v, n'] * q[seq, in_mul*m + off0 + j, v, n]   (frames outside [0, Tin): zeros)
 *   db[n'] += sum dy;  dy: [NM][Tz][V][Wp], q: [NM][Tin][V][Wp], dW: fp32 [ntaps][Wp][Wp], db: fp32 [Wp] or NULL. */
int istgcn_bneck_wgrad_taps(const void* dy, const void* q, float* dW, float* db, int NM, int Tin, int Tz, int V, int Wp,
                            int ntaps, int off0, int in_mul, int dtype, int grid_cap, float* ws, long long ws_floats,
                            void* stream);
int istgcn_bneck_wgrad(const void* wide, const void* nrw, const float* pre, int pre_relu, float* dW, float* db,
                       int wide_is_out, int db_wide, long long rows, int C, int Wp, int dtype, int grid_cap, float* ws,
                       long long ws_floats, void* stream);
int istgcn_bneck_in(const void* x, const float* W, long long w_rs, long long w_cs, const float* bias, const float* pre,
                    int pre_relu, void* y, long long rows, int C, int Wn, int Wp, int dtype, int grid_cap, void* stream);
int istgcn_bneck_out(const void* q, const float* Wt, long long wt_ts, long long wt_rs, long long wt_cs, const int* tap_sel,
                     int ntaps, int off0, const float* bt, void* yb, const float* We, long long we_rs, long long we_cs,
                     const float* be, void* z, const void* aux, const float* maux, double* stats, int stats_rep, int mode,
                     int NM, int Tin, int Tout, int Mlog, int V, int C, int Wn, int Wp, int in_mul, int out_mul, int out_off,
                     int dtype, int grid_cap, void* stream);

/* "Last workgroup finalises" (csrc/bn_tail.hpp): the arithmetic of istgcn_bn_finalize / istgcn_bn_bwd_coef as the TAIL of
 * the kernel that produces the batch sums.  istgcn_bn_tail_arm_* arms one tail for the calling host thread; the next launch
 * (same thread) of istgcn_gcn_fwd (register-chained variant), istgcn_tconv (wave-specialised variant), istgcn_bneck_out or
 * istgcn_block_out_bwd whose `stats` (stats2) pointer equals the armed one carries it: after its last workgroup's sums the
 * outputs are written, the running statistics updated, `stats` zeroed again and the ticket reset.  istgcn_bn_tail_disarm
 * returns 1 if the tail was NOT taken (a kernel variant without tails: launch the stand-alone entry point), else 0.
 * ticket: zero-initialised device word, one per stats buffer. */
int istgcn_bn_tail_arm_finalize(double* stats, int stats_rep, double count, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, float* coef, int C,
                                unsigned* ticket);
int istgcn_bn_tail_arm_bwd(double* stats, int stats_rep, double count, const float* gamma, const float* coef, int training,
                           float* abc, float* dgamma, float* dbeta, int C, unsigned* ticket);
int istgcn_bn_tail_disarm(void);

/* Test-only probes of the hardware conventions the kernels assume (MFMA lane maps, ds_read_b64_tr_b16). */
int istgcn_probe_mfma(const void* A, const void* Bt, float* D, int dtype, void* stream);
int istgcn_probe_tr16(const void* src, int nelem, const int* lane_byte_off, void* out, void* stream);

/* Dispatch trace (csrc/trace.hip; test instrumentation, own design -- the reference has no counterpart).  Off by default.
 * op 1: switch the trace on and clear it; op 0: off and clear; op 2: write "count<TAB>kernel symbol\n" for every distinct
 * kernel the library launched since the last clear into buf (NUL-terminated, at most cap bytes) and return the size of the
 * full dump.  The parity tests use it to assert WHICH kernel variant served a shape (tests/gpu_util.py: launched()). */
int istgcn_trace(int op, char* buf, int cap);

#ifdef __cplusplus
}
#endif
#endif
