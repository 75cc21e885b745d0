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
 *  - dtype: 0 = float32, 1 = bfloat16 storage; accumulation is always fp32, BatchNorm sums fp64.
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
 *   nnz_cap capacity of the in-LDS sparse column lists, >= nnz(A), <= K*V*V */
int istgcn_gcn_fwd(const void* x, const float* A, const void* Wp, const float* bterm, const void* addend,
                   void* y, double* stats, int stats_rep, int* status, int NM, int Tin, int Tout, int Tlog,
                   int V, int Cin, int Cout, int K, int in_t_stride, int out_t_stride, int nnz_cap,
                   int dtype, int grid_cap, void* stream);

#ifdef __cplusplus
}
#endif
#endif
