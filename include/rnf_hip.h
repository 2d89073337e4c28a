/*
 * rnf_hip.h -- C ABI of librnf_hip.so: the MI355X (gfx950) implementation of the SO(3) normalizing-flow density path
 * of PKU-EPIC/RotationNormFlow.  Plain pointers and sizes only; no framework types.  All `dev` pointers are HIP
 * device pointers, all other pointers are host pointers.  Every call is stream-ordered on `stream` (a hipStream_t
 * passed as void*; NULL = the default stream) and performs no host synchronisation.
 *
 * Return value: 0 on success, non-zero on error; rnf_last_error() returns a thread-local message.
 *
 * The reference has no FFI for this path (it is 100% Python); each entry point cites the reference code it replaces
 * (paths relative to the reference tree).  INTEGRATION.md shows the ctypes binding and the module-level drop-in.
 */
#ifndef RNF_HIP_H
#define RNF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RNF_ABI_VERSION 7

/* width of the conditioner MLP's hidden layers: flow/condition.py:9 (Nh=64, never overridden by any caller) */
#define RNF_HIDDEN 64

/* ---- layer table -------------------------------------------------------------------------------------------
 * A flow is described by an int32 table desc[n_layers][RNF_DESC_STRIDE] (host memory) in Flow.layers order
 * (flow/flow.py:36-51) plus one float32 parameter blob (device memory) built with the rnf_pack_* functions.
 *   desc[i][0] kind          RNF_LAYER_*
 *   desc[i][1] perm_row      row of the 6x3 permutation table used by this layer on the FORWARD pass
 *                            (flow/flow.py:13-15,64-70); the inverse pass uses the same row for the same layer
 *   desc[i][2] param_offset  offset of the layer's packed parameters in the blob, in floats (multiple of 4)
 *   desc[i][3] cond_slot     index of this layer among the layers that consume the feature vector, or -1
 *   desc[i][4] feat_offset   offset in the blob of the layer's packed feature-projection weights, or -1
 *   desc[i][5] precision     bits 0..7: RNF_PREC_* the layer's weight image was packed with (same for every MLP layer); bits 8..15
 *                            (ABI v7): RNF_PREC_* of the FALLBACK records of columns 6, 7 -- RNF_PREC_FP32 (0) or RNF_PREC_BF16X3;
 *                            bits 16..17 (ABI v7, RNF_LAYER_MOBIUS): order of the FIRST pass of the inverse root finder that stands in
 *                            for BinFind (flow/mobiusflow.py:189-224) -- 0 the library's default (third order), 1 third order, 2 fourth
 *                            order (pays on sharply peaked conditioner outputs, i.e. trained conditional flows); one value per flow
 *                            (the largest code of its layers), the returned grid cell is the same for either
 *   desc[i][6] fallback param_offset   } RNF_PREC_F16X2 flows only: offsets in the SAME blob of the layer's records packed with a strict
 *   desc[i][7] fallback feat_offset    } arithmetic (or -1; the feature-projection record is the RNF_PREC_FP32 image for either).  When
 *                            every MLP layer has them, each rnf_flow_forward / _inverse /
 *                            _log_prob call is GUARDED: a sample that ends non-finite (an fp16 operand left the fp16 range, |x| >= 65504;
 *                            flow/condition.py:24-30 has no such limit) sets a device flag and the strict kernels, launched right
 *                            behind on the same stream, redo the chunk -- they return at once when the flag is clear.  No host
 *                            synchronisation; int32 word 1 of the last 8 bytes of the first 32 KiB of the workspace is 1 after a call in
 *                            which the re-run happened.
 */
#define RNF_DESC_STRIDE 8
#define RNF_LAYER_MOBIUS 1        /* flow/mobiusflow.py:27-183  MobiusFlow                                  */
#define RNF_LAYER_AFFINE16 2      /* flow/squeezetrans.py:161-174 Uncondition16Trans (any constant 4x4 M)   */
#define RNF_LAYER_AFFINE16_COND 3 /* flow/squeezetrans.py:41-55  Condition16Trans (M = I + MLP(feature))    */
#define RNF_LAYER_GS9 4           /* flow/squeezetrans.py:250-261 Uncondition9Trans (Gram-Schmidt of M R)      */
#define RNF_LAYER_GS36 5          /* flow/squeezetrans.py:350-361 Uncondition36Trans (6x6 on two columns)      */
/* conditional 3x3 layers, M = I + reshape(MLP(feature), 3, 3) per sample (records from rnf_pack_cond9): */
#define RNF_LAYER_COND9_GS 6      /* flow/squeezetrans.py:234-247 Condition9Trans                              */
#define RNF_LAYER_COND9_SMITH 7   /* flow/rottrans.py:168-181     Condition9RotRSmith                          */
#define RNF_LAYER_COND9_POLAR_L 8 /* flow/rottrans.py:108-121     Condition9RotL                               */
#define RNF_LAYER_COND9_POLAR_R 9 /* flow/rottrans.py:138-151     Condition9RotR                               */
#define RNF_LAYER_COND36 10       /* flow/squeezetrans.py:334-347 Condition36Trans (record from rnf_pack_cond36) */
/* layers whose per-sample matrix the CALLER builds and hands in (rnf_flow_*_side; desc param_offset = slot in the side buffer, no blob
 * record): the reference forms these matrices with batched torch ops that are not a per-sample function -- ConditionRot's U^T V of a
 * batched SVD (flow/rottrans.py:37-66; depends on the SVD routine's sign conventions) and ConditionLU's torch.diag over the BATCH
 * dimension (flow/squeezetrans.py:121-131) -- so the host side reproduces them with the same torch calls on outputs of
 * rnf_cond_mlp_forward and the kernels apply the result: */
#define RNF_LAYER_SIDE16 11       /* flow/squeezetrans.py:134-144 Condition16TransLU: calculate_16 with the given 4x4 (log|det|) */
#define RNF_LAYER_SIDE16_ROT 12   /* flow/rottrans.py:37-66     ConditionRot: the given orthogonal 4x4 on the quaternion, log-det 0 */
#define RNF_LAYER_SIDE9 13        /* flow/squeezetrans.py:264-277 Condition9TransLU: calculate_9 with the given 3x3 (first 9 of 16 floats) */

/* ---- arithmetic of the conditioner GEMMs --------------------------------------------------------------------
 * RNF_PREC_FP32  : exact fp32 (v_mfma_f32_32x32x2_f32; bit-for-bit an fp32 fma chain).
 * RNF_PREC_F16X2 : every fp32 operand carried as two fp16 terms (hi + unscaled lo: 22 significant bits while |x| >= 2^-3, absolute
 *                  resolution 2^-25 below), three fp16 MFMAs into one fp32 accumulator per product-sum; ~2.8x faster than the fp32-input
 *                  MFMA, which shares the VALU's FMA datapath.  The absolute floor is kept 2^-22 below the signal of EVERY layer by the
 *                  packers: a ReLU network computes the same function under per-unit power-of-two rescalings of its hidden layers, and
 *                  rnf_pack_* / rnf_pack_flow_device first move the layer to the point of that orbit where every hidden pre-activation
 *                  has an estimated rms in (1/4, 1/2] (csrc/equalize.h; exact, power-of-two factors), THEN split.  The host packers
 *                  audit the packed image against the exact network on probe inputs (rnf_last_pack_audit) and return 2 -- pack
 *                  RNF_PREC_FP32 instead -- when it is off by more than 4e-6, or when a (scaled) weight leaves the fp16 range
 *                  (|x| < 65504).  Activations beyond the fp16 range at run time are caught by the range guard (see desc columns 6, 7).
 * RNF_PREC_BF16X3: (ABI v7) every fp32 operand -- weight and activation -- as THREE bf16 terms hi + mid + lo (truncated splits: 24
 *                  significant bits, fp32's exponent range), six bf16 MFMAs (every term down to 2^-16 of the leading one) into one fp32
 *                  accumulator per product-sum.  Nothing data- or scale-dependent: no equalisation, no audit, no feature calibration, no range
 *                  guard.  Records are larger (rnf_mobius_packed_floats_prec / rnf_cond_packed_floats_prec: a weight tile is 3072 floats
 *                  instead of 2048), the feature-projection record holds the RNF_PREC_FP32 image, and the kernels stage synchronously.
 *                  Host-packed flows only (rnf_pack_flow_device does not build it).
 */
#define RNF_PREC_FP32 0
#define RNF_PREC_F16X2 1
#define RNF_PREC_BF16X3 2

int rnf_abi_version(void);
const char *rnf_last_error(void);
/* largest relative error of the conditioner outputs the last rnf_pack_mobius / rnf_pack_cond* call of this thread measured on its probe
 * inputs (split precision only; 0 after an exact-fp32 pack) */
double rnf_last_pack_audit(void);
/* Process-wide measurement / test switches; both return the previous setting.  rnf_set_equalize(0): the packers split the weights as given
 * (also RNF_EQUALIZE=0 in the environment).  rnf_set_pack_audit(0): the host packers still measure, but no longer refuse. */
/* Mean square of a feature entry that the packers called from THIS thread assume when they equalise a conditional layer (default 1; the
 * one data-dependent input of csrc/equalize.h).  Returns the previous value. */
double rnf_set_feature_ms(double mean_square);
int rnf_set_equalize(int on);
int rnf_set_pack_audit(int on);
/* rnf_set_fused(1) (or RNF_FUSED=1): forward passes of conditional flows whose every MLP layer is conditional and feature_dim <= 256 run with
 * the feature projection INSIDE the stack kernel (no projection scratch in HBM).  Off by default: slower than the two-kernel path on C4. */
int rnf_set_fused(int on);
/* Block size of the training backward sweep: 16-rotation workgroups (csrc/train_block16.h) or 64-rotation ones (csrc/train_kernels.h, K <= 64
 * only); 0 (default) picks by batch size.  RNF_TRAIN_BLOCK=16|64 in the environment does the same.  Returns the previous setting. */
int rnf_set_train_block(int rotations);

/* ---- parameter packing (host side, pure CPU; called once per parameter version) ------------------------------
 * Sizes are in floats.  `segments` (K) is any positive count (flow/mobiusflow.py:7-14 takes any): records hold ceil(K / 8) fc_last tiles,
 * the last one zero padded, and the kernels give the pad segments weight 0.  rnf_flow_inverse keeps a layer's segment parameters in
 * registers (64 per lane; beyond K = 128 the rest streams through a per-wave stash, see rnf_workspace_bytes_segments); the training entry
 * points take K <= 512 (the conditioner outputs of a 16-rotation block live in LDS).  `feature_dim` (F) is the number of
 * feature inputs of the layer's MLP (0 for an unconditional Moebius layer).
 */
int64_t rnf_mobius_packed_floats(int32_t segments);                       /* RNF_PREC_FP32 / RNF_PREC_F16X2 records */
int64_t rnf_mobius_packed_floats_prec(int32_t segments, int32_t precision);   /* any RNF_PREC_* */
int64_t rnf_cond_packed_floats_prec(int32_t n_out /* 16, 9 or 36 */, int32_t precision);
int64_t rnf_affine16_packed_floats(void);
int64_t rnf_cond16_packed_floats(void);
int64_t rnf_featproj_packed_floats(int32_t feature_dim); /* 0 when feature_dim == 0 */

/* MobiusFlow.conditioner = ConditionalTransform(3+F, 4K) (flow/mobiusflow.py:40-43, flow/condition.py:10-22).
 * Weights are torch.nn.Linear layout [out, in], row-major.  fc_first_w is [64, 3+F] with the 3 y-inputs FIRST
 * (flow/mobiusflow.py:53-56).  fc_last_w is [4K, 64]: rows [0,K) raw segment weights, row K+3k+d = w_k[d]
 * (flow/mobiusflow.py:58-61).  out_layer receives rnf_mobius_packed_floats(K) floats; out_feat receives
 * rnf_featproj_packed_floats(F) floats (ignored when F == 0). */
int rnf_pack_mobius(const float *fc_first_w, const float *fc_first_b, const float *l1_w, const float *l1_b,
                    const float *l3_w, const float *l3_b, const float *l5_w, const float *l5_b,
                    const float *fc_last_w, const float *fc_last_b, int32_t segments, int32_t feature_dim,
                    int32_t precision, float *out_layer, float *out_feat);

/* Uncondition16Trans.mat [4,4] row-major (flow/squeezetrans.py:164-165).  Packs M, log|det M|, M^-1 and
 * log|det M^-1| (the reference recomputes inv/det every call: squeezetrans.py:38,171-174). */
int rnf_pack_affine16(const float *mat16, float *out_layer);

/* UnconditionRot (flow/rottrans.py:8-23): mat16 = U^T V of the SVD of the 4x4 parameter (an orthogonal matrix, computed by the
 * caller exactly as the reference does); the layer rotates the quaternion and contributes a log-det of exactly 0.  Same record
 * size and layer kind (RNF_LAYER_AFFINE16) as rnf_pack_affine16; the inverse pass uses the transpose (rottrans.py:26-28). */
int rnf_pack_rot16(const float *mat16, float *out_layer);

/* Uncondition9Trans (n = 3) / Uncondition36Trans (n = 6), flow/squeezetrans.py:250-261, 350-361 (and their LU
 * parameterisations, whose assembled matrix is passed here): layer kinds RNF_LAYER_GS9 / RNF_LAYER_GS36.
 * Record: M row-major, then M^-1 (used by the inverse pass, squeezetrans.py:259-261,359-361). */
int64_t rnf_gs_packed_floats(int32_t n);
int rnf_pack_gs(const float *mat, int32_t n, float *out);

/* Condition16Trans.net = ConditionalTransform(F, 16) (flow/squeezetrans.py:42-44). fc_first_w [64,F], fc_last_w [16,64]. */
int rnf_pack_cond16(const float *fc_first_w, const float *fc_first_b, const float *l1_w, const float *l1_b,
                    const float *l3_w, const float *l3_b, const float *l5_w, const float *l5_b,
                    const float *fc_last_w, const float *fc_last_b, int32_t feature_dim, int32_t precision,
                    float *out_layer, float *out_feat);

/* Same record for the conditional 3x3 layers (RNF_LAYER_COND9_*): fc_last has 9 rows. */
int rnf_pack_cond9(const float *fc_first_w, const float *fc_first_b, const float *l1_w, const float *l1_b,
                    const float *l3_w, const float *l3_b, const float *l5_w, const float *l5_b,
                    const float *fc_last_w, const float *fc_last_b, int32_t feature_dim, int32_t precision,
                    float *out_layer, float *out_feat);

/* Condition36Trans (RNF_LAYER_COND36): fc_last has 36 rows in two tiles; record length rnf_cond36_packed_floats(). */
int64_t rnf_cond36_packed_floats(void);
int rnf_pack_cond36(const float *fc_first_w, const float *fc_first_b, const float *l1_w, const float *l1_b,
                    const float *l3_w, const float *l3_b, const float *l5_w, const float *l5_b,
                    const float *fc_last_w, const float *fc_last_b, int32_t feature_dim, int32_t precision,
                    float *out_layer, float *out_feat);

/* ---- the flow -------------------------------------------------------------------------------------------------
 * rotation_dev   [n,3,3] float32 row-major contiguous
 * feature_dev    [n,F] float32 row-major contiguous (F % 8 == 0), or NULL for an unconditional flow
 * rotation_out   [n,3,3] or NULL;  ldj_out [n] or NULL
 * workspace_dev  scratch of at least rnf_workspace_bytes(n, n_cond_layers) bytes (may be NULL when that is 0)
 * Aliasing: rotation_out == rotation_dev (in place) is allowed -- every lane reads its rotation before it writes it -- but such a call
 * runs WITHOUT the range guard (the exact-fp32 re-run would start from the already overwritten input): an fp16 overflow then shows as
 * NaN outputs instead of being repaired.  No other overlap between inputs and outputs is supported.
 */
size_t rnf_workspace_bytes(int64_t n, int32_t n_cond_layers);
/* the same plus the per-wave stash an INVERSE pass with more than 128 segments needs (equal to rnf_workspace_bytes up to 128) */
size_t rnf_workspace_bytes_segments(int64_t n, int32_t n_cond_layers, int32_t segments);

/* Flow.forward (flow/flow.py:53-72): ldj = sum of forward log-det-Jacobians. */
int rnf_flow_forward(const float *rotation_dev, const float *feature_dev, int64_t n, int32_t feature_dim,
                     const float *blob_dev, const int32_t *desc, int32_t n_layers, int32_t segments,
                     float *rotation_out_dev, float *ldj_out_dev, void *workspace_dev, size_t workspace_bytes,
                     void *stream);

/* Flow.inverse (flow/flow.py:74-92): walks the table backwards; ldj = sum of inverse-map log-dets
 * (MobiusFlow.inverse returns -ldj: flow/mobiusflow.py:183). */
int rnf_flow_inverse(const float *rotation_dev, const float *feature_dev, int64_t n, int32_t feature_dim,
                     const float *blob_dev, const int32_t *desc, int32_t n_layers, int32_t segments,
                     float *rotation_out_dev, float *ldj_out_dev, void *workspace_dev, size_t workspace_bytes,
                     void *stream);

/* The same three calls for flows that contain RNF_LAYER_SIDE* layers: side_dev float[n_side_layers][n][16] (row-major matrices; 3x3 ones
 * in the first 9 floats). */
int rnf_flow_forward_side(const float *rotation_dev, const float *feature_dev, int64_t n, int32_t feature_dim, const float *side_dev,
                          const float *blob_dev, const int32_t *desc, int32_t n_layers, int32_t segments,
                          float *rotation_out_dev, float *ldj_out_dev, void *workspace_dev, size_t workspace_bytes, void *stream);
int rnf_flow_inverse_side(const float *rotation_dev, const float *feature_dev, int64_t n, int32_t feature_dim, const float *side_dev,
                          const float *blob_dev, const int32_t *desc, int32_t n_layers, int32_t segments,
                          float *rotation_out_dev, float *ldj_out_dev, void *workspace_dev, size_t workspace_bytes, void *stream);
int rnf_flow_log_prob_side(const float *rotation_dev, const float *feature_dev, int64_t n, int32_t feature_dim, const float *side_dev,
                           const float *blob_dev, const int32_t *desc, int32_t n_layers, int32_t segments, const float *fisher_A_dev,
                           const float *fisher_c_dev, int64_t fisher_B, float *rotation_out_dev, float *ldj_out_dev, float *logp_out_dev,
                           double *sum_out_dev, void *workspace_dev, size_t workspace_bytes, void *stream);

/* ConditionRot (flow/rottrans.py:37-66): the per-sample orthogonal 4x4 matrices U^T V of svd(I + reshape(mlp_out, 4, 4)), with the sign
 * conventions of the reference's torch.svd (LAPACK's dense-SVD path restated for 4x4, csrc/svd4_lapack.h; identical for >= 99.8 % of
 * random matrices, the rest differ like two LAPACK builds do).  mlp_out_dev [n][16] from rnf_cond_mlp_forward, rot_out_dev [n][16] = one
 * slot of the side buffer of an RNF_LAYER_SIDE16_ROT layer.  Stream-ordered, no host synchronisation.
 * fail_flag_dev (int32, may be null): bit 0 is OR-ed in when the QR iteration of a sample did not converge within LAPACK's sweep limit (the
 * slot then holds the factors of the last sweep); the caller zeroes it and reads it back when convenient.
 * rnf_condrot_svd also returns the factors -- u_out_dev [n][16] (U row-major), s_out_dev [n][4] (decreasing), vt_out_dev [n][16] (V^T
 * row-major) -- which is what the backward of U^T V needs (d rot = -wU rot + rot wV with wU, wV from U^T dM V; flow/rottrans.py here), so
 * that evaluation AND training see the same routine's sign choices (ABI v6; v5 trained through the host's torch.svd). */
int rnf_condrot_matrices(const float *mlp_out_dev, int64_t n, float *rot_out_dev, int32_t *fail_flag_dev, void *stream);
int rnf_condrot_svd(const float *mlp_out_dev, int64_t n, float *rot_out_dev, float *u_out_dev, float *s_out_dev, float *vt_out_dev,
                    int32_t *fail_flag_dev, void *stream);

/* ConditionLU's per-sample matrices on the device (replaces flow/squeezetrans.py:120-131, the `einsum` over `torch.diag` -- ABI v7; until v6
 * the host library assembled them with torch ops):
 *     weight[n] = w_p (reshape(wl[n], C, C) * l_mask + l_eye) (reshape(wu[n], C, C) * u_mask + dvec),  dvec[d] = s_sign[d] exp(ws[d][d])
 * where `torch.diag` of the 2-D tensor s_sign * exp(ws) is its diagonal ACROSS THE BATCH (rows 0..C-1) and the C-vector is broadcast over
 * the last axis -- added to every row of the upper factor.  Reproduced as the reference defines it: the matrix of sample n depends on the
 * first C rows of the batch; n < C is an error (the reference fails to broadcast).
 * wl_dev, wu_dev, ws_dev: the three conditioners' outputs (rnf_cond_mlp_forward), rows of stride_wl / stride_wu (>= C*C) and stride_ws
 * (>= C) floats; C = 3 or 4; consts_dev = w_p [C*C] | l_mask [C*C] | u_mask [C*C] | l_eye [C*C] | s_sign [C] (the module's buffers);
 * add_identity != 0: + I (Condition9TransLU, squeezetrans.py:269-271); side_out_dev [n][16]: one slot of the side buffer of an
 * RNF_LAYER_SIDE16 / RNF_LAYER_SIDE9 layer (C x C row-major in the leading floats, the rest 0).
 * rnf_condlu_backward: g_side_dev [n][16] = dL/d(weight) -> g_wl_dev [n][C*C], g_wu_dev [n][C*C], g_ws_dev [n][C] (dense rows); the
 * batch-coupled diagonal's gradient lands in g_ws[d][d] only (every other entry 0).  scratch_dev: 4 floats. */
int rnf_condlu_matrices(const float *wl_dev, const float *wu_dev, const float *ws_dev, int32_t stride_wl, int32_t stride_wu, int32_t stride_ws,
                        int64_t n, int32_t C, const float *consts_dev, int32_t add_identity, float *side_out_dev, void *stream);
int rnf_condlu_backward(const float *wl_dev, const float *wu_dev, const float *ws_dev, int32_t stride_wl, int32_t stride_wu, int32_t stride_ws,
                        int64_t n, int32_t C, const float *consts_dev, const float *g_side_dev, float *g_wl_dev, float *g_wu_dev,
                        float *g_ws_dev, float *scratch_dev, void *stream);

/* ConditionalTransform(feature_dim, <= 16 outputs)(feature) alone (flow/condition.py:24-30): records packed by rnf_pack_cond16 at
 * layer_offset / feat_offset (floats) of blob_dev; out_dev float[n][16], output o in column o.  Workspace: rnf_workspace_bytes(n, 1). */
int rnf_cond_mlp_forward(const float *feature_dev, int64_t n, int32_t feature_dim, const float *blob_dev, int32_t layer_offset,
                         int32_t feat_offset, int32_t precision, float *out_dev, void *workspace_dev, size_t workspace_bytes, void *stream);

/* ---- training (agent.py:75-92: loss = mean(-ldj), loss.backward(), Adam step) ----------------------------------------
 *
 * The backward pass works on the "plain" parameter blob: per layer, in reference state-dict order,
 *   Moebius / Condition16Trans conditioner (flow/condition.py): fc_first.weight [64][NI] | fc_first.bias | hidden[0] W,b |
 *       hidden[2] W,b | hidden[4] W,b | fc_last.weight [NO][64] | fc_last.bias  with NI = 3 + F, NO = 4 * segments
 *       (Moebius) or NI = F, NO = 16 (Condition16Trans);
 *   Uncondition16Trans (flow/squeezetrans.py:57-66): mat [16].
 * rnf_plain_layer_floats gives each layer's length; the gradient blob has the same layout.
 * train_desc: int32 [n_layers][3] = kind (| RNF_TRAIN_ORTHOGONAL for UnconditionRot, whose matrix is orthogonal and whose
 *             ldj is 0, flow/rottrans.py:14-23), perm_row, offset (floats) of the layer in the plain blob, in flow order.
 */
#define RNF_TRAIN_ORTHOGONAL 256
size_t rnf_plain_layer_floats(int32_t kind, int32_t segments, int32_t feature_dim);

/* Build the kernel blob on the device from the plain blob (same bits as rnf_pack_* on the host; the parameters change every
 * training iteration).  pack_desc: int32 [n_layers][4] = kind (| RNF_TRAIN_ORTHOGONAL), offset in the plain blob, offset of
 * the layer record in the kernel blob, offset of its feature-projection record (or -1), all in floats; record offsets are
 * multiples of 4.  feature_dim is the real (unpadded) width; records are laid out for it padded to a multiple of 8.
 * flags_dev: device int32, zeroed by the caller; bit 0 = a weight outside the fp16 range under RNF_PREC_F16X2 (the blob
 * then holds inf/NaN), bit 1 = a singular 4x4 matrix. */
int rnf_pack_flow_device(const float *plain_dev, const int32_t *pack_desc, int32_t n_layers, int32_t segments,
                         int32_t feature_dim, int32_t precision, float *blob_dev, int32_t *flags_dev, void *stream);

/* Flow.forward that also saves the rotation entering every layer: states_dev float[n_layers][n][9]. */
int rnf_flow_forward_train(const float *rotation_dev, const float *feature_dev, int64_t n, int32_t feature_dim,
                           const float *blob_dev, const int32_t *desc, int32_t n_layers, int32_t segments,
                           float *rotation_out_dev, float *ldj_out_dev, float *states_dev, void *workspace_dev,
                           size_t workspace_bytes, void *stream);

/* rnf_flow_forward_train for small batches, from the PLAIN parameter blob (the layout rnf_flow_backward reads: no packing step) on
 * 16-rotation workgroups in exact fp32 (csrc/train_block16.h) -- the arithmetic of the backward sweep's own forward recompute.  Replaces
 * Flow.forward inside the training step (agent.py:75-92) for flows made of Moebius, Uncondition16Trans / UnconditionRot and
 * Condition16Trans layers, n_layers <= 200, segments <= 512; other flows are refused (use rnf_flow_forward_train).  train_desc: the
 * table of rnf_flow_backward; feature_dev [n][feature_dim], unpadded. */
int rnf_flow_forward_train_plain(const float *rotation_dev, const float *feature_dev, int64_t n, int32_t feature_dim,
                                 const float *plain_dev, const int32_t *train_desc, int32_t n_layers, int32_t segments,
                                 float *rotation_out_dev, float *ldj_out_dev, float *states_dev, float *acts_dev, void *stream);
/* acts_dev (may be NULL): rnf_train_acts_floats(n, conditioner layers, segments) floats in which the forward leaves every conditioner's
 * activations (2 KB per rotation and layer at 64 segments); rnf_flow_backward_saved -- rnf_flow_backward with that buffer -- then reads them
 * back instead of recomputing each conditioner (16-rotation sweep; the 64-rotation sweep of large batches ignores the buffer). */
size_t rnf_train_acts_floats(int64_t n, int32_t n_conditioner_layers, int32_t segments);
int rnf_flow_backward_saved(const float *states_dev, const float *acts_dev, const float *feature_dev, int64_t n, int32_t feature_dim,
                            const float *plain_dev, const int32_t *train_desc, int32_t n_layers, int32_t segments,
                            const float *g_rotation_out_dev, const float *g_ldj_dev, float *grads_dev, float *g_rotation_in_dev,
                            float *g_feature_dev, float *layer_scratch_dev, void *stream);

/* Reverse sweep of Flow.forward (what autograd does for the reference, agent.py:79-80).
 * In : g_rotation_out_dev [n][9] (NULL = zeros), g_ldj_dev [n].
 * Out: grads_dev (plain layout, ACCUMULATED into: zero it first; NULL = skip every parameter gradient, for callers that only
 *      differentiate w.r.t. the inputs: pose refinement, eval.py:464-478), g_rotation_in_dev [n][9],
 *      g_feature_dev [n][F] (accumulated into; may be NULL).
 * Scratch: layer_scratch_dev float[n_layers], zeroed by the caller (batch sums of dL/dldj for the d log|det M| / dM term).
 * Any segment count 1..512 (the conditioner outputs of a 16-rotation block live in LDS), n_layers <= 400 like the forward passes (beyond 200
 * the sweep runs in chunks of 200 layers; g_rotation_in_dev then also carries the gradient between the chunks and may alias
 * g_rotation_out_dev). */
int rnf_flow_backward(const float *states_dev, const float *feature_dev, int64_t n, int32_t feature_dim,
                      const float *plain_dev, const int32_t *train_desc, int32_t n_layers, int32_t segments,
                      const float *g_rotation_out_dev, const float *g_ldj_dev, float *grads_dev, float *g_rotation_in_dev,
                      float *g_feature_dev, float *layer_scratch_dev, void *stream);

/* Gradients THROUGH Flow.inverse (flow/flow.py:74-92; MobiusFlow.inverse with BinFind.backward's implicit-function gradient of the
 * root, flow/mobiusflow.py:247-273; the affine layers apply M^-1, flow/squeezetrans.py:51-55,171-174).
 * rnf_flow_inverse_train is rnf_flow_inverse that also saves the rotation entering every ITERATION position of the inverse pass
 * (position 0 = the last flow layer): states_dev float[n_layers][n][9].  rnf_flow_inverse_backward is the reverse sweep:
 * train_desc lists the layers in that iteration order, rotation_out_dev is the output of the inverse pass (each Moebius layer reads its
 * root back from its own output, no second root search); everything else as in rnf_flow_backward.  g_rotation_in_dev is the gradient
 * w.r.t. the rotations GIVEN to Flow.inverse. */
int rnf_flow_inverse_train(const float *rotation_dev, const float *feature_dev, int64_t n, int32_t feature_dim,
                           const float *blob_dev, const int32_t *desc, int32_t n_layers, int32_t segments,
                           float *rotation_out_dev, float *ldj_out_dev, float *states_dev, void *workspace_dev,
                           size_t workspace_bytes, void *stream);
int rnf_flow_inverse_backward(const float *states_dev, const float *rotation_out_dev, const float *feature_dev, int64_t n,
                              int32_t feature_dim, const float *plain_dev, const int32_t *train_desc, int32_t n_layers, int32_t segments,
                              const float *g_rotation_out_dev, const float *g_ldj_dev, float *grads_dev, float *g_rotation_in_dev,
                              float *g_feature_dev, float *layer_scratch_dev, void *stream);

/* Training flows that contain side layers (RNF_LAYER_SIDE16 / SIDE16_ROT / SIDE9: Condition16TransLU, ConditionRot, Condition9TransLU;
 * flow/squeezetrans.py:134-144,264-277, flow/rottrans.py:37-66).  Their per-sample matrices are built by the caller with the reference's
 * own tensor ops (the LU layers' torch.diag couples the batch; ConditionRot's U^T V follows torch.svd's conventions), so the gradient
 * chain is split: rnf_flow_train_side (dir 0 = Flow.forward, 1 = Flow.inverse) is the pass that saves the layer inputs;
 * rnf_flow_backward_side is the reverse sweep, which also writes dL/d(matrix) into side_grad_dev float[n_side][n][16] (same layout as
 * side_dev) for the caller's autograd to carry through those ops; rnf_cond_mlp_backward is the backward of ONE conditioner network
 * evaluated by rnf_cond_mlp_forward.  train_desc: the side layer's slot goes into bits 16..23 of the kind word, its plain offset is
 * unused (rnf_plain_layer_floats = 0); pack_desc / desc as for any layer (an empty 4-float record). */
int rnf_flow_train_side(int32_t dir, const float *rotation_dev, const float *feature_dev, int64_t n, int32_t feature_dim,
                        const float *side_dev, const float *blob_dev, const int32_t *desc, int32_t n_layers, int32_t segments,
                        float *rotation_out_dev, float *ldj_out_dev, float *states_dev, void *workspace_dev, size_t workspace_bytes,
                        void *stream);
int rnf_flow_backward_side(int32_t dir, const float *states_dev, const float *rotation_out_dev, const float *feature_dev, int64_t n,
                           int32_t feature_dim, const float *plain_dev, const int32_t *train_desc, int32_t n_layers, int32_t segments,
                           const float *side_dev, float *side_grad_dev, const float *g_rotation_out_dev, const float *g_ldj_dev,
                           float *grads_dev, float *g_rotation_in_dev, float *g_feature_dev, float *layer_scratch_dev, void *stream);
/* plain_dev: the network's parameters in reference order (fc_first.weight [64][F], .bias, layers.{1,3,5}.weight / .bias, fc_last.weight
 * [n_out][64], .bias; flow/condition.py:14-22); g_out_dev float[n][n_out]; grads_dev (same layout as plain_dev) and g_feature_dev [n][F]
 * are ACCUMULATED into and may be NULL; scratch1_dev: one zeroed float.  n_out <= 64. */
int rnf_cond_mlp_backward(const float *feature_dev, int64_t n, int32_t feature_dim, const float *plain_dev, int32_t n_out,
                          const float *g_out_dev, float *grads_dev, float *g_feature_dev, float *scratch1_dev, void *stream);

/* Shared feature rows: feature_dev holds n / feature_div rows and row r conditions rotations [r * feature_div, (r + 1) * feature_div)
 * -- the pose-estimation pattern of Agent.eval_acc (agent.py:238-263), where the reference materialises feature.repeat(number_queries).
 * The feature projection runs once per row; workspace from rnf_workspace_bytes_shared.  n must be a multiple of feature_div. */
size_t rnf_workspace_bytes_shared(int64_t n, int32_t n_cond_layers, int64_t feature_div);
int rnf_flow_forward_shared(const float *rotation_dev, const float *feature_dev, int64_t n, int32_t feature_dim, int64_t feature_div,
                            const float *blob_dev, const int32_t *desc, int32_t n_layers, int32_t segments,
                            float *rotation_out_dev, float *ldj_out_dev, void *workspace_dev, size_t workspace_bytes, void *stream);
int rnf_flow_inverse_shared(const float *rotation_dev, const float *feature_dev, int64_t n, int32_t feature_dim, int64_t feature_div,
                            const float *blob_dev, const int32_t *desc, int32_t n_layers, int32_t segments,
                            float *rotation_out_dev, float *ldj_out_dev, void *workspace_dev, size_t workspace_bytes, void *stream);

/* rnf_flow_log_prob with shared feature rows (density of one image on a grid of rotations, eval.py:444-462). */
int rnf_flow_log_prob_shared(const float *rotation_dev, const float *feature_dev, int64_t n, int32_t feature_dim, int64_t feature_div,
                             const float *blob_dev, const int32_t *desc, int32_t n_layers, int32_t segments, const float *fisher_A,
                             const float *fisher_c, int64_t fisher_B, float *rotation_out_dev, float *ldj_out_dev, float *logp_out_dev,
                             double *sum_out, void *workspace_dev, size_t workspace_bytes, void *stream);

/* Fused density evaluation: Flow.forward + MatrixFisherN(A)._log_prob(R') + the NLL accumulation
 * (agent.py:54-65,217-229; utils/fisher.py:217-232).
 *   fisher_A_dev [B,3,3], fisher_c_dev [B] with c_b = sum(S_b) + log(norm_b) (host precomputes the proper singular
 *   values, utils/fisher.py:67-76,93-97); sample i uses row i / (n / B) (fisher.py:226).  Pass NULL/0 for a uniform base.
 *   logp_out_dev [n] or NULL: per-sample log p = ldj + base.
 *   sum_out_dev  double[2] or NULL: {sum_i log p_i, n}, accumulated in fp64 in a fixed order (deterministic).
 *   rotation_out_dev / ldj_out_dev optional as above. */
int rnf_flow_log_prob(const float *rotation_dev, const float *feature_dev, int64_t n, int32_t feature_dim,
                      const float *blob_dev, const int32_t *desc, int32_t n_layers, int32_t segments,
                      const float *fisher_A_dev, const float *fisher_c_dev, int64_t fisher_B,
                      float *rotation_out_dev, float *ldj_out_dev, float *logp_out_dev, double *sum_out_dev,
                      void *workspace_dev, size_t workspace_bytes, void *stream);

/* MatrixFisherN._log_prob alone (utils/fisher.py:217-232), same A/c convention; out [n]. */
int rnf_fisher_log_prob(const float *rotation_dev, int64_t n, const float *fisher_A_dev, const float *fisher_c_dev,
                        int64_t fisher_B, float *out_dev, void *stream);

/* Pose-accuracy epilogue of Agent.eval_acc (agent.py:266-283; utils/utils.py:231-235 min_geodesic_distance_rotmats): angle (radians)
 * between estimate i and the closest of its k ground-truth rotations.  est_dev float[n][9], gt_dev float[n][k][9], out_dev float[n]. */
int rnf_min_geodesic(const float *est_dev, const float *gt_dev, int64_t n, int32_t k, float *out_dev, void *stream);

/* Proper SVD of B parameter matrices on the device (utils/fisher.py:53-76): A = U diag(s) V^T with U, V rotations (row-major, singular
 * vectors as columns), s[2] carrying the sign of det A; lam [B,4] = the Bingham parameters the sampler takes (utils/fisher.py:151-158).
 * Any output pointer may be NULL.  Stream-ordered, no host synchronisation. */
int rnf_fisher_proper_svd(const float *A_dev, int64_t B, float *U_dev, float *V_dev, float *s_dev, float *lam_dev, void *stream);

/* Log-constants c[b] of MatrixFisherN(A[b]) with the default normaliser approximation (utils/fisher.py:67-76 proper singular
 * values, :93-97 norm_type = 1): log p(R) = tr(A^T R) - c.  A_dev float[B][9], c_out_dev float[B]; fp64 inside. */
int rnf_fisher_log_const(const float *A_dev, int64_t B, float *c_out_dev, void *stream);

/* The same for either closed-form normaliser approximation of matrix_fisher_norm_N (utils/fisher.py:79-97):
 *   norm_type 1 (default): 1/sqrt(8 pi (s0+s1)(s1+s2)(s0+s2));
 *   norm_type 0: (1 + Q/6 + s0 s1 s2/6)/exp(s0+s1+s2), where the reference's `(S**2).sum()` has no `dim`: Q runs over ALL B matrices of
 *   the call (batch-coupled, reproduced as is).  Types 2 (Monte-Carlo over pytorch3d random rotations) and 3 (scipy ODE; indexes rows
 *   of the [B,3] tensor) are refused.  scratch_dev: rnf_fisher_scratch_bytes(B) bytes of device memory (16 suffice here). */
size_t rnf_fisher_scratch_bytes(int64_t B);
/* norm_type 2 (utils/fisher.py:98-101): Monte-Carlo normaliser over approx_num uniform rotations for ONE matrix (B must be 1: the
 * reference broadcasts [approx_num,3,3] against [N,3,3]); counter-based Philox stream keyed by `seed` (statistical parity with the
 * reference's pytorch3d.random_rotations).  scratch_dev: 8 bytes. */
int rnf_fisher_log_const_mc(const float *A_dev, int64_t B, int64_t approx_num, uint64_t seed, void *scratch_dev, size_t scratch_bytes,
                            float *c_out_dev, void *stream);
int rnf_fisher_log_const_nt(const float *A_dev, int64_t B, int32_t norm_type, void *scratch_dev, size_t scratch_bytes,
                            float *c_out_dev, void *stream);

/* Gradient of MatrixFisherN._log_prob w.r.t. A (agent.py:57-65 keeps a network-predicted A in the autograd graph; the reference
 * differentiates torch.svd, utils/fisher.py:67-76,217-232):  g_A[b] = sum_i g_logp[i] R_i - (sum_i g_logp[i]) dc/dA_b over the n/B
 * samples of row b, dc/dA = U' diag(dc/ds) V'^T on the proper SVD (csrc/fisher_math.h), plus the batch coupling of norm_type 0.
 * rotation_dev float[n][9], g_A_dev float[B][9] (overwritten), scratch_dev: rnf_fisher_scratch_bytes(B) bytes.  fp64 accumulation. */
int rnf_fisher_log_prob_backward_param(const float *g_logp_dev, const float *rotation_dev, int64_t n, const float *A_dev, int64_t B,
                                   int32_t norm_type, void *scratch_dev, size_t scratch_bytes, float *g_A_dev, void *stream);

/* Gradient of rnf_fisher_log_prob w.r.t. the rotations (training with a matrix-Fisher base, agent.py:58-64):
 * g_rotation[i] = g_logp[i] * A[i / (n/B)]. */
int rnf_fisher_log_prob_backward(const float *g_logp_dev, int64_t n, const float *A_dev, int64_t B, float *g_rotation_dev,
                                 void *stream);

/* pytorch3d.transforms.matrix_to_quaternion as the reference calls it on sampled rotations (utils/fisher.py:242-243, context = 4) and
 * inside calculate_16 (flow/squeezetrans.py:34): rotation_dev float[n][9] -> quaternion_dev float[n][4], real part first, the candidate
 * with the largest component (published 0.7.5 rule). */
int rnf_matrix_to_quaternion(const float *rotation_dev, int64_t n, float *quaternion_dev, void *stream);

/* MatrixFisherN._sample (utils/fisher.py:117-207,234-243): n rotations per row of A, out [B,n,3,3].
 *   U_dev, V_dev [B,3,3]: proper SVD factors of A (det +1; utils/fisher.py:53-64); lam_dev [B,4]: the diagonal Bingham parameter
 *   (0, 2(S1+S2), 2(S0+S2), 2(S0+S1)) (utils/fisher.py:183-187).  Counter-based Philox stream keyed by `seed`: the same seed gives
 *   the same samples; parity with the reference's torch-generator samples is statistical.  fail_flag_dev (must point to one zeroed int32) is set to 1 if any sample exhausted its 4096 proposals. */
int rnf_fisher_sample(const float *U_dev, const float *V_dev, const float *lam_dev, int64_t B, int64_t n, uint64_t seed,
                      float *out_dev, int32_t *fail_flag_dev, void *stream);

/* ConditionalTransform.forward for one packed Moebius layer (flow/condition.py:24-30), unconditional input only:
 * y_dev [n,3] -> out_dev [n,4K] in the reference's output order.  Unit-test / bring-up entry point. */
int rnf_conditioner_forward(const float *y_dev, int64_t n, const float *layer_packed_dev, int32_t segments,
                            int32_t precision, float *out_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RNF_HIP_H */
