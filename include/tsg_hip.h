/* tsg_hip.h -- C ABI of libtsg_hip.so: MI355X (gfx950) kernels for the cross-modal matching hot
 * path of haojc/ShufflingVideosForTSG (grounding/model).
 *
 * The reference has no FFI of its own: its seam is the Python nn.Module protocol.
 * Each entry point below replaces the arithmetic of one reference forward (cited per function)
 * and is what a ctypes / cffi binding on the reference side would bind.
 *
 * Conventions
 *   - all pointers are DEVICE pointers on the current HIP device, 16-byte aligned, row-major
 *     contiguous, batch-first; the caller owns every buffer (outputs and workspaces included);
 *     the library never allocates, frees or synchronises.
 *   - `dtype`: TSG_F32 (0) everywhere; TSG_F32S (2) additionally in the K1 forwards, the LSTM entry points and tsg_mha_* (fp32
 *     storage, split-precision bf16 MFMA products; LSTM hidden sizes other than 128/256/384/512 and attention shapes outside
 *     the ones named at tsg_mha_bwd compute in plain fp32); TSG_BF16 (1) = bf16 STORAGE (ABI revision 3): the activation tensors
 *     of the entry point -- named at each -- are 2-byte bf16 elements in HBM (rows 8-byte aligned: widths % 4 == 0), converted on
 *     load and rounded to nearest-even on store; arithmetic, softmax, cell state and every accumulation stay fp32, as do the
 *     parameters, the small [B,T] / [B,J]-sized side tensors and all parameter gradients.  Taken by tsg_scdm_attn_*,
 *     tsg_scdm_gate_*, tsg_boundary_score_*, tsg_match_head_*, tsg_mha_* (head widths as stated there) and the persistent
 *     tsg_lstm_* paths; shapes an entry point does not take in this dtype return TSG_E_SHAPE (the host code then runs the fp32
 *     storage kernels on fp32 copies).
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); work is only enqueued.
 *   - return 0 on success; <0 = argument error (TSG_E_*); >0 = hipError_t from the launch.
 *     tsg_last_error() returns a thread-local message for the last non-zero return.
 *   - thread-safe to call from one host thread per device.  Process-wide state, all of it atomic words: a per-(device, kernel)
 *     co-residency cache, and the three LSTM switches set by tsg_lstm_error_sink / tsg_lstm_set_l2_exchange /
 *     tsg_lstm_set_persist (initialised from TSG_LSTM_L2X / TSG_LSTM_PERSIST in the environment on first use).
 *   - results are bitwise reproducible run to run except the sums formed with float atomics: dw of K1, dgbias of K1g, dcs / dw2 /
 *     db2 of K5, the K4 loss accumulators, the LSTM's dbias (one atomic add per batch slice and column), dcs / db1 / dw2 of K3, and
 *     dK / dV of tsg_mha_bwd with TSG_F32S when Tk <= 32 and Tq > 128 (one atomic add per query block).
 */
#ifndef TSG_HIP_H
#define TSG_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSG_VERSION 7   /* 2: K1 backward takes (ws, ws_bytes); input pipeline / span decode entry points
                           3: tsg_error_word (device-side expiry report); dtype TSG_BF16 (bf16 storage) in K1 / K1g / K2 / K3 / LSTM
                           4: tsg_boundary_score_bwd_ws (K3 backward in one launch); tsg_gemm_f32s
                           5: tsg_scdm_bwd_mode / tsg_scdm_bwd_fused_ok (path selection as a call and a predicate instead of an
                              environment variable and an error code); per-device error words
                           6: tsg_lstm_fwd_ws (the persistent LSTM forward's exchange ring in a caller-owned workspace), tsg_gemm_bf16, tsg_adam_step,
                              tsg_gemm_f32s_nn_acc, tsg_lstm_set_ring / _wide, tsg_wgrad_set_stream_k
                           7: tsg_time_next_launch / tsg_timed_launch_us (one launch bracketed by its own event pair), tsg_grads_nonfinite */
#define TSG_F32 0
#define TSG_BF16 1   /* bf16 storage of the activations, fp32 arithmetic (see Conventions)                               */
#define TSG_F32S 2   /* fp32 storage; matrix products as split-precision bf16 MFMAs (x = hi + lo; hi*hi + hi*lo + lo*hi,
                        fp32 accumulation) -- accepted by the LSTM entry points and by tsg_mha_fwd / tsg_mha_bwd[_rng] */

#define TSG_LSTM_SYNC_BYTES 2048   /* size of tsg_lstm_fwd's sync_ws */
#define TSG_E_NULL   (-1)   /* a required pointer is NULL                      */
#define TSG_E_SHAPE  (-2)   /* non-positive or unsupported dimension           */
#define TSG_E_ALIGN  (-3)   /* pointer / leading dimension not 16-B aligned    */
#define TSG_E_DTYPE  (-4)   /* dtype not supported by this entry point         */
#define TSG_E_LDS    (-5)   /* tile does not fit the 160 KiB LDS budget        */

int         tsg_version(void);
const char* tsg_last_error(void);

/* Measurement hook (bench.py's roofline line; never used by the product path).  tsg_time_next_launch(slot) arms the calling
 * thread: the NEXT launch of tsg_scdm_attn_fwd / tsg_scdm_gate_fwd (dtype TSG_F32S or TSG_BF16) or tsg_scdm_attn_bwd / tsg_scdm_gate_bwd
 * (their fused kernel) made by this thread is issued with hipExtLaunchKernel and the library-owned event pair of `slot`
 * (0 <= slot < 1024), which brackets that kernel alone -- the duration rocprofv3's kernel trace reports, without the dispatch gap an
 * event pair recorded around the call adds.  tsg_timed_launch_us(slot, &us) waits for the launch and returns its duration in
 * microseconds; TSG_E_SHAPE for a slot out of range or never used.  tsg_time_next_launch(-1) disarms.  Not valid while a stream capture is open.                        */
int tsg_time_next_launch(int slot);
int tsg_timed_launch_us(int slot, float* us);

/* ---- K1: SCDM additive cross-attention (SCDM_Attention.forward, networks/attention.py:109-121)
 * inputs are the PROJECTED tensors: a = W_a(video)+b [B,T,H], s = W_s(sent) [B,N,H], w [H] (the
 * [1,H] weight of self.w), sent [B,N,Ds].  Computes, without ever materialising [B,T,N,H]:
 *   e[b,t,n] = sum_k w[k]*tanh(a[b,t,k]+s[b,n,k]);  P = softmax_n(e);  C = P @ sent.
 * Outputs C [B,T,Ds] and P [B,T,N] (kept for the backward).  Limits: N <= 32, H%4==0, Ds%4==0,
 * roundup(N,4)*roundup(H,256)*4 B must fit LDS (TSG_E_LDS otherwise).
 * dtype TSG_F32S: C = P @ sent as split-precision bf16 MFMA products where H == Ds in {256, 512, 1024} (else as TSG_F32).
 * dtype TSG_BF16: a, s, sent, C (and dC, da, ds, dsent in the backward) are bf16; w, P, dw stay fp32.                  */
int tsg_scdm_attn_fwd(const void* a, const void* s, const void* w, const void* sent,
                      void* C, void* P, int B, int T, int N, int H, int Ds, int dtype, void* stream);

/* backward of the above.  dC [B,T,Ds] -> da [B,T,H], ds [B,N,H], dw [H], dsent [B,N,Ds] (direct
 * path through C = P@sent only; the path through s = W_s(sent) is the caller's GEMM).  One fused kernel: every input is
 * read once and every output written once (workgroup = batch item x column part; the parts of an item exchange their
 * partial <dC, sent> dot products -- T*N floats each -- through `ws`).  dw is fully overwritten.
 * `ws`: caller-owned workspace of at least tsg_scdm_bwd_ws_bytes(B,T,N,H,Ds,gate) bytes, contents irrelevant.
 * (TSG_K1_BWD=split in the environment selects the two-kernel path of revision 1, which moves dG / de through `ws`.)   */
long long tsg_scdm_bwd_ws_bytes(int B, int T, int N, int H, int Ds, int gate);
/* Path selection of the K1 / K1g backward (ABI revision 5).  tsg_scdm_bwd_mode(mode): 0 = automatic (the one-launch kernel where its
 * plan fits, else the two kernels), 1 = always the two-kernel path (no cross-workgroup exchange), 2 = one launch with the row phase
 * on the VALU; any other value only queries.  Returns the previous mode (initialised from TSG_K1_BWD = split / valu).
 * tsg_scdm_bwd_fused_ok: 1 when the current mode runs this shape in the one-launch kernel on the current device -- the only
 * backward dtype TSG_BF16 takes (the host code decides with this predicate whether to go through fp32 copies; it does not
 * probe with a failing call).                                                                                                  */
int tsg_scdm_bwd_mode(int mode);
int tsg_scdm_bwd_fused_ok(int B, int T, int N, int H, int Ds);
int tsg_scdm_attn_bwd(const void* a, const void* s, const void* w, const void* sent, const void* P,
                      const void* dC, void* da, void* ds, void* dw, void* dsent, void* ws, long long ws_bytes,
                      int B, int T, int N, int H, int Ds, int dtype, void* stream);

/* ---- K1g: SCDM attention fused with the channel gate of rnn_recalibration_layer.forward
 * (components/VideoEncoder.py:61-74):  out = r * sigmoid(sent_linear(C)),  C = P @ sent.
 * Because sent_linear(P sent) = P (sent W_l^T) + b_l, the caller passes VW = sent @ W_l^T [B,N,Ds] (a
 * [B*N,d]x[d,d] GEMM instead of the reference's [B*T,d]x[d,d] one) and the kernel's epilogue applies
 * bias, sigmoid and the gate: C is never materialised.  r [B,T,Ds] is the BiLSTM output being gated
 * (video_dim == Ds here), gbias [Ds].  Same limits and dtypes as tsg_scdm_attn_fwd (TSG_BF16: a, s, VW, r, out and
 * dout, da, ds, dVW, dr are bf16; w, gbias, P, dw, dgbias fp32).                                  */
int tsg_scdm_gate_fwd(const void* a, const void* s, const void* w, const void* VW, const void* gbias,
                      const void* r, void* out, void* P, int B, int T, int N, int H, int Ds, int dtype,
                      void* stream);

/* backward: dout [B,T,Ds] -> da, ds, dw (as tsg_scdm_attn_bwd), dVW [B,N,Ds], dgbias [Ds], dr [B,T,Ds]; `ws` as above
 * (gate = 1 in tsg_scdm_bwd_ws_bytes).  dw and dgbias are fully overwritten.                                       */
int tsg_scdm_gate_bwd(const void* a, const void* s, const void* w, const void* VW, const void* gbias,
                      const void* r, const void* P, const void* dout, void* da, void* ds, void* dw,
                      void* dVW, void* dgbias, void* dr, void* ws, long long ws_bytes,
                      int B, int T, int N, int H, int Ds, int dtype, void* stream);

/* ---- K3: boundary-score head (VideoSentenceConcat + MLP_predictor.forward,
 * components/CrossModalInteraction.py:44-47 + components/SpanPredictor.py:71-85; GMD gate
 * SpanGroundMatchDisc.py:86).  Start and end branches are stacked on the hidden axis: J = 2*Hm.
 * inputs: y [B,T,J] = video @ [W1s_v | W1e_v]^T (the video half of both first Linears, no bias),
 *         cs [B,J] = sent @ [W1s_s | W1e_s]^T (sentence half, no bias), b1 [J], w2 [J], b2 [2],
 *         gate [B,T] or NULL (GMD: raw matching logits multiply the concatenated feature),
 *         mask int32 [B,T] or NULL (mask_logits with -1e30, networks/attention.py:129-133).
 *   z = gate*(y + cs) + b1;  l = w2 . tanh(z) + b2 (per branch);  p = softmax over T.
 * outputs p_start, p_end [B,T].   Limits: J % 4 == 0, J <= 1024, T <= 8192.
 * dtype TSG_BF16: y (and dy in the backward) are bf16; cs, b1, w2, b2, gate, the probabilities and every other gradient fp32. */
int tsg_boundary_score_fwd(const void* y, const void* cs, const void* b1, const void* w2, const void* b2,
                           const void* gate, const int32_t* mask, void* p_start, void* p_end,
                           int B, int T, int Hm, int dtype, void* stream);

/* backward.  dp_start/dp_end [B,T] -> dy [B,T,J], dcs [B,J], per-sample partial sums db1_part [B,J],
 * dw2_part [B,J], db2_part [B,2] (the caller adds them over B), dgate [B,T] (may be NULL).
 * dl_ws: caller-owned workspace of 2*B*T floats.                                                 */
int tsg_boundary_score_bwd(const void* y, const void* cs, const void* b1, const void* w2, const void* gate,
                           const int32_t* mask, const void* p_start, const void* p_end,
                           const void* dp_start, const void* dp_end, void* dy, void* dcs, void* db1_part,
                           void* dw2_part, void* db2_part, void* dgate, void* dl_ws, int B, int T, int Hm,
                           int dtype, void* stream);
/* Same in ONE launch (ABI revision 4).  ws: caller-owned
 * workspace of tsg_boundary_score_bwd_ws_bytes(B,T,Hm) bytes, 16-byte aligned, whose first B 32-bit words (one ticket counter per
 * batch item) must be ZERO when the call is enqueued; the kernel leaves them zero, so one workspace zeroed once serves every later
 * call on the same stream (its other contents are scratch).  The T-sums are added in a fixed order: results are run-to-run
 * identical (the two-kernel entry point above accumulates dcs / db1_part / dw2_part with float atomics).                      */
long long tsg_boundary_score_bwd_ws_bytes(int B, int T, int Hm);
int tsg_boundary_score_bwd_ws(const void* y, const void* cs, const void* b1, const void* w2, const void* gate,
                              const int32_t* mask, const void* p_start, const void* p_end,
                              const void* dp_start, const void* dp_end, void* dy, void* dcs, void* db1_part,
                              void* dw2_part, void* db2_part, void* dgate, void* ws, long long ws_bytes,
                              int B, int T, int Hm, int dtype, void* stream);

/* ---- K2: multi-head dot-product attention between the wq/wk/wv projections and wo
 * (Attention.forward networks/attention.py:45-55, MultiHead.forward / A_forward :71-97).
 * Q [B,Tq,d_key], K [B,Tk,d_key], V [B,Tk,d_value] are split into n_heads chunks of the last axis;
 *   A_h = (Qh Kh^T - 1e10*triu(1) if causal) / scale;  S_h = softmax(A_h);  O = cat_h(S_h Vh).
 * `scale` is the DIVISOR: the reference passes sqrt(d_key) of the full width (attention.py:41,61).
 * Outputs O [B,Tq,d_value], lse [B,n_heads,Tq] (kept for the backward) and, when non-NULL, the
 * A_forward side outputs A_sum = sum_h A_h and S_sum = sum_h S_h, both [B,Tq,Tk].
 * p_drop in [0,1): attention dropout on the softmax (out = dropout(S) V, attention.py:53-54; S_sum stays un-dropped);
 * the keep mask is a counter-based hash of (seed, offset, b, head, q, key), regenerated by the backward from the
 * same (p_drop, seed, offset) -- nothing is stored.
 * Limits: head widths multiples of 4, d_value/n_heads <= 512; causal needs Tq == Tk.          */
int tsg_mha_fwd(const void* Q, const void* K, const void* V, void* O, void* A_sum, void* S_sum, void* lse,
                int B, int Tq, int Tk, int d_key, int d_value, int n_heads, float scale, int causal,
                float p_drop, uint64_t seed, uint64_t offset, int dtype, void* stream);

/* backward: dO [B,Tq,d_value] -> dQ, dK, dV (fully overwritten; no atomics, deterministic).
 * delta_ws: caller-owned workspace of B*n_heads*Tq floats (NULL selects the slower non-MFMA kernel).
 * dtype TSG_F32: exact fp32 products.  TSG_F32S (fp32 storage): the five products of the backward as split-precision bf16 MFMA
 * products (x = hi + lo, hi*hi + hi*lo + lo*hi, fp32 accumulate -- fp32-GEMM-level error) in two kernels (dK/dV, dQ), where
 * d_key == d_value and the head width is a multiple of 32 up to 256 (above 128: channel halves per wave pair); otherwise the exact
 * kernels run.  tsg_mha_fwd with
 * TSG_F32S: one split-precision kernel with an online softmax for head widths 32 .. 256 in steps of 32 (any Tk) when no A_sum /
 * S_sum side outputs are requested; otherwise the exact kernels.
 * dtype TSG_BF16: Q, K, V, O (and dO, dQ, dK, dV) are bf16, lse / delta_ws fp32 -- the split-precision kernels with 2-byte
 * elements (half the HBM bytes; the lo operand planes are zero).  Needs d_key == d_value, no A_sum / S_sum, head widths 32 .. 256
 * in steps of 32 in tsg_mha_fwd and 32 .. 128 in tsg_mha_bwd (TSG_E_SHAPE otherwise).                                           */
int tsg_mha_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const void* lse,
                void* dQ, void* dK, void* dV, void* delta_ws, int B, int Tq, int Tk, int d_key, int d_value,
                int n_heads, float scale, int causal, float p_drop, uint64_t seed, uint64_t offset, int dtype,
                void* stream);

/* Same two entry points with (seed, offset) read by the kernels from DEVICE memory: rng_dev -> two uint64 words, [0] = seed,
 * [1] = offset.  For launches captured into a HIP graph: the replays advance the offset with a captured add, so every replay
 * draws a fresh mask and the backward of the same replay regenerates it.                                                    */
int tsg_mha_fwd_rng(const void* Q, const void* K, const void* V, void* O, void* A_sum, void* S_sum, void* lse,
                    int B, int Tq, int Tk, int d_key, int d_value, int n_heads, float scale, int causal,
                    float p_drop, const void* rng_dev, int dtype, void* stream);
int tsg_mha_bwd_rng(const void* Q, const void* K, const void* V, const void* O, const void* dO, const void* lse,
                    void* dQ, void* dK, void* dV, void* delta_ws, int B, int Tq, int Tk, int d_key, int d_value,
                    int n_heads, float scale, int causal, float p_drop, const void* rng_dev, int dtype, void* stream);

/* ---- Adjacent glue: bidirectional LSTM recurrence (BiLSTM.forward networks/RNN.py:34-48 = one layer of
 * nn.LSTM(bidirectional), zero initial state).  Sequence tensors are TIME-MAJOR here.  The caller computes
 * the input projections of all steps and both directions with one GEMM:
 *   Gx [T,B,2,4h] = X W_ih^T + b_ih + b_hh  (gate order i,f,g,o).
 * tsg_lstm_fwd runs the T sequential steps (one launch per step covering both directions):
 *   Whh [2,4h,h];  out [T,B,2h] (forward half | reverse half);  saved for backward: R [T,2,B,h,4]
 *   (activated gates) and Cs [T,2,B,h] (cell states).  Limits: h % 4 == 0.
 * sync_ws: caller-owned workspace of TSG_LSTM_SYNC_BYTES bytes (may be NULL).  When given, T >= 8 (TSG_LSTM_PERSIST=0/1 in the
 *   environment or tsg_lstm_set_persist: never / always), h % 32 == 0 and h <= 512, ONE persistent launch runs all T steps (W_hh stationary in
 *   registers; workgroups hand h_t over by polling the sentinel-marked `out` slab itself); when the B rows need more workgroups
 *   than can be co-resident (one per CU: B > 128 at h = 512 on 256 CUs) the rows -- independent sequences -- are processed as
 *   consecutive persistent launches over balanced chunks of whole 16-row slices (B = 256 -> 2 x 128), never as step launches;
 *   word 0 of sync_ws is non-zero afterwards if a bounded wait expired (results then invalid).  Otherwise one launch per time step.
 * dtype TSG_BF16 (persistent path only: sync_ws given, T > 1, h in {128, 256, 384, 512}; TSG_E_SHAPE otherwise): Gx, out and R are
 *   bf16 (R: four bf16 per unit), Cs / bias / Whh stay fp32; W_hh is rounded to bf16 once in the kernel's prologue, each step is
 *   ONE bf16 MFMA per k block, and h_t is exchanged -- and fed back -- as the bf16 value stored to `out` (16-bit sentinels).
 *   tsg_lstm_bwd_ws[_layout] with TSG_BF16: R, dOut and dG bf16; Cs, dHn, dbias and the ring workspace fp32.               */
int tsg_lstm_fwd(const void* Gx, const void* Whh, void* out, void* R, void* Cs, void* sync_ws,
                 int B, int T, int h, int dtype, void* stream);
/* Same, with the bias b_ih + b_hh [2,4h] added inside the kernel (bias may be NULL): for callers whose input GEMM has
 * no bias epilogue, Gx = X W_ih^T.  batch_major != 0: Gx is [B,T,2,4h] and out [B,T,2h] -- the layout of the model's
 * activations (BiLSTM batch_first=True, RNN.py:27), no transposed copies around the recurrence; R and Cs stay time-major. */
int tsg_lstm_fwd_bias(const void* Gx, const void* bias, const void* Whh, void* out, void* R, void* Cs, void* sync_ws,
                      int B, int T, int h, int dtype, int batch_major, void* stream);
/* Same, with a workspace that also holds the EXCHANGE RING of the persistent kernel (ABI revision 6): ws = tsg_lstm_fwd_ws_bytes(B,T,h)
 * bytes (TSG_LSTM_SYNC_BYTES of sync words + [4 slots][2 ceil(B/16) groups][16 rows][h] dwords; 0 = no ring for this hidden size:
 * h % 128 != 0 or h > 512), 16-byte aligned, caller-owned, contents irrelevant on entry.  With it the workgroups hand h_t over through
 * the ring (which stays inside the L2s it is exchanged through; in the TSG_F32S arithmetic it carries h already split into bf16 hi / lo
 * halves) instead of polling `out`; `out` is written with ordinary stores.  Results are bit-identical to tsg_lstm_fwd_bias.  A smaller
 * workspace (>= TSG_LSTM_SYNC_BYTES), or a shape at which the ring is not the faster path (tsg_lstm_set_ring), selects the
 * tsg_lstm_fwd_bias behaviour.                                                                                                         */
long long tsg_lstm_fwd_ws_bytes(int B, int T, int h);
/* When tsg_lstm_fwd_ws takes the ring: -1 = automatic (default: where it measured faster -- B >= 96 rows, or TSG_BF16 at B <= 32; TSG_LSTM_XR=0/1
 * in the environment overrides the default), 0 = never, 1 = whenever the workspace holds one.  Process-wide, like tsg_lstm_set_persist. */
int tsg_lstm_set_ring(int mode);
/* With TSG_BF16 at h = 512 and a ring workspace, tsg_lstm_fwd_ws can run 64-UNIT workgroups (two A-tiles of W_hh per wave, 8 workgroups per
 * exchange group) on half the CUs: half the slab readers per L2 for twice the MFMA work per wave.  -1 = automatic (default: B >= 96 rows;
 * TSG_LSTM_W64=0/1 in the environment sets the initial mode), 0 = never, 1 = whenever the shape allows.  Same results bit for bit.           */
int tsg_lstm_set_wide(int mode);
int tsg_lstm_fwd_ws(const void* Gx, const void* bias, const void* Whh, void* out, void* R, void* Cs, void* ws, long long ws_bytes,
                    int B, int T, int h, int dtype, int batch_major, void* stream);

/* Error sink of the persistent kernels: an int in memory the host can read without synchronising (pinned / host-mapped, or
 * device memory), set to 1 by any persistent launch whose bounded wait expired (its results are then invalid; word 0 of
 * that launch's sync workspace is set as well).  NULL (default) disables it.  The Python host registers a pinned word and
 * checks it on every LSTM call, so a failed launch raises at the next call instead of passing silently.               */
int tsg_lstm_error_sink(void* flag);
int tsg_error_sink(void* flag);          /* the same sink under its general name: the K1 backward's bounded exchange wait reports there too */
/* The same report into DEVICE memory: a 4-byte word (caller-owned, cleared by the caller) every kernel with a bounded wait sets to
 * 1 on expiry, in addition to the host sink.  It exists for guards that run on the device -- the host code ORs it into the fused
 * optimizer's found_inf input, so the update of a step whose backward was corrupted by an expired wait is skipped even when the
 * step is replayed from a HIP graph and no host code runs between its launches.  On expiry the K1 backward also poisons the
 * affected item's da / ds / dw with NaN instead of summing incomplete partials.  NULL (default) disables it.  ONE WORD PER DEVICE
 * (ABI revision 5): the call registers the word for the current HIP device, and a launch reports into the word of the device it
 * runs on.                                                                                                                   */
int tsg_error_word(void* device_flag);
/* Allow (default, TSG_LSTM_L2X) or forbid the exchange that stays inside one XCD's L2 (plain stores when a group's one-XCD
 * placement is verified); forbidden = write-through stores always.  The Python host turns it off when its start-up self-test
 * of the persistent kernels reports an expired wait.                                                                       */
int tsg_lstm_set_l2_exchange(int on);
/* Persistent kernels: 0 = never (one launch per time step), 1 = whenever the shape allows, -1 = automatic (T >= 8; the
 * default, or TSG_LSTM_PERSIST).  The Python host selects 0 when its start-up self-test fails in both exchange modes.  */
int tsg_lstm_set_persist(int mode);

/* backward of the recurrence: dOut [T,B,2h] (+ optional dHn [2,B,h] added at each direction's last step)
 * -> dG [T,B,2,4h] = dL/d(pre-activation gates); the caller derives dX, dW_ih, dW_hh, db from it with
 * GEMMs.  WhhT [2,h,4h] is W_hh transposed per direction; dC_ws is a [2,B,h] float workspace (the cell-gradient carry
 * between the per-step launches).                                                                            */
int tsg_lstm_bwd(const void* WhhT, const void* R, const void* Cs, const void* dOut, const void* dHn,
                 void* dG, void* dC_ws, int B, int T, int h, int dtype, void* stream);

/* Same result through the persistent backward kernel when the caller provides its ring workspace: ws of at least
 * tsg_lstm_bwd_ws_bytes(B,T,h) bytes (0 = not available for this shape: h % 128 != 0 or h > 512), 16-byte aligned,
 * contents irrelevant.  One launch runs all T steps: workgroups exchange partial dh tiles through a 4-slot ring of
 * sentinel-marked, write-through 128-byte lines (no atomics or fences per step).  Falls back to tsg_lstm_bwd when ws is
 * NULL / too small or T < 8 (TSG_LSTM_PERSIST=0/1 or tsg_lstm_set_persist: never / always); rows beyond the co-resident
 * grid are processed in chunks as in the forward.                                                                   */
long long tsg_lstm_bwd_ws_bytes(int B, int T, int h);
/* 1 when tsg_lstm_bwd_ws will take the persistent path for this shape and workspace size (then, and only then, it also
 * fills dbias [2,4h] = d(b_ih + b_hh), the sum of dG over time and batch, when dbias is non-NULL).                  */
int tsg_lstm_bwd_ws_persistent(int B, int T, int h, long long ws_bytes);
int tsg_lstm_bwd_ws(const void* WhhT, const void* R, const void* Cs, const void* dOut, const void* dHn,
                    void* dG, void* dC_ws, void* ws, long long ws_bytes, void* dbias, int B, int T, int h, int dtype,
                    void* stream);
/* Same with the layout of dOut / dG selectable: batch_major != 0 -> dOut [B,T,2h], dG [B,T,2,4h].                    */
int tsg_lstm_bwd_ws_layout(const void* WhhT, const void* R, const void* Cs, const void* dOut, const void* dHn,
                           void* dG, void* dC_ws, void* ws, long long ws_bytes, void* dbias, int B, int T, int h, int dtype,
                           int batch_major, void* stream);

/* ---- K4: the four GMD training losses in one launch each way (grounding/loss.py:6-51, combined as train.py:142-165) -----
 * ps, pe [B,T] span start / end probabilities; om, pm [B,T] matching logits of the original / shuffled video; od, pd [B,2]
 * order-discriminator logits; fs, pfs int64 [B,2] ground-truth (start, end) clip indices of the original / shuffled video;
 * tl, ptl [B,T] temporal labels (0/1 as float); vm [B,T] video mask (float).  out[4] = (span_ground_loss,
 * BCE_loss(om,tl,vm) + BCE_loss(pm,ptl,vm), matching_KL_divergence(masked_softmax(om,tl), masked_softmax(pm,ptl), fs, pfs),
 * temporal_order_discrimination_loss(od, pd)), all un-weighted, and out[4] = out[0] + lam_match out[1] + lam_kl out[2] +
 * lam_disc out[3], the training loss of train.py:160 (out holds 5 floats).  ws: 32-byte workspace (zeroed by the call; kept
 * for the backward).  T <= 2048.  The backward writes the gradients of ps, pe, om, pm, od, pd given dL[4] (gradient of the
 * four losses) and / or dtotal[1] (gradient of out[4]); either may be NULL.                                              */
int tsg_gmd_losses_fwd(const void* ps, const void* pe, const void* om, const void* pm, const void* od, const void* pd,
                       const void* fs, const void* pfs, const void* tl, const void* ptl, const void* vm,
                       void* ws, void* out, int B, int T, float lam_match, float lam_kl, float lam_disc, void* stream);
int tsg_gmd_losses_bwd(const void* ps, const void* pe, const void* om, const void* pm, const void* od, const void* pd,
                       const void* fs, const void* pfs, const void* tl, const void* ptl, const void* vm,
                       const void* ws, const void* dL, const void* dtotal, void* dps, void* dpe, void* dom, void* dpm, void* dod,
                       void* dpd, int B, int T, float lam_match, float lam_kl, float lam_disc, void* stream);

/* ---- K5: matching head (VideoTextSemanticMatch = VideoTextConcat + TwoLayerdMLP, components/DistributionAlign.py:51-98)
 * after the video-half GEMM, as K3 does for the boundary head: y [B,T,H] = video @ W1v^T (no bias), cs [B,H] = query @ W1s^T
 * + b1, w2 [H], b2 [1]:   logits[b,t] = w2 . act(y[b,t,:] + cs[b,:]) + b2,   activation 0 = relu, 1 = tanh, 2 = sigmoid.
 * The backward writes dy [B,T,H] and accumulates dcs [B,H], dw2 [H], db2 [1] (zeroed by the call).  H % 4 == 0, H <= 1024.
 * dtype (ABI revision 3): TSG_F32, or TSG_BF16 = y and dy stored as bf16 (cs, w2, b2, logits and the sums stay fp32).    */
int tsg_match_head_fwd(const void* y, const void* cs, const void* w2, const void* b2, void* logits,
                       int B, int T, int H, int activation, int dtype, void* stream);
int tsg_match_head_bwd(const void* y, const void* cs, const void* w2, const void* dlogits, void* dy, void* dcs,
                       void* dw2, void* db2, int B, int T, int H, int activation, int dtype, void* stream);

/* ---- MomentPooling's masked means (temporal-order discriminator of the GMD, components/TemporalOrderDiscriminator.py:29-46;
 * ABI revision 5): pooled[b,k,:] = sum_t m_k[b,t] feat[b,t,:] / (sum_t m_k[b,t] + 1e-6) for the target / fore / back clip ranges
 * (k = 0, 1, 2) in ONE pass over feat [B,T,D]; masks float [B,T]; pooled fp32 [B,3,D].  The backward writes dfeat [B,T,D] from
 * dpooled [B,3,D] (the masks are labels: no gradient).  D % 4 == 0.  dtype TSG_F32, or TSG_BF16 = feat / dfeat stored as bf16.
 * Sums in a fixed order: run-to-run identical.                                                                              */
int tsg_moment_pool_fwd(const void* feat, const void* m_target, const void* m_fore, const void* m_back, void* pooled,
                        int B, int T, int D, int dtype, void* stream);
int tsg_moment_pool_bwd(const void* dpooled, const void* m_target, const void* m_fore, const void* m_back, void* dfeat,
                        int B, int T, int D, int dtype, void* stream);

/* ---- batched fp32 transpose: dst[b][c][r] = src[b][r][c]; src [batch][rows] rows of cols floats, ld floats apart; dst contiguous (ABI revision 5).  The weight operands of
 * the input-gradient GEMMs (dX = dY W on tsg_gemm_f32s, which takes W^T) and W_hh^T of the LSTM backward (tsg_lstm_bwd_ws*).
 * rows % 4 == 0, cols % 4 == 0, ld % 4 == 0, ld >= cols, 16-byte aligned.                                                         */
int tsg_transpose_f32(const void* src, long long ld, void* dst, int batch, int rows, int cols, void* stream);

/* ---- dropout without a stored mask (ABI revision 5): y[i] = keep(i) ? x[i] / (1 - p) : 0 over n contiguous elements, keep(i) a counter-based
 * hash of i and of the keys -- the dropout between the layers of the BiLSTMs (networks/RNN.py:27-31, nn.LSTM(dropout=...), training mode).
 * The BACKWARD is the same call on the gradient with the same keys (the mask is regenerated, nothing is stored).  key_mode 0: keys from the
 * host's (seed, offset); 1: keys derived in the kernel from the device-resident pair rng_dev ([0] = seed, [1] = offset, uint64 each) and
 * written to keys_io (two uint32) -- for launches captured into a HIP graph whose replays advance the offset; 2: keys read from keys_io
 * (the backward of a key_mode-1 launch).  0 <= p < 1; x, y 16-byte aligned; dtype TSG_F32 / TSG_F32S (float) or TSG_BF16.            */
int tsg_dropout(const void* x, void* y, long long n, float p, uint64_t seed, uint64_t offset, const void* rng_dev, void* keys_io,
                int key_mode, int dtype, void* stream);

/* ---- LayerNorm over the channel axis: the final nn.LayerNorm(d), eps 1e-5, of QueryAwareEncoder.forward (components/VideoEncoder.py:96,112)
 * on the [rows = 2B*T, d] encoder output (ABI revision 5).  y = (x - mean) * rstd * gamma + beta with the biased variance of the row, as
 * torch.nn.LayerNorm; mean / rstd [rows] are kept for the backward.  The backward reads x and dy ONCE: dx, and dgamma / dbeta through one
 * partial row per workgroup (ws of tsg_layer_norm_bwd_ws_bytes(rows, d) bytes) added in workgroup order by a second small kernel (no
 * float atomics: run-to-run identical).  d % 4 == 0, d <= 2048.  dtype TSG_F32, or TSG_BF16 = x / y / dy / dx stored as bf16 (gamma, beta,
 * the statistics and the sums fp32).                                                                                              */
int tsg_layer_norm_fwd(const void* x, const void* gamma, const void* beta, void* y, void* mean, void* rstd,
                       long long rows, int d, float eps, int dtype, void* stream);
long long tsg_layer_norm_bwd_ws_bytes(long long rows, int d);
int tsg_layer_norm_bwd(const void* x, const void* dy, const void* gamma, const void* mean, const void* rstd, void* dx,
                       void* dgamma, void* dbeta, void* ws, long long ws_bytes, long long rows, int d, int dtype, void* stream);

/* ---- dense projection GEMM on the fp32 matrix cores ("tsg_gemm_*" of SURVEY section 8b) ------------------------------
 * y[M,N] = x[M,K] w[N,K]^T (+ bias[N], may be NULL): torch.nn.Linear's layout, i.e. the d x d projections of the path
 * (SCDM W_a / W_s attention.py:104-106, sent_linear VideoEncoder.py:48, MultiHead wq/wk/wv/wo attention.py:63-66, the
 * first boundary Linear SpanPredictor.py:62-67).  Exact fp32 (v_mfma_f32_32x32x2_f32).  K % 4 == 0.                 */
int tsg_linear_fwd(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, int dtype, void* stream);

/* The same product in the split-precision ("f32s") arithmetic with the operands converted ON LOAD (ABI revision 4): every element is
 * split into hi = rne_bf16(x), lo = rne_bf16(x - hi) in registers and hi*hi + hi*lo + lo*hi is accumulated in fp32 on the bf16
 * MFMA -- tsg_split_bf16x3 + a bf16 GEMM over the 3x longer contraction, without the operand planes in memory.  fp32 in / out,
 * y[M,N] = x[M,K] w[N,K]^T (+ bias[N], may be NULL).  M % 64 == 0, N % 256 == 0, K % 32 == 0 (TSG_E_SHAPE otherwise).  The input
 * gradient dX = dY W of the same Linears is this call with the weight passed transposed ([K,N] contiguous).                       */
int tsg_gemm_f32s(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, void* stream);
/* The same with row strides (in elements, multiples of 4) for x, w and y: a column slice of a row-major matrix is an operand as it
 * stands (ABI revision 5) -- the video half W[:, :Dv] of a head's [H, Dv + Ds] first Linear, or a slice of a wider output.        */
int tsg_gemm_f32s_ld(const void* x, long long ldx, const void* w, long long ldw, const void* bias, void* y, long long ldy,
                     int M, int N, int K, void* stream);

/* The same arithmetic with the right operand stored CONTRACTION-major: y[M,N] = x[M,K] w[K,N] (+ bias), w row-major with row stride ldw,
 * optionally as two row segments (w0: rows < kseg, w1: the rest; kseg = K, w1 = NULL for one matrix).  This is the input gradient
 * dX = dY W of a Linear with W as the parameter stores it ([N_out][K_in]; the boundary head's two first Linears as two segments): no
 * transposed copy of the weight per call.  M % 256 == 0, N % 256 == 0, K % 32 == 0, kseg % 32 == 0 (ABI revision 5).                 */
int tsg_gemm_f32s_nn(const void* x, long long ldx, const void* w0, const void* w1, int kseg, long long ldw, const void* bias,
                     void* y, long long ldy, int M, int N, int K, void* stream);
/* y[M,N] += x[M,K] w[K,N] (ABI revision 6): tsg_gemm_f32s_nn with a read-add-store epilogue.  The second consumer of an activation adds its
 * input gradient dX = dY W onto the first consumer's (reference: the recalibration block reads the BiLSTM output twice -- W_a's input and the
 * gate's r, VideoEncoder.py:52-59 / attention.py:104-121 -- and the final clip features feed both heads, SpanGroundMatchDisc.py:80-96): no
 * elementwise add kernel over the [B, T, D] gradient.  Same shape rules as tsg_gemm_f32s_nn; no bias.                                        */
int tsg_gemm_f32s_nn_acc(const void* x, long long ldx, const void* w0, const void* w1, int kseg, long long ldw,
                         void* y, long long ldy, int M, int N, int K, void* stream);

/* ---- The same projections in the bf16 STORAGE mode (ABI revision 6; csrc/gemm_bf16.hip): y [M,N] = x [M,K] . w [N,K]^T (+ bias [N], fp32 or
 * NULL); x and w are bf16 matrices (row strides ldx / ldw in elements, multiples of 8; w as an nn.Linear stores its weight), fp32 accumulation,
 * y bf16 (out_dtype TSG_BF16) or fp32 (TSG_F32), row stride ldy elements (even).  Operands go global -> LDS by DMA and are read as MFMA
 * fragments without passing a register; replaces torch.mm -> hipBLASLt for W_s / W_a / sent_linear (networks/attention.py:104-113,
 * components/VideoEncoder.py:59), the heads' first Linear (SpanPredictor.py:71-85, DistributionAlign.py:83-118) and nn.LSTM's input
 * projection and input gradient (networks/RNN.py:31,42: dX = dG W^T^T takes a transposed bf16 copy of the weight).
 * M % 128 == 0, N % 256 == 0, K % 32 == 0; 16-byte aligned pointers.                                                                       */
int tsg_gemm_bf16(const void* x, long long ldx, const void* w, long long ldw, const void* bias, void* y, long long ldy,
                  int M, int N, int K, int out_dtype, void* stream);

/* ---- The optimizer update of the whole model in one launch per 64 tensors (ABI revision 6; csrc/adam.hip): Adam as the reference builds it
 * (grounding/train.py:367-371: torch.optim.Adam(lr, weight_decay = L2 added to the gradient, eps)), torch's single-tensor formulae in fp32:
 *   g' = g * grad_scale + weight_decay * p;  m += (1 - beta1) (g' - m);  v = beta2 v + (1 - beta2) g'^2;
 *   p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps),   t = updates so far + 1.
 * params / grads / exp_avg / exp_avg_sq: HOST arrays of n device pointers (fp32, contiguous; any 4-byte alignment, 16 bytes is faster), numel: host
 * array of element counts (< 2^31).  state: caller-owned DEVICE buffer of two words {float update count, unsigned ticket}, zeroed once by the
 * caller, advanced by the kernel (no "step += 1" launch; replayable from a HIP graph).  skip: device float or NULL -- non-zero leaves the
 * parameters, the moments and the count untouched (the guard of a step whose loss was not finite or whose bounded wait expired).
 * grad_scale: 1, or 1 / world after a SUM all-reduce.  The hyper-parameters are doubles: 1 - beta and log beta are formed in double, as torch does.                                                                                     */
int tsg_adam_step(int n, const void* const* params, const void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                  const long long* numel, double lr, double beta1, double beta2, double eps, double weight_decay, double grad_scale,
                  void* state, const void* skip, void* stream);
/* The same update, additionally rewriting a bf16 shadow of every parameter that has one (shadow: n pointers, entries may be NULL; NULL = none):
 * shadow[i][e] = rne_bf16(params[i][e]) after the update -- the operand the bf16 storage mode's GEMMs read, so the per-step casts of the weights
 * (`w.to(torch.bfloat16)` once per Linear and LSTM layer) disappear.  A skipped update leaves the shadows untouched (ABI revision 6).       */
int tsg_adam_step_shadow(int n, const void* const* params, const void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                         void* const* shadow, const long long* numel, double lr, double beta1, double beta2, double eps, double weight_decay,
                         double grad_scale, void* state, const void* skip, void* stream);

/* The gradient part of the optimizer guard (ABI revision 7): *flag (device float, caller-owned, not cleared by the call) = 1 when any element of
 * the n fp32 gradient tensors is a NaN or an infinity.  engine.TsgAdam launches it in front of a guarded update with flag = the update's `skip`
 * input, so a step whose backward produced non-finite gradients (an overflow; the NaN the K1 backward writes when its exchange wait expires)
 * never reaches the parameters or the moments, also under graph replay.                                                                  */
int tsg_grads_nonfinite(int n, const void* const* grads, const long long* numel, void* flag, void* stream);

/* ---- the heads as the EPILOGUE of their own first-Linear GEMM (ABI revision 5; round-3 review: SURVEY 8f #2 "split-W Linear + ReLU
 * + dot epilogue").  Same f32s arithmetic and tiling as tsg_gemm_f32s (row tiles of 256 / 128 / 64 so that a narrow head still
 * covers the chip); the accumulator tile goes through the head's tail in registers and only [rows]-sized logits leave the kernel.
 * Rows are (b, t) pairs: M = B*T, row = b*T + t.  `y` [M,N] (the pre-activation GEMM output the backward kernels
 * tsg_match_head_bwd / tsg_boundary_score_bwd_ws read) is written only when non-NULL: under no_grad it never exists.
 * ws: caller-owned workspace of tsg_head_gemm_ws_bytes(M, N, heads) bytes (tickets + per-tile partial rows of a head wider than one
 * 256-column tile), contents irrelevant (the call zeroes its tickets).  Sums are formed in a fixed order: run-to-run identical.
 *
 * tsg_match_head_gemm (K5; VideoTextSemanticMatch, components/DistributionAlign.py:83-118):
 *   logits[row] = w2 . act(x[row,:] @ w^T + cs[b,:]) + b2      x [M,K] (ldx), w [N,K] (ldw: the slice W1[:, :Dv]), cs [B,N] = q @ W1[:, Dv:]^T
 *   + b1, w2 [N], b2 [1]; activation 0 relu / 1 tanh / 2 sigmoid.  M % 64 == 0, N % 256 == 0, K % 32 == 0.
 * tsg_boundary_head_gemm (K3; VideoSentenceConcat + MLP_predictor, CrossModalInteraction.py:44-47, SpanPredictor.py:71-85, gate
 *   SpanGroundMatchDisc.py:86): the start and the end head's first Linears are read in place as two row segments (w_start, w_end:
 *   [Hm,K] slices with row stride ldw); cs [B,2Hm], b1 / w2 [2Hm], b2 [2], gate [B,T] or NULL, mask int32 [B,T] or NULL:
 *   z = gate (x @ w^T + cs) + b1;  l = w2 . tanh(z) + b2 per head;  mask_logits;  p_start, p_end = softmax over T (the second,
 *   [B,T]-sized kernel of tsg_boundary_score_fwd).  Hm % 256 == 0, (B*T) % 64 == 0, K % 32 == 0, T <= 8192.                     */
long long tsg_head_gemm_ws_bytes(int M, int N, int heads);
int tsg_match_head_gemm(const void* x, long long ldx, const void* w, long long ldw, const void* cs, const void* w2, const void* b2,
                        void* y, void* logits, void* ws, long long ws_bytes, int M, int T, int N, int K, int activation, void* stream);
int tsg_boundary_head_gemm(const void* x, long long ldx, const void* w_start, const void* w_end, long long ldw, const void* cs,
                           const void* b1, const void* w2, const void* b2, const void* gate, const int32_t* mask, void* y,
                           void* p_start, void* p_end, void* ws, long long ws_bytes, int B, int T, int Hm, int K, void* stream);
/* softmax over T of two [B,T] logit tensors, in place (the second kernel of tsg_boundary_score_fwd, exported for the fused head) */
int tsg_boundary_softmax(void* p_start, void* p_end, int B, int T, void* stream);

/* ---- split-precision operand preparation (optional "f32s" GEMM mode) ------------------------------------------------
 * Not a reference function: the reference's Linears / LSTM input GEMMs (torch.nn.Linear, nn.LSTM; e.g.
 * attention.py:104-106, RNN.py:27) run as fp32 GEMMs.  x [rows, cols] fp32 -> three bf16 planes at
 * out[r*ld_out + p*plane_stride + c], p = 0..2: (hi, hi, lo) for the left operand, (hi, lo, hi) when
 * right_operand != 0, hi = rne_bf16(x), lo = rne_bf16(x - hi).  One bf16 MFMA GEMM with fp32 accumulate over the
 * 3x longer contraction then gives hi·hi + hi·lo + lo·hi (fp32-GEMM-level error).  cols % 4 == 0.                    */
int tsg_split_bf16x3(const void* x, void* out, long long rows, long long cols, long long ld_out, long long plane_stride,
                     int right_operand, void* stream);
/* Same from a strided, row-shifted source: output row r is source row r - row_shift (row stride ld_in floats, a multiple
 * of 4), zeros when that row is outside [0, rows) or -- with period > 0, rows being consecutive sequences of `period`
 * steps -- outside r's own sequence.  Serves the h_{t-1} operand of the LSTM weight-gradient GEMM: a column slice of
 * out [T*B, 2h] shifted by +-B rows (time-major) or of out [B*T, 2h] shifted by +-1 with period T (batch-major), without
 * materialising the shifted copy.                                                                                    */
int tsg_split_bf16x3_shift(const void* x, long long ld_in, long long row_shift, long long period, void* out, long long rows,
                           long long cols, long long ld_out, long long plane_stride, int right_operand, void* stream);
/* Transposing variant for operands contracted over the ROWS of x: out[c*ld_out + p*plane_stride + r] (r contiguous), same
 * planes, shift and zero fill.  rows % 16 == 0.  dup_offset != 0: a second copy of the output is written dup_offset
 * elements after the first (the x planes of the weight-gradient bmm appear in both directions' batches).              */
int tsg_split_bf16x3_t(const void* x, long long ld_in, long long row_shift, long long period, void* out, long long rows,
                       long long cols, long long ld_out, long long plane_stride, int right_operand, long long dup_offset,
                       void* stream);

/* ---- Weight-gradient products in the split-precision mode, operands converted on load (csrc/wgrad_split.hip) ----------------
 * The dW = dY^T X products autograd forms for the path's Linears (networks/attention.py:105-106,113-114,
 * components/SpanPredictor.py:62-72, components/DistributionAlign.py:88-94) and for nn.LSTM's W_ih / W_hh (networks/RNN.py:31,42):
 *     C[g][n][k] = sum_{m < M} A[m][g*a_group_stride + n] * Bg[m][k],      g < groups (1 or 2), n < N, k < K0 + K1,
 *     Bg[m][k] = B0[m][k]                                                  for k <  K0   (row stride ldb0),
 *              = B1[m - s_g][g*b1_group_stride + (k - K0)]  or 0           for k >= K0   (row stride ldb1; s_0 = shift, s_1 = -shift;
 *                0 when row m - s_g is outside [0, M) or, with period > 0, outside m's own run of `period` rows -- the h_{t-1} /
 *                h_{t+1} operand of dW_hh read from the LSTM output, as in tsg_split_bf16x3_shift).
 * fp32 row-major operands (row strides lda / ldb0 / ldb1 / ldc floats), fp32 output C[g] at C + g*c_group_stride.  Arithmetic:
 * every operand element x is split into hi = rne_bf16(x), lo = rne_bf16(x - hi) and hi*hi + hi*lo + lo*hi is accumulated in
 * fp32 on the bf16 MFMA -- the "f32s" arithmetic of tsg_split_bf16x3 + a bf16 GEMM, without operand planes in memory.  (Unlike
 * tsg_split_bf16x3 an infinite element leaves a NaN, not an inf, in the result.)
 * Limits: M % 32 == 0, N % 256 == 0, K0 % 128 == 0, K1 % 128 == 0, strides % 4 == 0, pointers 16-byte aligned.
 * ws: scratch of tsg_wgrad_f32s_ws_bytes(...) bytes (0: may be NULL) -- partial tiles of the row ranges the contraction is cut
 * into, added in a fixed order by a second launch: results are run-to-run identical.                                        */
long long tsg_wgrad_f32s_ws_bytes(long long M, int N, int K0, int K1, int groups);
/* How row ranges of a tile are combined (ABI revision 6).  Where the tile count does not fill the chip evenly but is at least half of it, the
 * kernels can cut the flattened (tile, chunk) space into one equal piece per CU ("stream-K": a tile's last-arriving contributor adds the partial
 * tiles in a fixed order -- no reduce launch, run-to-run identical) instead of `splits` whole row ranges per tile + a reduce kernel.  mode -1 =
 * automatic (default: stream-K for bf16 operands, where it measured 5 % faster; the row-range scheme for fp32 operands, where it did not),
 * 0 = never, 1 = always; TSG_WGRAD_SK=0/1 in the environment sets the initial mode.  tsg_wgrad_f32s_ws_bytes covers either scheme.          */
int tsg_wgrad_set_stream_k(int mode);
int tsg_wgrad_f32s(const void* A, long long lda, long long a_group_stride, const void* B0, long long ldb0, int K0,
                   const void* B1, long long ldb1, long long b1_group_stride, int K1, long long shift, long long period,
                   void* C, long long ldc, long long c_group_stride, void* ws, long long ws_bytes,
                   long long M, int N, int groups, void* stream);

/* The same with TWO outputs (ABI revision 5): the B0 segment's columns to C (C[g][n][k], k < K0, ldc >= K0), the B1 segment's to C1
 * (C1[g][n][k - K0], ldc1 >= K1): nn.LSTM's dW_ih [2][4h][I] and dW_hh [2][4h][h] as the two parameter-shaped tensors.             */
int tsg_wgrad_f32s_out2(const void* A, long long lda, long long a_group_stride, const void* B0, long long ldb0, int K0,
                        const void* B1, long long ldb1, long long b1_group_stride, int K1, long long shift, long long period,
                        void* C, long long ldc, long long c_group_stride, void* C1, long long ldc1, long long c1_group_stride,
                        void* ws, long long ws_bytes, long long M, int N, int groups, void* stream);
/* The same for bf16 operands (the arithmetic of tsg_wgrad_bf16: one bf16 MFMA per product, fp32 outputs): the LSTM layers of the bf16 storage mode. */
int tsg_wgrad_bf16_out2(const void* A, long long lda, long long a_group_stride, const void* B0, long long ldb0, int K0,
                        const void* B1, long long ldb1, long long b1_group_stride, int K1, long long shift, long long period,
                        void* C, long long ldc, long long c_group_stride, void* C1, long long ldc1, long long c1_group_stride,
                        void* ws, long long ws_bytes, long long M, int N, int groups, void* stream);

/* The same product with bf16 OPERANDS (the bf16 storage mode, ABI revision 3): A, B0, B1 are bf16 matrices (strides in elements,
 * multiples of 4; 16-byte aligned bases), C and ws fp32 as above (workspace size: tsg_wgrad_f32s_ws_bytes).  One bf16 MFMA per
 * product with fp32 accumulation; the operands are transposed through LDS with 16-bit shuffles, nothing is converted.  The library's
 * bf16 GEMM runs this shape (1024 x 1024 output over a 16384-long contraction of two row-major operands) on 64 output tiles.    */
int tsg_wgrad_bf16(const void* A, long long lda, long long a_group_stride, const void* B0, long long ldb0, int K0,
                   const void* B1, long long ldb1, long long b1_group_stride, int K1, long long shift, long long period,
                   void* C, long long ldc, long long c_group_stride, void* ws, long long ws_bytes,
                   long long M, int N, int groups, void* stream);

/* ---- Device-side input pipeline and span decoding (csrc/input_pipeline.hip) ---------------------------
 * The reference does this per sample in numpy inside DataLoader workers; these entry points do it per batch on the GPU.
 * Integer outputs are bit-exact with the reference.  int32 index tensors, fp32 features (dtype TSG_F32).
 *
 * tsg_pool_clips: CharadesDataSentence.generate_video_fts_data (dataset/charades.py:177-196).  raw [sum_b n_b, D] = the
 *   batch's i3d clip features back to back, offsets [B+1] int64 (row offsets; n_b = offsets[b+1]-offsets[b]).
 *   out [B,T,D]: row j = mean of clips 2j, 2j+1 (a lone last clip is copied), zero rows from nfeats[b] = min(ceil(n_b/2), T) on.
 *   timestamps [B,2] double (seconds) -> framestps [B,2] = min(int(x), T-1) (charades.py:178); both NULL to skip.          */
int tsg_pool_clips(const void* raw, const int64_t* offsets, const double* timestamps, void* out, int32_t* nfeats,
                   int32_t* framestps, int B, int T, int D, int dtype, void* stream);
/* tsg_sequence_masks: Sequence_mask (charades.py:12-18: ones on [max(0,st), min(et,T-1)] INCLUSIVE) for the four masks a
 *   sample carries (charades.py:167-170, charades_pair_aug.py:96-107): video_mask = [0, nfeats], temporal_labels = [s, e],
 *   fore_mask = [0, s], back_mask = [e, nfeats].  Each output is int32 [B,T]; any of them may be NULL.                      */
int tsg_sequence_masks(const int32_t* nfeats, const int32_t* spans, int32_t* video_mask, int32_t* temporal_labels,
                       int32_t* fore_mask, int32_t* back_mask, int B, int T, void* stream);
/* tsg_moment_translate: the shuffling augmentation DataAugmentForTSG.gt_moment_translate (dataset/data_augment.py:135-156) as an
 *   index gather: the ground-truth moment spans[b] = [s, e] is cut out of the first nfeats[b] clips, the gap closed and the
 *   moment re-inserted in front of position cropin[b] of the gap-closed sequence; rows >= nfeats[b] become zero.  Moments of
 *   length <= 1 or covering every clip: out[b] = video[b], span unchanged.  cropin NULL: the position is drawn uniformly from
 *   [0, nfeats-len] by a counter-based hash of (seed, b) (the reference: unseeded random.randint, data_augment.py:149).
 *   video, out [B,T,D] (out != video); new_spans [B,2] = [cropin, cropin+len-1].                                            */
int tsg_moment_translate(const void* video, const int32_t* spans, const int32_t* nfeats, const int32_t* cropin,
                         uint64_t seed, void* out, int32_t* new_spans, int B, int T, int D, int dtype, void* stream);
/* tsg_span_pred: span_pred (grounding/loss.py:53-70): (i, j) = argmax of triu(start_i + end_j) -- the zero-filled lower
 *   triangle takes part, the first maximum wins along j and then along i (torch.max).  start, end [B,T] fp32 ->
 *   pred [B,2] int64 = (i, j), score [B] fp32 = the maximum.  T <= 16384.                                                   */
int tsg_span_pred(const void* start, const void* end, int64_t* pred, void* score, int B, int T, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TSG_HIP_H */
