/* sttode_hip.h -- C ABI of libsttode_hip.so (MI355X / gfx950), the drop-in boundary of the STTODE
 * forward trajectory-forecasting hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 (or int32 where noted) owned by the caller;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream), nothing
 *     synchronises, nothing allocates: all workspaces are caller-provided (graph-capturable; a workspace of the native pipeline is
 *     initialised ONCE with sttode_workspace_init before its first use);
 *   - threads: the per-kernel entry points are re-entrant (no state); the asynchronous entry points of ONE SttodeModel (sttode_inference_*_async,
 *     sttode_wait*, sttode_async_*, sttode_set_lagged) are serialised by a mutex inside the model and take everything a call needs as
 *     arguments (no arm-then-call state) -- but the ORDER of calls decides which slot's groups a launch carries and which stream
 *     sttode_async_next_stream names, so a model's pipelined calls belong to one host thread at a time; serial calls on different
 *     workspaces may come from different threads;
 *   - return 0 on success, non-zero on failure; sttode_last_error() gives the calling thread's message;
 *   - "PK16" = weights pre-packed in MFMA fragment order by sttode_amd/packing.py (csrc/chain.hpp);
 *   - "columns" are agents (n) or trajectories (m = n*K, row = agent*K + k, model/STTODE.py:322-328).
 *
 * Each entry point names the reference interface it replaces (file:line in joyecnu/STTODE).
 */
#ifndef STTODE_HIP_H
#define STTODE_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* Version of this header: the library returns it from sttode_abi_version(); a binding compares before its first call (round 1-2: 1). */
#define STTODE_ABI_VERSION 7
int sttode_abi_version(void);
const char* sttode_last_error(void);

/* Train-mode augmentation of set_data (model/STTODE.py:417-426, rotation_2d_torch :6-14): past [n,Tp,2] and fut [n,Tf,2] (optional) rotated in
 * place about the mean of the agents' last observed positions by the angle with cosine c and sine s.  One scene per call. */
int sttode_rotate_scene(float* past, float* fut, int n, int Tp, int Tf, float c, float s, void* stream);

/* STTODENet.set_data (model/STTODE.py:397-461), batched over S independent scenes (CSR scene_ptr ==
 * seq_start_end, utils/dataloader.py:177-181): scene_orig = mean_n(last obs) (:417), normalised past (:429),
 * first-difference velocities with the first duplicated (:432-433 / inference :588-589), cur_location (:461),
 * last-agent flag for add_category (:206).  past [n,Tp,2] world coords; outputs: scene_orig [S,2],
 * agent_scene [n] int32, xpad [n,16*TPX] (flattened (t,c), zero padded), enc_in [n,Tp,4], cur [n,2],
 * orig [n,2] (scene_orig per agent), last_flag [n] int32.  vel_from_norm: 1 = inference() semantics. */
int sttode_frontend_scenes(const float* past, const int* scene_ptr, int n, int S, int Tp, int TPX, int vel_from_norm,
                           float* scene_orig, int* agent_scene, float* xpad, float* enc_in, float* cur, float* orig,
                           int* last_flag, void* stream);

/* STTODENet.set_data_nba / inference() NBA branch (model/STTODE.py:463-486,578-583): past [B*N,Tp,2], no
 * normalisation, orig = 0, last_flag = (slot == N-1). */
int sttode_frontend_nba(const float* past, int n, int N, int Tp, int TPX, float* xpad, float* enc_in, float* cur, float* orig,
                        int* last_flag, void* stream);

/* inputs_for_posterior (model/STTODE.py:430,434,457 ; 477,481): future [n,Tf,2], past_last [n,2] world. */
int sttode_frontend_future(const float* future, const float* past_last, int n, int Tf, int mode, int nba_N,
                           const float* scene_orig, const int* agent_scene, const int* scene_ptr, float* enc_in, void* stream);

/* PastEncoder.forward up to the attention in-projection (model/STTODE.py:214-223; PositionalAgentEncoding
 * :167-176; add_category :199-210; Hyp_mhsa packed in-proj hyptransformerlib.py:113-115).
 * enc_in [n,Tlen,4] -> g [n,64] (ftraj_input), qkv [n,192] (raw q|k|v, q NOT yet scaled). */
int sttode_embed_qkv(const float* fc1P, const float* fc1b, const float* posP, const float* peb, const float* fc2P,
                     const float* fc2b, const float* fc3P, const float* fc3b, const float* fc3last, const float* inP,
                     const float* inb, const float* enc_in, const int* last_flag, float* g, float* qkv, int n, int Tlen,
                     void* stream);

/* Multi-head geodesic attention core, 8 heads x 8 dims (hyptransformerlib.py:191,214-218,251-300;
 * Oblique.proj/dist core/manifolds/oblique.py:15-16,36-43):
 *   out_i = sum_j softmax_j( -acos(clamp(<r_i/|r_i|, c_j/|c_j|>, -1+1e-4, 1-1e-4)) ) v_j
 * element (seq s, batch b, feature f) at base + s*seq_stride + b*batch_stride + f (strides in floats).
 * Self-attention L == S (the reference's live path): R = k, C = q (cscale = hd^-0.5), i.e. the untransposed-score
 * quirk (:261-265).  L != S: R = q (rscale = hd^-0.5), C = k.  rowsum [Nb,8,rows] and wout [Nb,rows,cols]
 * (head-averaged weights, :306-309) are optional. */
int sttode_mhgsa_attn(const float* R, const float* C, const float* V, float* out, float* rowsum, float* wout, int rows,
                      int cols, int Nb, long rs_seq, long rs_b, long cs_seq, long cs_b, long vs_seq, long vs_b, long os_seq,
                      long os_b, float rscale, float cscale, void* stream);
/* The same core over `groups` independent problems of one shape in ONE launch: group g reads R + g gs_r, C + g gs_c, V + g gs_v and writes
 * out + g gs_o (strides in floats).  The NBA branch attends over the batch dimension of ONE forward call (hyptransformerlib.py:261-265); a
 * test set is many such batches (test.py:520-524), and one launch over all of them fills the chip.  head_dim = hidden_dim / 8: 8 for the
 * reference default, 4 / 16 for --hidden_dim 32 / 128 (train.py:38; features of head h at offset h head_dim). */
int sttode_mhgsa_attn_groups(const float* R, const float* C, const float* V, float* out, int groups, long gs_r, long gs_c, long gs_v, long gs_o,
                             int rows, int cols, int Nb, long rs_seq, long rs_b, long cs_seq, long cs_b, long vs_seq, long vs_b, long os_seq,
                             long os_b, float rscale, float cscale, int head_dim, void* stream);

/* out_proj (hyptransformerlib.py:305) -> Hypattention gate tanh(info)*sigmoid(gate) (hypertransformer.py:81-83)
 * -> TransformerEncoderLayer post-LN + FFN (hypertransformer.py:148-152) -> ODEG_Encoder: one explicit Euler step of
 * size ode_time and relu (ode_demo.py:186-190,223-231) -> pf [n,128] = cat(g, ode) (model/STTODE.py:233-235). */
int sttode_post_attn(const float* outP, const float* outb, const float* infoP, const float* infob, const float* gateP,
                     const float* gateb, const float* ln1w, const float* ln1b, const float* l1P, const float* l1b,
                     const float* l2P, const float* l2b, const float* ln2w, const float* ln2b, const float* g, const float* attn,
                     int ld_attn, float* pf, int n, float ode_time, void* stream);

/* The same layer with the integrator as a kernel parameter (the north star names RK4 / multi-step updates; the reference itself only ever
 * runs ONE Euler step, ode_demo.py:186-190, so anything else is checked against the CPU oracle only: parity unpinned).
 * method 0 Euler | 1 torchdiffeq fixed-grid "rk4" (3/8 rule) | 2 classical RK4; `steps` uniform steps over [0, ode_time].  Attention
 * length 1 only (ETH/UCY/SDD path: the attention output of a stage's state is W_v y + b_v, taken from the packed in-projection
 * inP/inb and evaluated inside the kernel); (method 0, steps 1) equals sttode_post_attn. */
int sttode_post_attn_ode(const float* outP, const float* outb, const float* infoP, const float* infob, const float* gateP,
                         const float* gateb, const float* ln1w, const float* ln1b, const float* l1P, const float* l1b,
                         const float* l2P, const float* l2b, const float* ln2w, const float* ln2b, const float* inP, const float* inb,
                         const float* g, float* pf, int n, float ode_time, int method, int steps, void* stream);

/* Right-hand side of the encoder's tensor ODE at ONE state y [n,64] for any attention length: k = LN2(h + FFN(h)), h = LN1(y + gate(out_proj(a)))
 * with `a` [n, ld_attn] the attention output of state y (TransformerEncoder_ode.forward, ode_demo.py:25-72 = TransformerEncoderLayer.forward,
 * hypertransformer.py:134-153).  The native pipeline integrates with it when a non-default integrator meets an attention group > 1:
 * per stage in-projection of y (sttode_linear_cols) -> sttode_mhgsa_attn over the group -> this -> axpy combinations. */
int sttode_post_attn_rhs(const float* outP, const float* outb, const float* infoP, const float* infob, const float* gateP,
                         const float* gateb, const float* ln1w, const float* ln1b, const float* l1P, const float* l1b,
                         const float* l2P, const float* l2b, const float* ln2w, const float* ln2b, const float* y,
                         const float* attn, int ld_attn, float* kout, int n, void* stream);
/* past_feature [n,128] = cat(ftraj_input g, relu(integrated ODE state y)) (ode_demo.py:231, model/STTODE.py:233-235) */
int sttode_ode_state_to_pf(const float* g, const float* y, float* pf, int n, void* stream);

/* DecomposeBlock front half (model/STTODE.py:62-69): conv1d(2->32,k3,pad1)+relu, GRU(32->96) final state.
 * xin [ncols,16*TPX] = flattened (x_true - x_hat) -> state [ncols,96]. */
int sttode_gru_cols(const float* xin, const float* convP, const float* convB, const float* wihP, const float* whhP,
                    const float* gbias, float* state, int ncols, int Tp, int TPX, void* stream);

/* Process-wide crossover between the latency forms of sttode_gru_cols / sttode_mlp_block0 / sttode_mlp_block1 / sttode_embed_qkv (one
 * 16-column tile per WORKGROUP, its waves splitting the rows: a single scene of test.py:171-188) and their throughput forms (a tile per
 * wave): the latency form serves calls of at most this many 16-column tiles.  Negative = leave unchanged; defaults 512 / 1024 / 1024
 * (env STTODE_GRU_LAT_TILES / STTODE_MLP_LAT_TILES / STTODE_ENC_LAT_TILES).  Both forms sum in the same order: results are bitwise
 * independent of the setting. */
int sttode_set_latency_tiles(int gru_tiles, int mlp_tiles, int enc_tiles);

/* Generic per-column linear out[col, 0:N] = act(W [X1 | X2] + b) (nn.Linear; used for the per-agent part of
 * decoder_x/decoder_y layer 0, model/utils.py:86-95, and for the stage-2 Q-net, sampler.py:39,48-52 / utils/mlp.py:26-29).
 * K1, K2, N multiples of 16.  act: 0 none, 1 relu, 2 tanh. */
int sttode_linear_cols(const float* X1, int ld1, int K1, const float* X2, int ld2, int K2, const float* WP, const float* bias,
                       float* out, int ldo, int ncols, int N, int act, void* stream);

/* The three per-agent layer-0 pre-activations of decoder_x/decoder_y (block 0) and decoder_y (block 1) in one launch:
 * A0x, A0y = W[:, pf|state] [pf | state0] + b ; A1y = W[:, pf] pf + b   (model/STTODE.py:71,74-75 split). */
int sttode_agent_preact(const float* pf, const float* state0, const float* WAx, const float* b1x, const float* WAy,
                        const float* b1y, const float* WA1, const float* b11, float* A0x, float* A0y, float* A1y, int n,
                        void* stream);

/* DecomposeBlock 0 back half for all K samples (model/STTODE.py:71-75, Decoder.forward :336-339):
 * decoder_x and decoder_y MLPs; writes dbuf = x_true - x_hat0 [m,16*TPX] and ybuf = y_hat0 [m,16*NOY].
 * A0x/A0y [n,512]: per-agent part of layer 0 (sttode_linear_cols); stream: packed weight-chunk stream of both MLPs
 * (packing.mlp_stream: fragment-ordered 18 KiB chunks, layer-2/3 biases inside), total_chunks = (32 + TPX) + (32 + NOY). */
int sttode_mlp_block0(const float* A0x, const float* A0y, const float* stream, int total_chunks,
                      const float* z, const float* xpad, float* dbuf, float* ybuf, int ncols, int K, int TPX, int NOY,
                      void* stream_);

/* DecomposeBlock 1 back half + Decoder epilogue (model/STTODE.py:338,343-346) + "+ scene_orig" (:621-622):
 * pred [m,Tf,2] = ((y_hat0 + y_hat1) + cur_location) + scene_orig.  total_chunks = 32 + NOY (24 KiB chunks). */
int sttode_mlp_block1(const float* A1y, const float* stream, int total_chunks, const float* z,
                      const float* state1, const float* ybuf, const float* cur, const float* orig, float* pred, int ncols,
                      int K, int Tf, int NOY, void* stream_);

/* One decoder MLP of a non-first DecomposeBlock with B = [z | state] per column (model/STTODE.py:71-75): raw output tiles
 * out [ncols,16*NO].  Used by the training-forward path for the last block's decoder_x (recover_traj, :339-341). */
int sttode_mlp_cols(const float* A0, const float* stream, int total_chunks, const float* z,
                    const float* state, float* out, int ncols, int K, int NO, void* stream_);

/* Fused per-trajectory chain of Decoder.forward for all K samples (model/STTODE.py:320-347 with DecomposeBlock.forward :51-77
 * twice and the "+ scene_orig" of :621-622): block-0 decoder_x -> d = x_true - x_hat0 -> block-0 decoder_y -> block-1 conv + GRU
 * -> block-1 decoder_y -> pred [m,Tf,2] = ((y_hat0 + y_hat1) + cur) + orig, ONE persistent kernel, intermediates in registers
 * (replaces sttode_mlp_block0 + sttode_gru_cols + sttode_mlp_block1 on the inference path; csrc/chain32.hip).
 * A0x/A0y/A1y [n,512] as above; pool/prog/consts: packing.chain_stream (PK32 tile pool, per-group chunk program of
 * sttode_chain_prog_len(Tp,Tf) int32 pairs, bias block); xpad [n,ldx] (ldx = 16 or 32); counter: one int32 of scratch
 * (work queue, zeroed by the call on `stream`); wgs_per_cu: resident workgroups per CU, 2 for a stand-alone call, 1 when kernels of
 * other streams should run beside it (the pipelined forms use 1). */
int sttode_traj_chain(const float* A0x, const float* A0y, const float* A1y, const float* pool, const int* prog, int prog_len,
                      const float* consts, const float* z, const float* xpad, int ldx, const float* cur, const float* orig,
                      float* pred, int* counter, int ncols, int K, int Tp, int Tf, int wgs_per_cu, void* stream);
int sttode_chain_prog_len(int Tp, int Tf);
/* sttode_gru_cols in streaming form (same function: DecomposeBlock front half, model/STTODE.py:62-69): 32-column MFMA tiles, every
 * weight tile streamed per step, 24 KiB of LDS -- its workgroups co-reside with a running sttode_traj_chain of another stream, which
 * the resident-weights kernel (144 KiB of LDS) cannot.  pool / prog / consts: packing.gru32_stream; prog_len = 13 * Tp; ldx = 16 | 32. */
int sttode_gru_cols32(const float* xin, int ldx, const float* pool, const int* prog, int prog_len, const float* consts, float* state,
                      int ncols, int Tp, void* stream);

/* compute_ADE / compute_FDE per agent (utils/metrics.py:7-26): pred [n,K,Tf,2], gt [n,Tf,2] -> ade [n], fde [n]. */
int sttode_best_of_k(const float* pred, const float* gt, int n, int K, int Tf, float scale, float* ade, float* fde, void* stream);
/* The NBA evaluation's per-horizon metric (test.py:530-551): pred [n,K,Tf,2], gt [n,Tf,2] -> out [n,Tf,2]: for agent a and horizon h (frame
 * h - 1) out[a][h-1] = (min_k mean_{t < h} |scale (pred - gt)|, min_k |scale (pred_h - gt_h)|); the caller averages over agents and weighs by
 * the batch size like the reference.  K <= 64, K Tf <= 2048. */
int sttode_horizon_metrics(const float* pred, const float* gt, int n, int K, int Tf, float scale, float* out, void* stream);

/* Stage-2 latent sampler (sampler.py:47-54): z = b (eps_mode 0) or A*eps + b with eps shared [nz] (1, share_eps) or per agent
 * [n,nz] (2); logvar = log(A^2 + 1e-8).  A, b, z, logvar [n*K, nz] (row = agent*K + k). */
int sttode_sampler_latent(const float* A, const float* b, const float* eps, int eps_mode, float* z, float* logvar, int n, int K,
                          int nz, void* stream);

/* Stage-2 objective, per agent (samplerloss.py:4-20, utils/dist.py:22-30): kld[a] = sum_{k,d} KL(N(mu,logvar) || N(pmu,plogvar))
 * (pmu / plogvar NULL = standard normal); div[a] = mean over the K(K-1)/2 sample pairs of exp(-|m_i - m_j|^2 / scale),
 * motion [n,K,D] (D = 2*Tf).  The caller sums over agents, divides by agent_num, clamps and weights. */
int sttode_sampler_loss(const float* mu, const float* logvar, const float* pmu, const float* plogvar, const float* motion, int n,
                        int K, int nz, int D, float scale, float* kld, float* div, void* stream);

/* Backward of sttode_sampler_loss: upstream g_kld[a], g_div[a] -> dmu, dlogvar [n*K,nz], dmotion [n,K,D]. */
int sttode_sampler_loss_bwd(const float* mu, const float* logvar, const float* pmu, const float* plogvar, const float* motion,
                            const float* g_kld, const float* g_div, int n, int K, int nz, int D, float scale, float* dmu,
                            float* dlogvar, float* dmotion, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training step (csrc/train.hip): forward-with-tape + backward of STTODENet.forward() (model/STTODE.py:553-568; losses
 * :372-395; what train.py:81-87 drives through total_loss.backward()).  Generic kernels over row-major nn.Parameter storage;
 * gradients are ACCUMULATED into the caller's .grad buffers.  All matrices row-major with explicit leading dimensions.
 * ------------------------------------------------------------------------------------------------ */
/* out[c, i] = mask( act( sum_j X[c / xdiv, j] * Wop[i, j] + bias[i] (+ out[c, i] if accumulate) ) ), c < cols, j < J, i < I.
 * trans = 0: Wop[i, j] = W[i*ldw + j] (nn.Linear forward);  trans = 1: Wop[i, j] = W[j*ldw + i] (input gradient dX = dY W).
 * act: 0 none | 1 relu | 2 tanh | 3 sigmoid.  mask (optional, [cols, ldm]): result zeroed where mask <= 0 (relu backward). */
int sttode_tlinear(const float* X, long ldx, int xdiv, const float* W, long ldw, int trans, const float* bias, const float* mask,
                   long ldm, float* Y, long ldy, int cols, int J, int I, int act, int accumulate, void* stream);
/* Y[c, 0:I] = act(W X[c] + tab[c / tdiv] (+ bias)): nn.Linear on cat(shared, own) with the shared part's product -- identical for tdiv
 * consecutive columns -- handed over as a table tab [cols / tdiv, ldt].  The decoder MLPs' layer 1 (model/utils.py:86-95 on
 * cat(past_feature_rep, z, state), model/STTODE.py:71-75,322-328): tab = W1[:, pf] pf + b1 per AGENT (n rows), W = W1[:, z | state] per
 * trajectory -- half the layer's products, the inference chain's layer-1 split in the training step. */
int sttode_tlinear_tab(const float* X, long ldx, const float* W, long ldw, const float* bias, const float* tab, long ldt, int tdiv, float* Y,
                       long ldy, int cols, int J, int I, int act, void* stream);
/* dW[n, k] += sum_c dY[c, n] * X[c / xdiv, k];  db[n] += sum_c dY[c, n] (db may be NULL).  Deterministic: fixed column splits,
 * partials in scratch (scratch_floats capacity; NULL = single split). */
int sttode_twgrad(const float* dY, long ldy, const float* X, long ldx, int xdiv, float* dW, long ldw, float* db, int cols, int N,
                  int K, float* scratch, long scratch_floats, void* stream);
/* Backward of one nn.Linear: dX[c, 0:Kdx] = mask(sum_n dY[c, n] W[n, 0:Kdx] (+ dX)) and dW += dY^T X, db += sum_c dY, in ONE
 * launch when cols <= 1024 (the launch-bound training regime; dX must not alias dY or X), otherwise sttode_tlinear + sttode_twgrad. */
int sttode_tlinear_bwd(const float* dY, long ldy, const float* W, long ldw, const float* mask, long ldm, float* dX, long lddx,
                       int Kdx, int accumulate, const float* X, long ldx, int xdiv, float* dW, long ldgw, float* db, int cols, int N,
                       int K, float* scratch, long scratch_floats, void* stream);
/* Batch sizes (cols > 2048) split a weight gradient's reduction over the columns; the partial sums are added into dW / db by a reduction
 * launch.  sttode_twgrad_defer(1, buf, floats): from now on the partial sums are bump-allocated from buf (device memory nothing else
 * writes) and the reductions run as ONE launch per 16 gradients -- or earlier: buf full, a destination that is already pending, another
 * stream.  sttode_twgrad_defer(0, NULL, 0): run what is pending (on the stream of the gradients that queued it), back to a reduction per
 * gradient; sttode_twgrad_defer(-1, NULL, 0): forget what is pending (error paths).  sttode_twgrad_flush(): run what is pending, keep the
 * mode.  Host-side state of the calling thread; a training step brackets its backward pass with defer(1) / defer(0) (sttode_amd/training.py). */
int sttode_twgrad_defer(int on, float* buf, long floats);
int sttode_twgrad_flush(void);
/* Grouped launches: between sttode_tgemm_group(1) and sttode_tgemm_group(0) the batch-size products (cols > 2048) of sttode_tlinear,
 * sttode_twgrad and sttode_tlinear_bwd are queued and leave as ONE launch (at most 4 products per launch; a fifth starts the next).  The
 * caller promises that the products of a group do not depend on each other and do not write the same output; calls below the batch
 * size launch at once as usual.  sttode_tgemm_group(-1): forget what is queued (error paths).  Host-side state of the calling thread (a group is
 * opened, filled and closed by one thread).  Also queued inside a group: scene-size layers (cols <= 1024) of sttode_tlinear /
 * sttode_tlinear_bwd, element-wise pieces (sttode_train_ewise) and a second sttode_ttrunk_fwd. */
int sttode_tgemm_group(int on);
/* inp0 / inp1 [n K1, ld] (inp1 may be NULL), row (a, k): columns 0..pfw-1 = pf[a] (rows of ldpf floats; pfw = 2 hidden_dim), pfw..pfw+zd-1 =
 * qz[a] for k = 0, else eps[a, k - 1] (eps [n (K1 - 1), zd]): the layer-1 input prefix of the decompose blocks (model/STTODE.py:322-331,
 * 553-566).  pfw, zd multiples of 4 (128 / 32 at the reference's defaults). */
int sttode_decoder_inputs(float* inp0, float* inp1, long ld, const float* pf, long ldpf, const float* qz, const float* eps, int n, int K1,
                          int pfw, int zd, void* stream);
/* dst[r, 0:width] = src[(r / div) % mod, 0:width] (repeat_interleave: div = K; per-frame tables: mod = T). */
int sttode_rows_copy(float* dst, long ldd, const float* src, long lds, int rows, int width, int div, int mod, void* stream);
/* dst[a, f] (+)= sum_{k<K} src[a*K + k, f]  (backward of repeat_interleave). */
int sttode_rows_reduce(float* dst, long ldd, const float* src, long lds, int rows_out, int width, int K, int accumulate, void* stream);
/* Element-wise pieces, op codes: 0 p0=p1*p2 | 1 p0+=f0*p1 | 2 gate backward (dout=p0, tanh out p1, sigmoid out p2 -> p3, p4;
 * hypertransformer.py:81-83) | 3 p0=relu(p1+f0*p2) (Euler step + relu, ode_demo.py:188,228) | 4 its backward (dout p0, out p1 ->
 * p3 += d, p4 = f0*d) | 5 rsample z=mu+eps*exp(logvar/2) (model/STTODE.py:89-93; params p1 [rows,2*i0], eps p2) | 6 p0=p1*(p2>0) |
 * 7 p0=f0 | 8 rsample backward (dz p0, params p1, eps p2 -> dparams p3 +=) | 9 p0[c,d] += p1[c / K, d % 2] (row length i0,
 * K = f0: "+ cur_location", model/STTODE.py:343-344) | 10 p0=p1*(1-p2^2) (tanh backward from its output) | 11 stage-2 latent
 * backward (sampler.py:51-53): dz p0, dlogvar p1, A p2, eps p3 -> dA p4 = dz*eps + dlogvar*2A/(A^2+1e-8); i0 = nz*4 + eps_mode,
 * f0 = K*nz | 12 p0[c,d] = p1 + p2 (+ p3[c / K, d % 2]) (row length i0, K = f0: sum of the blocks' outputs + cur_location) |
 * 13 op 4 on dout = cat(dx0 | dode), rows of p0 with leading dimension i0 (64 features): d = dode*(out p1 > 0), p3 = dx0 + d, p4 = f0*d |
 * 14 p0 = f0*p0 (+ p1) | 15 p0[r, c] += f0*p1[r*ld + c], c < width (width = i0 & 0xffff, ld = i0 >> 16).
 * Inside a group (sttode_tgemm_group) up to four pieces leave as one launch. */
int sttode_train_ewise(int op, float* p0, const float* p1, const float* p2, float* p3, float* p4, long count, int i0, float f0,
                       void* stream);
/* y = LayerNorm(x + r) over D = hidden_dim features (32 / 64 / 128; hypertransformer.py:146,151); saves xhat [rows,D], rstd [rows]. */
int sttode_add_ln_fwd(const float* x, const float* r, const float* gamma, const float* beta, float* y, float* xhat, float* rstd,
                      int rows, int D, void* stream);
/* LayerNorm backward: dsum = grad wrt (x + r); dgamma, dbeta += ; scratch >= 64 * 2 D floats. */
int sttode_ln_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dsum, float* dgamma,
                  float* dbeta, int rows, int D, float* scratch, long scratch_floats, void* stream);
/* nn.GRU cell, gate order r|z|n (model/STTODE.py:68): gi [m rows, ld ldgi] = W_ih e_t + b_ih, gh [m,288] = W_hh h + b_hh;
 * tape [m,384] = r, z, n, gh_n.  hprev NULL = zero state. */
int sttode_gru_cell_fwd(const float* gi, long ldgi, const float* gh, const float* hprev, float* hnew, float* tape, int m, void* stream);
/* dh [m,96] (grad wrt h') -> dgi (ld ldgi), dgh [m,288], dhprev = dh * z (the W_hh^T dgh term is added by sttode_tlinear). */
int sttode_gru_cell_bwd(const float* dh, const float* tape, const float* hprev, float* dgi, long ldgi, float* dgh, float* dhprev,
                        int m, void* stream);
/* The same over the WHOLE sequence in one launch (columns are independent): gi [m*Tp,288] (row = col*Tp + t), H [(Tp+1),m,96]
 * with H[0] = 0 on entry (H[t+1] = h_t written), tapes [Tp,m,384].  W_hh fragments stay in registers across the steps. */
int sttode_gru_seq_fwd(const float* gi, const float* Whh, const float* bhh, float* H, float* tapes, float* hfinal, long ldhf, int m,
                       int Tp, void* stream);   /* hfinal (optional): the final state also goes to rows of ldhf floats */
/* Whole BPTT in one launch: dh_last [m,96] = grad wrt the final state -> dgi [m*Tp,288], dgh [Tp,m,288]
 * (dh_{t-1} = dh_t * z_t + dgh_t W_hh is carried in registers / LDS). */
int sttode_gru_seq_bwd(const float* dh_last, long lddh, const float* tapes, const float* H, const float* Whh, float* dgi, float* dgh,
                       int m, int Tp, void* stream);
/* conv1d(2->32,k3,pad1)+relu (model/STTODE.py:65) on x = xa[c / adiv] - xb[c] ([m,T,2], xb optional); saves x; e [m,T,32]. */
int sttode_conv_fwd(const float* xa, int adiv, const float* xb, const float* w, const float* b, float* x, float* e, int m, int T,
                    void* stream);
/* de (already relu-masked) -> dx [m,T,2] (optional), dw [32,2,3] +=, db [32] += (deterministic: per-WG partials in scratch,
 * >= 256*224 floats, then one ordered pass). */
int sttode_conv_bwd(const float* de, const float* x, const float* w, float* dx, float* dw, float* db, int m, int T, float* scratch,
                    long scratch_floats, void* stream);
/* Backward of the geodesic self-attention (hyptransformerlib.py:191-300, scores untransposed :261-265): qkv [L*Nb, 3 D] (q|k|v,
 * row = l*Nb + slot; D = 8 head_dim), dO [L*Nb, D] (grad wrt the merged-head output before out_proj) -> dqkv [L*Nb, 3 D].
 * head_dim 4 / 8 / 16; L (4 head_dim + 4) floats of LDS (L <= 1137 at head_dim 8). */
int sttode_mhgsa_attn_bwd(const float* qkv, const float* dO, float* dqkv, int L, int Nb, int head_dim, void* stream);
/* torch.optim.Adam's step (train.py:122 constructs it, :66,87 call it) for ALL parameters in ONE launch: items = DEVICE array of n records
 * {float* p, float* m, float* v, long goff, long numel, long chunk0} (48 bytes each): parameter, first / second moment, the gradient's offset
 * in floats from gbase, element count, first 1024-element chunk (ascending from 0; `chunks` = their total).  step = t >= 1 (bias
 * corrections 1 - beta^t).  m = beta1 m + (1 - beta1) g; v = beta2 v + (1 - beta2) g^2; p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 -
 * beta2^t) + eps); weight_decay adds weight_decay * p to g first (torch's L2 form).  amsgrad / maximize: not built (the caller falls back). */
int sttode_adam_step(const void* items, int n, long chunks, const float* gbase, double lr, double beta1, double beta2, double eps,
                     double weight_decay, long step, void* stream);
/* out[0] = scale * sum (pred - target)^2 (calculate_loss_pred / _recover, :372-376,384-388); dpred optional. */
int sttode_loss_sqerr(const float* pred, const float* target, long count, float scale, float* out, float* dpred, void* stream);
/* KL term (:378-382, utils/dist.py:26-29), params [rows,2*zd].  scene_ptr NULL: out[0] = clamp_min(sum KL / denom, min_clip).
 * scene_ptr [S+1] (a batch of independent scenes in one step): out[0] = sum_s clamp_min(sum_{a in s} KL_a / N_s, min_clip), i.e. the
 * sum of the per-scene objectives (what accumulating S reference steps gives).  scratch >= max(S, 1) floats. */
int sttode_loss_kl(const float* params, const int* scene_ptr, int S, int rows, int zd, float denom, float min_clip, float* out,
                   float* dparams, float* scratch, void* stream);
/* Best-of-K term (calculate_loss_diverse :390-395); pred [n,K,D], target [n,D], K <= 64.  scene_ptr NULL: mean over the n agents;
 * with scene_ptr / agent_scene: sum over scenes of the per-scene means.  scratch >= n floats. */
int sttode_loss_diverse(const float* pred, const float* target, const int* scene_ptr, const int* agent_scene, int n, int K, int D,
                        float* out, float* dpred, float* scratch, void* stream);
/* Training step, forward of ONE encoder trunk with its tape in one launch (PastEncoder / FutureEncoder trunk, model/STTODE.py:214-236 /
 * :276-300, hypertransformer.py:134-153, ode_demo.py:188,228) for attention length 1 (scene batches: softmax over one key == 1).
 * `ptrs`: table of STT_TT_COUNT device pointers -- the trunk's parameters in their row-major nn.Parameter storage, the inputs, and every
 * tensor sttode_amd/training.py's trunk_bwd reads.  feat: row stride ld_feat, [0,64) = ftraj_input, [64,128) = ODE encoder output.
 * T <= 12 (LDS); otherwise, and for attention groups > 1, the caller runs the layer-by-layer entry points. */
enum SttodeTrunkPtr {
    STT_TT_FC1_W, STT_TT_FC1_B, STT_TT_POS_W, STT_TT_POS_B, STT_TT_FC2_W, STT_TT_FC2_B, STT_TT_FC3_W, STT_TT_FC3_B, STT_TT_INPROJ_W,
    STT_TT_INPROJ_B, STT_TT_OUT_W, STT_TT_OUT_B, STT_TT_INFO_W, STT_TT_INFO_B, STT_TT_GATE_W, STT_TT_GATE_B, STT_TT_LN1_W, STT_TT_LN1_B,
    STT_TT_L1_W, STT_TT_L1_B, STT_TT_L2_W, STT_TT_L2_B, STT_TT_LN2_W, STT_TT_LN2_B,
    STT_TT_ENC_IN /* [n,T,4] */, STT_TT_LAST /* int32 [n] */, STT_TT_PE /* [>=T,64] */, STT_TT_DROP /* [n*T,64] keep mask / keep, or NULL */,
    STT_TT_POSIN /* [n*T,128] */, STT_TT_TP /* [n*T,64] */, STT_TT_H3IN /* [n,68] */, STT_TT_FEAT, STT_TT_XC /* [n,64] */,
    STT_TT_QKV /* [n,192] */, STT_TT_AO, STT_TT_TT, STT_TT_SS, STT_TT_H, STT_TT_XH1 /* [n,64] each */, STT_TT_RS1 /* [n] */,
    STT_TT_F1 /* [n,1024] */, STT_TT_XH2, STT_TT_RS2, STT_TT_ODE,
    STT_TT_ATTN /* [n,64]: phase 2 only -- the attention output (before out_proj) of sttode_mhgsa_attn over the forward-call batch */, STT_TT_COUNT
};
/* phase 0: the whole trunk (attention length 1).  Attention over the forward-call batch (the NBA branch): phase 1 = up to the in-projection
 * (writes qkv, xc, feat[:, :64]), then sttode_mhgsa_attn(_groups) as its own launch, then phase 2 = from ptrs[STT_TT_ATTN] on. */
int sttode_ttrunk_fwd(const void* const* ptrs, int count, int n, int T, long ld_feat, float ode_time, int phase, void* stream);

/* The four terms of forward()'s objective (:372-395,553-568) and all their gradients for ONE decoder pass over K1 = 1 + K samples per
 * agent (sample 0: decoded from the posterior draw, enters the prediction / recover terms; samples 1..K: the prior draws, best-of-K):
 * pred [n,K1,D], rec [n,K1,Dp], fut [n,D], past [n,Dp], qzp [n,2*zd] -> out[0..4] = (mse, recover, kl, diverse as the three entry
 * points above compute them, then their sum = total_loss), dpred [n,K1,D], drec [n,K1,Dp] (zero rows for samples 1..K), dqzp [n,2*zd].  Two launches.
 * scratch >= 3 n + max(S, 1) floats. */
int sttode_loss_objective(const float* pred, const float* rec, const float* fut, const float* past, const float* qzp,
                          const int* scene_ptr, const int* agent_scene, int S, int n, int K1, int D, int Dp, int zd, float scale_mse,
                          float scale_rec, float kl_denom, float min_clip, float* out, float* dpred, float* drec, float* dqzp,
                          float* scratch, long scratch_floats, void* stream);
/* The same objective, gradients for the LIVE decoder columns only.  Of an agent's K1 trajectory columns only sample 0 (posterior draw: the
 * mse and recover terms, model/STTODE.py:553-560) and the best of samples 1..K (loss_diverse takes the min over K, :390-395: torch's min
 * passes its gradient to the selected sample alone) receive a gradient; the decoder treats columns independently, so the backward pass of
 * every other column is exactly zero.  -> dpred2 [n,2,D], drec2 [n,2,Dp] (row 1 zero), best [n] (1..K: which sample row 1 is), out / dqzp
 * as above.  The reference's autograd multiplies through the zero rows; a backward pass over 2 n instead of K1 n columns is the same
 * gradient (tests/test_gpu_parity.py::test_training_step_vs_reference_*). */
int sttode_loss_objective_live(const float* pred, const float* rec, const float* fut, const float* past, const float* qzp,
                               const int* scene_ptr, const int* agent_scene, int S, int n, int K1, int D, int Dp, int zd, float scale_mse,
                               float scale_rec, float kl_denom, float min_clip, float* out, float* dpred2, float* drec2, float* dqzp,
                               int* best, float* scratch, long scratch_floats, void* stream);
/* ... and the rows of the forward pass's tape for those columns: items = HOST array of `count` <= 32 records {const float* src; float* dst;
 * long src_plane; long dst_plane; int row; int outer;} (40 bytes): src is `outer` planes (src_plane floats apart) of [n K1] rows of `row`
 * floats, dst `outer` planes (dst_plane apart) of [2 n] rows; dst row 2 a + j = src row a K1 + (j ? best[a] : 0).  One launch. */
int sttode_live_rows_gather(const void* items, int count, const int* best, int n, int K1, void* stream);
/* forward() returns its four loss terms as Python floats (model/STTODE.py:568: four `.item()`, each a synchronisation with the END of the
 * queue).  Inside a replayed step the values exist after the forward half: sttode_publish_values (one launch, capturable) copies
 * vals [n <= 64] (DEVICE) into host_vals [n] (pinned HOST memory, device-accessible), increments *dev_seq (DEVICE, zero before the first
 * launch) and stores the new count into *host_seq (pinned HOST); sttode_wait_value spins on the host until *host_seq == (unsigned)want (0) or
 * timeout_s seconds have passed (non-zero) -- no stream, no event: whatever is queued behind the publishing launch keeps running. */
int sttode_publish_values(const float* vals, int n, float* host_vals, unsigned* dev_seq, unsigned* host_seq, void* stream);
int sttode_wait_value(const unsigned* host_seq, long want, double timeout_s);

/* ------------------------------------------------------------------------------------------------
 * Stand-alone manifold op library (not on the model's data flow; op-level parity).
 * Row ops, x/y/out [rows,d] (scalar results: out [rows]); op codes (hyptorch/pmath.py unless noted):
 *  0 project :98-103   1 lambda_x :128-129  2 mobius_add :171-177  3 dist :205-208   4 dist0 :231-234
 *  5 expmap :268-277 (y = u)  6 expmap0 :300-304  7 logmap :334-339  8 logmap0 :365-368  9 p2k :440-442
 * 10 k2p :445-447  11 lorenz_factor :450-469  12 Oblique.proj (core/manifolds/oblique.py:15-16)
 * 13/14 internal halves of mobius_matvec / poincare_mean.
 * ------------------------------------------------------------------------------------------------ */
int sttode_pmath_rowop(int op, const float* x, const float* y, float* out, float* out2, int rows, int d, float c, void* stream);
/* which: 0 tanh (clamp 15, pmath.py:11-12), 1 artanh (:16-22), 2 arsinh (:51-55), 3 d artanh/dx at the clamped input
 * (Artanh.backward :25-27), 4 d arsinh/dx (Arsinh.backward :57-60) */
int sttode_pmath_scalar(int which, const float* x, float* out, long n, void* stream);
/* RiemannianGradient.backward (pmath.py:39-45): out[r,:] = g[r,:] * (1 - c |x_r|^2)^2 / 4. */
int sttode_pmath_riemannian_grad(const float* x, const float* g, float* out, int rows, int d, float c, void* stream);
/* mobius_matvec (pmath.py:399-408): m [O,d], x [rows,d] -> out [rows,O]; workspaces mx_ws [rows,O], xnorm_ws [rows]. */
int sttode_pmath_matvec(const float* m, const float* x, float* mx_ws, float* xnorm_ws, float* out, int rows, int d, int O, float c,
                        void* stream);
/* which: 0 dist_matrix (pmath.py:482-493) out [P,R]; 1 _mobius_addition_batch (:416-427) out [P,R,d];
 * 2 _hyperbolic_softmax (:430-437) with x = P [C,d], y = X [B,d], A [C,d] -> out [B,C]. */
int sttode_pmath_pair(int which, const float* x, const float* y, const float* A, float* out, int P, int R, int d, float c,
                      void* stream);
/* poincare_mean over dim 0 (pmath.py:472-479): x [rows,d] -> out [d]; workspaces yl_ws [rows,d], lam_ws [rows]. */
int sttode_pmath_mean(const float* x, float* yl_ws, float* lam_ws, float* out, int rows, int d, float c, void* stream);
/* Oblique.dist (core/manifolds/oblique.py:36-43): p1 [nb,n1,d], p2 [nb,n2,d] -> acos(clamp(p2 p1^T)) [nb,n2,n1]. */
int sttode_oblique_dist(const float* p1, const float* p2, float* out, int nb, int n1, int n2, int d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Native forward pipeline: one call enqueues STTODENet.inference (model/STTODE.py:574-623) end to end.
 * ------------------------------------------------------------------------------------------------ */
typedef struct SttodeModel SttodeModel;

/* order of the packed-weight pointer table handed to sttode_model_create (sttode_amd/packing.py names) */
enum SttodeWeight {
    STT_W_FC1P, STT_W_FC1B, STT_W_POSP, STT_W_PEB, STT_W_FC2P, STT_W_FC2B, STT_W_FC3P, STT_W_FC3B, STT_W_FC3LAST, STT_W_INP,
    STT_W_INB, STT_W_OUTP, STT_W_OUTB, STT_W_INFOP, STT_W_INFOB, STT_W_GATEP, STT_W_GATEB, STT_W_LN1W, STT_W_LN1B, STT_W_L1P,
    STT_W_L1B, STT_W_L2P, STT_W_L2B, STT_W_LN2W, STT_W_LN2B,
    STT_W_B0_CONVP, STT_W_B0_CONVB, STT_W_B0_WIHP, STT_W_B0_WHHP, STT_W_B0_GBIAS, STT_W_B0_XWA, STT_W_B0_XB1, STT_W_B0_YWA,
    STT_W_B0_YB1, STT_W_B0_STREAM,
    STT_W_B1_CONVP, STT_W_B1_CONVB, STT_W_B1_WIHP, STT_W_B1_WHHP, STT_W_B1_GBIAS, STT_W_B1_YWA, STT_W_B1_YB1, STT_W_B1_STREAM,
    STT_W_CHAIN_POOL, STT_W_CHAIN_PROG, STT_W_CHAIN_CONSTS, STT_W_G0_POOL, STT_W_G0_PROG, STT_W_G0_CONSTS,
    STT_W_CHAINB3_POOL, STT_W_CHAINB3_PROG,   /* exploratory bf16-split stream of the fused launch (packing.chain_stream_b3) */
    STT_W_ROLE32_POOL, STT_W_ROLE32_PROG_SCENES, STT_W_ROLE32_PROG_NBA, STT_W_ROLE32_CONSTS_SCENES,
    STT_W_ROLE32_CONSTS_NBA,                  /* throughput-form per-agent roles of the lagged launch (packing.role_stream, round 4) */
    STT_W_COUNT
};

/* workspace buffers (offsets in floats from sttode_workspace_layout) */
enum SttodeBuffer {
    STT_B_SCENE_ORIG, STT_B_AGENT_SCENE, STT_B_XPAD, STT_B_ENC_IN, STT_B_CUR, STT_B_ORIG, STT_B_LAST, STT_B_G, STT_B_QKV,
    STT_B_ATTN, STT_B_PF, STT_B_STATE0, STT_B_A0X, STT_B_A0Y, STT_B_A1Y, STT_B_DBUF, STT_B_YBUF, STT_B_STATE1, STT_B_QUEUE, STT_B_FLAGS /* tile flags of the fused launch */,
    STT_B_ODE /* [6][n][64]: state, k1..k4, scratch of the multi-stage integrator with attention groups > 1 */, STT_B_COUNT
};

/* pipeline stages reported by sttode_timing_read */
enum SttodeStage {
    STT_STAGE_FRONTEND, STT_STAGE_EMBED, STT_STAGE_ATTN, STT_STAGE_POST, STT_STAGE_GRU0, STT_STAGE_LINEAR, STT_STAGE_MLP0,
    STT_STAGE_GRU1, STT_STAGE_MLP1, STT_STAGE_CHAIN, STT_STAGE_AGENTS /* encoder + block-0 GRU in one launch (scene batches) */,
    STT_STAGE_FUSED /* per-agent roles + trajectory chain in ONE launch (scene batches, round 3) */, STT_STAGE_COUNT
};

/* STTODENet.__init__ + load_state_dict equivalent for the packed weights (model/STTODE.py:350-366).  The library keeps the weight pointers
 * (the caller keeps the tensors alive) and the model's events; the pipeline's HIP streams are created with the first model of a device
 * and shared by every later model of that device for the life of the process (the runtime deals its few hardware queues to streams
 * round-robin at creation: a later model's own streams could land on the caller's queue and serialise the pipeline). */
int sttode_model_create(SttodeModel** out, const void* const* weights, int count, int Tp, int Tf, int K, int n_chunks0,
                        int n_chunks1);
/* Hand over (or replace) one entry of the weight table after creation.  The two entries of the exploratory bf16-split stream
 * (STT_W_CHAINB3_*) may be NULL at creation and arrive here before sttode_set_mfma_mode(m, 1). */
int sttode_model_set_weight(SttodeModel* m, int index, const void* ptr);
int sttode_model_destroy(SttodeModel* m);
int sttode_workspace_layout(const SttodeModel* m, int n, int S, long* offsets /*[STT_B_COUNT]*/, long* total_floats);
/* ONCE per workspace (after allocation, before its first use by sttode_inference_*): zeroes the hand-off flag words (STT_B_FLAGS) and marks
 * the workspace initialised.  The one-launch scene form (sttode_set_scene_launch) keeps its flag words zero from then on -- the last
 * workgroup of every launch zeroes them again -- so it needs no memset in front of a launch and a captured sttode_inference_scenes replays
 * correctly; launched on a workspace that was never initialised it refuses the flag words it finds: NaN predictions, time-out word 2. */
int sttode_workspace_init(SttodeModel* m, float* workspace, int n, int S, void* stream);
/* number of column parts (1..8) the per-trajectory kernels are pipelined over on separate streams (default 1,
 * or env STTODE_COL_PARTS); results are bitwise independent of it. */
int sttode_set_col_parts(SttodeModel* m, int parts);
/* per-trajectory stage: 1 = fused chain kernel (sttode_traj_chain), 0 = the three-kernel form (mlp_block0 -> gru_cols -> mlp_block1),
 * -1 = automatic (fused when the batch has >= 128 trajectories per workgroup slot to fill; default, or env STTODE_CHAIN). */
int sttode_set_chain(SttodeModel* m, int mode);
/* Calls whose per-trajectory stage takes the fused chain: 1 (default, or env STTODE_FUSED) = the per-agent stage (encoder, block-0 GRU,
 * layer-1 pre-activation tables: PastEncoder.forward model/STTODE.py:214-236, DecomposeBlock.forward :62-75 of block 0) runs as the
 * leading workgroups of the chain launch, trajectory groups wait on one flag per 16-agent tile (attention groups > 1, the NBA branch:
 * the embedding and the attention stay launches in front, the roles start at the post-attention layer); 2 = as 1, and for scene batches
 * the roles also run set_data's normalisation of their tile (model/STTODE.py:397-461): the call is ONE launch; 0 = separate launches on
 * the pipeline's per-agent stream; 3 = as 1 with the roles interleaved 160 groups ahead of their consumers in the grid instead of all
 * in front (sttode_fused_block_of; measured neutral pipelined, slower serial); 4 = as 1 with a tile's stage split into FIVE role workgroups
 * -- encoder beside block-0 GRU, then one per layer-1 table: latency max(E, G) + one table instead of their sum; measured -1 % at 512
 * scenes, +3 % on the NBA-128 leg, not the default.  Results are bitwise the same in every mode. */
int sttode_set_fused(SttodeModel* m, int mode);
/* EXPLORATORY, opt-in, never the default (own dtype label in bench.py): 1 = the per-trajectory chain (the three decoder MLPs and block 1's
 * conv + GRU: DecomposeBlock.forward model/STTODE.py:51-77, Decoder.forward :320-347) runs its matrix products as a three-way bf16
 * split on the bf16 matrix cores -- x = hi + mid + lo, six products per k block, fp32 accumulate: fp32-class accuracy, held to the same
 * golden vectors at the same 1e-4 -- instead of fp32 MFMAs; 0 (default, or env STTODE_BF16X3) = fp32 everywhere.  The per-agent stage,
 * the attention and every call below the chain threshold stay fp32 in both modes. */
int sttode_set_mfma_mode(SttodeModel* m, int mode);
/* Fault injection for tests of the in-launch hand-off's give-up path: in fused launches the per-agent role of 16-agent tile `tile` computes
 * its tables but never publishes its flag (-1: off, the default).  The trajectory groups that read that tile then run into the bound of
 * their spin (~1 s), write NaN into their predictions and set the time-out word (workspace buffer STT_B_FLAGS, word [tiles]; with the split roles of
 * mode 4 the flag words are E [tiles] | time-out | G [tiles] | tables [3 tiles] and the withheld flags are the tile's three table flags) -- the
 * launch ends, it never hangs; every other group is unaffected.  The model's host-visible word (sttode_timeout_word) is set as well. */
int sttode_debug_drop_role_flag(SttodeModel* m, int tile);
/* Host staging of one scene for the one-scene-per-call loop (test.py:171-188 -> set_data, model/STTODE.py:397-404): pre [N][2][Tp] and
 * fut [N][2][Tf] (HOST pointers, the loader's layout; fut may be NULL with Tf = 0) are transposed into a pinned ring slot and copied to
 * dev [N*Tp*2 + N*Tf*2] (DEVICE: past [N][Tp][2] followed by future [N][Tf][2]) with one asynchronous copy on `stream`. */
int sttode_stage_scene(const float* pre, const float* fut, int N, int Tp, int Tf, float* dev, void* stream);
/* The same ring for a batch whose host layout IS the device layout (set_data_nba, model/STTODE.py:463-486: past_traj [B,N,Tp,2],
 * future_traj [B,N,Tf,2]): a [na] and b [nb] floats (HOST, pageable; b may be NULL with nb = 0) -> dev [na4 + nb] (DEVICE; a at 0, b at
 * na4 = na rounded up to a multiple of 4, so both start on 16-byte boundaries) with one
 * asynchronous copy on `stream` (a pageable `.to(device)` waits for everything queued on the stream: in train.py:61-67 the previous step). */
int sttode_stage_rows(const float* a, long na, const float* b, long nb, float* dev, void* stream);
/* The pipeline stream the next sttode_inference_*_async call of n agents will run on (*stream; NULL when the call will not take the
 * one-stream fused form).  Inputs prepared on that stream, and the call issued from it, need no cross-stream event. */
int sttode_async_next_stream(SttodeModel* m, int n, void** stream);
/* Best-of-K ADE / FDE of an asynchronous call's predictions (utils/metrics.py:7-26, as sttode_best_of_k) enqueued on the pipeline stream
 * the call of `slot` runs on: the metrics start the moment the call's launch drains, in stream order.  Re-records the slot's completion
 * event behind them (sttode_wait(slot) then covers the metrics).  gt must have been written before the sttode_inference_*_async call
 * (its stream waits for the caller's stream at that point). */
int sttode_async_best_of_k(SttodeModel* m, int slot, const float* pred, const float* gt, int n, int K, int Tf, float scale, float* ade,
                           float* fde);
/* Serial scene calls (sttode_inference_scenes) below the chain threshold -- the reference's evaluation loop hands over ONE scene per call
 * (test.py:171-188) -- run as ONE launch whose workgroups take the roles front-end + per-agent stage / block-0 decoder_y / block-0
 * decoder_x -> block-1 GRU -> block-1 decoder_y and hand tables over through flags (csrc/scene_lat.hip), when the call has at most
 * `max_tiles` 16-trajectory tiles: -1 = default (128, or env STTODE_SCENE_LAUNCH), 0 = never (six launches, as round 2).  Bitwise the
 * same predictions either way.  Replaces: model/STTODE.py:397-461 + :553-627 for one scene. */
int sttode_set_scene_launch(SttodeModel* m, int max_tiles);
/* Grid order of the fused launch (host-callable, no GPU): block -> group index (>= 0) or -1 - tile for the per-agent role of a 16-agent
 * tile; roles sit `lead` groups ahead of the first group that reads their tables, so every producer has a smaller block index than its
 * consumers (Decoder.forward's repeat_interleave layout, model/STTODE.py:322-328: trajectory = agent * K + k). */
int sttode_fused_block_of(long block, long tiles, long groups, long K, long lead);
/* integrator of the tensor-ODE encoder inside the native pipeline: method / steps as in sttode_post_attn_ode (default 0, 1 = reference).
 * Attention length 1 (scene batches): every stage inside the fused encoder kernel.  Attention groups > 1 (sttode_inference_nba): every
 * stage is a pass over the group -- in-projection of the state, sttode_mhgsa_attn, sttode_post_attn_rhs, axpy combinations -- enqueued
 * natively by the same call (odeint over TransformerEncoder_ode, ode_demo.py:186-190; parity unpinned: the reference only takes one Euler step). */
int sttode_set_ode(SttodeModel* m, int method, int steps);
/* every = 0: off; n > 0: bracket the stages of every n-th forward call with hipEvents recorded on the launch streams (the
 * per-trajectory stages are bracketed on every call while n > 0) */
int sttode_timing_enable(SttodeModel* m, int every);
/* total_ms: sum of the launches' durations; busy_ms (optional): length of the union of their [start, end] intervals -- launches of
 * consecutive pipelined calls overlap, so a stage's rate is work / busy time, not work / mean launch duration. */
int sttode_timing_read(SttodeModel* m, double* total_ms /*[STT_STAGE_COUNT]*/, int* launches /*[STT_STAGE_COUNT]*/,
                       double* busy_ms /*[STT_STAGE_COUNT] or NULL*/);

/* set_data (batched over scenes) + inference (model/STTODE.py:397-461,574-623; caller loop test.py:171-184).
 * past [n,Tp,2] world coords, scene_ptr [S+1], z [n*K,32] -> pred [n,K,Tf,2] world coords. */
int sttode_inference_scenes(SttodeModel* m, const float* past, const int* scene_ptr, int n, int S, const float* z,
                            float* workspace, float* pred, void* stream);
/* set_data_nba + inference, NBA branch (model/STTODE.py:463-486,578-583; caller test.py:520-524):
 * past [B*N,Tp,2]; attention length = B over the N agent slots. */
int sttode_inference_nba(SttodeModel* m, const float* past, int B, int N, const float* z, float* workspace, float* pred,
                         void* stream);
/* G forward-call batches of the NBA branch in ONE call (what G sttode_inference_nba calls compute; the reference's evaluation loop makes one
 * model call per DataLoader batch, test.py:520-524): past [G][B][N][Tp][2], z [G B N K][32] -> pred [G B N][K][Tf][2].  The attention
 * (Hyp_mhsa over the batch dimension, hyptransformerlib.py:261-265) runs within each batch of B scenes -- one launch with a group dimension
 * -- and everything per agent / per trajectory over all G B N agents: a test set of 128-scene batches fills the chip. */
int sttode_inference_nba_groups(SttodeModel* m, const float* past, int G, int B, int N, const float* z, float* workspace, float* pred,
                                void* stream);

/* Pipelined forms (STTODENet.inference as a stream of calls, model/STTODE.py:574-623; caller loop test.py:171-184): consecutive calls run
 * on the pipeline's internal streams so that the grid tail of one call's big launch is filled by the next call's.
 *   LAGGED form (round 4; default for every call whose per-trajectory stage takes the chain, reference integrator): calls rotate over
 *   `streams` pipeline streams (sttode_set_lagged; default 3); the ONE launch a call enqueues carries ITS per-agent stage in throughput form
 *   (128 agents per workgroup, csrc/role32.hpp) followed by the trajectory groups of the call made `streams` calls earlier on the same
 *   stream -- whose per-agent tables the previous launch of that stream produced.  Nothing inside a launch depends on anything else in it
 *   (no flags, no spinning).  A call's predictions are therefore produced by the launch of a LATER call -- or by sttode_wait(slot) /
 *   sttode_async_best_of_k(slot), which enqueue the outstanding groups of that slot as a launch of their own if no later call has done so.
 *   Slots: up to 8 (slot in [0, 8)); a loop that keeps 2 x streams calls in flight never waits on another stream.  Agrees with the serial
 *   forms to fp32 rounding (different MFMA tiling and host-folded embedding in the per-agent stage), not bitwise.
 *   sttode_set_lagged(m, 0): the round-3 forms (fused launches of one call each on three streams / separate per-agent launches), bitwise
 *   the serial forms.  Calls below the chain threshold or with a non-default integrator always take those forms.
 * workspace, pred and z of a slot must stay untouched until sttode_wait(slot) has been enqueued on the consuming stream. */
int sttode_set_lagged(SttodeModel* m, int streams /* 0 = off, 2 or 3 (default, or env STTODE_LAGGED) */);
/* Per-call options of the asynchronous entry points (NULL = none).  Everything a call needs travels WITH the call: a request the call's
 * form cannot honour makes the call fail before anything is enqueued or any state of the model changes (ABI <= 6 armed the model with
 * sttode_async_device_latents / sttode_async_fused_metrics before the call; a failed check in between left the request armed).
 *   device_latents != 0 (lagged form only; replaces the torch.randn_like of Normal.rsample, model/STTODE.py:89-93,609-616, for that call): the
 *     call treats its `z` argument as an OUTPUT buffer [n K][32] which its own per-agent roles fill with N(0, I) samples (Philox4x32-10, 64-bit
 *     key zkey, counter = the float4's index; two Box-Muller pairs per block) before its trajectory groups read it.
 *   metrics_gt != NULL (lagged form only; replaces a sttode_best_of_k launch per call): the call's trajectory groups also compute its
 *     min-over-K ADE / FDE (compute_ADE / compute_FDE, utils/metrics.py:7-26; the values of sttode_best_of_k on the same predictions, bit for
 *     bit) against metrics_gt [n][Tf][2], scaled by metrics_scale, into ade / fde [n]; valid once sttode_wait(slot) has passed.
 *   nba_groups > 1 (sttode_inference_nba_async only): the call carries that many forward-call batches [G][B][N][Tp][2]; the attention runs
 *     within each batch of B scenes (sttode_inference_nba_groups).
 * sttode_async_is_lagged(m, n) tells beforehand whether a call of n agents will take the lagged form. */
typedef struct SttodeAsyncOpts {
    int device_latents;
    unsigned long long zkey;
    const float* metrics_gt;
    float* ade;
    float* fde;
    float metrics_scale;
    int nba_groups;
} SttodeAsyncOpts;
/* Measurement aid: the shader clock at this moment.  out[0] = shader cycles, out[1] = ticks of the constant 100 MHz clock over ~20 us on
 * one lane (device memory, two int64): GHz = out[0] / (10 out[1]). */
int sttode_clock_probe(long long* out, void* stream);
/* Device -> pinned host copy of `bytes` (multiple of 16; both pointers 16-byte aligned; dst: device-accessible host memory) by `wgs`
 * persistent workgroups (<= 0: 8) on `stream`: a copy that leaves the chip's workgroup slots to the kernels running beside it. */
int sttode_copy_to_host(void* dst, const void* src, long bytes, int wgs, void* stream);
/* Enqueue the outstanding trajectory-group launch of the call in `slot` now (no-op when a later call has carried it already). */
int sttode_async_enqueue(SttodeModel* m, int slot);
/* Enqueue every outstanding trajectory-group launch of the lagged form (before buffers of pending calls are released or reused). */
int sttode_async_flush(SttodeModel* m);
int sttode_inference_scenes_async(SttodeModel* m, const float* past, const int* scene_ptr, int n, int S, const float* z,
                                  float* workspace, float* pred, int slot, const SttodeAsyncOpts* opts, void* stream);
int sttode_inference_nba_async(SttodeModel* m, const float* past, int B, int N, const float* z, float* workspace, float* pred,
                               int slot, const SttodeAsyncOpts* opts, void* stream);
/* sttode_horizon_metrics of an asynchronous call's predictions on the pipeline stream the call of `slot` runs on (as sttode_async_best_of_k:
 * starts the moment the call's groups drain; the slot's completion event is re-recorded behind it). */
int sttode_async_horizon_metrics(SttodeModel* m, int slot, const float* pred, const float* gt, int n, int K, int Tf, float scale, float* out);
int sttode_wait(SttodeModel* m, int slot, void* stream);
/* Zero-copy futures (lagged form): the trajectory groups of a lagged call only WRITE `pred` (block 0's y_hat0 waits in the workspace), so
 * `pred` may be pinned host memory addressed by its host pointer: the futures reach the host with the launch itself, no D2H copy
 * (test.py:186-188 moves every prediction to NumPy).  sttode_async_is_lagged: 1 if the next async call of n agents takes that form (a
 * query, not a status); sttode_wait_host: the HOST waits for the call of `slot` (its outstanding groups are enqueued first). */
int sttode_async_is_lagged(SttodeModel* m, int n);
int sttode_wait_host(SttodeModel* m, int slot);
/* Health of the in-launch hand-off (round-3 fused launches and the one-launch scene form: a group whose producer never signalled gives up
 * after ~1 s, poisons its predictions with NaN and sets the launch's time-out word -- in the workspace AND in the model's word in pinned
 * host memory).
 * sttode_timeout_word: *word = address of the model's host-visible time-out word (valid for the model's life): non-zero once ANY launch of
 * this model gave up since the last sttode_timeout_clear.  Reading it costs a host load -- no stream operation, no synchronisation -- so a
 * caller checks it at its next call and before handing results out (STTODENet.inference() / wait() do, and raise).  The word is set when
 * the group gives up, i.e. it is seen by whoever looks after the launch has finished.
 * sttode_check: reads the time-out word of the LAST launch that used `workspace` (laid out for n agents / S scenes) after synchronising
 * `stream`: returns 0 if it is clear, 3 (and sttode_last_error) if a group gave up or the workspace was never initialised -- the caller's
 * predictions of that call are not valid.  Lagged launches have no hand-off and always pass. */
int sttode_timeout_word(SttodeModel* m, const unsigned** word);
int sttode_timeout_clear(SttodeModel* m);
int sttode_check(SttodeModel* m, const float* workspace, int n, int S, void* stream);

#ifdef __cplusplus
}
#endif
#endif
