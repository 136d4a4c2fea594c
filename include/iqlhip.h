/*
 * iqlhip.h — C ABI of libiqlhip.so: the MI355X (gfx950) implementation of the
 * IQL gradient step of LaurenYTaylor/jsrl-CORL.
 *
 * The reference has no FFI: its boundary is the Python class surface of
 * algorithms/finetune/iql.py (ReplayBuffer :122-197, ImplicitQLearning :445-606).
 * Each entry point below names the reference method whose device work it
 * replaces; jsrl-corl_amd/iql.py is the Python shim that keeps those classes'
 * signatures and calls these symbols through ctypes.  INTEGRATION.md shows the
 * binding a reference maintainer would add.
 *
 * Conventions
 *  - plain C types only; every pointer named *_dev is device memory owned by the
 *    CALLER (torch tensors in the shim); the library never frees it.
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *    all calls are asynchronous on it unless stated otherwise.
 *  - return 0 on success, negative IQLHIP_E* otherwise; iqlhip_last_error() gives
 *    the message (thread-local).  No C++ exception crosses the boundary.
 *  - a context is not thread-safe; one context per (process, GPU).
 *  - all arithmetic is fp32 ("f32" in bench.py's dtype); GEMMs run on
 *    v_mfma_f32_16x16x4_f32 (exact fp32 fma chains).
 */
#ifndef IQLHIP_H
#define IQLHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IQLHIP_VERSION 300          /* 0.3.0 */
#define IQLHIP_HIDDEN 256           /* hidden width the kernels are tiled for (reference default, iql.py:352) */
#define IQLHIP_MAX_INPUT 128        /* max state_dim + action_dim */
#define IQLHIP_MAX_ACTION 32        /* max action_dim */
#define IQLHIP_MAX_WORLD 8          /* ranks of one data-parallel group (the GPUs of one node) */
#define IQLHIP_GRAPH_STEPS 64       /* steps per captured hipGraph chunk of iqlhip_train_steps */

enum {
  IQLHIP_OK = 0,
  IQLHIP_EINVAL = -1,    /* bad argument (maps to ValueError in the shim) */
  IQLHIP_EHIP = -2,      /* a HIP runtime call failed (RuntimeError) */
  IQLHIP_ENOTBOUND = -3, /* step before iqlhip_bind */
  IQLHIP_EUNSUPPORTED = -4, /* dims the kernels are not built for (NotImplementedError) */
  IQLHIP_EINDEX = -5,    /* a row index outside the buffer (IndexError, like the reference's tensor indexing iql.py:173-177) */
  IQLHIP_EEXCHANGE = -6  /* a peer of the P2P gradient exchange did not arrive in time: replicas out of sync (RuntimeError) */
};

enum { IQLHIP_NET_V = 0, IQLHIP_NET_Q1 = 1, IQLHIP_NET_Q2 = 2, IQLHIP_NET_PI = 3 };
enum { IQLHIP_POLICY_GAUSSIAN = 0, IQLHIP_POLICY_DETERMINISTIC = 1 };

/* ---- dimensions -------------------------------------------------------- */
typedef struct {
  int32_t state_dim;     /* S */
  int32_t action_dim;    /* A */
  int32_t hidden_dim;    /* must be IQLHIP_HIDDEN */
  int32_t n_hidden;      /* must be 2 (MLP [in,256,256,out], iql.py:351-356) */
  int32_t policy;        /* IQLHIP_POLICY_* (GaussianPolicy :347 / DeterministicPolicy :382) */
  int32_t max_batch;     /* largest batch a step will be called with */
} iqlhip_dims;

/* Offsets (in floats) of one MLP's tensors inside the flat parameter arena.
 * Weights keep torch.nn.Linear's [out,in] row-major layout. */
typedef struct {
  int64_t seg_begin, seg_end;  /* [begin,end) of this net's segment, multiples of 64 */
  int64_t w0, b0, w1, b1, w2, b2, log_std;  /* log_std = -1 when absent */
  int32_t k_in;    /* input width of layer 0: S (V, pi) or S+A (Q) */
  int32_t d_out;   /* 1 (V, Q) or A (pi) */
} iqlhip_net_layout;

typedef struct {
  iqlhip_net_layout net[4];   /* IQLHIP_NET_* order: V, Q1, Q2, PI */
  int64_t n_params;           /* floats in the trainable arena (= Adam exp_avg / exp_avg_sq arenas) */
  int64_t n_target;           /* floats in the target arena: copy of the [Q1,Q2] segments */
  int64_t target_src;         /* arena offset of Q1's segment: target[i] mirrors params[target_src+i] */
} iqlhip_layout;

/* Pure host function (no GPU needed).  Replaces nothing in the reference: the
 * reference keeps one tensor per nn.Parameter; the shim re-homes them as views
 * into ONE arena laid out by this function so a step touches three flat buffers. */
int iqlhip_arena_layout(const iqlhip_dims* dims, iqlhip_layout* out);

/* ---- hyper-parameters and per-step scalars ------------------------------ */
typedef struct {
  float iql_tau;    /* expectile, ImplicitQLearning(iql_tau)  iql.py:454,490 */
  float beta;       /* inverse temperature                    iql.py:455,524 */
  float discount;   /* gamma                                  iql.py:457,506 */
  float tau;        /* Polyak rate, cast of the python float  iql.py:458,515 */
  float one_minus_tau; /* (float)(1.0 - tau) formed in float64 first (iql.py:74) */
  float exp_adv_max;   /* EXP_ADV_MAX = 100                   iql.py:26 */
  float log_std_min, log_std_max; /* -20, 2                  iql.py:27-28 */
} iqlhip_hyper;

/* Host-computed (float64 -> float32) scalars of torch.optim.Adam's
 * _single_tensor_adam for THIS step; group order V, Q, PI. */
typedef struct {
  float step_size[3];   /* lr_g / (1 - beta1^t_g) */
  float bc2_sqrt[3];    /* sqrt(1 - beta2^t_g) */
  float beta2;          /* (float)beta2 */
  float one_minus_beta1;/* (float)(1 - beta1) : lerp weight */
  float one_minus_beta2;/* (float)(1 - beta2) */
  float eps;
  float grad_scale;     /* 1 for single GPU; 1/world after a summed all-reduce */
  float inv_batch;      /* 1 / (rows the batch means divide by): 1/B, or 1/(B*world) under DP */
} iqlhip_step_scalars;

/* One batch, either gathered already (idx_dev == NULL; five separate row-major
 * tensors as returned by ReplayBuffer.sample, iql.py:171-178) or addressed
 * through int64 row indices into buffer storage (ld_* = row strides in floats). */
typedef struct {
  const float* s_dev; const float* a_dev; const float* r_dev; const float* ns_dev; const float* d_dev;
  int64_t ld_s, ld_a, ld_r, ld_ns, ld_d;
  const int64_t* idx_dev;
  int32_t rows;
} iqlhip_batch;

typedef struct iqlhip_ctx iqlhip_ctx;

/* ---- life cycle --------------------------------------------------------- */
int iqlhip_version(void);
const char* iqlhip_last_error(void);

/* ImplicitQLearning.__init__ (iql.py:446-480): allocates library-owned scratch
 * (activations, gradient slabs, loss words) on `device`. */
int iqlhip_create(const iqlhip_dims* dims, const iqlhip_hyper* hyper, int device, iqlhip_ctx** out);
int iqlhip_destroy(iqlhip_ctx* ctx);
int iqlhip_set_hyper(iqlhip_ctx* ctx, const iqlhip_hyper* hyper);

/* Arithmetic of the large matrix products (layer 0 and layer 1 forward, dW1, dH0, dW0): 0 = fp32 MFMA (default; the
 * parity path), 1 = operands rounded to bf16, fp32 accumulate (v_mfma_f32_16x16x32_bf16).  The heads, master weights,
 * activations in memory, losses, Adam and Polyak stay fp32.  The reference has no reduced-precision mode; this
 * is BASELINE config 5's "MFMA bf16 path" and is checked against the fp32 fixtures at 2e-2. */
int iqlhip_set_precision(iqlhip_ctx* ctx, int mode);

/* Actor dropout (GaussianPolicy/DeterministicPolicy(..., dropout=p) -> nn.Dropout(p) after each hidden ReLU,
 * iql.py:331-333, active while the actor is in train mode): p in [0,1), 0 = off.  Keep-masks are drawn on
 * the device (Philox4x32-10 keyed by `seed` and a per-step counter). */
int iqlhip_set_dropout(iqlhip_ctx* ctx, float p, uint64_t seed);
/* The context's Philox stream positions {dropout step, act() call}: read them from a context that is about to be
 * replaced and set them on its successor, so that neither random stream restarts mid-run. */
int iqlhip_get_counters(const iqlhip_ctx* ctx, uint64_t out[2]);
int iqlhip_set_counters(iqlhip_ctx* ctx, const uint64_t in[2]);
/* Tests: inject keep-bits for the next steps instead of drawing them ([rows][8] uint32 per layer, bit j of
 * word w = hidden unit 32w + j); cleared by the next iqlhip_set_dropout. */
int iqlhip_debug_write_masks(iqlhip_ctx* ctx, const uint32_t* keep0_host, const uint32_t* keep1_host, int32_t rows,
                             void* stream);

/* Bind the caller-owned arenas: params (n_params), target (n_target; the
 * deepcopy q_target of iql.py:461), Adam exp_avg / exp_avg_sq (n_params each). */
int iqlhip_bind(iqlhip_ctx* ctx, float* params_dev, float* target_dev, float* exp_avg_dev, float* exp_avg_sq_dev);

/* ---- the step ----------------------------------------------------------- */
/* ImplicitQLearning.train(batch) (iql.py:542-563) minus the host syncs: forward
 * of V(s'), V(s), Qt1, Qt2, Q1, Q2, pi; the three losses; backward; Adam on the
 * three groups; Polyak.  Losses land in device words read by iqlhip_read_losses. */
int iqlhip_step(iqlhip_ctx* ctx, const iqlhip_batch* batch, const iqlhip_step_scalars* sc, void* stream);

/* The same step WITH the host synchronisation train() ends on (the three .item() calls of iql.py:491,509,535 as one):
 * the losses land in host-mapped pinned words followed by a completion word the host spins on — no stream synchronise
 * (12-17 us of host time after the GPU has finished, profiles/r03_sync_cost.txt).  Returns when out[3] = {value, q, actor}
 * is there; the parameter update itself stays ordered by `stream` like after iqlhip_step. */
int iqlhip_step_sync(iqlhip_ctx* ctx, const iqlhip_batch* batch, const iqlhip_step_scalars* sc, float out[3], void* stream);
/* ... in two halves, for a host that has work to do while the GPU runs the step: begin launches, wait returns the
 * losses of the step begun last. */
int iqlhip_step_begin(iqlhip_ctx* ctx, const iqlhip_batch* batch, const iqlhip_step_scalars* sc, void* stream);
int iqlhip_step_wait(iqlhip_ctx* ctx, float out[3], void* stream);

/* One iteration of the online fine-tuning loop's device work (algorithms/finetune/iql.py:741-773, jsrl_w_iql.py:
 * 512-548: add_transition -> sample -> train) in one call: the new packed transition row_host[ld] is stored at ring
 * row `pointer` of rows_dev (capacity rows), the batch rows_dev[idx_host[0..n)] (indices as np.random.randint drew them,
 * AFTER the insert, iql.py:172) is gathered and one IQL step runs on it.  row_host / idx_host are ordinary host
 * memory (copied into pinned staging inside the call).  Synchronous: returns the three losses in out[3].
 * act_state_host != NULL additionally evaluates the NEXT iteration's actor.act(state) (iql.py:371-379) with the
 * just-updated policy before the call's one synchronisation: state_dim floats in, action_dim floats out
 * (act_out_host); act_seed != 0 draws the training-mode noise on the device like iqlhip_actor_sample. */
int iqlhip_online_step(iqlhip_ctx* ctx, float* rows_dev, int64_t ld, int64_t capacity, int64_t pointer,
                       const float* row_host, const int64_t* idx_host, int32_t n, const iqlhip_step_scalars* sc,
                       float out[3], const float* act_state_host, float max_action, uint64_t act_seed,
                       float* act_out_host, void* stream);

/* Data-parallel split of the same step: forward+backward, then the
 * flat gradient (n_params floats + 4 tail words: 3 loss sums and a spare) is
 * written to grads_dev for the caller's all-reduce, then the update consumes it. */
int iqlhip_forward_backward(iqlhip_ctx* ctx, const iqlhip_batch* batch, const iqlhip_step_scalars* sc,
                            float* grads_dev, void* stream);
int iqlhip_apply_update(iqlhip_ctx* ctx, const float* grads_dev, const iqlhip_step_scalars* sc, void* stream);
int64_t iqlhip_grad_words(const iqlhip_ctx* ctx);   /* n_params + 4 */

/* n_steps consecutive `sample -> train` iterations without host round trips: row indices are drawn on the device
 * (Philox4x32-10, uniform with replacement over [0,size) like np.random.randint at iql.py:172; index j of the call comes
 * from counter stream_offset + j / 2), per-step scalars come from `sc` (host array of n_steps, free again on return).
 * The loss of every step is kept in a device ring read by iqlhip_read_loss_ring.
 * Replaces the offline loop body sample()->train() (algorithms/offline/iql.py:631-635).
 * A call = one set-up launch + replays of fixed chunk graphs (IQLHIP_GRAPH_STEPS = 64 steps, 16, 4, 2, 1) that chain on
 * the device; nothing is captured per value of n_steps.
 * flags: IQLHIP_TS_CONTINUE — the caller states that the replay rows have not been written since the previous
 * iqlhip_train_steps call; if that call ended where this one starts (same rows / size / batch_rows / seed,
 * stream_offset = its offset + n_steps * batch_rows / 2, an even n_steps, no other step entry point in between — the
 * library checks all of that) the rows its last forward staged for "the next step" ARE this call's step 0 and nothing
 * is gathered up front.  Results are identical with and without the flag. */
#define IQLHIP_TS_CONTINUE 1
int iqlhip_train_steps(iqlhip_ctx* ctx, const float* rows_dev, int64_t ld, int64_t size, int32_t batch_rows,
                       const iqlhip_step_scalars* sc, int32_t n_steps, uint64_t seed, uint64_t stream_offset,
                       int32_t flags, void* stream);

/* Capture, instantiate, upload AND rehearse (arenas saved and restored) every chunk graph iqlhip_train_steps replays for
 * this (buffer, batch_rows, inv_batch) — and, with an exchange attached, this exchange mode — so that no later
 * iqlhip_train_steps call pays for a capture or a first replay (bench.py calls it before its timed region).  The
 * rehearsal replays run on `stream` (pass the stream the later calls will use); synchronous. */
int iqlhip_train_steps_prepare(iqlhip_ctx* ctx, const float* rows_dev, int64_t ld, int32_t batch_rows, float inv_batch,
                               void* stream);

/* ---- data-parallel gradient exchange (the reference has no multi-device code) ----------------
 * One process per GPU.  With an exchange attached, iqlhip_step and iqlhip_train_steps run, per step:
 * forward, backward, flatten (this rank's flat gradient: n_params floats + 4 tail words with the loss
 * contributions; batch means divided by the GLOBAL row count: sc->inv_batch = 1 / (batch_rows * world)), the
 * exchange, and the fused Adam / Polyak update on the summed gradient — all on `stream`, inside the captured chunk
 * graphs too.  Two exchanges:
 *   RCCL  ncclAllReduce(sum, fp32) of the flat buffer, in place, in-stream (iqlhip_allreduce_init).
 *   P2P   every rank's flat buffers are mapped into every other rank (hipIpc); after a flag handshake in device
 *         memory the update kernel reads all ranks' buffers directly over xGMI and sums them in rank order
 *         (iqlhip_p2p_export + iqlhip_p2p_attach).  No collective library call per step. */
enum { IQLHIP_XCH_NONE = 0, IQLHIP_XCH_RCCL = 1, IQLHIP_XCH_P2P = 2 };
#define IQLHIP_UNIQUE_ID_BYTES 128  /* = NCCL_UNIQUE_ID_BYTES */
#define IQLHIP_IPC_HANDLE_BYTES 64  /* = sizeof(hipIpcMemHandle_t) */
/* ncclGetUniqueId: called by ONE rank; the caller ships the 128 bytes to the others (any channel). */
int iqlhip_comm_unique_id(void* id_out);
/* ncclCommInitRank on the context's device; collective over the `world` ranks.  Selects the RCCL exchange. */
int iqlhip_allreduce_init(iqlhip_ctx* ctx, const void* unique_id, int rank, int world);
/* P2P exchange, step 1: allocate this rank's exchange block (flags + two flat buffers) and export it
 * (hipIpcGetMemHandle) into handle_out[IQLHIP_IPC_HANDLE_BYTES]; the caller all-gathers the handles. */
int iqlhip_p2p_export(iqlhip_ctx* ctx, void* handle_out, int rank, int world);
/* step 2: map the peers' blocks (handles = world x IQLHIP_IPC_HANDLE_BYTES in rank order; the own slot is ignored).
 * Selects the P2P exchange.  timeout_ms bounds every in-stream wait for a peer (0 = 5000). */
int iqlhip_p2p_attach(iqlhip_ctx* ctx, const void* handles, int timeout_ms);
/* Switch between attached exchanges (IQLHIP_XCH_*); NONE detaches nothing, it only runs steps locally. */
int iqlhip_xch_select(iqlhip_ctx* ctx, int mode);
/* status[0] = mode in use, status[1] = first step at which a P2P wait timed out (0 = never), status[2] = steps
 * exchanged so far.  Synchronises `stream`. */
int iqlhip_xch_status(iqlhip_ctx* ctx, int64_t status[3], void* stream);
/* Forget a recorded P2P wait timeout (status[1] back to 0) once the caller has re-synchronised the replicas and
 * selected another exchange.  Until then iqlhip_read_losses / iqlhip_read_loss_ring / iqlhip_online_step — the entry
 * points that synchronise — return IQLHIP_EEXCHANGE; fully asynchronous callers poll iqlhip_xch_status. */
int iqlhip_xch_clear_status(iqlhip_ctx* ctx, void* stream);
/* Release communicator / peer mappings (also done by iqlhip_destroy). */
int iqlhip_xch_shutdown(iqlhip_ctx* ctx);

/* The three .item() calls of iql.py:491,509,535: synchronises `stream`. out = {value,q,actor}. */
int iqlhip_read_losses(iqlhip_ctx* ctx, float out[3], void* stream);
int iqlhip_read_loss_ring(iqlhip_ctx* ctx, float* out, int32_t n_steps, void* stream);

/* ---- replay buffer storage (packed rows [s | a | s' | r | d | pad]) ------ */
/* Row stride in floats for given dims (multiple of 4 floats = 16 B). */
int64_t iqlhip_row_stride(int32_t state_dim, int32_t action_dim);
/* ReplayBuffer.load_d4rl_dataset / add_transition (iql.py:153-169,180-196): write n
 * rows starting at row0 from five contiguous device arrays. */
int iqlhip_rows_write(float* rows_dev, int64_t ld, int32_t state_dim, int32_t action_dim, int64_t row0, int64_t n,
                      const float* s_dev, const float* a_dev, const float* r_dev, const float* ns_dev,
                      const float* d_dev, void* stream);
/* Synthetic D4RL-shaped rows written where they live (bench data of SURVEY §8d's distributions: obs / next_obs ~ N(0,1),
 * actions ~ U(-1,1) * 0.999, rewards ~ N(0,1) or the antmaze flavour {-1, 0}, dones ~ Bernoulli(p_done); Philox4x32-10
 * keyed by `seed`, counter = element number, so every rank of a data-parallel run fills identical rows without a
 * host-side generator or an upload).  Replaces nothing in the reference (its data come from d4rl.qlearning_dataset). */
int iqlhip_rows_fill_synth(float* rows_dev, int64_t ld, int32_t state_dim, int32_t action_dim, int64_t row0, int64_t n,
                           uint64_t seed, float p_done, int32_t antmaze_rewards, void* stream);
/* ReplayBuffer.sample's five advanced-index gathers (iql.py:173-177) in one launch.  n_rows = rows the buffer holds
 * (its capacity): the reference's indexing raises IndexError for an index outside the tensors; the entry points that
 * see the indices on the host return IQLHIP_EINDEX before anything is launched, the ones that take device indices never
 * dereference such an index (its output row is filled with NaN instead of faulting the GPU) — a caller that wants the
 * exception checks device indices itself, as the Python shim's ReplayBuffer.gather does. */
int iqlhip_rows_gather(const float* rows_dev, int64_t ld, int64_t n_rows, int32_t state_dim, int32_t action_dim,
                       const int64_t* idx_dev, int64_t n, float* s_dev, float* a_dev, float* r_dev, float* ns_dev,
                       float* d_dev, void* stream);
/* The same sample as whole packed rows: out[i] = rows[idx[i]] (one coalesced row copy per sample).  A batch whose
 * five pointers are the packed offsets of such a block (a = s + S, s' = s + S + A, r = s + 2S + A, d = r + 1, all
 * strides = iqlhip_row_stride, 16-byte aligned) is consumed IN PLACE by iqlhip_step / iqlhip_forward_backward. */
int iqlhip_rows_gather_packed(const float* rows_dev, int64_t ld, int64_t n_rows, const int64_t* idx_dev, int64_t n,
                              float* out_rows_dev, void* stream);
/* ... with the indices still in (pinned) host memory, as np.random.randint leaves them (iql.py:172): copies them to
 * idx_scratch_dev on `stream`, then gathers.  idx_host must stay untouched until that copy has run. */
int iqlhip_rows_gather_packed_h(const float* rows_dev, int64_t ld, int64_t n_rows, const int64_t* idx_host,
                                int64_t* idx_scratch_dev, int64_t n, float* out_rows_dev, void* stream);
/* ... or in ordinary host memory: the library stages them through its own pinned ring (event-guarded) — the whole
 * device side of ReplayBuffer.sample(batch_size) after the np.random.randint draw, in one call. */
int iqlhip_rows_sample_packed(const float* rows_dev, int64_t ld, int64_t n_rows, const int64_t* idx_host, int64_t n,
                              float* out_rows_dev, void* stream);
/* ---- dataset ingest on the device -------------------------------------------------------------
 * compute_mean_std (algorithms/finetune/iql.py:77-80): mean[c] = mean_r x[r][c], std[c] = sqrt(mean_r (x - mean)^2) + eps
 * over n rows of ncols columns (row stride ld floats; e.g. the state columns of packed replay rows: x = rows_dev,
 * ncols = state_dim).  Sums in float64, fixed order (deterministic); results are float32 like numpy's. */
int iqlhip_cols_mean_std(const float* x_dev, int64_t ld, int32_t ncols, int64_t n, float eps, float* mean_dev,
                         float* std_dev, void* stream);
/* normalize_states (:83-84) applied in place to the s and s' columns of n packed rows from row0:
 * x = (x - mean[c]) / std[c] in fp32 — bit-identical to numpy given the same mean / std. */
int iqlhip_rows_normalize(float* rows_dev, int64_t ld, int32_t state_dim, int32_t action_dim, int64_t row0, int64_t n,
                          const float* mean_dev, const float* std_dev, void* stream);
/* Device-side index draw used by iqlhip_train_steps, exposed for tests. */
int iqlhip_draw_indices(int64_t* idx_dev, int64_t n, int64_t size, uint64_t seed, uint64_t offset, void* stream);

/* ---- policy inference ---------------------------------------------------- */
/* GaussianPolicy.act (algorithms/finetune/iql.py:371-379), DeterministicPolicy.act (:404-413) and the batched policy
 * forward of evaluation loops (eval_actor, jsrl_w_iql.py:62-179):
 *   actions[r] = clamp(max_action * (tanh(MLP_pi(states[r])) + exp(clamp(log_std)) * noise[r]), -max_action, max_action)
 * noise_dev == NULL gives the mean (eval mode, or the deterministic policy); with a Gaussian policy in training mode
 * the caller passes standard-normal noise [rows][action_dim] (dist.sample() of iql.py:376).  Dropout is NOT applied
 * (eval-mode forward).  rows <= max(max_batch, IQLHIP_ACT_ROWS) per call.  Uses the bound parameter arena;
 * asynchronous on `stream`.  states / noise / actions may be device memory or host-mapped (pinned) memory. */
#define IQLHIP_ACT_ROWS 4096
int iqlhip_actor_forward(iqlhip_ctx* ctx, const float* states_dev, int64_t ld_s, int32_t rows, const float* noise_dev,
                         int64_t ld_noise, float max_action, float* actions_dev, int64_t ld_a, void* stream);

/* The training-mode act() of a Gaussian policy with the N(0,1) draw of dist.sample() (iql.py:376) made on the device
 * (Philox4x32-10 keyed by `seed` != 0, counter = (element, call number kept by the context), Box-Muller). */
int iqlhip_actor_sample(iqlhip_ctx* ctx, const float* states_dev, int64_t ld_s, int32_t rows, uint64_t seed,
                        float max_action, float* actions_dev, int64_t ld_a, void* stream);

/* Block the host until everything queued on `stream` has finished (hipStreamSynchronize): the completion point of
 * iqlhip_actor_forward when its buffers are host-mapped, i.e. the `.cpu()` of the reference's act() (iql.py:379). */
int iqlhip_stream_synchronize(void* stream);

/* ---- introspection (tests, profiling) ----------------------------------- */
/* Copy a named library-owned scratch array to host (synchronous).  Names:
 * "h0","h1" (activations [4][max_batch][256]), "heads" (partial head sums),
 * "grads" (flat summed gradient, n_params), "loss_parts". */
int iqlhip_debug_read(iqlhip_ctx* ctx, const char* name, float* host_out, int64_t max_floats, int64_t* n_out,
                      void* stream);
/* Micro-benchmark hook: `repeat` back-to-back launches of one kernel of the step (0 fwd, 1 bwd,
 * 2 update with zero step size, 3 all three); average microseconds per launch.  Synchronous. */
int iqlhip_debug_time_kernel(iqlhip_ctx* ctx, const iqlhip_batch* batch, int which, int repeat, float* avg_us,
                             void* stream);
/* Diagnostic: queue a flag kernel on `stream` and spin on its host-mapped word (no synchronise call): microseconds
 * until the host sees the stream drained. */
int iqlhip_debug_drain_spin(iqlhip_ctx* ctx, void* stream, double* spin_us);
/* Average device time (microseconds) of the kernels of the last iqlhip_step /
 * train_steps call measured with hipEvents on `stream`; 0 when timing is off. */
int iqlhip_set_timing(iqlhip_ctx* ctx, int enabled);
int iqlhip_get_timing(iqlhip_ctx* ctx, float out_us[4]); /* fwd, bwd, update, total */

#ifdef __cplusplus
}
#endif
#endif /* IQLHIP_H */
