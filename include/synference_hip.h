/*
 * synference_hip.h -- C ABI of the MI355X (gfx950) amortised-posterior flow engine.
 *
 * Drop-in boundary for ONE path of synthesizer-project/synference: the conditional
 * normalizing-flow density estimator (MAF / NSF) that the reference reaches through
 *     ili.utils.load_nde_sbi(...)            ref: src/synference/sbi_runner.py:5123-5146
 *     estimator_builder(batch_x=, batch_theta=)   ref: src/synference/custom_runner.py:320-326
 * and evaluates through
 *     posterior.sample((S,), x=)             ref: src/synference/sbi_runner.py:6438-6442
 *     posterior.log_prob(x=, theta=)         ref: src/synference/sbi_runner.py:7193-7196
 *     loss.backward(); clip; optimizer.step()  ref: src/synference/custom_runner.py:585-618
 *
 * The reference is pure Python (no FFI of its own), so these entry points are what a
 * ctypes binding for that path would bind; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain pointers and sizes only; every data pointer is a DEVICE pointer to
 *     contiguous row-major float32 unless the comment says "host".
 *   - the caller owns every buffer it passes; the library owns the handle, its packed
 *     weight image and its scratch.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are
 *     asynchronous on it unless stated.
 *   - return 0 on success, a negative sf_status otherwise; sf_last_error() returns the
 *     thread-local message.
 *   - one handle per (process, device); a handle is not thread-safe.
 *
 * Logical parameter layout (the flat vector of sf_flow_set_params / sf_flow_loss_grad),
 * per transform t = 0..T-1, torch nn.Linear convention W[out][in], row-major:
 *   MAF: W0[H,D] b0[H] Wc[H,C] bc[H] {Wk[H,H] bk[H]} x NB  Wf[2D,H] bf[2D]
 *   NSF: Win[H,d_id+C] bin[H] {Wg[H,C] bg[H] W1[H,H] b1[H] W2[H,H] b2[H]} x NB
 *        Wout[d_tr*(3K-1),H] bout[d_tr*(3K-1)]
 *        and, when D > 1: lower[D(D-1)/2] upper[D(D-1)/2] udiag[D] lubias[D]
 *        (tril / triu index order, row-major, as numpy tril_indices(D,-1) / triu_indices(D,1))
 *   NSF_AR: W0[H,D+C] b0[H] W1[H,H] b1[H] W2[D*(3K-1),H] b2[D*(3K-1)]   (NB = 2 hidden layers; row d*(3K-1)+j of the
 *        head: j in [0,K) widths, [K,2K) heights, [2K,3K-1) derivatives of dimension d; tail_bound = zuko's `bound` [5])
 * MADE masks are implied by (D, H) and applied inside the library (masked entries of
 * the flat vector are ignored on input and receive zero gradient).
 */
#ifndef SYNFERENCE_HIP_H
#define SYNFERENCE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sf_flow sf_flow;
typedef struct sf_opt sf_opt;

enum sf_status {
  SF_OK = 0,
  SF_ERR_INVALID = -1,     /* bad argument / unsupported shape */
  SF_ERR_HIP = -2,         /* a HIP runtime call failed */
  SF_ERR_NO_DEVICE = -3,   /* no gfx950 device visible */
  SF_ERR_STATE = -4        /* e.g. parameters not set */
};

/* SF_NSF_AR: the autoregressive NSF of the reference's second backend (backend="lampe" -> zuko.flows.NSF;
 * ref: src/synference/sbi_runner.py:5123-5125): masked hyper-network + zuko's monotonic rational-quadratic spline */
/* SF_MAF_AR: the MAF of the same backend (backend="lampe", model "maf" -> zuko.flows.MAF, sbi_runner.py:5123-5125): the same masked
 * hyper-network with two outputs per dimension and zuko's MonotonicAffineTransform as the univariate map
 * (y = x exp(s / (1 + |s / log slope|)) + shift); K, tail_bound and the spline fields are ignored. */
enum sf_kind { SF_MAF = 0, SF_NSF = 1, SF_NSF_AR = 2, SF_MAF_AR = 3 };

/* Static description of one flow.  Pointer members are HOST arrays read during
 * sf_flow_create only.  Defaults of the upstream stack (sbi/nflows) in brackets. */
typedef struct sf_flow_desc {
  int32_t kind;            /* sf_kind */
  int32_t D;               /* theta dimension, 1..16 */
  int32_t C;               /* context width seen by the transforms, 1..512 */
  int32_t H;               /* hidden_features [50], 1..128 */
  int32_t T;               /* num_transforms [5] */
  int32_t K;               /* num_bins (NSF) [10], 2..16 */
  int32_t NB;              /* num_blocks [2], 1..4 */
  int32_t scale_fn;        /* MAF scale: 0 = softplus(a)+eps [nflows>=0.14], 1 = sigmoid(a+2)+eps */
  int32_t hidden_bf16;     /* 0 = fp32 everywhere [default]; 1 = the hidden H x H layers of the INFERENCE kernels
                              (log_prob / inverse / sampler) use bf16 MFMA operands with fp32 accumulation
                              (BASELINE configs[4]); everything else, and all training, stays fp32 */
  float tail_bound;        /* [3.0] */
  float min_bin_width;     /* [1e-3] */
  float min_bin_height;    /* [1e-3] */
  float min_derivative;    /* [1e-3] */
  float maf_eps;           /* [1e-3] */
  float lu_eps;            /* [1e-3] */
  const float* theta_mean; /* host [D]  z-score buffers (sbi standardizing_transform) */
  const float* theta_std;  /* host [D] */
  const float* x_mean;     /* host [C]  (sbi standardizing_net) */
  const float* x_std;      /* host [C] */
  const int32_t* perms;    /* host [T*D] MAF RandomPermutation buffers, NULL = identity */
  float ar_slope;          /* SF_NSF_AR: slope of zuko's MonotonicRQSTransform [1e-3]; ignored by the other kinds */
} sf_flow_desc;

/* ---- lifetime ---------------------------------------------------------------------- */
/* Builds the layer tables and allocates the packed weight image on the current device.
 * Replaces: the nn.Module returned by build_fn (custom_runner.py:326). */
int sf_flow_create(const sf_flow_desc* desc, sf_flow** out);
void sf_flow_destroy(sf_flow* f);
int64_t sf_flow_num_params(const sf_flow* f);
/* floats in the MFMA-tiled weight image (for tests / diagnostics) */
int64_t sf_flow_packed_size(const sf_flow* f);

/* ---- parameters -------------------------------------------------------------------- */
/* flat: n = sf_flow_num_params floats in the logical layout; is_device selects host or
 * device source.  Re-tiles them into the MFMA operand image (a device gather kernel).
 * Replaces: estimator.load_state_dict (custom_runner.py:563, 709). */
int sf_flow_set_params(sf_flow* f, const float* flat, int64_t n, int is_device, void* stream);

/* Copies the logical vector last given to sf_flow_set_params back out (host or device destination).
 * SF_ERR_STATE after sf_flow_loss_grad*: during training the caller's vector is the master copy.
 * Replaces: estimator.state_dict() (custom_runner.py:658). */
int sf_flow_get_params(sf_flow* f, float* flat, int64_t n, int is_device, void* stream);

/* Host-only helpers (no GPU needed; used by the CPU test-suite):
 * src1/src2[i] = logical index feeding packed float i (or -1); packed = sum of both. */
int sf_flow_pack_table(const sf_flow* f, int32_t* src1, int32_t* src2, int64_t n_packed);
/* the same for the 16-row image of the incremental MAF sampler (size 0 when the flow has none) */
int64_t sf_flow_packed16_size(const sf_flow* f);
int sf_flow_pack_table16(const sf_flow* f, int32_t* src1, int32_t* src2, int64_t n_packed16);
/* split-bf16 image of the hidden blocks of the persistent 16-row sampler: src[i] = logical index | (part << 30) for
 * bf16 element i (part 0: hi = bf16(w), part 1: lo = bf16(w - hi)), -1 = zero; size 0 when the flow has none */
int64_t sf_flow_packed16b_size(const sf_flow* f);
int sf_flow_pack_table16b(const sf_flow* f, int32_t* src, int64_t n);
/* cooperative 16-row training image (MAF, num_blocks 2, D <= 8; csrc/sf_layout.h SfTrcDev): sizes (0 when the flow has
 * none), gather table, logical parameter -> gradient-partial index, and the descriptor as int32 words in declaration order */
int64_t sf_flow_trainc_size(const sf_flow* f);
int64_t sf_flow_trainc_grad_size(const sf_flow* f);
int sf_flow_trainc_table(const sf_flow* f, int32_t* src1, int32_t* src2, int64_t n, int32_t* gdst, int64_t n_params,
                         int32_t* desc /*[64]*/, float* cst, int64_t n_cst);
int64_t sf_flow_cst_size(const sf_flow* f);
/* which kernel sf_flow_loss_grad* runs for a batch of B rows: 0 = one producer wave per 32-sample tile (k_maf_train /
 * k_nsf_train), 1 / 2 = the cooperative 16-row MAF kernel with 4-wave / 8-wave workgroups (k_maf_trainc), 3 = the
 * cooperative 16-row NSF kernel (k_nsf_trainc) */
int sf_flow_train_path(const sf_flow* f, int64_t B, int want_dctx);
/* byte-for-byte description of the packed image for diagnostics (JSON, NUL-terminated) */
int sf_flow_describe(const sf_flow* f, char* buf, size_t buflen);

/* ---- density direction --------------------------------------------------------------
 * out[b] = log N(z;0,I) + sum log|det J|  for theta[b,:], x[b,:]   (raw estimator density)
 * Replaces: flow.log_prob(theta, context=x) (custom_runner.py:604, 646). */
int sf_flow_log_prob(sf_flow* f, const float* theta /*[B,D]*/, const float* x /*[B,C]*/,
                     int64_t B, float* out /*[B]*/, void* stream);

/* ---- sampling direction -------------------------------------------------------------
 * Parity hook: theta = inverse(z | x), logdet = log|det d theta / d z|  (may be NULL). */
int sf_flow_inverse_from_noise(sf_flow* f, const float* z /*[B,D]*/, const float* x /*[B,C]*/,
                               int64_t B, float* theta /*[B,D]*/, float* logdet /*[B]*/,
                               void* stream);
/* Parity hook of the SAMPLER's arithmetic: the same inverse from given noise, evaluated by the pass functions the
 * persistent sampler (sf_flow_sample) runs in the current mode (sf_set_sampler_fp32).  Returns which arithmetic that was
 * (negative: error):
 *   0  split-bf16 x3 hidden blocks with fp32 accumulation (the opt-in mode of a MAF with H <= 64; the default of the coupling NSF)
 *   1  the generic all-fp32 path (the result is sf_flow_inverse_from_noise's)
 *   2  the 16-row fp32 kernels, both layers of block 0 as they are stored (SF_FUSE=0)
 *   3  the 16-row fp32 kernels with the first block layer folded into the input layer (W' = W1 W0; the default of a MAF)
 * Replaces nothing in the reference: test surface of posterior.sample's numerics (sbi_runner.py:6442). */
int sf_flow_inverse_from_noise_sampler(sf_flow* f, const float* z /*[B,D]*/, const float* x /*[B,C]*/, int64_t B,
                                       float* theta /*[B,D]*/, void* stream);
/* Arithmetic of the persistent sampler's hidden blocks, process-wide (also: environment SF_SAMPLER_FP32):
 *    1  fp32 everywhere;  0  split-bf16 x3 hidden blocks where the flow has them;
 *   -1  the per-kind default (round 5): fp32 for a MAF -- configs[1] says fp32, and log p of a draw moved by up to 1.3e-3
 *       under the split -- , split for the coupling NSF.  Returns 0. */
int sf_set_sampler_fp32(int on);

/* One rejection round over a list of output slots (slot = g*S + p).  For every listed slot the
 * attempts attempt .. attempt+attempts_per_slot-1 are evaluated together (attempts_per_slot a power
 * of two <= 32) and the LOWEST accepted one is kept, so the result is the same as trying them one
 * after the other:
 *   noise = Philox4x32-10(seed, stream_id; slot, attempt)  ->  theta = inverse(noise | x[g])
 *   accepted (finite and lo <= theta <= hi on every dim; lo/hi NULL = accept all finite):
 *        theta written to out[slot*D ...]
 *   rejected: slot appended to rejected[] (order unspecified), *n_rejected incremented.
 * slots == NULL means the dense list slot = slot_base + i, i < n_slots.
 * n_drawn (may be NULL) [M] int32: += attempts consumed (up to and including the accepted one) for
 * galaxy g, in rounds with attempt > 0 (round 0 is accounted for by the caller: S per galaxy).
 * Replaces: DirectPosterior.sample -> accept_reject_sample (sbi_runner.py:6442; box
 * predicate custom_runner.py:982-987). */
int sf_flow_sample_round(sf_flow* f, const float* x /*[M,C]*/, int64_t S,
                         const uint32_t* slots, int64_t slot_base, int64_t n_slots,
                         uint32_t attempt, int32_t attempts_per_slot, uint64_t seed, uint32_t stream_id,
                         const float* lo /*[D]*/, const float* hi /*[D]*/,
                         float* out /*[M*S, D]*/, uint32_t* rejected, uint32_t* n_rejected,
                         int32_t* n_drawn, void* stream);

/* Optional, for callers that drive the rounds themselves: evaluate everything that depends on a context row
 * alone (the conditioner's context products; in the reference they are recomputed for every one of the S draws
 * of a galaxy inside DirectPosterior.sample, sbi_runner.py:6442) once per row of x[0..M), into a table owned by
 * the handle.  Later sf_flow_sample_round calls that pass the SAME x pointer read the table instead; x must not
 * change until sf_flow_release_context (or the next prepare / set_params / loss_grad, which drop the table).
 * Purely an optimisation: same noise stream, draws equal up to fp32 summation order.  Tables above SF_CTAB_MAX_MB (default
 * 4096) are not built.  sf_flow_sample and sf_flow_acceptance do this internally. */
int sf_flow_prepare_context(sf_flow* f, const float* x /*[M,C]*/, int64_t M, void* stream);
int sf_flow_release_context(sf_flow* f);

/* Whole sampler.  ONE persistent launch works the catalogue's M*S output slots -- first attempts and the retries of
 * rejected slots are scheduled on the device (no host round trip per rejection round) -- up to 1024 attempts per slot
 * (all kernels since round 3).  The few slots that are still empty then (their galaxy accepts less than about one draw
 * in a thousand) are continued chip-wide: a "find" launch evaluates a whole range of attempts of every open slot side by
 * side and records the lowest accepted one, a "resolve" launch re-evaluates exactly that attempt and writes the draw.
 * Every slot keeps the LOWEST accepted attempt of its Philox stream (slot, attempt): the draws do not depend on how the
 * work was scheduled.
 *   max_attempts > 0 : hard ceiling; a slot that used max_attempts attempts becomes a NaN row
 *                      (failure convention of ref: sbi_runner.py:6458-6460).
 *   max_attempts <= 0: no ceiling, like [UPSTREAM] accept_reject_sample, which keeps drawing until S draws are kept:
 *                      a slot is retried for as long as its galaxy still gets draws accepted.  Where an attempt window
 *                      ends (1024, 16384, 262144, ...), and once the galaxy's open slots have used 1e5 attempts since
 *                      the last look, the open slots of a galaxy that got NOT ONE draw accepted since then (counted
 *                      from its 64th attempt on) become NaN rows.
 * The host synchronises the stream once per launch pair (one pinned read-back).  n_drawn [M] may be NULL: attempts
 * consumed per galaxy (int32: pinned at 2^31 - 1 once S x attempts would pass it).  *n_unfilled (host): number of NaN slots. */
int sf_flow_sample(sf_flow* f, const float* x /*[M,C]*/, int64_t M, int64_t S,
                   const float* lo, const float* hi, uint64_t seed, int32_t max_attempts,
                   float* out /*[M,S,D]*/, int32_t* n_drawn /*[M]*/, int64_t* n_unfilled /*host*/,
                   void* stream);

/* Row offset of the NEXT sampling calls on this handle (sf_flow_sample, sf_flow_sample_slots, sf_flow_sample_round,
 * sf_flow_acceptance): the rows passed are rows [row_offset, row_offset + M) of a larger catalogue.  Only the random
 * streams see it (Philox counter = (row_offset + g) * S + p): a chunk of a catalogue, or one rank's contiguous shard of it
 * (rows sharded over GPUs, no collective), then draws exactly what those rows draw in ONE call over the whole
 * catalogue with the same seed.  Stays in force until changed; 0 after sf_flow_create.
 * Replaces: nothing in the reference (its per-galaxy loop, sbi_runner.py:6438-6442, has no batching to be independent of). */
int sf_flow_set_sample_row_offset(sf_flow* f, int64_t row_offset);
/* Output dtype of the next sf_flow_sample / sf_flow_sample_slots calls on this handle: on != 0 -- `out` is a DOUBLE array
 * [M,S,D] (passed through the float* parameter) and every accepted draw is widened fp32 -> float64 in its store (exact).
 * float64 is the container dtype of the reference's sample_posterior (sbi_runner.py:6436: np.zeros((N, S, D))); `out` may be
 * PINNED HOST memory mapped into the device's address space (hipHostMalloc): the draws then cross PCIe while the sampler
 * runs and the reference's host array is complete when the stream is -- no D2H copy and no widening pass afterwards
 * (measured on the cfg2 catalogue: +0.3 ms on a 2.6 ms launch against 1.0 ms for copy + widening).  Not offered for the
 * one-parameter / autoregressive NSF (fp32 + sf_copy_to_host_f64 there). */
int sf_flow_set_sample_output_f64(sf_flow* f, int on);
/* Wall-clock ceiling of later sf_flow_sample / sf_flow_sample_slots calls on this handle (seconds; <= 0 = none, the
 * default): once it is exceeded no further attempt window is opened and the slots still empty become NaN rows.
 * Replaces: the per-object timeout of sample_posterior (timeout_seconds_per_test, ref: sbi_runner.py:6358, 6443-6452). */
int sf_flow_set_sample_time_limit(sf_flow* f, double seconds);

/* The same over an explicit list of output slots (slot = g*S + p; DEVICE uint32 [n_slots], each slot once): only those
 * slots of out are written.  This is what an ensemble member runs on its share of every row's draws
 * ([UPSTREAM] sbi EnsemblePosterior.sample, built at ref: custom_runner.py:278-283). */
int sf_flow_sample_slots(sf_flow* f, const float* x /*[M,C]*/, int64_t M, int64_t S,
                         const uint32_t* slots, int64_t n_slots, const float* lo, const float* hi, uint64_t seed,
                         int32_t max_attempts, float* out /*[M,S,D]*/, int64_t* n_unfilled /*host*/, void* stream);

/* Figures of the last sf_flow_sample / sf_flow_sample_slots call (host [4]): duration in ms of its first persistent
 * launch (HIP events on the call's stream; the only launch unless some slot needed more than 1024 attempts), number of
 * launches, slots whose FIRST attempt was rejected, flow evaluations over all launches. */
int sf_flow_sample_stats(const sf_flow* f, float* stats4);

/* Accepted fraction of n unconstrained draws per row (stream_id 1): count[g] of n.
 * Replaces: DirectPosterior.leakage_correction (custom_runner.py:466-473). */
int sf_flow_acceptance(sf_flow* f, const float* x /*[M,C]*/, int64_t M, int64_t n,
                       const float* lo, const float* hi, uint64_t seed,
                       int32_t* count /*[M]*/, void* stream);

/* ---- training -----------------------------------------------------------------------
 * Forward + backward of  loss_b = -log_prob(theta_b | x_b)  with the parameters given in
 * `flat` (device, logical layout):
 *   loss[b] = loss_b                       (may be NULL)
 *   grad[i] = sum_b w * d loss_b / d flat[i], w = grad_scale (pass 1/B for the mean loss)
 * grad is overwritten (not accumulated).
 * Replaces: train_losses = -estimator.log_prob(...); mean().backward()
 * (custom_runner.py:604-610). */
int sf_flow_loss_grad(sf_flow* f, const float* flat /*[P]*/, const float* theta, const float* x,
                      int64_t B, float grad_scale, float* loss /*[B]*/, float* grad /*[P]*/,
                      void* stream);

/* Same, with the mini-batch gather fused into the kernel: batch row b reads theta[rows[b],:] / x[rows[b],:] of the
 * training arrays (rows: DEVICE int64 [B]); loss / dctx stay batch-indexed.  loss_sum (may be NULL): DEVICE double,
 * += sum_b (-log p_b), the epoch loss accumulator of custom_runner.py:608-611 without a per-step reduction. */
int sf_flow_loss_grad_rows(sf_flow* f, const float* flat, const float* theta /*[N,D]*/, const float* x /*[N,C]*/,
                           const int64_t* rows /*[B]*/, int64_t B, float grad_scale, const float* weights,
                           float* loss, double* loss_sum, float* grad, float* dctx, void* stream);


/* Same with per-sample weights: grad[i] = sum_b grad_scale * weights[b] * d loss_b / d flat[i]
 * (weights NULL = all ones).  This is the vector-Jacobian product torch.autograd needs for an
 * arbitrary reduction of the per-sample losses.
 * dctx (may be NULL) [B,C]: overwritten with grad_scale * weights[b] * d loss_b / d x[b,:] -- the
 * gradient reaching the context, i.e. what an embedding net in front of the flow back-propagates
 * (embedding_net kwarg, ref: sbi_runner.py:4432, custom_runner.py:321). */
int sf_flow_loss_grad_weighted(sf_flow* f, const float* flat, const float* theta, const float* x,
                               int64_t B, float grad_scale, const float* weights /*[B]*/,
                               float* loss, float* grad, float* dctx /*[B,C]*/, void* stream);

/* Measurement hook (bench.py's roofline_train): with profiling on, every sf_flow_loss_grad* call brackets its
 * forward+backward flow kernel with HIP events on the call's stream; sf_flow_train_stats waits for the last such call
 * and returns that kernel's duration in ms.  Off by default (two event records per step are not free at batch 64). */
int sf_flow_set_profiling(sf_flow* f, int on);
int sf_flow_train_stats(sf_flow* f, float* kernel_ms /*host*/);

/* Fused global-norm clip + Adam / AdamW step on flat vectors.
 * Replaces: clip_grad_norm_(max_norm) + optimizer.step() (custom_runner.py:613-618). */
typedef struct sf_adam_desc {
  float lr, beta1, beta2, eps, weight_decay; /* torch defaults 1e-3, .9, .999, 1e-8, 0 (AdamW .01) */
  int32_t decoupled;                           /* 0 = Adam (L2 in grad), 1 = AdamW */
} sf_adam_desc;
int sf_opt_create(int64_t n, const sf_adam_desc* d, sf_opt** out);
void sf_opt_destroy(sf_opt* o);
/* max_norm <= 0 disables clipping; *grad_norm_out (device float, may be NULL) gets the
 * pre-clip global L2 norm. */
int sf_adam_step(sf_opt* o, float* params, const float* grad, float max_norm,
                 float* grad_norm_out, void* stream);
/* Stateless form: the caller owns exp_avg / exp_avg_sq [n] (device) and the step counter (the
 * value AFTER this update, >= 1).  scratch: 2 device floats; scratch[1] receives the pre-clip
 * global L2 norm.  This is the form the Python runner uses so that optimizer state lives in
 * torch tensors and checkpoints like the reference's (custom_runner.py:693-704). */
int sf_adam_apply(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  const sf_adam_desc* d, int64_t step, float max_norm, float* scratch, void* stream);

/* One training epoch on a single device without returning to the host between steps
 * (ref: the batch loop of custom_runner.py:585-618): for b < n_batches:
 *   rows = order[b*batch .. (b+1)*batch);  grad = d/dflat sum_rows grad_scale * (-log p);  clip + Adam(W) step
 *   number step0 + b + 1 on flat (state exp_avg / exp_avg_sq, scratch [2] as in sf_adam_apply).
 * order: DEVICE int64 [n_batches*batch] (the epoch's shuffled training rows); grad: DEVICE scratch [P]. */
int sf_flow_train_epoch(sf_flow* f, float* flat, const float* theta, const float* x, const int64_t* order,
                        int64_t n_batches, int64_t batch, float grad_scale, float* exp_avg, float* exp_avg_sq,
                        const sf_adam_desc* d, int64_t step0, float max_norm, float* scratch /*[2]*/,
                        float* grad /*[P]*/, double* loss_sum, void* stream);

/* ---- data-parallel training: the gradient exchange --------------------------------
 * One process per GPU; every rank holds the same parameters and optimiser state and its own shard of the training rows.
 * Per optimiser step ONE sum all-reduce of the flat fp32 gradient (P floats: 129 KB for the cfg1 MAF) over RCCL -- xGMI
 * inside a node -- on the caller's stream, between the gradient gather and clip + Adam, so that the clip norm and the
 * update are those of the GLOBAL batch on every rank (grad_scale = 1 / (batch x ranks)).
 * Replaces nothing in the reference, which trains on CPU threads only (examples/sbi/slurm/train_final_model.slurm:26): the
 * call sits between `loss.backward()` and `optimizer.step()` of custom_runner.py:585-618.
 * RCCL is bound at run time (dlopen): sf_comm_set_library(path) before the first sf_comm_* call (or the environment variable
 * SF_RCCL_LIB) names the library -- a host that runs PyTorch passes torch's own librccl.so so that the process holds ONE
 * RCCL --, otherwise an already loaded librccl is used, then the loader's search path, then /opt/rocm/lib.
 * Bootstrap as in NCCL: rank 0 calls sf_comm_unique_id, the host ships the SF_COMM_ID_BYTES to every rank by its own means
 * (the torch.distributed store, MPI, a file), every rank calls sf_comm_create (collective) with its HIP device current. */
#define SF_COMM_ID_BYTES 128
typedef struct sf_comm sf_comm;
int sf_comm_set_library(const char* path);
int sf_comm_library(char* path_out, int64_t cap, int* version_out);    /* what was bound (loads it if necessary) */
int sf_comm_unique_id(void* id_out, int64_t bytes);
int sf_comm_create(const void* id, int64_t bytes, int nranks, int rank, sf_comm** out);
void sf_comm_destroy(sf_comm* c);
int sf_comm_info(const sf_comm* c, int* nranks, int* rank);
int sf_comm_all_reduce_sum(sf_comm* c, float* buf /*[n], in place*/, int64_t n, void* stream);
/* sf_flow_train_epoch with the exchange inside: per step  rows -> grad (this rank's shard, grad_scale = 1 / (batch x ranks))
 * -> all-reduce(SUM) over `comm` -> clip + Adam(W), all on `stream`, no host round trip per step.  `order` holds THIS rank's
 * rows; every rank must call it with the same n_batches.  loss_sum stays rank-local (the host reduces it once per epoch with
 * the validation sums, custom_runner.py:655-660).  comm == NULL is sf_flow_train_epoch. */
int sf_flow_train_epoch_dp(sf_flow* f, float* flat, const float* theta, const float* x, const int64_t* order,
                           int64_t n_batches, int64_t batch, float grad_scale, float* exp_avg, float* exp_avg_sq,
                           const sf_adam_desc* d, int64_t step0, float max_norm, float* scratch /*[2]*/,
                           float* grad /*[P]*/, double* loss_sum, sf_comm* comm, void* stream);
/* optimizer state access for checkpoints (custom_runner.py:693-704): exp_avg, exp_avg_sq
 * device pointers [n] and the step counter. */
int sf_opt_state(sf_opt* o, float** exp_avg, float** exp_avg_sq, int64_t** step_host);

/* ---- embedding MLP (context path) ---------------------------------------------------------
 * The optional fully connected embedding net in front of the flow ([UPSTREAM] ili FCN(n_hidden,
 * act_fn="SiLU"): Linear -> act -> ... -> Linear, no activation after the last layer; mentioned at
 * ref: examples/sbi/scripts/train_spectral_model.py:316-317, passed as the embedding_net kwarg,
 * ref: sbi_runner.py:4432).  Flat parameter layout: per layer W[out,in] row-major then b[out]. */
typedef struct sf_mlp sf_mlp;
enum sf_act { SF_ACT_SILU = 0, SF_ACT_RELU = 1, SF_ACT_TANH = 2 };
typedef struct sf_mlp_desc {
  int32_t n_in;        /* input width, 1..512 */
  int32_t n_layers;    /* 1..4 */
  int32_t widths[4];   /* output width of each layer, 1..128 */
  int32_t act;         /* sf_act */
  const float* x_mean; /* host [n_in] or NULL: inputs are standardised as (x - mean)/std in-kernel */
  const float* x_std;  /* host [n_in] or NULL */
} sf_mlp_desc;
int sf_mlp_create(const sf_mlp_desc* d, sf_mlp** out);
void sf_mlp_destroy(sf_mlp* m);
int64_t sf_mlp_num_params(const sf_mlp* m);
/* out[b,:] = MLP(x[b,:]) with the parameters in `flat` (device) */
int sf_mlp_forward(sf_mlp* m, const float* flat, const float* x /*[B,n_in]*/, int64_t B,
                   float* out /*[B,n_out]*/, void* stream);
/* grad[i] = sum_b dout[b,:] . d out[b,:] / d flat[i]   (overwritten) */
int sf_mlp_backward(sf_mlp* m, const float* flat, const float* x, const float* dout /*[B,n_out]*/,
                    int64_t B, float* grad /*[P]*/, void* stream);

/* ---- posterior summaries on the device ---------------------------------------------------
 * out[g,d,k] = quantile q[k] (numpy 'linear' rule) of samples[g,:,d], NaN draws ignored (all NaN -> NaN).
 * Replaces np.quantile(samples_i, quantiles, axis=1) of fit_catalogue (ref: sbi_runner.py:3270-3282)
 * without moving the (N,S,D) draws to the host.  q: DEVICE [Q]; 1 <= S <= 8192. */
int sf_quantiles(const float* samples /*[N,S,D]*/, int64_t N, int64_t S, int32_t D,
                 const float* q /*[Q]*/, int32_t Q, float* out /*[N,D,Q]*/, void* stream);

/* ---- feature transform on the device ------------------------------------------------------
 * mag = -2.5 log10(flux_nJy / 1000) + 23.9 ; negative flux -> mag_limit ; mag > mag_limit -> mag_limit ; NaN flux -> NaN
 * optional: mag_err = 2.5 err / (ln 10 flux) exactly as the reference computes it (inf for flux == 0, negative for
 * negative flux: the reference does not special-case them either).  All buffers [n] device, 16-byte aligned; err / mag_err may be NULL.
 * Replaces the numpy pass at ref: sbi_runner.py:1698-1716 (AB branch) and 1927-1932 (faint limit). */
int sf_flux_to_abmag(const float* flux_njy, const float* err_njy, int64_t n, float mag_limit,
                     float* mag, float* mag_err, void* stream);

/* asinh magnitudes with per-band softening f_b [C] (nJy, device): flux / err [N,C] row-major in nJy.
 *   mag = -2.5 log10(e) (asinh(f / 2 f_b) + ln(f_b / 3631 Jy)),  mag_err = 2.5 log10(e) err / sqrt(f^2 + (2 f_b)^2)
 * Replaces ref: utils.py:647-704 (f_jy_to_asinh, f_jy_err_to_asinh) as used at sbi_runner.py:1718-1731. */
int sf_flux_to_asinh(const float* flux_njy, const float* err_njy, int64_t N, int32_t C, const float* f_b_njy /*[C]*/,
                     float* mag, float* mag_err, void* stream);

/* Depth-noise scatter of library photometry: out[(i*n_scatters + s), c] = flux[i,c] + sigma_ic * N(0,1),
 * sigma_ic = max(sigma[c], |flux[i,c]| * min_flux_pc_error / 100); err_out (may be NULL) receives sigma_ic.
 * sigma [n_sigma_rows, C] device = depth / depth_sigma in the units of flux; n_sigma_rows = 1 (one depth per band)
 * or n_scatters (a depth set chosen per band and scatter copy: the reference's 2-D depths, the choice is the caller's).
 * Noise from the Philox stream (seed, stream 2).  Replaces ref: sbi_runner.py:580-691 (_apply_depths). */
int sf_scatter_depths(const float* flux /*[N,C]*/, int64_t N, int32_t C, const float* sigma, int32_t n_sigma_rows,
                      int32_t n_scatters, float min_flux_pc_error, uint64_t seed,
                      float* out /*[N*n_scatters,C]*/, float* err_out, void* stream);

/* PIT ranks of the truths among the posterior draws: out[g,d] = #{draws < truth} / #{finite draws}.
 * Replaces the host pass at ref: sbi_runner.py:7153-7158. */
int sf_pit_ranks(const float* samples /*[N,S,D]*/, const float* truth /*[N,D]*/, int64_t N, int64_t S, int32_t D,
                 float* out /*[N,D]*/, void* stream);

/* ---- hand-over to the host ----------------------------------------------------------------
 * host_dst[i] = (double) dev_src[i], i < n: the draws of a catalogue call leave HBM as fp32 in pieces through a ring of pinned
 * staging buffers on a private copy stream and are widened into the caller's float64 array (any host memory, not necessarily
 * pinned) by a pool of host threads with streaming stores, copy and widening overlapped.  Work queued on `stream` before the
 * call is waited for; blocking; one call at a time per process.  SF_HOSTIO_PIECE_MB [4], SF_HOSTIO_THREADS [min(8, cgroup
 * CPU quota)].
 * Replaces: `samples[i] = posterior.sample(...).detach().cpu().numpy()` into the float64 array (sbi_runner.py:6436-6457). */
int sf_copy_to_host_f64(const float* dev_src /*device [n]*/, double* host_dst /*host [n]*/, int64_t n, void* stream);


/* ---- misc --------------------------------------------------------------------------- */
const char* sf_last_error(void);
const char* sf_version(void);
/* number of visible HIP devices (0 on a CPU-only box; never fails) */
int sf_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* SYNFERENCE_HIP_H */
