/*
 * vaek.h -- C ABI of the MI355X-native ELBO train-step library (libvaek.so).
 *
 * The reference (virajmehta/vae-training) is pure Python on JAX/Flax and has no FFI; the
 * "native kernels" it runs are the XLA programs behind the calls cited on each entry point
 * below (paths relative to /root/reference).  This header is therefore the boundary a
 * maintainer would bind from Python with ctypes to replace
 *     VAE.train_step   networks.py:87-101   (forward + ELBO + backward + Adam, one jitted fn)
 *     VAE.loss         networks.py:103-113  (forward + ELBO, eval twin)
 *     VAE.apply        networks.py:61-84    (model(batch, z1, z2[, sampling=True]))
 *
 * Conventions
 *   - every function returns 0 on success, a negative vaek_status otherwise; it never throws
 *     and never aborts.  vaek_last_error() returns a thread-local message for the last failure.
 *   - every device buffer is owned by the caller (PyTorch-ROCm tensors on the Python side);
 *     the library receives raw device pointers and a hipStream_t (passed as void*), launches
 *     asynchronously on that stream and never synchronises or allocates after ctx_create
 *     (vaek_comm_* excepted: it maps peer memory once at init).
 *   - matrices are row-major; Dense kernels are [in, out] exactly as flax.nn.Dense stores
 *     them; all floating-point buffers are float32 unless a name says bf16.
 *   - parameters, gradients and both Adam moments each live in ONE flat float32 buffer with
 *     the fixed leaf order
 *         Encoder/FC0/kernel, Encoder/FC0/bias, ..., Decoder/FC0/kernel, ...,
 *         [SigDecoder/FC0/kernel, ...,]  epsilon_p (L),  [epsilon (1)]
 *     (names as in networks.py:67-78 and vae.py:73-80).  The gradient buffer has
 *     vaek_grad_len() = P + 4 floats: [P] = loss, [P+1] = mean Dkl, [P+2] = mean mse, [P+3] = 0,
 *     so that a data-parallel sum of the buffer also yields the global loss.
 *   - one host thread per context; one process per GPU for data parallelism.
 */
#ifndef VAEK_H
#define VAEK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VAEK_VERSION 100 /* 0.1.0 */
#define VAEK_MAX_HIDDEN 8

typedef enum vaek_status {
    VAEK_OK = 0,
    VAEK_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
    VAEK_ERR_HIP = -2,       /* a HIP runtime call failed                 */
    VAEK_ERR_NO_DEVICE = -3, /* no usable gfx950 device                   */
    VAEK_ERR_WORKSPACE = -4, /* workspace too small / misaligned          */
    VAEK_ERR_COMM = -5       /* peer-to-peer communicator failure         */
} vaek_status;

/* Activation codes of vaek_dense_fwd / vaek_dense_bwd_dx. */
enum { VAEK_ACT_NONE = 0, VAEK_ACT_RELU = 1 };

/* Compute dtype of the Dense GEMMs (ELBO, reductions and Adam are always float32). */
enum { VAEK_F32 = 0, VAEK_BF16 = 1 };

/* What VAE.partial binds at vae.py:57-59 plus the batch geometry. */
typedef struct vaek_config {
    int32_t struct_size;                  /* = sizeof(vaek_config), ABI guard                      */
    int32_t batch;                        /* rows handed to this rank per step (B_local)           */
    int32_t data_dim;                     /* D = prod(dataset.shape), vae.py:51                    */
    int32_t latent_dim;                   /* L, vae.py:50                                          */
    int32_t n_enc_hidden;                 /* encoder_layer_sizes without the appended L, vae.py:53 */
    int32_t enc_hidden[VAEK_MAX_HIDDEN];
    int32_t n_dec_hidden;                 /* layer_sizes without the appended D, vae.py:54         */
    int32_t dec_hidden[VAEK_MAX_HIDDEN];
    int32_t sigmoid_decoder;              /* dataset_name == "sigmoid": SigDecoder + Decoder, networks.py:75-78 */
    int32_t tunable_eps;                  /* -tdv: epsilon is a (1,) parameter times eps_cli, networks.py:70-71 */
    float   eps_cli;                      /* -e/--epsilon, run.py:31                               */
    int32_t dtype;                        /* VAEK_F32 | VAEK_BF16                                  */
    int32_t device;                       /* HIP device ordinal                                    */
    int32_t world;                        /* data-parallel ranks (1 = single GPU)                  */
    int32_t rank;
    int64_t global_batch;                 /* divisor of loss.mean() (networks.py:98); 0 -> batch*world */
    int32_t force_generic;                /* 1: never pick the fused small-model kernels (tests)   */
    int32_t reserved[7];
} vaek_config;

typedef struct vaek_ctx vaek_ctx;

/* ---- library ---------------------------------------------------------------------------- */
int vaek_version(void);
const char* vaek_last_error(void);

/* ---- context: replaces VAE.partial(...) + init_by_shape shape inference, vae.py:57-60 ---- */
int vaek_ctx_create(const vaek_config* cfg, vaek_ctx** out);
int vaek_ctx_destroy(vaek_ctx* ctx);
/* P = number of trainable floats; grad_len = P + 4 (see conventions). */
int vaek_param_count(const vaek_ctx* ctx, int64_t* P);
int vaek_grad_len(const vaek_ctx* ctx, int64_t* n);
/* Leaf table of the flat layout: n_leaves, then per leaf its offset and (rows, cols);
 * a bias / epsilon_p / epsilon leaf has rows == 1.  name buffers are NUL-terminated. */
int vaek_leaf_count(const vaek_ctx* ctx, int32_t* n_leaves);
int vaek_leaf_info(const vaek_ctx* ctx, int32_t leaf, char* name, int32_t name_cap,
                   int64_t* offset, int32_t* rows, int32_t* cols);
/* Bytes of caller-owned scratch (256-byte aligned) every entry point taking `workspace` needs. */
int vaek_workspace_bytes(const vaek_ctx* ctx, size_t* bytes);
/* 1 if the fused small-model path is used for train_step/loss_eval, 0 for layer-by-layer. */
int vaek_uses_fused_path(const vaek_ctx* ctx, int32_t* fused);

/* ---- building blocks (also used by the layer-by-layer path of vaek_train_step) ----------- */
/* flax.nn.Dense + relu, networks.py:34-39: y[rows,n_out] = act(x[rows,n_in] @ w[n_in,n_out] + b). */
int vaek_dense_fwd(vaek_ctx* ctx, const float* x, const float* w, const float* b, float* y,
                   int32_t rows, int32_t n_in, int32_t n_out, int32_t act, void* stream);
/* dx[rows,n_in] = (dy[rows,n_out] @ w^T) * (act == RELU ? x_post > 0 : 1); x_post is the
 * layer's INPUT as produced by the previous layer's relu (may be NULL for ACT_NONE).
 * accumulate != 0 adds into dx instead of overwriting (the two decoders of networks.py:76-78). */
int vaek_dense_bwd_dx(vaek_ctx* ctx, const float* dy, const float* w, const float* x_post, float* dx,
                      int32_t rows, int32_t n_in, int32_t n_out, int32_t act, int32_t accumulate, void* stream);
/* dwb[(n_in+1), n_out]: rows 0..n_in-1 = x^T @ dy (kernel gradient), row n_in = column sums of
 * dy (bias gradient) -- i.e. exactly the [kernel | bias] slice of the flat gradient buffer.
 * Deterministic (split over the batch into workspace slabs, then summed in a fixed order). */
int vaek_dense_bwd_dw(vaek_ctx* ctx, const float* x, const float* dy, float* dwb,
                      int32_t rows, int32_t n_in, int32_t n_out, void* workspace, void* stream);
/* networks.py:94-98 on explicit tensors.  x_hat_lin = Decoder(samples) WITHOUT the z2 noise;
 * x_hat_sig = SigDecoder pre-sigmoid output or NULL.  Writes out4 = {loss, mean Dkl, mean mse,
 * dL/d eps} (means over `batch_total`), and if d_lin != NULL the gradients w.r.t. the two
 * decoder outputs (d_sig may alias x_hat_sig, d_lin may alias x_hat_lin). */
int vaek_elbo_fwd_bwd(vaek_ctx* ctx, const float* x, const float* x_hat_lin, const float* x_hat_sig,
                      const float* z2, const float* mu, const float* logvar_e, float eps,
                      float* d_lin, float* d_sig, float* out4, int32_t rows, int32_t data_dim,
                      int32_t latent_dim, int64_t batch_total, void* workspace, void* stream);
/* The same with eps = eps_param_dev[0] * eps_scale read ON THE DEVICE (the tunable decoder variance of networks.py:70-71 is a
 * parameter: no host read, so a step built from the block entry points can be captured into a hipGraph). */
int vaek_elbo_fwd_bwd_dev(vaek_ctx* ctx, const float* x, const float* x_hat_lin, const float* x_hat_sig,
                          const float* z2, const float* mu, const float* logvar_e, const float* eps_param_dev, float eps_scale,
                          float* d_lin, float* d_sig, float* out4, int32_t rows, int32_t data_dim,
                          int32_t latent_dim, int64_t batch_total, void* workspace, void* stream);
/* flax.optim.Adam.apply_gradient, networks.py:100 (beta1 .9, beta2 .999, eps 1e-8).
 * `step_dev` (device int32, may be NULL) holds t of THIS update (1-based) when non-NULL,
 * otherwise `step` is used.  grad_scale multiplies the gradient first (1/world for means). */
int vaek_adam_step(vaek_ctx* ctx, float* params, const float* grads, float* m, float* v, int64_t n,
                   float lr, int32_t step, const int32_t* step_dev, float grad_scale, void* stream);

/* ---- the hot path ------------------------------------------------------------------------ */
/* VAE.train_step, networks.py:87-101, in place: reads params, x[B,D], z1[B,L], z2[B,D];
 * writes grads (vaek_grad_len floats; summed over ranks when the communicator is initialised),
 * updates params/m/v, increments *step_dev (device int32 Adam step counter), leaves the loss
 * in grads[P] (device; the reference keeps it un-synced too, vae.py:130).
 * Launches: linear VAEs with D, L <= 32 (the metric) run two kernels -- forward/backward/partial sums, then
 * reduction + Adam -- or ONE when the batch fits a workgroup (<= 256 rows, single GPU); any other MLP runs layer by
 * layer.  Asynchronous on `stream`, capturable into a hipGraph (no host-side state per step). */
int vaek_train_step(vaek_ctx* ctx, float* params, float* grads, float* m, float* v, int32_t* step_dev,
                    const float* x, const float* z1, const float* z2, float lr,
                    void* workspace, void* stream);
/* The two halves of vaek_train_step, for callers that all-reduce grads themselves (RCCL via
 * torch.distributed): grads_only leaves the LOCAL gradient sums (already divided by
 * global_batch) in grads and does not touch params; apply runs Adam on grads. */
int vaek_train_step_grads_only(vaek_ctx* ctx, const float* params, float* grads, int32_t* step_dev,
                               const float* x, const float* z1, const float* z2,
                               void* workspace, void* stream);
int vaek_train_step_apply(vaek_ctx* ctx, float* params, const float* grads, float* m, float* v,
                          const int32_t* step_dev, float lr, void* stream);
/* Bucketed variant for overlapping the data-parallel exchange with the rest of the backward pass
 * (layer-by-layer path): the flat gradient is final bucket by bucket, in backward order (decoder's last
 * layer first, encoder's first layer last, then the tail = epsilon_p, epsilon, loss slots).  Bucket i
 * covers grads[offset, offset + count) (vaek_bucket_info); ready_events[i] is a caller-owned hipEvent_t
 * (passed as void*) that the library records on `stream` the moment bucket i is complete, so a
 * communication stream can wait on it and all-reduce that slice while earlier layers are still in
 * their dW / dX GEMMs.  The fused small-model path has a single bucket. */
int vaek_bucket_count(const vaek_ctx* ctx, int32_t* n);
int vaek_bucket_info(const vaek_ctx* ctx, int32_t i, int64_t* offset, int64_t* count);
int vaek_train_step_grads_bucketed(vaek_ctx* ctx, const float* params, float* grads, int32_t* step_dev,
                                   const float* x, const float* z1, const float* z2, void* const* ready_events,
                                   void* workspace, void* stream);
/* VAE.loss, networks.py:103-113: out4 = {loss, mean Dkl, mean mse, eps}. */
int vaek_loss_eval(vaek_ctx* ctx, const float* params, const float* x, const float* z1, const float* z2,
                   float* out4, void* workspace, void* stream);
/* VAE.apply, networks.py:61-84.  sampling != 0: mu = 0, logvar_e = 0, samples = z1 and `eps`
 * is used as given (vae.py:199); otherwise x is encoded and eps comes from the parameters
 * (`eps` ignored).  x_hat[rows,D] includes the z2 noise; mu_out[rows,L] may be NULL. */
int vaek_forward(vaek_ctx* ctx, const float* params, const float* x, const float* z1, const float* z2,
                 int32_t sampling, float eps, float* x_hat, float* mu_out, int32_t rows,
                 void* workspace, void* stream);

/* ---- data-parallel exchange over xGMI (one-shot peer-to-peer all-reduce of the flat grads) -- */
/* The reference is single-device; this is the build's data-parallel addition.
 * Every rank calls vaek_comm_create (allocates ITS uncached exchange buffer -- the one allocation the
 * library makes after ctx_create -- and exports a 64-byte HIP IPC handle), the handles are
 * exchanged out of band (torch.distributed all_gather), and vaek_comm_init maps all peers.  After
 * that vaek_train_step sums gradients over ranks INSIDE its finalize kernel (tagged 8-byte granules
 * stored straight into every peer's buffer over xGMI; bounded spins).  Without a communicator and
 * world > 1 use vaek_train_step_grads_only + an RCCL all-reduce + vaek_train_step_apply. */
int vaek_comm_buffer_bytes(const vaek_ctx* ctx, size_t* bytes);
int vaek_comm_create(vaek_ctx* ctx, uint8_t handle_out[64]);
int vaek_comm_init(vaek_ctx* ctx, const uint8_t* all_handles /* world x 64 */);
int vaek_comm_destroy(vaek_ctx* ctx);
/* Stand-alone sum all-reduce of n floats in place through the communicator (n <= grad_len). */
int vaek_comm_allreduce(vaek_ctx* ctx, float* buf, int64_t n, void* stream);
/* Synchronous: *timed_out = 1 if any exchange on this rank ever gave up waiting for a peer. */
int vaek_comm_status(vaek_ctx* ctx, int32_t* timed_out);

/* ---- inputs of the hot path, generated on the device ---------------------- */
/* One Philox4x32-10 kernel replacing dataset.get_batch (datasets.py:75-84 sphere = kind 2, :183-195
 * linear_gaussian = kind 0 with A[dd][did], :240-249 sigmoid = kind 1 with A[dd]) and the latent draw
 * of model.py:225-228 split as vae.py:127-128 does (z1[rows,L], z2[rows,D]).  Row i of the call draws
 * from counter (row0 + i, block, step, tag) under key `seed`: reproducible, shardable by rows, and
 * graph-replayable when `step_dev` (the device Adam step counter) is given instead of step_host.
 * x may be NULL (latents only); z1 and z2 may both be NULL (dataset batch only, any D).  dd, did <= 16;
 * tag < 2^30. */
int vaek_make_batch(vaek_ctx* ctx, int32_t kind, const float* A, int32_t dd, int32_t did, int32_t pad, float var_added,
                    float* x, float* z1, float* z2, int32_t rows, int64_t row0, uint64_t seed,
                    const int32_t* step_dev, uint32_t step_host, uint32_t tag, void* stream);
/* The same draw for a loop that generates batch n+1 while step n trains (trainer.GraphLoop): the step is
 * counter[which] and the kernel ITSELF stores counter[which ^ 1] = step + 1 for the next launch, so a captured
 * launch needs no host argument and does not touch the Adam step counter a concurrently running train step is
 * incrementing.  The caller alternates `which` (0, 1, 0, ...) from launch to launch -- with two batch buffers that
 * is the buffer index -- and initialises counter[first which] to the first step.  Bit-identical to
 * vaek_make_batch(step_host = counter[which]). */
int vaek_make_batch_next(vaek_ctx* ctx, int32_t kind, const float* A, int32_t dd, int32_t did, int32_t pad, float var_added,
                         float* x, float* z1, float* z2, int32_t rows, int64_t row0, uint64_t seed,
                         int32_t* counter /* device int32[2] */, int32_t which, uint32_t tag, void* stream);
/* vaek_train_step on (x, z1, z2) AND vaek_make_batch_next into (x_next, z1_next, z2_next) -- the loop body of
 * model.py:221-222 with the draw for step n+1 taken off the critical path.  On the fused path the generator's work
 * items ride in the finalize launch (which by itself occupies 9 of 256 CUs): still two launches per step, one
 * stream, no cross-queue dependency; a batch that fits one workgroup (<= 256 rows, single GPU) is ONE launch, the draw on
 * its workgroups 1.. .  Elsewhere it is the two calls back to back.  Results are bit-identical to the
 * separate calls.  The next batch has ctx.batch rows and must not alias the current one. */
int vaek_train_step_gen(vaek_ctx* ctx, float* params, float* grads, float* m, float* v, int32_t* step_dev, const float* x,
                        const float* z1, const float* z2, float lr, void* workspace, int32_t kind, const float* A, int32_t dd,
                        int32_t did, int32_t pad, float var_added, float* x_next, float* z1_next, float* z2_next, int64_t row0,
                        uint64_t seed, int32_t* counter, int32_t which, uint32_t tag, void* stream);
/* N consecutive VAE.train_step's (the loop body of model.py:221-222 -> networks.py:87-101, N times) on N batches already
 * resident in HBM: xs / z1s / z2s are HOST arrays of n_steps device pointers (batch i = xs[i][B,D], z1s[i][B,L], z2s[i][B,D]).
 * On return (in stream order) params / m / v / *step_dev / grads are what n_steps calls of vaek_train_step on those batches
 * leave, to float32 summation-order tolerance -- NOT bitwise: the steps are evaluated through the batch's second-moment
 * matrix (csrc/linear_moments.hip), which takes the parameters off the streaming pass, so that the pass over batch n + 2,
 * the cross-workgroup sum of batch n + 1 and the Adam update of batch n run side by side -- as resident workgroup roles of one
 * persistent launch per 64 steps (arrival counters, bounded waits: vaek_train_steps_status), or where that form does not apply as
 * n_steps + 2 launches ordered by the stream alone.  Capturable into a hipGraph.  Linear encoder / decoder, one decoder, float32, L + 2 D + 1 <= 64,
 * and, with world > 1, an initialised P2P communicator (vaek_comm_create / vaek_comm_init: the moment matrix is additive over the
 * ranks' shards and is exchanged inside the launch; every rank must make the same calls, and the Adam step counter must not restart
 * while the communicator lives): vaek_supports_train_steps says whether this context qualifies; others return VAEK_ERR_INVALID. */
int vaek_supports_train_steps(const vaek_ctx* ctx, int32_t* yes);
int vaek_train_steps(vaek_ctx* ctx, float* params, float* grads, float* m, float* v, int32_t* step_dev,
                     const float* const* xs, const float* const* z1s, const float* const* z2s, int32_t n_steps, float lr,
                     void* workspace, void* stream);
/* Synchronous (reads one word back): *gave_up != 0 if a bounded in-launch wait of vaek_train_steps' / vaek_train_steps_gen's persistent
 * form has expired in ANY launch since the last call of this function (the results since then are invalid).  The word is STICKY on
 * the device -- no launch clears it, and while it is set every wait of every later launch returns at once (the grid drains) -- and
 * this call is read-and-clear.  It says which wait came first: 0x80000000 | role << 28 (1 the updater, 2 a reducer, 3 a reducer
 * waiting for a peer rank's moments) | batch index within the launch << 16 | the arrival count it last saw. */
int vaek_train_steps_status(vaek_ctx* ctx, void* workspace, int32_t* gave_up);
/* The same N train steps with the batches DRAWN inside the launch: step k of the call (the one that takes *step_dev from t to
 * t + 1) trains on the batch vaek_make_batch(kind, A, dd, did, pad, var_added, rows = ctx.batch, row0, seed, step = t, tag)
 * would write -- the same Philox4x32-10 counters, Box-Muller and dataset maps (csrc/rng_dev.h), bit for bit -- but the batch
 * never exists in HBM: the streamers of the persistent launch draw each tile straight into LDS between the products of an
 * earlier tile.  This is the loop body of the reference, model.py:221-222 (dataset.get_batch -> vae.py:123-130 sample_latent,
 * VAE.train_step), N times per launch.  kind: 0 linear_gaussian, 2 sphere (a linear VAE on the sigmoid dataset has two decoders:
 * not covered).  Data parallel: every rank passes its own row0 (global row indices).  Capturable into a hipGraph (the RNG step
 * is the device-resident Adam counter).  vaek_supports_train_steps_gen says whether this context / dataset kind qualifies;
 * status as vaek_train_steps. */
int vaek_supports_train_steps_gen(const vaek_ctx* ctx, int32_t kind, int32_t* yes);
/* The two halves of ONE step of vaek_train_steps' launch-per-step form, for data parallelism over a HOST collective (no P2P
 * communicator needed): the second-moment matrix of a linear VAE's batch is additive over the ranks' row shards -- the loss of
 * networks.py:97-98 is a batch mean -- so every rank calls vaek_train_steps_moments on its shard (x, z1, z2 of ctx.batch rows ->
 * M, a float64 image of vaek_train_steps_moment_len doubles), the caller sums M over the ranks (torch.distributed all_reduce:
 * RCCL / gloo; every rank receives the same bits), and vaek_train_steps_update turns the summed M into loss, gradients and the Adam
 * update of networks.py:99-101 with the GLOBAL batch as divisor (ctx.global_batch): replicas stay bitwise identical.  With world
 * == 1 the pair is one train step.  *len == 0: this context is not a linear VAE the moment form covers. */
int vaek_train_steps_moment_len(const vaek_ctx* ctx, int64_t* len);
int vaek_train_steps_moments(vaek_ctx* ctx, const float* x, const float* z1, const float* z2, double* M, void* workspace, void* stream);
int vaek_train_steps_update(vaek_ctx* ctx, float* params, float* grads, float* m, float* v, int32_t* step_dev, const double* M, float lr,
                            void* workspace, void* stream);
int vaek_train_steps_gen(vaek_ctx* ctx, float* params, float* grads, float* m, float* v, int32_t* step_dev, int32_t kind, const float* A,
                         int32_t dd, int32_t did, int32_t pad, float var_added, int64_t row0, uint64_t seed, uint32_t tag, int32_t n_steps,
                         float lr, void* workspace, void* stream);
/* Convolutional VAE of BASELINE config 5 -- NO reference counterpart (the reference has no convolutional model: its only image
 * code is utils.py:129-133); the layer is specified in DESIGN.md 3.4 and checked against oracle/conv_vae_oracle.py:conv_fwd.
 * 4 x 4 / stride 2 / pad 1 convolution, NHWC float32 tensors, HWIO kernel [4][4][c_in][c_out], bf16 matrix-core products with
 * float32 accumulation (the envelope of the bf16 Dense path, not the 1e-5 ELBO contract):
 *   y[n, i, j, o] = act(bias[o] + sum_{kh, kw, c} x[n, 2 i + kh - 1, 2 j + kw - 1, c] * w[kh, kw, c, o]),  y: [batch, height/2, width/2, c_out].
 * No context needed.  With the entry points below every product of both layer kinds' forward and backward passes exists
 * (vae_training_amd/conv_vae.py assembles the train step from them).
 * `workspace`: vaek_conv2d_forward_workspace bytes (transposed = 0 / 1 for the two entry points; 0 bytes = none needed), 16-byte
 * aligned, or NULL.  With a workspace, c_in a power of two (>= 8; >= 16 transposed), c_out a multiple of 32 and 16-byte aligned
 * tensors the layer runs on bf16 copies through the LDS-DMA GEMM (the same bf16 products, several times faster); otherwise, and
 * always for NULL, the register-staged kernel.  Layers with ONE channel on the thin side (c_in = 1 here, c_out = 1 transposed)
 * are streaming float32 kernels -- exact, no bf16 rounding.
 * bf16 copies (all optional, NULL = none; 16-byte aligned): `x_bf16` / `dy_bf16` / `y_bf16` INPUTS are bf16 images of the float32
 * tensor of the same name that the caller vouches for (an earlier call's output copy, or vaek_to_bf16) -- the LDS-DMA form then
 * skips its own conversion pass, the other forms ignore them; `y_bf16` / `out_bf16` OUTPUTS are written with the bf16 rounding of
 * the float32 result in every form (from the epilogue where the form can, by a conversion pass otherwise).
 * LEAN forms (round 3: a hidden tensor of the conv VAE lives in HBM as bf16 only; LDS-DMA shapes only, VAEK_ERR_INVALID elsewhere):
 *   - the float32 INPUT (x, y of the transposed call; x / dy of the kernel gradient) may be NULL when its bf16 copy is given;
 *   - the float32 RESULT (y, out) may be NULL when its bf16 copy is asked for -- also on the one-channel forward layer's matrix-core form;
 *   - relu bit 1 (relu = 2 or 3): `mask` points to the bf16 copy of the mask source instead of the float32 tensor
 *     ([mask > 0] is the same set either way: bf16 rounding keeps the sign and never reaches zero from a normal number). */
int vaek_conv2d_forward_workspace(int32_t batch, int32_t height, int32_t width, int32_t c_in, int32_t c_out, int32_t transposed, size_t* bytes);
int vaek_conv2d_forward(const float* x, const float* w, const float* bias, const float* mask, float* y, int32_t batch, int32_t height,
                        int32_t width, int32_t c_in, int32_t c_out, int32_t relu, void* workspace, const void* x_bf16, void* y_bf16,
                        void* stream);   /* mask: as below; NULL = none */
int vaek_to_bf16(const float* src, void* dst_bf16, int64_t n, void* stream);      /* n floats -> n bf16, round to nearest even */
/* The transposed convolution of the same specification = the adjoint of vaek_conv2d_forward with the SAME kernel array
 * (oracle: conv_t_fwd): y [batch, height, width, c_in], w [4][4][c_out][c_in] (the HWIO kernel of the convolution it is the adjoint
 * of), out [batch, 2 height, 2 width, c_out] = act(bias + ...).  It is also the convolution's input gradient (y := dL/d output, bias
 * NULL); `mask` (NULL or a tensor of out's shape) multiplies the result by [mask > 0] -- the relu of the layer below. */
int vaek_conv2d_transpose_forward(const float* y, const float* w, const float* bias, const float* mask, float* out, int32_t batch,
                                  int32_t height, int32_t width, int32_t c_in, int32_t c_out, int32_t relu, void* workspace,
                                  const void* y_bf16, void* out_bf16, void* stream);
/* Kernel gradient of vaek_conv2d_forward (oracle: conv_bwd): dw[kh, kw, c, o] = sum_{n, i, j} x[n, 2 i + kh - 1, 2 j + kw - 1, c] *
 * dy[n, i, j, o], dbias[o] = sum dy (NULL: not wanted); x [batch, height, width, c_in], dy [batch, height/2, width/2, c_out].
 * Batch-split slabs in `workspace` (vaek_conv2d_weight_grad_workspace bytes) + a fixed-order sum: bitwise repeatable.  With
 * (x := dL/d out, dy := the layer's input) it is the TRANSPOSED layer's kernel gradient in its [4][4][c_out][c_in] layout. */
int vaek_conv2d_weight_grad_workspace(int32_t batch, int32_t height, int32_t width, int32_t c_in, int32_t c_out, size_t* bytes);
int vaek_conv2d_weight_grad(const float* x, const float* dy, float* dw, float* dbias, void* workspace, int32_t batch, int32_t height,
                            int32_t width, int32_t c_in, int32_t c_out, const void* x_bf16, const void* dy_bf16, void* stream);
/* dbias[c] = sum over the pixels of dy[pixels][c] (the bias gradient of a transposed layer); workspace: 512 * c floats.
 * _bf16: the same sums (float32 accumulation, fixed order) of a bf16 tensor -- c a power of two in 8 .. 2048, 16-byte aligned. */
int vaek_conv2d_bias_grad(const float* dy, float* dbias, void* workspace, int64_t pixels, int32_t c, void* stream);
int vaek_conv2d_bias_grad_bf16(const void* dy_bf16, float* dbias, void* workspace, int64_t pixels, int32_t c, void* stream);
/* Dense + reparameterisation (networks.py:72-74) as one block: mu = x @ w + b, samples = mu + exp(logvar_e / 2) * z1. */
int vaek_dense_fwd_reparam(vaek_ctx* ctx, const float* x, const float* w, const float* b, float* mu, float* samples, const float* z1,
                           const float* logvar_e, int32_t rows, int32_t n_in, int32_t n_out, void* stream);
/* Backward of the reparameterisation + the KL term's mu and logvar_e parts (what jax.value_and_grad does with networks.py:73-74, 94):
 * in place d_samples -> d_mu = d_samples + mu / batch_total; d_logvar_e[l] = 0.5 exp(lv_l / 2) sum_rows d_samples z1
 * - 0.5 (1 - exp(lv_l)) rows / batch_total.  latent_dim <= 256. */
int vaek_reparam_bwd(vaek_ctx* ctx, float* d_samples, const float* mu, const float* z1, const float* logvar_e, float* d_logvar_e,
                     int32_t rows, int32_t latent_dim, int64_t batch_total, void* workspace, void* stream);
/* n standard normals and/or the raw Philox words they came from (block b = counter (b_lo, b_hi, step, tag)). */
int vaek_rng_fill(vaek_ctx* ctx, float* normals, uint32_t* bits, int64_t n, uint64_t seed, uint32_t step, uint32_t tag,
                  void* stream);
/* Optional device ring buffer: every vaek_train_step also stores its loss at buf[(t - 1) % cap],
 * t = Adam step (what the reference appends to vae_losses, vae.py:130, without a per-step copy). */
int vaek_set_loss_history(vaek_ctx* ctx, float* buf, int64_t cap);

/* ---- roofline denominators measured on the box (bench.py; not on the train-step path) -------------- */
/* float4 stream copy of `bytes` (multiple of 16) src -> dst; time it with vaek_profile_*. */
int vaek_microbench_copy(vaek_ctx* ctx, const void* src, void* dst, int64_t bytes, void* stream);
/* Back-to-back MFMA loop, independent accumulators: kind 0 = v_mfma_f32_16x16x4_f32, 1 =
 * v_mfma_f32_32x32x16_bf16; `waves_per_simd` workgroups of 4 waves per CU.  *flops_out = flops of the launch. */
int vaek_microbench_mfma(vaek_ctx* ctx, int32_t kind, int32_t iters, int32_t waves_per_simd, float* scratch,
                         double* flops_out, void* stream);

/* Launch floor of this platform: n back-to-back launches of a kernel of `blocks` workgroups that does nothing (kind 0)
 * or one dependent pair of loads per thread (kind 1; p = >= 65 small ints, out = >= blocks ints).  Time them with
 * vaek_profile_* (kernel timestamps) or around a hipGraph replay (launch-to-launch interval). */
int vaek_microbench_launch(vaek_ctx* ctx, int32_t kind, int32_t blocks, int32_t n, const int32_t* p, int32_t* out, void* stream);

/* ---- in-process kernel timing (bench.py's roofline leg) --------------------------------------- */
/* Between begin and report every kernel the library launches for this context is bracketed by a
 * pair of hipEvents recorded on the launch stream (pool of max_records pairs, allocated here, so
 * the launch path still allocates nothing).  vaek_profile_report synchronises the events, stops
 * profiling and writes a JSON object {"label": {"count": n, "total_ms": t}, ...} into buf. */
int vaek_profile_begin(vaek_ctx* ctx, int32_t max_records);
int vaek_profile_report(vaek_ctx* ctx, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* VAEK_H */
