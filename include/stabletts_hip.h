/*
 * stabletts_hip.h -- C ABI of libstabletts_hip.so: the MI355X (gfx950) native
 * conditional-flow-matching mel decoder of StableTTS.
 *
 * The reference (KdaiP/StableTTS) has no FFI layer; its boundary for this path is the
 * Python class models/flow_matching.py:11 `CFMDecoder`.  Each entry point below names the
 * reference interface it replaces.  Conventions:
 *   - every function returns 0 on success, a negative ST_ERR_* code on failure, and never
 *     throws across the ABI; st_last_error() returns the message of the last failure.
 *   - tensor pointers are BORROWED DEVICE pointers (fp32, contiguous, the reference's own
 *     layouts: (B, C, T) row-major), owned by the caller (torch).  The engine owns its packed
 *     16-bit weight copies and its workspace.
 *   - work is enqueued on the caller's HIP stream (`stream` = hipStream_t, e.g.
 *     torch.cuda.current_stream().cuda_stream) with no implicit device synchronisation.
 *   - one engine per device per process; an engine is not thread-safe (the reference is
 *     single-threaded per process: train.py:101-102, webui.py:128).
 */
#ifndef STABLETTS_HIP_H
#define STABLETTS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ST_ABI_VERSION 4

enum {
    ST_OK = 0,
    ST_ERR_INVALID = -1,      /* bad argument / unsupported configuration */
    ST_ERR_HIP = -2,          /* a HIP runtime call failed */
    ST_ERR_STATE = -3,        /* e.g. solve before all parameters were loaded */
    ST_ERR_UNSUPPORTED = -4   /* valid in the reference, not implemented natively (caller may fall back) */
};

/* MFMA operand type of the dense contractions (accumulation, residual stream, LayerNorm,
 * softmax statistics and the ODE state are always fp32). */
enum { ST_OPERAND_BF16 = 0, ST_OPERAND_F16 = 1 };

/* Fixed-grid solvers of torchdiffeq.odeint as used at models/flow_matching.py:54. */
enum { ST_SOLVER_EULER = 0, ST_SOLVER_MIDPOINT = 1, ST_SOLVER_RK4 = 2,
       ST_SOLVER_DOPRI5 = 3,  /* adaptive Dormand-Prince 5(4), rtol = atol = 1e-5: the reference default (solver=None) */
       /* the other explicit adaptive pairs of torchdiffeq offered by webui.py:110, same controller and tolerances */
       ST_SOLVER_BOSH3 = 4, ST_SOLVER_FEHLBERG2 = 5, ST_SOLVER_ADAPTIVE_HEUN = 6,
       ST_SOLVER_IMPLICIT_ADAMS = 7 };  /* torchdiffeq 'implicit_adams' (webui.py:110): Adams-Bashforth-Moulton on the fixed grid of n_steps,
                                           functional iteration of the corrector (<= 4 evaluations per step, rtol = atol = 1e-5) */

/* Constructor arguments of reference CFMDecoder.__init__ (models/flow_matching.py:12). */
typedef struct st_config {
    int32_t noise_channels;   /* = cond_channels = out_channels = n_mels (config.py:12: 128) */
    int32_t hidden_channels;  /* 256 (config.py:23) */
    int32_t filter_channels;  /* 1024 */
    int32_t n_heads;          /* 4 */
    int32_t n_layers;         /* 6 (n_dec_layers) */
    int32_t kernel_size;      /* 3 */
    int32_t gin_channels;     /* 256 */
    int32_t operand_dtype;    /* ST_OPERAND_* */
} st_config;

typedef struct st_engine st_engine;

/* ABI version of the loaded library (compare with ST_ABI_VERSION). */
int st_abi_version(void);

/* Replaces CFMDecoder.__init__ / Decoder.__init__ (models/flow_matching.py:12-22,
 * models/estimator.py:66-96): validates the architecture (same assertions: n_layers even,
 * hidden % n_heads == 0, even time-embedding dim) and creates an engine on HIP device `device`. */
int st_create(const st_config* cfg, int device, st_engine** out);

void st_destroy(st_engine* e);

/* Message of the last failure on this engine (e == NULL: last st_create failure). */
const char* st_last_error(const st_engine* e);

/* Replaces nn.Module.load_state_dict for `decoder.estimator.*` (api.py:49, utils/load.py:31-41):
 * uploads one tensor by its reference state_dict name, fp32,
 * reference shapes (Conv1d weights (Cout, Cin, K)).  `data` may be a host or a device pointer. */
int st_load_param(st_engine* e, const char* name, const float* data, const int64_t* shape, int ndim);

/* Number of state_dict tensors the configured architecture expects (116 for the 31M model). */
int st_num_params(const st_engine* e);

/* Enumerates the expected tensors (index in [0, st_num_params), name order): reference state_dict name and
 * shape, so that a host without the Python module tree can discover what to load.  Returns the number of
 * dimensions (<= 4) and writes them to shape[0..ndim) if shape != NULL; *name points into the engine. */
int st_param_info(const st_engine* e, int index, const char** name, int64_t* shape);

/* Packs the uploaded fp32 parameters into the engine's 16-bit MFMA operand layouts.  Must be
 * called after loading (and again after any parameter update).  Synchronises the device. */
int st_finalize(st_engine* e);

/* The training-loop form of the two calls above (nn.Parameter semantics: train.py:78-82 updates the weights in place
 * every iteration).  st_bind_param makes the engine READ the fp32 tensor at the caller's device pointer instead of
 * keeping a copy: `data` (device memory, reference shape) stays owned by the caller and must outlive the binding
 * (re-bind after the storage moves).  After an in-place update of bound tensors (optimizer.step()), st_repack
 * re-packs the 16-bit operand copies INTO THE EXISTING BUFFERS as kernels on `stream` -- no allocation, no host
 * copy, no device synchronisation; it needs one earlier st_finalize (which allocates them). */
int st_bind_param(st_engine* e, const char* name, const float* data, const int64_t* shape, int ndim);
int st_repack(st_engine* e, void* stream);

/* Replaces Decoder.forward(t, x, mask, mu, c) (models/estimator.py:103-138): ONE vector-field
 * evaluation.  t: device fp32, t_len = 1 (inference, 0-dim t) or B (training, flow_matching.py:99).
 * x, mu, out: (B, n_feats, T); mask: (B, 1, T) float 0/1; c: (B, gin). */
int st_estimator_forward(st_engine* e, const float* t, int t_len, const float* x, const float* mu,
                         const float* mask, const float* c, float* out, int B, int T, void* stream);

/* Replaces CFMDecoder.forward (models/flow_matching.py:25-55) + cfg_wrapper (:58-67) +
 * torchdiffeq.odeint fixed-grid stepping (:54): the whole ODE solve.
 *   z            : initial noise (B, n_feats, T), ALREADY multiplied by the temperature (:45).
 *   n_steps      : n_timesteps; the grid is linspace(0, 1, n_steps + 1) (:46).
 *   solver       : ST_SOLVER_*.  For the adaptive solvers (>= ST_SOLVER_DOPRI5) n_steps only names the output grid of the reference
 *                  call (flow_matching.py:46,54) and does not influence the steps taken.
 *   use_cfg != 0 : classifier-free guidance with fake_speaker (gin,), fake_content (n_feats,)
 *                  and cfg_strength (models/model.py:43-44,102); cond and uncond branches run as
 *                  one 2B batch.
 *   out          : trajectory[-1], (B, n_feats, T).
 * Environment: ST_HIP_GRAPH=1 replays the fixed-grid solve body (everything between the boundary layout
 * conversions) from a HIP graph captured at the second call with the same (B, T, n_steps, solver, CFG) signature. */
int st_cfm_solve(st_engine* e, const float* mu, const float* mask, const float* z, const float* c,
                 int n_steps, int solver, int use_cfg, float cfg_strength,
                 const float* fake_speaker, const float* fake_content,
                 float* out, int B, int T, void* stream);

/* ---- TextEncoder: the caller side of the path, on the same DiT block kernels ---------------- */

/* Replaces TextEncoder.__init__ (models/text_encoder.py:9-28).  cfg fields are read as: noise_channels =
 * out_channels (n_mels), hidden / filter / n_heads / kernel_size / gin_channels / operand_dtype as for the
 * decoder, n_layers = n_enc_layers (any 1..16).  Parameters are loaded with st_load_param under the reference
 * names ("emb.weight", "encoder.<i>.attn.conv_q.weight", ..., "proj.bias") and packed by st_finalize; the handle
 * is destroyed with st_destroy.  The decoder entry points reject such a handle and vice versa. */
int st_create_text_encoder(const st_config* cfg, int n_vocab, int device, st_engine** out);

/* Replaces TextEncoder.forward(x, c, x_lengths) (models/text_encoder.py:34-44).
 *   tokens  : (B, T) int64 phoneme ids (ids outside [0, n_vocab) are clamped; nn.Embedding would raise)
 *   lengths : (B,) int64 valid lengths;  c: (B, gin) fp32 speaker vectors          -- all device pointers
 *   x_out   : (B, hidden, T) fp32 encoder states;  mu_out: (B, n_mels, T) fp32 = proj(x) * mask;
 *   mask_out: (B, 1, T) fp32 sequence mask (utils/mask.py). */
int st_text_encoder_forward(st_engine* e, const int64_t* tokens, const int64_t* lengths, const float* c,
                            float* x_out, float* mu_out, float* mask_out, int B, int T, void* stream);

/* ---- duration -> alignment -> mu_y: the caller side between TextEncoder and CFMDecoder ------------ */
/* Stateless (no engine handle; a failure's message is st_last_error(NULL)); all pointers are device pointers.       */

/* Replaces models/model.py:85-87 (synthesise): w = exp(logw) * x_mask; w_ceil = ceil(w) * length_scale;
 * y_lengths = clamp_min(sum(w_ceil), 1).long().  logw, x_mask: (B, 1, Tx) fp32.  Outputs: w_ceil (B, Tx),
 * cum (B, Tx) = cumsum(w_ceil) (generate_path's first step, :19; sequential fp32), y_lengths (B) int64.  The host
 * reads y_lengths.max() to size the mel tensors, exactly as the reference does at :88. */
int st_durations(const float* logw, const float* x_mask, float length_scale, int B, int Tx, float* w_ceil, float* cum,
                 int64_t* y_lengths, void* stream);

/* Replaces generate_path(duration, mask) (models/model.py:17-27): duration (B, Tx), mask (B, Tx, Ty) -> path (B, Tx, Ty)
 * 0/1 monotonic alignment.  cum_scratch: (B, Tx) fp32 workspace. */
int st_generate_path(const float* duration, const float* mask, int B, int Tx, int Ty, float* cum_scratch, float* path, void* stream);

/* Replaces models/model.py:91-95: y_mask = sequence_mask(y_lengths, Ty); attn = generate_path(w_ceil, x_mask x y_mask);
 * mu_y = attn^T mu_x -- as ONE gather (every mel frame copies the text position its 0/1 alignment column selects).
 * cum (B, Tx) from st_durations; mu_x (B, M, Tx); outputs mu_y (B, M, Ty), y_mask (B, 1, Ty) (optional) and the
 * alignment attn (B, Tx, Ty) (optional, NULL to skip: synthesise returns it, the decoder does not need it). */
int st_align(const float* cum, const float* x_mask, const int64_t* y_lengths, const float* mu_x, int B, int M, int Tx, int Ty,
             float* attn, float* mu_y, float* y_mask, void* stream);

/* ---- CFMDecoder.compute_loss's own arithmetic (models/flow_matching.py:86-100), stateless like the alignment helpers ------
 * st_cfm_loss_prep: t = 1 - cos(t_rand pi / 2) (:88), y = (1 - (1 - sigma) t) z + t x1 (:93), u = x1 - (1 - sigma) z (:96).
 *   x1, z, y, u: (B, M, T); t_rand, t: (B).
 * st_cfm_loss: loss = sum((pred - u)^2) / (sum(mask) * M) (:100; u is NOT masked, as in the reference) -> *loss (device);
 *   scratch: st_cfm_loss_scratch_floats() floats, keeps the denominator for the backward; deterministic summation order.
 * st_cfm_loss_backward: grad_pred = grad_loss[0] * 2 (pred - u) / (sum(mask) * M)  (grad_loss: device scalar). */
int st_cfm_loss_prep(const float* x1, const float* z, const float* t_rand, float sigma_min, int B, int M, int T, float* t, float* y,
                     float* u, void* stream);
int st_cfm_loss(const float* pred, const float* u, const float* mask, int B, int M, int T, float* scratch, float* loss, void* stream);
int st_cfm_loss_backward(const float* pred, const float* u, const float* scratch, const float* grad_loss, int B, int M, int T,
                         float* grad_pred, void* stream);
int st_cfm_loss_scratch_floats(void);

/* ---- Vocos vocoder: mel -> waveform, the step after the decoder (api.py:76) ------------------------ */

/* Replaces Vocos.__init__(VocosConfig(), MelConfig()) (vocoders/vocos/models/model.py:11-15, config.py:4-19,46-50). */
typedef struct st_vocos_config {
    int32_t input_channels;    /* n_mels, config.py:47 (multiple of 64) */
    int32_t dim;               /* config.py:48; native kernels: 512 */
    int32_t intermediate_dim;  /* config.py:49 (multiple of 256) */
    int32_t num_layers;        /* config.py:50 (1..32) */
    int32_t n_fft;             /* MelConfig.n_fft, config.py:6; native kernels: 2048 */
    int32_t hop_length;        /* MelConfig.hop_length, config.py:8; native kernels: 512 */
    int32_t operand_dtype;     /* ST_OPERAND_* of the GEMM operands (accumulation, LayerNorms, residual stream, ISTFT: fp32) */
} st_vocos_config;

/* The handle takes the parameters of Vocos.state_dict() under their reference names ("backbone.embed.weight", ...,
 * "backbone.convnext.<i>.gamma", ..., "head.out.bias", "head.istft.window") through st_load_param / st_finalize and
 * is destroyed with st_destroy.  The decoder / text-encoder entry points reject it and vice versa. */
int st_create_vocoder(const st_vocos_config* cfg, int device, st_engine** out);

/* Replaces Vocos.forward(x) (model.py:17-20) = ISTFTHead(VocosBackbone(x)) (backbone.py:50-56, head.py:93-117 with
 * padding="same"):  mel (B, input_channels, T) fp32 -> audio (B, T * hop_length) fp32, device pointers.  Like the
 * reference there is no mask: every utterance is vocoded at the padded length T. */
int st_vocos_forward(st_engine* e, const float* mel, float* audio, int B, int T, void* stream);

/* ---- training: autograd counterpart of Decoder.forward ------------------------------------------- */

/* Replaces Decoder.forward(t, x, mask, mu, c) UNDER AUTOGRAD as CFMDecoder.compute_loss calls it (models/flow_matching.py:99,
 * train.py:78-81): one vector-field evaluation with a per-item t (B values) that keeps every activation the backward
 * pass needs in the engine.  Train-mode dropout of the reference (p_dropout on the FFN activations and on the attention
 * probabilities, models/diffusion_transformer.py:22,52,77) is counter-based: the same (seed, element) hash is
 * re-evaluated by the backward kernels, nothing is stored.  Pointers as for st_estimator_forward. */
int st_train_forward(st_engine* e, const float* t, const float* x, const float* mu, const float* mask, const float* c,
                     float* out, int B, int T, float p_dropout, uint64_t seed, void* stream);

/* Serial number of the st_train_forward whose activations the engine holds now (monotonically increasing from 1;
 * 0 = none: never run, or invalidated by a parameter update).  The engine keeps the activations of ONE forward. */
int64_t st_train_serial(const st_engine* e);

/* Replaces torch.autograd's backward of that call.  grad_out: d loss / d out (B, n_feats, T).  Writes d loss / d x,
 * d loss / d mu (B, n_feats, T) and d loss / d c (B, gin) where the pointer is not NULL, and the gradient of every
 * parameter into engine-owned fp32 buffers in the reference shapes (fetch with st_param_grad).
 * `serial`, B, T identify the forward this is the backward OF (st_train_serial right after that st_train_forward):
 * if another forward has replaced its activations, or the parameters were re-packed, or the shape differs, the
 * call fails with ST_ERR_STATE before touching any caller memory. */
int st_train_backward(st_engine* e, int64_t serial, int B, int T, const float* grad_out, float* grad_x, float* grad_mu,
                      float* grad_c, void* stream);

/* The same backward in three PARTS, so that a data-parallel wrapper (DDP, train.py:49-51) can reduce the gradients of finished
 * layers while the remaining ones are still being computed -- torch's autograd delivers them layer by layer, a single native call
 * would deliver all 116 at once:
 *     part 0: final_proj, blocks L-1 .. L/2, the long-skip convs      part 1: blocks L/2-1 .. 0
 *     part 2: in_proj, the cond prenet, the time MLP; writes d x, d mu, d c (grad_x / grad_mu / grad_c, each may be NULL)
 * st_train_param_part(name) says which part finishes a parameter's gradient.  Parts run in order 0, 1, 2 (ST_ERR_STATE otherwise);
 * grad_out is read by part 0 only.  grad_flat (part 0; may be NULL = the engine's own buffers, fetch with st_param_grad): a
 * caller-owned device buffer of st_train_grad_numel() floats that receives EVERY parameter gradient of this backward directly
 * -- no staging copy -- at st_train_grad_offset(name) (reference shape, 64-byte aligned slices, st_param_info's order); it must stay
 * valid until part 2 has run.  The legacy call above = the three parts with the engine's own buffers. */
int st_train_backward_part(st_engine* e, int64_t serial, int B, int T, int part, const float* grad_out, float* grad_flat,
                           int64_t grad_numel, float* grad_x, float* grad_mu, float* grad_c, void* stream);
int st_train_param_part(const st_engine* e, const char* name);        /* 0, 1, 2; < 0: unknown name */
int64_t st_train_grad_offset(const st_engine* e, const char* name);   /* element offset in the flat gradient layout; < 0: unknown */
int64_t st_train_grad_numel(const st_engine* e);                      /* floats of the flat layout (alignment gaps included) */

/* Copies the gradient of one parameter (reference state_dict name, `numel` fp32 values) to the device pointer dst. */
int st_param_grad(st_engine* e, const char* name, float* dst, int64_t numel, void* stream);

/* All parameter gradients in ONE copy: dst receives the flat layout described above (`numel` = st_train_grad_numel()). */
int st_param_grads_flat(st_engine* e, float* dst, int64_t numel, void* stream);

/* Non-finite guard (no reference analogue: the reference is fp32).  The kernel that writes the output of st_estimator_forward /
 * st_cfm_solve raises a flag when a value is NaN / Inf -- f16 MFMA operands overflow at 65504 (a checkpoint whose activations
 * exceed that needs operand_dtype = bf16), or the inputs were bad.  Synchronises `stream`, stores the flag of the calls completed
 * since the last query in *nonfinite (0 / 1) and clears it; after a flagged call the engine re-zeroes its workspace by itself. */
int st_output_status(st_engine* e, void* stream, int* nonfinite);

/* Function evaluations, attempted steps and rejected steps of the last st_cfm_solve (adaptive solvers vary). */
int st_last_solve_stats(const st_engine* e, int64_t* nfe, int64_t* steps, int64_t* rejects);

/* Engine options by name (no reference analogue: models/diffusion_transformer.py:70-77 computes the attention in fp32).
 *   "attention_precision"  0 (default): q, k, v enter the attention MFMAs as 16-bit operands.
 *                          1: q and k as hi + lo PAIRS of 16-bit operands, scores = q_hi k_hi + q_lo k_hi + q_hi k_lo (3x the QK^T
 *                             MFMAs) in st_estimator_forward / st_cfm_solve -- for checkpoints whose softmax has become an arg-max
 *                             (score maxima of 80-200), where the 2^-11 rounding of q and k moves the winning probability and
 *                             f16 operands miss the 1e-3 parity bar.  Takes effect at the next call.
 *   "fused_ffn"            read-only: 0 two-kernel FFN, 1 fused direct kernel (default), 3 fused Winograd kernel (ST_FUSED_FFN=3).
 * st_get_option returns ST_ERR_INVALID for an unknown name. */
int st_set_option(st_engine* e, const char* name, int value);
int st_get_option(const st_engine* e, const char* name, int* value);

/* Largest log-sum-exp (natural-log units; of the scaled, masked scores of softmax(q k^T / sqrt(d) + mask), diffusion_transformer.py:77)
 * over every valid attention row of the estimator evaluations completed on `stream` since the last query; -inf if there were none.  A
 * row's score maximum lies within log(T) below it, so this is the run-time sign of the arg-max regime above: seeded / freshly
 * initialised weights give ~10, values beyond ~50 mean "attention_precision" = 1 is needed for 1e-3 parity.  Synchronises `stream`,
 * reads 16 words, resets the statistic.  Costs one atomic per attention block while running. */
int st_attention_stats(st_engine* e, void* stream, float* max_lse);

/* ---- measurement / test hooks (no reference analogue) ------------------------------------- */

/* Copies a named internal tensor of the LAST st_estimator_forward call to `host_out` as fp32 in
 * the engine's time-major layout; returns the element count, or <0.  If host_out is
 * NULL only the count is returned.  Synchronises the device.  Names: "cond", "h0", "b<i>.x1",
 * "b<i>.h1", "b<i>.q", "b<i>.k", "b<i>.vt", "b<i>.attn", "b<i>.x2", "b<i>.h2", "b<i>.u",
 * "b<i>.x3", "lsc<j>", "v".  Capture must be enabled first (it snapshots after every stage). */
int st_debug_capture(st_engine* e, int enable);
int64_t st_debug_fetch(st_engine* e, const char* name, float* host_out, int64_t capacity);

/* Per-kernel-class timing with HIP events recorded on the launch stream.  Classes are listed by
 * st_profile_class_name(i), i in [0, st_profile_num_classes()).  st_profile_read synchronises,
 * returns launches and total milliseconds per class since the last reset, and resets. */
int st_profile_enable(st_engine* e, int enable);
/* Restrict event recording to the classes whose bit is set in class_mask (default: all). */
int st_profile_select(st_engine* e, uint64_t class_mask);
/* Record events around every stride-th launch of a selected class only (default 1 = every launch).  An event
 * pair costs ~10 us of idle stream time on MI355X, so a timed run samples: st_profile_read then reports the
 * SAMPLED launches and their total, i.e. total_ms / launches is still the mean launch duration. */
int st_profile_stride(st_engine* e, int stride);
int st_profile_num_classes(void);
const char* st_profile_class_name(int cls);
int st_profile_read(st_engine* e, int cls, int64_t* launches, double* total_ms, double* flops_per_launch);

/* Bytes of device memory currently held by the engine (weights + workspace). */
int64_t st_device_bytes(const st_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* STABLETTS_HIP_H */
