/*
 * lcmi.h - C ABI of liblcmi.so, the MI355X (gfx950) implementation of lightcurver's
 * PSF-fit + joint forward-model hot path.
 *
 * The reference (duxfrederic/lightcurver) has no FFI for this path: it calls the third-party
 * STARRED/JAX package from Python.  Each entry point below therefore cites the reference *call
 * site* whose arithmetic it replaces (paths relative to the reference repository root); the
 * ctypes stub a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, int status (0 = ok, < 0 = error; message through
 *     lc_last_error()).  No torch / numpy types.
 *   - All array arguments are HOST pointers unless the name ends in _dev.  Inputs are copied at
 *     call time, outputs are written into caller-allocated buffers.  float = IEEE fp32, the
 *     arithmetic type of the path (the reference stores stamps as float32:
 *     lightcurver/processes/cutout_making.py:48-49).
 *   - One context per device, one thread per context.
 *   - Layouts are C-contiguous; images are [row = y][col = x]; c_x, dx, x0 run along columns.
 */
#ifndef LCMI_H
#define LCMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LC_OK 0
#define LC_ERR_INVALID (-1)
#define LC_ERR_DEVICE (-2)
#define LC_ERR_UNSUPPORTED (-3)
#define LC_ERR_NONFINITE (-4)

typedef struct lc_ctx lc_ctx;
typedef struct lc_psf_batch lc_psf_batch;
typedef struct lc_joint lc_joint;

/* ---- context ------------------------------------------------------------------------------ */
int lc_version(void);
int lc_ctx_create(int device, lc_ctx **out);
void lc_ctx_destroy(lc_ctx *ctx);
const char *lc_last_error(const lc_ctx *ctx); /* ctx may be NULL: last error of a failed create */
int lc_ctx_synchronize(lc_ctx *ctx);
/* the hipStream_t every kernel of this context is enqueued on (and its device ordinal), so that a caller can
 * order its own work -- e.g. the RCCL all-reduce of the joint fit's shared block -- without a host sync */
int lc_ctx_stream(lc_ctx *ctx, void **hip_stream, int *device);
/* HIP-event timer on the context's stream (the stream every kernel of this library runs on). */
int lc_timer_start(lc_ctx *ctx);
int lc_timer_stop(lc_ctx *ctx, float *elapsed_ms);
/* device name + CU count, for reports */
int lc_device_info(lc_ctx *ctx, char *name, int name_len, int *n_cu, int64_t *hbm_bytes);

/* measured device copy bandwidth (read + write bytes / s) of a float4 grid-stride copy of `bytes` bytes, `reps`
 * launches: the box's own HBM figure that bench.py reports beside the 8 TB/s specification */
int lc_copy_bandwidth(lc_ctx *ctx, int64_t bytes, int reps, float *gb_per_s);
/* Measurement aid: synchronises the context's stream, launches `lc_marker_kernel` on a grid of `tag` (1 .. 65535) one-wave
 * workgroups and synchronises again - a dispatch whose grid size identifies it in a rocprofv3 trace or counter CSV, so that
 * the rows of a section (the iterations of one workload) can be cut out between two markers (bench.py, roofline.traffic). */
int lc_ctx_marker(lc_ctx *ctx, int tag);

/* ---- stamp pre-processing --------------------------------------------
 * One fused pass over K stamps of npix pixels, replacing the host-side NumPy of the reference:
 *   noise map from the background rms when `noisemap` is NULL: max(sqrt((t rms)^2 + |t data|), 1e-7) / t
 *     (lightcurver/processes/cutout_making.py:43-51; rms[K] in e-/s, exptime[K] = t);
 *   data, noisemap /= coefficient[K] when given (roi_file_preparation.py:162-164);
 *   pixels that are NaN in both inputs: data = 0, noisemap = nan_noise (1.0 at psf_modelling.py:139-143,
 *     1e7 at roi_file_preparation.py:194-196 and star_photometry.py:309-311), counted as masked;
 *   bad[K][npix] (1 = flagged cosmic / bad pixel, may be NULL): masked; with noise_boost > 0 the noise map is
 *     multiplied by it at the flagged pixels (roi_file_preparation.py:201) or, boost_whole_stamp != 0, once over
 *     every stamp that holds a flagged pixel (star_photometry.py:316);
 *   masked_count[K]: masked pixels per stamp, for the 40 % cut of psf_modelling.py:144-153.
 * Outputs (any may be NULL): data_out, noisemap_out, weight_out = good / noisemap^2 (what lc_psf_batch_create
 * takes), masked_count; kernel_ms = device time of the kernel alone (HIP events). */
int lc_prepare_stamps(lc_ctx *ctx, int K, int npix, const float *data, const float *noisemap, const float *rms,
                      const float *exptime, const float *coefficient, const uint8_t *bad, float nan_noise,
                      float noise_boost, int boost_whole_stamp, float *data_out, float *noisemap_out,
                      float *weight_out, int32_t *masked_count, float *kernel_ms);

/* ---- optimiser settings shared by both fits ----------------------------------------------- */
/* optax.adabelief as driven by STARRED's Optimizer(method='adabelief'):
 * lightcurver/processes/star_photometry.py:113-122, roi_modelling.py:326-334. */
typedef struct {
  float init_learning_rate;
  int32_t schedule_learning_rate; /* 0/1: lr_t = lr0 * decay_rate^(t / transition_steps) */
  float decay_rate;               /* default 0.99 */
  int32_t transition_steps;       /* default 10 */
  float b1, b2, eps, eps_root;    /* defaults 0.9, 0.999, 1e-16, 1e-16 */
} lc_adabelief_cfg;
void lc_adabelief_defaults(lc_adabelief_cfg *cfg);

/* ---- PSF fit: replaces starred.procedures.psf_routines.build_psf ----------------------------
 * Reference call site: lightcurver/processes/psf_modelling.py:164-171 (one call per frame inside
 * the serial loop at :92).  Here a whole batch of F frames is fitted at once, one workgroup per
 * frame.  S_max stars per frame; frames with fewer usable stars (the 40 % mask filter at
 * psf_modelling.py:144-153) pass weight == 0 for the padding stamps.
 *
 *   data    [F][S_max][n][n]  stamps, already divided by the caller's normalisation
 *   weight  [F][S_max][n][n]  mask / sigma^2 (0 where masked, NaN or padding)
 *   model per star i of a frame:  a_i * D_ss[ G(x0_i, y0_i) (*) (Moffat + B) ] + sky_i
 */
int lc_psf_supported(int n, int ss); /* 1 if a kernel is instantiated for this stamp size */
int lc_psf_batch_create(lc_ctx *ctx, int F, int S_max, int n, int ss, const float *data,
                        const float *weight, lc_psf_batch **out);
void lc_psf_batch_destroy(lc_psf_batch *b);
/* Moffat parameters [F][4] = fwhm_x, fwhm_y (data px), phi (rad), beta; rasterises the unit-sum
 * Moffat of every frame on the N x N grid (N = ss * n). */
int lc_psf_batch_set_moffat(lc_psf_batch *b, const float *moffat);
int lc_psf_batch_get_moffat(lc_psf_batch *b, float *moffat);
/* per-star parameters [F][S_max][4] = a, x0, y0, sky */
int lc_psf_batch_set_stars(lc_psf_batch *b, const float *stars);
int lc_psf_batch_get_stars(lc_psf_batch *b, float *stars);
/* pixel grid B [F][N*N]; NULL resets it (and the optimiser moments) to zero */
int lc_psf_batch_set_grid(lc_psf_batch *b, const float *grid);
int lc_psf_batch_get_grid(lc_psf_batch *b, float *grid);
/* starlet weights W [F][J][N][N] (J = floor(log2 N) detail scales) and strengths.  W == NULL keeps the weights the
 * batch already holds: the starlet scale norms of a new batch ("lambda is not normalized" case of STARRED), or the
 * maps left by lc_psf_batch_propagate_noise / an earlier call with W != NULL. */
int lc_psf_batch_set_regularization(lc_psf_batch *b, const float *W, float lam_scales, float lam_hf);
/* noise propagation of the chi2 gradient into the starlet domain of B (replaces the
 * propagate_noise call inside build_psf), using the current a, x0, y0.  Writes device W. */
int lc_psf_batch_propagate_noise(lc_psf_batch *b);
int lc_psf_batch_get_weights(lc_psf_batch *b, float *W);
/* One evaluation at the current parameters.  Any output may be NULL.
 *   loss [F] (0.5 chi2 + l1), chi2 [F], grad_moffat [F][4], grad_stars [F][S_max][4] (a,x0,y0,sky),
 *   grad_grid [F][N*N] (d loss / d B, regularisation included), model [F][S_max][n][n]. */
/* (after lc_psf_batch_set_moffat_q, grad_moffat is d loss / d (q11, q12, q22, beta)) */
int lc_psf_batch_eval(lc_psf_batch *b, float *loss, float *chi2, float *grad_moffat,
                      float *grad_stars, float *grad_grid, float *model);
/* Stage A of build_psf: Moffat + a, x0, y0 by bounded L-BFGS with B = 0 (n_iter_analytic).  One independent problem per
 * frame, optimiser state and two-loop recursion on the device (csrc/psf_lbfgs.h); replaces STARRED's
 * Optimizer(method='l-bfgs-b') inside build_psf (reference call site lightcurver/processes/psf_modelling.py:164-171). */
int lc_psf_batch_fit_moffat(lc_psf_batch *b, int n_iter, float *final_loss /* [F] or NULL */);
/* Stage B: n_iter AdaBelief steps on B, a, x0, y0, state and loop on device (the hot loop).
 * Asynchronous on the context stream.  loss history accumulates across calls. */
int lc_psf_batch_run_adabelief(lc_psf_batch *b, int n_iter, const lc_adabelief_cfg *cfg);
int lc_psf_batch_iterations_done(lc_psf_batch *b);
/* Small batches run the loop with two workgroups per frame that hand their halves of the gradient to each other inside
 * the launch; a launch whose partner workgroups cannot all be resident gives up and is redone by the library itself in
 * the one-workgroup form from the pre-launch state (same bits).  count = how often that happened for this batch. */
int lc_psf_batch_split_fallbacks(lc_psf_batch *b, int *count);
/* loss at theta_0 .. theta_T (T + 1 values per frame; history[f][t] is the loss BEFORE update t,
 * the last entry the loss of the final parameters) */
int lc_psf_batch_get_loss_history(lc_psf_batch *b, float *history, int stride);
/* narrow_psf, full_psf [F][N][N] (unit sum), residuals = data - model [F][S_max][n][n],
 * reduced chi2 [F] over unmasked pixels. */
int lc_psf_batch_get_results(lc_psf_batch *b, float *narrow_psf, float *full_psf, float *residuals,
                             float *chi2);

/* ---- build_psf(field_distortion=True): lightcurver/processes/psf_modelling.py:164-171 with field_distortion and
 * stamp_coordinates.  Star i of a frame sees T_i = Moffat_i + W_i[B] (csrc/psf_distort.h).  Two
 * batches work together: `frames` (F frames; holds B, the starlet term and the AdaBelief state of B) and `stars` (F * S
 * single-star frames, frame-major; holds the stamps, each star's Moffat by its quadratic form, and the AdaBelief state of
 * a, x0, y0).  One iteration = forward (resample B for every star) -> step of `stars` with export_grad -> backward
 * (adjoint resampling summed over the stars) -> step of `frames` with use_ext_grad; everything asynchronous. */
int lc_psf_batch_set_moffat_q(lc_psf_batch *b, const float *q /* [F][4] = q11, q12, q22, beta */);
int lc_psf_batch_set_distortion(lc_psf_batch *frames, int S_stars, const float *coeffs /* [F][9] */,
                                const float *xy /* [F][S_stars][2] rescaled frame coordinates */);
int lc_psf_distortion_forward(lc_psf_batch *frames, lc_psf_batch *stars);
int lc_psf_distortion_backward(lc_psf_batch *frames, lc_psf_batch *stars);
int lc_psf_batch_get_ext_grad(lc_psf_batch *b, float *grad /* [F][N*N] */);
int lc_psf_batch_step_adabelief(lc_psf_batch *b, const lc_adabelief_cfg *cfg, int use_ext_grad, int export_grad);
/* the whole pixel-grid stage in one call: n_iter times { forward; step(stars, export_grad); backward; step(frames,
 * use_ext_grad) } and a final forward, enqueued from C++ */
int lc_psf_distortion_run(lc_psf_batch *frames, lc_psf_batch *stars, int n_iter, const lc_adabelief_cfg *cfg);
/* The batched bounded L-BFGS of the analytic stage with a caller-supplied evaluation (the distortion fit adds nine
 * coefficients per frame to the variables): x, lo, hi [nb][D]; eval fills F [nb] and G [nb][D] for the trial points X. */
int lc_batched_lbfgs(int nb, int D, double *x, const double *lo, const double *hi, int maxiter,
                     int (*eval)(void *user, const double *X, double *F, double *G), void *user,
                     double *f_final /* [nb] or NULL */, int *evaluations /* or NULL */);

/* ---- field distortion of the narrow PSF: replaces starred.psf.psf.apply_distortion ------------
 * Reference call sites: lightcurver/processes/star_photometry.py:291-304, roi_file_preparation.py:169-180
 * (narrow_psf, kwargs_distortion read back from the regions file, rescaled star position).
 *   narrow_psf [N][N]; coeffs [9] = dilation_x (c0, c1, c2), dilation_y (c0, c1, c2), shear (c0, c1, c2), each evaluated
 *   as c0 + c1 x + c2 y; xy [K][2] rescaled frame coordinates of the K positions; out [K][N][N], unit sum each.
 * Host buffers, copied at call time; synchronous. */
int lc_apply_distortion(lc_ctx *ctx, int N, int K, const float *narrow_psf, const float *coeffs,
                        const float *xy, float *out);

/* ---- joint multi-epoch forward model: replaces starred Deconv / Loss / Optimizer ------------
 * Reference call sites: lightcurver/processes/star_photometry.py:66-137 (one star, all epochs)
 * and lightcurver/processes/roi_modelling.py:213-334 (ROI, M point sources + background).
 *   data, sigma2 [E][n][n];  psf [E][N][N] narrow PSF per epoch;  N = ss * n
 * Parameter blocks follow STARRED's kwargs: a [E*M] epoch-major (roi_modelling.py:462),
 * c_x, c_y [M], dx, dy, alpha [E] (alpha in degrees, never free), h [N*N], mean [E].
 */
enum { LC_P_A = 0, LC_P_CX = 1, LC_P_CY = 2, LC_P_DX = 3, LC_P_DY = 4, LC_P_ALPHA = 5, LC_P_H = 6,
       LC_P_MEAN = 7, LC_P_COUNT = 8 };

typedef struct {
  float lam_scales, lam_hf;  /* regularization_strength_scales / _hf (l1_starlet on h) */
  float lam_positivity;      /* regularization_strength_positivity (h) */
  float lam_positivity_ps;   /* regularization_strength_positivity_ps (a) */
  float lam_pts_source;      /* regularization_strength_pts_source */
  float lam_flux_uniformity; /* regularization_strength_flux_uniformity */
  int32_t n_prior;           /* Gaussian prior terms on c_x / c_y: arrays below, length M each */
  const float *prior_cx_mean, *prior_cx_sigma, *prior_cy_mean, *prior_cy_sigma; /* may be NULL */
} lc_joint_loss_cfg;

int lc_joint_supported(int n, int ss);
/* Diagnostic: when on, objects created afterwards for n = 16 or 32 (ss = 2) use the large-grid kernels
   (spectrum scratch in HBM, multi-block starlet / update) that n = 128 always uses, so that those kernels
   can be checked against the oracle at sizes the oracle finishes in seconds. Not for production use. */
int lc_joint_set_debug_global(int on);
int lc_joint_create(lc_ctx *ctx, int E, int M, int n, int ss, const float *data, const float *sigma2,
                    const float *psf, lc_joint **out);
void lc_joint_destroy(lc_joint *j);
int lc_joint_set_param(lc_joint *j, int which, const float *values, int count);
int lc_joint_get_param(lc_joint *j, int which, float *values, int count);
/* free_mask[LC_P_COUNT]: 1 = optimised, 0 = fixed (ParametersDeconv kwargs_fixed) */
int lc_joint_set_free(lc_joint *j, const int32_t *free_mask);
int lc_joint_set_loss(lc_joint *j, const lc_joint_loss_cfg *cfg, const float *W /* [J][N][N] or NULL */);
/* propagate_noise(method='SLIT', likelihood_type='chi2')[0]  ->  W [J+1][N][N] */
int lc_joint_propagate_noise(lc_joint *j, float *W_out /* may be NULL: keep on device only */);
/* loss and gradient at the current parameters; grads[which] may be NULL.  For L-BFGS drivers. */
int lc_joint_loss_grad(lc_joint *j, float *loss, float *const grads[LC_P_COUNT]);
/* Deconv.model(kwargs) -> [E][n][n];  chi2_per_epoch [E] = sum res^2/sigma2 (not normalised) */
int lc_joint_model(lc_joint *j, float *model, float *chi2_per_epoch);
/* Deconv.getDeconvolved(kwargs, epoch) -> high-res scene and background, [N][N] each */
int lc_joint_deconvolved(lc_joint *j, int epoch, float *scene, float *background);
int lc_joint_run_adabelief(lc_joint *j, int n_iter, const lc_adabelief_cfg *cfg);
/* Optimizer(method='l-bfgs-b').minimize(maxiter=...) (roi_modelling.py:278-280, starred_utilities.py:33-34): bounded
 * L-BFGS on the free blocks with parameters, gradients, direction and (s, y) history on the device; the host reads a few
 * scalars per trial point.  lower / upper: per block, length of the block, or NULL (unbounded); may be NULL altogether.
 * loss_history[i] = loss after accepted iteration i (up to history_capacity entries). */
int lc_joint_run_lbfgs(lc_joint *j, int maxiter, const float *const lower[LC_P_COUNT],
                       const float *const upper[LC_P_COUNT], float *loss_history, int history_capacity,
                       int *n_iterations, int *n_evaluations);
int lc_joint_get_loss_history(lc_joint *j, float *history, int count);
int lc_joint_iterations_done(lc_joint *j);
/* Few epochs per GPU (a rank's share of a sharded fit; the reference keeps all epochs on one device,
 * lightcurver/processes/roi_modelling.py:154-160,213, and has no counterpart): inside lc_joint_run_adabelief /
 * lc_joint_run_sharded the epoch kernel of a 64 x 64 fit runs as a CLUSTER launch - several workgroups per epoch in one
 * launch, phases separated by arrival counters in device memory - when all of them are resident together.  Every wait is
 * bounded; a run in which one gave up is redone by the library with one workgroup per epoch (lc_joint_run_adabelief) or
 * reported (lc_joint_run_sharded), and the object keeps the one-workgroup kernel afterwards.  parts_last: workgroups per epoch
 * of the most recent epoch launch (0 = one-workgroup kernel); fallbacks: runs in which a wait gave up.  LCMI_CLUSTER=0
 * switches the form off, LCMI_CLUSTER=<P> forces the count. */
int lc_joint_cluster_info(lc_joint *j, int *parts_last, int *fallbacks);
/* Optimizer.minimize(..., return_param_history=True) - what the reference's own call sites pass
 * (lightcurver/processes/star_photometry.py:115-122, roi_modelling.py:326-334).  begin: from now on every AdaBelief update
 * also stores the free parameter blocks (in block order a, c_x, c_y, dx, dy, h, mean; *n_params = their total length) into
 * a device-resident history of `capacity` rows, so the loop of lc_joint_run_adabelief stays on the device; updates past
 * the capacity are not recorded.  get: rows [first, first + count) -> out [count][n_params] (one D2H copy, whenever the
 * caller first looks at the history).  end: releases the buffer (also done by begin, set_free and destroy). */
int lc_joint_param_history_begin(lc_joint *j, int capacity, int *n_params);
int lc_joint_param_history_rows(lc_joint *j);
int lc_joint_param_history_get(lc_joint *j, int first, int count, float *out);
int lc_joint_param_history_end(lc_joint *j);
/* Batched star photometry.  The reference fits its reference stars one after the other (the loop at
 * lightcurver/processes/star_photometry.py:257 over do_one_star_forward_modelling, :23-151): G independent joint fits
 * without a background, each over its own epochs.  Here they are ONE object: star g owns epochs_per_group[g] consecutive
 * epochs of data / sigma2 / psf ([sum E_g][..]), M point sources with its own c_x, c_y (parameter blocks c_x, c_y have
 * G * M entries, star-major; a, dx, dy, mean follow the epochs), and every AdaBelief iteration is one kernel pair for all
 * stars (grid over (star, epoch), then one block per star for its reduction and update).  Each star's trajectory is bit
 * for bit that of lc_joint_create + lc_joint_run_adabelief on that star alone.  h and alpha stay fixed; no weight cube,
 * prior or point-source starlet term.  set/get_param, set_free, set_loss, run_adabelief, model, fisher_flux_sigma,
 * deconvolved (epoch counted over the epochs of all stars; the positions of its star) and iterations_done work as on a
 * plain object; the loss history is per star. */
int lc_joint_create_groups(lc_ctx *ctx, int G, const int32_t *epochs_per_group, int M, int n, int ss, const float *data,
                           const float *sigma2, const float *psf, lc_joint **out);
int lc_joint_get_group_loss_history(lc_joint *j, float *history /* [G][count_per_group] */, int count_per_group);
/* FisherCovariance(diagonal_only=True) with only `a` free -> sigma(a) [E*M]
 * (lightcurver/utilities/starred_utilities.py:36-38). */
int lc_joint_fisher_flux_sigma(lc_joint *j, float *sigma_a);
/* Multi-GPU epoch sharding: split one optimiser step around the caller's all-reduce of the shared block
 * [dL/dh (N*N) | dL/dc_x (M) | dL/dc_y (M) | sum_e (a - ref) (M) | sum_e (a - ref)^2 (M) | chi2 | n_epochs]
 * living in device memory.  The flux moments (flux-uniformity term, jnp.std over epochs in the reference:
 * roi_modelling.py:273-276) are centred on one reference flux per source so that the variance does not cancel in
 * fp32; lc_joint_set_param(LC_P_A) sets it to the mean of the local fluxes, a sharded fit must give every rank the
 * same reference (lightcurver_amd/distributed.py does) with lc_joint_set_flux_reference. */
int lc_joint_set_flux_reference(lc_joint *j, const float *ref, int count /* = M */);
int lc_joint_get_flux_reference(lc_joint *j, float *ref, int count /* = M */);
int lc_joint_step_local(lc_joint *j);                      /* forward/backward of local epochs */
int lc_joint_shared_buffer_dev(lc_joint *j, void **dev_ptr, int *count);
int lc_joint_step_update(lc_joint *j, const lc_adabelief_cfg *cfg); /* regularise + AdaBelief */
/* The gradient-only counterpart of lc_joint_step_update (the sharded L-BFGS stage: roi_modelling.py:278-280 with the epochs
 * spread over ranks): behind lc_joint_step_local and the all-reduce of the shared block it returns the loss of the WHOLE fit
 * and the gradients - those of the shared parameters complete, those of the per-epoch parameters for the local epochs -
 * without stepping anything.  grads as in lc_joint_loss_grad. */
int lc_joint_step_grad(lc_joint *j, float *loss, float *const grads[LC_P_COUNT]);
/* host-staged access to the shared block (gloo / CPU collectives: D2H, all-reduce, H2D) */
int lc_joint_shared_get(lc_joint *j, float *host, int count);
int lc_joint_shared_set(lc_joint *j, const float *host, int count);
/* The sharded loop without the host language in it: n_iter times { lc_joint_step_local; allreduce(user, block, count,
 * stream); lc_joint_step_update }, enqueued from C++.  `allreduce` must sum-all-reduce the `count` floats at `dev_buf`
 * over the ranks in place, ordered on `hip_stream` (= lc_ctx_stream: e.g. ncclAllReduce enqueued there, or
 * lc_peer_allreduce below with user = the peer group), and return 0.  Every rank must have agreed on the flux reference
 * (lc_joint_set_flux_reference) beforehand. */
typedef int (*lc_allreduce_fn)(void *user, void *dev_buf, int count, void *hip_stream);
int lc_joint_run_sharded(lc_joint *j, int n_iter, const lc_adabelief_cfg *cfg, lc_allreduce_fn allreduce, void *user);

/* ---- peer group: one-shot all-reduce of the shared block through peer memory (xGMI, one hop) -------------------------------
 * The block is 64 - 256 KiB and latency bound: every rank publishes it in an exchange buffer that the others map with HIP
 * IPC, reads the N - 1 peers directly and adds in rank order (the same bits on every rank).  csrc/peer.hip.
 *   create   count = lc_joint_shared_buffer_dev's count; rank / world of this process
 *   export   this rank's IPC handle (64 bytes used of handle_bytes) - the caller carries the handles between the processes
 *            (e.g. torch.distributed.all_gather_object)
 *   connect  handles [world][handle_bytes], own entry ignored
 *   lc_peer_allreduce   has the signature of lc_allreduce_fn with user = the group; all ranks must call it equally often
 *   status   LC_ERR_DEVICE if a rank did not show up within ~2 s in some call (the kernels never wait longer) */
typedef struct lc_peer_group lc_peer_group;
int lc_peer_group_create(lc_ctx *ctx, int count, int rank, int world, lc_peer_group **out);
int lc_peer_group_export(lc_peer_group *g, void *handle_out, int handle_bytes);
int lc_peer_group_connect(lc_peer_group *g, const void *handles, int handle_bytes);
int lc_peer_allreduce(void *group, void *dev_buf, int count, void *hip_stream);
int lc_peer_group_status(lc_peer_group *g);
void lc_peer_group_destroy(lc_peer_group *g);

/* RCCL group: the same all-reduce by RCCL, from the library's own loop ("RCCL all-reduce over xGMI only for the
 * shared-background gradient in the joint fit", BASELINE.json north_star; the reference keeps all epochs on one device,
 * lightcurver/processes/roi_modelling.py:154-160,213, and has no counterpart).  The library loads librccl at run time (it
 * does not link against it; a copy the process already carries - torch's - is shared) and owns the communicator:
 *   lc_rccl_available   1 when librccl could be loaded
 *   lc_rccl_unique_id   rank 0: a fresh ncclUniqueId (id_bytes >= 128), which the caller hands to every rank over any
 *                       channel it has (a torch.distributed broadcast, MPI, a file)
 *   lc_rccl_group_create  every rank, collectively: ncclCommInitRank on the context's device
 *   lc_rccl_allreduce   has the signature of lc_allreduce_fn with user = the group: ncclAllReduce (float32, sum) in place
 *                       on the given stream, enqueued, never synchronised - pass it to lc_joint_run_sharded and no host
 *                       code runs inside the loop
 *   lc_rccl_group_info  rank, size, all-reduces enqueued so far (each may be NULL) */
typedef struct lc_rccl_group lc_rccl_group;
int lc_rccl_available(void);
int lc_rccl_unique_id(void *id_out, int id_bytes);
int lc_rccl_group_create(lc_ctx *ctx, const void *unique_id, int id_bytes, int rank, int world, lc_rccl_group **out);
int lc_rccl_allreduce(void *group, void *dev_buf, int count, void *hip_stream);
int lc_rccl_group_info(lc_rccl_group *g, int *rank, int *world, long long *calls);
void lc_rccl_group_destroy(lc_rccl_group *g);

#ifdef __cplusplus
}
#endif
#endif /* LCMI_H */
