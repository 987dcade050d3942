/*
 * irsgmcmc.h -- C ABI of the MI355X-native SG-MCMC registration inner loop.
 *
 * Drop-in boundary for ONE path of dgrzech/ir-sgmcmc: `Trainer._SGLD_transition`
 * (reference trainer/trainer.py:291-356) and the operators it is built from.  The reference has no
 * FFI of its own -- its extension point is name-based construction of Python classes from JSON
 * (parse_config.py:251-266) -- so every entry point below cites the reference call site it replaces.
 * Python binds this header with ctypes (ir_sgmcmc_amd/_lib.py); see INTEGRATION.md.
 *
 * Conventions
 *  - every `float*` / `uint8_t*` argument is a CALLER-OWNED DEVICE pointer (e.g. torch allocation);
 *    fields are planar fp32 (C, 3, D, H, W), images (C, 1, D, H, W), channel 0 = x = last axis
 *    (utils/util.py:263-278); masks are 1 byte per voxel (torch.bool).
 *  - `stream` is a hipStream_t passed as void*; all work is asynchronous on it, nothing synchronises
 *    except the functions documented as "blocking".
 *  - every function returns 0 on success, non-zero on error (irs_last_error() gives the message);
 *    nothing aborts or throws across the boundary; no internal threads.
 *  - a context is not thread-safe; distinct contexts / streams are independent.
 */
#ifndef IRSGMCMC_H
#define IRSGMCMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: these are its only exports */

#define IRS_MAX_COMPONENTS 8
#define IRS_MAX_CHAINS 8
#define IRS_MAX_HALF_WIDTH 4 /* Sobolev / LCC half widths up to 4 */

enum { IRS_DATA_GMM_LCC = 0, IRS_DATA_SSD = 1 };
enum { IRS_REG_L2 = 0, IRS_REG_LOGNORMAL = 1, IRS_REG_STUDENT = 2, IRS_REG_LOGNORMAL_L2 = 3 };

/* ------------------------------------------------------------------------------------------------
 * stateless operators (unit-parity surface; the Python modules SVF_3D, RegistrationModule, GMM.map,
 * GradientOperator ... call these)
 * ---------------------------------------------------------------------------------------------- */

/* SGLD.forward + SobolevGrad.forward: out = S * (v + sqrt(2 tau) sigma eps)
 * (utils/functions.py:76-84,98-109; utils/util.py:48-58,394-404).
 * v, out, tmp: (C,3,D,H,W).  sigma: NULL (=1) or (C,3,D,H,W).  eps: standard-normal tensor, or NULL ->
 * in-kernel Philox4x32-10 keyed by (seed, iteration), one call per voxel pair (z even, z + 1): six 21-bit words ->
 * three Box-Muller pairs.  kernel: 2s+1 host floats; s == 0 -> no smoothing.
 * tau < 0 -> no noise.  `tmp` may not alias v or out; out may not alias v. */
int irs_perturb_smooth(const float* v, const float* sigma, const float* eps, float tau, const float* kernel, int s,
                       int C, int D, int H, int W, float* tmp, float* out, uint64_t seed, uint64_t iteration,
                       void* stream);

/* SVF_3D.forward (utils/transformation.py:63-76): scaling and squaring.
 * v: (C,3,D,H,W) voxel units.  steps: (no_steps, C,3,D,H,W) workspace receiving d_1..d_no_steps in
 * normalised units (kept for the backward).  transformation / displacement: outputs, either may be NULL. */
int irs_svf_exp_fwd(const float* v, float* steps, float* transformation, float* displacement, int no_steps, int C,
                    int D, int H, int W, void* stream);

/* adjoint of the above w.r.t. v (what autograd does through the 12 grid_sample calls).
 * g_last: gradient w.r.t. d_no_steps (normalised units), (C,3,D,H,W).  g_v: output, gradient w.r.t. v.
 * scratch: 2 x (C,3,D,H,W).  Owner-computes gather for max|d_k| < 2 voxels, fixed-point LDS accumulation above: no float
 * atomics, bitwise reproducible. */
int irs_svf_exp_bwd(const float* v, const float* steps, const float* g_last, float* scratch, float* g_v, int no_steps,
                    int C, int D, int H, int W, void* stream);

/* Cubic_B_spline_FFD_3D.forward (utils/transformation.py:126-153) and its adjoint.
 * v_cp: (C,3,G0,G1,G2), G = ceil((N-1)/cps)+3; dense: (C,3,D,H,W); tmp: 2 x (C,3,D,H,W) scratch. */
int irs_ffd_up(const float* v_cp, float* dense, float* tmp, int C, int D, int H, int W, int c0, int c1, int c2,
               void* stream);
int irs_ffd_adjoint(const float* g_dense, float* g_cp, float* tmp, int C, int D, int H, int W, int c0, int c1, int c2,
                    void* stream);

/* RegistrationModule.forward, float path (utils/registration.py:29-30) with the optional uniform grid
 * jitter of utils/util.py:44-53 folded in.  im: (Cim,1,D,H,W) with Cim in {1, C} (1 = shared by all chains).
 * d_last: d_no_steps in normalised units (C,3,D,H,W).  unif: U[0,1) tensor (C,3,D,H,W), or NULL -> Philox when
 * alpha > 0.  alpha <= 0 -> no jitter. */
int irs_warp_fwd(const float* im, int Cim, const float* d_last, const float* unif, float alpha, float* warped, int C,
                 int D, int H, int W, uint64_t seed, uint64_t iteration, void* stream);
/* grid-gradient of the warp: g_d = dL/d(d_last) given g_warped (C,1,D,H,W). */
int irs_warp_bwd(const float* im, int Cim, const float* d_last, const float* unif, float alpha, const float* g_warped,
                 float* g_d, int C, int D, int H, int W, uint64_t seed, uint64_t iteration, void* stream);
/* same warp applied to an arbitrary transformation tensor in [-1,1] (the public RegistrationModule call) */
int irs_warp_transformation(const float* im, int Cim, const float* transformation, float* warped, int C, int D, int H,
                            int W, void* stream);
/* nearest-neighbour path for masks (uint8) and label maps (int16) (utils/registration.py:20-27) */
int irs_warp_nearest_u8(const uint8_t* seg, int Cim, const float* transformation, uint8_t* out, int C, int D, int H,
                        int W, void* stream);
int irs_warp_nearest_i16(const int16_t* seg, int Cim, const float* transformation, int16_t* out, int C, int D, int H,
                         int W, void* stream);

/* LCC normalisation (I - u) / sqrt(var + 1e-10) of model/loss.py:103-109; sigma_out may be NULL. */
int irs_lcc_normalise(const float* im, float* out, float* sigma_out, int s, int C, int D, int H, int W, void* stream);
/* GMM.map (model/loss.py:102-111) with the fixed side pre-normalised: z = fhat - LCC(warped).
 * fhat: (Cf,1,D,H,W), Cf in {1,C}.  sigma_m: output, local std of the warped image (kept for the adjoint). */
int irs_lcc_map_fwd(const float* fhat, int Cf, const float* warped, float* z, float* sigma_m, int s, int C, int D,
                    int H, int W, void* stream);
/* adjoint of GMM.map w.r.t. the warped image given g_z. */
int irs_lcc_map_bwd(const float* fhat, int Cf, const float* z, const float* sigma_m, const float* g_z, float* g_warped,
                    int s, int C, int D, int H, int W, void* stream);

/* GradientOperator + energy (utils/diff_op.py:78-96, model/loss.py:158-159): y[c] = sum (fwd diff)^2.
 * y_out: C doubles on the device.  partials: scratch of irs_reduce_scratch_doubles() doubles. */
int irs_reg_energy(const float* v, double* y_out, double* partials, int C, int D, int H, int W, void* stream);
size_t irs_reduce_scratch_doubles(void);
/* GradientOperator.forward: nabla (C,3,D,H,W,3) as in utils/diff_op.py:92-96; transformation != 0 divides by the
 * pixel spacing 2/(N-1). */
int irs_gradient_operator(const float* v, float* nabla, int transformation, int C, int D, int H, int W, void* stream);
/* calc_det_J of GradientOperator(transformation=True) + NaN count of its log (utils/util.py:72-91,209-212).
 * log_det: (C,1,D,H,W) or NULL; nan_count: C int64 on the device (zeroed by the call). */
int irs_log_det_jacobian(const float* transformation, float* log_det, long long* nan_count, int C, int D, int H, int W,
                         void* stream);

/* ------------------------------------------------------------------------------------------------
 * fused transition (Trainer._SGLD_transition, trainer/trainer.py:291-356)
 * ---------------------------------------------------------------------------------------------- */

typedef struct irs_config {
    int32_t dims[3];        /* D, H, W of the images */
    int32_t cps[3];         /* control point spacing; all 0 -> SVF_3D, else SVFFD_3D */
    int32_t no_chains;      /* C */
    int32_t no_steps;       /* 12 */
    int32_t sobolev_s;      /* 0 -> disabled */
    float sobolev_kernel[2 * IRS_MAX_HALF_WIDTH + 1];
    float lr;               /* optimizer_SG_MCMC lr == tau (trainer.py:607) */
    float uniform_alpha;    /* <= 0 -> disabled */
    int32_t virtual_decimation;
    int32_t data_loss;      /* IRS_DATA_* */
    int32_t lcc_s;
    int32_t gmm_components;
    float ssd_sigma;
    /* optimizer_GMM (optimizers/adam_rate_decay.py) + hyper-priors (trainer.py:68-77) */
    float gmm_lr_log_std, gmm_lr_logits, gmm_lr_decay;
    float adam_beta1, adam_beta2, adam_eps;
    float scale_prior_loc, scale_prior_scale;             /* LogScaleNormalPrior */
    float dirichlet_concentration[IRS_MAX_COMPONENTS];    /* DirichletPrior */
    /* regulariser (model/loss.py:172-312) */
    int32_t reg_loss;       /* IRS_REG_* */
    int32_t reg_learnable;
    float w_reg;
    double dof;             /* 3 * prod(image dims), parse_config.py:120,128 */
    float reg_lr0, reg_lr1, reg_lr_decay; /* (lr_loc, lr_log_scale) or (lr_log_w_reg, -) */
    float loc_prior_nu, loc_prior_w_reg;  /* LogEnergyExpGammaPrior */
    float reg_scale_prior_loc, reg_scale_prior_scale; /* LogScaleNormalPrior on log_scale */
    double w_reg_prior_shape, w_reg_prior_rate;       /* LogPrecisionExpGammaPrior (learnable RegLoss_L2);
                                                         RegLoss_Student (model/loss.py:201-241): {a0, 2 b0} */
    uint64_t seed;          /* Philox key for in-kernel noise */
} irs_config;

/* hyper-parameter + optimiser state that lives on the device between transitions */
typedef struct irs_state {
    float gmm_log_std[IRS_MAX_COMPONENTS];
    float gmm_logits[IRS_MAX_COMPONENTS];
    double gmm_adam_m[2][IRS_MAX_COMPONENTS];
    double gmm_adam_v[2][IRS_MAX_COMPONENTS];
    int64_t gmm_adam_step[2];
    double reg_param[2];     /* L2: {log_w_reg, -}; LogNormal: {loc, log_scale} */
    double reg_adam_m[2], reg_adam_v[2];
    int64_t reg_adam_step[2];
    uint64_t iteration;      /* Philox counter, incremented by every transition */
} irs_state;

/* per-transition scalars (loss_terms / aux of the reference's return triple) */
typedef struct irs_scalars {
    double alpha[IRS_MAX_CHAINS];       /* VD factor */
    double data_term[IRS_MAX_CHAINS];   /* alpha * sum(-log p(z)) */
    double reg_term[IRS_MAX_CHAINS];
    double reg_energy[IRS_MAX_CHAINS];  /* y */
    double n_mask[IRS_MAX_CHAINS];
} irs_scalars;

typedef struct irs_io {
    const float* fixed_im;   /* (Cf,1,D,H,W) */
    const float* moving_im;  /* (Cm,1,D,H,W) */
    const uint8_t* mask;     /* (Cmask,1,D,H,W) */
    int32_t fixed_chains, moving_chains, mask_chains; /* 1 (shared) or C */
    float* v;                /* (C,3,Dv,Hv,Wv) in/out: v_curr_state */
    const float* sigma;      /* (C,3,Dv,Hv,Wv) or NULL (= 1) */
    const float* eps;        /* injected N(0,1) noise or NULL (Philox) */
    const float* unif;       /* injected U[0,1) noise (C,3,D,H,W) or NULL (Philox) */
    /* outputs (caller-owned; any may be NULL to skip) */
    float* curr_state;       /* (C,3,Dv,Hv,Wv) smoothed velocity = the recorded sample */
    float* im_moving_warped; /* (C,1,D,H,W) */
    float* residuals;        /* (C,1,D,H,W) dense z */
    float* displacement;     /* (C,3,D,H,W) voxels */
    float* transformation;   /* (C,3,D,H,W) [-1,1] */
    float* grad_v;           /* (C,3,Dv,Hv,Wv) v.grad of the reference: sigma^2 * dL/dv_s (debug / parity) */
} irs_io;

typedef struct irs_ctx irs_ctx;

/* allocates the workspace (saved scaling-and-squaring steps, gradients, tile scratch) with hipMalloc. blocking. */
int irs_create(const irs_config* cfg, irs_ctx** out);
void irs_destroy(irs_ctx* ctx);
size_t irs_workspace_bytes(const irs_ctx* ctx);
/* control-grid size for the configured transformation model (utils/util.py:61-69) */
int irs_velocity_dims(const irs_ctx* ctx, int32_t out[3]);

/* pre-normalise the fixed image for the LCC map (iteration-invariant half of model/loss.py:103-105). */
int irs_set_fixed(irs_ctx* ctx, const float* fixed_im, int fixed_chains, void* stream);
/* blocking copies of the small state / scalars.  (A slab context whose transport has failed -- a peer gone, irs_slab_transition /
 * irs_flush return that error -- still answers these two: the device holds the state after the last good transition.) */
int irs_get_state(irs_ctx* ctx, irs_state* out, void* stream);
int irs_set_state(irs_ctx* ctx, const irs_state* in, void* stream);
int irs_get_scalars(irs_ctx* ctx, irs_scalars* out, void* stream);

/* Trainer.__GMM_init (trainer/trainer.py:529-547): init log_std from std(z[mask]) at the given velocity sample
 * (NULL = zero field), then `warm_up` _step_GMM iterations.  blocking (reads one scalar back). */
int irs_gmm_init(irs_ctx* ctx, const irs_io* io, const float* v_sample, int warm_up, void* stream);

/* one SG-MCMC transition.  Asynchronous: nothing is allocated and the DEVICE is never waited for; the host, however, is
 * held back so that it runs at most two transitions ahead of the device (it waits on the end event of the transition before
 * the previous one -- the kernel-variant prediction reads bounds no older than that).  Under stream capture that wait is
 * skipped, so the call stays graph-capturable.  Which variants of the squaring-step kernels get launched is predicted from
 * the displacement bounds of earlier transitions; the device checks the prediction, and a transition that was launched
 * without a variant it turned out to need leaves everything untouched and is re-run by a later call (irs_flush).
 * Consequences for a caller that reads per-call results: the output arrays of irs_io (and grad_v, and the timings of
 * irs_transition_timed) belong to transition t only once irs_flush has returned -- until then a dropped transition leaves them
 * holding an earlier sample.  A re-run draws the in-kernel Philox noise of the transition it repeats (the counter did not
 * advance), but INJECTED eps / unif are taken from the irs_io of the call that performs the re-run: parity runs with injected
 * noise set predict_variants = 0.  Under stream capture no prediction is made and nothing is re-run (the captured sequence
 * launches every variant), so a replayed graph never depends on host state. */
int irs_transition(irs_ctx* ctx, const irs_io* io, void* stream);

/* Wait for everything enqueued on `stream` and complete the chain: a transition whose kernel-variant (or, on a slab, ghost-width)
 * prediction turned out wrong is a NO-OP on the device -- no parameter, optimiser moment, Philox counter or velocity changes --
 * and is re-run without predictions by the next irs_transition / irs_slab_transition call; this call re-runs the ones the last
 * calls left behind (with the irs_io of the most recent call, whose buffers must still be alive).  irs_get_state and
 * irs_get_scalars call it.  After it, v / state / scalars are those of a chain that never mispredicted.  blocking; collective
 * on a slab context (every rank reaches the same verdicts at the same call). */
int irs_flush(irs_ctx* ctx, void* stream);
/* transitions re-run so far because a prediction failed (statistics) */
int irs_recovered_transitions(const irs_ctx* ctx, uint64_t* out);

/* timing hook for bench.py: the same transition with hipEvents recorded on `stream` around the stages; blocking.
 * All times in milliseconds for THIS call. */
typedef struct irs_timings {
    float total_ms;            /* whole transition */
    float exp_fwd_ms;          /* the no_steps scaling-and-squaring forward launches */
    float exp_bwd_kernel_ms;   /* sum over the no_steps radius-1 adjoint KERNEL launches (events around each launch) */
    float exp_bwd_total_ms;    /* adjoint loop including the gradient-buffer memsets */
    float smooth_ms;           /* perturbation + Sobolev smoothing (+ FFD up-sampling) */
    float data_ms;             /* warp, LCC map, statistics, GMM step, data term + adjoints, warp backward */
    float update_ms;           /* gradient assembly + SGLD update + bookkeeping */
    float exp_bwd_primary_avg_ms; /* mean launch duration of the dominant kernel alone: the radius-1 adjoint step without
                                     input prescale (steps 1 .. no_steps-1), events around exactly that launch -- the
                                     number rocprofv3 reports for exp_bwd_march_kernel<false,1> */
} irs_timings;
int irs_transition_timed(irs_ctx* ctx, const irs_io* io, void* stream, irs_timings* out);

/* ------------------------------------------------------------------------------------------------
 * z-slab decomposition INSIDE the library (BASELINE.json config 4): one chain, the volume split
 * along z over the ranks of a node, one process per GPU.  What is sharded is the single-device loop body
 * trainer/trainer.py:291-356 (the reference has no multi-device code, base/base_trainer.py:16).
 *
 * Rank r owns the planes [a, b) = [r D / n, (r + 1) D / n) and HOLDS [lo, hi) = [a - margin, b + margin) clipped to the
 * volume: every array of the context and every array of irs_io -- except the moving image -- is slab-local,
 * (C, ch, hi - lo, H, W), so memory per rank falls with the number of ranks.  The moving image is static and is given
 * whole (67 MB at 256^3): the warp may then sample it at any displacement without traffic.
 *
 * irs_slab_transition runs the same kernels as irs_transition on windows of the slab and moves ghost planes between
 * neighbouring ranks with the communicator (RCCL ncclSend / ncclRecv groups, or producer-side stores into peer-mapped landing areas, on a
 * communication stream of its own; three small all-reduces per transition -- displacement bounds, statistics per chain, data-term
 * sums together with the regulariser energies) -- no host synchronisation, no Python between the stages:
 *   - stencil halos of fixed width (Sobolev s, LCC 4 s, update 1);
 *   - gather halos of the squaring steps, whose width floor(max|d_k|) + 1 follows the displacement: planned on the host from
 *     the all-reduced bounds of an earlier transition (never waited for), validated on the device afterwards
 *     (irs_slab_status_get); the first transition measures them step by step ("exact" mode, blocking);
 *   - forward squaring steps are grouped into communication-avoiding blocks (one exchange of up to `ghost_max` planes,
 *     then several steps on shrinking windows); every step around an exchange is split into its interior (launched
 *     while the exchange is in flight) and its two boundary strips (launched when the ghost planes have arrived);
 *   - the adjoint is an owner-computes gather, so the backward pass also only RECEIVES ghost planes (of the incoming
 *     gradient): no reverse halo accumulation.
 * With one rank the schedule degenerates to the single-GPU launch sequence.
 * SVFFD_3D: the control grid is small, so v / sigma / eps / curr_state / grad_v stay WHOLE (control-grid sized, replicated on
 * every rank: same Philox noise, same smoothing, same update); the dense velocity is up-sampled on the planes a rank needs,
 * and the adjoint of the up-sampling sums over a rank's own planes, one more all-reduce (floats) making the control-grid
 * gradient whole.
 * ---------------------------------------------------------------------------------------------- */
typedef struct irs_comm irs_comm;
#define IRS_COMM_ID_BYTES 128
/* rank 0: ncclGetUniqueId; the caller distributes the bytes to the other ranks (any channel). */
int irs_comm_unique_id(uint8_t id[IRS_COMM_ID_BYTES]);
/* ncclCommInitRank on the CURRENT device; collective over the `world` ranks, blocking. */
int irs_comm_create_rccl(const uint8_t id[IRS_COMM_ID_BYTES], int rank, int world, irs_comm** out);
/* Peer-mapped transport: every rank exports ONE device allocation (its landing area for ghost planes and all-reduce
 * contributions) with hipIpcGetMemHandle and maps its peers' (hipIpcOpenMemHandle); an exchange is a kernel of the PRODUCER storing
 * its strips into the consumer's landing area (xGMI stores on a node), ordered across processes by sequence flags -- no host or
 * stream synchronisation, no rendezvous.  `name` names a POSIX shared-memory segment through which the handles (and, by default,
 * the flags) travel: the same string on every rank, distributed by the caller like the id above; rank 0 creates the segment and
 * unlinks the name once every rank has attached.  Ranks may share a device (several processes on one GPU: how the asynchronous
 * schedule is exercised on a one-GPU box) or own one each (world <= 8, one node).  Collective, blocking.  Environment:
 * IRS_IPC_SLOT_MB (8) sizes the first landing area (MiB per neighbour and slot; it grows when a context needs more);
 * IRS_IPC_TIMEOUT_S (20) bounds every wait of a kernel for a peer -- a rank that waits longer raises an error that the next
 * irs_slab_transition / irs_flush returns. */
int irs_comm_create_ipc(const char* name, int rank, int world, irs_comm** out);
/* The same two transport operations as caller-supplied functions: rehearsal of the schedule with several ranks sharing ONE
 * GPU, which RCCL refuses (tests).  A callback must leave the data in place when it returns or enqueue its work on `stream`. */
typedef struct irs_xfer {
    void* ptr;      /* device pointer */
    size_t bytes;
    int32_t peer;   /* rank */
    int32_t recv;   /* 0 = send, 1 = receive */
} irs_xfer;
typedef int (*irs_exchange_fn)(void* user, const irs_xfer* xfers, int n, void* stream);
typedef int (*irs_allreduce_fn)(void* user, void* buf, size_t count, int kind, void* stream); /* 0 SUM f64 | 1 MAX u32 | 2 SUM f32 */
int irs_comm_create_callbacks(irs_exchange_fn ex, irs_allreduce_fn ar, void* user, int rank, int world, irs_comm** out);
void irs_comm_destroy(irs_comm* comm);
/* timing hook: `iters` neighbour exchanges of `bytes` (a multiple of 16) per direction and link, then `iters` all-reduces of
 * `ar_doubles` doubles, back to back; usec[0] / usec[1] = host-timed microseconds per exchange / per all-reduce. collective, blocking. */
int irs_comm_probe(irs_comm* comm, size_t bytes, size_t ar_doubles, int iters, void* stream, double usec[2]);
/* one line about the transport (kind, ranks; ipc: size and kind of the landing area, traffic so far) for logs */
int irs_comm_describe(const irs_comm* comm, char* out, size_t n);
int irs_comm_rank(const irs_comm* comm);
int irs_comm_world(const irs_comm* comm);
/* one all-reduce of each kind and one ring exchange on scratch memory, verified on the host. blocking, collective. */
int irs_comm_selftest(irs_comm* comm, void* stream);

typedef struct irs_slab_config {
    int32_t ghost_max;  /* widest ghost zone of one exchange, planes (0 -> 8); lowered to what the thinnest slab can send a
                         * neighbour (slab planes - sobolev_s): irs_slab_layout.ghost_max says what is in force */
    int32_t margin;     /* ghost planes held beyond a neighbour-facing edge (0 -> derived from the stencil widths) */
} irs_slab_config;
typedef struct irs_slab_layout {
    int32_t rank, world;
    int32_t a, b;       /* owned planes */
    int32_t lo, hi;     /* held planes: every slab-local array is (C, ch, hi - lo, H, W) */
    int32_t margin, ghost_max;
} irs_slab_layout;
/* planes a rank would own / hold, without creating anything (pure host arithmetic; what the caller cuts its arrays with) */
int irs_slab_plan_layout(const irs_config* cfg, const irs_slab_config* scfg, int rank, int world, irs_slab_layout* out);
/* context with slab-local workspace; `comm` may be NULL when world == 1.  The communicator is not owned. blocking. */
int irs_slab_create(const irs_config* cfg, const irs_slab_config* scfg, irs_comm* comm, irs_ctx** out);
int irs_slab_get_layout(const irs_ctx* ctx, irs_slab_layout* out);
/* one transition of the slab; irs_io arrays are slab-local except moving_im (whole volume, moving_chains in {1, C}).
 * curr_state / im_moving_warped / residuals outputs are valid on the owned planes.  asynchronous (after the first call). */
int irs_slab_transition(irs_ctx* ctx, const irs_io* io, void* stream);
/* Trainer.__GMM_init on the slabs (v_sample slab-local or NULL); collective, blocking. */
int irs_slab_gmm_init(irs_ctx* ctx, const irs_io* io, const float* v_sample, int warm_up, void* stream);
typedef struct irs_slab_status {
    uint64_t transitions, exact_transitions; /* total / run in exact (measuring, blocking) mode */
    uint64_t exchanges, exchanged_bytes;     /* point-to-point rounds / bytes sent by this rank */
    uint64_t mispredictions;                 /* transitions whose planned ghost widths turned out too narrow (results invalid) */
    int32_t last_fwd_rounds, last_bwd_rounds;/* exchange rounds of the squaring steps in the last transition */
} irs_slab_status;
/* blocking (waits for the enqueued transitions). A non-zero `mispredictions` is also returned as an error by the next
 * irs_slab_transition. */
int irs_slab_status_get(irs_ctx* ctx, irs_slab_status* out, void* stream);
/* Hand-over TIMELINE of one slab transition: what a first run on a node needs in order to tell a slow transport from load imbalance
 * from a serialised interior / boundary split (the reference has no counterpart: base/base_trainer.py:16 is single-device).
 * irs_slab_timeline_arm(ctx, t): the next `t` transitions record timing events around every exchange and all-reduce (a few
 * hipEventRecord per hand-over: off the timed path -- bench.py arms it on its trial transitions); the LAST armed one is kept.
 * irs_slab_timeline_get: blocking; entries in schedule order, *n_entries = how many there were (also when max_entries is smaller),
 * *total_us = first launch .. last launch of that transition on this rank. */
typedef struct irs_slab_timeline_entry {
    int32_t kind;       /* IRS_OP_EXCHANGE | IRS_OP_ALLREDUCE */
    int32_t stage;      /* exchange: the buffer (IRS_SB_*); all-reduce: IRS_AR_* */
    int32_t k, width;   /* exchange: adjoint step that consumes it (-1: none), ghost planes per side */
    float ready_us;     /* since the transition's first launch: the data to hand over was ready on the compute stream (event P) */
    float handover_us;  /* P .. the communication stream finished the hand-over (event R): push, waiting for the peer, drain / reduce */
    float wait_at_us;   /* since the first launch: the compute stream reached the launch that needs R (-1: never waited for) */
    float stall_us;     /* how long the compute stream stood still there (0: the hand-over was hidden behind interior work) */
} irs_slab_timeline_entry;
int irs_slab_timeline_arm(irs_ctx* ctx, int transitions);
int irs_slab_timeline_get(irs_ctx* ctx, irs_slab_timeline_entry* out, int max_entries, int32_t* n_entries, float* total_us, void* stream);

/* The schedule of the squaring steps as pure host arithmetic (tests, documentation): given the per-step ghost widths
 * h[0..n) (= floor(max|d_k|) + 1), the widest exchange and the smallest slab, fill fwd_round[k] / bwd_round[k] with the index
 * of the exchange round step k belongs to, and fwd_width[r] / bwd_width[r] with the planes that round exchanges (round 0 of
 * the forward pass is fed by the widened smoothing stage and exchanges the perturbed velocity instead).  n_buffers: gradient
 * fields the adjoint rotates through -- 2, or 3 as a context of several ranks has (a backward round then spans up to three
 * steps).  Returns the number of rounds through n_fwd / n_bwd; non-zero status if a width exceeds the limits. */
int irs_slab_plan_rounds(const int32_t* h, int n, int ghost_max, int min_slab, int n_buffers, int32_t* fwd_round, int32_t* fwd_width,
                         int32_t* n_fwd, int32_t* bwd_round, int32_t* bwd_width, int32_t* n_bwd);

/* The schedule of one planned transition of rank `rank`, as data -- the list the executor inside irs_slab_transition
 * interprets, built by the same host code (tests replay it on the CPU over two gloo ranks: tests/test_slab_schedule.py).
 * h[0..no_steps): ghost width of every squaring step.  n_ops receives the number of operations (also when max_ops is 0). */
enum { IRS_OP_LAUNCH = 0, IRS_OP_EXCHANGE = 1, IRS_OP_ALLREDUCE = 2, IRS_OP_WAIT = 3 };
enum {  /* launch stages */
    IRS_SG_PERTURB = 0, IRS_SG_COPY_V, IRS_SG_SMOOTH, IRS_SG_ENERGY, IRS_SG_EXP_FWD, IRS_SG_OUTPUTS, IRS_SG_WARP, IRS_SG_RESIDUAL,
    IRS_SG_STATS, IRS_SG_DATA_BWD, IRS_SG_WARP_BWD, IRS_SG_EXP_BWD, IRS_SG_UPDATE, IRS_SG_FFD_UP, IRS_SG_FFD_ADJ,
    IRS_SG_SCALARS = 32,  /* single-workgroup stages from here on (no output window) */
    IRS_SG_CHAIN_SCALAR = 32, IRS_SG_REG_SCALAR, IRS_SG_FINALIZE, IRS_SG_VERDICT
};
enum {  /* buffers */
    IRS_SB_V = 0, IRS_SB_NOISY, IRS_SB_VS, IRS_SB_WARPED, IRS_SB_Z, IRS_SB_GM, IRS_SB_GRAD_A, IRS_SB_GRAD_B,
    IRS_SB_DENSE,      /* SVFFD: the dense velocity (V, NOISY, VS then live on the control grid, replicated on every rank) */
    IRS_SB_GRAD_C,     /* third gradient field of the adjoint (contexts of several ranks) */
    IRS_SB_STEP0 = 16  /* + k: output of squaring step k */
};
enum { IRS_AR_ENERGY = 0, IRS_AR_DMAX = 1, IRS_AR_NLL = 2, IRS_AR_STATS = 3, IRS_AR_CPGRAD = 4, IRS_AR_MOMENTS = 7 };
typedef struct irs_slab_op {
    int32_t kind;              /* IRS_OP_* */
    int32_t stage;             /* launch: IRS_SG_*; exchange: the buffer (IRS_SB_*); all-reduce: IRS_AR_* */
    int32_t k;                 /* squaring step / chain */
    int32_t lo0, hi0, lo1, hi1;/* launch: output window [lo0, hi0) and, for boundary strips, a second one [lo1, hi1) */
    int32_t in0, in1;          /* launch: buffers read with a z reach (-1: none) */
    int32_t reach;             /* launch: planes beyond the output window(s) read from in0 / in1 */
    int32_t out;               /* launch: buffer written (-1: none) */
    int32_t width;             /* exchange: ghost planes per side */
    int32_t id;                /* exchange / all-reduce: its id; wait: the id waited for */
} irs_slab_op;
int irs_slab_trace(const irs_config* cfg, const irs_slab_config* scfg, int rank, int world, const int32_t* h, irs_slab_op* ops,
                   int max_ops, int32_t* n_ops);

/* Tuning / test switches (none of them selects other arithmetic: launch shapes, which always-correct kernel variants get
 * launched, test hooks).  The library reads its IRS_* environment variables ONCE per process, on first use; a context copies
 * them when it is created and no transition calls getenv.  This call changes one switch by name afterwards: on `ctx`, or --
 * ctx == NULL -- process-wide (the stateless operators and every context created later).  Names: predict_variants,
 * run_ahead, fuse_warp_bwd, energy_in_update, fuse_noise, recover, fwd_rows1, coarse_box, sobolev_tile, march_seg,
 * march_seg_fwd, swz_run, seg_min_blocks, seg_min_len, sobolev_seg, lcc_seg, stats_seg, update_seg, slab_split, slab_exact,
 * slab_force_h, slab_buffers, ps_rows, fwd_z2, tile_box, data_batch, chain_overlap, launch_log (csrc/common.h: Knobs).  The reference has no counterpart (it has one code path). */
int irs_option_set(irs_ctx* ctx, const char* name, int value);

const char* irs_last_error(void);
const char* irs_version(void);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* IRSGMCMC_H */
