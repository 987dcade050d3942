// Device-resident hyper-parameter state and the single-workgroup "scalar" kernels that replace the
// reference's host-side scalar autograd (trainer.py:68-77, 316-339; optimizers/adam_rate_decay.py) so that a
// whole transition runs without a host round trip (the reference syncs at trainer.py:308).
#pragma once
#include <string.h>

#include "common.h"

namespace irs {

constexpr int kStatVals = 5 + 2 * IRS_MAX_COMPONENTS;  // n, sum x^2, 3 lag-1 products, K dNLL/dlog_std, K resp. sums

struct DevState {
    irs_state st;  // parameters + Adam moments (host-visible layout)
    // mixture constants derived from st (refreshed whenever the parameters change)
    float A[IRS_MAX_COMPONENTS];          // log pi_k - log sigma_k - 0.5 log(2 pi)
    float inv_sigma[IRS_MAX_COMPONENTS];  // exp(-log sigma_k)
    float inv_var[IRS_MAX_COMPONENTS];    // inv_sigma_k^2
    int K;
    int mode;  // IRS_DATA_*
    float ssd_inv_sigma;
    irs_scalars sc;
    double coef[IRS_MAX_CHAINS];  // d(reg loss)/d(energy) per chain, consumed by the update kernel
    double moments[3];            // scratch: n, sum z, sum z^2 (GMM initialisation)
    unsigned bad_now;             // verdict about the transition in flight (Verdict below), written by the first scalar stage
    unsigned fails;               // transitions that ended as no-ops because an assumption about max|d_k| did not hold
    const unsigned* comm_err;     // slab over the peer-mapped transport: its sticky device error word (ipc.hip), else nullptr
    // the derived mixture constants as chain c's GMM step left them (chain_scalar_kernel, op bit 3): the data terms of ALL chains then
    // run as one launch after the serial statistics -> step loop, each chain against its own snapshot (api.hip: data_batch)
    float snapA[IRS_MAX_CHAINS][IRS_MAX_COMPONENTS];
    float snap_inv_var[IRS_MAX_CHAINS][IRS_MAX_COMPONENTS];
};

// A wait of the peer-mapped transport timed out (a peer is gone or late beyond IRS_IPC_TIMEOUT_S): whatever the exchanges after it
// delivered is stale, so every kernel that modifies persistent state treats the transition in flight as a no-op -- the same
// mechanism as a failed verdict -- and the host gets the error from its next call (comm_check).
__device__ __forceinline__ bool comm_bad(const DevState* s) {
    const unsigned* e = s->comm_err;
    return e != nullptr && __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}

// What the launch sequence of the transition in flight ASSUMED about the displacement bounds max|d_k| (decided on the host from
// bounds of earlier transitions, never waited for).  Every kernel that modifies persistent state -- the mixture and regulariser
// Adam steps, the velocity update, the Philox counter -- checks the assumptions against the bounds the forward pass has just
// measured; if one does not hold the transition becomes a NO-OP (no parameter, moment, counter or velocity changes), the
// failure is counted, and the host re-runs the transition without assumptions when it sees the count (api.hip / slab.hip).
struct Verdict {
    const unsigned* bounds;    // [steps + 1][C][4] max|d_k| per chain and axis (float bits); nullptr: nothing assumed
    int n, C;                  // squaring steps, chains
    unsigned need_lt1;         // bit k: step k only ran in a form that needs max|d_k| < 1 voxel (radius-1 kernels without fallback)
    unsigned need_lt2;         // bit k: the any-radius adjoint of step k was not launched -- beyond 2 voxels the radius-2 gather's generic
                               // fallback would own the step: correct, but not the bits of the any-radius kernel's fixed-point sums
    unsigned char width[32];   // slab: planned ghost width of step k (0: not planned) -- needs floor(max|d_k|) + 1 <= width
};
inline Verdict no_verdict() {
    Verdict v;
    memset(&v, 0, sizeof(v));
    return v;
}

// uniform over the block; every thread must call (one barrier)
__device__ __forceinline__ bool verdict_bad(const Verdict& v) {
    int bad = 0;
    if (v.bounds) {
        const int per = 4 * v.C;
        for (int i = threadIdx.x; i < v.n * per; i += blockDim.x) {
            const int k = i / per;
            const float m = __uint_as_float(v.bounds[i]);
            if (k < 32 && ((v.need_lt1 >> k) & 1u) && !(m < 1.0f)) bad = 1;
            if (k < 32 && ((v.need_lt2 >> k) & 1u) && !(m < 2.0f)) bad = 1;
            if (k < 32 && v.width[k] && (!(m >= 0.0f) || (int)floorf(m) + 1 > (int)v.width[k])) bad = 1;
        }
    }
    return __syncthreads_or(bad) != 0;
}

struct DevCfg {
    int K, mode, vd, C;
    float gmm_lr_log_std, gmm_lr_logits, gmm_lr_decay, beta1, beta2, eps;
    float scale_prior_loc, scale_prior_scale;
    float conc[IRS_MAX_COMPONENTS];
    int reg_loss, reg_learnable;
    double dof;
    float reg_lr0, reg_lr1, reg_lr_decay;
    float loc_prior_nu, loc_prior_w_reg, reg_scale_prior_loc, reg_scale_prior_scale;
    double w_reg_prior_shape, w_reg_prior_rate;
};

// per-voxel mixture evaluation shared by the statistics and the data-term kernels
struct MixEval {
    float nll;  // -log p(z)
    float gz;   // d(-log p)/dz = sum_k r_k z / sigma_k^2
    float x;    // VD-rescaled residual: sum_k r_k (z / sigma_k)^2   (utils/util.py:330-347, closed form)
};

// KMAX: compile-time bound of the component loops (the caller knows K <= KMAX): the arrays of a K = 4 mixture then take 4
// registers each instead of IRS_MAX_COMPONENTS
// mixA / mix_iv: the constants A_k and 1 / sigma_k^2 to evaluate with (DevState::A / inv_var, or a chain's snapshot of them)
template <bool WANT_RESP, int KMAX = IRS_MAX_COMPONENTS>
__device__ __forceinline__ MixEval mix_eval_with(float z, const DevState* __restrict__ s, const float* __restrict__ mixA,
                                                 const float* __restrict__ mix_iv, float* resp, float* q) {
    MixEval e;
    if (s->mode == IRS_DATA_SSD) {
        const float u = z * s->ssd_inv_sigma;
        e.x = u * u;
        e.nll = 0.5f * e.x;
        e.gz = u * s->ssd_inv_sigma;
        return e;
    }
    const int K = s->K;
    const float z2 = z * z;
    float t[KMAX], qq[KMAX];
    float m = -3.0e38f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            qq[k] = z2 * mix_iv[k];  // (z / sigma_k)^2 with the square of z shared by the components
            t[k] = fmaf(-0.5f, qq[k], mixA[k]);
            m = fmaxf(m, t[k]);
        }
    }
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            t[k] = __expf(t[k] - m);
            sum += t[k];
        }
    }
    const float inv = __builtin_amdgcn_rcpf(sum);  // sum in [1, K]; 1 ulp
    e.nll = -(m + __logf(sum));
    e.x = 0.0f;
    float gs = 0.0f;  // sum_k r_k / sigma_k^2
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            const float r = t[k] * inv;
            e.x += r * qq[k];
            gs += r * mix_iv[k];
            if (WANT_RESP) {
                resp[k] = r;
                q[k] = qq[k];
            }
        }
    }
    e.gz = z * gs;
    return e;
}
template <bool WANT_RESP, int KMAX = IRS_MAX_COMPONENTS>
__device__ __forceinline__ MixEval mix_eval(float z, const DevState* __restrict__ s, float* resp, float* q) {
    return mix_eval_with<WANT_RESP, KMAX>(z, s, s->A, s->inv_var, resp, q);
}

void launch_refresh_derived(DevState* s, DevCfg cfg, hipStream_t st);
// op bit 0: recompute the VD factor alpha; bit 1: take one GMM Adam step (with the stored alpha); bit 2: evaluate the verdict
// about the transition in flight into DevState::bad_now (the first scalar stage of a transition; the others read it); bit 3: leave
// a snapshot of the mixture constants for this chain's data term (DevState::snapA)
void launch_chain_scalar(DevState* s, const double* stat_partials, int nblocks, int chain, int op, DevCfg cfg,
                         hipStream_t st, Verdict vd = no_verdict());
void launch_reg_scalar(DevState* s, const double* energy_partials, int nblocks, DevCfg cfg, hipStream_t st,
                       Verdict vd = no_verdict());
// Publishes the bounds of this transition and the cumulative count of failed (no-op) transitions (hint[flag_word]) to pinned
// host memory; advances the Philox counter unless the verdict is bad.  `zero_bounds`: clear the bound scratch for the next
// transition.  `reg_partials` (optional): the regulariser scalar stage runs first, in the same launch (energy from the update).
void launch_finalize(DevState* s, const double* nll_partials, int nblocks_per_chain, DevCfg cfg, bool advance,
                     unsigned* bounds, unsigned* hint, int nbounds, Verdict vd, int flag_word, bool zero_bounds,
                     hipStream_t st, const double* reg_partials = nullptr, int reg_blocks = 0);
void launch_gmm_init_from_moments(DevState* s, const double* moment_partials, int nblocks, DevCfg cfg, hipStream_t st);

}  // namespace irs
