// Single-workgroup kernels for the hyper-parameter side of the transition: deterministic (fixed-order)
// reduction of per-block partial sums in fp64, the virtual-decimation factor, the GMM Adam step, the
// regulariser energy terms / coefficients / Adam step, and the loss bookkeeping.
// Reference: trainer/trainer.py:68-77,316-339,507-514; utils/util.py:446-485; optimizers/adam_rate_decay.py:32-99;
// model/loss.py:172-312; model/distributions.py.
#include "scalar_kernels.h"

namespace irs {

#ifndef IRS_REDUCE_U
#define IRS_REDUCE_U 4
#endif
#ifdef IRS_SCALAR_TRACE
__device__ unsigned long long g_scalar_trace[8];
extern "C" __attribute__((visibility("default"))) int irs_debug_scalar_trace(unsigned long long* out) {  // (trace builds only: the library is compiled -fvisibility=hidden)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scalar_trace), sizeof(g_scalar_trace)) == hipSuccess ? 0 : 1;
}
#define IRS_ST(i) do { if (threadIdx.x == 0) g_scalar_trace[i] = wall_clock64(); } while (0)
#else
#define IRS_ST(i)
#endif
// out[j] (j < nvals) = sum_b partials[b][j], same order every run.  All threads must call.  BY_COLUMN: the partials are stored
// [nvals][nblocks] (the statistics kernel writes them that way): consecutive lanes then read consecutive doubles.  With [nblocks][21]
// rows every load instruction touched ~45 cache lines, and reading 2048 rows was 11 of chain_scalar_kernel's 22 us.
template <int NV, bool BY_COLUMN = false>
__device__ void reduce_partials(const double* __restrict__ partials, int nblocks, int nvals, double (&out)[NV],
                                double* smem) {
    double acc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] = 0.0;
    // BY_COLUMN: the loads of kU rows are in flight together (a thread still adds its rows in ascending order): the partials were
    // written by workgroups on all eight XCDs, so every row is a trip to memory, and 2048 rows were eight of them in a row
    constexpr int kU = BY_COLUMN ? IRS_REDUCE_U : 1;
    for (int b0 = threadIdx.x; b0 < nblocks; b0 += kBlock * kU) {
        double tmp[kU][NV];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int b = b0 + u * kBlock;
#pragma unroll
            for (int j = 0; j < NV; ++j)
                tmp[u][j] = (b < nblocks && j < nvals) ? (BY_COLUMN ? partials[(int64_t)j * nblocks + b] : partials[(int64_t)b * nvals + j]) : 0.0;
        }

#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (b0 + u * kBlock < nblocks) {
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    if (j < nvals) acc[j] += tmp[u][j];
            }
    }
    if (NV > 3) IRS_ST(6);
    block_sum<NV>(acc, smem);
#pragma unroll
    for (int j = 0; j < NV; ++j) out[j] = acc[j];  // valid in thread 0
}

// sum of one column of per-block partial sums: a thread's rows in ascending order as before, 8 loads in flight instead of one
// round trip per row (finalize_kernel 12.1 -> 7.8 us at 256^3); then the fixed-order block sum.  (The same for the 21-column
// statistics rows of chain_scalar_kernel makes it SLOWER: 18 -> 28 / 32 us with 2 / 4 rows in flight.)
__device__ __forceinline__ void sum_rows(const double* __restrict__ p, int nblocks, double (&acc)[1], double* smem) {
    acc[0] = 0.0;
    constexpr int kU = 8;
    for (int b0 = threadIdx.x; b0 < nblocks; b0 += kBlock * kU) {
        double tmp[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) tmp[u] = b0 + u * kBlock < nblocks ? p[b0 + u * kBlock] : 0.0;
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (b0 + u * kBlock < nblocks) acc[0] += tmp[u];
    }
    block_sum<1>(acc, smem);
}

__device__ void refresh_derived(DevState* s, const DevCfg& cfg) {
    s->K = cfg.K;
    s->mode = cfg.mode;
    // log_softmax(logits + 1e-2) (model/loss.py:67-69), fp32 like the reference
    float m = -3.0e38f;
    for (int k = 0; k < cfg.K; ++k) m = fmaxf(m, s->st.gmm_logits[k] + 1e-2f);
    float sum = 0.0f;
    for (int k = 0; k < cfg.K; ++k) sum += expf(s->st.gmm_logits[k] + 1e-2f - m);
    const float lse = m + logf(sum);
    for (int k = 0; k < cfg.K; ++k) {
        const float lp = s->st.gmm_logits[k] + 1e-2f - lse;
        s->A[k] = (lp - s->st.gmm_log_std[k]) - 0.91893853320467274178f;
        s->inv_sigma[k] = expf(-1.0f * s->st.gmm_log_std[k]);
        s->inv_var[k] = s->inv_sigma[k] * s->inv_sigma[k];
    }
}

// optimizers/adam_rate_decay.py:32-99 for one scalar parameter (the state is never re-initialised on this path):
// clr = lr / (1 + step * lr_decay); bias corrections count from step 0
// the part that depends on the step count only: chain_scalar_kernel forms it while the partial sums are still on their way
struct AdamCoef {
    double clr, bc1, bc2;
};
__device__ __forceinline__ AdamCoef adam_coef(int64_t step_before, double lr, double lr_decay, double b1, double b2) {
    AdamCoef a;
    a.clr = lr / (1.0 + (double)step_before * lr_decay);
    const double t = (double)(step_before + 1);
    a.bc1 = 1.0 - pow(b1, t);
    a.bc2 = 1.0 - pow(b2, t);
    return a;
}
__device__ __forceinline__ double adam_step_with(double p, double g, double& m, double& v, const AdamCoef& a, double b1, double b2, double eps) {
    m = b1 * m + (1.0 - b1) * g;
    v = b2 * v + (1.0 - b2) * g * g;
    const double denom = sqrt(v) / sqrt(a.bc2) + eps;
    return p - (a.clr / a.bc1) * m / denom;
}
__device__ double adam_step_decay(double p, double g, double& m, double& v, int64_t step_before, double lr,
                                  double lr_decay, double b1, double b2, double eps) {
    return adam_step_with(p, g, m, v, adam_coef(step_before, lr, lr_decay, b1, b2), b1, b2, eps);
}

__global__ __launch_bounds__(kBlock) void refresh_kernel(DevState* s, DevCfg cfg) {
    if (threadIdx.x == 0) refresh_derived(s, cfg);
}

void launch_refresh_derived(DevState* s, DevCfg cfg, hipStream_t st) {
    hipLaunchKernelGGL(refresh_kernel, dim3(1), dim3(kBlock), 0, st, s, cfg);
}

// ------------------------------------------------------------------------------------------------
// per chain: VD factor (utils/util.py:446-485) and one _step_GMM (trainer.py:68-77)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void chain_scalar_kernel(DevState* s, const double* __restrict__ partials,
                                                              int nblocks, int chain, int op, DevCfg cfg, Verdict vd) {
    __shared__ double smem[kStatVals * (kBlock / kWave)];
    IRS_ST(0);
    // the first scalar stage of a transition evaluates the verdict (scalar_kernels.h); a bad one freezes every parameter
    bool bad;
    if (op & 4) {
        bad = verdict_bad(vd);
        if (threadIdx.x == 0) s->bad_now = bad ? 1u : 0u;
    } else {
        bad = vd.bounds != nullptr && s->bad_now != 0u;
    }
    bad = bad || comm_bad(s);
    __shared__ double rs[kStatVals];
    __shared__ double alpha_s;
    // step-count part of the Adam update (two fp64 pow), one lane per parameter, BEFORE the reduction: it overlaps the memory
    // round trips of the partial sums instead of following them (same expressions, same values)
    // (computed before the reduction rather than after it: 19.0 -> 16.6 us; placed between the issue of the first loads and their
    // use it gains nothing more -- the reduction is not waiting on memory)
    AdamCoef coef = {0.0, 1.0, 1.0};
    if (cfg.mode == IRS_DATA_GMM_LCC && (op & 2) && (int)threadIdx.x < 2 * cfg.K) {
        const bool ls = (int)threadIdx.x < cfg.K;
        coef = adam_coef(s->st.gmm_adam_step[ls ? 0 : 1], ls ? cfg.gmm_lr_log_std : cfg.gmm_lr_logits, cfg.gmm_lr_decay, cfg.beta1, cfg.beta2);
    }
    double r[kStatVals];
    IRS_ST(1);
    reduce_partials<kStatVals, true>(partials, nblocks, kStatVals, r, smem);
    IRS_ST(2);
    if (threadIdx.x == 0) {
        const double n = r[0];
        double alpha = (op & 1) ? 1.0 : s->sc.alpha[chain];
        if (cfg.vd && (op & 1)) {
            const double var = r[1] / n;
            double prod = 1.0;
            for (int a = 0; a < 3; ++a) {
                const double corr = (r[2 + a] / n) / var;
                prod *= fmin(-2.0 / 3.14159265358979323846 * log(corr), 1.0);
            }
            alpha = sqrt(prod);  // NaN if a lag-1 correlation is negative, as in the reference
        }
        s->sc.alpha[chain] = alpha;
        s->sc.n_mask[chain] = n;
        alpha_s = alpha;
#pragma unroll
        for (int j = 0; j < kStatVals; ++j) rs[j] = r[j];
    }
    __syncthreads();
    IRS_ST(3);
    if (!(cfg.mode == IRS_DATA_GMM_LCC && (op & 2)) || bad) {
        if ((op & 8) && (int)threadIdx.x < cfg.K) {  // no step: the snapshot is the mixture as it stands
            s->snapA[chain][threadIdx.x] = s->A[threadIdx.x];
            s->snap_inv_var[chain][threadIdx.x] = s->inv_var[threadIdx.x];
        }
        return;
    }
    // one GMM Adam step (trainer.py:68-77), one lane per parameter: lanes 0 .. K-1 the log std, K .. 2K-1 the logits (the
    // fp64 pow / exp / sqrt of sixteen serial updates on one lane were 20 us of every transition)
    const int K = cfg.K;
    const int t = threadIdx.x;
    // every lane's inputs are staged in shared memory first and the new values are written after a barrier at uniform control
    // flow: the logit lanes read all K logits, and nothing may overwrite them before every lane has read
    __shared__ float old_ls[IRS_MAX_COMPONENTS], old_lg[IRS_MAX_COMPONENTS];
    if (t < K) {
        old_ls[t] = s->st.gmm_log_std[t];
        old_lg[t] = s->st.gmm_logits[t];
    }
    __syncthreads();
    float newv = 0.0f;
    const int k = t < K ? t : t - K;
    if (t < 2 * K) {
        const double n = rs[0], alpha = alpha_s;
        const double* Gs = rs + 5;
        const double* Gl = rs + 5 + IRS_MAX_COMPONENTS;
        // gradient and parameter per lane, then ONE call of the update for both kinds (in two divergent branches the fp64 square
        // roots and divisions of the update ran twice, one after the other)
        double g_par, p_old;
        double *mp, *vp;
        if (t < K) {
            const double sp2 = (double)cfg.scale_prior_scale * (double)cfg.scale_prior_scale;
            // d/dlog_std_k [alpha NLL - log N(log_std; loc, scale)]
            g_par = alpha * Gs[k] + ((double)old_ls[k] - (double)cfg.scale_prior_loc) / sp2;
            p_old = (double)old_ls[k];
            mp = &s->st.gmm_adam_m[0][k];
            vp = &s->st.gmm_adam_v[0][k];
        } else {
            // proportions pi = softmax(logits + 1e-2)
            double mx = -1e300, sum = 0.0, csum = 0.0;
            for (int j = 0; j < K; ++j) mx = fmax(mx, (double)old_lg[j]);
            for (int j = 0; j < K; ++j) {
                sum += exp((double)old_lg[j] - mx);
                csum += (double)cfg.conc[j] - 1.0;
            }
            const double pik = exp((double)old_lg[k] - mx) / sum;
            // d/dlogit_k [alpha NLL - log Dir(log pi)]
            g_par = alpha * (-Gl[k] + pik * n) + (-((double)cfg.conc[k] - 1.0) + pik * csum);
            p_old = (double)old_lg[k];
            mp = &s->st.gmm_adam_m[1][k];
            vp = &s->st.gmm_adam_v[1][k];
        }
        newv = (float)adam_step_with(p_old, g_par, *mp, *vp, coef, cfg.beta1, cfg.beta2, cfg.eps);
    }
    __syncthreads();
    IRS_ST(4);
    if (t < K) s->st.gmm_log_std[k] = newv;
    else if (t < 2 * K) s->st.gmm_logits[k] = newv;
    __syncthreads();
    if (t == 0) {
        s->st.gmm_adam_step[0] += 1;
        s->st.gmm_adam_step[1] += 1;
        refresh_derived(s, cfg);
        if (op & 8)
            for (int j = 0; j < K; ++j) {
                s->snapA[chain][j] = s->A[j];
                s->snap_inv_var[chain][j] = s->inv_var[j];
            }
    }
    IRS_ST(5);
}

void launch_chain_scalar(DevState* s, const double* stat_partials, int nblocks, int chain, int op, DevCfg cfg,
                         hipStream_t st, Verdict vd) {
    hipLaunchKernelGGL(chain_scalar_kernel, dim3(1), dim3(kBlock), 0, st, s, stat_partials, nblocks, chain, op, cfg, vd);
}

// ------------------------------------------------------------------------------------------------
// regulariser: energies -> loss terms, d(loss)/d(energy) coefficients, Adam step on its hyper-parameters
// energy_partials: [C][nblocks]
// ------------------------------------------------------------------------------------------------
// all threads of the block call; `bad`: the transition is a no-op, the hyper-parameters and their moments stay as they are
__device__ void reg_scalar_body(DevState* s, const double* __restrict__ partials, int nblocks, const DevCfg& cfg, bool bad) {
    __shared__ double smem[kBlock / kWave];
    __shared__ double ysh[IRS_MAX_CHAINS];
    for (int c = 0; c < cfg.C; ++c) {
        double acc[1];
        sum_rows(partials + (int64_t)c * nblocks, nblocks, acc, smem);
        if (threadIdx.x == 0) ysh[c] = acc[0];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;

    const double dof = cfg.dof;
    if (cfg.reg_loss == IRS_REG_L2) {
        // model/loss.py:197-198: 0.5 w y - 0.5 dof log w
        const double lw = s->st.reg_param[0], w = exp(lw);
        double g_lw = 0.0;
        for (int c = 0; c < cfg.C; ++c) {
            const double y = ysh[c];
            s->sc.reg_energy[c] = y;
            s->sc.reg_term[c] = 0.5 * w * y - 0.5 * dof * lw;
            s->coef[c] = 0.5 * w;
            g_lw += 0.5 * w * y - 0.5 * dof;
        }
        if (cfg.reg_learnable && !bad) {
            // minus LogPrecisionExpGammaPrior(log w): d/dx [shape log rate + (shape-1) x - rate e^x - lgamma + x]
            g_lw -= cfg.w_reg_prior_shape - cfg.w_reg_prior_rate * w;
            const int64_t st0 = s->st.reg_adam_step[0];
            s->st.reg_param[0] = adam_step_decay(lw, g_lw, s->st.reg_adam_m[0], s->st.reg_adam_v[0], st0, cfg.reg_lr0,
                                                 cfg.reg_lr_decay, cfg.beta1, cfg.beta2, cfg.eps);
            s->st.reg_adam_step[0] = st0 + 1;
        }
    } else if (cfg.reg_loss == IRS_REG_STUDENT) {
        // model/loss.py:234-241: (a0 + dof/2) log(2 b0 + y)      (a0, 2 b0 travel in w_reg_prior_shape / _rate)
        const double a0 = cfg.w_reg_prior_shape, b2 = cfg.w_reg_prior_rate;
        for (int c = 0; c < cfg.C; ++c) {
            const double y = ysh[c];
            s->sc.reg_energy[c] = y;
            s->sc.reg_term[c] = log(b2 + y) * (a0 + 0.5 * dof);
            s->coef[c] = (a0 + 0.5 * dof) / (b2 + y);
        }
    } else if (cfg.reg_loss == IRS_REG_LOGNORMAL_L2) {
        // model/loss.py:315-321 + :262-270: -log Gamma(y; dof/2, w/2) + (dof/2 - 1) log y  (= w y / 2 + const)
        const double shape = 0.5 * dof, rate = 0.5 * exp(s->st.reg_param[0]);
        for (int c = 0; c < cfg.C; ++c) {
            const double y = ysh[c], ly = log(y);
            s->sc.reg_energy[c] = y;
            s->sc.reg_term[c] = -(shape * log(rate) + (shape - 1.0) * ly - rate * y - lgamma(shape)) + (0.5 * dof - 1.0) * ly;
            s->coef[c] = rate;
        }
    } else {
        // model/loss.py:266-312: log y + log s + 0.5 ((log y - loc)/s)^2 + (dof/2 - 1) log y
        const double loc = s->st.reg_param[0], ls = s->st.reg_param[1], sc = exp(ls);
        double g_loc = 0.0, g_ls = 0.0;
        for (int c = 0; c < cfg.C; ++c) {
            const double y = ysh[c], ly = log(y), u = (ly - loc) / sc;
            s->sc.reg_energy[c] = y;
            s->sc.reg_term[c] = ly + ls + 0.5 * u * u + (0.5 * dof - 1.0) * ly;
            double dly = 1.0 + u / sc + (0.5 * dof - 1.0);
            if (cfg.reg_learnable) {
                // minus LogEnergyExpGammaPrior evaluated AT log y (trainer.py:336): -(a - 1) - 1 + b y
                const double a = 0.5 * (double)cfg.loc_prior_nu * dof, b = 0.5 * (double)cfg.loc_prior_nu * (double)cfg.loc_prior_w_reg;
                dly += -a + b * y;
            }
            s->coef[c] = dly / y;
            g_loc += -u / sc;
            g_ls += 1.0 - u * u;
        }
        if (cfg.reg_learnable && !bad) {
            const double ps = (double)cfg.reg_scale_prior_scale;
            g_ls += (ls - (double)cfg.reg_scale_prior_loc) / (ps * ps);  // minus LogScaleNormalPrior(log_scale)
            const int64_t st0 = s->st.reg_adam_step[0], st1 = s->st.reg_adam_step[1];
            s->st.reg_param[0] = adam_step_decay(loc, g_loc, s->st.reg_adam_m[0], s->st.reg_adam_v[0], st0, cfg.reg_lr0,
                                                 cfg.reg_lr_decay, cfg.beta1, cfg.beta2, cfg.eps);
            s->st.reg_param[1] = adam_step_decay(ls, g_ls, s->st.reg_adam_m[1], s->st.reg_adam_v[1], st1, cfg.reg_lr1,
                                                 cfg.reg_lr_decay, cfg.beta1, cfg.beta2, cfg.eps);
            s->st.reg_adam_step[0] = st0 + 1;
            s->st.reg_adam_step[1] = st1 + 1;
        }
    }
}

__global__ __launch_bounds__(kBlock) void reg_scalar_kernel(DevState* s, const double* __restrict__ partials, int nblocks,
                                                            DevCfg cfg, Verdict vd) {
    const bool bad = verdict_bad(vd) || comm_bad(s);  // (this stage may run before the first chain's scalar stage)
    reg_scalar_body(s, partials, nblocks, cfg, bad);
}

void launch_reg_scalar(DevState* s, const double* energy_partials, int nblocks, DevCfg cfg, hipStream_t st, Verdict vd) {
    hipLaunchKernelGGL(reg_scalar_kernel, dim3(1), dim3(kBlock), 0, st, s, energy_partials, nblocks, cfg, vd);
}

// data_term[c] = alpha_c * sum(-log p(z_c)) with the parameters in force for chain c; advance the Philox counter; publish
// the bounds and the count of failed transitions.  Optionally the regulariser scalar stage first (one launch less).
__global__ __launch_bounds__(kBlock) void finalize_kernel(DevState* s, const double* __restrict__ partials,
                                                          int nblocks_per_chain, DevCfg cfg, int advance,
                                                          unsigned* __restrict__ bounds, unsigned* __restrict__ hint,
                                                          int nbounds, Verdict vd, int flag_word, int zero_bounds,
                                                          const double* __restrict__ reg_partials, int reg_blocks) {
    __shared__ double smem[kBlock / kWave];
    const bool bad = verdict_bad(vd) || comm_bad(s);
    if (reg_partials) {
        reg_scalar_body(s, reg_partials, reg_blocks, cfg, bad);
        __syncthreads();
    }
    // publish the displacement bounds of this transition to pinned host memory (the host reads them unsynchronised, as a
    // hint for which kernel variants to launch next time)
    if (bounds && hint) {
        for (int i = threadIdx.x; i < nbounds; i += kBlock) {
            hint[i] = bounds[i];
            if (zero_bounds) bounds[i] = 0u;
        }
    }
    for (int c = 0; c < cfg.C; ++c) {
        double acc[1];
        sum_rows(partials + (int64_t)c * nblocks_per_chain, nblocks_per_chain, acc, smem);
        if (threadIdx.x == 0) s->sc.data_term[c] = s->sc.alpha[c] * acc[0];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (bad) s->fails += 1u;
        else if (advance) s->st.iteration += 1;
        if (hint) hint[flag_word] = s->fails;  // cumulative: the host compares it with the count it has already acted on
    }
}

void launch_finalize(DevState* s, const double* nll_partials, int nblocks_per_chain, DevCfg cfg, bool advance,
                     unsigned* bounds, unsigned* hint, int nbounds, Verdict vd, int flag_word, bool zero_bounds,
                     hipStream_t st, const double* reg_partials, int reg_blocks) {
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(kBlock), 0, st, s, nll_partials, nblocks_per_chain, cfg,
                       advance ? 1 : 0, bounds, hint, nbounds, vd, flag_word, zero_bounds ? 1 : 0, reg_partials, reg_blocks);
}

// GMM.init_parameters (model/loss.py:61-65) from the unbiased std of the masked residuals (trainer.py:537-541)
__global__ __launch_bounds__(kBlock) void gmm_init_kernel(DevState* s, const double* __restrict__ partials, int nblocks,
                                                          DevCfg cfg) {
    __shared__ double smem[3 * (kBlock / kWave)];
    double r[3];
    reduce_partials<3>(partials, nblocks, 3, r, smem);
    if (threadIdx.x != 0) return;
    const double n = r[0], mean = r[1] / n;
    const double var = (r[2] - n * mean * mean) / (n - 1.0);
    const float sd = (float)sqrt(var);
    s->moments[0] = n;
    s->moments[1] = mean;
    s->moments[2] = (double)sd;
    const float lo = logf(sd / 100.0f), hi = logf(sd * 5.0f);
    const int K = cfg.K;
    // torch.linspace(lo, hi, K): start + step*i below the midpoint, end - step*(K-1-i) above
    const float step = K > 1 ? (hi - lo) / (float)(K - 1) : 0.0f;
    for (int k = 0; k < K; ++k) s->st.gmm_log_std[k] = k < K / 2 ? lo + step * (float)k : hi - step * (float)(K - 1 - k);
    refresh_derived(s, cfg);
}

void launch_gmm_init_from_moments(DevState* s, const double* moment_partials, int nblocks, DevCfg cfg, hipStream_t st) {
    hipLaunchKernelGGL(gmm_init_kernel, dim3(1), dim3(kBlock), 0, st, s, moment_partials, nblocks, cfg);
}

}  // namespace irs
