// z-slab decomposition of ONE chain over the GPUs of a node, inside the library (include/irsgmcmc.h: irs_slab_*).
// What is sharded is the single-device loop body of the reference, trainer/trainer.py:291-356.
//
// Every rank owns the planes [a, b) and holds [lo, hi) of every array (ctx.h: SlabInfo, common.h: Vol).  A transition is
// the launch sequence of irs_transition on windows of the slab, with ghost-plane exchanges between neighbouring ranks on a
// communication stream `cs` (RCCL send / recv, comm.hip) tied to the compute stream `st` by events:
//
//      st:  ... producer(boundary strips) | record P | interior launches ............... | wait R | boundary launches ...
//      cs:                                  wait P   | grouped send / recv of the strips | record R
//
// A ROUND is one exchange plus the squaring steps that live off it: the forward pass groups steps into rounds whose ghost
// widths add up to <= ghost_max (the steps of a round run on shrinking windows, recomputing ghost planes instead of
// exchanging them: the early steps need one plane each); the backward pass groups adjoint steps as far as the ghost
// planes of d_k that the forward pass left behind allow (plan_rounds below).  The ghost width of step k is
// floor(max|d_k|) + 1 planes; the plan of a transition comes from bounds the host has already seen (all-reduced, published
// to pinned memory by the finalize kernel of an earlier transition) with a safety factor, and is validated on the device.
#include <math.h>
#include <new>

#include "comm.h"
#include "ctx.h"

using namespace irs;

namespace {

constexpr int kMaxSteps = 32;

struct Plan {
    int n = 0;
    int h[kMaxSteps];                       // ghost width of step k (forward step k and adjoint step k)
    float m[kMaxSteps];                     // exact mode: the measured bound max|d_k| (voxels, all chains and axes)
    int fr[kMaxSteps], fw[kMaxSteps], nf;   // forward: round of step k, exchange width of round r
    int br[kMaxSteps], bw[kMaxSteps], nb;   // backward: round of step k (rounds in execution order, k descending)
};

// Rounds of the squaring steps (pure arithmetic, also exported for the tests).
//   forward   greedy: a round takes steps while the sum of their ghost widths stays <= gmax (a single step wider than gmax
//             is a round of its own).  Round [k0, k1] needs d_k0 on slab +- w, w = sum h; step k then leaves d_k valid on
//             slab +- E_k, E_k = sum_{j = k .. k1} h_j.
//   backward  adjoint step k needs the incoming gradient AND d_k on (its output window) +- h_k.  A round [k0', k1'] (run from
//             k1' down to k0', output of k0' on the bare slab) needs d_k on slab +- N_k, N_k = sum_{j = k0' .. k} h_j: it can
//             only be as long as N_k <= E_k holds for all its steps -- the ghost planes of d_k the forward pass computed.
int plan_rounds(Plan& p, int gmax, int min_slab) {
    const int n = p.n;
    int r = 0, w = 0, k1_of[kMaxSteps];
    for (int k = 0; k < n; ++k) {
        if (p.h[k] < 1) return fail("slab plan: ghost width of step %d is %d", k, p.h[k]);
        if (w > 0 && w + p.h[k] > gmax) {
            p.fw[r++] = w;
            w = 0;
        }
        p.fr[k] = r;
        w += p.h[k];
    }
    p.fw[r] = w;
    p.nf = r + 1;
    for (int k = n - 1, last = n - 1; k >= 0; --k) {
        if (k < n - 1 && p.fr[k] != p.fr[k + 1]) last = k;
        k1_of[k] = last;
    }
    int E[kMaxSteps];
    for (int k = 0; k < n; ++k) {
        E[k] = 0;
        for (int j = k; j <= k1_of[k]; ++j) E[k] += p.h[j];
    }
    int rb = 0;
    for (int k1 = n - 1; k1 >= 0;) {
        int k0 = k1;  // a single step is always possible: N = h_k <= E_k
        while (k0 - 1 >= 0) {
            const int c0 = k0 - 1;
            int N = 0;
            bool ok = true;
            for (int k = c0; k <= k1 && ok; ++k) {
                N += p.h[k];
                ok = N <= E[k];
            }
            // at most two steps: the adjoint ping-pongs between two gradient buffers, so the interior of a third step would
            // overwrite planes of the buffer whose boundary strips are still being sent
            if (!ok || N > gmax || k1 - c0 + 1 > 2) break;
            k0 = c0;
        }
        int wsum = 0;
        for (int k = k0; k <= k1; ++k) {
            p.br[k] = rb;
            wsum += p.h[k];
        }
        p.bw[rb++] = wsum;
        k1 = k0 - 1;
    }
    p.nb = rb;
    for (int i = 0; i < p.nf; ++i)
        if (p.fw[i] > min_slab) return fail("slab plan: a ghost zone of %d planes exceeds the smallest slab (%d planes): fewer ranks for this displacement", p.fw[i], min_slab);
    for (int i = 0; i < p.nb; ++i)
        if (p.bw[i] > min_slab) return fail("slab plan: a ghost zone of %d planes exceeds the smallest slab (%d planes): fewer ranks for this displacement", p.bw[i], min_slab);
    return 0;
}

int default_margin(const irs_config* cfg, int gmax) {
    const int ls = cfg->data_loss == IRS_DATA_GMM_LCC ? cfg->lcc_s : 0;
    int m = cfg->sobolev_s + gmax;
    if (4 * ls > m) m = 4 * ls;
    return m + 2;
}

int plan_layout(const irs_config* cfg, const irs_slab_config* scfg, int rank, int world, SlabInfo* sl) {
    if (!cfg || !sl || world < 1 || rank < 0 || rank >= world) return fail("slab layout: bad arguments");
    const int D = cfg->dims[0];
    if (world > D / 2) return fail("slab layout: %d ranks for %d planes", world, D);
    sl->on = true;
    sl->rank = rank;
    sl->world = world;
    sl->a = (int)(((int64_t)rank * D) / world);
    sl->b = (int)(((int64_t)(rank + 1) * D) / world);
    sl->gmax = scfg && scfg->ghost_max > 0 ? scfg->ghost_max : 4;
    sl->margin = scfg && scfg->margin > 0 ? scfg->margin : default_margin(cfg, sl->gmax);
    sl->has_lo = rank > 0;
    sl->has_hi = rank + 1 < world;
    sl->lo = sl->has_lo ? (sl->a - sl->margin > 0 ? sl->a - sl->margin : 0) : 0;
    sl->hi = sl->has_hi ? (sl->b + sl->margin < D ? sl->b + sl->margin : D) : D;
    sl->min_slab = D;
    for (int r = 0; r < world; ++r) {
        const int n = (int)(((int64_t)(r + 1) * D) / world) - (int)(((int64_t)r * D) / world);
        if (n < sl->min_slab) sl->min_slab = n;
    }
    return 0;
}

// ---- slab-local arrays: virtual base pointers (global z indexing lands inside the held planes) ---------------------------
struct Views {
    int64_t P;      // first held plane * H * W
    int64_t field;  // elements of one (C,3,held,H,W) field
};

inline Views views(const irs_ctx* c) { return Views{(int64_t)c->sl.lo * c->vol.H * c->vol.W, (int64_t)c->C * 3 * c->vol.V}; }
template <typename T>
inline T* planar(T* p, const Views& v) { return p ? p - v.P : nullptr; }  // planar 3-channel fields and images alike
inline float* aos(float* p, const Views& v) { return p ? p - 3 * v.P : nullptr; }

inline float* step_buf(const irs_ctx* c, const Views& v, int k) {  // output of squaring step k = d_{k+1}
    float* raw = c->steps + (int64_t)k * v.field;
    return (fwd_lay(c, k) & 4) ? aos(raw, v) : planar(raw, v);
}
inline bool step_is_aos(const irs_ctx* c, int k) { return (fwd_lay(c, k) & 4) != 0; }
// dL/d(d_last) lands in A; adjoint step n-1 writes B, the next one A, ...
inline float* grad_raw(const irs_ctx* c, int k, bool input) {
    const bool odd = ((c->cfg.no_steps - k) & 1) != 0;
    return (odd == input) ? c->gA : c->gB;
}

// ---- one grouped exchange of `w` ghost planes on both sides of the slab --------------------------------------------------
enum { F_PLANAR3 = 0, F_AOS3 = 1, F_IMAGE = 2 };

int exchange_planes(irs_ctx* c, float* virt, int kind, int chains, int w, hipStream_t cs) {
    const SlabInfo& s = c->sl;
    if (s.world == 1 || w <= 0 || !c->comm) return 0;
    if (w > s.min_slab) return fail("slab: ghost zone of %d planes exceeds the smallest slab (%d planes)", w, s.min_slab);
    if (w > s.margin) return fail("slab: ghost zone of %d planes exceeds the held margin (%d planes): larger irs_slab_config.margin", w, s.margin);
    const int64_t HW = (int64_t)c->vol.H * c->vol.W, V = c->vol.V;
    irs_xfer x[4 * 3 * IRS_MAX_CHAINS];
    int n = 0;
    const int blocks = kind == F_PLANAR3 ? 3 * chains : chains;  // contiguous runs per direction
    const int64_t bstride = kind == F_PLANAR3 ? V : (kind == F_AOS3 ? 3 * V : V);
    const int64_t pe = kind == F_AOS3 ? 3 * HW : HW;  // elements per plane of a run
    auto add = [&](int z0, int peer, int recv) {
        for (int b = 0; b < blocks; ++b) x[n++] = irs_xfer{virt + b * bstride + (int64_t)z0 * pe, (size_t)w * pe * sizeof(float), peer, recv};
    };
    // same order on both sides of a link: the i-th send of rank r to r + 1 meets the i-th receive of r + 1 from r
    if (s.has_hi) {
        add(s.b - w, s.rank + 1, 0);
        add(s.b, s.rank + 1, 1);
    }
    if (s.has_lo) {
        add(s.a, s.rank - 1, 0);
        add(s.a - w, s.rank - 1, 1);
    }
    if (comm_exchange(c->comm, x, n, cs)) return 1;
    c->slab_exchanges += 1;
    c->slab_exchanged_bytes += (uint64_t)(s.has_hi + s.has_lo) * blocks * w * pe * sizeof(float);
    return 0;
}

// events of the pipeline: a rotating pool (an event may be re-recorded once the waits that named it have been enqueued)
struct Pipe {
    irs_ctx* c;
    hipStream_t st, cs;
    int next = 0;
    hipEvent_t take() { return c->sev[(next++) % 16]; }
};

// exchange `w` planes of a buffer whose boundary strips are final on `st` NOW; returns the event `st` must wait for before
// it reads the ghost planes (nullptr: nothing was exchanged)
int start_exchange(Pipe& p, float* virt, int kind, int chains, int w, hipEvent_t* recv_ev) {
    *recv_ev = nullptr;
    if (p.c->sl.world == 1 || w <= 0) return 0;
    hipEvent_t prod = p.take(), recv = p.take();
    HIP_TRY(hipEventRecord(prod, p.st));
    HIP_TRY(hipStreamWaitEvent(p.cs, prod, 0));
    if (exchange_planes(p.c, virt, kind, chains, w, p.cs)) return 1;
    HIP_TRY(hipEventRecord(recv, p.cs));
    *recv_ev = recv;
    return 0;
}
int wait_exchange(Pipe& p, hipEvent_t recv_ev) {
    if (recv_ev) HIP_TRY(hipStreamWaitEvent(p.st, recv_ev, 0));
    return 0;
}
// small all-reduce of `buf`, which `st` has just produced; returns the event to wait for before `st` reads it again.
// `slot` names a dedicated event pair: these results are waited for much later than the exchanges in between.
enum { AR_ENERGY = 0, AR_DMAX, AR_NLL, AR_STATS };
int start_allreduce(Pipe& p, void* buf, size_t count, int max_u32, int slot, hipEvent_t* done) {
    *done = nullptr;
    if (p.c->sl.world == 1) return 0;
    hipEvent_t prod = p.c->sev[16 + 2 * slot], fin = p.c->sev[17 + 2 * slot];
    HIP_TRY(hipEventRecord(prod, p.st));
    HIP_TRY(hipStreamWaitEvent(p.cs, prod, 0));
    if (comm_allreduce(p.c->comm, buf, count, max_u32, p.cs)) return 1;
    HIP_TRY(hipEventRecord(fin, p.cs));
    *done = fin;
    return 0;
}

// all-reduce whose result `st` needs at once (one-off stages, the measuring mode)
int allreduce_now(Pipe& p, void* buf, size_t count, int max_u32) {
    hipEvent_t done;
    if (start_allreduce(p, buf, count, max_u32, AR_STATS, &done)) return 1;
    if (done) HIP_TRY(hipStreamWaitEvent(p.st, done, 0));
    return 0;
}

// windows of one step of a round: e = planes beyond the slab its output must cover, r = how far into the slab outputs
// depend on ghost planes.  Sides without a neighbour have neither.
struct StepWin {
    Vol interior, boundary;
};
StepWin step_windows(const irs_ctx* c, int e, int r, bool split) {
    const SlabInfo& s = c->sl;
    const int elo = s.has_lo ? e : 0, ehi = s.has_hi ? e : 0;
    const int ilo = s.a + (s.has_lo ? r : 0), ihi = s.b - (s.has_hi ? r : 0);
    StepWin w;
    if (!split || s.world == 1 || ihi <= ilo) {  // one launch over everything (counted as "boundary": it needs the ghosts)
        w.interior = window(c->vol, 0, 0);
        w.boundary = window(c->vol, s.a - elo, s.b + ehi);
        if (!split || s.world == 1) {
            w.interior = w.boundary;
            w.boundary = window(c->vol, 0, 0);
        }
        return w;
    }
    w.interior = window(c->vol, ilo, ihi);
    w.boundary = window2(c->vol, s.has_lo ? s.a - e : 0, s.has_lo ? ilo : 0, s.has_hi ? ihi : 0, s.has_hi ? s.b + e : 0);
    return w;
}

void fwd_step(irs_ctx* c, const Views& v, const float* vs, int k, int chains, Vol w, hipStream_t st) {
    if (w.nz + w.nzb <= 0) return;
    const float* in = k == 0 ? vs : step_buf(c, v, k - 1);
    launch_exp_step_fwd_march(in, step_buf(c, v, k), k == 0, c->cfg.no_steps, chains, w, c->lin.lin(), c->dmax + (int64_t)k * c->C * 4,
                              c->dmax + (int64_t)(k + 1) * c->C * 4, predicted_small(c, k), fwd_lay(c, k), st);
}

// `hplan`: the ghost width this transition was planned with for step k (0: unknown, launch every variant).  The plan is
// validated against the measured bound afterwards (validate_widths_kernel: floor(max|d_k|) + 1 <= hplan, else the transition
// is flagged invalid), so the variants that can only be selected above it need not be launched at all: hplan = 1 leaves the
// radius-1 gather alone -- 24 idle launches less per transition than the fused path, which has no such guarantee.
void bwd_step(irs_ctx* c, const Views& v, const float* vs, int k, int hplan, Vol w, hipStream_t st) {
    if (w.nz + w.nzb <= 0) return;
    const int lay = bwd_lay(c, k);
    float* gi = grad_raw(c, k, true);
    float* go = grad_raw(c, k, false);
    const float* G = (lay & 2) ? aos(gi, v) : planar(gi, v);
    float* out = (lay & 4) ? aos(go, v) : planar(go, v);
    const float* dk = k == 0 ? vs : step_buf(c, v, k - 1);
    const unsigned* dm = c->dmax + (int64_t)k * c->C * 4;
    const bool skip_any = (hplan >= 1 && hplan <= 2) || predicted_below(c, k, 1.5f);
    launch_exp_step_bwd_march(G, dk, out, k == 0, c->cfg.no_steps, c->C, w, c->lin.lin(), dm, hplan == 1 ? 1 : 2, skip_any, nullptr, lay, nullptr, st);
    // the any-radius kernel bounds its sources by the global bound around the tile (no coarse grid: that one spans the volume)
    if (!skip_any) launch_exp_step_bwd_lds(G, dk, out, k == 0, c->cfg.no_steps, c->C, w, c->lin.lin(), dm, 2, 2, nullptr, lay, nullptr, st);
}

// one round: [exchange of `xbuf`] + steps ks[0..m) in execution order with ghost widths hs[]
template <typename StepFn>
int run_round(Pipe& p, float* xbuf, int xkind, int chains, int w, const int* ks, const int* hs, int m, StepFn step) {
    irs_ctx* c = p.c;
    hipEvent_t recv = nullptr;
    if (start_exchange(p, xbuf, xkind, chains, w, &recv)) return 1;
    const bool split = recv != nullptr;
    StepWin wins[kMaxSteps];
    int r = 0;
    for (int i = 0; i < m; ++i) {
        r += hs[i];
        wins[i] = step_windows(c, w - r, r, split);
        step(ks[i], wins[i].interior);
    }
    if (split) {
        if (wait_exchange(p, recv)) return 1;
        for (int i = 0; i < m; ++i) step(ks[i], wins[i].boundary);
    }
    LAUNCH_CHECK();
    return 0;
}

// validation of the planned ghost widths against the (all-reduced) bounds of this transition; sticky flag in pinned memory
struct Used {
    int h[kMaxSteps];
};
__global__ void validate_widths_kernel(const unsigned* __restrict__ dmax, Used used, int n, int C, unsigned* __restrict__ flags) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned bad = 0;
    for (int k = 0; k < n; ++k) {
        float m = 0.0f;
        for (int i = 0; i < C * 4; ++i) m = fmaxf(m, __uint_as_float(dmax[k * C * 4 + i]));
        if (!(m >= 0.0f) || (int)floorf(m) + 1 > used.h[k]) bad = 1;
    }
    if (bad) flags[0] += 1;
}

int ghost_width_from_bound(float m, bool safety) {
    if (!(m >= 0.0f) || m > 1.0e6f) return -1;
    return (int)floorf(safety ? 1.25f * m + 0.25f : m) + 1;
}

// global bound (all chains, all axes) of d_k as the host last saw it
float hint_bound(const irs_ctx* c, int k) {
    const volatile unsigned* h = c->hint + (size_t)k * c->C * 4;
    float m = 0.0f;
    for (int i = 0; i < c->C * 4; ++i) {
        const unsigned bits = h[i];
        float f;
        memcpy(&f, &bits, sizeof(f));
        if (!(f >= 0.0f)) return INFINITY;
        m = f > m ? f : m;
    }
    return m;
}

struct FwdOpts {
    bool noise, jitter, energy;
    int chains;
};

// perturbation .. residual on the slab.  `plan` in: predicted widths (exact == false) / out: measured widths (exact == true)
int slab_forward(irs_ctx* c, Pipe& p, const irs_io& io, const float* v_src, float* vs, float* warped, float* z, Plan& plan,
                 bool exact, const FwdOpts& o, hipEvent_t* energy_done) {
    const irs_config& cfg = c->cfg;
    const SlabInfo& s = c->sl;
    const Views v = views(c);
    const int n = cfg.no_steps, C = o.chains, D = c->vol.D;
    hipStream_t st = p.st;
    const Lin lin = c->lin.lin();
    const uint64_t* it = &c->state->st.iteration;
    auto W = [&](int e) { return window(c->vol, s.a - (s.has_lo ? e : 0), s.b + (s.has_hi ? e : 0)); };
    float* noisy = planar(c->tmpA, v);

    if (!c->dmax_clean) HIP_TRY(hipMemsetAsync(c->dmax, 0, sizeof(unsigned) * 4 * c->C * (n + 1), st));
    c->dmax_clean = false;
    // the first forward round lives off ghost planes of v_s that the smoothing stage computes itself from a wider exchange
    // of the perturbed velocity
    const int e0 = exact ? 1 : (plan.fw[0] > 1 ? plan.fw[0] : 1);
    float* first = cfg.sobolev_s > 0 ? noisy : vs;
    if (o.noise) launch_perturb(v_src, io.sigma, io.eps, (float)sqrt(2.0 * (double)cfg.lr), first, C, W(0), cfg.seed, 0, it, st);
    else {
        for (int ch = 0; ch < 3 * C; ++ch)  // own planes of every channel
            HIP_TRY(hipMemcpyAsync(first + (int64_t)ch * c->vol.V + (int64_t)s.a * c->vol.H * c->vol.W,
                                   v_src + (int64_t)ch * c->vol.V + (int64_t)s.a * c->vol.H * c->vol.W,
                                   (size_t)(s.b - s.a) * c->vol.H * c->vol.W * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    {
        hipEvent_t recv;
        if (start_exchange(p, first, F_PLANAR3, C, cfg.sobolev_s + e0, &recv) || wait_exchange(p, recv)) return 1;
    }
    if (cfg.sobolev_s > 0) launch_sobolev_march(noisy, vs, c->sob, C * 3, W(e0), c->dmax, n, st);
    else launch_field_absmax(vs, true, n, c->dmax, C, W(e0), st);
    if (o.energy) {
        launch_reg_energy(vs, c->energy_partials, C, W(0), st);
        launch_reduce_partials(c->energy_partials, energy_blocks(W(0)), C, c->energy_sum, st);
        if (start_allreduce(p, c->energy_sum, C, 0, AR_ENERGY, energy_done)) return 1;
    }
    LAUNCH_CHECK();

    auto fstep = [&](int k, Vol w) { fwd_step(c, v, vs, k, C, w, st); };
    if (exact) {
        // measure: the bound of d_k is all-reduced and read back before step k (one host synchronisation per step)
        plan.n = n;
        for (int k = 0; k < n; ++k) {
            unsigned* dm = c->dmax + (int64_t)k * c->C * 4;
            if (allreduce_now(p, dm, (size_t)c->C * 4, 1)) return 1;
            unsigned host[4 * IRS_MAX_CHAINS];
            HIP_TRY(hipMemcpyAsync(host, dm, sizeof(unsigned) * 4 * c->C, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            float m = 0.0f;
            for (int i = 0; i < 4 * c->C; ++i) {
                float f;
                memcpy(&f, &host[i], sizeof(f));
                if (!(f >= 0.0f)) return fail("slab: displacement bound of step %d is not finite", k);
                m = f > m ? f : m;
            }
            const int h = ghost_width_from_bound(m, false);
            if (h < 1) return fail("slab: displacement bound of step %d out of range", k);
            if (k == 0 && h > 1) return fail("slab: |d_0| >= 1 voxel: velocity field too large for %d squaring steps", n);
            plan.h[k] = h;
            plan.m[k] = m;
            plan.fr[k] = k;
            plan.fw[k] = h;
            if (k == 0) {  // reads v_s, whose ghost plane the smoothing stage made
                fstep(0, W(0));
                LAUNCH_CHECK();
            } else {
                const int ks[1] = {k}, hs[1] = {h};
                if (run_round(p, step_buf(c, v, k - 1), step_is_aos(c, k - 1) ? F_AOS3 : F_PLANAR3, C, h, ks, hs, 1, fstep)) return 1;
            }
        }
        plan.nf = n;
    } else {
        for (int r = 0; r < plan.nf; ++r) {
            int ks[kMaxSteps], hs[kMaxSteps], m = 0;
            for (int k = 0; k < n; ++k)
                if (plan.fr[k] == r) {
                    ks[m] = k;
                    hs[m++] = plan.h[k];
                }
            const int k0 = ks[0];
            if (r == 0) {
                // no exchange: v_s is valid on slab +- e0 >= fw[0]; single launches on the shrinking windows
                int rr = 0;
                for (int i = 0; i < m; ++i) {
                    rr += hs[i];
                    fstep(ks[i], W(plan.fw[0] - rr));
                }
                LAUNCH_CHECK();
            } else if (run_round(p, step_buf(c, v, k0 - 1), step_is_aos(c, k0 - 1) ? F_AOS3 : F_PLANAR3, C, plan.fw[r], ks, hs, m, fstep))
                return 1;
        }
    }

    const float* d_last = step_buf(c, v, n - 1);
    if (io.transformation || io.displacement) launch_svf_outputs(d_last, io.transformation, io.displacement, C, W(0), lin, st);
    const float alpha = o.jitter && cfg.uniform_alpha > 0.0f ? cfg.uniform_alpha : 0.0f;
    launch_warp_fwd(io.moving_im, io.moving_chains == 1 ? 0 : c->vol.Vg, d_last, io.unif, alpha, warped, nullptr, 0, C, W(0), lin, cfg.seed,
                    0, it, st);
    if (cfg.data_loss == IRS_DATA_GMM_LCC) {
        const int ls = cfg.lcc_s;
        hipEvent_t recv;
        if (start_exchange(p, warped, F_IMAGE, C, 4 * ls, &recv) || wait_exchange(p, recv)) return 1;
        launch_lcc_fwd_march(planar(c->fhat, v), c->fhat_chains == 1 ? 0 : c->vol.V, warped, z, planar(c->sigM, v), ls, C, W(2 * ls), st);
    } else {
        const int e = cfg.virtual_decimation ? 1 : 0;  // the lag-1 products of the VD statistics read one plane up
        hipEvent_t recv;
        if (start_exchange(p, warped, F_IMAGE, C, e, &recv) || wait_exchange(p, recv)) return 1;
        launch_residual_ssd(io.fixed_im, io.fixed_chains == 1 ? 0 : c->vol.V, warped, z, C, window(c->vol, s.a, s.b + (s.has_hi ? e : 0)), st);
    }
    (void)D;
    LAUNCH_CHECK();
    return 0;
}

// irs_io with every slab-local pointer shifted to its virtual base
irs_io shifted_io(const irs_ctx* c, const irs_io* io) {
    const Views v = views(c);
    irs_io o = *io;
    o.fixed_im = planar(io->fixed_im, v);
    o.mask = planar(io->mask, v);
    o.v = planar(io->v, v);
    o.sigma = planar(io->sigma, v);
    o.eps = planar(io->eps, v);
    o.unif = planar(io->unif, v);
    o.curr_state = planar(io->curr_state, v);
    o.im_moving_warped = planar(io->im_moving_warped, v);
    o.residuals = planar(io->residuals, v);
    o.displacement = planar(io->displacement, v);
    o.transformation = planar(io->transformation, v);
    o.grad_v = planar(io->grad_v, v);
    return o;  // moving_im stays: it is the whole volume
}

int stats_for_chain(irs_ctx* c, Pipe& p, const irs_io& io, const float* z, int ch, int want_vd, int op) {
    const SlabInfo& s = c->sl;
    const Vol w0 = window(c->vol, s.a, s.b);
    const uint8_t* mask = io.mask + (io.mask_chains == 1 ? 0 : (int64_t)ch * c->vol.V);
    launch_stats(want_vd, z + (int64_t)ch * c->vol.V, mask, c->state, c->stat_partials, w0, p.st);
    launch_reduce_cols(c->stat_partials, stats_blocks(w0), kStatVals, c->stat_sum, p.st);
    hipEvent_t done;
    if (start_allreduce(p, c->stat_sum, kStatVals, 0, AR_STATS, &done)) return 1;
    if (done) HIP_TRY(hipStreamWaitEvent(p.st, done, 0));
    launch_chain_scalar(c->state, c->stat_sum, 1, ch, op, c->dcfg, p.st);
    LAUNCH_CHECK();
    return 0;
}

}  // namespace

namespace irs {
void slab_release(irs_ctx* c) {
    if (!c) return;
    for (int i = 0; i < 24; ++i)
        if (c->sev[i]) (void)hipEventDestroy(c->sev[i]);
    if (c->cs) (void)hipStreamDestroy(c->cs);
}
}  // namespace irs

extern "C" {

int irs_slab_plan_rounds(const int32_t* h, int n, int ghost_max, int min_slab, int32_t* fwd_round, int32_t* fwd_width,
                         int32_t* n_fwd, int32_t* bwd_round, int32_t* bwd_width, int32_t* n_bwd) {
    if (!h || n < 1 || n > kMaxSteps || !fwd_round || !fwd_width || !n_fwd || !bwd_round || !bwd_width || !n_bwd)
        return fail("irs_slab_plan_rounds: bad arguments");
    Plan p;
    p.n = n;
    for (int k = 0; k < n; ++k) p.h[k] = h[k];
    if (plan_rounds(p, ghost_max > 0 ? ghost_max : 4, min_slab)) return 1;
    for (int k = 0; k < n; ++k) {
        fwd_round[k] = p.fr[k];
        bwd_round[k] = p.br[k];
    }
    for (int r = 0; r < p.nf; ++r) fwd_width[r] = p.fw[r];
    for (int r = 0; r < p.nb; ++r) bwd_width[r] = p.bw[r];
    *n_fwd = p.nf;
    *n_bwd = p.nb;
    return 0;
}

int irs_slab_plan_layout(const irs_config* cfg, const irs_slab_config* scfg, int rank, int world, irs_slab_layout* out) {
    if (!out) return fail("irs_slab_plan_layout: null argument");
    SlabInfo s;
    if (plan_layout(cfg, scfg, rank, world, &s)) return 1;
    *out = irs_slab_layout{s.rank, s.world, s.a, s.b, s.lo, s.hi, s.margin, s.gmax};
    return 0;
}

int irs_slab_create(const irs_config* cfg, const irs_slab_config* scfg, irs_comm* comm, irs_ctx** out) {
    if (!cfg || !out) return fail("irs_slab_create: null argument");
    const int world = comm ? comm->world : 1, rank = comm ? comm->rank : 0;
    if (cfg->cps[0] || cfg->cps[1] || cfg->cps[2]) return fail("irs_slab_create: the slab decomposition supports SVF_3D only");
    if (!use_lds_exp()) return fail("irs_slab_create: needs the LDS squaring kernels (IRS_EXP_LDS=1)");
    if (cfg->no_steps > kMaxSteps) return fail("irs_slab_create: at most %d squaring steps", kMaxSteps);
    SlabInfo s;
    if (plan_layout(cfg, scfg, rank, world, &s)) return 1;
    const int ls = cfg->data_loss == IRS_DATA_GMM_LCC ? cfg->lcc_s : 0;
    if (world > 1 && (s.margin < cfg->sobolev_s + 1 || s.margin < 4 * ls || s.margin < 2))
        return fail("irs_slab_create: margin %d too small for the stencils (Sobolev %d, LCC %d)", s.margin, cfg->sobolev_s, ls);
    if (world > 1 && (s.min_slab < cfg->sobolev_s + 1 || s.min_slab < 4 * ls))
        return fail("irs_slab_create: slabs of %d planes are thinner than the stencil halos: fewer ranks", s.min_slab);
    irs_ctx* c = nullptr;
    if (create_ctx(cfg, &s, &c)) return 1;
    c->comm = comm;
    hipError_t e = hipStreamCreateWithFlags(&c->cs, hipStreamNonBlocking);
    for (int i = 0; i < 24 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->sev[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        irs_destroy(c);
        return fail("irs_slab_create: stream / event creation failed: %s", hipGetErrorString(e));
    }
    c->have_pred = false;
    *out = c;
    return 0;
}

int irs_slab_get_layout(const irs_ctx* c, irs_slab_layout* out) {
    if (!c || !out || !c->sl.on) return fail("irs_slab_get_layout: not a slab context");
    const SlabInfo& s = c->sl;
    *out = irs_slab_layout{s.rank, s.world, s.a, s.b, s.lo, s.hi, s.margin, s.gmax};
    return 0;
}

int irs_slab_status_get(irs_ctx* c, irs_slab_status* out, void* stream) {
    if (!c || !out || !c->sl.on) return fail("irs_slab_status_get: not a slab context");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize(c->cs));
    memset(out, 0, sizeof(*out));
    out->transitions = c->n_enqueued;
    out->exact_transitions = c->slab_exact;
    out->exchanges = c->slab_exchanges;
    out->exchanged_bytes = c->slab_exchanged_bytes;
    out->mispredictions = c->hint[kHintWords - 8];
    out->last_fwd_rounds = c->last_nf;
    out->last_bwd_rounds = c->last_nb;
    return 0;
}

int irs_slab_transition(irs_ctx* c, const irs_io* io_in, void* stream) {
    if (check_io(c, io_in, "irs_slab_transition")) return 1;
    if (!c->sl.on) return fail("irs_slab_transition: not a slab context (irs_slab_create)");
    if (!io_in->v) return fail("irs_slab_transition: v is required");
    const irs_config& cfg = c->cfg;
    const SlabInfo& s = c->sl;
    hipStream_t st = (hipStream_t)stream;
    const int C = c->C, n = cfg.no_steps;
    const Views v = views(c);
    const irs_io io = shifted_io(c, io_in);
    if (c->hint[kHintWords - 8]) return fail("irs_slab_transition: an earlier transition ran with ghost zones narrower than its displacement needed (results invalid)");
    {   // bounded run-ahead, as irs_transition: the width plan below reads bounds at most two transitions old
        const int depth = env_int("IRS_RUN_AHEAD", 2);
        if (depth > 0 && depth <= 3 && c->n_enqueued >= (uint64_t)depth) HIP_TRY(hipEventSynchronize(c->ra_ev[(c->n_enqueued - depth) % 4]));
    }
    Pipe p{c, st, c->cs};
    // ---- plan of the ghost widths
    Plan plan;
    plan.n = n;
    bool exact = !c->have_pred || env_int("IRS_SLAB_EXACT", 0) != 0;
    if (!exact) {
        const bool fresh = c->n_enqueued >= 2;  // the hint holds the all-reduced bounds of a finished transition
        for (int k = 0; k < n && !exact; ++k) {
            const int hh = fresh ? ghost_width_from_bound(hint_bound(c, k), true) : c->pred[k];
            if (hh < 1) exact = true;
            plan.h[k] = hh;
        }
        if (!exact) {
            plan.h[0] = 1;  // |d_0| = |v_s| / 2^n voxels (validated like the others)
            if (plan_rounds(plan, s.gmax, s.world > 1 ? s.min_slab : 1 << 30)) return 1;
        }
    }
    float* vs = io.curr_state ? io.curr_state : planar(c->vs, v);
    float* warped = io.im_moving_warped ? io.im_moving_warped : planar(c->warped, v);
    float* z = io.residuals ? io.residuals : planar(c->z, v);
    const Vol w0 = window(c->vol, s.a, s.b);
    const Lin lin = c->lin.lin();
    const uint64_t* it = &c->state->st.iteration;

    hipEvent_t energy_done = nullptr;
    const FwdOpts fo{true, cfg.uniform_alpha > 0.0f, true, C};
    if (slab_forward(c, p, io, io.v, vs, warped, z, plan, exact, fo, &energy_done)) return 1;
    if (exact) {
        // the forward pass ran one step per round (E_k = h_k): so does the backward pass
        for (int k = 0; k < n; ++k) {
            plan.br[k] = n - 1 - k;
            plan.bw[n - 1 - k] = plan.h[k];
            c->pred[k] = ghost_width_from_bound(plan.m[k], true);
        }
        plan.nb = n;
        c->have_pred = true;
        c->slab_exact += 1;
    }
    c->last_nf = plan.nf;
    c->last_nb = plan.nb;
    // bounds of every d_k, all-reduced: variant selection of the adjoint steps, the next plans, the validation of this one
    hipEvent_t dmax_done = nullptr;
    if (start_allreduce(p, c->dmax, (size_t)4 * C * (n + 1), 1, AR_DMAX, &dmax_done)) return 1;

    // ---- per chain, serially (trainer.py:316-327): VD factor -> GMM step -> data term with the UPDATED mixture
    // (SSD without virtual decimation has alpha = 1 and no mixture: the statistics stage and its all-reduce drop out)
    const bool need_stats = cfg.data_loss == IRS_DATA_GMM_LCC || cfg.virtual_decimation || exact;  // (once, for n_mask)
    for (int ch = 0; ch < C; ++ch) {
        if (need_stats && stats_for_chain(c, p, io, z, ch, cfg.virtual_decimation, 3)) return 1;
        const uint8_t* mask = io.mask + (io.mask_chains == 1 ? 0 : (int64_t)ch * c->vol.V);
        const float* f = cfg.data_loss == IRS_DATA_GMM_LCC ? planar(c->fhat, v) + (c->fhat_chains == 1 ? 0 : (int64_t)ch * c->vol.V) : nullptr;
        double* part = c->nll_partials + (int64_t)ch * c->nll_blocks;
        launch_data_bwd(cfg.data_loss, f, 0, z + (int64_t)ch * c->vol.V, planar(c->sigM, v) + (int64_t)ch * c->vol.V, mask, 0, nullptr, c->state, ch,
                        planar(c->gM, v) + (int64_t)ch * c->vol.V, part, cfg.lcc_s, 1, w0, st);
        launch_reduce_partials(part, data_bwd_blocks(cfg.data_loss, w0), 1, c->nll_sum + ch, st);
    }
    hipEvent_t nll_done = nullptr;
    if (start_allreduce(p, c->nll_sum, C, 0, AR_NLL, &nll_done)) return 1;
    // ---- back through the warp and the squaring steps
    const float* d_last = step_buf(c, v, n - 1);
    launch_warp_bwd(io.moving_im, io.moving_chains == 1 ? 0 : c->vol.Vg, d_last, io.unif, cfg.uniform_alpha > 0.0f ? cfg.uniform_alpha : 0.0f,
                    planar(c->gM, v), planar(c->gA, v), C, w0, lin, cfg.seed, 0, it, st);
    LAUNCH_CHECK();
    if (dmax_done) HIP_TRY(hipStreamWaitEvent(st, dmax_done, 0));
    auto bstep = [&](int k, Vol w) { bwd_step(c, v, vs, k, plan.h[k], w, st); };
    for (int r = 0; r < plan.nb; ++r) {
        int ks[kMaxSteps], hs[kMaxSteps], m = 0;
        for (int k = n - 1; k >= 0; --k)
            if (plan.br[k] == r) {
                ks[m] = k;
                hs[m++] = plan.h[k];
            }
        const int k1 = ks[0];
        float* gi = grad_raw(c, k1, true);
        const bool gaos = (bwd_lay(c, k1) & 2) != 0;
        if (run_round(p, gaos ? aos(gi, v) : planar(gi, v), gaos ? F_AOS3 : F_PLANAR3, C, plan.bw[r], ks, hs, m, bstep)) return 1;
    }
    // ---- regulariser scalars (its all-reduce has been in flight since the smoothing stage), update, bookkeeping
    if (energy_done) HIP_TRY(hipStreamWaitEvent(st, energy_done, 0));
    launch_reg_scalar(c->state, c->energy_sum, 1, c->dcfg, st);
    float sc3[3];
    prescale_factors(c->vol, n, sc3);
    float* g0 = grad_raw(c, 0, false);
    launch_sgld_update(io.v, io.sigma, planar(g0, v), vs, c->state, cfg.lr, sc3[0], sc3[1], sc3[2], io.grad_v, C, w0, st);
    if (nll_done) HIP_TRY(hipStreamWaitEvent(st, nll_done, 0));
    Used used;
    for (int k = 0; k < kMaxSteps; ++k) used.h[k] = k < n ? plan.h[k] : 0;
    hipLaunchKernelGGL(validate_widths_kernel, dim3(1), dim3(64), 0, st, c->dmax, used, n, C, c->hint + (kHintWords - 8));
    launch_finalize(c->state, c->nll_sum, 1, c->dcfg, true, c->dmax, c->hint, 4 * C * (n + 1), 0u, 0, true, st);
    c->dmax_clean = true;
    LAUNCH_CHECK();
    HIP_TRY(hipEventRecord(c->ra_ev[c->n_enqueued % 4], st));
    ++c->n_enqueued;
    return 0;
}

int irs_slab_gmm_init(irs_ctx* c, const irs_io* io_in, const float* v_sample, int warm_up, void* stream) {
    if (check_io(c, io_in, "irs_slab_gmm_init")) return 1;
    if (!c->sl.on) return fail("irs_slab_gmm_init: not a slab context (irs_slab_create)");
    if (c->cfg.data_loss != IRS_DATA_GMM_LCC) return 0;
    hipStream_t st = (hipStream_t)stream;
    const Views v = views(c);
    const SlabInfo& s = c->sl;
    const irs_io io = shifted_io(c, io_in);
    Pipe p{c, st, c->cs};
    // trainer.py:529-547: one velocity sample (no Langevin noise, no jitter), batch of one
    float* vsrc = planar(c->gB, v);
    const int64_t HW = (int64_t)c->vol.H * c->vol.W;
    for (int ch = 0; ch < 3; ++ch) {
        float* dst = c->gB + (int64_t)ch * c->vol.V + (int64_t)(s.a - s.lo) * HW;
        const size_t bytes = (size_t)(s.b - s.a) * HW * sizeof(float);
        if (v_sample) HIP_TRY(hipMemcpyAsync(dst, v_sample + (int64_t)ch * c->vol.V + (int64_t)(s.a - s.lo) * HW, bytes, hipMemcpyDeviceToDevice, st));
        else HIP_TRY(hipMemsetAsync(dst, 0, bytes, st));
    }
    Plan plan;
    hipEvent_t unused = nullptr;
    irs_io io1 = io;
    io1.sigma = nullptr;
    io1.eps = nullptr;
    io1.unif = nullptr;
    io1.transformation = nullptr;
    io1.displacement = nullptr;
    const FwdOpts fo{false, false, false, 1};
    if (slab_forward(c, p, io1, vsrc, planar(c->vs, v), planar(c->warped, v), planar(c->z, v), plan, true, fo, &unused)) return 1;
    const Vol w0 = window(c->vol, s.a, s.b);
    const float* z = planar(c->z, v);
    launch_masked_moments(z, io.mask, c->stat_partials, w0, st);
    launch_reduce_cols(c->stat_partials, stats_blocks(w0), 3, c->stat_sum, st);
    if (allreduce_now(p, c->stat_sum, 3, 0)) return 1;
    launch_gmm_init_from_moments(c->state, c->stat_sum, 1, c->dcfg, st);
    LAUNCH_CHECK();
    auto stats = [&](int want_vd, int op) {
        launch_stats(want_vd, z, io.mask, c->state, c->stat_partials, w0, st);
        launch_reduce_cols(c->stat_partials, stats_blocks(w0), kStatVals, c->stat_sum, st);
        if (allreduce_now(p, c->stat_sum, kStatVals, 0)) return 1;
        launch_chain_scalar(c->state, c->stat_sum, 1, 0, op, c->dcfg, st);
        return 0;
    };
    if (stats(c->cfg.virtual_decimation, 1)) return 1;  // alpha, fixed below
    for (int i = 0; i < warm_up; ++i)
        if (stats(0, 2)) return 1;
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

}  // extern "C"
