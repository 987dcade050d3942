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
#include <stdio.h>
#include <new>
#include <vector>

#include "comm.h"
#include "ctx.h"

using namespace irs;

namespace {

constexpr int kMaxSteps = 32;
// widest ghost zone of one exchange: 8 planes = two forward rounds and six backward rounds for twelve sub-voxel steps (4: four and
// eight).  Measured with real concurrent ranks over the peer-mapped transport (tools/slab_probe.py --transport ipc, 256^3): 4 ranks
// 8.90 -> 8.19 ms, 2 ranks 7.12 -> 6.89 ms per transition; the ghost planes recomputed instead cost nothing measurable
// (one rank of eight without transport: 1.269 -> 1.263 ms).
constexpr int kDefaultGhostMax = 8;

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
int plan_rounds(Plan& p, int gmax, int min_slab, int nbuf = 2) {
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
            // The adjoint rotates through `nbuf` gradient buffers (two; three on a slab of several ranks): step i of a round (i = 1
            // is the top step k1) reads B_{i-1} and writes B_i, indices mod nbuf, B_0 being the buffer the round's exchange carries.
            //  (1) B_0 is being SENT from its planes [a, a + w) until the wait: a step that writes B_0 must start beyond them -- only
            //      the last step of the round does (its interior begins at r_m = w): at most nbuf steps.
            //  (2) the boundary strips of step i run after all interiors and read B_{i-1} up to a + r_i + h_i (r_i = h_1 + .. + h_i);
            //      step j = i - 1 + nbuf writes the interior of that same buffer from a + r_j on: r_i + h_i <= r_j.  For two buffers
            //      that is h_1 <= h_2; for three h_1 <= h_2 + h_3 (tests/test_slab_schedule.py replays exactly this).
            //  (3) planes are only disjoint within ONE layout: adjoint step 0 writes the planar field the update reads, every other
            //      step an interleaved one (ctx.h: bwd_lay) -- in the buffer the round's exchange carries and its first strips still
            //      read, step 0's interior would land on other bytes than "its" planes.  It never closes a full rotation.
            const int m = k1 - c0 + 1;
            bool hazard = m > nbuf || (c0 == 0 && m == nbuf);
            if (!hazard && m == nbuf) {  // (j = i - 1 + nbuf <= m only for i = 1)
                int rj = 0;
                for (int k = k1; k >= c0; --k) rj += p.h[k];
                hazard = 2 * p.h[k1] > rj;
            }
            if (!ok || N > gmax || hazard) break;
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
    // widest read beyond the slab: the smoothing stage on slab +- e0 (e0 <= gmax) reads sobolev_s planes further; the LCC map on
    // slab +- 2 ls reads 2 ls further.  + 1: the adjoint kernel prefetches one plane past the last one it uses.
    int m = cfg->sobolev_s + gmax;
    if (4 * ls > m) m = 4 * ls;
    return m + 1;
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
    sl->min_slab = D;
    for (int r = 0; r < world; ++r) {
        const int n = (int)(((int64_t)(r + 1) * D) / world) - (int)(((int64_t)r * D) / world);
        if (n < sl->min_slab) sl->min_slab = n;
    }
    // The widest exchange a round limit of g planes leads to is the velocity's: sobolev_s + e0 planes with e0 <= g.  An exchange
    // reaches the NEIGHBOUR only, so it must fit the thinnest slab: the limit is lowered to what fits (same on every rank).  It
    // used to be taken as given, and the default of 8 -- or any generous request -- made thin slabs refuse their first exchange
    // ("ghost zone of 14 planes exceeds the smallest slab (13 planes)": tests/test_gpu_slab_fuzz.py).
    sl->gmax = scfg && scfg->ghost_max > 0 ? scfg->ghost_max : kDefaultGhostMax;
    if (world > 1 && sl->gmax > sl->min_slab - cfg->sobolev_s) sl->gmax = sl->min_slab - cfg->sobolev_s > 1 ? sl->min_slab - cfg->sobolev_s : 1;
    sl->margin = scfg && scfg->margin > 0 ? scfg->margin : default_margin(cfg, sl->gmax);
    sl->has_lo = rank > 0;
    sl->has_hi = rank + 1 < world;
    sl->lo = sl->has_lo ? (sl->a - sl->margin > 0 ? sl->a - sl->margin : 0) : 0;
    sl->hi = sl->has_hi ? (sl->b + sl->margin < D ? sl->b + sl->margin : D) : D;
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
// dL/d(d_last) lands in A; adjoint step n-1 writes B, the next one C (or A again where there are only two buffers), ...
inline int grad_buffers(const irs_ctx* c) { return c->gC && c->kn.slab_buffers >= 3 ? 3 : 2; }
inline int grad_slot(const irs_ctx* c, int k, bool input) {
    const int nbuf = grad_buffers(c);
    return (c->cfg.no_steps - k - (input ? 1 : 0)) % nbuf;
}
inline float* grad_raw(const irs_ctx* c, int k, bool input) {
    const int slot = grad_slot(c, k, input);
    return slot == 0 ? c->gA : (slot == 1 ? c->gB : c->gC);
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

// ================================================================================================
// The schedule of a transition as DATA: a pure-host builder emits a list of operations (launches with their output windows,
// exchanges, all-reduces, the waits that tie the two streams together); the executor below interprets it on the GPU, and
// irs_slab_trace hands the same list to the tests, which replay it on the CPU with two gloo ranks (tests/test_slab_schedule.py).
// ================================================================================================
struct Sched {
    const SlabInfo& s;
    const irs_config& cfg;
    Vol vol;
    int chains;
    std::vector<irs_slab_op> ops;
    int next_id = 0;

    bool want_split_ = true;  // Knobs::slab_split (measurements): no interior / boundary split when false
    bool fuse_noise_ = true;  // Knobs::fuse_noise
    bool ffd;   // SVFFD_3D: v / noisy / v_s live on the control grid, whole on every rank; d_0 is the up-sampled DENSE field
    Vol volv;   // the velocity grid (control grid, or the image grid for SVF_3D)

    Sched(const SlabInfo& s_, const irs_config& cfg_, int chains_) : s(s_), cfg(cfg_), vol(make_vol(cfg_.dims[0], cfg_.dims[1], cfg_.dims[2])), chains(chains_) {
        ffd = cfg.cps[0] || cfg.cps[1] || cfg.cps[2];
        volv = ffd ? make_vol(control_points(cfg.dims[0], cfg.cps[0]), control_points(cfg.dims[1], cfg.cps[1]), control_points(cfg.dims[2], cfg.cps[2])) : vol;
    }
    static int control_points(int n, int cps) { return (int)ceil((double)(n - 1) / (double)cps) + 1 + 2; }  // utils/util.py:61-69
    Vol W(int e) const { return window(vol, s.a - (s.has_lo ? e : 0), s.b + (s.has_hi ? e : 0)); }
    int step_buf_id(int k) const { return k < 0 ? (ffd ? IRS_SB_DENSE : IRS_SB_VS) : IRS_SB_STEP0 + k; }
    int nbuf_ = 2;  // gradient buffers of the adjoint (three on a slab of several ranks: plan_rounds)
    // dL/d(d_last) lands in A; adjoint step n-1 writes B, the next one C (or A again), ...
    int grad_id(int k, bool input) const {
        const int slot = (cfg.no_steps - k - (input ? 1 : 0)) % nbuf_;
        return slot == 0 ? IRS_SB_GRAD_A : (slot == 1 ? IRS_SB_GRAD_B : IRS_SB_GRAD_C);
    }
    void launch(int stage, int k, Vol w, int in0, int in1, int reach, int out) {
        if (w.nz + w.nzb <= 0 && stage < IRS_SG_SCALARS) return;  // empty window: nothing to launch
        irs_slab_op o;
        memset(&o, 0, sizeof(o));
        o.kind = IRS_OP_LAUNCH;
        o.stage = stage;
        o.k = k;
        o.lo0 = w.z0;
        o.hi0 = w.z0 + w.nz;
        o.lo1 = w.nzb > 0 ? w.z0b : 0;
        o.hi1 = w.nzb > 0 ? w.z0b + w.nzb : 0;
        o.in0 = in0;
        o.in1 = in1;
        o.reach = reach;
        o.out = out;
        ops.push_back(o);
    }
    int comm(int kind, int what, int width, int k = 0) {  // EXCHANGE of buffer `what` / ALLREDUCE of reduction `what`; returns its id
        irs_slab_op o;
        memset(&o, 0, sizeof(o));
        o.kind = kind;
        o.stage = what;
        o.width = width;
        o.k = k;  // exchange: the squaring step that consumes the ghost planes (names the layout of a gradient buffer)
        o.id = next_id++;
        o.in0 = o.in1 = o.out = -1;
        ops.push_back(o);
        return o.id;
    }
    int exchange(int buf, int width, int k = 0) {
        if (s.world == 1 || width <= 0) return -1;
        return comm(IRS_OP_EXCHANGE, buf, width, k);
    }
    int allreduce(int which) { return s.world == 1 ? -1 : comm(IRS_OP_ALLREDUCE, which, 0); }
    void wait(int id) {
        if (id < 0) return;
        irs_slab_op o;
        memset(&o, 0, sizeof(o));
        o.kind = IRS_OP_WAIT;
        o.id = id;
        o.in0 = o.in1 = o.out = -1;
        ops.push_back(o);
    }

    // windows of one step of a round: e = planes beyond the slab its output must cover, r = how far into the slab outputs
    // depend on ghost planes.  Sides without a neighbour have neither.
    void step_windows(int e, int r, bool split, Vol* interior, Vol* boundary) const {
        const int elo = s.has_lo ? e : 0, ehi = s.has_hi ? e : 0;
        const int ilo = s.a + (s.has_lo ? r : 0), ihi = s.b - (s.has_hi ? r : 0);
        if (!split) {
            *interior = window(vol, s.a - elo, s.b + ehi);
            *boundary = window(vol, 0, 0);
        } else if (ihi <= ilo) {  // the slab is all boundary: one launch, after the ghost planes have arrived
            *interior = window(vol, 0, 0);
            *boundary = window(vol, s.a - elo, s.b + ehi);
        } else {
            *interior = window(vol, ilo, ihi);
            *boundary = window2(vol, s.has_lo ? s.a - e : 0, s.has_lo ? ilo : 0, s.has_hi ? ihi : 0, s.has_hi ? s.b + e : 0);
        }
    }

    // one round: [exchange of `xbuf`, w planes] + steps ks[0..m) in execution order with ghost widths hs[]: the interiors while
    // the exchange is in flight, the boundary strips when the ghost planes have arrived
    void round(bool backward, int xbuf, int w, const int* ks, const int* hs, int m) {
        const int id = exchange(xbuf, w, ks[0]);
        // IRS_SLAB_SPLIT=0 (measurements): no interior / boundary split -- every step one launch after the ghost planes have arrived
        const bool want_split = want_split_;
        const bool split = id >= 0 && want_split;
        if (id >= 0 && !want_split) wait(id);
        Vol in[kMaxSteps], bd[kMaxSteps];
        int r = 0;
        for (int i = 0; i < m; ++i) {
            r += hs[i];
            step_windows(w - r, r, split, &in[i], &bd[i]);
            step(backward, ks[i], hs[i], in[i]);
        }
        if (split) {
            wait(id);
            for (int i = 0; i < m; ++i) step(backward, ks[i], hs[i], bd[i]);
        }
    }
    void step(bool backward, int k, int h, Vol w) {
        if (!backward) launch(IRS_SG_EXP_FWD, k, w, step_buf_id(k - 1), -1, h, step_buf_id(k));
        else launch(IRS_SG_EXP_BWD, k, w, grad_id(k, true), step_buf_id(k - 1), h, grad_id(k, false));
    }

    // perturbation, smoothing (on slab +- e0, fed by a wider exchange of the perturbed velocity), regulariser energy
    void head(bool noise, bool energy, int e0, int* energy_ar) {
        const int sb = cfg.sobolev_s;
        const int first = sb > 0 ? IRS_SB_NOISY : IRS_SB_VS;
        if (ffd) {
            // the control grid is whole on every rank: no exchange, no all-reduce (every rank computes the same numbers); the
            // dense velocity is up-sampled on the planes the first forward round reads
            *energy_ar = -1;
            launch(noise ? IRS_SG_PERTURB : IRS_SG_COPY_V, 0, volv, -1, -1, 0, -1);
            launch(IRS_SG_SMOOTH, 0, volv, -1, -1, 0, -1);
            // (the regulariser's scalar stage steps hyper-parameters: it runs behind the verdict about this transition's plan, in
            // backward_and_update -- here the all-reduced bounds it is judged by do not exist yet)
            if (energy) launch(IRS_SG_ENERGY, 0, volv, -1, -1, 0, -1);
            launch(IRS_SG_FFD_UP, 0, W(e0), -1, -1, 0, IRS_SB_DENSE);
            return;
        }
        if (noise && sb > 0 && fuse_noise_) {
            // the smoothing kernel generates the noise itself while staging (Philox keyed by the GLOBAL voxel index: a ghost plane
            // draws what its owner draws): what travels is the velocity, and the perturbed field is never materialised
            wait(exchange(IRS_SB_V, sb + e0));
            launch(IRS_SG_SMOOTH, 1, W(e0), IRS_SB_V, -1, sb, IRS_SB_VS);   // k = 1: fused perturbation
        } else {
            launch(noise ? IRS_SG_PERTURB : IRS_SG_COPY_V, 0, W(0), IRS_SB_V, -1, 0, first);
            wait(exchange(first, sb + e0));
            launch(IRS_SG_SMOOTH, 0, W(e0), first, -1, sb, IRS_SB_VS);
        }
        // the energy partial sum of the slab stays local for now: it travels with the data-term sums (IRS_AR_NLL carries both:
        // three all-reduces per transition -- bounds, statistics, loss terms -- instead of four)
        *energy_ar = -1;
        if (energy) launch(IRS_SG_ENERGY, 0, W(0), IRS_SB_VS, -1, 1, -1);
    }
    void forward_planned(const Plan& p) {
        const int n = cfg.no_steps;
        for (int r = 0; r < p.nf; ++r) {
            int ks[kMaxSteps], hs[kMaxSteps], m = 0;
            for (int k = 0; k < n; ++k)
                if (p.fr[k] == r) {
                    ks[m] = k;
                    hs[m++] = p.h[k];
                }
            if (r == 0) {  // no exchange: v_s is valid on slab +- e0 >= fw[0]; single launches on the shrinking windows
                int rr = 0;
                for (int i = 0; i < m; ++i) {
                    rr += hs[i];
                    step(false, ks[i], hs[i], W(p.fw[0] - rr));
                }
            } else round(false, step_buf_id(ks[0] - 1), p.fw[r], ks, hs, m);
        }
    }
    void forward_exact_step(int k, int h) {  // measuring mode: one step per round, width just read back
        if (k == 0) step(false, 0, h, W(0));
        else round(false, step_buf_id(k - 1), h, &k, &h, 1);
    }
    // outputs, warp, residual
    void middle(bool outputs) {
        const int n = cfg.no_steps;
        if (outputs) launch(IRS_SG_OUTPUTS, 0, W(0), step_buf_id(n - 1), -1, 0, -1);
        launch(IRS_SG_WARP, 0, W(0), step_buf_id(n - 1), -1, 0, IRS_SB_WARPED);
        if (cfg.data_loss == IRS_DATA_GMM_LCC) {
            const int ls = cfg.lcc_s;
            wait(exchange(IRS_SB_WARPED, 4 * ls));
            launch(IRS_SG_RESIDUAL, 0, W(2 * ls), IRS_SB_WARPED, -1, 2 * ls, IRS_SB_Z);
        } else {
            const int e = cfg.virtual_decimation ? 1 : 0;  // the lag-1 products of the VD statistics read one plane up
            wait(exchange(IRS_SB_WARPED, e));
            launch(IRS_SG_RESIDUAL, 0, window(vol, s.a, s.b + (s.has_hi ? e : 0)), IRS_SB_WARPED, -1, 0, IRS_SB_Z);
        }
    }
    // statistics / mixture step / data term per chain, backward warp, the adjoint rounds, update
    void backward_and_update(const Plan& p, bool stats, int energy_ar) {
        const int n = cfg.no_steps, ls = cfg.data_loss == IRS_DATA_GMM_LCC ? cfg.lcc_s : 0;
        // bounds of every d_k, all-reduced: variant selection of the adjoint steps, the next plans, the validation of this one
        const int dmax_ar = allreduce(IRS_AR_DMAX);
        // The first scalar stage evaluates the verdict about this transition's ghost-width plan from the all-reduced bounds
        // (every rank arrives at the same one); the bounds travel on the communication stream ahead of the statistics.
        if (!stats) {
            wait(dmax_ar);
            launch(IRS_SG_VERDICT, 0, W(0), -1, -1, 0, -1);
        }
        for (int ch = 0; ch < chains; ++ch) {
            if (stats) {
                launch(IRS_SG_STATS, ch, W(0), IRS_SB_Z, -1, cfg.virtual_decimation ? 1 : 0, -1);
                wait(allreduce(IRS_AR_STATS));
                if (ch == 0) wait(dmax_ar);
                launch(IRS_SG_CHAIN_SCALAR, ch, W(0), -1, -1, 0, -1);
            }
            launch(IRS_SG_DATA_BWD, ch, W(0), IRS_SB_Z, -1, 2 * ls, IRS_SB_GM);
        }
        // SVFFD: the energy is whole on every rank (head); its scalar stage -- an Adam step on the regulariser's parameters -- must
        // see the verdict, which the stages above have waited for: a dropped transition then leaves them alone
        if (ffd) launch(IRS_SG_REG_SCALAR, 0, volv, -1, -1, 0, -1);
        const int nll_ar = allreduce(IRS_AR_NLL);
        launch(IRS_SG_WARP_BWD, 0, W(0), IRS_SB_GM, step_buf_id(n - 1), 0, IRS_SB_GRAD_A);
        for (int r = 0; r < p.nb; ++r) {
            int ks[kMaxSteps], hs[kMaxSteps], m = 0;
            for (int k = n - 1; k >= 0; --k)
                if (p.br[k] == r) {
                    ks[m] = k;
                    hs[m++] = p.h[k];
                }
            round(true, grad_id(ks[0], true), p.bw[r], ks, hs, m);
        }
        if (ffd) {
            // adjoint of the up-sampling over the owned planes -> partial control-grid gradient -> whole; the update then runs on
            // the whole control grid on every rank (replicated state stays replicated)
            launch(IRS_SG_FFD_ADJ, 0, W(0), grad_id(0, false), -1, 0, -1);
            wait(allreduce(IRS_AR_CPGRAD));
            launch(IRS_SG_UPDATE, 0, volv, -1, -1, 0, -1);
            wait(nll_ar);
            launch(IRS_SG_FINALIZE, 0, W(0), -1, -1, 0, -1);
            return;
        }
        // regulariser scalars (the energies came with the data-term sums), update, bookkeeping
        (void)energy_ar;
        wait(nll_ar);
        launch(IRS_SG_REG_SCALAR, 0, W(0), -1, -1, 0, -1);
        launch(IRS_SG_UPDATE, 0, W(0), grad_id(0, false), IRS_SB_VS, 1, IRS_SB_V);
        launch(IRS_SG_FINALIZE, 0, W(0), -1, -1, 0, -1);
    }
};

// ---- executor --------------------------------------------------------------------------------------------------------------
struct Exec {
    irs_ctx* c;
    hipStream_t st, cs;
    const irs_io& io;     // shifted
    const float* v_src;   // velocity the forward pass starts from (io.v, or the staged sample of the mixture initialisation)
    float *vs, *warped, *z;
    const Plan* plan;     // ghost widths of the adjoint steps (variant selection); may be null before they are known
    bool jitter;
    int chains;
    int next_ev = 0;
    hipEvent_t pending[128];  // event to wait for, by comm id (ids of one transition are < 128)

    hipEvent_t take() { return c->sev[(next_ev++) % 16]; }
    // timeline sampling (irs_slab_timeline_arm): a timing event on `s`, or -1 when this transition is not sampled / the pool is out
    bool tl = false;
    int stamp(hipStream_t s) {
        if (!tl || c->tl_ev_used >= c->tl_ev_n) return -1;
        const int i = c->tl_ev_used++;
        return hipEventRecord(c->tl_ev[i], s) == hipSuccess ? i : -1;
    }
    void tl_open(int id, int kind, int stage, int k, int width, int eP) {
        if (!tl || id < 0 || id >= 128) return;
        c->tl_rec[id] = irs_ctx::TlRec{kind, stage, k, width, eP, -1, -1, -1};
        if (id + 1 > c->tl_n) c->tl_n = id + 1;
    }
    float* buffer(int id, const Views& v, int* kind) const {
        const int n = c->cfg.no_steps;
        switch (id) {
            case IRS_SB_V: *kind = F_PLANAR3; return const_cast<float*>(v_src);
            case IRS_SB_NOISY: *kind = F_PLANAR3; return planar(c->tmpA, v);
            case IRS_SB_VS: *kind = F_PLANAR3; return vs;
            case IRS_SB_WARPED: *kind = F_IMAGE; return warped;
            case IRS_SB_Z: *kind = F_IMAGE; return z;
            case IRS_SB_GM: *kind = F_IMAGE; return planar(c->gM, v);
            case IRS_SB_DENSE: *kind = F_PLANAR3; return planar(c->dense, v);
            case IRS_SB_GRAD_A:
            case IRS_SB_GRAD_B:
            case IRS_SB_GRAD_C: {
                // the layout of a gradient buffer is that of the adjoint step that READS it next
                float* raw = id == IRS_SB_GRAD_A ? c->gA : (id == IRS_SB_GRAD_B ? c->gB : c->gC);
                const bool a = cur_bwd_k >= 0 && cur_bwd_k < n && (bwd_lay(c, cur_bwd_k) & 2) != 0;
                *kind = a ? F_AOS3 : F_PLANAR3;
                return a ? aos(raw, v) : planar(raw, v);
            }
            default:
                if (id >= IRS_SB_STEP0 && id < IRS_SB_STEP0 + n) {
                    *kind = step_is_aos(c, id - IRS_SB_STEP0) ? F_AOS3 : F_PLANAR3;
                    return step_buf(c, v, id - IRS_SB_STEP0);
                }
                *kind = -1;
                return nullptr;
        }
    }
    int cur_bwd_k = -1;  // the adjoint step whose input buffer an exchange is about to carry
    // the planned ghost widths as assumptions the device validates (scalar_kernels.h: Verdict); nothing planned -> nothing assumed
    Verdict verdict() const {
        Verdict vd = no_verdict();
        if (!plan || !planned_) return vd;
        vd.bounds = c->dmax;
        vd.n = c->cfg.no_steps;
        vd.C = c->C;
        for (int k = 0; k < vd.n && k < 32; ++k) vd.width[k] = (unsigned char)(plan->h[k] > 255 ? 255 : plan->h[k]);
        return vd;
    }
    bool planned_ = false;  // the widths come from a prediction (not measured by this transition)

    int run(const irs_slab_op* ops, int n_ops) {
        for (int i = 0; i < n_ops; ++i) {
            const irs_slab_op& o = ops[i];
            if (o.kind == IRS_OP_LAUNCH) {
                if (launch(o)) return 1;
            } else if (o.kind == IRS_OP_EXCHANGE) {
                cur_bwd_k = o.k;  // the adjoint step that consumes the ghost planes names the layout of a gradient buffer
                if (exchange(o)) return 1;
            } else if (o.kind == IRS_OP_ALLREDUCE) {
                if (allreduce(o)) return 1;
            } else if (o.kind == IRS_OP_WAIT) {
                if (o.id < 0 || o.id >= 128) return fail("slab: bad wait id %d", o.id);
                if (pending[o.id]) {
                    const bool first = tl && c->tl_rec[o.id].eW0 < 0 && c->tl_rec[o.id].eP >= 0;
                    if (first) c->tl_rec[o.id].eW0 = stamp(st);
                    HIP_TRY(hipStreamWaitEvent(st, pending[o.id], 0));
                    if (first) c->tl_rec[o.id].eW1 = stamp(st);
                }
            } else return fail("slab: unknown op kind %d", o.kind);
        }
        LAUNCH_CHECK();
        return 0;
    }
    int exchange(const irs_slab_op& o) {
        if (o.id < 0 || o.id >= 128) return fail("slab: bad exchange id %d", o.id);
        const Views v = views(c);
        int kind;
        float* buf = buffer(o.stage, v, &kind);
        if (!buf) return fail("slab: exchange of unknown buffer %d", o.stage);
        hipEvent_t prod = take(), recv = take();
        HIP_TRY(hipEventRecord(prod, st));
        tl_open(o.id, IRS_OP_EXCHANGE, o.stage, o.k, o.width, stamp(st));
        HIP_TRY(hipStreamWaitEvent(cs, prod, 0));
        if (exchange_planes(c, buf, kind, chains, o.width, cs)) return 1;
        HIP_TRY(hipEventRecord(recv, cs));
        if (tl) c->tl_rec[o.id].eR = stamp(cs);
        pending[o.id] = recv;
        return 0;
    }
    int allreduce(const irs_slab_op& o) {
        if (o.id < 0 || o.id >= 128) return fail("slab: bad all-reduce id %d", o.id);
        const int n = c->cfg.no_steps;
        void* buf;
        size_t count;
        int mx = 0;
        switch (o.stage) {
            case IRS_AR_ENERGY: buf = c->energy_sum; count = (size_t)chains; break;
            case IRS_AR_STATS: buf = c->stat_sum; count = kStatVals; break;
            case IRS_AR_NLL:  // data-term sums + (SVF) the regulariser energies, adjacent in the workspace: one payload
                if (c->ffd) { buf = c->nll_sum; count = (size_t)chains; }  // (SVFFD: the energy is computed whole on every rank)
                else { buf = c->energy_sum; count = (size_t)2 * IRS_MAX_CHAINS; }
                break;
            case IRS_AR_DMAX: buf = c->dmax; count = (size_t)4 * c->C * (n + 1); mx = 1; break;
            case IRS_AR_MOMENTS: buf = c->stat_sum; count = 3; break;
            case IRS_AR_CPGRAD: buf = c->tmpB; count = (size_t)chains * 3 * c->volv.V; mx = 2; break;
            default: return fail("slab: unknown all-reduce %d", o.stage);
        }
        // a dedicated event pair per KIND of reduction: these results are waited for much later than the exchanges in between
        const int slot = o.stage == IRS_AR_MOMENTS ? 5 : o.stage;  // kinds 0 .. 4 and 7
        hipEvent_t prod = c->sev[16 + 2 * slot], fin = c->sev[17 + 2 * slot];
        HIP_TRY(hipEventRecord(prod, st));
        tl_open(o.id, IRS_OP_ALLREDUCE, o.stage, -1, 0, stamp(st));
        HIP_TRY(hipStreamWaitEvent(cs, prod, 0));
        if (comm_allreduce(c->comm, buf, count, mx, cs)) return 1;
        HIP_TRY(hipEventRecord(fin, cs));
        if (tl) c->tl_rec[o.id].eR = stamp(cs);
        pending[o.id] = fin;
        return 0;
    }
    int launch(const irs_slab_op& o) {
        const irs_config& cfg = c->cfg;
        const Views v = views(c);
        const int n = cfg.no_steps, C = chains;
        const Vol w = o.hi1 > o.lo1 ? window2(c->vol, o.lo0, o.hi0, o.lo1, o.hi1) : window(c->vol, o.lo0, o.hi0);
        const Lin lin = c->lin.lin();
        const uint64_t* it = &c->state->st.iteration;
        // SVFFD: the velocity-grid arrays (v, sigma, eps, noisy, v_s, grad_v) are whole control-grid arrays, not slab-local
        float* noisy = c->ffd ? c->tmpA : planar(c->tmpA, v);
        const Vol wv = c->ffd ? c->volv : w;  // window of the velocity-grid stages
        const float* d0 = c->ffd ? planar(c->dense, v) : vs;  // what squaring step 0 reads
        const int64_t HW = (int64_t)c->vol.H * c->vol.W;
        switch (o.stage) {
            case IRS_SG_PERTURB:
                launch_perturb(v_src, io.sigma, io.eps, (float)sqrt(2.0 * (double)cfg.lr), cfg.sobolev_s > 0 ? noisy : vs, C, wv, cfg.seed, 0, it, st);
                break;
            case IRS_SG_COPY_V: {
                float* first = cfg.sobolev_s > 0 ? noisy : vs;
                if (c->ffd) {
                    HIP_TRY(hipMemcpyAsync(first, v_src, (size_t)C * 3 * c->volv.V * sizeof(float), hipMemcpyDeviceToDevice, st));
                    break;
                }
                for (int ch = 0; ch < 3 * C; ++ch)
                    HIP_TRY(hipMemcpyAsync(first + (int64_t)ch * c->vol.V + (int64_t)w.z0 * HW, v_src + (int64_t)ch * c->vol.V + (int64_t)w.z0 * HW,
                                           (size_t)w.nz * HW * sizeof(float), hipMemcpyDeviceToDevice, st));
                break;
            }
            case IRS_SG_SMOOTH:
                if (o.k == 1)  // perturbation fused into the smoothing kernel (SVF_3D, s > 0)
                    launch_perturb_sobolev_march(v_src, io.sigma, io.eps, (float)sqrt(2.0 * (double)cfg.lr), vs, c->sob, C, w, c->dmax, n, cfg.seed, 0, it, st);
                else if (cfg.sobolev_s > 0) launch_sobolev_march(noisy, vs, c->sob, C * 3, wv, c->ffd ? nullptr : c->dmax, n, st);
                else if (!c->ffd) launch_field_absmax(vs, true, n, c->dmax, C, w, st);
                break;
            case IRS_SG_ENERGY:
                launch_reg_energy(vs, c->energy_partials, C, wv, st);
                launch_reduce_partials(c->energy_partials, energy_blocks(wv), C, c->energy_sum, st);
                break;
            case IRS_SG_FFD_UP: {
                const int G[3] = {c->volv.D, c->volv.H, c->volv.W};
                ffd_up(vs, c->dense, c->tmpA, C, c->vol, G, c->spl, st, w.z0, w.nz, c->sl.lo, c->sl.hi - c->sl.lo);
                launch_field_absmax(d0, true, n, c->dmax, C, w, st);  // bound of d_0
                break;
            }
            case IRS_SG_FFD_ADJ: {
                float sc3[3];
                prescale_factors(c->vol, n, sc3);
                float* g0 = grad_raw(c, 0, false);
                float* scaled = g0 == c->gA ? c->gB : c->gA;  // (any field other than g0)
                launch_scale_channels(planar(g0, v), planar(scaled, v), sc3[0], sc3[1], sc3[2], C, w, st);
                const int G[3] = {c->volv.D, c->volv.H, c->volv.W};
                ffd_adjoint(scaled, c->tmpB, c->tmpA, C, c->vol, G, c->spl, st, w.z0, w.nz, c->sl.lo, c->sl.hi - c->sl.lo);
                break;
            }
            case IRS_SG_EXP_FWD: {
                const int k = o.k;
                const float* in = k == 0 ? d0 : step_buf(c, v, k - 1);
                launch_exp_step_fwd_march(in, step_buf(c, v, k), k == 0, n, C, w, lin, c->dmax + (int64_t)k * c->C * 4,
                                          c->dmax + (int64_t)(k + 1) * c->C * 4, predicted_small(c, k), fwd_lay(c, k), st);
                break;
            }
            case IRS_SG_OUTPUTS:
                launch_svf_outputs(step_buf(c, v, n - 1), io.transformation, io.displacement, C, w, lin, st);
                break;
            case IRS_SG_WARP:
                launch_warp_fwd(io.moving_im, io.moving_chains == 1 ? 0 : c->vol.Vg, step_buf(c, v, n - 1), io.unif,
                                jitter && cfg.uniform_alpha > 0.0f ? cfg.uniform_alpha : 0.0f, warped, nullptr, 0, C, w, lin, cfg.seed, 0, it, st);
                break;
            case IRS_SG_RESIDUAL:
                if (cfg.data_loss == IRS_DATA_GMM_LCC)
                    launch_lcc_fwd_march(planar(c->fhat, v), c->fhat_chains == 1 ? 0 : c->vol.V, warped, z, planar(c->sigM, v), cfg.lcc_s, C, w, st);
                else launch_residual_ssd(io.fixed_im, io.fixed_chains == 1 ? 0 : c->vol.V, warped, z, C, w, st);
                break;
            case IRS_SG_STATS: {
                const uint8_t* mask = io.mask + (io.mask_chains == 1 ? 0 : (int64_t)o.k * c->vol.V);
                launch_stats(stats_vd, z + (int64_t)o.k * c->vol.V, mask, c->state, c->stat_partials, w, st, c->dcfg.K);
                launch_reduce_partials(c->stat_partials, stats_blocks(w), kStatVals, c->stat_sum, st);  // stored [kStatVals][blocks]
                break;
            }
            case IRS_SG_CHAIN_SCALAR:
                launch_chain_scalar(c->state, c->stat_sum, 1, o.k, stats_op | (o.k == 0 && in_transition ? 4 : 0), c->dcfg, st, in_transition ? verdict_always() : no_verdict());
                break;
            case IRS_SG_VERDICT:  // no statistics stage in this configuration: the verdict alone
                launch_chain_scalar(c->state, c->stat_sum, 1, 0, 4, c->dcfg, st, verdict_always());
                break;
            case IRS_SG_DATA_BWD: {
                const int ch = o.k;
                const uint8_t* mask = io.mask + (io.mask_chains == 1 ? 0 : (int64_t)ch * c->vol.V);
                const float* f = cfg.data_loss == IRS_DATA_GMM_LCC ? planar(c->fhat, v) + (c->fhat_chains == 1 ? 0 : (int64_t)ch * c->vol.V) : nullptr;
                double* part = c->nll_partials + (int64_t)ch * c->nll_blocks;
                launch_data_bwd(cfg.data_loss, f, 0, z + (int64_t)ch * c->vol.V, planar(c->sigM, v) + (int64_t)ch * c->vol.V, mask, 0, nullptr, c->state, ch,
                                planar(c->gM, v) + (int64_t)ch * c->vol.V, part, cfg.lcc_s, 1, w, st);
                launch_reduce_partials(part, data_bwd_blocks(cfg.data_loss, w), 1, c->nll_sum + ch, st);
                break;
            }
            case IRS_SG_WARP_BWD:
                launch_warp_bwd(io.moving_im, io.moving_chains == 1 ? 0 : c->vol.Vg, step_buf(c, v, n - 1), io.unif,
                                cfg.uniform_alpha > 0.0f ? cfg.uniform_alpha : 0.0f, planar(c->gM, v), planar(c->gA, v), C, w, lin, cfg.seed, 0, it, st);
                break;
            case IRS_SG_EXP_BWD: {
                // The plan is validated against the measured bounds afterwards (finalize stage: floor(max|d_k|) + 1 <= planned
                // width, else the transition is flagged invalid), so the variants that can only be selected above it need not be
                // launched at all: a planned width of 1 leaves the radius-1 gather alone.
                const int k = o.k, hplan = plan ? plan->h[k] : 0;
                const int lay = bwd_lay(c, k);
                float* gi = grad_raw(c, k, true);
                float* go = grad_raw(c, k, false);
                const float* G = (lay & 2) ? aos(gi, v) : planar(gi, v);
                float* out = (lay & 4) ? aos(go, v) : planar(go, v);
                const float* dk = k == 0 ? d0 : step_buf(c, v, k - 1);
                const unsigned* dm = c->dmax + (int64_t)k * c->C * 4;
                // (only from the PLAN, which the verdict checks -- a host guess here could hand a step beyond two voxels to the radius-2
                // gather's generic fallback, whose sums are not the any-radius kernel's bit for bit)
                const bool skip_any = hplan >= 1 && hplan <= 2;
                launch_exp_step_bwd_march(G, dk, out, k == 0, n, c->C, w, lin, dm, hplan == 1 ? 1 : 2, skip_any, nullptr, lay, nullptr, st);
                // the any-radius kernel bounds its sources by the global bound around the tile (no coarse grid: that one spans the volume)
                if (!skip_any) launch_exp_step_bwd_lds(G, dk, out, k == 0, n, c->C, w, lin, dm, 2, 2, nullptr, lay, nullptr, st);
                break;
            }
            case IRS_SG_REG_SCALAR:
                launch_reg_scalar(c->state, c->energy_sum, 1, c->dcfg, st, verdict_always());
                break;
            case IRS_SG_UPDATE: {
                float sc3[3];
                prescale_factors(c->vol, n, sc3);
                if (c->ffd) launch_sgld_update(io.v, io.sigma, c->tmpB, vs, c->state, cfg.lr, 1.0f, 1.0f, 1.0f, io.grad_v, C, wv, st);
                else launch_sgld_update(io.v, io.sigma, planar(grad_raw(c, 0, false), v), vs, c->state, cfg.lr, sc3[0], sc3[1], sc3[2], io.grad_v, C, w, st);
                break;
            }
            case IRS_SG_FINALIZE:
                // bounds + the cumulative count of failed (no-op) transitions into the plan-hint slot of this transition
                launch_finalize(c->state, c->nll_sum, 1, c->dcfg, true, c->dmax, c->plan_hint + (c->n_enqueued % 2) * kHintWords,
                                4 * c->C * (n + 1), verdict_always(), kHintWords - 8, true, st);
                c->dmax_clean = true;
                break;
            default:
                return fail("slab: unknown stage %d", o.stage);
        }
        return 0;
    }
    int stats_vd = 0, stats_op = 3;
    bool in_transition = false;  // (the mixture initialisation runs the same stages without a verdict)
    // the stages that read DevState::bad_now need it written by THIS transition: a transition without a plan passes an empty
    // assumption set with valid bounds, whose verdict is "fine"
    Verdict verdict_always() const {
        Verdict vd = verdict();
        if (!vd.bounds) {
            vd.bounds = c->dmax;
            vd.n = 0;
            vd.C = c->C;
        }
        return vd;
    }
};

// validation of the planned ghost widths against the (all-reduced) bounds of this transition; sticky flag in pinned memory
int ghost_width_from_bound(float m, bool safety) {
    if (!(m >= 0.0f) || m > 1.0e6f) return -1;
    return (int)floorf(safety ? 1.25f * m + 0.25f : m) + 1;
}

// global bound (all chains, all axes) of d_k in a slot of the plan hints
float hint_bound(const irs_ctx* c, const unsigned* slot, int k) {
    const volatile unsigned* h = slot + (size_t)k * c->C * 4;
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

// the measuring forward pass: the bound of d_k is all-reduced and read back before step k (one host synchronisation per step)
int forward_exact(Exec& ex, Sched& sch, Plan& plan) {
    irs_ctx* c = ex.c;
    const int n = c->cfg.no_steps;
    plan.n = n;
    for (int k = 0; k < n; ++k) {
        size_t from = sch.ops.size();
        sch.wait(sch.allreduce(IRS_AR_DMAX));  // (every bound published so far; only row k is read)
        if (ex.run(sch.ops.data() + from, (int)(sch.ops.size() - from))) return 1;
        unsigned host[4 * IRS_MAX_CHAINS];
        HIP_TRY(hipMemcpyAsync(host, c->dmax + (int64_t)k * c->C * 4, sizeof(unsigned) * 4 * c->C, hipMemcpyDeviceToHost, ex.st));
        HIP_TRY(hipStreamSynchronize(ex.st));
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
        // the backward pass lives off E_k = h_k: one step per round as well
        plan.br[k] = n - 1 - k;
        plan.bw[n - 1 - k] = h;
        from = sch.ops.size();
        sch.forward_exact_step(k, h);
        if (ex.run(sch.ops.data() + from, (int)(sch.ops.size() - from))) return 1;
    }
    plan.nf = plan.nb = n;
    return 0;
}

// irs_io with every slab-local pointer shifted to its virtual base
irs_io shifted_io(const irs_ctx* c, const irs_io* io) {
    const Views v = views(c);
    irs_io o = *io;
    o.fixed_im = planar(io->fixed_im, v);
    o.mask = planar(io->mask, v);
    if (!c->ffd) {  // (SVFFD: these live on the control grid, whole on every rank)
        o.v = planar(io->v, v);
        o.sigma = planar(io->sigma, v);
        o.eps = planar(io->eps, v);
        o.curr_state = planar(io->curr_state, v);
        o.grad_v = planar(io->grad_v, v);
    }
    o.unif = planar(io->unif, v);
    o.im_moving_warped = planar(io->im_moving_warped, v);
    o.residuals = planar(io->residuals, v);
    o.displacement = planar(io->displacement, v);
    o.transformation = planar(io->transformation, v);
    return o;  // moving_im stays: it is the whole volume
}

}  // namespace

namespace irs {
void slab_release(irs_ctx* c) {
    if (!c) return;
    for (int i = 0; i < 28; ++i)
        if (c->sev[i]) (void)hipEventDestroy(c->sev[i]);
    if (c->cs) (void)hipStreamDestroy(c->cs);
    if (c->plan_hint) (void)hipHostFree(c->plan_hint);
    for (int i = 0; i < c->tl_ev_n; ++i) (void)hipEventDestroy(c->tl_ev[i]);
    free(c->tl_ev);
    c->tl_ev = nullptr;
    c->tl_ev_n = 0;
}
}  // namespace irs

extern "C" {

int irs_slab_plan_rounds(const int32_t* h, int n, int ghost_max, int min_slab, int n_buffers, int32_t* fwd_round, int32_t* fwd_width,
                         int32_t* n_fwd, int32_t* bwd_round, int32_t* bwd_width, int32_t* n_bwd) {
    if (!h || n < 1 || n > kMaxSteps || !fwd_round || !fwd_width || !n_fwd || !bwd_round || !bwd_width || !n_bwd)
        return fail("irs_slab_plan_rounds: bad arguments");
    Plan p;
    p.n = n;
    for (int k = 0; k < n; ++k) p.h[k] = h[k];
    if (plan_rounds(p, ghost_max > 0 ? ghost_max : kDefaultGhostMax, min_slab, n_buffers >= 3 ? 3 : 2)) return 1;
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
    if (comm && world > 1) {
        // the peer-mapped transport sizes its landing area here (collective): the widest exchange is `margin` planes of a
        // three-channel field per chain and side; the largest all-reduce is the control-grid gradient (SVFFD) or the bounds
        const size_t xbytes = (size_t)3 * c->C * s.margin * c->vol.H * c->vol.W * sizeof(float);
        size_t arbytes = sizeof(double) * (kStatVals > 2 * IRS_MAX_CHAINS ? kStatVals : 2 * IRS_MAX_CHAINS);
        const size_t bounds = sizeof(unsigned) * 4 * c->C * (cfg->no_steps + 1);
        arbytes = bounds > arbytes ? bounds : arbytes;
        if (c->ffd && sizeof(float) * c->C * 3 * c->volv.V > arbytes) arbytes = sizeof(float) * c->C * 3 * c->volv.V;
        if (comm_reserve(comm, xbytes, arbytes)) {
            irs_destroy(c);
            return 1;
        }
        // a timed-out wait of the peer-mapped transport freezes the transition in flight (scalar_kernels.h: comm_bad)
        const unsigned* ef = comm_error_flag(comm);
        if (ef && hipMemcpy((char*)c->state + offsetof(DevState, comm_err), &ef, sizeof(ef), hipMemcpyHostToDevice) != hipSuccess) {
            irs_destroy(c);
            return fail("irs_slab_create: publishing the transport's error word failed");
        }
    }
    // the communication stream gets the HIGHEST priority: its send / recv kernels are enqueued while interior launches of a
    // thousand workgroups occupy every CU, and the overlap the schedule is built on needs them to be dispatched ahead of those
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    hipError_t e = hipStreamCreateWithPriority(&c->cs, hipStreamNonBlocking, prio_greatest);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->plan_hint, sizeof(unsigned) * 2 * kHintWords, hipHostMallocDefault);
    if (e == hipSuccess)
        for (int i = 0; i < 2 * kHintWords; ++i) c->plan_hint[i] = (i % kHintWords) < kHintWords - 8 ? 0x7f800000u : 0u;  // +inf: nothing known; flags clear
    for (int i = 0; i < 28 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->sev[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        irs_destroy(c);
        return fail("irs_slab_create: stream / event creation failed: %s", hipGetErrorString(e));
    }
    c->have_pred = false;
    *out = c;
    return 0;
}

int irs_slab_timeline_arm(irs_ctx* c, int transitions) {
    if (!c || !c->sl.on) return fail("irs_slab_timeline_arm: not a slab context");
    if (transitions > 0 && !c->tl_ev) {  // four events per hand-over (at most 128 of them) + the two ends of the transition
        const int n = 4 * 128 + 2;
        c->tl_ev = (hipEvent_t*)calloc((size_t)n, sizeof(hipEvent_t));
        if (!c->tl_ev) return fail("irs_slab_timeline_arm: out of host memory");
        for (c->tl_ev_n = 0; c->tl_ev_n < n; ++c->tl_ev_n)
            if (hipEventCreate(&c->tl_ev[c->tl_ev_n]) != hipSuccess) return fail("irs_slab_timeline_arm: hipEventCreate failed");
    }
    c->tl_arm = transitions > 0 ? transitions : 0;
    return 0;
}

int irs_slab_timeline_get(irs_ctx* c, irs_slab_timeline_entry* out, int max_entries, int32_t* n_entries, float* total_us, void* stream) {
    if (!c || !c->sl.on || !n_entries) return fail("irs_slab_timeline_get: not a slab context / null argument");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize(c->cs));
    *n_entries = 0;
    if (total_us) *total_us = 0.0f;
    if (!c->tl_have) return fail("irs_slab_timeline_get: no sampled transition (irs_slab_timeline_arm, then irs_slab_transition)");
    auto us = [&](int a, int b, float* o) {  // elapsed a -> b in microseconds; false if either event is missing
        if (a < 0 || b < 0) return false;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->tl_ev[a], c->tl_ev[b]) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        *o = 1000.0f * ms;
        return true;
    };
    float t = 0.0f;
    if (total_us && us(c->tl_t0, c->tl_t1, &t)) *total_us = t;
    int n = 0;
    for (int id = 0; id < c->tl_n; ++id) {
        const irs_ctx::TlRec& r = c->tl_rec[id];
        if (r.kind < 0) continue;
        if (out && n < max_entries) {
            irs_slab_timeline_entry e;
            e.kind = r.kind;
            e.stage = r.stage;
            e.k = r.k;
            e.width = r.width;
            e.ready_us = e.handover_us = e.stall_us = 0.0f;
            e.wait_at_us = -1.0f;
            (void)us(c->tl_t0, r.eP, &e.ready_us);
            (void)us(r.eP, r.eR, &e.handover_us);
            if (us(c->tl_t0, r.eW0, &e.wait_at_us)) (void)us(r.eW0, r.eW1, &e.stall_us);
            out[n] = e;
        }
        ++n;
    }
    *n_entries = n;
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
    // (all enqueued transitions have finished: both slots are final; cumulative counts)
    const unsigned f0 = c->plan_hint[kHintWords - 8], f1 = c->plan_hint[kHintWords + kHintWords - 8], fmax = f0 > f1 ? f0 : f1;
    out->mispredictions = c->slab_mispredictions + (fmax > c->fails_seen ? fmax - c->fails_seen : 0);
    out->last_fwd_rounds = c->last_nf;
    out->last_bwd_rounds = c->last_nb;
    return 0;
}

// ghost widths of this transition: from bounds the host has already seen (safety factor), or unknown -> measure
static bool plan_widths(irs_ctx* c, Plan& plan) {
    const int n = c->cfg.no_steps;
    plan.n = n;
    if (!c->have_pred || c->kn.slab_exact != 0) return false;
    // Every rank must arrive at the SAME widths (they size the messages both sides of a link post): the source is the
    // all-reduced bounds of transition t - 2, which the caller has just waited for (slot t % 2 of the plan hints; the
    // finalize kernel of t - 1 writes the other slot).  Transition 1 plans from the widths transition 0 measured.
    const bool fresh = c->n_enqueued >= 2;
    const unsigned* slot = c->plan_hint + (c->n_enqueued % 2) * kHintWords;
    for (int k = 0; k < n; ++k) {
        const int hh = fresh ? ghost_width_from_bound(hint_bound(c, slot, k), true) : c->pred[k];
        if (hh < 1) return false;
        plan.h[k] = hh;
    }
    // the launch heuristics (which variants to launch) read the same snapshot
    if (fresh) memcpy(c->hint, slot, sizeof(unsigned) * 4 * c->C * (n + 1));
    plan.h[0] = 1;  // |d_0| = |v_s| / 2^n voxels (validated like the others)
    const int forced = c->kn.slab_force_h;  // test hook: a deliberately wrong plan (the validation must catch it)
    if (forced > 0)
        for (int k = 0; k < n; ++k) plan.h[k] = forced;
    return true;
}

}  // extern "C"

namespace {

// one transition of the slab, enqueued (planned from the bounds of t - 2, or measuring)
int slab_transition_once(irs_ctx* c, const irs_io* io_in, hipStream_t st) {
    const irs_config& cfg = c->cfg;
    const SlabInfo& s = c->sl;
    const int C = c->C, n = cfg.no_steps;
    const Views v = views(c);
    const irs_io io = shifted_io(c, io_in);
    Plan plan;
    const bool planned = c->n_enqueued >= c->exact_until && plan_widths(c, plan);
    if (planned && plan_rounds(plan, s.gmax, s.world > 1 ? s.min_slab : 1 << 30, grad_buffers(c))) return 1;

    Exec ex{c, st, c->cs, io, io.v, io.curr_state ? io.curr_state : (c->ffd ? c->vs : planar(c->vs, v)), io.im_moving_warped ? io.im_moving_warped : planar(c->warped, v),
            io.residuals ? io.residuals : planar(c->z, v), planned ? &plan : nullptr, cfg.uniform_alpha > 0.0f, C};
    memset(ex.pending, 0, sizeof(ex.pending));
    ex.stats_vd = cfg.virtual_decimation;
    ex.stats_op = 3;
    ex.in_transition = true;
    ex.planned_ = planned;
    if (c->tl_arm > 0 && c->tl_ev) {  // sampled transition: timing events around every hand-over (irs_slab_timeline_get)
        --c->tl_arm;
        ex.tl = true;
        c->tl_ev_used = 0;
        c->tl_n = 0;
        for (int i = 0; i < 128; ++i) c->tl_rec[i] = irs_ctx::TlRec{-1, -1, -1, 0, -1, -1, -1, -1};
        c->tl_t0 = ex.stamp(st);
        c->tl_have = false;
    }
    Sched sch(s, cfg, C);
    sch.nbuf_ = grad_buffers(c);
    sch.want_split_ = c->kn.slab_split != 0;
    sch.fuse_noise_ = c->kn.fuse_noise != 0 && !(io.sigma && io.eps);  // (sigma field AND injected noise, tests only: the two-kernel form, as in the fused engine)
    if (!c->dmax_clean) HIP_TRY(hipMemsetAsync(c->dmax, 0, sizeof(unsigned) * 4 * c->C * (n + 1), st));
    c->dmax_clean = false;
    int energy_ar = -1;
    if (planned) {
        // the first forward round lives off ghost planes of v_s that the smoothing stage computes itself
        sch.head(true, true, plan.fw[0] > 1 ? plan.fw[0] : 1, &energy_ar);
        sch.forward_planned(plan);
    } else {
        sch.head(true, true, 1, &energy_ar);
        if (ex.run(sch.ops.data(), (int)sch.ops.size())) return 1;
        if (forward_exact(ex, sch, plan)) return 1;
        for (int k = 0; k < n; ++k) c->pred[k] = ghost_width_from_bound(plan.m[k], true);
        c->have_pred = true;
        c->slab_exact += 1;
        ex.plan = &plan;
    }
    const size_t done = planned ? 0 : sch.ops.size();
    sch.middle(io.transformation || io.displacement);
    // (SSD without virtual decimation has alpha = 1 and no mixture: the statistics stage and its all-reduce drop out; the
    // measuring transition runs it once, for n_mask)
    sch.backward_and_update(plan, cfg.data_loss == IRS_DATA_GMM_LCC || cfg.virtual_decimation || !planned, energy_ar);
    if (sch.next_id > 128) return fail("irs_slab_transition: schedule with %d exchanges", sch.next_id);
    if (ex.run(sch.ops.data() + done, (int)(sch.ops.size() - done))) return 1;
    c->last_nf = plan.nf;
    c->last_nb = plan.nb;
    if (getenv("IRS_SLAB_DEBUG") && s.rank == 0) {  // (debugging aid: the ghost widths this transition ran with)
        fprintf(stderr, "[slab] transition %llu %s: h =", (unsigned long long)c->n_enqueued, planned ? "planned" : "exact");
        for (int k = 0; k < n; ++k) fprintf(stderr, " %d", plan.h[k]);
        fprintf(stderr, "\n");
    }
    if (ex.tl) {
        c->tl_t1 = ex.stamp(st);
        c->tl_have = c->tl_t0 >= 0 && c->tl_t1 >= 0;
    }
    HIP_TRY(hipEventRecord(c->ra_ev[c->n_enqueued % 4], st));
    ++c->n_enqueued;
    return 0;
}

// failed (no-op) transitions reported in a plan-hint slot (cumulative count; every rank reads the same number at the same call)
void slab_poll(irs_ctx* c, int slot) {
    const unsigned f = ((volatile unsigned*)c->plan_hint)[slot * kHintWords + (kHintWords - 8)];
    if (f == c->fails_seen || (int)(f - c->fails_seen) < 0) return;
    const uint64_t d = f - c->fails_seen;
    c->makeup += d;
    c->fails_total += d;
    c->slab_mispredictions += d;
    c->fails_seen = f;
    // the bounds that misled the plan are still the newest every rank has seen: measure (exact mode) until the plan source --
    // the bounds of transition t - 2 -- is one that was measured after the jump
    c->exact_until = c->n_enqueued + c->makeup + 2;
}

}  // namespace

namespace irs {
void slab_drop_pending(irs_ctx* c) {  // irs_set_state: the chain is being replaced, dropped transitions of the old one are not re-run
    if (c->cs) (void)hipStreamSynchronize(c->cs);
    slab_poll(c, 0);
    slab_poll(c, 1);
    c->makeup = 0;
}
int slab_flush(irs_ctx* c, hipStream_t st) {
    for (int guard = 0; guard < 8; ++guard) {
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipStreamSynchronize(c->cs));
        if (comm_check(c->comm)) return 1;
        slab_poll(c, 0);  // everything enqueued has finished: both slots are final
        slab_poll(c, 1);
        if (!c->makeup) return 0;
        if (!c->kn.recover || !c->have_last_io) return fail("irs_flush: %llu slab transition(s) were dropped after a failed ghost-width plan", (unsigned long long)c->makeup);
        while (c->makeup > 0) {
            --c->makeup;
            if (slab_transition_once(c, &c->last_io, st)) return 1;
        }
    }
    return fail("irs_flush: slab transitions keep failing");
}
}  // namespace irs

extern "C" {

int irs_slab_transition(irs_ctx* c, const irs_io* io_in, void* stream) {
    if (check_io(c, io_in, "irs_slab_transition")) return 1;
    if (!c->sl.on) return fail("irs_slab_transition: not a slab context (irs_slab_create)");
    if (!io_in->v) return fail("irs_slab_transition: v is required");
    hipStream_t st = (hipStream_t)stream;
    // the host runs at most two transitions ahead: the width plan reads the bounds of transition t - 2, which must have
    // FINISHED on this rank (not an option here, unlike run_ahead of the fused path)
    if (c->n_enqueued >= 2) {
        HIP_TRY(hipEventSynchronize(c->ra_ev[(c->n_enqueued - 2) % 4]));
        // ... and so has its verdict.  A transition whose ghost zones turned out narrower than its displacement needed was a
        // no-op on the device (scalar_kernels.h: Verdict) -- on EVERY rank, the verdict comes from the all-reduced bounds -- and
        // every rank reads the same count at the same call: all of them re-run it here, measuring instead of predicting.
        slab_poll(c, (int)(c->n_enqueued % 2));
    }
    if (comm_check(c->comm)) return 1;
    if (c->makeup && !c->kn.recover)
        return fail("irs_slab_transition: a transition ran with ghost zones narrower than its displacement needed and was dropped (recover = 0)");
    while (c->makeup > 0) {
        --c->makeup;
        if (slab_transition_once(c, io_in, st)) return 1;
    }
    c->last_io = *io_in;
    c->have_last_io = true;
    return slab_transition_once(c, io_in, st);
}

int irs_slab_gmm_init(irs_ctx* c, const irs_io* io_in, const float* v_sample, int warm_up, void* stream) {
    if (check_io(c, io_in, "irs_slab_gmm_init")) return 1;
    if (!c->sl.on) return fail("irs_slab_gmm_init: not a slab context (irs_slab_create)");
    if (c->cfg.data_loss != IRS_DATA_GMM_LCC) return 0;
    hipStream_t st = (hipStream_t)stream;
    const Views v = views(c);
    const SlabInfo& s = c->sl;
    const int n = c->cfg.no_steps;
    irs_io io = shifted_io(c, io_in);
    io.sigma = nullptr;
    io.eps = nullptr;
    io.unif = nullptr;
    io.transformation = nullptr;
    io.displacement = nullptr;
    // trainer.py:529-547: one velocity sample (no Langevin noise, no jitter), batch of one; staged in gB (free until the backward pass)
    const int64_t HW = (int64_t)c->vol.H * c->vol.W;
    if (c->ffd) {  // the sample is a whole control-grid field: staged in tmpB (velocity-grid sized, untouched by the forward pass)
        const size_t bytes = (size_t)3 * c->volv.V * sizeof(float);
        if (v_sample) HIP_TRY(hipMemcpyAsync(c->tmpB, v_sample, bytes, hipMemcpyDeviceToDevice, st));
        else HIP_TRY(hipMemsetAsync(c->tmpB, 0, bytes, st));
    } else
        for (int ch = 0; ch < 3; ++ch) {
            float* dst = c->gB + (int64_t)ch * c->vol.V + (int64_t)(s.a - s.lo) * HW;
            const size_t bytes = (size_t)(s.b - s.a) * HW * sizeof(float);
            if (v_sample) HIP_TRY(hipMemcpyAsync(dst, v_sample + (int64_t)ch * c->vol.V + (int64_t)(s.a - s.lo) * HW, bytes, hipMemcpyDeviceToDevice, st));
            else HIP_TRY(hipMemsetAsync(dst, 0, bytes, st));
        }
    Plan plan;
    Exec ex{c, st, c->cs, io, c->ffd ? c->tmpB : planar(c->gB, v), c->ffd ? c->vs : planar(c->vs, v), planar(c->warped, v), planar(c->z, v), nullptr, false, 1};
    memset(ex.pending, 0, sizeof(ex.pending));
    Sched sch(s, c->cfg, 1);
    HIP_TRY(hipMemsetAsync(c->dmax, 0, sizeof(unsigned) * 4 * c->C * (n + 1), st));
    c->dmax_clean = false;
    int unused = -1;
    sch.head(false, false, 1, &unused);
    if (ex.run(sch.ops.data(), (int)sch.ops.size())) return 1;
    if (forward_exact(ex, sch, plan)) return 1;
    size_t from = sch.ops.size();
    sch.middle(false);
    if (ex.run(sch.ops.data() + from, (int)(sch.ops.size() - from))) return 1;
    // std of the masked residual -> initial mixture; VD factor; warm-up steps with that factor (each a blocking all-reduce:
    // a one-off stage)
    const Vol w0 = window(c->vol, s.a, s.b);
    launch_masked_moments(ex.z, io.mask, c->stat_partials, w0, st);
    launch_reduce_cols(c->stat_partials, stats_blocks(w0), 3, c->stat_sum, st);
    from = sch.ops.size();
    sch.wait(sch.allreduce(IRS_AR_MOMENTS));
    if (ex.run(sch.ops.data() + from, (int)(sch.ops.size() - from))) return 1;
    launch_gmm_init_from_moments(c->state, c->stat_sum, 1, c->dcfg, st);
    LAUNCH_CHECK();
    for (int i = 0; i <= warm_up; ++i) {
        ex.stats_vd = i == 0 ? c->cfg.virtual_decimation : 0;
        ex.stats_op = i == 0 ? 1 : 2;  // first: alpha (kept for the warm-up); then: one mixture step each
        Sched one(s, c->cfg, 1);
        one.launch(IRS_SG_STATS, 0, w0, IRS_SB_Z, -1, ex.stats_vd ? 1 : 0, -1);
        one.wait(one.allreduce(IRS_AR_STATS));
        one.launch(IRS_SG_CHAIN_SCALAR, 0, w0, -1, -1, 0, -1);
        memset(ex.pending, 0, sizeof(ex.pending));
        if (ex.run(one.ops.data(), (int)one.ops.size())) return 1;
    }
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int irs_slab_trace(const irs_config* cfg, const irs_slab_config* scfg, int rank, int world, const int32_t* h, irs_slab_op* ops,
                   int max_ops, int32_t* n_ops) {
    if (!cfg || !h || !n_ops || (max_ops > 0 && !ops)) return fail("irs_slab_trace: bad arguments");
    if (cfg->no_steps < 1 || cfg->no_steps > kMaxSteps) return fail("irs_slab_trace: no_steps out of range");
    SlabInfo s;
    if (plan_layout(cfg, scfg, rank, world, &s)) return 1;
    Plan plan;
    plan.n = cfg->no_steps;
    for (int k = 0; k < plan.n; ++k) plan.h[k] = h[k];
    if (plan_rounds(plan, s.gmax, world > 1 ? s.min_slab : 1 << 30, world > 1 ? 3 : 2)) return 1;
    Sched sch(s, *cfg, cfg->no_chains);
    sch.nbuf_ = world > 1 ? 3 : 2;
    sch.fuse_noise_ = global_knobs().fuse_noise != 0;
    int energy_ar = -1;
    sch.head(true, true, plan.fw[0] > 1 ? plan.fw[0] : 1, &energy_ar);
    sch.forward_planned(plan);
    sch.middle(true);
    sch.backward_and_update(plan, cfg->data_loss == IRS_DATA_GMM_LCC || cfg->virtual_decimation != 0, energy_ar);
    *n_ops = (int32_t)sch.ops.size();
    if ((int)sch.ops.size() > max_ops) return max_ops > 0 ? fail("irs_slab_trace: %zu operations, room for %d", sch.ops.size(), max_ops) : 0;
    memcpy(ops, sch.ops.data(), sch.ops.size() * sizeof(irs_slab_op));
    return 0;
}

}  // extern "C"
