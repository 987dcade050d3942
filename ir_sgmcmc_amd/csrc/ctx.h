// The context behind include/irsgmcmc.h (shared by api.hip and slab.hip): workspace views, variant prediction, error plumbing.
#pragma once
#include <stdlib.h>
#include <string.h>

#include "kernels.h"
#include "scalar_kernels.h"

struct irs_comm;

namespace irs {

int fail(const char* fmt, ...);  // formats irs_last_error(), returns 1 (api.hip)

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return irs::fail("%s failed: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

#define LAUNCH_CHECK()                                                                              \
    do {                                                                                            \
        hipError_t e_ = hipGetLastError();                                                          \
        if (e_ != hipSuccess) return irs::fail("kernel launch failed: %s", hipGetErrorString(e_)); \
    } while (0)


// z-slab decomposition (slab.hip): this rank owns the planes [a, b) of the volume and HOLDS the planes [lo, hi) (its slab
// plus a ghost margin) of every array; everything else about the context is the single-GPU one.
struct SlabInfo {
    bool on = false;
    int rank = 0, world = 1;
    int a = 0, b = 0;    // owned planes
    int lo = 0, hi = 0;  // held planes
    int margin = 0;      // ghost planes held beyond a neighbour-facing edge
    int gmax = 8;        // widest ghost zone one exchange may carry (communication-avoiding blocks of squaring steps)
    int min_slab = 0;    // smallest slab of any rank: a rank cannot send more planes than it owns
    bool has_lo = false, has_hi = false;  // neighbours below / above
};

}  // namespace irs

struct irs_ctx {
    irs_config cfg;
    irs::DevCfg dcfg;
    irs::Vol vol, volv;
    bool ffd;
    int C;
    irs::LinTables lin;
    irs::Taps sob;
    irs::SplineTaps spl[3];
    char* slab;
    size_t slab_bytes;
    // workspace views (slab mode: UNSHIFTED bases of slab-local arrays; slab.hip applies the plane shift)
    float *steps, *tmpA, *tmpB, *vs, *gA, *gB, *gC, *warped, *z, *sigM, *fhat, *gM, *dense;
    double *stat_partials, *energy_partials, *nll_partials;
    double *stat_sum, *energy_sum, *nll_sum;  // reduced partial sums (staged / slab path)
    unsigned* dmax;  // [no_steps + 1][C][4] max |d_k| in voxels per axis (float bits), by-product of the forward steps
    float* cmm;      // coarse (8^3 cells) min / max of d_k for the source boxes of the any-radius adjoint (kernels.h)
    unsigned* hint = nullptr;  // pinned host copy of dmax as of the last finished transition (written by finalize_kernel, read
                     // by the host WITHOUT synchronisation: a hint that only decides which variants are launched)
    irs::DevState* state;
    int fhat_chains;
    bool fixed_set;
    int nll_blocks;
    int nll_seg_C;  // chains the data term's segment length (and so nll_blocks) was chosen for: C with `data_batch`, else 1
    hipEvent_t ev[8];
    hipEvent_t ev_bwd[64];
    hipEvent_t ra_ev[4];     // end of the last transitions: bounds how far the host may run ahead of the device
    // several chains: the data term of chain c runs on a side stream while the statistics of chain c + 1 run on the caller's
    // (api.hip: the per-chain stage); created with the context when C > 1
    hipStream_t side = nullptr;
    hipEvent_t ev_side[2 * IRS_MAX_CHAINS] = {};
    uint64_t n_enqueued = 0;
    irs::Knobs kn;            // copy of the process-wide switches, taken by irs_create (irs_option_set changes it)
    bool dmax_clean = false;  // the bound scratch was cleared by the finalize kernel of the last transition
    // ---- recovery from failed variant / ghost-width predictions (scalar_kernels.h: Verdict)
    unsigned fails_seen = 0;        // DevState::fails as of the last look
    uint64_t makeup = 0;            // failed (no-op) transitions still to be re-run
    uint64_t fails_total = 0;       // statistics
    uint64_t force_all_until = 0;   // transitions enqueued before this count make no assumptions
    uint64_t exact_until = 0;       // slab: transitions enqueued before this count measure their ghost widths
    float hint_seen[32] = {}, hint_before[32] = {};  // per squaring step: the bound (max over chains / axes) the host last saw published,
                                    // and the DIFFERENT value it saw before that: how fast the bound moves per published snapshot
    bool hint_trend[32] = {};       // both are real observations (not the +inf of a fresh context)
    irs_io last_io;                 // the io of the last irs_transition call (irs_flush re-runs with it)
    bool have_last_io = false;
    // ---- z-slab decomposition (slab.hip)
    irs::SlabInfo sl;
    irs_comm* comm = nullptr;      // not owned
    hipStream_t cs = nullptr;      // communication stream (owned)
    hipEvent_t sev[28];            // 0..15 rotating producer / receive events of the exchanges, 16.. one pair per KIND of all-reduce
    unsigned* plan_hint = nullptr; // pinned, TWO slots of kHintWords: the all-reduced bounds of transition t land in slot t % 2.  The
                                   // plan of transition t reads slot t % 2 = the bounds of t - 2, a transition every rank has
                                   // seen FINISH (the host waits for it): every rank plans from the same numbers.  (The single
                                   // `hint` above is read whenever -- fine for launch decisions, fatal for exchange widths.)
    double* hsum = nullptr;        // pinned: host copy of small reductions in exact mode
    int pred[32];                  // host: ghost-width plan source (bounds of the last exact transition); -1 = none
    bool have_pred = false;
    uint64_t slab_exchanged_bytes = 0;  // bookkeeping for tests / reports
    uint64_t slab_exchanges = 0;
    uint64_t slab_exact = 0;           // transitions run in measuring mode
    uint64_t slab_mispredictions = 0;  // transitions found to have run with too narrow a plan (reported as errors)
    int last_nf = 0, last_nb = 0;      // exchange rounds of the last transition
    // ---- hand-over timeline of one sampled slab transition (irs_slab_timeline_*): timing events around every exchange / all-reduce
    struct TlRec {
        int32_t kind, stage, k, width;
        int eP, eR, eW0, eW1;  // indices into tl_ev (-1: not recorded)
    };
    int tl_arm = 0;                  // > 0: the next transition records
    hipEvent_t* tl_ev = nullptr;     // pool of timing-enabled events, created when first armed
    int tl_ev_n = 0, tl_ev_used = 0;
    TlRec tl_rec[128];               // by comm id
    int tl_n = 0, tl_t0 = -1, tl_t1 = -1;
    bool tl_have = false;
};

namespace irs {

constexpr int kHintWords = 4 * IRS_MAX_CHAINS * 32 + 8;  // dmax scratch + [flags] (slab.hip: misprediction flag)

// Host-side guess of "max |d_k| stays well below `bound`" from the bounds of the last transition the host has seen
// finish (never waited for: stale by a transition or two, and displacements move by O(lr) per transition).  Only a launch
// decision: the kernels that remain are correct for any displacement, so a wrong guess costs time, not parity.
inline bool predicted_below(const irs_ctx* c, int k, float bound) {
    const int mode = c->kn.predict_variants;
    if (mode == 2 || mode == 3) return true;
    if (!mode || !c->hint) return false;
    const volatile unsigned* h = c->hint + (size_t)k * c->C * 4;
    float m = 0.0f;
    for (int i = 0; i < c->C * 4; ++i) {
        const unsigned bits = h[i];
        float f;
        memcpy(&f, &bits, sizeof(f));
        if (!(f >= 0.0f)) return false;  // NaN / garbage
        m = f > m ? f : m;
    }
    return m < bound;
}
// (0.9: the radius-2 FORWARD variant only has work at max|d_k| >= 1, and the radius-1 kernel that remains reads a tap that leaves its
// ring from global memory -- a bound that crosses 1 between this guess and the launch costs that one step some speed, nothing else.
// With 0.75 a chain whose d_11 sits between 0.75 and 1 voxel -- the bench's -- paid an idle launch (~5 us) per transition for nothing.)
// (Round 5: ... and from 0.97 while the bound has been moving by less than 10 % per published snapshot -- the same observed trend as
// predicted_tiny below.  The chain at rest of the bench carries max|d_11| ~ 0.9 voxel: its radius-2 forward launch was idle every time.)
inline bool bound_is_steady(const irs_ctx* c, int k) {
    if (k >= 32 || !c->hint_trend[k]) return false;
    const float a = c->hint_before[k], b = c->hint_seen[k];
    return a > 0.0f && b < 1.1f * a;
}
inline bool predicted_small(const irs_ctx* c, int k) {
    if (predicted_below(c, k, 0.9f)) return true;
#ifdef IRS_NO_TREND
    return false;
#endif
    return c->kn.predict_variants == 1 && bound_is_steady(c, k) && predicted_below(c, k, 0.97f);
}
// "max |d_k| is nowhere near one voxel": the radius-2 ADJOINT variant is not even launched.  Unlike the guesses above this one
// is not backed by a fallback inside the kernel that remains (the radius-1 gather only covers |d| < 1), so it is (a) taken
// from the production heuristic only, with a 2.5x margin -- d_k moves by O(0.1) voxel per transition and the host is at most
// two transitions behind -- and (b) VALIDATED on the device: finalize_kernel compares the bounds of the steps whose variant
// was skipped with 1 and raises a sticky flag that the next irs_transition returns as an error.
// Round 5: ... or below 0.66 voxel while the bound has been moving by less than 10 % per published snapshot (observed, not assumed:
// irs_ctx::hint_seen / hint_before, refreshed by note_hint_trend at every call): two snapshots on that is < 0.8 voxel.  A chain at
// rest in the bench regime carries max|d_10| ~ 0.45 voxel: with the fixed 0.4 rule alone its radius-2 adjoint variant was launched,
// found nothing to do and cost ~8 us (5 us of launch + the gap to the next kernel) every transition -- 1 % of a 128^3 transition.
inline bool predicted_tiny(const irs_ctx* c, int k) {
    if (c->kn.predict_variants == 3) return true;  // test hook, always
    if (c->kn.predict_variants != 1) return false;
    if (predicted_below(c, k, 0.4f)) return true;
#ifdef IRS_NO_TREND  // (A/B builds: the fixed rule alone)
    return false;
#endif
    return bound_is_steady(c, k) && predicted_below(c, k, 0.66f);
}
// refresh the observed trend of the published bounds (host side, unsynchronised reads of the pinned hint -- a hint, like the rest)
inline void note_hint_trend(irs_ctx* c) {
    if (!c->hint) return;
    for (int k = 0; k < c->cfg.no_steps && k < 32; ++k) {
        const volatile unsigned* h = c->hint + (size_t)k * c->C * 4;
        float m = 0.0f;
        bool ok = true;
        for (int i = 0; i < c->C * 4; ++i) {
            const unsigned bits = h[i];
            float f;
            memcpy(&f, &bits, sizeof(f));
            if (!(f >= 0.0f) || f > 1.0e6f) ok = false;  // NaN / +inf (nothing published yet) / garbage
            m = f > m ? f : m;
        }
        if (!ok) {
            c->hint_trend[k] = false;
            c->hint_seen[k] = 0.0f;
            continue;
        }
        if (m != c->hint_seen[k]) {
            c->hint_trend[k] = c->hint_seen[k] > 0.0f;  // a second DIFFERENT observation
            c->hint_before[k] = c->hint_seen[k];
            c->hint_seen[k] = m;
        }
    }
}

// Layouts of the INTERNAL fields of the fused path (exp_kernels.hip: Lay3; bits 1 displacement in, 2 gradient in, 4 out):
// d_1 .. d_{n-1} and the gradients handed from one adjoint step to the next are interleaved ([V][3]); everything that crosses
// into another kernel family -- the velocity in, d_n into the warp, the gradient into the first and out of the last adjoint
// step -- stays planar like the reference's tensors.
inline int fwd_lay(const irs_ctx* c, int k) { return (k > 0 ? 1 : 0) | (k < c->cfg.no_steps - 1 ? 4 : 0); }
inline int bwd_lay(const irs_ctx* c, int k) {
    return (k > 0 ? 1 : 0) | (k < c->cfg.no_steps - 1 ? 2 : 0) | (k > 0 ? 4 : 0);
}

inline void prescale_factors(Vol vol, int no_steps, float s[3]) {
    const double p = 1.0 / (double)(1 << no_steps);
    s[0] = (float)(2.0 / (double)(vol.W - 1) * p);  // x <-> W, y <-> H, z <-> D
    s[1] = (float)(2.0 / (double)(vol.H - 1) * p);
    s[2] = (float)(2.0 / (double)(vol.D - 1) * p);
}

// shared by irs_create and irs_slab_create (api.hip)
int create_ctx(const irs_config* cfg, const SlabInfo* sl, irs_ctx** out);
int check_io(const irs_ctx* c, const irs_io* io, const char* who);
void slab_release(irs_ctx* c);  // slab.hip
int slab_flush(irs_ctx* c, hipStream_t st);  // slab.hip
void slab_drop_pending(irs_ctx* c);          // slab.hip
// cubic B-spline FFD up-sampling / adjoint over the three axes (api.hip); the window arguments are for slab-local dense arrays
int ffd_up(const float* v_cp, float* dense, float* tmp, int C, Vol vol, const int G[3], const SplineTaps spl[3], hipStream_t st,
           int w_lo = 0, int w_n = -1, int store_lo = 0, int store_n = -1);
int ffd_adjoint(const float* g_dense, float* g_cp, float* tmp, int C, Vol vol, const int G[3], const SplineTaps spl[3], hipStream_t st,
                int w_lo = 0, int w_n = -1, int store_lo = 0, int store_n = -1);

}  // namespace irs
