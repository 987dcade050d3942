// C ABI (include/irsgmcmc.h): argument validation, workspace ownership and the launch sequence of one
// SG-MCMC transition.  No exceptions / aborts cross this boundary; errors come back as codes + irs_last_error().
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>

#include "comm.h"
#include "ctx.h"

using namespace irs;

namespace irs {
thread_local char g_err[512] = "";

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}
}  // namespace irs

namespace irs {

static int env_or(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

static Knobs knobs_from_env() {
    Knobs k;
    k.predict_variants = env_or("IRS_PREDICT_VARIANTS", k.predict_variants);
    k.run_ahead = env_or("IRS_RUN_AHEAD", k.run_ahead);
    k.fuse_warp_bwd = env_or("IRS_FUSE_WARP_BWD", k.fuse_warp_bwd);
    k.energy_in_update = env_or("IRS_ENERGY_IN_UPDATE", k.energy_in_update);
    k.fuse_noise = env_or("IRS_FUSE_NOISE", k.fuse_noise);
    k.recover = env_or("IRS_RECOVER", k.recover);
    k.fwd_rows1 = env_or("IRS_FWD_ROWS1", k.fwd_rows1);
    k.coarse_box = env_or("IRS_COARSE_BOX", k.coarse_box);
    k.lds_from = env_or("IRS_LDS_FROM", k.lds_from);
    k.seg_fit = env_or("IRS_SEG_FIT", k.seg_fit);
    k.fwd_pf = env_or("IRS_FWD_PF", k.fwd_pf);
    k.fwd_z2 = env_or("IRS_FWD_Z2", k.fwd_z2);
    k.tile_box = env_or("IRS_TILE_BOX", k.tile_box);
    k.fwd_r2_rows1 = env_or("IRS_FWD_R2_ROWS1", k.fwd_r2_rows1);
    k.ps_rows = env_or("IRS_PS_ROWS", k.ps_rows);
    const char* tile = getenv("IRS_SOBOLEV_TILE");
    if (tile && *tile) k.sobolev_tile = tile[0] == 'b' ? 2 : (tile[0] == 's' ? 1 : atoi(tile));
    k.march_seg = env_or("IRS_MARCH_SEG", k.march_seg);
    k.march_seg_fwd = env_or("IRS_MARCH_SEG_FWD", k.march_seg_fwd);
    k.swz_run = env_or("IRS_SWZ_RUN", k.swz_run);
    k.seg_min_blocks = env_or("IRS_SEG_MIN_BLOCKS", k.seg_min_blocks);
    k.seg_min_len = env_or("IRS_SEG_MIN_LEN", k.seg_min_len);
    k.sobolev_seg = env_or("IRS_SOBOLEV_SEG", k.sobolev_seg);
    k.lcc_seg = env_or("IRS_LCC_SEG", k.lcc_seg);
    k.stats_seg = env_or("IRS_STATS_SEG", k.stats_seg);
    k.update_seg = env_or("IRS_UPDATE_SEG", k.update_seg);
    k.slab_split = env_or("IRS_SLAB_SPLIT", k.slab_split);
    k.slab_buffers = env_or("IRS_SLAB_BUFFERS", k.slab_buffers);
    k.slab_exact = env_or("IRS_SLAB_EXACT", k.slab_exact);
    k.slab_force_h = env_or("IRS_SLAB_FORCE_H", k.slab_force_h);
    k.launch_log = env_or("IRS_LAUNCH_LOG", k.launch_log);
    k.chain_overlap = env_or("IRS_CHAIN_OVERLAP", k.chain_overlap);
    k.data_batch = env_or("IRS_DATA_BATCH", k.data_batch);
    return k;
}

Knobs& global_knobs() {
    static Knobs k = knobs_from_env();  // the only place the library reads IRS_* tuning variables, once per process
    return k;
}

void log_launch(const char* kernel, int tile_x, int tile_y, int64_t blocks, int threads, int seg_len, int run_in, int planes_out,
                int chains, int64_t resident) {
    if (!global_knobs().launch_log) return;
    static uint64_t seen[256];
    static int n_seen = 0;
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
    for (const char* p = kernel; *p; ++p) mix((uint64_t)*p);
    mix((uint64_t)blocks); mix((uint64_t)threads); mix((uint64_t)seg_len); mix((uint64_t)planes_out); mix((uint64_t)chains);
    for (int i = 0; i < n_seen; ++i)
        if (seen[i] == h) return;
    if (n_seen < 256) seen[n_seen++] = h;
    const int steps = (seg_len < planes_out ? seg_len : planes_out) + run_in;
    const double rounds = resident > 0 ? (double)blocks / (double)resident : 0.0;
    fprintf(stderr, "[irs launch] {\"kernel\": \"%s\", \"tile\": [%d, %d], \"workgroups\": %lld, \"threads\": %d, \"seg_len\": %d, \"run_in\": %d, "
                    "\"planes_out\": %d, \"chains\": %d, \"plane_steps\": %d, \"resident\": %lld, \"rounds\": %.3f, \"run_in_overhead\": %.3f}\n",
            kernel, tile_x, tile_y, (long long)blocks, threads, seg_len, run_in, planes_out, chains, steps, (long long)resident, rounds,
            (double)steps / (double)(steps - run_in > 0 ? steps - run_in : 1));
}

static int g_live_contexts = 0;  // contexts alive in this process (a context is not thread-safe, and neither is this count)
void context_born() { ++g_live_contexts; }
void context_gone() { --g_live_contexts; }

// Scope of a switch.  CTX: copied into a context at creation and read from there (`irs_option_set(ctx, ...)` changes that copy,
// `irs_option_set(NULL, ...)` the default of contexts created later -- and the stateless operators).  GLOBAL: the launchers read
// the process-wide value at every launch; naming a context for it is an error (it used to be accepted and ignored).  LAYOUT:
// GLOBAL, and the value also sizes the per-block partial sums a context lays out when it is created (irs_ctx::nll_blocks ...):
// changing it while a context is alive would make launches disagree with that layout, so it is refused then.
enum { KN_CTX = 0, KN_GLOBAL = 1, KN_LAYOUT = 2 };

int knob_set(Knobs& k, const char* name, int value, bool on_context) {
    struct Entry {
        const char* name;
        int Knobs::*field;
        int scope;
    };
    static const Entry table[] = {
        {"predict_variants", &Knobs::predict_variants, KN_CTX}, {"run_ahead", &Knobs::run_ahead, KN_CTX}, {"fuse_warp_bwd", &Knobs::fuse_warp_bwd, KN_CTX},
        {"energy_in_update", &Knobs::energy_in_update, KN_CTX}, {"fuse_noise", &Knobs::fuse_noise, KN_CTX}, {"recover", &Knobs::recover, KN_CTX},
        {"chain_overlap", &Knobs::chain_overlap, KN_CTX}, {"data_batch", &Knobs::data_batch, KN_CTX}, {"slab_split", &Knobs::slab_split, KN_CTX}, {"slab_buffers", &Knobs::slab_buffers, KN_CTX}, {"slab_exact", &Knobs::slab_exact, KN_CTX}, {"slab_force_h", &Knobs::slab_force_h, KN_CTX},
        {"fwd_rows1", &Knobs::fwd_rows1, KN_GLOBAL}, {"coarse_box", &Knobs::coarse_box, KN_GLOBAL}, {"lds_from", &Knobs::lds_from, KN_GLOBAL},
        {"fwd_pf", &Knobs::fwd_pf, KN_GLOBAL}, {"fwd_z2", &Knobs::fwd_z2, KN_GLOBAL}, {"tile_box", &Knobs::tile_box, KN_GLOBAL}, {"fwd_r2_rows1", &Knobs::fwd_r2_rows1, KN_GLOBAL}, {"sobolev_tile", &Knobs::sobolev_tile, KN_GLOBAL},
        {"march_seg", &Knobs::march_seg, KN_GLOBAL}, {"march_seg_fwd", &Knobs::march_seg_fwd, KN_GLOBAL}, {"swz_run", &Knobs::swz_run, KN_GLOBAL},
        {"sobolev_seg", &Knobs::sobolev_seg, KN_GLOBAL}, {"ps_rows", &Knobs::ps_rows, KN_GLOBAL}, {"launch_log", &Knobs::launch_log, KN_GLOBAL},
        {"seg_fit", &Knobs::seg_fit, KN_LAYOUT}, {"seg_min_blocks", &Knobs::seg_min_blocks, KN_LAYOUT}, {"seg_min_len", &Knobs::seg_min_len, KN_LAYOUT},
        {"lcc_seg", &Knobs::lcc_seg, KN_LAYOUT}, {"stats_seg", &Knobs::stats_seg, KN_LAYOUT}, {"update_seg", &Knobs::update_seg, KN_LAYOUT},
    };
    if (!name) return fail("irs_option_set: null name");
    for (const Entry& e : table)
        if (!strcmp(e.name, name)) {
            if (on_context && e.scope != KN_CTX)
                return fail("irs_option_set: '%s' is a process-wide switch (the launchers read it at every launch): set it with ctx == NULL", name);
            if (!on_context && e.scope == KN_LAYOUT && g_live_contexts > 0 && k.*(e.field) != value)
                return fail("irs_option_set: '%s' sizes the partial-sum layout of a context at creation; %d context(s) are alive -- set it before irs_create", name, g_live_contexts);
            k.*(e.field) = value;
            return 0;
        }
    return fail("irs_option_set: unknown option '%s'", name);
}
}  // namespace irs

namespace {

bool dims_ok(int C, int D, int H, int W) {
    // < 2^30 voxels per volume: kernels address within a volume with 32-bit byte offsets (a 1024^3 transition would need
    // 240 GB of workspace anyway)
    return C >= 1 && D >= 2 && H >= 2 && W >= 2 && (int64_t)D * H * W < ((int64_t)1 << 30);
}

SplineTaps make_spline(int cps) {
    // sampled cubic B-spline (utils/transformation.py:79-102); evaluated in double, stored as float like the reference
    SplineTaps t;
    memset(&t, 0, sizeof(t));
    t.cps = cps;
    const int n = 4 * cps - 1, r = n / 2;
    for (int i = 0; i < n; ++i) {
        const double x = fabs((double)(i - r) / (double)cps);
        double v = 0.0;
        if (x < 1.0) v = 2.0 / 3.0 + (0.5 * x - 1.0) * x * x;
        else if (x < 2.0) v = -1.0 * ((x - 2.0) * (x - 2.0) * (x - 2.0)) / 6.0;
        t.k[i] = (float)v;
    }
    return t;
}

int control_points(int n, int cps) { return (int)ceil((double)(n - 1) / (double)cps) + 1 + 2; }  // utils/util.py:61-69

}  // namespace

extern "C" {

const char* irs_last_error(void) { return irs::g_err; }
const char* irs_version(void) { return "ir-sgmcmc-amd 0.1 (gfx950)"; }
size_t irs_reduce_scratch_doubles(void) { return (size_t)kMaxPartialBlocks * IRS_MAX_CHAINS; }

int irs_option_set(irs_ctx* ctx, const char* name, int value) { return knob_set(ctx ? ctx->kn : global_knobs(), name, value, ctx != nullptr); }

// ================================================================================================
// stateless operators
// ================================================================================================

int irs_perturb_smooth(const float* v, const float* sigma, const float* eps, float tau, const float* kernel, int s,
                       int C, int D, int H, int W, float* tmp, float* out, uint64_t seed, uint64_t iteration,
                       void* stream) {
    if (!v || !out || !dims_ok(C, D, H, W)) return fail("irs_perturb_smooth: bad arguments");
    if (s < 0 || s > IRS_MAX_HALF_WIDTH || (s > 0 && (!kernel || !tmp))) return fail("irs_perturb_smooth: bad kernel/s");
    hipStream_t st = (hipStream_t)stream;
    const Vol vol = make_vol(D, H, W);
    const size_t bytes = (size_t)C * 3 * vol.V * sizeof(float);
    if (s == 0) {
        if (tau >= 0.0f) launch_perturb(v, sigma, eps, sqrtf(2.0f * tau), out, C, vol, seed, iteration, nullptr, st);
        else HIP_TRY(hipMemcpyAsync(out, v, bytes, hipMemcpyDeviceToDevice, st));
        LAUNCH_CHECK();
        return 0;
    }
    Taps taps;
    taps.s = s;
    for (int i = 0; i <= 2 * s; ++i) taps.k[i] = kernel[i];
    const float* src = v;
    if (tau >= 0.0f && global_knobs().fuse_noise) {  // the noise is generated while the smoothing kernel stages its planes
        launch_perturb_sobolev_march(v, sigma, eps, (float)sqrt(2.0 * (double)tau), out, taps, C, vol, nullptr, 12, seed, iteration, nullptr, st);
        LAUNCH_CHECK();
        return 0;
    }
    if (tau >= 0.0f) {
        launch_perturb(v, sigma, eps, (float)sqrt(2.0 * (double)tau), out, C, vol, seed, iteration, nullptr, st);
        src = out;
    }
    if (src == out) {  // the marching kernel cannot run in place
        HIP_TRY(hipMemcpyAsync(tmp, out, bytes, hipMemcpyDeviceToDevice, st));
        src = tmp;
    }
    launch_sobolev_march(src, out, taps, C * 3, vol, nullptr, 12, st);
    LAUNCH_CHECK();
    return 0;
}

int irs_svf_exp_fwd(const float* v, float* steps, float* transformation, float* displacement, int no_steps, int C,
                    int D, int H, int W, void* stream) {
    if (!v || !steps || !dims_ok(C, D, H, W) || no_steps < 1 || no_steps > 30) return fail("irs_svf_exp_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const Vol vol = make_vol(D, H, W);
    Lin lin;
    if (cached_lin(D, H, W, st, &lin)) return fail("irs_svf_exp_fwd: identity grid allocation failed");
    const int64_t field = (int64_t)C * 3 * vol.V;
    for (int k = 0; k < no_steps; ++k) {
        const float* in = k == 0 ? v : steps + (int64_t)(k - 1) * field;
        launch_exp_step_fwd_march(in, steps + (int64_t)k * field, k == 0, no_steps, C, vol, lin, nullptr, nullptr, false, 0, st);
    }
    if (transformation || displacement)
        launch_svf_outputs(steps + (int64_t)(no_steps - 1) * field, transformation, displacement, C, vol, lin, st);
    LAUNCH_CHECK();
    return 0;
}

static int exp_backward(const float* v, const float* steps, const float* g_last, float* gA, float* gB, int no_steps, int C,
                        Vol vol, Lin lin, hipStream_t st, float** result) {
    const int64_t field = (int64_t)C * 3 * vol.V;
    const float* G = g_last;
    float* bufs[2] = {gA, gB};
    int cur = 0;
    // Scratch of the stateless operator, one set PER DEVICE (a pointer of device 0 is no use to a launch on device 1), guarded
    // by a mutex and regrown only after the WHOLE device has drained (another stream may still be reading the old block).
    // [32 steps][8 chains][4] bounds + the coarse displacement extrema of the any-radius adjoint (kernels.h).
    struct Scratch {
        unsigned* dmax = nullptr;
        float* cmm = nullptr;
        size_t cmm_bytes = 0;
    };
    static Scratch per_device[64];
    static std::mutex mu;
    int dev_id = 0;
    HIP_TRY(hipGetDevice(&dev_id));
    if (dev_id < 0 || dev_id >= 64) return fail("irs_svf_exp_bwd: device ordinal %d out of range", dev_id);
    unsigned* dmax;
    float* cmm;
    {
        std::lock_guard<std::mutex> lock(mu);
        Scratch& sc = per_device[dev_id];
        if (!sc.dmax) HIP_TRY(hipMalloc((void**)&sc.dmax, sizeof(unsigned) * 4 * IRS_MAX_CHAINS * 32));
        if (coarse_minmax_bytes(vol, C) > sc.cmm_bytes) {
            HIP_TRY(hipDeviceSynchronize());
            if (sc.cmm) HIP_TRY(hipFree(sc.cmm));
            sc.cmm = nullptr;
            sc.cmm_bytes = 0;
            HIP_TRY(hipMalloc((void**)&sc.cmm, coarse_minmax_bytes(vol, C)));
            sc.cmm_bytes = coarse_minmax_bytes(vol, C);
        }
        dmax = sc.dmax;
        cmm = sc.cmm;
    }
    if (C > IRS_MAX_CHAINS || no_steps > 32) return fail("irs_svf_exp_bwd: at most %d chains / 32 steps", IRS_MAX_CHAINS);
    HIP_TRY(hipMemsetAsync(dmax, 0, sizeof(unsigned) * 4 * IRS_MAX_CHAINS * 32, st));
    for (int k = no_steps - 1; k >= 0; --k) {
        float* out = bufs[cur];
        const float* dk = k == 0 ? v : steps + (int64_t)(k - 1) * field;
        // every variant is launched (radius-1 / radius-2 gather, any-radius fixed-point scatter); the device picks by max|d_k|
        launch_field_absmax(dk, k == 0, no_steps, dmax + (int64_t)k * IRS_MAX_CHAINS * 4, C, vol, st);
        launch_exp_step_bwd_march(G, dk, out, k == 0, no_steps, C, vol, lin, dmax + (int64_t)k * IRS_MAX_CHAINS * 4, 2, false, nullptr, 0, nullptr, st);
        launch_exp_step_bwd_lds(G, dk, out, k == 0, no_steps, C, vol, lin, dmax + (int64_t)k * IRS_MAX_CHAINS * 4, 2, 2, nullptr, 0, cmm, st);
        G = out;
        cur ^= 1;
    }
    *result = const_cast<float*>(G);
    return 0;
}

int irs_svf_exp_bwd(const float* v, const float* steps, const float* g_last, float* scratch, float* g_v, int no_steps,
                    int C, int D, int H, int W, void* stream) {
    if (!v || !steps || !g_last || !scratch || !g_v || !dims_ok(C, D, H, W) || no_steps < 1 || no_steps > 30)
        return fail("irs_svf_exp_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const Vol vol = make_vol(D, H, W);
    Lin lin;
    if (cached_lin(D, H, W, st, &lin)) return fail("irs_svf_exp_bwd: identity grid allocation failed");
    const int64_t field = (int64_t)C * 3 * vol.V;
    float* res = nullptr;
    if (exp_backward(v, steps, g_last, scratch, scratch + field, no_steps, C, vol, lin, st, &res)) return 1;
    float s[3];
    prescale_factors(vol, no_steps, s);
    launch_scale_channels(res, g_v, s[0], s[1], s[2], C, vol, st);
    LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

// `dense` / `g_dense` hold the planes [store_lo, store_lo + store_n) of the volume (whole volume: 0, D); up-sampling produces the
// planes [w_lo, w_lo + w_n), the adjoint sums over them (a rank's own planes: partial control-grid gradients, all-reduced)
int irs::ffd_up(const float* v_cp, float* dense, float* tmp, int C, Vol vol, const int G[3], const SplineTaps spl[3], hipStream_t st,
                int w_lo, int w_n, int store_lo, int store_n) {
    // axis order of utils/transformation.py:146-149: tensor axis 2 (D, cps[0]), 3 (H, cps[1]), 4 (W, cps[2])
    if (w_n < 0) { w_lo = 0; w_n = vol.D; }
    if (store_n < 0) { store_lo = 0; store_n = vol.D; }
    const int64_t CC = (int64_t)C * 3;
    float* t1 = tmp;
    float* t2 = tmp + CC * store_n * G[1] * G[2];
    launch_ffd_axis(v_cp, t1, spl[0], false, CC, G[0], vol.D, (int64_t)G[1] * G[2], st, w_lo, w_n, store_lo, store_n);
    launch_ffd_axis(t1, t2, spl[1], false, CC * store_n, G[1], vol.H, G[2], st);
    launch_ffd_axis(t2, dense, spl[2], false, CC * store_n * vol.H, G[2], vol.W, 1, st);
    return 0;
}

int irs::ffd_adjoint(const float* g_dense, float* g_cp, float* tmp, int C, Vol vol, const int G[3], const SplineTaps spl[3],
                     hipStream_t st, int w_lo, int w_n, int store_lo, int store_n) {
    if (w_n < 0) { w_lo = 0; w_n = vol.D; }
    if (store_n < 0) { store_lo = 0; store_n = vol.D; }
    const int64_t CC = (int64_t)C * 3;
    float* t1 = tmp;
    float* t2 = tmp + CC * store_n * vol.H * G[2];
    launch_ffd_axis(g_dense, t1, spl[2], true, CC * store_n * vol.H, vol.W, G[2], 1, st);
    launch_ffd_axis(t1, t2, spl[1], true, CC * store_n, vol.H, G[1], G[2], st);
    launch_ffd_axis(t2, g_cp, spl[0], true, CC, vol.D, G[0], (int64_t)G[1] * G[2], st, w_lo, w_n, store_lo, store_n);
    return 0;
}

extern "C" {

int irs_ffd_up(const float* v_cp, float* dense, float* tmp, int C, int D, int H, int W, int c0, int c1, int c2,
               void* stream) {
    if (!v_cp || !dense || !tmp || !dims_ok(C, D, H, W) || c0 < 1 || c1 < 1 || c2 < 1 || c0 > 8 || c1 > 8 || c2 > 8)
        return fail("irs_ffd_up: bad arguments");
    const Vol vol = make_vol(D, H, W);
    const int G[3] = {control_points(D, c0), control_points(H, c1), control_points(W, c2)};
    const SplineTaps spl[3] = {make_spline(c0), make_spline(c1), make_spline(c2)};
    ffd_up(v_cp, dense, tmp, C, vol, G, spl, (hipStream_t)stream);
    LAUNCH_CHECK();
    return 0;
}

int irs_ffd_adjoint(const float* g_dense, float* g_cp, float* tmp, int C, int D, int H, int W, int c0, int c1, int c2,
                    void* stream) {
    if (!g_dense || !g_cp || !tmp || !dims_ok(C, D, H, W) || c0 < 1 || c1 < 1 || c2 < 1 || c0 > 8 || c1 > 8 || c2 > 8)
        return fail("irs_ffd_adjoint: bad arguments");
    const Vol vol = make_vol(D, H, W);
    const int G[3] = {control_points(D, c0), control_points(H, c1), control_points(W, c2)};
    const SplineTaps spl[3] = {make_spline(c0), make_spline(c1), make_spline(c2)};
    ffd_adjoint(g_dense, g_cp, tmp, C, vol, G, spl, (hipStream_t)stream);
    LAUNCH_CHECK();
    return 0;
}

int irs_warp_fwd(const float* im, int Cim, const float* d_last, const float* unif, float alpha, float* warped, int C,
                 int D, int H, int W, uint64_t seed, uint64_t iteration, void* stream) {
    if (!im || !d_last || !warped || !dims_ok(C, D, H, W) || (Cim != 1 && Cim != C)) return fail("irs_warp_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const Vol vol = make_vol(D, H, W);
    Lin lin;
    if (cached_lin(D, H, W, st, &lin)) return fail("irs_warp_fwd: identity grid allocation failed");
    launch_warp_fwd(im, Cim == 1 ? 0 : vol.V, d_last, unif, alpha, warped, nullptr, 0, C, vol, lin, seed, iteration, nullptr, st);
    LAUNCH_CHECK();
    return 0;
}

int irs_warp_bwd(const float* im, int Cim, const float* d_last, const float* unif, float alpha, const float* g_warped,
                 float* g_d, int C, int D, int H, int W, uint64_t seed, uint64_t iteration, void* stream) {
    if (!im || !d_last || !g_warped || !g_d || !dims_ok(C, D, H, W) || (Cim != 1 && Cim != C))
        return fail("irs_warp_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const Vol vol = make_vol(D, H, W);
    Lin lin;
    if (cached_lin(D, H, W, st, &lin)) return fail("irs_warp_bwd: identity grid allocation failed");
    launch_warp_bwd(im, Cim == 1 ? 0 : vol.V, d_last, unif, alpha, g_warped, g_d, C, vol, lin, seed, iteration, nullptr, st);
    LAUNCH_CHECK();
    return 0;
}

int irs_warp_transformation(const float* im, int Cim, const float* transformation, float* warped, int C, int D, int H,
                            int W, void* stream) {
    if (!im || !transformation || !warped || !dims_ok(C, D, H, W) || (Cim != 1 && Cim != C))
        return fail("irs_warp_transformation: bad arguments");
    const Vol vol = make_vol(D, H, W);
    launch_warp_transformation(im, Cim == 1 ? 0 : vol.V, transformation, warped, C, vol, (hipStream_t)stream);
    LAUNCH_CHECK();
    return 0;
}

int irs_warp_nearest_u8(const uint8_t* seg, int Cim, const float* transformation, uint8_t* out, int C, int D, int H,
                        int W, void* stream) {
    if (!seg || !transformation || !out || !dims_ok(C, D, H, W) || (Cim != 1 && Cim != C))
        return fail("irs_warp_nearest_u8: bad arguments");
    const Vol vol = make_vol(D, H, W);
    launch_warp_nearest_u8(seg, Cim == 1 ? 0 : vol.V, transformation, out, C, vol, (hipStream_t)stream);
    LAUNCH_CHECK();
    return 0;
}

int irs_warp_nearest_i16(const int16_t* seg, int Cim, const float* transformation, int16_t* out, int C, int D, int H,
                         int W, void* stream) {
    if (!seg || !transformation || !out || !dims_ok(C, D, H, W) || (Cim != 1 && Cim != C))
        return fail("irs_warp_nearest_i16: bad arguments");
    const Vol vol = make_vol(D, H, W);
    launch_warp_nearest_i16(seg, Cim == 1 ? 0 : vol.V, transformation, out, C, vol, (hipStream_t)stream);
    LAUNCH_CHECK();
    return 0;
}

static bool lcc_ok(int s, int D, int H, int W) { return (s == 1 || s == 2) && D > 2 * s && H > 2 * s && W > 2 * s; }

int irs_lcc_normalise(const float* im, float* out, float* sigma_out, int s, int C, int D, int H, int W, void* stream) {
    if (!im || !out || !dims_ok(C, D, H, W)) return fail("irs_lcc_normalise: bad arguments");
    if (!lcc_ok(s, D, H, W)) return fail("irs_lcc_normalise: LCC half width must be 1 or 2 and smaller than half the volume");
    launch_lcc_fwd_march(nullptr, 0, im, out, sigma_out, s, C, make_vol(D, H, W), (hipStream_t)stream);
    LAUNCH_CHECK();
    return 0;
}

int irs_lcc_map_fwd(const float* fhat, int Cf, const float* warped, float* z, float* sigma_m, int s, int C, int D,
                    int H, int W, void* stream) {
    if (!fhat || !warped || !z || !sigma_m || !dims_ok(C, D, H, W) || (Cf != 1 && Cf != C))
        return fail("irs_lcc_map_fwd: bad arguments");
    if (!lcc_ok(s, D, H, W)) return fail("irs_lcc_map_fwd: LCC half width must be 1 or 2 and smaller than half the volume");
    const Vol vol = make_vol(D, H, W);
    launch_lcc_fwd_march(fhat, Cf == 1 ? 0 : vol.V, warped, z, sigma_m, s, C, vol, (hipStream_t)stream);
    LAUNCH_CHECK();
    return 0;
}

int irs_lcc_map_bwd(const float* fhat, int Cf, const float* z, const float* sigma_m, const float* g_z, float* g_warped,
                    int s, int C, int D, int H, int W, void* stream) {
    if (!fhat || !z || !sigma_m || !g_z || !g_warped || !dims_ok(C, D, H, W) || (Cf != 1 && Cf != C))
        return fail("irs_lcc_map_bwd: bad arguments");
    if (!lcc_ok(s, D, H, W)) return fail("irs_lcc_map_bwd: LCC half width must be 1 or 2 and smaller than half the volume");
    const Vol vol = make_vol(D, H, W);
    for (int c = 0; c < C; ++c)
        launch_data_bwd(IRS_DATA_GMM_LCC, fhat + (Cf == 1 ? 0 : (int64_t)c * vol.V), 0, z + (int64_t)c * vol.V,
                        sigma_m + (int64_t)c * vol.V, nullptr, 0, g_z + (int64_t)c * vol.V, nullptr, c,
                        g_warped + (int64_t)c * vol.V, nullptr, s, 1, vol, (hipStream_t)stream);
    LAUNCH_CHECK();
    return 0;
}

int irs_reg_energy(const float* v, double* y_out, double* partials, int C, int D, int H, int W, void* stream) {
    if (!v || !y_out || !partials || !dims_ok(C, D, H, W) || C > IRS_MAX_CHAINS) return fail("irs_reg_energy: bad arguments");
    const Vol vol = make_vol(D, H, W);
    launch_reg_energy(v, partials, C, vol, (hipStream_t)stream);
    launch_reduce_partials(partials, energy_blocks(vol), C, y_out, (hipStream_t)stream);
    LAUNCH_CHECK();
    return 0;
}

int irs_gradient_operator(const float* v, float* nabla, int transformation, int C, int D, int H, int W, void* stream) {
    if (!v || !nabla || !dims_ok(C, D, H, W)) return fail("irs_gradient_operator: bad arguments");
    launch_gradient_operator(v, nabla, transformation, C, make_vol(D, H, W), (hipStream_t)stream);
    LAUNCH_CHECK();
    return 0;
}

int irs_log_det_jacobian(const float* transformation, float* log_det, long long* nan_count, int C, int D, int H, int W,
                         void* stream) {
    if (!transformation || !nan_count || !dims_ok(C, D, H, W)) return fail("irs_log_det_jacobian: bad arguments");
    launch_log_det_jacobian(transformation, log_det, nan_count, C, make_vol(D, H, W), (hipStream_t)stream);
    LAUNCH_CHECK();
    return 0;
}

// ================================================================================================
// context
// ================================================================================================

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

}  // extern "C"

int irs::create_ctx(const irs_config* cfg, const SlabInfo* sl, irs_ctx** out) {
    if (!cfg || !out) return fail("irs_create: null argument");
    const int D = cfg->dims[0], H = cfg->dims[1], W = cfg->dims[2], C = cfg->no_chains;
    if (!dims_ok(C, D, H, W) || C > IRS_MAX_CHAINS) return fail("irs_create: bad dims / chains (C <= %d)", IRS_MAX_CHAINS);
    if (cfg->no_steps < 1 || cfg->no_steps > 30) return fail("irs_create: no_steps out of range");
    if (cfg->sobolev_s < 0 || cfg->sobolev_s > IRS_MAX_HALF_WIDTH) return fail("irs_create: sobolev_s out of range");
    if (cfg->data_loss != IRS_DATA_GMM_LCC && cfg->data_loss != IRS_DATA_SSD) return fail("irs_create: unknown data loss");
    if (cfg->data_loss == IRS_DATA_GMM_LCC) {
        if (!lcc_ok(cfg->lcc_s, D, H, W)) return fail("irs_create: LCC half width must be 1 or 2");
        if (cfg->gmm_components < 1 || cfg->gmm_components > IRS_MAX_COMPONENTS) return fail("irs_create: 1..%d mixture components", IRS_MAX_COMPONENTS);
    } else if (!(cfg->ssd_sigma > 0.0f)) return fail("irs_create: ssd_sigma must be positive");
    if (cfg->reg_loss < IRS_REG_L2 || cfg->reg_loss > IRS_REG_LOGNORMAL_L2) return fail("irs_create: unknown regulariser");
    if ((cfg->reg_loss == IRS_REG_STUDENT || cfg->reg_loss == IRS_REG_LOGNORMAL_L2) && cfg->reg_learnable)
        return fail("irs_create: RegLoss_Student / RegLoss_LogNormal_L2 have no learnable parameters");
    if (cfg->reg_loss == IRS_REG_STUDENT && !(cfg->w_reg_prior_rate > 0.0 && cfg->w_reg_prior_shape > 0.0))
        return fail("irs_create: RegLoss_Student needs a0 > 0 and b0 > 0 (w_reg_prior_shape / w_reg_prior_rate)");
    const bool any_cps = cfg->cps[0] || cfg->cps[1] || cfg->cps[2];
    if (any_cps && (cfg->cps[0] < 1 || cfg->cps[1] < 1 || cfg->cps[2] < 1 || cfg->cps[0] > 8 || cfg->cps[1] > 8 || cfg->cps[2] > 8))
        return fail("irs_create: control point spacing must be 1..8 on every axis");

    irs_ctx* c = new (std::nothrow) irs_ctx();
    if (!c) return fail("irs_create: out of host memory");
    memset((void*)c, 0, sizeof(*c));
    context_born();
    c->kn = global_knobs();
    c->cfg = *cfg;
    c->C = C;
    c->vol = make_vol(D, H, W);
    if (sl && sl->on) {  // slab-local arrays: the channel / chain stride is the number of HELD planes (common.h: Vol)
        c->sl = *sl;
        c->vol.V = (int64_t)(sl->hi - sl->lo) * H * W;
    }
    c->ffd = any_cps;
    c->volv = c->ffd ? make_vol(control_points(D, cfg->cps[0]), control_points(H, cfg->cps[1]), control_points(W, cfg->cps[2]))
                     : c->vol;
    // (a velocity grid narrower than the 2 s + 1 taps of the Sobolev kernel -- an SVFFD control grid of a small volume -- is fine: every
    // smoothing kernel reads through clamped coordinates, which IS the reference's replicate padding, however often a tap folds back)
    c->sob.s = cfg->sobolev_s;
    for (int i = 0; i <= 2 * cfg->sobolev_s; ++i) c->sob.k[i] = cfg->sobolev_kernel[i];
    if (c->ffd)
        for (int a = 0; a < 3; ++a) c->spl[a] = make_spline(cfg->cps[a]);

    DevCfg& d = c->dcfg;
    d.K = cfg->data_loss == IRS_DATA_GMM_LCC ? cfg->gmm_components : 1;
    d.mode = cfg->data_loss;
    d.vd = cfg->virtual_decimation;
    d.C = C;
    d.gmm_lr_log_std = cfg->gmm_lr_log_std;
    d.gmm_lr_logits = cfg->gmm_lr_logits;
    d.gmm_lr_decay = cfg->gmm_lr_decay;
    d.beta1 = cfg->adam_beta1 > 0 ? cfg->adam_beta1 : 0.9f;
    d.beta2 = cfg->adam_beta2 > 0 ? cfg->adam_beta2 : 0.999f;
    d.eps = cfg->adam_eps > 0 ? cfg->adam_eps : 1e-8f;
    d.scale_prior_loc = cfg->scale_prior_loc;
    d.scale_prior_scale = cfg->scale_prior_scale;
    for (int k = 0; k < IRS_MAX_COMPONENTS; ++k) d.conc[k] = cfg->dirichlet_concentration[k];
    d.reg_loss = cfg->reg_loss;
    d.reg_learnable = cfg->reg_learnable;
    d.dof = cfg->dof;
    d.reg_lr0 = cfg->reg_lr0;
    d.reg_lr1 = cfg->reg_lr1;
    d.reg_lr_decay = cfg->reg_lr_decay;
    d.loc_prior_nu = cfg->loc_prior_nu;
    d.loc_prior_w_reg = cfg->loc_prior_w_reg;
    d.reg_scale_prior_loc = cfg->reg_scale_prior_loc;
    d.reg_scale_prior_scale = cfg->reg_scale_prior_scale;
    d.w_reg_prior_shape = cfg->w_reg_prior_shape;
    d.w_reg_prior_rate = cfg->w_reg_prior_rate;

    // ---- one slab for the whole workspace
    const size_t fieldI = (size_t)C * 3 * c->vol.V * sizeof(float);   // image-grid field
    const size_t fieldV = (size_t)C * 3 * c->volv.V * sizeof(float);  // velocity-grid field
    const size_t imageI = (size_t)C * c->vol.V * sizeof(float);
    // scratch of the three axis passes: (C 3, planes held, G1 G2) + the larger of (., H, G2) [up] and (., H, G2) after (., H W -> G2) [adjoint]
    size_t ffd_tmp = 0;
    if (c->ffd) {
        const size_t planes = (size_t)(c->vol.V / ((int64_t)H * W)), g1 = c->volv.H, g2 = c->volv.W;
        ffd_tmp = sizeof(float) * (size_t)C * 3 * planes * ((size_t)g1 * g2 + (size_t)H * g2 + (size_t)H * W);
    }
    // several chains in one (unsharded) engine: their data terms run as ONE launch (transition: `data_batch`), so the segment length
    // is the one that fits ALL chains into a resident set -- at 128^3, C = 2: 7-plane segments, 1216 workgroups of 11 plane steps in
    // one round instead of two launches of 1024 with 8 each
    c->nll_seg_C = (C > 1 && !sl && cfg->data_loss == IRS_DATA_GMM_LCC && c->kn.data_batch != 0) ? C : 1;
    c->nll_blocks = data_bwd_blocks(cfg->data_loss, c->vol, c->nll_seg_C);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    const size_t o_steps = take(fieldI * cfg->no_steps);
    const size_t o_tmpA = take(fieldV > ffd_tmp ? fieldV : ffd_tmp);
    const size_t o_tmpB = take(fieldV);
    const size_t o_vs = take(fieldV);
    const size_t o_gA = take(fieldI), o_gB = take(fieldI);
    // a z-slab of several ranks rotates the adjoint's gradient through THREE fields: a backward exchange round may then span three
    // squaring steps (slab.hip: plan_rounds) -- held planes only, 1 / world of the volume
    const bool third = sl && sl->on && sl->world > 1;
    const size_t o_gC = third ? take(fieldI) : 0;
    const size_t o_warped = take(imageI), o_z = take(imageI), o_sig = take(imageI), o_fhat = take(imageI), o_gM = take(imageI);
    const size_t o_dense = take(c->ffd ? fieldI : 0);
    const size_t o_stat = take(sizeof(double) * kMaxPartialBlocks * kStatVals);
    const size_t o_energy = take(sizeof(double) * kMaxPartialBlocks * IRS_MAX_CHAINS);
    const size_t o_nll = take(sizeof(double) * (size_t)c->nll_blocks * C);
    const size_t o_sums = take(sizeof(double) * (kStatVals + 2 * IRS_MAX_CHAINS));
    const size_t o_dmax = take(sizeof(unsigned) * 4 * IRS_MAX_CHAINS * 32);
    const size_t o_cmm = take(coarse_minmax_bytes(c->vol, C));
    const size_t o_state = take(sizeof(DevState));
    c->slab_bytes = off;
    if (hipMalloc((void**)&c->slab, off) != hipSuccess) {
        delete c;
        context_gone();
        return fail("irs_create: hipMalloc of %zu workspace bytes failed", off);
    }
    if (c->sl.on) (void)hipMemset(c->slab, 0, off);  // ghost planes nobody has written yet must hold finite values
    c->steps = (float*)(c->slab + o_steps);
    c->tmpA = (float*)(c->slab + o_tmpA);
    c->tmpB = (float*)(c->slab + o_tmpB);
    c->vs = (float*)(c->slab + o_vs);
    c->gA = (float*)(c->slab + o_gA);
    c->gB = (float*)(c->slab + o_gB);
    c->gC = third ? (float*)(c->slab + o_gC) : nullptr;
    c->warped = (float*)(c->slab + o_warped);
    c->z = (float*)(c->slab + o_z);
    c->sigM = (float*)(c->slab + o_sig);
    c->fhat = (float*)(c->slab + o_fhat);
    c->gM = (float*)(c->slab + o_gM);
    c->dense = c->ffd ? (float*)(c->slab + o_dense) : nullptr;
    c->stat_partials = (double*)(c->slab + o_stat);
    c->energy_partials = (double*)(c->slab + o_energy);
    c->nll_partials = (double*)(c->slab + o_nll);
    c->stat_sum = (double*)(c->slab + o_sums);
    c->energy_sum = c->stat_sum + kStatVals;
    c->nll_sum = c->energy_sum + IRS_MAX_CHAINS;
    c->dmax = (unsigned*)(c->slab + o_dmax);
    c->cmm = (float*)(c->slab + o_cmm);
    c->state = (DevState*)(c->slab + o_state);

    if (ensure_lin_tables(c->lin, D, H, W, nullptr)) {
        (void)hipFree(c->slab);
        delete c;
        context_gone();
        return fail("irs_create: identity grid allocation failed");
    }
    // initial hyper-parameters as the reference constructors set them (model/loss.py:49-50,191-192,298-303)
    DevState init;
    memset(&init, 0, sizeof(init));
    init.K = d.K;
    init.mode = d.mode;
    init.ssd_inv_sigma = cfg->data_loss == IRS_DATA_SSD ? 1.0f / cfg->ssd_sigma : 0.0f;
    if (cfg->reg_loss == IRS_REG_L2 || cfg->reg_loss == IRS_REG_LOGNORMAL_L2) init.st.reg_param[0] = log((double)cfg->w_reg);
    // (RegLoss_LogNormal's loc / log_scale need digamma: the host wrapper sets them through irs_set_state)
    for (int ch = 0; ch < IRS_MAX_CHAINS; ++ch) init.sc.alpha[ch] = 1.0;
    hipError_t e = hipMemcpy(c->state, &init, sizeof(init), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_refresh_derived(c->state, c->dcfg, nullptr);
        e = hipDeviceSynchronize();
    }
    for (int i = 0; i < 8 && e == hipSuccess; ++i) e = hipEventCreate(&c->ev[i]);
    for (int i = 0; i < 64 && e == hipSuccess; ++i) e = hipEventCreate(&c->ev_bwd[i]);
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->ra_ev[i], hipEventDisableTiming);
    if (c->C > 1 && !c->sl.on && c->kn.chain_overlap) {  // the fused engine with several chains: side stream of the per-chain stage (off by default)
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
        for (int i = 0; i < 2 * c->C && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->ev_side[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->hint, sizeof(unsigned) * kHintWords, hipHostMallocDefault);
    if (e == hipSuccess)
        for (int i = 0; i < kHintWords; ++i) c->hint[i] = i < kHintWords - 8 ? 0x7f800000u : 0u;  // +inf: nothing known yet, launch every variant; flags clear
    if (e != hipSuccess) {
        (void)hipFree(c->slab);
        delete c;
        context_gone();
        return fail("irs_create: state initialisation failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return 0;
}

extern "C" {

int irs_create(const irs_config* cfg, irs_ctx** out) { return irs::create_ctx(cfg, nullptr, out); }

void irs_destroy(irs_ctx* c) {
    if (!c) return;
    (void)hipDeviceSynchronize();
    if (c->sl.on) slab_release(c);
    for (int i = 0; i < 8; ++i)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < 64; ++i)
        if (c->ev_bwd[i]) (void)hipEventDestroy(c->ev_bwd[i]);
    for (int i = 0; i < 4; ++i)
        if (c->ra_ev[i]) (void)hipEventDestroy(c->ra_ev[i]);
    for (int i = 0; i < 2 * IRS_MAX_CHAINS; ++i)
        if (c->ev_side[i]) (void)hipEventDestroy(c->ev_side[i]);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->lin.dev) (void)hipFree(c->lin.dev);
    if (c->hint) (void)hipHostFree(c->hint);
    if (c->slab) (void)hipFree(c->slab);
    delete c;
    context_gone();
}

size_t irs_workspace_bytes(const irs_ctx* c) { return c ? c->slab_bytes : 0; }

int irs_velocity_dims(const irs_ctx* c, int32_t out[3]) {
    if (!c || !out) return fail("irs_velocity_dims: null argument");
    out[0] = c->volv.D;
    out[1] = c->volv.H;
    out[2] = c->volv.W;
    return 0;
}

int irs_set_fixed(irs_ctx* c, const float* fixed_im, int fixed_chains, void* stream) {
    if (!c || !fixed_im || (fixed_chains != 1 && fixed_chains != c->C)) return fail("irs_set_fixed: bad arguments");
    if (c->cfg.data_loss == IRS_DATA_GMM_LCC) {
        Vol w = c->vol;
        int64_t shift = 0;
        if (c->sl.on) {  // slab-local image: normalise where the 2 s input planes either side are held (or are replicate padding)
            const int ls = c->cfg.lcc_s;
            w = window(c->vol, c->sl.lo + (c->sl.lo > 0 ? 2 * ls : 0), c->sl.hi - (c->sl.hi < c->vol.D ? 2 * ls : 0));
            shift = (int64_t)c->sl.lo * c->vol.H * c->vol.W;
        }
        launch_lcc_fwd_march(nullptr, 0, fixed_im - shift, c->fhat - shift, nullptr, c->cfg.lcc_s, fixed_chains, w, (hipStream_t)stream);
        LAUNCH_CHECK();
    }
    c->fhat_chains = fixed_chains;
    c->fixed_set = true;
    return 0;
}

// A slab context whose transport has FAILED (a peer gone: csrc/ipc.hip, fail-safe timeout) cannot flush -- nothing can be re-run --
// but what the device holds is well defined: the state after the last GOOD transition (the failed one was a no-op).  Reading it
// must still work: it is what a dying run checkpoints.
static int flush_or_failed_transport(irs_ctx* c, void* stream) {
    if (!irs_flush(c, stream)) return 0;
    if (!(c->sl.on && c->comm && irs::comm_check(c->comm))) return 1;
    (void)hipStreamSynchronize((hipStream_t)stream);
    if (c->cs) (void)hipStreamSynchronize(c->cs);
    return 0;
}

int irs_get_state(irs_ctx* c, irs_state* out, void* stream) {
    if (!c || !out) return fail("irs_get_state: null argument");
    if (flush_or_failed_transport(c, stream)) return 1;  // transitions that ended as no-ops are re-run first: the state is final
    HIP_TRY(hipMemcpyAsync(out, &c->state->st, sizeof(irs_state), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

static void drop_pending(irs_ctx* c);

int irs_set_state(irs_ctx* c, const irs_state* in, void* stream) {
    if (!c || !in) return fail("irs_set_state: null argument");
    // The chain is being replaced (resume, hand-over from the VI stage): transitions of the OLD chain that were dropped by a
    // failed prediction and not re-run yet must not be re-run on the restored one.  Wait, take note of the count, forget them.
    {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing((hipStream_t)stream, &cap);
        if (cap != hipStreamCaptureStatusNone) return fail("irs_set_state: the stream is being captured -- this call waits for the stream, which a capture forbids");
    }
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (c->sl.on) irs::slab_drop_pending(c);
    else drop_pending(c);
    HIP_TRY(hipMemcpyAsync(&c->state->st, in, sizeof(irs_state), hipMemcpyHostToDevice, (hipStream_t)stream));
    launch_refresh_derived(c->state, c->dcfg, (hipStream_t)stream);
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int irs_get_scalars(irs_ctx* c, irs_scalars* out, void* stream) {
    if (!c || !out) return fail("irs_get_scalars: null argument");
    if (flush_or_failed_transport(c, stream)) return 1;
    HIP_TRY(hipMemcpyAsync(out, &c->state->sc, sizeof(irs_scalars), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

// ================================================================================================
// the transition
// ================================================================================================

}  // extern "C"

int irs::check_io(const irs_ctx* c, const irs_io* io, const char* who) {
    if (!c || !io) return fail("%s: null argument", who);
    if (!io->fixed_im || !io->moving_im || !io->mask) return fail("%s: fixed_im, moving_im and mask are required", who);
    auto ok = [&](int n) { return n == 1 || n == c->C; };
    if (!ok(io->fixed_chains) || !ok(io->moving_chains) || !ok(io->mask_chains))
        return fail("%s: *_chains must be 1 or no_chains", who);
    if (c->cfg.data_loss == IRS_DATA_GMM_LCC && !c->fixed_set) return fail("%s: call irs_set_fixed first", who);
    return 0;
}

extern "C" {

// velocity (v + noise, smoothed) -> vs; dense velocity -> d_1..d_n; warp -> warped; residual -> z (+ sigM)
static int forward_pass(irs_ctx* c, const irs_io* io, const float* v, bool with_noise, bool with_jitter, float* vs,
                        float* warped, float* z, float* gradm, int chains, hipStream_t st, int timed) {
    const irs_config& cfg = c->cfg;
    const int C = chains;
    const uint64_t* it = &c->state->st.iteration;
    // (the finalize kernel of a transition leaves the bound scratch cleared for the next one)
    if (!c->dmax_clean) HIP_TRY(hipMemsetAsync(c->dmax, 0, sizeof(unsigned) * 4 * c->C * (cfg.no_steps + 1), st));
    c->dmax_clean = false;
    // 1. SGLD perturbation + Sobolev smoothing: one kernel that generates the noise while it stages its planes (v + noise is
    //    never materialised); the two-kernel form for the mixture initialisation (no noise) and the small SVFFD control grid
    bool have_dmax0 = false;
    const float amp = (float)sqrt(2.0 * (double)cfg.lr);
    // (a sigma FIELD -- the preconditioner of a chain started from the VI posterior -- takes the kernel's 32 x 16 tile: 99 VGPRs, two
    // workgroups per CU; only sigma together with INJECTED noise, which tests use, keeps the two-kernel form)
    if (with_noise && cfg.sobolev_s > 0 && !c->ffd && c->kn.fuse_noise && !(io->sigma && io->eps)) {
        have_dmax0 = true;
        launch_perturb_sobolev_march(v, io->sigma, io->eps, amp, vs, c->sob, C, c->volv, c->dmax, cfg.no_steps, cfg.seed, 0, it, st);
    } else {
        float* first = cfg.sobolev_s > 0 ? c->tmpA : vs;
        if (with_noise) launch_perturb(v, io->sigma, io->eps, amp, first, C, c->volv, cfg.seed, 0, it, st);
        else HIP_TRY(hipMemcpyAsync(first, v, (size_t)C * 3 * c->volv.V * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (cfg.sobolev_s > 0) {
            have_dmax0 = !c->ffd;
            launch_sobolev_march(c->tmpA, vs, c->sob, C * 3, c->volv, have_dmax0 ? c->dmax : nullptr, cfg.no_steps, st);
        }
    }
    // 2. dense velocity
    const float* dense = vs;
    if (c->ffd) {
        const int G[3] = {c->volv.D, c->volv.H, c->volv.W};
        ffd_up(vs, c->dense, c->tmpA, C, c->vol, G, c->spl, st);
        dense = c->dense;
    }
    // 3. scaling and squaring
    if (timed) HIP_TRY(hipEventRecord(c->ev[1], st));
    const int64_t field = (int64_t)c->C * 3 * c->vol.V;
    const Lin lin = c->lin.lin();
    if (!have_dmax0) launch_field_absmax(dense, true, cfg.no_steps, c->dmax, C, c->vol, st);  // bound of d_0
    for (int k = 0; k < cfg.no_steps; ++k) {
        const float* in = k == 0 ? dense : c->steps + (int64_t)(k - 1) * field;
        float* out = c->steps + (int64_t)k * field;
        launch_exp_step_fwd_march(in, out, k == 0, cfg.no_steps, C, c->vol, lin, c->dmax + (int64_t)k * c->C * 4,
                                  c->dmax + (int64_t)(k + 1) * c->C * 4, predicted_small(c, k), fwd_lay(c, k), st);
    }
    if (timed) HIP_TRY(hipEventRecord(c->ev[2], st));
    const float* d_last = c->steps + (int64_t)(cfg.no_steps - 1) * field;
    // 4. warp (+ jitter) and residual
    const float alpha = with_jitter ? cfg.uniform_alpha : 0.0f;
    launch_warp_fwd(io->moving_im, io->moving_chains == 1 ? 0 : c->vol.V, d_last, io->unif, alpha, warped, gradm, 1, C, c->vol, lin,
                    cfg.seed, 0, it, st);
    if (cfg.data_loss == IRS_DATA_GMM_LCC)
        launch_lcc_fwd_march(c->fhat, c->fhat_chains == 1 ? 0 : c->vol.V, warped, z, c->sigM, cfg.lcc_s, C, c->vol, st);
    else
        launch_residual_ssd(io->fixed_im, io->fixed_chains == 1 ? 0 : c->vol.V, warped, z, C, c->vol, st);
    LAUNCH_CHECK();
    return 0;
}

int irs_gmm_init(irs_ctx* c, const irs_io* io, const float* v_sample, int warm_up, void* stream) {
    if (check_io(c, io, "irs_gmm_init")) return 1;
    if (c->cfg.data_loss != IRS_DATA_GMM_LCC) return 0;
    hipStream_t st = (hipStream_t)stream;
    // trainer.py:529-547: one velocity sample (no Langevin noise, no jitter), batch of one
    // staged in tmpB: a velocity-grid-sized buffer the forward pass does not touch (gA is image-grid-sized, and the control
    // grid of SVFFD with cps = 1 is LARGER than the image grid)
    const size_t bytes = (size_t)3 * c->volv.V * sizeof(float);
    if (v_sample) HIP_TRY(hipMemcpyAsync(c->tmpB, v_sample, bytes, hipMemcpyDeviceToDevice, st));
    else HIP_TRY(hipMemsetAsync(c->tmpB, 0, bytes, st));
    if (forward_pass(c, io, c->tmpB, false, false, c->vs, c->warped, c->z, nullptr, 1, st, 0)) return 1;
    launch_masked_moments(c->z, io->mask, c->stat_partials, c->vol, st);
    launch_gmm_init_from_moments(c->state, c->stat_partials, stats_blocks(c->vol), c->dcfg, st);
    launch_stats(c->cfg.virtual_decimation, c->z, io->mask, c->state, c->stat_partials, c->vol, st, c->dcfg.K);
    launch_chain_scalar(c->state, c->stat_partials, stats_blocks(c->vol), 0, 1, c->dcfg, st);  // alpha, fixed below
    for (int i = 0; i < warm_up; ++i) {
        launch_stats(0, c->z, io->mask, c->state, c->stat_partials, c->vol, st, c->dcfg.K);
        launch_chain_scalar(c->state, c->stat_partials, stats_blocks(c->vol), 0, 2, c->dcfg, st);
    }
    LAUNCH_CHECK();
    return 0;
}

// One transition, enqueued.  `no_assumptions`: launch every kernel variant (nothing about max|d_k| is assumed, the transition
// cannot end as a no-op) -- the mode of the re-runs after a failed prediction.
static int enqueue_transition(irs_ctx* c, const irs_io* io, hipStream_t st, int timed, bool no_assumptions) {

    const irs_config& cfg = c->cfg;
    const int C = c->C;
    const Vol vol = c->vol, volv = c->volv;
    const Lin lin = c->lin.lin();
    float* vs = io->curr_state ? io->curr_state : c->vs;
    float* warped = io->im_moving_warped ? io->im_moving_warped : c->warped;
    float* z = io->residuals ? io->residuals : c->z;
    const uint64_t* it = &c->state->st.iteration;
    const int saved_mode = c->kn.predict_variants;
    if (no_assumptions) c->kn.predict_variants = 0;
    struct Restore {
        irs_ctx* c;
        int mode;
        ~Restore() { c->kn.predict_variants = mode; }
    } restore{c, saved_mode};

    // Which adjoint variants this transition launches, decided NOW from the bounds the host last saw (never waited for).  The
    // any-radius LDS-scatter kernel is launched only when the bound of d_k is near 2 voxels; otherwise the (rarely selected)
    // radius-2 kernel owns everything above one voxel -- through its generic in-kernel fallback if the bound exceeds its ring
    // after all.  Below 0.4 voxel (2.5x margin) the radius-2 variant is not launched either: that one IS an assumption
    // (max|d_k| < 1, the radius-1 gather has no fallback), so it goes into the verdict the device evaluates after the forward
    // pass (scalar_kernels.h: Verdict): if it does not hold the transition is a no-op and is re-run (irs_transition below).
    bool skip_any[32], skip_r2[32];
    note_hint_trend(c);
    Verdict vd = no_verdict();
    vd.bounds = c->dmax;
    vd.n = cfg.no_steps;
    vd.C = C;
    for (int k = 0; k < cfg.no_steps && k < 32; ++k) {
        skip_any[k] = predicted_below(c, k, global_knobs().lds_from <= 2 ? 0.75f : 1.5f);
        skip_r2[k] = skip_any[k] && predicted_tiny(c, k);
        if (skip_r2[k]) vd.need_lt1 |= 1u << k;
        // ... and with the any-radius kernel left out a step must stay within the radius-2 gather's ring: its generic fallback beyond
        // it is correct, but sums in another order than the kernel a chain that launches every variant uses there -- and WHICH of the
        // two ran would depend on how old the bounds were that the host happened to see.  Part of the verdict instead: the chain is
        // the same chain, bit for bit, whatever the host guessed (tests/test_gpu_recovery_fuzz.py found the difference).
        // (lds_from 2: the any-radius kernel owns everything beyond ONE voxel when it is launched, so leaving it out assumes that)
        else if (skip_any[k]) (global_knobs().lds_from > 2 ? vd.need_lt2 : vd.need_lt1) |= 1u << k;
    }

    if (timed) HIP_TRY(hipEventRecord(c->ev[0], st));
    // fused backward warp: the forward warp also writes d(warped)/d(d_last) into gA, and the first adjoint squaring step
    // multiplies it with g_warped while staging (kernels.h: gscale)
    const bool fuse_warp_bwd = c->kn.fuse_warp_bwd != 0;
    if (forward_pass(c, io, io->v, true, cfg.uniform_alpha > 0.0f, vs, warped, z, fuse_warp_bwd ? c->gA : nullptr, C, st, timed)) return 1;
    const int64_t field = (int64_t)C * 3 * vol.V;
    const float* d_last = c->steps + (int64_t)(cfg.no_steps - 1) * field;
    if (io->transformation || io->displacement) launch_svf_outputs(d_last, io->transformation, io->displacement, C, vol, lin, st);

    // regulariser energy -> loss terms, coefficients, hyper-parameter step.  For the L2 family the coefficient (w / 2) does not
    // depend on the energy, so the update kernel produces the energy as a by-product of its stencil and the scalar stage runs
    // after it, inside the finalize launch (same values in, same order of the hyper-parameter step: the update still sees the
    // w of this transition)
    const int upd_blocks = sgld_update_blocks_per_chain(volv, C);
    const bool energy_in_update = (cfg.reg_loss == IRS_REG_L2 || cfg.reg_loss == IRS_REG_LOGNORMAL_L2) &&
                                  (int64_t)upd_blocks * C <= (int64_t)kMaxPartialBlocks * IRS_MAX_CHAINS &&
                                  c->kn.energy_in_update != 0;
    if (!energy_in_update) {
        launch_reg_energy(vs, c->energy_partials, C, volv, st);
        launch_reg_scalar(c->state, c->energy_partials, energy_blocks(volv), c->dcfg, st, vd);
    }

    // per chain, serially (trainer.py:316-327): VD factor -> GMM step -> data term with the UPDATED mixture.  The serial part is the
    // mixture: the statistics of chain c + 1 need the parameters chain c's step left, and that step must not touch them while the
    // data term of chain c still reads them -- but the data term of chain c (a 36 us launch of 1024 workgroups at 128^3, alone on
    // the chip) and the statistics of chain c + 1 (24 us) only READ the same parameters: with `chain_overlap` the former runs on a
    // side stream, the next chain's scalar stage waits for it.  Same kernels, same inputs, same order of every sum: chains
    // bit-identical (tests/test_gpu_transition.py).  Measured SLOWER than the serial form (the two half-filled launches get in each
    // other's way and every chain pays two event hand-overs; profiles/r05_chain_overlap_ab.txt): off by default.
    //
    // `data_batch` (default): what the data term of chain c needs from the mixture are 2K derived constants -- chain c's scalar stage
    // leaves a snapshot of them (DevState::snapA), the serial loop is then statistics -> step only, and the data terms of ALL chains
    // run as one launch behind it (grid.z = segments x C: one launch that fills the chip instead of C half-filled ones, C - 1 launch
    // gaps less).  Same kernel arithmetic on the same values, same partial-sum slots: chains bit-identical to the serial form.
    const int sb = stats_blocks(vol);
    const bool overlap = C > 1 && c->side && c->kn.chain_overlap != 0;
    const bool batch = C > 1 && !overlap && cfg.data_loss == IRS_DATA_GMM_LCC && c->kn.data_batch != 0;
    for (int ch = 0; ch < C; ++ch) {
        const uint8_t* mask = io->mask + (io->mask_chains == 1 ? 0 : (int64_t)ch * vol.V);
        const float* zc = z + (int64_t)ch * vol.V;
        launch_stats(cfg.virtual_decimation, zc, mask, c->state, c->stat_partials, vol, st, c->dcfg.K);
        if (overlap && ch > 0) HIP_TRY(hipStreamWaitEvent(st, c->ev_side[2 * (ch - 1) + 1], 0));  // data term of chain ch - 1 has read the mixture
        launch_chain_scalar(c->state, c->stat_partials, sb, ch, (ch == 0 ? 7 : 3) | (batch ? 8 : 0), c->dcfg, st, vd);  // chain 0: + the verdict
        if (batch) continue;
        const float* f = cfg.data_loss == IRS_DATA_GMM_LCC ? c->fhat + (c->fhat_chains == 1 ? 0 : (int64_t)ch * vol.V) : nullptr;
        hipStream_t ds = st;
        if (overlap && ch + 1 < C) {  // (the last chain's data term has nothing to overlap with: it stays on the caller's stream)
            HIP_TRY(hipEventRecord(c->ev_side[2 * ch], st));
            HIP_TRY(hipStreamWaitEvent(c->side, c->ev_side[2 * ch], 0));
            ds = c->side;
        }
        launch_data_bwd(cfg.data_loss, f, 0, zc, c->sigM + (int64_t)ch * vol.V, mask, 0, nullptr, c->state, ch,
                        c->gM + (int64_t)ch * vol.V, c->nll_partials + (int64_t)ch * c->nll_blocks, cfg.lcc_s, 1, vol, ds, c->nll_seg_C);
        if (ds != st) HIP_TRY(hipEventRecord(c->ev_side[2 * ch + 1], c->side));
    }
    if (batch)
        launch_data_bwd(cfg.data_loss, c->fhat, c->fhat_chains == 1 ? 0 : vol.V, z, c->sigM, io->mask, io->mask_chains == 1 ? 0 : vol.V, nullptr,
                        c->state, 0, c->gM, c->nll_partials, cfg.lcc_s, C, vol, st, c->nll_seg_C);
    // back through the warp and the squaring steps
    if (!fuse_warp_bwd)
        launch_warp_bwd(io->moving_im, io->moving_chains == 1 ? 0 : vol.V, d_last, io->unif,
                        cfg.uniform_alpha > 0.0f ? cfg.uniform_alpha : 0.0f, c->gM, c->gA, C, vol, lin, cfg.seed, 0, it, st);
    LAUNCH_CHECK();
    if (timed) HIP_TRY(hipEventRecord(c->ev[3], st));
    const float* dense = c->ffd ? c->dense : vs;
    float* g0 = nullptr;
    {
        // the adjoint ping-pongs between two buffers; the incoming gradient sits in gA, so start writing into gB
        const float* G = c->gA;
        float* bufs[2] = {c->gB, c->gA};
        int cur = 0;
        for (int k = cfg.no_steps - 1; k >= 0; --k) {
            float* out = bufs[cur];
            const float* dk = k == 0 ? dense : c->steps + (int64_t)(k - 1) * field;
            if (timed) HIP_TRY(hipEventRecord(c->ev_bwd[2 * k], st));
            const unsigned* dm = c->dmax + (int64_t)k * C * 4;
            // fused backward warp: gA holds d(warped)/d(d_last); the first step scales it by g_warped while staging
            const float* gscale = fuse_warp_bwd && k == cfg.no_steps - 1 ? c->gM : nullptr;
            // with the fused backward warp the first step's incoming gradient is the interleaved d(warped)/d(d_n)
            const int lay = bwd_lay(c, k) | (gscale ? 2 : 0);
            const bool sa = k < 32 && skip_any[k], s2 = k < 32 && skip_r2[k];
            // timed mode: the end event of step k closes right after the radius-1 kernel, so that exp_bwd_kernel_ms is the time
            // of the dominant kernel alone (as rocprofv3 reports it), not of the idle variants after it
            // lds_from 2: the any-radius kernel, when launched, takes every step beyond the radius-1 gather (no radius-2 gather then)
            const int gr = global_knobs().lds_from <= 2 ? 1 : 2;
            launch_exp_step_bwd_march(G, dk, out, k == 0, cfg.no_steps, C, vol, lin, dm, (s2 || (gr == 1 && !sa)) ? 1 : 2, sa, gscale, lay,
                                      timed ? c->ev_bwd[2 * k + 1] : nullptr, st);
            if (!sa) launch_exp_step_bwd_lds(G, dk, out, k == 0, cfg.no_steps, C, vol, lin, dm, 2, gr, gscale, lay, c->cmm, st);
            G = out;
            cur ^= 1;
        }
        g0 = const_cast<float*>(G);
    }
    if (timed) HIP_TRY(hipEventRecord(c->ev[4], st));
    float s[3];
    prescale_factors(vol, cfg.no_steps, s);
    if (c->ffd) {
        float* scaled = g0 == c->gA ? c->gB : c->gA;
        launch_scale_channels(g0, scaled, s[0], s[1], s[2], C, vol, st);
        const int G[3] = {volv.D, volv.H, volv.W};
        ffd_adjoint(scaled, c->tmpB, c->tmpA, C, vol, G, c->spl, st);
        launch_sgld_update(io->v, io->sigma, c->tmpB, vs, c->state, cfg.lr, 1.0f, 1.0f, 1.0f, io->grad_v, C, volv, st,
                           energy_in_update ? c->energy_partials : nullptr, energy_in_update);
    } else {
        launch_sgld_update(io->v, io->sigma, g0, vs, c->state, cfg.lr, s[0], s[1], s[2], io->grad_v, C, volv, st,
                           energy_in_update ? c->energy_partials : nullptr, energy_in_update);
    }
    // bookkeeping (+ the regulariser scalar stage of the L2 family, whose energy the update has just produced: one launch)
    launch_finalize(c->state, c->nll_partials, c->nll_blocks, c->dcfg, true, c->dmax, c->hint, 4 * C * (cfg.no_steps + 1), vd,
                    kHintWords - 7, true, st, energy_in_update ? c->energy_partials : nullptr, upd_blocks);
    c->dmax_clean = true;
    LAUNCH_CHECK();
    if (timed) HIP_TRY(hipEventRecord(c->ev[5], st));
    {
        // (under stream capture nothing has been enqueued on the device -- the launch sequence became graph nodes, and an event
        // recorded inside a capture cannot be waited for by the host: the run-ahead bookkeeping counts executed transitions only)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        if (cap != hipStreamCaptureStatusNone) return 0;
    }
    HIP_TRY(hipEventRecord(c->ra_ev[c->n_enqueued % 4], st));
    ++c->n_enqueued;
    return 0;
}

// failed (no-op) transitions the device has reported since the host last looked -> transitions to re-run
static void poll_failures(irs_ctx* c) {
    if (!c->hint) return;
    const unsigned f = ((volatile unsigned*)c->hint)[kHintWords - 7];
    if (f == c->fails_seen || (int)(f - c->fails_seen) < 0) return;  // (a count that went backwards is not 4e9 failures)
    c->makeup += (uint64_t)(f - c->fails_seen);
    c->fails_total += (uint64_t)(f - c->fails_seen);
    c->fails_seen = f;
    // the bounds that misled the prediction are still the ones the host sees: no assumptions for the next few transitions
    c->force_all_until = c->n_enqueued + c->makeup + 3;
}

static void drop_pending(irs_ctx* c) {
    poll_failures(c);
    c->makeup = 0;
}

static int transition_impl(irs_ctx* c, const irs_io* io, hipStream_t st, int timed) {
    if (check_io(c, io, "irs_transition")) return 1;
    if (!io->v) return fail("irs_transition: v is required");
    {
        // Under stream capture the launch sequence becomes a graph that is replayed without this host code: no prediction may be
        // baked into it (a replay whose verdict failed would stay a no-op on every replay) and no pending re-run belongs in it.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        if (cap != hipStreamCaptureStatusNone) {
            // (timing events are read back by the host right after the call, which a captured stream never executed: refused.
            // The io of this call is still "the last one": irs_flush re-runs with it.)
            if (timed) return fail("irs_transition_timed: the stream is being captured -- per-stage timings need a stream that executes");
            c->last_io = *io;
            c->have_last_io = true;
            return enqueue_transition(c, io, st, 0, true);
        }
    }
    // Bounded run-ahead: the host may be at most IRS_RUN_AHEAD (default 2) transitions ahead of the device.  The variant
    // prediction reads bounds the device published at the end of an earlier transition; a host that has queued twenty
    // transitions would predict from a state twenty transitions old, and while the displacement is still growing (burn-in)
    // that mispredicts into the slow always-correct fallbacks.  Two queued transitions keep the device busy all the same.
    {
        const int depth = c->kn.run_ahead;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        if (depth > 0 && depth <= 3 && cap == hipStreamCaptureStatusNone && c->n_enqueued >= (uint64_t)depth)
            HIP_TRY(hipEventSynchronize(c->ra_ev[(c->n_enqueued - depth) % 4]));
    }
    // A transition whose assumptions about max|d_k| failed was a no-op on the device (nothing changed, the Philox counter did
    // not advance): it is re-run here, without assumptions, before the transition of this call -- the chain continues as if
    // every variant had been launched all along (same noise, same order; with injected eps / unif the re-run uses THIS call's).
    poll_failures(c);
    if (c->makeup && !c->kn.recover)
        return fail("irs_transition: an earlier transition skipped a kernel variant its displacement then needed and was dropped "
                    "(recover = 0); predict_variants = 0 launches every variant");
    while (c->makeup > 0) {
        --c->makeup;
        if (enqueue_transition(c, io, st, 0, true)) return 1;
    }
    c->last_io = *io;
    c->have_last_io = true;
    return enqueue_transition(c, io, st, timed, c->n_enqueued < c->force_all_until);
}

// wait for everything enqueued, then re-run what failed (with the io of the last call); afterwards state, scalars and v are final
static int flush_impl(irs_ctx* c, hipStream_t st) {
    for (int guard = 0; guard < 8; ++guard) {
        HIP_TRY(hipStreamSynchronize(st));
        if (c->n_enqueued) HIP_TRY(hipEventSynchronize(c->ra_ev[(c->n_enqueued - 1) % 4]));
        poll_failures(c);
        if (!c->makeup) return 0;
        if (!c->kn.recover || !c->have_last_io) return fail("irs_flush: %llu transition(s) were dropped after a failed variant prediction", (unsigned long long)c->makeup);
        while (c->makeup > 0) {
            --c->makeup;
            if (enqueue_transition(c, &c->last_io, st, 0, true)) return 1;
        }
    }
    return fail("irs_flush: transitions keep failing");
}

int irs_flush(irs_ctx* c, void* stream) {
    if (!c) return fail("irs_flush: null argument");
    if (c->sl.on) return irs::slab_flush(c, (hipStream_t)stream);
    return flush_impl(c, (hipStream_t)stream);
}

int irs_recovered_transitions(const irs_ctx* c, uint64_t* out) {
    if (!c || !out) return fail("irs_recovered_transitions: null argument");
    *out = c->fails_total;
    return 0;
}

int irs_transition(irs_ctx* c, const irs_io* io, void* stream) { return transition_impl(c, io, (hipStream_t)stream, 0); }

int irs_transition_timed(irs_ctx* c, const irs_io* io, void* stream, irs_timings* out) {
    if (!out) return fail("irs_transition_timed: null output");
    if (c && c->cfg.no_steps > 32) return fail("irs_transition_timed: at most 32 steps");
    if (transition_impl(c, io, (hipStream_t)stream, 1)) return 1;
    HIP_TRY(hipEventSynchronize(c->ev[5]));
    memset(out, 0, sizeof(*out));
    HIP_TRY(hipEventElapsedTime(&out->total_ms, c->ev[0], c->ev[5]));
    HIP_TRY(hipEventElapsedTime(&out->smooth_ms, c->ev[0], c->ev[1]));
    HIP_TRY(hipEventElapsedTime(&out->exp_fwd_ms, c->ev[1], c->ev[2]));
    HIP_TRY(hipEventElapsedTime(&out->data_ms, c->ev[2], c->ev[3]));
    HIP_TRY(hipEventElapsedTime(&out->exp_bwd_total_ms, c->ev[3], c->ev[4]));
    HIP_TRY(hipEventElapsedTime(&out->update_ms, c->ev[4], c->ev[5]));
    for (int k = 0; k < c->cfg.no_steps; ++k) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_bwd[2 * k], c->ev_bwd[2 * k + 1]));
        out->exp_bwd_kernel_ms += ms;
        if (k >= 1) out->exp_bwd_primary_avg_ms += ms / (float)(c->cfg.no_steps > 1 ? c->cfg.no_steps - 1 : 1);
    }
    return 0;
}

}  // extern "C"
