// C ABI of the MI355X-native VISTAF FTP path (see include/vistaf_ftp.h).  Host orchestration only:
// every image operation runs in a HIP kernel of this library; the host builds constant tables
// (Gaussian taps, Hann window, DFT twiddles, ROI / apodisation planes) in double precision.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/vistaf_ftp.h"
#include "kernels.hpp"

using namespace vf;
#include "test_hooks.h"
#ifdef VISTAF_DEBUG
namespace vf { void telea_debug_dump(); void telea_window_debug_dump(int B); void telea_window_mw_debug_dump(int B); void fit_debug_dump(); void unwrap_big_debug_dump(); void unwrap_batch_debug_dump(); }
#endif

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
namespace vf { int set_error(int code, const std::string &msg) { return fail(code, msg); } }   // for the other translation units of the ABI
#define HIPCHK(x)                                                                                         \
    do {                                                                                                  \
        hipError_t e_ = (x);                                                                              \
        if (e_ != hipSuccess) return fail(VISTAF_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace {

struct GKern { float *d = nullptr; int k = 0; };

enum Stage { ST_GRAY_BAD = 0, ST_INPAINT, ST_PREPROC, ST_DEMOD, ST_RELIABLE, ST_UNWRAP_RANK, ST_UNWRAP, ST_UNWRAP_TREE, ST_DETREND, ST_SMOOTH_FLIP, ST_COMPOSE, ST_MM_BLOB,
             ST_TAIL, ST_COUNT };
const char *kStageNames[ST_COUNT] = {"gray+badpix", "inpaint (k_telea_window_mw)", "illum+blur+apod+median", "pruned-dft demod", "reliable mask",
                                     "unwrap check (k_unwrap_fast)", "unwrap flood (fallback)", "unwrap tree (fallback)", "detrend (3x IRLS)", "smooth+flip", "frontier+compose", "mm+blob filter", "tail"};

int cv_round(double v) { return (int)std::nearbyint(v); }

}  // namespace

struct vistaf_ftp_handle {
    vistaf_ftp_config cfg;
    int h = 0, w = 0, P = 0, cx = 0, cy = 0, r = 0, maxB = 0;
    Curve hcurve, fcurve;
    int use_neg = 1;
    bool have_ref = false;

    // static planes
    uint8_t *roi = nullptr, *valid = nullptr;
    float *apo = nullptr, *roi_den = nullptr;
    GKern g_illum, g_pre, g_qual, g_rel, g_unrel;
    RowSpanSE se_bad, se_close, se_contact;

    // reference state (session mode: one carrier, tables shared by the batch)
    double peak_x = 0, peak_y = 0, kx = 0, ky = 0, period = 0, mm_per_px = 0;
    int Hf = 0, Wf = 0, ph = 0, pw = 0, pmax = 0;
    double2 *Ex = nullptr, *Ey = nullptr, *Gx = nullptr, *Gy = nullptr, *cref = nullptr;
    float *win = nullptr, *amp_ref = nullptr;
    CarrierGeom *geom = nullptr;                      // [1] session carrier, device
    // carrier search (full spectrum), allocated on first use for `search_cap` frames
    double2 *Exf = nullptr, *Eyf = nullptr, *search_tmp = nullptr;
    double *search_mag = nullptr, *search_peaks = nullptr;
    int search_cap = 0;
    // uncached-pair mode (vistaf_ftp_predict_pairs): per-frame carriers, tables and reference fields, allocated on first use
    CarrierGeom *pgeom = nullptr;
    double2 *pEx = nullptr, *pEy = nullptr, *pGx = nullptr, *pGy = nullptr, *pcref = nullptr;
    float *pamp_ref = nullptr, *win_full = nullptr;
    bool pairs_ready = false;
    Tiers tiers;                                       // kernel tier selection (test hook; defaults = production kernels)
    bool keep_planes = false;                          // test hook: also write planes that only the parity tests read (float64 field)

    // workspace (maxB frames)
    std::map<std::string, std::pair<void *, size_t>> named;   // name -> (ptr, bytes per frame)
    std::vector<void *> allocs;
    float *img, *grad, *tmpf, *blurA, *inorm, *iw, *amp, *prod, *quality, *wrapped, *unwrapped, *phase1, *resid0, *detr, *z0, *mplane,
        *num, *den, *hmap, *dist, *z0f, *snum, *unitless, *depth;
    double2 *field, *patch;
    double2 *tmpT;
    uint8_t *bad0, *bad1, *rel0, *rel1, *rel2, *reliable, *contact, *contact_d, *background, *cand, *kept;
    uint8_t *out_rel = nullptr, *hole_cand = nullptr;      // hole stage only (reliable_smooth_sigma_px == 0)
    float *hole_med = nullptr, *hole_fill = nullptr;
    int32_t *labels, *area, *rowdist, *parent;
    unsigned int *peak_bits;
    uint16_t *morph_pre;
    void *inpaint_scratch, *inpaint_cl_scratch, *inpaint_win_scratch, *unwrap_scratch;
    void *big_scratch = nullptr;      // k_big.hip (frames of 512 x 512 and more), else null
    int32_t *unwrap_need;       // [max_batch] 1: the frame went through the priority flood, 0: the consistency check settled it
    // small per-frame arrays
    float *thr_hi, *thr_g, *mu, *amp_thr, *thr3, *thr_used, *bg_med, *core_thr, *core_med, *coef;
    int *cnt_a, *cnt_valid, *rel_count, *contact_count, *bg_count, *bad_count, *flipped;
    unsigned int *gmax;
    int32_t *status;
    unsigned long long *cc_best;        // [maxB] largest-component key of launch_cc_largest on large frames
    double *scalars;
    float *req_hi, *req_g, *req_med, *req_amp, *req_contact, *req_core;   // device percentile requests

    bool timing = false;
    hipEvent_t ev[ST_COUNT + 1];
    bool ev_made = false;
    float stage_ms[ST_COUNT];
};

namespace {

template <typename T>
int dalloc(vistaf_ftp_handle *hd, T **p, size_t count, const char *name = nullptr, size_t per_frame = 0)
{
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(T) + 256);
    if (e != hipSuccess) return fail(VISTAF_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    hd->allocs.push_back(q);
    *p = (T *)q;
    if (name) hd->named[name] = {q, per_frame};
    return 0;
}

int make_gkern(vistaf_ftp_handle *hd, double sigma, GKern *g)
{
    g->k = 0; g->d = nullptr;
    if (!(sigma > 0)) return 0;
    int n = cv_round(sigma * 4 * 2 + 1) | 1;       // cv::GaussianBlur ksize rule, CV_32F
    if (n > 511) return fail(VISTAF_E_INVALID, "gaussian sigma too large (ksize > 511)");
    std::vector<double> t(n);
    double s2 = -0.5 / (sigma * sigma), sum = 0;
    for (int i = 0; i < n; i++) { double x = i - (n - 1) * 0.5; t[i] = std::exp(s2 * x * x); sum += t[i]; }
    std::vector<float> f(n);
    for (int i = 0; i < n; i++) f[i] = (float)(t[i] * (1.0 / sum));
    int rc = dalloc(hd, &g->d, n);
    if (rc) return rc;
    if (hipMemcpy(g->d, f.data(), n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail(VISTAF_E_HIP, "memcpy gkern");
    g->k = n;
    return 0;
}

// Row spans of the set { (dx, dy) : 3x3 chamfer metric <= margin } (integer metric of cv::distanceTransform DIST_L2 3x3: 0.955 -> 62587,
// 1.3693 -> 89738 in 1/65536 units; dist = c / 65536 as float, exact, so "dist <= margin" is "c <= margin * 65536").  false: too large.
bool chamfer_ball(float margin, RowSpanSE *se)
{
    const long long HV = 62587, DG = 89738;
    const long long M = (long long)std::floor((double)margin * 65536.0);
    if (M < 0) return false;
    const int R = (int)(M / HV);
    if (2 * R + 1 > 33 || R > 63) return false;
    se->k = std::max(3, 2 * R + 1);
    const int r = se->k / 2;
    for (int i = 0; i < se->k; i++) {
        const int dy = std::abs(i - r);
        int a = -1;
        for (int dx = 0; dx <= 63; dx++) {
            const long long mn = std::min(dx, dy), mx = std::max(dx, dy);
            if (HV * (mx - mn) + DG * mn <= M) a = dx; else break;
        }
        if (a < 0) { se->lo[i] = 1; se->hi[i] = -1; }       // empty row
        else { se->lo[i] = (int8_t)-a; se->hi[i] = (int8_t)a; }
    }
    return true;
}

// cv2.getStructuringElement(MORPH_ELLIPSE, (k, k)), anchor at (k/2, k/2).  force_odd: the callers that first do `max(3, k | 1)`
// (shape_ftp.py:641-644, :756-758); the contact dilation passes DILATE_KERNEL_SIZE as it is (:1734), even sizes included.
int make_se(int k, RowSpanSE *se, bool force_odd = true)
{
    if (force_odd) k = std::max(3, k | 1);
    if (k > 33) return fail(VISTAF_E_INVALID, "structuring element larger than 33");
    se->k = k;
    int r = k / 2, c = k / 2;
    double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < k; i++) {
        int dy = i - r;
        int dx = cv_round(c * std::sqrt((r * r - dy * dy) * inv_r2));
        int j1 = std::max(c - dx, 0), j2 = std::min(c + dx + 1, k);
        se->lo[i] = (int8_t)(j1 - c);
        se->hi[i] = (int8_t)(j2 - 1 - c);
    }
    return 0;
}

void blur(vistaf_ftp_handle *hd, const float *src, float *dst, const GKern &g, int B, hipStream_t st)
{
    launch_gauss_blur(src, hd->tmpf, dst, g.d, g.k, B, hd->h, hd->w, st);
}

float q32_of(double pct) { return (float)pct / 100.0f; }   // np.true_divide(q, float32(100))

int upload_req(vistaf_ftp_handle *hd, float **d, const std::vector<float> &v)
{
    int rc = dalloc(hd, d, v.size());
    if (rc) return rc;
    if (hipMemcpy(*d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail(VISTAF_E_HIP, "memcpy req");
    return 0;
}

// preprocessing shared by reference and deformed frames: frames -> iw (apodised, normalised) and mu
// frames2 (pair mode, when the workspace holds 2 * nframes): a second set of nframes frames preprocessed in the SAME launches, as frames
// [nframes, 2 * nframes) of every plane -- the one-wave-per-frame march then runs once over twice as many CUs instead of twice
void preprocess(vistaf_ftp_handle *hd, const void *frames, int format, int nframes, hipStream_t st, bool timed, const void *frames2 = nullptr)
{
    const vistaf_ftp_config &c = hd->cfg;
    int h = hd->h, w = hd->w, P = hd->P;
    if (timed) hipEventRecord(hd->ev[ST_GRAY_BAD], st);
    launch_to_gray(frames, format, hd->img, nframes, P, st);
    if (frames2) launch_to_gray(frames2, format, hd->img + (size_t)nframes * P, nframes, P, st);
    const int B = frames2 ? 2 * nframes : nframes;
    hipMemsetAsync(hd->bad_count, 0, sizeof(int) * B, st);
    if (c.bad_pixel_enable) {
        launch_sobel_mag(hd->img, hd->grad, B, h, w, st);
        launch_select(hd->img, hd->valid, 0, nullptr, false, hd->req_hi, 1, hd->thr_hi, hd->cnt_valid, B, P, st, hd->tiers.big_chain ? hd->big_scratch : nullptr);
        launch_select(hd->grad, hd->valid, 0, nullptr, false, hd->req_g, 1, hd->thr_g, nullptr, B, P, st, hd->tiers.big_chain ? hd->big_scratch : nullptr);
        launch_bad_flags(hd->img, hd->grad, hd->valid, hd->thr_hi, hd->thr_g, hd->bad0, B, P, st);
        uint8_t *src = hd->bad0, *dst = hd->bad1;
        if (c.bad_dilate_ksize > 1)
            for (int it = 0; it < c.bad_dilate_iters; it++) { launch_morph(src, dst, B, h, w, hd->se_bad, true, nullptr, nullptr, st); std::swap(src, dst); }
        if (src != hd->bad1) hipMemcpyAsync(hd->bad1, src, (size_t)B * P, hipMemcpyDeviceToDevice, st);
        launch_count_u8(hd->bad1, hd->bad_count, B, P, st);
        {
            int range = std::min(100, std::max(1, cv_round((double)c.bad_inpaint_radius)));   // cv::inpaint clamps the radius
            const uint8_t *seq_mask = hd->bad1;
            const int32_t *only = nullptr;
            // default (Tiers::inpaint 2): the frame-window kernel (LDS-resident march), then the whole-frame kernel for the frames it hands
            // back; 1: whole-frame kernel only; 0: cluster by cluster -- every independent cluster of hole pixels on its own wave, in an LDS
            // window when it fits, on the frame's global planes otherwise.  At 224 x 224 the hole pixels form ONE cluster per frame, so the
            // frame window is the better tool; frames much larger than its capacity (14 464 cells) -- the native 1182 x 1182 crops -- never
            // fit one window, and there the crests are ~60 px apart: many independent clusters (march of eight native crops: 1.95 s with the
            // whole-frame kernel, 0.54 s with LDS clusters + whole-frame kernel for the rest, 18 ms since k_inpaint_big.hip: DESIGN.md section 5a).
            int mode = hd->tiers.inpaint;
            if (mode == 2 && (size_t)h * w > (size_t)8 * 14464) mode = 0;
            if (mode == 0 && inpaint_clusters_supported(range)) {
                // every independent cluster of hole pixels on its own wave: LDS windows for those that fit, the frame's global planes for the rest
                uint8_t *bad_big = nullptr;
                ClusterPlanes left;
                launch_inpaint_clusters(hd->img, hd->bad1, range, hd->inpaint_cl_scratch, &bad_big, &left, B, h, w, st);
                if (timed) hipEventRecord(hd->ev[ST_INPAINT], st);      // (after the LDS cluster pass: its bookkeeping counts as mask work)
                launch_inpaint_big_clusters(hd->img, bad_big, range, hd->inpaint_scratch, hd->status, left, B, h, w, st, hd->tiers.big_queue_lds != 0);
            } else {
                if (timed && mode == 1) hipEventRecord(hd->ev[ST_INPAINT], st);
                if (mode != 1) only = launch_inpaint_window(hd->img, seq_mask, range, hd->inpaint_win_scratch, B, h, w, st, timed ? hd->ev[ST_INPAINT] : nullptr,
                                                               hd->tiers.telea_two_tier != 0, hd->tiers.telea_mw != 0);
                launch_inpaint_telea(hd->img, seq_mask, range, hd->inpaint_scratch, hd->status, only, B, h, w, st);
            }
        }
    } else if (timed) hipEventRecord(hd->ev[ST_INPAINT], st);
    if (timed) hipEventRecord(hd->ev[ST_PREPROC], st);
    blur(hd, hd->img, hd->blurA, hd->g_illum, B, st);
    launch_illum_norm(hd->img, hd->blurA, hd->inorm, B, P, st);
    const float *in = hd->inorm;
    if (hd->g_pre.k) { blur(hd, hd->inorm, hd->blurA, hd->g_pre, B, st); in = hd->blurA; }
    launch_mul_static(in, hd->apo, hd->iw, B, P, st);
    launch_select(hd->iw, hd->valid, 0, nullptr, false, hd->req_med, 1, hd->mu, nullptr, B, P, st, hd->tiers.big_chain ? hd->big_scratch : nullptr);
}

// np.hanning(ph)[:,None] * np.hanning(pw)[None,:] in float32 (shape_ftp.py:800-807); np.hanning(M) = 0.5 + 0.5*cos(pi*n/(M-1)), n = 1-M, 3-M, ...
std::vector<float> hann_patch(int ph, int pw)
{
    auto hann = [](int M, int i) -> float { return M == 1 ? 1.0f : (float)(0.5 + 0.5 * std::cos(3.14159265358979323846 * (double)(1 - M + 2 * i) / (double)(M - 1))); };
    std::vector<float> win((size_t)ph * pw);
    for (int a = 0; a < ph; a++)
        for (int c = 0; c < pw; c++) win[(size_t)a * pw + c] = hann(ph, a) * hann(pw, c);
    return win;
}

template <typename T>
int upload(vistaf_ftp_handle *hd, T **d, const std::vector<T> &v)
{
    int rc = dalloc(hd, d, v.size());
    if (rc) return rc;
    if (hipMemcpy(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return fail(VISTAF_E_HIP, "memcpy upload");
    return 0;
}

}  // namespace

extern "C" {

int vistaf_ftp_abi_version(void) { return VISTAF_FTP_ABI_VERSION; }
const char *vistaf_ftp_last_error(void) { return g_err.c_str(); }

int vistaf_ftp_default_config(vistaf_ftp_config *c)
{
    if (!c) return fail(VISTAF_E_INVALID, "null config");
    memset(c, 0, sizeof(*c));
    c->patch_half_width_bins = 10; c->dc_exclusion = 10; c->fft_pad_px = 96; c->roi_erode_px = 0; c->apod_taper_px = 120;
    c->reliable_edge_margin_px = 6; c->poly_order = 2; c->frontier_zero_band_px = 200; c->valid_close_kernel = 7;
    c->valid_close_iters = 1; c->bad_pixel_enable = 1; c->bad_dilate_ksize = 5; c->bad_dilate_iters = 1; c->bad_inpaint_radius = 3;
    c->dilate_kernel_size = 15; c->dilate_iters = 2; c->n_fft_peaks = 12; c->plane_order_for_removal = 1; c->irls_iters = 6;
    c->pre_blur_sigma_px = 1.5; c->amp_valid_percentile = 25.0; c->quality_smooth_sigma_px = 6.0; c->reliable_smooth_sigma_px = 2.5;
    c->illum_sigma_px = 45.0; c->bad_intensity_percentile = 99.9; c->bad_gradient_percentile = 99.7; c->contact_core_percentile = 8.0;
    c->contact_percentile = 92.0; c->min_contact_frac = 0.002; c->max_contact_frac = 0.40; c->unreliable_smooth_sigma_px = 9.0;
    c->contact_blob_min_peak_mm = 0.1; c->contact_blob_min_peak_rel_frac = 1.0 / 3.0; c->peak_max_dy_from_center = 0.12;
    c->irls_c = 4.685; c->grating_pitch_mm = 2.0; c->depth_eps_mm = 0.01;
    c->hole_neighborhood_px = 11; c->hole_min_dist_px = 4; c->inpaint_radius = 5; c->hole_known_fraction = 0.70;
    return 0;
}

void vistaf_ftp_destroy(vistaf_ftp_handle *hd)
{
    if (!hd) return;
    for (void *p : hd->allocs) hipFree(p);
    if (hd->ev_made) for (int i = 0; i <= ST_COUNT; i++) hipEventDestroy(hd->ev[i]);
    delete hd;
}

int vistaf_ftp_create(const vistaf_ftp_config *cfg, int h, int w, int cx, int cy, int r, int max_batch,
                      const vistaf_curve *height_curve, int use_negated_height, const vistaf_curve *force_curve,
                      vistaf_ftp_handle **out)
{
    if (!cfg || !out || !height_curve || !force_curve) return fail(VISTAF_E_INVALID, "null argument");
    if (h < 8 || w < 8 || max_batch < 1 || r < 1) return fail(VISTAF_E_INVALID, "bad geometry");
    if (height_curve->type < 0 || height_curve->type > 5 || force_curve->type < 0 || force_curve->type > 5)
        return fail(VISTAF_E_INVALID, "Unknown model type in calibration");
    if (cfg->poly_order < 1 || cfg->poly_order > 2 || cfg->plane_order_for_removal < 0 || cfg->plane_order_for_removal > 2)
        return fail(VISTAF_E_INVALID, "polynomial order must be 1 or 2 (plane_order_for_removal: 0 = no pre-removal)");
    if (cfg->patch_half_width_bins < 3 && cfg->patch_half_width_bins != 0) {}
    int bwp = std::max(3, cfg->patch_half_width_bins);
    if (2 * bwp + 1 > 255) return fail(VISTAF_E_INVALID, "patch too wide");
    vistaf_ftp_handle *hd = new vistaf_ftp_handle();
    hd->cfg = *cfg; hd->h = h; hd->w = w; hd->P = h * w; hd->cx = cx; hd->cy = cy; hd->r = r; hd->maxB = max_batch;
    hd->hcurve = Curve{height_curve->type, height_curve->a, height_curve->b, height_curve->c};
    hd->fcurve = Curve{force_curve->type, force_curve->a, force_curve->b, force_curve->c};
    hd->use_neg = use_negated_height ? 1 : 0;
    int P = hd->P;
    size_t n = (size_t)max_batch * P;
    int rc = 0;
#define TRY(x) do { rc = (x); if (rc) { vistaf_ftp_destroy(hd); return rc; } } while (0)
    // static planes (shape_ftp.py:383-403, :1519-1528)
    std::vector<uint8_t> roi(P), valid(P);
    std::vector<float> apo(P), roif(P);
    int r_valid = std::max(0, r - cfg->roi_erode_px);
    double r_in = std::max(0.0, (double)(r - cfg->apod_taper_px));
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            long d2 = (long)(x - cx) * (x - cx) + (long)(y - cy) * (y - cy);
            roi[(size_t)y * w + x] = d2 <= (long)r_valid * r_valid;
            double d = std::sqrt((double)d2);
            float a = 0.f;
            if (d <= r_in) a = 1.0f;
            else if (d <= r && cfg->apod_taper_px > 0) {
                double t = (d - r_in) / std::max(1e-6, (double)cfg->apod_taper_px);
                a = (float)(0.5 * (1.0 + std::cos(3.14159265358979323846 * t)));
            }
            apo[(size_t)y * w + x] = a;
            valid[(size_t)y * w + x] = a > 1e-6f;
            roif[(size_t)y * w + x] = roi[(size_t)y * w + x] ? 1.f : 0.f;
        }
    TRY(upload(hd, &hd->roi, roi)); TRY(upload(hd, &hd->valid, valid)); TRY(upload(hd, &hd->apo, apo));
    hd->named["roi"] = {hd->roi, 0};
    TRY(make_gkern(hd, cfg->illum_sigma_px, &hd->g_illum));
    if (!hd->g_illum.k) { vistaf_ftp_destroy(hd); return fail(VISTAF_E_INVALID, "illum_sigma_px must be > 0"); }
    TRY(make_gkern(hd, cfg->pre_blur_sigma_px, &hd->g_pre));
    TRY(make_gkern(hd, cfg->quality_smooth_sigma_px, &hd->g_qual));
    TRY(make_gkern(hd, cfg->reliable_smooth_sigma_px, &hd->g_rel));
    TRY(make_gkern(hd, cfg->unreliable_smooth_sigma_px, &hd->g_unrel));
    TRY(make_se(cfg->bad_dilate_ksize, &hd->se_bad));
    TRY(make_se(cfg->valid_close_kernel, &hd->se_close));
    {   // cv2.getStructuringElement(ELLIPSE, (k,k)) with k as given (shape_ftp.py:1734)
        int k = cfg->dilate_kernel_size;
        if (k < 1 || k > 33) { vistaf_ftp_destroy(hd); return fail(VISTAF_E_INVALID, "dilate_kernel_size out of range"); }
        TRY(make_se(k, &hd->se_contact, false));
    }
    if (!(cfg->reliable_smooth_sigma_px > 0)) {     // the hole stage is live (shape_ftp.py:1770-1801)
        if (cfg->hole_neighborhood_px < 1 || cfg->hole_neighborhood_px > 255 || cfg->hole_min_dist_px < 0 || cfg->inpaint_radius < 1 ||
            !(cfg->hole_known_fraction >= 0.0)) {
            vistaf_ftp_destroy(hd);
            return fail(VISTAF_E_INVALID, "hole-stage constants out of range");
        }
    }
    // workspace
#define PLANE(T, name) TRY(dalloc(hd, &hd->name, n, #name, (size_t)P * sizeof(T)))
    PLANE(float, img); PLANE(float, grad); PLANE(float, tmpf); PLANE(float, blurA); PLANE(float, inorm); PLANE(float, iw);
    PLANE(float, amp); PLANE(float, prod); PLANE(float, quality); PLANE(float, wrapped); PLANE(float, unwrapped); PLANE(float, phase1);
    PLANE(float, resid0); PLANE(float, detr); PLANE(float, z0); PLANE(float, mplane); PLANE(float, num); PLANE(float, den);
    PLANE(float, hmap); PLANE(float, dist); PLANE(float, z0f); PLANE(float, snum); PLANE(float, unitless); PLANE(float, depth);
    PLANE(double2, field);
    PLANE(uint8_t, bad0); PLANE(uint8_t, bad1); PLANE(uint8_t, rel0); PLANE(uint8_t, rel1); PLANE(uint8_t, rel2); PLANE(uint8_t, reliable);
    PLANE(uint8_t, contact); PLANE(uint8_t, contact_d); PLANE(uint8_t, background); PLANE(uint8_t, cand); PLANE(uint8_t, kept);
    if (!(cfg->reliable_smooth_sigma_px > 0)) { PLANE(uint8_t, out_rel); PLANE(uint8_t, hole_cand); }
    PLANE(int32_t, labels); PLANE(int32_t, area); PLANE(int32_t, rowdist); PLANE(int32_t, parent);
    PLANE(unsigned int, peak_bits);
    PLANE(uint16_t, morph_pre);
#undef PLANE
    {
        void *p = nullptr;
        TRY(dalloc(hd, (uint8_t **)&p, inpaint_scratch_bytes_per_frame(h, w) * max_batch)); hd->inpaint_scratch = p;
        TRY(dalloc(hd, (uint8_t **)&p, inpaint_cl_scratch_bytes_per_frame(h, w) * max_batch + 2048)); hd->inpaint_cl_scratch = p;
        TRY(dalloc(hd, (uint8_t **)&p, inpaint_win_scratch_bytes(max_batch))); hd->inpaint_win_scratch = p;
        TRY(dalloc(hd, (uint8_t **)&p, unwrap_scratch_bytes_per_frame(h, w) * max_batch + 1024)); hd->unwrap_scratch = p;
        if ((size_t)h * w >= 262144) { TRY(dalloc(hd, (uint8_t **)&p, big_scratch_bytes(max_batch, h, w))); hd->big_scratch = p; }
        TRY(dalloc(hd, &hd->unwrap_need, (size_t)max_batch, "unwrap_need", sizeof(int32_t)));
        HIPCHK(hipMemset(hd->unwrap_need, 0xff, (size_t)max_batch * sizeof(int32_t)));        // -1: never written (the check is off or does not cover this frame size)
    }
    int pmax = 2 * bwp + 1;
    hd->pmax = pmax;
    TRY(dalloc(hd, &hd->patch, (size_t)max_batch * pmax * pmax, "patch", (size_t)pmax * pmax * sizeof(double2)));
    TRY(dalloc(hd, &hd->tmpT, (size_t)max_batch * (size_t)std::max(h, w) * pmax));
    TRY(dalloc(hd, &hd->Ex, (size_t)w * pmax)); TRY(dalloc(hd, &hd->Gx, (size_t)w * pmax));
    TRY(dalloc(hd, &hd->Ey, (size_t)h * pmax)); TRY(dalloc(hd, &hd->Gy, (size_t)h * pmax));
    TRY(dalloc(hd, &hd->win, (size_t)pmax * pmax)); TRY(dalloc(hd, &hd->geom, 1));
    TRY(dalloc(hd, &hd->cref, (size_t)P)); TRY(dalloc(hd, &hd->amp_ref, (size_t)P));
    hd->named["cref"] = {hd->cref, 0}; hd->named["amp_ref"] = {hd->amp_ref, 0};
    size_t mb = max_batch;
    TRY(dalloc(hd, &hd->thr_hi, mb)); TRY(dalloc(hd, &hd->thr_g, mb)); TRY(dalloc(hd, &hd->mu, mb)); TRY(dalloc(hd, &hd->amp_thr, mb));
    TRY(dalloc(hd, &hd->thr3, mb * 3)); TRY(dalloc(hd, &hd->thr_used, mb)); TRY(dalloc(hd, &hd->bg_med, mb));
    TRY(dalloc(hd, &hd->core_thr, mb)); TRY(dalloc(hd, &hd->core_med, mb)); TRY(dalloc(hd, &hd->coef, mb * 6));
    TRY(dalloc(hd, &hd->cnt_a, mb)); TRY(dalloc(hd, &hd->cnt_valid, mb)); TRY(dalloc(hd, &hd->rel_count, mb));
    TRY(dalloc(hd, &hd->contact_count, mb)); TRY(dalloc(hd, &hd->bg_count, mb)); TRY(dalloc(hd, &hd->bad_count, mb));
    TRY(dalloc(hd, &hd->flipped, mb)); TRY(dalloc(hd, &hd->gmax, mb)); TRY(dalloc(hd, &hd->status, mb)); TRY(dalloc(hd, &hd->cc_best, mb));
    TRY(dalloc(hd, &hd->scalars, mb * VISTAF_NSCALARS));
    TRY(dalloc(hd, &hd->hole_med, mb)); TRY(dalloc(hd, &hd->hole_fill, mb));
    hd->named["mu"] = {hd->mu, sizeof(float)}; hd->named["thr_hi"] = {hd->thr_hi, sizeof(float)}; hd->named["thr_g"] = {hd->thr_g, sizeof(float)};
    hd->named["coef"] = {hd->coef, 6 * sizeof(float)}; hd->named["thr3"] = {hd->thr3, 3 * sizeof(float)};
    hd->named["core_thr"] = {hd->core_thr, sizeof(float)}; hd->named["core_med"] = {hd->core_med, sizeof(float)};
    TRY(upload_req(hd, &hd->req_hi, {q32_of(cfg->bad_intensity_percentile)}));
    TRY(upload_req(hd, &hd->req_g, {q32_of(cfg->bad_gradient_percentile)}));
    TRY(upload_req(hd, &hd->req_med, {-1.0f}));
    TRY(upload_req(hd, &hd->req_amp, {q32_of(cfg->amp_valid_percentile)}));
    TRY(upload_req(hd, &hd->req_contact, {q32_of(cfg->contact_percentile), q32_of(95.0), q32_of(98.0)}));
    TRY(upload_req(hd, &hd->req_core, {q32_of(cfg->contact_core_percentile)}));
    // den of the ROI-wide masked smooth is frame independent: blur(roi) + 1e-6 (shape_ftp.py:1146, :1821)
    TRY(dalloc(hd, &hd->roi_den, (size_t)P));
    if (hd->g_unrel.k) {
        float *tmp_roi = nullptr;
        TRY(upload(hd, &tmp_roi, roif));
        launch_gauss_rows(tmp_roi, hd->tmpf, hd->g_unrel.d, hd->g_unrel.k, 1, h, w, 0);
        launch_gauss_cols(hd->tmpf, hd->roi_den, hd->g_unrel.d, hd->g_unrel.k, 1, h, w, 0);
        std::vector<float> hden(P);
        if (hipMemcpy(hden.data(), hd->roi_den, P * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { vistaf_ftp_destroy(hd); return fail(VISTAF_E_HIP, "roi_den readback"); }
        for (int i = 0; i < P; i++) hden[i] = hden[i] + 1e-6f;
        if (hipMemcpy(hd->roi_den, hden.data(), P * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { vistaf_ftp_destroy(hd); return fail(VISTAF_E_HIP, "roi_den upload"); }
    }
    hd->Hf = h + 2 * std::max(0, cfg->fft_pad_px);
    hd->Wf = w + 2 * std::max(0, cfg->fft_pad_px);
    if (hipDeviceSynchronize() != hipSuccess) { vistaf_ftp_destroy(hd); return fail(VISTAF_E_HIP, "create sync"); }
#undef TRY
    *out = hd;
    return 0;
}

// carrier search + tables + demodulation of nb reference frames that preprocess() has left in hd->iw / hd->mu (shape_ftp.py:867-961).
// Asynchronous; geometry lands in geom_dev[0..nb).  ph / pw: patch size the DFT launches use (session mode: read back from the geometry
// between the two halves; pair mode: the full 2*bw+1 square, frames whose patch is clipped by the spectrum border are refused).
static int reference_search(vistaf_ftp_handle *hd, int nb, CarrierGeom *geom_dev, hipStream_t st)
{
    const vistaf_ftp_config &c = hd->cfg;
    const int h = hd->h, w = hd->w, Hf = hd->Hf, Wf = hd->Wf, pad = std::max(0, c.fft_pad_px);
    if (hd->search_cap < nb) {
        int rc;
        if (!hd->Exf) {
            if ((rc = dalloc(hd, &hd->Exf, (size_t)w * (Wf / 2 + 1))) || (rc = dalloc(hd, &hd->Eyf, (size_t)Hf * h))) return rc;
            launch_build_full_tables(hd->Exf, hd->Eyf, h, w, pad, Hf, Wf, st);
        }
        // The search buffers grow geometrically (at least doubling, at most max_batch frames) and the previous set is RELEASED here, not
        // kept until destroy: a caller whose pair batches grow (1, 2, ..., max_batch) holds one set of <= max_batch frames at any time
        // instead of piling up O(max_batch^2) frames of Hf x Wf doubles.  (The first pair-mode call and every growth allocate, i.e. synchronise.)
        const int cap = std::min(std::max(nb, 2 * hd->search_cap), std::max(nb, hd->maxB));
        if (hd->search_cap > 0) {
            HIPCHK(hipStreamSynchronize(st));          // earlier searches on this stream may still read the old set
            for (void *q : {(void *)hd->search_tmp, (void *)hd->search_mag, (void *)hd->search_peaks}) {
                auto it = std::find(hd->allocs.begin(), hd->allocs.end(), q);
                if (it != hd->allocs.end()) { hipFree(q); hd->allocs.erase(it); }
            }
            hd->search_tmp = nullptr; hd->search_mag = nullptr; hd->search_peaks = nullptr; hd->search_cap = 0;
        }
        if ((rc = dalloc(hd, &hd->search_tmp, (size_t)cap * h * (Wf / 2 + 1) + 64)) || (rc = dalloc(hd, &hd->search_mag, (size_t)cap * Hf * Wf)) ||
            (rc = dalloc(hd, &hd->search_peaks, (size_t)cap * 192)))
            return rc;
        hd->search_cap = cap;
    }
    const int npk = std::min(std::max(1, c.n_fft_peaks), 64);
    launch_dft_full_mag(hd->iw, hd->mu, hd->Exf, hd->Eyf, hd->search_tmp, hd->search_mag, nb, h, w, Hf, Wf, st);
    launch_top_peaks(hd->search_mag, nb, Hf, Wf, c.dc_exclusion, npk, hd->search_peaks, st);
    launch_carrier_choose(hd->search_peaks, npk, hd->search_mag, Hf, Wf, std::max(3, c.patch_half_width_bins), c.peak_max_dy_from_center, geom_dev, nb, st);
    return 0;
}

int vistaf_ftp_set_reference(vistaf_ftp_handle *hd, const void *d_ref, int format, void *stream)
{
    if (!hd || !d_ref) return fail(VISTAF_E_INVALID, "null argument");
    if (format < 0 || format > 3) return fail(VISTAF_E_INVALID, "bad frame format");
    hipStream_t st = (hipStream_t)stream;
    const vistaf_ftp_config &c = hd->cfg;
    int h = hd->h, w = hd->w, P = hd->P, Hf = hd->Hf, Wf = hd->Wf, pad = std::max(0, c.fft_pad_px);
    hd->have_ref = false;
    HIPCHK(hipMemsetAsync(hd->status, 0, sizeof(int32_t) * hd->maxB, st));
    preprocess(hd, d_ref, format, 1, st, false);
    int rc = reference_search(hd, 1, hd->geom, st);
    if (rc) return rc;
    CarrierGeom g;
    HIPCHK(hipMemcpyAsync(&g, hd->geom, sizeof(g), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (!g.ok) return fail(VISTAF_E_NOCARRIER, "no carrier peak in the reference spectrum");
    hd->peak_x = g.peak_x; hd->peak_y = g.peak_y; hd->kx = g.kx; hd->ky = g.ky;
    hd->ph = g.ph; hd->pw = g.pw;
    {
        std::vector<float> win = hann_patch(g.ph, g.pw);
        HIPCHK(hipMemcpyAsync(hd->win, win.data(), win.size() * sizeof(float), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));         // `win` is a host temporary
    }
    launch_build_tables(hd->geom, 0, hd->Ex, hd->Ey, hd->Gx, hd->Gy, 0, 0, 1, h, w, pad, Hf, Wf, hd->pmax, st);
    launch_dft_forward(hd->iw, hd->mu, hd->Ex, hd->Ey, 0, 0, hd->win, hd->tmpT, hd->patch, hd->pmax * hd->pmax, 1, h, w, hd->ph, hd->pw, st);
    launch_dft_inverse(hd->patch, hd->pmax * hd->pmax, hd->Gx, hd->Gy, 0, 0, hd->tmpT, hd->cref, hd->amp_ref, nullptr, nullptr, 0, nullptr, nullptr,
                       1, h, w, hd->ph, hd->pw, st);
    HIPCHK(hipStreamSynchronize(st));
    // period estimate (shape_ftp.py:2015-2027) and scale (force_sensor.py:173-187); carrier is locked so k_def == k_ref
    hd->period = g.period;
    hd->mm_per_px = hd->period > 1e-12 ? c.grating_pitch_mm / hd->period : 0.0;
    hd->have_ref = true;
    (void)P;
    return 0;
}

int vistaf_ftp_get_reference_info(const vistaf_ftp_handle *hd, double *o)
{
    if (!hd || !o) return fail(VISTAF_E_INVALID, "null argument");
    if (!hd->have_ref) return fail(VISTAF_E_STATE, "set_reference has not been called");
    o[0] = hd->peak_x; o[1] = hd->peak_y; o[2] = hd->kx; o[3] = hd->ky; o[4] = hd->Hf; o[5] = hd->Wf; o[6] = hd->period; o[7] = hd->mm_per_px;
    return 0;
}

// Everything after the demodulation (shape_ftp.py:1655-2037 + the force tail), shared by the session mode and the uncached-pair mode.
// In: hd->amp-independent planes `prod` (amp_ref * amp_def) and `wrapped`.  pair_geom: per-frame carriers (pair mode) or null.
static int post_demod(vistaf_ftp_handle *hd, int B, float *d_height_mm, uint8_t *d_reliable, double *d_scalars, int32_t *d_status,
                      const CarrierGeom *pair_geom, hipStream_t st, const int *bad_count = nullptr)
{
    if (!bad_count) bad_count = hd->bad_count;
    const vistaf_ftp_config &c = hd->cfg;
    int h = hd->h, w = hd->w, P = hd->P;
    bool timed = hd->timing;

    // ---- reliable mask (shape_ftp.py:739-775)
    if (timed) hipEventRecord(hd->ev[ST_RELIABLE], st);
    const float *qual = hd->prod;
    if (hd->g_qual.k) { blur(hd, hd->prod, hd->quality, hd->g_qual, B, st); qual = hd->quality; }
    else HIPCHK(hipMemcpyAsync(hd->quality, hd->prod, (size_t)B * P * sizeof(float), hipMemcpyDeviceToDevice, st));
    qual = hd->quality;
    launch_select(qual, hd->roi, 0, nullptr, false, hd->req_amp, 1, hd->amp_thr, nullptr, B, P, st, hd->tiers.big_chain ? hd->big_scratch : nullptr);
    launch_threshold_mask(qual, hd->roi, hd->amp_thr, hd->rel0, B, P, st);
    {
        // MORPH_CLOSE with n iterations = n dilations then n erosions; the eroded-ROI mask applies to the result
        uint8_t *src = hd->rel0, *dst = hd->rel1;
        if (c.valid_close_iters >= 1 && 2 * c.valid_close_iters <= 4) {
            int ops[4];
            for (int it = 0; it < c.valid_close_iters; it++) { ops[it] = 1; ops[c.valid_close_iters + it] = 0; }
            launch_morph_seq(hd->rel0, hd->rel1, hd->rel2, B, h, w, hd->se_close, ops, 2 * c.valid_close_iters, hd->roi, nullptr, st, hd->morph_pre);
            src = hd->rel1;
        } else {
            for (int it = 0; it < c.valid_close_iters; it++) { launch_morph(src, dst, B, h, w, hd->se_close, true, nullptr, nullptr, st, hd->morph_pre); std::swap(src, dst); }
            for (int it = 0; it < c.valid_close_iters; it++) {
                launch_morph(src, dst, B, h, w, hd->se_close, false, it == c.valid_close_iters - 1 ? hd->roi : nullptr, nullptr, st, hd->morph_pre);
                std::swap(src, dst);
            }
        }
        launch_cc_label(src, hd->labels, B, h, w, st);
        launch_cc_largest(hd->labels, hd->area, hd->cc_best, hd->roi, hd->rel2, B, P, st);
    }
    if (c.reliable_edge_margin_px > 0) {
        // erode_by_distance (shape_ftp.py:368-377): keep a pixel when its 3x3-chamfer distance to the nearest zero pixel exceeds the margin,
        // i.e. when no zero pixel lies in the chamfer ball of that radius: an erosion by the ball (image border: nothing to erode, as in
        // cv::distanceTransform where the outside holds no zero pixel).  The ball is a symmetric row-span element: bit-plane kernel.
        RowSpanSE ball;
        if (chamfer_ball((float)c.reliable_edge_margin_px, &ball))
            launch_morph(hd->rel2, hd->reliable, B, h, w, ball, false, nullptr, nullptr, st, hd->morph_pre);
        else {
            launch_chamfer(hd->rel2, false, hd->rowdist, hd->dist, B, h, w, c.reliable_edge_margin_px + 1, st, hd->tiers.chamfer_twopass != 0);
            launch_erode_by_dist(hd->dist, hd->rel2, (float)c.reliable_edge_margin_px, hd->reliable, B, P, st);
        }
    } else HIPCHK(hipMemcpyAsync(hd->reliable, hd->rel2, (size_t)B * P, hipMemcpyDeviceToDevice, st));
    launch_count_u8(hd->reliable, hd->rel_count, B, P, st);
    launch_mark_empty(hd->rel_count, hd->status, B, st);

    // ---- unwrap (shape_ftp.py:1702)
    if (timed) hipEventRecord(hd->ev[ST_UNWRAP_RANK], st);
    launch_unwrap(hd->wrapped, qual, hd->reliable, hd->unwrapped, hd->parent, hd->unwrap_scratch, hd->status, B, h, w, st,
                  timed ? hd->ev[ST_UNWRAP_TREE] : nullptr, timed ? hd->ev[ST_UNWRAP] : nullptr, hd->tiers.flood, hd->tiers.unwrap_fast ? hd->unwrap_need : nullptr);

    // ---- plane removal + two-pass detrend (shape_ftp.py:1706, :1716-1751)
    if (timed) hipEventRecord(hd->ev[ST_DETREND], st);
    if (c.plane_order_for_removal > 0)
        // debug_ramp gates on the reliable count, NaN pixels included (:1364-1366), robust_polyfit2d on 200 finite samples (:1103)
        launch_robust_polyfit(hd->unwrapped, hd->reliable, c.plane_order_for_removal, c.irls_iters, (float)c.irls_c, 200, 500, hd->coef, hd->phase1, B, h, w, st, hd->tiers.fit_capped, hd->tiers.big_chain ? hd->big_scratch : nullptr);
    else   // no debug_ramp (the constants of Code/phase_to_height.py): the unwrapped phase goes to the detrend as it is
        HIPCHK(hipMemcpyAsync(hd->phase1, hd->unwrapped, (size_t)B * P * sizeof(float), hipMemcpyDeviceToDevice, st));
    launch_robust_polyfit(hd->phase1, hd->reliable, c.poly_order, c.irls_iters, (float)c.irls_c, 200, 0, hd->coef, hd->resid0, B, h, w, st, hd->tiers.fit_capped, hd->tiers.big_chain ? hd->big_scratch : nullptr);
    launch_select(hd->resid0, hd->reliable, (size_t)P, nullptr, true, hd->req_contact, 3, hd->thr3, nullptr, B, P, st, hd->tiers.big_chain ? hd->big_scratch : nullptr);
    launch_contact_mask(hd->resid0, hd->reliable, hd->thr3, hd->rel_count, hd->contact_count, (float)c.min_contact_frac, (float)c.max_contact_frac,
                        hd->contact, hd->thr_used, B, P, st);
    {
        int iters = std::max(1, c.dilate_iters);
        if (iters <= 4) {
            int ops[4] = {1, 1, 1, 1};
            launch_morph_seq(hd->contact, hd->contact_d, hd->cand, B, h, w, hd->se_contact, ops, iters, nullptr, hd->reliable, st, hd->morph_pre);   // cand is free until the blob filter
        } else {
            uint8_t *src = hd->contact, *dst = hd->contact_d;
            uint8_t *bufs[2] = {hd->contact_d, hd->cand};
            for (int it = 0; it < iters; it++) {
                dst = bufs[it & 1];
                launch_morph(src, dst, B, h, w, hd->se_contact, true, nullptr, it == iters - 1 ? hd->reliable : nullptr, st, hd->morph_pre);
                src = dst;
            }
            if (src != hd->contact_d) HIPCHK(hipMemcpyAsync(hd->contact_d, src, (size_t)B * P, hipMemcpyDeviceToDevice, st));
        }
    }
    launch_background(hd->reliable, hd->contact_d, hd->rel_count, hd->bg_count, hd->background, B, P, st);
    launch_robust_polyfit(hd->phase1, hd->background, c.poly_order, c.irls_iters, (float)c.irls_c, 200, 0, hd->coef, hd->detr, B, h, w, st, hd->tiers.fit_capped, hd->tiers.big_chain ? hd->big_scratch : nullptr);
    launch_select(hd->detr, hd->background, (size_t)P, nullptr, false, hd->req_med, 1, hd->bg_med, nullptr, B, P, st, hd->tiers.big_chain ? hd->big_scratch : nullptr);

    // ---- reliable-only smoothing + sign flip (shape_ftp.py:1753-1768)
    if (timed) hipEventRecord(hd->ev[ST_SMOOTH_FLIP], st);
    const bool smooth_rel = hd->g_rel.k > 0;
    if (smooth_rel) {
        launch_sub_scalar_mask(hd->detr, hd->bg_med, hd->reliable, hd->z0, hd->mplane, B, P, st);
        blur(hd, hd->z0, hd->num, hd->g_rel, B, st);
        blur(hd, hd->mplane, hd->den, hd->g_rel, B, st);
        launch_div_planes(hd->num, hd->den, hd->hmap, B, P, st);
    } else launch_zeroed_keep_nan(hd->detr, hd->bg_med, hd->reliable, hd->hmap, B, P, st);      // NaN stays NaN: the hole stage below is live
    launch_select(hd->hmap, hd->reliable, (size_t)P, nullptr, false, hd->req_core, 1, hd->core_thr, nullptr, B, P, st, hd->tiers.big_chain ? hd->big_scratch : nullptr);
    launch_select(hd->hmap, hd->reliable, (size_t)P, hd->core_thr, false, hd->req_med, 1, hd->core_med, nullptr, B, P, st, hd->tiers.big_chain ? hd->big_scratch : nullptr);
    launch_core_flip(hd->hmap, hd->core_med, hd->flipped, B, P, st);

    // ---- internal holes of the reliable region (shape_ftp.py:1770-1801): with the smoothing every reliable pixel is finite here and
    // upstream's compute_internal_holes_within_mask returns at its first test; without it the unreached pixels are NaN
    const uint8_t *orel = hd->reliable;                      // output_reliable (:1801)
    if (!smooth_rel) {
        const int ksz = std::max(3, c.hole_neighborhood_px | 1);
        launch_chamfer(hd->reliable, false, hd->rowdist, hd->dist, B, h, w, c.hole_min_dist_px + 1, st, hd->tiers.chamfer_twopass != 0);
        launch_hole_candidates(hd->hmap, hd->reliable, hd->dist, ksz, (float)c.hole_known_fraction, (float)c.hole_min_dist_px, hd->hole_cand, B, h, w, st);
        launch_select(hd->hmap, hd->reliable, (size_t)P, nullptr, false, hd->req_med, 1, hd->hole_med, nullptr, B, P, st, hd->tiers.big_chain ? hd->big_scratch : nullptr);
        launch_hole_tmp(hd->hmap, hd->reliable, hd->hole_cand, hd->hole_med, hd->z0, B, P, st);
        launch_select(hd->z0, hd->reliable, (size_t)P, nullptr, false, hd->req_med, 1, hd->hole_fill, nullptr, B, P, st, hd->tiers.big_chain ? hd->big_scratch : nullptr);
        launch_hole_zin(hd->z0, hd->hole_fill, B, P, st);
        {
            const int range = std::min(100, std::max(1, cv_round((double)c.inpaint_radius)));
            const int32_t *only = nullptr;
            if (hd->tiers.inpaint != 1) only = launch_inpaint_window(hd->z0, hd->hole_cand, range, hd->inpaint_win_scratch, B, h, w, st, nullptr, hd->tiers.telea_two_tier != 0, hd->tiers.telea_mw != 0);
            launch_inpaint_telea(hd->z0, hd->hole_cand, range, hd->inpaint_scratch, hd->status, only, B, h, w, st);
        }
        launch_hole_merge(hd->hmap, hd->reliable, hd->hole_cand, hd->z0, hd->out_rel, B, P, st);
        orel = hd->out_rel;
    }

    // ---- frontier taper, composition, unreliable-region smoothing, clamp (shape_ftp.py:1770-1841)
    if (timed) hipEventRecord(hd->ev[ST_COMPOSE], st);
    bool use_band = c.frontier_zero_band_px > 0;
    float band = (float)c.frontier_zero_band_px;
    // both distance transforms of the reliable mask (to its outside for the taper, to its inside for the final blend) in one launch; the
    // second one lands in planes that are idle at this point (`area` as the integer temporary, `depth` -- written by to_mm below)
    if (use_band) launch_chamfer_pair(orel, hd->rowdist, hd->dist, hd->area, hd->depth, B, h, w, c.frontier_zero_band_px + 2, st, hd->tiers.chamfer_twopass != 0);
    else HIPCHK(hipMemsetAsync(hd->dist, 0x7f, (size_t)B * P * sizeof(float), st));   // huge distance: taper weight 1
    launch_frontier_compose(hd->hmap, orel, hd->roi, hd->dist, use_band ? band : 1.0f, hd->z0f, B, P, st);
    if (hd->g_unrel.k) blur(hd, hd->z0f, hd->snum, hd->g_unrel, B, st);
    launch_finalize_unitless(hd->z0f, hd->g_unrel.k ? hd->snum : nullptr, hd->roi_den, orel, hd->roi, use_band ? hd->depth : hd->dist, band, use_band ? 1 : 0,
                             hd->unitless, B, P, st);

    // ---- unitless -> mm, blob filter (shape_ftp.py:1850-1873)
    if (timed) hipEventRecord(hd->ev[ST_MM_BLOB], st);
    launch_to_mm(hd->unitless, hd->roi, hd->hcurve, hd->use_neg, hd->depth, hd->cand, hd->gmax, B, P, st);
    launch_cc_label(hd->cand, hd->labels, B, h, w, st);
    launch_blob_filter(hd->depth, hd->cand, hd->labels, hd->peak_bits, hd->gmax, (float)c.contact_blob_min_peak_mm,
                       c.contact_blob_min_peak_rel_frac, hd->kept, B, P, st);

    // ---- force tail (multimodal_sensor.py:388-419) + arg-extrema
    if (timed) hipEventRecord(hd->ev[ST_TAIL], st);
    PostParams pp;
    pp.mm_per_px = hd->mm_per_px; pp.depth_eps_mm = c.depth_eps_mm; pp.period_px = hd->period; pp.force_curve = hd->fcurve;
    pp.pair_geom = pair_geom; pp.grating_pitch_mm = c.grating_pitch_mm;
    launch_tail(hd->depth, nullptr, hd->unitless, hd->roi, pp, hd->scalars, VISTAF_NSCALARS, nullptr, B, P, st, hd->tiers.big_chain ? hd->big_scratch : nullptr,
                hd->big_scratch ? big_scratch_bytes(hd->maxB, h, w) : 0);
    launch_fill_scalars(hd->scalars, VISTAF_NSCALARS, hd->rel_count, hd->flipped, hd->amp_thr, hd->thr_used, hd->bg_med, bad_count, B, st);
    launch_copy_out(hd->depth, orel, hd->status, d_height_mm, d_reliable, B, P, st);
    if (d_scalars) HIPCHK(hipMemcpyAsync(d_scalars, hd->scalars, sizeof(double) * VISTAF_NSCALARS * B, hipMemcpyDeviceToDevice, st));
    if (d_status) HIPCHK(hipMemcpyAsync(d_status, hd->status, sizeof(int32_t) * B, hipMemcpyDeviceToDevice, st));
    if (timed) {
        hipEventRecord(hd->ev[ST_COUNT], st);
        hipEventSynchronize(hd->ev[ST_COUNT]);
        for (int i = 0; i < ST_COUNT; i++) hipEventElapsedTime(&hd->stage_ms[i], hd->ev[i], hd->ev[i + 1]);
    }
#ifdef VISTAF_DEBUG
    if (getenv("VISTAF_TELEA_DBG")) { hipStreamSynchronize(st); telea_debug_dump(); telea_window_debug_dump(B); telea_window_mw_debug_dump(B); fit_debug_dump(); unwrap_big_debug_dump(); unwrap_batch_debug_dump(); }
#endif
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VISTAF_E_HIP, std::string("launch: ") + hipGetErrorString(e));
    return 0;
}

int vistaf_ftp_predict_batch(vistaf_ftp_handle *hd, const void *d_frames, int format, int B, float *d_height_mm, uint8_t *d_reliable,
                             double *d_scalars, int32_t *d_status, void *stream)
{
    if (!hd || !d_frames) return fail(VISTAF_E_INVALID, "null argument");
    if (!hd->have_ref) return fail(VISTAF_E_STATE, "set_reference has not been called");
    if (B < 1 || B > hd->maxB) return fail(VISTAF_E_STATE, "batch exceeds max_batch");
    if (format < 0 || format > 3) return fail(VISTAF_E_INVALID, "bad frame format");
    if (!(hd->period > 1e-12)) return fail(VISTAF_E_STATE, "Invalid estimated_grating_period_px");
    hipStream_t st = (hipStream_t)stream;
    int h = hd->h, w = hd->w;
    bool timed = hd->timing;
    if (timed && !hd->ev_made) { for (int i = 0; i <= ST_COUNT; i++) hipEventCreate(&hd->ev[i]); hd->ev_made = true; }
    HIPCHK(hipMemsetAsync(hd->status, 0, sizeof(int32_t) * B, st));

    preprocess(hd, d_frames, format, B, st, timed);

    // ---- demodulation, carrier locked to the reference (shape_ftp.py:1643-1653, :1681-1689)
    if (timed) hipEventRecord(hd->ev[ST_DEMOD], st);
    launch_dft_forward(hd->iw, hd->mu, hd->Ex, hd->Ey, 0, 0, hd->win, hd->tmpT, hd->patch, hd->pmax * hd->pmax, B, h, w, hd->ph, hd->pw, st);
    launch_dft_inverse(hd->patch, hd->pmax * hd->pmax, hd->Gx, hd->Gy, 0, 0, hd->tmpT, hd->keep_planes ? hd->field : nullptr, hd->amp, hd->cref,
                       hd->amp_ref, 0, hd->prod, hd->wrapped, B, h, w, hd->ph, hd->pw, st);
    return post_demod(hd, B, d_height_mm, d_reliable, d_scalars, d_status, nullptr, st);
}

// Uncached pairs: every sample brings its own reference frame, as Code/height_to_force.py:384 runs shape_ftp.main per image (reference
// demodulation with carrier search :1632-1639, then the deformed frame locked to it :1643-1653).  BASELINE configs[4] restated
// (SURVEY.md 8d): B (reference, deformed) pairs, two demodulations per sample, no fusion of any kind exists upstream.
int vistaf_ftp_predict_pairs(vistaf_ftp_handle *hd, const void *d_refs, const void *d_defs, int format, int B, float *d_height_mm,
                             uint8_t *d_reliable, double *d_scalars, int32_t *d_status, void *stream)
{
    if (!hd || !d_refs || !d_defs) return fail(VISTAF_E_INVALID, "null argument");
    if (B < 1 || B > hd->maxB) return fail(VISTAF_E_STATE, "batch exceeds max_batch");
    if (format < 0 || format > 3) return fail(VISTAF_E_INVALID, "bad frame format");
    hipStream_t st = (hipStream_t)stream;
    const vistaf_ftp_config &c = hd->cfg;
    int h = hd->h, w = hd->w, P = hd->P, pad = std::max(0, c.fft_pad_px), pm = hd->pmax;
    bool timed = hd->timing;
    if (timed && !hd->ev_made) { for (int i = 0; i <= ST_COUNT; i++) hipEventCreate(&hd->ev[i]); hd->ev_made = true; }
    if (!hd->pairs_ready) {
        int rc;
        size_t mb = hd->maxB;
        if ((rc = dalloc(hd, &hd->pgeom, mb)) || (rc = dalloc(hd, &hd->pEx, mb * w * pm)) || (rc = dalloc(hd, &hd->pGx, mb * w * pm)) ||
            (rc = dalloc(hd, &hd->pEy, mb * h * pm)) || (rc = dalloc(hd, &hd->pGy, mb * h * pm)) || (rc = dalloc(hd, &hd->pcref, mb * P)) ||
            (rc = dalloc(hd, &hd->pamp_ref, mb * P)) || (rc = dalloc(hd, &hd->win_full, (size_t)pm * pm)))
            return rc;
        std::vector<float> win = hann_patch(pm, pm);
        HIPCHK(hipMemcpy(hd->win_full, win.data(), win.size() * sizeof(float), hipMemcpyHostToDevice));
        hd->pairs_ready = true;
    }
    HIPCHK(hipMemsetAsync(hd->status, 0, sizeof(int32_t) * B, st));
    const size_t sx = (size_t)w * pm, sy = (size_t)h * pm;
    const bool together = 2 * B <= hd->maxB;      // room for both frame sets in the workspace: preprocess them in the same launches
    const float *iw_def = hd->iw, *mu_def = hd->mu;
    const int *bad_def = hd->bad_count;
    if (together) {
        HIPCHK(hipMemsetAsync(hd->status + B, 0, sizeof(int32_t) * B, st));
        preprocess(hd, d_refs, format, B, st, timed, d_defs);
        iw_def = hd->iw + (size_t)B * P; mu_def = hd->mu + B; bad_def = hd->bad_count + B;
    } else preprocess(hd, d_refs, format, B, st, false);
    // ---- reference frames: carrier search, tables, demodulation
    int rc = reference_search(hd, B, hd->pgeom, st);
    if (rc) return rc;
    launch_pair_status(hd->pgeom, pm, hd->status, together ? hd->status + B : nullptr, B, st);
    launch_build_tables(hd->pgeom, 1, hd->pEx, hd->pEy, hd->pGx, hd->pGy, sx, sy, B, h, w, pad, hd->Hf, hd->Wf, pm, st);
    launch_dft_forward(hd->iw, hd->mu, hd->pEx, hd->pEy, sx, sy, hd->win_full, hd->tmpT, hd->patch, pm * pm, B, h, w, pm, pm, st);
    launch_dft_inverse(hd->patch, pm * pm, hd->pGx, hd->pGy, sx, sy, hd->tmpT, hd->pcref, hd->pamp_ref, nullptr, nullptr, 0, nullptr, nullptr, B, h, w,
                       pm, pm, st);
    // ---- deformed frames, carrier locked to their own reference
    if (!together) preprocess(hd, d_defs, format, B, st, timed);
    if (timed) hipEventRecord(hd->ev[ST_DEMOD], st);
    launch_dft_forward(iw_def, mu_def, hd->pEx, hd->pEy, sx, sy, hd->win_full, hd->tmpT, hd->patch, pm * pm, B, h, w, pm, pm, st);
    launch_dft_inverse(hd->patch, pm * pm, hd->pGx, hd->pGy, sx, sy, hd->tmpT, hd->keep_planes ? hd->field : nullptr, hd->amp, hd->pcref, hd->pamp_ref,
                       (size_t)P, hd->prod, hd->wrapped, B, h, w, pm, pm, st);
    return post_demod(hd, B, d_height_mm, d_reliable, d_scalars, d_status, hd->pgeom, st, bad_def);
}

int vistaf_ftp_get_pair_info(vistaf_ftp_handle *hd, int batch, double *out /* [batch][VISTAF_NREFINFO] */, void *stream)
{
    if (!hd || !out) return fail(VISTAF_E_INVALID, "null argument");
    if (!hd->pairs_ready || batch < 1 || batch > hd->maxB) return fail(VISTAF_E_STATE, "predict_pairs has not been called");
    std::vector<CarrierGeom> g(batch);
    HIPCHK(hipMemcpyAsync(g.data(), hd->pgeom, sizeof(CarrierGeom) * batch, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    for (int b = 0; b < batch; b++) {
        double *o = out + (size_t)b * VISTAF_NREFINFO;
        o[0] = g[b].peak_x; o[1] = g[b].peak_y; o[2] = g[b].kx; o[3] = g[b].ky; o[4] = hd->Hf; o[5] = hd->Wf; o[6] = g[b].period;
        o[7] = g[b].period > 1e-12 ? hd->cfg.grating_pitch_mm / g[b].period : 0.0;
    }
    return 0;
}

int vistaf_ftp_get_intermediate(vistaf_ftp_handle *hd, const char *name, void *d_dst, int batch, size_t *bytes_per_frame, void *stream)
{
    if (!hd || !name) return fail(VISTAF_E_INVALID, "null argument");
    auto it = hd->named.find(name);
    if (it == hd->named.end()) return fail(VISTAF_E_INVALID, std::string("unknown intermediate: ") + name);
    size_t per = it->second.second;
    size_t total = per ? per * (size_t)batch : (std::string(name) == "cref" ? (size_t)hd->P * sizeof(double2) : std::string(name) == "amp_ref" ? (size_t)hd->P * sizeof(float) : (size_t)hd->P);
    if (bytes_per_frame) *bytes_per_frame = per ? per : total;
    if (d_dst) HIPCHK(hipMemcpyAsync(d_dst, it->second.first, total, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

int vistaf_ftp_stage_count(void) { return ST_COUNT; }
const char *vistaf_ftp_stage_name(int i) { return (i >= 0 && i < ST_COUNT) ? kStageNames[i] : ""; }
int vistaf_ftp_enable_stage_timing(vistaf_ftp_handle *hd, int enable)
{
    if (!hd) return fail(VISTAF_E_INVALID, "null handle");
    hd->timing = enable != 0;
    return 0;
}
int vistaf_ftp_get_stage_times(vistaf_ftp_handle *hd, float *ms_out, int n)
{
    if (!hd || !ms_out) return fail(VISTAF_E_INVALID, "null argument");
    for (int i = 0; i < n && i < ST_COUNT; i++) ms_out[i] = hd->stage_ms[i];
    return 0;
}

// csrc/test_hooks.h (not part of the public header): kernel tier selection and debug planes for the parity tests
int vistaf_ftp_test_set(vistaf_ftp_handle *hd, const char *name, int value)
{
    if (!hd || !name) return fail(VISTAF_E_INVALID, "null argument");
    const std::string n(name);
    if (n == "inpaint_tier" && value >= 0 && value <= 2) hd->tiers.inpaint = value;
    else if (n == "flood_tier" && value >= 0 && value <= 3) hd->tiers.flood = value;
    else if (n == "chamfer_twopass") hd->tiers.chamfer_twopass = value != 0;
    else if (n == "telea_two_tier") hd->tiers.telea_two_tier = value != 0;
    else if (n == "fit_capped") hd->tiers.fit_capped = value != 0;
    else if (n == "telea_mw") hd->tiers.telea_mw = value != 0;
    else if (n == "big_queue_lds") hd->tiers.big_queue_lds = value != 0;
    else if (n == "unwrap_fast") hd->tiers.unwrap_fast = value != 0;
    else if (n == "big_chain") hd->tiers.big_chain = value != 0;
    else if (n == "keep_planes") hd->keep_planes = value != 0;
    else return fail(VISTAF_E_INVALID, "unknown test hook or value: " + n);
    return 0;
}

int vistaf_depth_map_to_volume(const float *d_height, const uint8_t *d_roi, int batch, int h, int w, double mm_per_px,
                               double depth_eps_mm, double *d_out, void *stream)
{
    if (!d_height || !d_out || batch < 1 || h < 1 || w < 1) return fail(VISTAF_E_INVALID, "bad argument");
    PostParams pp;
    pp.mm_per_px = mm_per_px; pp.depth_eps_mm = depth_eps_mm; pp.period_px = 0; pp.force_curve = Curve{0, 0, 0, 0};
    launch_tail(d_height, d_roi, nullptr, nullptr, pp, nullptr, 0, d_out, batch, h * w, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VISTAF_E_HIP, std::string("launch: ") + hipGetErrorString(e));
    return 0;
}

int vistaf_predict_force_from_volume(const vistaf_curve *curve, double volume_cm3, double *force_out)
{
    if (!curve || !force_out) return fail(VISTAF_E_INVALID, "null argument");
    if (curve->type < 0 || curve->type > 5) return fail(VISTAF_E_INVALID, "Unknown model type in force calibration JSON");
    Curve cv{curve->type, curve->a, curve->b, curve->c};
    *force_out = curve_eval(cv, volume_cm3);
    return 0;
}

}  // extern "C"
