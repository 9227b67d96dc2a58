// C ABI of the MI355X-native VISTAF FTP path (see include/vistaf_ftp.h).  Host orchestration only:
// every image operation runs in a HIP kernel of this library; the host builds constant tables
// (Gaussian taps, Hann window, DFT twiddles, ROI / apodisation planes) in double precision.
#include <hip/hip_runtime.h>
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
namespace vf { void telea_debug_dump(); void telea_window_debug_dump(int B); }

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
const char *kStageNames[ST_COUNT] = {"gray+badpix", "inpaint (k_telea_window)", "illum+blur+apod+median", "pruned-dft demod", "reliable mask",
                                     "unwrap rank (k_unwrap_rank)", "unwrap flood (k_unwrap_flood_batch)", "unwrap tree", "detrend (3x IRLS)", "smooth+flip", "frontier+compose", "mm+blob filter", "tail"};

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

    // reference state
    double peak_x = 0, peak_y = 0, kx = 0, ky = 0, period = 0, mm_per_px = 0;
    int Hf = 0, Wf = 0, ph = 0, pw = 0;
    float2 *Ex = nullptr, *Ey = nullptr, *Gx = nullptr, *Gy = nullptr, *cref = nullptr;
    float *win = nullptr, *amp_ref = nullptr;

    // workspace (maxB frames)
    std::map<std::string, std::pair<void *, size_t>> named;   // name -> (ptr, bytes per frame)
    std::vector<void *> allocs;
    float *img, *grad, *tmpf, *blurA, *inorm, *iw, *amp, *prod, *quality, *wrapped, *unwrapped, *phase1, *resid0, *detr, *z0, *mplane,
        *num, *den, *hmap, *dist, *z0f, *snum, *unitless, *depth;
    float2 *field, *patch;
    double2 *tmpT;
    uint8_t *bad0, *bad1, *rel0, *rel1, *rel2, *reliable, *contact, *contact_d, *background, *cand, *kept;
    int32_t *labels, *area, *rowdist, *parent;
    unsigned int *peak_bits;
    uint16_t *morph_pre;
    void *inpaint_scratch, *inpaint_cl_scratch, *inpaint_win_scratch, *unwrap_scratch;
    // small per-frame arrays
    float *thr_hi, *thr_g, *mu, *amp_thr, *thr3, *thr_used, *bg_med, *core_thr, *core_med, *coef;
    int *cnt_a, *cnt_valid, *rel_count, *contact_count, *bg_count, *bad_count, *flipped;
    unsigned int *gmax;
    int32_t *status;
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

int make_se(int k, RowSpanSE *se)
{
    k = std::max(3, k | 1);
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
void preprocess(vistaf_ftp_handle *hd, const void *frames, int format, int B, hipStream_t st, bool timed)
{
    const vistaf_ftp_config &c = hd->cfg;
    int h = hd->h, w = hd->w, P = hd->P;
    if (timed) hipEventRecord(hd->ev[ST_GRAY_BAD], st);
    launch_to_gray(frames, format, hd->img, B, P, st);
    hipMemsetAsync(hd->bad_count, 0, sizeof(int) * B, st);
    if (c.bad_pixel_enable) {
        launch_sobel_mag(hd->img, hd->grad, B, h, w, st);
        launch_select(hd->img, hd->valid, 0, nullptr, false, hd->req_hi, 1, hd->thr_hi, hd->cnt_valid, B, P, st);
        launch_select(hd->grad, hd->valid, 0, nullptr, false, hd->req_g, 1, hd->thr_g, nullptr, B, P, st);
        launch_bad_flags(hd->img, hd->grad, hd->valid, hd->thr_hi, hd->thr_g, hd->bad0, B, P, st);
        uint8_t *src = hd->bad0, *dst = hd->bad1;
        if (c.bad_dilate_ksize > 1)
            for (int it = 0; it < c.bad_dilate_iters; it++) { launch_morph(src, dst, B, h, w, hd->se_bad, true, nullptr, nullptr, st); std::swap(src, dst); }
        if (src != hd->bad1) hipMemcpyAsync(hd->bad1, src, (size_t)B * P, hipMemcpyDeviceToDevice, st);
        launch_count_u8(hd->bad1, hd->bad_count, B, P, st);
        {
            bool ev_done = false;
            int range = std::min(100, std::max(1, cv_round((double)c.bad_inpaint_radius)));   // cv::inpaint clamps the radius
            const uint8_t *seq_mask = hd->bad1;
            const int32_t *only = nullptr;
            // default: the frame-window kernel (LDS-resident march), then the whole-frame kernel for the frames it hands back.
            // VISTAF_INPAINT=cluster: first march every small independent cluster of hole pixels on its own window (pays off when
            // the hole mask is many separate blobs; the fringe crests of this path form a few large clusters per frame, so it is off
            // by default); =seq: whole-frame kernel only
            const char *ev = getenv("VISTAF_INPAINT");          // read on every call: the parity tests switch between the tiers
            const int mode = (ev && !strcmp(ev, "seq")) ? 1 : (ev && !strcmp(ev, "cluster")) ? 0 : 2;
            if (mode == 0 && inpaint_clusters_supported(range)) {
                uint8_t *bad_big = nullptr;
                launch_inpaint_clusters(hd->img, hd->bad1, range, hd->inpaint_cl_scratch, &bad_big, B, h, w, st);
                seq_mask = bad_big;
            }
            if (timed && mode != 2) { hipEventRecord(hd->ev[ST_INPAINT], st); ev_done = true; }
            if (mode != 1) only = launch_inpaint_window(hd->img, seq_mask, range, hd->inpaint_win_scratch, B, h, w, st, (timed && !ev_done) ? hd->ev[ST_INPAINT] : nullptr);
            launch_inpaint_telea(hd->img, seq_mask, range, hd->inpaint_scratch, hd->status, only, B, h, w, st);
        }
    } else if (timed) hipEventRecord(hd->ev[ST_INPAINT], st);
    if (timed) hipEventRecord(hd->ev[ST_PREPROC], st);
    blur(hd, hd->img, hd->blurA, hd->g_illum, B, st);
    launch_illum_norm(hd->img, hd->blurA, hd->inorm, B, P, st);
    const float *in = hd->inorm;
    if (hd->g_pre.k) { blur(hd, hd->inorm, hd->blurA, hd->g_pre, B, st); in = hd->blurA; }
    launch_mul_static(in, hd->apo, hd->iw, B, P, st);
    launch_select(hd->iw, hd->valid, 0, nullptr, false, hd->req_med, 1, hd->mu, nullptr, B, P, st);
}

void build_pruned_twiddles(vistaf_ftp_handle *hd, int x0, int y0, double dpx, double dpy, std::vector<float2> &Ex, std::vector<float2> &Ey,
                           std::vector<float2> &Gx, std::vector<float2> &Gy, std::vector<float> &win)
{
    const double PI2 = 6.283185307179586476925286766559;
    int h = hd->h, w = hd->w, pad = hd->cfg.fft_pad_px, Hf = hd->Hf, Wf = hd->Wf, ph = hd->ph, pw = hd->pw;
    int cxs = Wf / 2, cys = Hf / 2;
    std::vector<double> tcx(Wf), tsx(Wf), tcy(Hf), tsy(Hf);
    for (int m = 0; m < Wf; m++) { tcx[m] = std::cos(PI2 * m / Wf); tsx[m] = -std::sin(PI2 * m / Wf); }
    for (int m = 0; m < Hf; m++) { tcy[m] = std::cos(PI2 * m / Hf); tsy[m] = -std::sin(PI2 * m / Hf); }
    std::vector<double> er((size_t)w * pw, 0.0), ei((size_t)w * pw, 0.0);
    for (int X = 0; X < Wf; X++) {
        int xs = reflect_edge(X - pad, w);
        for (int c = 0; c < pw; c++) {
            long f = ((long)(x0 + c - cxs) % Wf + Wf) % Wf;
            int m = (int)((f * X) % Wf);
            er[(size_t)xs * pw + c] += tcx[m]; ei[(size_t)xs * pw + c] += tsx[m];
        }
    }
    Ex.resize((size_t)w * pw);
    for (size_t i = 0; i < Ex.size(); i++) Ex[i] = make_float2((float)er[i], (float)ei[i]);
    std::vector<double> fr((size_t)ph * h, 0.0), fi((size_t)ph * h, 0.0);
    for (int Y = 0; Y < Hf; Y++) {
        int ys = reflect_edge(Y - pad, h);
        for (int a = 0; a < ph; a++) {
            long f = ((long)(y0 + a - cys) % Hf + Hf) % Hf;
            int m = (int)((f * Y) % Hf);
            fr[(size_t)a * h + ys] += tcy[m]; fi[(size_t)a * h + ys] += tsy[m];
        }
    }
    Ey.resize((size_t)ph * h);
    for (size_t i = 0; i < Ey.size(); i++) Ey[i] = make_float2((float)fr[i], (float)fi[i]);
    // inverse: patch element (a,c) sits at frequency (a - ph/2, c - pw/2) after re-centring (shape_ftp.py:945-948)
    Gx.resize((size_t)pw * w);
    for (int c = 0; c < pw; c++)
        for (int x = 0; x < w; x++) {
            double ang = PI2 * ((double)(c - pw / 2) - dpx) * (double)(x + pad) / (double)Wf;
            Gx[(size_t)c * w + x] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
    Gy.resize((size_t)h * ph);
    double scale = 1.0 / ((double)Hf * (double)Wf);
    for (int y = 0; y < h; y++)
        for (int a = 0; a < ph; a++) {
            double ang = PI2 * ((double)(a - ph / 2) - dpy) * (double)(y + pad) / (double)Hf;
            Gy[(size_t)y * ph + a] = make_float2((float)(std::cos(ang) * scale), (float)(std::sin(ang) * scale));
        }
    // np.hanning(ph)[:,None] * np.hanning(pw)[None,:] in float32 (shape_ftp.py:800-807)
    auto hann = [](int M, int n) -> float { return M == 1 ? 1.0f : (float)(0.5 - 0.5 * std::cos(6.283185307179586476925286766559 * n / (M - 1))); };
    win.resize((size_t)ph * pw);
    for (int a = 0; a < ph; a++)
        for (int c = 0; c < pw; c++) win[(size_t)a * pw + c] = hann(ph, a) * hann(pw, c);
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
        TRY(make_se(k, &hd->se_contact));
    }
    // workspace
#define PLANE(T, name) TRY(dalloc(hd, &hd->name, n, #name, (size_t)P * sizeof(T)))
    PLANE(float, img); PLANE(float, grad); PLANE(float, tmpf); PLANE(float, blurA); PLANE(float, inorm); PLANE(float, iw);
    PLANE(float, amp); PLANE(float, prod); PLANE(float, quality); PLANE(float, wrapped); PLANE(float, unwrapped); PLANE(float, phase1);
    PLANE(float, resid0); PLANE(float, detr); PLANE(float, z0); PLANE(float, mplane); PLANE(float, num); PLANE(float, den);
    PLANE(float, hmap); PLANE(float, dist); PLANE(float, z0f); PLANE(float, snum); PLANE(float, unitless); PLANE(float, depth);
    PLANE(float2, field);
    PLANE(uint8_t, bad0); PLANE(uint8_t, bad1); PLANE(uint8_t, rel0); PLANE(uint8_t, rel1); PLANE(uint8_t, rel2); PLANE(uint8_t, reliable);
    PLANE(uint8_t, contact); PLANE(uint8_t, contact_d); PLANE(uint8_t, background); PLANE(uint8_t, cand); PLANE(uint8_t, kept);
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
    }
    int pmax = 2 * bwp + 1;
    TRY(dalloc(hd, &hd->patch, (size_t)max_batch * pmax * pmax, "patch", (size_t)pmax * pmax * sizeof(float2)));
    TRY(dalloc(hd, &hd->tmpT, (size_t)max_batch * (size_t)std::max(h, w) * pmax));
    size_t mb = max_batch;
    TRY(dalloc(hd, &hd->thr_hi, mb)); TRY(dalloc(hd, &hd->thr_g, mb)); TRY(dalloc(hd, &hd->mu, mb)); TRY(dalloc(hd, &hd->amp_thr, mb));
    TRY(dalloc(hd, &hd->thr3, mb * 3)); TRY(dalloc(hd, &hd->thr_used, mb)); TRY(dalloc(hd, &hd->bg_med, mb));
    TRY(dalloc(hd, &hd->core_thr, mb)); TRY(dalloc(hd, &hd->core_med, mb)); TRY(dalloc(hd, &hd->coef, mb * 6));
    TRY(dalloc(hd, &hd->cnt_a, mb)); TRY(dalloc(hd, &hd->cnt_valid, mb)); TRY(dalloc(hd, &hd->rel_count, mb));
    TRY(dalloc(hd, &hd->contact_count, mb)); TRY(dalloc(hd, &hd->bg_count, mb)); TRY(dalloc(hd, &hd->bad_count, mb));
    TRY(dalloc(hd, &hd->flipped, mb)); TRY(dalloc(hd, &hd->gmax, mb)); TRY(dalloc(hd, &hd->status, mb));
    TRY(dalloc(hd, &hd->scalars, mb * VISTAF_NSCALARS));
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

int vistaf_ftp_set_reference(vistaf_ftp_handle *hd, const void *d_ref, int format, void *stream)
{
    if (!hd || !d_ref) return fail(VISTAF_E_INVALID, "null argument");
    if (format < 0 || format > 3) return fail(VISTAF_E_INVALID, "bad frame format");
    hipStream_t st = (hipStream_t)stream;
    const vistaf_ftp_config &c = hd->cfg;
    int h = hd->h, w = hd->w, P = hd->P, Hf = hd->Hf, Wf = hd->Wf, pad = std::max(0, c.fft_pad_px);
    HIPCHK(hipMemsetAsync(hd->status, 0, sizeof(int32_t) * hd->maxB, st));
    preprocess(hd, d_ref, format, 1, st, false);
    // ---- full spectrum magnitude + top peaks (shape_ftp.py:867-905)
    const double PI2 = 6.283185307179586476925286766559;
    std::vector<double> tcx(Wf), tsx(Wf), tcy(Hf), tsy(Hf);
    for (int m = 0; m < Wf; m++) { tcx[m] = std::cos(PI2 * m / Wf); tsx[m] = -std::sin(PI2 * m / Wf); }
    for (int m = 0; m < Hf; m++) { tcy[m] = std::cos(PI2 * m / Hf); tsy[m] = -std::sin(PI2 * m / Hf); }
    std::vector<float2> exf((size_t)w * Wf), eyf((size_t)Hf * h);
    {
        std::vector<double> er((size_t)w * Wf, 0.0), ei((size_t)w * Wf, 0.0);
        for (int X = 0; X < Wf; X++) {
            int xs = reflect_edge(X - pad, w);
            for (int f = 0; f < Wf; f++) { int m = (int)(((long)f * X) % Wf); er[(size_t)xs * Wf + f] += tcx[m]; ei[(size_t)xs * Wf + f] += tsx[m]; }
        }
        for (size_t i = 0; i < exf.size(); i++) exf[i] = make_float2((float)er[i], (float)ei[i]);
        std::vector<double> fr((size_t)Hf * h, 0.0), fi((size_t)Hf * h, 0.0);
        for (int Y = 0; Y < Hf; Y++) {
            int ys = reflect_edge(Y - pad, h);
            for (int f = 0; f < Hf; f++) { int m = (int)(((long)f * Y) % Hf); fr[(size_t)f * h + ys] += tcy[m]; fi[(size_t)f * h + ys] += tsy[m]; }
        }
        for (size_t i = 0; i < eyf.size(); i++) eyf[i] = make_float2((float)fr[i], (float)fi[i]);
    }
    float2 *d_ex = nullptr, *d_ey = nullptr;
    double2 *d_tmp = nullptr;
    float *d_mag = nullptr, *d_peaks = nullptr;
    HIPCHK(hipMalloc((void **)&d_ex, exf.size() * sizeof(float2)));
    HIPCHK(hipMalloc((void **)&d_ey, eyf.size() * sizeof(float2)));
    HIPCHK(hipMalloc((void **)&d_tmp, (size_t)h * Wf * sizeof(double2)));
    HIPCHK(hipMalloc((void **)&d_mag, (size_t)Hf * Wf * sizeof(float)));
    int npk = std::min(std::max(1, c.n_fft_peaks), 64);
    HIPCHK(hipMalloc((void **)&d_peaks, 3 * 64 * sizeof(float)));
    HIPCHK(hipMemcpyAsync(d_ex, exf.data(), exf.size() * sizeof(float2), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_ey, eyf.data(), eyf.size() * sizeof(float2), hipMemcpyHostToDevice, st));
    launch_dft_full_mag(hd->iw, hd->mu, d_ex, d_ey, (float2 *)d_tmp, d_mag, h, w, Hf, Wf, c.dc_exclusion, st);
    launch_top_peaks(d_mag, Hf, Wf, c.dc_exclusion, npk, d_peaks, st);
    std::vector<float> pk(3 * npk);
    HIPCHK(hipMemcpyAsync(pk.data(), d_peaks, pk.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // choose_carrier_peak (shape_ftp.py:444-463): right half plane, near the centre row, largest magnitude
    int cys = Hf / 2, cxs = Wf / 2;
    std::vector<int> cand(npk);
    for (int i = 0; i < npk; i++) cand[i] = i;
    auto filt = [&](auto pred) { std::vector<int> o; for (int i : cand) if (pred(i)) o.push_back(i); if (!o.empty()) cand = o; };
    filt([&](int i) { return pk[3 * i] > (float)cxs; });
    int max_dy = (int)(c.peak_max_dy_from_center * Hf);
    filt([&](int i) { return std::abs((int)pk[3 * i + 1] - cys) <= max_dy; });
    int best = cand[0];
    for (int i : cand) if (pk[3 * i + 2] > pk[3 * best + 2]) best = i;
    int px = (int)pk[3 * best], py = (int)pk[3 * best + 1];
    if (!(pk[3 * best + 2] > 0.f)) {
        hipFree(d_ex); hipFree(d_ey); hipFree(d_tmp); hipFree(d_mag); hipFree(d_peaks);
        return fail(VISTAF_E_NOCARRIER, "no carrier peak in the reference spectrum");
    }
    // refine_peak_parabolic_log (shape_ftp.py:473-483), float32 arithmetic as NumPy scalars give
    double pxf = px, pyf = py;
    if (px > 0 && px < Wf - 1 && py > 0 && py < Hf - 1) {
        float m[5];
        size_t offs[5] = {(size_t)py * Wf + px - 1, (size_t)py * Wf + px, (size_t)py * Wf + px + 1, (size_t)(py - 1) * Wf + px, (size_t)(py + 1) * Wf + px};
        for (int i = 0; i < 5; i++) HIPCHK(hipMemcpy(&m[i], d_mag + offs[i], sizeof(float), hipMemcpyDeviceToHost));
        auto lg = [](float v) { return std::log(v + 1e-12f); };
        auto par = [](float fm1, float f0, float fp1) -> float {
            float den = (fm1 - 2.0f * f0) + fp1;
            if (std::fabs(den) < 1e-12f) return 0.0f;
            return 0.5f * (fm1 - fp1) / den;
        };
        float dx = par(lg(m[0]), lg(m[1]), lg(m[2])), dy = par(lg(m[3]), lg(m[1]), lg(m[4]));
        pxf = (double)((float)px + dx); pyf = (double)((float)py + dy);
    }
    hipFree(d_ex); hipFree(d_ey); hipFree(d_tmp); hipFree(d_mag); hipFree(d_peaks);
    hd->peak_x = pxf; hd->peak_y = pyf;
    hd->kx = pxf - cxs; hd->ky = pyf - cys;
    // patch geometry (shape_ftp.py:930-948)
    int px_i = (int)std::nearbyint(pxf), py_i = (int)std::nearbyint(pyf);
    int bw = std::max(3, c.patch_half_width_bins);
    int x0 = std::max(0, px_i - bw), x1 = std::min(Wf, px_i + bw + 1), y0 = std::max(0, py_i - bw), y1 = std::min(Hf, py_i + bw + 1);
    hd->ph = y1 - y0; hd->pw = x1 - x0;
    if (hd->ph < 1 || hd->pw < 1) return fail(VISTAF_E_NOCARRIER, "carrier patch is empty");
    double dpx = pxf - px_i, dpy = pyf - py_i;
    if (!(std::fabs(dpx) > 1e-6 || std::fabs(dpy) > 1e-6)) { dpx = 0; dpy = 0; }
    std::vector<float2> Ex, Ey, Gx, Gy;
    std::vector<float> win;
    build_pruned_twiddles(hd, x0, y0, dpx, dpy, Ex, Ey, Gx, Gy, win);
    int rc;
    if ((rc = upload(hd, &hd->Ex, Ex)) || (rc = upload(hd, &hd->Ey, Ey)) || (rc = upload(hd, &hd->Gx, Gx)) || (rc = upload(hd, &hd->Gy, Gy)) ||
        (rc = upload(hd, &hd->win, win)))
        return rc;
    if (!hd->cref) { if ((rc = dalloc(hd, &hd->cref, (size_t)P)) || (rc = dalloc(hd, &hd->amp_ref, (size_t)P))) return rc; }
    launch_dft_forward(hd->iw, hd->mu, hd->Ex, hd->Ey, hd->win, (float2 *)hd->tmpT, hd->patch, 1, h, w, hd->ph, hd->pw, st);
    launch_dft_inverse(hd->patch, hd->Gx, hd->Gy, (float2 *)hd->tmpT, hd->field, hd->amp, 1, h, w, hd->ph, hd->pw, st);
    HIPCHK(hipMemcpyAsync(hd->cref, hd->field, (size_t)P * sizeof(float2), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(hd->amp_ref, hd->amp, (size_t)P * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    hd->named["cref"] = {hd->cref, 0}; hd->named["amp_ref"] = {hd->amp_ref, 0};
    // period estimate (shape_ftp.py:2015-2027) and scale (force_sensor.py:173-187); carrier is locked so k_def == k_ref
    hd->period = std::fabs(hd->kx) > 1e-9 ? (double)Wf / std::fabs(hd->kx) : 0.0;
    hd->mm_per_px = hd->period > 1e-12 ? c.grating_pitch_mm / hd->period : 0.0;
    hd->have_ref = true;
    return 0;
}

int vistaf_ftp_get_reference_info(const vistaf_ftp_handle *hd, double *o)
{
    if (!hd || !o) return fail(VISTAF_E_INVALID, "null argument");
    if (!hd->have_ref) return fail(VISTAF_E_STATE, "set_reference has not been called");
    o[0] = hd->peak_x; o[1] = hd->peak_y; o[2] = hd->kx; o[3] = hd->ky; o[4] = hd->Hf; o[5] = hd->Wf; o[6] = hd->period; o[7] = hd->mm_per_px;
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
    const vistaf_ftp_config &c = hd->cfg;
    int h = hd->h, w = hd->w, P = hd->P;
    bool timed = hd->timing;
    if (timed && !hd->ev_made) { for (int i = 0; i <= ST_COUNT; i++) hipEventCreate(&hd->ev[i]); hd->ev_made = true; }
    HIPCHK(hipMemsetAsync(hd->status, 0, sizeof(int32_t) * B, st));

    preprocess(hd, d_frames, format, B, st, timed);

    // ---- demodulation, carrier locked to the reference (shape_ftp.py:1643-1653, :1681-1689)
    if (timed) hipEventRecord(hd->ev[ST_DEMOD], st);
    launch_dft_forward(hd->iw, hd->mu, hd->Ex, hd->Ey, hd->win, (float2 *)hd->tmpT, hd->patch, B, h, w, hd->ph, hd->pw, st);
    launch_dft_inverse(hd->patch, hd->Gx, hd->Gy, (float2 *)hd->tmpT, hd->field, hd->amp, B, h, w, hd->ph, hd->pw, st);
    launch_phase_diff(hd->field, hd->cref, hd->amp, hd->amp_ref, hd->prod, hd->wrapped, B, P, st);

    // ---- reliable mask (shape_ftp.py:739-775)
    if (timed) hipEventRecord(hd->ev[ST_RELIABLE], st);
    const float *qual = hd->prod;
    if (hd->g_qual.k) { blur(hd, hd->prod, hd->quality, hd->g_qual, B, st); qual = hd->quality; }
    else HIPCHK(hipMemcpyAsync(hd->quality, hd->prod, (size_t)B * P * sizeof(float), hipMemcpyDeviceToDevice, st));
    qual = hd->quality;
    launch_select(qual, hd->roi, 0, nullptr, false, hd->req_amp, 1, hd->amp_thr, nullptr, B, P, st);
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
        launch_cc_largest(hd->labels, hd->area, nullptr, hd->roi, hd->rel2, B, P, st);
    }
    if (c.reliable_edge_margin_px > 0) {
        // erode_by_distance (shape_ftp.py:368-377): keep a pixel when its 3x3-chamfer distance to the nearest zero pixel exceeds the margin,
        // i.e. when no zero pixel lies in the chamfer ball of that radius: an erosion by the ball (image border: nothing to erode, as in
        // cv::distanceTransform where the outside holds no zero pixel).  The ball is a symmetric row-span element: bit-plane kernel.
        RowSpanSE ball;
        if (chamfer_ball((float)c.reliable_edge_margin_px, &ball))
            launch_morph(hd->rel2, hd->reliable, B, h, w, ball, false, nullptr, nullptr, st, hd->morph_pre);
        else {
            launch_chamfer(hd->rel2, false, hd->rowdist, hd->dist, B, h, w, c.reliable_edge_margin_px + 1, st);
            launch_erode_by_dist(hd->dist, hd->rel2, (float)c.reliable_edge_margin_px, hd->reliable, B, P, st);
        }
    } else HIPCHK(hipMemcpyAsync(hd->reliable, hd->rel2, (size_t)B * P, hipMemcpyDeviceToDevice, st));
    launch_count_u8(hd->reliable, hd->rel_count, B, P, st);
    launch_mark_empty(hd->rel_count, hd->status, B, st);

    // ---- unwrap (shape_ftp.py:1702)
    if (timed) hipEventRecord(hd->ev[ST_UNWRAP_RANK], st);
    launch_unwrap(hd->wrapped, qual, hd->reliable, hd->unwrapped, hd->parent, hd->unwrap_scratch, hd->status, B, h, w, st,
                  timed ? hd->ev[ST_UNWRAP_TREE] : nullptr, timed ? hd->ev[ST_UNWRAP] : nullptr);

    // ---- plane removal + two-pass detrend (shape_ftp.py:1706, :1716-1751)
    if (timed) hipEventRecord(hd->ev[ST_DETREND], st);
    if (c.plane_order_for_removal > 0)
        launch_robust_polyfit(hd->unwrapped, hd->reliable, c.plane_order_for_removal, c.irls_iters, (float)c.irls_c, 500, hd->coef, hd->phase1, B, h, w, st);
    else   // no debug_ramp (the constants of Code/phase_to_height.py): the unwrapped phase goes to the detrend as it is
        HIPCHK(hipMemcpyAsync(hd->phase1, hd->unwrapped, (size_t)B * P * sizeof(float), hipMemcpyDeviceToDevice, st));
    launch_robust_polyfit(hd->phase1, hd->reliable, c.poly_order, c.irls_iters, (float)c.irls_c, 200, hd->coef, hd->resid0, B, h, w, st);
    launch_select(hd->resid0, hd->reliable, (size_t)P, nullptr, true, hd->req_contact, 3, hd->thr3, nullptr, B, P, st);
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
    launch_robust_polyfit(hd->phase1, hd->background, c.poly_order, c.irls_iters, (float)c.irls_c, 200, hd->coef, hd->detr, B, h, w, st);
    launch_select(hd->detr, hd->background, (size_t)P, nullptr, false, hd->req_med, 1, hd->bg_med, nullptr, B, P, st);

    // ---- reliable-only smoothing + sign flip (shape_ftp.py:1753-1768)
    if (timed) hipEventRecord(hd->ev[ST_SMOOTH_FLIP], st);
    launch_sub_scalar_mask(hd->detr, hd->bg_med, hd->reliable, hd->z0, hd->mplane, B, P, st);
    if (hd->g_rel.k) {
        blur(hd, hd->z0, hd->num, hd->g_rel, B, st);
        blur(hd, hd->mplane, hd->den, hd->g_rel, B, st);
        launch_div_planes(hd->num, hd->den, hd->hmap, B, P, st);
    } else HIPCHK(hipMemcpyAsync(hd->hmap, hd->z0, (size_t)B * P * sizeof(float), hipMemcpyDeviceToDevice, st));
    launch_select(hd->hmap, hd->reliable, (size_t)P, nullptr, false, hd->req_core, 1, hd->core_thr, nullptr, B, P, st);
    launch_select(hd->hmap, hd->reliable, (size_t)P, hd->core_thr, false, hd->req_med, 1, hd->core_med, nullptr, B, P, st);
    launch_core_flip(hd->hmap, hd->core_med, hd->flipped, B, P, st);

    // ---- frontier taper, composition, unreliable-region smoothing, clamp (shape_ftp.py:1770-1841)
    if (timed) hipEventRecord(hd->ev[ST_COMPOSE], st);
    bool use_band = c.frontier_zero_band_px > 0;
    float band = (float)c.frontier_zero_band_px;
    // both distance transforms of the reliable mask (to its outside for the taper, to its inside for the final blend) in one launch; the
    // second one lands in planes that are idle at this point (`area` as the integer temporary, `depth` -- written by to_mm below)
    if (use_band) launch_chamfer_pair(hd->reliable, hd->rowdist, hd->dist, hd->area, hd->depth, B, h, w, c.frontier_zero_band_px + 2, st);
    else HIPCHK(hipMemsetAsync(hd->dist, 0x7f, (size_t)B * P * sizeof(float), st));   // huge distance: taper weight 1
    launch_frontier_compose(hd->hmap, hd->reliable, hd->roi, hd->dist, use_band ? band : 1.0f, hd->z0f, hd->status, B, P, st);
    if (hd->g_unrel.k) blur(hd, hd->z0f, hd->snum, hd->g_unrel, B, st);
    launch_finalize_unitless(hd->z0f, hd->g_unrel.k ? hd->snum : nullptr, hd->roi_den, hd->reliable, hd->roi, use_band ? hd->depth : hd->dist, band, use_band ? 1 : 0,
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
    launch_tail(hd->depth, nullptr, hd->unitless, hd->roi, pp, hd->scalars, VISTAF_NSCALARS, nullptr, B, P, st);
    launch_fill_scalars(hd->scalars, VISTAF_NSCALARS, hd->rel_count, hd->flipped, hd->amp_thr, hd->thr_used, hd->bg_med, hd->bad_count, B, st);
    launch_copy_out(hd->depth, hd->reliable, hd->status, d_height_mm, d_reliable, B, P, st);
    if (d_scalars) HIPCHK(hipMemcpyAsync(d_scalars, hd->scalars, sizeof(double) * VISTAF_NSCALARS * B, hipMemcpyDeviceToDevice, st));
    if (d_status) HIPCHK(hipMemcpyAsync(d_status, hd->status, sizeof(int32_t) * B, hipMemcpyDeviceToDevice, st));
    if (timed) {
        hipEventRecord(hd->ev[ST_COUNT], st);
        hipEventSynchronize(hd->ev[ST_COUNT]);
        for (int i = 0; i < ST_COUNT; i++) hipEventElapsedTime(&hd->stage_ms[i], hd->ev[i], hd->ev[i + 1]);
    }
    if (getenv("VISTAF_TELEA_DBG")) { hipStreamSynchronize(st); telea_debug_dump(); telea_window_debug_dump(B); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VISTAF_E_HIP, std::string("launch: ") + hipGetErrorString(e));
    return 0;
}

int vistaf_ftp_get_intermediate(vistaf_ftp_handle *hd, const char *name, void *d_dst, int batch, size_t *bytes_per_frame, void *stream)
{
    if (!hd || !name) return fail(VISTAF_E_INVALID, "null argument");
    auto it = hd->named.find(name);
    if (it == hd->named.end()) return fail(VISTAF_E_INVALID, std::string("unknown intermediate: ") + name);
    size_t per = it->second.second;
    size_t total = per ? per * (size_t)batch : (std::string(name) == "cref" ? (size_t)hd->P * sizeof(float2) : std::string(name) == "amp_ref" ? (size_t)hd->P * sizeof(float) : (size_t)hd->P);
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
