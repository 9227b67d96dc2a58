/*
 * oracle/cvlite.c -- TEST INFRASTRUCTURE ONLY (the parity oracle). Never linked into, imported by
 * or called from the product path; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load the library built from this file.
 *
 * CPU restatement, in plain C, of the third-party image primitives the reference FTP path calls
 * (OpenCV `cv2.*`, which is NOT present under /root/reference nor installable here) and of the
 * reference's own sequential loops that are too slow as Python for test-sized inputs.
 *
 * The reference pins no OpenCV version (README.md:63 lists bare package names).  Every cv2 stage
 * below follows OpenCV 4.x's documented/published behaviour; at the individual-stage level this is
 * "parity unpinned" (no golden holds a cv2 intermediate) and is checked end-to-end against the
 * reference's stored height_map_bundle.npz files only (see oracle/README.md).
 *
 * Call sites restated (all in /root/reference/Code/shape_ftp.py):
 *   cv2.GaussianBlur((0,0),sigma)      :530 :557 :746 :831 :836 :1145 :1146
 *   cv2.Sobel(ksize=3)                 :633 :634
 *   cv2.getStructuringElement(ELLIPSE) :644 :733 :758 :1734
 *   cv2.dilate / morphologyEx(CLOSE)   :646 :735 :760 :1736
 *   cv2.connectedComponentsWithStats   :712 :1244
 *   cv2.distanceTransform(DIST_L2,3)   :725 :790 :1172 :1309 :1312
 *   cv2.boxFilter(normalize=False)     :1166 :1167
 *   cv2.copyMakeBorder(BORDER_REFLECT) :859
 *   cv2.inpaint(INPAINT_TELEA)         :665 :1199
 *   unwrap_quality_guided (pure Python):1043-1080
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* ------------------------------------------------------------------------------------------- */
/* helpers                                                                                      */
/* ------------------------------------------------------------------------------------------- */

/* cv::borderInterpolate, BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba) */
static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

/* cv::borderInterpolate, BORDER_REFLECT (fedcba|abcdefgh|hgfedcb) */
static int reflect(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p - 1;
        else p = 2 * len - 1 - p;
    }
    return p;
}

/* cvRound: round half to even */
static int cv_round(double v) { return (int)lrint(v); }

/* ------------------------------------------------------------------------------------------- */
/* GaussianBlur, ksize=(0,0), float32 source                                                    */
/* ------------------------------------------------------------------------------------------- */

/* ksize rule of cv::GaussianBlur for a CV_32F source: cvRound(sigma*4*2+1)|1 */
int cvl_gaussian_ksize_f32(double sigma) { return cv_round(sigma * 4 * 2 + 1) | 1; }

/* cv::getGaussianKernel(n, sigma, CV_32F) for sigma>0: exp(-x^2/(2 sigma^2)) in double,
 * normalised in double, cast to float last (OpenCV 4.x getGaussianKernelBitExact path). */
void cvl_gaussian_kernel_f32(int n, double sigma, float *out)
{
    double scale2x = -0.5 / (sigma * sigma);
    double *t = (double *)malloc(sizeof(double) * (size_t)n);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        t[i] = exp(scale2x * x * x);
        sum += t[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < n; i++) out[i] = (float)(t[i] * sum);
    free(t);
}

/* Separable filter, BORDER_REFLECT_101, float accumulation.
 * Row pass: ascending-k sum s = k[0]*S[0]; s = fma(k[j], S[j], s) (cv::RowVec_32f / RowFilter);
 * column pass: symmetric form s = k[c]*S[c]; s = fma(k[c+j], S[c+j]+S[c-j], s) (cv::SymmColumnVec_32f /
 * SymmColumnFilter).  The multiply-adds are FUSED: OpenCV's universal-intrinsic loops use v_muladd, which is
 * a hardware fma in every build that dispatches to AVX2/FMA3 or runs on AArch64 NEON (the builds the pip wheels
 * select on current machines); the SSE2-only baseline would round the product first.  Which one produced the
 * reference's stored outputs is unknowable (OpenCV version and machine unpinned), so the fused form -- one
 * rounding per tap, the more accurate of the two -- is the restatement, and the GPU kernels (k_gauss_*) execute
 * exactly this sequence, so the blurred planes agree bit for bit.
 * target_clones: the "fma" clone inlines fmaf as vfmadd*, the default clone calls libm's (same result). */
void cvl_gaussian_blur_xy_f32(const float *src, float *dst, int h, int w, double sigmax, double sigmay);
void cvl_gaussian_blur_f32(const float *src, float *dst, int h, int w, double sigma) { cvl_gaussian_blur_xy_f32(src, dst, h, w, sigma, sigma); }

/* cv::GaussianBlur(src, (0, 0), sigmaX, sigmaY): each direction with its own kernel; sigmaY <= 0 takes sigmaX (cv::createGaussianKernels) */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__) && !defined(CVL_NO_CLONES)
__attribute__((target_clones("fma", "default")))
#endif
void cvl_gaussian_blur_xy_f32(const float *src, float *dst, int h, int w, double sigmax, double sigmay)
{
    if (sigmay <= 0) sigmay = sigmax;
    int n = cvl_gaussian_ksize_f32(sigmax);
    int r = n / 2;
    float *k = (float *)malloc(sizeof(float) * (size_t)n);
    cvl_gaussian_kernel_f32(n, sigmax, k);
    float *tmp = (float *)malloc(sizeof(float) * (size_t)h * (size_t)w);
    int *xi = (int *)malloc(sizeof(int) * (size_t)(w + 2 * r));
    for (int x = -r; x < w + r; x++) xi[x + r] = reflect101(x, w);
    for (int y = 0; y < h; y++) {
        const float *s = src + (size_t)y * w;
        float *t = tmp + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            float acc = k[0] * s[xi[x]];
            for (int j = 1; j < n; j++) acc = fmaf(k[j], s[xi[x + j]], acc);
            t[x] = acc;
        }
    }
    free(k);
    n = cvl_gaussian_ksize_f32(sigmay);
    r = n / 2;
    k = (float *)malloc(sizeof(float) * (size_t)n);
    cvl_gaussian_kernel_f32(n, sigmay, k);
    int *yi = (int *)malloc(sizeof(int) * (size_t)(h + 2 * r));
    for (int y = -r; y < h + r; y++) yi[y + r] = reflect101(y, h);
    for (int y = 0; y < h; y++) {
        float *d = dst + (size_t)y * w;
        const float *c0 = tmp + (size_t)yi[y + r] * w;
        for (int x = 0; x < w; x++) d[x] = k[r] * c0[x];
        for (int j = 1; j <= r; j++) {
            const float *a = tmp + (size_t)yi[y + r + j] * w;
            const float *b = tmp + (size_t)yi[y + r - j] * w;
            float kj = k[r + j];
            for (int x = 0; x < w; x++) d[x] = fmaf(kj, a[x] + b[x], d[x]);
        }
    }
    free(yi); free(xi); free(tmp); free(k);
}

/* ------------------------------------------------------------------------------------------- */
/* Sobel 3x3 (dx=1,dy=0) and (dx=0,dy=1), CV_32F, BORDER_REFLECT_101                            */
/* ------------------------------------------------------------------------------------------- */
void cvl_sobel3_f32(const float *src, float *gx, float *gy, int h, int w)
{
    for (int y = 0; y < h; y++) {
        int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
        for (int x = 0; x < w; x++) {
            int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
            float a00 = src[(size_t)ym * w + xm], a01 = src[(size_t)ym * w + x], a02 = src[(size_t)ym * w + xp];
            float a10 = src[(size_t)y * w + xm], a12 = src[(size_t)y * w + xp];
            float a20 = src[(size_t)yp * w + xm], a21 = src[(size_t)yp * w + x], a22 = src[(size_t)yp * w + xp];
            gx[(size_t)y * w + x] = (a02 - a00) + 2.0f * (a12 - a10) + (a22 - a20);
            gy[(size_t)y * w + x] = (a20 - a00) + 2.0f * (a21 - a01) + (a22 - a02);
        }
    }
}

/* ------------------------------------------------------------------------------------------- */
/* Structuring element + binary morphology                                                     */
/* ------------------------------------------------------------------------------------------- */

/* cv::getStructuringElement(MORPH_ELLIPSE, (k,k)); se is k*k bytes of 0/1 */
void cvl_ellipse_se(int k, uint8_t *se)
{
    int r = k / 2, c = k / 2;
    double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    memset(se, 0, (size_t)k * k);
    for (int i = 0; i < k; i++) {
        int j1 = 0, j2 = 0;
        int dy = i - r;
        if (abs(dy) <= r) {
            int dx = cv_round(c * sqrt((r * r - dy * dy) * inv_r2));
            j1 = c - dx; if (j1 < 0) j1 = 0;
            j2 = c + dx + 1; if (j2 > k) j2 = k;
        }
        for (int j = j1; j < j2; j++) se[i * k + j] = 1;
    }
}

/* cv::dilate / cv::erode on a 0/255 (or 0/1) image, anchor at centre, iterations applied one
 * after another, border pixels do not contribute (morphologyDefaultBorderValue). */
static void morph_once(const uint8_t *src, uint8_t *dst, int h, int w, const uint8_t *se, int k, int is_dilate)
{
    int r = k / 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int v = is_dilate ? 0 : 1;
            for (int i = 0; i < k && (is_dilate ? !v : v); i++) {
                int yy = y + i - r;
                if (yy < 0 || yy >= h) continue;
                for (int j = 0; j < k; j++) {
                    if (!se[i * k + j]) continue;
                    int xx = x + j - r;
                    if (xx < 0 || xx >= w) continue;
                    int s = src[(size_t)yy * w + xx] != 0;
                    if (is_dilate) { if (s) { v = 1; break; } }
                    else { if (!s) { v = 0; break; } }
                }
            }
            dst[(size_t)y * w + x] = (uint8_t)(v ? 255 : 0);
        }
}

static void morph(const uint8_t *src, uint8_t *dst, int h, int w, const uint8_t *se, int k, int iters, int is_dilate)
{
    size_t n = (size_t)h * w;
    uint8_t *a = (uint8_t *)malloc(n), *b = (uint8_t *)malloc(n);
    memcpy(a, src, n);
    for (int it = 0; it < iters; it++) {
        morph_once(a, b, h, w, se, k, is_dilate);
        uint8_t *t = a; a = b; b = t;
    }
    memcpy(dst, a, n);
    free(a); free(b);
}

void cvl_dilate_u8(const uint8_t *src, uint8_t *dst, int h, int w, const uint8_t *se, int k, int iters)
{ morph(src, dst, h, w, se, k, iters, 1); }

void cvl_erode_u8(const uint8_t *src, uint8_t *dst, int h, int w, const uint8_t *se, int k, int iters)
{ morph(src, dst, h, w, se, k, iters, 0); }

/* ------------------------------------------------------------------------------------------- */
/* connectedComponents, 8-connectivity. Labels 1.. numbered by raster order of each component's */
/* first pixel; returns number of labels including background 0; areas[] (size >= returned n).  */
/* ------------------------------------------------------------------------------------------- */
int cvl_cc8_label(const uint8_t *src, int32_t *labels, int32_t *areas, int areas_cap, int h, int w)
{
    size_t n = (size_t)h * w;
    memset(labels, 0, n * sizeof(int32_t));
    int32_t *stack = (int32_t *)malloc(n * sizeof(int32_t));
    int next = 0;
    if (areas && areas_cap > 0) areas[0] = 0;
    for (size_t p0 = 0; p0 < n; p0++) {
        if (!src[p0] || labels[p0]) continue;
        next++;
        int area = 0;
        size_t sp = 0;
        stack[sp++] = (int32_t)p0; labels[p0] = next;
        while (sp) {
            int32_t p = stack[--sp];
            area++;
            int y = p / w, x = p % w;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    if (!dy && !dx) continue;
                    int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                    size_t q = (size_t)yy * w + xx;
                    if (src[q] && !labels[q]) { labels[q] = next; stack[sp++] = (int32_t)q; }
                }
        }
        if (areas && next < areas_cap) areas[next] = area;
    }
    free(stack);
    return next + 1;
}

/* ------------------------------------------------------------------------------------------- */
/* distanceTransform(src, DIST_L2, 3): two-pass 3x3 chamfer in 16.16 fixed point               */
/* (cv distanceTransform_3x3: a=0.955, b=1.3693, DIST_SHIFT=16, border = DIST_MAX)             */
/* ------------------------------------------------------------------------------------------- */
#define CVL_DIST_SHIFT 16
#define CVL_DIST_MAX (INT32_MAX >> 2)
void cvl_dist_l2_3x3(const uint8_t *src, float *dst, int h, int w)
{
    const int HV = cv_round(0.955 * (1 << CVL_DIST_SHIFT));
    const int DG = cv_round(1.3693 * (1 << CVL_DIST_SHIFT));
    const float scale = 1.0f / (1 << CVL_DIST_SHIFT);
    int tw = w + 2;
    int32_t *t = (int32_t *)malloc(sizeof(int32_t) * (size_t)(h + 2) * tw);
    for (size_t i = 0; i < (size_t)(h + 2) * tw; i++) t[i] = CVL_DIST_MAX;
    for (int y = 0; y < h; y++) {
        int32_t *row = t + (size_t)(y + 1) * tw + 1;
        const uint8_t *s = src + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            if (!s[x]) row[x] = 0;
            else {
                int t0 = row[x - tw - 1] + DG, v;
                v = row[x - tw] + HV; if (v < t0) t0 = v;
                v = row[x - tw + 1] + DG; if (v < t0) t0 = v;
                v = row[x - 1] + HV; if (v < t0) t0 = v;
                row[x] = t0;
            }
        }
    }
    for (int y = h - 1; y >= 0; y--) {
        int32_t *row = t + (size_t)(y + 1) * tw + 1;
        float *d = dst + (size_t)y * w;
        for (int x = w - 1; x >= 0; x--) {
            int t0 = row[x];
            if (t0 > HV) {
                int v;
                v = row[x + tw + 1] + DG; if (v < t0) t0 = v;
                v = row[x + tw] + HV; if (v < t0) t0 = v;
                v = row[x + tw - 1] + DG; if (v < t0) t0 = v;
                v = row[x + 1] + HV; if (v < t0) t0 = v;
                row[x] = t0;
            }
            if (t0 > CVL_DIST_MAX) t0 = CVL_DIST_MAX;
            d[x] = (float)t0 * scale;
        }
    }
    free(t);
}

/* ------------------------------------------------------------------------------------------- */
/* boxFilter(ddepth=-1, ksize=(k,k), normalize=False), BORDER_REFLECT_101                       */
/* ------------------------------------------------------------------------------------------- */
void cvl_box_sum_f32(const float *src, float *dst, int h, int w, int k)
{
    int r = k / 2;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)h * w);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float acc = 0;
            for (int j = -r; j <= r; j++) acc += src[(size_t)y * w + reflect101(x + j, w)];
            tmp[(size_t)y * w + x] = acc;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float acc = 0;
            for (int j = -r; j <= r; j++) acc += tmp[(size_t)reflect101(y + j, h) * w + x];
            dst[(size_t)y * w + x] = acc;
        }
    free(tmp);
}

/* ------------------------------------------------------------------------------------------- */
/* copyMakeBorder(BORDER_REFLECT), equal pad on all four sides                                  */
/* ------------------------------------------------------------------------------------------- */
void cvl_pad_reflect_f32(const float *src, float *dst, int h, int w, int pad)
{
    int H = h + 2 * pad, W = w + 2 * pad;
    for (int y = 0; y < H; y++) {
        int sy = reflect(y - pad, h);
        for (int x = 0; x < W; x++) dst[(size_t)y * W + x] = src[(size_t)sy * w + reflect(x - pad, w)];
    }
}

/* ------------------------------------------------------------------------------------------- */
/* cv::inpaint(src f32 1ch, mask, radius, INPAINT_TELEA)                                        */
/* Restates OpenCV photo/inpaint.cpp: FMM over a stable priority queue ordered by (T, push      */
/* sequence) [CvPriorityQueueFloat inserts after all elements with T' <= T, pops from the head],*/
/* the outside-band T field computed by a first FMM pass and negated, and Telea's weighted      */
/* first-order estimate with OpenCV's gradient quirks (x2 on central differences, km/lm index   */
/* shifts at the first/last row/column). Float variant: no +0.5 rounding bias.                  */
/* ------------------------------------------------------------------------------------------- */
#define F_KNOWN 0
#define F_BAND 1
#define F_INSIDE 2
#define F_CHANGE 3

typedef struct { float T; uint32_t seq; int i, j; } pq_elem;
typedef struct { pq_elem *e; size_t n, cap; uint32_t seq; } pq_t;

static int pq_less(const pq_elem *a, const pq_elem *b)
{ return a->T < b->T || (a->T == b->T && a->seq < b->seq); }

static void pq_push(pq_t *q, int i, int j, float T)
{
    if (q->n == q->cap) { q->cap = q->cap ? q->cap * 2 : 1024; q->e = (pq_elem *)realloc(q->e, q->cap * sizeof(pq_elem)); }
    pq_elem v = { T, q->seq++, i, j };
    size_t k = q->n++;
    while (k) {
        size_t p = (k - 1) / 2;
        if (!pq_less(&v, &q->e[p])) break;
        q->e[k] = q->e[p]; k = p;
    }
    q->e[k] = v;
}

static int pq_pop(pq_t *q, int *i, int *j)
{
    if (!q->n) return 0;
    *i = q->e[0].i; *j = q->e[0].j;
    pq_elem v = q->e[--q->n];
    size_t k = 0;
    for (;;) {
        size_t c = 2 * k + 1;
        if (c >= q->n) break;
        if (c + 1 < q->n && pq_less(&q->e[c + 1], &q->e[c])) c++;
        if (!pq_less(&q->e[c], &v)) break;
        q->e[k] = q->e[c]; k = c;
    }
    if (q->n) q->e[k] = v;
    return 1;
}

static float fmm_solve(int i1, int j1, int i2, int j2, const uint8_t *f, const float *t, int ec)
{
    double sol, a11, a22, m12;
    a11 = t[(size_t)i1 * ec + j1];
    a22 = t[(size_t)i2 * ec + j2];
    m12 = a11 < a22 ? a11 : a22;
    if (f[(size_t)i1 * ec + j1] != F_INSIDE) {
        if (f[(size_t)i2 * ec + j2] != F_INSIDE) {
            if (fabs(a11 - a22) >= 1.0) sol = 1 + m12;
            else sol = (a11 + a22 + sqrt((double)(2 - (a11 - a22) * (a11 - a22)))) * 0.5;
        } else sol = 1 + a11;
    } else if (f[(size_t)i2 * ec + j2] != F_INSIDE) sol = 1 + a22;
    else sol = 1 + m12;
    return (float)sol;
}

static float min4f(float a, float b, float c, float d)
{ a = a < b ? a : b; c = c < d ? c : d; return a < c ? a : c; }

/* optional log of the march's fill sequence (diagnostics and the test of the GPU's ordering pass): fill number per pixel, -1 = not filled */
static int g_round_u8 = 0;             /* 8-bit variant: the estimate gets + 0.5f, cvRound and saturate_cast<uchar> (values stay floats holding 0..255) */
static int32_t *g_fill_index = NULL;
static int32_t g_fill_count = 0;
/* optional log of both FMM passes' pops: rows (pass, i, j) in padded coordinates and the popped T */
static int32_t *g_pop_log = NULL;
static float *g_pop_T = NULL;
static int32_t g_pop_cap = 0, g_pop_count = 0;
static void log_pop(int pass, int i, int j, float T)
{
    if (g_pop_log && g_pop_count < g_pop_cap) {
        g_pop_log[3 * g_pop_count] = pass; g_pop_log[3 * g_pop_count + 1] = i; g_pop_log[3 * g_pop_count + 2] = j;
        g_pop_T[g_pop_count] = T;
    }
    g_pop_count++;
}

static void calc_fmm(uint8_t *f, float *t, pq_t *q, int er, int ec, int negate)
{
    int ii, jj;
    while (pq_pop(q, &ii, &jj)) {
        log_pop(0, ii, jj, t[(size_t)ii * ec + jj]);
        f[(size_t)ii * ec + jj] = (uint8_t)(negate ? F_CHANGE : F_KNOWN);
        for (int k = 0; k < 4; k++) {
            int i = ii, j = jj;
            if (k == 0) i = ii - 1; else if (k == 1) j = jj - 1; else if (k == 2) i = ii + 1; else j = jj + 1;
            if (i <= 0 || j <= 0 || i >= er - 1 || j >= ec - 1) continue;
            if (f[(size_t)i * ec + j] == F_INSIDE) {
                float dist = min4f(fmm_solve(i - 1, j, i, j - 1, f, t, ec), fmm_solve(i + 1, j, i, j - 1, f, t, ec),
                                   fmm_solve(i - 1, j, i, j + 1, f, t, ec), fmm_solve(i + 1, j, i, j + 1, f, t, ec));
                t[(size_t)i * ec + j] = dist;
                f[(size_t)i * ec + j] = F_BAND;
                pq_push(q, i, j, dist);
            }
        }
    }
    if (negate)
        for (size_t p = 0; p < (size_t)er * ec; p++)
            if (f[p] == F_CHANGE) { f[p] = F_KNOWN; t[p] = -t[p]; }
}

void cvl_inpaint_telea_f32(const float *src, const uint8_t *inpaint_mask, float *dst, int h, int w, double radius)
{
    int range = cv_round(radius);
    if (range < 1) range = 1;
    if (range > 100) range = 100;
    int er = h + 2, ec = w + 2;
    size_t en = (size_t)er * ec;
    uint8_t *mask = (uint8_t *)calloc(en, 1), *band = (uint8_t *)calloc(en, 1), *out = (uint8_t *)calloc(en, 1);
    float *t = (float *)malloc(en * sizeof(float));
    memcpy(dst, src, (size_t)h * w * sizeof(float));
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            if (inpaint_mask[(size_t)y * w + x]) mask[(size_t)(y + 1) * ec + x + 1] = F_INSIDE;
    for (size_t p = 0; p < en; p++) t[p] = 1.0e6f;
    /* band = dilate(mask, 3x3 cross) - mask, border cleared */
    for (int i = 0; i < er; i++)
        for (int j = 0; j < ec; j++) {
            int v = mask[(size_t)i * ec + j];
            if (i > 0 && mask[(size_t)(i - 1) * ec + j]) v = 1;
            if (i < er - 1 && mask[(size_t)(i + 1) * ec + j]) v = 1;
            if (j > 0 && mask[(size_t)i * ec + j - 1]) v = 1;
            if (j < ec - 1 && mask[(size_t)i * ec + j + 1]) v = 1;
            band[(size_t)i * ec + j] = (uint8_t)((v && !mask[(size_t)i * ec + j]) ? 1 : 0);
        }
    for (int i = 0; i < er; i++) { band[(size_t)i * ec] = 0; band[(size_t)i * ec + ec - 1] = 0; }
    for (int j = 0; j < ec; j++) { band[j] = 0; band[(size_t)(er - 1) * ec + j] = 0; }
    pq_t heap = { 0 }, outq = { 0 };
    for (int i = 0; i < er; i++)
        for (int j = 0; j < ec; j++)
            if (band[(size_t)i * ec + j]) { pq_push(&heap, i, j, 0.f); t[(size_t)i * ec + j] = 0.f; }
    /* outside ring: dilate(mask, rect (2r+1)^2) - mask - band, border cleared; T computed by an
     * FMM pass seeded from the band and negated */
    for (int i = 0; i < er; i++)
        for (int j = 0; j < ec; j++) {
            int v = 0;
            for (int a = -range; a <= range && !v; a++) {
                int ii = i + a; if (ii < 0 || ii >= er) continue;
                for (int b = -range; b <= range; b++) {
                    int jj = j + b; if (jj < 0 || jj >= ec) continue;
                    if (mask[(size_t)ii * ec + jj]) { v = 1; break; }
                }
            }
            out[(size_t)i * ec + j] = (uint8_t)((v && !mask[(size_t)i * ec + j] && !band[(size_t)i * ec + j]) ? F_INSIDE : 0);
        }
    for (int i = 0; i < er; i++) { out[(size_t)i * ec] = 0; out[(size_t)i * ec + ec - 1] = 0; }
    for (int j = 0; j < ec; j++) { out[j] = 0; out[(size_t)(er - 1) * ec + j] = 0; }
    for (int i = 0; i < er; i++)
        for (int j = 0; j < ec; j++)
            if (band[(size_t)i * ec + j]) pq_push(&outq, i, j, 0.f);
    calc_fmm(out, t, &outq, er, ec, 1);

    /* Telea march; flags array is `mask` (INSIDE in the hole, KNOWN elsewhere) */
    uint8_t *f = mask;
    int ii, jj;
    while (pq_pop(&heap, &ii, &jj)) {
        log_pop(1, ii, jj, t[(size_t)ii * ec + jj]);
        f[(size_t)ii * ec + jj] = F_KNOWN;
        for (int q = 0; q < 4; q++) {
            int i = ii, j = jj;
            if (q == 0) i = ii - 1; else if (q == 1) j = jj - 1; else if (q == 2) i = ii + 1; else j = jj + 1;
            if (i <= 0 || j <= 0 || i > er - 1 || j > ec - 1) continue;
            if (i >= er - 1 || j >= ec - 1) continue; /* border cells are never INSIDE */
            if (f[(size_t)i * ec + j] != F_INSIDE) continue;
            float dist = min4f(fmm_solve(i - 1, j, i, j - 1, f, t, ec), fmm_solve(i + 1, j, i, j - 1, f, t, ec),
                               fmm_solve(i - 1, j, i, j + 1, f, t, ec), fmm_solve(i + 1, j, i, j + 1, f, t, ec));
            t[(size_t)i * ec + j] = dist;
#define FF(a, b) f[(size_t)(a) * ec + (b)]
#define TT(a, b) t[(size_t)(a) * ec + (b)]
#define OO(a, b) dst[(size_t)(a) * w + (b)]
            float gtx, gty;
            if (FF(i, j + 1) != F_INSIDE) {
                if (FF(i, j - 1) != F_INSIDE) gtx = (float)((TT(i, j + 1) - TT(i, j - 1))) * 0.5f;
                else gtx = (float)((TT(i, j + 1) - TT(i, j)));
            } else {
                if (FF(i, j - 1) != F_INSIDE) gtx = (float)((TT(i, j) - TT(i, j - 1)));
                else gtx = 0;
            }
            if (FF(i + 1, j) != F_INSIDE) {
                if (FF(i - 1, j) != F_INSIDE) gty = (float)((TT(i + 1, j) - TT(i - 1, j))) * 0.5f;
                else gty = (float)((TT(i + 1, j) - TT(i, j)));
            } else {
                if (FF(i - 1, j) != F_INSIDE) gty = (float)((TT(i, j) - TT(i - 1, j)));
                else gty = 0;
            }
            float Ia = 0, Jx = 0, Jy = 0, s = 1.0e-20f;
            for (int k = i - range; k <= i + range; k++) {
                int km = k - 1 + (k == 1), kp = k - 1 - (k == er - 2);
                for (int l = j - range; l <= j + range; l++) {
                    int lm = l - 1 + (l == 1), lp = l - 1 - (l == ec - 2);
                    if (k > 0 && l > 0 && k < er - 1 && l < ec - 1) {
                        if (FF(k, l) != F_INSIDE && ((l - j) * (l - j) + (k - i) * (k - i) <= range * range)) {
                            float ry = (float)(i - k), rx = (float)(j - l);
                            float len2 = rx * rx + ry * ry;
                            float dstw = (float)(1. / (len2 * sqrtf(len2)));
                            float lev = (float)(1. / (1 + fabs(TT(k, l) - TT(i, j))));
                            float dir = rx * gtx + ry * gty;
                            if (fabs(dir) <= 0.01) dir = 0.000001f;
                            float wgt = (float)fabs(dstw * lev * dir);
                            float gix, giy;
                            if (FF(k, l + 1) != F_INSIDE) {
                                if (FF(k, l - 1) != F_INSIDE) gix = (float)((OO(km, lp + 1) - OO(km, lm - 1))) * 2.0f;
                                else gix = (float)((OO(km, lp + 1) - OO(km, lm)));
                            } else {
                                if (FF(k, l - 1) != F_INSIDE) gix = (float)((OO(km, lp) - OO(km, lm - 1)));
                                else gix = 0;
                            }
                            if (FF(k + 1, l) != F_INSIDE) {
                                if (FF(k - 1, l) != F_INSIDE) giy = (float)((OO(kp + 1, lm) - OO(km - 1, lm))) * 2.0f;
                                else giy = (float)((OO(kp + 1, lm) - OO(km, lm)));
                            } else {
                                if (FF(k - 1, l) != F_INSIDE) giy = (float)((OO(kp, lm) - OO(km - 1, lm)));
                                else giy = 0;
                            }
                            Ia += (float)wgt * (float)(OO(km, lm));
                            Jx -= (float)wgt * (float)(gix * rx);
                            Jy -= (float)wgt * (float)(giy * ry);
                            s += wgt;
                        }
                    }
                }
            }
            if (g_round_u8) {
                float sat = (float)((double)(Ia / s) + (double)(Jx + Jy) / (sqrt((double)(Jx * Jx + Jy * Jy)) + (double)1.0e-20f) + (double)0.5f);
                long isat = lrint((double)sat);
                OO(i - 1, j - 1) = (float)(isat < 0 ? 0 : isat > 255 ? 255 : isat);
            } else
            OO(i - 1, j - 1) = (float)((double)(Ia / s) + (double)(Jx + Jy) / (sqrt((double)(Jx * Jx + Jy * Jy)) + (double)1.0e-20f));
#undef FF
#undef TT
#undef OO
            f[(size_t)i * ec + j] = F_BAND;
            if (g_fill_index) g_fill_index[(size_t)(i - 1) * w + (j - 1)] = g_fill_count++;
            pq_push(&heap, i, j, dist);
        }
    }
    free(heap.e); free(outq.e); free(t); free(mask); free(band); free(out);
}

/* cv::inpaint on an 8-bit single-channel image (photo/inpaint.cpp, the uchar branch of icvTeleaInpaintFMM): src / dst hold the 0..255 values
 * as floats; every estimate is rounded as OpenCV rounds it (+ 0.5f, cvRound, saturate).  Not pinned by any file of the reference. */
void cvl_inpaint_telea_u8_f32(const float *src, const uint8_t *inpaint_mask, float *dst, int h, int w, double radius)
{
    g_round_u8 = 1;
    cvl_inpaint_telea_f32(src, inpaint_mask, dst, h, w, radius);
    g_round_u8 = 0;
}

/* the same call, also logging every pop of the outside pass (pass 0) and of the march (pass 1): returns the number of pops */
int cvl_inpaint_telea_f32_pops(const float *src, const uint8_t *inpaint_mask, float *dst, int32_t *pop_log, float *pop_T, int cap, int h, int w, double radius)
{
    g_pop_log = pop_log; g_pop_T = pop_T; g_pop_cap = cap; g_pop_count = 0;
    cvl_inpaint_telea_f32(src, inpaint_mask, dst, h, w, radius);
    g_pop_log = NULL; g_pop_T = NULL;
    return g_pop_count;
}

/* the same call, also reporting the order in which the march filled the hole pixels (0, 1, 2, ...; -1 elsewhere); returns the count */
int cvl_inpaint_telea_f32_order(const float *src, const uint8_t *inpaint_mask, float *dst, int32_t *fill_index, int h, int w, double radius)
{
    for (size_t p = 0; p < (size_t)h * w; p++) fill_index[p] = -1;
    g_fill_index = fill_index; g_fill_count = 0;
    cvl_inpaint_telea_f32(src, inpaint_mask, dst, h, w, radius);
    g_fill_index = NULL;
    return g_fill_count;
}

/* ------------------------------------------------------------------------------------------- */
/* unwrap_quality_guided  (shape_ftp.py:1043-1080)                                              */
/* Python heapq over tuples (-float(q), ny, nx, py, px): a total order, so any exact min-heap   */
/* on the full 5-tuple pops in the same sequence.  Arithmetic follows numpy >= 2 scalar         */
/* semantics (float32 throughout): dw = angle(exp(1j*(w[y,x]-w[py,px]))) in complex64.          */
/* ------------------------------------------------------------------------------------------- */
typedef struct { float nq; int32_t y, x, py, px; } uq_elem;

static int uq_less(const uq_elem *a, const uq_elem *b)
{
    if (a->nq != b->nq) return a->nq < b->nq;
    if (a->y != b->y) return a->y < b->y;
    if (a->x != b->x) return a->x < b->x;
    if (a->py != b->py) return a->py < b->py;
    return a->px < b->px;
}

typedef struct { uq_elem *e; size_t n, cap; } uq_t;

static void uq_push(uq_t *q, uq_elem v)
{
    if (q->n == q->cap) { q->cap = q->cap ? q->cap * 2 : 4096; q->e = (uq_elem *)realloc(q->e, q->cap * sizeof(uq_elem)); }
    size_t k = q->n++;
    while (k) {
        size_t p = (k - 1) / 2;
        if (!uq_less(&v, &q->e[p])) break;
        q->e[k] = q->e[p]; k = p;
    }
    q->e[k] = v;
}

static uq_elem uq_pop(uq_t *q)
{
    uq_elem top = q->e[0];
    uq_elem v = q->e[--q->n];
    size_t k = 0;
    for (;;) {
        size_t c = 2 * k + 1;
        if (c >= q->n) break;
        if (c + 1 < q->n && uq_less(&q->e[c + 1], &q->e[c])) c++;
        if (!uq_less(&q->e[c], &v)) break;
        q->e[k] = q->e[c]; k = c;
    }
    if (q->n) q->e[k] = v;
    return top;
}

/* parent_out (optional, h*w int32): linear index of the parent each pixel was unwrapped from
 * (-1 outside the flood, own index for the seed); order_out (optional): visit rank. */
void cvl_unwrap_quality_guided(const float *wrapped, const uint8_t *mask, const float *quality,
                               float *unwrapped, int32_t *parent_out, int32_t *order_out, int h, int w)
{
    size_t n = (size_t)h * w;
    for (size_t p = 0; p < n; p++) unwrapped[p] = NAN;
    if (parent_out) for (size_t p = 0; p < n; p++) parent_out[p] = -1;
    if (order_out) for (size_t p = 0; p < n; p++) order_out[p] = -1;
    /* seed = np.argmax(q) with q[~m] = -inf: first occurrence of the maximum */
    long seed = -1; float best = -INFINITY;
    int any = 0;
    for (size_t p = 0; p < n; p++) {
        if (!mask[p]) continue;
        any = 1;
        if (quality[p] > best) { best = quality[p]; seed = (long)p; }
    }
    if (!any) return;
    if (seed < 0) seed = 0; /* every masked q is -inf: argmax of an all -inf array is index 0 */
    static const int ndy[8] = { -1, 1, 0, 0, -1, -1, 1, 1 };
    static const int ndx[8] = { 0, 0, -1, 1, -1, 1, -1, 1 };
    uq_t heap = { 0 };
    int32_t rank = 0;
    unwrapped[seed] = wrapped[seed];
    if (parent_out) parent_out[seed] = (int32_t)seed;
    if (order_out) order_out[seed] = rank++;
    {
        int py = (int)(seed / w), px = (int)(seed % w);
        for (int k = 0; k < 8; k++) {
            int ny = py + ndy[k], nx = px + ndx[k];
            if (ny < 0 || ny >= h || nx < 0 || nx >= w) continue;
            size_t q = (size_t)ny * w + nx;
            if (mask[q] && !isfinite(unwrapped[q])) { uq_elem e = { -quality[q], ny, nx, py, px }; uq_push(&heap, e); }
        }
    }
    while (heap.n) {
        uq_elem e = uq_pop(&heap);
        size_t p = (size_t)e.y * w + e.x, pp = (size_t)e.py * w + e.px;
        if (!mask[p]) continue;
        if (isfinite(unwrapped[p])) continue;
        if (!isfinite(unwrapped[pp])) continue;
        float d = wrapped[p] - wrapped[pp];
        float dw = atan2f(sinf(d), cosf(d));
        unwrapped[p] = unwrapped[pp] + dw;
        if (parent_out) parent_out[p] = (int32_t)pp;
        if (order_out) order_out[p] = rank++;
        for (int k = 0; k < 8; k++) {
            int ny = e.y + ndy[k], nx = e.x + ndx[k];
            if (ny < 0 || ny >= h || nx < 0 || nx >= w) continue;
            size_t q = (size_t)ny * w + nx;
            if (mask[q] && !isfinite(unwrapped[q])) { uq_elem v = { -quality[q], ny, nx, e.y, e.x }; uq_push(&heap, v); }
        }
    }
    free(heap.e);
}
