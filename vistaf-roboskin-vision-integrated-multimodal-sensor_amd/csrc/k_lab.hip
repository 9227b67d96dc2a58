// Feature planes of the temperature modality (Code/temperature_sensor.py:278-293) and the colour-support test of its main() (:790-799).
//
//   compute_feature_planes(image_bgr, blur_ksize = 5):
//       img = cv2.GaussianBlur(image_bgr, (5, 5), 0)         8-bit path: tabulated kernel [1 4 6 4 1] / 16, both passes exact in fixed point,
//                                                            ONE rounding (sum + 128) >> 8, BORDER_REFLECT_101
//       L, a, b = cv2.cvtColor(img, COLOR_BGR2LAB)           8-bit path: sRGB gamma table (x 255 * 8), matrix in 2^12 fixed point, cube-root
//                                                            table (x 2^15), L * 255 / 100, a + 128, b + 128, every step integer
//       gray = cv2.cvtColor(img, COLOR_BGR2GRAY)             the same fixed-point form as k_to_gray
//   all four returned as float32 planes.  One kernel: a 64 x 16 output tile per block, the 68 x 20 x 3 byte halo tile and both tables staged
//   in LDS (18 KB), row pass into 16-bit LDS, column pass + colour arithmetic in registers, four coalesced float stores.
//   Bandwidth-bound: 3 B read + 16 B written per pixel.
#include "kernels.hpp"

namespace vf {

namespace {
constexpr int FP_TX = 64, FP_TY = 16, FP_HALO = 2;
constexpr int FP_LW = FP_TX + 2 * FP_HALO, FP_LH = FP_TY + 2 * FP_HALO;

__device__ inline int fp_descale(int v, int s) { return (v + (1 << (s - 1))) >> s; }
__device__ inline int fp_sat8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
}  // namespace

__global__ __launch_bounds__(256) void k_feature_planes(const uint8_t *__restrict__ bgr, const uint16_t *__restrict__ gamma_tab,
                                                        const uint16_t *__restrict__ cbrt_tab, LabCoef cf, int blur, float *__restrict__ oL,
                                                        float *__restrict__ oa, float *__restrict__ ob, float *__restrict__ ogray, int H, int W)
{
    __shared__ uint8_t tile[FP_LH][FP_LW * 3];
    __shared__ uint16_t hz[FP_LH][FP_TX * 3];
    __shared__ uint16_t sg[256];
    __shared__ uint16_t sc[LAB_CBRT_N];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * FP_TX, y0 = blockIdx.y * FP_TY;
    for (int i = tid; i < 256; i += 256) sg[i] = gamma_tab[i];
    for (int i = tid; i < LAB_CBRT_N; i += 256) sc[i] = cbrt_tab[i];
    for (int i = tid; i < FP_LH * FP_LW; i += 256) {
        const int ty = i / FP_LW, tx = i - ty * FP_LW;
        // columns / rows of a partial tile that lie beyond the reflected border are never used by a stored pixel: clamp them into the frame
        const int gy = reflect101(min(y0 + ty - FP_HALO, 2 * (H - 1)), H), gx = reflect101(min(x0 + tx - FP_HALO, 2 * (W - 1)), W);
        const uint8_t *s = bgr + ((size_t)gy * W + gx) * 3;
        tile[ty][3 * tx] = s[0]; tile[ty][3 * tx + 1] = s[1]; tile[ty][3 * tx + 2] = s[2];
    }
    __syncthreads();
    if (blur) {
        for (int i = tid; i < FP_LH * FP_TX * 3; i += 256) {
            const int ty = i / (FP_TX * 3), j = i - ty * (FP_TX * 3);          // j = 3 * column + channel
            const uint8_t *t = &tile[ty][j + 3 * FP_HALO];
            hz[ty][j] = (uint16_t)((int)t[-6] + (int)t[6] + 4 * ((int)t[-3] + (int)t[3]) + 6 * (int)t[0]);
        }
        __syncthreads();
    }
    const int tx = tid & 63, x = x0 + tx;
    if (x >= W) return;
#pragma unroll
    for (int k = 0; k < FP_TY / 4; k++) {
        const int ty = (tid >> 6) + 4 * k, y = y0 + ty;
        if (y >= H) break;
        int c[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            if (blur) {
                const int j = 3 * tx + ch;
                const int acc = (int)hz[ty][j] + (int)hz[ty + 4][j] + 4 * ((int)hz[ty + 1][j] + (int)hz[ty + 3][j]) + 6 * (int)hz[ty + 2][j];
                c[ch] = (acc + 128) >> 8;
            } else c[ch] = tile[ty + FP_HALO][3 * (tx + FP_HALO) + ch];
        }
        const int B = sg[c[0]], G = sg[c[1]], R = sg[c[2]];
        const int fx = sc[fp_descale(R * cf.c[0] + G * cf.c[1] + B * cf.c[2], LAB_SHIFT)];
        const int fy = sc[fp_descale(R * cf.c[3] + G * cf.c[4] + B * cf.c[5], LAB_SHIFT)];
        const int fz = sc[fp_descale(R * cf.c[6] + G * cf.c[7] + B * cf.c[8], LAB_SHIFT)];
        const int l = fp_sat8(fp_descale(cf.lscale * fy + cf.lshift, LAB_SHIFT2));
        const int a = fp_sat8(fp_descale(500 * (fx - fy) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2));
        const int b = fp_sat8(fp_descale(200 * (fy - fz) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2));
        const int gr = (c[0] * 3735 + c[1] * 19235 + c[2] * 9798 + (1 << 14)) >> 15;
        const size_t o = (size_t)y * W + x;
        if (oL) oL[o] = (float)l;
        if (oa) oa[o] = (float)a;
        if (ob) ob[o] = (float)b;
        if (ogray) ogray[o] = (float)gr;
    }
}

void launch_feature_planes(const uint8_t *bgr, const uint16_t *gamma_tab, const uint16_t *cbrt_tab, const LabCoef &cf, bool blur, float *L, float *a,
                           float *b, float *gray, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL(k_feature_planes, dim3((W + FP_TX - 1) / FP_TX, (H + FP_TY - 1) / FP_TY), dim3(256), 0, st, bgr, gamma_tab, cbrt_tab, cf,
                       blur ? 1 : 0, L, a, b, gray, H, W);
}

// chroma = float32 sqrt((a - 128)^2 + (b - 128)^2) (:793-795); color_support = light_d & roi_eff & ~sat & (chroma >= COLOR_CHROMA_MIN) (:799)
__global__ void k_color_support(const float *__restrict__ a, const float *__restrict__ b, const uint8_t *__restrict__ light_d,
                                const uint8_t *__restrict__ roi_eff, const uint8_t *__restrict__ sat, float chroma_min, float *__restrict__ chroma,
                                uint8_t *__restrict__ support, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float da = __fsub_rn(a[i], 128.0f), db = __fsub_rn(b[i], 128.0f);
    const float ch = sqrtf(__fadd_rn(__fmul_rn(da, da), __fmul_rn(db, db)));
    if (chroma) chroma[i] = ch;
    if (support) support[i] = (uint8_t)(light_d[i] && roi_eff[i] && !sat[i] && ch >= chroma_min);
}

void launch_color_support(const float *a, const float *b, const uint8_t *light_d, const uint8_t *roi_eff, const uint8_t *sat, float chroma_min,
                          float *chroma, uint8_t *support, size_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_color_support, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, light_d, roi_eff, sat, chroma_min, chroma, support, n);
}

}  // namespace vf
