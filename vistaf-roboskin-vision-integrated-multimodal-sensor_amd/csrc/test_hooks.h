/* Test-only entry points of libvistaf_ftp.so.  NOT part of the drop-in boundary (include/vistaf_ftp.h): nothing a caller of the
 * path needs lives here.  The parity tests use them to run the fallback / opt-in kernels of a stage against the default ones and
 * to read planes that the production path does not materialise. */
#ifndef VISTAF_TEST_HOOKS_H
#define VISTAF_TEST_HOOKS_H
#include "../../include/vistaf_ftp.h"
#ifdef __cplusplus
extern "C" {
#endif
/* name: "inpaint_tier"  2 window march + whole-frame fallback (default), 1 whole-frame kernel only, 0 cluster front end first
 *       "flood_tier"    2 batched pops (default), 1 one pop per step, 0 frontier scan; 3 (frames beyond the uint16 rank range only): the
 *                       bitmap flood hands every frame back to the generic kernel, as it does for masks larger than its bitmap
 *       "chamfer_twopass" 1 forces the one-wave two-pass chamfer transform (also for the wide frontier band of native crops, where batches
 *                      below 16 frames take the closed form)
 *       "telea_two_tier" 1 (default) 111 KB first tier of the window march + full-size retry, 0 full-size march only
 *       "unwrap_fast"  1 (default) frames whose wrapped field is verified path-independent take the parallel integration (k_unwrap_fast.hip)
 *                      instead of the priority flood, 0 always the flood (the parent plane is only produced by the flood)
 *       "big_chain"    1 (default) large frames (>= 512 x 512): selections and IRLS fits as chains of streaming kernels over the whole batch (k_big.hip),
 *                      0 one workgroup per frame (k_select / k_robust_polyfit)
 *       "telea_mw"     1 (default) 16-wave window kernel (ordering pass + dataflow fills) in front of the single-wave tiers, 0 single-wave tiers only
 *       "big_queue_lds" 1 (default) the march of a cluster no LDS window takes (k_inpaint_big.hip) keeps its queue in LDS whenever the cluster's cell
 *                      counts bound it, 0 always in the wave's slice of global memory (the path of clusters beyond that bound)
 *       "fit_capped"   0 (default) the 128-VGPR column polyfit, 1 the register-capped variant (96 VGPRs) that shares a CU with LDS-heavy one-wave kernels
 *       "keep_planes"   1 also writes the float64 demodulated field of every frame ("field" of vistaf_ftp_get_intermediate) */
int vistaf_ftp_test_set(vistaf_ftp_handle *hd, const char *name, int value);
#ifdef __cplusplus
}
#endif
#endif
