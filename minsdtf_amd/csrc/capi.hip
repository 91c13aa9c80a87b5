// Library-level entry points of the C ABI (include/minsdtf_hip.h).
#include <stdarg.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void msd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int msd_conv_gemm_init();
int msd_attention_init();

extern "C" int msd_abi_version(void) { return MSD_ABI_VERSION; }
extern "C" const char* msd_last_error(void) { return g_err; }
extern "C" int msd_init(void) {
    int rc = msd_conv_gemm_init();
    if (rc) return rc;
    return msd_attention_init();
}

void msd_set_gn_impl(int v);
void msd_set_conv_dense(int v);
void msd_set_gn_rows_q(int v);
void msd_set_gn_wide(int v);
void msd_set_gn_cluster(int v);
void msd_set_gn_rows(int v);
void msd_set_gn_xmap(int v);
void msd_set_gn_poll_limit(int v);
void msd_set_attn_qf(int v);
void msd_set_attn_qf4_min(int v);
void msd_set_xattn_nw(int v);
int msd_set_xattn160_mode(int v);
void msd_set_attn_form(int v);
void msd_set_attn_d160_pipe(int v);
/* Tuning / A-B switches (not needed for normal use). Known keys: "gn_impl" (1 = single-launch per-group
 * GroupNorm where the group slab fits in registers [default], 0 = always stats/finalize/apply). */
extern "C" int msd_set_option(const char* key, int value) {
    if (key && strcmp(key, "gn_rows_q") == 0) {   // row-major GroupNorm: log2 of the workgroups sharing a part by channels (0 .. 2), -1 = by launch size [default]; placement only
        if (value < -1 || value > 2) MSD_FAIL(MSD_E_ARG, "set_option: gn_rows_q %d (-1 .. 2)", value);
        msd_set_gn_rows_q(value);
        return MSD_OK;
    }
    if (key && strcmp(key, "conv_dense") == 0) {   // 1 = DENSE loader for 1x1 / Dense layers [default], 0 = general loader
        msd_set_conv_dense(value);
        return MSD_OK;
    }
    if (key && strcmp(key, "attn_qf") == 0) {     // 0 = automatic [default], 1 / 2 = 64 / 128 queries per workgroup
        if (value < 0 || value > 4 || value == 3) MSD_FAIL(MSD_E_ARG, "set_option: attn_qf takes 0, 1, 2 or 4");
        msd_set_attn_qf(value);
        return MSD_OK;
    }
    if (key && strcmp(key, "attn_qf4_min") == 0) {   // d = 40 software-pipelined attention: 256-query workgroups from this many 128-query workgroups on (scheduling only: same bits)
        if (value < 128) MSD_FAIL(MSD_E_ARG, "set_option: attn_qf4_min takes a workgroup count >= 128");
        msd_set_attn_qf4_min(value);
        return MSD_OK;
    }
    if (key && strcmp(key, "attn_form") == 0) {   // d = 40 / 80: 2 = 32x32x16 MFMAs, software-pipelined on long key walks [default], 1 = 32x32x16 plain, 0 = 16x16x32
        if (value < 0 || value > 2) MSD_FAIL(MSD_E_ARG, "set_option: attn_form takes 0, 1 or 2");
        msd_set_attn_form(value);
        return MSD_OK;
    }
    if (key && strcmp(key, "attn_d160_pipe") == 0) {   // d = 160, 64-query workgroups: 1 = K/V register prefetch [default], 0 = serial staging (same bits)
        msd_set_attn_d160_pipe(value ? 1 : 0);
        return MSD_OK;
    }
    if (key && strcmp(key, "xattn160_mode") == 0) {   // experiment modes of xattn_q160_kernel (wrong results by design): `make stamps` build only
        if (!msd_set_xattn160_mode(value)) MSD_FAIL(MSD_E_ARG, "set_option: xattn160_mode %d exists in the instrumented build (make stamps) only", value);
        return MSD_OK;
    }
    if (key && strcmp(key, "xattn_nw") == 0) {    // 0 = automatic [default], 4 / 8 = 64 / 128 queries per fused cross-attention workgroup
        if (value != 0 && value != 4 && value != 8) MSD_FAIL(MSD_E_ARG, "set_option: xattn_nw takes 0, 4 or 8");   // grid and kernel must agree on the tile
        msd_set_xattn_nw(value);
        return MSD_OK;
    }
    if (key && strcmp(key, "gn_wide") == 0) {   // 1 = 1024-thread GroupNorm workgroups for mid-sized tensors [default]
        msd_set_gn_wide(value);
        return MSD_OK;
    }
    if (key && strcmp(key, "gn_xmap") == 0) { msd_set_gn_xmap(value); return MSD_OK; }   // A/B runs: 0 = the cluster GroupNorm's parts of a group on consecutive workgroup ids
    if (key && strcmp(key, "gn_rows") == 0) {   // row-major cluster GroupNorm for samples of at least this many pixels [default 9216; 4096 pays at batch >= 2 per GPU]; 0 = never
        if (value < 0) MSD_FAIL(MSD_E_ARG, "set_option: gn_rows takes a pixel count >= 0, got %d", value);
        msd_set_gn_rows(value);
        return MSD_OK;
    }
    if (key && strcmp(key, "gn_cluster") == 0) {   // pixels per part of the cluster GroupNorm (P = pixels / this, a power of two <= 8) [default 256]; 0 = never
        if (value != 0 && (value < 64 || value > (1 << 20))) MSD_FAIL(MSD_E_ARG, "set_option: gn_cluster takes 0 or 64 .. 2^20 pixels per part");
        msd_set_gn_cluster(value);
        return MSD_OK;
    }
    if (key && strcmp(key, "gn_poll_limit") == 0) {   // bound of the cluster GroupNorm's exchange poll [default 2^18]; tests of the give-up path shorten it
        if (value < 1) MSD_FAIL(MSD_E_ARG, "set_option: gn_poll_limit takes a positive count");
        msd_set_gn_poll_limit(value);
        return MSD_OK;
    }
    if (key && strcmp(key, "gn_impl") == 0) {
        msd_set_gn_impl(value);
        return MSD_OK;
    }
    MSD_FAIL(MSD_E_ARG, "set_option: unknown key");
}
