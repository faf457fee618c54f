// mke_api.hip — error plumbing and version of libmultike_hip.so.
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "mke_common.h"

namespace mke {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return MKE_OK;
}

}  // namespace mke

namespace mke {
int g_score_splits = 0;
int g_score_half_max = -1;   // largest neg_per_pos scored two groups per wavefront; -1 = by row width (mke_score.hip), 0 = never
int g_score_o32 = 1;         // 32-bit row offsets in the training kernel when the tables allow (mke_score.hip, row_at)
int g_count_in_score = 1;    // runner: the next step's reference counting rides in the score launch (0: in the update launch)
int g_score_lane_ids = 1;    // training kernel: a group's ids and reference counts fetched once, one negative per lane (mke_score.hip)
int g_update_chunk = 0;      // rows per wavefront of the row-update kernel on large tables: 0 = by table size, 16, 64
int g_deterministic = 0;     // fixed-order gradient reduction (parity / debugging mode)
int g_sampler_fast = 1;      // mke_sampler.hip: coin block in the draw evaluation's idle lane, LDS duplicate test (0: the earlier form)
int g_attr_fused_bwd = 1;    // mke_attr_cnn.hip: dflat product inside the convolution-backward launch, dW on rider blocks (every dim <= 80)
int g_oc_em_keys64 = 0;      // mke_oc_em.hip: sort the epoch's (step, row) keys as 64-bit words even when they fit 32 bits (the path large KGs take)
int g_oc_score_quarter = -1; // mke_oc.hip: quarter-wave per positive in the owner-computes score kernel: -1 = by shape, 0 / 1
thread_local const mke_tuning* tl_tuning = nullptr;   // the tuning of the API call in progress on this thread (mke_common.h)
}

extern "C" int mke_tuning_init(mke_tuning* t) {
  if (!t) { mke::set_error("mke_tuning_init: NULL"); return MKE_E_NULL; }
  int* f = reinterpret_cast<int*>(t);
  for (size_t i = 0; i < sizeof(mke_tuning) / sizeof(int); ++i) f[i] = MKE_TUNE_DEFAULT;
  return MKE_OK;
}

extern "C" int mke_set_option(const char* name, int value, int* old_value) {
  if (!name) { mke::set_error("mke_set_option: NULL name"); return MKE_E_NULL; }
  if (!strcmp(name, "score_splits")) {
    if (old_value) *old_value = mke::g_score_splits;
    mke::g_score_splits = value < 0 ? 0 : value;
    return MKE_OK;
  }
  if (!strcmp(name, "score_half_groups")) {
    if (old_value) *old_value = mke::g_score_half_max;
    mke::g_score_half_max = value < 0 ? -1 : (value > 64 ? 64 : value);
    return MKE_OK;
  }
  if (!strcmp(name, "score_offsets32")) {
    if (old_value) *old_value = mke::g_score_o32;
    mke::g_score_o32 = value != 0;
    return MKE_OK;
  }
  if (!strcmp(name, "count_in_score")) {
    if (old_value) *old_value = mke::g_count_in_score;
    mke::g_count_in_score = value != 0;
    return MKE_OK;
  }
  if (!strcmp(name, "score_lane_ids")) {
    if (old_value) *old_value = mke::g_score_lane_ids;
    mke::g_score_lane_ids = value != 0;
    return MKE_OK;
  }
  if (!strcmp(name, "update_chunk")) {
    if (old_value) *old_value = mke::g_update_chunk;
    mke::g_update_chunk = value == 16 ? 16 : (value == 64 ? 64 : 0);
    return MKE_OK;
  }
  if (!strcmp(name, "oc_score_quarter")) {
    if (old_value) *old_value = mke::g_oc_score_quarter;
    mke::g_oc_score_quarter = value < 0 ? -1 : (value != 0);
    return MKE_OK;
  }
  if (!strcmp(name, "attr_fused_bwd")) {
    if (old_value) *old_value = mke::g_attr_fused_bwd;
    mke::g_attr_fused_bwd = value != 0;
    return MKE_OK;
  }
  if (!strcmp(name, "sampler_fast")) {
    if (old_value) *old_value = mke::g_sampler_fast;
    mke::g_sampler_fast = value != 0;
    return MKE_OK;
  }
  if (!strcmp(name, "oc_em_keys64")) {
    if (old_value) *old_value = mke::g_oc_em_keys64;
    mke::g_oc_em_keys64 = value != 0;
    return MKE_OK;
  }
  if (!strcmp(name, "deterministic")) {
    if (old_value) *old_value = mke::g_deterministic;
    mke::g_deterministic = value != 0;
    return MKE_OK;
  }
  mke::set_error("mke_set_option: unknown option '%s'", name);
  return MKE_E_UNSUPPORTED;
}

extern "C" int mke_version(void) { return MKE_VERSION; }
extern "C" const char* mke_last_error(void) { return mke::g_err; }

