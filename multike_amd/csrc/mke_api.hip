// mke_api.hip — error plumbing and version of libmultike_hip.so.
#include <cstdarg>
#include <cstdio>

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

extern "C" int mke_version(void) { return MKE_VERSION; }
extern "C" const char* mke_last_error(void) { return mke::g_err; }
