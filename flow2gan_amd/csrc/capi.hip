// Library-level entry points and error plumbing of libflow2gan_hip.so.
#include <stdio.h>
#include <string.h>

#include "common.h"
#include "version_gen.h"   // F2G_SRC_HASH: digest of the sources (written by the Makefile)

static char g_err[256] = "";

void f2g_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

int f2g_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    f2g_set_error(hipGetErrorString(e));
    return F2G_ELAUNCH;
  }
  return F2G_OK;
}

// "flow2gan_hip <version> src:<first 12 hex digits of the sha256 over csrc/*.hip, common.h, the header> gfx950"
extern "C" const char* f2g_version(void) { return "flow2gan_hip 0.4.0 src:" F2G_SRC_HASH " gfx950"; }
extern "C" const char* f2g_last_error(void) { return g_err; }
