// Library-level entry points and error plumbing of libflow2gan_hip.so.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

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

// ---- options -----------------------------------------------------------------------------------------------
namespace {
struct Opt { const char* name; int value; };
Opt g_opts[F2G_OPT_COUNT] = {
    {"lean", 1}, {"lean_tall", 1}, {"lean_tap", 1}, {"lean_wgrad", 1}, {"x6_tap", 1}, {"x6_wide", 1}, {"x6p", 1},
    {"w6t", 1}, {"deterministic", 0}, {"streamk", 1}, {"conv2ch_v2", 1}, {"conv32_v2", 1}, {"conv32_wgrad_v2", 1},
    {"mlp_rt", 0}, {"mlp_split", 0}, {"multi_rt384", 4}, {"multi_rt512", 3}, {"streamk_min", 4}};
std::once_flag g_opts_once;

int opt_index(const char* name, size_t len) {
  for (int i = 0; i < F2G_OPT_COUNT; ++i)
    if (strlen(g_opts[i].name) == len && strncmp(g_opts[i].name, name, len) == 0) return i;
  return -1;
}

// F2G_OPTS="x6p=2,streamk=0": names this table does not know belong to the Python side and are skipped
void opts_init() {
  const char* e = getenv("F2G_OPTS");
  while (e && *e) {
    const char* end = strchr(e, ',');
    const size_t len = end ? (size_t)(end - e) : strlen(e);
    const char* eq = (const char*)memchr(e, '=', len);
    if (eq) {
      const int i = opt_index(e, (size_t)(eq - e));
      if (i >= 0) g_opts[i].value = atoi(eq + 1);
    }
    e = end ? end + 1 : nullptr;
  }
  const char* det = getenv("F2G_DETERMINISTIC");      // (the one documented user-facing switch keeps its own name)
  if (det) g_opts[F2G_OPT_DETERMINISTIC].value = atoi(det) != 0;
}
}  // namespace

int f2g_opt(int id) {
  std::call_once(g_opts_once, opts_init);
  return g_opts[id].value;
}

extern "C" int f2g_set_option(const char* name, int32_t value) {
  std::call_once(g_opts_once, opts_init);
  const int i = name ? opt_index(name, strlen(name)) : -1;
  if (i < 0) {
    f2g_set_error("f2g_set_option: unknown option");
    return F2G_EINVAL;
  }
  g_opts[i].value = value;
  return F2G_OK;
}

extern "C" int f2g_get_option(const char* name, int32_t* value) {
  std::call_once(g_opts_once, opts_init);
  const int i = name ? opt_index(name, strlen(name)) : -1;
  if (i < 0 || !value) return F2G_EINVAL;
  *value = g_opts[i].value;
  return F2G_OK;
}
