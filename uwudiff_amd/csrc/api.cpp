// Error plumbing + version for libuwu_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/uwu_hip.h"

static thread_local char g_err[512] = "";

void uwu_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* uwu_last_error(void) { return g_err; }
extern "C" int uwu_version(void) { return 1; }

// generation counter of the cached environment switches (common.h UwuEnv)
static int g_env_gen = 0;
int uwu_env_generation() { return g_env_gen; }
extern "C" int uwu_env_refresh(void) { return ++g_env_gen; }
