// Cached environment switches of libuwu_hip.so (host only; shared by the .hip files through common.h and by dit.cpp).
#pragma once
#include <stdlib.h>

// ---- environment switches (A/B comparisons, sweeps, tests) --------------------------------------------------
// Read once per switch and cached: a launch does not call getenv().  uwu_env_refresh() (api.cpp) bumps the generation so
// that the next use re-reads -- the tests flip switches inside one process.
int uwu_env_generation();
struct UwuEnv {
  const char* name;
  int gen = -1, ival = 0;
  bool set = false;
  char c0 = 0;
  explicit UwuEnv(const char* n) : name(n) {}
  const UwuEnv& get() {
    const int g = uwu_env_generation();
    if (gen != g) {
      const char* e = getenv(name);
      set = e != nullptr;
      c0 = e ? e[0] : 0;
      ival = e ? atoi(e) : 0;
      gen = g;
    }
    return *this;
  }
  bool is(char c) const { return set && c0 == c; }
};

