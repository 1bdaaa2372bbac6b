// fa_error.h -- the error type and argument checks of libfastani_hip: plain C++, no HIP (the host-only pieces -- packer,
// FASTA reader, statistics, workspace leases -- build from this alone, which is what scripts/host_sanitize.sh compiles).
#pragma once

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/fastani_hip.h"

namespace fa {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &msg) : std::runtime_error(msg), code(c) {}
};

#define FA_REQUIRE(cond, code, msg) do { if (!(cond)) throw ::fa::Error((code), (msg)); } while (0)

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
inline uint32_t next_pow2(uint32_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace fa
