// Shared helpers for the gfx950 C-ABI library (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/cmbpo_hip.h"

void cmbpo_set_error(const char *fmt, ...);

#define CMBPO_HIP_CHECK(expr)                                                  \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) {                                                    \
      cmbpo_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                      __FILE__, __LINE__);                                     \
      return CMBPO_EHIP;                                                       \
    }                                                                          \
  } while (0)

#define CMBPO_REQUIRE(cond, ...)                                               \
  do {                                                                         \
    if (!(cond)) {                                                             \
      cmbpo_set_error(__VA_ARGS__);                                            \
      return CMBPO_EINVAL;                                                     \
    }                                                                          \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int cmbpo_ceil_div(int a, int b) { return (a + b - 1) / b; }
