// Shared helpers for the gfx950 RefVSR kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/refvsr_hip.h"

typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void refvsr_set_error(const char* fmt, ...);

#define RV_CHECK(cond, ...)                        \
    do {                                           \
        if (!(cond)) {                             \
            refvsr_set_error(__VA_ARGS__);         \
            return 1;                              \
        }                                          \
    } while (0)

#define RV_HIP(call)                                                                  \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            refvsr_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

#define RV_LAUNCH_CHECK() RV_HIP(hipGetLastError())

static inline int rv_cdiv(int a, int b) { return (a + b - 1) / b; }

// ReflectionPad2d index (no edge repeat): -1 -> 1, n -> n-2.  Valid for -n < i < 2n-1.
__device__ __forceinline__ int rv_reflect(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

__device__ __forceinline__ float rv_lrelu(float v, float slope) { return v >= 0.f ? v : v * slope; }

// torch.linspace(-1, 1, n)[j] as computed by ATen's CPU kernel (symmetric two-sided evaluation).
__device__ __forceinline__ float rv_linspace_m1p1(int j, int n) {
    const float step = 2.0f / (float)(n - 1);
    return (j < n / 2) ? (-1.0f + step * (float)j) : (1.0f - step * (float)(n - 1 - j));
}
