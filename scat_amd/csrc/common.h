// Shared host/device helpers for libscat_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/scat_hip.h"

namespace scat {

// Timing-experiment code (in-kernel time stamps that overwrite outputs, ablation variants) exists only in the
// tools build (python -m scat_amd.build --diag -> tools/_bin/libscat_hip_diag.so, -DSCAT_DIAG): the shipped
// library has no switch that changes results.
#ifdef SCAT_DIAG
constexpr bool kDiag = true;
#else
constexpr bool kDiag = false;
#endif

void set_error(const char* fmt, ...);
void set_kernel_label(const char* fmt, ...);   // which engine instantiation the last call launched
void append_kernel_label(const char* suffix);
// scat_epilogue_stats_arm: the next contraction launched from this thread may write per-tile row sums into the armed
// buffer.  A launcher that supports it calls this with the rows and column groups of its grid: returns the buffer (and
// records the group count for scat_epilogue_stats_groups) when one is armed and large enough, else nullptr.
// *shift (optional out): the per-row reference the sums are taken about (scat_epilogue_stats_arm_shift), or nullptr
float* epi_stats_take(int rows, int groups, const float** shift = nullptr);
// BatchNorm-backward epilogue armed by the caller for the next qualifying accumulate launch of this host thread
// (scat_epilogue_bnb_arm): true when armed and the buffers fit (rows x groups partial pairs, a tensor of n floats)
struct EpiBnb { const float* x; const uint8_t* mask; const float* mean; float* part; int64_t cap; int64_t n; int groups; };
bool epi_bnb_take(int rows, int groups, int64_t n, EpiBnb* out);

// Every entry point returns through these: no exception crosses the C boundary.
#define SCAT_REQUIRE(cond, code, ...)          \
    do {                                       \
        if (!(cond)) {                         \
            ::scat::set_error(__VA_ARGS__);    \
            return (code);                     \
        }                                      \
    } while (0)

#define SCAT_LAUNCH_CHECK(name)                                                        \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            ::scat::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));  \
            return SCAT_E_LAUNCH;                                                      \
        }                                                                              \
    } while (0)

// Unsigned division by a runtime-constant divisor, valid for n < 2^31
// (Granlund–Montgomery add-shift form): q = (umulhi(n, mul) + n) >> shift.
struct FastDiv {
    uint32_t mul, shift, d;
    __host__ static FastDiv make(uint32_t d) {
        FastDiv f;
        f.d = d;
        uint32_t s = 0;
        while ((1ull << s) < d) ++s;
        f.shift = s;
        f.mul = (uint32_t)((((1ull << s) - d) << 32) / d + 1);
        return f;
    }
    __host__ __device__ __forceinline__ uint32_t div(uint32_t n) const {
#ifdef __HIP_DEVICE_COMPILE__
        return (__umulhi(n, mul) + n) >> shift;
#else
        return (uint32_t)(((((uint64_t)n * mul) >> 32) + n) >> shift);
#endif
    }
};

// the reference of the shifted BatchNorm sums (OutDesc::stats_shift, gemm_engine.h): a non-finite one (a diverged step's batch mean) must
// not poison every later sum — the epilogue and the finish (norm.hip bn_partials_fin_kernel) both read it through this
__device__ __forceinline__ float finite_or_zero(float c) {
    return (__float_as_uint(c) & 0x7f800000u) != 0x7f800000u ? c : 0.f;
}

// Tuning switches (tile targets, schedule variants whose losing side is a recorded negative result: DESIGN.md 1b / 1c)
// exist only in the tools build.  In the shipped library the default is a constant: its dispatch depends on the shapes of a
// call and on the five documented switches (SCAT_MATH, SCAT_PC, SCAT_WG_ROWS, SCAT_BN_LASTBLOCK, SCAT_BN_ONEPASS), which
// the parity tests use to reach both sides of a dispatch.
static inline long long diag_env_int(const char* name, long long dflt) {
#ifdef SCAT_DIAG
    const char* e = getenv(name);
    return e ? atoll(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// Cache policy of a kernel's bulk output stores, by the size of the tensor written (misc.hip): 0 = default, 2 = nt
// (non-temporal: the lines are not kept in the caches), 16 = sc1.  See DESIGN.md 1d.
int store_policy(int64_t out_bytes);

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

static inline bool fits_i32(int64_t n) { return n >= 0 && n < (1ll << 31); }

}  // namespace scat
