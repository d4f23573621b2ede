// Shared host-side helpers of libmagnet_hip.so (error text, argument checks).
#pragma once
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <hip/hip_runtime.h>

#include "../../include/magnet_hip.h"

namespace mh {

char* error_buffer();                       // thread-local, 512 bytes
int fail(int code, const char* fmt, ...);   // formats into error_buffer, returns code

inline int hip_fail(hipError_t e, const char* what) {
    return fail(MH_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

}  // namespace mh

#define MH_HIP(call)                                          \
    do {                                                      \
        hipError_t e__ = (call);                              \
        if (e__ != hipSuccess) return mh::hip_fail(e__, #call); \
    } while (0)

#define MH_LAUNCH_CHECK(name)                                  \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) return mh::hip_fail(e__, name); \
    } while (0)

#define MH_REQUIRE(cond, ...)                                 \
    do {                                                      \
        if (!(cond)) return mh::fail(MH_ERR_ARG, __VA_ARGS__); \
    } while (0)

static inline bool mh_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
