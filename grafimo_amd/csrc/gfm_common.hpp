// gfm_common.hpp -- constants, error reporting and small host helpers
// Part of libgrafimo_hip.so (one translation unit: included by grafimo_hip.hip only).
// Reference lines cited as file:line are relative to /root/reference/src/grafimo/.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "grafimo_hip.h"

namespace {


constexpr int kRange = 1000;             // utils.py:26
constexpr double kLogFactor = 1.44269504;  // utils.py:25 (truncated 1/ln2, verbatim)

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(GFM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                                 \
    } while (0)

// device allocation that frees itself (error paths of the host-side helpers)
template <typename T> struct DevBuf {
    T *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { return hipMalloc(&p, sizeof(T) * count); }
    operator T *() const { return p; }
};


}  // namespace
