// Shared declarations of the orbx library's translation units.
#ifndef ORBX_INTERNAL_H
#define ORBX_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/orbx.h"

#define ORBX_MAX_LEVELS 16

// Thread-local error text behind orbx_last_error().
void orbx_set_error(const char* fmt, ...);

#define ORBX_HIP_CHECK(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            orbx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                           __LINE__);                                                         \
            return ORBX_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

// Candidate / selected keypoint packing: x[31:20] y[19:8] response[7:0].
#define ORBX_PACK(x, y, r) (((uint32_t)(x) << 20) | ((uint32_t)(y) << 8) | (uint32_t)(r))
#define ORBX_PX(p) ((int)((p) >> 20))
#define ORBX_PY(p) ((int)(((p) >> 8) & 0xFFF))
#define ORBX_PR(p) ((int)((p)&0xFF))

#endif
