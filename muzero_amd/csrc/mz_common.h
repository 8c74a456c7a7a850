// Shared host-side helpers for the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "muzero_hip.h"

namespace mz {

// Thread-local text of the last failure, returned by mz_last_error().
char *error_buffer();

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MZ_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return MZ_OK;
}

inline int lanes_per_tree(int A) {
    int l = 2;
    while (l < A) l <<= 1;
    return l;
}

inline int check_pool(const mz_pool *p, bool need_latent) {
    if (!p) return fail(MZ_ERR_ARG, "pool is NULL");
    if (p->G <= 0 || p->A <= 0 || p->A > 64 || p->cap < 1 + p->A)
        return fail(MZ_ERR_ARG, "bad pool dims G=%d A=%d cap=%d (need G>0, 1<=A<=64, cap>=1+A)", p->G, p->A, p->cap);
    if (!p->node || !p->n_nodes) return fail(MZ_ERR_ARG, "pool has no node records or node counts");
    if (((uintptr_t)p->node & 31) != 0) return fail(MZ_ERR_ARG, "pool node records must be 32-byte aligned");
    if (p->cap > (1 << 23)) return fail(MZ_ERR_ARG, "cap %d exceeds the 2^23 nodes a parent link can name", p->cap);
    if (need_latent && (!p->latent || p->D <= 0)) return fail(MZ_ERR_ARG, "pool has no latent storage (D=%d)", p->D);
    if ((long long)p->G * p->cap > 0x7fffffffLL) return fail(MZ_ERR_ARG, "G*cap exceeds 2^31-1 nodes");
    return MZ_OK;
}

}  // namespace mz

#define MZ_TRY(expr)               \
    do {                           \
        int _rc = (expr);          \
        if (_rc != MZ_OK) return _rc; \
    } while (0)
