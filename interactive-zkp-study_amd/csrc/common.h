// common.h -- error plumbing shared by the HIP translation units of libzkhip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string>
#include "../../include/zkhip.h"

namespace zk {

void set_last_error(const std::string &msg);

struct HipError {
    hipError_t code;
    const char *what;
    const char *file;
    int line;
};

#define ZK_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) throw zk::HipError{_e, #expr, __FILE__, __LINE__};             \
    } while (0)

// Runs body(), translating exceptions into C-ABI error codes.
template <class Fn> int guarded(Fn body) {
    try {
        return body();
    } catch (const HipError &e) {
        char buf[512];
        snprintf(buf, sizeof(buf), "HIP error %d (%s) in %s at %s:%d", (int)e.code, hipGetErrorString(e.code), e.what, e.file, e.line);
        set_last_error(buf);
        return e.code == hipErrorNoDevice || e.code == hipErrorInvalidDevice ? ZK_ERR_NO_DEVICE
               : e.code == hipErrorOutOfMemory                               ? ZK_ERR_NOMEM
                                                                             : ZK_ERR_HIP;
    } catch (const std::bad_alloc &) {
        set_last_error("host allocation failed");
        return ZK_ERR_NOMEM;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return ZK_ERR_INVALID;
    }
}

inline int invalid(const char *msg) {
    set_last_error(msg);
    return ZK_ERR_INVALID;
}

// Fails loudly when no HIP device is usable (there is no CPU fallback).
int require_device();

// RAII device / pinned buffers.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() {}
    explicit DevBuf(size_t n) { alloc(n); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    void alloc(size_t n) {
        release();
        if (n) ZK_HIP(hipMalloc(&p, n));
        bytes = n;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    ~DevBuf() { release(); }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};
struct PinnedBuf {
    void *p = nullptr;
    PinnedBuf() {}
    explicit PinnedBuf(size_t n) { alloc(n); }
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    void alloc(size_t n) {
        release();
        if (n) ZK_HIP(hipHostMalloc(&p, n, hipHostMallocDefault));
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
    }
    ~PinnedBuf() { release(); }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

}  // namespace zk
