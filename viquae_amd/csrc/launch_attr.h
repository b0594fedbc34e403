// Shared by the .hip translation units of libmeerqat_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>

namespace mq_detail {

constexpr int LDS_PER_CU = 160 * 1024;  // gfx950

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device): raise it the first time a
// kernel is launched on a device instead of on every call.  `done` = one bit per device ordinal.
inline hipError_t ensure_dynamic_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    if (bytes >= LDS_PER_CU) {  // "as much as the kernel may ever ask for": what its static __shared__ objects leave
        hipFuncAttributes attr;
        e = hipFuncGetAttributes(&attr, fn);
        if (e != hipSuccess) return e;
        bytes = LDS_PER_CU - (int)attr.sharedSizeBytes;
    }
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

}  // namespace mq_detail

// CHECK = the translation unit's "return MQ_EHIP on failure" macro; the kernel expression may hold commas.
#define MQ_DYNAMIC_LDS_WITH(CHECK, bytes, ...)                                                         \
    do {                                                                                               \
        static std::atomic<unsigned long long> _mq_lds_done{0};                                        \
        CHECK(mq_detail::ensure_dynamic_lds((const void*)(__VA_ARGS__), (int)(bytes), _mq_lds_done));  \
    } while (0)
