// launch_util.h -- host-side launch helpers shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <stdint.h>

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to a function ON A DEVICE.  A process-wide "done" flag applies it to the first
// device only, and a process that later launches on a second GPU with more than 64 KB of dynamic LDS fails there.  One bit per device.
struct hsLdsAttrOnce {
    std::atomic<uint64_t> done{0};
    void set(const void *fn, int bytes) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const uint64_t bit = 1ull << (dev & 63);
        if (!(done.load(std::memory_order_relaxed) & bit)) {
            (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            done.fetch_or(bit, std::memory_order_relaxed);
        }
    }
};
