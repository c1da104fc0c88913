// Diagnostic device allocator for PyTorch (torch.cuda.memory.CUDAPluggableAllocator): every allocation gets a region of its
// own (hipMalloc, a multiple of 2 MiB) and is placed so that it ENDS at the end of the region (16-byte aligned start), with
// the next 2 MiB of address space allocated and released again right after, so that -- as far as the driver's address
// assignment allows -- nothing is mapped behind it.  A kernel that reads or writes past the end of ANY tensor then faults
// at once instead of quietly touching a neighbour in the caching allocator's pool (where such a read only faults when the
// tensor happens to sit at the end of a mapped segment: the intermittent "Memory access fault ... on address 0x...600000").
// Test infrastructure only (tests/test_gpu_guard.py); built on the GPU box by the test that uses it.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>

namespace {
constexpr size_t kPage = 2u << 20;
std::mutex g_mu;
std::map<void *, void *> g_base;  // user pointer -> region base
}  // namespace

extern "C" void *guard_malloc(ssize_t size, int device, hipStream_t stream)
{
    (void)stream;
    if (size <= 0) size = 16;
    int cur = 0;
    hipGetDevice(&cur);
    if (cur != device) hipSetDevice(device);
    const size_t want = ((size_t)size + 15) & ~(size_t)15;
    const size_t region = (want + kPage - 1) / kPage * kPage;
    void *base = nullptr, *hole = nullptr;
    if (hipMalloc(&base, region) != hipSuccess) return nullptr;
    if (hipMalloc(&hole, kPage) == hipSuccess) hipFree(hole);  // usually the next addresses: leaves them unmapped
    void *user = static_cast<char *>(base) + (region - want);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_base[user] = base;
    }
    if (cur != device) hipSetDevice(cur);
    return user;
}

extern "C" void guard_free(void *ptr, ssize_t size, int device, hipStream_t stream)
{
    (void)size;
    (void)device;
    (void)stream;
    void *base = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_base.find(ptr);
        if (it == g_base.end()) return;
        base = it->second;
        g_base.erase(it);
    }
    hipFree(base);  // synchronises the device: nothing in flight still uses the region
}
